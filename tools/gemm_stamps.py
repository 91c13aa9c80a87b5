"""In-kernel timeline of conv_gemm_dma_kernel on one layer shape (needs `make -C minsdtf_amd/csrc stamps`).

    python tools/gemm_stamps.py --m 512 --n 1280 --k 1280 --tile 64x64 [--stages 0] [--splitk 1] [--ksize 1]

Thread 0 of every workgroup stamps the 100 MHz wall clock at: 0 entry, 1 ring primed (S-1 tiles issued),
2 first tile landed (first barrier passed), 5 half of the K loop, 3 K loop done, 4 stores retired.  Printed:
percentiles over workgroups of each interval, the spread of entry times (launch ramp) and the span from the
first entry to the last exit (= the kernel's duration)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="512x1280x1280:64x64:0:1,8192x320x320:128x128:0:1",
                    help="comma list of MxNxK:tile:stages:splitk (M = 2*h*h; K = channels, or 9*channels with --ks 3)")
    ap.add_argument("--ks", type=int, default=1, help="1 = dense / 1x1, 3 = 3x3 conv (tile 1128 / 1256 / 2128 x N = halo kernel)")
    ap.add_argument("--dense", type=int, default=1, help="0 = general loader for 1x1 layers (A/B)")
    ap.add_argument("--hot", action="store_true", help="one weight buffer (L2 / Infinity-Cache resident) instead of > 256 MiB of rotating copies")
    args = ap.parse_args()
    from minsdtf_amd import _lib

    _lib.LIB_PATH = os.path.join(ROOT, "tools", "_build", "libminsdtf_hip_stamps.so")
    from minsdtf_amd import ops

    lib = _lib.load()
    lib.msd_init()
    lib.msd_set_option(b"conv_dense", args.dense)
    for fn in (lib.msd_debug_stamps, lib.msd_debug_stamps_halo, lib.msd_debug_stamps_wreg, lib.msd_debug_stamps_big, lib.msd_debug_phase_stamps_big):
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int]
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream()
    for spec in args.shapes.split(","):
        mnk, tile, stg, sk = spec.split(":")
        M, N, K = (int(v) for v in mnk.split("x"))
        tm, tn = (int(v) for v in tile.split("x"))
        stg, sk = int(stg), int(sk)
        h = int(round((M // 2) ** 0.5))
        assert 2 * h * h == M
        cin = K // (args.ks * args.ks)
        x = torch.randn(M, cin, device=dev).to(torch.bfloat16)
        ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(1 if args.hot else max(2, (300 << 20) // (N * K * 2)))][:24]
        if args.hot:
            ws = ws * 6
        bias = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        wsf = torch.empty(max(1, sk * M * N), device=dev, dtype=torch.float32)
        wreg = 4000 <= tm < 5000   # wreg form: tile 4000 + rows, fragment-major weights
        big = tm >= 5000           # big form (conv_big.hip): tile 5000 + rows
        if wreg:
            from minsdtf_amd import packing

            ws = [packing.fragment_major(wi) for wi in ws]
        calls = [ops.conv_gemm(a0=x, w=wi, out=out, batch=2, h_in=h, w_in=h, c0=cin, N=N, ksize=args.ks, bias=bias, residual=res,
                               workspace=wsf, workspace_floats=wsf.numel(), splitk=sk, tile_m=tm, tile_n=tn, stages=stg, w_layout=2 if wreg else 0)
                 for wi in ws]
        for c in calls:          # the last call's stamps survive; weights rotate so they come from HBM
            c(st.cuda_stream)
        torch.cuda.synchronize()
        halo = 1000 <= tm < 3000
        bm = (tm % 1000) if tm >= 1000 else tm
        nwg = ((M + bm - 1) // bm) * ((N + tn - 1) // tn) * sk
        buf = np.zeros(16 * 8192, np.uint64)
        rc = (lib.msd_debug_stamps_halo if halo else lib.msd_debug_stamps_wreg if wreg else lib.msd_debug_stamps_big if big else lib.msd_debug_stamps)(buf.ctypes.data, buf.size)
        assert rc == 0, rc
        t = buf.reshape(8192, 16)[:min(nwg, 8192)].astype(np.int64)
        t0 = t[:, 0].min()
        us = lambda a: a / 100.0  # 100 MHz ticks -> us

        def pct(a):
            return " ".join(f"{us(np.percentile(a, q)):6.2f}" for q in (0, 25, 50, 75, 100))

        nk = K // 64 // sk
        if sk > 1:
            print("   (split-K: the stamps cover the slab kernel only, not splitk_finalize)")
        print(f"--- M={M} N={N} K={K} tile {tm}x{tn} stages {stg} splitk {sk}: {nwg} workgroups, {nk} K tiles each")
        print(f"   percentiles over workgroups (us)        min    p25    p50    p75    max")
        print(f"   entry time after first entry        {pct(t[:, 0] - t0)}")
        print(f"   prologue (entry -> ring primed)      {pct(t[:, 1] - t[:, 0])}")
        print(f"   first tile (primed -> landed)        {pct(t[:, 2] - t[:, 1])}")
        print(f"   K loop first half                    {pct(t[:, 5] - t[:, 2])}")
        print(f"   K loop second half                   {pct(t[:, 3] - t[:, 5])}")
        print(f"   epilogue: loop done -> loads back    {pct(t[:, 6] - t[:, 3])}")
        print(f"   epilogue: arithmetic + store issue   {pct(t[:, 7] - t[:, 6])}")
        print(f"   epilogue: stores retired             {pct(t[:, 4] - t[:, 7])}")
        if halo and t[:, 15].min() > 0:
            print(f"   one K step: vmcnt wait              {pct(t[:, 11] - t[:, 10])}")
            print(f"   one K step: barrier                 {pct(t[:, 12] - t[:, 11])}")
            print(f"   one K step: ds_read + DMA issue     {pct(t[:, 13] - t[:, 12])}")
            print(f"   one K step: fragments landed        {pct(t[:, 14] - t[:, 13])}")
            print(f"   one K step: MFMAs issued + retired  {pct(t[:, 15] - t[:, 14])}")
        if big:   # shader-clock sums per phase of waves 0 / 4 (the first wave of either half-workgroup)
            pb = np.zeros(1024 * 2 * 8, np.uint64)
            assert lib.msd_debug_phase_stamps_big(pb.ctypes.data, pb.size) == 0
            pr = pb.reshape(1024, 2, 8)[:min(nwg, 1024)].astype(np.float64)
            names = ("fragment reads issued", "DMAs issued", "counted wait", "barrier 1", "MFMAs issued", "barrier 2")
            for hh in (0, 1):
                nph = pr[:, hh, 6]
                ok = nph > 0
                tot = 0.0
                line = []
                for i in range(6):
                    v = (pr[ok, hh, i] / nph[ok]).mean()
                    tot += v
                    line.append(f"{names[i]} {v:6.0f}")
                print(f"   half {hh}: cycles per phase: " + " | ".join(line) + f" | sum {tot:6.0f}  ({nph[ok].mean():.0f} phases)")
        print(f"   workgroup lifetime                   {pct(t[:, 4] - t[:, 0])}")
        print(f"   exit time after first entry          {pct(t[:, 4] - t0)}")
        print(f"   kernel span {us(t[:, 4].max() - t0):.2f} us; per K tile in the loop {us(np.median(t[:, 3] - t[:, 2])) / max(1, nk - 1):.3f} us",
              flush=True)
        # the same launch back to back on the stream (dependent by stream order, weights rotating): wall time per launch minus
        # the span above = what a launch boundary + dispatch ramp costs around this kernel
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record(st)
        for _ in range(reps):
            for c in calls:
                c(st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        per = e0.elapsed_time(e1) * 1e3 / (reps * len(calls))
        g = torch.cuda.CUDAGraph()
        s2 = torch.cuda.Stream()
        s2.wait_stream(st)
        with torch.cuda.stream(s2):
            with torch.cuda.graph(g, stream=s2):
                for _ in range(4):
                    for c in calls:
                        c(torch.cuda.current_stream().cuda_stream)
        st.wait_stream(s2)
        g.replay()
        torch.cuda.synchronize()
        e0.record(st)
        for _ in range(5):
            g.replay()
        e1.record(st)
        torch.cuda.synchronize()
        perg = e0.elapsed_time(e1) * 1e3 / (5 * 4 * len(calls))
        print(f"   back to back: {per:.2f} us per launch eager, {perg:.2f} us per launch in a replayed graph", flush=True)


if __name__ == "__main__":
    main()

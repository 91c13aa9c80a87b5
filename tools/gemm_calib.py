"""Calibration of msd_conv_gemm on the UNet's dense / conv-as-GEMM shapes against the vendor GEMM
(torch.nn.functional.linear -> hipBLASLt), same operands, same layout (W[N][K]), HIP-event timing,
weights rotated through > 256 MiB of copies so that they stream from HBM as in the real pipeline.

    python tools/gemm_calib.py [--iters 20]

The vendor GEMM is a yardstick for the tuner only; nothing in the product path calls it."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # (batch, h, w, cin, N): 1x1 / dense shapes of one fused cond+uncond UNet step
    (2, 64, 64, 320, 320), (2, 64, 64, 320, 960), (2, 64, 64, 320, 2560), (2, 64, 64, 1280, 320), (2, 64, 64, 960, 320),
    (2, 32, 32, 640, 640), (2, 32, 32, 640, 1920), (2, 32, 32, 640, 5120), (2, 32, 32, 2560, 640), (2, 32, 32, 1920, 640),
    (2, 16, 16, 1280, 1280), (2, 16, 16, 1280, 3840), (2, 16, 16, 1280, 10240), (2, 16, 16, 5120, 1280), (2, 16, 16, 2560, 1280),
    (2, 8, 8, 1280, 1280), (2, 8, 8, 1280, 3840), (2, 8, 8, 1280, 10240), (2, 8, 8, 5120, 1280), (2, 8, 8, 2560, 1280),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--sweep", action="store_true", help="also time every tile / ring depth / split-K of msd_conv_gemm")
    args = ap.parse_args()
    from minsdtf_amd import _lib, ops, tuning

    lib = _lib.load()
    lib.msd_init()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream()

    def timed(fns):
        for f in fns[:2]:
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(args.iters):
            fns[i % len(fns)]()
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.iters

    # launch floor: a kernel that does nothing useful, back to back on the stream
    tiny = torch.zeros(64, device=dev)
    print(f"launch floor (torch add_ on 64 floats, back to back): {timed([lambda: tiny.add_(1.0)]):.2f} us")
    for (b, h, w, cin, N) in SHAPES:
        M, K = b * h * w, cin
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        ncopy = max(1, min(16, (300 << 20) // (N * K * 2)))
        ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(ncopy)]
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_vendor = timed([lambda wi=wi: torch.nn.functional.linear(x, wi, out=None) for wi in ws])
        bm, bn, sk, stg = tuning.lookup(b, h, w, cin, N, 1, 1, False, M, K // 64, True)
        wsf = torch.empty(max(1, sk * M * N), device=dev, dtype=torch.float32)
        calls = [ops.conv_gemm(a0=x, w=wi, out=out, batch=b, h_in=h, w_in=w, c0=cin, N=N, ksize=1, bias=bias, workspace=wsf,
                               workspace_floats=wsf.numel(), splitk=sk, tile_m=bm, tile_n=bn, stages=stg) for wi in ws]
        t_ours = timed([lambda c=c: c(st.cuda_stream) for c in calls])
        fl = 2.0 * M * N * K
        sweep = ""
        if args.sweep:
            res = []
            for (tm, tn, stg2) in tuning.TILES:
                if tn == 80 and N % 80:
                    continue
                if tm == 256 and M < 1024:
                    continue
                for sk2 in (1, 2, 4):
                    if sk2 > 1 and (K // 64) // sk2 < 2:
                        continue
                    wsf2 = torch.empty(max(1, sk2 * M * N), device=dev, dtype=torch.float32)
                    cs = [ops.conv_gemm(a0=x, w=wi, out=out, batch=b, h_in=h, w_in=w, c0=cin, N=N, ksize=1, bias=bias, workspace=wsf2,
                                        workspace_floats=wsf2.numel(), splitk=sk2, tile_m=tm, tile_n=tn, stages=stg2) for wi in ws]
                    res.append((timed([lambda c=c: c(st.cuda_stream) for c in cs]), f"{tm}x{tn}s{stg2}/k{sk2}"))
            sweep = "\n      " + " ".join(f"{n}:{t:.1f}" for t, n in sorted(res))
        print(f"M={M:5d} N={N:5d} K={K:5d}  vendor {t_vendor:6.1f} us {fl / t_vendor / 1e6:6.1f} TF/s | ours {bm}x{bn}s{stg}/k{sk} "
              f"{t_ours:6.1f} us {fl / t_ours / 1e6:6.1f} TF/s{sweep}", flush=True)


if __name__ == "__main__":
    main()

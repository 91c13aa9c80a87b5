"""Micro-benchmark of msd_cross_attention_q (attn2.to_q + attention over the text context in one launch) against the two
launches it replaces, at the UNet's shapes, timed as the per-call time of a replayed hipGraph of N calls (HIP events).

    python tools/xattn_bench.py [--only IDX] [--calls 60] [--copies 90]

`--copies` weight matrices are rotated through (90 x 3.3 MB > the 256-MB Infinity Cache: every call streams its weights
from HBM, as in the denoise loop, where 1.7 GB of weights pass between two uses of a layer); --copies 1 = hot weights.
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [
    ("16x16 level  B=2 S=256 d=160", 2, 160, 256, 10),
    ("8x8 mid      B=2 S=64  d=160", 2, 160, 64, 20),
    ("16x16 level  B=8 S=256 d=160", 8, 160, 256, 10),
    ("32x32 level  B=2 S=1024 d=80", 2, 80, 1024, 10),
    ("64x64 level  B=2 S=4096 d=40", 2, 40, 4096, 5),
]


def graph_time(calls, reps=5):
    st = torch.cuda.current_stream()
    for c in calls[:2]:
        c(st.cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = torch.cuda.current_stream().cuda_stream
        for c in calls:
            c(s)
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / len(calls))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--calls", type=int, default=60)
    ap.add_argument("--copies", type=int, default=90)
    ap.add_argument("--modes", default="0", help="comma-separated xattn160_mode values (experiments of xattn_q160_kernel) to time the fused call under")
    args = ap.parse_args()
    from minsdtf_amd import _lib, ops, packing

    lib = _lib.load()
    lib.msd_init()
    dev = torch.device("cuda:0")
    T, H = 77, 8
    Tp = (T + 7) // 8 * 8
    for idx, (name, B, d, S, slots) in enumerate(SHAPES):
        if args.only >= 0 and idx != args.only:
            continue
        C = H * d
        torch.manual_seed(idx)
        x = (torch.randn(B * S, C) * 1.5 + 0.3).to(torch.bfloat16).to(dev)
        stats = torch.randn(B * S, slots, 2).abs().to(dev)
        k = torch.randn(B, T, C).to(torch.bfloat16).to(dev)
        vt = torch.randn(B, C, Tp).to(torch.bfloat16).to(dev)
        ncopy = max(1, min(args.copies, int(300e6 // (C * C * 2)) + 1))
        ws = []
        w0, cs, cb = packing.fold_layer_norm((torch.randn(C, C) / math.sqrt(C)).t().contiguous(), None, torch.ones(C).numpy(), torch.zeros(C).numpy(), dev)
        for _ in range(ncopy):
            ws.append(packing.chunk_major(w0.clone()))
        out = torch.empty(B * S, C, dtype=torch.bfloat16, device=dev)
        q2 = torch.empty(B * S, C, dtype=torch.bfloat16, device=dev)
        fused, two = [], []
        for i in range(args.calls):
            w = ws[i % ncopy]
            fused.append(ops.cross_attention_q(x=x, ln_in=stats, ln_in_slots=slots, wq=w, ln_colsum=cs, bias=cb, k=k, vt=vt, out=out, batch=B,
                                               heads=H, head_dim=d, s=S, t=T, k_ld=C, vt_ld=Tp, o_ld=C, w_layout=1))
            two.append(ops.conv_gemm(a0=x, w=w, out=q2, batch=1, h_in=B * S, w_in=1, c0=C, N=C, bias=cb, ln_in=stats, ln_in_slots=slots,
                                     ln_colsum=cs, w_layout=1))
            two.append(ops.attention(q=q2, k=k, vt=vt, out=out, batch=B, heads=H, head_dim=d, s=S, t=T, q_ld=C, k_ld=C, vt_ld=Tp, o_ld=C,
                                     scale=d ** -0.5, q_prescaled=True))
        tfs = []
        for m in [int(v) for v in args.modes.split(",")]:
            # (modes other than 0 exist in the instrumented build only: make -C minsdtf_amd/csrc stamps, MSD_HIP_LIB=tools/_build/libminsdtf_hip_stamps.so)
            _lib.check(lib.msd_set_option(b"xattn160_mode", m), "xattn160_mode")
            tfs.append(f"{graph_time(fused):7.2f}")
        lib.msd_set_option(b"xattn160_mode", 0)
        tt = graph_time(two) * 2
        print(f"{idx} {name:32s} weights x{ncopy:3d}: fused {' / '.join(tfs)} us   two launches {tt:7.2f} us (default tile of msd_conv_gemm, not the tuned one)", flush=True)


if __name__ == "__main__":
    main()

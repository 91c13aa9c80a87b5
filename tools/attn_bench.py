"""Micro-benchmark of msd_attention on the UNet's attention shapes (random bf16 data, HIP events).

    python tools/attn_bench.py [--only IDX] [--iters N]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [
    ("L0 self  S=4096 d=40", 2, 8, 40, 4096, 4096),
    ("L0 cross S=4096 T=77 d=40", 2, 8, 40, 4096, 77),
    ("L1 self  S=1024 d=80", 2, 8, 80, 1024, 1024),
    ("L2 self  S=256 d=160", 2, 8, 160, 256, 256),
    ("L0 self  B=8", 8, 8, 40, 4096, 4096),
    ("768^2 self S=9216 d=40", 2, 8, 40, 9216, 9216),
    ("L1 self  B=8", 8, 8, 80, 1024, 1024),
    ("L2 self  B=8", 8, 8, 160, 256, 256),
    ("L1 cross S=1024 T=77 d=80", 2, 8, 80, 1024, 77),
    ("L2 cross S=256 T=77 d=160", 2, 8, 160, 256, 77),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--qf", type=int, default=0, help="0 = automatic, 1 / 2 = 64 / 128 queries per workgroup")
    ap.add_argument("--presc", type=int, default=1, help="1 = q carries scale*log2(e) (the UNet's projections), 0 = generic")
    ap.add_argument("--form", type=int, default=2, help="2 = 32x32x16 MFMA kernel, software-pipelined, 1 = 32x32x16 plain, 0 = 16x16x32 MFMA kernel")
    args = ap.parse_args()
    from minsdtf_amd import _lib, ops

    lib = _lib.load()
    lib.msd_init()
    lib.msd_set_option(b"attn_qf", args.qf)
    _lib.check(lib.msd_set_option(b"attn_form", args.form), "attn_form")
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream()
    for idx, (name, B, H, d, S, T) in enumerate(SHAPES):
        if args.only >= 0 and idx != args.only:
            continue
        C = H * d
        Tp = (T + 7) // 8 * 8
        q = torch.randn(B, S, C, device=dev).to(torch.bfloat16)
        k = torch.randn(B, T, C, device=dev).to(torch.bfloat16)
        vt = torch.randn(B, C, Tp, device=dev).to(torch.bfloat16)
        out = torch.empty(B, S, C, device=dev, dtype=torch.bfloat16)
        call = ops.attention(q=q, k=k, vt=vt, out=out, batch=B, heads=H, head_dim=d, s=S, t=T, q_ld=C, k_ld=C, vt_ld=Tp, o_ld=C,
                             scale=d ** -0.5, q_prescaled=bool(args.presc))
        for _ in range(3):
            call(st.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(args.iters):
            call(st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        fl = 4.0 * B * H * S * T * d
        print(f"{idx} {name:28s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s (algorithmic)")


if __name__ == "__main__":
    main()

"""Micro-benchmark of msd_conv_gemm on the UNet's layer shapes (random bf16 data, HIP-event timing).

    python tools/conv_bench.py                 # table over the main shapes
    python tools/conv_bench.py --only 0 --iters 20    # one shape, e.g. under rocprofv3 --pmc ...
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [
    # (name, B, H, W, c0, c1, N, ksize, splitk)
    ("L0 conv3x3 320->320", 2, 64, 64, 320, 0, 320, 3, 1),
    ("L0 conv3x3 640->320 (concat)", 2, 64, 64, 320, 320, 320, 3, 1),
    ("L0 dense 320->320", 2, 64, 64, 320, 0, 320, 1, 1),
    ("L0 qkv 320->960", 2, 64, 64, 320, 0, 960, 1, 1),
    ("L0 geglu 320->2560", 2, 64, 64, 320, 0, 2560, 1, 1),
    ("L0 ff2 1280->320", 2, 64, 64, 1280, 0, 320, 1, 1),
    ("L1 conv3x3 640->640", 2, 32, 32, 640, 0, 640, 3, 4),
    ("L1 conv3x3 640->640 nosplit", 2, 32, 32, 640, 0, 640, 3, 1),
    ("L1 geglu 640->5120", 2, 32, 32, 640, 0, 5120, 1, 1),
    ("L2 conv3x3 1280->1280", 2, 16, 16, 1280, 0, 1280, 3, 6),
    ("L2 dense 1280->1280", 2, 16, 16, 1280, 0, 1280, 1, 1),
    ("L3 conv3x3 1280->1280", 2, 8, 8, 1280, 0, 1280, 3, 16),
    ("B8 L0 conv3x3 320->320", 8, 64, 64, 320, 0, 320, 3, 1),
    ("VAE conv3x3 128->128 @512", 1, 512, 512, 128, 0, 128, 3, 1),
    ("VAE conv3x3 512->512 @128", 1, 128, 128, 512, 0, 512, 3, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--tile-n", type=int, default=0)
    args = ap.parse_args()
    from minsdtf_amd import _lib, ops

    lib = _lib.load()
    lib.msd_init()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream()
    for idx, (name, B, H, W, c0, c1, N, ks, sk) in enumerate(SHAPES):
        if args.only >= 0 and idx != args.only:
            continue
        cin = c0 + c1
        x0 = torch.randn(B, H, W, c0, device=dev).to(torch.bfloat16)
        x1 = torch.randn(B, H, W, c1, device=dev).to(torch.bfloat16) if c1 else None
        w = (torch.randn(N, ks * ks * cin, device=dev) * 0.02).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        M = B * H * W
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ws = torch.empty(max(1, sk * M * N), device=dev, dtype=torch.float32)
        call = ops.conv_gemm(a0=x0, a1=x1, c1=c1, w=w, out=out, batch=B, h_in=H, w_in=W, c0=c0, N=N, ksize=ks, bias=bias,
                             workspace=ws, workspace_floats=ws.numel(), splitk=sk, tile_n=args.tile_n)
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
        fl = 2.0 * M * N * ks * ks * cin
        print(f"{idx:2d} {name:34s} M={M:6d} N={N:5d} K={ks * ks * cin:6d} sk={sk:2d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s")


if __name__ == "__main__":
    main()

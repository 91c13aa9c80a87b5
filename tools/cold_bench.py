"""What do HBM-cold weights cost a weight-streaming layer?  The tuned msd_conv_gemm launch of the UNet's weight-heavy shapes
(batch-1 step = fused batch 2), timed as the per-call time of a replayed hipGraph whose calls rotate through `copies`
copies of the weight matrix: few copies (together < the 256-MB Infinity Cache) = every call finds its weights there,
many copies (together > 2 x 256 MB) = every call streams them from HBM, as in the denoise loop, where 1.7 GB of weights
pass between two uses of a layer.

    python tools/cold_bench.py [--only IDX] [--calls 48]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [
    # (name, B, H, W, c0, c1, N, ksize)
    ("8x8   conv3x3 1280->1280", 2, 8, 8, 1280, 0, 1280, 3),
    ("8x8   conv3x3 2560->1280 (concat)", 2, 8, 8, 1280, 1280, 1280, 3),
    ("16x16 conv3x3 1280->1280", 2, 16, 16, 1280, 0, 1280, 3),
    ("16x16 conv3x3 2560->1280 (concat)", 2, 16, 16, 1280, 1280, 1280, 3),
    ("16x16 dense 1280->1280", 2, 16, 16, 1280, 0, 1280, 1),
    ("32x32 conv3x3 640->640", 2, 32, 32, 640, 0, 640, 3),
    ("32x32 conv3x3 1280->640 (concat)", 2, 32, 32, 640, 640, 640, 3),
    ("64x64 conv3x3 320->320", 2, 64, 64, 320, 0, 320, 3),
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
    ap.add_argument("--calls", type=int, default=48)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT", help="msd_set_option switch, e.g. --opt splitk_xcd=0")
    args = ap.parse_args()
    from minsdtf_amd import _lib, ops, packing, tuning

    lib = _lib.load()
    lib.msd_init()
    for kv in args.opt:
        k, v = kv.split("=")
        _lib.check(lib.msd_set_option(k.encode(), int(v)), kv)
    dev = torch.device("cuda:0")
    for idx, (name, B, H, W, c0, c1, N, ks) in enumerate(SHAPES):
        if args.only >= 0 and idx != args.only:
            continue
        cin = c0 + c1
        K = ks * ks * cin
        M = B * H * W
        tile_m, tile_n, sk, stages = tuning.lookup(B, H, W, cin, N, ks, 1, False, M, K // 64, True, 0)
        x0 = torch.randn(B, H, W, c0, device=dev).to(torch.bfloat16)
        x1 = torch.randn(B, H, W, c1, device=dev).to(torch.bfloat16) if c1 else None
        w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        wreg = tuning.is_wreg(tile_m)
        wbytes = N * K * 2
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ws = torch.empty(max(1, sk * M * N), device=dev, dtype=torch.float32)
        res = []
        for label, total in (("cache", 100e6), ("HBM", 600e6)):
            ncopy = max(2, int(total // wbytes)) if label == "cache" else max(4, int(total // wbytes) + 1)
            if label == "cache" and ncopy * wbytes > 200e6:
                ncopy = max(1, int(200e6 // wbytes))
            mats = [(packing.fragment_major(w.clone()) if wreg else packing.chunk_major(w.clone())) for _ in range(ncopy)]
            calls = [ops.conv_gemm(a0=x0, a1=x1, c1=c1, w=mats[i % ncopy], out=out, batch=B, h_in=H, w_in=W, c0=c0, N=N, ksize=ks, bias=bias,
                                   workspace=ws if sk > 1 else None, workspace_floats=ws.numel() if sk > 1 else 0, splitk=sk, tile_m=tile_m,
                                   tile_n=tile_n, stages=stages, w_layout=2 if wreg else 1) for i in range(args.calls)]
            res.append(f"{label} (x{ncopy:3d}) {graph_time(calls):7.2f} us")
            del calls, mats
            torch.cuda.empty_cache()
        print(f"{idx} {name:34s} M={M:5d} K={K:6d} {wbytes / 1e6:5.1f} MB  tile {tile_m}x{tile_n} sk {sk} st {stages}:  " + "   ".join(res), flush=True)


if __name__ == "__main__":
    main()

"""GroupNorm forms per shape: the per-(sample, group) cluster / one-workgroup forms against the row-major cluster form (csrc/norm.hip
gn_rows_kernel), isolated launches of one stream (HIP events over 200 back-to-back calls).

    python tools/gn_bench.py [--batches 2 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(4096, 320, 0), (4096, 640, 0), (4096, 320, 320), (4096, 640, 320), (1024, 320, 0), (1024, 640, 0), (1024, 640, 640), (1024, 1280, 640),
          (1024, 640, 320), (256, 1280, 0), (256, 1280, 1280), (256, 1280, 640), (256, 640, 0), (9216, 320, 0), (9216, 320, 320), (2304, 640, 0),
          (4096, 512, 0), (16384, 512, 0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="*", default=[2, 8])
    ap.add_argument("--kb", type=int, default=1)
    args = ap.parse_args()
    from minsdtf_amd import _lib, ops

    lib = _lib.load()
    lib.msd_init()
    d = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for hw, c0, c1 in SHAPES:
        C = c0 + c1
        for B in args.batches:
            if B * hw * C * 2 > (1 << 30):
                continue
            x0 = torch.randn(B, hw, c0, device=d).to(torch.bfloat16)
            x1 = torch.randn(B, hw, c1, device=d).to(torch.bfloat16) if c1 else None
            g, b_ = torch.ones(C, device=d), torch.zeros(C, device=d)
            stats = torch.empty(B * 64, device=d)
            partials = torch.empty(B * ops.GN_MAX_CHUNKS * 64, device=d)
            sync = torch.zeros(B * ops.GN_SYNC_WORDS_PER_SAMPLE, dtype=torch.int32, device=d)
            out = torch.empty(B, hw, C, device=d, dtype=torch.bfloat16)
            res = []
            for kb in (0, args.kb):
                lib.msd_set_option(b"gn_rows", kb)
                call = ops.group_norm(partials=partials, x0=x0, x1=x1, gamma=g, beta=b_, stats=stats, out=out, batch=B, hw=hw, c0=c0, c1=c1, silu=True, sync=sync)
                for _ in range(5):
                    call(st)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200):
                    call(st)
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) * 5.0)
            mb = B * hw * C * 4 / 1e6
            print(f"hw {hw:6d} C {c0:4d}+{c1:<4d} batch {B}: groups {res[0]:7.2f} us  rows {res[1]:7.2f} us  ratio {res[1] / res[0]:.2f}   ({mb:.1f} MB: {mb / res[0] / 1e3:.2f} / {mb / res[1] / 1e3:.2f} TB/s)", flush=True)
    lib.msd_set_option(b"gn_rows", 9216)


if __name__ == "__main__":
    main()

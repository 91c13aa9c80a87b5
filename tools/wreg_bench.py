"""Isolated timing of the wreg form (csrc/conv_wreg.hip) against the other forms of msd_conv_gemm on chosen layer shapes
(tools/tune_conv.py's method: HIP events around back-to-back launches, random bf16 data, weights rotating through > 256 MiB):

    python tools/wreg_bench.py [--fused-batch 2] [--match SUBSTR] [--iters 20] [--json OUT.json]

Prints, per shape of one UNet step, the fastest configuration of each form (tile / halo / row-panel / wreg) and the wreg
candidates in order."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def form(bm):
    return "wreg" if bm >= 4000 else "rowpanel" if bm >= 3000 else "halo" if bm >= 1000 else "tile"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fused-batch", type=int, default=2)
    ap.add_argument("--match", default="")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from minsdtf_amd import _lib, tuning
    from tools.tune_conv import tune_one
    from tools.vendor_yardstick import unet_shapes

    _lib.load().msd_init()
    counts = unet_shapes(args.fused_batch)
    allr = {}
    tot = {"best_other": 0.0, "best_any": 0.0}
    for shape, n in counts.items():
        key = tuning.shape_key(*shape)
        if shape[4] < 16 or args.match not in key:
            continue
        _best, results, flop = tune_one(shape, iters=args.iters)
        allr[key] = [list(r) for r in results]
        per = {}
        for us, bm, bn, sk, stg in results:
            f = form(bm)
            if f not in per:
                per[f] = (us, bm, bn, sk, stg)
        other = min((v for f, v in per.items() if f != "wreg"), default=None)
        w = per.get("wreg")
        tot["best_other"] += n * other[0]
        tot["best_any"] += n * min(other[0], w[0] if w else 1e9)
        line = f"{key:42s} x{n:2d} " + "  ".join(f"{f} {v[0]:6.1f} ({v[1]}x{v[2]} s{v[4]} k{v[3]})" for f, v in sorted(per.items()))
        if w:
            line += f"   wreg/other {w[0] / other[0]:.2f}  [{flop / w[0] / 1e6:.0f} TF/s]"
        print(line, flush=True)
        wl = [r for r in results if r[1] >= 4000][:6]
        print("      wreg: " + "  ".join(f"{bm - 4000}x{bn} s{stg} k{sk}: {us:.1f}" for us, bm, bn, sk, stg in wl), flush=True)
    print(f"sum over one step's launches: best non-wreg {tot['best_other'] / 1e3:.3f} ms, best of all forms {tot['best_any'] / 1e3:.3f} ms")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(allr, f)


if __name__ == "__main__":
    main()

"""Round 6: ONE numerics class for the shortcut-folded 3x3 convs (delete the second profile).

The staged-halo big form walks conv2(h) + conv_shortcut(x) chunk-major (a slice's main chunks, then its shortcut chunks) and
wins from two images per GPU; at one image it lost to the tile kernels (tap-major: another class), hence round 5's
MSD_PROFILE=throughput overlay.  The halo-tile kernel now walks the same order (csrc/conv_halo.hip), so every batch of those
layers can live in the chunk-major class: this tool times, per layer of minsdtf_amd/conv_tuning_throughput.json and per
measured batch, the table's current entry, every halo-tile configuration and every staged-halo configuration at the
overlay's split count (isolated launches, tools/tune_conv.py tune_one), and writes the per-batch winners INSIDE the class.

    python tools/unify_shortcut.py --out gpurun_out/unified_shortcut.json [--iters 20]

Whether the loop gains at batch 1 is tools/ab_loop.py's question (--variant base --variant tune:FILE)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--overlay", default=os.path.join(ROOT, "tools", "shortcut_layers.json"))
    args = ap.parse_args()
    import tune_conv
    from big_retarget import parse_key
    from minsdtf_amd import _lib, tuning

    _lib.load().msd_init()
    table = json.load(open(os.path.join(ROOT, "minsdtf_amd", "conv_tuning.json")))
    overlay = json.load(open(args.overlay))
    out, log = {}, {}
    for key in sorted(overlay, key=lambda k: (k.split("x", 1)[1], int(k.split("x")[0]))):
        s = parse_key(key)
        sk = int(overlay[key][2])
        cur = table[key]
        ctm, ctn, csk, cstg = int(cur[0]), int(cur[1]), int(cur[2]), int(cur[3]) if len(cur) > 4 else 0
        t_cur, _, flop = tune_conv.tune_one(s, iters=args.iters, only=lambda t: t == (ctm, ctn, cstg), sks_only=[csk])
        in_class = lambda t: tuning.is_halo(t[0]) or (tuning.is_big(t[0]) and t[2] >= 20)   # noqa: E731
        best, results, _ = tune_conv.tune_one(s, iters=args.iters, only=in_class, sks_only=[sk])
        halo = [r for r in results if tuning.is_halo(r[1])]
        big = [r for r in results if tuning.is_big(r[1])]
        fmt = lambda r: "-" if r is None else f"{r[1]}x{r[2]}s{r[4]}k{r[3]} {r[0]:7.1f} us"   # noqa: E731
        print(f"{key:40s} table {fmt(t_cur)} | best halo {fmt(halo[0] if halo else None)} | best staged halo {fmt(big[0] if big else None)} "
              f"| class winner vs table {1.0 - best[0] / t_cur[0]:+.1%}", flush=True)
        out[key] = [int(best[1]), int(best[2]), int(best[3]), int(best[4])]
        log[key] = {"table": list(t_cur), "halo": [list(r) for r in halo[:4]], "staged_halo": [list(r) for r in big[:3]]}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=0)
    with open(args.out.replace(".json", "_log.json"), "w") as f:
        json.dump(log, f)
    print("wrote", args.out)


if __name__ == "__main__":
    main()

"""Targeted, class-preserving replacement of conv_tuning.json entries by the big form (csrc/conv_big.hip).

    python tools/big_retarget.py [--min-gain 0.04] [--out minsdtf_amd/conv_tuning.json] [--log gpurun_out/big_retarget.json]

For every table entry with M >= tuning.BIG_MIN_ROWS: re-time the entry, time every big candidate INSIDE THE ENTRY'S NUMERICS
CLASS (same split-K; tap-major candidates for entries of the tile / wreg / row-panel class, chunk-major ones (code + 10) for
entries of the halo class; never for the LayerNorm-producer Dense layers) on isolated launches (tools/tune_conv.py tune_one), and
replace the entry when the best candidate is faster by more than --min-gain.  The table's one-class-per-layer invariant
(tests/test_host_cpu.py) holds by construction; whether the loop gains is tools/ab_loop.py's question."""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_key(key):
    m = re.match(r"(\d+)x(\d+)x(\d+)x(\d+)->(\d+)k(\d)s(\d)u([01])(n?)(?:\+x(\d+))?$", key)
    b, h, w, cin, n, ks, st, ups, nos, cx = m.groups()
    return (int(b), int(h), int(w), int(cin), int(n), int(ks), int(st), bool(int(ups)), nos != "n", int(cx or 0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-gain", type=float, default=0.04)
    ap.add_argument("--out", default=os.path.join(ROOT, "minsdtf_amd", "conv_tuning.json"))
    ap.add_argument("--log", default=None)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    import tune_conv
    from minsdtf_amd import _lib, tuning

    _lib.load().msd_init()
    path = os.path.join(ROOT, "minsdtf_amd", "conv_tuning.json")
    table = json.load(open(path))
    log = {}
    n_rep = 0
    for key in sorted(table):
        ent = table[key]
        s = parse_key(key)
        batch, h_in, w_in, cin, N, ks, stride, ups, allow_split, cx = s
        pad = 1 if ks == 3 else 0
        hl, wl = (2 * h_in, 2 * w_in) if ups else (h_in, w_in)
        M = batch * ((hl + 2 * pad - ks) // stride + 1) * ((wl + 2 * pad - ks) // stride + 1)
        if M < tuning.BIG_MIN_ROWS or N < 64:
            continue
        tm, tn, sk, stg = int(ent[0]), int(ent[1]), int(ent[2]), int(ent[3]) if len(ent) > 4 else 0
        if tuning.is_big(tm):
            continue
        cls = tune_conv.numerics_class(s, tm, tn, sk, stg)
        best_t, _, flop = tune_conv.tune_one(s, iters=args.iters, only=lambda t: t == (tm, tn, stg), sks_only=[sk])
        if best_t is None:
            continue
        ok = lambda t: t[0] >= 5000 and tune_conv.numerics_class(s, t[0], t[1], sk, t[2]) == cls   # noqa: E731
        best_b, res_b, _ = tune_conv.tune_one(s, iters=args.iters, only=ok, sks_only=[sk])
        if best_b is None:
            continue
        gain = 1.0 - best_b[0] / best_t[0]
        rep = gain > args.min_gain
        log[key] = dict(table=[tm, tn, sk, stg, round(best_t[0], 1)], big=[best_b[1], best_b[2], best_b[3], best_b[4], round(best_b[0], 1)],
                        tf_table=round(flop / best_t[0] / 1e6), tf_big=round(flop / best_b[0] / 1e6), replaced=rep)
        print(f"{key:44s} {tm}x{tn}s{stg}k{sk} {best_t[0]:8.1f} us ({flop / best_t[0] / 1e6:5.0f} TF) | {best_b[1]}x{best_b[2]}s{best_b[4]} {best_b[0]:8.1f} us "
              f"({flop / best_b[0] / 1e6:5.0f} TF) {'-> REPLACED' if rep else ''} ({gain:+.1%})", flush=True)
        if rep:
            table[key] = [best_b[1], best_b[2], best_b[3], best_b[4], round(best_b[0], 1)]
            n_rep += 1
    with open(args.out, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    if args.log:
        with open(args.log, "w") as f:
            json.dump(log, f, indent=1, sort_keys=True)
    print(f"{n_rep} entries replaced; wrote {args.out}")


if __name__ == "__main__":
    main()

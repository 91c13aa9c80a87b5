"""Time the big form (csrc/conv_big.hip) against the table's entry on every conv / dense shape with M >= 4096 rows of the
configurations named on the command line (isolated launches, rotating weights: tools/tune_conv.py tune_one).

    python tools/big_bench.py [b4] [b1] [768] [vae] [--all-forms]

For every shape: the table's entry re-timed, the best big candidate at the table's split-K (the numerics class), the ratio."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import tune_conv
    from minsdtf_amd import _lib, tuning

    what = [a for a in sys.argv[1:] if not a.startswith("--")] or ["b4"]
    shapes = tune_conv.collect_shapes(False)
    _lib.load().msd_init()
    table = tuning._load()
    rows = []
    tot_tab = tot_best = 0.0
    for s in shapes:
        batch, h_in, w_in, cin, N, ks, stride, ups, allow_split, cx = s
        pad = 1 if ks == 3 else 0
        hl, wl = (2 * h_in, 2 * w_in) if ups else (h_in, w_in)
        M = batch * ((hl + 2 * pad - ks) // stride + 1) * ((wl + 2 * pad - ks) // stride + 1)
        if M < int(os.environ.get("BIG_BENCH_MIN_ROWS", tuning.BIG_MIN_ROWS)):
            continue
        key = tuning.shape_key(*s)
        tag = "vae" if (cin in (128, 256, 512) and N in (128, 256, 512)) or h_in >= 128 else ("768" if h_in in (96, 48, 24, 12) and w_in == h_in else ("b4" if batch == 8 else ("b2" if batch == 4 else "b1")))
        if tag not in what:
            continue
        ent = table.get(key)
        if ent is None:
            continue
        tm, tn, sk, stg = int(ent[0]), int(ent[1]), int(ent[2]), int(ent[3])
        best_t, res_t, flop = tune_conv.tune_one(s, iters=20, only=lambda t: t == (tm, tn, stg), sks_only=[sk])
        best_b, res_b, _ = tune_conv.tune_one(s, iters=20, only=lambda t: t[0] >= 5000, sks_only=[sk])
        if best_b is None:
            continue
        us_t, us_b = best_t[0], best_b[0]
        rows.append((key, M, N, ks * ks * cin + cx, sk, f"{tm}x{tn}s{stg}", us_t, f"{best_b[1]}x{best_b[2]}s{best_b[4]}", us_b, flop))
        tot_tab += us_t
        tot_best += min(us_t, us_b)
        alt = " ".join(f"{bm}x{bn}s{st}:{us:.0f}" for us, bm, bn, k_, st in res_b[:5])
        print(f"{key:40s} M={M:6d} N={N:5d} K={ks*ks*cin+cx:6d} k{sk} table {tm}x{tn}s{stg} {us_t:7.1f} us ({flop/us_t/1e6:5.0f} TF) | big {best_b[1]}x{best_b[2]}s{best_b[4]} {us_b:7.1f} us ({flop/us_b/1e6:5.0f} TF) ratio {us_b/us_t:.2f} [{alt}]", flush=True)
    print(f"sum table {tot_tab:.0f} us, best-of-two {tot_best:.0f} us ({tot_best/tot_tab-1:+.1%})")
    out = os.environ.get("BIG_BENCH_OUT")
    if out:
        with open(out, "w") as f:
            json.dump(rows, f)


if __name__ == "__main__":
    main()

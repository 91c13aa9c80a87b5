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
    ap.add_argument("--halo-image", action="store_true",
                    help="only the halo-image candidates (stages >= 20), also against entries that are already on the big form")
    ap.add_argument("--reclass-upsample", action="store_true",
                    help="upsampling 3x3 layers whose every measured batch has M >= BIG_MIN_ROWS: move the LAYER (all its batches) from the "
                         "tile class to the chunk-major class when the chunk-major candidates are faster for every batch")
    ap.add_argument("--reclass-shortcut", action="store_true",
                    help="the THROUGHPUT profile's overlay (minsdtf_amd/conv_tuning_throughput.json, --out): shortcut-folded 3x3 layers (+x keys) whose "
                         "every measured batch is whole 16 x 16-pixel tiles move - all their batches - to the staged-halo big form when the batches of "
                         "two or more images per GPU (fused batch >= 4) gain; the one-image batch pays (that is the profile's trade)")
    ap.add_argument("--only", default=None, help="regular expression on the key")
    args = ap.parse_args()
    import tune_conv
    from minsdtf_amd import _lib, tuning

    _lib.load().msd_init()
    path = os.path.join(ROOT, "minsdtf_amd", "conv_tuning.json")
    table = json.load(open(path))
    log = {}
    n_rep = 0
    def rows_of(s):
        batch, h_in, w_in, cin, N, ks, stride, ups, allow_split, cx = s
        pad = 1 if ks == 3 else 0
        hl, wl = (2 * h_in, 2 * w_in) if ups else (h_in, w_in)
        return batch * ((hl + 2 * pad - ks) // stride + 1) * ((wl + 2 * pad - ks) // stride + 1)

    if args.reclass_shortcut:
        groups, overlay = {}, {}
        for key in sorted(table):
            s = parse_key(key)
            if s[9] and s[5] == 3 and s[6] == 1 and not s[7] and s[1] % 16 == 0 and s[2] % 16 == 0 and (args.only is None or re.search(args.only, key)):
                groups.setdefault(key.split("x", 1)[1], []).append(key)
        for rest, keys in sorted(groups.items()):
            if any(rows_of(parse_key(k)) < tuning.HALO_IMAGE_MIN_ROWS for k in keys):
                print(f"  layer {rest}: a measured batch below {tuning.HALO_IMAGE_MIN_ROWS} rows: left alone")
                continue
            plan, t_old, t_new = {}, 0.0, 0.0
            for key in keys:
                s = parse_key(key)
                ent = table[key]
                tm, tn, sk, stg = int(ent[0]), int(ent[1]), int(ent[2]), int(ent[3]) if len(ent) > 4 else 0
                best_t, _, flop = tune_conv.tune_one(s, iters=args.iters, only=lambda t: t == (tm, tn, stg), sks_only=[sk])
                best_b, _, _ = tune_conv.tune_one(s, iters=args.iters, only=lambda t: t[0] >= 5000 and t[2] >= 20, sks_only=[sk])
                if best_t is None or best_b is None:
                    plan = None
                    break
                print(f"{key:44s} {tm}x{tn}s{stg}k{sk} {best_t[0]:8.1f} us ({flop / best_t[0] / 1e6:5.0f} TF) | {best_b[1]}x{best_b[2]}s{best_b[4]} {best_b[0]:8.1f} us "
                      f"({flop / best_b[0] / 1e6:5.0f} TF) ({1.0 - best_b[0] / best_t[0]:+.1%})", flush=True)
                if s[0] >= 4:
                    t_old += best_t[0]
                    t_new += best_b[0]
                plan[key] = ([tm, tn, sk, stg, round(best_t[0], 1)], [best_b[1], best_b[2], best_b[3], best_b[4], round(best_b[0], 1)])
            if plan and t_old > 0 and t_new < (1.0 - args.min_gain) * t_old:
                for key, (old_e, new_e) in plan.items():
                    overlay[key] = new_e
                    log[key] = dict(table=old_e, big=new_e, replaced=True, profile="throughput")
                    n_rep += 1
                print(f"  layer {rest}: into the throughput overlay ({len(plan)} entries; fused batches >= 4: {t_old:.0f} -> {t_new:.0f} us)")
            else:
                print(f"  layer {rest}: left alone")
        with open(args.out, "w") as f:
            json.dump(overlay, f, indent=0, sort_keys=True)
        if args.log:
            with open(args.log, "w") as f:
                json.dump(log, f, indent=1, sort_keys=True)
        print(f"{n_rep} entries in the overlay; wrote {args.out}")
        return
    if args.reclass_upsample:
        groups = {}
        for key in sorted(table):
            s = parse_key(key)
            if s[7] and s[5] == 3 and s[6] == 1 and not s[9] and (args.only is None or re.search(args.only, key)):
                groups.setdefault(key.split("x", 1)[1], []).append(key)
        for rest, keys in sorted(groups.items()):
            if any(rows_of(parse_key(k)) < tuning.HALO_IMAGE_MIN_ROWS for k in keys):
                continue
            plan, ok_all, any_gain = {}, True, False
            for key in keys:
                s = parse_key(key)
                ent = table[key]
                tm, tn, sk, stg = int(ent[0]), int(ent[1]), int(ent[2]), int(ent[3]) if len(ent) > 4 else 0
                if tune_conv.numerics_class(s, tm, tn, sk, stg)[0]:
                    ok_all = False   # (already chunk-major)
                    break
                best_t, _, flop = tune_conv.tune_one(s, iters=args.iters, only=lambda t: t == (tm, tn, stg), sks_only=[sk])
                best_b, _, _ = tune_conv.tune_one(s, iters=args.iters, only=lambda t: t[0] >= 5000 and t[2] >= 10, sks_only=[sk])
                if best_t is None or best_b is None:
                    ok_all = False
                    break
                gain = 1.0 - best_b[0] / best_t[0]
                print(f"{key:44s} {tm}x{tn}s{stg}k{sk} {best_t[0]:8.1f} us ({flop / best_t[0] / 1e6:5.0f} TF) | {best_b[1]}x{best_b[2]}s{best_b[4]} {best_b[0]:8.1f} us "
                      f"({flop / best_b[0] / 1e6:5.0f} TF) ({gain:+.1%})", flush=True)
                ok_all = ok_all and gain > -0.02
                any_gain = any_gain or gain > args.min_gain
                plan[key] = ([tm, tn, sk, stg, round(best_t[0], 1)], [best_b[1], best_b[2], best_b[3], best_b[4], round(best_b[0], 1)])
            if ok_all and any_gain:
                for key, (old_e, new_e) in plan.items():
                    table[key] = new_e
                    log[key] = dict(table=old_e, big=new_e, replaced=True, reclassed=True)
                    n_rep += 1
                print(f"  layer {rest}: moved to the chunk-major class ({len(plan)} entries)")
            else:
                print(f"  layer {rest}: left alone")
    for key in sorted(table):
        if args.reclass_upsample:
            break
        if args.only is not None and not re.search(args.only, key):
            continue
        ent = table[key]
        s = parse_key(key)
        batch, h_in, w_in, cin, N, ks, stride, ups, allow_split, cx = s
        pad = 1 if ks == 3 else 0
        hl, wl = (2 * h_in, 2 * w_in) if ups else (h_in, w_in)
        M = batch * ((hl + 2 * pad - ks) // stride + 1) * ((wl + 2 * pad - ks) // stride + 1)
        if M < (tuning.HALO_IMAGE_MIN_ROWS if args.halo_image else tuning.BIG_MIN_ROWS) or N < 64:
            continue
        tm, tn, sk, stg = int(ent[0]), int(ent[1]), int(ent[2]), int(ent[3]) if len(ent) > 4 else 0
        if tuning.is_big(tm) and not (args.halo_image and stg < 20):
            continue
        cls = tune_conv.numerics_class(s, tm, tn, sk, stg)
        best_t, _, flop = tune_conv.tune_one(s, iters=args.iters, only=lambda t: t == (tm, tn, stg), sks_only=[sk])
        if best_t is None:
            continue
        ok = lambda t: t[0] >= 5000 and (t[2] >= 20 or not args.halo_image) and tune_conv.numerics_class(s, t[0], t[1], sk, t[2]) == cls   # noqa: E731
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

"""In-place ranking of the chunk-major shortcut class (halo-tile kernel with shortcut steps / staged-halo big form) for the
ResBlocks with a 1x1 shortcut, against what the table launches today - measured where it counts: HIP events around each
layer's launches in an eager sampler step of the real engine (weights streaming from HBM, the producer's output where it
left it), one pass per candidate configuration (tools/tune_insitu.py's method).  Isolated timing (tools/unify_shortcut.py)
ranks these layers up to 15 % differently.

    python tools/unify_shortcut_insitu.py --out gpurun_out/unified_insitu.json [--batches 1,2,4] [--fold-all]

--fold-all also folds the 64x64-level blocks (engine.SHORTCUT_FOLD_MAX_PIXELS off) and compares each block's ONE folded launch
with the conv2 + conv_shortcut pair the table runs there today."""
import argparse
import collections
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--batches", default="1,2,4")
    ap.add_argument("--fold-all", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import tune_insitu as ti
    from minsdtf_amd import engine, tuning
    from minsdtf_amd.stable_diffusion import StableDiffusion

    dev = torch.device("cuda:0")
    sd = StableDiffusion(512, 512, jit_compile=False, device=dev)
    sd.diffusion_model.load_synthetic(seed=0)
    overlay = json.load(open(os.path.join(ROOT, "tools", "shortcut_layers.json")))
    sk_of = {k.split("x", 1)[1]: int(v[2]) for k, v in overlay.items()}     # the class's split count per layer
    cands = [t for t in tuning.HALO_TILES] + [t for t in tuning.BIG_TILES_HALO_IMAGE]
    result, log = {}, {}
    for B in [int(b) for b in args.batches.split(",")]:
        nb = 2 * B
        hook = ti.Hook(tuning, engine)
        try:
            # pass 0: the table as it is (folded where the engine folds today)
            eng = ti.build_engine(sd, B, True, False)
            base_t = dict(zip([c.name for c in eng.calls], ti.time_calls(eng, args.reps)))
            key_of0 = dict(hook.key_of)
            del eng
            if args.fold_all:
                engine.SHORTCUT_FOLD_MAX_PIXELS = 1 << 30
            per_key = collections.defaultdict(dict)   # key -> {candidate: us}
            names_of = collections.defaultdict(list)
            for cand in cands:
                tm, tn, stg = cand
                hook.override = {}
                for rest, sk in sk_of.items():
                    key = f"{nb}x{rest}"
                    hook.override[key] = (tm, tn, sk, stg)
                hook.key_of = {}
                try:
                    eng = ti.build_engine(sd, B, True, False)
                    ts = dict(zip([c.name for c in eng.calls], ti.time_calls(eng, args.reps)))
                except Exception as e:   # (a candidate the kernel refuses for one of the layers: skip the pass)
                    print(f"batch {B} candidate {cand}: {type(e).__name__}: {str(e)[:120]}", flush=True)
                    continue
                for name, key in hook.key_of.items():
                    if key in hook.override and name in ts:
                        per_key[key].setdefault(cand, 0.0)
                        per_key[key][cand] += ts[name]
                        if name not in names_of[key]:
                            names_of[key].append(name)
                del eng
            for key in sorted(per_key):
                n_calls = len(names_of[key])
                # what the same ResBlocks cost today: the folded launch, or conv2 + conv_shortcut where the engine does not fold
                today = 0.0
                for name in names_of[key]:
                    rb = name[: -len(".conv2sc")]
                    if name in base_t:
                        today += base_t[name]
                    else:
                        today += base_t.get(rb + ".conv2", 0.0) + base_t.get(rb + ".conv_shortcut", 0.0)
                ranked = sorted(per_key[key].items(), key=lambda kv: kv[1])
                best, t_best = ranked[0]
                print(f"batch {B} {key:40s} x{n_calls}: today {today / n_calls:7.1f} us | " +
                      " | ".join(f"{c[0]}x{c[1]}s{c[2]} {t / n_calls:6.1f}" for c, t in ranked[:4]) + f" | best vs today {1.0 - t_best / today:+.1%}", flush=True)
                result[key] = [best[0], best[1], sk_of[key.split("x", 1)[1]], best[2]]
                log[key] = {"today_us": today / n_calls, "calls": n_calls, "ranked": [[list(c), t / n_calls] for c, t in ranked]}
        finally:
            hook.close()
            engine.SHORTCUT_FOLD_MAX_PIXELS = int(os.environ.get("MSD_SHORTCUT_FOLD_MAX_PIXELS", str(1 << 30)))
    with open(args.out, "w") as f:
        json.dump(result, f, indent=0)
    with open(args.out.replace(".json", "_log.json"), "w") as f:
        json.dump(log, f)
    print("wrote", args.out)


if __name__ == "__main__":
    main()

"""Second tuning stage: re-rank the launch configurations of msd_conv_gemm IN PLACE, inside the real sampler step.

    python tools/tune_conv.py --dump gpurun_out/tune_all.json          # stage 1: every candidate, isolated, random data
    python tools/tune_insitu.py --micro gpurun_out/tune_all.json [--out minsdtf_amd/conv_tuning.json] [--top 8]

Stage 1 times a layer alone, back to back on the same buffers; in the pipeline the same launch finds its weights in HBM
(1.7 GB stream through every step), its input in whatever state the producer left it in L2 / MALL, and carries its real
epilogue (LayerNorm fold, GEGLU, time-embedding row, shortcut operand).  The two rankings differ by up to 15 % per layer
(profiles/ r2 notes).  This stage takes, per layer shape FAMILY (shape without the batch), the union of the stage-1 top-N
of every batch plus the installed entry, and measures each candidate with HIP events around the layer's launches in an
eager sampler step of the real engine (synthetic weights): pass k runs candidate k of every layer at once, so the number
of passes is the longest candidate list, not the number of layers x candidates.  One numerics class per family is then
chosen exactly as in stage 1 (tools/tune_conv.py: pin_classes) from the in-place times.  Layers of plans that are not
walked here (VAE, preparation plans) keep their stage-1 entries."""
import argparse
import collections
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

# (image batch, latent size, guidance on, ControlNet): the sampler steps whose layers are re-ranked
PIPELINES = ((1, 64, True, False), (2, 64, True, False), (4, 64, True, False), (1, 64, False, False), (1, 96, True, False),
             (1, 64, True, True))


def parse_key(key):
    """shape tuple of tools/tune_conv.py from a tuning.shape_key string."""
    import re

    m = re.match(r"(\d+)x(\d+)x(\d+)x(\d+)->(\d+)k(\d)s(\d)u(\d)(n?)(?:\+x(\d+))?$", key)
    b, h, w, cin, N, ks, st, up, n, cx = m.groups()
    return (int(b), int(h), int(w), int(cin), int(N), int(ks), int(st), bool(int(up)), n != "n", int(cx or 0))


class Hook:
    """Replaces tuning.lookup while engines are built: serves the override of the current pass and notes which layer
    (call name) asked for which key."""

    def __init__(self, tuning, engine):
        self.tuning, self.engine = tuning, engine
        self.orig_lookup, self.orig_conv = tuning.lookup, engine.Emitter.conv
        self.override = {}
        self.key_of = {}
        self.cur = None
        hook = self

        def lookup(batch, h_in, w_in, cin, N, ksize, stride, upsample, M, nk, allow_split, cx=0):
            key = tuning.shape_key(batch, h_in, w_in, cin, N, ksize, stride, upsample, allow_split, cx)
            hook.key_of[hook.cur] = key
            o = hook.override.get(key)
            if o is not None:
                bm, bn, sk, stg = o
                if bm == 256 and M < 1024:
                    bm = 128
                return bm, bn, (sk if allow_split else 1), stg
            return hook.orig_lookup(batch, h_in, w_in, cin, N, ksize, stride, upsample, M, nk, allow_split, cx)

        def conv(self_, x, name, *a, **k):
            hook.cur = name
            return hook.orig_conv(self_, x, name, *a, **k)

        tuning.lookup = lookup
        engine.Emitter.conv = conv

    def close(self):
        self.tuning.lookup = self.orig_lookup
        self.engine.Emitter.conv = self.orig_conv


def build_engine(sd, B, cfg, control):
    from minsdtf_amd.stable_diffusion import DenoiseEngine

    g = 7.5 if cfg else 0.0
    eng = DenoiseEngine(sd.diffusion_model, B, 77, 77, 25, g, 0.7 if cfg else 0.0, control_net=sd.control_net if control else None,
                        hint_net=sd.hint_net if control else None, use_graph=False)
    rng = np.random.default_rng(1234)
    ctx = rng.standard_normal((B, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((B, 77, 768)).astype(np.float32)
    h = sd.img_height // 8
    noise = np.random.default_rng(0).standard_normal((B, h, h, 4)).astype(np.float32)
    hint = np.random.default_rng(7).random((B, 8 * h, 8 * h, 3)).astype(np.float32) if control else None
    sd.scheduler.set_timesteps(25)
    eng.prepare(eng.contexts(unc, ctx), noise, sd.scheduler, None, 0, hint)
    return eng


def time_calls(eng, reps):
    """median device time (us) of every launch of one eager sampler step"""
    calls = eng.calls
    st = torch.cuda.current_stream()
    times = [[] for _ in calls]
    for rep in range(reps + 1):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(calls) + 1)]
        evs[0].record(st)
        for i, c in enumerate(calls):
            c(st.cuda_stream)
            evs[i + 1].record(st)
        torch.cuda.synchronize()
        eng.step_ptr.zero_()
        if rep:
            for i in range(len(calls)):
                times[i].append(evs[i].elapsed_time(evs[i + 1]) * 1e3)
    return [statistics.median(t) for t in times]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--micro", required=True, help="stage-1 dump (tools/tune_conv.py --dump)")
    ap.add_argument("--base", default=os.path.join(ROOT, "minsdtf_amd", "conv_tuning.json"))
    ap.add_argument("--out", default=os.path.join(ROOT, "minsdtf_amd", "conv_tuning.json"))
    ap.add_argument("--top", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dump", default=None)
    ap.add_argument("--pipelines", default=None, help="comma list of indices into PIPELINES (default all)")
    args = ap.parse_args()
    import tune_conv
    from minsdtf_amd import _lib, engine, tuning
    from minsdtf_amd.stable_diffusion import StableDiffusion

    micro = json.load(open(args.micro))    # key -> [[us, bm, bn, sk, stg], ...] sorted
    base = json.load(open(args.base))
    _lib.load().msd_init()
    dev = torch.device("cuda:0")
    pipes = [PIPELINES[int(i)] for i in args.pipelines.split(",")] if args.pipelines else list(PIPELINES)
    hook = Hook(tuning, engine)
    t0 = time.time()
    # in-place time per key and configuration, summed over the layers (and pipelines) that use the key
    insitu = collections.defaultdict(lambda: collections.defaultdict(float))
    seen_in = collections.defaultdict(set)   # key -> pipelines it occurs in
    try:
        for pi, (B, h, cfg, control) in enumerate(pipes):
            sd = StableDiffusion(8 * h, 8 * h, jit_compile=False, device=dev)
            sd.diffusion_model.load_synthetic(seed=0)
            if control:
                sd.control_net.load_synthetic(seed=1, bias_scale=0.05)
                sd.hint_net.load_synthetic(seed=2, bias_scale=0.05)
            # pass 0: the installed table; also tells which keys this step uses
            hook.override, hook.key_of = {}, {}
            eng = build_engine(sd, B, cfg, control)
            names = [c.name for c in eng.calls if isinstance(c.keep, _lib.MsdConvGemm)]
            keys = sorted({hook.key_of[n] for n in names})
            base_us = time_calls(eng, args.reps)
            print(f"pipeline {pi} (batch {B}, latent {h}, cfg {cfg}, controlnet {control}): {len(names)} conv/dense launches, {len(keys)} keys, "
                  f"step {sum(base_us):.0f} us with the installed table", flush=True)
            # candidates per key: union over the family's batches of the stage-1 top-N, plus the installed entries
            cands = {}
            for key in keys:
                fam = key.split("x", 1)[1]
                cs = []
                for k2, res in micro.items():
                    if k2.split("x", 1)[1] != fam:
                        continue
                    for r in res[:args.top]:
                        c = (int(r[1]), int(r[2]), int(r[3]), int(r[4]))
                        if c not in cs:
                            cs.append(c)
                    ent = base.get(k2)
                    if ent is not None:
                        c = (int(ent[0]), int(ent[1]), int(ent[2]), int(ent[3]) if len(ent) > 4 else 0)
                        if c not in cs:
                            cs.append(c)
                if not cs:
                    # a layer stage 1 has not seen: every plain tile, and the split counts stage 1 would have tried
                    s = parse_key(key)
                    for (bm, bn, stg) in tuning.TILES:
                        if (bn == 128 and s[4] <= 64) or (bn == 80 and (s[4] % 80 or not s[8])):
                            continue
                        nkt = s[5] * s[5] * s[3] // 64 + s[9] // 64
                        for sk in (1, 2, 4) if (s[8] and nkt >= 16 and (s[1] * s[2]) // (s[6] * s[6]) <= 1024) else (1,):
                            cs.append((bm, bn, sk, stg))
                    print(f"  (no stage-1 results for {key}: {len(cs)} generic candidates)")
                cands[key] = cs
            npass = max(len(c) for c in cands.values())
            for k in range(npass):
                hook.override = {key: cs[k] for key, cs in cands.items() if k < len(cs)}
                del eng
                eng = build_engine(sd, B, cfg, control)
                us = time_calls(eng, args.reps)
                acc = collections.defaultdict(float)
                for c, t in zip(eng.calls, us):
                    if isinstance(c.keep, _lib.MsdConvGemm):
                        key = hook.key_of[c.name]
                        if key in hook.override:
                            acc[key] += t
                for key, t in acc.items():
                    insitu[key][hook.override[key]] += t
                    seen_in[key].add(pi)
                print(f"  pass {k + 1}/{npass}: step {sum(us):.0f} us  ({time.time() - t0:.0f}s)", flush=True)
            del eng, sd
            torch.cuda.empty_cache()
    finally:
        hook.close()

    # one numerics class per family from the in-place times (same rule as stage 1)
    shapes = [parse_key(k) for k in insitu]
    key_of_shape = {parse_key(k): k for k in insitu}
    all_results = {s: sorted((us, c[0], c[1], c[2], c[3]) for c, us in insitu[key_of_shape[s]].items()) for s in shapes}
    chosen, price = tune_conv.pin_classes(shapes, all_results)
    table = dict(base)
    for s in shapes:
        key = key_of_shape[s]
        us, bm, bn, sk, stg = chosen[s]
        old = base.get(key)
        old_c = None if old is None else (int(old[0]), int(old[1]), int(old[2]), int(old[3]) if len(old) > 4 else 0)
        old_us = insitu[key].get(old_c) if old_c else None
        table[key] = [bm, bn, sk, stg, round(us, 1)]
        tag = "" if old_c == (bm, bn, sk, stg) else f"   (was {old_c}: {old_us if old_us is None else round(old_us, 1)} us)"
        print(f"{key:44s} -> {bm}x{bn}s{stg} splitk {sk:2d}  {us:8.1f} us in place{tag}", flush=True)
    # a family's other batches (not walked here) must stay in the family's class: re-pick them from stage 1 inside the class
    fam_class = {}
    for s in shapes:
        us, bm, bn, sk, stg = chosen[s]
        fam_class[key_of_shape[s].split("x", 1)[1]] = tune_conv.numerics_class(s, bm, bn, sk, stg)
    for key, res in micro.items():
        fam = key.split("x", 1)[1]
        if key in insitu or fam not in fam_class:
            continue
        s = parse_key(key)
        inclass = [r for r in res if tune_conv.numerics_class(s, int(r[1]), int(r[2]), int(r[3]), int(r[4])) == fam_class[fam]]
        if not inclass:
            raise SystemExit(f"{key}: stage 1 has no result in the family's class {fam_class[fam]}")
        r = inclass[0]
        table[key] = [int(r[1]), int(r[2]), int(r[3]), int(r[4]), round(r[0], 1)]
        print(f"{key:44s} -> {table[key][:4]} (stage 1, inside the family's class)")
    for b, (free, pinned) in sorted(price.items()):
        print(f"batch {b}: in-place layer time {free:9.1f} us free, {pinned:9.1f} us with pinned numerics classes ({pinned / free - 1:+.1%})")
    if args.dump:
        with open(args.dump, "w") as f:
            json.dump({k: [[us] + list(c) for c, us in sorted(v.items(), key=lambda kv: kv[1])] for k, v in insitu.items()}, f)
    with open(args.out, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    print(f"wrote {args.out} ({len(table)} entries) in {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()

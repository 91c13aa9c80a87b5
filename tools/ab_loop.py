"""Same-box, same-process A/B of library switches on the BASELINE denoise loop (512x512, 25 steps, batch B, hipGraph).

    python tools/ab_loop.py --variant base --variant gn_impl=0 [--rounds 4] [--batch 1]

Each variant is a comma-separated list of msd_set_option KEY=INT pairs ("base" = no switch).  The UNet is loaded once;
`py:NAME=INT` sets the module switch minsdtf_amd.engine.NAME instead (e.g. py:XATTN_FUSED_D160=0);
for every round every variant gets a fresh DenoiseEngine (switches are read when a launch is recorded, so the loop graph
is re-captured), and the rounds are interleaved (cdna_hip_programming.md rule 24).  Prints ms per 25-step loop: median,
min, and every round.
"""
import argparse
import json
import os
import statistics
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", action="append", default=[])
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--reps", type=int, default=3, help="timed loops per variant and round")
    args = ap.parse_args()
    from minsdtf_amd import _lib
    from minsdtf_amd.stable_diffusion import StableDiffusion

    lib = _lib.load()
    dev = torch.device("cuda:0")
    sd = StableDiffusion(args.size, args.size, jit_compile=True, device=dev)
    sd.diffusion_model.load_synthetic(seed=0)
    B, h = args.batch, args.size // 8
    rng = np.random.default_rng(1234)
    ctx = rng.standard_normal((B, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((B, 77, 768)).astype(np.float32)
    noise = np.random.default_rng(0).standard_normal((B, h, h, 4)).astype(np.float32)
    sd.scheduler.set_timesteps(args.steps)
    variants = args.variant or ["base"]
    keys = sorted({kv.split("=")[0] for v in variants if v != "base" for kv in v.split(",") if not kv.startswith(("tune:", "py:"))})
    from minsdtf_amd import engine as _engine

    py_defaults = {}   # py:NAME=INT sets minsdtf_amd.engine.NAME (the emitters' module switches) for that variant
    from minsdtf_amd import tuning

    tuning._load()
    pristine = dict(tuning._table)
    defaults = {}   # restore between variants: every key any variant touches gets its default back first
    times = {v: [] for v in variants}
    for rnd in range(args.rounds):
        for v in variants:
            for k in keys:
                lib.msd_set_option(k.encode(), defaults.get(k, DEFAULTS.get(k, 0)))
            tuning._table = dict(pristine)
            for k, val in py_defaults.items():
                setattr(_engine, k, val)
            if v != "base":
                for kv in v.split(","):
                    if kv.startswith("tune:"):   # tune:FILE = JSON {shape key: [tile_m, tile_n, splitk, stages]} laid over the tuning table
                        with open(kv[5:]) as f:
                            tuning._table.update({k: list(e) + [0.0] for k, e in json.load(f).items()})
                        continue
                    if kv.startswith("py:"):
                        k, val = kv[3:].split("=")
                        py_defaults.setdefault(k, getattr(_engine, k))
                        setattr(_engine, k, type(py_defaults[k])(int(val)))
                        continue
                    k, val = kv.split("=")
                    _lib.check(lib.msd_set_option(k.encode(), int(val)), kv)
            sd._engines = {}
            eng = sd._engine(B, 77, 77, args.steps, 7.5, 0.7, False)
            eng.prepare(eng.contexts(unc, ctx), noise, sd.scheduler, None, 0, None)
            eng.run_steps(args.steps, None)          # capture + warm
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                eng.step_ptr.zero_()
                eng.run_steps(args.steps, None)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / args.reps)
    for v in variants:
        t = times[v]
        print(f"{v:40s} median {statistics.median(t):8.3f} ms  min {min(t):8.3f} ms   rounds {[round(x, 3) for x in t]}")


DEFAULTS = {"gn_rows_q": -1, "attn_qf4_min": 512, "gn_xmap": 1, "gn_rows": 9216, "attn_qf": 0, "conv_dense": 1, "gn_wide": 1, "gn_impl": 1, "xattn_nw": 0, "gn_cluster": 256, "attn_form": 2, "attn_d160_pipe": 1}

if __name__ == "__main__":
    main()

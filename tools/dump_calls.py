"""Per-call timing of one eager denoise step (512x512, batch 1): `python tools/dump_calls.py OUT.txt [reps]`.
Each launch is bracketed by HIP events (includes ~1.5 us of event gap per call); the median over `reps` passes is written."""
import os
import statistics
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 7
    from minsdtf_amd import _lib
    from minsdtf_amd.stable_diffusion import StableDiffusion

    dev = torch.device("cuda:0")
    sd = StableDiffusion(512, 512, jit_compile=False, device=dev)
    sd.diffusion_model.load_synthetic(seed=0)
    rng = np.random.default_rng(1234)
    ctx = rng.standard_normal((1, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((1, 77, 768)).astype(np.float32)
    noise = np.random.default_rng(0).standard_normal((1, 64, 64, 4)).astype(np.float32)
    sd.scheduler.set_timesteps(25)
    eng = sd._engine(1, 77, 77, 25, 7.5, 0.7, False)
    eng.prepare(eng.contexts(unc, ctx), noise, sd.scheduler, None, 0, None)
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
    with open(out, "w") as f:
        tot = 0.0
        for i, c in enumerate(calls):
            us = statistics.median(times[i])
            tot += us
            line = f"{i:4d} {c.name:62s} {us:9.1f} us"
            s = c.keep
            if isinstance(s, _lib.MsdConvGemm):
                line += f"  tile {s.tile_m}x{s.tile_n} s{s.stages} k{s.splitk}"
            f.write(line + "\n")
        f.write(f"total {tot:.1f} us\n")


if __name__ == "__main__":
    main()

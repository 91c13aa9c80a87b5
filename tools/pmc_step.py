"""Two eager (no hipGraph) denoise steps at the BASELINE shape, for counter collection:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_step.py [--batch B] [--size S] [--controlnet]

(PMC collection serialises dispatches; under a replayed 8,700-node graph it does not finish in
reasonable time, so the counters are taken on the same launch list run eagerly.)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import argparse

    from minsdtf_amd.stable_diffusion import StableDiffusion

    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--controlnet", action="store_true")
    args = ap.parse_args()
    size, steps, B = args.size, 25, args.batch
    dev = torch.device("cuda:0")
    sd = StableDiffusion(size, size, jit_compile=False, device=dev)
    sd.diffusion_model.load_synthetic(seed=0)
    hint = None
    if args.controlnet:
        sd.control_net.load_synthetic(seed=0, bias_scale=0.05)
        sd.hint_net.load_synthetic(seed=0, bias_scale=0.05)
        hint = np.random.default_rng(7).integers(0, 256, (B, size, size, 3)).astype(np.float32) / 255.0
    rng = np.random.default_rng(1234)
    ctx = rng.standard_normal((B, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((B, 77, 768)).astype(np.float32)
    noise = np.random.default_rng(0).standard_normal((B, size // 8, size // 8, 4)).astype(np.float32)
    sd.scheduler.set_timesteps(steps)
    eng = sd._engine(B, 77, 77, steps, 7.5, 0.7, args.controlnet)
    eng.prepare(eng.contexts(unc, ctx), noise, sd.scheduler, None, 0, hint)
    eng.run_steps(2, None)
    torch.cuda.synchronize()
    print("done", float(eng.latent.abs().mean()))


if __name__ == "__main__":
    main()

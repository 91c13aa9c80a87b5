"""Two eager (no hipGraph) denoise steps at the BASELINE shape, for counter collection:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_step.py

(PMC collection serialises dispatches; under a replayed 8,700-node graph it does not finish in
reasonable time, so the counters are taken on the same launch list run eagerly.)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from minsdtf_amd.stable_diffusion import StableDiffusion

    size, steps = 512, 25
    dev = torch.device("cuda:0")
    sd = StableDiffusion(size, size, jit_compile=False, device=dev)
    sd.diffusion_model.load_synthetic(seed=0)
    rng = np.random.default_rng(1234)
    ctx = rng.standard_normal((1, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((1, 77, 768)).astype(np.float32)
    noise = np.random.default_rng(0).standard_normal((1, size // 8, size // 8, 4)).astype(np.float32)
    sd.scheduler.set_timesteps(steps)
    eng = sd._engine(1, 77, 77, steps, 7.5, 0.7, False)
    eng.prepare(eng.contexts(unc, ctx), noise, sd.scheduler, None, 0, None)
    eng.run_steps(2, None)
    torch.cuda.synchronize()
    print("done", float(eng.latent.abs().mean()))


if __name__ == "__main__":
    main()

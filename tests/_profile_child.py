"""Child of tests/test_baseline_configs_gpu.py::test_throughput_profile_keeps_batch_independence_and_parity: run under
MSD_PROFILE=throughput (read when minsdtf_amd is imported, hence a process of its own).  At the REAL layer shapes: a 2-step job of three
images, the first and last sample again alone (bit-identical: the overlay moves whole layers, so a sample's bits still do not depend on its
batch), and the full 25-step latent of one image against the committed fp32-oracle latent (the overlay's order of sums is another one, the
parity bar is the same).  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    assert os.environ.get("MSD_PROFILE") == "throughput"
    from minsdtf_amd import _lib, tuning
    from minsdtf_amd.models import DiffusionModel
    from minsdtf_amd.stable_diffusion import StableDiffusion
    from oracle import sd_oracle as O

    assert tuning.profile() == "throughput"
    tab = tuning._load()
    moved = [k for k, e in tab.items() if "+x" in k and int(e[0]) == 5256 and int(e[3]) >= 20 and k.startswith("2x32x32")]
    assert moved, "the overlay is not in the table"
    dev = torch.device("cuda:0")
    unet = DiffusionModel(512, 512, device=dev)
    unet.load_synthetic(seed=0)
    sd = StableDiffusion(512, 512, jit_compile=True, device=dev)
    sd._diffusion_model = unet
    rng = np.random.default_rng(1234)
    B = 3
    ctx = rng.standard_normal((B, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((B, 77, 768)).astype(np.float32)
    noise = np.random.default_rng(0).standard_normal((B, 64, 64, 4)).astype(np.float32)
    kw = dict(num_steps=2, unconditional_guidance_scale=7.5, guidance_rescale=0.7, return_latent=True)
    both = sd.generate_image(ctx, negative_prompt=unc, batch_size=B, diffusion_noise=noise, **kw)
    same = True
    for i in (0, B - 1):
        one = sd.generate_image(ctx[i], negative_prompt=unc[i], batch_size=1, diffusion_noise=noise[i], **kw)
        same = same and bool(np.array_equal(both[i:i + 1], one))
    g = np.load(os.path.join(ROOT, "tests", "golden", "oracle_latent_512_25.npz"))   # inputs re-drawn from the recorded seeds, as tests/test_e2e_gpu.py does
    rng = np.random.default_rng(int(g["context_seed"]))
    c1 = rng.standard_normal((1, 77, 768)).astype(np.float32)
    u1 = rng.standard_normal((1, 77, 768)).astype(np.float32)
    n1 = np.random.default_rng(int(g["noise_seed"])).standard_normal((1, 64, 64, 4)).astype(np.float32)
    lat = sd.generate_image(c1[0], negative_prompt=u1[0], batch_size=1, diffusion_noise=n1[0], num_steps=int(g["steps"]),
                            unconditional_guidance_scale=float(g["guidance"]), guidance_rescale=float(g["guidance_rescale"]), return_latent=True)
    psnr = float(O.psnr(lat, g["latent"]))
    opt = _lib.load()   # (the profile also selected the row-major GroupNorm for the 64x64 level: nothing to assert here but that the run was clean)
    print(json.dumps({"batch_independent": same, "psnr_db": psnr, "moved_entries": len(moved), "finite": bool(np.isfinite(lat).all())}))


if __name__ == "__main__":
    main()

"""Parity cost of the pack-time folds (engine.LN_FOLD / FF_PROJ_FOLD / MFMA_CONV_OUT): final-latent PSNR against the fp32
oracle at 64x64, 2 and 6 steps, with each fold on its own and all together (GPU + oracle on the host)."""
import sys, itertools, numpy as np, torch
sys.path.insert(0, '/root/repo')
from minsdtf_amd import engine, weights as Wt
from minsdtf_amd.stable_diffusion import StableDiffusion
from oracle import sd_oracle as O
dev = torch.device('cuda:0')
rng = np.random.default_rng(3)
ctx = rng.standard_normal((1, 77, 768)).astype(np.float32); unc = rng.standard_normal((1, 77, 768)).astype(np.float32)
noise = rng.standard_normal((1, 8, 8, 4)).astype(np.float32)
ref = None
for ln, ff, co in [(1,1,1),(0,0,0),(1,0,0),(0,1,0),(0,0,1)]:
    engine.LN_FOLD, engine.FF_PROJ_FOLD, engine.MFMA_CONV_OUT = bool(ln), bool(ff), bool(co)
    sd = StableDiffusion(64, 64, jit_compile=True, device=dev)
    arrays = sd.diffusion_model.load_synthetic(seed=0)
    if ref is None:
        W = O.named_weights(Wt.table("civitai_model"), arrays)
        refs = {n: O.denoise_loop(lambda l, t, c, ctl: O.unet_forward(W, l, t, c), ctx, unc, noise, num_steps=n, guidance=7.5, guidance_rescale=0.7) for n in (2, 6)}
        ref = True
    sd.unconditional_context = unc[0]
    out = []
    for n in (2, 6):
        got = sd.generate_image(ctx[0], batch_size=1, num_steps=n, unconditional_guidance_scale=7.5, diffusion_noise=noise[0], guidance_rescale=0.7, return_latent=True)
        out.append(O.psnr(got, refs[n]))
    print(f"LN_FOLD={ln} FF_PROJ_FOLD={ff} MFMA_CONV_OUT={co}: PSNR 2 steps {out[0]:.1f} dB, 6 steps {out[1]:.1f} dB", flush=True)

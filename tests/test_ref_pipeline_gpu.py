"""The HIP pipeline against the REFERENCE'S OWN PIPELINE, executed end to end (G12, tests/golden/g12_ref_pipeline.npz: minSDTF's
`StableDiffusion.generate_image` run over tools/keras_shim.py in the build container, see tests/test_ref_pipeline_cpu.py).  The same
seven jobs through minsdtf_amd's `StableDiffusion.generate_image` - the same keyword arguments, the drop-in API - on the same seeded
weights: unconditional context from the device CLIP models, fused device loop, ControlNet, image-to-image, inpainting, the TCD sampler, decode and
the fused uint8 cast.  Bar: >= 40 dB PSNR on the final latent (R = max - min of the reference's latent) and on the picture (R = 255)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PSNR_MIN = 40.0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g12(gpu):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ref_pipeline_goldens as g
    from minsdtf_amd.models import ControlNet, DiffusionModel, HintNet, ImageDecoder, ImageEncoder, TextClipEmbedding, TextEncoder
    from minsdtf_amd.stable_diffusion import StableDiffusion

    nets = dict(_diffusion_model=DiffusionModel(g.IMG, g.IMG, device=gpu), _image_decoder=ImageDecoder(device=gpu),
                _image_encoder=ImageEncoder(device=gpu), _text_clip_embedding=TextClipEmbedding(device=gpu), _text_encoder=TextEncoder(clip_skip=-1, device=gpu))
    cnets = dict(_control_net=ControlNet(g.IMG, g.IMG, device=gpu), _hint_net=HintNet(g.IMG, g.IMG, device=gpu))
    for m in list(nets.values()) + list(cnets.values()):
        m.load_synthetic(seed=g.SEED, bias_scale=g.BIAS_SCALE)
    plain = StableDiffusion(g.IMG, g.IMG, jit_compile=True, device=gpu)
    with_cn = StableDiffusion(g.IMG, g.IMG, jit_compile=True, device=gpu, controlnet_path="synthetic")
    for k, m in nets.items():
        setattr(plain, k, m)
        setattr(with_cn, k, m)
    for k, m in cnets.items():
        setattr(with_cn, k, m)
    tcd = StableDiffusion(g.IMG, g.IMG, jit_compile=True, device=gpu, active_tcd=True)
    for k, m in nets.items():
        setattr(tcd, k, m)
    return g, np.load(os.path.join(ROOT, "tests", "golden", "g12_ref_pipeline.npz")), g.inputs(), plain, dict(controlnet=with_cn, tcd=tcd)


def test_unconditional_context_vs_reference_pipeline(gpu, g12):
    from oracle import sd_oracle as O

    g, gold, x, plain, _ = g12
    got = plain._get_unconditional_context()
    assert got.shape == gold["uncond_context"].shape
    p = O.psnr(got, gold["uncond_context"])
    print(f"unconditional context vs the reference's own run: {p:.1f} dB")
    assert p >= PSNR_MIN


@pytest.mark.parametrize("name", ["txt2img", "txt2img_plain_cfg", "no_cfg", "controlnet", "img2img", "inpaint", "tcd"])
def test_job_vs_reference_pipeline(gpu, g12, name):
    from oracle import sd_oracle as O

    g, gold, x, plain, special = g12
    sd = special.get(name, plain)
    ctx, kw = g.case_kwargs(name, x)
    np.random.seed(g.TCD_NP_SEED)   # (the TCD sampler's per-step noise is numpy's global stream on every side, scheduler.py:301)
    lat = sd.generate_image(ctx, return_latent=True, **kw)
    ref = gold[name + ".latent"]
    assert lat.shape == ref.shape and np.isfinite(lat).all()
    p = O.psnr(lat, ref)
    np.random.seed(g.TCD_NP_SEED)
    img = sd.generate_image(ctx, **kw)
    rimg = gold[name + ".image"]
    assert img.dtype == np.uint8 and img.shape == rimg.shape
    pi = O.psnr(img.astype(np.int32), rimg.astype(np.int32), data_range=255.0)
    print(f"{name}: final latent {p:.1f} dB, uint8 picture {pi:.1f} dB (R = 255) vs the reference's own run")
    assert p >= PSNR_MIN and pi >= PSNR_MIN

"""The HIP path against the REFERENCE'S OWN network code (G11, tests/golden/g11_ref_graph.npz: minSDTF's classes executed over
tools/keras_shim.py in the build container on seeded synthetic checkpoints, see tests/test_ref_graph_cpu.py).  Every other GPU
test compares with oracle/sd_oracle.py; these compare the library's networks - through the duck-typed model classes, i.e.
through the C ABI - directly with what the reference's diffusion_model.py / control_net.py / image_decoder.py / image_encoder.py
/ text_encoder.py computed for the same weights and inputs.  Bar: >= 40 dB PSNR (R = max - min of the reference tensor)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PSNR_MIN = 40.0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g11():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ref_graph_goldens as g

    return g, np.load(os.path.join(ROOT, "tests", "golden", "g11_ref_graph.npz")), g.inputs()


def _psnr(got, ref):
    from oracle import sd_oracle as O

    assert got.shape == ref.shape and np.isfinite(got).all()
    return O.psnr(got, ref)


def test_unet_and_controlnet_vs_reference_graph(gpu, g11):
    from minsdtf_amd.models import ControlNet, DiffusionModel, HintNet

    g, gold, x = g11
    unet = DiffusionModel(g.IMG, g.IMG, device=gpu)
    unet.load_synthetic(seed=g.SEED, bias_scale=g.BIAS_SCALE)
    p = _psnr(unet.predict_on_batch([x["latent"], x["t_emb"], x["context"]]), gold["unet"])
    pc = _psnr(unet.predict_on_batch([x["latent"], x["t_emb"], x["context"]] + x["controls"]), gold["unet_controls"])
    print(f"UNet vs reference graph code: {p:.1f} dB; with the 13 ControlNet residuals: {pc:.1f} dB")
    assert p >= PSNR_MIN and pc >= PSNR_MIN
    hn, cn = HintNet(g.IMG, g.IMG, device=gpu), ControlNet(g.IMG, g.IMG, device=gpu)
    hn.load_synthetic(seed=g.SEED, bias_scale=g.BIAS_SCALE)
    cn.load_synthetic(seed=g.SEED, bias_scale=g.BIAS_SCALE)
    ph = _psnr(hn.predict_on_batch(x["hint_image"]), gold["hintnet"])
    outs = cn.predict_on_batch([x["latent"], x["t_emb"], x["context"], gold["hintnet"]])
    per = [_psnr(o, gold[f"controlnet.{i}"]) for i, o in enumerate(outs)]
    print(f"HintNet {ph:.1f} dB; ControlNet outputs {[round(v, 1) for v in per]} dB")
    assert ph >= PSNR_MIN and len(per) == 13 and min(per) >= PSNR_MIN


def test_vae_vs_reference_graph(gpu, g11):
    from minsdtf_amd.models import ImageDecoder, ImageEncoder

    g, gold, x = g11
    dec, enc = ImageDecoder(device=gpu), ImageEncoder(device=gpu)
    dec.load_synthetic(seed=g.SEED, bias_scale=g.BIAS_SCALE)
    enc.load_synthetic(seed=g.SEED, bias_scale=g.BIAS_SCALE)
    pd = _psnr(dec.predict_on_batch(x["vae_latent"]), gold["decoder"])
    pe = _psnr(enc.predict_on_batch(x["image"]), gold["encoder"])
    print(f"VAE decoder vs reference graph code: {pd:.1f} dB; encoder: {pe:.1f} dB")
    assert pd >= PSNR_MIN and pe >= PSNR_MIN


@pytest.mark.parametrize("clip_skip", [-1, -2])
def test_text_encoder_vs_reference_graph(gpu, g11, clip_skip):
    from minsdtf_amd.models import TextClipEmbedding, TextEncoder

    g, gold, x = g11
    emb = TextClipEmbedding(device=gpu)
    emb.load_synthetic(seed=g.SEED, bias_scale=g.BIAS_SCALE)
    pe = _psnr(emb.predict_on_batch([x["tokens"], x["positions"]]), gold["clip_embedding"])
    enc = TextEncoder(clip_skip=clip_skip, device=gpu)
    enc.load_synthetic(seed=g.SEED, bias_scale=g.BIAS_SCALE)
    pt = _psnr(enc.predict_on_batch(gold["clip_embedding"]), gold[f"text_encoder{clip_skip}"])
    print(f"CLIP embedding {pe:.1f} dB; text encoder clip_skip={clip_skip}: {pt:.1f} dB vs the reference graph code")
    assert pe >= PSNR_MIN and pt >= PSNR_MIN

"""The oracle against the REFERENCE'S OWN PIPELINE, executed end to end (G12).

tests/golden/g12_ref_pipeline.npz holds what minSDTF's own `StableDiffusion.generate_image` (reference stable_diffusion.py:317-486,
with its scheduler.py, its model properties :672-760, its pre-processing :217-302, its checkpoint loader) produced in the build
container over tools/keras_shim.py's pipeline mode for seven jobs on seeded synthetic checkpoints: text-to-image with guidance +
rescale (batch 2), plain guidance, no guidance (one UNet call per step), ControlNet (hint picture of another size), image-to-image
(strength 0.6 of 5 steps), inpainting (mask blur 3, per-step latent blend, final pixel blend) and the TCD sampler (4 stochastic steps) - the final latent handed to the
decoder, the uint8 picture, and the unconditional context from the CLIP models (tools/make_ref_pipeline_goldens.py).  G1-G10 pin the
reference's host arithmetic piece by piece and G11 its networks one by one; here the oracle's whole job (networks + denoise_loop +
decode + cast, with the product's picture / mask pre-processing in front) must land where the reference's whole job landed.
Not pinned, as in G11: the inside of the Keras primitives (restated in the shim from the Keras documentation).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "g12_ref_pipeline.npz")
TOL = 5e-5   # relative RMS of the final latent; measured 1.2e-6 .. 3.4e-6 (fp32 sums in another order, 2-5 steps deep)


def _gen():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ref_pipeline_goldens as g

    return g


@pytest.mark.timeout(1500)
def test_oracle_reproduces_the_reference_pipeline():
    g = _gen()
    gold = np.load(GOLD)
    assert (int(gold["seed"]), float(gold["bias_scale"]), int(gold["img"])) == (g.SEED, g.BIAS_SCALE, g.IMG)
    got = g.oracle_outputs()
    e = g.rel_rms(got["uncond_context"], gold["uncond_context"])
    print(f"unconditional context (CLIP embedding + text encoder of the empty prompt): relative RMS error {e:.2e}")
    assert e < TOL
    for name in g.CASES:
        lat, ref = got[name + ".latent"], gold[name + ".latent"]
        assert lat.shape == ref.shape and np.isfinite(lat).all(), name
        e = g.rel_rms(lat, ref)
        a = g.image_agreement(got[name + ".image"], gold[name + ".image"])
        print(f"{name}: oracle vs the reference's own run: final latent relative RMS error {e:.2e}, uint8 picture within one level {a:.5f}")
        assert e < TOL, (name, e)
        assert a > 0.999, (name, a)
    # the reference's scheduler hands back float64 (scheduler.py:52-55 builds its tables in float64): so does the oracle's loop
    assert str(gold["txt2img.latent_dtype"]) == "float64"


def test_the_comparison_has_teeth():
    """The jobs differ from one another by far more than the tolerance (a loop that ignored the rescale, the guidance scale,
    the ControlNet residuals, the start latent or the mask would be caught)."""
    g = _gen()
    gold = np.load(GOLD)
    one = gold["txt2img.latent"][:1]
    for other in ("txt2img_plain_cfg", "no_cfg", "controlnet", "img2img", "inpaint", "tcd"):
        assert g.rel_rms(gold[other + ".latent"], one) > 1000 * TOL, other
    assert g.rel_rms(gold["inpaint.latent"], gold["img2img.latent"]) > 1000 * TOL
    assert g.rel_rms(gold["txt2img.latent"][1:], one) > 1000 * TOL      # the two samples of the batch are different jobs
    for name in g.CASES:
        im = gold[name + ".image"]
        assert im.dtype == np.uint8 and int(im.max()) > int(im.min())

"""SURVEY.md §8f rank 2 on the GPU: the checkpoint-FILE route into the HIP models.

`StableDiffusion(unet_ckpt=, vae_ckpt=, lora_path=)` (reference stable_diffusion.py:620-643,692-725 ->
ckpt_loader.load_weights_from_file / load_weights_from_lora, ckpt_loader.py:2136-2276): a .safetensors file under the
reference's own checkpoint keys is read, mapped positionally onto the model's weight list, packed and run.  The loaders
themselves are pinned bit-exact against the reference on the CPU (tests/test_loaders_cpu.py); here the file route must
produce the same bits on the device as handing the same arrays to `set_weights`, and a LoRA file must land on the
oracle's result for the merged weights."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    from minsdtf_amd import weights as Wt

    p = str(tmp_path_factory.mktemp("ckpt") / "synthetic_sd15.safetensors")
    Wt.write_synthetic_checkpoint(p, kinds=("civitai_model", "decoder"), seed=0, bias_scale=0.05)   # UNet + VAE keys in one file
    return p


def _inputs():
    rng = np.random.default_rng(31)
    return (rng.standard_normal((77, 768)).astype(np.float32), rng.standard_normal((77, 768)).astype(np.float32),
            rng.standard_normal((8, 8, 4)).astype(np.float32))


def _run(sd, ctx, unc, noise, **kw):
    sd.unconditional_context = unc
    return sd.generate_image(ctx, batch_size=1, num_steps=2, unconditional_guidance_scale=7.5, diffusion_noise=noise,
                             guidance_rescale=0.7, **kw)


def test_checkpoint_file_equals_set_weights(gpu, ckpt):
    from minsdtf_amd.stable_diffusion import StableDiffusion

    ctx, unc, noise = _inputs()
    a = StableDiffusion(64, 64, jit_compile=True, unet_ckpt=ckpt, vae_ckpt=ckpt, device=gpu)
    lat_a = _run(a, ctx, unc, noise, return_latent=True)
    img_a = _run(a, ctx, unc, noise)
    b = StableDiffusion(64, 64, jit_compile=True, device=gpu)
    b.diffusion_model.load_synthetic(seed=0, bias_scale=0.05)
    b.image_decoder.load_synthetic(seed=0, bias_scale=0.05)
    lat_b = _run(b, ctx, unc, noise, return_latent=True)
    img_b = _run(b, ctx, unc, noise)
    assert np.isfinite(lat_a).all() and img_a.dtype == np.uint8 and img_a.shape == (1, 64, 64, 3)
    np.testing.assert_array_equal(lat_a, lat_b)   # the file route is layout only: the same bits as set_weights
    np.testing.assert_array_equal(img_a, img_b)


class _Collector:
    """The loader surface of a Keras model (name / weights / set_weights) that just keeps the arrays."""

    def __init__(self, specs):
        from minsdtf_amd.models import WeightVar

        self.name = "collector"
        self.weights = [WeightVar(s.name, s.shape) for s in specs]
        self.arrays = None

    def set_weights(self, arrays):
        self.arrays = list(arrays)


def test_lora_file_vs_oracle_on_merged_weights(gpu, ckpt, tmp_path):
    """kohya-named LoRA file (Linear, 1x1-conv and 3x3-conv deltas) -> StableDiffusion(lora_path=) on the device against
    the oracle run on W + delta; and the LoRA really changes the result."""
    from safetensors.torch import save_file

    from minsdtf_amd import weights as Wt
    from minsdtf_amd.stable_diffusion import StableDiffusion
    from oracle import sd_oracle as O

    rng = np.random.default_rng(32)
    names = Wt._lora_unet_name_map()   # kohya name -> diffusers key
    spec_of = {s.alt_key: s for s in Wt.table("civitai_model") if s.alt_key}
    picked = [n for n in names if n.endswith(("down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q", "mid_block_attentions_0_proj_in",
                                              "up_blocks_1_resnets_0_conv1", "down_blocks_1_resnets_0_conv_shortcut",
                                              "up_blocks_2_attentions_1_transformer_blocks_0_ff_net_0_proj"))]
    assert len(picked) == 5
    sd_lora = {}
    for n in picked:
        ts = spec_of[names[n]].torch_shape
        r = 4
        if len(ts) == 2:
            up, down = (ts[0], r), (r, ts[1])
        else:
            up, down = (ts[0], r, 1, 1), (r, ts[1], ts[2], ts[3])
        sd_lora[n + ".lora_up.weight"] = torch.from_numpy((rng.standard_normal(up) * 0.2).astype(np.float32))
        sd_lora[n + ".lora_down.weight"] = torch.from_numpy((rng.standard_normal(down) * 0.2).astype(np.float32))
        sd_lora[n + ".alpha"] = torch.tensor(4.0)
    lp = str(tmp_path / "lora.safetensors")
    save_file(sd_lora, lp)

    ctx, unc, noise = _inputs()
    with_lora = StableDiffusion(64, 64, jit_compile=True, unet_ckpt=ckpt, vae_ckpt=ckpt, lora_path=lp, device=gpu)
    assert with_lora.unet_lora_dict is not None and len(with_lora.unet_lora_dict) == 5
    got = _run(with_lora, ctx, unc, noise, return_latent=True)
    plain = _run(StableDiffusion(64, 64, jit_compile=True, unet_ckpt=ckpt, vae_ckpt=ckpt, device=gpu), ctx, unc, noise, return_latent=True)

    # oracle on the merged weights: the same loader, collecting the Keras-layout arrays instead of packing them
    specs = Wt.table("civitai_model")
    col = _Collector(specs)
    _te, unet_deltas = Wt.load_weights_from_lora(lp)
    Wt.load_weights_from_file(col, ckpt, "civitai_model", lora_dict=unet_deltas, specs=specs)
    W = O.named_weights(specs, col.arrays)
    ref = O.denoise_loop(lambda l, t, c, ctl: O.unet_forward(W, l, t, c), ctx[None], unc[None], noise[None], num_steps=2, guidance=7.5,
                         guidance_rescale=0.7)
    p, p_plain = O.psnr(got, ref), O.psnr(plain, ref)
    print(f"LoRA file route: final-latent PSNR {p:.1f} dB vs the oracle on merged weights (without the LoRA: {p_plain:.1f} dB)")
    assert p >= 40.0 and p_plain < p - 6.0


def test_new_weights_retire_the_cached_engine(gpu):
    """set_weights / load_synthetic on a model AFTER an image was generated: the cached DenoiseEngine (plans + captured
    hipGraphs hold raw addresses of the packed weights) must not be reused — the second result equals a fresh pipeline's
    with the new weights, bit for bit, and differs from the first."""
    from minsdtf_amd.stable_diffusion import StableDiffusion

    ctx, unc, noise = _inputs()
    a = StableDiffusion(64, 64, jit_compile=True, device=gpu)
    a.diffusion_model.load_synthetic(seed=0, bias_scale=0.05)
    first = _run(a, ctx, unc, noise, return_latent=True)
    a.diffusion_model.load_synthetic(seed=1, bias_scale=0.05)      # frees the old packed weights
    junk = [torch.full((1 << 22,), float("nan"), device=gpu) for _ in range(16)]   # recycle the freed memory with NaNs
    second = _run(a, ctx, unc, noise, return_latent=True)
    del junk
    b = StableDiffusion(64, 64, jit_compile=True, device=gpu)
    b.diffusion_model.load_synthetic(seed=1, bias_scale=0.05)
    fresh = _run(b, ctx, unc, noise, return_latent=True)
    assert np.isfinite(second).all()
    np.testing.assert_array_equal(second, fresh)
    assert not np.array_equal(first, second)

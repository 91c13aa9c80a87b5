"""BASELINE.json configurations beyond C2, at sizes the oracle finishes in seconds:
C5 ControlNet inside the fused device loop (HintNet once, ControlNet + UNet per step, residual adds),
C4 shape coverage (latent side not a power of two: 96x96 -> here 24x24 = 192x192 image),
C3 batch > 1 through the fused loop (per-sample independence: a batch equals its samples run alone)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
PSNR_MIN = 40.0


@pytest.fixture(scope="module")
def nets(gpu):
    from minsdtf_amd import weights as Wt
    from minsdtf_amd.models import ControlNet, DiffusionModel, HintNet
    from oracle import sd_oracle as O

    out = {}
    u = DiffusionModel(64, 64, device=gpu)
    out["unet"], out["Wu"] = u, O.named_weights(Wt.table("civitai_model"), u.load_synthetic(seed=0, bias_scale=0.05))
    c = ControlNet(64, 64, device=gpu)
    out["cn"], out["Wc"] = c, O.named_weights(Wt.table("controlnet"), c.load_synthetic(seed=0, bias_scale=0.05))
    h = HintNet(64, 64, device=gpu)
    out["hn"], out["Wh"] = h, O.named_weights(Wt.table("hintnet"), h.load_synthetic(seed=0, bias_scale=0.05))
    return out


def test_controlnet_fused_loop_vs_oracle(gpu, nets):
    """C5: stable_diffusion.py:427-452 — hint once per image, then per step control_net -> 13 residuals
    -> diffusion_model, uncond before cond; here all of it on the device, 3 steps, against the oracle."""
    from minsdtf_amd.stable_diffusion import StableDiffusion
    from oracle import sd_oracle as O

    sd = StableDiffusion(64, 64, jit_compile=True, device=gpu, controlnet_path="synthetic")
    sd._diffusion_model, sd._control_net, sd._hint_net = nets["unet"], nets["cn"], nets["hn"]
    rng = np.random.default_rng(21)
    ctx = rng.standard_normal((1, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((1, 77, 768)).astype(np.float32)
    noise = rng.standard_normal((1, 8, 8, 4)).astype(np.float32)
    image = rng.integers(0, 256, (64, 64, 3)).astype(np.float32)
    hint = O.hintnet_forward(nets["Wh"], image[None] / 255.0)
    ref = O.denoise_loop(lambda l, t, c, ctl: O.unet_forward(nets["Wu"], l, t, c, controls=ctl), ctx, unc, noise, num_steps=3,
                         guidance=7.5, guidance_rescale=0.7,
                         controlnet_fn=lambda l, t, c, h: O.controlnet_forward(nets["Wc"], l, t, c, h), hint=hint)
    sd.unconditional_context = unc[0]
    got = sd.generate_image(ctx[0], batch_size=1, num_steps=3, unconditional_guidance_scale=7.5, diffusion_noise=noise[0],
                            guidance_rescale=0.7, control_net_image=image, return_latent=True)
    p = O.psnr(got, ref)
    print(f"ControlNet fused loop: final-latent PSNR {p:.1f} dB")
    assert p >= PSNR_MIN
    host = sd.generate_image(ctx[0], batch_size=1, num_steps=3, unconditional_guidance_scale=7.5, diffusion_noise=noise[0],
                             guidance_rescale=0.7, control_net_image=image, return_latent=True, host_loop=True)
    assert O.psnr(host, ref) >= PSNR_MIN
    # the ControlNet encoder ran on a side stream beside the UNet's down path (one fork / join per step inside the captured
    # loop): in line on one stream it gives the same bits, with and without the graph
    from minsdtf_amd import stable_diffusion as SDm

    assert sd._engine(1, 77, 77, 3, 7.5, 0.7, True).cn_plan is not None
    try:
        SDm.CONTROLNET_OVERLAP = False
        sd._engines = {}
        inline = sd.generate_image(ctx[0], batch_size=1, num_steps=3, unconditional_guidance_scale=7.5, diffusion_noise=noise[0],
                                   guidance_rescale=0.7, control_net_image=image, return_latent=True)
        assert sd._engine(1, 77, 77, 3, 7.5, 0.7, True).cn_plan is None
    finally:
        SDm.CONTROLNET_OVERLAP = True
        sd._engines = {}
    np.testing.assert_array_equal(got, inline)
    calls = []
    stepped = sd.generate_image(ctx[0], batch_size=1, num_steps=3, unconditional_guidance_scale=7.5, diffusion_noise=noise[0],
                                guidance_rescale=0.7, control_net_image=image, return_latent=True, callback=calls.append)
    assert calls == [1, 2, 3]
    np.testing.assert_array_equal(got, stepped)   # per-step graph (fork / join inside each replay) == whole-loop graph


def test_batch_equals_independent_samples(gpu, nets):
    """C3: a batch of 3 through the fused loop == each sample run alone (what batch sharding relies on)."""
    from minsdtf_amd.stable_diffusion import StableDiffusion
    from oracle import sd_oracle as O

    sd = StableDiffusion(64, 64, jit_compile=True, device=gpu)
    sd._diffusion_model = nets["unet"]
    rng = np.random.default_rng(22)
    ctx = rng.standard_normal((3, 77, 768)).astype(np.float32)
    sd.unconditional_context = rng.standard_normal((77, 768)).astype(np.float32)
    noise = rng.standard_normal((3, 8, 8, 4)).astype(np.float32)
    kw = dict(num_steps=3, unconditional_guidance_scale=7.5, guidance_rescale=0.7, return_latent=True)
    both = sd.generate_image(ctx, batch_size=3, diffusion_noise=noise, **kw)
    for i in range(3):
        one = sd.generate_image(ctx[i], batch_size=1, diffusion_noise=noise[i], **kw)
        # BIT-identical: everything that orders a layer's fp32 sums (kernel family, split-K slices, column tile of the
        # LayerNorm partials, GroupNorm chunking, the attention's reference-maximum moves) is fixed per layer, not per
        # batch (minsdtf_amd/tuning.py numerics_class) — so a sample does not depend on how a global batch is sharded
        np.testing.assert_array_equal(both[i:i + 1], one)


@pytest.mark.parametrize("jit", [True, False])
def test_two_stream_step_equals_fused_batch(gpu, nets, jit):
    """denoise_streams=2: the cond and uncond halves of a step as two batch-B forwards on two HIP
    streams (fork after the sampler step, join before the next) == the fused batch-2B forward."""
    from minsdtf_amd.stable_diffusion import StableDiffusion
    from oracle import sd_oracle as O

    rng = np.random.default_rng(24)
    ctx = rng.standard_normal((2, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((77, 768)).astype(np.float32)
    noise = rng.standard_normal((2, 8, 8, 4)).astype(np.float32)
    kw = dict(batch_size=2, num_steps=4, unconditional_guidance_scale=7.5, guidance_rescale=0.7, diffusion_noise=noise,
              return_latent=True)
    out = {}
    for streams in (1, 2):
        sd = StableDiffusion(64, 64, jit_compile=jit, device=gpu)
        sd._diffusion_model, sd.unconditional_context, sd.denoise_streams = nets["unet"], unc, streams
        out[streams] = sd.generate_image(ctx, **kw)
        assert sd._engine(2, 77, 77, 4, 7.5, 0.7, False).dual == (streams == 2)
        again = sd.generate_image(ctx, **kw)                      # replay (graph or eager): same bits
        np.testing.assert_array_equal(out[streams], again)
    np.testing.assert_array_equal(out[2], out[1])   # batch-2 forwards vs the batch-4 fused one: same bits per sample


def test_non_power_of_two_latent(gpu):
    """C4-like geometry: 192x192 image -> 24x24 latent (S = 576, 144, 36, 9 tokens per level)."""
    from minsdtf_amd import weights as Wt
    from minsdtf_amd.models import DiffusionModel
    from oracle import sd_oracle as O

    m = DiffusionModel(192, 192, device=gpu)
    W = O.named_weights(Wt.table("civitai_model"), m.load_synthetic(seed=1, bias_scale=0.05))
    rng = np.random.default_rng(23)
    lat = rng.standard_normal((1, 24, 24, 4)).astype(np.float32)
    ctx = rng.standard_normal((1, 77, 768)).astype(np.float32)
    te = O.timestep_embedding(980, 1).astype(np.float32)
    ref = O.unet_forward(W, lat, te, ctx)
    got = m.predict_on_batch([lat, te, ctx])
    assert O.psnr(got, ref) >= PSNR_MIN


@pytest.mark.parametrize("jit", [True, False])
def test_tcd_sampler_fused_loop(gpu, nets, jit):
    """StableDiffusion(active_tcd=True): TCD schedule + stochastic step (scheduler.py:136-237,286-307).  The
    device loop (coefficients + pre-drawn per-step noise inside the sampler kernel) against the oracle's
    loop and the host loop, all seeded through numpy's global generator like the reference."""
    from minsdtf_amd.stable_diffusion import StableDiffusion
    from oracle import sd_oracle as O

    sd = StableDiffusion(64, 64, jit_compile=jit, device=gpu, active_tcd=True)
    sd._diffusion_model = nets["unet"]
    rng = np.random.default_rng(25)
    ctx = rng.standard_normal((1, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((1, 77, 768)).astype(np.float32)
    noise = rng.standard_normal((2, 8, 8, 4)).astype(np.float32)
    sd.unconditional_context = unc[0]
    kw = dict(batch_size=2, num_steps=4, unconditional_guidance_scale=7.5, guidance_rescale=0.7, diffusion_noise=noise,
              return_latent=True)
    if "tcd_ref" not in nets:   # (the oracle's 16 forwards on the CPU: once for both parameters)
        np.random.seed(77)
        nets["tcd_ref"] = O.denoise_loop(lambda l, t, c, ctl: O.unet_forward(nets["Wu"], l, t, c), np.repeat(ctx, 2, 0), np.repeat(unc, 2, 0),
                                         noise, num_steps=4, guidance=7.5, guidance_rescale=0.7, active_tcd=True)
    ref = nets["tcd_ref"]
    np.random.seed(77)
    got = sd.generate_image(ctx[0], **kw)
    assert list(sd.scheduler.timesteps) == [999, 759, 499, 259]
    p = O.psnr(got, ref)
    print(f"TCD fused loop: final-latent PSNR {p:.1f} dB")
    assert p >= PSNR_MIN
    np.random.seed(77)
    host = sd.generate_image(ctx[0], host_loop=True, **kw)
    assert O.psnr(host, ref) >= PSNR_MIN
    np.random.seed(78)
    other = sd.generate_image(ctx[0], **kw)      # a different draw of the per-step noise gives a different sample
    assert O.psnr(other, ref) < 30.0


@pytest.mark.parametrize("control", [False, True])
def test_cfg_prefix_sharing_is_exact(gpu, nets, control):
    """engine.SHARE_CFG_PREFIX: the unconditional and the conditioned forward of a step get the same latent and time embedding
    (stable_diffusion.py:454-457) and differ only from the first cross-attention on (diffusion_model.py:88-95), so the fused batch
    computes conv_in / down_blocks.0.resnets.0 / the front of down_blocks.0.attentions.0 once per image and replicates three tensors.
    Because a sample's bits do not depend on its batch, the result must be THE SAME BITS as computing both halves - with and without
    the ControlNet (whose encoder shares the same prefix: conv_in + hint is identical in both halves too), batch 1 and 3."""
    from minsdtf_amd import engine
    from minsdtf_amd.stable_diffusion import StableDiffusion

    sd = StableDiffusion(64, 64, jit_compile=True, device=gpu, **(dict(controlnet_path="synthetic") if control else {}))
    sd._diffusion_model = nets["unet"]
    if control:
        sd._control_net, sd._hint_net = nets["cn"], nets["hn"]
    rng = np.random.default_rng(61)
    for B in (1, 3):
        ctx = rng.standard_normal((B, 77, 768)).astype(np.float32)
        unc = rng.standard_normal((B, 77, 768)).astype(np.float32)
        noise = rng.standard_normal((B, 8, 8, 4)).astype(np.float32)
        kw = dict(negative_prompt=unc, batch_size=B, num_steps=3, unconditional_guidance_scale=7.5, diffusion_noise=noise,
                  guidance_rescale=0.7, return_latent=True)
        if control:
            kw["control_net_image"] = rng.integers(0, 256, (64, 64, 3)).astype(np.float32)
        assert engine.SHARE_CFG_PREFIX
        eng = sd._engine(B, 77, 77, 3, 7.5, 0.7, control)
        names = [c.name for c in eng.calls]
        assert sum(n.endswith(".replicate") for n in names) >= (3 if not control else 6), "the shared prefix is not in the launch list"
        shared = sd.generate_image(ctx, **kw)
        try:
            engine.SHARE_CFG_PREFIX = False
            sd._engines = {}
            both = sd.generate_image(ctx, **kw)
            assert not any(c.name.endswith(".replicate") for c in sd._engine(B, 77, 77, 3, 7.5, 0.7, control).calls)
        finally:
            engine.SHARE_CFG_PREFIX = True
            sd._engines = {}
        assert np.isfinite(shared).all()
        np.testing.assert_array_equal(shared, both)

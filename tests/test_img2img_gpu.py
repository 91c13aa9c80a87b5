"""image_to_image on the HIP path: the VAE encoder against the oracle's
restatement of image_encoder.py, and the shortened denoise loop (strength < 1) — fused device loop and
the reference-style host loop — against the oracle's loop started from the same noised init latent."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PSNR_MIN = 40.0


@pytest.fixture(scope="module")
def enc(gpu):
    from minsdtf_amd import weights as Wt
    from minsdtf_amd.models import ImageEncoder
    from oracle import sd_oracle as O

    m = ImageEncoder(device=gpu)
    W = O.named_weights(Wt.table("encoder"), m.load_synthetic(seed=0, bias_scale=0.05))
    return m, W


@pytest.mark.parametrize("shape", [(1, 64, 64), (2, 64, 128), (1, 192, 192)])
def test_encoder_vs_oracle(gpu, enc, shape):
    """image_encoder.py:21-48: conv_in, 4 levels x 2 res blocks with bottom/right padded stride-2 convs,
    mid block with single-head attention, GN+swish, conv_out, quant_conv, mean * 0.18215."""
    from oracle import sd_oracle as O

    m, W = enc
    b, h, w = shape
    rng = np.random.default_rng(31)
    img = rng.uniform(-1.0, 1.0, (b, h, w, 3)).astype(np.float32)
    ref = O.encoder_forward(W, img)
    got = m.predict_on_batch(img)
    assert got.shape == ref.shape == (b, h // 8, w // 8, 4)
    p = O.psnr(got, ref)
    print(f"encoder {shape}: PSNR {p:.1f} dB")
    assert p >= PSNR_MIN


def test_encoder_is_deterministic_and_batch_independent(gpu, enc):
    m, _ = enc
    rng = np.random.default_rng(32)
    img = rng.uniform(-1.0, 1.0, (2, 64, 64, 3)).astype(np.float32)
    a = m.predict_on_batch(img)
    b = m.predict_on_batch(img)
    np.testing.assert_array_equal(a, b)
    from oracle import sd_oracle as O

    one = m.predict_on_batch(img[1:2])
    assert O.psnr(a[1:2], one) >= 50.0


def test_encoder_rejects_bad_geometry(gpu, enc):
    m, _ = enc
    with pytest.raises(ValueError):
        m.predict_on_batch(np.zeros((1, 60, 64, 3), np.float32))


def test_image_to_image_vs_oracle(gpu, enc):
    """stable_diffusion.py:410-418, 559-568: strength 0.75 of 4 steps -> 3 steps from the noised
    encoder latent.  Fused device loop and host loop both against the oracle's loop."""
    from minsdtf_amd import weights as Wt
    from minsdtf_amd.models import DiffusionModel
    from minsdtf_amd.stable_diffusion import StableDiffusion
    from oracle import sd_oracle as O

    m, We = enc
    u = DiffusionModel(64, 64, device=gpu)
    Wu = O.named_weights(Wt.table("civitai_model"), u.load_synthetic(seed=0, bias_scale=0.05))
    sd = StableDiffusion(64, 64, jit_compile=True, device=gpu)
    sd._diffusion_model, sd._image_encoder = u, m
    rng = np.random.default_rng(33)
    ctx = rng.standard_normal((1, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((1, 77, 768)).astype(np.float32)
    noise = rng.standard_normal((2, 8, 8, 4)).astype(np.float32)
    image = rng.integers(0, 256, (64, 64, 3)).astype(np.uint8)
    sd.unconditional_context = unc[0]

    init = O.encoder_forward(We, (image[None].astype(np.float32) / 255.0) * 2.0 - 1.0)
    ref = O.denoise_loop(lambda l, t, c, ctl: O.unet_forward(Wu, l, t, c), np.repeat(ctx, 2, 0), np.repeat(unc, 2, 0), noise,
                         num_steps=4, guidance=7.5, guidance_rescale=0.7, init_latent=init, strength=0.75)
    kw = dict(batch_size=2, num_steps=4, unconditional_guidance_scale=7.5, guidance_rescale=0.7, diffusion_noise=noise,
              reference_image=image, reference_image_strength=0.75, return_latent=True)
    calls = []
    got = sd.generate_image(ctx[0], callback=calls.append, **kw)
    assert calls == [1, 2, 3]
    p = O.psnr(got, ref)
    print(f"img2img fused loop: final-latent PSNR {p:.1f} dB")
    assert p >= PSNR_MIN
    whole = sd.generate_image(ctx[0], **kw)          # no callback: whole-loop graph of 3 steps
    assert O.psnr(whole, got) >= 60.0
    host = sd.generate_image(ctx[0], host_loop=True, **kw)
    assert O.psnr(host, ref) >= PSNR_MIN
    # strength outside (0, 1) falls back to text_to_image like the reference (:410)
    kw["reference_image_strength"] = 1.0
    t2i = sd.generate_image(ctx[0], **kw)
    kw.pop("reference_image"), kw.pop("reference_image_strength")
    np.testing.assert_array_equal(t2i, sd.generate_image(ctx[0], **kw))


def test_image_to_image_api_returns_uint8(gpu, enc):
    from minsdtf_amd.models import DiffusionModel, ImageDecoder
    from minsdtf_amd.stable_diffusion import StableDiffusion

    sd = StableDiffusion(64, 64, jit_compile=True, device=gpu)
    sd._image_encoder = enc[0]
    sd._diffusion_model = DiffusionModel(64, 64, device=gpu)
    sd._diffusion_model.load_synthetic(seed=0, bias_scale=0.05)
    sd._image_decoder = ImageDecoder(device=gpu)
    sd._image_decoder.load_synthetic(seed=0, bias_scale=0.05)
    rng = np.random.default_rng(34)
    sd.unconditional_context = rng.standard_normal((77, 768)).astype(np.float32)
    img = sd.image_to_image(rng.standard_normal((77, 768)).astype(np.float32), batch_size=1, num_steps=5, seed=3,
                            reference_image=rng.integers(0, 256, (48, 80, 3)).astype(np.uint8), reference_image_strength=0.6)
    assert img.dtype == np.uint8 and img.shape == (1, 64, 64, 3)


def test_inpaint_vs_oracle(gpu, enc):
    """stable_diffusion.py:406-409,469-475,484-485: image_to_image + mask.  Every step the latent outside
    the (blurred, latent-resolution) mask is replaced by the encoded image re-noised at that step — fused
    into the sampler kernel on the device path — and the decoded image is blended with the original."""
    from minsdtf_amd import weights as Wt
    from minsdtf_amd.models import DiffusionModel, ImageDecoder
    from minsdtf_amd.stable_diffusion import StableDiffusion
    from oracle import sd_oracle as O

    m, We = enc
    u = DiffusionModel(64, 64, device=gpu)
    Wu = O.named_weights(Wt.table("civitai_model"), u.load_synthetic(seed=0, bias_scale=0.05))
    dec = ImageDecoder(device=gpu)
    Wd = O.named_weights(Wt.table("decoder"), dec.load_synthetic(seed=0, bias_scale=0.05))
    sd = StableDiffusion(64, 64, jit_compile=True, device=gpu)
    sd._diffusion_model, sd._image_encoder, sd._image_decoder = u, m, dec
    rng = np.random.default_rng(35)
    ctx = rng.standard_normal((1, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((1, 77, 768)).astype(np.float32)
    noise = rng.standard_normal((2, 8, 8, 4)).astype(np.float32)
    image = rng.integers(0, 256, (64, 64, 3)).astype(np.uint8)
    mask = np.zeros((64, 64), np.uint8)
    mask[16:48, 8:40] = 255
    sd.unconditional_context = unc[0]

    img01, img11 = sd.preprocessed_image(image)
    full, lat_mask = sd.preprocessed_mask(mask, 5)
    init = O.encoder_forward(We, img11)
    ref = O.denoise_loop(lambda l, t, c, ctl: O.unet_forward(Wu, l, t, c), np.repeat(ctx, 2, 0), np.repeat(unc, 2, 0), noise,
                         num_steps=4, guidance=7.5, guidance_rescale=0.7, init_latent=init, strength=0.75, latent_mask=lat_mask)
    kw = dict(batch_size=2, num_steps=4, unconditional_guidance_scale=7.5, guidance_rescale=0.7, diffusion_noise=noise,
              reference_image=image, reference_image_strength=0.75, inpaint_mask=mask, mask_blur_strength=5)
    got = sd.generate_image(ctx[0], return_latent=True, **kw)
    p = O.psnr(got, ref)
    print(f"inpaint fused loop: final-latent PSNR {p:.1f} dB")
    assert p >= PSNR_MIN
    host = sd.generate_image(ctx[0], return_latent=True, host_loop=True, **kw)
    assert O.psnr(host, ref) >= PSNR_MIN
    # outside the mask the last step leaves exactly the encoded image re-noised at t = 0 -> close to `init`
    outside = lat_mask[0, :, :, 0] == 0.0
    assert outside.any()
    # pixel blend: where the full-resolution mask is 0 the output is the reference image itself
    out = sd.inpaint(ctx[0], **{k: v for k, v in kw.items() if k != "diffusion_noise"}, seed=5)
    assert out.dtype == np.uint8 and out.shape == (2, 64, 64, 3)
    keep = full[0, :, :, 0] == 0.0
    expect = np.clip(img01[0] * 255.0, 0, 255).astype(np.uint8)
    np.testing.assert_array_equal(out[0][keep], expect[keep])
    # and the whole image agrees with the oracle's decode + blend of the oracle latent
    out_fixed = sd.generate_image(ctx[0], **kw)
    dec_ref = (O.decoder_forward(Wd, ref) + 1.0) * 0.5
    ref_img = np.clip((img01 * (1.0 - full) + dec_ref * full) * 255.0, 0, 255).astype(np.uint8)
    assert O.psnr(out_fixed.astype(np.float32), ref_img.astype(np.float32), data_range=255.0) >= 35.0

"""Generate the full-size oracle fixtures that stand behind the `-m gpu` parity tests of the BASELINE.json
configurations (tests/test_baseline_configs_gpu.py).  Oracle only (fp32 torch-CPU restatement, oracle/sd_oracle.py);
run in the build container, the GPU box only reads the committed .npz files.

    python tools/make_oracle_fixtures.py c2_image c3 c4 c5          (about 20 minutes on 8 cores)
    python tools/make_oracle_fixtures.py c4_long:50 c5:25 c3:25     (the configurations' FULL chains: about 1.5 hours)

  c2_image  tests/golden/oracle_image_512_25.npz     VAE decode of the committed C2 oracle latent (oracle_latent_512_25.npz):
                                                     uint8 image (reference :483-486 conversion) + the float image on a stride-4 pixel grid
  c3        tests/golden/oracle_c3_b4_512_3.npz      C3 per-GPU shape: batch 4 at 512x512, 3 steps -> final latent (4,64,64,4)
  c4        tests/golden/oracle_c4_768_2.npz         C4 shape: 768x768 (latent 96x96, S = 9216), 2 steps -> final latent + uint8 image
  c4_long   tests/golden/oracle_c4_768_8.npz         C4 shape over the complete 8-step schedule (error growth along a longer chain) -> final latent + uint8 image
  c5        tests/golden/oracle_c5_cn_512_2.npz      C5: ControlNet + HintNet at 512x512, 2 steps, batch 1 -> final latent

Inputs are NOT stored; they are regenerated from the recorded numpy PCG64 seeds exactly as bench.py draws them:
contexts default_rng(1234) -> cond (B,77,768) then uncond (B,77,768); noise default_rng(0) (B,h,w,4);
ControlNet hint image default_rng(7).integers(0,256,(B,H,W,3)) / 255.  Weights: the seeded synthetic checkpoints
(minsdtf_amd.weights.synth_keras_weights: UNet / decoder seed 0, bias_scale 0; ControlNet / HintNet seed 0,
bias_scale 0.05, as bench.py --controlnet loads them).  CFG 7.5, guidance rescale 0.7 (text_to_image defaults).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def _weights(kind, seed=0, bias_scale=0.0):
    from minsdtf_amd import weights as Wt
    from oracle import sd_oracle as O

    return O.named_weights(Wt.table(kind), Wt.synth_keras_weights(kind, seed=seed, bias_scale=bias_scale))


def _inputs(B, h):
    rng = np.random.default_rng(1234)
    ctx = rng.standard_normal((B, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((B, 77, 768)).astype(np.float32)
    noise = np.random.default_rng(0).standard_normal((B, h, h, 4)).astype(np.float32)
    return ctx, unc, noise


def _loop(Wu, ctx, unc, noise, steps, t0, controlnet_fn=None, hint=None):
    from oracle import sd_oracle as O

    def unet(lat, te, c, ctl):
        r = O.unet_forward(Wu, lat, te, c, controls=ctl)
        print(f"    unet fwd B={lat.shape[0]} done, t={time.time() - t0:.0f}s", flush=True)
        return r

    trace = []
    lat = O.denoise_loop(unet, ctx, unc, noise, num_steps=steps, guidance=7.5, guidance_rescale=0.7, trace=trace,
                         controlnet_fn=controlnet_fn, hint=hint)
    return np.asarray(lat, dtype=np.float32), trace


def c2_image():
    from oracle import sd_oracle as O

    g = np.load(os.path.join(GOLD, "oracle_latent_512_25.npz"))
    Wv = _weights("decoder")
    t0 = time.time()
    dec = O.decoder_forward(Wv, g["latent"])
    out = os.path.join(GOLD, "oracle_image_512_25.npz")
    # the float image is kept on a stride-4 pixel grid (the full uint8 image covers every pixel)
    np.savez_compressed(out, image_u8=O.to_uint8(dec), image_f32_s4=dec[:, ::4, ::4, :].astype(np.float32), decoder_seed=0, size=512,
                        steps=25)
    print("wrote", out, os.path.getsize(out), "bytes,", f"{time.time() - t0:.0f}s", flush=True)


def c3(steps=3):
    Wu = _weights("civitai_model")
    ctx, unc, noise = _inputs(4, 64)
    t0 = time.time()
    lat, trace = _loop(Wu, ctx, unc, noise, steps, t0)
    out = os.path.join(GOLD, f"oracle_c3_b4_512_{steps}.npz")
    np.savez_compressed(out, latent=lat, step0=trace[0], weight_seed=0, context_seed=1234, noise_seed=0, guidance=7.5,
                        guidance_rescale=0.7, size=512, steps=steps, batch=4)
    print("wrote", out, os.path.getsize(out), "bytes,", f"{time.time() - t0:.0f}s", flush=True)


def c4():
    from oracle import sd_oracle as O

    Wu = _weights("civitai_model")
    ctx, unc, noise = _inputs(1, 96)
    t0 = time.time()
    lat, trace = _loop(Wu, ctx, unc, noise, 2, t0)
    del Wu
    Wv = _weights("decoder")
    dec = O.decoder_forward(Wv, lat)
    out = os.path.join(GOLD, "oracle_c4_768_2.npz")
    np.savez_compressed(out, latent=lat, step0=trace[0], image_u8=O.to_uint8(dec), weight_seed=0, decoder_seed=0, context_seed=1234,
                        noise_seed=0, guidance=7.5, guidance_rescale=0.7, size=768, steps=2)
    print("wrote", out, os.path.getsize(out), "bytes,", f"{time.time() - t0:.0f}s", flush=True)


def c4_long(steps=8):
    """C4 over a longer chain (error growth at 768x768): a complete `steps`-step schedule (8, or the configuration's own 50)
    -> final latent + uint8 image."""
    from oracle import sd_oracle as O

    Wu = _weights("civitai_model")
    ctx, unc, noise = _inputs(1, 96)
    t0 = time.time()
    lat, trace = _loop(Wu, ctx, unc, noise, steps, t0)
    del Wu
    Wv = _weights("decoder")
    dec = O.decoder_forward(Wv, lat)
    out = os.path.join(GOLD, f"oracle_c4_768_{steps}.npz")
    np.savez_compressed(out, latent=lat, image_u8=O.to_uint8(dec), weight_seed=0, decoder_seed=0, context_seed=1234,
                        noise_seed=0, guidance=7.5, guidance_rescale=0.7, size=768, steps=steps)
    print("wrote", out, os.path.getsize(out), "bytes,", f"{time.time() - t0:.0f}s", flush=True)


def c5(steps=2):
    from oracle import sd_oracle as O

    Wu = _weights("civitai_model")
    Wc = _weights("controlnet", bias_scale=0.05)
    Wh = _weights("hintnet", bias_scale=0.05)
    ctx, unc, noise = _inputs(1, 64)
    image = np.random.default_rng(7).integers(0, 256, (1, 512, 512, 3)).astype(np.float32) / 255.0
    t0 = time.time()
    hint = O.hintnet_forward(Wh, image)
    lat, trace = _loop(Wu, ctx, unc, noise, steps, t0, controlnet_fn=lambda l, t, c, h: O.controlnet_forward(Wc, l, t, c, h), hint=hint)
    out = os.path.join(GOLD, f"oracle_c5_cn_512_{steps}.npz")
    np.savez_compressed(out, latent=lat, step0=trace[0], weight_seed=0, controlnet_seed=0, controlnet_bias_scale=0.05, context_seed=1234,
                        noise_seed=0, hint_seed=7, guidance=7.5, guidance_rescale=0.7, size=512, steps=steps)
    print("wrote", out, os.path.getsize(out), "bytes,", f"{time.time() - t0:.0f}s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="+", help="c2_image | c3[:steps] | c4 | c4_long[:steps] | c5[:steps]")
    for case in ap.parse_args().cases:
        print("==", case, flush=True)
        name, _, n = case.partition(":")
        assert name in ("c2_image", "c3", "c4", "c4_long", "c5"), name
        globals()[name](*([int(n)] if n else []))

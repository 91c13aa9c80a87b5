"""Generate tests/golden/oracle_latent_<size>_<steps>.npz: the fp32 CPU oracle's final latent for
the BASELINE configuration (seeded synthetic SD1.5 weights, N(0,1) contexts and noise, CFG 7.5,
rescale 0.7).  The GPU tests / bench compare the HIP path against it at full size without having
to run ~40 TFLOP of fp32 on the GPU box's host.

    python tools/make_oracle_latent.py --size 512 --steps 25      (about 10 minutes on 8 cores)

Inputs are NOT stored: they are regenerated from the recorded numpy PCG64 seeds
(contexts: default_rng(1234) -> cond then uncond; noise: default_rng(0)), exactly as bench.py and
the tests draw them, so the fixture is just the final latent (+ every 5th step for the error curve).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from minsdtf_amd import weights as Wt
    from oracle import sd_oracle as O

    h = args.size // 8
    t0 = time.time()
    W = O.named_weights(Wt.table("civitai_model"), Wt.synth_keras_weights("civitai_model", seed=args.seed))
    print(f"weights in {time.time() - t0:.0f}s", flush=True)
    rng = np.random.default_rng(1234)
    ctx = rng.standard_normal((1, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((1, 77, 768)).astype(np.float32)
    noise = np.random.default_rng(0).standard_normal((1, h, h, 4)).astype(np.float32)
    trace = []
    t0 = time.time()

    def unet(lat, te, c, ctl):
        r = O.unet_forward(W, lat, te, c)
        print(f"  unet fwd done, t={time.time() - t0:.0f}s", flush=True)
        return r

    lat = O.denoise_loop(unet, ctx, unc, noise, num_steps=args.steps, guidance=7.5, guidance_rescale=0.7, trace=trace)
    out = args.out or os.path.join(ROOT, "tests", "golden", f"oracle_latent_{args.size}_{args.steps}.npz")
    keep = list(range(4, args.steps, 5))
    np.savez_compressed(out, latent=np.asarray(lat, dtype=np.float32), trace_steps=np.asarray(keep),
                        trace=np.stack([trace[i] for i in keep]).astype(np.float32), weight_seed=args.seed, context_seed=1234,
                        noise_seed=0, guidance=7.5, guidance_rescale=0.7, size=args.size, steps=args.steps)
    print("wrote", out, os.path.getsize(out), "bytes in", time.time() - t0, "s")


if __name__ == "__main__":
    main()

"""Per-launch timing of one launch plan (HIP events around every C-ABI call, eager).

    python tools/plan_profile.py decoder [--size 512] [--batch 1] [--out gpurun_out/decoder_calls.txt]
    python tools/plan_profile.py encoder
    python tools/plan_profile.py unet            (one fused cond+uncond denoise step)

Each line: index, call name, microseconds (median of 5 runs; includes ~1-2 us of event gap), and for
conv_gemm calls the GEMM shape and achieved TFLOP/s.  The whole-plan time with no events in between
is printed last."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("net", choices=["decoder", "encoder", "unet"])
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from minsdtf_amd import _lib
    from minsdtf_amd.models import ImageDecoder, ImageEncoder
    from minsdtf_amd.stable_diffusion import StableDiffusion

    dev = torch.device("cuda:0")
    h = args.size // 8
    if args.net == "decoder":
        m = ImageDecoder(device=dev)
        m.load_synthetic(seed=0)
        m.decode_to_uint8(torch.randn(args.batch, h, h, 4, device=dev))
        calls = next(iter(m._plans.values())).plan.calls   # (the one plan this process recorded; keys carry engine.GN_EPOCH since round 5)
    elif args.net == "encoder":
        m = ImageEncoder(device=dev)
        m.load_synthetic(seed=0)
        m.predict_on_batch(np.zeros((args.batch, args.size, args.size, 3), np.float32))
        calls = next(iter(m._plans.values())).plan.calls
    else:
        sd = StableDiffusion(args.size, args.size, device=dev)
        sd.diffusion_model.load_synthetic(seed=0)
        eng = sd._engine(args.batch, 77, 77, 25, 7.5, 0.7, False)
        rng = np.random.default_rng(0)
        sd.scheduler.set_timesteps(25)
        eng.prepare(eng.contexts(rng.standard_normal((args.batch, 77, 768)), rng.standard_normal((args.batch, 77, 768))),
                    rng.standard_normal((args.batch, h, h, 4)), sd.scheduler, None, 0, None)
        calls = eng.calls
    st = torch.cuda.current_stream()
    runs = []
    for rep in range(6):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(calls) + 1)]
        evs[0].record(st)
        for i, c in enumerate(calls):
            c(st.cuda_stream)
            evs[i + 1].record(st)
        torch.cuda.synchronize()
        if rep:
            runs.append([evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(len(calls))])
    us = np.median(np.array(runs), axis=0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5):
        for c in calls:
            c(st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    lines = []
    for i, c in enumerate(calls):
        line = f"{i:4d} {c.name:58s} {us[i]:9.1f} us"
        if isinstance(c.keep, _lib.MsdConvGemm):
            s = c.keep
            M, K = s.batch * s.h_out * s.w_out, s.ksize * s.ksize * (s.c0 + s.c1)
            line += (f"  M={M:7d} N={s.N:5d} K={K:6d} tile={s.tile_m}x{s.tile_n} splitk={s.splitk:2d}"
                     f" {2.0 * M * s.N * K / us[i] / 1e6:8.1f} TF/s")
        lines.append(line)
    lines.append(f"sum of per-call times {us.sum() / 1e3:.3f} ms; plan back-to-back {e0.elapsed_time(e1) / 5:.3f} ms ({len(calls)} calls)")
    text = "\n".join(lines)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()

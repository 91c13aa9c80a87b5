"""Two eager (no hipGraph) denoise steps at the BASELINE shape, for counter collection:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_step.py [--batch B] [--size S] [--controlnet]

(PMC collection serialises dispatches; under a replayed 8,700-node graph it does not finish in
reasonable time, so the counters are taken on the same launch list run eagerly.)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import argparse

    from minsdtf_amd import host
    from minsdtf_amd.stable_diffusion import StableDiffusion

    host.fit_torch_threads()   # (weight packing on the CPUs the container is granted, not on os.cpu_count() threads)
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--controlnet", action="store_true")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT", help="msd_set_option switch, e.g. --opt gn_cluster=0")
    ap.add_argument("--graph-loops", type=int, default=0,
                    help="instead of two eager steps: capture the WHOLE 25-step loop as one hipGraph (as bench.py runs it) and replay it "
                         "this many times - for `rocprofv3 --kernel-trace` (no --pmc: counter collection under a replayed graph does not "
                         "finish); the trace then holds the kernels exactly as they run inside the timed region")
    ap.add_argument("--denoise-steps", type=int, default=25)
    ap.add_argument("--calls-json", default=None,
                    help="write, in launch order, the algorithmic FLOP of every msd_conv_gemm call this process makes and the "
                         "shader clock sampled from sysfs while the steps run (tools/pmc_mfma.py joins it with the kernel trace)")
    args = ap.parse_args()
    if args.opt:
        from minsdtf_amd import _lib as _l

        for kv in args.opt:
            k, v = kv.split("=")
            _l.check(_l.load().msd_set_option(k.encode(), int(v)), kv)
    conv_log, clk = [], {"mhz": [], "stop": False}
    if args.calls_json:
        from minsdtf_amd import _lib, ops

        orig_call = ops.Call.__call__

        def logged(self, stream):   # every library call passes here: record the conv / dense ones (M, N, K, split-K)
            s = self.keep
            if isinstance(s, _lib.MsdConvGemm):
                K = s.ksize * s.ksize * (s.c0 + s.c1) + s.c2 + s.c3
                M = s.batch * s.h_out * s.w_out
                conv_log.append({"name": self.name, "M": M, "N": s.N, "K": K, "splitk": int(s.splitk), "flop": 2.0 * M * s.N * K})
            return orig_call(self, stream)

        ops.Call.__call__ = logged

        def sample_clock():   # current sclk level of every card (the line marked '*'), a few times per millisecond of work
            import glob
            import time

            files = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
            while not clk["stop"]:
                for fn in files:
                    try:
                        for ln in open(fn):
                            if "*" in ln:
                                clk["mhz"].append(float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()))
                    except (OSError, ValueError, IndexError):
                        pass
                time.sleep(0.002)
    size, steps, B = args.size, args.denoise_steps, args.batch
    dev = torch.device("cuda:0")
    sd = StableDiffusion(size, size, jit_compile=args.graph_loops > 0, device=dev)
    sd.diffusion_model.load_synthetic(seed=0)
    hint = None
    if args.controlnet:
        sd.control_net.load_synthetic(seed=0, bias_scale=0.05)
        sd.hint_net.load_synthetic(seed=0, bias_scale=0.05)
        hint = np.random.default_rng(7).integers(0, 256, (B, size, size, 3)).astype(np.float32) / 255.0
    rng = np.random.default_rng(1234)
    ctx = rng.standard_normal((B, 77, 768)).astype(np.float32)
    unc = rng.standard_normal((B, 77, 768)).astype(np.float32)
    noise = np.random.default_rng(0).standard_normal((B, size // 8, size // 8, 4)).astype(np.float32)
    sd.scheduler.set_timesteps(steps)
    eng = sd._engine(B, 77, 77, steps, 7.5, 0.7, args.controlnet)
    eng.prepare(eng.contexts(unc, ctx), noise, sd.scheduler, None, 0, hint)
    if args.calls_json:
        import threading

        th = threading.Thread(target=sample_clock, daemon=True)
        th.start()
    n_before = len(conv_log)
    if args.graph_loops > 0:
        eng.run_steps(steps, None)   # capture + first replay (in the trace: the eager capture pass is not dispatched)
        torch.cuda.synchronize()
        for _ in range(args.graph_loops):
            eng.step_ptr.zero_()
            eng.run_steps(steps, None)
        torch.cuda.synchronize()
        print(f"graph loops {args.graph_loops + 1} steps {steps}")
    else:
        eng.run_steps(2, None)
    torch.cuda.synchronize()
    if args.calls_json:
        import json

        clk["stop"] = True
        th.join(timeout=1)
        mhz = sorted(x for x in clk["mhz"] if x > 1000.0)   # (a GPU of the box that idles — or this one between launches — reads its sleep level)
        with open(args.calls_json, "w") as f:
            json.dump({"conv_calls": conv_log, "first_step_call": n_before,
                       "sclk_mhz_samples": len(mhz), "sclk_mhz_median": mhz[len(mhz) // 2] if mhz else None,
                       "sclk_mhz_min": mhz[0] if mhz else None, "sclk_mhz_max": mhz[-1] if mhz else None}, f)
    print("done", float(eng.latent.abs().mean()))


if __name__ == "__main__":
    main()

"""Does it cost a launch anything that the PREVIOUS launch was a different kernel?  Same-shape dense launches on three tile
configurations (= three code objects), timed back to back as AAAA..., BBBB..., CCCC... and interleaved ABCABC...: if the
interleaved average exceeds the mean of the homogeneous ones, the difference is what a cold instruction cache (and whatever
else a kernel switch drags along) costs per launch.  python tools/icache_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from minsdtf_amd import _lib, ops

    _lib.load().msd_init()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream()
    for (M, N, K) in ((8192, 320, 320), (512, 1280, 1280)):
        h = int(round((M // 2) ** 0.5))
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(24)]
        bias = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        cfgs = {"A 128x64 s13": (128, 64, 13), "B 64x64": (64, 64, 0), "C 64x128": (64, 128, 0), "D 128x128": (128, 128, 0)}
        calls = {k: [ops.conv_gemm(a0=x, w=w, out=out, batch=2, h_in=h, w_in=h, c0=K, N=N, ksize=1, bias=bias, residual=res, tile_m=tm,
                                   tile_n=tn, stages=sg) for w in ws] for k, (tm, tn, sg) in cfgs.items()}

        def timed(seq, reps=6):
            for c in seq:
                c(st.cuda_stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                for c in seq:
                    c(st.cuda_stream)
            e1.record(st)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / (reps * len(seq))

        homo = {k: timed(v) for k, v in calls.items()}
        inter = []
        for i in range(24):
            for k in calls:
                inter.append(calls[k][i])
        t_inter = timed(inter)
        mean = sum(homo.values()) / len(homo)
        print(f"M={M} N={N} K={K}: " + ", ".join(f"{k} {v:.2f} us" for k, v in homo.items()) +
              f"; mean {mean:.2f} us; interleaved ABCD {t_inter:.2f} us per launch  (+{t_inter - mean:.2f})", flush=True)
        # the same launch on ROTATING activation buffers (input, residual, output each from a pool bigger than the Infinity Cache),
        # as in the pipeline, where a launch's input was written by the previous kernel and has left the L2s
        nset = max(2, (600 << 20) // (3 * M * max(N, K) * 2))
        xs = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(nset)]
        rs = [torch.randn(M, N, device=dev).to(torch.bfloat16) for _ in range(nset)]
        os_ = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nset)]
        tm, tn, sg = cfgs["A 128x64 s13"] if M > 1024 else cfgs["B 64x64"]
        rot = [ops.conv_gemm(a0=xs[i], w=ws[i % len(ws)], out=os_[i], batch=2, h_in=h, w_in=h, c0=K, N=N, ksize=1, bias=bias, residual=rs[i],
                             tile_m=tm, tile_n=tn, stages=sg) for i in range(nset)]
        # producer -> consumer chain: launch i reads what launch i-1 wrote (x := previous out; needs N == K)
        chain = [ops.conv_gemm(a0=os_[(i - 1) % nset], w=ws[i % len(ws)], out=os_[i], batch=2, h_in=h, w_in=h, c0=K, N=N, ksize=1, bias=bias,
                               residual=rs[i], tile_m=tm, tile_n=tn, stages=sg) for i in range(nset)]
        print(f"      rotating activations ({nset} sets): {timed(rot, 3):.2f} us per launch; chained (input = previous output): {timed(chain, 3):.2f} us", flush=True)


if __name__ == "__main__":
    main()

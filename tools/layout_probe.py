"""[N][K] against chunk-major [K/64][N][64] weights, layer by layer, with the installed launch configurations:
`python tools/layout_probe.py` (weights rotate through > 256 MiB of copies so that they stream from HBM as in the pipeline)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def chunk_major(w):
    N, K = w.shape
    return w.view(N, K // 64, 64).permute(1, 0, 2).contiguous()


def main():
    import tune_conv
    from minsdtf_amd import _lib, ops, tuning

    _lib.load().msd_init()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream()
    shapes = [s for s in tune_conv.collect_shapes(quick=True) if s[0] == 2]
    tot = [0.0, 0.0]
    for shape in shapes:
        batch, h_in, w_in, cin, N, ks, stride, ups, allow_split, cx = shape
        pad = 1 if ks == 3 else 0
        hl, wl = (2 * h_in, 2 * w_in) if ups else (h_in, w_in)
        ho, wo = (hl + 2 * pad - ks) // stride + 1, (wl + 2 * pad - ks) // stride + 1
        M, K = batch * ho * wo, ks * ks * cin + cx
        bm, bn, sk, stg = tuning.lookup(batch, h_in, w_in, cin, N, ks, stride, ups, M, K // 64, allow_split, cx)
        x = torch.randn(batch, h_in, w_in, cin, device=dev).to(torch.bfloat16)
        xx = torch.randn(batch, ho, wo, cx, device=dev).to(torch.bfloat16) if cx else None
        ncopy = max(1, min(16, (300 << 20) // (N * K * 2)))
        ws0 = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(ncopy)]
        ws1 = [chunk_major(w) for w in ws0]
        bias = torch.randn(N, device=dev)
        outs = []
        res = []
        for layout, ws_ in ((0, ws0), (1, ws1)):
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            wsf = torch.empty(max(1, sk * M * N), device=dev, dtype=torch.float32) if sk > 1 else None
            calls = [ops.conv_gemm(a0=x, w=w, out=out, batch=batch, h_in=h_in, w_in=w_in, c0=cin, N=N, ksize=ks, stride=stride, upsample=ups,
                                   bias=bias, workspace=wsf, workspace_floats=0 if wsf is None else wsf.numel(), splitk=sk, tile_m=bm,
                                   tile_n=bn, stages=stg, a2=xx, c2=cx, w_layout=layout) for w in ws_]
            calls[0](st.cuda_stream)
            torch.cuda.synchronize()
            outs.append(out.clone())
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for i in range(16):
                    calls[i % ncopy](st.cuda_stream)
                e1.record(st)
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / 16)
            res.append(best)
        same = torch.equal(outs[0], outs[1])
        tot[0] += res[0]
        tot[1] += res[1]
        print(f"{tuning.shape_key(*shape):40s} {bm}x{bn}s{stg}k{sk:<2d} W {N * K * 2 / 1e6:6.1f} MB  rows {res[0]:7.1f} us  chunk-major {res[1]:7.1f} us "
              f"({res[1] / res[0] - 1:+.1%})  {'same bits' if same else 'DIFFERENT'}", flush=True)
    print(f"sum {tot[0]:.0f} -> {tot[1]:.0f} us ({tot[1] / tot[0] - 1:+.1%})")


if __name__ == "__main__":
    main()

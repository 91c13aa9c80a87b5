"""Vendor yardstick for the conv / dense family — OFF the product path (nothing under minsdtf_amd/ or in bench.py's timed
region imports this file): every distinct conv / dense shape of one UNet step at batch 1 (fused cond+uncond = 2 samples) and
at batch 4 (8 samples) timed in isolation, tools/tune_conv.py's way (HIP events around back-to-back launches on random bf16
data, weights rotating through > 256 MiB of copies so that weight-streaming layers read HBM), on

  (i)   this library's entry for the shape (minsdtf_amd/conv_tuning.json, i.e. what the pipeline launches; bias epilogue),
  (ii)  torch.matmul bf16  [M, K] @ [K, N]            — hipBLASLt / rocBLAS behind PyTorch; for the 3x3 shapes this is the
        GEMM of an ALREADY im2col'ed operand (no gather, no padding): the vendor library's time for the contraction alone,
  (iii) F.conv2d bf16 channels-last for the 3x3 shapes — MIOpen behind PyTorch.

    python tools/vendor_yardstick.py --out profiles/r4_vendor_yardstick.md [--json gpurun_out/yardstick.json] [--iters 20]

What the table is for: DESIGN §6 argues the family cannot leave ~17 % of the MFMA peak at batch 1; a vendor kernel that is
> 15 % faster on a shape turns that shape into a to-do with a known-achievable time.  Reference shapes:
/root/reference/stable_diffusion/layers.py:17-25, diffusion_model.py:22-51,99-153,184-279."""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def unet_shapes(nb):
    """Distinct (batch, h, w, cin, N, ksize, stride, upsample, allow_split, cx) of one UNet forward at fused batch nb, with
    the number of launches of each per step."""
    from minsdtf_amd import engine, tuning

    rec = []
    orig = tuning.lookup

    def hook(batch, h_in, w_in, cin, N, ksize, stride, upsample, M, nk, allow_split, cx=0):
        rec.append((batch, h_in, w_in, cin, N, ksize, stride, bool(upsample), bool(allow_split), cx))
        return orig(batch, h_in, w_in, cin, N, ksize, stride, upsample, M, nk, allow_split, cx)

    tuning.lookup = hook
    engine.tuning.lookup = hook

    class _AnyW(dict):
        def __contains__(self, k):
            return True

        def __missing__(self, k):
            return None

    class _T:
        ptr = 0

        def at(self, off):
            return self

    try:
        p = engine.Plan("cpu")
        e = engine.Emitter(p, _AnyW())
        ctx = engine.Act(p.alloc(nb * 77 * 768 * 2), nb, 77, 1, 768)
        kv = engine.emit_context_kv(e, ctx, engine.UNET_ATTN_LAYERS, p)
        n_kv = len(rec)
        engine.emit_unet(e, _T(), nb, nb, 64, 64, (_T(), 0, 0, engine.temb_columns(False)), kv, 77, _T(), None)
    finally:
        tuning.lookup = orig
        engine.tuning.lookup = orig
    counts = {}
    for r in rec[n_kv:]:   # (the context K/V projections run once per prompt, not per step)
        counts[r] = counts.get(r, 0) + 1
    return counts


def time_calls(fns, iters):
    """Microseconds per call, measured INSIDE a replayed hipGraph of >= 50 calls (the functions in rotation, i.e. rotating weights):
    eager timing put every small vendor entry at 17.9-18.7 us whatever its shape - torch's host dispatch, not a kernel time
    (round 4's table).  Both sides are timed this way; a function that cannot be captured falls back to eager timing and says so."""
    st = torch.cuda.current_stream()
    for f in fns:       # warm every variant (library heuristics / solver search / lazy initialisation happen here, outside the graph)
        f()
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    n = max(50, iters, len(fns))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        g = torch.cuda.CUDAGraph()
        s2 = torch.cuda.Stream()
        s2.wait_stream(st)
        with torch.cuda.stream(s2):
            with torch.cuda.graph(g, stream=s2):
                for i in range(n):
                    fns[i % len(fns)]()
        st.wait_stream(s2)
        g.replay()
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(3):
            e0.record(st)
            g.replay()
            e1.record(st)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / n)
        return best
    except Exception as e:   # pragma: no cover
        print(f"   (not capturable, timed eagerly: {str(e)[:100]})", flush=True)
        torch.cuda.synchronize()
        e0.record(st)
        for i in range(iters):
            fns[i % len(fns)]()
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters


def measure(shape, iters):
    from minsdtf_amd import ops, packing, tuning

    batch, h_in, w_in, cin, N, ks, stride, ups, allow_split, cx = shape
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream()
    pad = 1 if ks == 3 else 0
    hl, wl = (2 * h_in, 2 * w_in) if ups else (h_in, w_in)
    ho, wo = (hl + 2 * pad - ks) // stride + 1, (wl + 2 * pad - ks) // stride + 1
    M, K = batch * ho * wo, ks * ks * cin + cx
    nk = K // 64
    x = torch.randn(batch, h_in, w_in, cin, device=dev).to(torch.bfloat16)
    xx = torch.randn(batch, ho, wo, cx, device=dev).to(torch.bfloat16) if cx else None
    ncopy = max(1, min(16, (300 << 20) // (N * K * 2)))
    ws_ = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(ncopy)]
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {"M": M, "N": N, "K": K, "gflop": 2.0 * M * N * K / 1e9}

    # (i) this library, the table's entry
    bm, bn, sk, stg = tuning.lookup(batch, h_in, w_in, cin, N, ks, stride, ups, M, nk, allow_split, cx)
    wsf = torch.empty(max(1, sk * M * N), device=dev, dtype=torch.float32) if sk > 1 else None
    wl_ = 2 if 4000 <= bm < 5000 else 1
    packed = [packing.fragment_major(w) if wl_ == 2 else packing.chunk_major(w) for w in ws_]
    calls = [ops.conv_gemm(a0=x, w=w, out=out, batch=batch, h_in=h_in, w_in=w_in, c0=cin, N=N, ksize=ks, stride=stride, upsample=ups,
                           bias=bias, workspace=wsf, workspace_floats=0 if wsf is None else wsf.numel(), splitk=sk, tile_m=bm,
                           tile_n=bn, stages=stg, a2=xx, c2=cx, w_layout=wl_) for w in packed]
    res["ours_us"] = time_calls([lambda c=c: c(torch.cuda.current_stream().cuda_stream) for c in calls], iters)
    res["ours_cfg"] = f"{bm}x{bn} s{stg} k{sk}"
    del packed, calls

    # (ii) hipBLASLt through torch.matmul on the im2col'ed operand (given, not timed)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    wts = [w.t().contiguous() for w in ws_]   # [K, N]
    omm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res["matmul_kn_us"] = time_calls([lambda w=w: torch.matmul(a, w, out=omm) for w in wts], iters)
    # ... and with the weights as they are stored, [N, K] (A @ W^T): the layout a library may prefer
    res["matmul_nk_us"] = time_calls([lambda w=w: torch.matmul(a, w.t(), out=omm) for w in ws_], iters)
    # addmm with bias: what a framework would call for a Dense layer with bias
    bb = bias.to(torch.bfloat16)
    res["addmm_us"] = time_calls([lambda w=w: torch.addmm(bb, a, w, out=omm) for w in wts], iters)
    del a, wts, omm

    # (iii) MIOpen through F.conv2d, channels-last bf16, for the 3x3 shapes without a folded shortcut
    if ks == 3 and not cx:
        xin = x.permute(0, 3, 1, 2)   # NCHW view of NHWC memory = channels_last
        if ups:
            xin = F.interpolate(xin, scale_factor=2, mode="nearest").contiguous(memory_format=torch.channels_last)
        wc = [w.view(N, ks, ks, cin).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last) for w in ws_]
        bbias = bias.to(torch.bfloat16)
        try:
            res["conv2d_us"] = time_calls([lambda w=w: F.conv2d(xin, w, bbias, stride=stride, padding=pad) for w in wc], iters)
        except RuntimeError as e:   # pragma: no cover
            res["conv2d_err"] = str(e)[:80]
    return res


def fair_vendor(r):
    """The vendor kernel for the SAME problem: F.conv2d (MIOpen) for a 3x3 conv - it gathers / pads itself, as this library's kernel
    does - and the plain GEMM for a 1x1 / Dense layer.  (`matmul` on an already im2col'ed operand is the contraction alone.)"""
    if r.get("ksize") == 3 and "conv2d_us" in r:
        return r["conv2d_us"]
    return min(r[k] for k in ("matmul_kn_us", "matmul_nk_us", "addmm_us"))


def write_md(rows, out, where):
    """rows (the --json dump) -> the markdown table committed under profiles/."""
    with open(out, "w") as f:
        f.write(f"# Vendor yardstick for the conv / dense family ({where})\n\n")
        f.write("`python tools/vendor_yardstick.py` — isolated launches timed INSIDE a replayed hipGraph of >= 50 calls (HIP events around the "
                "replay; both sides; no host dispatch in the number), random bf16 data, weights rotating through > 256 MiB.  `ours` = the "
                "conv_tuning.json entry the pipeline launches (bias epilogue, split-K reduction launch included).  `matmul` = `torch.matmul` "
                "bf16 (hipBLASLt) on an ALREADY im2col'ed `[M, K]` operand, weights `[K, N]` / as stored `[N, K]`; `addmm` adds the bias; "
                "`conv2d` = `F.conv2d` bf16 channels-last (MIOpen), 3x3 shapes only.  Two ratios: **best / ours** = the fastest vendor number of "
                "the row, the im2col'ed GEMM included (the contraction alone: what a kernel could reach if its operand gather were free); "
                "**same problem / ours** = `conv2d` for the 3x3 convs, the GEMM for the 1x1 / Dense layers.  < 1: the vendor kernel is faster.  "
                "Off the product path.\n\n")
        for nb in sorted({r["fused_batch"] for r in rows}):
            sel = [r for r in rows if r["fused_batch"] == nb]
            f.write(f"## fused batch {nb} (batch {nb // 2} per GPU with CFG)\n\n")
            f.write("| shape | launches / step | M x N x K | ours us (TF/s) | config | matmul [K,N] | matmul [N,K] | addmm | conv2d | best vendor TF/s | best / ours | same problem / ours |\n")
            f.write("|---|---|---|---|---|---|---|---|---|---|---|---|\n")
            tot_o = tot_v = tot_f = 0.0
            targets = []
            for r in sel:
                vend = {k: v for k, v in r.items() if k.endswith("_us") and k != "ours_us"}
                bv = min(vend.values())
                fv = fair_vendor(r)
                tot_o += r["ours_us"] * r["launches_per_step"]
                tot_v += bv * r["launches_per_step"]
                tot_f += fv * r["launches_per_step"]
                if bv < 0.9 * r["ours_us"]:
                    targets.append((r["key"], r["M"], r["N"], r["K"], r["ours_us"], bv, min(vend, key=vend.get)))
                f.write(f"| `{r['key']}` | {r['launches_per_step']} | {r['M']} x {r['N']} x {r['K']} | {r['ours_us']:.1f} ({r['gflop'] / r['ours_us'] * 1e3:.0f}) | "
                        f"{r['ours_cfg']} | {r['matmul_kn_us']:.1f} | {r['matmul_nk_us']:.1f} | {r['addmm_us']:.1f} | "
                        f"{r.get('conv2d_us', float('nan')):.1f} | {r['gflop'] / bv * 1e3:.0f} | {bv / r['ours_us']:.2f} | {fv / r['ours_us']:.2f} |\n")
            f.write(f"\nSum over one step's launches: ours {tot_o / 1e3:.3f} ms; per-shape best vendor number {tot_v / 1e3:.3f} ms "
                    f"(ratio {tot_v / tot_o:.2f}); vendor kernels for the same problems {tot_f / 1e3:.3f} ms (ratio {tot_f / tot_o:.2f}).\n\n")
            if targets:
                f.write("Shapes where a vendor number is more than 10 % below ours (targets):\n\n| shape | M x N x K | ours us | vendor us | which |\n|---|---|---|---|---|\n")
                for (k, M, N, K, o, v, w) in targets:
                    f.write(f"| `{k}` | {M} x {N} x {K} | {o:.1f} | {v:.1f} | {w[:-3]} |\n")
                f.write("\n")
            else:
                f.write("No shape where a vendor number is more than 10 % below ours.\n\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="profiles/r5_vendor_yardstick.md")
    ap.add_argument("--json", default=None)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batches", default="2,8", help="fused batches (2 = batch 1 with CFG, 8 = batch 4)")
    ap.add_argument("--from-json", default=None, help="only re-write the markdown table from an earlier --json dump (no GPU needed)")
    ap.add_argument("--where", default="MI355X", help="device / software line for --from-json")
    args = ap.parse_args()
    if args.from_json:
        write_md(json.load(open(args.from_json)), args.out, args.where)
        return
    from minsdtf_amd import _lib, tuning

    _lib.load().msd_init()
    torch.backends.cudnn.benchmark = True   # let MIOpen search its solvers (untimed warm-up calls)
    rows = []
    t0 = time.time()
    for nb in (int(b) for b in args.batches.split(",")):
        counts = unet_shapes(nb)
        for shape, n in counts.items():
            if shape[4] < 16:   # conv_out (N = 4): not an MFMA-shaped layer
                continue
            r = measure(shape, args.iters)
            r.update(key=tuning.shape_key(*shape), fused_batch=nb, launches_per_step=n, ksize=shape[5])
            rows.append(r)
            best_v = min(v for k, v in r.items() if k.endswith("_us") and k != "ours_us")
            print(f"{r['key']:44s} x{n:2d}  ours {r['ours_us']:7.1f} ({r['ours_cfg']})  matmul {r['matmul_kn_us']:7.1f}/{r['matmul_nk_us']:7.1f} "
                  f"addmm {r['addmm_us']:7.1f} conv2d {r.get('conv2d_us', float('nan')):7.1f}  best vendor / ours {best_v / r['ours_us']:.2f}",
                  flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)
    dev_name = torch.cuda.get_device_properties(0).name
    write_md(rows, args.out, f"{dev_name}, torch {torch.__version__}")
    print(f"wrote {args.out} in {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()

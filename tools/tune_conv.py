"""Measure the fastest (tile, split-K) configuration of msd_conv_gemm for every conv / dense shape the
SD1.5 UNet / ControlNet / VAE decoder plans emit, and write minsdtf_amd/conv_tuning.json.

    python tools/tune_conv.py [--out minsdtf_amd/conv_tuning.json] [--quick] [--dump all_results.json]

NUMERICS CLASSES.  Three launch parameters change the ORDER of a layer's fp32 sums, i.e. its bits: the kernel family
(halo-tile 3x3 kernel walks K chunk-major, the general kernel tap-major), the split-K slice count (partial slabs summed in
slice order) and, for the 1x1 / dense layers, the column tile (the grouping of the LayerNorm-fold row-moment partials).
A sample's result must not depend on the batch it runs in (north star: the same images however a global batch is sharded
over GPUs), so these three are chosen ONCE per layer shape, for all batch sizes together (the class with the smallest
weighted relative time over the measured batches), and only the batch-invariant parameters (row tile, column tile of the
3x3 convs, LDS ring depth, wave layout) are tuned per batch.  tests/test_host_cpu.py checks the table for it.

Shapes are collected by walking the launch plans with a recording hook (no weights needed).  Each
candidate is timed with HIP events on random bf16 data; weight buffers are rotated through > 256 MiB
of copies so that small-M (weight streaming) layers are measured from HBM, not from the Infinity
Cache, as they run in the real pipeline where 1.7 GB of weights stream through every step."""
import argparse
import collections
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def collect_shapes(quick=False):
    from minsdtf_amd import engine, tuning
    from minsdtf_amd import weights as wtab

    rec = []
    orig = tuning.lookup

    def hook(batch, h_in, w_in, cin, N, ksize, stride, upsample, M, nk, allow_split, cx=0):
        rec.append((batch, h_in, w_in, cin, N, ksize, stride, bool(upsample), bool(allow_split), cx))
        return tuning.heuristic(M, N, nk, allow_split)

    tuning.lookup = hook
    engine.tuning.lookup = hook
    class _AnyW(dict):   # every weight "exists" (so the folded layer forms are the ones walked), none is real
        def __contains__(self, k):
            return True

        def __missing__(self, k):
            return None

    W = _AnyW()

    class _T:  # stands in for device tensors / buffers while walking the topology
        ptr = 0

        def at(self, off):
            return self

    def unet(nb, h, T=77):
        p = engine.Plan("cpu")
        e = engine.Emitter(p, W)
        ctx = engine.Act(p.alloc(nb * T * 768 * 2), nb, T, 1, 768)
        kv = engine.emit_context_kv(e, ctx, engine.UNET_ATTN_LAYERS, p)
        cols = engine.temb_columns(False)
        engine.emit_unet(e, _T(), nb, nb, h, h, (_T(), 0, 0, cols), kv, T, _T(), None)

    def controlnet(nb, h, T=77):
        p = engine.Plan("cpu")
        e = engine.Emitter(p, W)
        ctx = engine.Act(p.alloc(nb * T * 768 * 2), nb, T, 1, 768)
        kv = engine.emit_context_kv(e, ctx, engine.ENCODER_ATTN_LAYERS, p)
        cols = engine.temb_columns(True)
        outs = [p.act(nb, h >> l, h >> l, ch) for l, ch in zip((0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 3), wtab.UNET_SKIP_CH + (1280,))]
        engine.emit_controlnet(e, _T(), nb, nb, h, h, (_T(), 0, 0, cols), kv, T, p.act(nb, h, h, 320), outs)

    def vae(b, h):
        p = engine.Plan("cpu")
        e = engine.Emitter(p, W)
        engine.emit_decoder(e, _T(), b, h, h, _T(), 0)

    def vae_enc(b, h):
        p = engine.Plan("cpu")
        e = engine.Emitter(p, W)
        engine.emit_encoder(e, _T(), b, 8 * h, 8 * h, _T())

    unet(2, 64)
    vae(1, 64)
    if not quick:
        for nb in (4, 8):
            unet(nb, 64)
        unet(2, 96)
        unet(1, 64)
        vae(4, 64)
        vae(1, 96)
        controlnet(2, 64)
        vae_enc(1, 64)
    tuning.lookup = orig
    engine.tuning.lookup = orig
    uniq = []
    for r in rec:
        if r not in uniq:
            uniq.append(r)
    return uniq


def wreg_nj(bm, bn, stg):
    """16-column blocks per wave of a wreg configuration (csrc/conv_wreg.hip MSD_WREG_CFGS): 4 waves over N, or 8 (stages code
    10 + depth), except the 256-row tile whose 8 waves are a 2 x 4 grid."""
    waves_n = 4 if (stg % 20 < 10 or bm == 4256) else 8
    return bn // 16 // waves_n


def tune_one(shape, iters=10, only=None, sks_only=None):
    from minsdtf_amd import ops, tuning

    batch, h_in, w_in, cin, N, ks, stride, ups, allow_split, cx = shape
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream()
    pad = 1 if ks == 3 else 0
    hl, wl = (2 * h_in, 2 * w_in) if ups else (h_in, w_in)
    ho, wo = (hl + 2 * pad - ks) // stride + 1, (wl + 2 * pad - ks) // stride + 1
    M, K = batch * ho * wo, ks * ks * cin + cx
    xx = torch.randn(batch, ho, wo, cx, device=dev).to(torch.bfloat16) if cx else None   # shortcut operand (extra K tiles)
    nk = K // 64
    x = torch.randn(batch, h_in, w_in, cin, device=dev).to(torch.bfloat16)
    wbytes = N * K * 2
    ncopy = max(1, min(16, (300 << 20) // wbytes))
    ws_ = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(ncopy)]
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    best = None
    results = []
    cands = list(tuning.TILES)
    if ks == 3 and stride == 1 and not ups and w_in % 16 == 0:  # halo-tile kernel (tile_m = 1000 + pixels per tile; round 6: with a shortcut operand too)
        cands += [t for t in tuning.HALO_TILES if h_in % ((t[0] % 1000) // 16) == 0]
    if ks == 1 and stride == 1 and not ups and not cx and cin in tuning.ROWPANEL_ROWS and N % 32 == 0:   # row-panel Dense kernel
        cands += [(rows, cols, 0) for rows in tuning.ROWPANEL_ROWS[cin] for cols in tuning.ROWPANEL_COLS if N % cols == 0]
    if N % 16 == 0 and not os.environ.get("MSD_TUNE_NO_WREG"):   # wreg form (fragment-major weights straight to registers)
        cands += [t for t in tuning.WREG_TILES if not (t[1] > 64 and N <= 64)]
    # big form (256-row macro tiles): not for the LayerNorm-producer Dense layers (K = N, 1x1: it has no ln_out epilogue)
    if M >= tuning.BIG_MIN_ROWS and not (ks == 1 and allow_split and cin == N and not cx) and not os.environ.get("MSD_TUNE_NO_BIG"):
        cands += [t for t in tuning.BIG_TILES if not (t[1] == 160 and N % 160) and not (t[1] > 128 and N <= 128)]
        if ks == 3 and stride == 1 and not cx:   # ... walking K chunk-major: the halo-tile kernel's class
            cands += [t for t in tuning.BIG_TILES_CHUNK_MAJOR if not (t[1] == 160 and N % 160) and not (t[1] > 128 and N <= 128)]
    if M >= tuning.HALO_IMAGE_MIN_ROWS and ks == 3 and stride == 1 and not (cx and ups) and hl % 16 == 0 and wl % 16 == 0 and not os.environ.get("MSD_TUNE_NO_BIG"):
        # ... on a staged halo (whole 16 x 16-pixel output tiles)
        cands += [t for t in tuning.BIG_TILES_HALO_IMAGE if not (t[1] == 160 and N % 160) and not (t[1] > 128 and N <= 128)]
    if only is not None:
        cands = [t for t in cands if only(t)]
    frag = None
    for (bm, bn, stg) in cands:
        if 4000 <= bm < 5000 and allow_split is False and wreg_nj(bm, bn, stg) % 2:
            continue   # (the 'n' shapes include GEGLU, which pairs the two blocks of a wave)
        if bm >= 5000 and allow_split is False and bn == 160:
            continue   # (5 blocks per wave: no x | gate pairs)
        if bm < 3000:
            if bm == 256 and M < 1024:
                continue
            if bn == 128 and N <= 64:
                continue
            if bn == 80 and (N % 80 or not allow_split):   # (the 'n' shapes include GEGLU, which pairs fragments)
                continue
            if bn == 160 and (N % 160 or N < 1280):
                continue
        sks = [1]
        if allow_split and (bm < 3000 or bm >= 4000):
            # candidates from the PER-SAMPLE shape, so every batch of a layer is measured on the same set of slice counts
            bme = bm % 1000 if bm >= 1000 else bm
            tiles = ((M // batch + bme - 1) // bme) * ((N + bn - 1) // bn)
            kmax = (cin // 64) if (1000 <= bm < 3000 or (bm >= 5000 and stg >= 10)) else nk // 4   # the halo kernel (and the chunk-major big form) split over 64-channel chunks
            sks += [s for s in (2, 3, 4, 6, 8, 12, 16) if s <= kmax and tiles * s <= 2048 and tiles < 512]
        if sks_only is not None:
            sks = [s for s in sks_only if s == 1 or allow_split]
        for sk in sks:
            wsf = torch.empty(max(1, sk * M * N), device=dev, dtype=torch.float32) if sk > 1 else None
            if 4000 <= bm < 5000 and frag is None:
                from minsdtf_amd import packing

                frag = [packing.fragment_major(w) for w in ws_]
            calls = [ops.conv_gemm(a0=x, w=w, out=out, batch=batch, h_in=h_in, w_in=w_in, c0=cin, N=N, ksize=ks, stride=stride,
                                   upsample=ups, bias=bias, workspace=wsf, workspace_floats=0 if wsf is None else wsf.numel(),
                                   splitk=sk, tile_m=bm, tile_n=bn, stages=stg, a2=xx, c2=cx, w_layout=2 if 4000 <= bm < 5000 else 0)
                     for w in (frag if 4000 <= bm < 5000 else ws_)]
            for c in calls[:2]:
                c(st.cuda_stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for i in range(iters):
                calls[i % ncopy](st.cuda_stream)
            e1.record(st)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            results.append((us, bm, bn, sk, stg))
            if best is None or us < best[0]:
                best = (us, bm, bn, sk, stg)
            del wsf
    return best, sorted(results), 2.0 * M * N * K


BATCH_WEIGHT = {1: 1.0, 2: 4.0, 4: 1.5, 8: 2.0}   # fused cond+uncond batch: 2 = the headline batch-1 run, 8 = 4 images per GPU


def numerics_class(shape, bm, bn, sk, stg=0):
    """What of a configuration changes the order of the fp32 sums (see the module docstring)."""
    from minsdtf_amd import tuning

    # (a LayerNorm producer: 1x1, C -> C, splittable key, no shortcut operand - tuning.key_is_ln_producer)
    ln_prod = bool(shape[8]) and shape[5] == 1 and shape[3] == shape[4] and not shape[9]
    return tuning.numerics_class(shape[5], bm, bn, sk, ln_producer=ln_prod, stages=stg)


def pin_classes(shapes, all_results):
    """One numerics class per batch-agnostic layer shape; returns {shape: (us, bm, bn, sk, stg)} = the fastest measured
    configuration of that class for every batch, and the per-batch price of the pinning."""
    fams = collections.defaultdict(list)
    for s in shapes:
        fams[s[1:]].append(s)
    chosen, price = {}, collections.defaultdict(lambda: [0.0, 0.0])
    for fam, members in fams.items():
        best_free = {s: min(r[0] for r in all_results[s]) for s in members}
        per_class = collections.defaultdict(dict)   # class -> {shape: best result}
        for s in members:
            for r in all_results[s]:
                c = numerics_class(s, r[1], r[2], r[3], r[4])
                if s not in per_class[c] or r[0] < per_class[c][s][0]:
                    per_class[c][s] = r
        full = {c: d for c, d in per_class.items() if len(d) == len(members)}
        assert full, fam
        score = lambda d: sum(BATCH_WEIGHT.get(s[0], 1.0) * d[s][0] / best_free[s] for s in members)  # noqa: E731
        c_best = min(full, key=lambda c: score(full[c]))
        for s in members:
            chosen[s] = full[c_best][s]
            price[s[0]][0] += best_free[s]
            price[s[0]][1] += chosen[s][0]
    return chosen, price


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "minsdtf_amd",
                                                  "conv_tuning.json"))
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--dump", default=None, help="also write every measured (configuration, time) per shape to this JSON file")
    args = ap.parse_args()
    from minsdtf_amd import _lib, tuning

    shapes = collect_shapes(args.quick)
    print(f"{len(shapes)} distinct conv/dense shapes", flush=True)
    lib = _lib.load()
    lib.msd_init()
    table = {}
    t0 = time.time()
    all_results, flops = {}, {}
    for s in shapes:
        best, results, flop = tune_one(s)
        all_results[s], flops[s] = results, flop
        alt = " ".join(f"{bm}x{bn}s{stg}/k{sk}:{us:.0f}" for us, bm, bn, sk, stg in results[:4])
        print(f"{tuning.shape_key(*s):44s} free best {best[1]}x{best[2]}s{best[4]} splitk {best[3]:2d}  {best[0]:7.1f} us   [{alt}]", flush=True)
    chosen, price = pin_classes(shapes, all_results)
    for s in shapes:
        us, bm, bn, sk, stg = chosen[s]
        key = tuning.shape_key(*s)
        table[key] = [bm, bn, sk, stg, round(us, 1)]
        print(f"{key:44s} -> {bm}x{bn}s{stg} splitk {sk:2d}  {us:7.1f} us {flops[s] / us / 1e6:7.1f} TF/s", flush=True)
    for b, (free, pinned) in sorted(price.items()):
        print(f"batch {b}: sum of layer times {free:9.1f} us free, {pinned:9.1f} us with pinned numerics classes ({pinned / free - 1:+.1%})")
    if args.dump:
        with open(args.dump, "w") as f:
            json.dump({tuning.shape_key(*s): [list(r) for r in all_results[s]] for s in shapes}, f)
    with open(args.out, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    print(f"wrote {args.out} ({len(table)} entries) in {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()

"""Per-layer-shape launch configuration of msd_conv_gemm (tile size, LDS ring depth, split-K).

The kernel is bounded by the per-CU L2->LDS operand rate, so the best tile is a trade between
bytes per FLOP (bigger tiles, the halo-tile 3x3 variant), bytes in flight per CU (ring depth) and
workgroups in flight (smaller tiles / split-K) that depends on the layer shape.
`conv_tuning.json` holds the configuration measured fastest on an MI355X for every conv / dense
shape of the SD1.5 UNet, ControlNet and VAE decoder at the benchmarked batch sizes (produced by
tools/tune_conv.py); shapes that are not in the table fall back to a size heuristic.
"""
from __future__ import annotations

import json
import os
import re
from typing import Dict, Optional, Tuple

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv_tuning.json")
_table: Optional[Dict[str, list]] = None
_families: Optional[Dict[str, Dict[int, list]]] = None   # batch-agnostic key -> {batch: entry}

# (tile_m, tile_n, stages): stages 0 = the tile's default ring depth.  tile_m 1128 / 1256 = halo-tile
# 3x3 kernel with 8x16 / 16x16 pixel tiles.
TILES = ((128, 128, 0), (128, 64, 0), (64, 64, 0), (64, 128, 0), (256, 128, 0),
         (128, 128, 4), (64, 64, 8), (64, 128, 5), (128, 64, 5), (128, 80, 0), (128, 80, 4),
         (64, 64, 14), (128, 64, 13), (64, 128, 13),   # stages 10 + depth: the tile on 8 waves
         (128, 128, 23), (128, 128, 24), (128, 64, 24), (64, 128, 24),   # 20 + depth: 64x64 per wave (4 / 2 waves)
         (128, 160, 0))
HALO_TILES = ((1128, 64, 0), (1128, 128, 0), (1256, 128, 0), (1128, 80, 0), (1256, 80, 0),
              (1128, 64, 8), (1128, 128, 6), (1128, 80, 8), (1256, 80, 5), (2128, 64, 0), (2128, 80, 0),
              (1128, 64, 33), (1128, 64, 34), (1128, 80, 33), (2128, 64, 33),   # 30 + depth: 3 taps (a filter row) per K step
              (1128, 64, 63), (1128, 80, 63),   # 60 + depth: ... staged by two loader waves behind the compute waves
              (1128, 64, 93), (1128, 80, 93),   # 90 + depth: the 60s' form with the K loop rotated (every fragment read issued in front of MFMAs that do not need it)
              (1128, 64, 153), (1128, 64, 158), (1128, 128, 153), (1128, 128, 156), (1256, 128, 153), (1128, 80, 158),
              (1256, 80, 153), (1256, 80, 155))   # 150 + depth: the one-tap form rotated (tap it + 1's fragments read under tap it's MFMAs)


# (4000 + rows per workgroup, columns per workgroup, ring depth [+ 10: 8 waves] [+ 20: two K tiles per stage]): wreg form (csrc/conv_wreg.hip) — weights global ->
# VGPR from the fragment-major image, every wave owns all rows and its own 16-column blocks
WREG_TILES = ((4128, 128, 3), (4128, 128, 4), (4128, 64, 3), (4128, 64, 4), (4064, 128, 3), (4064, 128, 4), (4064, 256, 3), (4064, 256, 4),
              (4064, 64, 4), (4256, 64, 3), (4128, 128, 13),
              # stages 20 + depth: two K tiles (128 channels) per ring stage / wait / barrier
              (4064, 64, 23), (4064, 64, 24), (4064, 128, 23), (4064, 128, 24), (4064, 256, 23), (4128, 64, 23), (4128, 128, 23),
              # 256 rows on 8 waves as a 2 x 4 wave grid (128 rows x 2 blocks per wave)
              (4256, 128, 13), (4256, 128, 14))


# (5000 + rows per workgroup, columns per workgroup, configuration code): big form (csrc/conv_big.hip) for M >= 8192 — 8 waves in two
# half-workgroups one barrier apart; the tile kernel's K walk and epilogue (same bits), no LayerNorm-producer epilogue
BIG_TILES = ((5256, 256, 0), (5256, 160, 0), (5256, 160, 1), (5256, 160, 2), (5256, 128, 0), (5256, 128, 1), (5128, 256, 0), (5128, 256, 1))
# ... code + 10: the chunk-major K walk = the halo-tile kernel's order of sums (3x3, stride 1, no upsampling, no shortcut operand): its numerics class
BIG_TILES_CHUNK_MAJOR = ((5256, 256, 10), (5256, 160, 10), (5256, 160, 11), (5256, 128, 10), (5128, 256, 10))
# ... code + 20: the same walk over a staged 18 x 18-pixel halo per chunk (also pad 1, h_in and w_in multiples of 16; csrc/conv_big.hip conv_bighalo_kernel)
BIG_TILES_HALO_IMAGE = ((5256, 160, 20), (5256, 128, 20), (5256, 128, 21))
BIG_MIN_ROWS = 4096   # rows (M) below which the tuner does not try the big form
HALO_IMAGE_MIN_ROWS = 256   # ... its halo-image variant (one 16 x 16-pixel tile per sample at the 16x16 level, split-K over chunks)


def is_halo(tile_m: int) -> bool:
    return 1000 <= tile_m < 3000


def is_rowpanel(tile_m: int) -> bool:
    return 3000 <= tile_m < 4000


def is_wreg(tile_m: int) -> bool:
    """The same ranges as csrc/conv_gemm.hip cg_is_wreg / cg_is_big (tile_m >= 6000 is refused there)."""
    return 4000 <= tile_m < 5000


def is_big(tile_m: int) -> bool:
    return 5000 <= tile_m < 6000


# (rows per workgroup + 3000, columns per workgroup): row-panel Dense kernel (csrc/conv_rowpanel.hip) for the LayerNorm-
# consumer GEMMs of the transformer blocks (K = 320 / 640, 128 rows per workgroup)
ROWPANEL_ROWS = {320: (3128,), 640: (3128,)}
ROWPANEL_COLS = (64, 96, 128, 160, 192, 320, 480, 640, 960)


def shape_key(batch, h_in, w_in, cin, N, ksize, stride, upsample, allow_split=True, cx=0) -> str:
    """cx: channels of the shortcut operand (extra K tiles at the output pixel), 0 = none."""
    return (f"{batch}x{h_in}x{w_in}x{cin}->{N}k{ksize}s{stride}u{int(bool(upsample))}{'' if allow_split else 'n'}"
            f"{'+x' + str(cx) if cx else ''}")


def _load() -> Dict[str, list]:
    global _table, _families
    if _table is None:
        try:
            with open(_PATH) as f:
                _table = json.load(f)
        except (OSError, ValueError):
            _table = {}
        # A/B runs: $MSD_TUNE_OVERRIDE = JSON {shape key: [tile_m, tile_n, splitk, stages]} (or @file) laid over the table
        ov = os.environ.get("MSD_TUNE_OVERRIDE")
        if ov:
            if ov.startswith("@"):
                with open(ov[1:]) as f:
                    ov = f.read()
            _table.update({k: list(v) + [0.0] for k, v in json.loads(ov).items()})
        _families = {}
        for key, ent in _table.items():
            b, rest = key.split("x", 1)
            _families.setdefault(rest, {})[int(b)] = ent
    return _table


def heuristic(M: int, N: int, nk: int, allow_split: bool) -> Tuple[int, int, int, int]:
    """(tile_m, tile_n, splitk, stages) on the plain tile kernel from the sizes alone (the tuner's starting point and the
    tensor-less shape walks).  M = rows of ONE sample: split-K and the column tile are part of a layer's arithmetic (order of
    the fp32 sums) and must not depend on the batch."""
    bn = 128 if (N % 128 == 0 or N > 1024) else 64
    bm = 128
    if M <= 64:
        bm = 64
    tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    sk = 1
    if allow_split and tiles < 160 and nk >= 32:
        sk = max(1, min((256 + tiles - 1) // tiles, nk // 8, 16))
    return bm, bn, sk, 0


def _ceil_div(a: int, b: int) -> int:
    return -(-a // b)


def shape_class(h_in, w_in, cin, N, ksize, stride, upsample, allow_split, cx=0) -> Tuple[str, int, int]:
    """The ARITHMETIC of a layer that is not in the table, from the shape of ONE sample only: (K walk, split-K slices,
    column tile of the LayerNorm-fold partials or 0).  K walk: "chunk" = chunk-major (the halo-tile kernel's order: for every
    64-channel chunk its nine taps; split-K over chunks), "chunk_sc" = the same with the shortcut chunks behind a slice's main
    chunks (staged-halo big form), "tap" = tap-major (tile / wreg / big forms; split-K over K tiles).  What the measured table
    shows per shape class (tools/tune_conv.py + tune_insitu.py, 512x512 / 768x768, batches 1-8) written as rules: the forms this
    library's speed lives in reach any image size, not only the tuned ones."""
    hl, wl = (2 * h_in, 2 * w_in) if upsample else (h_in, w_in)
    pad = 1 if ksize == 3 else 0
    ho, wo = (hl + 2 * pad - ksize) // stride + 1, (wl + 2 * pad - ksize) // stride + 1
    hw_out = ho * wo
    nkc = cin // 64
    nk = ksize * ksize * nkc + cx // 64
    tile16 = stride == 1 and hl % 16 == 0 and wl % 16 == 0
    walk = "tap"
    if ksize == 3 and stride == 1:
        if cx == 0 and not upsample and w_in % 16 == 0 and h_in % 8 == 0 and hw_out >= 256:
            walk = "chunk"       # halo-tile kernel (8 x 16-pixel tiles), staged-halo big form on whole 16 x 16 tiles
        elif cx == 0 and upsample and tile16 and h_in * w_in >= 256:
            walk = "chunk"       # nearest x2 + conv on the staged halo (chunk-major big form where the tiles are not whole)
        elif cx and w_in % 16 == 0 and h_in % 8 == 0 and hw_out >= 256:
            walk = "chunk_sc"    # shortcut-folded conv: halo-tile kernel (one or two images per GPU) / staged-halo big form, one class (round 6)
    sk = 1
    if allow_split:
        if walk != "tap":
            wgs = _ceil_div(hw_out, 128) * _ceil_div(N, 64)      # 8 x 16-pixel tiles x 64 columns, per sample
            if wgs < 256 and hw_out < 4096:
                sk = max(1, min(_ceil_div(512, 2 * wgs), nkc // (2 if N < 64 else 6)))
        elif ksize == 3 or nk >= 32:                             # 3x3 on the tile forms, long-K Dense (ff.net.2 + proj_out, K = 5 C)
            if ksize == 3 and hw_out < 4096:
                sk = max(1, min(_ceil_div(96, max(1, int(hw_out ** 0.5))), nk // 8, 16))
            elif ksize == 1 and hw_out < 1024:
                sk = max(1, min(_ceil_div(48, max(1, int(hw_out ** 0.5))), nk // 8, 16))
        elif nk >= 16 and N <= 2048:                              # short-K Dense at the lowest levels (few column tiles)
            sk = 4 if hw_out <= 64 else 2 if hw_out <= 144 else 1
    ln_tile = 64 if (ksize == 1 and allow_split and cx == 0 and cin == N) else 0   # (the layers that may produce LayerNorm-fold partials)
    return walk, sk, ln_tile


def shape_config(batch, h_in, w_in, cin, N, ksize, stride, upsample, M, nk, allow_split, cx=0) -> Tuple[int, int, int, int]:
    """(tile_m, tile_n, splitk, stages) for a layer the table has no entry of: the arithmetic from shape_class() (per-sample
    shape: a sample's bits do not depend on its batch), the kernel FORM inside that class from the size of the whole launch
    (bit-neutral: row tile, 3x3 column tile, ring depth, wave layout, row-panel / wreg / big / staged-halo form).  Every
    configuration named here is one the library builds and the table uses somewhere."""
    walk, sk, ln_tile = shape_class(h_in, w_in, cin, N, ksize, stride, upsample, allow_split, cx)
    hl, wl = (2 * h_in, 2 * w_in) if upsample else (h_in, w_in)
    tile16 = stride == 1 and hl % 16 == 0 and wl % 16 == 0
    wide = 128 if (N % 128 == 0 or N > 1024) else 64
    if walk == "chunk_sc":
        wgs16 = batch * (hl // 16) * (wl // 16) * _ceil_div(N, 128) * sk if tile16 else 0
        if N >= 128 and wgs16 >= 128:
            return 5256, (160 if (N % 160 == 0 and M >= 32768 and allow_split) else 128), sk, 20
        if M <= 4096 or N < 80:
            return 1128, 64, sk, 0        # (in place the plain 8 x 16 loop won at one image per GPU: tools/unify_shortcut_insitu.py)
        return 1128, (80 if (N % 80 == 0 and allow_split) else 64), sk, 93
    if walk == "chunk":
        wgs16 = batch * (hl // 16) * (wl // 16) * _ceil_div(N, 128) * sk if tile16 else 0
        if upsample:
            if tile16:
                return 5256, (160 if (N % 160 == 0 and M >= 32768) else 128), sk, 20
            return 5256, 128, sk, 10      # (never reached today: "chunk" + upsample implies whole tiles)
        if N >= 128 and wgs16 >= 128:     # the staged-halo big form once it fills half the chip
            return 5256, (160 if (N % 160 == 0 and M >= 32768 and allow_split) else 128), sk, 20
        if M <= 4096 or N < 80:
            return 2128, 64, sk, (33 if M <= 16384 else 0)
        return 1128, (80 if (N % 80 == 0 and allow_split) else 64), sk, 93
    # ---- tap-major: tile / row-panel / wreg / big forms
    if ln_tile:                            # C -> C 1x1 that may write LayerNorm partials: the column tile is part of the class
        return (64, 64, sk, 0) if M <= 2048 else (128, 64, sk, 13)
    if ksize == 1 and not allow_split:     # q|k|v, GEGLU and context k|v projections (never split, never LayerNorm producers)
        if cin in ROWPANEL_ROWS and N % 32 == 0 and M >= 4096 and stride == 1 and not upsample and not cx:
            panels = _ceil_div(M, 128)
            cols = [c for c in ROWPANEL_COLS if N % c == 0]
            fit = [c for c in cols if panels * (N // c) >= 512]
            if cols:
                return ROWPANEL_ROWS[cin][0], (max(fit) if fit else min(cols)), 1, 0
        if M <= 512:
            return 64, 64, 1, 14
        if M <= 4096 or N <= 64:
            return 128, wide, 1, 0
        return 256, 128, 1, 0
    if ksize == 1 and nk < 32:             # short-K 1x1 with cin != N (conv_shortcut, zero convs, time-embedding projections)
        if N > 4096:                       # (the concatenated time-embedding projections, N = 20,160: at most 255 column tiles per launch)
            return (64, 128, sk, 0) if M <= 2048 else (128, 128, sk, 0)
        if M <= 2048:
            return 64, 64, sk, 0
        if M < 16384 or N < 128:
            return 128, 64, sk, 13
        return 5256, (160 if N % 160 == 0 else 128), sk, (1 if N % 160 == 0 else 0)
    # 3x3 (stride 2, odd geometry, shortcut-folded, 8x8 level) and long-K Dense
    if M <= 256:
        return 64, 64, sk, 14
    if M <= 1024:
        return (4064, 128, sk, 23) if N % 128 == 0 else (128, 64, sk, 0)
    if M <= 4096 or N < 128:
        return (128, 128, sk, 4) if wide == 128 else (128, 64, sk, 0)
    if M < 8192:
        return 256, 128, sk, 0
    return 5256, (160 if N % 160 == 0 else 128), sk, (1 if N % 160 == 0 else 0)


def lookup(batch, h_in, w_in, cin, N, ksize, stride, upsample, M, nk, allow_split, cx=0) -> Tuple[int, int, int, int]:
    key = shape_key(batch, h_in, w_in, cin, N, ksize, stride, upsample, allow_split, cx)
    ent = _load().get(key)
    if ent is None:
        # a batch that was not measured takes the entry of the nearest measured batch of the same layer: every entry of
        # a layer is in the layer's one numerics class (tools/tune_conv.py), so the sample's bits stay the same
        fam = _families.get(key.split("x", 1)[1])
        if fam:
            ent = fam[min(fam, key=lambda b: (abs(b - batch) / (b + batch), b))]
            if int(ent[0]) == 256 and M < 1024:   # (the 256-row tile needs >= 1024 rows)
                ent = [128] + list(ent[1:])
            if is_big(int(ent[0])) and M < BIG_MIN_ROWS:   # (a 256-row macro tile on a small launch: the same class on small tiles)
                stg = int(ent[3]) if len(ent) > 4 else 0
                if stg >= 20 and cx:   # (shortcut-folded conv of the chunk-major class: a small launch runs on the halo-tile kernel, the same bits)
                    if w_in % 16 == 0 and h_in % 8 == 0:
                        ent = [1128, 80 if (N % 80 == 0 and M > 4096) else 64, int(ent[2]), 93 if M > 4096 else 0, 0.0]
                elif stg >= 10 and not upsample and stride == 1 and w_in % 16 == 0 and h_in % 8 == 0:
                    ent = [1128, 80 if N % 80 == 0 else 64, int(ent[2]), 0, 0.0]
                elif stg >= 10:   # (an upsampling layer of the chunk-major class: the halo-tile kernel does not take it; the big form walks any M)
                    ent = [5256, 128, int(ent[2]), 10, 0.0]
                else:
                    ent = [128, 128 if N % 128 == 0 else 64, int(ent[2]), 0, 0.0]
    if ent is not None:
        bm, bn, sk = int(ent[0]), int(ent[1]), int(ent[2])
        stages = int(ent[3]) if len(ent) > 4 else 0   # [bm, bn, splitk, stages, us] (older tables: [bm, bn, splitk, us])
        if not allow_split:
            sk = 1
        return bm, bn, sk, stages
    if os.environ.get("MSD_SHAPE_CONFIG", "1") == "0":   # (A/B runs: the plain-tile fallback of rounds 1-5)
        return heuristic(M // max(1, batch), N, nk, allow_split)
    return shape_config(batch, h_in, w_in, cin, N, ksize, stride, upsample, M, nk, allow_split, cx)


def numerics_class(ksize: int, tile_m: int, tile_n: int, splitk: int, ln_producer: bool = True, stages: int = 0) -> Tuple[bool, int, int]:
    """The part of a launch configuration that decides the ORDER of the layer's fp32 sums: kernel family (halo-tile 3x3
    kernel: chunk-major K walk), split-K slices, and for the 1x1 / dense layers that may PRODUCE LayerNorm-fold row moments
    the column tile (how the moments are grouped into partials).  `ln_producer` = False for the shapes that never do — the
    GEGLU and q|k|v projections, the keys ending in "n": their column tile orders nothing.  The table holds ONE class per
    layer shape for all batch sizes (tools/tune_conv.py)."""
    if tile_m >= 6000:
        raise ValueError(f"tile_m {tile_m} names no kernel form")
    if is_big(tile_m) and stages >= 10:   # big form walking K chunk-major: the halo-tile kernel's sums
        return (True, splitk, 0)
    if is_rowpanel(tile_m):   # row-panel Dense kernel: the tile kernel's bits (a launch it cannot take runs on the 128x64 tile)
        return (False, splitk, 64 if (ksize == 1 and ln_producer) else 0)
    if is_wreg(tile_m) or is_big(tile_m):   # wreg / big form: the tile kernel's K walk and epilogue, partials per column tile as requested
        return (False, splitk, tile_n if (ksize == 1 and ln_producer) else 0)
    return (is_halo(tile_m), splitk, tile_n if (ksize == 1 and ln_producer) else 0)


def key_is_ln_producer(key: str) -> bool:
    """Whether a launch of this shape may write LayerNorm-fold row moments (ln_out), i.e. whether its column tile is part of its
    numerics class.  Only three layers of a transformer block do (engine.Emitter.attentions: proj_in and the two to_out): 1x1,
    C -> C, no shortcut operand.  Not the GEGLU / q|k|v projections (keys ending in 'n': shape_key(allow_split=False)), not the
    1x1 layers with cin != N (conv_shortcut, the folded ff.net.2 + proj_out with K = 5 C)."""
    m = re.match(r"\d+x\d+x\d+x(\d+)->(\d+)k(\d)s\du[01](n?)(\+x\d+)?$", key)
    if m is None:
        return True
    cin, n, ks, nos, cx = m.groups()
    return ks == "1" and nos != "n" and cx is None and cin == n

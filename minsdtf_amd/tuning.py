"""Per-layer-shape launch configuration of msd_conv_gemm (tile size, split-K).

The kernel is bounded by the per-CU L2->LDS operand rate, so the best tile is a trade between
bytes per FLOP (bigger tiles) and workgroups in flight (smaller tiles / split-K) that depends on the
layer shape.  `conv_tuning.json` holds the configuration measured fastest on an MI355X for every
conv / dense shape of the SD1.5 UNet, ControlNet and VAE decoder at the benchmarked batch sizes
(produced by tools/tune_conv.py); shapes that are not in the table fall back to a size heuristic.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional, Tuple

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv_tuning.json")
_table: Optional[Dict[str, list]] = None

TILES = ((128, 128), (128, 64), (64, 64), (64, 128), (256, 128))


def shape_key(batch, h_in, w_in, cin, N, ksize, stride, upsample, allow_split=True) -> str:
    return f"{batch}x{h_in}x{w_in}x{cin}->{N}k{ksize}s{stride}u{int(bool(upsample))}{'' if allow_split else 'n'}"


def _load() -> Dict[str, list]:
    global _table
    if _table is None:
        try:
            with open(_PATH) as f:
                _table = json.load(f)
        except (OSError, ValueError):
            _table = {}
    return _table


def heuristic(M: int, N: int, nk: int, allow_split: bool) -> Tuple[int, int, int]:
    """(tile_m, tile_n, splitk) when the shape has not been measured."""
    bn = 128 if (N % 128 == 0 or N > 1024) else 64
    bm = 128
    if M <= 64:
        bm = 64
    tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    sk = 1
    if allow_split and tiles < 160 and nk >= 32:
        sk = max(1, min((256 + tiles - 1) // tiles, nk // 8, 16))
    return bm, bn, sk


def lookup(batch, h_in, w_in, cin, N, ksize, stride, upsample, M, nk, allow_split) -> Tuple[int, int, int]:
    ent = _load().get(shape_key(batch, h_in, w_in, cin, N, ksize, stride, upsample, allow_split))
    if ent is not None:
        bm, bn, sk = int(ent[0]), int(ent[1]), int(ent[2])
        if not allow_split:
            sk = 1
        return bm, bn, sk
    return heuristic(M, N, nk, allow_split)

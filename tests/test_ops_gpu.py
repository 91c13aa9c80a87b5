"""Per-kernel parity: each C-ABI entry point against the oracle's fp32 primitive on the same
(bf16-rounded) inputs.  Tolerances are written next to each check; with identical rounded inputs
the only differences are fp32 accumulation order and the final bf16 round of the output
(relative 2^-8), so bf16 outputs are compared with rtol 1e-2 and a small absolute floor."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import run_calls

pytestmark = pytest.mark.gpu


def bf(x):
    """Round an fp32 CPU tensor to bf16 and back (what the device will see)."""
    return x.to(torch.bfloat16).to(torch.float32)


def close(got, ref, rtol=1e-2, atol=None, what="", rms=None):
    """Element-wise: |got - ref| <= atol + rtol |ref| everywhere.  Aggregate: relative RMS error ||got - ref|| / ||ref|| <=
    `rms`, by default 2^-7 for a bf16 result (its own rounding is 2^-9 relative per element) and 2^-12 for an fp32 result of
    bf16 operands — the element-wise floor alone would let a dropped K chunk of one tap of a K = 2,304 contraction through;
    the aggregate does not."""
    is_f32 = got.dtype == torch.float32
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    if atol is None:
        atol = 1e-2 * float(ref.abs().max()) + 1e-6
    assert bool(torch.isfinite(got).all()), f"{what}: {int((~torch.isfinite(got)).sum())}/{got.numel()} non-finite outputs"
    err = (got - ref).abs()
    bad = err > (atol + rtol * ref.abs())
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} mismatches, max err {float(err.max()):.4g}, ref max {float(ref.abs().max()):.4g}"
    if rms is None:
        rms = 2.0 ** -12 if is_f32 else 2.0 ** -7
    rel = float((got - ref).double().norm() / max(float(ref.double().norm()), 1e-30))
    assert rel <= rms, f"{what}: relative RMS error {rel:.3g} > {rms:.3g}"


def conv_ref(x_nhwc, w_hwio, bias=None, stride=1, pad=1, upsample=False):
    x = x_nhwc.permute(0, 3, 1, 2)
    if upsample:
        x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    y = F.conv2d(x, w_hwio.permute(3, 2, 0, 1), bias, stride=stride, padding=pad)
    return y.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("case", [
    dict(B=2, H=16, W=16, c0=64, N=128, ks=3),                       # basic 3x3
    dict(B=1, H=8, W=8, c0=128, N=320, ks=3),                        # M=64 < tile, N=320 (64-wide tiles)
    dict(B=2, H=12, W=20, c0=64, N=192, ks=3, tile_n=128),           # ragged M and N edges
    dict(B=2, H=16, W=16, c0=64, N=64, ks=3, stride=2),              # stride-2 downsampler
    dict(B=1, H=8, W=8, c0=128, N=128, ks=3, upsample=True),         # nearest x2 fused into the gather
    dict(B=2, H=8, W=8, c0=128, c1=64, N=128, ks=3),                 # channel concat of two tensors
    dict(B=2, H=8, W=8, c0=64, c1=128, N=64, ks=1),                  # 1x1 shortcut over a concat
    dict(B=1, H=16, W=16, c0=320, N=320, ks=1),                      # dense / proj
    dict(B=2, H=8, W=8, c0=256, N=128, ks=3, splitk=4),              # split-K slabs + finalize
    dict(B=1, H=8, W=8, c0=1280, N=1280, ks=3, splitk=9),            # 8x8-level shape
    dict(B=2, H=16, W=16, c0=64, N=128, ks=3, f32out=True, act="silu"),
    dict(B=2, H=12, W=20, c0=64, N=192, ks=3, tile_m=64, tile_n=64),     # every tile configuration, ragged edges
    dict(B=2, H=12, W=20, c0=128, N=192, ks=3, tile_m=64, tile_n=128),
    dict(B=2, H=12, W=20, c0=128, N=400, ks=3, tile_m=128, tile_n=160),      # 128x160 tile: ragged M (480) and N (2.5 tiles)
    dict(B=1, H=16, W=16, c0=320, N=320, ks=1, tile_m=128, tile_n=160),      # ... on the dense form
    dict(B=3, H=12, W=20, c0=64, N=192, ks=3, tile_m=256, tile_n=128),
    dict(B=1, H=8, W=8, c0=256, N=320, ks=3, tile_m=64, tile_n=64, splitk=3),
    dict(B=2, H=16, W=16, c0=128, c1=64, N=128, ks=3, tile_m=256, tile_n=128, upsample=True),
    dict(B=2, H=16, W=16, c0=64, N=128, ks=3, tile_m=1128, tile_n=64),               # halo-tile 3x3 kernel, 8x16 tiles
    dict(B=2, H=16, W=32, c0=128, c1=64, N=192, ks=3, tile_m=1128, tile_n=128),      # halo + concat + ragged N
    dict(B=1, H=32, W=16, c0=128, N=320, ks=3, tile_m=1256, tile_n=128),             # halo 16x16 tiles
    dict(B=2, H=16, W=16, c0=256, N=128, ks=3, tile_m=1128, tile_n=64, splitk=2),    # halo + split over chunks
    dict(B=3, H=16, W=48, c0=320, N=320, ks=3, tile_m=1256, tile_n=128, splitk=3),
    dict(B=2, H=12, W=20, c0=64, N=64, ks=3, tile_m=1128, tile_n=64),                # not tileable -> generic fallback
    dict(B=2, H=16, W=16, c0=128, N=320, ks=3, tile_m=1128, tile_n=80),              # 80-wide tiles (5 fragments per wave): halo 8x16
    dict(B=1, H=32, W=16, c0=64, c1=64, N=200, ks=3, tile_m=1256, tile_n=80, splitk=2),   # halo 16x16 x 80, ragged N, split
    dict(B=2, H=16, W=16, c0=192, N=128, ks=3, tile_m=1128, tile_n=64, stages=8),    # deep weight rings of the halo kernel
    dict(B=2, H=16, W=32, c0=128, c1=64, N=192, ks=3, tile_m=1128, tile_n=128, stages=6, splitk=2),
    dict(B=1, H=16, W=16, c0=64, N=320, ks=3, tile_m=1128, tile_n=80, stages=8),     # ring deeper than the K loop of a chunk
    dict(B=1, H=32, W=32, c0=256, N=160, ks=3, tile_m=1256, tile_n=80, stages=5),
    dict(B=2, H=16, W=16, c0=128, N=192, ks=3, tile_m=2128, tile_n=64),              # 8x16 tiles on 8 waves
    dict(B=1, H=16, W=32, c0=64, c1=64, N=320, ks=3, tile_m=2128, tile_n=80, splitk=2),
    dict(B=2, H=12, W=20, c0=128, N=192, ks=3, tile_m=64, tile_n=64, stages=14),     # 8-wave variants of the small tiles
    dict(B=2, H=12, W=20, c0=64, c1=64, N=192, ks=1, tile_m=128, tile_n=64, stages=13, splitk=2),
    dict(B=1, H=8, W=8, c0=256, N=320, ks=3, tile_m=64, tile_n=128, stages=13, stride=2),
    dict(B=2, H=12, W=20, c0=128, N=320, ks=1, tile_m=128, tile_n=80),               # generic 128x80, ragged M
    dict(B=2, H=12, W=20, c0=192, N=168, ks=3, tile_m=128, tile_n=80, stages=4, splitk=3),
    dict(B=2, H=12, W=20, c0=128, N=320, ks=1, tile_m=128, tile_n=128, stages=23),    # 64x64 per wave (4 waves), DENSE loader, ragged M / N
    dict(B=2, H=12, W=20, c0=64, c1=128, N=192, ks=1, tile_m=128, tile_n=128, stages=24, splitk=2),   # + concat + split-K
    dict(B=3, H=12, W=20, c0=128, N=200, ks=3, tile_m=128, tile_n=64, stages=24),     # 2 waves of 64x64, general loader
    dict(B=2, H=12, W=20, c0=192, N=320, ks=1, tile_m=64, tile_n=128, stages=24),
    dict(B=1, H=12, W=20, c0=64, N=100, ks=1, tile_m=64, tile_n=64),                  # DENSE loader: M and N tails inside one tile
    dict(B=2, H=16, W=16, c0=128, N=128, ks=3, stride=2, asym=True),                 # VAE encoder downsampler: pad bottom/right only
    dict(B=1, H=10, W=18, c0=64, N=192, ks=3, stride=2, asym=True, splitk=3, tile_m=64, tile_n=64),
    # halo kernel, 3 taps (one filter row) per K step: stages 30 + ring depth
    dict(B=2, H=16, W=16, c0=192, N=128, ks=3, tile_m=1128, tile_n=64, stages=33, same_as=(1128, 64, 0)),
    dict(B=1, H=16, W=32, c0=64, c1=64, N=100, ks=3, tile_m=1128, tile_n=64, stages=34, same_as=(1128, 64, 0)),
    dict(B=2, H=8, W=16, c0=320, N=320, ks=3, tile_m=1128, tile_n=80, stages=33, same_as=(1128, 80, 0)),
    dict(B=2, H=16, W=16, c0=128, N=192, ks=3, tile_m=2128, tile_n=64, stages=33, same_as=(1128, 64, 0)),
    dict(B=2, H=16, W=16, c0=256, N=128, ks=3, tile_m=1128, tile_n=64, stages=33, splitk=2),       # + split over chunks
    # ... with two loader waves behind the compute waves (stages 60 + depth): a different DMA row distribution, the same bits
    dict(B=2, H=8, W=16, c0=320, N=320, ks=3, tile_m=1128, tile_n=80, stages=63, same_as=(1128, 80, 0)),
    dict(B=1, H=16, W=32, c0=64, c1=64, N=100, ks=3, tile_m=1128, tile_n=64, stages=63, same_as=(1128, 64, 0)),
    dict(B=2, H=16, W=16, c0=256, N=128, ks=3, tile_m=1128, tile_n=64, stages=63, splitk=2),
    # ... rotated K loop (stages 90 + depth, with the loader waves): every fragment read in front of MFMAs that do not need it, the
    # barrier in the middle of a step; same taps in the same order, the same bits
    dict(B=2, H=8, W=16, c0=320, N=320, ks=3, tile_m=1128, tile_n=80, stages=93, same_as=(1128, 80, 0)),
    dict(B=1, H=16, W=32, c0=64, c1=64, N=100, ks=3, tile_m=1128, tile_n=64, stages=93, same_as=(1128, 64, 0)),     # one chunk per tensor of the concat, ragged N
    dict(B=2, H=16, W=16, c0=256, N=128, ks=3, tile_m=1128, tile_n=64, stages=93, splitk=2),
    dict(B=2, H=16, W=16, c0=64, N=128, ks=3, tile_m=1128, tile_n=64, stages=93, same_as=(1128, 64, 0)),            # a single chunk: 3 steps
    dict(B=1, H=16, W=16, c0=320, N=160, ks=3, tile_m=1128, tile_n=80, stages=93, splitk=2, same_as=(1128, 80, 0)), # 5 chunks in slices of 3 + 2
    dict(B=2, H=32, W=32, c0=192, c1=128, N=320, ks=3, tile_m=1128, tile_n=80, stages=93, same_as=(1128, 80, 0)),   # concat boundary inside the walk, 16 tiles per sample
    # ... and the one-tap form rotated (stages 150 + depth): tap it + 1's fragments read under tap it's MFMAs, all S stages in flight
    dict(B=2, H=16, W=16, c0=192, N=128, ks=3, tile_m=1128, tile_n=64, stages=153, same_as=(1128, 64, 0)),
    dict(B=1, H=16, W=32, c0=64, c1=64, N=100, ks=3, tile_m=1128, tile_n=64, stages=158, same_as=(1128, 64, 0)),    # ring deeper than a chunk's first steps, concat, ragged N
    dict(B=2, H=16, W=16, c0=256, N=128, ks=3, tile_m=1128, tile_n=64, stages=153, splitk=2, same_as=(1128, 64, 0)),
    dict(B=2, H=16, W=32, c0=128, c1=64, N=192, ks=3, tile_m=1128, tile_n=128, stages=153, same_as=(1128, 128, 0)),
    dict(B=2, H=16, W=32, c0=128, c1=64, N=192, ks=3, tile_m=1128, tile_n=128, stages=156, splitk=2, same_as=(1128, 128, 0)),
    dict(B=2, H=16, W=16, c0=128, N=256, ks=3, tile_m=1256, tile_n=128, stages=153, same_as=(1256, 128, 0)),
    dict(B=1, H=32, W=32, c0=64, N=128, ks=3, tile_m=1256, tile_n=128, stages=153, same_as=(1256, 128, 0)),         # a single chunk: 9 steps
    dict(B=1, H=16, W=16, c0=64, N=320, ks=3, tile_m=1128, tile_n=80, stages=158, same_as=(1128, 80, 0)),           # ring of 8 on a 9-step walk
    dict(B=2, H=16, W=16, c0=320, N=320, ks=3, tile_m=1256, tile_n=80, stages=153, same_as=(1256, 80, 0)),
    dict(B=2, H=16, W=16, c0=128, c1=192, N=160, ks=3, tile_m=1256, tile_n=80, stages=155, splitk=2, same_as=(1256, 80, 0)),
    # wreg form (conv_wreg.hip; tile_m 4000 + rows): weights global -> VGPR from the fragment-major image (w_layout 2), all waves
    # split over N; the tile kernel's K walk and epilogue, so its bits
    dict(B=2, H=16, W=16, c0=64, N=128, ks=3, tile_m=4128, tile_n=128, stages=3, same_as=(128, 128, 0)),
    dict(B=2, H=12, W=20, c0=128, N=192, ks=3, tile_m=4128, tile_n=128, stages=4, same_as=(128, 128, 0)),      # ragged M (480) and N (1.5 tiles)
    dict(B=1, H=16, W=16, c0=320, N=320, ks=1, tile_m=4128, tile_n=64, stages=3, same_as=(128, 64, 0)),        # dense loader, one block per wave
    dict(B=1, H=16, W=16, c0=320, N=320, ks=1, tile_m=4128, tile_n=64, stages=4, same_as=(128, 64, 0)),
    dict(B=2, H=8, W=8, c0=64, c1=128, N=64, ks=1, tile_m=4064, tile_n=64, stages=4, same_as=(64, 64, 0)),      # 1x1 over a concat, K = 3 tiles
    dict(B=1, H=8, W=8, c0=1280, N=1280, ks=3, splitk=9, tile_m=4128, tile_n=128, stages=3, same_as=(64, 128, 0)),   # 8x8 level: BM = M, split-K
    dict(B=1, H=8, W=8, c0=1280, N=1280, ks=3, splitk=12, tile_m=4128, tile_n=64, stages=4, same_as=(64, 128, 0)),
    dict(B=2, H=16, W=16, c0=64, N=64, ks=3, stride=2, tile_m=4064, tile_n=128, stages=3, same_as=(64, 64, 0)),      # stride 2; N < the column tile
    dict(B=1, H=8, W=8, c0=128, N=128, ks=3, upsample=True, tile_m=4064, tile_n=128, stages=4, same_as=(64, 64, 0)),
    dict(B=2, H=8, W=8, c0=128, c1=64, N=272, ks=3, tile_m=4064, tile_n=256, stages=3, same_as=(64, 64, 0)),       # 4 blocks per wave, ragged N (17 blocks)
    dict(B=2, H=8, W=8, c0=128, c1=64, N=272, ks=3, tile_m=4064, tile_n=256, stages=4, same_as=(64, 64, 0)),
    dict(B=3, H=12, W=20, c0=64, N=192, ks=3, tile_m=4256, tile_n=64, stages=3, same_as=(128, 64, 0)),             # 256 rows x one block per wave
    dict(B=2, H=12, W=20, c0=128, N=320, ks=1, tile_m=4128, tile_n=128, stages=13, same_as=(128, 64, 0)),          # 8 waves
    dict(B=2, H=16, W=16, c0=128, N=128, ks=3, stride=2, asym=True, tile_m=4064, tile_n=64, stages=4, same_as=(64, 64, 0)),
    dict(B=2, H=16, W=16, c0=64, N=128, ks=3, f32out=True, act="silu", tile_m=4128, tile_n=64, stages=3, same_as=(128, 64, 0)),
    dict(B=2, H=12, W=20, c0=64, N=128, ks=3, tile_m=4064, tile_n=64, stages=4, splitk=3, same_as=(64, 64, 0)),    # K = 9 tiles in 3 slices of 3 (< ring depth + 1)
    # ... two K tiles per ring stage (stages 20 + depth): odd tile counts end in a half-full stage
    dict(B=2, H=12, W=20, c0=64, N=128, ks=3, tile_m=4064, tile_n=64, stages=23, same_as=(64, 64, 0)),             # 9 tiles = 4 stages + 1 tile
    dict(B=2, H=12, W=20, c0=64, N=128, ks=3, tile_m=4064, tile_n=64, stages=24, splitk=3, same_as=(64, 64, 0)),   # slices of 3 tiles: 1.5 stages each
    dict(B=1, H=16, W=16, c0=320, N=320, ks=1, tile_m=4128, tile_n=64, stages=23, same_as=(128, 64, 0)),           # dense, 5 tiles
    dict(B=2, H=8, W=8, c0=64, c1=128, N=64, ks=1, tile_m=4064, tile_n=128, stages=23, same_as=(64, 64, 0)),       # concat boundary inside a stage
    dict(B=1, H=8, W=8, c0=1280, N=1280, ks=3, splitk=12, tile_m=4064, tile_n=128, stages=24, same_as=(64, 128, 0)),   # 8x8 level: 15 tiles per slice
    dict(B=2, H=8, W=8, c0=128, c1=64, N=272, ks=3, tile_m=4064, tile_n=256, stages=23, same_as=(64, 64, 0)),      # 27 tiles, 4 blocks per wave
    dict(B=2, H=16, W=16, c0=128, N=128, ks=3, stride=2, tile_m=4128, tile_n=128, stages=23, same_as=(128, 128, 0)),
    dict(B=1, H=8, W=8, c0=128, N=128, ks=3, upsample=True, tile_m=4064, tile_n=64, stages=23, splitk=2, same_as=(64, 64, 0)),
    # ... 256 rows on 8 waves as a 2 x 4 wave grid (stages 10 + depth, + 20 for two K tiles per stage)
    dict(B=3, H=12, W=20, c0=64, N=192, ks=3, tile_m=4256, tile_n=128, stages=13, same_as=(256, 128, 0)),         # ragged M (720) and N
    dict(B=2, H=16, W=16, c0=128, c1=64, N=128, ks=3, tile_m=4256, tile_n=128, stages=14, upsample=True, same_as=(256, 128, 0)),
    dict(B=3, H=12, W=20, c0=128, N=320, ks=1, tile_m=4256, tile_n=128, stages=13, same_as=(128, 128, 0)),        # dense loader
    # big form (conv_big.hip: tile_m 5000 + rows; 8 waves in two half-workgroups one barrier apart, operands by LDS-DMA in need order):
    # the tile kernel's K walk and epilogue, so the tile kernel's bits
    dict(B=2, H=16, W=16, c0=64, N=256, ks=3, tile_m=5256, tile_n=256, stages=0, same_as=(128, 128, 0)),           # 2 row tiles, 9 K tiles
    dict(B=3, H=12, W=20, c0=128, N=272, ks=3, tile_m=5256, tile_n=256, stages=0, same_as=(128, 128, 0)),          # ragged M (720) and N (17 blocks), 18 K tiles
    dict(B=3, H=12, W=20, c0=128, N=320, ks=3, tile_m=5256, tile_n=160, stages=0, same_as=(128, 64, 0)),           # 256 x 160, quadrant phases, 3 buffers
    dict(B=3, H=12, W=20, c0=128, N=320, ks=3, tile_m=5256, tile_n=160, stages=1, same_as=(128, 64, 0)),           # ... row-half phases
    dict(B=3, H=12, W=20, c0=128, N=320, ks=3, tile_m=5256, tile_n=160, stages=2, same_as=(128, 64, 0)),           # ... quadrants, 2 buffers
    dict(B=2, H=16, W=16, c0=64, c1=64, N=100, ks=3, tile_m=5256, tile_n=128, stages=0, same_as=(128, 64, 0)),     # concat, N < the column tile
    dict(B=2, H=16, W=16, c0=64, c1=64, N=192, ks=3, tile_m=5256, tile_n=128, stages=1, same_as=(128, 64, 0)),
    dict(B=2, H=12, W=20, c0=64, N=272, ks=3, tile_m=5128, tile_n=256, stages=0, same_as=(128, 128, 0)),           # 128-row tiles, ragged M (480)
    dict(B=2, H=12, W=20, c0=64, N=512, ks=3, tile_m=5128, tile_n=256, stages=1, same_as=(128, 128, 0)),
    dict(B=2, H=16, W=16, c0=64, N=256, ks=1, tile_m=5256, tile_n=256, stages=0, same_as=(128, 128, 0)),           # dense loader, ONE K tile (nothing to prefetch)
    dict(B=2, H=16, W=16, c0=128, N=320, ks=1, tile_m=5256, tile_n=160, stages=0, same_as=(128, 64, 0)),           # two K tiles = the prologue's whole lead (3 buffers)
    dict(B=2, H=16, W=16, c0=64, c1=128, N=320, ks=1, tile_m=5256, tile_n=160, stages=1, same_as=(128, 64, 0)),    # 1x1 over a concat, 3 K tiles
    dict(B=3, H=12, W=20, c0=320, N=320, ks=1, tile_m=5256, tile_n=256, stages=0, same_as=(128, 64, 0)),           # 5 K tiles, 2 buffers, ragged M and N
    dict(B=2, H=32, W=32, c0=64, N=160, ks=3, stride=2, tile_m=5256, tile_n=160, stages=0, same_as=(128, 64, 0)),  # stride 2
    dict(B=2, H=32, W=32, c0=64, N=128, ks=3, stride=2, asym=True, tile_m=5256, tile_n=128, stages=0, same_as=(128, 64, 0)),
    dict(B=2, H=8, W=8, c0=128, c1=64, N=256, ks=3, upsample=True, tile_m=5256, tile_n=256, stages=0, same_as=(128, 128, 0)),   # nearest x2 in the loader
    dict(B=2, H=8, W=8, c0=128, N=160, ks=3, upsample=True, tile_m=5256, tile_n=160, stages=1, same_as=(128, 64, 0)),
    dict(B=2, H=16, W=16, c0=64, N=256, ks=3, f32out=True, act="silu", tile_m=5256, tile_n=256, stages=0, same_as=(128, 128, 0)),
    dict(B=2, H=16, W=16, c0=128, N=320, ks=3, splitk=3, tile_m=5256, tile_n=160, stages=0, same_as=(128, 64, 0)), # 18 K tiles in 3 slices of 6
    dict(B=2, H=16, W=16, c0=64, N=256, ks=3, splitk=4, tile_m=5256, tile_n=256, stages=0, same_as=(128, 128, 0)), # slices of 3, 3, 3, 0 -> 3 slices
    # ... walking K chunk-major (stages code + 10): the HALO-tile kernel's order of sums, so that kernel's bits
    dict(B=2, H=16, W=16, c0=192, N=128, ks=3, tile_m=5256, tile_n=128, stages=10, same_as=(1128, 64, 0)),
    dict(B=1, H=16, W=32, c0=64, c1=64, N=100, ks=3, tile_m=5256, tile_n=128, stages=10, same_as=(1128, 64, 0)),        # one chunk per tensor of the concat, ragged N
    dict(B=3, H=16, W=16, c0=320, N=320, ks=3, tile_m=5256, tile_n=160, stages=11, same_as=(1128, 80, 0)),              # 45 K steps
    dict(B=2, H=16, W=16, c0=256, N=256, ks=3, splitk=2, tile_m=5256, tile_n=256, stages=10, same_as=(1128, 64, 0)),    # split over chunks: 2 + 2
    dict(B=2, H=16, W=16, c0=320, N=160, ks=3, splitk=2, tile_m=5256, tile_n=160, stages=10, same_as=(1128, 80, 0)),    # 5 chunks in slices of 3 + 2
    # big form on a staged halo (stages 20 + code): 16x16-pixel tiles, every chunk's 18x18 halo staged once - again conv_halo's bits
    dict(B=2, H=16, W=16, c0=64, N=160, ks=3, tile_m=5256, tile_n=160, stages=20, same_as=(1128, 80, 0)),               # ONE chunk: nothing to prefetch but fillers; every border
    dict(B=1, H=32, W=48, c0=192, N=128, ks=3, tile_m=5256, tile_n=128, stages=20, same_as=(1128, 64, 0)),              # 2 x 3 tiles: interior borders read neighbours, 3 chunks
    dict(B=3, H=16, W=32, c0=320, N=320, ks=3, tile_m=5256, tile_n=160, stages=20, same_as=(1128, 80, 0)),              # 5 chunks (both halo buffers reused), 2 column tiles
    dict(B=1, H=16, W=32, c0=64, c1=64, N=100, ks=3, tile_m=5256, tile_n=128, stages=21, same_as=(1128, 64, 0)),        # one chunk per tensor of the concat, ragged N, weight ring of 4
    dict(B=2, H=32, W=16, c0=128, c1=192, N=272, ks=3, tile_m=5256, tile_n=160, stages=20, same_as=(1128, 80, 0)),      # concat boundary inside the walk, N = 17 blocks (ragged second column tile)
    dict(B=2, H=16, W=16, c0=320, N=160, ks=3, splitk=2, tile_m=5256, tile_n=160, stages=20, same_as=(1128, 80, 0)),    # 5 chunks in slices of 3 + 2
    dict(B=2, H=16, W=16, c0=256, N=256, ks=3, splitk=4, tile_m=5256, tile_n=128, stages=21, same_as=(1128, 64, 0)),    # one chunk per slice
    dict(B=2, H=16, W=16, c0=128, N=128, ks=3, f32out=True, act="silu", tile_m=5256, tile_n=128, stages=20, same_as=(1128, 64, 0)),
    dict(B=2, H=8, W=8, c0=128, c1=64, N=256, ks=3, upsample=True, tile_m=5256, tile_n=128, stages=20, same_as=(5256, 128, 10)),   # nearest x2 in the halo staging (one 16x16 tile per sample)
    dict(B=1, H=16, W=24, c0=64, N=160, ks=3, upsample=True, tile_m=5256, tile_n=160, stages=20, same_as=(5256, 160, 11)),          # 2 x 3 tiles of the 32 x 48 image
    dict(B=2, H=16, W=16, c0=128, c1=192, N=256, ks=3, tile_m=5128, tile_n=256, stages=10, same_as=(1256, 128, 0)),     # concat boundary inside the walk
])
def test_conv_gemm(gpu, case):
    from minsdtf_amd import ops, packing

    torch.manual_seed(1)
    B, H, W, c0, N, ks = case["B"], case["H"], case["W"], case["c0"], case["N"], case["ks"]
    c1 = case.get("c1", 0)
    stride, ups = case.get("stride", 1), case.get("upsample", False)
    splitk = case.get("splitk", 1)
    cin = c0 + c1
    x0 = bf(torch.randn(B, H, W, c0))
    x1 = bf(torch.randn(B, H, W, c1)) if c1 else None
    w = bf(torch.randn(ks, ks, cin, N) / math.sqrt(ks * ks * cin))
    bias = torch.randn(N)
    pad = 1 if ks == 3 else 0
    xin = torch.cat([x0, x1], dim=-1) if c1 else x0
    asym = case.get("asym", False)
    if asym:   # image_encoder.py PaddedConv2D(padding=((0,1),(0,1)), strides=2)
        ref = conv_ref(F.pad(xin, (0, 0, 0, 1, 0, 1)), w, bias, stride=stride, pad=0)
    else:
        ref = conv_ref(xin, w, bias, stride=stride, pad=pad, upsample=ups)
    Ho, Wo = ref.shape[1], ref.shape[2]
    M = B * Ho * Wo
    steps = 3
    temb = torch.randn(steps, B, N)
    resid = bf(torch.randn(B, Ho, Wo, N))
    ref = ref + temb[2][:, None, None, :]
    if case.get("act") == "silu":
        ref = ref * torch.sigmoid(ref)
    ref = ref + resid

    d = gpu
    wp = packing.pack_conv(w.numpy(), d)
    x0d, x1d = x0.to(torch.bfloat16).to(d), (x1.to(torch.bfloat16).to(d) if c1 else None)
    f32out = case.get("f32out", False)
    out = torch.full((M, N), float("nan"), dtype=torch.float32 if f32out else torch.bfloat16, device=d)
    ws = torch.empty(max(1, splitk * M * N), dtype=torch.float32, device=d)
    step = torch.tensor([2], dtype=torch.int32, device=d)
    biasd, tembd, residd = bias.to(d), temb.to(d), resid.to(torch.bfloat16).to(d)   # (a Call holds addresses, not tensors)
    wreg = 4000 <= case.get("tile_m", 0) < 5000
    wmain = packing.fragment_major(wp) if wreg else wp
    call = ops.conv_gemm(a0=x0d, a1=x1d, c1=c1, w=wmain, w_layout=2 if wreg else 0, out=out, batch=B, h_in=H, w_in=W, c0=c0, N=N, ksize=ks, stride=stride,
                         upsample=ups, bias=biasd, rowvec=tembd, rv_step_stride=B * N, rv_batch_stride=N,
                         step_ptr=step, residual=residd, act=ops.ACT_SILU if case.get("act") else ops.ACT_NONE,
                         out_dtype=ops.OUT_F32 if f32out else ops.OUT_BF16, workspace=ws, workspace_floats=ws.numel(),
                         splitk=splitk, tile_n=case.get("tile_n", 0), tile_m=case.get("tile_m", 0), stages=case.get("stages", 0),
                         **(dict(pad=0, pad_end=1) if asym else {}))
    run_calls(call)
    close(out.reshape(B, Ho, Wo, N), ref, what=str(case))
    if case.get("same_as"):   # tile shape / kernel form never changes a result's bits (K tiles are summed in the same order)
        tm, tn, stg = case["same_as"]
        out2 = torch.full_like(out, float("nan"))
        run_calls(ops.conv_gemm(a0=x0d, a1=x1d, c1=c1, w=wp, out=out2, batch=B, h_in=H, w_in=W, c0=c0, N=N, ksize=ks, stride=stride,
                                upsample=ups, bias=biasd, rowvec=tembd, rv_step_stride=B * N, rv_batch_stride=N, step_ptr=step,
                                residual=residd, act=ops.ACT_SILU if case.get("act") else ops.ACT_NONE,
                                out_dtype=ops.OUT_F32 if f32out else ops.OUT_BF16, tile_n=tn, tile_m=tm, stages=stg,
                                workspace=ws, workspace_floats=ws.numel(), splitk=splitk, **(dict(pad=0, pad_end=1) if asym else {})))
        bits = torch.int32 if f32out else torch.int16
        assert torch.equal(out.view(bits), out2.view(bits)), "tile shape / kernel form changed the bits"
    if wreg:   # (the fragment-major image IS the layout this form reads)
        return
    # chunk-major weights [K/64][N][64] (w_layout = 1, the form the models keep): storage order only, the same bits
    out3 = torch.full_like(out, float("nan"))
    run_calls(ops.conv_gemm(a0=x0d, a1=x1d, c1=c1, w=packing.chunk_major(wp), w_layout=1, out=out3, batch=B, h_in=H, w_in=W, c0=c0, N=N,
                            ksize=ks, stride=stride, upsample=ups, bias=biasd, rowvec=tembd, rv_step_stride=B * N,
                            rv_batch_stride=N, step_ptr=step, residual=residd,
                            act=ops.ACT_SILU if case.get("act") else ops.ACT_NONE, out_dtype=ops.OUT_F32 if f32out else ops.OUT_BF16,
                            workspace=ws, workspace_floats=ws.numel(), splitk=splitk, tile_n=case.get("tile_n", 0),
                            tile_m=case.get("tile_m", 0), stages=case.get("stages", 0), **(dict(pad=0, pad_end=1) if asym else {})))
    assert torch.equal(out.view(torch.int32 if f32out else torch.int16), out3.view(torch.int32 if f32out else torch.int16)), "weight layout changed the bits"


@pytest.mark.parametrize("case", [
    dict(B=2, H=12, W=20, c=128, cx0=64, cx1=0, N=128, ks=3),                          # ResBlock conv2 + 1x1 shortcut, ragged M
    dict(B=2, H=8, W=8, c=192, cx0=128, cx1=64, N=192, ks=3, splitk=3),                # shortcut over a concat, split-K across both parts
    dict(B=1, H=16, W=16, c=64, cx0=64, cx1=0, N=320, ks=3, tile_m=64, tile_n=64),
    dict(B=2, H=12, W=20, c=64, cx0=128, cx1=0, N=100, ks=1, tile_m=128, tile_n=64, stages=13),   # 1x1 main part
    dict(B=2, H=8, W=8, c=192, cx0=128, cx1=64, N=192, ks=3, splitk=3, tile_m=4128, tile_n=128, stages=3),   # wreg form (fragment-major weights)
    dict(B=1, H=16, W=16, c=64, cx0=64, cx1=0, N=320, ks=3, tile_m=4064, tile_n=256, stages=4),
    dict(B=2, H=8, W=8, c=192, cx0=128, cx1=64, N=192, ks=3, splitk=3, tile_m=4064, tile_n=128, stages=23),   # ... two K tiles per stage
    dict(B=3, H=12, W=20, c=128, cx0=64, cx1=0, N=320, ks=3, tile_m=5256, tile_n=160, stages=0),              # big form, ragged M
    dict(B=2, H=16, W=16, c=192, cx0=128, cx1=64, N=256, ks=3, splitk=3, tile_m=5256, tile_n=256, stages=0),  # ... shortcut over a concat, split-K across both parts
    dict(B=2, H=16, W=16, c=64, cx0=128, cx1=0, N=256, ks=1, tile_m=5128, tile_n=256, stages=0),              # ... 1x1 main part
    # big form on a staged halo: the slice's shortcut chunks follow its main chunks (chunk-major: its own numerics class - compared
    # with another configuration of the same form)
    dict(B=2, H=16, W=32, c=128, cx0=64, cx1=0, N=160, ks=3, tile_m=5256, tile_n=160, stages=20, other=(5256, 128, 20)),              # ONE shortcut chunk (nothing to prefetch)
    dict(B=1, H=32, W=16, c=192, cx0=128, cx1=64, N=272, ks=3, tile_m=5256, tile_n=128, stages=20, other=(5256, 160, 20)),            # shortcut over a concat: 3 chunks, ragged N
    dict(B=2, H=16, W=16, c=320, cx0=640, cx1=0, N=320, ks=3, splitk=3, tile_m=5256, tile_n=128, stages=21, other=(5256, 160, 20)),   # 5 + 10 chunks over 3 slices (2, 2, 1 main; 4, 4, 2 extra)
    dict(B=1, H=16, W=16, c=128, cx0=64, cx1=0, N=128, ks=3, splitk=2, tile_m=5256, tile_n=128, stages=20, other=(5256, 128, 21)),    # the second slice has no shortcut chunk
    # round 6: the halo-tile kernel walks the shortcut chunks behind the slice's main chunks too (conv_halo.hip) - every loop form of it, against
    # the staged-halo big form: ONE numerics class for these layers at every batch (the table no longer needs a second profile)
    dict(B=2, H=16, W=32, c=128, cx0=64, cx1=0, N=160, ks=3, tile_m=1128, tile_n=64, stages=0, other=(5256, 128, 20)),                 # plain loop, ONE shortcut chunk, ragged N
    dict(B=1, H=32, W=16, c=192, cx0=128, cx1=64, N=272, ks=3, tile_m=2128, tile_n=64, stages=33, other=(5256, 160, 20)),              # 3 taps per step on 8 waves; shortcut over a concat
    dict(B=2, H=16, W=16, c=320, cx0=640, cx1=0, N=320, ks=3, splitk=3, tile_m=1128, tile_n=80, stages=93, other=(5256, 128, 21)),     # rotated loop + loader waves; 5 + 10 chunks over 3 slices
    dict(B=1, H=16, W=16, c=128, cx0=64, cx1=0, N=128, ks=3, splitk=2, tile_m=1128, tile_n=64, stages=153, other=(5256, 128, 20)),     # one-tap rotated loop; the second slice has no shortcut chunk
    dict(B=2, H=16, W=16, c=128, cx0=192, cx1=0, N=256, ks=3, tile_m=1256, tile_n=128, stages=0, other=(5256, 128, 20)),               # 16 x 16-pixel tiles: a 256-row shortcut tile
    dict(B=2, H=16, W=32, c=192, cx0=64, cx1=128, N=160, ks=3, tile_m=1128, tile_n=80, stages=63, other=(5256, 160, 20)),              # loader waves, lock-step loop; concat boundary between shortcut chunks
    dict(B=3, H=32, W=32, c=64, cx0=320, cx1=0, N=128, ks=3, tile_m=1128, tile_n=128, stages=6, other=(5256, 128, 20)),                # more shortcut chunks (5) than main chunks (1), deep ring
    dict(B=2, H=8, W=16, c=128, cx0=128, cx1=0, N=64, ks=3, splitk=2, tile_m=2128, tile_n=64, stages=33),                              # an 8-row image (no 16 x 16 tile): the kernel against the fp32 answer alone
])
def test_conv_gemm_shortcut_operand(gpu, case):
    """conv(h) + conv1x1(x) as one contraction (diffusion_model.py:34-38,50): K = taps of h, then the channels of x."""
    from minsdtf_amd import ops

    torch.manual_seed(21)
    B, H, W, c, N, ks = case["B"], case["H"], case["W"], case["c"], case["N"], case["ks"]
    cx0, cx1 = case["cx0"], case["cx1"]
    cx = cx0 + cx1
    h = bf(torch.randn(B, H, W, c))
    x0 = bf(torch.randn(B, H, W, cx0))
    x1 = bf(torch.randn(B, H, W, cx1)) if cx1 else None
    w2 = bf(torch.randn(ks, ks, c, N) / math.sqrt(ks * ks * c))
    wsc = bf(torch.randn(1, 1, cx, N) / math.sqrt(cx))
    b2, bs = torch.randn(N), torch.randn(N)
    xin = torch.cat([x0, x1], dim=-1) if cx1 else x0
    ref = conv_ref(h, w2, b2, pad=1 if ks == 3 else 0) + conv_ref(xin, wsc, bs, pad=0)
    d = gpu
    wcat = torch.cat([w2.permute(3, 0, 1, 2).reshape(N, -1), wsc.permute(3, 0, 1, 2).reshape(N, -1)], dim=1).to(torch.bfloat16).contiguous().to(d)
    M = B * H * W
    sk = case.get("splitk", 1)
    keep = [h.to(torch.bfloat16).to(d), x0.to(torch.bfloat16).to(d), x1.to(torch.bfloat16).to(d) if cx1 else None, (b2 + bs).to(d),
            torch.empty(max(1, sk * M * N), dtype=torch.float32, device=d)]
    from minsdtf_amd import packing

    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=d)
    wreg = 4000 <= case.get("tile_m", 0) < 5000
    wmain = packing.fragment_major(wcat) if wreg else wcat
    call = ops.conv_gemm(a0=keep[0], w=wmain, w_layout=2 if wreg else 0, out=out, batch=B, h_in=H, w_in=W, c0=c, N=N, ksize=ks, bias=keep[3],
                         a2=keep[1], c2=cx0, a3=keep[2], c3=cx1, workspace=keep[4], workspace_floats=keep[4].numel(), splitk=sk,
                         tile_m=case.get("tile_m", 0), tile_n=case.get("tile_n", 0), stages=case.get("stages", 0))
    run_calls(call)
    close(out.reshape(B, H, W, N), ref, what=str(case))
    # chunk-major weights on the same tile (wreg: on the tile kernel): storage order / kernel form only, the same bits
    big = case.get("tile_m", 0) >= 5000
    other = dict(tile_m=64, tile_n=128) if (wreg or big) else dict(tile_m=case.get("tile_m", 0), tile_n=case.get("tile_n", 0), stages=case.get("stages", 0))
    if case.get("other"):
        other = dict(tile_m=case["other"][0], tile_n=case["other"][1], stages=case["other"][2])
    out2 = torch.full_like(out, float("nan"))
    run_calls(ops.conv_gemm(a0=keep[0], w=packing.chunk_major(wcat), w_layout=1, out=out2, batch=B, h_in=H, w_in=W, c0=c, N=N, ksize=ks,
                            bias=keep[3], a2=keep[1], c2=cx0, a3=keep[2], c3=cx1, workspace=keep[4], workspace_floats=keep[4].numel(),
                            splitk=sk, **other))
    assert torch.equal(out.view(torch.int16), out2.view(torch.int16)), "weight layout / kernel form changed the bits"


@pytest.mark.parametrize("tile", [(0, 0), (128, 160), (64, 128), (3128, 320), (4128, 128), (4064, 256), (5256, 256), (5128, 256), (5256, 128)])   # heuristic tile, the 128x160 tile (16x16 level), 64x128, row panels, wreg, big
def test_conv_gemm_geglu(gpu, tile):
    from minsdtf_amd import ops, packing

    torch.manual_seed(2)
    M, C = (192, 64) if not 3000 <= tile[0] < 4000 else (200, 320)   # (the row-panel kernel takes K = 320 / 640)
    if tile[0] >= 5000:
        M, C = 600, 128   # three ragged 256-row tiles, two K tiles
    x = bf(torch.randn(M, C))
    w = bf(torch.randn(C, 8 * C) / math.sqrt(C))
    b = torch.randn(8 * C) * 0.1
    h = x @ w + b
    a, gate = h[:, :4 * C], h[:, 4 * C:]
    ref = a * 0.5 * gate * (1 + torch.tanh(gate * 0.7978845608 * (1 + 0.044715 * gate ** 2)))
    wp, bp = packing.pack_geglu(w.numpy(), b.numpy(), gpu)
    out = torch.full((M, 4 * C), float("nan"), dtype=torch.bfloat16, device=gpu)
    wreg = 4000 <= tile[0] < 5000
    call = ops.conv_gemm(a0=x.to(torch.bfloat16).to(gpu), w=packing.fragment_major(wp) if wreg else wp, w_layout=2 if wreg else 0, out=out, batch=1,
                         h_in=M, w_in=1, c0=C, N=8 * C, bias=bp, act=ops.ACT_GEGLU, tile_m=tile[0], tile_n=tile[1], stages=3 if wreg else 0)
    run_calls(call)
    close(out, ref, what=f"geglu {tile}")


@pytest.mark.parametrize("case", [
    dict(M=200, C=320, tile=(128, 128, 0), mode="dense"),      # 3 column tiles x 4 wave slabs, ragged M
    dict(M=192, C=320, tile=(128, 80, 0), mode="qkv"),         # 80-wide producer tiles, q|k|v^T epilogue
    dict(M=128, C=640, tile=(64, 64, 0), mode="geglu"),        # 10 partials per row, GEGLU consumer
    dict(M=130, C=1280, tile=(64, 64, 14), mode="dense"),      # 20 partials per row (the maximum), 8-wave producer
    dict(M=256, C=320, tile=(64, 128, 24), mode="dense"),      # one slab per tile
    dict(M=200, C=320, tile=(128, 80, 0), mode="geglu", ctile=(128, 128, 0)),     # consumer tiles chosen explicitly
    dict(M=300, C=1280, tile=(64, 64, 0), mode="geglu", ctile=(128, 160, 0)),     # the 16x16 level's GEGLU projection on the 128x160 tile
    dict(M=136, C=640, tile=(64, 64, 0), mode="qkv", ctile=(64, 128, 13)),
    # consumer on the row-panel kernel (conv_rowpanel.hip: tile_m = 3000 + rows per workgroup, tile_n = columns per
    # workgroup); `csame`: the same bits as the tile kernel
    dict(M=600, C=320, tile=(128, 128, 0), mode="geglu", ctile=(3128, 320, 0), csame=True),   # 5 ragged row panels x 8 column shares
    dict(M=300, C=320, tile=(128, 64, 0), mode="qkv", ctile=(3128, 192, 0), csame=True),      # q | k | v^T, a share that straddles k | v
    dict(M=200, C=640, tile=(64, 64, 0), mode="geglu", ctile=(3128, 640, 0), csame=True),     # K = 640
    dict(M=136, C=640, tile=(128, 128, 0), mode="qkv", ctile=(3128, 480, 0), csame=True),
    dict(M=520, C=320, tile=(128, 80, 0), mode="dense", ctile=(3128, 160, 0), csame=True),    # attn2.to_q
    dict(M=260, C=640, tile=(64, 64, 0), mode="dense", ctile=(3128, 320, 0), csame=True),
    dict(M=130, C=1280, tile=(64, 64, 14), mode="dense", ctile=(3128, 320, 0), csame=True),   # K = 1280: not eligible, runs on the 128x64 tile
    dict(M=200, C=320, tile=(3128, 320, 0), mode="dense", ctile=(3128, 160, 0), csame=True),  # a producer (ln_out) asked for more than 128 columns per workgroup: 128x64 tile, 5 partials
    # producer on the row-panel kernel's residual / LayerNorm-producer form (64 or 128 columns per workgroup); `psame`: output
    # AND row-moment partials bit-identical to the 128x64 tile kernel's
    dict(M=600, C=320, tile=(3128, 64, 0), mode="dense", ctile=(3128, 160, 0), csame=True, psame=True),    # attn1.to_out -> attn2.to_q
    dict(M=300, C=320, tile=(3128, 128, 0), mode="geglu", ctile=(3128, 320, 0), csame=True, psame=True),   # 128 + 128 + 64 column shares; attn2.to_out -> GEGLU
    dict(M=260, C=640, tile=(3128, 128, 0), mode="qkv", ctile=(3128, 480, 0), csame=True, psame=True),     # K = 640; proj_in -> q|k|v
    dict(M=136, C=640, tile=(3128, 64, 0), mode="dense", psame=True, nores=True),                          # proj_in: no residual
    # producer and / or consumer on the wreg form (conv_wreg.hip: fragment-major weights straight to registers): the tile kernel's bits
    dict(M=600, C=320, tile=(4128, 64, 3), mode="dense", ctile=(4128, 64, 4), csame=True, psame=True),     # both; ragged M
    dict(M=300, C=320, tile=(4128, 64, 4), mode="geglu", ctile=(4128, 128, 3), csame=True, psame=True),    # GEGLU pairs inside a wave's two blocks
    dict(M=260, C=640, tile=(4064, 64, 4), mode="qkv", ctile=(4064, 256, 3), csame=True, psame=True),      # q | k | v^T split from 4 blocks per wave
    dict(M=136, C=1280, tile=(4128, 64, 3), mode="dense", ctile=(4256, 64, 3), psame=True, csame=True, nores=True),   # 20 partials per row
    dict(M=200, C=320, tile=(4128, 128, 3), mode="dense", ctile=(4128, 128, 13), csame=True),              # 128-column producer tiles (3 partials), 8-wave consumer
    dict(M=300, C=320, tile=(4128, 64, 23), mode="geglu", ctile=(4064, 128, 23), csame=True, psame=True),  # two K tiles per stage, 5 tiles
    dict(M=136, C=1280, tile=(4064, 64, 24), mode="qkv", ctile=(4064, 256, 23), csame=True, psame=True),
    dict(M=600, C=320, tile=(4256, 128, 13), mode="geglu", ctile=(4256, 128, 14), csame=True),            # 2 x 4 wave grid: producer partials from both wave rows
    # consumer on the big form (conv_big.hip; it has no producer epilogue): the tile kernel's bits
    dict(M=600, C=320, tile=(128, 64, 0), mode="geglu", ctile=(5256, 256, 0), csame=True),
    dict(M=520, C=640, tile=(64, 64, 0), mode="qkv", ctile=(5256, 160, 0), csame=True),                   # q | k | v^T split from 5 blocks per wave
    dict(M=300, C=320, tile=(128, 80, 0), mode="dense", ctile=(5128, 256, 1), csame=True),
])
def test_conv_gemm_layer_norm_fold(gpu, case):
    """LayerNormalization folded into the GEMMs around it (diffusion_model.py:84-88 + Dense): the producer
    (Dense + residual) writes row-moment partials, the consumer normalises through its epilogue.  Reference:
    fp32 LayerNorm of the producer's bf16 output, then the Dense.  rtol 1e-2 (bf16 output) + 2 % of max."""
    from minsdtf_amd import ops, packing

    torch.manual_seed(11)
    M, C = case["M"], case["C"]
    tm, tn, stg = case["tile"]
    ctm, ctn, cstg = case.get("ctile", (0, 0, 0))
    ckw = dict(tile_m=ctm, tile_n=ctn, stages=cstg)

    def lay(w, tile_m):   # the weight image a launch reads: fragment-major for the wreg form
        return dict(w=packing.fragment_major(w), w_layout=2) if 4000 <= tile_m < 5000 else dict(w=w)

    x = bf(torch.randn(M, C))
    res = bf(torch.randn(M, C) * 2 + 0.5)                      # non-zero row means
    if case.get("nores"):
        res = torch.zeros(M, C)
    w0 = bf(torch.randn(C, C) / math.sqrt(C))
    b0 = torch.randn(C) * 0.1
    t = bf(x @ w0 + b0 + res)                                  # what the producer stores (bf16)
    gamma, beta = 1 + 0.3 * torch.randn(C), 0.2 * torch.randn(C)
    ln = F.layer_norm(t, (C,), gamma, beta, eps=1e-5)
    d = gpu
    tdev = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=d)
    slots = ops.conv_gemm_ln_slots(N=C, tile_n=tn, tile_m=tm)
    assert slots == -(-C // (64 if 3000 <= tm < 4000 else tn))
    stats = torch.full((M, slots, 2), float("nan"), dtype=torch.float32, device=d)
    keep = [x.to(torch.bfloat16).to(d), packing.pack_dense(w0.numpy(), d), b0.to(d), res.to(torch.bfloat16).to(d)]   # (Calls hold raw pointers)
    if case.get("nores"):
        keep[3] = None
    keep.append(lay(keep[1], tm))
    prod = ops.conv_gemm(a0=keep[0], **keep[4], out=tdev, batch=1, h_in=M, w_in=1, c0=C, N=C, bias=keep[2], residual=keep[3],
                         tile_m=tm, tile_n=tn, stages=stg, ln_out=stats, ln_out_slots=slots)
    if case.get("psame"):
        tdev2, stats2 = torch.full_like(tdev, float("nan")), torch.full_like(stats, float("nan"))
        run_calls([prod, ops.conv_gemm(a0=keep[0], w=keep[1], out=tdev2, batch=1, h_in=M, w_in=1, c0=C, N=C, bias=keep[2], residual=keep[3],
                                       tile_m=128, tile_n=64, ln_out=stats2, ln_out_slots=slots)])
        assert torch.equal(tdev.view(torch.int16), tdev2.view(torch.int16)), "row-panel producer changed the output bits"
        assert torch.equal(stats.view(torch.int32), stats2.view(torch.int32)), "row-panel producer changed the row-moment partials"
    mode = case["mode"]
    if mode == "geglu":
        w1 = bf(torch.randn(C, 8 * C) / math.sqrt(C))
        b1 = torch.randn(8 * C) * 0.1
        h = ln @ w1 + b1
        a, gate = h[:, :4 * C], h[:, 4 * C:]
        ref = a * 0.5 * gate * (1 + torch.tanh(gate * 0.7978845608 * (1 + 0.044715 * gate ** 2)))
        order = torch.from_numpy(packing.geglu_row_order(4 * C))
        wf, cs, cb = packing.fold_layer_norm(w1.t().contiguous()[order], b1.numpy()[order.numpy()], gamma.numpy(), beta.numpy(), d)
        out = torch.full((M, 4 * C), float("nan"), dtype=torch.bfloat16, device=d)
        wfl = lay(wf, ctm)
        cons = ops.conv_gemm(a0=tdev, **wfl, out=out, batch=1, h_in=M, w_in=1, c0=C, N=8 * C, bias=cb, act=ops.ACT_GEGLU,
                             ln_in=stats, ln_in_slots=slots, ln_colsum=cs, **ckw)
        run_calls([prod, cons])
        close(out, ref, atol=2e-2 * float(ref.abs().max()), what=str(case))
        if case.get("csame"):
            out2 = torch.full_like(out, float("nan"))
            run_calls(ops.conv_gemm(a0=tdev, w=wf, out=out2, batch=1, h_in=M, w_in=1, c0=C, N=8 * C, bias=cb, act=ops.ACT_GEGLU,
                                    ln_in=stats, ln_in_slots=slots, ln_colsum=cs))
            assert torch.equal(out.view(torch.int16), out2.view(torch.int16)), "row-panel kernel changed the bits"
    elif mode == "qkv":
        w1 = bf(torch.randn(C, 3 * C) / math.sqrt(C))
        ref = ln @ w1
        wf, cs, cb = packing.fold_layer_norm(w1.t().contiguous(), None, gamma.numpy(), beta.numpy(), d)
        q = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=d)
        k = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=d)
        sp = (M + 7) // 8 * 8
        vt = torch.zeros((1, C, sp), dtype=torch.bfloat16, device=d)
        wfl = lay(wf, ctm)
        cons = ops.conv_gemm(a0=tdev, **wfl, out=q, batch=1, h_in=M, w_in=1, c0=C, N=3 * C, bias=cb, split=(C, C, k, C, vt, sp),
                             ln_in=stats, ln_in_slots=slots, ln_colsum=cs, **ckw)
        run_calls([prod, cons])
        atol = 2e-2 * float(ref.abs().max())
        close(q, ref[:, :C], atol=atol, what=str(case) + " q")
        close(k, ref[:, C:2 * C], atol=atol, what=str(case) + " k")
        close(vt[0, :, :M].t(), ref[:, 2 * C:], atol=atol, what=str(case) + " v^T")
        if case.get("csame"):
            q2, k2, vt2 = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.zeros_like(vt)
            run_calls(ops.conv_gemm(a0=tdev, w=wf, out=q2, batch=1, h_in=M, w_in=1, c0=C, N=3 * C, bias=cb, split=(C, C, k2, C, vt2, sp),
                                    ln_in=stats, ln_in_slots=slots, ln_colsum=cs))
            for a_, b_ in ((q, q2), (k, k2), (vt, vt2)):
                assert torch.equal(a_.view(torch.int16), b_.view(torch.int16)), "row-panel kernel changed the bits"
    else:
        w1 = bf(torch.randn(C, C) / math.sqrt(C))
        b1 = torch.randn(C) * 0.1
        ref = ln @ w1 + b1
        wf, cs, cb = packing.fold_layer_norm(w1.t().contiguous(), b1.numpy(), gamma.numpy(), beta.numpy(), d)
        out = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=d)
        wfl = lay(wf, ctm)
        cons = ops.conv_gemm(a0=tdev, **wfl, out=out, batch=1, h_in=M, w_in=1, c0=C, N=C, bias=cb, ln_in=stats, ln_in_slots=slots,
                             ln_colsum=cs, **ckw)
        run_calls([prod, cons])
        close(out, ref, atol=2e-2 * float(ref.abs().max()), what=str(case))
        if case.get("csame"):
            out2 = torch.full_like(out, float("nan"))
            run_calls(ops.conv_gemm(a0=tdev, w=wf, out=out2, batch=1, h_in=M, w_in=1, c0=C, N=C, bias=cb, ln_in=stats, ln_in_slots=slots,
                                    ln_colsum=cs))
            assert torch.equal(out.view(torch.int16), out2.view(torch.int16)), "row-panel kernel changed the bits"
    close(tdev, t, what=str(case) + " producer output")
    # the partials themselves: sum / sum of squares of the stored bf16 values
    tt = tdev.float().cpu()
    st = stats.cpu()
    assert torch.allclose(st[:, :, 0].sum(1), tt.sum(1), rtol=1e-4, atol=1e-2)
    assert torch.allclose(st[:, :, 1].sum(1), (tt * tt).sum(1), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("nq", [1, 0])
def test_conv_gemm_qkv_split(gpu, nq):
    """q|k|v^T epilogue (nq=1) and the k|v^T form used for the text context (nq=0)."""
    from minsdtf_amd import ops, packing

    torch.manual_seed(3)
    B, S, Cin, C = 2, 77 if nq == 0 else 64, 128, 64
    Sp = (S + 7) // 8 * 8
    x = bf(torch.randn(B, S, Cin))
    ws = [bf(torch.randn(Cin, C) / math.sqrt(Cin)) for _ in range(2 + nq)]
    wp = packing.pack_dense_stack([w.numpy() for w in ws], gpu)
    refs = [x @ w for w in ws]
    N = C * (2 + nq)
    q = torch.full((B * S, C), float("nan"), dtype=torch.bfloat16, device=gpu)
    k = torch.full((B * S, C), float("nan"), dtype=torch.bfloat16, device=gpu)
    vt = torch.zeros((B, C, Sp), dtype=torch.bfloat16, device=gpu)
    call = ops.conv_gemm(a0=x.to(torch.bfloat16).to(gpu), w=wp, out=q, batch=B, h_in=S, w_in=1, c0=Cin, N=N,
                         split=(C * nq, C, k, C, vt, Sp), out_ld=C)
    run_calls(call)
    if nq:
        close(q.reshape(B, S, C), refs[0], what="q")
    close(k.reshape(B, S, C), refs[nq], what="k")
    close(vt[:, :, :S].permute(0, 2, 1), refs[nq + 1], what="v^T")
    assert float(vt[:, :, S:].abs().sum()) == 0.0


@pytest.mark.parametrize("case", [
    dict(B=2, mod=1, H=8, W=8, cin=4, cout=320, in_f32=True),                    # UNet conv_in, latent shared by both halves
    dict(B=2, mod=2, H=8, W=8, cin=320, cout=4, in_f32=False, out="f32"),        # UNet conv_out
    dict(B=1, mod=1, H=16, W=16, cin=128, cout=3, in_f32=False, out="u8"),       # VAE conv_out -> uint8
    dict(B=1, mod=1, H=16, W=16, cin=3, cout=16, in_f32=True, act=True),         # HintNet first conv
    dict(B=1, mod=1, H=16, W=16, cin=32, cout=96, in_f32=False, stride=2, act=True),
    dict(B=1, mod=1, H=8, W=8, cin=4, cout=4, in_f32=True, ks=1, scale=1 / 0.18215, out="f32"),  # post_quant
    dict(B=5, mod=5, H=1, W=1, cin=320, cout=1280, in_f32=True, ks=1, act=True, out="f32"),      # time-embedding dense
    dict(B=25, mod=25, H=1, W=1, cin=1280, cout=1280, in_f32=True, ks=1, act=True),               # ... 25 steps: 2 row chunks, bf16 out (feeds the MFMA projection)
    dict(B=50, mod=50, H=1, W=1, cin=320, cout=1344, in_f32=True, ks=1, out="f32"),               # ... 50 steps, a last 64-column block that is cut
    dict(B=1, mod=1, H=8, W=8, cin=4, cout=320, in_f32=True, resid=True),        # ControlNet conv_in + hint
    dict(B=2, mod=1, H=64, W=64, cin=4, cout=320, in_f32=True, resid=True),      # ... at the real size: 32-pixel row segments
    dict(B=3, mod=3, H=24, W=48, cin=4, cout=512, in_f32=True, scale=1 / 0.18215),  # VAE decoder.conv_in: 16-pixel segments, 2 px groups
    dict(B=1, mod=1, H=5, W=8, cin=4, cout=64, in_f32=True),                     # 8-pixel segments, 16 pixel groups
    dict(B=1, mod=1, H=6, W=6, cin=4, cout=320, in_f32=True),                    # width not a multiple of 8: generic kernel
])
def test_conv_direct(gpu, case):
    from minsdtf_amd import ops

    torch.manual_seed(4)
    B, mod, H, W, cin, cout = case["B"], case["mod"], case["H"], case["W"], case["cin"], case["cout"]
    ks, stride = case.get("ks", 3), case.get("stride", 1)
    scale = case.get("scale", 1.0)
    x = torch.randn(mod, H, W, cin)
    if not case["in_f32"]:
        x = bf(x)
    w = torch.randn(ks, ks, cin, cout) / math.sqrt(ks * ks * cin)
    b = torch.randn(cout) * 0.1
    xb = x[torch.arange(B) % mod]
    ref = conv_ref(xb * scale, w, b, stride=stride, pad=1 if ks == 3 else 0)
    if case.get("act"):
        ref = ref * torch.sigmoid(ref)
    Ho, Wo = ref.shape[1], ref.shape[2]
    resid = None
    if case.get("resid"):
        resid = bf(torch.randn(B, Ho, Wo, cout))
        ref = ref + resid
    outk = case.get("out", "bf16")
    dt = {"bf16": torch.bfloat16, "f32": torch.float32, "u8": torch.uint8}[outk]
    out = torch.zeros((B, Ho, Wo, cout), dtype=dt, device=gpu)
    xd = x.to(gpu) if case["in_f32"] else x.to(torch.bfloat16).to(gpu)
    call = ops.conv_direct(x=xd, w=w.to(gpu), bias=b.to(gpu), out=out, batch=B, in_batch_mod=mod, h_in=H, w_in=W, c_in=cin,
                           c_out=cout, ksize=ks, stride=stride, in_dtype=ops.OUT_F32 if case["in_f32"] else ops.OUT_BF16,
                           out_dtype={"bf16": ops.OUT_BF16, "f32": ops.OUT_F32, "u8": ops.OUT_U8}[outk],
                           act=ops.ACT_SILU if case.get("act") else ops.ACT_NONE, in_scale=scale,
                           residual=None if resid is None else resid.to(torch.bfloat16).to(gpu))
    run_calls(call)
    if outk == "u8":
        refu = np.clip(((ref.numpy() + 1.0) * 0.5) * 255.0, 0, 255)
        got = out.cpu().numpy().astype(np.int32)
        # truncation: allow off-by-one where the fp32 value sits on an integer boundary
        assert np.all(np.abs(got - np.floor(refu)) <= 1), "uint8 conversion"
        assert np.mean(got == np.floor(refu).astype(np.int32)) > 0.995
    elif outk == "f32":
        close(out, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()), what=str(case))
    else:
        close(out, ref, what=str(case))


@pytest.mark.parametrize("case", [
    dict(B=2, hw=64, c0=320, silu=True),
    dict(B=2, hw=256, c0=1280, c1=640, silu=True),     # concat, groups straddle the two tensors
    dict(B=1, hw=64, c0=1280, c1=1280, silu=True),     # C=2560: two channel vectors per thread
    dict(B=1, hw=4096, c0=128, silu=False),            # VAE-like: few channels, many pixels
    dict(B=3, hw=100, c0=512, silu=True),
    dict(B=2, hw=65536, c0=128, silu=True),            # > 64 chunks: separate ordered finalize launch
    dict(B=2, hw=3264, c0=320, silu=True),             # single-launch kernel: 4-byte units, 16 per thread
    dict(B=1, hw=3000, c0=320, c1=320, silu=True),     # 8-byte units, 15 per thread, ragged last pass
    dict(B=2, hw=1024, c0=640, c1=320, silu=True),     # C=960: group 21 straddles x0|x1, 4-byte units
    dict(B=1, hw=1024, c0=1280, c1=640, silu=False),   # C=1920: 8-byte units, 16 per thread
    dict(B=2, hw=1000, c0=1280, silu=True),            # 16-byte units, ragged last pass
    dict(B=1, hw=4096, c0=640, c1=320, silu=True),     # slab too big for registers -> three-launch path
    dict(B=2, hw=4096, c0=320, silu=True),             # the UNet's 64x64 level: 1024-thread stats / apply workgroups
    dict(B=1, hw=4100, c0=320, c1=320, silu=False),    # same path, concat, ragged last chunk
    dict(B=2, hw=256, c0=1280, c1=640, silu=True, impl=0),   # three-launch path forced on a small tensor
    dict(B=2, hw=64, c0=320, silu=True, impl=0),
])
def test_group_norm(gpu, case):
    from minsdtf_amd import _lib, ops

    _lib.load().msd_set_option(b"gn_impl", case.get("impl", 1))

    torch.manual_seed(5)
    B, hw, c0, c1 = case["B"], case["hw"], case["c0"], case.get("c1", 0)
    C = c0 + c1
    x0 = bf(torch.randn(B, hw, c0) * 2 + 0.7)
    x1 = bf(torch.randn(B, hw, c1) - 0.3) if c1 else None
    x = torch.cat([x0, x1], -1) if c1 else x0
    gamma, beta = torch.randn(C) * 0.2 + 1, torch.randn(C) * 0.2
    ref = F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, eps=1e-5).permute(0, 2, 1)
    if case["silu"]:
        ref = ref * torch.sigmoid(ref)
    out = torch.full((B, hw, C), float("nan"), dtype=torch.bfloat16, device=gpu)
    stats = torch.full((B * 64,), float("nan"), dtype=torch.float32, device=gpu)
    partials = torch.full((B * ops.GN_MAX_CHUNKS * 64,), float("nan"), dtype=torch.float32, device=gpu)
    call = ops.group_norm(partials=partials, x0=x0.to(torch.bfloat16).to(gpu), x1=None if x1 is None else x1.to(torch.bfloat16).to(gpu),
                          gamma=gamma.to(gpu), beta=beta.to(gpu), stats=stats, out=out, batch=B, hw=hw, c0=c0, c1=c1,
                          silu=case["silu"])
    try:
        run_calls(call)
    finally:
        _lib.load().msd_set_option(b"gn_impl", 1)
    close(out, ref, atol=2e-2, what=str(case))
    # {mean, rstd} per (sample, group) are part of the contract on both paths
    xs = x.reshape(B, hw, 32, C // 32).permute(0, 2, 1, 3).reshape(B, 32, -1)
    st = stats.cpu().reshape(B, 32, 2)
    np.testing.assert_allclose(st[..., 0].numpy(), xs.mean(-1).numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(st[..., 1].numpy(), (xs.var(-1, unbiased=False) + 1e-5).rsqrt().numpy(), rtol=1e-3)
    out2 = torch.full_like(out, float("nan"))
    call2 = ops.group_norm(partials=partials, x0=x0.to(torch.bfloat16).to(gpu), x1=None if x1 is None else x1.to(torch.bfloat16).to(gpu),
                           gamma=gamma.to(gpu), beta=beta.to(gpu), stats=stats, out=out2, batch=B, hw=hw, c0=c0, c1=c1,
                           silu=case["silu"])
    run_calls(call2)
    if case.get("impl", 1) == 1:
        assert torch.equal(out.view(torch.int16), out2.view(torch.int16))   # bit-reproducible run to run


@pytest.mark.parametrize("case", [
    dict(B=2, hw=4096, c0=320, silu=True),               # the UNet's 64x64 level: 4 parts, 4-byte units
    dict(B=1, hw=4096, c0=640, c1=320, silu=True),       # C = 960: 16 units per thread, a group straddles x0 | x1
    dict(B=1, hw=4100, c0=320, c1=320, silu=False),      # ragged last part, 8-byte units
    dict(B=2, hw=9216, c0=320, silu=True),               # 768x768 (BASELINE config 4): 8 parts
    dict(B=1, hw=16384, c0=512, silu=True),              # VAE 128x128 stage: 16-byte units, 8 parts
    dict(B=3, hw=2048, c0=1280, silu=False),             # 2 parts
    dict(B=2, hw=1024, c0=640, silu=True, opt=256),      # the 32x32 level on 4 parts (gn_cluster = 256 pixels per part)
    dict(B=2, hw=256, c0=1280, silu=True, opt=128),      # ... and the 16x16 level on 2
])
@pytest.mark.parametrize("form", ["groups", "rows"])
def test_group_norm_cluster(gpu, case, form):
    """GroupNorm as ONE launch in which 2 / 4 / 8 workgroups share a (sample, group) slab and exchange partial moments through the
    caller's sync block (norm.hip gn_cluster_kernel), or - form "rows", gn_rows_kernel - 4-64 workgroups share a SAMPLE by pixel range,
    all channels each (the threshold is lowered so that every case whose parts fit in registers takes it; the 128x128 x 512 case does
    not and stays on the group form): against fp32 group_norm; the same bits on every repetition (each launch
    runs under a new epoch of the same counters), for a sample alone and inside a batch (with the SAME sync block, so the
    block's layout may not depend on the batch), and with launches of another part count in between; no workgroup ever gave
    up waiting (error words stay 0)."""
    from minsdtf_amd import _lib, ops

    torch.manual_seed(15)
    B, hw, c0, c1 = case["B"], case["hw"], case["c0"], case.get("c1", 0)
    C = c0 + c1
    x0 = bf(torch.randn(B, hw, c0) * 2 + 0.7)
    x1 = bf(torch.randn(B, hw, c1) - 0.3) if c1 else None
    x = torch.cat([x0, x1], -1) if c1 else x0
    gamma, beta = torch.randn(C) * 0.2 + 1, torch.randn(C) * 0.2
    ref = F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, eps=1e-5).permute(0, 2, 1)
    if case["silu"]:
        ref = ref * torch.sigmoid(ref)
    d = gpu
    keep = [x0.to(torch.bfloat16).to(d), None if x1 is None else x1.to(torch.bfloat16).to(d), gamma.to(d), beta.to(d)]
    stats = torch.full((B * 64,), float("nan"), dtype=torch.float32, device=d)
    partials = torch.full((B * ops.GN_MAX_CHUNKS * 64,), float("nan"), dtype=torch.float32, device=d)
    sync = torch.zeros(B * ops.GN_SYNC_WORDS_PER_SAMPLE, dtype=torch.int32, device=d)

    def launch(out, b=B, xs=(keep[0], keep[1]), hw_=hw):
        return ops.group_norm(partials=partials, x0=xs[0], x1=xs[1], gamma=keep[2], beta=keep[3], stats=stats, out=out, batch=b, hw=hw_,
                              c0=c0, c1=c1, silu=case["silu"], sync=sync)

    lib = _lib.load()
    lib.msd_set_option(b"gn_cluster", case.get("opt", 256))
    lib.msd_set_option(b"gn_rows", 1 if form == "rows" else 0)
    try:
        outs = [torch.full((B, hw, C), float("nan"), dtype=torch.bfloat16, device=d) for _ in range(3)]
        run_calls([launch(o) for o in outs])                       # three epochs back to back on one stream
        counters = sync.view(-1, 64)[:, 0].clone()
        assert int(counters.max()) > 0, "the cluster form did not run"     # (tickets were taken)
        close(outs[0], ref, atol=2e-2, what=str(case))
        for o in outs[1:]:
            assert torch.equal(outs[0].view(torch.int16), o.view(torch.int16))
        xs = x.reshape(B, hw, 32, C // 32).permute(0, 2, 1, 3).reshape(B, 32, -1)
        st = stats.cpu().reshape(B, 32, 2)
        np.testing.assert_allclose(st[..., 0].numpy(), xs.mean(-1).numpy(), rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(st[..., 1].numpy(), (xs.var(-1, unbiased=False) + 1e-5).rsqrt().numpy(), rtol=1e-3)
        # another part count on the same sync block in between (half the pixels -> half the parts where parts follow the size)
        half = torch.full((B, hw // 2, C), float("nan"), dtype=torch.bfloat16, device=d)
        x0h = keep[0][:, :hw // 2].contiguous()
        x1h = None if keep[1] is None else keep[1][:, :hw // 2].contiguous()
        one = torch.full((1, hw, C), float("nan"), dtype=torch.bfloat16, device=d)
        again = torch.full((B, hw, C), float("nan"), dtype=torch.bfloat16, device=d)
        last = B - 1   # the LAST sample alone: in a batch of one it uses sample 0's counters, at another epoch than sample 0's
        x0l = keep[0][last:last + 1].contiguous()
        x1l = None if keep[1] is None else keep[1][last:last + 1].contiguous()
        run_calls([launch(half, xs=(x0h, x1h), hw_=hw // 2), launch(one, b=1, xs=(x0l, x1l)), launch(again)])
        assert bool(torch.isfinite(half.float()).all())
        assert torch.equal(one[0].view(torch.int16), outs[0][last].view(torch.int16)), "a sample's bits depend on its batch"
        assert torch.equal(again.view(torch.int16), outs[0].view(torch.int16))
        # the cluster form's workgroups dealt by XCD (norm.hip gn_cluster_kernel xsh, the default) against a group's parts on consecutive
        # workgroup ids: which workgroup works on which (sample, group, part) is placement only
        lib.msd_set_option(b"gn_xmap", 0)
        plain = torch.full((B, hw, C), float("nan"), dtype=torch.bfloat16, device=d)
        run_calls([launch(plain)])
        assert torch.equal(plain.view(torch.int16), outs[0].view(torch.int16)), "the XCD dealing of the cluster form changed bits"
        # the row-major form's parts shared by CHANNELS among 1 / 2 / 4 workgroups (norm.hip gn_rows_kernel qshift; default: from the launch's
        # size): who computes which of a part's granules is placement only
        for qv in (0, 1, 2):
            lib.msd_set_option(b"gn_rows_q", qv)
            split = torch.full((B, hw, C), float("nan"), dtype=torch.bfloat16, device=d)
            st0 = stats.clone()
            stats.fill_(float("nan"))
            run_calls([launch(split)])
            assert torch.equal(split.view(torch.int16), outs[0].view(torch.int16)), f"gn_rows_q={qv} changed bits"
            assert torch.equal(stats.view(torch.int32), st0.view(torch.int32)), f"gn_rows_q={qv} changed the statistics"
        # (error words: word 8 of the 96 group slots and of the row-major form's header slot of every sample; behind them lie granules)
        assert int(sync.view(B, -1, 64)[:, :97, 8].max()) == 0, "a workgroup gave up waiting for its group's partial moments"
        rows_ran = int(sync.view(B, -1)[:, 3 * 32 * 64].max()) > 0     # the row-major form's ticket: first word behind the 96 group slots
        rows_per_pass = 1024 // (C // 8)     # (eligible: some P in 4 .. 64 leaves a thread at most 6 pixels; the half-size launch counts too)
        fits = any(-(-(-(-n // P)) // rows_per_pass) <= 6 for n in (hw, hw // 2) for P in (4, 8, 16, 32, 64))
        assert rows_ran == (form == "rows" and fits), "which form ran"
    finally:
        lib.msd_set_option(b"gn_cluster", 256)
        lib.msd_set_option(b"gn_rows", 9216)
        lib.msd_set_option(b"gn_xmap", 1)
        lib.msd_set_option(b"gn_rows_q", -1)


@pytest.mark.parametrize("form", ["groups", "rows"])
def test_group_norm_cluster_give_up_is_loud(gpu, form):
    """The cluster GroupNorm's bounded poll (norm.hip gn_cluster_kernel; form "rows": gn_rows_kernel, whose 128 workgroups per sample add 2
    each to ONE ticket) must never end in a silently wrong image: a ticket
    counter knocked off its multiple-of-P phase makes one workgroup of a group wait for an epoch nobody publishes; it gives
    up (poll bound shortened through set_option so the test takes milliseconds), raises word [8] of the sync block, and the
    engine's once-per-job check (engine.check_gn_sync) turns that into HipExtensionError and clears the word."""
    from minsdtf_amd import _lib, engine, ops

    lib = _lib.load()
    lib.msd_set_option(b"gn_rows", 4096 if form == "rows" else 0)
    B, H, W, C = 1, 64, 64, 320
    plan = engine.Plan(gpu)
    e = engine.Emitter(plan, {"n.g": torch.ones(C, device=gpu), "n.b": torch.zeros(C, device=gpu)})
    x = plan.act(B, H, W, C)
    y = e.group_norm(x, "n", True)
    plan.finalize()
    x.buf.tensor(torch.bfloat16, (B, H * W, C)).copy_(torch.randn(B, H * W, C, device=gpu).to(torch.bfloat16))
    st = torch.cuda.current_stream().cuda_stream
    plan.run(st)
    torch.cuda.synchronize()
    engine.check_gn_sync(device=gpu)                     # a healthy launch: nothing raised
    good = y.buf.tensor(torch.bfloat16, (B, H * W, C)).clone()
    words = plan._gn_sync_buf.tensor(torch.int32, (B * ops.GN_SYNC_WORDS_PER_SAMPLE,)).view(-1, 64)
    used = torch.nonzero(words[:, 0]).flatten()
    if form == "rows":   # (behind slot 96 lie granules, not counters)
        assert int(torch.count_nonzero(words[:96, 0])) == 0 and int(words[96, 0]) == 256, "the row-major form did not run (one ticket per sample, 256 per launch)"
        used, P, knock = [96] * 6, 255, 2                # 32 parts x 4 channel quarters x 2: off by 2, the last ticket of the next launch lands in the next epoch
    else:
        assert used.numel() == 32, "the cluster form did not run (one ticket counter per group expected)"
        P = int(words[used[5], 0])                           # one launch added exactly P to the counter
        assert P in (2, 4, 8)
        knock = 1
    words[used[5], 0] += knock                           # group 5: the P tickets of the next launch straddle two epochs
    lib.msd_set_option(b"gn_poll_limit", 256)
    try:
        plan.run(st)
        torch.cuda.synchronize()
        assert int(words[0, engine.GN_GIVE_UP_WORD]) == 1 and int(words[used[5], engine.GN_GIVE_UP_WORD]) == 1
        epoch = engine.GN_EPOCH
        with pytest.raises(_lib.HipExtensionError, match="gave up"):
            engine.check_gn_sync(device=gpu)
        assert int(words[0, engine.GN_GIVE_UP_WORD]) == 0        # reported once, then cleared
        engine.check_gn_sync(device=gpu)
        # the in-process answer: plan caches are retired (they key on the epoch) and a plan recorded from now on does not use the
        # cluster kernel, so the caller's retry cannot meet the same hazard
        assert engine.GN_EPOCH == epoch + 1
        plan2 = engine.Plan(gpu)
        e2 = engine.Emitter(plan2, {"n.g": torch.ones(C, device=gpu), "n.b": torch.zeros(C, device=gpu)})
        x2 = plan2.act(B, H, W, C)
        y2 = e2.group_norm(x2, "n", True)
        plan2.finalize()
        x2.buf.tensor(torch.bfloat16, (B, H * W, C)).copy_(x.buf.tensor(torch.bfloat16, (B, H * W, C)))
        plan2.run(st)
        torch.cuda.synchronize()
        w2 = plan2._gn_sync_buf.tensor(torch.int32, (B * ops.GN_SYNC_WORDS_PER_SAMPLE,)).view(-1, 64)
        assert int(torch.count_nonzero(w2[:, 0])) == 0, "a plan recorded after the give-up still ran the cluster kernel"
        close(y2.buf.tensor(torch.bfloat16, (B, H * W, C)).float().cpu(), good.float().cpu(), what="three-launch GroupNorm after the give-up")
        # the groups that were not disturbed still carry the right numbers
        got = y.buf.tensor(torch.bfloat16, (B, H * W, C))
        if form == "groups":
            assert torch.equal(got[..., :50].view(torch.int16), good[..., :50].view(torch.int16))
    finally:
        lib.msd_set_option(b"gn_poll_limit", 1 << 18)
        lib.msd_set_option(b"gn_cluster", 256)           # (the default: the rest of the suite runs the cluster kernel again)
        lib.msd_set_option(b"gn_rows", 9216)
        words[used[5], 0] += P - knock if form == "rows" else P - 1   # back on a multiple (the plan dies with the test anyway)


@pytest.mark.parametrize("B,S,spike", [(1, 4096, False), (2, 320, True), (1, 9216, False), (1, 64, False), (3, 200, False)])
def test_attention_d512(gpu, B, S, spike):
    """VAE AttentionBlock (layers.py:28-59): single head, d = 512, softmax(q k^T / sqrt(512)) v with the scores kept on
    chip (msd_attention head_dim 512), at the decoder's real sizes (S = 4096 at 512x512, 9216 at 768x768), a query count
    that is not a multiple of the 64-query workgroup, and forced reference-maximum moves (`spike`)."""
    from minsdtf_amd import ops

    torch.manual_seed(12)
    d = 512
    T = (S + 31) // 32 * 32 if S % 32 else S
    q, k, v = bf(torch.randn(B, S, d)), bf(torch.randn(B, T, d)), bf(torch.randn(B, T, d))
    if spike:
        k[:, T // 2 + 3] *= 5.0
        k[:, T - 5] *= 9.0
        k = bf(k)
    scale = 1.0 / math.sqrt(d)
    ref = torch.softmax((q @ k.transpose(-1, -2)) * scale, -1) @ v
    vt = v.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(gpu)
    out = torch.full((B, S, d), float("nan"), dtype=torch.bfloat16, device=gpu)
    qd, kd = q.to(torch.bfloat16).to(gpu), k.to(torch.bfloat16).to(gpu)
    run_calls(ops.attention(q=qd, k=kd, vt=vt, out=out, batch=B, heads=1, head_dim=d,
                            s=S, t=T, q_ld=d, k_ld=d, vt_ld=T, o_ld=d, scale=scale))
    close(out, ref, rtol=2e-2, atol=1.5e-2 * max(1.0, float(ref.abs().max())), what=f"d512 B={B} S={S}")
    # with a workspace the key walk is split four ways (t >= 2048, t % 128 == 0) and merged in part order (ABI 9): the fp32 answer
    # again, and the same bits for a sample whether it runs alone or in a batch (the split follows the key count only)
    wsf = 4 * B * S * (d + 2)
    ws = torch.full((wsf,), float("nan"), dtype=torch.float32, device=gpu)
    out2 = torch.full((B, S, d), float("nan"), dtype=torch.bfloat16, device=gpu)
    run_calls(ops.attention(q=qd, k=kd, vt=vt, out=out2, batch=B, heads=1, head_dim=d, s=S, t=T, q_ld=d, k_ld=d, vt_ld=T, o_ld=d, scale=scale,
                            workspace=ws, workspace_floats=wsf))
    close(out2, ref, rtol=2e-2, atol=1.5e-2 * max(1.0, float(ref.abs().max())), what=f"d512 split B={B} S={S}")
    split_ran = T >= 2048 and T % 128 == 0
    assert bool(torch.isnan(ws).all()) != split_ran, "the key split must run exactly when t >= 2048 and t % 128 == 0"
    if not split_ran:
        assert torch.equal(out.view(torch.int16), out2.view(torch.int16))
    if split_ran and B == 1:
        q3 = torch.cat([qd, torch.randn(2, S, d, device=gpu).to(torch.bfloat16)])
        k3 = torch.cat([kd, torch.randn(2, T, d, device=gpu).to(torch.bfloat16)])
        vt3 = torch.cat([vt, torch.randn(2, d, T, device=gpu).to(torch.bfloat16)])
        ws3 = torch.empty(3 * wsf, dtype=torch.float32, device=gpu)
        out3 = torch.full((3, S, d), float("nan"), dtype=torch.bfloat16, device=gpu)
        run_calls(ops.attention(q=q3, k=k3, vt=vt3, out=out3, batch=3, heads=1, head_dim=d, s=S, t=T, q_ld=d, k_ld=d, vt_ld=T, o_ld=d, scale=scale,
                                workspace=ws3, workspace_floats=3 * wsf))
        assert torch.equal(out3[0].view(torch.int16), out2[0].view(torch.int16)), "the split result of a sample depends on its batch"


@pytest.mark.parametrize("rows,c", [(256, 320), (100, 640), (64, 1280), (7, 2048)])
def test_layer_norm(gpu, rows, c):
    from minsdtf_amd import ops

    torch.manual_seed(6)
    x = bf(torch.randn(rows, c) * 3 + 1)
    gamma, beta = torch.randn(c) * 0.2 + 1, torch.randn(c) * 0.2
    ref = F.layer_norm(x, (c,), gamma, beta, eps=1e-5)
    out = torch.full((rows, c), float("nan"), dtype=torch.bfloat16, device=gpu)
    run_calls(ops.layer_norm(x=x.to(torch.bfloat16).to(gpu), gamma=gamma.to(gpu), beta=beta.to(gpu), out=out, rows=rows, c=c))
    close(out, ref, atol=2e-2, what=f"ln {rows}x{c}")


def test_attention_partial_round_split(gpu):
    """256-query workgroups whose last round would be at most half full hand that round's queries to a second launch of
    64-query workgroups (attention.hip, msd_attention): scheduling only — the bits of the one-launch form, and the fp32 answer."""
    from minsdtf_amd import _lib, ops

    torch.manual_seed(11)
    B, H, d, S, T = 8, 8, 40, 1280, 1280     # 5 x 64 = 320 workgroups of 256 queries on 256 CUs: 64 in the second round
    C = H * d
    scale = d ** -0.5
    q = bf(torch.randn(B, S, C) * (scale * 1.4426950408889634))
    k, v = bf(torch.randn(B, T, C)), bf(torch.randn(B, T, C))
    k[:, T - 7] *= 10.0   # a late reference-maximum move inside the tail range's walk too
    k = bf(k)
    qh = q.view(B, S, H, d).permute(0, 2, 1, 3)
    kh = k.view(B, T, H, d).permute(0, 2, 1, 3)
    vh = v.view(B, T, H, d).permute(0, 2, 1, 3)
    ref = (torch.softmax((qh @ kh.transpose(-1, -2)) * math.log(2.0), -1) @ vh).permute(0, 2, 1, 3).reshape(B, S, C)
    qd, kd = q.to(torch.bfloat16).to(gpu), k.to(torch.bfloat16).to(gpu)
    vt = v.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(gpu)
    outs = []
    for qf in (0, 4):   # automatic (splits the partial round) / 256 queries forced (one launch)
        out = torch.full((B, S, C), float("nan"), dtype=torch.bfloat16, device=gpu)
        call = ops.attention(q=qd, k=kd, vt=vt, out=out, batch=B, heads=H, head_dim=d, s=S, t=T, q_ld=C, k_ld=C, vt_ld=T, o_ld=C,
                             scale=scale, q_prescaled=True)
        _lib.load().msd_set_option(b"attn_qf", qf)
        try:
            run_calls(call)
        finally:
            _lib.load().msd_set_option(b"attn_qf", 0)
        outs.append(out)
    close(outs[0], ref, rtol=2e-2, atol=1.5e-2 * max(1.0, float(ref.abs().max())), what="partial-round split")
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), "the two-launch form changed the bits"


ATTENTION_CASES = [
    dict(B=2, H=8, d=40, S=256, T=256),
    dict(B=1, H=8, d=40, S=1024, T=1024),
    dict(B=2, H=8, d=80, S=128, T=77),       # text cross-attention: ragged key tail
    dict(B=1, H=8, d=160, S=64, T=64),       # 8x8 level: half-empty query tile
    dict(B=2, H=8, d=160, S=256, T=154),     # long prompt (2 x 77)
    dict(B=1, H=2, d=40, S=72, T=72),        # ragged queries
    dict(B=1, H=8, d=80, S=256, T=256, spike=True),  # forces online-softmax rescales
    dict(B=1, H=4, d=40, S=200, T=520, spike=True),  # ... with a ragged last tile, d = 40 (ones-row denominator)
    dict(B=1, H=2, d=160, S=64, T=320, spike=True, ramp=True),  # reference maximum creeps up below the threshold, then jumps
    dict(B=1, H=2, d=40, S=4096, T=4096, spike=True),  # the 64x64 level's real size (the most expensive kernel of a step); late spikes
    dict(B=1, H=1, d=40, S=9216, T=9216, spike=True),  # 768x768 (BASELINE config 4): 144 key tiles, reference moves late in the walk
    dict(B=1, H=1, d=40, S=4090, T=4090),              # long walk ending in a ragged query tile AND a ragged key tile
    dict(B=1, H=2, d=80, S=64, T=257, spike=True),     # software-pipelined form: shortest walk it takes + a 1-key ragged tile; one busy wave
    dict(B=3, H=3, d=40, S=300, T=256),                # ... exactly four tiles (the ring's depth), odd batch x heads, ragged queries
]


def _attention_params():
    """(case, qf, presc, form) without duplicates: qf = 128 / 64 / 256 queries per workgroup (the library picks by grid size; all
    forced here; 256 exists in the software-pipelined d = 40 form only, on key walks of 4 tiles and more); presc = q carrying
    scale * log2(e) already (the UNet's projections) or not; form, d = 40 / 80: the 32x32x16 MFMA kernel, software-pipelined (2) or
    plain (1), or the 16x16x32 one (0) — other head sizes run the 16x16x32 kernel whatever the form says, and the long walks are
    checked on one 16x16x32 form only (the CPU reference is the expensive part), so those combinations are not generated."""
    out = []
    for case in ATTENTION_CASES:
        for form in (2, 1, 0):
            if form != 1 and (case["d"] not in (40, 80) or (form == 0 and case["S"] > 2048)):
                continue
            for presc in (False, True):
                for qf in (2, 1, 4):
                    if qf == 4 and not (form == 2 and case["d"] == 40 and case["T"] >= 256):
                        continue
                    out.append(pytest.param(case, qf, presc, form, id=f"form{form}-presc{int(presc)}-qf{qf}-" +
                                            "-".join(f"{k}{int(v)}" for k, v in case.items())))
    return out


@pytest.mark.parametrize("case,qf,presc,form", _attention_params())
def test_attention(gpu, case, qf, presc, form):
    """The lazy rescale (attention.hip ATTN_THR) is a rare data-dependent branch: the `spike` cases force it at chosen
    tiles (one key row scaled so the tile maximum jumps far past the threshold), `ramp` makes the maximum grow by
    less than the threshold over several tiles first (the deferred path), and the reference is the full fp32 softmax."""
    from minsdtf_amd import _lib, ops

    torch.manual_seed(7)
    B, H, d, S, T = case["B"], case["H"], case["d"], case["S"], case["T"]
    C = H * d
    Tp = (T + 7) // 8 * 8
    q, k, v = bf(torch.randn(B, S, C)), bf(torch.randn(B, T, C)), bf(torch.randn(B, T, C))
    if case.get("ramp"):
        k *= (1.0 + 0.5 * torch.arange(T).float() / T)[None, :, None]
    if case.get("spike"):
        # later keys score far higher than earlier ones for some queries -> the running max jumps
        k[:, T // 2 + 3] *= 6.0
        k[:, T - 5] *= 12.0
        k = bf(k)
    scale = d ** -0.5
    if presc:   # what the prescaled projection delivers: bf16(q * scale * log2 e); the kernel computes exp2(q' k^T)
        q = bf(q * (scale * 1.4426950408889634))
        score_scale = math.log(2.0)
    else:
        score_scale = scale
    qh = q.view(B, S, H, d).permute(0, 2, 1, 3)
    kh = k.view(B, T, H, d).permute(0, 2, 1, 3)
    vh = v.view(B, T, H, d).permute(0, 2, 1, 3)
    ref = (torch.softmax((qh @ kh.transpose(-1, -2)) * score_scale, -1) @ vh).permute(0, 2, 1, 3).reshape(B, S, C)
    # q lives inside a wider fused buffer (leading dimension 3C) like the QKV GEMM output would
    qbuf = torch.zeros(B, S, 3 * C, dtype=torch.bfloat16, device=gpu)
    qbuf[:, :, C:2 * C] = q.to(torch.bfloat16).to(gpu)
    vt = torch.zeros(B, C, Tp, dtype=torch.bfloat16, device=gpu)
    vt[:, :, :T] = v.permute(0, 2, 1).to(torch.bfloat16).to(gpu)
    out = torch.full((B, S, C), float("nan"), dtype=torch.bfloat16, device=gpu)
    kd = k.to(torch.bfloat16).to(gpu)
    call = ops.attention(q=qbuf.data_ptr() + 2 * C, k=kd, vt=vt, out=out, batch=B, heads=H,
                         head_dim=d, s=S, t=T, q_ld=3 * C, k_ld=C, vt_ld=Tp, o_ld=C, scale=scale, q_prescaled=presc)
    _lib.load().msd_set_option(b"attn_qf", qf)
    _lib.load().msd_set_option(b"attn_form", form)
    try:
        run_calls(call)
    finally:
        _lib.load().msd_set_option(b"attn_qf", 0)
        _lib.load().msd_set_option(b"attn_form", 2)
    # P is rounded to bf16 before the PV product (relative 2^-9 per term): the error scales with the
    # magnitude of the summed terms, so the absolute floor is 1.5e-2 of max(1, max|O|)
    close(out, ref, rtol=2e-2, atol=1.5e-2 * max(1.0, float(ref.abs().max())), what=f"{case} qf={qf} presc={presc} form={form}")


@pytest.mark.parametrize("case", [
    dict(B=2, S=256, d=40, T=77, slots=5),      # the UNet's 64x64-level shape family (C = 320), text context
    dict(B=1, S=200, d=40, T=77, slots=4),      # ragged query tile
    dict(B=2, S=128, d=80, T=77, slots=10),     # 32x32 level (C = 640)
    dict(B=1, S=64, d=80, T=96, slots=5),       # the largest context the kernel takes
    dict(B=1, S=64, d=40, T=13, slots=3, spike=True),   # short context, one dominant key
    dict(B=2, S=256, d=160, T=77, slots=10),    # 16x16 level (C = 1280): the streamed-projection form (xattn_q160_kernel)
    dict(B=2, S=64, d=160, T=77, slots=20),     # the 8x8 mid block, the most partial slots a producer leaves
    dict(B=1, S=100, d=160, T=96, slots=5),     # ragged query tile, the largest context
    dict(B=3, S=64, d=160, T=13, slots=3, spike=True),
])
@pytest.mark.parametrize("layout,nw", [(1, 4), (0, 4), (1, 8)])   # weight layout; 64 / 128 queries per workgroup (picked by grid size: both forced)
def test_cross_attention_q(gpu, case, layout, nw):
    """msd_cross_attention_q: attn2.to_q (LayerNorm folded in) + attention over the text context in one launch, against the
    fp32 reference LayerNorm -> Dense -> softmax(q k^T) v and against the two launches it replaces."""
    from minsdtf_amd import _lib, ops, packing

    torch.manual_seed(17)
    B, S, d, T, slots = case["B"], case["S"], case["d"], case["T"], case["slots"]
    H, C = 8, 8 * case["d"]
    Tp = (T + 7) // 8 * 8
    x = bf(torch.randn(B * S, C) * 1.5 + 0.3)
    gamma, beta = 1 + 0.3 * torch.randn(C), 0.2 * torch.randn(C)
    wq = bf(torch.randn(C, C) / math.sqrt(C))            # (in, out)
    k, v = bf(torch.randn(B, T, C)), bf(torch.randn(B, T, C))
    if case.get("spike"):
        k[:, T // 2] *= 8.0
        k = bf(k)
    c2 = (d ** -0.5) * 1.4426950408889634                 # folded into the projection, as models._q_prescale does
    ln = F.layer_norm(x, (C,), gamma, beta, eps=1e-5)
    q = bf(ln @ (wq * c2))
    qh = q.view(B, S, H, d).permute(0, 2, 1, 3)
    kh = k.view(B, T, H, d).permute(0, 2, 1, 3)
    vh = v.view(B, T, H, d).permute(0, 2, 1, 3)
    ref = (torch.softmax((qh @ kh.transpose(-1, -2)) * math.log(2.0), -1) @ vh).permute(0, 2, 1, 3).reshape(B * S, C)

    dev = gpu
    wf, cs, cb = packing.fold_layer_norm((wq * c2).t().contiguous(), None, gamma.numpy(), beta.numpy(), dev)
    wdev = packing.chunk_major(wf) if layout else wf
    # row-moment partials as a producing GEMM would leave them: (sum, sum of squares) of the stored bf16 row per column group
    edges = [round(i * C / slots) for i in range(slots + 1)]
    st = torch.stack([torch.stack([x[:, a:b_].sum(1), (x[:, a:b_] ** 2).sum(1)], -1) for a, b_ in zip(edges[:-1], edges[1:])], 1)
    stats = st.to(torch.float32).contiguous().to(dev)
    xd, kd = x.to(torch.bfloat16).to(dev), k.to(torch.bfloat16).to(dev)
    vt = torch.full((B, C, Tp), float("nan"), dtype=torch.bfloat16, device=dev)   # (padding keys hold garbage, as in the arena)
    vt[:, :, :T] = v.permute(0, 2, 1).to(torch.bfloat16).to(dev)
    out = torch.full((B * S, C), float("nan"), dtype=torch.bfloat16, device=dev)
    _lib.load().msd_set_option(b"xattn_nw", nw)
    try:
        run_calls(ops.cross_attention_q(x=xd, ln_in=stats, ln_in_slots=slots, wq=wdev, ln_colsum=cs, bias=cb, k=kd, vt=vt, out=out, batch=B,
                                        heads=H, head_dim=d, s=S, t=T, k_ld=C, vt_ld=Tp, o_ld=C, w_layout=layout))
    finally:
        _lib.load().msd_set_option(b"xattn_nw", 0)
    tol = dict(rtol=2e-2, atol=1.5e-2 * max(1.0, float(ref.abs().max())))
    close(out, ref, what=f"{case} fused", **tol)
    # the two launches it replaces
    q2 = torch.full((B * S, C), float("nan"), dtype=torch.bfloat16, device=dev)
    out2 = torch.full_like(out, float("nan"))
    run_calls([ops.conv_gemm(a0=xd, w=wdev, out=q2, batch=1, h_in=B * S, w_in=1, c0=C, N=C, bias=cb, ln_in=stats, ln_in_slots=slots,
                             ln_colsum=cs, w_layout=layout),
               ops.attention(q=q2, k=kd, vt=vt, out=out2, batch=B, heads=H, head_dim=d, s=S, t=T, q_ld=C, k_ld=C, vt_ld=Tp, o_ld=C,
                             scale=d ** -0.5, q_prescaled=True)])
    close(out2, ref, what=f"{case} two launches", **tol)
    close(out, out2.float().cpu(), what=f"{case} fused vs two launches", **tol)


def test_softmax_rows(gpu):
    from minsdtf_amd import ops

    torch.manual_seed(8)
    rows, cols = 50, 4096
    x = torch.randn(rows, cols) * 20
    scale = 1 / math.sqrt(512)
    ref = torch.softmax(x * scale, -1)
    out = torch.zeros(rows, cols, dtype=torch.bfloat16, device=gpu)
    run_calls(ops.softmax_rows(x=x.to(gpu), out=out, rows=rows, cols=cols, ld_in=cols, ld_out=cols, scale=scale))
    close(out, ref, rtol=1e-2, atol=1e-6, what="softmax")


@pytest.mark.parametrize("guidance,rescale", [(7.5, 0.7), (7.5, 0.0), (0.0, 0.0)])
@pytest.mark.parametrize("advance", [2, 1])   # the step counter moved inside the launch (by the last workgroup), or by a second launch
@pytest.mark.parametrize("hw", [8, 64, 96, 100])   # 8x8 latent (most threads idle), 64x64 / 96x96 (the register forms' full sizes), 100x100 (the re-reading loop)
def test_cfg_step(gpu, guidance, rescale, advance, hw):
    """CFG + rescale + sampler step against the oracle restatement, for every step of a 5-step run (batch 3: three
    workgroups race for the last ticket of the in-kernel advance)."""
    from minsdtf_amd import ops
    from minsdtf_amd.scheduler import Scheduler
    from oracle import sd_oracle as O

    rng = np.random.default_rng(9)
    B, n, steps = 3, hw * hw * 4, 5
    sch = Scheduler()
    sch.set_timesteps(steps)
    coef = torch.from_numpy(sch.coefficient_table()).to(gpu)
    osch = O.OracleScheduler()
    osch.set_timesteps(steps)
    lat = rng.standard_normal((B, hw, hw, 4)).astype(np.float32)
    lat_d = torch.from_numpy(lat.reshape(B, n).copy()).to(gpu)
    step = torch.zeros(2, dtype=torch.int32, device=gpu)   # {step, ticket}
    ref = lat.astype(np.float64)
    for i, t in enumerate(osch.timesteps):
        u = rng.standard_normal((B, hw, hw, 4)).astype(np.float32)
        c = (u + 0.3 * rng.standard_normal((B, hw, hw, 4))).astype(np.float32)
        if guidance > 0:
            e = u + guidance * (c - u)
            if rescale > 0:
                e = O.rescale_noise_cfg(e, c, rescale)
            eps_d = torch.from_numpy(np.concatenate([u, c]).reshape(2 * B, n)).to(gpu)
        else:
            e = c
            eps_d = torch.from_numpy(c.reshape(B, n)).to(gpu)
        ref = osch.step(e, int(t), ref)
        run_calls(ops.cfg_step(eps=eps_d, latent=lat_d, coef=coef, step_ptr=step, batch=B, n=n, num_steps=steps,
                               guidance=guidance, guidance_rescale=rescale, advance=advance))
        got = lat_d.cpu().numpy().reshape(B, hw, hw, 4)
        # fp32 device math vs the reference's float64 numpy path
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4 * np.abs(ref).max(), err_msg=f"step {i}")
        assert step.tolist() == [i + 1, 0]


def test_elementwise(gpu):
    from minsdtf_amd import ops

    torch.manual_seed(10)
    a, b = bf(torch.randn(4096)), bf(torch.randn(4096))
    out = torch.zeros(4096, dtype=torch.bfloat16, device=gpu)
    run_calls(ops.add_bf16(a=a.to(torch.bfloat16).to(gpu), b=b.to(torch.bfloat16).to(gpu), out=out, n=4096))
    assert torch.equal(out.cpu(), (a + b).to(torch.bfloat16))
    x = torch.randn(1000)
    o = torch.zeros(1000, dtype=torch.bfloat16, device=gpu)
    run_calls(ops.cast_f32_to_bf16(x=x.to(gpu), out=o, n=1000))
    assert torch.equal(o.cpu(), x.to(torch.bfloat16))
    o2 = torch.zeros(1000, dtype=torch.float32, device=gpu)
    run_calls(ops.cast_bf16_to_f32(x=o, out=o2, n=1000))
    assert torch.equal(o2.cpu(), x.to(torch.bfloat16).float())
    z = torch.ones(1024, dtype=torch.float32, device=gpu)
    run_calls(ops.memset_zero(ptr=z, nbytes=4096))
    assert float(z.abs().sum()) == 0


def test_argument_errors(gpu):
    """The ABI rejects bad shapes with an error instead of launching."""
    from minsdtf_amd import _lib, ops

    x = torch.zeros(1, 8, 8, 48, dtype=torch.bfloat16, device=gpu)
    w = torch.zeros(64, 48, dtype=torch.bfloat16, device=gpu)
    o = torch.zeros(64, 64, dtype=torch.bfloat16, device=gpu)
    with pytest.raises(_lib.HipExtensionError):
        run_calls(ops.conv_gemm(a0=x, w=w, out=o, batch=1, h_in=8, w_in=8, c0=48, N=64))
    # the big form on a staged halo works on whole 16 x 16-pixel output tiles of a 3x3 / stride-1 conv and takes no shortcut operand: anything else
    # is refused, never run on another kernel (its numerics class is part of the request)
    x64 = torch.zeros(1, 24, 16, 64, dtype=torch.bfloat16, device=gpu)
    w9 = torch.zeros(128, 9 * 64, dtype=torch.bfloat16, device=gpu)
    o24 = torch.zeros(24 * 16, 128, dtype=torch.bfloat16, device=gpu)
    for kw, what in [(dict(h_in=24, w_in=16), "halo-image"),                       # 24 rows: one and a half tiles
                     (dict(h_in=16, w_in=16, stride=2), "halo-image"),
                     (dict(h_in=16, w_in=16, stages=29), "halo-image"),            # no such configuration
                     (dict(h_in=16, w_in=16, tile_m=5128), "halo-image")]:
        args = dict(a0=x64, w=w9, out=o24, batch=1, c0=64, N=128, ksize=3, tile_m=5256, tile_n=128, stages=20)
        args.update(kw)
        with pytest.raises(_lib.HipExtensionError, match=what):
            run_calls(ops.conv_gemm(**args))


def test_replicate(gpu):
    """msd_replicate (ABI 11): `copies` replicas back to back, in place (src == dst: replica 0 stays) or into another buffer; a
    source that overlaps the replicas in any other way is refused."""
    from minsdtf_amd import _lib, ops

    torch.manual_seed(5)
    src = torch.randint(0, 2 ** 15, (3, 8, 8, 320), dtype=torch.int16, device=gpu)
    nbytes = src.numel() * 2
    dst = torch.full((3 * 3, 8, 8, 320), -1, dtype=torch.int16, device=gpu)
    run_calls(ops.replicate(src=src, dst=dst, nbytes=nbytes, copies=3))
    assert torch.equal(dst, src.repeat(3, 1, 1, 1))
    wide = torch.full((2 * 3, 8, 8, 320), -1, dtype=torch.int16, device=gpu)
    wide[:3] = src
    run_calls(ops.replicate(src=wide, dst=wide, nbytes=nbytes, copies=2))
    assert torch.equal(wide, src.repeat(2, 1, 1, 1))
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    assert lib.msd_replicate(wide.data_ptr() + 16, wide.data_ptr(), nbytes, 2, st) == -1    # overlapping, not in place
    assert lib.msd_replicate(src.data_ptr(), dst.data_ptr(), nbytes + 8, 3, st) == -1       # not whole 16-byte vectors
    assert lib.msd_replicate(src.data_ptr(), dst.data_ptr(), nbytes, 0, st) == -1
    torch.cuda.synchronize()


@pytest.mark.parametrize("hw,C,sync_on,rows", [(64, 1280, False, 0), (1024, 640, True, 0), (4096, 320, True, 0), (4096, 320, True, 4096), (16384, 128, False, 0)])
def test_group_norm_of_a_constant_tensor(gpu, hw, C, sync_on, rows):
    """Zero variance in every group (a constant image, and a tensor constant per channel with a large offset: E[x^2] - E[x]^2 cancels
    to rounding noise, clamped at 0): (x - mean) * rsqrt(var + eps) must stay finite and the output must be beta (+ swish) to bf16
    round-off, in every form - one workgroup per group, the cluster, the row-major parts, statistics / apply."""
    from minsdtf_amd import _lib, ops

    d = gpu
    lib = _lib.load()
    lib.msd_set_option(b"gn_rows", rows if rows else 9216)
    try:
        B = 2
        gamma, beta = (torch.randn(C) * 0.2 + 1).to(d), (torch.randn(C) * 0.2).to(d)
        stats = torch.full((B * 64,), float("nan"), dtype=torch.float32, device=d)
        partials = torch.full((B * ops.GN_MAX_CHUNKS * 64,), float("nan"), dtype=torch.float32, device=d)
        sync = torch.zeros(B * ops.GN_SYNC_WORDS_PER_SAMPLE, dtype=torch.int32, device=d) if sync_on else None
        for value in (0.0, 3.0, -117.0):
            x = torch.full((B, hw, C), value, dtype=torch.bfloat16, device=d)
            out = torch.full((B, hw, C), float("nan"), dtype=torch.bfloat16, device=d)
            run_calls(ops.group_norm(partials=partials, x0=x, gamma=gamma, beta=beta, stats=stats, out=out, batch=B, hw=hw, c0=C, silu=False, sync=sync))
            assert bool(torch.isfinite(out.float()).all()), value
            st = stats.view(B, 32, 2)
            assert float((st[..., 0] - float(x[0, 0, 0])).abs().max()) <= 1e-3 * max(1.0, abs(value))
            # var in [0, a few ulp of mean^2]: rstd <= rsqrt(eps); the output is beta up to (x - mean) * rstd * gamma, which the clamp keeps tiny
            assert float(st[..., 1].max()) <= (1e-5) ** -0.5 * 1.0001
            err = (out.float() - beta.to(torch.bfloat16).float()[None, None, :]).abs().max()
            assert float(err) <= 0.05, (value, float(err))
    finally:
        lib.msd_set_option(b"gn_rows", 9216)


@pytest.mark.parametrize("d,S,T", [(40, 4096, 4096), (40, 300, 77), (80, 1024, 1024), (160, 256, 256), (160, 64, 77), (64, 77, 77)])
@pytest.mark.parametrize("presc", [False, True])
def test_attention_of_zero_queries_is_the_mean_of_v(gpu, d, S, T, presc):
    """q = 0: every score is 0, the softmax is uniform and the output is the mean of V over the keys, for every query - whatever the
    keys hold (here: large) and with NaN in V^T's padding columns (>= t: never read).  Pins the denominator (the ones-row / row-sum
    bookkeeping of every head size's form) apart from the exponentials."""
    from minsdtf_amd import ops

    torch.manual_seed(29)
    B, H = 2, 8 if d != 64 else 12
    C = H * d
    Tp = (T + 7) // 8 * 8
    dev = gpu
    q = torch.zeros(B, S, C, dtype=torch.bfloat16, device=dev)
    k = (bf(torch.randn(B, T, C)) * 30.0).to(torch.bfloat16).to(dev)
    v = bf(torch.randn(B, T, C))
    vt = torch.full((B, C, Tp), float("nan"), dtype=torch.bfloat16, device=dev)
    vt[:, :, :T] = v.permute(0, 2, 1).to(torch.bfloat16).to(dev)
    out = torch.full((B, S, C), float("nan"), dtype=torch.bfloat16, device=dev)
    run_calls(ops.attention(q=q, k=k, vt=vt, out=out, batch=B, heads=H, head_dim=d, s=S, t=T, q_ld=C, k_ld=C, vt_ld=Tp, o_ld=C,
                            scale=d ** -0.5, q_prescaled=presc))
    ref = v.mean(dim=1, keepdim=True).expand(B, S, C)
    # (P = 1 exactly, so the only rounding is the fp32 sum over the keys and the bf16 store)
    close(out, ref, rtol=1e-2, atol=4e-3, what=f"d={d} S={S} T={T} presc={presc}")

"""Launch plans: the reference's Keras graphs re-expressed as flat lists of C-ABI calls.

A :class:`Plan` is built once per (network, batch, resolution, context length): it walks the
network topology, assigns every activation a byte range in one HBM arena (ranges are recycled as
soon as the last consumer has been recorded, which keeps the working set of a UNet forward inside
the 256 MiB Infinity Cache), and records one :class:`~minsdtf_amd.ops.Call` per kernel.  Running
a plan is a Python loop of foreign calls on one HIP stream; captured once into a hipGraph it is
replayed with no host work at all.  Nothing here computes: all arithmetic is in the HIP library.

Topology sources (reference, paths relative to stable_diffusion/):
  UNet            diffusion_model.py:166-283      ResBlock :22-51   Attentions :54-78
  TransformerBlock :81-96   CrossAttention :99-129   GEGLU :142-153   Upsamplers :132-139
  VAE decoder     image_decoder.py:22-55, layers.py:28-80
  ControlNet / HintNet   control_net.py:10-107
"""
from __future__ import annotations

import os
import weakref
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops, packing, tuning
from . import weights as wtab

EPS = 1e-5  # every normalisation layer of the reference uses epsilon=1e-5


# ----------------------------------------------------------------------------- memory
class Arena:
    """Plan-time bump/free-list allocator over one device buffer (materialised after planning)."""

    ALIGN = 256

    def __init__(self):
        self.free_list: List[Tuple[int, int]] = []  # (offset, size), sorted by offset
        self.top = 0
        self.base: Optional[int] = None
        self.storage: Optional[torch.Tensor] = None

    def alloc(self, nbytes: int) -> "Buf":
        n = (int(nbytes) + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        best = None
        for i, (off, sz) in enumerate(self.free_list):
            if sz >= n and (best is None or sz < self.free_list[best][1]):
                best = i
        if best is not None:
            off, sz = self.free_list.pop(best)
            if sz > n:
                self.free_list.insert(best, (off + n, sz - n))
            return Buf(self, off, n)
        off = self.top
        self.top += n
        return Buf(self, off, n)

    def free(self, buf: "Buf") -> None:
        if buf is None or buf.arena is not self or buf.freed:
            return
        buf.freed = True
        self.free_list.append((buf.offset, buf.nbytes))
        self.free_list.sort()
        merged: List[Tuple[int, int]] = []
        for off, sz in self.free_list:
            if merged and merged[-1][0] + merged[-1][1] == off:
                merged[-1] = (merged[-1][0], merged[-1][1] + sz)
            else:
                merged.append((off, sz))
        if merged and merged[-1][0] + merged[-1][1] == self.top:  # give the tail back
            self.top = merged[-1][0]
            merged.pop()
        self.free_list = merged

    def materialize(self, device, high_water: int) -> None:
        self.storage = torch.zeros(max(high_water, self.ALIGN), dtype=torch.uint8, device=device)
        self.base = self.storage.data_ptr()


class Buf:
    __slots__ = ("arena", "offset", "nbytes", "freed")

    def __init__(self, arena, offset, nbytes):
        self.arena, self.offset, self.nbytes, self.freed = arena, offset, nbytes, False

    @property
    def ptr(self) -> int:
        assert self.arena.base is not None, "arena not materialised"
        return self.arena.base + self.offset

    def at(self, byte_offset: int) -> "BufView":
        return BufView(self, byte_offset)

    def tensor(self, dtype, shape) -> torch.Tensor:
        """A torch view of this range (host <-> device copies at the model boundary only)."""
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        assert n <= self.nbytes
        return self.arena.storage[self.offset:self.offset + n].view(dtype).view(*shape)


class BufView:
    __slots__ = ("buf", "off")

    def __init__(self, buf, off):
        self.buf, self.off = buf, off

    @property
    def ptr(self) -> int:
        return self.buf.ptr + self.off


@dataclass
class Act:
    """bf16 NHWC activation (tokens are the same memory: [B][H*W][C])."""
    buf: Buf
    B: int
    H: int
    W: int
    C: int
    ln: Optional[tuple] = None   # (buffer, slots): row-moment partials written by the producing GEMM (LayerNorm fold)

    @property
    def M(self) -> int:
        return self.B * self.H * self.W


class Plan:
    def __init__(self, device):
        self.device = device
        self.arena = Arena()
        self.high_water = 0
        self.recs: List[Callable[[], ops.Call]] = []
        self.calls: List[ops.Call] = []
        self.gn_slots = 0
        self.gn_batch = 0
        self.ws_floats = 0
        self._ws_buf: Optional[Buf] = None
        self._stats_buf: Optional[Buf] = None
        self._gn_part_buf: Optional[Buf] = None
        self._gn_sync_buf: Optional[Buf] = None   # ticket counters + granules of the cluster GroupNorm (zero at start, this plan's only)
        self.keep: list = []  # device tensors that must outlive the plan
        self.marks: Dict[str, int] = {}

    # -- memory
    def alloc(self, nbytes: int) -> Buf:
        b = self.arena.alloc(nbytes)
        self.high_water = max(self.high_water, self.arena.top)
        return b

    def _alloc_tail(self, nbytes: int) -> Buf:
        n = (int(nbytes) + Arena.ALIGN - 1) // Arena.ALIGN * Arena.ALIGN
        b = Buf(self.arena, self.high_water, n)
        self.high_water += n
        return b

    def free(self, *bufs) -> None:
        for b in bufs:
            if isinstance(b, Act):
                b = b.buf
            if b is not None and getattr(b, "arena", self.arena) is self.arena:   # (another plan's buffer is not ours to recycle)
                self.arena.free(b)

    def mark(self, name: str) -> None:
        """Remember the position of the next recorded call under `name` (run_range: a plan split over two streams)."""
        self.marks[name] = len(self.recs)

    def act(self, B, H, W, C) -> Act:
        return Act(self.alloc(B * H * W * C * 2), B, H, W, C)

    # -- recording
    def rec(self, fn, **kw) -> None:
        self.recs.append(lambda: fn(**kw))

    @property
    def ws(self):
        return _Lazy(lambda: self._ws_buf.ptr if self._ws_buf is not None else None)

    @property
    def gn_partials(self):
        return _Lazy(lambda: self._gn_part_buf.ptr)

    @property
    def gn_sync(self):
        return _Lazy(lambda: self._gn_sync_buf.ptr)

    def gn_stats_slot(self, batch: int):
        slot = self.gn_slots
        self.gn_slots += 1
        self.gn_batch = max(self.gn_batch, batch)
        return _Lazy(lambda: self._stats_buf.ptr + slot * self.gn_batch * 64 * 4)

    def finalize(self) -> None:
        """Fix the arena, then turn the recorded closures into typed calls."""
        # scratch that is live across the whole plan goes ABOVE every recycled range
        if self.ws_floats:
            self._ws_buf = self._alloc_tail(self.ws_floats * 4)
        if self.gn_slots:
            self._stats_buf = self._alloc_tail(self.gn_slots * self.gn_batch * 64 * 4)
            self._gn_part_buf = self._alloc_tail(self.gn_batch * ops.GN_MAX_CHUNKS * 64 * 4)
            # (the arena is allocated zeroed and this range is never recycled: the cluster kernel's counters start at 0 and
            #  only its launches on this plan's stream ever touch them)
            self._gn_sync_buf = self._alloc_tail(self.gn_batch * ops.GN_SYNC_WORDS_PER_SAMPLE * 4)
        self.arena.materialize(self.device, self.high_water)
        if self._gn_sync_buf is not None and torch.device(self.device).type == "cuda":
            _gn_sync_plans.add(self)   # check_gn_sync() reads this plan's give-up word once per job
        self.calls = [r() for r in self.recs]
        self.recs = []

    def run(self, stream: int) -> None:
        self.run_range(stream, 0, None)

    def run_range(self, stream: int, lo: int, hi: Optional[int] = None) -> None:
        if not TRACE:
            for c in self.calls[lo:hi]:
                c(stream)
            return
        # $MSD_TRACE=1: a roctx range per network block (rocprofv3 --marker-trace attributes the kernel trace to
        # down_blocks.0.resnets.0, ...attentions.1, mid_block, ...; eager runs — ranges are host-side, a replayed graph has none)
        cur = None
        for c in self.calls[lo:hi]:
            blk = _block_of(c.name)
            if blk != cur:
                if cur is not None:
                    torch.cuda.nvtx.range_pop()
                torch.cuda.nvtx.range_push(blk)
                cur = blk
            c(stream)
        if cur is not None:
            torch.cuda.nvtx.range_pop()


TRACE = os.environ.get("MSD_TRACE", "0") == "1"

# ---- give-up flag of the cluster GroupNorm (include/minsdtf_hip.h, msd_group_norm) --------------------------------------
# A workgroup of gn_cluster_kernel that stops waiting for its group's partial moments sets word [8] of the plan's sync block
# and the launch ends with wrong numbers.  A launch cannot return that, so every live plan's word is read ONCE PER JOB, after
# the job's work has completed (one 4-byte word per plan, gathered by one tiny copy kernel, outside any captured graph), and
# a non-zero word raises: the pipeline never returns an image computed from abandoned moments with a clean return code.
_gn_sync_plans: "weakref.WeakSet[Plan]" = weakref.WeakSet()
GN_GIVE_UP_WORD = 8
# Bumped by every reported give-up: plan caches (StableDiffusion._engines, the models' bound plans) key on it, so the job a caller
# retries after the exception is RECORDED AGAIN - with the cluster kernel switched off (check_gn_sync sets gn_cluster = 0 for the
# rest of the process): the retry does not meet the hazard a second time.
GN_EPOCH = 0


def retire_stale(cache: dict) -> int:
    """Drop the entries of a plan cache whose key ends in another GN_EPOCH than the current one (every such cache appends
    the epoch to its keys), BEFORE the replacement is built: a retired engine / bound plan owns a device arena, a captured
    hipGraph and a sync block, and the retry after a give-up must not have to fit next to them (768x768 or batch 4 would
    run out of memory on the very call the mechanism exists for).  Returns the number of entries dropped."""
    stale = [k for k in cache if k[-1] != GN_EPOCH]
    for k in stale:
        owner = cache.pop(k)
        rel = getattr(owner, "release_graphs", None)
        if rel is not None:
            rel()
    if stale:
        import gc

        gc.collect()   # (plans and their recorded closures reference one another: free the arenas now, not at some later collection)
    return len(stale)


def gn_sync_flags(device=None) -> Optional[torch.Tensor]:
    """Device int32 tensor with the give-up word of every live plan on `device` that owns a cluster-GroupNorm sync block
    (None when there is none).  Stream-ordered: queue it behind the job, copy it with the job's D2H, hand it to
    check_gn_sync() once the copy has landed."""
    plans = _gn_plans_on(device)
    if not plans:
        return None
    words = [p._gn_sync_buf.tensor(torch.int32, (GN_GIVE_UP_WORD + 1,))[GN_GIVE_UP_WORD:] for p in plans]
    return torch.cat(words)


def _gn_plans_on(device):
    return [p for p in _gn_sync_plans if device is None or torch.device(p.device) == torch.device(device)]


def check_gn_sync(flags=None, device=None, group_wide: bool = False) -> None:
    """Raise HipExtensionError if any cluster GroupNorm launch since the last check gave up (`flags`: the HOST copy of a
    gn_sync_flags() tensor; None = read them now, synchronously).  `group_wide` (the sharded jobs: generate_image, bench.py —
    calls that every rank of the process group makes once per job): under a process group whose exchanges run
    (dist.collectives_on) the verdict is the MAXIMUM over the ranks: a rank whose slice is corrupt has already shipped it to
    every rank in the all-gather, so every rank must raise - together, or the others hang in the next job's broadcast.  Only
    the words of the plans that were read (those on `device`) are cleared, and the process stops using the cluster kernel
    (GN_EPOCH)."""
    global GN_EPOCH
    if flags is None:
        dev_flags = gn_sync_flags(device)
        flags = None if dev_flags is None else dev_flags.cpu()
    worst = 0 if flags is None else int(flags.max())
    from . import dist as mdist

    if group_wide and mdist.collectives_on():   # (every rank calls this once per job, after its D2H: a matched collective)
        import torch.distributed as tdist

        backend_cuda = tdist.get_backend() == "nccl"
        t = torch.tensor([worst], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()) if backend_cuda else "cpu")
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        worst = int(t.item())
    if worst == 0:
        return
    for p in _gn_plans_on(device):
        p._gn_sync_buf.tensor(torch.int32, (GN_GIVE_UP_WORD + 1,))[GN_GIVE_UP_WORD].zero_()
    GN_EPOCH += 1
    _lib.check(_lib.load().msd_set_option(b"gn_cluster", 0), "gn_cluster")
    raise _lib.HipExtensionError(
        "msd_group_norm: a workgroup of the cluster GroupNorm gave up waiting for its group's partial moments (sync word [8] "
        "set, on this rank or another one of the process group): this job's result is wrong and has been discarded.  The cluster "
        "kernel is now off for this process (msd_set_option('gn_cluster', 0): the one-workgroup-per-group path has no exchange) "
        "and the launch plans will be recorded again: retry the call")


def _block_of(call_name: str) -> str:
    """'down_blocks.0.attentions.1.transformer_blocks.0.attn1.qkv' -> 'down_blocks.0.attentions.1'."""
    parts = call_name.split(".")
    n = 4 if parts[0] in ("down_blocks", "up_blocks") else 3 if parts[0] == "mid_block" else 2 if parts[0] in ("decoder", "encoder") else 1
    return ".".join(parts[:n])


class _Lazy:
    """Pointer known only after the arena is materialised."""

    def __init__(self, fn):
        self.fn = fn

    @property
    def ptr(self):
        return self.fn()


def _tile_n(N: int) -> int:
    return 128 if (N % 128 == 0 or N > 1024) else 64


def pick_splitk(M: int, N: int, nk: int) -> int:
    """Spread small-M layers over the chip: aim at >= ~256 workgroups, >= 4 K-tiles per slice."""
    tiles = ((M + 127) // 128) * ((N + _tile_n(N) - 1) // _tile_n(N))
    # every split costs a second launch (slab reduction, ~5 us): only worth it for long K loops
    if tiles >= 160 or nk < 32:
        return 1
    s = min((256 + tiles - 1) // tiles, nk // 8, 16)
    return max(1, s)


# LayerNorm fold (include/minsdtf_hip.h): the three LayerNormalizations of a transformer block run inside the
# epilogues of the GEMMs around them instead of as 48 launches per UNet forward.  False = separate
# msd_layer_norm launches (A/B runs, and any weight dict without the folded tensors).
LN_FOLD = True
# ff.net.2 + proj_out of a transformer block as one GEMM over the concat [ff | t2] (weights folded at pack time)
FF_PROJ_FOLD = True
# UNet conv_out (320 -> 4, 3x3) on the MFMA path instead of the vector-FMA kernel
MFMA_CONV_OUT = True
# ResBlock with a 1x1 shortcut: conv2 and conv_shortcut as one GEMM (the shortcut's channels are extra K tiles)
SHORTCUT_FOLD = os.environ.get("MSD_SHORTCUT_FOLD", "1") != "0"   # (env switch: same-box A/B runs)
# ... below this many pixels per SAMPLE.  Rounds 2-5 stopped folding at 64x64 (4096): conv2 ran on the halo-tile kernel, which had no shortcut
# operand, and conv2 + a separate 1x1 launch beat the folded general-kernel contraction there.  Round 6: the halo-tile kernel walks the
# shortcut chunks itself (csrc/conv_halo.hip) and the fold wins at every level (in place, batch 1: 36.5 -> 29.2 us per 64x64-level block), so
# there is no limit any more; the switch stays for same-box A/B runs.  The rule reads the per-sample shape only: folding changes the order of
# a layer's sums, and a sample's bits must not depend on the batch it runs in.
SHORTCUT_FOLD_MAX_PIXELS = int(os.environ.get("MSD_SHORTCUT_FOLD_MAX_PIXELS", str(1 << 30)))
# Classifier-free guidance runs the UNet twice per step on the SAME latent and time embedding (stable_diffusion.py:454-457); the two
# forwards differ only in the text context, which first enters at down_blocks.0.attentions.0's attn2 (diffusion_model.py:191-196,88-95).
# In the fused batch (rows [0, B) unconditional, [B, 2B) conditioned) everything in front of that point - conv_in, down_blocks.0.resnets.0,
# and norm / proj_in / q|k|v / self-attention / to_out of the first transformer block, 10 % of a step's FLOP - is therefore computed ONCE per
# image and three tensors are replicated (msd_replicate).  Exact: a sample's bits do not depend on the batch it runs in, so
# the replicated rows are the rows the second computation would have produced, bit for bit (tests/test_configs_gpu.py).
SHARE_CFG_PREFIX = os.environ.get("MSD_SHARE_CFG_PREFIX", "1") != "0"   # (env switch: same-box A/B runs)
# packed msd_conv_gemm weights stored chunk-major [K/64][N][64] (packing.chunk_major) instead of [N][K] rows
W_CHUNK_MAJOR = os.environ.get("MSD_W_CHUNK_MAJOR", "1") != "0"
# the concatenated time_emb_proj Dense of the preparation plan on the MFMA path (bf16 weights and input, fp32 table)
MFMA_TEMB_PROJ = os.environ.get("MSD_MFMA_TEMB_PROJ", "1") != "0"
# attn2.to_q + the attention over the text context as one launch at the 64x64 / 32x32 levels (msd_cross_attention_q)
XATTN_FUSED = os.environ.get("MSD_XATTN_FUSED", "1") != "0"
# ... and at the 16x16 / 8x8 levels (C = 1280, d = 160: the streamed-projection form, round 5).  Off: in the replayed loop the
# one launch takes 19.9 us where the two take 8.6 + 9.9 (25-step loop 97.30 vs 97.06 ms); the kernel stays for
# tools/xattn_bench.py and its parity tests.
XATTN_FUSED_D160 = os.environ.get("MSD_XATTN_FUSED_D160", "0") != "0"


# ----------------------------------------------------------------------------- layer emitters
class Emitter:
    """Emits the calls of the reference's layer types into a Plan. `W` maps logical weight names to
    device tensors prepared by the model classes (minsdtf_amd/models.py)."""

    def __init__(self, plan: Plan, W: Dict[str, torch.Tensor], step_ptr=None):
        self.p, self.W, self.step_ptr = plan, W, step_ptr
        plan.keep.append(W)   # the calls hold raw addresses of these tensors: the plan owns a reference to them

    def w_layout(self, key: str) -> int:
        """Storage layout of weight matrix `key` as its owner recorded it (packing.PackedWeights); a plain mapping (the
        tuner's shape walk) has no tensors and takes the default."""
        lay = getattr(self.W, "layout", None)
        return lay(key) if lay is not None else (1 if W_CHUNK_MAJOR else 0)

    # conv / dense on the MFMA path; x may be a single Act or a (Act, Act) channel concat
    def conv(self, x, name, N, ksize=1, stride=1, upsample=False, act=ops.ACT_NONE, residual: Optional[Act] = None,
             rowvec=None, out_dtype=ops.OUT_BF16, bias=True, wkey=None, split=None, out: Optional[Act] = None,
             asym_pad: bool = False, ln_out: bool = False, ln_in: Optional[tuple] = None, extra=None) -> Act:
        """ln_out: also write the row-moment partials of the output (returned Act carries them in .ln);
        ln_in = (buffer, slots): the input rows are LayerNormalized through the fold (weights <name>.lnw/.lncs/.lnb).
        extra: Act or (Act, Act) read at the output pixel as extra K tiles (ResBlock shortcut folded into conv2)."""
        p = self.p
        x0, x1 = (x if isinstance(x, tuple) else (x, None))
        cin = x0.C + (x1.C if x1 is not None else 0)
        hl, wl = (2 * x0.H, 2 * x0.W) if upsample else (x0.H, x0.W)
        pad = 1 if ksize == 3 else 0
        pad_lead, pad_end = ((0, 1) if asym_pad else (pad, pad))  # asym: padding=((0,1),(0,1)) of image_encoder.py:28
        ho, wo = (hl + pad_lead + pad_end - ksize) // stride + 1, (wl + pad_lead + pad_end - ksize) // stride + 1
        n_out = N // 2 if act == ops.ACT_GEGLU else N
        M = x0.B * ho * wo
        nk = ksize * ksize * (cin // 64)
        can_split = not (split is not None or act == ops.ACT_GEGLU)
        e0, e1 = (extra if isinstance(extra, tuple) else (extra, None))
        cx = 0 if e0 is None else e0.C + (e1.C if e1 is not None else 0)
        nk += cx // 64
        tile_m, tile_n, sk, stages = tuning.lookup(x0.B, x0.H, x0.W, cin, N, ksize, stride, upsample, M, nk, can_split, cx)
        if ln_out or ln_in is not None:
            sk = 1   # the fold is plain-K only
        if sk > 1:
            p.ws_floats = max(p.ws_floats, sk * M * N)
        if out is None and split is None:
            esz = 4 if out_dtype == ops.OUT_F32 else 2
            out = Act(p.alloc(M * n_out * esz), x0.B, ho, wo, n_out)
        kw = dict(a0=x0.buf, a1=None if x1 is None else x1.buf, c1=0 if x1 is None else x1.C,
                  w=self.W[(wkey or name) + ".w"], out=out.buf if out is not None else None, batch=x0.B, h_in=x0.H,
                  w_in=x0.W, c0=x0.C, N=N, ksize=ksize, stride=stride, upsample=upsample,
                  bias=self.W[(wkey or name) + ".b"] if bias else None, act=act, out_dtype=out_dtype,
                  residual=None if residual is None else residual.buf, res_ld=None if residual is None else residual.C,
                  workspace=p.ws if sk > 1 else None, workspace_floats=sk * M * N if sk > 1 else 0, splitk=sk,
                  tile_m=tile_m, tile_n=tile_n, stages=stages, pad=pad_lead, pad_end=pad_end,
                  step_ptr=self.step_ptr if rowvec is not None else None, w_layout=self.w_layout((wkey or name) + ".w"), name=name)
        if e0 is not None:
            kw.update(a2=e0.buf, c2=e0.C, a3=None if e1 is None else e1.buf, c3=0 if e1 is None else e1.C)
        if ln_in is not None:
            wn = wkey or name
            kw.update(w=self.W[wn + ".lnw"], bias=self.W[wn + ".lnb"], ln_in=ln_in[0], ln_in_slots=ln_in[1],
                      ln_colsum=self.W[wn + ".lncs"], ln_eps=EPS, w_layout=self.w_layout(wn + ".lnw"))
        if tuning.is_wreg(tile_m):   # wreg form (csrc/conv_wreg.hip): reads the fragment-major image of the same matrix
            wk = (wkey or name) + (".lnw" if ln_in is not None else ".w")
            frag = getattr(self.W, "fragment_major", None)
            if frag is not None:
                kw.update(w=frag(wk), w_layout=2)
            elif self.W[wk] is None:    # the tuner's tensor-less shape walk
                kw.update(w=None, w_layout=2)
            else:                       # a plain mapping of real tensors: convert, never re-label a row / chunk-major matrix
                from . import packing
                src = self.W[wk]
                if self.w_layout(wk) == 1:
                    raise ValueError(f"{wk}: the wreg form needs packing.PackedWeights (a chunk-major tensor in a plain mapping cannot be re-laid)")
                kw.update(w=packing.fragment_major(src), w_layout=2)
        if ln_out:
            slots = ops.conv_gemm_ln_slots(N=N, tile_n=tile_n, tile_m=tile_m, ksize=ksize, act=act)
            out.ln = (p.alloc(M * slots * 8), slots)
            kw.update(ln_out=out.ln[0], ln_out_slots=slots)
        if rowvec is not None:
            kw.update(rowvec=rowvec[0], rv_step_stride=rowvec[1], rv_batch_stride=rowvec[2])
        if split is not None:
            ns0, ns1, out0, out1, out2, out2_ld = split
            kw.update(out=out0, out_ld=max(ns0, 4), split=(ns0, ns1, out1, ns1, out2, out2_ld))
        p.rec(ops.conv_gemm, **kw)
        return out

    def widen(self, x: Act, copies: int, name: str) -> Act:
        """`x` is the first x.B samples of a buffer that was allocated for copies * x.B: fill the rest with replicas of it and
        return the whole (the fused cond + uncond batch behind the part of the UNet both halves share, SHARE_CFG_PREFIX)."""
        if copies == 1:
            return x
        nbytes = x.B * x.H * x.W * x.C * 2
        assert x.buf.nbytes >= copies * nbytes
        self.p.rec(ops.replicate, src=x.buf, dst=x.buf, nbytes=nbytes, copies=copies, name=name + ".replicate")
        return Act(x.buf, x.B * copies, x.H, x.W, x.C)

    def group_norm(self, x, name, silu: bool) -> Act:
        p = self.p
        x0, x1 = (x if isinstance(x, tuple) else (x, None))
        C = x0.C + (x1.C if x1 is not None else 0)
        out = p.act(x0.B, x0.H, x0.W, C)
        p.rec(ops.group_norm, x0=x0.buf, x1=None if x1 is None else x1.buf, c1=0 if x1 is None else x1.C,
              gamma=self.W[name + ".g"], beta=self.W[name + ".b"], stats=p.gn_stats_slot(x0.B), partials=p.gn_partials, out=out.buf, batch=x0.B,
              hw=x0.H * x0.W, c0=x0.C, silu=silu, eps=EPS, sync=p.gn_sync, name=name)
        return out

    def layer_norm(self, x: Act, name) -> Act:
        out = self.p.act(x.B, x.H, x.W, x.C)
        self.p.rec(ops.layer_norm, x=x.buf, gamma=self.W[name + ".g"], beta=self.W[name + ".b"], out=out.buf, rows=x.M,
                   c=x.C, eps=EPS, name=name)
        return out

    # ---- reference layer types
    def res_block(self, x, name, cout, temb=None, free_input=False, out_copies: int = 1) -> Act:
        """ResBlock (diffusion_model.py:22-51) / VAE ResnetBlock (layers.py:62-80, temb=None).  out_copies > 1: the result is written
        into the head of a buffer for out_copies * B samples and replicated (SHARE_CFG_PREFIX); the returned Act is the wide one."""
        p = self.p
        x0 = x[0] if isinstance(x, tuple) else x
        cin = sum(a.C for a in x) if isinstance(x, tuple) else x.C
        g1 = self.group_norm(x, name + ".norm1", silu=True)
        h = self.conv(g1, name + ".conv1", cout, ksize=3, rowvec=temb)
        p.free(g1)
        g2 = self.group_norm(h, name + ".norm2", silu=True)
        p.free(h)
        fold = SHORTCUT_FOLD and (temb is None or x0.H * x0.W < SHORTCUT_FOLD_MAX_PIXELS)   # (VAE blocks: measured equal, stay folded)
        head = None
        if out_copies > 1:   # the block's output in the head of the wide buffer
            wide = p.act(x0.B * out_copies, x0.H, x0.W, cout)
            head = Act(wide.buf, x0.B, x0.H, x0.W, cout)
        if cin != cout and fold and (name + ".conv2sc.w") in self.W:
            # conv2(h) + conv_shortcut(x) as ONE contraction: K = 9 C_out taps of g2, then the C_in channels of x
            out = self.conv(g2, name + ".conv2sc", cout, ksize=3, extra=x, out=head)
            p.free(g2)
        else:
            if cin != cout:
                res = self.conv(x, name + ".conv_shortcut", cout, ksize=1)
            else:
                assert not isinstance(x, tuple)
                res = x0
            out = self.conv(g2, name + ".conv2", cout, ksize=3, residual=res, out=head)
            p.free(g2)
            if res is not x0:
                p.free(res)
        if out_copies > 1:
            out = self.widen(out, out_copies, name)
        if free_input:
            for a in (x if isinstance(x, tuple) else (x,)):
                p.free(a)
        return out

    def attentions(self, x: Act, name, ctx_kv, ctx_len, heads=8, free_input=False, shared: int = 1) -> Act:
        """Attentions / TransformerBlock / CrossAttention / GEGLU (diffusion_model.py:54-153).  shared > 1 (SHARE_CFG_PREFIX): `x` holds
        `shared` identical copies of x.B / shared samples (the cond and uncond halves in front of the first cross-attention): norm,
        proj_in, q|k|v, the self-attention and its to_out run on ONE copy, their result (rows + LayerNorm partials) is replicated, and the
        block continues on the whole batch from attn2 on, where the halves' text contexts differ."""
        p = self.p
        x_all = x
        if shared > 1:
            assert x.B % shared == 0
            x = Act(x.buf, x.B // shared, x.H, x.W, x.C)   # the first copy
        B, H, Wd, C = x.B, x.H, x.W, x.C
        S = H * Wd
        d = C // heads
        tb = name + ".transformer_blocks.0"
        fold = LN_FOLD and (tb + ".attn1.qkv.lnw") in self.W
        g = self.group_norm(x, name + ".norm", silu=False)
        t0 = self.conv(g, name + ".proj_in", C, ln_out=fold)
        p.free(g)
        # self-attention: fused q|k|v projection, v written transposed for the PV product
        q, k = p.act(B, H, Wd, C), p.act(B, H, Wd, C)
        sp = (S + 7) // 8 * 8  # V^T rows are read in 16-byte chunks
        vt = p.alloc(B * C * sp * 2)
        if fold:   # LayerNorm(norm1) inside the q|k|v GEMM
            self.conv(t0, tb + ".attn1.qkv", 3 * C, bias=False, split=(C, C, q.buf, k.buf, vt, sp), ln_in=t0.ln)
            p.free(t0.ln[0])
        else:
            n1 = self.layer_norm(t0, tb + ".norm1")
            self.conv(n1, tb + ".attn1.qkv", 3 * C, bias=False, split=(C, C, q.buf, k.buf, vt, sp))
            p.free(n1)
        a1 = p.act(B, H, Wd, C)
        p.rec(ops.attention, q=q.buf, k=k.buf, vt=vt, out=a1.buf, batch=B, heads=heads, head_dim=d, s=S, t=S, q_ld=C,
              k_ld=C, vt_ld=sp, o_ld=C, scale=d ** -0.5, q_prescaled=True, name=tb + ".attn1")
        p.free(q, k, vt)
        if shared > 1:
            t1_wide = p.act(B * shared, H, Wd, C)
            t1 = self.conv(a1, tb + ".attn1.to_out.0", C, residual=t0, ln_out=fold, out=Act(t1_wide.buf, B, H, Wd, C))
            p.free(a1, t0)
            ln = t1.ln
            t1 = self.widen(t1, shared, tb + ".attn1.to_out.0")
            if ln is not None:   # the row-moment partials of the rows, replicated like the rows
                nb_ln = B * H * Wd * ln[1] * 8
                ln_wide = p.alloc(shared * nb_ln)
                p.rec(ops.replicate, src=ln[0], dst=ln_wide, nbytes=nb_ln, copies=shared, name=tb + ".attn1.to_out.0.ln.replicate")
                p.free(ln[0])
                t1.ln = (ln_wide, ln[1])
            x, B = x_all, B * shared
        else:
            t1 = self.conv(a1, tb + ".attn1.to_out.0", C, residual=t0, ln_out=fold)
            p.free(a1, t0)
        # cross-attention over the text context (k, v^T precomputed once per prompt)
        kc, vtc, tp = ctx_kv[tb + ".attn2"]
        a2 = p.act(B, H, Wd, C)
        if fold and XATTN_FUSED and heads == 8 and (d in (40, 80) or (d == 160 and XATTN_FUSED_D160)) and ctx_len <= 96:
            # norm2 -> to_q -> attention over the 77 context tokens as ONE launch (msd_cross_attention_q)
            wn = tb + ".attn2.to_q"
            p.rec(ops.cross_attention_q, x=t1.buf, ln_in=t1.ln[0], ln_in_slots=t1.ln[1], wq=self.W[wn + ".lnw"],
                  ln_colsum=self.W[wn + ".lncs"], bias=self.W[wn + ".lnb"], k=kc, vt=vtc, out=a2.buf, batch=B, heads=heads,
                  head_dim=d, s=S, t=ctx_len, k_ld=C, vt_ld=tp, o_ld=C, ln_eps=EPS, w_layout=self.w_layout(wn + ".lnw"),
                  name=tb + ".attn2")
            p.free(t1.ln[0])
        else:
            if fold:
                q2 = self.conv(t1, tb + ".attn2.to_q", C, bias=False, ln_in=t1.ln)
                p.free(t1.ln[0])
            else:
                n2 = self.layer_norm(t1, tb + ".norm2")
                q2 = self.conv(n2, tb + ".attn2.to_q", C, bias=False)
                p.free(n2)
            p.rec(ops.attention, q=q2.buf, k=kc, vt=vtc, out=a2.buf, batch=B, heads=heads, head_dim=d, s=S, t=ctx_len, q_ld=C,
                  k_ld=C, vt_ld=tp, o_ld=C, scale=d ** -0.5, q_prescaled=True, name=tb + ".attn2")
            p.free(q2)
        t2 = self.conv(a2, tb + ".attn2.to_out.0", C, residual=t1, ln_out=fold)
        p.free(a2, t1)
        # feed-forward: GEGLU fused into the first GEMM's epilogue
        if fold:
            ff = self.conv(t2, tb + ".ff.net.0.proj", 8 * C, act=ops.ACT_GEGLU, bias=False, ln_in=t2.ln)
            p.free(t2.ln[0])
        else:
            n3 = self.layer_norm(t2, tb + ".norm3")
            ff = self.conv(n3, tb + ".ff.net.0.proj", 8 * C, act=ops.ACT_GEGLU)
            p.free(n3)
        if FF_PROJ_FOLD and (name + ".ffproj.w") in self.W:   # ff.net.2 and proj_out as one GEMM over [ff | t2]
            out = self.conv((ff, t2), name + ".ffproj", C, residual=x)
            p.free(ff, t2)
        else:
            t3 = self.conv(ff, tb + ".ff.net.2", C, residual=t2)
            p.free(ff, t2)
            out = self.conv(t3, name + ".proj_out", C, residual=x)
            p.free(t3)
        if free_input:
            p.free(x)
        return out


# ----------------------------------------------------------------------------- networks
def emit_context_kv(e: Emitter, ctx: Act, attn_names: Sequence[Tuple[str, int]], persistent: Plan):
    """K and V^T of the text context for every cross-attention layer (constant over all steps).
    ctx: bf16 [NB][T][768] as an Act with H=T, W=1.  Buffers come from `persistent` (never recycled)."""
    out = {}
    T = ctx.H
    tp = (T + 7) // 8 * 8
    for name, C in attn_names:
        kb = persistent.alloc(ctx.B * T * C * 2)
        vb = persistent.alloc(ctx.B * C * tp * 2)
        e.conv(ctx, name + ".kv", 2 * C, bias=False, split=(0, C, None, kb, vb, tp))
        out[name] = (kb, vb, tp)
    return out


UNET_ATTN_LAYERS: List[Tuple[str, int]] = []
for _lvl, _ch in enumerate(wtab.UNET_CH[:3]):
    for _r in range(2):
        UNET_ATTN_LAYERS.append((f"down_blocks.{_lvl}.attentions.{_r}.transformer_blocks.0.attn2", _ch))
UNET_ATTN_LAYERS.append(("mid_block.attentions.0.transformer_blocks.0.attn2", 1280))
ENCODER_ATTN_LAYERS = list(UNET_ATTN_LAYERS)
for _ui, _lvl in enumerate((3, 2, 1, 0)):
    if _lvl < 3:
        for _r in range(3):
            UNET_ATTN_LAYERS.append((f"up_blocks.{_ui}.attentions.{_r}.transformer_blocks.0.attn2", wtab.UNET_CH[_lvl]))


def resblock_names(encoder_only: bool) -> List[Tuple[str, int]]:
    """(name, c_out) of every ResBlock in weight-table order: column layout of the time-projection table."""
    out = []
    for lvl, ch in enumerate(wtab.UNET_CH):
        for r in range(2):
            out.append((f"down_blocks.{lvl}.resnets.{r}", ch))
    out += [("mid_block.resnets.0", 1280), ("mid_block.resnets.1", 1280)]
    if not encoder_only:
        for ui, lvl in enumerate((3, 2, 1, 0)):
            for r in range(3):
                out.append((f"up_blocks.{ui}.resnets.{r}", wtab.UNET_CH[lvl]))
    return out


def emit_time_embedding(e: Emitter, t_emb_f32, rows: int, out_table, encoder_only: bool):
    """time_embedding MLP + every ResBlock's time_emb_proj for `rows` embeddings at once
    (diffusion_model.py:184-188,30,47).  The MLP runs in fp32 on the vector-FMA path; the 22 (UNet) / 10 (ControlNet)
    projections, concatenated to ONE 1280 -> 20,160 / 9,600 Dense, run on the MFMA path when the model packed them in
    bf16 (MFMA_TEMB_PROJ): 105 MB of fp32 weights through the vector-FMA kernel took 1.4 ms per image."""
    p = e.p
    total = sum(c for _, c in resblock_names(encoder_only))
    mfma = e.W["time_emb_proj_cat.w"].dtype == torch.bfloat16
    h1 = p.alloc(rows * 1280 * 4)
    h2 = p.alloc(rows * 1280 * (2 if mfma else 4))
    common = dict(batch=rows, h_in=1, w_in=1, ksize=1, in_dtype=ops.OUT_F32)
    p.rec(ops.conv_direct, x=t_emb_f32, w=e.W["time_embedding.linear_1.w"], bias=e.W["time_embedding.linear_1.b"], out=h1,
          c_in=320, c_out=1280, act=ops.ACT_SILU, out_dtype=ops.OUT_F32, name="time_embedding.linear_1", **common)
    # (the swish in front of every time_emb_proj, diffusion_model.py:30, is the output activation of linear_2)
    p.rec(ops.conv_direct, x=h1, w=e.W["time_embedding.linear_2.w"], bias=e.W["time_embedding.linear_2.b"], out=h2,
          c_in=1280, c_out=1280, act=ops.ACT_SILU, out_dtype=ops.OUT_BF16 if mfma else ops.OUT_F32, name="time_embedding.linear_2", **common)
    if mfma:
        e.conv(Act(h2, 1, rows, 1, 1280), "time_emb_proj_cat", total, out_dtype=ops.OUT_F32, out=Act(out_table, 1, rows, 1, total))
    else:
        p.rec(ops.conv_direct, x=h2, w=e.W["time_emb_proj_cat.w"], bias=e.W["time_emb_proj_cat.b"], out=out_table,
              c_in=1280, c_out=total, out_dtype=ops.OUT_F32, name="time_emb_proj_cat", **common)
    p.free(h1, h2)
    return total


def temb_columns(encoder_only: bool) -> Dict[str, int]:
    cols, off = {}, 0
    for name, c in resblock_names(encoder_only):
        cols[name] = off
        off += c
    return cols


def _emit_encoder(e: Emitter, x: Act, temb_of, ctx_kv, ctx_len, outputs: List[Act], shared: int = 1) -> Act:
    """Down path + mid block shared by the UNet (diffusion_model.py:193-229) and the ControlNet.  shared > 1: `x` (conv_in's output) is
    `shared` identical copies of x.B / shared samples - the first ResBlock and the front of the first transformer block run on one."""
    for lvl, ch in enumerate(wtab.UNET_CH):
        for r in range(2):
            name = f"down_blocks.{lvl}.resnets.{r}"
            if shared > 1 and lvl == 0 and r == 0:
                one = Act(x.buf, x.B // shared, x.H, x.W, x.C)
                x = e.res_block(one, name, ch, temb=temb_of(name), out_copies=shared)
                x = e.attentions(x, f"down_blocks.{lvl}.attentions.{r}", ctx_kv, ctx_len, free_input=True, shared=shared)
                outputs.append(x)
                continue
            x = e.res_block(x, name, ch, temb=temb_of(name))
            if lvl < 3:
                x = e.attentions(x, f"down_blocks.{lvl}.attentions.{r}", ctx_kv, ctx_len, free_input=True)
            outputs.append(x)
        if lvl < 3:
            x = e.conv(x, f"down_blocks.{lvl}.downsamplers.0.conv", ch, ksize=3, stride=2)
            outputs.append(x)
    x = e.res_block(x, "mid_block.resnets.0", 1280, temb=temb_of("mid_block.resnets.0"))
    x = e.attentions(x, "mid_block.attentions.0", ctx_kv, ctx_len, free_input=True)
    x = e.res_block(x, "mid_block.resnets.1", 1280, temb=temb_of("mid_block.resnets.1"), free_input=True)
    return x


def emit_unet(e: Emitter, latent_f32, latent_batch_mod: int, NB: int, h: int, w: int, temb, ctx_kv, ctx_len: int,
              eps_out_f32, controls=None, control_taps=None) -> None:
    """DiffusionModel graph (diffusion_model.py:184-279).

    latent_f32: fp32 [latent_batch_mod][h][w][4] (sample b reads row b % latent_batch_mod);
    temb = (table, step_stride, batch_stride, {resblock: column}); eps_out_f32: fp32 [NB][h][w][4].
    ControlNet residuals (diffusion_model.py:230-234: added to the 12 skips and to the mid-block output AFTER the down
    path), two forms:
      control_taps = (ControlNet emitter, its 13 feature maps from emit_controlnet_features): the fused device loop.  Each
        residual is a 1x1 "zero conv" of a ControlNet feature map (control_net.py:92-106), so that conv runs HERE with the
        skip as its epilogue residual and the skip's buffer as its output: skip += zero_conv(feature) in one launch, summed
        in fp32 — the 13 elementwise adds do not exist;
      controls = 13 fp32 buffers [NB][h_i][w_i][c_i] handed over the model boundary (predict_on_batch): one
        add-and-round launch each."""
    p = e.p
    table, sstride, bstride, cols = temb

    def temb_of(name):
        return (table.at(cols[name] * 4), sstride, bstride)

    x = p.act(NB, h, w, 320)
    p.rec(ops.conv_direct, x=latent_f32, w=e.W["conv_in.w"], bias=e.W["conv_in.b"], out=x.buf, batch=NB,
          in_batch_mod=latent_batch_mod, h_in=h, w_in=w, c_in=4, c_out=320, ksize=3, in_dtype=ops.OUT_F32,
          out_dtype=ops.OUT_BF16, name="conv_in")
    outputs: List[Act] = [x]
    # (conv_in itself stays on the whole batch: every row reads latent row b % latent_batch_mod, a 10-us launch whose output is skip 0)
    shared = NB // latent_batch_mod if (SHARE_CFG_PREFIX and latent_batch_mod > 0 and NB % latent_batch_mod == 0) else 1
    x = _emit_encoder(e, x, temb_of, ctx_kv, ctx_len, outputs, shared=shared)
    p.mark("controls")   # everything above is independent of the ControlNet (its encoder may run beside it on another stream)
    if control_taps is not None:
        e_c, feats = control_taps
        assert len(outputs) == 12 and len(feats) == 13
        for i, (o, f) in enumerate(zip(outputs + [x], feats)):
            assert (o.B, o.H, o.W, o.C) == (f.B, f.H, f.W, f.C)
            e_c.conv(f, f"zero_convs.{i}", o.C, ksize=1, residual=o, out=o)
            p.free(f)
    elif controls is not None:
        assert len(outputs) == 12 and len(controls) == 13
        for i, (o, c) in enumerate(zip(outputs + [x], controls)):
            p.rec(ops.add_f32_bf16, a=o.buf, b=c, out=o.buf, n=o.M * o.C, name=f"control.{i}")
    for ui, lvl in enumerate((3, 2, 1, 0)):
        ch = wtab.UNET_CH[lvl]
        for r in range(3):
            skip = outputs.pop()
            name = f"up_blocks.{ui}.resnets.{r}"
            x = e.res_block((x, skip), name, ch, temb=temb_of(name), free_input=True)
            if lvl < 3:
                x = e.attentions(x, f"up_blocks.{ui}.attentions.{r}", ctx_kv, ctx_len, free_input=True)
        if lvl > 0:
            y = e.conv(x, f"up_blocks.{ui}.upsamplers.0.conv", ch, ksize=3, upsample=True)
            p.free(x)
            x = y
    g = e.group_norm(x, "conv_norm_out", silu=True)
    p.free(x)
    if MFMA_CONV_OUT and "conv_out.m.w" in e.W:
        # 320 -> 4 channels: N is tiny but K = 2880, so the implicit-GEMM kernel (bf16 weights, fp32 accumulate,
        # fp32 output) beats the vector-FMA kernel 3x; the 60 unused columns of the tile cost nothing that matters
        e.conv(g, "conv_out", 4, ksize=3, out_dtype=ops.OUT_F32, wkey="conv_out.m", out=Act(eps_out_f32, NB, h, w, 4))
    else:
        p.rec(ops.conv_direct, x=g.buf, w=e.W["conv_out.w"], bias=e.W["conv_out.b"], out=eps_out_f32, batch=NB, h_in=h, w_in=w,
              c_in=320, c_out=4, ksize=3, in_dtype=ops.OUT_BF16, out_dtype=ops.OUT_F32, name="conv_out")
    p.free(g)


def emit_controlnet_features(e: Emitter, latent_f32, latent_batch_mod: int, NB: int, h: int, w: int, temb, ctx_kv, ctx_len: int,
                             hint: Act) -> List[Act]:
    """ControlNet (control_net.py:45-90) up to its 13 taps: conv_in(latent)+hint and the encoder.  The taps' 1x1 zero
    convs are emitted by the consumer (emit_unet: fused with the residual adds) or by emit_controlnet."""
    p = e.p
    table, sstride, bstride, cols = temb

    def temb_of(name):
        return (table.at(cols[name] * 4), sstride, bstride)

    x = p.act(NB, h, w, 320)
    p.rec(ops.conv_direct, x=latent_f32, w=e.W["conv_in.w"], bias=e.W["conv_in.b"], residual=hint.buf, out=x.buf, batch=NB,
          in_batch_mod=latent_batch_mod, h_in=h, w_in=w, c_in=4, c_out=320, ksize=3, in_dtype=ops.OUT_F32,
          out_dtype=ops.OUT_BF16, name="conv_in+hint")
    outputs: List[Act] = [x]
    # (the hint is tiled to both halves - emit_hintnet's copies - so conv_in + hint is the same in every copy of the batch too)
    shared = NB // latent_batch_mod if (SHARE_CFG_PREFIX and latent_batch_mod > 0 and NB % latent_batch_mod == 0) else 1
    x = _emit_encoder(e, x, temb_of, ctx_kv, ctx_len, outputs, shared=shared)
    outputs.append(x)
    assert len(outputs) == 13
    return outputs


def emit_controlnet(e: Emitter, latent_f32, latent_batch_mod: int, NB: int, h: int, w: int, temb, ctx_kv, ctx_len: int,
                    hint: Act, outs: List[Act]) -> None:
    """ControlNet (control_net.py:45-107): conv_in(latent)+hint, encoder, 13 1x1 'zero' convs."""
    p = e.p
    outputs = emit_controlnet_features(e, latent_f32, latent_batch_mod, NB, h, w, temb, ctx_kv, ctx_len, hint)
    assert len(outs) == 13
    for i, (o, dst) in enumerate(zip(outputs, outs)):
        e.conv(o, f"zero_convs.{i}", o.C, ksize=1, out=dst)
    for o in outputs:
        p.free(o)


def emit_hintnet(e: Emitter, image_f32, B: int, H: int, W: int, out: Act, copies: int = 1) -> None:
    """HintNet (control_net.py:10-31): 8 convs with swish between; last conv (256->320) on MFMA.  `copies`: `out` holds
    that many replicas of the B-sample result back to back (the cond and uncond halves of a fused forward read the same
    hint): the last conv writes each of them, nothing else runs twice."""
    p = e.p
    x_buf, x_f32 = image_f32, True
    h, w = H, W
    cur: Optional[Act] = None
    for i, (cin, cout, s) in enumerate(wtab.HINT_CH):
        ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
        if i < 7:
            nxt = p.act(B, ho, wo, cout)
            p.rec(ops.conv_direct, x=x_buf, w=e.W[f"input_hint_block.{i}.w"], bias=e.W[f"input_hint_block.{i}.b"], out=nxt.buf,
                  batch=B, h_in=h, w_in=w, c_in=cin, c_out=cout, ksize=3, stride=s,
                  in_dtype=ops.OUT_F32 if x_f32 else ops.OUT_BF16, out_dtype=ops.OUT_BF16, act=ops.ACT_SILU,
                  name=f"input_hint_block.{i}")
            if cur is not None:
                p.free(cur)
            cur, x_buf, x_f32 = nxt, nxt.buf, False
        else:
            for rep in range(copies):
                e.conv(cur, f"input_hint_block.{i}", cout, ksize=3, out=Act(out.buf.at(rep * B * ho * wo * cout * 2), B, ho, wo, cout))
            p.free(cur)
        h, w = ho, wo


# VAE attention: the fused d = 512 kernel (scores stay on chip), or the three-launch route that materialises them
VAE_FUSED_ATTN = os.environ.get("MSD_VAE_FUSED_ATTN", "1") != "0"   # (env switch: same-box A/B runs)
VAE_ATTN_SPLIT = os.environ.get("MSD_VAE_ATTN_SPLIT", "1") != "0"   # 4-way key split of the d = 512 attention (env switch: A/B runs)


def emit_vae_attention(e: Emitter, x: Act, name: str) -> Act:
    """AttentionBlock (layers.py:28-59): single head, d = C = 512, scale 1/sqrt(C); q/k/v/proj Dense with bias.
    softmax(q k^T / sqrt(C)) v is one msd_attention launch (the d = 512 kernel: scores never leave the chip; the reference
    materialises S x S fp32 scores, 64 MB per image at 512x512, 340 MB at 768x768)."""
    p = e.p
    B, H, Wd, C = x.B, x.H, x.W, x.C
    S = H * Wd
    g = e.group_norm(x, name + ".group_norm", silu=False)
    q, k = p.act(B, H, Wd, C), p.act(B, H, Wd, C)
    vt = p.alloc(B * C * S * 2)
    e.conv(g, name + ".qkv", 3 * C, split=(C, C, q.buf, k.buf, vt, S))
    p.free(g)
    o = p.act(B, H, Wd, C)
    if VAE_FUSED_ATTN and C == 512 and S % 32 == 0:
        # scratch of the kernel's 4-way key split (msd_attention, ABI 9): one head at 512x512 is 64 query tiles on 256 CUs
        wsf = 4 * B * S * (C + 2) if (VAE_ATTN_SPLIT and S >= 2048 and S % 128 == 0) else 0
        ws = p.alloc(wsf * 4) if wsf else None
        p.rec(ops.attention, q=q.buf, k=k.buf, vt=vt, out=o.buf, batch=B, heads=1, head_dim=C, s=S, t=S, q_ld=C, k_ld=C, vt_ld=S,
              o_ld=C, scale=1.0 / float(np.sqrt(C)), workspace=ws, workspace_floats=wsf, name=name + ".attention")
        p.free(q, k, vt)
        if ws is not None:
            p.free(ws)
        out = e.conv(o, name + ".proj_attn", C, residual=x)
        p.free(o, x)
        return out
    # (any other geometry: scores materialised per sample — S = Q K^T on MFMA, row softmax, O = P V on MFMA)
    scores = p.alloc(S * S * 4)
    probs = p.alloc(S * S * 2)
    for b in range(B):
        # scores[s, t] = q[b, s, :] . k[b, t, :]   (k rows act as the [N][K] "weight" operand)
        p.rec(ops.conv_gemm, a0=q.buf.at(b * S * C * 2), w=k.buf.at(b * S * C * 2), out=scores, batch=1, h_in=S, w_in=1, c0=C,
              N=S, out_dtype=ops.OUT_F32, name=name + ".qk")
        p.rec(ops.softmax_rows, x=scores, out=probs, rows=S, cols=S, ld_in=S, ld_out=S, scale=1.0 / float(np.sqrt(C)),
              name=name + ".softmax")
        p.rec(ops.conv_gemm, a0=probs, w=vt.at(b * C * S * 2), out=o.buf.at(b * S * C * 2), batch=1, h_in=S, w_in=1, c0=S,
              N=C, name=name + ".pv")
    p.free(q, k, vt, scores, probs)
    out = e.conv(o, name + ".proj_attn", C, residual=x)
    p.free(o, x)
    return out


def emit_decoder(e: Emitter, latent_f32, B: int, h: int, w: int, out_buf, out_dtype: int) -> None:
    """ImageDecoder (image_decoder.py:22-55). out: fp32 or uint8 [B][8h][8w][3]."""
    p = e.p
    z = p.alloc(B * h * w * 4 * 4)
    p.rec(ops.conv_direct, x=latent_f32, w=e.W["post_quant_conv.w"], bias=e.W["post_quant_conv.b"], out=z, batch=B, h_in=h,
          w_in=w, c_in=4, c_out=4, ksize=1, in_dtype=ops.OUT_F32, out_dtype=ops.OUT_F32, in_scale=1.0 / 0.18215,
          name="post_quant_conv")
    x = p.act(B, h, w, 512)
    p.rec(ops.conv_direct, x=z, w=e.W["decoder.conv_in.w"], bias=e.W["decoder.conv_in.b"], out=x.buf, batch=B, h_in=h, w_in=w,
          c_in=4, c_out=512, ksize=3, in_dtype=ops.OUT_F32, out_dtype=ops.OUT_BF16, name="decoder.conv_in")
    p.free(z)
    x = e.res_block(x, "decoder.mid_block.resnets.0", 512, free_input=True)
    x = emit_vae_attention(e, x, "decoder.mid_block.attentions.0")
    x = e.res_block(x, "decoder.mid_block.resnets.1", 512, free_input=True)
    for bi, (cin, cout, up) in enumerate(wtab.VAE_DEC_BLOCKS):
        for r in range(3):
            x = e.res_block(x, f"decoder.up_blocks.{bi}.resnets.{r}", cout, free_input=True)
        if up:
            y = e.conv(x, f"decoder.up_blocks.{bi}.upsamplers.0.conv", cout, ksize=3, upsample=True)
            p.free(x)
            x = y
    g = e.group_norm(x, "decoder.conv_norm_out", silu=True)
    p.free(x)
    p.rec(ops.conv_direct, x=g.buf, w=e.W["decoder.conv_out.w"], bias=e.W["decoder.conv_out.b"], out=out_buf, batch=B,
          h_in=g.H, w_in=g.W, c_in=128, c_out=3, ksize=3, in_dtype=ops.OUT_BF16, out_dtype=out_dtype, name="decoder.conv_out")
    p.free(g)


def emit_encoder(e: Emitter, image_f32, B: int, H: int, W: int, latent_out_f32) -> None:
    """ImageEncoder (image_encoder.py:21-48): image (B,H,W,3) in [-1,1] -> mean latent * 0.18215,
    fp32 (B,H/8,W/8,4).  The three stride-2 convs use the asymmetric ((0,1),(0,1)) padding; the
    final 1x1 quant_conv + split(...)[0] * 0.18215 (:46-47) is one 8->4 conv with pre-scaled weights."""
    p = e.p
    x = p.act(B, H, W, 128)
    p.rec(ops.conv_direct, x=image_f32, w=e.W["encoder.conv_in.w"], bias=e.W["encoder.conv_in.b"], out=x.buf, batch=B, h_in=H,
          w_in=W, c_in=3, c_out=128, ksize=3, in_dtype=ops.OUT_F32, out_dtype=ops.OUT_BF16, name="encoder.conv_in")
    for bi, (cin, cout, down) in enumerate(wtab.VAE_ENC_BLOCKS):
        for r in range(2):
            x = e.res_block(x, f"encoder.down_blocks.{bi}.resnets.{r}", cout, free_input=True)
        if down:
            y = e.conv(x, f"encoder.down_blocks.{bi}.downsamplers.0.conv", cout, ksize=3, stride=2, asym_pad=True)
            p.free(x)
            x = y
    x = e.res_block(x, "encoder.mid_block.resnets.0", 512, free_input=True)
    x = emit_vae_attention(e, x, "encoder.mid_block.attentions.0")
    x = e.res_block(x, "encoder.mid_block.resnets.1", 512, free_input=True)
    g = e.group_norm(x, "encoder.conv_norm_out", silu=True)
    p.free(x)
    m = e.conv(g, "encoder.conv_out", 8, ksize=3, out_dtype=ops.OUT_F32)   # 512 -> 8 moments (mean | logvar), fp32
    p.free(g)
    p.rec(ops.conv_direct, x=m.buf, w=e.W["quant_conv.mean.w"], bias=e.W["quant_conv.mean.b"], out=latent_out_f32, batch=B,
          h_in=m.H, w_in=m.W, c_in=8, c_out=4, ksize=1, in_dtype=ops.OUT_F32, out_dtype=ops.OUT_F32, name="quant_conv.mean")
    p.free(m)


def emit_text_encoder(e: Emitter, x: Act, n_layers: int, heads: int = 12) -> Act:
    """CLIP text transformer (text_encoder.py:36-58,104-113): n_layers x [LN -> causal self-attention
    (q/k/v/out Dense with bias, scale on the scores) -> +x -> LN -> fc1 -> quick_gelu -> fc2 -> +x],
    then the final LayerNorm.  x: bf16 tokens [B][T][768]; the layers after out[clip_skip] are not built."""
    p = e.p
    B, T, C = x.B, x.H, x.C
    d = C // heads
    Tp = (T + 7) // 8 * 8
    for i in range(n_layers):
        ln = f"text_model.encoder.layers.{i}"
        h = e.layer_norm(x, ln + ".layer_norm1")
        q, k = p.act(B, T, 1, C), p.act(B, T, 1, C)
        vt = p.alloc(B * C * Tp * 2)
        e.conv(h, ln + ".self_attn.qkv", 3 * C, split=(C, C, q.buf, k.buf, vt, Tp))
        p.free(h)
        o = p.act(B, T, 1, C)
        p.rec(ops.attention, q=q.buf, k=k.buf, vt=vt, out=o.buf, batch=B, heads=heads, head_dim=d, s=T, t=T, q_ld=C, k_ld=C,
              vt_ld=Tp, o_ld=C, scale=float(d) ** -0.5, causal=True, name=ln + ".self_attn")
        p.free(q, k, vt)
        x2 = e.conv(o, ln + ".self_attn.out_proj", C, residual=x)
        p.free(o, x)
        h = e.layer_norm(x2, ln + ".layer_norm2")
        f = e.conv(h, ln + ".mlp.fc1", 4 * C, act=ops.ACT_QUICK_GELU)
        p.free(h)
        x = e.conv(f, ln + ".mlp.fc2", C, residual=x2)
        p.free(f, x2)
    out = e.layer_norm(x, "text_model.final_layer_norm")
    p.free(x)
    return out

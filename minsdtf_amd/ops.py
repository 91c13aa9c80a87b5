"""Typed launch records for the C ABI (include/minsdtf_hip.h).

Every function here only *describes* one library call: it fills the ctypes argument block from
raw device addresses + shapes and returns a :class:`Call`.  Running a Call on a stream is one
foreign call, so a whole UNet forward is a flat Python list of Calls that is recorded once and
then replayed from a hipGraph.  No arithmetic happens in this module and nothing falls back to
PyTorch: if the library is missing, :func:`_lib.load` raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

from . import _lib
from ._lib import ACT_GEGLU, ACT_NONE, ACT_QUICK_GELU, ACT_SILU, OUT_BF16, OUT_F32, OUT_U8  # noqa: F401


class Call:
    """One stream-ordered library call: ``fn(*args, stream)``."""

    __slots__ = ("fn", "args", "name", "keep")

    def __init__(self, fn, args, name, keep=None):
        self.fn, self.args, self.name, self.keep = fn, args, name, keep

    def __call__(self, stream: int) -> None:
        rc = self.fn(*self.args, stream)
        if rc != 0:
            _lib.check(rc, self.name)


def _p(x) -> Optional[int]:
    """Device address of a torch tensor / Buf / int / None."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "ptr"):
        return x.ptr
    return x.data_ptr()


def conv_gemm(*, a0, w, out, batch, h_in, w_in, c0, N, a1=None, c1=0, ksize=1, stride=1, upsample=False, bias=None,
              rowvec=None, rv_step_stride=0, rv_batch_stride=0, step_ptr=None, residual=None, res_ld=None, act=ACT_NONE,
              out_dtype=OUT_BF16, out_ld=None, split=None, workspace=None, workspace_floats=0, splitk=1, tile_n=0,
              tile_m=0, stages=0, pad=None, pad_end=None, ln_in=None, ln_in_slots=0, ln_colsum=None, ln_out=None,
              ln_out_slots=0, ln_eps=1e-5, a2=None, c2=0, a3=None, c3=0, w_layout=0, name="conv_gemm") -> Call:
    """split = (ns0, ns1, out1, out1_ld, out2, out2_ld) selects the q|k|v^T epilogue.
    pad / pad_end: leading / trailing zero padding (default: symmetric 1 for 3x3, 0 for 1x1);
    (0, 1) is the VAE encoder's stride-2 padding ((0,1),(0,1))."""
    lib = _lib.load()
    if pad is None:
        pad = 1 if ksize == 3 else 0
    if pad_end is None:
        pad_end = pad
    hl, wl = (2 * h_in, 2 * w_in) if upsample else (h_in, w_in)
    h_out = (hl + pad + pad_end - ksize) // stride + 1
    w_out = (wl + pad + pad_end - ksize) // stride + 1
    n_out = N // 2 if act == ACT_GEGLU else N
    s = _lib.MsdConvGemm()
    s.a0, s.a1, s.w = _p(a0), _p(a1), _p(w)
    s.bias, s.rowvec, s.step_ptr, s.residual = _p(bias), _p(rowvec), _p(step_ptr), _p(residual)
    s.out = _p(out)
    s.workspace, s.workspace_floats = _p(workspace), int(workspace_floats)
    s.batch, s.h_in, s.w_in, s.c0, s.c1 = batch, h_in, w_in, c0, c1
    s.h_out, s.w_out, s.ksize, s.stride, s.pad, s.upsample = h_out, w_out, ksize, stride, pad, int(bool(upsample))
    s.N, s.act, s.out_dtype = N, act, out_dtype
    s.out_ld = n_out if out_ld is None else out_ld
    s.res_ld = (n_out if res_ld is None else res_ld) if residual is not None else 0
    s.rv_step_stride, s.rv_batch_stride = rv_step_stride, rv_batch_stride
    if split is not None:
        ns0, ns1, out1, out1_ld, out2, out2_ld = split
        s.split_mode, s.ns0, s.ns1 = 1, ns0, ns1
        s.out1, s.out1_ld, s.out2, s.out2_ld = _p(out1), out1_ld, _p(out2), out2_ld
        if out_ld is None:
            s.out_ld = max(ns0, 4)
    s.splitk, s.tile_n, s.tile_m, s.stages = splitk, tile_n, tile_m, stages
    # LayerNorm fold (include/minsdtf_hip.h): row-moment partials in / out, column sums of the folded weights
    s.ln_in, s.ln_colsum, s.ln_out = _p(ln_in), _p(ln_colsum), _p(ln_out)
    s.ln_in_slots, s.ln_out_slots, s.ln_eps = ln_in_slots, ln_out_slots, float(ln_eps)
    s.a2, s.a3, s.c2, s.c3 = _p(a2), _p(a3), c2, c3   # shortcut operand: extra K tiles read at the output pixel
    s.w_layout = int(w_layout)   # 0: [N][K]; 1: chunk-major [K/64][N][64] (packing.chunk_major); 2: fragment-major (packing.fragment_major)
    return Call(lib.msd_conv_gemm, (C.byref(s),), name, keep=s)


def conv_gemm_ln_slots(*, N, tile_n, tile_m=0, ksize=1, act=ACT_NONE) -> int:
    """Row-moment partials per row a launch with these parameters writes to ``ln_out``."""
    lib = _lib.load()
    s = _lib.MsdConvGemm()
    s.N, s.tile_n, s.tile_m, s.ksize, s.act = N, tile_n, tile_m, ksize, act
    n = lib.msd_conv_gemm_ln_slots(C.byref(s))
    if n <= 0:
        _lib.check(n, "conv_gemm_ln_slots")
    return n


def conv_direct(*, x, w, out, batch, h_in, w_in, c_in, c_out, ksize=3, stride=1, bias=None, residual=None,
                in_batch_mod=None, in_dtype=OUT_BF16, out_dtype=OUT_BF16, act=ACT_NONE, act_in=False, in_scale=1.0,
                name="conv_direct") -> Call:
    lib = _lib.load()
    pad = 1 if ksize == 3 else 0
    s = _lib.MsdConvDirect()
    s.in_, s.w, s.bias, s.residual, s.out = _p(x), _p(w), _p(bias), _p(residual), _p(out)
    s.batch, s.in_batch_mod = batch, (batch if in_batch_mod is None else in_batch_mod)
    s.h_in, s.w_in, s.c_in = h_in, w_in, c_in
    s.h_out = (h_in + 2 * pad - ksize) // stride + 1
    s.w_out = (w_in + 2 * pad - ksize) // stride + 1
    s.c_out, s.ksize, s.stride, s.pad = c_out, ksize, stride, pad
    s.in_dtype, s.out_dtype, s.act, s.act_in, s.in_scale = in_dtype, out_dtype, act, int(bool(act_in)), float(in_scale)
    return Call(lib.msd_conv_direct, (C.byref(s),), name, keep=s)


GN_MAX_CHUNKS = 1024


GN_SYNC_WORDS_PER_SAMPLE = 16384   # MSD_GN_SYNC_WORDS_PER_SAMPLE


def group_norm(*, x0, gamma, beta, stats, partials, out, batch, hw, c0, x1=None, c1=0, silu=False, eps=1e-5,
               partials_floats=None, sync=None, sync_words=None, name="group_norm") -> Call:
    """sync: zero-initialised uint32 block of >= batch * GN_SYNC_WORDS_PER_SAMPLE words owned by the calling stream (see
    include/minsdtf_hip.h): enables the single-launch cluster form for the larger tensors."""
    lib = _lib.load()
    s = _lib.MsdGroupNorm()
    s.x0, s.x1, s.gamma, s.beta, s.stats, s.out = _p(x0), _p(x1), _p(gamma), _p(beta), _p(stats), _p(out)
    s.partials = _p(partials)
    s.partials_floats = batch * GN_MAX_CHUNKS * 64 if partials_floats is None else partials_floats
    s.batch, s.hw, s.c0, s.c1, s.silu, s.eps = batch, hw, c0, c1, int(bool(silu)), float(eps)
    s.sync = _p(sync)
    s.sync_words = 0 if sync is None else (batch * GN_SYNC_WORDS_PER_SAMPLE if sync_words is None else sync_words)
    return Call(lib.msd_group_norm, (C.byref(s),), name, keep=s)


def layer_norm(*, x, gamma, beta, out, rows, c, eps=1e-5, name="layer_norm") -> Call:
    lib = _lib.load()
    return Call(lib.msd_layer_norm, (_p(x), _p(gamma), _p(beta), _p(out), rows, c, C.c_float(eps)), name)


def attention(*, q, k, vt, out, batch, heads, head_dim, s, t, q_ld, k_ld, vt_ld, o_ld, scale, causal=False,
              q_prescaled=False, workspace=None, workspace_floats=0, name="attention") -> Call:
    lib = _lib.load()
    a = _lib.MsdAttention()
    a.q, a.k, a.vt, a.out = _p(q), _p(k), _p(vt), _p(out)
    a.batch, a.heads, a.head_dim, a.s, a.t = batch, heads, head_dim, s, t
    a.q_ld, a.k_ld, a.vt_ld, a.o_ld, a.scale = q_ld, k_ld, vt_ld, o_ld, float(scale)
    a.causal = int(bool(causal))
    a.q_prescaled = int(bool(q_prescaled))
    a.workspace, a.workspace_floats = _p(workspace), int(workspace_floats)   # (head_dim 512: scratch of the 4-way key split)
    return Call(lib.msd_attention, (C.byref(a),), name, keep=a)


def cross_attention_q(*, x, ln_in, ln_in_slots, wq, ln_colsum, bias, k, vt, out, batch, heads, head_dim, s, t, k_ld, vt_ld, o_ld,
                      ln_eps=1e-5, w_layout=0, name="cross_attention_q") -> Call:
    """attn2.to_q (LayerNorm folded in) + attention over the text context in one launch (msd_cross_attention_q)."""
    lib = _lib.load()
    a = _lib.MsdCrossAttnQ()
    a.x, a.ln_in, a.wq, a.ln_colsum, a.bias, a.k, a.vt, a.out = _p(x), _p(ln_in), _p(wq), _p(ln_colsum), _p(bias), _p(k), _p(vt), _p(out)
    a.batch, a.heads, a.head_dim, a.s, a.t = batch, heads, head_dim, s, t
    a.k_ld, a.vt_ld, a.o_ld, a.ln_in_slots, a.ln_eps, a.w_layout = k_ld, vt_ld, o_ld, ln_in_slots, float(ln_eps), int(w_layout)
    return Call(lib.msd_cross_attention_q, (C.byref(a),), name, keep=a)


def softmax_rows(*, x, out, rows, cols, ld_in, ld_out, scale, name="softmax_rows") -> Call:
    lib = _lib.load()
    return Call(lib.msd_softmax_rows, (_p(x), _p(out), rows, cols, ld_in, ld_out, C.c_float(scale)), name)


def cfg_step(*, eps, latent, coef, step_ptr, batch, n, num_steps, guidance, guidance_rescale, advance=True,
             inpaint_init=None, inpaint_noise=None, inpaint_mask=None, step_noise=None, noise_coef=None,
             name="cfg_step") -> Call:
    lib = _lib.load()
    s = _lib.MsdCfgStep()
    s.eps, s.latent, s.coef, s.step_ptr = _p(eps), _p(latent), _p(coef), _p(step_ptr)
    s.inpaint_init, s.inpaint_noise, s.inpaint_mask = _p(inpaint_init), _p(inpaint_noise), _p(inpaint_mask)
    s.step_noise, s.noise_coef = _p(step_noise), _p(noise_coef)
    s.batch, s.n, s.num_steps = batch, n, num_steps
    s.guidance, s.guidance_rescale, s.advance = float(guidance), float(guidance_rescale), int(advance)   # 2: in-kernel ({step, ticket})
    return Call(lib.msd_cfg_step, (C.byref(s),), name, keep=s)


def add_bf16(*, a, b, out, n, name="add_bf16") -> Call:
    lib = _lib.load()
    return Call(lib.msd_add_bf16, (_p(a), _p(b), _p(out), n), name)


def add_f32_bf16(*, a, b, out, n, name="add_f32_bf16") -> Call:
    lib = _lib.load()
    return Call(lib.msd_add_f32_bf16, (_p(a), _p(b), _p(out), n), name)


def cast_f32_to_bf16(*, x, out, n, name="cast_f2b") -> Call:
    lib = _lib.load()
    return Call(lib.msd_cast_f32_to_bf16, (_p(x), _p(out), n), name)


def cast_bf16_to_f32(*, x, out, n, name="cast_b2f") -> Call:
    lib = _lib.load()
    return Call(lib.msd_cast_bf16_to_f32, (_p(x), _p(out), n), name)


def embedding_sum(*, tokens, positions, tok_table, pos_table, out, rows, dim, vocab, max_len, status=None,
                  name="embedding_sum") -> Call:
    lib = _lib.load()
    return Call(lib.msd_embedding_sum, (_p(tokens), _p(positions), _p(tok_table), _p(pos_table), _p(out), rows, dim, vocab, max_len,
                                        _p(status)), name)


def replicate(*, src, dst, nbytes, copies, name="replicate") -> Call:
    """dst = `copies` replicas of src's nbytes back to back (src may be dst: replica 0 in place)."""
    lib = _lib.load()
    return Call(lib.msd_replicate, (_p(src), _p(dst), int(nbytes), int(copies)), name)


def memset_zero(*, ptr, nbytes, name="memset") -> Call:
    lib = _lib.load()
    return Call(lib.msd_memset_zero, (_p(ptr), nbytes), name)

"""Weight pre-packing: Keras-layout fp32 weights -> the layouts the gfx950 kernels consume.

Done once at ``set_weights`` time on the host (layout only, no arithmetic beyond the bf16 round):

* MFMA path (``msd_conv_gemm``): ``W[N][K]`` bf16, K contiguous, ``k = (ky*ks + kx)*C_in + c`` —
  i.e. HWIO ``(kh,kw,cin,cout)`` -> ``(cout,kh,kw,cin)``; Dense ``(in,out)`` -> ``(out,in)``.
* GEGLU (diffusion_model.py:142-153): the ``8C`` projection rows are interleaved in 16-wide
  ``x | gate`` groups so both halves of an output element land in one lane's accumulators.
* q|k|v (diffusion_model.py:102-104): the three bias-free projections are stacked into one
  ``[3C][C]`` matrix so one GEMM feeds the attention kernel (q, k row-major, v transposed).
* direct path (``msd_conv_direct``): fp32 Keras layout unchanged.
"""
from __future__ import annotations

import numpy as np
import torch


def _dev_bf16(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(torch.bfloat16).contiguous().to(device)


class PackedWeights(dict):
    """name -> device tensor, plus the storage layout of every msd_conv_gemm / msd_cross_attention_q weight matrix in it:
    ``layout(key)`` is the value for MsdConvGemm.w_layout (0: [N][K] rows, 1: chunk-major).  The layout is recorded per
    key when the tensor is re-laid out, so an op can never be told a layout its operand does not have."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.chunk_major_keys = set()
        self._fragment = {}   # key -> fragment-major copy (w_layout 2), made when a launch plan first asks for it

    def layout(self, key: str) -> int:
        return 1 if key in self.chunk_major_keys else 0

    def to_chunk_major(self, key: str) -> None:
        t = self[key]
        if key in self.chunk_major_keys or t.dim() != 2 or t.shape[1] % 64:
            return   # (a K that is no multiple of 64 stays in rows: the kernels take either layout)
        self[key] = chunk_major(t)
        self.chunk_major_keys.add(key)


    def fragment_major(self, key: str) -> torch.Tensor:
        """The fragment-major image (MsdConvGemm.w_layout = 2, fragment_major below) of weight matrix `key`, for the launches
        the tuning table sends to the wreg form (csrc/conv_wreg.hip).  A second copy beside the stored one, made on first use and
        kept: which kernel reads a matrix is a per-(shape, batch) decision of the table, the other forms (halo, row-panel, the
        fused cross-attention) read rows or chunk-major, and 1.7 GB of weights are nothing in 288 GB of HBM."""
        t = self._fragment.get(key)
        if t is None:
            w = self[key]
            if key in self.chunk_major_keys:   # [K/64][N][64] -> [N][K]
                w = w.permute(1, 0, 2).reshape(w.shape[1], -1)
            t = self._fragment[key] = fragment_major(w)
        return t


FRAGMENT_ROW_ORDER = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)   # MFMA row r <- weight row of the 16-column block (cg_wrow, conv_common.h)


def fragment_major(w_nk: torch.Tensor) -> torch.Tensor:
    """[N][K] -> [K/64][N/16][2][64 lanes][8] (MsdConvGemm.w_layout = 2): the operand registers of `v_mfma_f32_16x16x32_bf16`
    laid out in memory.  Lane l = 16 g + r of the fragment (K tile kt, column block nb, half ks) holds the 8 values
    W[16 nb + FRAGMENT_ROW_ORDER[r]][64 kt + 32 ks + 8 g .. + 8], so a wave loads a fragment as ONE contiguous, aligned KiB
    (global_load_dwordx4, lane l at byte 16 l) and a wave's NJ column blocks of a K tile are 2 NJ consecutive KiB.  Same
    values, same K order as the row / chunk-major layouts: a launch that reads it (csrc/conv_wreg.hip) computes the same bits."""
    n, k = w_nk.shape
    assert n % 16 == 0 and k % 64 == 0, (n, k)
    order = torch.tensor(FRAGMENT_ROW_ORDER, device=w_nk.device)
    w = w_nk.reshape(n // 16, 16, k)[:, order, :]                       # (nb, r, k)
    w = w.reshape(n // 16, 16, k // 64, 2, 4, 8)                        # (nb, r, kt, ks, g, e)
    return w.permute(2, 0, 3, 4, 1, 5).contiguous()                     # (kt, nb, ks, g, r, e)


def chunk_major(w_nk: torch.Tensor) -> torch.Tensor:
    """[N][K] -> [K/64][N][64] (MsdConvGemm.w_layout = 1): the 64-element K chunk of ALL output columns is one contiguous
    run, so the weight tile of a K step is a single block of HBM whatever the column tile.  Same values, same K order."""
    n, k = w_nk.shape
    assert k % 64 == 0, (n, k)
    return w_nk.view(n, k // 64, 64).permute(1, 0, 2).contiguous()


def pack_conv(w_hwio: np.ndarray, device) -> torch.Tensor:
    """(kh,kw,cin,cout) fp32 -> [cout][kh*kw*cin] bf16 on device."""
    t = torch.from_numpy(np.ascontiguousarray(w_hwio))
    kh, kw, cin, cout = t.shape
    return _dev_bf16(t.permute(3, 0, 1, 2).reshape(cout, kh * kw * cin), device)


def pack_dense(w_io: np.ndarray, device) -> torch.Tensor:
    """(in,out) fp32 -> [out][in] bf16 on device."""
    return _dev_bf16(torch.from_numpy(np.ascontiguousarray(w_io)).t(), device)


def pack_dense_stack(ws, device) -> torch.Tensor:
    """Stack several (in,out_i) matrices along the output axis -> [sum out_i][in] bf16."""
    return _dev_bf16(torch.cat([torch.from_numpy(np.ascontiguousarray(w)).t() for w in ws], dim=0), device)


def geglu_row_order(n_half: int) -> np.ndarray:
    """Packed row p -> original column of the (in, 2*n_half) GEGLU projection.

    Packed rows [32i, 32i+16) are value columns [16i, 16i+16); rows [32i+16, 32i+32) are the gate
    columns n_half + [16i, 16i+16)."""
    assert n_half % 16 == 0
    p = np.arange(2 * n_half)
    grp, within = p // 32, p % 32
    return np.where(within < 16, grp * 16 + within, n_half + grp * 16 + (within - 16))


def pack_geglu(w_io: np.ndarray, b: np.ndarray, device):
    n_half = w_io.shape[1] // 2
    order = geglu_row_order(n_half)
    wt = torch.from_numpy(np.ascontiguousarray(w_io)).t()[torch.from_numpy(order)]
    return _dev_bf16(wt, device), dev_f32(np.asarray(b)[order], device)


def fold_layer_norm(w_oi: torch.Tensor, bias, gamma: np.ndarray, beta: np.ndarray, device):
    """LayerNormalization (diffusion_model.py:84-88) folded into the Dense that follows it.

    ``w_oi``: fp32 [out][in] (already in the row order the kernel wants); ``bias``: fp32 [out] in the same
    order or None.  LN(x) W^T + b = rstd * (x (gamma*W)^T - mean * colsum) + (W beta + b), so this returns
    (bf16 gamma-folded weights on device, fp32 colsum of the ROUNDED folded weights, fp32 W beta + b)."""
    g = torch.from_numpy(np.ascontiguousarray(gamma, dtype=np.float32))
    bt = torch.from_numpy(np.ascontiguousarray(beta, dtype=np.float32))
    wf = (w_oi * g[None, :]).to(torch.bfloat16)
    colsum = wf.to(torch.float64).sum(dim=1).to(torch.float32)
    c = (w_oi.to(torch.float64) @ bt.to(torch.float64)).to(torch.float32)
    if bias is not None:
        c = c + torch.from_numpy(np.ascontiguousarray(bias, dtype=np.float32))
    return wf.contiguous().to(device), colsum.contiguous().to(device), c.contiguous().to(device)


def dev_f32(a: np.ndarray, device) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)


# ---- packed weights on disk (one rank packs, the others map) ----------------------------------------------------------------
def save_packed(W: PackedWeights, path: str, meta: dict = None) -> None:
    """Write a PackedWeights (the device tensors the launch plans read: bf16 matrices in their stored layout, fp32 vectors)
    to `path`, atomically (temp file + rename: a reader never sees a partial file).  Meant for a RAM-backed directory
    (/dev/shm): the ranks of one node then map what rank 0 packed instead of each generating and packing its own copy
    (bench.py, N > 1).  Fragment-major copies are not saved: every plan makes the ones its table asks for."""
    import os

    blob = {"meta": dict(meta or {}), "chunk_major": sorted(W.chunk_major_keys),
            "tensors": {k: (t.detach().cpu() if isinstance(t, torch.Tensor) else t) for k, t in W.items()}}
    tmp = f"{path}.tmp.{os.getpid()}"
    torch.save(blob, tmp)
    os.replace(tmp, path)


def load_packed(path: str, device, meta: dict = None) -> PackedWeights:
    """The inverse of save_packed, onto `device`; `meta` (if given) must equal what the writer recorded (network kind,
    table arguments, seed): a file of another model raises instead of being launched on."""
    blob = torch.load(path, map_location="cpu", mmap=True, weights_only=True)
    if meta is not None and blob["meta"] != dict(meta):
        raise ValueError(f"{path}: packed for {blob['meta']}, wanted {dict(meta)}")
    W = PackedWeights({k: (t.to(device) if isinstance(t, torch.Tensor) else t) for k, t in blob["tensors"].items()})
    W.chunk_major_keys = set(blob["chunk_major"])
    return W

"""ctypes binding of libminsdtf_hip.so (C ABI declared in include/minsdtf_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If the shared object is
missing or does not export the ABI the header declares, importing any op raises
:class:`HipExtensionError` (loudly) instead of silently computing on some other path.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
import sys
import weakref

LIB_NAME = "libminsdtf_hip.so"
# $MSD_HIP_LIB: another build of the same library (A/B runs of two kernel versions on one box); default = the in-tree build
LIB_PATH = os.environ.get("MSD_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)
ABI_VERSION = 11

ACT_NONE, ACT_SILU, ACT_GEGLU, ACT_QUICK_GELU = 0, 1, 2, 3
OUT_BF16, OUT_F32, OUT_U8 = 0, 1, 2


class HipExtensionError(RuntimeError):
    pass


class MsdConvGemm(C.Structure):
    _fields_ = [
        ("a0", C.c_void_p), ("a1", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("rowvec", C.c_void_p),
        ("step_ptr", C.c_void_p), ("residual", C.c_void_p), ("out", C.c_void_p), ("out1", C.c_void_p),
        ("out2", C.c_void_p), ("workspace", C.c_void_p), ("workspace_floats", C.c_int64),
        ("batch", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32), ("c0", C.c_int32), ("c1", C.c_int32),
        ("h_out", C.c_int32), ("w_out", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("upsample", C.c_int32), ("N", C.c_int32), ("act", C.c_int32), ("out_dtype", C.c_int32),
        ("out_ld", C.c_int32), ("res_ld", C.c_int32), ("rv_step_stride", C.c_int32), ("rv_batch_stride", C.c_int32),
        ("split_mode", C.c_int32), ("ns0", C.c_int32), ("ns1", C.c_int32), ("out1_ld", C.c_int32),
        ("out2_ld", C.c_int32), ("splitk", C.c_int32), ("tile_n", C.c_int32), ("tile_m", C.c_int32), ("stages", C.c_int32),
        ("ln_in", C.c_void_p), ("ln_colsum", C.c_void_p), ("ln_out", C.c_void_p),
        ("ln_in_slots", C.c_int32), ("ln_out_slots", C.c_int32), ("ln_eps", C.c_float),
        ("a2", C.c_void_p), ("a3", C.c_void_p), ("c2", C.c_int32), ("c3", C.c_int32), ("w_layout", C.c_int32),
    ]


class MsdConvDirect(C.Structure):
    _fields_ = [
        ("in_", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p), ("out", C.c_void_p),
        ("batch", C.c_int32), ("in_batch_mod", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32),
        ("c_in", C.c_int32), ("h_out", C.c_int32), ("w_out", C.c_int32), ("c_out", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("in_dtype", C.c_int32),
        ("out_dtype", C.c_int32), ("act", C.c_int32), ("act_in", C.c_int32), ("in_scale", C.c_float),
    ]


class MsdGroupNorm(C.Structure):
    _fields_ = [
        ("x0", C.c_void_p), ("x1", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("stats", C.c_void_p),
        ("partials", C.c_void_p), ("partials_floats", C.c_int64),
        ("out", C.c_void_p), ("batch", C.c_int32), ("hw", C.c_int32), ("c0", C.c_int32), ("c1", C.c_int32),
        ("silu", C.c_int32), ("eps", C.c_float), ("sync", C.c_void_p), ("sync_words", C.c_int64),
    ]


class MsdCrossAttnQ(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ln_in", C.c_void_p), ("wq", C.c_void_p), ("ln_colsum", C.c_void_p), ("bias", C.c_void_p),
        ("k", C.c_void_p), ("vt", C.c_void_p), ("out", C.c_void_p),
        ("batch", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32), ("s", C.c_int32), ("t", C.c_int32),
        ("k_ld", C.c_int32), ("vt_ld", C.c_int32), ("o_ld", C.c_int32), ("ln_in_slots", C.c_int32), ("ln_eps", C.c_float),
        ("w_layout", C.c_int32),
    ]


class MsdAttention(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("vt", C.c_void_p), ("out", C.c_void_p),
        ("batch", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32), ("s", C.c_int32), ("t", C.c_int32),
        ("q_ld", C.c_int32), ("k_ld", C.c_int32), ("vt_ld", C.c_int32), ("o_ld", C.c_int32), ("scale", C.c_float),
        ("causal", C.c_int32), ("q_prescaled", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_floats", C.c_int64),   # ABI 9: key-split scratch of the d = 512 kernel, or NULL
    ]


class MsdCfgStep(C.Structure):
    _fields_ = [
        ("eps", C.c_void_p), ("latent", C.c_void_p), ("coef", C.c_void_p), ("step_ptr", C.c_void_p),
        ("batch", C.c_int32), ("n", C.c_int32), ("num_steps", C.c_int32), ("guidance", C.c_float),
        ("guidance_rescale", C.c_float), ("advance", C.c_int32),
        ("inpaint_init", C.c_void_p), ("inpaint_noise", C.c_void_p), ("inpaint_mask", C.c_void_p),
        ("step_noise", C.c_void_p), ("noise_coef", C.c_void_p),
    ]


# every symbol include/minsdtf_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "msd_abi_version": (C.c_int, []),
    "msd_last_error": (C.c_char_p, []),
    "msd_init": (C.c_int, []),
    "msd_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "msd_conv_gemm": (C.c_int, [C.POINTER(MsdConvGemm), C.c_void_p]),
    "msd_conv_gemm_ln_slots": (C.c_int, [C.POINTER(MsdConvGemm)]),
    "msd_conv_direct": (C.c_int, [C.POINTER(MsdConvDirect), C.c_void_p]),
    "msd_group_norm": (C.c_int, [C.POINTER(MsdGroupNorm), C.c_void_p]),
    "msd_cross_attention_q": (C.c_int, [C.POINTER(MsdCrossAttnQ), C.c_void_p]),
    "msd_layer_norm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                 C.c_void_p]),
    "msd_attention": (C.c_int, [C.POINTER(MsdAttention), C.c_void_p]),
    "msd_softmax_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                   C.c_void_p]),
    "msd_memset_zero": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "msd_replicate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "msd_embedding_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_void_p, C.c_void_p]),
    "msd_cfg_step": (C.c_int, [C.POINTER(MsdCfgStep), C.c_void_p]),
    "msd_add_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "msd_add_f32_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "msd_cast_f32_to_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "msd_cast_bf16_to_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
}

_lib = None

# ---- orderly shutdown ---------------------------------------------------------------------------------------------------
# Captured hipGraphs hold kernel nodes of this library's code object.  At process exit the C runtime unregisters the code
# object (an atexit handler installed when the .so was loaded, i.e. AFTER torch's: it runs BEFORE torch's static
# destructors), and only then would torch destroy whatever CUDAGraph objects are still alive — a graph whose kernels'
# module is already gone.  So everything that owns a graph registers here, and `shutdown()` (also run by Python's own
# atexit, which precedes the C-level handlers) releases the graphs first, then drains the device.
_graph_owners = weakref.WeakSet()
_shutdown_registered = False


def track_graph_owner(owner) -> None:
    """`owner.release_graphs()` will be called by shutdown()."""
    _graph_owners.add(owner)


def shutdown() -> None:
    """Destroy every captured hipGraph of this package and wait for the device (idempotent; engines and models stay
    usable: their graphs are re-captured on the next use)."""
    import gc

    for o in list(_graph_owners):
        try:
            o.release_graphs()
        except Exception:   # pragma: no cover - best effort at exit
            pass
    gc.collect()
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.synchronize()


def _apply_env_options(lib) -> None:
    """The one process-wide arithmetic choice made through the environment, validated before the library handle is cached:
    MSD_GN_ROWS=<pixels> - the row-major cluster GroupNorm from that sample size on (default 9216; 4096 pays from
    two images per GPU and costs at one).  Under either setting a sample's bits do not depend on its batch.  MSD_PROFILE (round
    5's second tuning table) no longer exists: the layers it moved are in ONE numerics class at every batch since round 6, and
    a process that still sets it is told so instead of silently running something else."""
    if os.environ.get("MSD_PROFILE") not in (None, "", "latency"):
        raise HipExtensionError(f"MSD_PROFILE={os.environ['MSD_PROFILE']!r}: the throughput profile was removed in round 6 (its launch "
                                "configurations are in the one tuning table now); unset it.  MSD_GN_ROWS=4096 is still available.")
    gn_rows = os.environ.get("MSD_GN_ROWS") or ""
    if not gn_rows:
        return
    try:
        rows = int(gn_rows)
    except ValueError as e:
        raise HipExtensionError(f"MSD_GN_ROWS={gn_rows!r}: a pixel count (integer >= 0)") from e
    if lib.msd_set_option(b"gn_rows", rows) != 0:
        raise HipExtensionError(f"MSD_GN_ROWS={gn_rows}: {lib.msd_last_error().decode(errors='replace')}")


def load() -> C.CDLL:
    """Load (once) and type the shared library.  Raises HipExtensionError when it is unusable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no fallback path.")
    import torch  # noqa: F401  — torch's bundled libamdhip64.so.7 must be the HIP runtime the library binds to

    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise HipExtensionError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipExtensionError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.msd_abi_version() != ABI_VERSION:
        raise HipExtensionError(f"ABI version mismatch: library {lib.msd_abi_version()} != binding {ABI_VERSION}")
    _apply_env_options(lib)   # (raises: the library is NOT cached then, so the next load() fails the same way instead of running on defaults)
    _lib = lib
    global _shutdown_registered
    if not _shutdown_registered:
        atexit.register(shutdown)
        _shutdown_registered = True
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().msd_last_error().decode(errors="replace")
        raise HipExtensionError(f"{what} failed (rc={rc}): {msg}")

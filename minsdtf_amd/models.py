"""Drop-in model objects: the duck type ``StableDiffusion`` uses for its Keras models.

The reference pipeline only touches its models through ``ctor(...)``, ``.compile(jit_compile=True)``,
``.predict_on_batch(x)`` and — from the checkpoint loader — ``.name``, ``.weights`` (ordered, Keras
layout, with ``.shape`` / ``.name``) and ``.set_weights(list)`` (reference
``stable_diffusion.py:650-725``, ``ckpt_loader.py:2136-2193``).  The classes below provide exactly
that surface over the HIP library:

=================  =============================================  ==========================
class              replaces (reference)                           predict_on_batch
=================  =============================================  ==========================
DiffusionModel     diffusion_model.py:163-296                     [latent, t_emb, context(, 13 controls)] -> (B,h,w,4)
ImageDecoder       image_decoder.py:22-66                         latent -> (B,8h,8w,3)
ControlNet         control_net.py:45-118                          [latent, t_emb, context, hint] -> 13 arrays
HintNet            control_net.py:10-42                           (B,H,W,3) -> (B,H/8,W/8,320)
=================  =============================================  ==========================

Inputs / outputs at this boundary are host numpy arrays like the reference's; the fused,
device-resident loop (``minsdtf_amd/pipeline.py``) bypasses the boundary and keeps everything in
HBM.  There is no CPU fallback: constructing a model without a GPU + built library raises.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib, engine, ops, packing
from . import weights as wtab


class WeightVar:
    """What ``model.weights[i]`` needs to be for the positional loader: a name and a Keras shape."""
    __slots__ = ("name", "shape")

    def __init__(self, name, shape):
        self.name, self.shape = name, tuple(shape)

    def __repr__(self):
        return f"<WeightVar {self.name} {self.shape}>"


def default_device() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.HipExtensionError("no HIP device visible: the MI355X path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


# names whose convs / denses run on the fp32 vector-FMA path (channel counts below an MFMA tile)
_DIRECT = {
    "conv_in", "conv_out", "time_embedding.linear_1", "time_embedding.linear_2",
    "post_quant_conv", "decoder.conv_in", "decoder.conv_out", "encoder.conv_in", "quant_conv",
} | {f"input_hint_block.{i}" for i in range(7)}


UNET_HEADS = 8   # CrossAttention(num_heads=8) everywhere in the UNet / ControlNet (diffusion_model.py:60-65)


def _q_prescale(c_out: int) -> np.float32:
    """Factor folded into the UNet's query projections (attn1.to_q, attn2.to_q) at pack time: the attention scale
    head_size**-0.5 (diffusion_model.py:105,123) times log2(e), so the attention kernel takes exp2 of q k^T directly
    (MsdAttention.q_prescaled).  Exact up to the bf16 rounding of the weights, which happens once either way."""
    return np.float32((c_out // UNET_HEADS) ** -0.5 * 1.4426950408889634)


class HipModel:
    kind = ""  # weight-table kind

    def __init__(self, name=None, device=None):
        self.name = name or type(self).__name__.lower()
        self.device = device if device is not None else default_device()
        lib = _lib.load()
        _lib.check(lib.msd_init(), "msd_init")
        self._specs = wtab.table(self.kind, **self._table_kw())
        self.weights: List[WeightVar] = [
            WeightVar(s.name + (".kernel" if s.kind.endswith("_w") else "." + s.kind), s.shape) for s in self._specs]
        self._W: Optional[Dict[str, torch.Tensor]] = None
        self._plans: Dict[tuple, "_BoundPlan"] = {}
        self._use_graph = False
        # bumped by every set_weights(): launch plans hold raw device addresses of the packed weights, so whatever caches
        # a plan built on this model (its own _plans, StableDiffusion._engines) keys on it
        self.weights_version = 0

    def _table_kw(self) -> dict:
        return {}

    # ---- Keras-like surface
    def compile(self, jit_compile=True, **_):
        """The reference calls ``compile(jit_compile=True)``; here it switches predict_on_batch to
        hipGraph replay (captured on first use per input shape)."""
        self._use_graph = bool(jit_compile)
        self._plans.clear()

    def count_params(self) -> int:
        return wtab.param_count(self.kind)

    def set_weights(self, arrays: Sequence[np.ndarray]) -> None:
        if len(arrays) != len(self._specs):
            raise ValueError(f"{self.name}: expected {len(self._specs)} weight arrays, got {len(arrays)}")
        named = {}
        for s, a in zip(self._specs, arrays):
            a = np.asarray(a)
            if tuple(a.shape) != tuple(s.shape):
                raise ValueError(f"{self.name}: {s.name} has shape {a.shape}, expected {s.shape}")
            named[(s.name, s.kind)] = a
        W = packing.PackedWeights(self._pack(named))
        if engine.W_CHUNK_MAJOR:
            # the bf16 matrices the MFMA kernels read (msd_conv_gemm, msd_cross_attention_q: keys *.w / *.lnw) are stored
            # chunk-major; W records which ones, and the emitters pass that per-key layout to the op (Emitter.conv)
            for k in [k for k, t in W.items() if t.dtype == torch.bfloat16 and t.dim() == 2 and k.endswith((".w", ".lnw"))]:
                W.to_chunk_major(k)
        self._W = W
        self.weights_version += 1
        self._plans.clear()

    def share_weights(self, other: "HipModel") -> None:
        """Use `other`'s packed device weights (same network kind, e.g. one checkpoint served at two image sizes):
        no second copy in HBM, no second packing pass."""
        if type(other) is not type(self) or other._W is None:
            raise ValueError(f"{self.name}: share_weights needs a loaded model of the same kind")
        if other.device != self.device:   # the plans hold raw device addresses
            raise ValueError(f"{self.name}: share_weights across devices ({other.device} -> {self.device})")
        self._W = other._W
        self.weights_version += 1
        self._plans.clear()

    def load_synthetic(self, seed=0, bias_scale=0.0) -> List[np.ndarray]:
        """Fill with the seeded synthetic checkpoint; returns the Keras-layout list."""
        arrays = wtab.synth_keras_weights(self.kind, seed=seed, bias_scale=bias_scale, **self._table_kw())
        self.set_weights(arrays)
        return arrays

    def save_packed(self, path: str, **meta) -> None:
        """The packed device weights of this model -> `path` (packing.save_packed; `meta`: whatever identifies the
        checkpoint, compared by load_packed)."""
        self._require_weights()
        packing.save_packed(self._W, path, dict(meta, kind=self.kind, table_kw=self._table_kw()))

    def load_packed(self, path: str, **meta) -> None:
        """Adopt weights another process packed (save_packed): no generation, no packing pass, one upload."""
        self._W = packing.load_packed(path, self.device, dict(meta, kind=self.kind, table_kw=self._table_kw()))
        self.weights_version += 1
        self._plans.clear()

    def _maybe_load(self, ckpt_path, lora_dict=None):
        """Reference constructors load a local checkpoint if given one; no network download here."""
        if ckpt_path is not None and os.path.exists(ckpt_path):
            wtab.load_weights_from_file(self, ckpt_path, self.kind, lora_dict=lora_dict, specs=self._specs)

    # ---- packing
    def _pack(self, named) -> Dict[str, torch.Tensor]:
        d = self.device
        W: Dict[str, torch.Tensor] = {}
        get = lambda n, k: named.get((n, k))  # noqa: E731
        names = []
        for s in self._specs:
            if s.name not in names:
                names.append(s.name)
        tproj_w, tproj_b = [], []
        for n in names:
            cw, dw = get(n, "conv_w"), get(n, "dense_w")
            b, g, beta = get(n, "bias"), get(n, "gamma"), get(n, "beta")
            if g is not None:
                W[n + ".g"], W[n + ".b"] = packing.dev_f32(g, d), packing.dev_f32(beta, d)
                continue
            if n.endswith(".time_emb_proj"):
                tproj_w.append(dw)
                tproj_b.append(b)
                continue
            if n in _DIRECT:
                w = cw if cw is not None else dw.reshape(1, 1, *dw.shape)
                W[n + ".w"] = packing.dev_f32(w, d)
                W[n + ".b"] = packing.dev_f32(b, d)
                if n == "conv_out" and cw is not None and cw.shape[2] % 64 == 0:
                    # 320 -> 4: K = 2880 fills MFMA tiles even though N does not; 48 -> ~15 us per step (engine.MFMA_CONV_OUT)
                    W[n + ".m.w"], W[n + ".m.b"] = packing.pack_conv(cw, d), W[n + ".b"]
                continue
            if n.endswith(".ff.net.0.proj"):
                W[n + ".w"], W[n + ".b"] = packing.pack_geglu(dw, b, d)
                continue
            if n.endswith((".attn1.to_q", ".attn1.to_k", ".attn1.to_v", ".attn2.to_k", ".attn2.to_v",
                           ".query", ".key", ".value", ".self_attn.q_proj", ".self_attn.k_proj", ".self_attn.v_proj")):
                continue  # stacked below
            if cw is not None:
                W[n + ".w"] = packing.pack_conv(cw, d)
            else:
                W[n + ".w"] = packing.pack_dense(dw, d)
            if b is not None:
                W[n + ".b"] = packing.dev_f32(b, d)
        for n in names:
            if n.endswith(".attn1.to_q"):
                base = n[: -len(".to_q")]
                wq = get(base + ".to_q", "dense_w")
                W[base + ".qkv.w"] = packing.pack_dense_stack(
                    [wq * _q_prescale(wq.shape[1]), get(base + ".to_k", "dense_w"), get(base + ".to_v", "dense_w")], d)
            elif n.endswith(".attn2.to_q"):   # (the generic branch above packed it unscaled)
                wq = get(n, "dense_w")
                W[n + ".w"] = packing.pack_dense(wq * _q_prescale(wq.shape[1]), d)
            elif n.endswith(".attn2.to_k"):
                base = n[: -len(".to_k")]
                W[base + ".kv.w"] = packing.pack_dense_stack([get(base + ".to_k", "dense_w"), get(base + ".to_v", "dense_w")], d)
            elif n.endswith(".query") or n.endswith(".self_attn.q_proj"):
                trio = ("query", "key", "value") if n.endswith(".query") else ("q_proj", "k_proj", "v_proj")
                base = n[: -len("." + trio[0])]
                W[base + ".qkv.w"] = packing.pack_dense_stack([get(base + "." + k, "dense_w") for k in trio], d)
                W[base + ".qkv.b"] = packing.dev_f32(np.concatenate([get(base + "." + k, "bias") for k in trio]), d)
        # LayerNorm folds of the transformer blocks (engine.Emitter.attentions): norm1 -> q|k|v, norm2 -> attn2.to_q,
        # norm3 -> GEGLU projection.  Keys: <consumer>.lnw (bf16 gamma-folded), .lncs (column sums), .lnb (W beta + b)
        for n in names:
            if not n.endswith(".transformer_blocks.0.norm1"):
                continue
            tb = n[: -len(".norm1")]
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).t().contiguous()  # noqa: E731  (in,out) -> [out][in]
            cq = _q_prescale(get(tb + ".attn1.to_q", "dense_w").shape[1])
            qkv = torch.cat([t(get(tb + ".attn1." + k, "dense_w")) * (cq if k == "to_q" else 1.0) for k in ("to_q", "to_k", "to_v")],
                            dim=0)
            W[tb + ".attn1.qkv.lnw"], W[tb + ".attn1.qkv.lncs"], W[tb + ".attn1.qkv.lnb"] = packing.fold_layer_norm(
                qkv, None, get(tb + ".norm1", "gamma"), get(tb + ".norm1", "beta"), d)
            W[tb + ".attn2.to_q.lnw"], W[tb + ".attn2.to_q.lncs"], W[tb + ".attn2.to_q.lnb"] = packing.fold_layer_norm(
                t(get(tb + ".attn2.to_q", "dense_w")) * cq, None, get(tb + ".norm2", "gamma"), get(tb + ".norm2", "beta"), d)
            gw, gb = get(tb + ".ff.net.0.proj", "dense_w"), get(tb + ".ff.net.0.proj", "bias")
            order = packing.geglu_row_order(gw.shape[1] // 2)
            W[tb + ".ff.net.0.proj.lnw"], W[tb + ".ff.net.0.proj.lncs"], W[tb + ".ff.net.0.proj.lnb"] = packing.fold_layer_norm(
                t(gw)[torch.from_numpy(order)], np.asarray(gb)[order], get(tb + ".norm3", "gamma"), get(tb + ".norm3", "beta"), d)
        # ff.net.2 followed by proj_out (diffusion_model.py:146-147 and :66-67: two Dense layers with only the
        # residual add of t2 between them) as ONE GEMM over the channel concat [ff | t2]:
        #   proj_out(ff W2 + b2 + t2) = ff (W2 Wp) + t2 Wp + (b2 Wp + bp)        key <attentions>.ffproj
        for n in names:
            if not n.endswith(".transformer_blocks.0.ff.net.2"):
                continue
            att = n[: -len(".transformer_blocks.0.ff.net.2")]
            w2, b2 = get(n, "dense_w"), get(n, "bias")
            wp, bp = get(att + ".proj_out", "conv_w"), get(att + ".proj_out", "bias")
            if w2 is None or wp is None:
                continue
            w2d, wpd = np.asarray(w2, np.float64), np.asarray(wp, np.float64).reshape(wp.shape[-2], wp.shape[-1])
            wcat = np.concatenate([w2d @ wpd, wpd], axis=0)                       # (4C + C, C) as (in, out)
            W[att + ".ffproj.w"] = packing.pack_dense(wcat.astype(np.float32), d)
            W[att + ".ffproj.b"] = packing.dev_f32((np.asarray(b2, np.float64) @ wpd + np.asarray(bp, np.float64)).astype(np.float32), d)
        # ResBlock conv2 + conv_shortcut (diffusion_model.py:34-38,50) as one contraction: W = [conv2 taps | shortcut], b = b2 + bs
        for n in names:
            if not n.endswith(".conv_shortcut"):
                continue
            rb = n[: -len(".conv_shortcut")]
            w2, b2 = get(rb + ".conv2", "conv_w"), get(rb + ".conv2", "bias")
            ws, bs = get(n, "conv_w"), get(n, "bias")
            if w2 is None or ws is None or w2.shape[0] != 3 or ws.shape[0] != 1 or ws.shape[2] % 64:
                continue
            t2 = torch.from_numpy(np.ascontiguousarray(w2)).permute(3, 0, 1, 2).reshape(w2.shape[3], -1)      # [N][9 C_out]
            tsc = torch.from_numpy(np.ascontiguousarray(ws)).permute(3, 0, 1, 2).reshape(ws.shape[3], -1)     # [N][C_in]
            W[rb + ".conv2sc.w"] = torch.cat([t2, tsc], dim=1).to(torch.bfloat16).contiguous().to(d)
            W[rb + ".conv2sc.b"] = packing.dev_f32(np.asarray(b2, np.float32) + np.asarray(bs, np.float32), d)
        if tproj_w:
            cat = np.concatenate(tproj_w, axis=1)   # (1280, sum of the ResBlocks' C_out)
            W["time_emb_proj_cat.w"] = (packing.pack_dense(cat, d) if engine.MFMA_TEMB_PROJ
                                        else packing.dev_f32(cat.reshape(1, 1, 1280, -1), d))
            W["time_emb_proj_cat.b"] = packing.dev_f32(np.concatenate(tproj_b), d)
        return W

    def _require_weights(self):
        if self._W is None:
            raise RuntimeError(f"{self.name}: no weights set (pass ckpt_path=, call set_weights() or load_synthetic())")

    # ---- plan cache
    def _bound(self, key, builder) -> "_BoundPlan":
        key = (key, engine.GN_EPOCH)   # (a reported cluster-GroupNorm give-up retires every recorded plan: engine.check_gn_sync)
        engine.retire_stale(self._plans)   # ... and frees their arenas / graphs before the replacement is built
        bp = self._plans.get(key)
        if bp is None:
            self._require_weights()
            bp = builder()
            self._plans[key] = bp
        return bp


class _BoundPlan:
    """A finalised plan + its boundary tensors, optionally captured into a hipGraph."""

    def __init__(self, plan: engine.Plan, use_graph: bool):
        self.plan = plan
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.use_graph = use_graph
        self.io: Dict[str, torch.Tensor] = {}
        _lib.track_graph_owner(self)

    def release_graphs(self) -> None:
        self.graph = None

    def run(self):
        if not self.use_graph:
            self.plan.run(torch.cuda.current_stream().cuda_stream)
            return
        if self.graph is None:
            # warm-up run outside capture (first-touch of code objects), then capture
            self.plan.run(torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                with torch.cuda.graph(g, stream=s):
                    self.plan.run(torch.cuda.current_stream().cuda_stream)
            torch.cuda.current_stream().wait_stream(s)
            self.graph = g
        self.graph.replay()

    def host(self, name):
        """Boundary tensor `name` (or a list of names) as host array(s) — the duck type's D2H — followed by this plan's
        cluster-GroupNorm give-up word (engine.check_gn_sync: 4 more bytes on a stream the copy has just drained); raises
        instead of returning a result computed from abandoned moments."""
        names = [name] if isinstance(name, str) else list(name)
        out = [self.io[n].cpu().numpy() for n in names]
        buf = self.plan._gn_sync_buf
        if buf is not None and self.io[names[0]].device.type == "cuda":
            word = buf.tensor(torch.int32, (engine.GN_GIVE_UP_WORD + 1,))[engine.GN_GIVE_UP_WORD:]
            engine.check_gn_sync(word.cpu())
        return out[0] if isinstance(name, str) else out


def _np32(x) -> np.ndarray:
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x), dtype=np.float32)


def _stage_inputs(plan: engine.Plan, shapes: Dict[str, tuple]) -> Dict[str, engine.Buf]:
    return {k: plan.alloc(int(np.prod(s)) * 4) for k, s in shapes.items()}


class DiffusionModel(HipModel):
    """SD1.5 UNet (reference diffusion_model.py:163-296) on the HIP path."""
    kind = "civitai_model"

    def __init__(self, img_height=512, img_width=512, apply_control_net=False, name=None, ckpt_path=None, lora_dict=None,
                 device=None):
        super().__init__(name or "diffusion_model", device)
        if img_height % 64 or img_width % 64:
            raise ValueError("img_height / img_width must be multiples of 64 (three stride-2 levels after the /8 VAE)")
        self.h, self.w = img_height // 8, img_width // 8
        self.apply_control_net = apply_control_net
        self._maybe_load(ckpt_path, lora_dict)

    def _build(self, B: int, T: int, with_controls: bool) -> _BoundPlan:
        h, w = self.h, self.w
        plan = engine.Plan(self.device)
        e = engine.Emitter(plan, self._W)
        ins = _stage_inputs(plan, dict(latent=(B, h, w, 4), t_emb=(B, 320), context=(B, T, 768)))
        ctx16 = engine.Act(plan.alloc(B * T * 768 * 2), B, T, 1, 768)
        plan.rec(ops.cast_f32_to_bf16, x=ins["context"], out=ctx16.buf, n=B * T * 768, name="context.bf16")
        ctx_kv = engine.emit_context_kv(e, ctx16, engine.UNET_ATTN_LAYERS, plan)
        cols = engine.temb_columns(False)
        total = sum(c for _, c in engine.resblock_names(False))
        table = plan.alloc(B * total * 4)
        engine.emit_time_embedding(e, ins["t_emb"], B, table, encoder_only=False)
        controls, cstage = None, []
        if with_controls:   # fp32 staging buffers; emit_unet adds them to the skips in fp32 (one add-and-round launch each)
            controls = []
            for i, ch in enumerate(wtab.UNET_SKIP_CH + (1280,)):
                hh, ww = _skip_hw(i, h, w)
                st = plan.alloc(B * hh * ww * ch * 4)
                cstage.append((st, (B, hh, ww, ch)))
                controls.append(st)
        eps = plan.alloc(B * h * w * 4 * 4)
        engine.emit_unet(e, ins["latent"], B, B, h, w, (table, 0, total, cols), ctx_kv, T, eps, controls)
        plan.finalize()
        bp = _BoundPlan(plan, self._use_graph)
        bp.io = {k: b.tensor(torch.float32, s) for (k, b), s in
                 zip(ins.items(), [(B, h, w, 4), (B, 320), (B, T, 768)])}
        bp.io["eps"] = eps.tensor(torch.float32, (B, h, w, 4))
        for i, (st, shp) in enumerate(cstage):
            bp.io[f"control.{i}"] = st.tensor(torch.float32, shp)
        return bp

    def predict_on_batch(self, x):
        latent, t_emb, context = _np32(x[0]), _np32(x[1]), _np32(x[2])
        controls = [_np32(c) for c in x[3:]]
        if controls and len(controls) != 13:
            raise ValueError("expected 13 control tensors")
        B, T = latent.shape[0], context.shape[1]
        if latent.shape[1:] != (self.h, self.w, 4):
            raise ValueError(f"latent shape {latent.shape} does not match the model ({self.h},{self.w},4)")
        bp = self._bound((B, T, bool(controls)), lambda: self._build(B, T, bool(controls)))
        bp.io["latent"].copy_(torch.from_numpy(latent))
        bp.io["t_emb"].copy_(torch.from_numpy(t_emb))
        bp.io["context"].copy_(torch.from_numpy(context))
        for i, c in enumerate(controls):
            bp.io[f"control.{i}"].copy_(torch.from_numpy(c))
        bp.run()
        return bp.host("eps")

    __call__ = predict_on_batch


def _skip_hw(i: int, h: int, w: int):
    """Spatial size of skip / control tensor i (index 12 = mid block output)."""
    lvl = (0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 3)[i]
    return h >> lvl, w >> lvl


class ImageDecoder(HipModel):
    """VAE decoder (reference image_decoder.py:22-66) on the HIP path."""
    kind = "decoder"

    def __init__(self, name=None, ckpt_path=None, device=None):
        super().__init__(name or "image_decoder", device)
        self._maybe_load(ckpt_path)

    def _build(self, B, h, w, out_u8: bool) -> _BoundPlan:
        plan = engine.Plan(self.device)
        e = engine.Emitter(plan, self._W)
        lat = plan.alloc(B * h * w * 4 * 4)
        out = plan.alloc(B * 8 * h * 8 * w * 3 * (1 if out_u8 else 4))
        engine.emit_decoder(e, lat, B, h, w, out, ops.OUT_U8 if out_u8 else ops.OUT_F32)
        plan.finalize()
        bp = _BoundPlan(plan, self._use_graph)
        bp.io["latent"] = lat.tensor(torch.float32, (B, h, w, 4))
        bp.io["image"] = out.tensor(torch.uint8 if out_u8 else torch.float32, (B, 8 * h, 8 * w, 3))
        return bp

    def predict_on_batch(self, x):
        latent = _np32(x)
        B, h, w, _ = latent.shape
        bp = self._bound((B, h, w, False), lambda: self._build(B, h, w, False))
        bp.io["latent"].copy_(torch.from_numpy(latent))
        bp.run()
        return bp.host("image")

    def decode_to_uint8(self, latent_dev: torch.Tensor) -> torch.Tensor:
        """Device-resident variant used by the fused pipeline: fp32 latent tensor on the GPU ->
        uint8 image tensor on the GPU with the reference's truncating conversion fused in."""
        B, h, w, _ = latent_dev.shape
        bp = self._bound((B, h, w, True), lambda: self._build(B, h, w, True))
        bp.io["latent"].copy_(latent_dev)
        bp.run()
        return bp.io["image"]

    __call__ = predict_on_batch


class ControlNet(HipModel):
    """ControlNet (reference control_net.py:45-118) on the HIP path."""
    kind = "controlnet"

    def __init__(self, img_height=512, img_width=512, name=None, controlnet_path=None, device=None):
        super().__init__(name or "control_net", device)
        self.h, self.w = img_height // 8, img_width // 8
        self._maybe_load(controlnet_path)

    def _build(self, B: int, T: int) -> _BoundPlan:
        h, w = self.h, self.w
        plan = engine.Plan(self.device)
        e = engine.Emitter(plan, self._W)
        ins = _stage_inputs(plan, dict(latent=(B, h, w, 4), t_emb=(B, 320), context=(B, T, 768), hint=(B, h, w, 320)))
        ctx16 = engine.Act(plan.alloc(B * T * 768 * 2), B, T, 1, 768)
        plan.rec(ops.cast_f32_to_bf16, x=ins["context"], out=ctx16.buf, n=B * T * 768, name="context.bf16")
        hint16 = plan.act(B, h, w, 320)
        plan.rec(ops.cast_f32_to_bf16, x=ins["hint"], out=hint16.buf, n=B * h * w * 320, name="hint.bf16")
        ctx_kv = engine.emit_context_kv(e, ctx16, engine.ENCODER_ATTN_LAYERS, plan)
        cols = engine.temb_columns(True)
        total = sum(c for _, c in engine.resblock_names(True))
        table = plan.alloc(B * total * 4)
        engine.emit_time_embedding(e, ins["t_emb"], B, table, encoder_only=True)
        outs, outs32 = [], []
        for i, ch in enumerate(wtab.UNET_SKIP_CH + (1280,)):
            hh, ww = _skip_hw(i, h, w)
            outs.append(plan.act(B, hh, ww, ch))
        engine.emit_controlnet(e, ins["latent"], B, B, h, w, (table, 0, total, cols), ctx_kv, T, hint16, outs)
        for i, a in enumerate(outs):
            o32 = plan.alloc(a.M * a.C * 4)
            plan.rec(ops.cast_bf16_to_f32, x=a.buf, out=o32, n=a.M * a.C, name=f"control.{i}.f32")
            outs32.append((o32, (a.B, a.H, a.W, a.C)))
        plan.finalize()
        bp = _BoundPlan(plan, self._use_graph)
        shapes = dict(latent=(B, h, w, 4), t_emb=(B, 320), context=(B, T, 768), hint=(B, h, w, 320))
        bp.io = {k: ins[k].tensor(torch.float32, s) for k, s in shapes.items()}
        for i, (o32, shp) in enumerate(outs32):
            bp.io[f"out.{i}"] = o32.tensor(torch.float32, shp)
        return bp

    def predict_on_batch(self, x):
        latent, t_emb, context, hint = (_np32(v) for v in x)
        B, T = latent.shape[0], context.shape[1]
        bp = self._bound((B, T), lambda: self._build(B, T))
        for k, v in (("latent", latent), ("t_emb", t_emb), ("context", context), ("hint", hint)):
            bp.io[k].copy_(torch.from_numpy(v))
        bp.run()
        return bp.host([f"out.{i}" for i in range(13)])

    __call__ = predict_on_batch


class HintNet(HipModel):
    """HintNet (reference control_net.py:10-42) on the HIP path."""
    kind = "hintnet"

    def __init__(self, img_height=512, img_width=512, name=None, controlnet_path=None, device=None):
        super().__init__(name or "hint_net", device)
        self.H, self.W = img_height, img_width
        self._maybe_load(controlnet_path)

    def _build(self, B: int) -> _BoundPlan:
        plan = engine.Plan(self.device)
        e = engine.Emitter(plan, self._W)
        img = plan.alloc(B * self.H * self.W * 3 * 4)
        out = plan.act(B, self.H // 8, self.W // 8, 320)
        engine.emit_hintnet(e, img, B, self.H, self.W, out)
        o32 = plan.alloc(out.M * 320 * 4)
        plan.rec(ops.cast_bf16_to_f32, x=out.buf, out=o32, n=out.M * 320, name="hint.f32")
        plan.finalize()
        bp = _BoundPlan(plan, self._use_graph)
        bp.io["image"] = img.tensor(torch.float32, (B, self.H, self.W, 3))
        bp.io["hint"] = o32.tensor(torch.float32, (B, self.H // 8, self.W // 8, 320))
        return bp

    def predict_on_batch(self, x):
        img = _np32(x)
        B = img.shape[0]
        bp = self._bound((B,), lambda: self._build(B))
        bp.io["image"].copy_(torch.from_numpy(img))
        bp.run()
        return bp.host("hint")

    __call__ = predict_on_batch


class ImageEncoder(HipModel):
    """VAE encoder (reference image_encoder.py:21-59) on the HIP path: image in [-1, 1] ->
    mean latent * 0.18215 (no sampling, like the reference's ``split(x, 2)[0] * 0.18215``)."""
    kind = "encoder"

    def __init__(self, ckpt_path=None, name=None, device=None):
        super().__init__(name or "image_encoder", device)
        self._maybe_load(ckpt_path)

    def _pack(self, named):
        W = super()._pack(named)
        # quant_conv (1x1, 8 -> 8) followed by "take the first 4 channels, times 0.18215" is one
        # 8 -> 4 conv with pre-scaled weights (exact: both steps are linear)
        w = np.asarray(named[("quant_conv", "conv_w")], dtype=np.float32)[:, :, :, :4] * np.float32(0.18215)
        b = np.asarray(named[("quant_conv", "bias")], dtype=np.float32)[:4] * np.float32(0.18215)
        W["quant_conv.mean.w"] = packing.dev_f32(w, self.device)
        W["quant_conv.mean.b"] = packing.dev_f32(b, self.device)
        return W

    def _build(self, B, H, Wd) -> _BoundPlan:
        plan = engine.Plan(self.device)
        e = engine.Emitter(plan, self._W)
        img = plan.alloc(B * H * Wd * 3 * 4)
        lat = plan.alloc(B * (H // 8) * (Wd // 8) * 4 * 4)
        engine.emit_encoder(e, img, B, H, Wd, lat)
        plan.finalize()
        bp = _BoundPlan(plan, self._use_graph)
        bp.io["image"] = img.tensor(torch.float32, (B, H, Wd, 3))
        bp.io["latent"] = lat.tensor(torch.float32, (B, H // 8, Wd // 8, 4))
        return bp

    def predict_on_batch(self, x):
        img = _np32(x)
        B, H, Wd, _ = img.shape
        if H % 8 or Wd % 8 or ((H // 8) * (Wd // 8)) % 64:
            # the reference documents multiples of 128 only (stable_diffusion.py:589-593); the mid-block
            # attention GEMMs here need a token count that is a multiple of 64
            raise ValueError("image height / width must be multiples of 8 with (H/8)*(W/8) a multiple of 64")
        bp = self._bound((B, H, Wd), lambda: self._build(B, H, Wd))
        bp.io["image"].copy_(torch.from_numpy(img))
        bp.run()
        return bp.host("latent")

    __call__ = predict_on_batch


class TextClipEmbedding(HipModel):
    """CLIP token + position embedding (reference text_encoder.py:104-121): [tokens, positions] int32
    (B, 77) -> (B, 77, 768)."""
    kind = "text_clip_embedding"

    def __init__(self, max_length=77, embed_dim=768, vocab_size=49408, name=None, ckpt_path=None, device=None):
        if (max_length, embed_dim, vocab_size) != (wtab.CLIP_MAX_LEN, wtab.CLIP_DIM, wtab.CLIP_VOCAB):
            raise ValueError("only the SD1.5 CLIP ViT-L/14 text model geometry is built (77, 768, 49408)")
        super().__init__(name or "text_clip_embedding", device)
        self._maybe_load(ckpt_path)

    def _pack(self, named):
        return {s.name: packing.dev_f32(named[(s.name, s.kind)], self.device) for s in self._specs}

    def emit(self, plan: engine.Plan, tokens, positions, out: engine.Act, status) -> None:
        plan.rec(ops.embedding_sum, tokens=tokens, positions=positions,
                 tok_table=self._W["text_model.embeddings.token_embedding"],
                 pos_table=self._W["text_model.embeddings.position_embedding"], out=out.buf, rows=out.M, dim=wtab.CLIP_DIM,
                 vocab=wtab.CLIP_VOCAB, max_len=wtab.CLIP_MAX_LEN, status=status, name="text_model.embeddings")

    def _build(self, B, T) -> _BoundPlan:
        plan = engine.Plan(self.device)
        tok, pos = plan.alloc(B * T * 4), plan.alloc(B * T * 4)
        x = plan.act(B, T, 1, wtab.CLIP_DIM)
        status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.emit(plan, tok, pos, x, status)
        o32 = plan.alloc(x.M * wtab.CLIP_DIM * 4)
        plan.rec(ops.cast_bf16_to_f32, x=x.buf, out=o32, n=x.M * wtab.CLIP_DIM, name="clip_emb.f32")
        plan.finalize()
        bp = _BoundPlan(plan, self._use_graph)
        bp.io["tokens"], bp.io["positions"] = tok.tensor(torch.int32, (B, T)), pos.tensor(torch.int32, (B, T))
        bp.io["emb"], bp.io["status"] = o32.tensor(torch.float32, (B, T, wtab.CLIP_DIM)), status
        return bp

    def predict_on_batch(self, x):
        tokens, positions = (np.ascontiguousarray(np.asarray(a), dtype=np.int32) for a in x)
        positions = np.array(np.broadcast_to(positions, tokens.shape), dtype=np.int32)
        B, T = tokens.shape
        bp = self._bound((B, T), lambda: self._build(B, T))
        bp.io["tokens"].copy_(torch.from_numpy(tokens))
        bp.io["positions"].copy_(torch.from_numpy(positions))
        bp.run()
        if int(bp.io["status"].item()):
            bp.io["status"].zero_()
            raise ValueError("token / position id outside the embedding table")
        return bp.host("emb")

    __call__ = predict_on_batch


class TextEncoder(HipModel):
    """CLIP text transformer (reference text_encoder.py:123-169): clip_emb (B, 77, 768) -> final
    LayerNorm of the output of layer `clip_skip` (B, 77, 768)."""
    kind = "text_encoder"

    def __init__(self, max_length=77, embed_dim=768, num_heads=12, num_layers=12, clip_skip=-2, name=None, ckpt_path=None,
                 lora_dict=None, device=None):
        if (embed_dim, num_heads, num_layers) != (wtab.CLIP_DIM, wtab.CLIP_HEADS, wtab.CLIP_LAYERS):
            raise ValueError("only the SD1.5 CLIP ViT-L/14 text model geometry is built (768, 12 heads, 12 layers)")
        if not -num_layers <= clip_skip <= -1:
            raise ValueError("clip_skip must be in [-num_layers, -1]")
        self.clip_skip, self.max_length = clip_skip, max_length
        super().__init__(name or "text_encoder", device)
        self._maybe_load(ckpt_path, lora_dict)

    def _table_kw(self):
        return {"clip_skip": self.clip_skip}

    def count_params(self) -> int:
        return int(sum(int(np.prod(s.shape)) for s in self._specs))

    @property
    def n_layers(self) -> int:
        return wtab.CLIP_LAYERS + self.clip_skip + 1

    def _build(self, B, T) -> _BoundPlan:
        plan = engine.Plan(self.device)
        e = engine.Emitter(plan, self._W)
        C = wtab.CLIP_DIM
        x32 = plan.alloc(B * T * C * 4)
        x = plan.act(B, T, 1, C)
        plan.rec(ops.cast_f32_to_bf16, x=x32, out=x.buf, n=B * T * C, name="clip_emb.bf16")
        y = engine.emit_text_encoder(e, x, self.n_layers)
        o32 = plan.alloc(B * T * C * 4)
        plan.rec(ops.cast_bf16_to_f32, x=y.buf, out=o32, n=B * T * C, name="context.f32")
        plan.finalize()
        bp = _BoundPlan(plan, self._use_graph)
        bp.io["emb"], bp.io["context"] = x32.tensor(torch.float32, (B, T, C)), o32.tensor(torch.float32, (B, T, C))
        return bp

    def predict_on_batch(self, x):
        emb = _np32(x)
        B, T, _ = emb.shape
        bp = self._bound((B, T), lambda: self._build(B, T))
        bp.io["emb"].copy_(torch.from_numpy(emb))
        bp.run()
        return bp.host("context")

    __call__ = predict_on_batch

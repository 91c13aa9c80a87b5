"""Checkpoint layout contract, synthetic checkpoints and checkpoint loading.

The reference loads weights *positionally*: ``load_weights_from_file`` walks an ordered table
of ``(checkpoint_key, perm)`` pairs, transposes each tensor with ``perm`` and hands the list to
``model.set_weights`` (reference ``stable_diffusion/ckpt_loader.py:2136-2193``; tables
``ckpt_loader.py:708-2133``).  This module *generates* the same ordered tables from the network
topology (reference ``diffusion_model.py:184-279``, ``image_decoder.py:26-53``,
``control_net.py:14-106``) instead of carrying 2k lines of literals; ``tests/test_weights.py``
checks the generated tables against the reference's (when ``/root/reference`` is present) and
against a committed digest (always).

Each table row is a :class:`WeightSpec`:

* ``key``      checkpoint key as the reference's table spells it (LDM naming for the UNet /
               ControlNet, diffusers naming for the VAE),
* ``alt_key``  the diffusers spelling the reference falls back to (``UNET_KEY_MAPPING``) or None,
* ``name``     logical module path used by this package (diffusers-style, no ``.weight``),
* ``kind``     one of ``conv_w dense_w bias gamma beta``,
* ``perm``     transpose taking the checkpoint (PyTorch) layout to the Keras layout the reference
               model holds: ``(2,3,1,0)`` OIHW->HWIO, ``(1,0)`` (out,in)->(in,out), ``None``,
* ``shape``    the Keras-layout shape (what ``model.weights[i].shape`` is in the reference).
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

CONV_PERM = (2, 3, 1, 0)
DENSE_PERM = (1, 0)


@dataclass(frozen=True)
class WeightSpec:
    key: str
    alt_key: Optional[str]
    name: str
    kind: str
    perm: Optional[Tuple[int, ...]]
    shape: Tuple[int, ...]

    @property
    def torch_shape(self) -> Tuple[int, ...]:
        """Shape of the tensor as stored in the checkpoint (PyTorch layout)."""
        if self.perm is None:
            return self.shape
        inv = np.argsort(self.perm)
        return tuple(self.shape[i] for i in inv)


class _Table:
    def __init__(self, prefix: str, use_alt: bool):
        self.rows: List[WeightSpec] = []
        self.prefix = prefix
        self.use_alt = use_alt

    def _add(self, key, name, suffix, kind, perm, shape):
        alt = (name + "." + suffix) if self.use_alt else None
        self.rows.append(WeightSpec(self.prefix + key + "." + suffix, alt, name, kind, perm, tuple(shape)))

    def conv(self, key, name, cin, cout, k):
        self._add(key, name, "weight", "conv_w", CONV_PERM, (k, k, cin, cout))
        self._add(key, name, "bias", "bias", None, (cout,))

    def dense(self, key, name, cin, cout, bias=True):
        self._add(key, name, "weight", "dense_w", DENSE_PERM, (cin, cout))
        if bias:
            self._add(key, name, "bias", "bias", None, (cout,))

    def norm(self, key, name, c):
        self._add(key, name, "weight", "gamma", None, (c,))
        self._add(key, name, "bias", "beta", None, (c,))


# --------------------------------------------------------------------------------------
# UNet (reference diffusion_model.py:184-279; table order ckpt_loader.py:709-1394)
# --------------------------------------------------------------------------------------
UNET_CH = (320, 640, 1280, 1280)
UNET_HEADS = 8
CTX_DIM = 768
TEMB_DIM = 1280


def _ldm_resblock(t: _Table, key, name, cin, cout):
    t.norm(key + ".in_layers.0", name + ".norm1", cin)
    t.conv(key + ".in_layers.2", name + ".conv1", cin, cout, 3)
    t.dense(key + ".emb_layers.1", name + ".time_emb_proj", TEMB_DIM, cout)
    t.norm(key + ".out_layers.0", name + ".norm2", cout)
    t.conv(key + ".out_layers.3", name + ".conv2", cout, cout, 3)
    if cin != cout:
        t.conv(key + ".skip_connection", name + ".conv_shortcut", cin, cout, 1)


def _ldm_attention(t: _Table, key, name, c):
    t.norm(key + ".norm", name + ".norm", c)
    t.conv(key + ".proj_in", name + ".proj_in", c, c, 1)
    tb_k, tb_n = key + ".transformer_blocks.0", name + ".transformer_blocks.0"
    t.norm(tb_k + ".norm1", tb_n + ".norm1", c)
    for p in ("to_q", "to_k", "to_v"):
        t.dense(tb_k + ".attn1." + p, tb_n + ".attn1." + p, c, c, bias=False)
    t.dense(tb_k + ".attn1.to_out.0", tb_n + ".attn1.to_out.0", c, c)
    t.norm(tb_k + ".norm2", tb_n + ".norm2", c)
    t.dense(tb_k + ".attn2.to_q", tb_n + ".attn2.to_q", c, c, bias=False)
    t.dense(tb_k + ".attn2.to_k", tb_n + ".attn2.to_k", CTX_DIM, c, bias=False)
    t.dense(tb_k + ".attn2.to_v", tb_n + ".attn2.to_v", CTX_DIM, c, bias=False)
    t.dense(tb_k + ".attn2.to_out.0", tb_n + ".attn2.to_out.0", c, c)
    t.norm(tb_k + ".norm3", tb_n + ".norm3", c)
    t.dense(tb_k + ".ff.net.0.proj", tb_n + ".ff.net.0.proj", c, 8 * c)
    t.dense(tb_k + ".ff.net.2", tb_n + ".ff.net.2", 4 * c, c)
    t.conv(key + ".proj_out", name + ".proj_out", c, c, 1)


def _unet_encoder(t: _Table, control: bool):
    """Shared down + mid path of the UNet and the ControlNet (control_net.py:47-90)."""
    t.dense("time_embed.0", "time_embedding.linear_1", 320, TEMB_DIM)
    t.conv("input_blocks.0.0", "conv_in", 4, 320, 3)
    t.dense("time_embed.2", "time_embedding.linear_2", TEMB_DIM, TEMB_DIM)
    ib = 1
    cin = 320
    for lvl, ch in enumerate(UNET_CH):
        for r in range(2):
            _ldm_resblock(t, f"input_blocks.{ib}.0", f"down_blocks.{lvl}.resnets.{r}", cin, ch)
            cin = ch
            if lvl < 3:
                _ldm_attention(t, f"input_blocks.{ib}.1", f"down_blocks.{lvl}.attentions.{r}", ch)
            ib += 1
        if lvl < 3:
            t.conv(f"input_blocks.{ib}.0.op", f"down_blocks.{lvl}.downsamplers.0.conv", ch, ch, 3)
            ib += 1
    _ldm_resblock(t, "middle_block.0", "mid_block.resnets.0", 1280, 1280)
    _ldm_attention(t, "middle_block.1", "mid_block.attentions.0", 1280)
    _ldm_resblock(t, "middle_block.2", "mid_block.resnets.1", 1280, 1280)


# skip-stack channel counts popped by the up path
UNET_SKIP_CH = (320, 320, 320, 320, 640, 640, 640, 1280, 1280, 1280, 1280, 1280)


def unet_table() -> List[WeightSpec]:
    t = _Table("model.diffusion_model.", use_alt=True)
    _unet_encoder(t, control=False)
    skips = list(UNET_SKIP_CH)
    cin = 1280
    ob = 0
    for ui, lvl in enumerate((3, 2, 1, 0)):
        ch = UNET_CH[lvl]
        for r in range(3):
            sk = skips.pop()
            _ldm_resblock(t, f"output_blocks.{ob}.0", f"up_blocks.{ui}.resnets.{r}", cin + sk, ch)
            cin = ch
            sub = 1
            if lvl < 3:
                _ldm_attention(t, f"output_blocks.{ob}.1", f"up_blocks.{ui}.attentions.{r}", ch)
                sub = 2
            if r == 2 and lvl > 0:
                t.conv(f"output_blocks.{ob}.{sub}.conv", f"up_blocks.{ui}.upsamplers.0.conv", ch, ch, 3)
            ob += 1
    t.norm("out.0", "conv_norm_out", 320)
    t.conv("out.2", "conv_out", 320, 4, 3)
    return t.rows


def controlnet_table() -> List[WeightSpec]:
    t = _Table("control_model.", use_alt=False)
    _unet_encoder(t, control=True)
    for i, ch in enumerate(UNET_SKIP_CH):
        t.conv(f"zero_convs.{i}.0", f"zero_convs.{i}", ch, ch, 1)
    t.conv("middle_block_out.0", "zero_convs.12", 1280, 1280, 1)
    return t.rows


HINT_CH = ((3, 16, 1), (16, 16, 1), (16, 32, 2), (32, 32, 1), (32, 96, 2), (96, 96, 1), (96, 256, 2), (256, 320, 1))


def hintnet_table() -> List[WeightSpec]:
    t = _Table("control_model.", use_alt=False)
    for i, (cin, cout, _s) in enumerate(HINT_CH):
        t.conv(f"input_hint_block.{2 * i}", f"input_hint_block.{i}", cin, cout, 3)
    return t.rows


# --------------------------------------------------------------------------------------
# VAE decoder (reference image_decoder.py:26-53, layers.py:28-80; table ckpt_loader.py:1505-1646)
# --------------------------------------------------------------------------------------
def _vae_resnet(t: _Table, key, cin, cout):
    t.norm(key + ".norm1", key + ".norm1", cin)
    t.conv(key + ".conv1", key + ".conv1", cin, cout, 3)
    t.norm(key + ".norm2", key + ".norm2", cout)
    t.conv(key + ".conv2", key + ".conv2", cout, cout, 3)
    if cin != cout:
        t.conv(key + ".conv_shortcut", key + ".conv_shortcut", cin, cout, 1)


def _vae_attention(t: _Table, key, c):
    t.norm(key + ".group_norm", key + ".group_norm", c)
    for p in ("query", "key", "value", "proj_attn"):
        t.dense(key + "." + p, key + "." + p, c, c)


VAE_DEC_BLOCKS = ((512, 512, True), (512, 512, True), (512, 256, True), (256, 128, False))


def decoder_table() -> List[WeightSpec]:
    t = _Table("", use_alt=False)
    t.conv("post_quant_conv", "post_quant_conv", 4, 4, 1)
    t.conv("decoder.conv_in", "decoder.conv_in", 4, 512, 3)
    _vae_resnet(t, "decoder.mid_block.resnets.0", 512, 512)
    _vae_attention(t, "decoder.mid_block.attentions.0", 512)
    _vae_resnet(t, "decoder.mid_block.resnets.1", 512, 512)
    for bi, (cin, cout, up) in enumerate(VAE_DEC_BLOCKS):
        for r in range(3):
            _vae_resnet(t, f"decoder.up_blocks.{bi}.resnets.{r}", cin if r == 0 else cout, cout)
        if up:
            t.conv(f"decoder.up_blocks.{bi}.upsamplers.0.conv", f"decoder.up_blocks.{bi}.upsamplers.0.conv",
                   cout, cout, 3)
    t.norm("decoder.conv_norm_out", "decoder.conv_norm_out", 128)
    t.conv("decoder.conv_out", "decoder.conv_out", 128, 3, 3)
    return t.rows


VAE_ENC_BLOCKS = ((128, 128, True), (128, 256, True), (256, 512, True), (512, 512, False))


def encoder_table() -> List[WeightSpec]:
    """VAE encoder (reference image_encoder.py:21-48; table ckpt_loader.py 'encoder', 108 tensors)."""
    t = _Table("", use_alt=False)
    t.conv("encoder.conv_in", "encoder.conv_in", 3, 128, 3)
    for bi, (cin, cout, down) in enumerate(VAE_ENC_BLOCKS):
        for r in range(2):
            _vae_resnet(t, f"encoder.down_blocks.{bi}.resnets.{r}", cin if r == 0 else cout, cout)
        if down:
            t.conv(f"encoder.down_blocks.{bi}.downsamplers.0.conv", f"encoder.down_blocks.{bi}.downsamplers.0.conv", cout, cout, 3)
    _vae_resnet(t, "encoder.mid_block.resnets.0", 512, 512)
    _vae_attention(t, "encoder.mid_block.attentions.0", 512)
    _vae_resnet(t, "encoder.mid_block.resnets.1", 512, 512)
    t.norm("encoder.conv_norm_out", "encoder.conv_norm_out", 512)
    t.conv("encoder.conv_out", "encoder.conv_out", 512, 8, 3)
    t.conv("quant_conv", "quant_conv", 8, 8, 1)
    return t.rows


TABLES = {
    "civitai_model": unet_table,
    "encoder": encoder_table,
    "decoder": decoder_table,
    "controlnet": controlnet_table,
    "hintnet": hintnet_table,
}


# --------------------------------------------------------------------------------------
# CLIP text encoder (reference text_encoder.py:104-169).  The reference builds
# these mappings in the constructors, not in CKPT_MAPPING: 16 tensors per encoder layer for layers
# 0 .. num_layers + clip_skip (the layers after out[clip_skip] are never loaded), then the final
# LayerNorm; the embedding model holds the two lookup tables.
# --------------------------------------------------------------------------------------
CLIP_DIM, CLIP_HEADS, CLIP_LAYERS, CLIP_VOCAB, CLIP_MAX_LEN = 768, 12, 12, 49408, 77


def text_encoder_table(clip_skip: int = -1) -> List[WeightSpec]:
    t = _Table("", use_alt=False)
    for idx in range(0, CLIP_LAYERS + clip_skip + 1):
        ln = f"text_model.encoder.layers.{idx}"
        t.norm(ln + ".layer_norm1", ln + ".layer_norm1", CLIP_DIM)
        for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
            t.dense(f"{ln}.self_attn.{proj}", f"{ln}.self_attn.{proj}", CLIP_DIM, CLIP_DIM)
        t.norm(ln + ".layer_norm2", ln + ".layer_norm2", CLIP_DIM)
        t.dense(ln + ".mlp.fc1", ln + ".mlp.fc1", CLIP_DIM, 4 * CLIP_DIM)
        t.dense(ln + ".mlp.fc2", ln + ".mlp.fc2", 4 * CLIP_DIM, CLIP_DIM)
    t.norm("text_model.final_layer_norm", "text_model.final_layer_norm", CLIP_DIM)
    return t.rows


def text_clip_embedding_table() -> List[WeightSpec]:
    return [WeightSpec("text_model.embeddings.token_embedding.weight", None, "text_model.embeddings.token_embedding", "embedding",
                       None, (CLIP_VOCAB, CLIP_DIM)),
            WeightSpec("text_model.embeddings.position_embedding.weight", None, "text_model.embeddings.position_embedding",
                       "embedding", None, (CLIP_MAX_LEN, CLIP_DIM))]


TABLES["text_encoder"] = text_encoder_table
TABLES["text_clip_embedding"] = text_clip_embedding_table


def table(kind: str, **kw) -> List[WeightSpec]:
    return TABLES[kind](**kw)


def table_digest(kind: str) -> str:
    """SHA-256 over the ordered (key, perm) list — the positional contract (golden G6)."""
    h = hashlib.sha256()
    for s in table(kind):
        h.update(repr((s.key, s.perm)).encode())
    return h.hexdigest()


def param_count(kind: str) -> int:
    return int(sum(int(np.prod(s.shape)) for s in table(kind)))


# --------------------------------------------------------------------------------------
# Synthetic checkpoints: seeded Glorot-uniform, zeros bias, ones/zeros norm,
# emitted in *checkpoint* (PyTorch) layout under the reference's key names so the same file can
# be given to real minSDTF through unet_ckpt= / vae_ckpt= / controlnet_path=.
# --------------------------------------------------------------------------------------
def _glorot_limit(spec: WeightSpec) -> float:
    if spec.kind == "conv_w":
        kh, kw, cin, cout = spec.shape
        fan_in, fan_out = kh * kw * cin, kh * kw * cout
    else:
        fan_in, fan_out = spec.shape
    return float(np.sqrt(6.0 / (fan_in + fan_out)))


def synth_tensors(kind: str, seed: int = 0, bias_scale: float = 0.0, **table_kw) -> Iterator[Tuple[WeightSpec, np.ndarray]]:
    """Yield (spec, tensor in checkpoint/PyTorch layout) in table order.

    One PCG64 stream per table, consumed in table order, so the values do not depend on how the
    caller batches the work.  ``bias_scale`` > 0 draws biases / norm offsets from U(-s, s) (used by
    parity tests so that bias / beta paths are not vacuous); 0 gives the Keras default init.
    """
    salt = {"civitai_model": 0, "decoder": 1, "controlnet": 2, "hintnet": 3, "encoder": 4, "text_encoder": 5,
            "text_clip_embedding": 6}[kind]
    rng = np.random.Generator(np.random.PCG64([seed, salt]))
    for spec in table(kind, **table_kw):
        if spec.kind == "embedding":   # Keras Embedding default: uniform(-0.05, 0.05)
            yield spec, rng.uniform(-0.05, 0.05, size=spec.shape).astype(np.float32)
            continue
        if spec.kind in ("conv_w", "dense_w"):
            lim = np.float32(_glorot_limit(spec))
            w = rng.random(size=spec.torch_shape, dtype=np.float32)
            w *= np.float32(2.0) * lim
            w -= lim
        elif spec.kind == "gamma":
            w = np.ones(spec.shape, np.float32)
            if bias_scale > 0:
                w += rng.uniform(-bias_scale, bias_scale, size=spec.shape).astype(np.float32)
        else:
            w = np.zeros(spec.shape, np.float32)
            if bias_scale > 0:
                w = rng.uniform(-bias_scale, bias_scale, size=spec.shape).astype(np.float32)
        yield spec, w


def to_keras_layout(spec: WeightSpec, w: np.ndarray) -> np.ndarray:
    """Apply the table's perm (ckpt_loader.py:2181-2182); returns a contiguous array."""
    if spec.perm is None:
        return w
    import torch  # torch's permute+contiguous is multi-threaded; numpy's is not

    return torch.from_numpy(np.ascontiguousarray(w)).permute(*spec.perm).contiguous().numpy()


def synth_keras_weights(kind: str, seed: int = 0, bias_scale: float = 0.0, **table_kw) -> List[np.ndarray]:
    """Ordered list in Keras layout — what ``set_weights`` receives in the reference."""
    return [to_keras_layout(s, w) for s, w in synth_tensors(kind, seed, bias_scale, **table_kw)]


def write_synthetic_checkpoint(path: str, kinds=("civitai_model",), seed: int = 0, bias_scale: float = 0.0) -> None:
    """Write a .safetensors file under the reference's checkpoint keys (PyTorch layout)."""
    import torch
    from safetensors.torch import save_file

    sd = {}
    for kind in kinds:
        for spec, w in synth_tensors(kind, seed, bias_scale):
            sd[spec.key] = torch.from_numpy(np.ascontiguousarray(w))
    save_file(sd, path)


# --------------------------------------------------------------------------------------
# Loading real / synthetic checkpoint files (mirrors ckpt_loader.load_weights_from_file)
# --------------------------------------------------------------------------------------
def read_state_dict(ckpt_path: str) -> Dict[str, np.ndarray]:
    if ckpt_path.endswith(".safetensors"):
        from safetensors import safe_open

        out = {}
        with safe_open(ckpt_path, framework="pt", device="cpu") as f:
            for k in f.keys():
                out[k] = f.get_tensor(k)
        return out
    import torch

    sd = torch.load(ckpt_path, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    return sd


def load_weights_from_file(model, ckpt_path: str, kind: str, lora_dict: Optional[dict] = None,
                           specs: Optional[List[WeightSpec]] = None) -> None:
    """Positional load with the reference's semantics (ckpt_loader.py:2136-2193).

    ``model`` exposes ``name``, ``weights`` (ordered objects with ``.shape``/``.name`` in Keras
    layout) and ``set_weights(list)`` — the same surface the reference loader uses.  ``specs``
    overrides the table of ``kind`` (the reference passes the table as an argument too).
    """
    import os

    print("{} loading:[{}]".format(model.name, os.path.basename(ckpt_path)))
    sd = read_state_dict(ckpt_path)
    if specs is None:
        specs = table(kind)
    out = []
    lora_keys = list(lora_dict.keys()) if lora_dict is not None else []
    lora_count, lora_idx = len(lora_keys), 0
    for i, spec in enumerate(specs):
        if spec.key in sd:
            w = sd[spec.key]
        elif spec.alt_key is not None and spec.alt_key in sd:
            w = sd[spec.alt_key]
        else:
            raise KeyError(spec.key)
        if hasattr(w, "detach"):
            w = w.detach().float().numpy()
        if lora_dict is not None:
            lw = lora_dict.get(spec.alt_key if spec.alt_key is not None else spec.key, None)
            if lw is not None:
                w = w + lw
                lora_idx += 1
        w = to_keras_layout(spec, np.asarray(w, dtype=np.float32))
        if tuple(model.weights[i].shape) != tuple(w.shape):
            print("Wrong :[{},{}]".format(model.weights[i].name, spec.key))
        out.append(w)
    if lora_count > 0:
        print("Apply {}/{} lora weights".format(lora_idx, lora_count))
    model.set_weights(out)
    print("Loaded %d weights for %s" % (len(out), model.name))


# --------------------------------------------------------------------------------------
# LoRA files (kohya naming) -> additive weight deltas (mirrors ckpt_loader.load_weights_from_lora)
# --------------------------------------------------------------------------------------
# The reference restores module names from the flattened kohya keys with a fixed list of string
# substitutions (ckpt_loader.py:2236-2273); only these layer types come out with a usable name, the
# others (conv_in, conv_out, time_embedding, norms) are never matched by the loader.  Here the
# inverse map is built from the weight table instead: kohya name = "lora_unet_" + module path with
# "." -> "_", restricted to the same layer types so that exactly the same deltas get applied.
_LORA_UNET_SUFFIXES = ("proj_in", "proj_out", "attn1.to_q", "attn1.to_k", "attn1.to_v", "attn1.to_out.0", "attn2.to_q",
                       "attn2.to_k", "attn2.to_v", "attn2.to_out.0", "ff.net.0.proj", "ff.net.2", "time_emb_proj",
                       "conv_shortcut", "downsamplers.0.conv", "upsamplers.0.conv", "conv1", "conv2")
_LORA_TE_SUFFIXES = ("mlp.fc1", "mlp.fc2", "self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj")
_lora_unet_names: Optional[Dict[str, str]] = None


def _lora_unet_name_map() -> Dict[str, str]:
    global _lora_unet_names
    if _lora_unet_names is None:
        m = {}
        for spec in table("civitai_model"):
            if spec.alt_key is None or not spec.alt_key.endswith(".weight"):
                continue
            path = spec.alt_key[:-len(".weight")]
            if path.endswith(_LORA_UNET_SUFFIXES):
                m["lora_unet_" + path.replace(".", "_")] = spec.alt_key
        _lora_unet_names = m
    return _lora_unet_names


def _lora_te_name(name: str) -> Optional[str]:
    """lora_te_text_model_encoder_layers_<i>_<module> -> text_model.encoder.layers.<i>.<module>.weight"""
    head = "lora_te_text_model_encoder_layers_"
    if not name.startswith(head):
        return None
    idx, _, rest = name[len(head):].partition("_")
    for suf in _LORA_TE_SUFFIXES:
        if rest == suf.replace(".", "_"):
            return f"text_model.encoder.layers.{idx}.{suf}.weight"
    return None


def lora_delta(lora_up, lora_down, alpha) -> np.ndarray:
    """(alpha / rank) * up . down in the checkpoint (PyTorch) layout: Linear (out,in), 1x1 conv
    (out,in,1,1) and 3x3 conv (out,in,3,3, `down` carries the taps) — ckpt_loader.py:2216-2230."""
    import torch

    up = torch.as_tensor(np.asarray(lora_up.detach().float().cpu() if hasattr(lora_up, "detach") else lora_up, dtype=np.float32))
    down = torch.as_tensor(np.asarray(lora_down.detach().float().cpu() if hasattr(lora_down, "detach") else lora_down, dtype=np.float32))
    a = np.asarray(alpha.detach().float().cpu() if hasattr(alpha, "detach") else alpha, dtype=np.float32)
    scale = a / float(up.shape[1])
    if down.dim() == 2:
        w = up @ down
    elif tuple(down.shape[2:4]) == (1, 1):
        w = (up[:, :, 0, 0] @ down[:, :, 0, 0])[:, :, None, None]
    else:
        # out[o, i] = sum_r up[o, r] (*) down[r, i]: a conv of the per-input-channel taps with `up` as the filter
        w = torch.nn.functional.conv2d(down.permute(1, 0, 2, 3), up).permute(1, 0, 2, 3)
    return w.numpy() * scale


def load_weights_from_lora(ckpt_path: str) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    """LoRA file -> (text_encoder_deltas, unet_deltas), keyed by the diffusers weight names that
    ``load_weights_from_file(..., lora_dict=)`` looks up (reference ckpt_loader.py:2196-2276).
    The text-encoder half is returned for API parity; the CLIP front-end is outside this path."""
    print("loading:[{}]".format(ckpt_path))
    sd = read_state_dict(ckpt_path)
    text_encoder, unet = {}, {}
    names = _lora_unet_name_map()
    for key in list(sd.keys()):
        key = str(key)
        if not key.endswith(".alpha"):
            continue
        name = key[:-len(".alpha")]
        w = lora_delta(sd[name + ".lora_up.weight"], sd[name + ".lora_down.weight"], sd[key])
        if name.startswith("lora_te_text_model"):
            restored = _lora_te_name(name)
            if restored is not None:
                text_encoder[restored] = w
        elif name.startswith("lora_unet_"):
            restored = names.get(name)
            if restored is not None:
                unet[restored] = w
    print("lora dict:[{}] done.".format(ckpt_path))
    return text_encoder, unet

"""SURVEY.md §8f rank 2 — checkpoint and LoRA loading with the reference's semantics
(ckpt_loader.py:2136-2276): positional load in table order, LDM key with diffusers-key fallback,
LoRA deltas added in the checkpoint layout before the Keras transpose, kohya LoRA names restored to
diffusers weight names.  Expected values were produced by running the reference's own
load_weights_from_file / load_weights_from_lora on the same generated files (tools/make_goldens.py,
tests/golden/g8_loaders.json); when /root/reference is present the two are also compared live."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = "/root/reference/stable_diffusion/ckpt_loader.py"


@pytest.fixture(scope="module")
def g8():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "g8_loaders.json")))


@pytest.fixture(scope="module")
def lora_file(tmp_path_factory):
    from make_goldens import lora_fixture
    from safetensors.torch import save_file

    p = str(tmp_path_factory.mktemp("lora") / "lora.safetensors")
    save_file(lora_fixture(), p)
    return p


def test_lora_file_to_deltas_matches_reference(g8, lora_file):
    from make_goldens import digest_arrays

    from minsdtf_amd import weights as W

    te, un = W.load_weights_from_lora(lora_file)
    assert sorted(un.keys()) == g8["lora_unet_keys"]          # every layer type the reference restores, nothing else
    assert sorted(te.keys()) == g8["lora_te_keys"]
    assert not (set(un) & set(g8["lora_unet_unmatched"]))
    assert digest_arrays([un[k] for k in sorted(un)]) == g8["lora_unet_digest"]
    assert digest_arrays([te[k] for k in sorted(te)]) == g8["lora_te_digest"]
    # shapes follow the checkpoint (PyTorch) layout of each layer class
    assert un["down_blocks.0.resnets.0.conv1.weight"].shape == (4, 3, 3, 3)
    assert un["down_blocks.0.attentions.0.proj_in.weight"].shape == (4, 3, 1, 1)
    assert un["down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight"].shape == (4, 3)


def test_lora_delta_math():
    """(alpha / rank) * up . down for the three layer classes, against plain einsum."""
    from minsdtf_amd.weights import lora_delta

    rng = np.random.default_rng(3)
    up, down = rng.standard_normal((6, 4)).astype(np.float32), rng.standard_normal((4, 5)).astype(np.float32)
    np.testing.assert_allclose(lora_delta(up, down, 2.0), 0.5 * up @ down, rtol=1e-5, atol=1e-6)
    up4, down4 = up[:, :, None, None], rng.standard_normal((4, 5, 3, 3)).astype(np.float32)
    np.testing.assert_allclose(lora_delta(up4, down4, torch.tensor(8.0)), 2.0 * np.einsum("or,rikl->oikl", up, down4), rtol=1e-5,
                               atol=1e-5)
    np.testing.assert_allclose(lora_delta(up4, down[:, :, None, None], 4.0), (up @ down)[:, :, None, None], rtol=1e-5, atol=1e-6)


def _synthetic_case(kind, nrows, use_alt, with_lora, lora_names, tmpdir):
    """The same generated checkpoint / LoRA dict as tools/make_goldens.make_loader_goldens."""
    from safetensors.torch import save_file

    from minsdtf_amd import weights as W

    specs = W.table(kind)[:nrows]
    rng = np.random.default_rng(5)
    sd, lora = {}, {}
    for i, s in enumerate(specs):
        w = rng.standard_normal(s.torch_shape).astype(np.float32)
        key = s.alt_key if (use_alt and s.alt_key is not None and i % 2) else s.key
        sd[key] = torch.from_numpy(w)
        if with_lora and s.alt_key is not None and s.alt_key in lora_names and i % 3 == 0:
            lora[s.alt_key] = rng.standard_normal(s.torch_shape).astype(np.float32) * 0.1
    path = os.path.join(str(tmpdir), kind + ".safetensors")
    save_file(sd, path)
    return specs, path, lora


CASES = [("hintnet", None, False, False), ("civitai_model", 120, True, True), ("encoder", 30, False, False)]


@pytest.mark.parametrize("kind,nrows,use_alt,with_lora", CASES)
def test_positional_load_matches_reference(g8, tmp_path, kind, nrows, use_alt, with_lora):
    from make_goldens import FakeKerasModel, digest_arrays

    from minsdtf_amd import weights as W

    specs, path, lora = _synthetic_case(kind, nrows, use_alt, with_lora, set(g8["lora_unet_keys"]), tmp_path)
    fake = FakeKerasModel(kind, specs)
    W.load_weights_from_file(fake, path, kind, lora_dict=dict(lora) if with_lora else None, specs=specs)
    assert len(fake.loaded) == g8[f"load_{kind}_count"] and len(lora) == g8[f"load_{kind}_lora_applied"]
    for w, s in zip(fake.loaded, specs):
        assert tuple(w.shape) == tuple(s.shape)                  # Keras layout, table order
    assert digest_arrays(fake.loaded) == g8[f"load_{kind}_digest"]


@pytest.mark.skipif(not os.path.exists(REF), reason="reference not present")
@pytest.mark.parametrize("kind,nrows,use_alt,with_lora", CASES)
def test_positional_load_equals_reference_live(g8, tmp_path, kind, nrows, use_alt, with_lora):
    import importlib.util

    from make_goldens import FakeKerasModel

    from minsdtf_amd import weights as W

    spec = importlib.util.spec_from_file_location("ref_ckpt_live", REF)
    ck = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ck)
    specs, path, lora = _synthetic_case(kind, nrows, use_alt, with_lora, set(g8["lora_unet_keys"]), tmp_path)
    mine, theirs = FakeKerasModel(kind, specs), FakeKerasModel(kind, specs)
    W.load_weights_from_file(mine, path, kind, lora_dict=dict(lora) if with_lora else None, specs=specs)
    mapping = ck.CKPT_MAPPING[kind][:nrows] if nrows else ck.CKPT_MAPPING[kind]
    ck.load_weights_from_file(theirs, path, mapping, key_mapping=ck.UNET_KEY_MAPPING if use_alt else None,
                              lora_dict=dict(lora) if with_lora else None)
    for a, b in zip(mine.loaded, theirs.loaded):
        np.testing.assert_array_equal(np.asarray(a, np.float32), np.asarray(b, np.float32))


def test_pipeline_reads_lora_path(lora_file):
    """StableDiffusion(lora_path=...) builds the UNet delta dict at construction (reference :641-643);
    a path that does not exist is ignored, like the reference."""
    from minsdtf_amd.stable_diffusion import StableDiffusion

    sd = StableDiffusion(64, 64, lora_path=lora_file, device="cpu")
    assert sd.lora_path == lora_file and len(sd.unet_lora_dict) == 278 and len(sd.text_encoder_lora_dict) == 12
    sd2 = StableDiffusion(64, 64, lora_path="/nonexistent/lora.safetensors", device="cpu")
    assert sd2.lora_path is None and sd2.unet_lora_dict is None


@pytest.mark.parametrize("clip_skip", [-1, -2, -12])
def test_text_encoder_tables_match_reference(clip_skip):
    """The (key, perm) lists text_encoder.py:133-157 builds in its constructor (captured by
    tools/make_goldens.py under the keras stub) == the generated table, for several clip_skip values."""
    import hashlib

    from minsdtf_amd import weights as W

    g9 = json.load(open(os.path.join(ROOT, "tests", "golden", "g9_text_tables.json")))
    specs = W.table("text_encoder", clip_skip=clip_skip)
    h = hashlib.sha256()
    for s in specs:
        h.update(repr((s.key, s.perm)).encode())
    want = g9[f"text_encoder_clip_skip_{clip_skip}"]
    assert len(specs) == want["count"] and h.hexdigest() == want["sha256"]
    assert [[s.key, s.perm] for s in W.table("text_clip_embedding")] == g9["text_clip_embedding"]
    # 12 layers: 85,056,000 + embeddings 38,004,480 = the 123 M parameters of CLIP ViT-L/14's text model
    if clip_skip == -1:
        assert sum(int(np.prod(s.shape)) for s in specs) == 85056000
        assert sum(int(np.prod(s.shape)) for s in W.table("text_clip_embedding")) == 49408 * 768 + 77 * 768


def test_text_encoder_lora_keys_are_table_keys(g8):
    """The text-encoder half of a LoRA file is keyed by names of the text-encoder table, so
    TextEncoder(lora_dict=...) merges it through the same positional loader."""
    from minsdtf_amd import weights as W

    keys = {s.key for s in W.table("text_encoder", clip_skip=-1)}
    assert set(g8["lora_te_keys"]) <= keys

"""CPU tests (no GPU): the oracle and the host-side mirror of the reference interface against the
golden vectors captured from the reference itself (tools/make_goldens.py -> tests/golden/g*.{npz,json}),
the checkpoint-layout contract, the launch-plan allocator, and the C-ABI library's exports."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
ROOT = os.path.dirname(HERE)


def gold(name):
    p = os.path.join(GOLD, name)
    return np.load(p) if name.endswith(".npz") else json.load(open(p))


# ------------------------------------------------------------------ G1: scheduler
@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_scheduler_matches_reference(impl):
    g = gold("g1_scheduler.npz")
    if impl == "oracle":
        from oracle.sd_oracle import OracleScheduler as Sch
    else:
        from minsdtf_amd.scheduler import Scheduler as Sch
    s = Sch()
    np.testing.assert_array_equal(s.alphas_cumprod[g["alphas_cumprod_idx"]], g["alphas_cumprod"])
    np.testing.assert_array_equal(s.signal_rates, g["signal_rates_full"])
    np.testing.assert_array_equal(s.noise_rates, g["noise_rates_full"])
    for n in (1, 4, 25, 50):
        s.set_timesteps(n)
        np.testing.assert_array_equal(np.asarray(s.timesteps), g[f"timesteps_{n}"])
        assert np.asarray(s.timesteps).dtype == np.int32
    for n in (4, 25):
        s = Sch()
        s.set_timesteps(n)
        x = g[f"run{n}_latent0"]
        for i, t in enumerate(s.timesteps):
            x = s.step(g[f"run{n}_eps"][i], int(t), x)
            assert np.asarray(x).dtype == np.float64  # f64 coefficients promote, like the reference
            np.testing.assert_array_equal(np.asarray(x), g[f"run{n}_out"][i])


def test_coefficient_table_reproduces_step():
    """The fp32 table the cfg_step kernel indexes == the scheduler's own float64 coefficients."""
    from minsdtf_amd.scheduler import Scheduler

    s = Scheduler()
    s.set_timesteps(25)
    tab = s.coefficient_table()
    assert tab.shape == (25, 4) and tab.dtype == np.float32
    ts = s.timesteps
    for i, t in enumerate(ts):
        nxt = [s.signal_rates[ts[i + 1]], s.noise_rates[ts[i + 1]]] if i + 1 < 25 else [1.0, 0.0]   # last step: x' = x0
        np.testing.assert_allclose(tab[i], [s.signal_rates[t], s.noise_rates[t]] + nxt, rtol=1e-7)
    assert not s.noise_coefficients().any()


@pytest.mark.parametrize("impl", ["product", "oracle"])
def test_tcd_scheduler_matches_reference(impl):
    """TCD schedule and stochastic step (scheduler.py:136-237,286-307) against the reference's outputs under the
    same seed of numpy's global generator; and the {A, B, C} coefficient form the sampler kernel evaluates."""
    if impl == "oracle":
        from oracle.sd_oracle import OracleScheduler as Scheduler
    else:
        from minsdtf_amd.scheduler import Scheduler

    g = gold("g1_scheduler.npz")
    for n in (1, 4, 25, 50):
        s = Scheduler(active_tcd=True)
        s.set_timesteps(n)
        np.testing.assert_array_equal(np.asarray(s.timesteps), g[f"tcd_timesteps_{n}"])
    for n in (4, 8):
        s = Scheduler(active_tcd=True)
        s.set_timesteps(n)
        x = g[f"tcd_run{n}_latent0"]
        np.random.seed(1000 + n)
        for i, t in enumerate(s.timesteps):
            x = s.step(g[f"tcd_run{n}_eps"][i], int(t), x)
            np.testing.assert_array_equal(np.asarray(x), g[f"tcd_run{n}_out"][i])
        if impl == "oracle":
            continue
        tab, cz = s.coefficient_table().astype(np.float64), s.noise_coefficients().astype(np.float64)
        assert cz[-1] == 0.0 and (cz[:-1] > 0).all()
        x = g[f"tcd_run{n}_latent0"].astype(np.float64)
        np.random.seed(1000 + n)
        for i in range(n):
            e = g[f"tcd_run{n}_eps"][i]
            x = tab[i, 2] * ((x - tab[i, 1] * e) / tab[i, 0]) + tab[i, 3] * e
            if i < n - 1:
                x = x + cz[i] * np.random.randn(*e.shape).astype(np.float32)
            np.testing.assert_allclose(x, g[f"tcd_run{n}_out"][i], rtol=0, atol=2e-5)


# ------------------------------------------------------------------ G3 / G4 / G7: host helpers
def test_timestep_embedding_matches_reference():
    g = gold("g3_timestep_embedding.npz")
    from minsdtf_amd.stable_diffusion import get_timestep_embedding
    from oracle.sd_oracle import timestep_embedding

    for fn in (get_timestep_embedding, timestep_embedding):
        tab = np.stack([fn(int(t), 1)[0] for t in g["timesteps"]])
        assert str(tab.dtype) == str(g["dtype"])
        np.testing.assert_array_equal(tab, g["table"])
        np.testing.assert_array_equal(fn(960, 3), g["batch3"])


def test_rescale_noise_cfg_matches_reference():
    g = gold("g4_rescale.npz")
    from minsdtf_amd.stable_diffusion import rescale_noise_cfg
    from oracle import sd_oracle as O

    for phi in (0.3, 0.7, 1.0):
        for fn in (rescale_noise_cfg, O.rescale_noise_cfg):
            np.testing.assert_array_equal(fn(g["noise_cfg"], g["noise_text"], guidance_rescale=phi), g[f"out_{phi}"])


def test_resize_and_expand_match_reference():
    g = gold("g7_host_utils.npz")
    from minsdtf_amd.stable_diffusion import StableDiffusionBase

    np.testing.assert_allclose(StableDiffusionBase.resize(g["image"], 16, 24), g["resized_16_24"], rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(StableDiffusionBase.resize(g["image"], 5, 4), g["resized_5_4"], rtol=1e-6, atol=1e-4)
    sd = StableDiffusionBase(64, 64)
    np.testing.assert_array_equal(sd._expand_tensor(g["expand_in"], 3), g["expand_out"])
    np.testing.assert_array_equal(sd._expand_tensor(g["expand_in_b"], 1), g["expand_out_b"])


# ------------------------------------------------------------------ G2: the host loop itself
def _fake_unet(latent, t_emb, context):
    latent = np.asarray(latent, dtype=np.float32)
    c = np.asarray(context, dtype=np.float32).mean(axis=(1, 2))[:, None, None, None]
    t = np.asarray(t_emb, dtype=np.float32)[:, :8].mean(axis=1)[:, None, None, None]
    return (0.6 * latent + 0.25 * np.sin(3.0 * latent) + 0.2 * c + 0.1 * t).astype(np.float32)


def _fake_decoder(latent):
    latent = np.asarray(latent, dtype=np.float32)
    up = np.repeat(np.repeat(latent[..., :3], 8, axis=1), 8, axis=2)
    return np.tanh(up * 0.7).astype(np.float32)


def _fake_encoder(img):
    img = np.asarray(img, dtype=np.float32)
    b, h, w, _ = img.shape
    m = img.reshape(b, h // 8, 8, w // 8, 8, 3).mean(axis=(2, 4))
    return np.concatenate([m, m.mean(axis=-1, keepdims=True)], axis=-1).astype(np.float32) * 0.7


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e"])
def test_host_loop_matches_reference(tag):
    """generate_image(host_loop=True) with the same numpy fake models the reference's loop was run
    with: same call order and arguments (uncond before cond, one call when guidance <= 0), same
    uint8 image bit for bit."""
    from minsdtf_amd.stable_diffusion import StableDiffusionBase

    g = gold("g2_host_loop.npz")
    tr = gold("g2_host_loop_trace.json")[tag]
    trace = []

    class Fake:
        def __init__(self, fn, kind):
            self.fn, self.kind = fn, kind

        def predict_on_batch(self, x):
            if self.kind == "unet":
                trace.append(["unet", float(np.asarray(x[1])[0, 0]), float(np.asarray(x[2]).mean())])
                return self.fn(*x[:3])
            trace.append([self.kind])
            return self.fn(x)

    class Pipe(StableDiffusionBase):
        diffusion_model = property(lambda self: Fake(_fake_unet, "unet"))
        image_decoder = property(lambda self: Fake(_fake_decoder, "decoder"))
        image_encoder = property(lambda self: Fake(_fake_encoder, "encoder"))

    p = Pipe(64, 64)
    p.unconditional_context = g["uncond"]
    extra = {"reference_image": g["d_reference_image"]} if tag in ("d", "e") else {}   # image_to_image / inpaint runs
    if tag == "e":
        extra["inpaint_mask"] = g["e_inpaint_mask"]
    img = p.generate_image(g["context"], diffusion_noise=g[f"{tag}_noise"], host_loop=True, **tr["kwargs"], **extra)
    assert img.dtype == np.uint8 and img.shape == g[f"{tag}_image"].shape
    assert len(trace) == len(tr["calls"])
    for mine, ref in zip(trace, tr["calls"]):
        assert mine[0] == ref[0]
        np.testing.assert_allclose(mine[1:], ref[1:], rtol=1e-6)
    np.testing.assert_array_equal(img, g[f"{tag}_image"])


def test_oracle_loop_matches_reference_loop():
    """The oracle's denoise_loop + to_uint8 reproduce the reference loop's uint8 image (run 'a')."""
    from oracle import sd_oracle as O

    g = gold("g2_host_loop.npz")
    lat = O.denoise_loop(lambda l, t, c, ctl: _fake_unet(l, t, c), g["context"][None], g["uncond"], g["a_noise"], num_steps=25,
                         guidance=7.5, guidance_rescale=0.7)
    np.testing.assert_array_equal(O.to_uint8(_fake_decoder(lat)), g["a_image"])


def test_mask_preprocessing_matches_reference():
    """gaussian_blur (binomial filter, reflected borders) and preprocessed_mask (resize, channel mean,
    /255, blur, latent-resolution resize) against the reference's outputs (stable_diffusion.py:217-240,288-302)."""
    from minsdtf_amd.stable_diffusion import StableDiffusionBase

    g = gold("g7_host_utils.npz")
    p = StableDiffusionBase(64, 64)
    for r in (1, 3, 5):
        np.testing.assert_array_equal(p.gaussian_blur(g["blur_in"], radius=r, h_axis=0, v_axis=1), g[f"blur_{r}"])
    full, lat = p.preprocessed_mask(g["mask_in"], 5)
    np.testing.assert_array_equal(full, g["mask_full"])
    np.testing.assert_array_equal(lat, g["mask_latent"])
    assert full.shape == (1, 64, 64, 1) and lat.shape == (1, 8, 8, 1)
    np.testing.assert_array_equal(p.preprocessed_mask(np.stack([g["mask_in"]] * 3, -1), None)[1], g["mask_noblur_latent"])


def test_oracle_inpaint_loop_matches_reference_loop():
    """The oracle's loop with init latent + latent mask reproduces the reference's inpaint run 'e' (uint8 image)."""
    from minsdtf_amd.stable_diffusion import StableDiffusionBase
    from oracle import sd_oracle as O

    g = gold("g2_host_loop.npz")
    p = StableDiffusionBase(64, 64)
    img01, img11 = p.preprocessed_image(g["d_reference_image"])
    full, lat_mask = p.preprocessed_mask(g["e_inpaint_mask"], 5)
    lat = O.denoise_loop(lambda l, t, c, ctl: _fake_unet(l, t, c), np.repeat(g["context"][None], 2, 0), np.repeat(g["uncond"], 2, 0),
                         g["e_noise"], num_steps=10, guidance=5.0, guidance_rescale=0.7, init_latent=_fake_encoder(img11), strength=0.6,
                         latent_mask=lat_mask)
    decoded = (_fake_decoder(lat) + 1.0) * 0.5
    decoded = img01 * (1.0 - full) + np.array(decoded, dtype=np.float32) * full
    np.testing.assert_array_equal(np.clip(decoded * 255.0, 0, 255).astype("uint8"), g["e_image"])


def test_generate_image_argument_errors():
    from minsdtf_amd.stable_diffusion import StableDiffusionBase

    p = StableDiffusionBase(64, 64)
    with pytest.raises(ValueError):
        p.generate_image(np.zeros((77, 768), np.float32), diffusion_noise=np.zeros((8, 8, 4)), seed=1)
    with pytest.raises(NotImplementedError):
        p.encode_text("a string prompt")


# ------------------------------------------------------------------ G5: prompt parser cases (data only)
def test_prompt_attention_golden_is_wellformed():
    for case in gold("g5_prompt_attention.json"):
        assert isinstance(case["prompt"], str) and all(len(p) == 2 for p in case["parsed"])


# ------------------------------------------------------------------ G6: checkpoint layout contract
def test_weight_tables_match_reference_digests():
    from minsdtf_amd import weights as W

    g = gold("g6_ckpt_tables.json")
    for kind in ("civitai_model", "decoder", "controlnet", "hintnet", "encoder"):
        assert len(W.table(kind)) == g[kind]["count"]
        assert W.table_digest(kind) == g[kind]["sha256"]
    assert (W.param_count("civitai_model"), W.param_count("decoder"), W.param_count("controlnet"), W.param_count("hintnet")) == \
        (859520964, 49490199, 360192640, 1086480)
    assert W.param_count("encoder") == 34163664
    import hashlib

    h = hashlib.sha256()
    for s in W.table("civitai_model"):
        h.update(repr((s.key, s.alt_key + "" if s.alt_key else None)).encode())
    assert h.hexdigest() == g["UNET_KEY_MAPPING"]["sha256"]


@pytest.mark.skipif(not os.path.exists("/root/reference/stable_diffusion/ckpt_loader.py"), reason="reference not present")
def test_weight_tables_equal_reference_tables():
    import importlib.util

    from minsdtf_amd import weights as W

    spec = importlib.util.spec_from_file_location("ref_ckpt", "/root/reference/stable_diffusion/ckpt_loader.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for kind in m.CKPT_MAPPING:   # (the CLIP text tables are built in text_encoder.py's constructors: test_loaders_cpu.py)
        assert [(s.key, s.perm) for s in W.table(kind)] == [tuple(x) for x in m.CKPT_MAPPING[kind]]
    assert {s.key: s.alt_key for s in W.table("civitai_model")} == m.UNET_KEY_MAPPING


def test_synthetic_checkpoint_roundtrip(tmp_path):
    """Synthetic checkpoint written under the reference's keys (PyTorch layout) loads back through
    the positional loader into Keras layout, diffusers key spelling accepted as well."""
    from minsdtf_amd import weights as W

    path = str(tmp_path / "hint.safetensors")
    W.write_synthetic_checkpoint(path, kinds=("hintnet",), seed=3, bias_scale=0.1)

    class Model:
        name = "m"
        weights = [type("V", (), {"shape": s.shape, "name": s.name})() for s in W.table("hintnet")]

        def set_weights(self, arrs):
            self.got = arrs

    m = Model()
    W.load_weights_from_file(m, path, "hintnet")
    ref = W.synth_keras_weights("hintnet", seed=3, bias_scale=0.1)
    assert len(m.got) == 16
    for a, b, s in zip(m.got, ref, W.table("hintnet")):
        assert a.shape == s.shape
        np.testing.assert_array_equal(a, b)
    # deterministic and seed dependent
    again = W.synth_keras_weights("hintnet", seed=3, bias_scale=0.1)
    other = W.synth_keras_weights("hintnet", seed=4, bias_scale=0.1)
    assert all(np.array_equal(a, b) for a, b in zip(ref, again))
    assert not np.array_equal(ref[0], other[0])
    lim = np.sqrt(6.0 / (9 * 3 + 9 * 16))
    assert np.abs(ref[0]).max() <= lim and np.abs(ref[0]).max() > 0.9 * lim  # Glorot-uniform bound of conv 3->16


def test_geglu_packing_order():
    from minsdtf_amd.packing import geglu_row_order

    o = geglu_row_order(64)
    assert sorted(o.tolist()) == list(range(128))
    assert o[:16].tolist() == list(range(16)) and o[16:32].tolist() == list(range(64, 80)) and o[32] == 16


# ------------------------------------------------------------------ launch-plan allocator
def test_arena_recycles_and_never_overlaps_live_ranges():
    from minsdtf_amd.engine import Arena

    rng = np.random.default_rng(0)
    a = Arena()
    live = []
    high = 0
    for _ in range(2000):
        if live and rng.random() < 0.45:
            a.free(live.pop(rng.integers(len(live))))
        else:
            b = a.alloc(int(rng.integers(1, 5000)))
            assert b.offset % Arena.ALIGN == 0
            for o in live:
                assert b.offset + b.nbytes <= o.offset or o.offset + o.nbytes <= b.offset
            live.append(b)
        high = max(high, a.top)
    assert high < 2000 * 5000 / 4  # recycling keeps the footprint far below the sum of requests


def test_splitk_policy():
    from minsdtf_amd.engine import pick_splitk

    assert pick_splitk(8192, 320, 45) == 1          # 64x64 level: enough tiles already
    assert pick_splitk(128, 1280, 180) > 1          # 8x8 level at batch 1: spread K over the chip
    assert pick_splitk(128, 1280, 4) == 1           # short K: never split


def test_tuning_table_pins_one_numerics_class_per_layer():
    """A sample's bits must not depend on the batch it runs in: the launch parameters that order a layer's fp32 sums
    (kernel family, split-K slices, column tile of 1x1 / dense layers) are the same for every batch of a layer shape."""
    import json

    from minsdtf_amd import tuning

    table = json.load(open(os.path.join(ROOT, "minsdtf_amd", "conv_tuning.json")))
    assert not os.path.exists(os.path.join(ROOT, "minsdtf_amd", "conv_tuning_throughput.json")), "round 6: ONE table, one set of bits per layer"
    fams = {}
    for key, ent in table.items():
        b, rest = key.split("x", 1)
        ks = int(re.search(r"k(\d)s", rest).group(1))
        fams.setdefault(rest, set()).add(tuning.numerics_class(ks, int(ent[0]), int(ent[1]), int(ent[2]), tuning.key_is_ln_producer(key), stages=int(ent[3]) if len(ent) > 4 else 0))
    bad = {k: v for k, v in fams.items() if len(v) != 1}
    assert not bad, bad
    # the shortcut-folded 3x3 convs of the 16x16 / 32x32 / 64x64 levels: chunk-major (halo-tile kernel or staged-halo big form) at EVERY batch
    sc = {k: e for k, e in table.items() if re.search(r"x(16x16|32x32|64x64)x\d+->\d+k3s1u0\+x", k)}
    assert len(sc) >= 36 and all(tuning.is_halo(int(e[0])) or (tuning.is_big(int(e[0])) and int(e[3]) >= 20) for e in sc.values()), sc
    # an unmeasured batch lands in the same class as the measured ones; an unknown layer falls back per SAMPLE
    a = tuning.lookup(2, 8, 8, 1280, 1280, 3, 1, False, 128, 180, True)
    b = tuning.lookup(6, 8, 8, 1280, 1280, 3, 1, False, 384, 180, True)
    assert tuning.numerics_class(3, a[0], a[1], a[2]) == tuning.numerics_class(3, b[0], b[1], b[2])
    h1 = tuning.lookup(1, 40, 40, 704, 704, 3, 1, False, 1600, 99, True)
    h5 = tuning.lookup(5, 40, 40, 704, 704, 3, 1, False, 8000, 99, True)
    assert tuning.numerics_class(3, h1[0], h1[1], h1[2], stages=h1[3]) == tuning.numerics_class(3, h5[0], h5[1], h5[2], stages=h5[3])


def test_engine_contexts_accept_mixed_host_and_device_inputs():
    """DenoiseEngine.contexts: `context` a device tensor and `unconditional_context` a host array (or the reverse) end up as
    ONE concat on the tensor's device (the "meta" device stands in for the GPU: nothing can be copied OUT of it, which is
    what the earlier form tried)."""
    from types import SimpleNamespace

    import torch

    from minsdtf_amd.stable_diffusion import DenoiseEngine

    eng = SimpleNamespace(cfg=True, passes=[None])
    host = np.zeros((1, 77, 768), dtype=np.float32)
    on_dev = torch.empty(1, 77, 768, device="meta")
    for u, c in ((host, on_dev), (on_dev, host), (on_dev, on_dev)):
        both = DenoiseEngine.contexts(eng, u, c)["both"]
        assert both.device.type == "meta" and tuple(both.shape) == (2, 77, 768)
    assert isinstance(DenoiseEngine.contexts(eng, host, host)["both"], np.ndarray)


# ------------------------------------------------------------------ the C ABI
def test_library_exports_every_declared_symbol():
    """libminsdtf_hip.so loads and exports exactly what include/minsdtf_hip.h declares."""
    from minsdtf_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    header = open(os.path.join(ROOT, "include", "minsdtf_hip.h")).read()
    declared = set(re.findall(r"^MSD_API\s+(?:int|const char\*)\s+(msd_\w+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    assert not re.search(r"^(?:int|const char\*)\s+msd_\w+\s*\(", header, flags=re.M), "a declaration without MSD_API"
    # the dynamic symbol table is the header and nothing else (built with -fvisibility=hidden + csrc/exports.map: no mangled
    # C++ internals, no kernel handles)
    import subprocess
    nm = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.strip()}
    assert exported == declared, sorted(exported ^ declared)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.msd_abi_version() == _lib.ABI_VERSION
    # argument validation works without a GPU: nothing is launched for a bad call
    assert lib.msd_conv_gemm(None, None) == -1
    assert b"null" in lib.msd_last_error()
    assert lib.msd_set_option(b"no_such_key", 1) == -1


def test_struct_layouts_match_header():
    """ctypes mirrors have the field order / sizes of the C structs (checked against a compiled probe)."""
    import subprocess
    import tempfile

    from minsdtf_amd import _lib

    src = '#include <stdio.h>\n#include "minsdtf_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(MsdConvGemm), ' \
          'sizeof(MsdConvDirect), sizeof(MsdGroupNorm), sizeof(MsdAttention), sizeof(MsdCfgStep));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "p.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "p")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    mine = [ctypes.sizeof(t) for t in (_lib.MsdConvGemm, _lib.MsdConvDirect, _lib.MsdGroupNorm, _lib.MsdAttention, _lib.MsdCfgStep)]
    assert sizes == mine


def test_product_path_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under minsdtf_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "minsdtf_amd")):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


# ------------------------------------------------------------------ round-6 hygiene
def test_plan_caches_retire_stale_epochs(monkeypatch):
    """A reported cluster-GroupNorm give-up bumps engine.GN_EPOCH; every cache keyed on it must DROP the old-epoch entries
    (arena + hipGraph + sync block each) before the replacement is built, not keep them beside it."""
    from minsdtf_amd import engine, models

    released = []

    class Owner:
        def __init__(self, tag):
            self.tag = tag

        def release_graphs(self):
            released.append(self.tag)

    monkeypatch.setattr(engine, "GN_EPOCH", 3)
    cache = {(("a",), 2): Owner("old-a"), (("b",), 3): Owner("cur-b"), (("c",), 1): Owner("old-c")}
    assert engine.retire_stale(cache) == 2
    assert list(cache) == [(("b",), 3)] and sorted(released) == ["old-a", "old-c"]
    assert engine.retire_stale(cache) == 0
    # HipModel._bound purges before it builds
    m = models.HipModel.__new__(models.HipModel)
    m._plans = {(("x",), 2): Owner("old-x")}
    m._W = object()
    m.name = "stub"
    built = m._bound(("x",), lambda: Owner("new-x"))
    assert built.tag == "new-x" and list(m._plans) == [(("x",), 3)] and "old-x" in released


def test_engine_cache_holds_one_engine(monkeypatch):
    """StableDiffusion._engine: the resident engine is released BEFORE its replacement is constructed."""
    import minsdtf_amd.stable_diffusion as sdm
    from minsdtf_amd import engine

    events = []

    class FakeEngine:
        def __init__(self, *a, **kw):
            events.append("build")

        def release_graphs(self):
            events.append("release")

    class FakeModel:
        weights_version = 1

    monkeypatch.setattr(sdm, "DenoiseEngine", FakeEngine)
    sd = sdm.StableDiffusion.__new__(sdm.StableDiffusion)
    sd._engines, sd.denoise_streams, sd.active_tcd, sd.jit_compile = {}, 1, False, True
    sd._diffusion_model = FakeModel()
    sd._control_net = sd._hint_net = None
    e1 = sd._engine(1, 77, 77, 25, 7.5, 0.7, False)
    assert sd._engine(1, 77, 77, 25, 7.5, 0.7, False) is e1 and events == ["build"]
    monkeypatch.setattr(engine, "GN_EPOCH", engine.GN_EPOCH + 1)
    e2 = sd._engine(1, 77, 77, 25, 7.5, 0.7, False)
    assert e2 is not e1 and events == ["build", "release", "build"] and list(sd._engines.values()) == [e2]


def test_env_options_are_validated_before_the_library_is_cached(monkeypatch):
    """_lib.load(): a bad MSD_GN_ROWS / MSD_PROFILE raises HipExtensionError and leaves NO cached handle behind (the next
    load() must fail the same way, never run silently on the defaults)."""
    from minsdtf_amd import _lib

    _lib.load()   # (make sure the file exists / is loadable at all)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("MSD_GN_ROWS", "lots")
    for _ in range(2):
        with pytest.raises(_lib.HipExtensionError, match="MSD_GN_ROWS"):
            _lib.load()
        assert _lib._lib is None
    monkeypatch.setenv("MSD_GN_ROWS", "-5")
    with pytest.raises(_lib.HipExtensionError, match="MSD_GN_ROWS"):
        _lib.load()
    monkeypatch.delenv("MSD_GN_ROWS")
    monkeypatch.setenv("MSD_PROFILE", "throughput")   # round 5's second table is gone: a process that still asks for it is told so
    with pytest.raises(_lib.HipExtensionError, match="removed in round 6"):
        _lib.load()
    assert _lib._lib is None
    monkeypatch.delenv("MSD_PROFILE")
    lib = _lib.load()
    assert _lib._lib is lib
    assert lib.msd_set_option(b"gn_rows", 9216) == 0   # (back to the default for whatever test runs next in this process)


def test_release_library_refuses_experiment_modes():
    """msd_set_option("xattn160_mode", != 0) selects kernel instantiations that compute wrong results on purpose: they exist
    in the instrumented build (make stamps) only, the release library says so instead of switching."""
    from minsdtf_amd import _lib

    lib = _lib.load()
    assert lib.msd_set_option(b"xattn160_mode", 0) == 0
    for mode in (1, 2, 8, 15):
        assert lib.msd_set_option(b"xattn160_mode", mode) == -1
        assert b"instrumented build" in lib.msd_last_error()


def _walk_shapes(nb, h, w, what="unet"):
    """Every msd_conv_gemm shape of a network at a latent size, through the emitters (tensor-less weights): the tuner's walk."""
    from minsdtf_amd import engine, tuning

    rec = []
    orig = tuning.lookup

    def hook(batch, h_in, w_in, cin, N, ksize, stride, upsample, M, nk, allow_split, cx=0):
        rec.append((batch, h_in, w_in, cin, N, ksize, stride, bool(upsample), M, nk, bool(allow_split), cx))
        return tuning.heuristic(M, N, nk, allow_split)

    class AnyW(dict):
        def __contains__(self, k):
            return True

        def __missing__(self, k):
            return None

    class T:
        ptr = 0

        def at(self, off):
            return self

    tuning.lookup = hook
    try:
        p = engine.Plan("cpu")
        e = engine.Emitter(p, AnyW())
        if what == "unet":
            ctx = engine.Act(p.alloc(nb * 77 * 768 * 2), nb, 77, 1, 768)
            kv = engine.emit_context_kv(e, ctx, engine.UNET_ATTN_LAYERS, p)
            engine.emit_unet(e, T(), nb, nb, h, w, (T(), 0, 0, engine.temb_columns(False)), kv, 77, T(), None)
        else:
            engine.emit_decoder(e, T(), nb, h, w, T(), 0)
    finally:
        tuning.lookup = orig
    return rec


def _config_is_built(cfg, shape):
    """The (tile_m, tile_n, stages) of a launch configuration names a kernel the library builds AND that takes this shape (the
    tuner's candidate filters, tools/tune_conv.py): the wreg / big / staged-halo forms fail the launch otherwise."""
    from minsdtf_amd import tuning as t

    bm, bn, sk, stg = cfg
    batch, h_in, w_in, cin, N, ks, stride, ups, M, nk, allow_split, cx = shape
    hl, wl = (2 * h_in, 2 * w_in) if ups else (h_in, w_in)
    key = (bm, bn, stg)
    if t.is_halo(bm):
        return key in t.HALO_TILES and ks == 3 and stride == 1 and not ups and w_in % 16 == 0 and h_in % ((bm % 1000) // 16) == 0   # (round 6: with a shortcut operand too)
    if t.is_rowpanel(bm):
        return ks == 1 and stride == 1 and not ups and not cx and cin in t.ROWPANEL_ROWS and bm in t.ROWPANEL_ROWS[cin] and bn in t.ROWPANEL_COLS and N % bn == 0 and N % 32 == 0
    if t.is_wreg(bm):
        waves_n = 4 if (stg % 20 < 10 or bm == 4256) else 8
        return key in t.WREG_TILES and N % 16 == 0 and not (bn > 64 and N <= 64) and (allow_split or (bn // 16 // waves_n) % 2 == 0)
    if t.is_big(bm):
        ok = not (bn == 160 and (N % 160 or not allow_split)) and not (bn > 128 and N <= 128) and not (ks == 1 and allow_split and cin == N and not cx)
        if stg >= 20:
            return ok and key in t.BIG_TILES_HALO_IMAGE and ks == 3 and stride == 1 and not (cx and ups) and hl % 16 == 0 and wl % 16 == 0 and M >= t.HALO_IMAGE_MIN_ROWS
        if stg >= 10:
            return ok and key in t.BIG_TILES_CHUNK_MAJOR and ks == 3 and stride == 1 and not cx
        return ok and key in t.BIG_TILES
    if key not in t.TILES:
        return False
    if -(-N // bn) >= 256:   # column tiles travel in 8 bits of a packed launch argument (cg_hot_ok)
        return False
    return not (bm == 256 and M < 1024) and not (bn == 128 and N <= 64) and not (bn == 80 and (N % 80 or not allow_split)) and not (bn == 160 and (N % 160 or N < 1280))


def test_shape_config_reaches_the_fast_forms_at_untuned_sizes():
    """speed is a property of the shape class, not of a table row.  For image sizes the table has no entry of
    (640x640, 512x768, 1024x1024, 704x576) every conv / dense launch of the UNet and the VAE decoder gets (a) a configuration
    the library builds and that takes the shape, (b) ONE numerics class per layer whatever the batch, (c) the halo / staged-halo
    forms on the 3x3 convs whose geometry allows them (what a table row would have chosen) - never just the plain tile."""
    from minsdtf_amd import tuning

    table = tuning._load()
    for (lh, lw) in ((80, 80), (64, 96), (128, 128), (88, 72)):
        classes, forms = {}, {}
        for net in ("unet", "vae"):
            for nb in ((1, 2, 4, 8) if net == "unet" else (1, 4)):
                for sh in _walk_shapes(nb, lh, lw, net):
                    batch, h_in, w_in, cin, N, ks, stride, ups, M, nk, allow_split, cx = sh
                    key = tuning.shape_key(batch, h_in, w_in, cin, N, ks, stride, ups, allow_split, cx)
                    cfg = tuning.lookup(*sh)
                    layer = key.split("x", 1)[1]
                    if key not in table and layer not in tuning._families:   # (a table row was measured on the GPU: it launched)
                        assert _config_is_built(cfg, sh), (key, cfg)
                    cls = tuning.numerics_class(ks, cfg[0], cfg[1], cfg[2], tuning.key_is_ln_producer(key), cfg[3])
                    classes.setdefault(layer, set()).add(cls)
                    if key not in table and layer not in tuning._families:
                        forms.setdefault((ks, stride, ups, bool(cx)), set()).add((cfg[0], h_in, w_in))
        bad = {k: v for k, v in classes.items() if len(v) != 1}
        assert not bad, (lh, lw, bad)
        plain3 = {(bm, h, w) for (bm, h, w) in forms.get((3, 1, False, False), ()) if bm < 1000 and w % 16 == 0 and h % 8 == 0 and h * w >= 256}
        assert not plain3, f"{lh}x{lw}: 3x3 convs on the plain tile although the halo forms take them: {sorted(plain3)}"
        ups3 = {(bm, h, w) for (bm, h, w) in forms.get((3, 1, True, False), ()) if not tuning.is_big(bm) and (2 * h) % 16 == 0 and (2 * w) % 16 == 0 and h * w >= 256}
        assert not ups3, f"{lh}x{lw}: upsampling convs off the staged halo: {sorted(ups3)}"


def test_shape_class_reads_the_sample_only():
    """shape_class() has no batch argument; shape_config() may move the FORM with the batch but never the class."""
    from minsdtf_amd import tuning

    for (h, w, cin, N, ks, st, up, split, cx) in ((80, 80, 320, 320, 3, 1, False, True, 0), (40, 40, 640, 640, 3, 1, False, True, 1280),
                                                  (10, 10, 2560, 1280, 3, 1, False, True, 0), (80, 80, 320, 2560, 1, 1, False, False, 0),
                                                  (20, 20, 6400, 1280, 1, 1, False, True, 0), (40, 40, 1280, 1280, 3, 1, True, True, 0),
                                                  (80, 80, 320, 320, 3, 2, False, True, 0), (160, 160, 512, 512, 3, 1, False, True, 0),
                                                  (25, 1, 1280, 20160, 1, 1, False, True, 0)):
        pad = 1 if ks == 3 else 0
        hl, wl = (2 * h, 2 * w) if up else (h, w)
        ho, wo = (hl + 2 * pad - ks) // st + 1, (wl + 2 * pad - ks) // st + 1
        nk = ks * ks * (cin // 64) + cx // 64
        seen = set()
        for b in (1, 2, 3, 4, 8, 16, 32):
            cfg = tuning.shape_config(b, h, w, cin, N, ks, st, up, b * ho * wo, nk, split, cx)
            seen.add(tuning.numerics_class(ks, cfg[0], cfg[1], cfg[2], ks == 1 and split and cin == N and not cx, cfg[3]))
        assert len(seen) == 1, ((h, w, cin, N, ks, st, up, split, cx), seen)


def test_effective_cpus_respects_affinity_and_quota(tmp_path, monkeypatch):
    """minsdtf_amd.host.effective_cpus: never more than os.cpu_count() or the affinity mask, and the cgroup quota where one is
    set; fit_torch_threads only ever lowers torch's thread count."""
    import builtins

    import torch

    from minsdtf_amd import host

    n = host.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            p = tmp_path / "cpu.max"
            p.write_text("300000 100000\n")
            return real_open(p, *a, **k)
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, "open", fake_open)
    assert host.effective_cpus() == min(3, n if n < 3 else 3)
    before = torch.get_num_threads()
    try:
        assert host.fit_torch_threads() == min(before, host.effective_cpus())
    finally:
        torch.set_num_threads(before)

"""Capture golden vectors from the importable parts of the reference (run in the build container,
where /root/reference exists; the outputs under tests/golden/ are data, the reference sources never
leave this container).

What the reference lets us pin without Keras:
  G1 scheduler.py (numpy only, imported by path): schedule constants, timestep lists, step() outputs
  G2 StableDiffusionBase.generate_image (imported with a stub `keras` module, numpy fake models):
     call order / arguments of the model calls, CFG + rescale + scheduler composition, final-step
     branch, batch tiling, uint8 conversion
  G3 _get_timestep_embedding table     G4 rescale_noise_cfg      G7 resize (bilinear)
  G6 ckpt_loader tables: ordered (key, perm) lists -> counts + SHA-256 digests
  G5 long_prompt_weighting.parse_prompt_attention doctest cases (host text munging, informative)

    python tools/make_goldens.py
"""
import hashlib
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def install_keras_stub():
    """Inert `keras` so that `import stable_diffusion` succeeds; nothing numeric is provided."""
    import torch  # noqa: F401  (must be imported before the stub: torch inspects modules at import)

    class _Inert:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

    class _Mod(types.ModuleType):
        def __getattr__(self, item):
            if item.startswith("__"):
                raise AttributeError(item)
            return _Inert

    names = ["keras", "keras.layers", "keras.utils", "keras.random", "keras.activations", "keras.ops"]
    mods = {n: _Mod(n) for n in names}
    for n, m in mods.items():
        m.__file__ = "<stub>"
        m.__path__ = []
        sys.modules[n] = m
    k = mods["keras"]
    for sub in ("layers", "utils", "random", "activations", "ops"):
        setattr(k, sub, mods["keras." + sub])
    k.Model = _Inert
    k.Sequential = _Inert
    mods["keras.layers"].Layer = _Inert

    class Progbar:
        def __init__(self, *a, **k):
            pass

        def update(self, *a, **k):
            pass

    mods["keras.utils"].Progbar = Progbar
    mods["keras.utils"].get_file = lambda *a, **k: "/nonexistent"

    def _no_rng(*a, **k):
        raise RuntimeError("keras.random is backend specific: inject diffusion_noise")

    mods["keras.random"].normal = _no_rng


def fake_unet(latent, t_emb, context):
    """Deterministic numpy stand-in for DiffusionModel.predict_on_batch (depends on all 3 inputs)."""
    latent = np.asarray(latent, dtype=np.float32)
    c = np.asarray(context, dtype=np.float32).mean(axis=(1, 2))[:, None, None, None]
    t = np.asarray(t_emb, dtype=np.float32)[:, :8].mean(axis=1)[:, None, None, None]
    return (0.6 * latent + 0.25 * np.sin(3.0 * latent) + 0.2 * c + 0.1 * t).astype(np.float32)


def fake_decoder(latent):
    latent = np.asarray(latent, dtype=np.float32)
    up = np.repeat(np.repeat(latent[..., :3], 8, axis=1), 8, axis=2)
    return np.tanh(up * 0.7).astype(np.float32)


def fake_encoder(img):
    """numpy stand-in for ImageEncoder: (1,H,W,3) in [-1,1] -> (1,H/8,W/8,4)."""
    img = np.asarray(img, dtype=np.float32)
    b, h, w, _ = img.shape
    m = img.reshape(b, h // 8, 8, w // 8, 8, 3).mean(axis=(2, 4))
    return np.concatenate([m, m.mean(axis=-1, keepdims=True)], axis=-1).astype(np.float32) * 0.7


TOKENIZER_STRINGS = [
    "a photograph of an astronaut riding a horse", "A PHOTOGRAPH, of an Astronaut!!  riding\ta horse...", "the cat's hat isn't here; they'll say we've won",
    "3 dogs and 42 cats in 2024", "caf\u00e9 na\u00efve \u65e5\u672c\u8a9e \U0001f600", "&lt;b&gt;bold&amp;amp;brave&lt;/b&gt;", "  ", "*", "masterpiece,best quality,(ultra-detailed:1.2)",
    "<|startoftext|>hello<|endoftext|>", "supercalifragilisticexpialidocious antidisestablishmentarianism",
]
PROMPTS = [
    "a photograph of an astronaut riding a horse",
    "a (very beautiful:1.3) [blurry] ((masterpiece)) of a cat, \\(literal\\)",
    ", ".join(["a photograph of an (astronaut:1.2) riding a [horse] on the moon, highly detailed"] * 12),   # 150 < tokens <= 225: 3 windows
    "",
]


def train_toy_bpe(n_merges=400):
    """A small merge list in CLIP's file format, learnt from a toy corpus (the real list is a download)."""
    sys.path.insert(0, ROOT)
    from minsdtf_amd.text import byte_alphabet

    corpus = ("a photograph of an astronaut riding a horse on the moon highly detailed masterpiece best quality ultra detailed "
              "the cat hat is not here they will say we have won dogs and cats in bold brave very beautiful blurry literal "
              "hello super cali fragilistic expiali docious anti dis establishment arianism cafe naive riding photo graph ") * 3
    enc = byte_alphabet()
    words = {}
    for w in corpus.split():
        sym = tuple(enc[b] for b in w.encode("utf-8"))
        sym = sym[:-1] + (sym[-1] + "</w>",)
        words[sym] = words.get(sym, 0) + 1
    merges = []
    for _ in range(n_merges):
        counts = {}
        for sym, c in words.items():
            for p in zip(sym, sym[1:]):
                counts[p] = counts.get(p, 0) + c
        if not counts:
            break
        best = max(sorted(counts), key=lambda p: counts[p])
        merges.append(best)
        new_words = {}
        for sym, c in words.items():
            out, i = [], 0
            while i < len(sym):
                if i + 1 < len(sym) and (sym[i], sym[i + 1]) == best:
                    out.append(sym[i] + sym[i + 1])
                    i += 2
                else:
                    out.append(sym[i])
                    i += 1
            new_words[tuple(out)] = new_words.get(tuple(out), 0) + c
        words = new_words
    return "#version: toy\n" + "\n".join(" ".join(m) for m in merges)


class FakeClipEmbedding:
    """Deterministic numpy stand-ins for TextClipEmbedding / TextEncoder (dim 8) for the prompt-weighting goldens."""

    def predict_on_batch(self, x):
        ids, pos = np.asarray(x[0], dtype=np.float64), np.asarray(x[1], dtype=np.float64)
        base = ids[..., None] * 0.37 + pos[..., None] * 0.11 + np.arange(8)[None, None, :] * 0.5
        return np.sin(base).astype(np.float32)


class FakeTextEncoder:
    def predict_on_batch(self, emb):
        emb = np.asarray(emb, dtype=np.float32)
        mix = np.cumsum(emb, axis=1) / np.arange(1, emb.shape[1] + 1, dtype=np.float32)[None, :, None]   # causal mixing
        return np.tanh(emb * 1.3 + mix + 0.2).astype(np.float32)


def make_text_goldens():
    """G10: the reference's SimpleTokenizer and get_weighted_text_embeddings on a toy merge list (needs the keras stub)."""
    import gzip

    import stable_diffusion.clip_tokenizer as ref_tok
    import stable_diffusion.long_prompt_weighting as ref_lpw

    bpe_path = os.path.join(OUT, "g10_toy_bpe_merges.txt.gz")
    with gzip.GzipFile(bpe_path, "wb", mtime=0) as f:
        f.write(train_toy_bpe().encode("utf-8"))
    tok = ref_tok.SimpleTokenizer(bpe_path)
    g10 = {"vocab_size": len(tok.vocab), "start": tok.start_of_text, "end": tok.end_of_text,
           "encode": [[s, tok.encode(s)] for s in TOKENIZER_STRINGS],
           "decode": [tok.decode(tok.encode(s)) for s in TOKENIZER_STRINGS[:4]]}
    tok.add_tokens(["<cat-toy>", "the"])
    g10["after_add"] = {"vocab_size": len(tok.vocab), "encode": tok.encode("a <cat-toy> on the moon")}
    json.dump(g10, open(os.path.join(OUT, "g10_tokenizer.json"), "w"), indent=0)
    tok = ref_tok.SimpleTokenizer(bpe_path)
    arrays = {}
    ti = np.random.default_rng(9).standard_normal((1, 3, 8)).astype(np.float32)
    arrays["ti_embedding"] = ti
    for i, prompt in enumerate(PROMPTS):
        for nbm in (False, True):
            arrays[f"p{i}_nbm{int(nbm)}"] = ref_lpw.get_weighted_text_embeddings(tok, FakeClipEmbedding(), FakeTextEncoder(), prompt,
                                                                                  no_boseos_middle=nbm, pad_token_id=tok.end_of_text)
    arrays["p1_skipw"] = ref_lpw.get_weighted_text_embeddings(tok, FakeClipEmbedding(), FakeTextEncoder(), PROMPTS[1], skip_weighting=True,
                                                             pad_token_id=tok.end_of_text)
    arrays["p2_mult2"] = ref_lpw.get_weighted_text_embeddings(tok, FakeClipEmbedding(), FakeTextEncoder(), PROMPTS[2], max_embeddings_multiples=2,
                                                             pad_token_id=tok.end_of_text)
    for i in (0, 2):
        arrays[f"p{i}_ti"] = ref_lpw.get_weighted_text_embeddings(tok, FakeClipEmbedding(), FakeTextEncoder(), PROMPTS[i], embedding=ti,
                                                                  embedding_tokens_count=3, pad_token_id=tok.end_of_text)
    arrays["batch"] = ref_lpw.get_weighted_text_embeddings(tok, FakeClipEmbedding(), FakeTextEncoder(), [PROMPTS[0], PROMPTS[1]],
                                                          pad_token_id=tok.end_of_text)
    np.savez_compressed(os.path.join(OUT, "g10_prompt_weighting.npz"), **arrays)
    cases = ["normal text", "an (important) word", "(unbalanced", "\\(literal\\]", "(unnecessary)(parens)", "a (((house:1.3)) [on] a (hill:0.5), sun, (((sky))).",
             "", "\\", "a:b (c:d) :1.5) x", "[[x]] (y:+.5) (z:-2.)", "]) unopened ([ nested (a [b:1.2) c]"]
    json.dump([{"prompt": c, "parsed": ref_lpw.parse_prompt_attention(c)} for c in cases],
              open(os.path.join(OUT, "g5_prompt_attention.json"), "w"), indent=1)


def lora_fixture(seed=11):
    """A small kohya-style LoRA state dict: every UNet layer type the reference can restore (names made
    from its own UNET_KEY_MAPPING), a few text-encoder entries and a few names it cannot restore."""
    import torch

    sys.path.insert(0, ROOT)
    from minsdtf_amd import weights as W

    rng = np.random.default_rng(seed)
    sd = {}

    def add(name, shape_up, shape_down):
        sd[name + ".lora_up.weight"] = torch.from_numpy(rng.standard_normal(shape_up).astype(np.float32))
        sd[name + ".lora_down.weight"] = torch.from_numpy(rng.standard_normal(shape_down).astype(np.float32))
        sd[name + ".alpha"] = torch.tensor(float(rng.integers(1, 9)))

    for spec in W.table("civitai_model"):
        if spec.alt_key is None or not spec.alt_key.endswith(".weight") or spec.kind not in ("conv_w", "dense_w"):
            continue
        name = "lora_unet_" + spec.alt_key[:-7].replace(".", "_")
        ts = spec.torch_shape
        if len(ts) == 2:
            add(name, (4, 2), (2, 3))
        elif ts[2] == 1:
            add(name, (4, 2, 1, 1), (2, 3, 1, 1))
        else:
            add(name, (4, 2, 1, 1), (2, 3, 3, 3))
    for i in (0, 11):
        for mod in ("mlp_fc1", "mlp_fc2", "self_attn_q_proj", "self_attn_k_proj", "self_attn_v_proj", "self_attn_out_proj"):
            add(f"lora_te_text_model_encoder_layers_{i}_{mod}", (4, 2), (2, 3))
    return sd


class FakeKerasModel:
    """The loader surface of a Keras model: name, weights[i].shape/.name, set_weights."""

    class _W:
        def __init__(self, shape, name):
            self.shape, self.name = tuple(shape), name

    def __init__(self, name, specs):
        self.name = name
        self.weights = [self._W(s.shape, s.name) for s in specs]
        self.loaded = None

    def set_weights(self, ws):
        self.loaded = [np.asarray(w) for w in ws]


def digest_arrays(arrs):
    h = hashlib.sha256()
    for a in arrs:
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def make_loader_goldens(ck):
    import tempfile

    import torch
    from safetensors.torch import save_file

    sys.path.insert(0, ROOT)
    from minsdtf_amd import weights as W

    g8 = {}
    with tempfile.TemporaryDirectory() as d:
        # (a) LoRA file -> delta dicts through the reference's load_weights_from_lora
        lpath = os.path.join(d, "lora.safetensors")
        save_file(lora_fixture(), lpath)
        te, un = ck.load_weights_from_lora(lpath)
        # the reference also files the layers its substitutions cannot restore (conv_in, conv_out, time_embedding)
        # under names no weight has; the loader never matches them — keep them apart
        real = {s.alt_key for s in W.table("civitai_model") if s.alt_key}
        g8["lora_unet_keys"] = sorted(k for k in un if k in real)
        g8["lora_unet_unmatched"] = sorted(k for k in un if k not in real)
        g8["lora_te_keys"] = sorted(te.keys())
        g8["lora_unet_digest"] = digest_arrays([un[k] for k in g8["lora_unet_keys"]])
        g8["lora_te_digest"] = digest_arrays([te[k] for k in sorted(te.keys())])
        # (b) positional load through the reference's load_weights_from_file
        #     hintnet: whole table, LDM keys only;  UNet: first 120 rows, half the tensors stored under the
        #     diffusers (fallback) keys, with the LoRA deltas resized to the real shapes of those rows
        for kind, nrows, use_alt, with_lora in (("hintnet", None, False, False), ("civitai_model", 120, True, True),
                                                ("encoder", 30, False, False)):
            specs = W.table(kind)[:nrows]
            rng = np.random.default_rng(5)
            sd, lora = {}, {}
            for i, s in enumerate(specs):
                w = rng.standard_normal(s.torch_shape).astype(np.float32)
                key = s.alt_key if (use_alt and s.alt_key is not None and i % 2) else s.key
                sd[key] = torch.from_numpy(w)
                if with_lora and s.alt_key is not None and s.alt_key in un and i % 3 == 0:
                    lora[s.alt_key] = rng.standard_normal(s.torch_shape).astype(np.float32) * 0.1
            path = os.path.join(d, kind + ".safetensors")
            save_file(sd, path)
            fake = FakeKerasModel(kind, specs)
            mapping = ck.CKPT_MAPPING[kind][:nrows] if nrows else ck.CKPT_MAPPING[kind]
            ck.load_weights_from_file(fake, path, mapping, key_mapping=ck.UNET_KEY_MAPPING if use_alt else None,
                                      lora_dict=dict(lora) if with_lora else None)
            g8[f"load_{kind}_digest"] = digest_arrays(fake.loaded)
            g8[f"load_{kind}_count"] = len(fake.loaded)
            g8[f"load_{kind}_lora_applied"] = len(lora)
    json.dump(g8, open(os.path.join(OUT, "g8_loaders.json"), "w"), indent=0, sort_keys=True)


def main():
    os.makedirs(OUT, exist_ok=True)
    # ---------------------------------------------------------------- G1 scheduler
    sch_mod = by_path("ref_scheduler", "stable_diffusion/scheduler.py")
    S = sch_mod.Scheduler(active_tcd=False)
    g1 = {"alphas_cumprod_idx": np.array([0, 1, 500, 999]), "alphas_cumprod": S.alphas_cumprod[[0, 1, 500, 999]],
          "signal_rates_full": S.signal_rates, "noise_rates_full": S.noise_rates}
    for n in (1, 4, 25, 50):
        S.set_timesteps(n)
        g1[f"timesteps_{n}"] = np.asarray(S.timesteps)
    rng = np.random.default_rng(42)
    for n in (4, 25):
        S = sch_mod.Scheduler(active_tcd=False)
        S.set_timesteps(n)
        lat = rng.standard_normal((1, 8, 8, 4)).astype(np.float32)
        g1[f"run{n}_latent0"] = lat
        x = lat
        outs, eps_all = [], []
        for t in S.timesteps:
            eps = rng.standard_normal((1, 8, 8, 4)).astype(np.float32)
            x = S.step(eps, t, x)
            outs.append(np.asarray(x, dtype=np.float64))
            eps_all.append(eps)
        g1[f"run{n}_eps"] = np.stack(eps_all)
        g1[f"run{n}_out"] = np.stack(outs)
    # TCD sampler (scheduler.py:136-237,286-307): schedule and the stochastic step under a seeded global numpy RNG
    rng_t = np.random.default_rng(43)
    for n in (1, 4, 25, 50):
        S = sch_mod.Scheduler(active_tcd=True)
        S.set_timesteps(n)
        g1[f"tcd_timesteps_{n}"] = np.asarray(S.timesteps)
    for n in (4, 8):
        S = sch_mod.Scheduler(active_tcd=True)
        S.set_timesteps(n)
        x = rng_t.standard_normal((2, 8, 8, 4)).astype(np.float32)
        g1[f"tcd_run{n}_latent0"] = x
        np.random.seed(1000 + n)
        outs, eps_all = [], []
        for t in S.timesteps:
            eps = rng_t.standard_normal((2, 8, 8, 4)).astype(np.float32)
            x = S.step(eps, t, x)
            outs.append(np.asarray(x, dtype=np.float64))
            eps_all.append(eps)
        g1[f"tcd_run{n}_eps"], g1[f"tcd_run{n}_out"] = np.stack(eps_all), np.stack(outs)
    np.savez_compressed(os.path.join(OUT, "g1_scheduler.npz"), **g1)

    # ---------------------------------------------------------------- G6 checkpoint tables
    ck = by_path("ref_ckpt_loader", "stable_diffusion/ckpt_loader.py")
    g6 = {}
    for kind, tab in ck.CKPT_MAPPING.items():
        h = hashlib.sha256()
        for key, perm in tab:
            h.update(repr((key, tuple(perm) if perm is not None else None)).encode())
        g6[kind] = {"count": len(tab), "sha256": h.hexdigest()}
    h = hashlib.sha256()
    for k, v in ck.UNET_KEY_MAPPING.items():
        h.update(repr((k, v)).encode())
    g6["UNET_KEY_MAPPING"] = {"count": len(ck.UNET_KEY_MAPPING), "sha256": h.hexdigest()}
    json.dump(g6, open(os.path.join(OUT, "g6_ckpt_tables.json"), "w"), indent=1, sort_keys=True)

    # ---------------------------------------------------------------- G5 prompt parser
    lpw = by_path("ref_lpw", "stable_diffusion/long_prompt_weighting.py")
    cases = ["normal text", "an (important) word", "(unbalanced", "\\(literal\\]", "(unnecessary)(parens)",
             "a (((house:1.3)) [on] a (hill:0.5), sun, (((sky)))."]
    json.dump([{"prompt": c, "parsed": lpw.parse_prompt_attention(c)} for c in cases],
              open(os.path.join(OUT, "g5_prompt_attention.json"), "w"), indent=1)

    # ---------------------------------------------------------------- G2/G3/G4/G7 host loop (stub keras)
    install_keras_stub()
    sys.path.insert(0, REF)
    import stable_diffusion.stable_diffusion as ref_sd

    trace = []

    class FakeModel:
        def __init__(self, fn, tag):
            self.fn, self.tag = fn, tag

        def predict_on_batch(self, x):
            if self.tag == "unet":
                trace.append(("unet", float(np.asarray(x[1])[0, 0]), float(np.asarray(x[2]).mean())))
                return self.fn(*x[:3])
            trace.append((self.tag,))
            return self.fn(x)

    class RefPipe(ref_sd.StableDiffusionBase):
        @property
        def diffusion_model(self):
            return FakeModel(fake_unet, "unet")

        @property
        def image_decoder(self):
            return FakeModel(fake_decoder, "decoder")

        @property
        def image_encoder(self):
            return FakeModel(fake_encoder, "encoder")

    g2 = {}
    rng = np.random.default_rng(7)
    ctx = rng.standard_normal((77, 768)).astype(np.float32)
    unc = rng.standard_normal((1, 77, 768)).astype(np.float32)
    RefPipe._get_unconditional_context = lambda self: unc
    g2["context"], g2["uncond"] = ctx, unc
    runs = {"a": dict(batch_size=1, num_steps=25, unconditional_guidance_scale=7.5, guidance_rescale=0.7),
            "b": dict(batch_size=2, num_steps=4, unconditional_guidance_scale=5.0, guidance_rescale=0.0),
            "c": dict(batch_size=1, num_steps=3, unconditional_guidance_scale=0.0, guidance_rescale=0.0),
            # image_to_image: strength 0.8 of 25 steps -> 20 steps from t=760, init latent noised at t=800
            "d": dict(batch_size=2, num_steps=25, unconditional_guidance_scale=7.5, guidance_rescale=0.7,
                      reference_image_strength=0.8),
            # inpaint: image_to_image + a blurred mask; per-step latent blend and final pixel blend
            "e": dict(batch_size=2, num_steps=10, unconditional_guidance_scale=5.0, guidance_rescale=0.7,
                      reference_image_strength=0.6, mask_blur_strength=5)}
    ref_img = rng.integers(0, 256, (40, 56, 3)).astype(np.uint8)   # not the model size: exercises the bilinear resize
    g2["d_reference_image"] = ref_img
    mask_img = np.zeros((40, 56), np.uint8)
    mask_img[8:30, 10:40] = 255
    g2["e_inpaint_mask"] = mask_img
    traces = {}
    for tag, kw in runs.items():
        pipe = RefPipe(64, 64)
        noise = rng.standard_normal((kw["batch_size"], 8, 8, 4)).astype(np.float32)
        del trace[:]
        extra = {"reference_image": ref_img} if "reference_image_strength" in kw else {}
        if "mask_blur_strength" in kw:
            extra["inpaint_mask"] = mask_img
        img = pipe.generate_image(ctx, diffusion_noise=noise, **kw, **extra)
        g2[f"{tag}_noise"], g2[f"{tag}_image"] = noise, img
        traces[tag] = {"kwargs": kw, "calls": list(trace)}
        # the final latent is not returned by the reference; recover it by re-running its own loop pieces
    np.savez_compressed(os.path.join(OUT, "g2_host_loop.npz"), **g2)
    json.dump(traces, open(os.path.join(OUT, "g2_host_loop_trace.json"), "w"), indent=0)

    pipe = RefPipe(64, 64)
    pipe.scheduler.set_timesteps(25)
    emb = np.stack([np.asarray(pipe._get_timestep_embedding(int(t), 1))[0] for t in pipe.scheduler.timesteps])
    g3 = {"timesteps": np.asarray(pipe.scheduler.timesteps), "table": emb, "dtype": np.array(str(emb.dtype)),
          "batch3": np.asarray(pipe._get_timestep_embedding(960, 3))}
    np.savez_compressed(os.path.join(OUT, "g3_timestep_embedding.npz"), **g3)

    g4 = {}
    a = rng.standard_normal((2, 8, 8, 4)).astype(np.float32)
    b = (a * 0.4 + 0.3 * rng.standard_normal((2, 8, 8, 4))).astype(np.float32)
    g4["noise_cfg"], g4["noise_text"] = a, b
    for phi in (0.3, 0.7, 1.0):
        g4[f"out_{phi}"] = pipe.rescale_noise_cfg(a, b, guidance_rescale=phi)
    np.savez_compressed(os.path.join(OUT, "g4_rescale.npz"), **g4)

    img = rng.uniform(0, 255, (13, 9, 3)).astype(np.float32)
    g7 = {"image": img, "resized_16_24": ref_sd.StableDiffusionBase.resize(img, 16, 24),
          "resized_5_4": ref_sd.StableDiffusionBase.resize(img, 5, 4),
          "blur_in": img[..., :1] / 255.0, "blur_3": pipe.gaussian_blur(img[..., :1] / 255.0, radius=3, h_axis=0, v_axis=1),
          "blur_5": pipe.gaussian_blur(img[..., :1] / 255.0, radius=5, h_axis=0, v_axis=1),
          "blur_1": pipe.gaussian_blur(img[..., :1] / 255.0, radius=1, h_axis=0, v_axis=1),
          "mask_in": mask_img, "mask_full": pipe.preprocessed_mask(mask_img, 5)[0], "mask_latent": pipe.preprocessed_mask(mask_img, 5)[1],
          "mask_noblur_latent": pipe.preprocessed_mask(np.stack([mask_img] * 3, -1), None)[1],
          "expand_in": ctx[:5, :6], "expand_out": pipe._expand_tensor(ctx[:5, :6], 3),
          "expand_in_b": ctx[None, :5, :6], "expand_out_b": pipe._expand_tensor(ctx[None, :5, :6], 1)}
    np.savez_compressed(os.path.join(OUT, "g7_host_utils.npz"), **g7)
    # ---------------------------------------------------------------- G8 checkpoint / LoRA loaders (§8f rank 2)
    make_loader_goldens(ck)
    # ---------------------------------------------------------------- G9 CLIP text model checkpoint mappings (§8f rank 3)
    # text_encoder.py builds its (key, perm) lists inside the constructors; capture them by constructing the
    # classes under the keras stub with the loader call intercepted
    import tempfile

    import stable_diffusion.text_encoder as ref_te

    captured = {}

    def capture(model, ckpt_path, ckpt_mapping, key_mapping=None, lora_dict=None):
        captured["mapping"] = list(ckpt_mapping)

    ref_te.load_weights_from_file = capture
    g9 = {}
    with tempfile.NamedTemporaryFile(suffix=".safetensors") as tmp:
        for skip in (-1, -2, -12):
            ref_te.TextEncoder(77, clip_skip=skip, ckpt_path=tmp.name)
            h = hashlib.sha256()
            for key, perm in captured["mapping"]:
                h.update(repr((key, tuple(perm) if perm is not None else None)).encode())
            g9[f"text_encoder_clip_skip_{skip}"] = {"count": len(captured["mapping"]), "sha256": h.hexdigest()}
        ref_te.TextClipEmbedding(77, ckpt_path=tmp.name)
        g9["text_clip_embedding"] = [[k, p] for k, p in captured["mapping"]]
    json.dump(g9, open(os.path.join(OUT, "g9_text_tables.json"), "w"), indent=0, sort_keys=True)
    make_text_goldens()
    print("goldens written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f:36s} {os.path.getsize(os.path.join(OUT, f)):8d} B")


if __name__ == "__main__":
    main()

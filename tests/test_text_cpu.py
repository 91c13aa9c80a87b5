"""Host text front-end: CLIP BPE tokenizer and prompt weighting against the
reference's SimpleTokenizer / parse_prompt_attention / get_weighted_text_embeddings run on the same toy
merge list and the same numpy stand-in text models (tools/make_goldens.py -> tests/golden/g5*, g10*)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tools"))
BPE = os.path.join(GOLD, "g10_toy_bpe_merges.txt.gz")


def test_byte_alphabet_is_the_gpt2_table():
    from minsdtf_amd.text import byte_alphabet

    t = byte_alphabet()
    assert len(t) == 256 and len(set(t.values())) == 256
    assert t[ord("a")] == "a" and t[ord(" ")] == chr(256 + 32) and t[0] == chr(256) and t[173] == chr(256 + 67)
    assert list(t)[:3] == [33, 34, 35] and list(t)[-1] == 173          # printable bytes first: this order is the vocabulary order


def test_tokenizer_matches_reference():
    from minsdtf_amd.text import SimpleTokenizer

    g = json.load(open(os.path.join(GOLD, "g10_tokenizer.json")))
    tok = SimpleTokenizer(BPE)
    assert len(tok.vocab) == g["vocab_size"] and tok.start_of_text == g["start"] and tok.end_of_text == g["end"]
    for text, ids in g["encode"]:
        assert tok.encode(text) == ids, text
    for (text, _), dec in zip(g["encode"][:4], g["decode"]):
        assert tok.decode(tok.encode(text)) == dec
    assert tok.add_tokens(["<cat-toy>", "the"]) >= 1
    assert len(tok.vocab) == g["after_add"]["vocab_size"]
    assert tok.encode("a <cat-toy> on the moon") == g["after_add"]["encode"]


def test_parse_prompt_attention_matches_reference():
    from minsdtf_amd.text import parse_prompt_attention

    for case in json.load(open(os.path.join(GOLD, "g5_prompt_attention.json"))):
        assert parse_prompt_attention(case["prompt"]) == case["parsed"], case["prompt"]   # float weights bit for bit


def test_weighted_text_embeddings_match_reference():
    from make_goldens import PROMPTS, FakeClipEmbedding, FakeTextEncoder

    from minsdtf_amd.text import SimpleTokenizer, get_weighted_text_embeddings

    g = np.load(os.path.join(GOLD, "g10_prompt_weighting.npz"))
    tok = SimpleTokenizer(BPE)
    kw = dict(pad_token_id=tok.end_of_text)
    emb, enc = FakeClipEmbedding(), FakeTextEncoder()
    for i, prompt in enumerate(PROMPTS):
        for nbm in (False, True):
            out = get_weighted_text_embeddings(tok, emb, enc, prompt, no_boseos_middle=nbm, **kw)
            np.testing.assert_array_equal(out, g[f"p{i}_nbm{int(nbm)}"])
    assert g["p2_nbm0"].shape == (1, 3 * 77, 8) and g["p2_nbm1"].shape == (1, 3 * 75 + 2, 8)    # a 3-window prompt
    np.testing.assert_array_equal(get_weighted_text_embeddings(tok, emb, enc, PROMPTS[1], skip_weighting=True, **kw), g["p1_skipw"])
    np.testing.assert_array_equal(get_weighted_text_embeddings(tok, emb, enc, PROMPTS[2], max_embeddings_multiples=2, **kw), g["p2_mult2"])
    for i in (0, 2):   # textual inversion: 3 learned vectors injected after the start token of the first window
        np.testing.assert_array_equal(get_weighted_text_embeddings(tok, emb, enc, PROMPTS[i], embedding=g["ti_embedding"],
                                                                   embedding_tokens_count=3, **kw), g[f"p{i}_ti"])
    np.testing.assert_array_equal(get_weighted_text_embeddings(tok, emb, enc, [PROMPTS[0], PROMPTS[1]], **kw), g["batch"])


def test_pipeline_string_prompt_path(tmp_path):
    """encode_text("...") = tokenizer + weighting + the two text models (reference stable_diffusion.py:176-215);
    without a merge list the error says what to provide."""
    from make_goldens import FakeClipEmbedding, FakeTextEncoder

    from minsdtf_amd.stable_diffusion import StableDiffusion
    from minsdtf_amd.text import SimpleTokenizer, get_weighted_text_embeddings

    sd = StableDiffusion(64, 64, device="cpu")
    old = os.environ.pop("MSD_BPE_PATH", None)
    try:
        with pytest.raises(NotImplementedError, match="bpe_simple_vocab"):
            sd.encode_text("a string prompt")
    finally:
        if old is not None:
            os.environ["MSD_BPE_PATH"] = old
    sd.bpe_path = BPE
    sd._text_clip_embedding, sd._text_encoder = FakeClipEmbedding(), FakeTextEncoder()
    ctx = sd.encode_text("a (very beautiful:1.3) cat")
    tok = SimpleTokenizer(BPE)
    np.testing.assert_array_equal(ctx, get_weighted_text_embeddings(tok, FakeClipEmbedding(), FakeTextEncoder(),
                                                                    "a (very beautiful:1.3) cat"))
    assert ctx.shape == (1, 77, 8)
    import torch

    p = str(tmp_path / "ti.pt")
    torch.save({"string_to_param": {"*": torch.arange(16, dtype=torch.float32).reshape(2, 8)}}, p)
    assert sd.load_embedding(p).shape == (2, 8) and sd.load_embedding(str(tmp_path / "missing.pt")) is None
    assert sd.encode_text("a cat", p).shape == (1, 77, 8)


@pytest.mark.skipif(not os.path.exists("/root/reference/stable_diffusion/clip_tokenizer.py"), reason="reference not present")
def test_tokenizer_and_weighting_equal_reference_live():
    """Random strings / prompts through the reference's code (imported under the keras stub) and through ours."""
    import make_goldens as mg

    mg.install_keras_stub()
    sys.path.insert(0, mg.REF)
    import stable_diffusion.clip_tokenizer as ref_tok
    import stable_diffusion.long_prompt_weighting as ref_lpw

    from minsdtf_amd.text import SimpleTokenizer, get_weighted_text_embeddings, parse_prompt_attention

    a, b = ref_tok.SimpleTokenizer(BPE), SimpleTokenizer(BPE)
    rng = np.random.default_rng(5)
    words = ["photo", "graph", "astronaut", "riding", "horse", "moon", "the", "cat's", "Hat", "42", "café", "!!", "...", "(", ")", "[", "]",
             ":1.3)", "\\(", "&amp;", "beautiful", "masterpiece", ",", "\t", "  "]
    for _ in range(200):
        s = " ".join(rng.choice(words, size=rng.integers(1, 40)))
        assert a.encode(s) == b.encode(s), s
        assert ref_lpw.parse_prompt_attention(s) == parse_prompt_attention(s), s
    for _ in range(25):
        s = " ".join(rng.choice(words, size=rng.integers(1, 120)))
        kw = dict(no_boseos_middle=bool(rng.integers(0, 2)), pad_token_id=a.end_of_text)
        x = ref_lpw.get_weighted_text_embeddings(a, mg.FakeClipEmbedding(), mg.FakeTextEncoder(), s, **kw)
        y = get_weighted_text_embeddings(b, mg.FakeClipEmbedding(), mg.FakeTextEncoder(), s, **kw)
        np.testing.assert_array_equal(x, y)

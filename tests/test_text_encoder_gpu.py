"""SURVEY.md §8f rank 3 — the CLIP text encoder on the HIP path: embedding lookup, 12 causal
self-attention layers (d=64, bias on q/k/v/out), quick-GELU MLP, clip_skip, final LayerNorm; against the
oracle's restatement of text_encoder.py."""
import math

import numpy as np
import pytest
import torch

from conftest import run_calls

pytestmark = pytest.mark.gpu
PSNR_MIN = 40.0


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("B,S", [(1, 77), (2, 77), (1, 200)])
def test_causal_attention_d64(gpu, B, S):
    """msd_attention with head_dim 64 and the causal mask of text_encoder.py:75-78."""
    from minsdtf_amd import ops

    torch.manual_seed(11)
    H, d = 12, 64
    q, k, v = (bf(torch.randn(B, S, H * d)) for _ in range(3))
    qh, kh, vh = (t.view(B, S, H, d).permute(0, 2, 1, 3) for t in (q, k, v))
    mask = torch.triu(torch.full((S, S), float("-inf")), diagonal=1)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5 + mask, -1) @ vh).permute(0, 2, 1, 3).reshape(B, S, H * d)
    Sp = (S + 7) // 8 * 8
    vt = torch.full((B, H * d, Sp), float("nan"), dtype=torch.bfloat16, device=gpu)   # padding columns are unspecified
    vt[:, :, :S] = v.permute(0, 2, 1).to(torch.bfloat16).to(gpu)
    out = torch.full((B, S, H * d), float("nan"), dtype=torch.bfloat16, device=gpu)
    run_calls(ops.attention(q=q.to(torch.bfloat16).to(gpu), k=k.to(torch.bfloat16).to(gpu), vt=vt, out=out, batch=B, heads=H,
                            head_dim=d, s=S, t=S, q_ld=H * d, k_ld=H * d, vt_ld=Sp, o_ld=H * d, scale=d ** -0.5, causal=True))
    err = (out.float().cpu() - ref).abs().max()
    assert float(err) <= 2e-2 * float(ref.abs().max()) + 1e-3, float(err)
    # without the mask the same call is plain attention
    run_calls(ops.attention(q=q.to(torch.bfloat16).to(gpu), k=k.to(torch.bfloat16).to(gpu), vt=vt, out=out, batch=B, heads=H,
                            head_dim=d, s=S, t=S, q_ld=H * d, k_ld=H * d, vt_ld=Sp, o_ld=H * d, scale=d ** -0.5))
    ref2 = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh).permute(0, 2, 1, 3).reshape(B, S, H * d)
    assert float((out.float().cpu() - ref2).abs().max()) <= 2e-2 * float(ref2.abs().max()) + 1e-3


def test_quick_gelu_epilogue(gpu):
    from minsdtf_amd import ops, packing

    torch.manual_seed(12)
    M, C, N = 154, 128, 192
    x = bf(torch.randn(M, C))
    w = bf(torch.randn(C, N) / math.sqrt(C))
    b = torch.randn(N) * 0.3
    h = x @ w + b
    ref = h * torch.sigmoid(1.702 * h)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=gpu)
    run_calls(ops.conv_gemm(a0=x.to(torch.bfloat16).to(gpu), w=packing.pack_dense(w.numpy(), gpu), out=out, batch=2, h_in=77, w_in=1,
                            c0=C, N=N, bias=b.to(gpu), act=ops.ACT_QUICK_GELU))
    assert float((out.float().cpu() - ref).abs().max()) <= 1e-2 * float(ref.abs().max()) + 1e-2


def test_embedding_sum(gpu):
    from minsdtf_amd import ops

    rng = np.random.default_rng(13)
    tok_t = torch.from_numpy(rng.standard_normal((1000, 768)).astype(np.float32)).to(gpu)
    pos_t = torch.from_numpy(rng.standard_normal((77, 768)).astype(np.float32)).to(gpu)
    tokens = torch.from_numpy(rng.integers(0, 1000, (2, 77)).astype(np.int32)).to(gpu)
    positions = torch.arange(77, dtype=torch.int32, device=gpu).repeat(2, 1)
    out = torch.zeros(2 * 77, 768, dtype=torch.bfloat16, device=gpu)
    status = torch.zeros(1, dtype=torch.int32, device=gpu)
    run_calls(ops.embedding_sum(tokens=tokens, positions=positions, tok_table=tok_t, pos_table=pos_t, out=out, rows=154, dim=768,
                                vocab=1000, max_len=77, status=status))
    ref = (tok_t[tokens.long().view(-1)] + pos_t[positions.long().view(-1)]).to(torch.bfloat16)
    assert torch.equal(out, ref) and int(status.item()) == 0
    tokens[1, 5] = 1000   # out of range: flagged, not dereferenced
    run_calls(ops.embedding_sum(tokens=tokens, positions=positions, tok_table=tok_t, pos_table=pos_t, out=out, rows=154, dim=768,
                                vocab=1000, max_len=77, status=status))
    assert int(status.item()) == 1


@pytest.mark.parametrize("clip_skip,B", [(-1, 1), (-2, 2)])
def test_text_encoder_vs_oracle(gpu, clip_skip, B):
    from minsdtf_amd import weights as Wt
    from minsdtf_amd.models import TextClipEmbedding, TextEncoder
    from oracle import sd_oracle as O

    emb_m = TextClipEmbedding(device=gpu)
    We = O.named_weights(Wt.table("text_clip_embedding"), emb_m.load_synthetic(seed=0))
    enc = TextEncoder(clip_skip=clip_skip, device=gpu)
    assert len(enc.weights) == 16 * (12 + clip_skip + 1) + 2
    W = O.named_weights(Wt.table("text_encoder", clip_skip=clip_skip), enc.load_synthetic(seed=0, bias_scale=0.05))
    rng = np.random.default_rng(14)
    tokens = np.concatenate([np.full((B, 1), 49406), rng.integers(0, 49406, (B, 30)), np.full((B, 46), 49407)], axis=1).astype(np.int32)
    pos = np.arange(77, dtype=np.int32)[None]
    emb_ref = O.clip_embedding(We, tokens, pos)
    emb = emb_m.predict_on_batch([tokens, pos])
    assert emb.shape == (B, 77, 768) and O.psnr(emb, emb_ref) >= 45.0          # bf16 rounding of the sum
    # a stronger input than the 0.05-uniform synthetic table so that the attention rows are not flat
    x = (emb_ref * 20.0).astype(np.float32)
    ref = O.text_encoder_forward(W, x, clip_skip=clip_skip)
    got = enc.predict_on_batch(x)
    assert got.shape == ref.shape == (B, 77, 768)
    p = O.psnr(got, ref)
    print(f"text encoder clip_skip={clip_skip} B={B}: PSNR {p:.1f} dB")
    assert p >= PSNR_MIN
    # causality: changing a later token must not change earlier positions
    x2 = x.copy()
    x2[:, 40:] += 1.0
    got2 = enc.predict_on_batch(x2)
    np.testing.assert_array_equal(got2[:, :40], got[:, :40])
    with pytest.raises(ValueError):
        emb_m.predict_on_batch([np.full((1, 77), 60000, np.int32), pos])


def test_pipeline_unconditional_context_from_text_models(gpu):
    """stable_diffusion.py:488-493: without a supplied unconditional context the pipeline encodes the
    start + end tokens through the embedding and the text encoder."""
    from minsdtf_amd import weights as Wt
    from minsdtf_amd.models import TextClipEmbedding, TextEncoder
    from minsdtf_amd.stable_diffusion import StableDiffusion
    from oracle import sd_oracle as O

    sd = StableDiffusion(64, 64, device=gpu)
    sd._text_clip_embedding, sd._text_encoder = TextClipEmbedding(device=gpu), TextEncoder(clip_skip=sd.clip_skip, device=gpu)
    We = O.named_weights(Wt.table("text_clip_embedding"), sd._text_clip_embedding.load_synthetic(seed=0))
    W = O.named_weights(Wt.table("text_encoder", clip_skip=-1), sd._text_encoder.load_synthetic(seed=0, bias_scale=0.05))
    ids = np.asarray([[49406] + [49407] * 76], dtype=np.int32)
    ref = O.text_encoder_forward(W, O.clip_embedding(We, ids, np.arange(77)[None]), clip_skip=-1)
    got = sd._get_unconditional_context()
    assert got.shape == (1, 77, 768) and O.psnr(got, ref) >= PSNR_MIN
    ctx = sd.encode_text(np.concatenate([ids, ids], axis=0))      # two 77-token chunks -> (154, 768)
    assert ctx.shape == (154, 768)
    with pytest.raises(NotImplementedError):
        sd.encode_text("a string prompt")

"""The oracle checked against itself: the torch-based restatement (oracle/sd_oracle.py) vs the
independent plain-C restatement (oracle/c/ref_ops.c, double accumulation) of every primitive op,
on small seeded shapes.  Agreement to fp32 round-off pins the op semantics (NHWC / HWIO layouts,
group partition, biased variance, nearest upsampling, tanh-GELU, scale-after-QK^T) that the
reference leaves to Keras."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_build", "libref_ops.so")


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "c")])
    return C.CDLL(SO)


def fp(a):
    return a.ctypes.data_as(C.c_void_p)


def test_conv_groupnorm_upsample(ref):
    from oracle import sd_oracle as O

    rng = np.random.default_rng(0)
    for (ks, stride, pad) in ((3, 1, 1), (3, 2, 1), (1, 1, 0)):
        B, H, W, Ci, Co = 2, 6, 5, 8, 12
        x = rng.standard_normal((B, H, W, Ci)).astype(np.float32)
        w = rng.standard_normal((ks, ks, Ci, Co)).astype(np.float32)
        b = rng.standard_normal(Co).astype(np.float32)
        Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        y = np.zeros((B, Ho, Wo, Co), np.float32)
        ref.ref_conv2d_nhwc(fp(x), fp(w), fp(b), fp(y), B, H, W, Ci, Co, ks, stride, pad)
        Wd = {"c.weight": torch.from_numpy(w), "c.bias": torch.from_numpy(b)}
        t = O.padded_conv(torch.from_numpy(x).permute(0, 3, 1, 2), Wd, "c", stride=stride, pad=pad).permute(0, 2, 3, 1).numpy()
        np.testing.assert_allclose(t, y, rtol=1e-5, atol=1e-5)
    x = rng.standard_normal((2, 3, 4, 64)).astype(np.float32) * 3 + 1
    g, bt = rng.standard_normal(64).astype(np.float32), rng.standard_normal(64).astype(np.float32)
    y = np.zeros_like(x)
    ref.ref_group_norm(fp(x), fp(g), fp(bt), fp(y), 2, 12, 64, 32, C.c_float(1e-5))
    t = O.group_norm(torch.from_numpy(x).permute(0, 3, 1, 2), {"n.weight": torch.from_numpy(g), "n.bias": torch.from_numpy(bt)}, "n")
    np.testing.assert_allclose(t.permute(0, 2, 3, 1).numpy(), y, rtol=1e-4, atol=1e-5)
    y = np.zeros((2, 6, 8, 64), np.float32)
    ref.ref_upsample2_nhwc(fp(x), fp(y), 2, 3, 4, 64)
    np.testing.assert_array_equal(O.upsample2(torch.from_numpy(x).permute(0, 3, 1, 2)).permute(0, 2, 3, 1).numpy(), y)


def test_dense_layernorm_geglu_swish(ref):
    from oracle import sd_oracle as O

    rng = np.random.default_rng(1)
    x = rng.standard_normal((7, 40)).astype(np.float32)
    w, b = rng.standard_normal((40, 24)).astype(np.float32), rng.standard_normal(24).astype(np.float32)
    y = np.zeros((7, 24), np.float32)
    ref.ref_dense(fp(x), fp(w), fp(b), fp(y), 7, 40, 24)
    Wd = {"d.weight": torch.from_numpy(w), "d.bias": torch.from_numpy(b)}
    np.testing.assert_allclose(O.dense(torch.from_numpy(x), Wd, "d").numpy(), y, rtol=1e-5, atol=1e-5)
    g, bt = rng.standard_normal(40).astype(np.float32), rng.standard_normal(40).astype(np.float32)
    y = np.zeros_like(x)
    ref.ref_layer_norm(fp(x), fp(g), fp(bt), fp(y), 7, 40, C.c_float(1e-5))
    np.testing.assert_allclose(O.layer_norm(torch.from_numpy(x), {"n.weight": torch.from_numpy(g), "n.bias": torch.from_numpy(bt)}, "n").numpy(),
                               y, rtol=1e-4, atol=1e-5)
    # GEGLU: Dense(40 -> 2*12) then value * gelu_tanh(gate)
    wg, bg = rng.standard_normal((40, 24)).astype(np.float32), rng.standard_normal(24).astype(np.float32)
    h = np.zeros((7, 24), np.float32)
    ref.ref_dense(fp(x), fp(wg), fp(bg), fp(h), 7, 40, 24)
    y = np.zeros((7, 12), np.float32)
    ref.ref_geglu(fp(h), fp(y), 7, 12)
    t = O.geglu(torch.from_numpy(x), {"g.proj.weight": torch.from_numpy(wg), "g.proj.bias": torch.from_numpy(bg)}, "g").numpy()
    np.testing.assert_allclose(t, y, rtol=1e-4, atol=1e-5)
    y = np.zeros_like(x)
    ref.ref_swish(fp(x), fp(y), C.c_long(x.size))
    np.testing.assert_allclose(O.swish(torch.from_numpy(x)).numpy(), y, rtol=1e-5, atol=1e-6)


def test_attention(ref):
    from oracle import sd_oracle as O

    rng = np.random.default_rng(2)
    B, S, T, heads, d = 2, 9, 7, 8, 5
    Cm = heads * d
    x = rng.standard_normal((B, S, Cm)).astype(np.float32)
    ctx = rng.standard_normal((B, T, 16)).astype(np.float32)
    Wd = {"a.to_q.weight": torch.from_numpy(rng.standard_normal((Cm, Cm)).astype(np.float32) * 0.3),
          "a.to_k.weight": torch.from_numpy(rng.standard_normal((16, Cm)).astype(np.float32) * 0.3),
          "a.to_v.weight": torch.from_numpy(rng.standard_normal((16, Cm)).astype(np.float32) * 0.3),
          "a.to_out.0.weight": torch.eye(Cm), "a.to_out.0.bias": torch.zeros(Cm)}
    t = O.cross_attention(torch.from_numpy(x), torch.from_numpy(ctx), Wd, "a", heads=heads).numpy()
    q = (torch.from_numpy(x) @ Wd["a.to_q.weight"]).numpy().copy()
    k = (torch.from_numpy(ctx) @ Wd["a.to_k.weight"]).numpy().copy()
    v = (torch.from_numpy(ctx) @ Wd["a.to_v.weight"]).numpy().copy()
    o = np.zeros_like(q)
    ref.ref_attention(fp(q), fp(k), fp(v), fp(o), B, S, T, heads, d, C.c_float(d ** -0.5))
    np.testing.assert_allclose(t, o, rtol=1e-4, atol=1e-5)


def test_host_math(ref):
    from oracle import sd_oracle as O

    emb = np.zeros(320, np.float32)
    for t in (0, 40, 960):
        ref.ref_timestep_embedding(t, fp(emb), 320, C.c_float(10000.0))
        # freqs are float32: one ulp of a frequency times t=960 moves the argument by ~6e-5, so the two
        # float32 evaluation orders agree to ~1e-4 absolute (the bit-exact pin is golden G3)
        np.testing.assert_allclose(O.timestep_embedding(t, 1)[0], emb, rtol=0, atol=1.5e-4)
    rng = np.random.default_rng(3)
    u, c = rng.standard_normal((2, 256)).astype(np.float32), rng.standard_normal((2, 256)).astype(np.float32)
    out = np.zeros_like(u)
    ref.ref_cfg_rescale(fp(u), fp(c), fp(out), 2, 256, C.c_float(7.5), C.c_float(0.7))
    e = O.rescale_noise_cfg(u + 7.5 * (c - u), c, 0.7)
    np.testing.assert_allclose(e, out, rtol=2e-5, atol=2e-5)
    s = O.OracleScheduler()
    s.set_timesteps(5)
    x = rng.standard_normal(256)
    for i, t in enumerate(s.timesteps):
        eps = rng.standard_normal(256).astype(np.float32)
        tp = s.timesteps[i + 1] if i + 1 < 5 else t
        o = np.zeros(256)
        ref.ref_sched_step(fp(x), fp(eps), fp(o), C.c_long(256), C.c_double(s.signal_rates[t]), C.c_double(s.noise_rates[t]),
                           C.c_double(s.signal_rates[tp]), C.c_double(s.noise_rates[tp]), int(i == 4))
        x = s.step(eps, int(t), x)
        np.testing.assert_allclose(x, o, rtol=1e-12, atol=1e-12)

"""World-size-2 test of the batch-sharding path on CPU (gloo): inputs broadcast from rank 0, each
rank runs its contiguous slice, images all-gathered in batch order — and the result equals the
single-process run on the global batch (the property the multi-GPU bench relies on)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_generate(ctx, unc, noise):
    """Per-sample deterministic stand-in for the local denoise+decode (no cross-sample coupling).  Inputs are host arrays
    (single process) or the device-resident views of the packed broadcast (world > 1; CPU tensors under gloo)."""
    ctx, unc, noise = (np.asarray(a) for a in (ctx, unc, noise))
    v = noise.reshape(noise.shape[0], -1)[:, :48] * 20 + ctx.mean(axis=(1, 2))[:, None] * 100 + unc.std(axis=(1, 2))[:, None] * 10
    img = np.clip(v + 128, 0, 255).astype(np.uint8).reshape(noise.shape[0], 4, 4, 3)
    return torch.from_numpy(img)


def _inputs(gb):
    rng = np.random.default_rng(5)
    return (rng.standard_normal((gb, 7, 768)).astype(np.float32), rng.standard_normal((gb, 7, 768)).astype(np.float32),
            rng.standard_normal((gb, 8, 8, 4)).astype(np.float32))


def _worker(rank, world, port, gb, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from minsdtf_amd import dist as mdist

    r, w = mdist.init("gloo")
    assert (r, w) == (rank, world)
    ctx, unc, noise = _inputs(gb)
    if rank != 0:  # only rank 0 holds the real inputs; the others pass right-shaped garbage
        ctx, unc, noise = np.zeros_like(ctx), np.ones_like(unc), np.full_like(noise, 7.0)
    seen = {}

    def local(c, u, z):
        seen["n"] = z.shape[0]
        return _fake_generate(c, u, z)

    img = mdist.generate_sharded(local, ctx, unc, noise, torch.device("cpu"))
    assert seen["n"] == gb // world
    np.save(os.path.join(out_dir, f"img_{rank}.npy"), img.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_generate_sharded_world2(tmp_path):
    gb, world = 6, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, gb, str(tmp_path)), nprocs=world, join=True)
    ctx, unc, noise = _inputs(gb)
    ref = _fake_generate(ctx, unc, noise).numpy()
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"img_{r}.npy"))
        assert got.shape == (gb, 4, 4, 3)
        np.testing.assert_array_equal(got, ref)  # every rank ends with the full batch, in batch order


def _bench_worker(rank, world, port, gb, out_dir):
    """bench.py's own job + timing code (sharded_job / timed_jobs: packed broadcast, all-gather, rank-0 host copy, barriers,
    all_reduce(MAX) of the elapsed time) on 2 gloo ranks with a stub local generator in place of the GPU pipeline."""
    import sys

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from minsdtf_amd import dist as mdist

    mdist.init("gloo")
    dev = torch.device("cpu")
    ctx, unc, noise = _inputs(gb)
    if rank != 0:
        ctx, unc, noise = np.zeros_like(ctx), np.ones_like(unc), np.full_like(noise, 7.0)
    calls = {"n": 0}

    def local(c, u, z):
        calls["n"] += 1
        assert isinstance(z, torch.Tensor) and z.shape[0] == gb // world   # device-resident slice of the ONE broadcast buffer
        return _fake_generate(c, u, z)

    elapsed, img = bench.timed_jobs(lambda: bench.sharded_job(local, ctx, unc, noise, dev), steps=3, warmup=1, dev=dev)
    assert calls["n"] == 4 and elapsed > 0
    np.save(os.path.join(out_dir, f"bench_{rank}.npy"), np.asarray([elapsed]))
    np.save(os.path.join(out_dir, f"bimg_{rank}.npy"), img.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bench_job_and_timing_world2(tmp_path):
    gb, world = 4, 2
    port = _free_port()
    mp.spawn(_bench_worker, args=(world, port, gb, str(tmp_path)), nprocs=world, join=True)
    ctx, unc, noise = _inputs(gb)
    ref = _fake_generate(ctx, unc, noise).numpy()
    t = [float(np.load(os.path.join(str(tmp_path), f"bench_{r}.npy"))[0]) for r in range(world)]
    assert t[0] == t[1]                       # MAX over ranks, the same number on every rank
    for r in range(world):
        np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), f"bimg_{r}.npy")), ref)


def test_algorithmic_flop_table():
    """bench.py prices an image with the reference graph's algorithmic FLOP, exact per size."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    assert abs(bench.algorithmic_tflop_per_image(512, 25) - 42.68) < 0.01
    assert abs(bench.algorithmic_tflop_per_image(768, 50) - 220.57) < 0.5     # SURVEY: 100 x 2.148 + 5.754
    assert abs(bench.algorithmic_tflop_per_image(512, 25, controlnet=True) - 56.12) < 0.02


def test_shard_bounds():
    from minsdtf_amd.dist import shard_bounds

    assert [shard_bounds(32, r, 8) for r in range(8)] == [(4 * r, 4 * r + 4) for r in range(8)]
    with pytest.raises(ValueError):
        shard_bounds(6, 0, 4)


def test_single_process_passthrough():
    from minsdtf_amd import dist as mdist

    ctx, unc, noise = _inputs(2)
    img = mdist.generate_sharded(_fake_generate, ctx, unc, noise, torch.device("cpu"))
    np.testing.assert_array_equal(img.numpy(), _fake_generate(ctx, unc, noise).numpy())

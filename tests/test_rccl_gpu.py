"""The RCCL path on the one GPU a test box has (reference batch tiling: stable_diffusion.py:384-397).

No 8-GPU node is available to the builder, and at world size 1 the sharding helpers normally short-circuit — so without
this test librccl would never even be loaded before a driver SCALE run.  A fresh child process creates a ONE-rank process
group with backend `nccl` (= RCCL on ROCm) before anything else touches the GPU and runs the real collectives: the packed
device-resident fp32 broadcast, all_gather_into_tensor of uint8 [b, 512, 512, 3], barrier, destroy, and the pipeline itself
with its exchanges forced through them (bit-identical images)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rccl_world1_collectives_and_pipeline(gpu):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, os.path.join(HERE, "_collectives_world1_child.py"), "nccl", "--pipeline"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=570)
    assert p.returncode == 0, f"child failed ({p.returncode}):\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    ok = [ln for ln in p.stdout.splitlines() if ln.startswith("OK ")]
    assert ok, p.stdout[-2000:]
    info = json.loads(ok[-1][3:])
    print("RCCL world-1:", info)
    assert info["backend"] == "nccl" and info["world"] == 1 and info["device"] == "cuda:0" and info["pipeline"] == "bit-identical"


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_through_rccl_with_shared_weights(gpu, tmp_path):
    """First-contact rehearsal of the driver's SCALE command.  On a box with >= 2 GPUs: `python bench.py --gpus 2` - the
    launcher starts two fresh ranks before anything touches a GPU, rank 1 maps the weights rank 0 packed (/dev/shm), both run
    the real pipeline over RCCL, the line carries two devices, two first-job times and rank 0's image hash.  On a one-GPU box
    (this pool): the same bench through a ONE-rank RCCL group with its exchanges forced, so the RCCL code path of bench.py
    itself (packed broadcast, all-gather, the timing collectives) still runs here."""
    import torch

    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    n = 2 if torch.cuda.device_count() >= 2 else 1
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "1", "--size", "64",
           "--denoise-steps", "3", "--no-cpu-baseline", "--no-roofline"] + (["--force-collectives"] if n == 1 else [])
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=850)
    assert p.returncode == 0, f"bench failed ({p.returncode}):\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    lines = [ln for ln in p.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["n_ranks_seen"] == n and out["backend"].startswith("nccl") and out["value"] > 0
    assert len(out["per_rank_ms"]) == n and len(out["start_to_first_job_s"]) == n and all(t and t > 0 for t in out["start_to_first_job_s"])
    assert out["config"]["global_batch"] == n
    if n == 2:
        assert len({d["hip_device"] for d in out["rank_devices"]}) == 2
        assert "mapped from rank 0" in p.stderr   # rank 1 did not generate its own weights
    else:
        assert out.get("collectives_forced") is True


@pytest.mark.gpu
def test_packed_weights_file_gives_the_same_bits(gpu, tmp_path):
    """What ranks > 0 of an N-GPU bench do (bench.load_synthetic_shared): adopt the packed weights another process wrote to a
    RAM-backed file instead of generating and packing their own - on the real networks and the real device here: the mapped
    UNet and VAE decoder give the bits of the models that generated theirs, through plans with every kernel form (the
    fragment-major copies the wreg form reads are not in the file: each plan makes its own)."""
    import numpy as np

    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    from minsdtf_amd.models import DiffusionModel, ImageDecoder

    rng = np.random.default_rng(5)
    x = [rng.standard_normal((2, 8, 8, 4)).astype(np.float32), rng.standard_normal((2, 320)).astype(np.float32),
         rng.standard_normal((2, 77, 768)).astype(np.float32)]
    lat = (rng.standard_normal((1, 8, 8, 4)) * 0.5).astype(np.float32)
    sdir = str(tmp_path)
    for make, inp in ((lambda: DiffusionModel(64, 64, device=gpu), x), (lambda: ImageDecoder(device=gpu), lat)):
        r0, r1 = make(), make()
        assert bench.load_synthetic_shared(r0, 0, sdir, seed=0, bias_scale=0.05) is not None      # rank 0: generates, packs, writes
        assert bench.load_synthetic_shared(r1, 1, sdir, seed=0, bias_scale=0.05) is None           # rank 1: maps the file
        a, b = r0.predict_on_batch(inp), r1.predict_on_batch(inp)
        assert np.isfinite(a).all()
        np.testing.assert_array_equal(a, b)
        with pytest.raises(ValueError):
            bench.load_synthetic_shared(make(), 1, sdir, seed=1, bias_scale=0.05)                 # another checkpoint's file is refused

"""Child process of tests/test_rccl_gpu.py (backend nccl = RCCL, on cuda:0) and tests/test_dist_cpu.py (backend gloo):
a ONE-rank torch.distributed process group whose exchanges really run (minsdtf_amd.dist.FORCE_COLLECTIVES), i.e. the
multi-GPU code path of SURVEY.md §8e — packed device-resident fp32 broadcast, all_gather_into_tensor of uint8 images,
barrier, destroy — executed on the one device a test box has.  A fresh interpreter: the process group is created before
anything else touches the GPU.  Prints one line `OK {...json...}`; any failure is a non-zero exit.

    python tests/_collectives_world1_child.py nccl|gloo [--pipeline]
"""
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    backend = sys.argv[1]
    pipeline = "--pipeline" in sys.argv[2:]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch
    import torch.distributed as dist

    from minsdtf_amd import dist as mdist

    mdist.FORCE_COLLECTIVES = True
    r, w = mdist.init(backend, force=True)
    assert (r, w) == (0, 1) and dist.is_initialized() and dist.get_backend() == backend, (r, w, dist.get_backend())
    assert mdist.collectives_on()
    dev = torch.device("cuda", 0) if backend == "nccl" else torch.device("cpu")
    info = {"backend": dist.get_backend(), "world": dist.get_world_size(), "device": str(dev)}

    # 1. the packed broadcast: host arrays (rank 0 of a real run) and device-resident tensors (bench.py at N = 1)
    rng = np.random.default_rng(11)
    arrays = [rng.standard_normal((2, 77, 768)).astype(np.float32), rng.standard_normal((2, 77, 768)).astype(np.float32),
              rng.standard_normal((2, 64, 64, 4)).astype(np.float32)]
    for resident in (False, True):
        src = [torch.from_numpy(a).to(dev) for a in arrays] if resident else arrays
        got = mdist.broadcast_inputs(src, dev)
        assert all(isinstance(g, torch.Tensor) and g.device.type == dev.type and g.dtype == torch.float32 for g in got)
        base = got[0].untyped_storage().data_ptr()
        assert all(g.untyped_storage().data_ptr() == base for g in got), "the inputs must travel as ONE packed buffer"
        for g, a in zip(got, arrays):
            assert np.array_equal(g.cpu().numpy(), a)

    # 2. the all-gather of finished uint8 images [b, 512, 512, 3] (and of fp32 latents: return_latent)
    img = torch.from_numpy(rng.integers(0, 256, (2, 512, 512, 3), dtype=np.uint8)).to(dev)
    out = mdist.all_gather_images(img)
    assert out.dtype == torch.uint8 and tuple(out.shape) == (2, 512, 512, 3) and out.data_ptr() != img.data_ptr()
    assert torch.equal(out, img)
    lat = torch.randn(2, 64, 64, 4, device=dev)
    assert torch.equal(mdist.all_gather_images(lat), lat)

    # 3. generate_sharded end to end through both collectives
    def local(c, u, z):
        assert all(isinstance(t, torch.Tensor) and t.device.type == dev.type for t in (c, u, z))
        return (z.reshape(z.shape[0], -1)[:, :48] * 20 + c.mean(dim=(1, 2))[:, None] * 100 + 128).clamp(0, 255).to(torch.uint8)

    a = mdist.generate_sharded(local, *arrays, dev)
    mdist.FORCE_COLLECTIVES = False
    b = mdist.generate_sharded(lambda c, u, z: local(*(torch.as_tensor(t).to(dev) for t in (c, u, z))), *arrays, dev)
    mdist.FORCE_COLLECTIVES = True
    assert torch.equal(a, b)

    if pipeline:
        # 4. the real pipeline (64x64 image, 3 steps, hipGraph loop) with its exchanges on RCCL: the same bits as without
        from minsdtf_amd.stable_diffusion import StableDiffusion

        sd = StableDiffusion(64, 64, jit_compile=True, device=dev)
        sd.shard_batch = True   # (opt in: the default keeps the reference's replica semantics)
        sd.diffusion_model.load_synthetic(seed=0)
        sd.image_decoder.load_synthetic(seed=0)
        ctx = rng.standard_normal((77, 768)).astype(np.float32)
        sd.unconditional_context = rng.standard_normal((77, 768)).astype(np.float32)
        noise = rng.standard_normal((2, 8, 8, 4)).astype(np.float32)
        kw = dict(batch_size=2, num_steps=3, unconditional_guidance_scale=7.5, diffusion_noise=noise, guidance_rescale=0.7)
        # count the exchanges: the forced run must really broadcast its packed inputs and all-gather its images (a comparison of
        # the short-cut path with itself would pass as well)
        calls = {"broadcast": 0, "all_gather_into_tensor": 0}
        real = {k: getattr(dist, k) for k in calls}

        def counted(name):
            def f(*a, **k):
                calls[name] += 1
                return real[name](*a, **k)
            return f

        for k in calls:
            setattr(dist, k, counted(k))
        t0 = time.perf_counter()
        forced = sd.generate_image(ctx, **kw)
        info["pipeline_s"] = round(time.perf_counter() - t0, 2)
        assert calls["broadcast"] >= 1 and calls["all_gather_into_tensor"] >= 1, f"generate_image took the single-process short cut: {calls}"
        info["pipeline_collectives"] = dict(calls)
        mdist.FORCE_COLLECTIVES = False
        n0 = dict(calls)
        plain = sd.generate_image(ctx, **kw)
        assert calls == n0, "the un-forced run must not touch the process group"
        for k in calls:
            setattr(dist, k, real[k])
        mdist.FORCE_COLLECTIVES = True
        assert forced.shape == (2, 64, 64, 3) and forced.dtype == np.uint8
        assert np.array_equal(forced, plain), "the collectives changed the images"
        info["pipeline"] = "bit-identical"

    dist.barrier()
    dist.destroy_process_group()
    print("OK " + json.dumps(info), flush=True)


if __name__ == "__main__":
    main()

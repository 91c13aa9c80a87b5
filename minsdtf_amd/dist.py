"""Batch sharding of the denoise loop over the GPUs of one node.

The path shards by independent samples: no op couples samples (GroupNorm / LayerNorm / attention
are per sample, the CFG-rescale std is per sample, reference ``stable_diffusion.py:309-310``), so
rank r owns the contiguous slice ``[r*B/G, (r+1)*B/G)`` of the global batch and NOTHING crosses
GPUs inside a step.  The only exchanges are the inputs and the outputs:

* ONE broadcast from rank 0 of the text contexts and of the initial noise drawn for the GLOBAL
  batch (so results do not depend on the number of ranks), packed into a single device buffer,
* one all-gather of the finished uint8 images.

Both are a few MB at most (latency-bound), so they are plain RCCL collectives
(``torch.distributed`` backend ``nccl`` on ROCm; ``gloo`` in the CPU tests).
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


# True: a process group of ONE rank still goes through the real collectives (packed device broadcast, all_gather_into_tensor)
# instead of the single-process short cuts.  That is how the RCCL path is exercised on a one-GPU box (bench.py
# --force-collectives, tests/test_rccl_gpu.py): library loading, dtype support and stream ordering against the captured
# graph's stream are the same code at world 1 as at world 8.  $MSD_FORCE_COLLECTIVES=1 sets it at import.
FORCE_COLLECTIVES = os.environ.get("MSD_FORCE_COLLECTIVES", "0") == "1"


def env_rank() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend: Optional[str] = None, force: bool = False) -> Tuple[int, int]:
    """Initialise the default process group from the torchrun environment (no-op for 1 rank unless `force` /
    FORCE_COLLECTIVES: then a one-rank group is created and the collectives below really run)."""
    rank, local_rank, world = env_rank()
    if (world > 1 or force or FORCE_COLLECTIVES) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def shard_bounds(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split; the global batch must divide evenly (weak scaling: fixed work per GPU)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by the {world} ranks of the initialised torch.distributed "
                         "process group: with `shard_batch = True` `batch_size` is the GLOBAL batch, sharded over the ranks; "
                         "for independent replicas per rank leave `shard_batch` at its default (False)")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def world_size() -> int:
    """Ranks of the initialised default process group (1 when torch.distributed is not in use)."""
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def collectives_on() -> bool:
    """Whether the exchanges below go through torch.distributed: more than one rank, or one rank with FORCE_COLLECTIVES."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def broadcast_inputs(arrays, device, src: int = 0):
    """ONE broadcast from `src` of a list of fp32 arrays (same shapes on every rank; contents only matter on `src`),
    packed into a single flat device buffer; returns device-resident views of it, one per array — nothing comes back to
    the host (the denoise engine copies device -> device).  Single process: host inputs come back as fp32 numpy arrays,
    tensors as they are (so a caller sees the same kinds of object at world = 1 and world > 1: something with
    ``.shape`` that slices along the batch)."""
    if not collectives_on():
        return [a if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float32) for a in arrays]
    shapes = [tuple(np.shape(a)) for a in arrays]
    sizes = [int(np.prod(sh)) for sh in shapes]
    if dist.get_rank() == src:
        if all(isinstance(a, torch.Tensor) and a.device == torch.device(device) for a in arrays):
            buf = torch.cat([a.detach().reshape(-1).to(torch.float32) for a in arrays])   # already resident: packed on the device
        else:
            flat = np.concatenate([_host_f32(a).reshape(-1) for a in arrays])
            buf = torch.from_numpy(flat).to(device)
    else:
        buf = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    dist.broadcast(buf, src=src)
    out, off = [], 0
    for sh, n in zip(shapes, sizes):
        out.append(buf[off:off + n].view(*sh))
        off += n
    return out


def _host_f32(a) -> np.ndarray:
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def all_gather_images(local: torch.Tensor) -> torch.Tensor:
    """[b, ...] per rank (uint8 images, or fp32 latents) -> [world*b, ...] on every rank, rank order = batch order."""
    if not collectives_on():
        return local
    world = dist.get_world_size()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def generate_sharded(generate_local: Callable[..., torch.Tensor], context, uncond_context, noise, device,
                     per_sample: Sequence = (), shared: Sequence = (), shard: bool = True) -> torch.Tensor:
    """Run ``generate_local(ctx_slice, uncond_slice, noise_slice, *per_sample_slices, *shared) -> tensor [b, ...]`` on this
    rank's slice of the global batch and gather the results of all ranks in batch order.

    context / uncond_context: (B, T, 768); noise: (B, h, w, 4) — for the GLOBAL batch, valid on rank 0, right shape
    elsewhere (host arrays or tensors).  ``per_sample``: further arrays with the global batch as leading dimension (the
    ControlNet hint images, reference stable_diffusion.py:427-441; the inpaint noise) — sliced like the noise.  ``shared``:
    arrays every rank needs whole (an encoded reference image).  ALL of them travel in the ONE packed broadcast.
    `generate_local` receives device tensors when world > 1 (views of the broadcast buffer) and fp32 host arrays /
    the caller's tensors when world = 1.  ``shard=False``: no exchange at all, this rank runs the whole batch it was given
    (independent replicas under a process group that exists for other reasons)."""
    r, world = (rank(), world_size()) if shard else (0, 1)
    n_ps = len(per_sample)
    arrays = [context, uncond_context, noise, *per_sample, *shared]
    if world == 1 and not (shard and collectives_on()):
        got = [a if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float32) for a in arrays]
        return generate_local(*got)
    got = broadcast_inputs(arrays, device)
    lo, hi = shard_bounds(int(got[2].shape[0]), r, world)
    sliced = [a[lo:hi] for a in got[:3 + n_ps]]
    img = generate_local(*sliced, *got[3 + n_ps:])
    return all_gather_images(img)

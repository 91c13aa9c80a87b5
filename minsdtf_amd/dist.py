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
from typing import Callable, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend: Optional[str] = None) -> Tuple[int, int]:
    """Initialise the default process group from the torchrun environment (no-op for 1 rank)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
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
        raise ValueError(f"global batch {global_batch} is not divisible by {world} ranks")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def broadcast_inputs(arrays, device, src: int = 0):
    """ONE broadcast from `src` of a list of fp32 arrays (same shapes on every rank; contents only matter on `src`),
    packed into a single flat device buffer; returns device-resident views of it, one per array — nothing comes back to
    the host (the denoise engine copies device -> device).  Single process: the arrays are returned as they are."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(arrays)
    shapes = [tuple(np.shape(a)) for a in arrays]
    sizes = [int(np.prod(sh)) for sh in shapes]
    if dist.get_rank() == src:
        flat = np.concatenate([np.ascontiguousarray(a, dtype=np.float32).reshape(-1) for a in arrays])
        buf = torch.from_numpy(flat).to(device)
    else:
        buf = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    dist.broadcast(buf, src=src)
    out, off = [], 0
    for sh, n in zip(shapes, sizes):
        out.append(buf[off:off + n].view(*sh))
        off += n
    return out


def all_gather_images(local: torch.Tensor) -> torch.Tensor:
    """uint8 [b, H, W, 3] per rank -> [world*b, H, W, 3] on every rank, rank order = batch order."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def generate_sharded(generate_local: Callable[[np.ndarray, np.ndarray, np.ndarray], torch.Tensor], context: np.ndarray,
                     uncond_context: np.ndarray, noise: np.ndarray, device) -> torch.Tensor:
    """Run `generate_local(ctx_slice, uncond_slice, noise_slice) -> uint8 tensor [b,H,W,3]` on this
    rank's slice of the global batch and gather all images.

    context / uncond_context: (B, T, 768); noise: (B, h, w, 4) for the GLOBAL batch (valid on
    rank 0, right shape elsewhere)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    context, uncond_context, noise = broadcast_inputs([context, uncond_context, noise], device)   # device tensors when world > 1
    lo, hi = shard_bounds(noise.shape[0], rank, world)
    img = generate_local(context[lo:hi], uncond_context[lo:hi], noise[lo:hi])
    return all_gather_images(img)

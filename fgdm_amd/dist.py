"""Multi-GPU layer: one process per GPU, prompt-batch sharding, ONE collective.

The denoising path has no cross-sample dependence (GroupNorm is per sample; no BatchNorm anywhere), so ranks
never talk during sampling.  The only collective is the broadcast of the frozen weights from rank 0
at start-up -- over RCCL/xGMI on GPUs ("nccl" backend), over gloo in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_global, rank, world):
    """Contiguous block of the global prompt batch owned by `rank` (CFG pairs stay together)."""
    if n_global % world:
        raise ValueError(f'global batch {n_global} not divisible by {world} ranks')
    per = n_global // world
    return rank * per, (rank + 1) * per


def shard(t, rank, world):
    lo, hi = shard_bounds(t.shape[0], rank, world)
    return t[lo:hi]


def broadcast_weights(shapes, make_tensor, rank, world, device):
    """Rank 0 materialises every parameter (make_tensor(key, shape) -> float32 ndarray) into one flat buffer;
    a single broadcast ships it; every rank returns {key: view}.  With world == 1 nothing is communicated."""
    total = sum(int(np.prod(s)) for s in shapes.values())
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if rank == 0:
        off = 0
        for k, s in shapes.items():
            n = int(np.prod(s))
            flat[off:off + n].copy_(torch.from_numpy(np.ascontiguousarray(make_tensor(k, s)).ravel()))
            off += n
    if world > 1:
        dist.broadcast(flat, src=0)
    out, off = {}, 0
    for k, s in shapes.items():
        n = int(np.prod(s))
        out[k] = flat[off:off + n].view(*s)
        off += n
    return out, flat


def gather_latents(x_local, rank, world):
    """Optional: collect the per-rank result latents [N/R,4,H,W] on every rank (64 KB per sample)."""
    if world == 1:
        return x_local
    parts = [torch.empty_like(x_local) for _ in range(world)]
    dist.all_gather(parts, x_local.contiguous())
    return torch.cat(parts)

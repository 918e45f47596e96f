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


# Weights the engine transforms in fp32 BEFORE it rounds them to fp16 (fgdm_finalize_weights): a tensor shipped as fp16 would be
# rounded twice there.  to_q: log2(e) d^-1/2 is folded in; the consumers of a folded LayerNorm (norm1 -> attn1.to_q|to_k|to_v,
# norm2 -> attn2.to_q, norm3 -> ff.net.0.proj) are packed as fp16(gamma_k W_nk) with bias / u sums taken over the fp32 W; the text
# encoder keeps its token / position embeddings in fp32 (fp32 residual stream), the first stage its 4x4 post_quant_conv.
_FP32_SUFFIXES = ('attn1.to_q.weight', 'attn1.to_k.weight', 'attn1.to_v.weight', 'attn2.to_q.weight', 'ff.net.0.proj.weight',
                  'token_embedding.weight', 'position_embedding.weight', 'post_quant_conv.weight')


def _ships_as_fp16(key, shape):
    """Tensors the engine stores as fp16 UNCHANGED may travel as fp16: every >= 2-D weight except the ones listed in
    _FP32_SUFFIXES.  1-D tensors (biases, norm affine) stay fp32 in the engine and travel as fp32.  With this rule a rank fed
    from the broadcast holds bit-identical packed weights to one that loaded the fp32 state dict directly
    (tests/test_gpu_dropin.py::test_broadcast_layout_is_bit_identical_to_fp32_load)."""
    return len(shape) >= 2 and not key.endswith(_FP32_SUFFIXES)


def broadcast_weights(shapes, make_tensor, rank, world, device, timing=None, force=False):
    """Rank 0 materialises every parameter (make_tensor(key, shape) -> float32 ndarray) into ONE flat byte buffer -- fp16
    where the engine keeps the tensor as fp16 unchanged (3.2 GB for SD-v1.5 + one ControlNet instead of 4.9 GB of fp32), fp32 otherwise -- a single
    broadcast ships it (RCCL over xGMI on GPUs, gloo in the CPU tests) and every rank returns {key: view} plus the
    buffer.  With world == 1 nothing is communicated.  timing: optional dict, receives 'bcast_s' and 'bcast_bytes'."""
    import time
    layout, off = {}, 0
    for k, s in shapes.items():
        n = int(np.prod(s))
        dt = torch.float16 if _ships_as_fp16(k, s) else torch.float32
        nbytes = n * (2 if dt == torch.float16 else 4)
        layout[k] = (off, nbytes, dt)
        off += (nbytes + 15) & ~15            # 16-byte aligned views
    flat = torch.empty(off, dtype=torch.uint8, device=device)
    if rank == 0:
        # the tensors are independent (one counter-based stream per key: fgdm_amd/synth.py; a checkpoint loader would read them
        # from disk): materialise them on a few host threads -- the other ranks wait for this broadcast and nothing else
        import concurrent.futures as cf
        import os

        def fill(k):
            o, nbytes, dt = layout[k]
            src = torch.from_numpy(np.ascontiguousarray(make_tensor(k, shapes[k]), dtype=np.float32).ravel()).to(dt)
            flat[o:o + nbytes].view(dt).copy_(src)
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        with cf.ThreadPoolExecutor(max_workers=max(1, min(cores, 16))) as pool:
            list(pool.map(fill, list(shapes)))
    if world > 1 or force:      # force: issue the collective even for one rank (rehearsal of the RCCL call)
        if flat.is_cuda:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.broadcast(flat, src=0)
        if flat.is_cuda:
            torch.cuda.synchronize()
        if timing is not None:
            timing['bcast_s'] = time.perf_counter() - t0
    if timing is not None:
        timing['bcast_bytes'] = int(off)
    out = {k: flat[o:o + nbytes].view(dt).view(*shapes[k]) for k, (o, nbytes, dt) in layout.items()}
    return out, flat


def gather_latents(x_local, rank, world):
    """Optional: collect the per-rank result latents [N/R,4,H,W] on every rank (64 KB per sample)."""
    if world == 1:
        return x_local
    parts = [torch.empty_like(x_local) for _ in range(world)]
    dist.all_gather(parts, x_local.contiguous())
    return torch.cat(parts)

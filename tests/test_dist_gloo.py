"""N>1 path on CPU: two gloo ranks.  Weight broadcast delivers rank 0's bytes; a prompt batch sharded over the
ranks gives exactly the unsharded result (no cross-rank dependence during denoising)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    import golden_inputs as gi
    import kernel_stubs
    from fgdm_amd import dist as fd, models, samplers, synth
    from test_samplers_host import AnalyticLDM
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        samplers._k = kernel_stubs
        models._k = kernel_stubs
        shapes = {'a.weight': (8, 4, 3, 3), 'a.bias': (8,), 'b.weight': (16, 8)}
        # only rank 0 can produce the weights; other ranks would produce garbage if asked
        make = (lambda k, s: synth.make_tensor(k, s)) if rank == 0 else (lambda k, s: np.full(s, np.nan, np.float32))
        sd, _ = fd.broadcast_weights(shapes, make, rank, world, 'cpu')
        # >= 2-D weights travel as fp16 (what the engine stores), 1-D tensors as fp32: rank 0's bytes, exactly
        ok_w = all(torch.equal(sd[k], torch.from_numpy(synth.make_tensor(k, s)).to(sd[k].dtype)) for k, s in shapes.items())
        ok_w = ok_w and sd['a.weight'].dtype == torch.float16 and sd['a.bias'].dtype == torch.float32
        # global batch generated identically everywhere, then sliced
        N = 4
        x_T = torch.from_numpy(synth.latents(N, 8, 8, seed=42))
        c, uc = torch.from_numpy(synth.context(N, seed=43)), torch.from_numpy(synth.context(N, seed=44))
        run = lambda xs, cs, us: samplers.DDIMSampler(AnalyticLDM()).sample(
            6, xs.shape[0], (4, 8, 8), conditioning=cs, x_T=xs, verbose=False, unconditional_guidance_scale=7.5,
            unconditional_conditioning=us)[0]
        full = run(x_T, c, uc)
        mine = run(fd.shard(x_T, rank, world), fd.shard(c, rank, world), fd.shard(uc, rank, world))
        allx = fd.gather_latents(mine, rank, world)
        q.put((rank, ok_w, bool(torch.equal(allx, full)), fd.shard_bounds(N, rank, world)))
    finally:
        dist.destroy_process_group()


def test_two_rank_broadcast_and_shard_invariance():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] for r in res), 'broadcast weights differ from rank 0'
    assert all(r[2] for r in res), 'sharded sampling != unsharded'
    assert [r[3] for r in res] == [(0, 2), (2, 4)]


def test_shard_bounds_rejects_ragged():
    from fgdm_amd import dist as fd
    with pytest.raises(ValueError):
        fd.shard_bounds(10, 0, 4)


def test_shipping_rule_keeps_fp32_for_tensors_the_packer_transforms():
    """fp16 shipping must not add a rounding in front of a fold (LayerNorm gamma, attention scale)."""
    from fgdm_amd import dist as fd
    two_d = (320, 320)
    for k in ('x.attn1.to_q.weight', 'x.attn1.to_k.weight', 'x.attn1.to_v.weight', 'x.attn2.to_q.weight', 'x.ff.net.0.proj.weight',
              'c.embeddings.token_embedding.weight', 'c.embeddings.position_embedding.weight', 'first_stage_model.post_quant_conv.weight'):
        assert not fd._ships_as_fp16(k, two_d), k
    for k in ('x.attn2.to_k.weight', 'x.attn2.to_v.weight', 'x.attn1.to_out.0.weight', 'x.ff.net.2.weight', 'x.in_layers.2.weight', 'x.proj_in.weight'):
        assert fd._ships_as_fp16(k, two_d), k
    assert not fd._ships_as_fp16('x.in_layers.2.bias', (320,))

"""Block-level parity through the C ABI (fgdm_run_block): one block of the engine's own graph at a time against the golden
vectors of the reference's block modules (tests/golden/ops.npz: ResBlock x4, Downsample, Upsample, SpatialTransformer at
head dims 40 / 80 / 160, Adapter; tools/make_goldens.py g_ops).

Three comparisons per block, all normwise relative:
  e_engine   vs the CPU oracle in 'engine' precision (oracle/precision.py: the same fp16 storage policy, so the difference
             is fp32 summation order plus what a few roundings amplify) -- held to the north-star 1e-3;
  e_fp32     vs the reference's fp32 golden, next to floor = |reference under its CUDA-autocast policy - reference fp32|
             (ops_ac.npz): the engine must sit no farther from the PyTorch-CPU path than the reference's own GPU policy;
  e_autocast vs that autocast golden (reported).
A block is a handful of roundings deep, so this is where a kernel that loses more than its share shows up; whole networks
are chaotic at the fp16 level (tests/test_oracle_autocast.py::test_fp16_storage_is_chaotic)."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from common import gold, relerr, report
from fgdm_amd import synth
from oracle import arch, nn as onn, precision

pytestmark = pytest.mark.gpu

TOL = 1e-3
U = 'model.diffusion_model.'
# engine block / layer prefix -> golden tag (the generator hashes parameter names WITH the tag, tools/make_goldens.py)
MAP = {
    U + 'input_blocks.1.0.': 'res_320_320.',
    U + 'input_blocks.4.0.': 'res_320_640.',
    U + 'output_blocks.0.0.': 'res_2560_1280.',
    U + 'output_blocks.9.0.': 'res_960_320.',
    U + 'input_blocks.3.0.': 'down.',
    U + 'output_blocks.8.2.': 'up.',
    U + 'input_blocks.4.1.': 'st.',
    U + 'input_blocks.1.1.': 'st320.',
    U + 'input_blocks.7.1.': 'st1280.',
    U + 'adapter.': 'adapter.',
}


def rename(k):
    for pre, tag in MAP.items():
        if k.startswith(pre):
            return tag + k[len(pre):]
    return k


@pytest.fixture(scope='module')
def eng():
    from test_gpu_nets import build_engine
    e = build_engine(gi.SD_CFG, rename, use_adapter=True)
    yield e
    e.close()


def _params(prefix, tag, e):
    """oracle parameter dict {tag + suffix: tensor} for the engine tensors under `prefix`"""
    return {tag + k[len(prefix):]: torch.from_numpy(synth.make_tensor(tag + k[len(prefix):], s))
            for k, s in e.param_shapes().items() if k.startswith(prefix)}


def _check(name, got, key, oracle_fn):
    g32 = torch.from_numpy(gold('ops')[key])
    gac = torch.from_numpy(gold('ops_ac')[key].astype(np.float32))
    with torch.no_grad(), precision.mode('engine'):
        want = oracle_fn()
    got = got.cpu().view_as(g32)
    floor = relerr(gac, g32)
    e_en = report(f'block {name} vs oracle[engine]', relerr(got, want), TOL)
    e_32 = report(f'block {name} vs reference fp32 golden (floor {floor:.2e})', relerr(got, g32), max(TOL, 1.2 * floor))
    report(f'block {name} vs reference autocast golden', relerr(got, gac), 2 * floor)
    assert e_en < TOL
    assert e_32 < max(TOL, 1.2 * floor)


@pytest.mark.parametrize('prefix,split', [(U + 'input_blocks.1.0.', None), (U + 'input_blocks.4.0.', None),
                                          (U + 'output_blocks.0.0.', 1280), (U + 'output_blocks.9.0.', 640)])
def test_resblock(eng, prefix, split):
    tag = MAP[prefix]
    x, emb = gi.get(f'ops/{tag}x'.replace('.x', '_x')), gi.get('ops/emb')
    xs = (x, None) if split is None else (x[:, :split].contiguous(), x[:, split:].contiguous())   # th.cat([h, hs.pop()], 1)
    got = eng.run_block(prefix, xs[0], emb=emb, x_skip=xs[1])
    p = _params(prefix, tag, eng)
    _check(tag[:-1], got, tag[:-1] + '_y', lambda: onn.resblock(p, tag, x, emb))


def test_down_up(eng):
    import torch.nn.functional as F
    from oracle.precision import st, wt
    pre = U + 'input_blocks.3.0.'
    p = _params(pre, 'down.', eng)
    x = gi.get('ops/down_x')
    _check('Downsample', eng.run_block(pre, x), 'down_y',
           lambda: st(F.conv2d(st(x), wt(p['down.op.weight']), p['down.op.bias'], stride=2, padding=1)))
    pre = U + 'output_blocks.8.2.'
    p = _params(pre, 'up.', eng)
    x = gi.get('ops/up_x')
    _check('Upsample', eng.run_block(pre, x), 'up_y',
           lambda: st(F.conv2d(F.interpolate(st(x), scale_factor=2, mode='nearest'), wt(p['up.conv.weight']), p['up.conv.bias'], padding=1)))


@pytest.mark.parametrize('prefix,xkey', [(U + 'input_blocks.1.1.', 'ops/st320_x'), (U + 'input_blocks.4.1.', 'ops/st_x'),
                                         (U + 'input_blocks.7.1.', 'ops/st1280_x')])
def test_spatial_transformer(eng, prefix, xkey):
    tag = MAP[prefix]
    x, ctx = gi.get(xkey), gi.get('ops/ctx')
    got = eng.run_block(prefix, x, ctx=ctx)
    p = _params(prefix, tag, eng)
    _check(f'SpatialTransformer {tag[:-1]} (C={x.shape[1]})', got, tag[:-1] + '_y',
           lambda: onn.spatial_transformer(p, tag, precision.st(x), precision.st(ctx), 8))


@pytest.mark.parametrize('reps', [8, 32], ids=['128x320 tiles', '256x320 tiles'])
def test_spatial_transformer_pipelined_tiles(eng, reps):
    """The same block with the batch replicated until the 64x64-level tile configurations are chosen (M = 4096: 128x320,
    M = 16384: 256x320): the stacked q|k|v GEMM with its transposed V^T destination, LayerNorm statistics from the
    producers' epilogues, the straight-line epilogue paths.  Samples are independent, so every replica must repeat
    the first pair bit for bit and that pair must match the golden."""
    prefix, tag = U + 'input_blocks.1.1.', 'st320.'
    x, ctx = gi.get('ops/st320_x'), gi.get('ops/ctx')
    got = eng.run_block(prefix, x.repeat(reps, 1, 1, 1), ctx=ctx.repeat(reps, 1, 1)).cpu()
    got = got.view(reps, -1)
    assert torch.equal(got, got[:1].expand_as(got))
    p = _params(prefix, tag, eng)
    _check(f'SpatialTransformer st320 x{reps}', got[0], 'st320_y',
           lambda: onn.spatial_transformer(p, tag, precision.st(x), precision.st(ctx), 8))


def test_adapter(eng):
    pre = U + 'adapter.'
    x = gi.get('ops/adapter_x')
    got = eng.run_block(pre, x).cpu()
    p = _params(pre, 'adapter.', eng)
    with torch.no_grad(), precision.mode('engine'):
        want = onn.adapter_forward(p, 'adapter.', precision.st(x))
    g32, gac = gold('ops'), gold('ops_ac')
    off = 0
    for i, w in enumerate(want):
        part = got[off:off + w.numel()].view_as(w)
        off += w.numel()
        floor = relerr(gac[f'adapter_f{i}'].astype(np.float32), g32[f'adapter_f{i}'])
        assert report(f'block Adapter feature {i} vs oracle[engine]', relerr(part, w), TOL) < TOL
        assert report(f'block Adapter feature {i} vs reference fp32 golden (floor {floor:.2e})', relerr(part, g32[f'adapter_f{i}']),
                      max(TOL, 1.2 * floor)) < max(TOL, 1.2 * floor)
    assert off == got.numel()

"""The CPU oracle against golden vectors produced by the imported reference
(tools/make_goldens.py).  fp32 CPU on both sides -> tight tolerances."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_inputs as gi
from common import GOLD, gold, params, relerr
from oracle import arch, nn as onn, samplers, schedule

TOL = 2e-5     # fp32 vs fp32, different op order (e.g. fused attention reshape) only


def test_param_keys_match_reference():
    ref = json.load(open(os.path.join(GOLD, 'param_keys.json')))
    def chk(mine, theirs):
        assert list(mine.keys()) == list(theirs.keys())
        for k, s in mine.items():
            assert list(s) == theirs[k], k
    chk(arch.unet_param_shapes(gi.SD_CFG, adapter=True), ref['unet_fgdm'])
    chk(arch.unet_param_shapes(gi.SD_CFG, adapter=False), ref['unet_plain'])
    chk(arch.unet_param_shapes(gi.SD_CFG, adapter='time'), ref['unet_time_adapter'])
    chk(arch.unet_param_shapes(gi.SD_CFG, adapter=False), ref['controlled_unet'])
    chk(arch.controlnet_param_shapes(gi.SD_CFG), ref['controlnet'])
    chk(arch.unet_param_shapes(gi.SMALL_CFG, adapter=False), ref['unet_small'])
    chk(arch.controlnet_param_shapes(gi.SMALL_CFG), ref['controlnet_small'])
    chk(arch.unet_param_shapes(gi.NARROW_CFG, adapter=False), ref['unet_narrow'])


def test_schedule_tables():
    g = gold('schedule')
    s = schedule.register_schedule()
    np.testing.assert_array_equal(s['betas'], g['betas'])
    np.testing.assert_array_equal(s['alphas_cumprod'], g['alphas_cumprod'])
    for S in (20, 50):
        for eta in (0, 1):
            tab = schedule.ddim_tables(s['alphas_cumprod'], S, float(eta))
            tag = f'S{S}_eta{eta}'
            np.testing.assert_array_equal(tab['timesteps'], g[f'ts_{tag}'])
            np.testing.assert_array_equal(tab['alphas'], g[f'alphas_{tag}'])
            np.testing.assert_array_equal(tab['alphas_prev'], g[f'alphas_prev_{tag}'])
            np.testing.assert_allclose(tab['sigmas'], g[f'sigmas_{tag}'], rtol=1e-6, atol=0)
            np.testing.assert_allclose(tab['sqrt_one_minus_alphas'], g[f'sqrt1m_{tag}'], rtol=1e-6)
    assert list(schedule.ddim_timesteps(50)[:3]) == [1, 21, 41] and schedule.ddim_timesteps(50)[-1] == 981


def test_ddpm_schedule_buffers():
    g = gold('ddpm_schedule')
    s = schedule.register_schedule()
    for k in g.files:
        np.testing.assert_array_equal(s[k], g[k], err_msg=k)


def test_timestep_embedding():
    g = gold('schedule')
    e = onn.timestep_embedding(torch.from_numpy(g['temb_t']), 320)
    np.testing.assert_allclose(e.numpy(), g['temb_320'], rtol=0, atol=1e-6)


@pytest.mark.parametrize('tag,cin,cout', [('res_320_320', 320, 320), ('res_320_640', 320, 640),
                                           ('res_2560_1280', 2560, 1280), ('res_960_320', 960, 320)])
def test_resblock(tag, cin, cout):
    g = gold('ops')
    sh = {}
    arch._res_params(sh, '', cin, cout, 1280)
    p = params(sh, tag + '.')
    y = onn.resblock(p, tag + '.', gi.get(f'ops/{tag}_x'), gi.get('ops/emb'))
    assert relerr(y, g[tag + '_y']) < TOL


def test_down_up_gn():
    g = gold('ops')
    p = params({'op.weight': (320, 320, 3, 3), 'op.bias': (320,)}, 'down.')
    y = F.conv2d(gi.get('ops/down_x'), p['down.op.weight'], p['down.op.bias'], stride=2, padding=1)
    assert relerr(y, g['down_y']) < TOL
    p = params({'conv.weight': (640, 640, 3, 3), 'conv.bias': (640,)}, 'up.')
    y = onn.run_block({'up.0.' + k[3:]: v for k, v in p.items()}, 'up.', [('up', 640)], gi.get('ops/up_x'), None, None)
    assert relerr(y, g['up_y']) < TOL
    for tag, eps in (('gn5', 1e-5), ('gn6', 1e-6)):
        p = params({'weight': (320,), 'bias': (320,)}, tag + '.')
        y = onn._gn(gi.get(f'ops/{tag}_x'), p, tag, eps)
        assert relerr(y, g[tag + '_y']) < TOL


def _attn_shapes(c, ctx):
    return {'to_q.weight': (c, c), 'to_k.weight': (c, ctx), 'to_v.weight': (c, ctx),
            'to_out.0.weight': (c, c), 'to_out.0.bias': (c,)}


def test_attention():
    g = gold('ops')
    x, ctx = gi.get('ops/attn_x'), gi.get('ops/ctx')
    p = params(_attn_shapes(320, 320), 'attn_self.')
    assert relerr(onn.attention(p, 'attn_self.', x, None, 8), g['attn_self_y']) < TOL
    p = params(_attn_shapes(320, 768), 'attn_cross.')
    assert relerr(onn.attention(p, 'attn_cross.', x, ctx, 8), g['attn_cross_y']) < TOL
    p = params(_attn_shapes(1280, 1280), 'attn_self160.')
    assert relerr(onn.attention(p, 'attn_self160.', gi.get('ops/attn160_x'), None, 8), g['attn_self160_y']) < TOL


def test_transformer_and_spatial():
    g = gold('ops')
    ctx = gi.get('ops/ctx')
    sh = {}
    arch._attn_params(sh, 'X.', 320, 768)
    tb = {k[len('X.transformer_blocks.0.'):]: v for k, v in sh.items() if 'transformer_blocks.0.' in k}
    p = params(tb, 'tblock.')
    y = onn.transformer_block(p, 'tblock.', gi.get('ops/ff_x'), ctx, 8)
    assert relerr(y, g['tblock_y']) < TOL
    # FeedForward alone: same math as the ff part of the block
    pf = params({'net.0.proj.weight': (2560, 320), 'net.0.proj.bias': (2560,),
                 'net.2.weight': (320, 1280), 'net.2.bias': (320,)}, 'ff.')
    h = F.linear(gi.get('ops/ff_x'), pf['ff.net.0.proj.weight'], pf['ff.net.0.proj.bias'])
    a, gate = h.chunk(2, dim=-1)
    y = F.linear(a * F.gelu(gate), pf['ff.net.2.weight'], pf['ff.net.2.bias'])
    assert relerr(y, g['ff_y']) < TOL
    sh = {}
    arch._attn_params(sh, '', 640, 768)
    p = params(sh, 'st.')
    y = onn.spatial_transformer(p, 'st.', gi.get('ops/st_x'), ctx, 8)
    assert relerr(y, g['st_y']) < TOL


def test_adapter():
    g = gold('ops')
    p = params(arch.adapter_param_shapes(4, prefix=''), 'adapter.')
    feats = onn.adapter_forward(p, 'adapter.', gi.get('ops/adapter_x'))
    for i, f in enumerate(feats):
        assert relerr(f, g[f'adapter_f{i}']) < TOL, i
    # single ResnetBlocks (down / same) through the same code path
    x = gi.get('ops/arb_x')
    pd = params({'in_conv.weight': (640, 320, 1, 1), 'in_conv.bias': (640,), 'block1.weight': (640, 640, 3, 3),
                 'block1.bias': (640,), 'block2.weight': (640, 640, 1, 1), 'block2.bias': (640,)}, 'arb_down.')
    h = F.avg_pool2d(x, 2, 2)
    h = F.conv2d(h, pd['arb_down.in_conv.weight'], pd['arb_down.in_conv.bias'])
    y = F.conv2d(F.relu(F.conv2d(h, pd['arb_down.block1.weight'], pd['arb_down.block1.bias'], padding=1)),
                 pd['arb_down.block2.weight'], pd['arb_down.block2.bias']) + h
    assert relerr(y, g['arb_down_y']) < TOL


def test_unet_full_width():
    g = gold('unet_full')
    p = params(arch.unet_param_shapes(gi.SD_CFG, adapter=True), 'model.diffusion_model.')
    ctx = gi.get('unet/ctx')
    t = torch.from_numpy(g['t'])
    for hw in (8, 16):
        x = gi.get(f'unet/x{hw}')
        e = onn.unet_forward(p, gi.SD_CFG, x, t, ctx, prefix='model.diffusion_model.')
        assert relerr(e, g[f'eps_orig{hw}']) < TOL
        e = onn.unet_forward(p, gi.SD_CFG, x, t, ctx, prefix='model.diffusion_model.', use_adapter=True)
        assert relerr(e, g[f'eps_fgdm{hw}']) < TOL
    pt = params(arch.unet_param_shapes(gi.SD_CFG, adapter='time'), 'model.diffusion_model.')
    for hw in (8, 16):
        e = onn.unet_forward(pt, gi.SD_CFG, gi.get(f'unet/x{hw}'), t, ctx, prefix='model.diffusion_model.', use_adapter='time')
        assert relerr(e, g[f'eps_tadapt{hw}']) < TOL


def test_controlnet_full_width():
    g = gold('controlnet_full')
    p = params(arch.unet_param_shapes(gi.SD_CFG, adapter=False), 'model.diffusion_model.')
    p.update(params(arch.controlnet_param_shapes(gi.SD_CFG), 'control_model.'))
    ctx, x = gi.get('cn/ctx'), gi.get('cn/x')
    t = torch.from_numpy(g['t'])
    ctrl = onn.controlnet_forward(p, gi.SD_CFG, x, gi.hint(2, 64, 45), t, ctx, prefix='control_model.')
    assert len(ctrl) == 13
    for i, c in enumerate(ctrl):
        assert relerr(c, g[f'ctrl{i}']) < TOL, i
    e = onn.control_ldm_apply(p, gi.SD_CFG, x, t, ctx, [gi.hint(2, 64, 45)], scales=gi.CTRL_SCALES)
    assert relerr(e, g['eps_ctrl']) < TOL
    e = onn.control_ldm_apply(p, gi.SD_CFG, x, t, ctx, None)
    assert relerr(e, g['eps_noctrl']) < TOL
    assert relerr(onn.hint_block(p, 'control_model.', gi.hint(1, 64, 46)), g['guided']) < TOL


def test_reduced_nets_64():
    g = gold('small_nets')
    ctx, x = gi.get('small/ctx'), gi.get('small/x')
    t = torch.from_numpy(g['t'])
    p = params(arch.unet_param_shapes(gi.SMALL_CFG, adapter=False), 'small.')
    assert relerr(onn.unet_forward(p, gi.SMALL_CFG, x, t, ctx, prefix='small.'), g['eps_small']) < TOL
    p.update(params(arch.controlnet_param_shapes(gi.SMALL_CFG), 'small_cn.'))
    e = onn.control_ldm_apply(p, gi.SMALL_CFG, x, t, ctx, [gi.hint(2, 512, 47)],
                              unet_prefix='small.', cn_prefixes=('small_cn.',))
    assert relerr(e, g['eps_small_ctrl']) < TOL
    pn = params(arch.unet_param_shapes(gi.NARROW_CFG, adapter=False), 'narrow.')
    e = onn.unet_forward(pn, gi.NARROW_CFG, x[:, :, :32, :32].contiguous(), t, ctx, prefix='narrow.')
    assert relerr(e, g['eps_narrow32']) < TOL


# ----------------------------------------------------------------------------- samplers
def analytic_eps(x, t, c):
    """same closed form as tools/make_goldens.py:analytic_eps"""
    if isinstance(c, dict):
        cc = c['c_crossattn'][0]
        hint = c['c_concat'][0] if c.get('c_concat') is not None else None
    else:
        cc, hint = c, None
    s = cc.mean(dim=(1, 2)).reshape(-1, 1, 1, 1)
    tt = (t.float() / 1000.0).reshape(-1, 1, 1, 1)
    e = 0.7 * x + 0.2 * torch.sin(3.0 * x + s) + 0.1 * tt * torch.roll(x, 1, dims=3) + 0.05 * s
    if hint is not None:
        e = e + 0.1 * F.avg_pool2d(hint, 8).mean(dim=1, keepdim=True)
    return e


STOL = 1e-5


def test_ddim_trajectories():
    g = gold('samplers')
    sched = schedule.register_schedule()
    x_T, c, uc = gi.get('samp/x_T'), gi.get('samp/c'), gi.get('samp/uc')
    for S, scale, eta in ((50, 7.5, 0.0), (20, 9.0, 0.0), (20, 7.5, 1.0), (10, 1.0, 0.0)):
        calls = [0]
        def fn(x, t, cc):
            calls[0] += 1
            return analytic_eps(x, t, cc)
        torch.manual_seed(123)
        out, inter = samplers.ddim_sample(fn, sched, S, x_T.shape, c, x_T, eta=eta, scale=scale, uc=uc, log_every_t=5)
        tag = f'ddim_S{S}_s{scale}_eta{eta}'
        assert relerr(out, g[tag]) < STOL, tag
        assert relerr(torch.stack(inter['x_inter']), g[tag + '_xinter']) < STOL
        assert relerr(torch.stack(inter['pred_x0']), g[tag + '_predx0']) < STOL
        assert calls[0] == int(g[tag + '_calls'][0])


def test_plms_controlnet_mask_ancestral():
    g = gold('samplers')
    sched = schedule.register_schedule()
    x_T, c, uc = gi.get('samp/x_T'), gi.get('samp/c'), gi.get('samp/uc')
    calls = [0]
    def fn(x, t, cc):
        calls[0] += 1
        return analytic_eps(x, t, cc)
    out, inter = samplers.plms_sample(fn, sched, 50, x_T.shape, c, x_T, scale=7.5, uc=uc, log_every_t=5)
    assert relerr(out, g['plms_S50']) < STOL
    assert relerr(torch.stack(inter['x_inter']), g['plms_S50_xinter']) < STOL
    assert calls[0] == int(g['plms_S50_calls'][0]) == 51
    # ControlNet sampler: dict conds, two sequential calls per step
    hint = gi.hint(2, 64, 48)
    cond = {'c_concat': [hint], 'c_crossattn': [c]}
    ucond = {'c_concat': [hint], 'c_crossattn': [uc]}
    calls[0] = 0
    out, _ = samplers.ddim_sample(fn, sched, 20, x_T.shape, cond, x_T, scale=9.0, uc=ucond, cfg_mode='sequential')
    assert relerr(out, g['cn_ddim_S20']) < STOL
    assert calls[0] == int(g['cn_ddim_S20_calls'][0]) == 40
    torch.manual_seed(321)
    out, _ = samplers.ddim_sample(fn, sched, 10, x_T.shape, c, x_T, scale=7.5, uc=uc,
                                  mask=gi.get('samp/mask'), x0=gi.get('samp/x0'))
    assert relerr(out, g['ddim_mask_S10']) < STOL
    torch.manual_seed(99)
    out, inter = samplers.p_sample_loop(fn, sched, c, x_T.shape, x_T, timesteps=12, log_every_t=4)
    assert relerr(out, g['ancestral_T12']) < STOL
    assert relerr(torch.stack(inter), g['ancestral_T12_inter']) < STOL


def test_ddim_over_reduced_unet():
    g = gold('sampler_unet')
    sched = schedule.register_schedule()
    p = params(arch.unet_param_shapes(gi.SMALL_CFG, adapter=False), 'small.')
    fn = lambda x, t, c: onn.unet_forward(p, gi.SMALL_CFG, x, t, c, prefix='small.')
    x_T = gi.get('sunet/x_T')
    out, inter = samplers.ddim_sample(fn, sched, 10, x_T.shape, gi.get('sunet/c'), x_T, scale=7.5,
                                      uc=gi.get('sunet/uc'), log_every_t=1)
    assert relerr(out, g['out']) < 1e-4
    assert relerr(torch.stack(inter['x_inter']), g['xinter']) < 1e-4


def test_adapt_unet_multi_adapter():
    """AdaptUNetModel (openaimodel.py:887-1320), num_prompts = 3: keys and the summed-adapter forward."""
    ref = json.load(open(os.path.join(GOLD, 'param_keys.json')))['adapt_unet_3']
    mine = arch.unet_param_shapes(gi.SD_CFG, adapter=True, num_prompts=3)
    assert list(mine.keys()) == list(ref.keys())
    assert all(tuple(ref[k]) == tuple(v) for k, v in mine.items())
    g = gold('adapt_unet')
    p = params(mine, 'model.diffusion_model.')
    x, ctx = gi.get('unet/x16'), gi.get('unet/ctx')
    t = torch.tensor([981, 1])
    conds = [gi.get('adapt/cond0'), gi.get('adapt/cond1')]
    kw = dict(prefix='model.diffusion_model.', use_adapter=True)
    with torch.no_grad():
        assert relerr(onn.unet_forward(p, gi.SD_CFG, x, t, ctx, conds=conds, **kw), g['eps_conds']) < 2e-5
        assert relerr(onn.unet_forward(p, gi.SD_CFG, x, t, ctx, conds=conds, pcond=gi.get('adapt/control'), **kw),
                      g['eps_conds_control']) < 2e-5
        assert relerr(onn.unet_forward(p, gi.SD_CFG, x, t, ctx, **kw), g['eps_plain']) < 2e-5


def test_full_size_fixture_is_certified_against_reference_apply_model():
    """tests/golden/full_size*.npz hold what the reference's own ControlLDM.apply_model (controlnet/cldm/cldm.py:836-849) returns:
    tools/make_goldens.py --only full_size_check re-ran the CFG pair at t = 981 through that method (the class itself, its heavy
    LatentDiffusion constructor bypassed) in the build container and recorded the comparison next to the fixtures."""
    import json
    import os
    from common import GOLD
    for name in ('full_size', 'full_size_ac'):
        rec = json.load(open(os.path.join(GOLD, name + '_apply_model_check.json')))
        assert rec['bit_identical'] and rec['max_abs_diff'] == 0.0
        assert 'ControlLDM.apply_model' in rec['through'] and rec['fixture'].startswith(name + '.npz')

"""Whole-network parity through the C ABI: HIP engine vs golden vectors from the imported reference
(tests/golden, fp32 CPU) and vs the CPU oracle at full latent size.

Tolerance for one complete UNet / ControlNet evaluation: normwise relative error <= 4e-3.
Why not 1e-3: rounding ONLY the GEMM operands (weights + inputs) of the fp32 reference to fp16 -- which any
fp16-MFMA implementation must do -- already moves the output of this 60-layer network by 1.5e-3 (measured with
the oracle, see DESIGN.md "Numerics"); with fp16 activation storage the emulated floor is 1.7e-3.  Per-kernel
tests (test_gpu_ops.py) hold the 1e-3 bar."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from common import gold, relerr, report
from fgdm_amd import synth

pytestmark = pytest.mark.gpu

NET_TOL = 4e-3


def build_engine(cfg, rename, use_adapter=False, n_controlnets=0):
    """Engine with synthetic weights; `rename(engine_key) -> generator key` (the goldens hash names with
    the prefixes tools/make_goldens.py used)."""
    from fgdm_amd.engine import Engine
    e = Engine(cfg, use_adapter=use_adapter, n_controlnets=n_controlnets)
    for k, shape in e.param_shapes().items():
        e.load_tensor(k, synth.make_tensor(rename(k), shape))
    e.finalize()
    return e


@pytest.fixture(scope='module')
def sd_engine():
    e = build_engine(gi.SD_CFG, lambda k: k, use_adapter=True, n_controlnets=1)
    yield e
    e.close()


def small_rename(k):
    return k.replace('model.diffusion_model.', 'small.').replace('control_model.', 'small_cn.')


@pytest.fixture(scope='module')
def small_engine():
    e = build_engine(gi.SMALL_CFG, small_rename, n_controlnets=1)
    yield e
    e.close()


def test_unet_full_width_vs_reference_goldens(sd_engine):
    from fgdm_amd import _lib
    g = gold('unet_full')
    ctx = gi.get('unet/ctx')
    t = torch.from_numpy(g['t'])
    for hw in (8, 16):
        x = gi.get(f'unet/x{hw}')
        e = sd_engine.apply_model(x, t, ctx, flags=_lib.FLAG_USE_ORIGINAL | _lib.FLAG_NO_CONTROL)
        assert report(f'unet forward_original {hw}x{hw} vs reference golden', relerr(e.cpu(), g[f'eps_orig{hw}']), NET_TOL) < NET_TOL
        e = sd_engine.apply_model(x, t, ctx, flags=_lib.FLAG_NO_CONTROL)
        assert report(f'unet FG-DM adapter {hw}x{hw} vs reference golden', relerr(e.cpu(), g[f'eps_fgdm{hw}']), NET_TOL) < NET_TOL


def test_time_adapter_unet_vs_reference_goldens():
    """use_time_adapter=True: TimeAdapter features from time-conditioned ResBlocks (adapter.py:387-417)."""
    from fgdm_amd import _lib
    e = build_engine(gi.SD_CFG, lambda k: k, use_adapter='time')
    try:
        g = gold('unet_full')
        ctx = gi.get('unet/ctx')
        t = torch.from_numpy(g['t'])
        for hw in (8, 16):
            eps = e.apply_model(gi.get(f'unet/x{hw}'), t, ctx, flags=_lib.FLAG_NO_CONTROL)
            assert report(f'unet TimeAdapter {hw}x{hw} vs reference golden', relerr(eps.cpu(), g[f'eps_tadapt{hw}']), NET_TOL) < NET_TOL
    finally:
        e.close()


def test_controlnet_full_width_vs_reference_goldens():
    from fgdm_amd import _lib
    # ControlledUnetModel has no adapter: separate engine without it (keys identical to the golden's)
    e = build_engine(gi.SD_CFG, lambda k: k, use_adapter=False, n_controlnets=1)
    try:
        g = gold('controlnet_full')
        ctx, x = gi.get('cn/ctx'), gi.get('cn/x')
        t = torch.from_numpy(g['t'])
        e.set_hint(0, gi.hint(2, 64, 45).cuda())
        ctrl = e.controlnet(0, x, t, ctx)
        assert len(ctrl) == 13
        for i, c in enumerate(ctrl):
            assert tuple(c.shape) == g[f'ctrl{i}'].shape
            assert report(f'controlnet residual {i} vs reference golden', relerr(c.cpu(), g[f'ctrl{i}']), NET_TOL) < NET_TOL
        eps = e.apply_model(x, t, ctx, control_scales=gi.CTRL_SCALES)
        assert report('ControlLDM.apply_model (scaled control) vs reference golden', relerr(eps.cpu(), g['eps_ctrl']), NET_TOL) < NET_TOL
        eps = e.apply_model(x, t, ctx, flags=_lib.FLAG_NO_CONTROL)
        assert report('ControlledUnet control=None vs reference golden', relerr(eps.cpu(), g['eps_noctrl']), NET_TOL) < NET_TOL
        # hint block alone: a zero latent/ctx isolates it?  No -- check through the public cache instead:
        # the cached guided hint feeds ctrl0 (= zero_conv0(conv_in(x) + guided)), already covered above.
    finally:
        e.close()


def test_reduced_nets_at_full_latent_size(small_engine):
    from fgdm_amd import _lib
    g = gold('small_nets')
    ctx, x = gi.get('small/ctx'), gi.get('small/x')
    t = torch.from_numpy(g['t'])
    e = small_engine.apply_model(x, t, ctx, flags=_lib.FLAG_NO_CONTROL)
    assert report('reduced UNet 64x64 vs reference golden', relerr(e.cpu(), g['eps_small']), NET_TOL) < NET_TOL
    small_engine.set_hint(0, gi.hint(2, 512, 47).cuda())
    e = small_engine.apply_model(x, t, ctx)
    assert report('reduced UNet+ControlNet 64x64, hint 512 vs reference golden', relerr(e.cpu(), g['eps_small_ctrl']), NET_TOL) < NET_TOL


def test_batch_rows_are_independent_and_deterministic(small_engine):
    """Sharding invariant: a sample's eps does not depend on its batch neighbours or position,
    bit for bit -- this is what makes N-rank results identical to 1-rank results."""
    from fgdm_amd import _lib
    x = torch.from_numpy(synth.latents(4, 32, 32, seed=5))
    ctx = torch.from_numpy(synth.context(4, seed=6))
    t = torch.tensor([500, 500, 500, 500])
    f = _lib.FLAG_NO_CONTROL
    full = small_engine.apply_model(x, t, ctx, flags=f).cpu()
    again = small_engine.apply_model(x, t, ctx, flags=f).cpu()
    assert torch.equal(full, again)
    lo = small_engine.apply_model(x[:2], t[:2], ctx[:2], flags=f).cpu()
    hi = small_engine.apply_model(x[2:], t[2:], ctx[2:], flags=f).cpu()
    assert torch.equal(full[:2], lo) and torch.equal(full[2:], hi)


def test_ddim_trajectory_vs_reference_sampler(small_engine):
    """End-to-end compounding: 10 DDIM steps with CFG 7.5 on a 16x16 latent, reference DDIMSampler + reference
    UNet (golden) vs the device-side loop.  Errors accumulate over 20 network evaluations -> 1e-2."""
    from fgdm_amd import _lib
    from oracle import schedule
    g = gold('sampler_unet')
    sched = schedule.register_schedule()
    tab = schedule.ddim_tables(sched['alphas_cumprod'], 10, 0.0)
    out = small_engine.sample_ddim(gi.get('sunet/x_T'), gi.get('sunet/c'), gi.get('sunet/uc'), 7.5,
                                   tab['timesteps'], tab['alphas'], tab['alphas_prev'], tab['sqrt_one_minus_alphas'],
                                   flags=_lib.FLAG_NO_CONTROL)
    assert report('10-step DDIM CFG7.5 trajectory vs reference sampler+UNet', relerr(out.cpu(), g['out']), 1e-2) < 1e-2


def test_pipelined_and_two_stage_kernels_both_match_oracle(small_engine):
    """At a batch large enough for the automatic tile choice to pick the pipelined big-tile kernels (igemm2.hip),
    the output is checked against the CPU oracle, and so is the same evaluation forced onto the 2-stage 128x128
    kernel.  The two GPU runs differ from each other by fp16 rounding placement + summation order only."""
    from fgdm_amd import _lib
    from common import params
    from oracle import arch, nn as onn
    lib = _lib.load()
    B = 4
    x = torch.from_numpy(synth.latents(B, 64, 64, seed=21))
    ctx = torch.from_numpy(synth.context(B, seed=22))
    t = torch.full((B,), 601, dtype=torch.long)
    hint = torch.from_numpy(synth.hint(B, 512, seed=23))
    small_engine.set_hint(0, hint.cuda())
    xx, cc, tt = torch.cat([x, x]), torch.cat([ctx, ctx]), torch.cat([t, t])       # 2B rows: big-tile territory
    auto = small_engine.apply_model(xx, tt, cc).cpu()
    try:
        lib.fgdm_debug_force_igemm_cfg(1)
        forced = small_engine.apply_model(xx, tt, cc).cpu()
    finally:
        lib.fgdm_debug_force_igemm_cfg(0)
    p = params(arch.unet_param_shapes(gi.SMALL_CFG, adapter=False), 'small.')
    p.update(params(arch.controlnet_param_shapes(gi.SMALL_CFG), 'small_cn.'))
    want = onn.control_ldm_apply(p, gi.SMALL_CFG, x, t, ctx, [hint], unet_prefix='small.', cn_prefixes=('small_cn.',))
    assert torch.equal(auto[:B], auto[B:])
    assert report('auto tiles (pipelined kernels) vs oracle, 8 rows 64x64', relerr(auto[:B], want), NET_TOL) < NET_TOL
    assert report('forced 2-stage kernel vs oracle, 8 rows 64x64', relerr(forced[:B], want), NET_TOL) < NET_TOL
    assert report('auto vs forced tiles', relerr(auto, forced), NET_TOL) < NET_TOL


def test_fractional_timesteps(small_engine):
    """DPM-Solver evaluates the network at fractional t: float timesteps must equal the int path at integers and
    follow the oracle's timestep_embedding in between."""
    from fgdm_amd import _lib
    from common import params
    from oracle import arch, nn as onn
    x = torch.from_numpy(synth.latents(2, 16, 16, seed=31))
    ctx = torch.from_numpy(synth.context(2, seed=32))
    f = _lib.FLAG_NO_CONTROL
    a = small_engine.apply_model(x, torch.tensor([500, 37]), ctx, flags=f).cpu()
    b = small_engine.apply_model(x, torch.tensor([500.0, 37.0]), ctx, flags=f).cpu()
    assert torch.equal(a, b)
    tf = torch.tensor([949.05, 0.5])
    got = small_engine.apply_model(x, tf, ctx, flags=f).cpu()
    p = params(arch.unet_param_shapes(gi.SMALL_CFG, adapter=False), 'small.')
    want = onn.unet_forward(p, gi.SMALL_CFG, x, tf, ctx, prefix='small.')
    assert report('UNet at fractional timesteps vs oracle', relerr(got, want), NET_TOL) < NET_TOL


def test_context_cache_is_exact_and_invalidated(small_engine):
    """fgdm_set_context: K/V projections of a registered context are reused across calls.  Results are bit-identical
    to passing the context every time, and an in-place change of the context tensor is noticed."""
    from fgdm_amd import _lib
    e = small_engine
    x = gi.get('small/x')[:, :, :16, :16].contiguous().cuda()
    t = torch.tensor([981, 1]).cuda()
    ctx = gi.get('small/ctx').cuda()
    flags = _lib.FLAG_NO_CONTROL
    e.cache_context = False
    want = e.apply_model(x, t, ctx, flags=flags).clone()
    e.cache_context = True
    a = e.apply_model(x, t, ctx, flags=flags).clone()      # registers the context
    b = e.apply_model(x, t, ctx, flags=flags).clone()      # served from the cached projections
    assert torch.equal(a, want) and torch.equal(b, want)
    ctx.mul_(0.5)                                          # in-place: must be re-projected
    c = e.apply_model(x, t, ctx, flags=flags)
    e.cache_context = False
    assert torch.equal(c, e.apply_model(x, t, ctx, flags=flags))
    assert not torch.equal(c, want)
    e.cache_context = True
    # a context of another batch size replaces the cache
    d = e.apply_model(x[:1], t[:1], ctx[:1].contiguous(), flags=flags)
    assert torch.equal(d[0], c[0])


def test_adapt_unet_multi_adapter_vs_reference_goldens():
    """AdaptUNetModel (openaimodel.py:887-1320), num_prompts = 3: `conds` through two further adapters, `control` as the
    adapter prompt; through the LatentDiffusion mirror's apply_model(..., conds=, control=) pass-through."""
    from fgdm_amd import models
    m = models.LatentDiffusion(gi.SD_CFG, use_adapter=True, num_prompts=3)
    try:
        sd = {k: synth.make_tensor(k, s) for k, s in m.engine.param_shapes().items()}
        assert not m.load_state_dict(sd)[0]
        g = gold('adapt_unet')
        x, ctx = gi.get('unet/x16').cuda(), gi.get('unet/ctx').cuda()
        t = torch.tensor([981, 1]).cuda()
        conds = [gi.get('adapt/cond0').cuda(), gi.get('adapt/cond1').cuda()]
        e = m.apply_model(x, t, ctx, conds=conds)
        assert report('AdaptUNetModel conds vs reference golden', relerr(e.cpu(), g['eps_conds']), NET_TOL) < NET_TOL
        e2 = m.apply_model(x, t, ctx, conds=conds)                      # cached adapter features
        assert torch.equal(e, e2)
        e = m.apply_model(x, t, ctx, conds=conds, control=gi.get('adapt/control').cuda())
        assert report('AdaptUNetModel conds + control vs reference golden', relerr(e.cpu(), g['eps_conds_control']), NET_TOL) < NET_TOL
        e = m.apply_model(x, t, ctx, conds=None)                        # conds=None: only the main adapter
        assert report('AdaptUNetModel without conds vs reference golden', relerr(e.cpu(), g['eps_plain']), NET_TOL) < NET_TOL
    finally:
        m.engine.close()


def test_non_square_latent_and_batch_one_vs_oracle(small_engine):
    """Latents need not be square (the scripts take --H / --W): 16x24 latent, hint 128x192, batch 1, against the oracle."""
    from oracle import arch, nn as onn
    cfg = gi.SMALL_CFG
    x = torch.from_numpy(synth.latents(1, 16, 24, seed=31))
    ctx = torch.from_numpy(synth.context(1, seed=32))
    hint = torch.from_numpy(synth.hint(1, res=128, seed=33))
    hint = torch.cat([hint, hint.flip(-1)[..., :64]], dim=-1).contiguous()          # 128 x 192
    t = torch.tensor([501])
    small_engine.set_hint(0, hint.cuda())
    got = small_engine.apply_model(x, t, ctx, control_scales=gi.CTRL_SCALES)
    p = {}
    for pre, name, shapes in (('model.diffusion_model.', 'small.', arch.unet_param_shapes(cfg, adapter=False)),
                              ('control_model.', 'small_cn.', arch.controlnet_param_shapes(cfg))):
        for k, s in shapes.items():
            p[pre + k] = torch.from_numpy(synth.make_tensor(name + k, s))
    with torch.no_grad():
        want = onn.control_ldm_apply(p, cfg, x, t, ctx, [hint], scales=gi.CTRL_SCALES)
    assert tuple(got.shape) == (1, 4, 16, 24)
    assert report('non-square 16x24 latent, batch 1, UNet+ControlNet vs oracle', relerr(got.cpu(), want), NET_TOL) < NET_TOL


def test_two_controlnets_sum_of_residuals_vs_oracle():
    """BASELINE configs 4/5: several ControlNets on one UNet.  Not in the reference (one control_model per ControlLDM);
    defined as the element-wise sum of the scaled 13-tensor residual lists, each encoder pinned separately by
    the ControlNet goldens.  Through the ControlLDM mirror with one hint per control model."""
    from fgdm_amd import models
    from oracle import nn as onn
    cfg = gi.SMALL_CFG
    m = models.ControlLDM(cfg, n_controlnets=2)
    try:
        sd = {k: synth.make_tensor(k, s) for k, s in m.engine.param_shapes().items()}
        assert any(k.startswith('control_model_1.') for k in sd)
        assert not m.load_state_dict(sd)[0]
        m.control_scales = [0.7] * 13
        B, H = 2, 16
        x = torch.from_numpy(synth.latents(B, H, H, seed=41))
        ctx = torch.from_numpy(synth.context(B, seed=42))
        h0 = torch.from_numpy(synth.hint(B, res=8 * H, seed=43))
        h1 = torch.from_numpy(synth.hint(B, res=8 * H, seed=44))
        t = torch.tensor([741, 21])
        got = m.apply_model(x.cuda(), t.cuda(), {'c_concat': [h0.cuda(), h1.cuda()], 'c_crossattn': [ctx.cuda()]})
        p = {k: torch.from_numpy(v) for k, v in sd.items()}
        with torch.no_grad():
            want = onn.control_ldm_apply(p, cfg, x, t, ctx, [h0, h1], scales=[0.7] * 13,
                                         cn_prefixes=('control_model.', 'control_model_1.'))
            one = onn.control_ldm_apply(p, cfg, x, t, ctx, [h0], scales=[0.7] * 13)
        assert report('two ControlNets (summed residuals) vs oracle', relerr(got.cpu(), want), NET_TOL) < NET_TOL
        assert relerr(want, one) > 1e-2          # the second control model really contributes
    finally:
        m.engine.close()


def test_cfg_pairs_shared_prefix_is_bit_identical(small_engine, sd_engine):
    """FGDM_FLAG_CFG_PAIRS: for a cat([x]*2) classifier-free-guidance batch the network prefix up to the first
    cross-attention (conv_in, first ResBlock, first self-attention, adapter, ControlNet stem) runs once on B/2 rows.
    Output must equal the plain evaluation bit for bit."""
    from fgdm_amd import _lib
    # UNet + ControlNet (reduced depth), hint shared by both halves
    xs = gi.get('small/x')[:, :, :32, :32].contiguous()
    x = torch.cat([xs, xs]).cuda()
    t = torch.tensor([981, 21, 981, 21]).cuda()
    ctx = torch.cat([gi.get('small/ctx'), torch.from_numpy(synth.context(2, seed=77))]).cuda()
    hint = torch.from_numpy(synth.hint(2, res=256, seed=78)).cuda()
    small_engine.set_hint(0, hint)
    a = small_engine.apply_model(x, t, ctx, control_scales=gi.CTRL_SCALES).clone()
    b = small_engine.apply_model(x, t, ctx, control_scales=gi.CTRL_SCALES, flags=_lib.FLAG_CFG_PAIRS)
    assert torch.equal(a, b)
    assert not torch.equal(a[:2], a[2:])            # the halves really differ (different contexts)
    # full SD UNet with the FG-DM adapter, no control
    xs = gi.get('unet/x16')
    x = torch.cat([xs, xs]).cuda()
    ctx = torch.cat([gi.get('unet/ctx'), torch.from_numpy(synth.context(2, seed=79))]).cuda()
    a = sd_engine.apply_model(x, t, ctx, flags=_lib.FLAG_NO_CONTROL).clone()
    b = sd_engine.apply_model(x, t, ctx, flags=_lib.FLAG_NO_CONTROL | _lib.FLAG_CFG_PAIRS)
    assert torch.equal(a, b)
    a = sd_engine.apply_model(x, t, ctx, flags=_lib.FLAG_NO_CONTROL | _lib.FLAG_USE_ORIGINAL).clone()
    b = sd_engine.apply_model(x, t, ctx, flags=_lib.FLAG_NO_CONTROL | _lib.FLAG_USE_ORIGINAL | _lib.FLAG_CFG_PAIRS)
    assert torch.equal(a, b)

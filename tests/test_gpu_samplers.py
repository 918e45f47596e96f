"""Drop-in samplers with the REAL fused HIP kernels on the GPU: reference golden trajectories (analytic eps
model -> pins the sampler arithmetic exactly) and the full drop-in stack (ControlLDM mirror + ControlNet
DDIM sampler over the HIP engine) against the CPU oracle."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from common import CAP_CHAIN, check_net_vs_oracle, gold, relerr, report
from fgdm_amd import models, samplers, synth
from test_oracle_golden import analytic_eps
from test_samplers_host import AnalyticLDM

pytestmark = pytest.mark.gpu
STOL = 2e-5     # fp32 on both sides; device sin/FMA contraction differ from the CPU by a few ulp per step


@pytest.fixture()
def cpu_noise(monkeypatch):
    """draw sampler noise from the CPU generator (as the reference run that produced the goldens did)"""
    monkeypatch.setattr(samplers, '_randn', lambda shape, device: torch.randn(shape).to(device))
    monkeypatch.setattr(models, '_randn', lambda shape, device: torch.randn(shape).to(device))
    monkeypatch.setattr(torch, 'randn_like', lambda x, **k: torch.randn(x.shape).to(x.device))


def _cuda(*ts):
    return [t.cuda() for t in ts]


def test_ddim_trajectories_on_device(cpu_noise):
    g = gold('samplers')
    x_T, c, uc = _cuda(gi.get('samp/x_T'), gi.get('samp/c'), gi.get('samp/uc'))
    for S, scale, eta in ((50, 7.5, 0.0), (20, 9.0, 0.0), (20, 7.5, 1.0), (10, 1.0, 0.0)):
        m = AnalyticLDM('cuda')
        torch.manual_seed(123)
        out, inter = samplers.DDIMSampler(m).sample(S, 2, (4, 8, 8), conditioning=c, x_T=x_T, eta=eta, verbose=False,
                                                    unconditional_guidance_scale=scale, unconditional_conditioning=uc,
                                                    log_every_t=5)
        tag = f'ddim_S{S}_s{scale}_eta{eta}'
        assert out.is_cuda
        assert report(f'DDIMSampler {tag} (HIP kernels) vs reference', relerr(out.cpu(), g[tag]), STOL) < STOL
        assert relerr(torch.stack(inter['pred_x0']).cpu(), g[tag + '_predx0']) < STOL


def test_plms_mask_ancestral_on_device(cpu_noise):
    g = gold('samplers')
    x_T, c, uc = _cuda(gi.get('samp/x_T'), gi.get('samp/c'), gi.get('samp/uc'))
    m = AnalyticLDM('cuda')
    out, _ = samplers.PLMSSampler(m).sample(50, 2, (4, 8, 8), conditioning=c, x_T=x_T, eta=0.0, verbose=False,
                                            unconditional_guidance_scale=7.5, unconditional_conditioning=uc)
    assert report('PLMSSampler S50 (HIP kernels) vs reference', relerr(out.cpu(), g['plms_S50']), STOL) < STOL
    assert m.calls == 51
    torch.manual_seed(321)
    out, _ = samplers.DDIMSampler(AnalyticLDM('cuda')).sample(
        10, 2, (4, 8, 8), conditioning=c, x_T=x_T, eta=0.0, verbose=False, mask=gi.get('samp/mask').cuda(),
        x0=gi.get('samp/x0').cuda(), unconditional_guidance_scale=7.5, unconditional_conditioning=uc)
    assert report('DDIM inpainting mask blend vs reference', relerr(out.cpu(), g['ddim_mask_S10']), STOL) < STOL
    m = AnalyticLDM('cuda')
    m.log_every_t = 4
    torch.manual_seed(99)
    img, inter = m.p_sample_loop(c, (2, 4, 8, 8), return_intermediates=True, x_T=x_T, verbose=False, timesteps=12)
    assert report('p_sample_loop 12 steps vs reference', relerr(img.cpu(), g['ancestral_T12']), STOL) < STOL
    assert relerr(torch.stack(inter).cpu(), g['ancestral_T12_inter']) < STOL


def test_controlnet_sampler_over_engine_vs_oracle():
    """ControlLDM mirror + ControlNet DDIM sampler (dict conds, CFG 9.0, control_scales) on the HIP engine vs the
    oracle's ddim_hacked restatement (two sequential calls per step), 4 steps on a 16x16 latent."""
    from oracle import arch, nn as onn, samplers as osamp, schedule
    cfg = gi.SMALL_CFG
    model = models.ControlLDM(cfg, n_controlnets=1)
    sd = {k: synth.make_tensor(k, s) for k, s in model.engine.param_shapes().items()}
    missing, _ = model.load_state_dict(sd)
    assert not missing
    model.control_scales = [0.9] * 13
    B, H = 2, 16
    x_T = torch.from_numpy(synth.latents(B, H, H, seed=11))
    c, uc = torch.from_numpy(synth.context(B, seed=12)), torch.from_numpy(synth.context(B, seed=13))
    hint = torch.from_numpy(synth.hint(B, 8 * H, seed=14))
    hint_d = hint.cuda()
    cond = {'c_concat': [hint_d], 'c_crossattn': [c.cuda()]}
    ucond = {'c_concat': [hint_d], 'c_crossattn': [uc.cuda()]}
    out, inter = samplers.ControlDDIMSampler(model).sample(4, B, (4, H, H), cond, verbose=False, eta=0.0, x_T=x_T.cuda(),
                                                           unconditional_guidance_scale=9.0,
                                                           unconditional_conditioning=ucond)
    p = {k: torch.from_numpy(v) for k, v in sd.items()}
    fn = lambda x, t, cc: onn.control_ldm_apply(p, cfg, x, t, cc['c_crossattn'][0], [cc['c_concat'][0]], scales=[0.9] * 13)
    run = lambda: osamp.ddim_sample(fn, schedule.register_schedule(), 4, x_T.shape, {'c_concat': [hint], 'c_crossattn': [c]},
                                    x_T, scale=9.0, uc={'c_concat': [hint], 'c_crossattn': [uc]}, cfg_mode='sequential')[0]
    check_net_vs_oracle('drop-in ControlLDM + ControlDDIMSampler, 4 steps CFG 9', out.cpu(), run, cap=CAP_CHAIN)
    assert len(inter['x_inter']) == 3
    model.engine.close()


def test_dpm_solver_and_encode_on_device():
    g = gold('samplers2')
    x_T, c, uc = _cuda(gi.get('samp/x_T'), gi.get('samp/c'), gi.get('samp/uc'))
    for S, scale in ((20, 7.5), (10, 7.5), (12, 1.0)):
        m = AnalyticLDM('cuda')
        out, _ = samplers.DPMSolverSampler(m).sample(S, 2, (4, 8, 8), conditioning=c, x_T=x_T, verbose=False,
                                                     unconditional_guidance_scale=scale, unconditional_conditioning=uc)
        assert report(f'DPMSolverSampler S{S} scale {scale} (HIP kernels) vs reference',
                      relerr(out.cpu(), g[f'dpm_S{S}_s{scale}']), 5e-5) < 5e-5
    m = AnalyticLDM('cuda')
    smp = samplers.ControlDDIMSampler(m)
    smp.make_schedule(20, ddim_eta=0.0, verbose=False)
    enc, _ = smp.encode(gi.get('samp/x0').cuda(), c, 12, unconditional_guidance_scale=5.0, unconditional_conditioning=uc)
    assert report('DDIM encode (inversion) 12 steps vs reference', relerr(enc.cpu(), g['encode_cfg']), STOL) < STOL


def test_sampling_is_bitwise_reproducible():
    """No atomics anywhere on the path: the same sample() call twice gives identical latents, also after other work
    (a different batch size) ran in between and with the context cache cold or warm."""
    cfg = gi.SMALL_CFG
    model = models.ControlLDM(cfg, n_controlnets=1)
    try:
        sd = {k: synth.make_tensor(k, s) for k, s in model.engine.param_shapes().items()}
        assert not model.load_state_dict(sd)[0]
        B, H = 2, 16
        x_T = torch.from_numpy(synth.latents(B, H, H, seed=51)).cuda()
        c, uc = torch.from_numpy(synth.context(B, seed=52)).cuda(), torch.from_numpy(synth.context(B, seed=53)).cuda()
        hint = torch.from_numpy(synth.hint(B, 8 * H, seed=54)).cuda()
        cond = {'c_concat': [hint], 'c_crossattn': [c]}
        ucond = {'c_concat': [hint], 'c_crossattn': [uc]}

        def run(xT, cd, ucd, n):
            return samplers.ControlDDIMSampler(model).sample(5 if n else 4, xT.shape[0], (4, H, H), cd, verbose=False, eta=0.0,
                                                             x_T=xT, unconditional_guidance_scale=9.0,
                                                             unconditional_conditioning=ucd)[0].clone()
        a = run(x_T, cond, ucond, 0)
        run(x_T[:1], {'c_concat': [hint[:1]], 'c_crossattn': [c[:1]]}, {'c_concat': [hint[:1]], 'c_crossattn': [uc[:1]]}, 1)
        b = run(x_T, cond, ucond, 0)
        assert torch.equal(a, b)
    finally:
        model.engine.close()


def test_sampler_branches_on_device(cpu_noise):
    """The reference sampler branches no script exercises, with the HIP update kernels (tests/sampler_branches.py)."""
    import sampler_branches

    def make_model(hint=None):
        if hint is None:
            return AnalyticLDM('cuda')

        class M(AnalyticLDM):
            def apply_model(self, x, t, cc, **kw):
                self.calls += 1
                if cc['c_crossattn'][0].shape[0] == 2 * hint.shape[0]:
                    cc = {'c_concat': [torch.cat([hint, hint])], 'c_crossattn': cc['c_crossattn']}
                return analytic_eps(x, t, cc)
        return M('cuda')
    for k, v in sampler_branches.run(make_model, 'cuda', STOL).items():
        report(f'sampler branch {k} (HIP kernels) vs reference', v, STOL)

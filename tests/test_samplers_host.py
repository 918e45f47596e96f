"""Host logic of the drop-in samplers / model mirrors on CPU: control flow, schedules, call counts, kwargs
semantics -- against the golden trajectories produced by the REFERENCE samplers (tests/golden/samplers.npz).
The fused update kernels are replaced by torch-CPU stubs (tests/kernel_stubs.py) for these tests only; the
same trajectories are re-checked with the real HIP kernels in test_gpu_samplers.py."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
import kernel_stubs
from common import gold, relerr
from fgdm_amd import models, samplers
from test_oracle_golden import analytic_eps

STOL = 1e-5


class _DummyEngine:
    def __init__(self, device):
        self.device = torch.device(device)


class AnalyticLDM(models.LatentDiffusion):
    """LatentDiffusion mirror whose network is the closed-form eps of the goldens."""

    def __init__(self, device='cpu'):
        super().__init__(engine=_DummyEngine(device))
        self.calls = 0

    def apply_model(self, x, t, c, **kw):
        self.calls += 1
        return analytic_eps(x, t, c)


@pytest.fixture()
def stubs(monkeypatch):
    monkeypatch.setattr(samplers, '_k', kernel_stubs)
    monkeypatch.setattr(models, '_k', kernel_stubs)
    monkeypatch.setattr(samplers, '_randn', lambda shape, device: torch.randn(shape))
    monkeypatch.setattr(models, '_randn', lambda shape, device: torch.randn(shape))


def test_make_schedule_attributes_match_reference(stubs):
    g = gold('schedule')
    s = samplers.DDIMSampler(AnalyticLDM())
    for S in (20, 50):
        for eta in (0, 1):
            s.make_schedule(S, ddim_eta=float(eta), verbose=False)
            tag = f'S{S}_eta{eta}'
            np.testing.assert_array_equal(s.ddim_timesteps, g[f'ts_{tag}'])
            np.testing.assert_array_equal(s.ddim_alphas, g[f'alphas_{tag}'])
            np.testing.assert_array_equal(s.ddim_alphas_prev, g[f'alphas_prev_{tag}'])
            np.testing.assert_allclose(np.asarray(s.ddim_sigmas, dtype=np.float32), g[f'sigmas_{tag}'], rtol=1e-6)
            np.testing.assert_allclose(s.ddim_sqrt_one_minus_alphas, g[f'sqrt1m_{tag}'], rtol=1e-6)
    np.testing.assert_array_equal(s.alphas_cumprod.numpy(), g['alphas_cumprod'])
    gd = gold('ddpm_schedule')
    m = AnalyticLDM()
    for k in gd.files:
        np.testing.assert_array_equal(getattr(m, k).numpy(), gd[k], err_msg=k)
    assert m.num_timesteps == 1000


def test_ddim_sampler_trajectories(stubs):
    g = gold('samplers')
    x_T, c, uc = gi.get('samp/x_T'), gi.get('samp/c'), gi.get('samp/uc')
    for S, scale, eta in ((50, 7.5, 0.0), (20, 9.0, 0.0), (20, 7.5, 1.0), (10, 1.0, 0.0)):
        m = AnalyticLDM()
        smp = samplers.DDIMSampler(m)
        torch.manual_seed(123)
        out, inter = smp.sample(S, 2, (4, 8, 8), conditioning=c, x_T=x_T, eta=eta, verbose=False,
                                unconditional_guidance_scale=scale, unconditional_conditioning=uc, log_every_t=5)
        tag = f'ddim_S{S}_s{scale}_eta{eta}'
        assert relerr(out, g[tag]) < STOL, tag
        assert relerr(torch.stack(inter['x_inter']), g[tag + '_xinter']) < STOL
        assert relerr(torch.stack(inter['pred_x0']), g[tag + '_predx0']) < STOL
        # the reference makes one 2B call per step under CFG, one B call otherwise: same count here
        assert m.calls == int(g[tag + '_calls'][0])


def test_plms_sampler(stubs):
    g = gold('samplers')
    x_T, c, uc = gi.get('samp/x_T'), gi.get('samp/c'), gi.get('samp/uc')
    m = AnalyticLDM()
    smp = samplers.PLMSSampler(m)
    out, inter = smp.sample(50, 2, (4, 8, 8), conditioning=c, x_T=x_T, eta=0.0, verbose=False,
                            unconditional_guidance_scale=7.5, unconditional_conditioning=uc, log_every_t=5)
    assert relerr(out, g['plms_S50']) < STOL
    assert relerr(torch.stack(inter['x_inter']), g['plms_S50_xinter']) < STOL
    assert m.calls == 51
    with pytest.raises(ValueError):
        smp.sample(10, 2, (4, 8, 8), conditioning=c, x_T=x_T, eta=0.5, verbose=False)


def test_controlnet_sampler_dict_conds(stubs):
    g = gold('samplers')
    x_T, c, uc = gi.get('samp/x_T'), gi.get('samp/c'), gi.get('samp/uc')
    hint = gi.hint(2, 64, 48)
    cond = {'c_concat': [hint], 'c_crossattn': [c]}
    ucond = {'c_concat': [hint], 'c_crossattn': [uc]}

    class M(AnalyticLDM):
        def apply_model(self, x, t, cc, **kw):
            self.calls += 1
            if cc['c_crossattn'][0].shape[0] == 2 * hint.shape[0]:      # batched CFG: hint serves both halves
                cc = {'c_concat': [torch.cat([hint, hint])], 'c_crossattn': cc['c_crossattn']}
            return analytic_eps(x, t, cc)
    m = M()
    smp = samplers.ControlDDIMSampler(m)
    out, _ = smp.sample(20, 2, (4, 8, 8), cond, verbose=False, eta=0.0, x_T=x_T,
                        unconditional_guidance_scale=9.0, unconditional_conditioning=ucond)
    assert relerr(out, g['cn_ddim_S20']) < STOL
    assert m.calls == 20            # one 2B call per step instead of the reference's 40 B-sized calls
    # different hints on the two branches (guess mode): falls back to two calls per step
    m2 = M()
    smp = samplers.ControlDDIMSampler(m2)
    ucond2 = {'c_concat': None, 'c_crossattn': [uc]}
    smp.sample(4, 2, (4, 8, 8), cond, verbose=False, x_T=x_T, unconditional_guidance_scale=9.0,
               unconditional_conditioning=ucond2)
    assert m2.calls == 8
    with pytest.raises(NotImplementedError):
        smp.sample(4, 2, (4, 8, 8), cond, verbose=False, x_T=x_T, dynamic_threshold=0.9)


def test_mask_blend_callbacks_and_ancestral(stubs):
    g = gold('samplers')
    x_T, c, uc = gi.get('samp/x_T'), gi.get('samp/c'), gi.get('samp/uc')
    m = AnalyticLDM()
    smp = samplers.DDIMSampler(m)
    seen, imgs = [], []
    torch.manual_seed(321)
    out, _ = smp.sample(10, 2, (4, 8, 8), conditioning=c, x_T=x_T, eta=0.0, verbose=False, mask=gi.get('samp/mask'),
                        x0=gi.get('samp/x0'), unconditional_guidance_scale=7.5, unconditional_conditioning=uc,
                        callback=seen.append, img_callback=lambda p, i: imgs.append(i))
    assert relerr(out, g['ddim_mask_S10']) < STOL
    assert seen == list(range(10)) and imgs == list(range(10))
    m = AnalyticLDM()
    m.log_every_t = 4
    torch.manual_seed(99)
    img, inter = m.p_sample_loop(c, (2, 4, 8, 8), return_intermediates=True, x_T=x_T, verbose=False, timesteps=12)
    assert relerr(img, g['ancestral_T12']) < STOL
    assert relerr(torch.stack(inter), g['ancestral_T12_inter']) < STOL
    assert m.calls == 12


def test_batch_size_warning_and_unsupported_paths(stubs, capsys):
    x_T, c = gi.get('samp/x_T'), gi.get('samp/c')
    smp = samplers.DDIMSampler(AnalyticLDM())
    smp.sample(2, 3, (4, 8, 8), conditioning=c[:1], x_T=torch.cat([x_T, x_T[:1]]), verbose=False)
    assert 'Warning: Got 1 conditionings but batch-size is 3' in capsys.readouterr().out
    with pytest.raises(NotImplementedError):
        smp.sample(2, 2, (4, 8, 8), conditioning=c, x_T=x_T, verbose=False, inference_loss=True)


def test_dropin_import_paths():
    import fgdm_amd.dropin as dropin
    dropin.install()
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.plms import PLMSSampler
    from ldm.models.diffusion.ddpm import LatentDiffusion
    from controlnet.cldm.ddim_hacked import DDIMSampler as CNSampler
    from controlnet.cldm.cldm import ControlLDM
    assert DDIMSampler is samplers.DDIMSampler and PLMSSampler is samplers.PLMSSampler
    assert CNSampler is samplers.ControlDDIMSampler and LatentDiffusion is models.LatentDiffusion
    assert ControlLDM is models.ControlLDM


def test_dpm_solver_and_ddim_encode(stubs):
    """DPM-Solver++ 2M (fractional timesteps, lower_order_final below 15 steps) and DDIM inversion vs the reference."""
    g = gold('samplers2')
    x_T, c, uc = gi.get('samp/x_T'), gi.get('samp/c'), gi.get('samp/uc')
    for S, scale in ((20, 7.5), (10, 7.5), (12, 1.0)):
        m = AnalyticLDM()
        out, _ = samplers.DPMSolverSampler(m).sample(S, 2, (4, 8, 8), conditioning=c, x_T=x_T, verbose=False,
                                                     unconditional_guidance_scale=scale, unconditional_conditioning=uc)
        assert relerr(out, g[f'dpm_S{S}_s{scale}']) < 2e-5, (S, scale)
        assert m.calls == int(g[f'dpm_S{S}_s{scale}_calls'][0]) == S
    m = AnalyticLDM()
    smp = samplers.ControlDDIMSampler(m)
    smp.make_schedule(20, ddim_eta=0.0, verbose=False)
    x0 = gi.get('samp/x0')
    enc, info = smp.encode(x0, c, 12, unconditional_guidance_scale=5.0, unconditional_conditioning=uc,
                           return_intermediates=3)
    assert relerr(enc, g['encode_cfg']) < STOL
    assert relerr(torch.stack(info['intermediates']), g['encode_cfg_inter']) < STOL
    assert list(info['intermediate_steps']) == list(g['encode_cfg_steps'])
    hint = gi.hint(2, 64, 48)
    enc, _ = smp.encode(x0, {'c_concat': [hint], 'c_crossattn': [c]}, 15)
    assert relerr(enc, g['encode_plain']) < STOL


def test_batched_conditioning_is_built_once_per_sample(stubs):
    """cat([uc, c]) is loop-invariant: every step must hand the SAME tensor object to apply_model (the engine keeps the
    context's K/V projections keyed on that object), and an in-place change of a source tensor must invalidate it."""
    x_T, c, uc = gi.get('samp/x_T'), gi.get('samp/c'), gi.get('samp/uc')
    seen = []

    class M(AnalyticLDM):
        def apply_model(self, x, t, cc, **kw):
            seen.append(cc)
            return analytic_eps(x, t, cc)
    smp = samplers.DDIMSampler(M())
    smp.sample(S=5, batch_size=2, shape=[4, 8, 8], conditioning=c, verbose=False, unconditional_guidance_scale=7.5,
               unconditional_conditioning=uc, x_T=x_T)
    assert len(seen) == 5 and all(s is seen[0] for s in seen)
    assert torch.equal(seen[0], torch.cat([uc, c]))
    first = seen[0]
    c.mul_(1.0)                       # in-place write: same values, new _version
    smp.sample(S=2, batch_size=2, shape=[4, 8, 8], conditioning=c, verbose=False, unconditional_guidance_scale=7.5,
               unconditional_conditioning=uc, x_T=x_T)
    assert seen[5] is not first and seen[6] is seen[5]
    # dict conditionings (ControlNet sampler): the concatenated cross-attention tensor is reused as well
    hint = gi.hint(2, 64, 48)
    seen.clear()

    class MC(AnalyticLDM):
        def apply_model(self, x, t, cc, **kw):
            seen.append(cc['c_crossattn'][0])
            return analytic_eps(x, t, {'c_concat': [torch.cat([hint, hint])], 'c_crossattn': cc['c_crossattn']})
    smp = samplers.ControlDDIMSampler(MC())
    smp.sample(4, 2, (4, 8, 8), {'c_concat': [hint], 'c_crossattn': [c]}, verbose=False, x_T=x_T,
               unconditional_guidance_scale=9.0, unconditional_conditioning={'c_concat': [hint], 'c_crossattn': [uc]})
    assert len(seen) == 4 and all(s is seen[0] for s in seen)


def test_sampler_branches_vs_reference(stubs):
    """composable_diffusion, augmented_conditoning, score_corrector, timesteps=, decode, use_original_steps,
    stochastic_encode, ucg_schedule: written in round 1, first exercised here."""
    import sampler_branches
    from test_oracle_golden import analytic_eps as ae

    def make_model(hint=None):
        if hint is None:
            return AnalyticLDM()

        class M(AnalyticLDM):
            def apply_model(self, x, t, cc, **kw):
                self.calls += 1
                if cc['c_crossattn'][0].shape[0] == 2 * hint.shape[0]:      # batched CFG: the hint serves both halves
                    cc = {'c_concat': [torch.cat([hint, hint])], 'c_crossattn': cc['c_crossattn']}
                return ae(x, t, cc)
        return M()
    err = sampler_branches.run(make_model, 'cpu', STOL)
    print(err)


def test_context_cache_policy_registers_once_under_alternation():
    """guess mode alternates two contexts (ddim_hacked.py:190-191); the engine must not re-register (free + ~50
    hipMallocs + sync) every third call.  One of the two stays registered, the other is passed along."""
    from fgdm_amd.engine import ContextCachePolicy
    a, b, c = torch.zeros(2, 77, 8), torch.zeros(2, 77, 8), torch.zeros(2, 77, 8)
    pol = ContextCachePolicy()
    seq = [pol.see(t) for t in (a, b) * 10]
    assert pol.registrations == 2 and seq[:2] == ['register', 'register'] and set(seq[2:]) == {'hit', 'bypass'}
    assert seq[2::2] == ['bypass'] * 9 and seq[3::2] == ['hit'] * 9
    # the ordinary case: the same object every step, a new one per sample() call
    pol = ContextCachePolicy()
    assert [pol.see(a) for _ in range(5)] == ['register'] + ['hit'] * 4
    assert [pol.see(c) for _ in range(3)] == ['register', 'hit', 'hit']
    # an in-place write invalidates (torch bumps _version)
    c.add_(1.0)
    assert pol.see(c) == 'register' and pol.see(c) == 'hit'
    assert pol.registrations == 3

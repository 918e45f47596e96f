"""The metric's own workload at full size (BASELINE configs[2]): SD-v1.5-width UNet + ControlNet, latent 64x64, hint 512x512,
against fixtures produced by the REFERENCE's modules at that size (tests/golden/full_size.npz = fp32 CPU path,
full_size_ac.npz = the same modules under the reference's CUDA-autocast policy; tools/make_goldens.py g_full_size).

  * one classifier-free-guidance pair through fgdm_apply_model (with and without FGDM_FLAG_CFG_PAIRS: bit-identical);
  * a complete 50-step eta = 0 DDIM sampling, CFG 9.0, through the drop-in ControlLDM + ControlDDIMSampler mirrors, the
    error recorded per step (gpurun_out/parity_50step.json; a copy is committed under profiles/).
Tolerance (tests/common.py: check_net): max(1e-3, NET_K x floor) with NET_K = 1.09, floor = |reference autocast - reference fp32| at the same
point of the same trajectory."""
import json
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi
from common import CAP_CHAIN, check_net, gold, net_tol, relerr, report
from fgdm_amd import synth

pytestmark = pytest.mark.gpu


def _inputs():
    x = torch.from_numpy(synth.latents(1, 64, 64, seed=42))
    c = torch.from_numpy(synth.context(1, seed=43))
    uc = torch.from_numpy(synth.context(1, seed=44))
    hint = torch.from_numpy(synth.hint(1, 512, seed=45))
    return x, c, uc, hint


@pytest.fixture(scope='module')
def model():
    from fgdm_amd import models
    m = models.ControlLDM(gi.SD_CFG, n_controlnets=1)
    sd = {k: synth.make_tensor(k, s) for k, s in m.engine.param_shapes().items()}
    assert not m.load_state_dict(sd)[0]
    del sd
    yield m
    m.engine.close()


def test_cfg_pair_at_full_size_vs_reference(model):
    from fgdm_amd import _lib
    g, ga = gold('full_size'), gold('full_size_ac')
    x, c, uc, hint = _inputs()
    e = model.engine
    e.set_hint(0, hint.cuda())
    xx, cc = torch.cat([x, x]).cuda(), torch.cat([uc, c]).cuda()
    for tv in (981, 21):
        t = torch.full((2,), tv, dtype=torch.long).cuda()
        plain = e.apply_model(xx, t, cc).clone()
        pairs = e.apply_model(xx, t, cc, flags=_lib.FLAG_CFG_PAIRS)
        assert torch.equal(plain, pairs)
        check_net(f'full-size SD UNet + ControlNet, CFG pair, t={tv}', plain.cpu(), g[f'eps_pair_t{tv}'], ga[f'eps_pair_t{tv}'])
        # classifier-free guidance amplifies the difference of the two halves by the scale: check the combined eps too
        comb = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32))[:1] + 9.0 * (torch.as_tensor(np.asarray(a, dtype=np.float32))[1:] - torch.as_tensor(np.asarray(a, dtype=np.float32))[:1])
        check_net(f'full-size CFG-combined eps (scale 9), t={tv}', comb(plain.cpu()), comb(g[f'eps_pair_t{tv}']), comb(ga[f'eps_pair_t{tv}']), cap=CAP_CHAIN)


def test_50_step_sampling_at_full_size_vs_reference(model):
    from fgdm_amd import samplers
    g, ga = gold('full_size'), gold('full_size_ac')
    x, c, uc, hint = _inputs()
    hint_d = hint.cuda()
    cond = {'c_concat': [hint_d], 'c_crossattn': [c.cuda()]}
    ucond = {'c_concat': [hint_d], 'c_crossattn': [uc.cuda()]}
    model.control_scales = [1.0] * 13
    out, inter = samplers.ControlDDIMSampler(model).sample(50, 1, (4, 64, 64), cond, verbose=False, eta=0.0, x_T=x.cuda(),
                                                           unconditional_guidance_scale=9.0, log_every_t=1,
                                                           unconditional_conditioning=ucond)
    xs = [v.cpu() for v in inter['x_inter'][1:]]
    ps = [v.cpu() for v in inter['pred_x0'][1:]]
    assert len(xs) == 50
    rec = {'steps': [], 'kept': {}}
    s32, sac = g['traj_sums'], ga['traj_sums']
    for i in range(50):
        # (sum, L2) of every step's latent: engine next to the two reference trajectories
        rec['steps'].append({'step': i + 1, 'x_l2': float(xs[i].double().norm()), 'x_l2_ref_fp32': float(s32[i, 1]),
                             'x_l2_ref_autocast': float(sac[i, 1]), 'x_sum': float(xs[i].double().sum()),
                             'x_sum_ref_fp32': float(s32[i, 0]), 'x_sum_ref_autocast': float(sac[i, 0]),
                             'pred_x0_l2': float(ps[i].double().norm()), 'pred_x0_l2_ref_fp32': float(s32[i, 3])})
        assert abs(rec['steps'][-1]['x_l2'] / s32[i, 1] - 1.0) < 2e-3, i
    worst = 0.0
    for k in (1, 2, 5, 10, 20, 30, 40, 50):
        e32, floor = check_net(f'50-step full-size sampling, latent after step {k}', xs[k - 1], g[f'x_step{k}'], ga[f'x_step{k}'], cap=CAP_CHAIN)
        rec['kept'][k] = {'err_vs_ref_fp32': e32, 'floor_ref_autocast_vs_ref_fp32': floor,
                          'err_vs_ref_autocast': relerr(xs[k - 1], ga[f'x_step{k}'].astype(np.float32))}
        worst = max(worst, e32 / max(floor, 1e-12))
    e32, floor = check_net('50-step full-size sampling, final latent', out.cpu(), g['out'], ga['out'], cap=CAP_CHAIN)
    rec['final'] = {'err_vs_ref_fp32': e32, 'floor_ref_autocast_vs_ref_fp32': floor, 'worst_err_over_floor': worst}
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, 'parity_50step.json'), 'w') as f:
            json.dump(rec, f, indent=1)

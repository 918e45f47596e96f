"""Shared body of the sampler-branch tests (CPU with kernel stubs: tests/test_samplers_host.py; GPU with the HIP kernels:
tests/test_gpu_samplers.py) against tests/golden/samplers3.npz (reference samplers, analytic model; tools/make_goldens.py
g_samplers3)."""
import numpy as np
import torch

import golden_inputs as gi
from common import gold, relerr
from fgdm_amd import samplers


class Corrector:
    """same score_corrector as tools/make_goldens.py"""

    def modify_score(self, model, e_t, x, t, c, gain=1.0):
        return e_t + gain * 0.1 * torch.tanh(x) * (t.float() / 1000.0).reshape(-1, 1, 1, 1)


def run(make_model, dev, tol):
    g = gold('samplers3')
    d = lambda v: v.to(dev)
    shape = (4, 8, 8)
    x_T, c, uc = d(gi.get('samp/x_T')), d(gi.get('samp/c')), d(gi.get('samp/uc'))
    err = {}
    m = make_model()
    out, _ = samplers.DDIMSampler(m).sample(10, 1, shape, conditioning=c, x_T=x_T[:1], eta=0.0, verbose=False,
                                            unconditional_guidance_scale=7.5, unconditional_conditioning=uc[:1],
                                            composable_diffusion=2)
    err['composable_diffusion'] = relerr(out.cpu(), g['compose'])
    assert m.calls == int(g['compose_calls'][0])
    m = make_model()
    out, _ = samplers.DDIMSampler(m).sample(10, 2, shape, conditioning=c, x_T=x_T, eta=0.0, verbose=False,
                                            unconditional_guidance_scale=3.0, unconditional_conditioning=uc,
                                            augmented_conditoning=True, ac=d(gi.get('samp/ac')))
    err['augmented_conditoning'] = relerr(out.cpu(), g['augmented'])
    assert m.calls == int(g['augmented_calls'][0])
    out, _ = samplers.DDIMSampler(make_model()).sample(10, 2, shape, conditioning=c, x_T=x_T, eta=0.0, verbose=False,
                                                       unconditional_guidance_scale=7.5, unconditional_conditioning=uc,
                                                       score_corrector=Corrector(), corrector_kwargs={'gain': 0.5})
    err['score_corrector'] = relerr(out.cpu(), g['corrector'])
    m = make_model()
    smp = samplers.DDIMSampler(m)
    smp.make_schedule(20, ddim_eta=0.0, verbose=False)
    out, inter = smp.ddim_sampling(c, (2,) + shape, x_T=x_T, timesteps=10, unconditional_guidance_scale=7.5,
                                   unconditional_conditioning=uc, log_every_t=1)
    err['timesteps= truncation'] = relerr(out.cpu(), g['truncated'])
    assert m.calls == int(g['truncated_n'][0]) == 9
    x_lat = d(gi.get('samp/x0'))
    smp = samplers.DDIMSampler(make_model())
    smp.make_schedule(20, ddim_eta=0.0, verbose=False)
    err['decode (ddim steps)'] = relerr(smp.decode(x_lat, c, 12, unconditional_guidance_scale=5.0, unconditional_conditioning=uc).cpu(),
                                        g['decode_ddim12'])
    err['decode (use_original_steps)'] = relerr(smp.decode(x_lat, c, 15, unconditional_guidance_scale=5.0, unconditional_conditioning=uc,
                                                           use_original_steps=True).cpu(), g['decode_orig15'])
    noise = d(gi.get('samp/noise'))
    err['stochastic_encode'] = relerr(smp.stochastic_encode(x_lat, torch.tensor([7, 7]).to(dev), noise=noise).cpu(), g['stoch_ddim7'])
    err['stochastic_encode (use_original_steps)'] = relerr(
        smp.stochastic_encode(x_lat, torch.tensor([300, 300]).to(dev), use_original_steps=True, noise=noise).cpu(), g['stoch_orig300'])
    hint = d(gi.hint(2, 64, 48))
    out, _ = samplers.ControlDDIMSampler(make_model(hint)).sample(
        10, 2, shape, {'c_concat': [hint], 'c_crossattn': [c]}, verbose=False, eta=0.0, x_T=x_T, unconditional_guidance_scale=9.0,
        ucg_schedule=[float(v) for v in g['cn_ucg_schedule']], unconditional_conditioning={'c_concat': [hint], 'c_crossattn': [uc]})
    err['ucg_schedule'] = relerr(out.cpu(), g['cn_ucg'])
    for k, v in err.items():
        assert v < tol, (k, v)
    return err

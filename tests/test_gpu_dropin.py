"""Script-level drop-in on the device: the exact call sequence of scripts/txt2img_fgdm_inference.py:23-38,179-185,216-245 and of
controlnet/seg2image_inference.py:36-94, written against the REFERENCE's import paths (fgdm_amd.dropin.install()), with the
YAML contents of models/config.yaml / cldm_v15_canny.yaml as literals, synthetic checkpoints and a stand-in tokenizer (the
BPE vocabulary needs the network)."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from fgdm_amd import synth
from test_dropin_configs import CLDM_YAML, FGDM_YAML

pytestmark = pytest.mark.gpu


def _tokenizer(prompts):
    ids = gi.clip_ids(8, seed=11)
    return torch.stack([ids[(len(p) + sum(map(ord, p))) % 8] for p in prompts])


def test_txt2img_fgdm_inference_call_sequence():
    import fgdm_amd.dropin as dropin
    dropin.install()
    from ldm.util import instantiate_from_config
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.plms import PLMSSampler

    # load_model_from_config (scripts/txt2img_fgdm_inference.py:23-38)
    model = instantiate_from_config(FGDM_YAML['model'])
    sd = {k: synth.make_tensor(k, s) for k, s in model.engine.param_shapes().items()}
    sd['model_ema.decay'] = np.zeros((), np.float32)            # checkpoints carry keys the sampling path does not use
    m, u = model.load_state_dict(sd, strict=False)
    assert len(m) == 0 and u == ['model_ema.decay']
    model.cuda()
    model.eval()
    model.tokenizer = _tokenizer
    try:
        n_samples, H, W, C, f, scale, steps = 2, 256, 256, 4, 8, 7.5, 4     # (S = 3 fails in the reference too: timestep 1000)
        for sampler in (DDIMSampler(model), PLMSSampler(model)):            # :182-185
            with torch.no_grad():
                with model.ema_scope():
                    uc = model.get_learned_conditioning(n_samples * [''])
                    c = model.get_learned_conditioning(n_samples * ['a bedroom with a large window'])
                    shape = [C, H // f, W // f]
                    torch.manual_seed(7)
                    samples_ddim, _ = sampler.sample(S=steps, conditioning=c, batch_size=n_samples, shape=shape, num=1, x_T=None,
                                                     verbose=False, unconditional_guidance_scale=scale,
                                                     unconditional_conditioning=uc, eta=0.0)
                    x = model.decode_first_stage(samples_ddim)
                    x = torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0)
            assert tuple(samples_ddim.shape) == (n_samples, C, H // f, W // f) and tuple(x.shape) == (n_samples, 3, H, W)
            assert torch.isfinite(x).all() and float(x.std()) > 1e-3
            img = (255. * x[0].permute(1, 2, 0).cpu().numpy()).astype(np.uint8)          # :249-252
            assert img.shape == (H, W, 3)
        # the same seed reproduces the same latents bit for bit (x_T=None draws from torch's generator, ddim.py:126-129)
        torch.manual_seed(7)
        again, _ = DDIMSampler(model).sample(S=steps, conditioning=c, batch_size=n_samples, shape=shape, num=1, x_T=None,
                                             verbose=False, unconditional_guidance_scale=scale, unconditional_conditioning=uc,
                                             eta=0.0)
        torch.manual_seed(7)
        first, _ = DDIMSampler(model).sample(S=steps, conditioning=c, batch_size=n_samples, shape=shape, num=1, x_T=None,
                                             verbose=False, unconditional_guidance_scale=scale, unconditional_conditioning=uc,
                                             eta=0.0)
        assert torch.equal(again, first)
    finally:
        model.engine.close()


def test_seg2image_inference_call_sequence():
    import fgdm_amd.dropin as dropin
    dropin.install()
    from cldm.model import create_model
    from cldm.ddim_hacked import DDIMSampler
    import controlnet.seg2image_inference as s2i

    model = create_model(CLDM_YAML).cpu()                       # controlnet/seg2image_inference.py:37 (a YAML path there)
    sd = {k: torch.from_numpy(synth.make_tensor(k, s)) for k, s in model.engine.param_shapes().items()}
    model.load_state_dict(sd)                                   # :38
    model = model.cuda()
    ddim_sampler = DDIMSampler(model)                           # :40
    model.tokenizer = _tokenizer
    s2i.setup(model, ddim_sampler)
    try:
        seg = (synth.hint(1, 256, seed=91)[0].transpose(1, 2, 0) * 255).astype(np.uint8)      # a 256x256 "segmentation map"
        inp = np.zeros((300, 300, 3), np.uint8)                 # input_image only fixes the working resolution (:45-47)
        x_T = torch.from_numpy(synth.latents(2, 32, 32, seed=92)).cuda()
        res = s2i.process(inp, 'a bedroom', 'best quality, extremely detailed', 'lowres, bad anatomy', 2, 256, 256, 4, False, 1.0,
                          9.0, 12345, 0.0, spath=seg, x_T=x_T)
        assert len(res) == 3 and all(r.dtype == np.uint8 and r.shape == (256, 256, 3) for r in res)
        assert np.array_equal(res[0], seg)                      # [detected_map] + results
        assert res[1].std() > 1.0 and not np.array_equal(res[1], res[2])
        # guess_mode: unconditional branch without control, geometric control scales (:78,81)
        res_g = s2i.process(inp, 'a bedroom', 'best quality', 'lowres', 1, 256, 256, 2, True, 1.0, 9.0, 12345, 0.0, spath=seg,
                            x_T=x_T[:1])
        assert len(res_g) == 2 and abs(model.control_scales[0] - 0.825 ** 12) < 1e-6
        with pytest.raises(FileNotFoundError):
            s2i.process(inp, 'a', 'b', 'c', 1, 256, 256, 2, False, 1.0, 9.0, 1, 0.0, spath='/nonexistent', index=3)
    finally:
        model.engine.close()

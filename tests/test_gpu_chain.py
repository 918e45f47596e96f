"""The two-factor FG-DM chain end to end on the device (scripts/txt2img_fgdm_inference.py:231-292):

  stage A  FG-DM UNet (self-prompt adapter) @32x32 latent, DDIM + CFG 7.5 -> decode_first_stage -> 256x256 condition image
  boundary uint8 truncation -> cv2-style bilinear 2x -> /255                -> 512x512 hint
  stage B  initialize_cn.process: SD-v1.5 UNet + ControlNet @64x64, DDIM + CFG 9.0 -> decode -> uint8 image

against the CPU oracle doing the same steps.  Two DDIM steps per stage keep the oracle within a minute.  The uint8
truncation turns the ~2e-3 fp16 deviation of a decoded image into occasional one-LSB differences, so bytes are compared
by counting: hint bytes equal or one LSB apart; final bytes on average well under one LSB apart."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from common import CAP_CHAIN, relerr, report
from fgdm_amd import boundary, initialize_cn, models, samplers, synth

pytestmark = pytest.mark.gpu


def _ctx(prompts, seed0):
    """Stand-in text encoder (CLIP is outside the path): a deterministic context per prompt string."""
    rows = [synth.context(1, seed=seed0 + sum(map(ord, p)) % 1000) for p in prompts]
    return torch.from_numpy(np.concatenate(rows))


def test_two_stage_chain_vs_oracle():
    from oracle import arch, boundary as ob, nn as onn, samplers as osamp, schedule, vae as ovae
    cfg = gi.SD_CFG
    S = 2
    # ---------------- stage A on the engine
    a = models.LatentDiffusion(cfg, use_adapter=True, first_stage_config=True)
    sd_a = {k: synth.make_tensor(k, s) for k, s in a.engine.param_shapes().items()}
    assert not a.load_state_dict(sd_a)[0]
    a.cond_stage_model = lambda prompts: _ctx(prompts, 100).cuda()
    prompt = 'a bedroom with a large window'
    c, uc = a.get_learned_conditioning([prompt]), a.get_learned_conditioning([''])
    xT_a = torch.from_numpy(synth.latents(1, 32, 32, seed=21))
    z_a, _ = samplers.DDIMSampler(a).sample(S=S, conditioning=c, batch_size=1, shape=[4, 32, 32], verbose=False,
                                            unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0,
                                            x_T=xT_a.cuda())
    img_a = a.decode_first_stage(z_a)
    hint, u8_a = boundary.hint_from_image(img_a, 512)
    big_u8 = boundary.resize_linear_uint8(u8_a, 512, 512)
    a.engine.close()
    # ---------------- stage B on the engine, through the initialize_cn.process mirror
    from fgdm_amd import engine as eng
    shapes_b = eng.param_shapes(eng.make_config(initialize_cn.CLDM_V15, n_controlnets=1, vae=True, clip=True))
    sd_b = {k: synth.make_tensor(k, s) for k, s in shapes_b.items()}
    model, sampler = initialize_cn.initialize_controlnet('seg', state_dict=sd_b)
    model.cond_stage_model = lambda prompts: _ctx(prompts, 200).cuda()
    xT_b = torch.from_numpy(synth.latents(1, 64, 64, seed=22))
    res = initialize_cn.process(model, sampler, big_u8, prompt, 'best quality', 'lowres', 1, 1, 512, eta=0.0,
                                detect_resolution=512, ddim_steps=S, guess_mode=False, strength=1.0, scale=9.0, seed=5,
                                x_T=xT_b.cuda())
    assert len(res) == 1 and res[0].dtype == np.uint8 and res[0].shape == (512, 512, 3)
    model.engine.close()

    # ---------------- the same chain on the CPU oracle
    sched = schedule.register_schedule()
    pa = {k: torch.from_numpy(v) for k, v in sd_a.items()}
    fn_a = lambda x, t, cc: onn.unet_forward(pa, cfg, x, t, cc, prefix='model.diffusion_model.', use_adapter=True)
    from common import check_net
    from oracle import precision
    with torch.no_grad():
        z_ref, _ = osamp.ddim_sample(fn_a, sched, S, xT_a.shape, c.cpu(), xT_a, scale=7.5, uc=uc.cpu())
        img_ref = ovae.decode_first_stage(pa, z_ref)
        with precision.mode('autocast'):     # the reference's own GPU numerics on the same chain: the floor
            z_ac, _ = osamp.ddim_sample(fn_a, sched, S, xT_a.shape, c.cpu(), xT_a, scale=7.5, uc=uc.cpu())
            img_ac = ovae.decode_first_stage(pa, z_ac)
    check_net('chain stage A latent (2 DDIM steps, adapter, CFG 7.5)', z_a.cpu(), z_ref, z_ac.float(), cap=CAP_CHAIN)
    check_net('chain stage A decoded 256x256 image', img_a.cpu(), img_ref, img_ac.float(), cap=CAP_CHAIN)
    u8_ref = ob.image_to_uint8(img_ref.numpy(), 0)
    big_ref = ob.resize_linear_u8(u8_ref, 512, 512)
    d = np.abs(big_u8.cpu().numpy().astype(np.int32) - big_ref.astype(np.int32))
    # a 4.7e-3 relative deviation of the decoded image is ~0.3 LSB after the x255 scaling, so truncation flips roughly
    # that fraction of the bytes by one step; anything beyond a couple of LSB would be a real error
    report('chain boundary: fraction of 512x512 hint bytes that differ (all by <= 2 LSB)', float((d > 0).mean()), 0.35)
    assert d.max() <= 2 and (d > 0).mean() < 0.35, (d.max(), (d > 0).mean())
    # stage B of the oracle starts from the ENGINE's hint so that the two stages are judged separately
    hint_cpu = hint.cpu()
    assert np.array_equal(hint_cpu.numpy(), ob.uint8_to_hint(big_u8.cpu().numpy()))
    pb = {k: torch.from_numpy(v) for k, v in sd_b.items()}
    cb = _ctx([prompt + ', best quality'], 200)
    ub = _ctx(['lowres'], 200)
    fn_b = lambda x, t, cc: onn.control_ldm_apply(pb, cfg, x, t, cc['c_crossattn'][0], [cc['c_concat'][0]], scales=[1.0] * 13)
    with torch.no_grad():
        zb_ref, _ = osamp.ddim_sample(fn_b, sched, S, xT_b.shape, {'c_concat': [hint_cpu], 'c_crossattn': [cb]}, xT_b,
                                      scale=9.0, uc={'c_concat': [hint_cpu], 'c_crossattn': [ub]}, cfg_mode='sequential')
        out_ref = ob.image_to_uint8(ovae.decode_first_stage(pb, zb_ref).numpy(), 1)
    d = np.abs(res[0].astype(np.int32) - out_ref[0].astype(np.int32))
    report('chain stage B final uint8 image: mean |diff| in LSB', float(d.mean()), 1.0)
    assert d.mean() < 1.0 and (d > 8).mean() < 0.01, (d.mean(), d.max())

"""Mirror of controlnet/seg2image_inference.py:43-94 -- the stand-alone ControlNet stage: `process(input_image, prompt, ...)`
with the reference's signature, over the module-level `model` / `ddim_sampler` pair the script creates at import time
(:36-40; here: `setup(model, sampler)` or `setup()` = create_model + checkpoint as the script does).

The condition map is read from `spath` (a .png, or `spath/<index:05d>.png`) exactly as the reference does; an ndarray may be
passed as `spath` instead (the FG-DM chain hands the map over in memory).  HWC3 / resize_image restate
controlnet/annotator/util.py:9-38; the bilinear resize of the map runs in the engine's cv2-compatible kernel
(fgdm_amd/boundary.py)."""
import os
import random

import numpy as np
import torch

from . import boundary

model = None
ddim_sampler = None


def setup(m=None, sampler=None, config_path='./models/cldm_v15_canny.yaml', ckpt='./models/control_sd15_seg.pth'):
    """seg2image_inference.py:36-40"""
    global model, ddim_sampler
    if m is None:
        from . import config, initialize_cn
        m = config.create_model(config_path).cpu()
        m.load_state_dict(initialize_cn.load_state_dict(ckpt, location='cuda'))
        m = m.cuda()
    if sampler is None:
        from . import samplers
        sampler = samplers.ControlDDIMSampler(m)
    model, ddim_sampler = m, sampler
    return model, ddim_sampler


def HWC3(x):
    """annotator/util.py:9-26"""
    assert x.dtype == np.uint8
    if x.ndim == 2:
        x = x[:, :, None]
    assert x.ndim == 3
    H, W, C = x.shape
    assert C in (1, 3, 4)
    if C == 3:
        return x
    if C == 1:
        return np.concatenate([x, x, x], axis=2)
    color = x[:, :, 0:3].astype(np.float32)
    alpha = x[:, :, 3:4].astype(np.float32) / 255.0
    y = color * alpha + 255.0 * (1.0 - alpha)
    return y.clip(0, 255).astype(np.uint8)


def resized_shape(H, W, resolution):
    """annotator/util.py:29-38 (resize_image): the target size, a multiple of 64"""
    k = float(resolution) / min(float(H), float(W))
    return int(np.round(H * k / 64.0)) * 64, int(np.round(W * k / 64.0)) * 64


def process(input_image, prompt, a_prompt, n_prompt, num_samples, image_resolution, detect_resolution, ddim_steps, guess_mode,
            strength, scale, seed, eta, class_map=None, spath='', index=0, x_T=None):
    """-> [detected_map] + results (uint8 HWC), as the reference.  `x_T` (start noise) is an addition for reproducible tests."""
    if model is None:
        raise RuntimeError('seg2image.setup(model, ddim_sampler) has not been called')
    with torch.no_grad():
        input_image = HWC3(np.asarray(input_image))
        H, W = resized_shape(*input_image.shape[:2], image_resolution)
        if isinstance(spath, np.ndarray):
            detected_map = spath
        else:
            path = spath if spath.endswith('.png') else os.path.join(spath, f'{index:05d}.png')
            if not os.path.exists(path):
                raise FileNotFoundError(path)       # the reference prints the path and exits
            from PIL import Image
            detected_map = np.array(Image.open(path))
        detected_map = HWC3(detected_map)
        dm = torch.from_numpy(np.ascontiguousarray(detected_map)).to(model.device)[None]
        if dm.shape[1:3] != (H, W):
            dm = boundary.resize_linear_uint8(dm, H, W)          # cv2.resize(detected_map, (W, H), INTER_LINEAR)
        detected_map = dm[0].cpu().numpy()
        control = boundary.uint8_to_hint(dm)                      # float / 255, b h w c -> b c h w
        control = control.expand(num_samples, -1, -1, -1).contiguous()
        if seed == -1:
            seed = random.randint(0, 65535)
        torch.manual_seed(seed)
        np.random.seed(seed)
        random.seed(seed)
        cond = {'c_concat': [control],
                'c_crossattn': [model.get_learned_conditioning([prompt + ', ' + a_prompt] * num_samples)]}
        un_cond = {'c_concat': None if guess_mode else [control],
                   'c_crossattn': [model.get_learned_conditioning([n_prompt] * num_samples)]}
        shape = (4, H // 8, W // 8)
        model.control_scales = ([strength * (0.825 ** float(12 - i)) for i in range(13)] if guess_mode else ([strength] * 13))
        samples, _ = ddim_sampler.sample(ddim_steps, num_samples, shape, cond, verbose=False, eta=eta, x_T=x_T,
                                         unconditional_guidance_scale=scale, unconditional_conditioning=un_cond, cond_mask=None)
        x_samples = model.decode_first_stage(samples)
        u8 = boundary.image_to_uint8(x_samples, 1).cpu().numpy()
        return [detected_map] + [u8[i] for i in range(num_samples)]

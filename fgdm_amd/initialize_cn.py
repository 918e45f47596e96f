"""Mirror of controlnet/initialize_cn.py (the ControlNet stage of the FG-DM chain) over the HIP engine.

  initialize_controlnet(cond)  <-  initialize_cn.py:25-43   (create_model(cldm_v15_canny.yaml) + checkpoint -> model, sampler)
  process(...)                 <-  initialize_cn.py:74-104  (hint from uint8 images, DDIM + CFG, decode, uint8 results)

`process` keeps the reference's signature and host-side types (uint8 numpy in, list of uint8 numpy out) and also accepts
a device tensor for `input_image` (uint8 NHWC) or a ready hint (fp32 NCHW), so that the two-stage chain can stay on the
GPU (fgdm_amd/boundary.py).  Text conditioning comes from `model.get_learned_conditioning` exactly as in the reference: the
model is built with its cond_stage_config, so the checkpoint's `cond_stage_model.transformer.text_model.*` tensors load into
the engine's text encoder (fgdm_clip_encode); only the BPE tokenizer stays on the host (`model.tokenizer`)."""
import os
import random

import numpy as np
import torch

from . import boundary, models, samplers

# controlnet/models/cldm_v15_canny.yaml: unet_config / control_stage_config parameters (SD-v1.5 widths)
CLDM_V15 = dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1), num_res_blocks=2,
                channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768, transformer_depth=1)
CHECKPOINTS = {'seg': './models/fgdm_control_sd15_seg.pth', 'depth': './models/fgdm_control_sd15_depth.pth',
               'normal': './models/fgdm_control_sd15_normal.pth', 'sketch': './models/fgdm_control_sd15_scribble.pth'}


def get_state_dict(d):
    return d.get('state_dict', d)


def load_state_dict(ckpt_path, location='cpu'):
    """controlnet/cldm/model.py:12-21"""
    _, extension = os.path.splitext(ckpt_path)
    if extension.lower() == '.safetensors':
        import safetensors.torch
        state_dict = safetensors.torch.load_file(ckpt_path, device=location)
    else:
        state_dict = get_state_dict(torch.load(ckpt_path, map_location=torch.device(location)))
    state_dict = get_state_dict(state_dict)
    print(f'Loaded state_dict from [{ckpt_path}]')
    return state_dict


def initialize_controlnet(cond='seg', state_dict=None, device=0, cond_stage_config=True):
    """-> (ControlLDM mirror, ControlNet DDIMSampler mirror).  `state_dict` overrides reading the checkpoint file;
    cond_stage_config=None builds the model without the text encoder (then plug `model.cond_stage_model`)."""
    if cond not in CHECKPOINTS:
        raise NotImplementedError
    model = models.ControlLDM(CLDM_V15, n_controlnets=1, device=device, first_stage_config=True,
                              cond_stage_config=cond_stage_config)
    sd = state_dict if state_dict is not None else load_state_dict(CHECKPOINTS[cond], location='cpu')
    m, u = model.load_state_dict(sd, strict=False)
    print('Missing keys: ', m)
    return model, samplers.ControlDDIMSampler(model)


def _control_from(input_image, device):
    """initialize_cn.py:76-80: uint8 [B,H,W,C] -> fp32 [B,C,H,W] / 255 (device kernel); fp32 NCHW passes through."""
    if isinstance(input_image, np.ndarray):
        input_image = torch.from_numpy(np.ascontiguousarray(input_image))
    if input_image.dtype == torch.uint8:
        return boundary.uint8_to_hint(input_image.to(device))
    return input_image.to(device, torch.float32)


def process(model, ddim_sampler, input_image, prompt, a_prompt, n_prompt, num_samples, num_repeats, image_resolution,
            detect_resolution, ddim_steps, guess_mode, strength, scale, seed, eta, class_map=None, spath='', index=0,
            x_T=None, return_tensors=False):
    """Same arguments as the reference; `x_T` (optional start noise) and `return_tensors` (uint8 NHWC device tensor
    instead of a list of numpy images) are additions."""
    with torch.no_grad():
        control = _control_from(input_image, model.device)
        B, C, H, W = control.shape
        if H % 64 or W % 64:          # the reference's annotator.util.resize_image always hands over multiples of 64
            raise ValueError(f'control image {H}x{W}: height and width must be multiples of 64 (latent of 8 px, three stride-2 '
                             'levels); resize it first (fgdm_amd.seg2image.resized_shape)')
        control = torch.cat([control for _ in range(num_repeats)], dim=0)
        if seed == -1:
            seed = random.randint(0, 65535)
        torch.manual_seed(seed)                       # seed_everything(seed)
        np.random.seed(seed)
        random.seed(seed)
        cond = {'c_concat': [control],
                'c_crossattn': [model.get_learned_conditioning([prompt + ', ' + a_prompt] * num_samples)]}
        un_cond = {'c_concat': None if guess_mode else [control],
                   'c_crossattn': [model.get_learned_conditioning([n_prompt] * num_samples)]}
        shape = (4, H // 8, W // 8)
        model.control_scales = ([strength * (0.825 ** float(12 - i)) for i in range(13)] if guess_mode
                                else ([strength] * 13))
        samples, _ = ddim_sampler.sample(ddim_steps, num_samples, shape, cond, verbose=False, eta=eta, x_T=x_T,
                                         unconditional_guidance_scale=scale, unconditional_conditioning=un_cond)
        x_samples = model.decode_first_stage(samples)
        u8 = boundary.image_to_uint8(x_samples, 1)    # (x * 127.5 + 127.5).clip(0, 255).astype(np.uint8), b h w c
        if return_tensors:
            return u8
        u8 = u8.cpu().numpy()
        return [u8[i] for i in range(num_samples)]

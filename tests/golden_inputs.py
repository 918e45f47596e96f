"""Seeded inputs shared by tools/make_goldens.py (which feeds them to the imported
reference) and by the tests (which feed them to the oracle / the HIP engine).
Inputs are regenerated from the seed instead of being stored in the fixtures."""
import numpy as np
import torch

from fgdm_amd import synth

# key -> (rng stream name, shape)
TABLE = {
    'ops/emb': ('emb', (2, 1280)),
    'ops/res_320_320_x': ('res_320_320.x', (2, 320, 8, 8)),
    'ops/res_320_640_x': ('res_320_640.x', (2, 320, 8, 8)),
    'ops/res_2560_1280_x': ('res_2560_1280.x', (2, 2560, 8, 8)),
    'ops/res_960_320_x': ('res_960_320.x', (2, 960, 8, 8)),
    'ops/down_x': ('down.x', (2, 320, 16, 16)),
    'ops/up_x': ('up.x', (2, 640, 8, 8)),
    'ops/gn5_x': ('gn5.x', (2, 320, 8, 8)),
    'ops/gn6_x': ('gn6.x', (2, 320, 8, 8)),
    'ops/ctx': ('ctx', (2, 77, 768)),
    'ops/attn_x': ('attn_self.x', (2, 64, 320)),
    'ops/attn160_x': ('attn_self160.x', (1, 64, 1280)),
    'ops/ff_x': ('ff.x', (2, 64, 320)),
    'ops/st_x': ('st.x', (2, 640, 8, 8)),
    'ops/st320_x': ('st320.x', (2, 320, 16, 16)),
    'ops/st1280_x': ('st1280.x', (2, 1280, 8, 8)),
    'ops/arb_x': ('arb.x', (2, 320, 16, 16)),
    'ops/adapter_x': ('adapter.x', (2, 4, 16, 16)),
    'unet/ctx': ('unet.ctx', (2, 77, 768)),
    'unet/x8': ('unet.x8', (2, 4, 8, 8)),
    'unet/x16': ('unet.x16', (2, 4, 16, 16)),
    'cn/ctx': ('cn.ctx', (2, 77, 768)),
    'cn/x': ('cn.x', (2, 4, 8, 8)),
    'small/ctx': ('small.ctx', (2, 77, 768)),
    'small/x': ('small.x', (2, 4, 64, 64)),
    'samp/x_T': ('samp.xT', (2, 4, 8, 8)),
    'samp/c': ('samp.c', (2, 77, 768)),
    'samp/uc': ('samp.uc', (2, 77, 768)),
    'samp/mask': ('samp.mask', (2, 1, 8, 8)),
    'samp/x0': ('samp.x0', (2, 4, 8, 8)),
    'samp/ac': ('samp.ac', (2, 77, 768)),
    'samp/noise': ('samp.noise', (2, 4, 8, 8)),
    'sunet/x_T': ('sunet.xT', (2, 4, 16, 16)),
    'sunet/c': ('sunet.c', (2, 77, 768)),
    'sunet/uc': ('sunet.uc', (2, 77, 768)),
    'adapt/cond0': ('adapt.cond0', (2, 4, 16, 16)),
    'adapt/cond1': ('adapt.cond1', (2, 4, 16, 16)),
    'adapt/control': ('adapt.control', (2, 4, 16, 16)),
    'vae/z8': ('vae.z8', (2, 4, 8, 8)),
    'vae/z16': ('vae.z16', (1, 4, 16, 16)),
}


def get(key, seed=7):
    name, shape = TABLE[key]
    x = torch.from_numpy(synth._rng(name, seed).standard_normal(shape, dtype=np.float32))
    if key in ('ops/gn5_x', 'ops/gn6_x'):
        x = x * 3.0 + 0.5
    if key.startswith('vae/'):
        x = x * 0.18215          # latents as the sampler returns them (scaled by scale_factor)
    if key == 'samp/mask':
        x = (x > 0).float()
    return x


def clip_ids(n=2, T=77, seed=7):
    """Token ids shaped like the CLIP tokenizer's output: BOS, words, EOS, EOS padding (max_length 77)."""
    rng = synth._rng('clip.ids', seed)
    ids = np.full((n, T), 49407, dtype=np.int64)
    ids[:, 0] = 49406
    for b, L in enumerate([9, 40, 75, 0, 23][:n]):
        ids[b, 1:1 + L] = rng.integers(0, 49406, size=L)
    return torch.from_numpy(ids)


def hint(n, res, seed):
    return torch.from_numpy(synth.hint(n, res=res, seed=seed))


# model configs used by the goldens (keys follow the reference's UNetModel kwargs)
SD_CFG = dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1),
              num_res_blocks=2, channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768, transformer_depth=1)
# reduced depth, SD widths: d_head 40 / 80
SMALL_CFG = dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(1, 2),
                 num_res_blocks=1, channel_mult=(1, 2), num_heads=8, context_dim=768, transformer_depth=1)
# reduced width, SD depth: d_head 40 / 80 / 160 with 4 heads
NARROW_CFG = dict(in_channels=4, out_channels=4, model_channels=160, attention_resolutions=(4, 2, 1),
                  num_res_blocks=2, channel_mult=(1, 2, 4, 4), num_heads=4, context_dim=768, transformer_depth=1)
CTRL_SCALES = [0.5 + 0.05 * i for i in range(13)]

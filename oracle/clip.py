"""CLIP text encoder (FrozenCLIPEmbedder's transformer)  --  CPU oracle, TEST INFRASTRUCTURE ONLY.

The reference does not contain this algorithm: `FrozenCLIPEmbedder` (ldm/modules/encoders/modules.py:137-162;
controlnet/ldm/modules/encoders/modules.py:88-121) calls the THIRD-PARTY `transformers.CLIPTextModel`
("openai/clip-vit-large-patch14"; the reference's environment pins transformers==4.19.2) on the tokenizer's ids and returns
`outputs.last_hidden_state`.  Restated here is that model's published algorithm (CLIP text transformer, pre-LN, causal):
    h = token_embedding[ids] + position_embedding[0..T)
    per layer:  h += out_proj(softmax(causal(q k^T d^-1/2)) v),  q/k/v = proj(LN1(h));   h += fc2(quick_gelu(fc1(LN2(h))))
    last_hidden_state = final_layer_norm(h)                      quick_gelu(x) = x * sigmoid(1.702 x)
PINNED against the `transformers` installed in this image (5.15.0, same architecture; tests/golden/clip.npz made by
tools/make_goldens.py g_clip with synthetic weights -- the real weights need the network).  Tokenisation (BPE) is host
code of the same third-party package and stays outside: inputs are token ids.
Parameter names are those of the reference's checkpoints: `cond_stage_model.transformer.text_model.*`.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

# openai/clip-vit-large-patch14 text tower
SD_CLIP = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
               num_attention_heads=12, max_position_embeddings=77)
PREFIX = 'cond_stage_model.transformer.text_model.'


def param_shapes(cfg=SD_CLIP, prefix=PREFIX):
    """State-dict keys in module-registration order (transformers 4.19.2 modeling_clip.py; `position_ids` is a buffer,
    not a parameter, and is not loaded)."""
    W, I = cfg['hidden_size'], cfg['intermediate_size']
    p = OrderedDict()
    p[prefix + 'embeddings.token_embedding.weight'] = (cfg['vocab_size'], W)
    p[prefix + 'embeddings.position_embedding.weight'] = (cfg['max_position_embeddings'], W)
    for i in range(cfg['num_hidden_layers']):
        lp = f'{prefix}encoder.layers.{i}.'
        for n in ('k_proj', 'v_proj', 'q_proj', 'out_proj'):
            p[lp + f'self_attn.{n}.weight'] = (W, W)
            p[lp + f'self_attn.{n}.bias'] = (W,)
        p[lp + 'layer_norm1.weight'] = (W,)
        p[lp + 'layer_norm1.bias'] = (W,)
        p[lp + 'mlp.fc1.weight'] = (I, W)
        p[lp + 'mlp.fc1.bias'] = (I,)
        p[lp + 'mlp.fc2.weight'] = (W, I)
        p[lp + 'mlp.fc2.bias'] = (W,)
        p[lp + 'layer_norm2.weight'] = (W,)
        p[lp + 'layer_norm2.bias'] = (W,)
    p[prefix + 'final_layer_norm.weight'] = (W,)
    p[prefix + 'final_layer_norm.bias'] = (W,)
    return p


def _ln(x, p, name):
    return F.layer_norm(x, (x.shape[-1],), p[name + '.weight'], p[name + '.bias'], 1e-5)


def _lin(x, p, name):
    return F.linear(x, p[name + '.weight'], p[name + '.bias'])


def text_encode(p, ids, cfg=SD_CLIP, prefix=PREFIX):
    """ids int64 [B, T] -> last_hidden_state fp32 [B, T, W]."""
    B, T = ids.shape
    H = cfg['num_attention_heads']
    W = cfg['hidden_size']
    d = W // H
    h = p[prefix + 'embeddings.token_embedding.weight'][ids] + p[prefix + 'embeddings.position_embedding.weight'][:T]
    mask = torch.full((T, T), float('-inf')).triu(1)
    for i in range(cfg['num_hidden_layers']):
        lp = f'{prefix}encoder.layers.{i}.'
        x = _ln(h, p, lp + 'layer_norm1')
        q = _lin(x, p, lp + 'self_attn.q_proj').view(B, T, H, d).transpose(1, 2) * d ** -0.5
        k = _lin(x, p, lp + 'self_attn.k_proj').view(B, T, H, d).transpose(1, 2)
        v = _lin(x, p, lp + 'self_attn.v_proj').view(B, T, H, d).transpose(1, 2)
        w = torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1)
        a = (w @ v).transpose(1, 2).reshape(B, T, W)
        h = h + _lin(a, p, lp + 'self_attn.out_proj')
        x = _lin(_ln(h, p, lp + 'layer_norm2'), p, lp + 'mlp.fc1')
        x = x * torch.sigmoid(1.702 * x)
        h = h + _lin(x, p, lp + 'mlp.fc2')
    return _ln(h, p, prefix + 'final_layer_norm')

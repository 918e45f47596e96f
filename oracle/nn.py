"""Functional CPU forward of the denoiser networks.  TEST INFRASTRUCTURE.

Plain ``torch.nn.functional`` on CPU tensors, parameters looked up by the
reference's state-dict key names.  Three precision modes (oracle/precision.py): 'fp32' (the golden-pinned
default: every ``st`` / ``wt`` / ``like`` below is then the identity), 'autocast' (the reference's CUDA fp16 policy,
applied op by op by oracle/autocast.py; this file only carries the reference's explicit dtype casts) and 'engine'
(the HIP engine's storage policy: ``st`` marks each tensor the engine writes to HBM, ``wt`` each packed weight).
Restates:
  * timestep_embedding        ldm/modules/diffusionmodules/util.py:160-180
  * ResBlock._forward         ldm/modules/diffusionmodules/openaimodel.py:275-301
  * Downsample / Upsample     openaimodel.py:171-180 / :114-130
  * SpatialTransformer        ldm/modules/attention.py:275-292
  * BasicTransformerBlock     attention.py:234-240
  * CrossAttention            attention.py:177-216  (tuple's 2nd element is discarded by callers)
  * GEGLU / FeedForward       attention.py:37-64    (exact-erf GELU)
  * UNetModel.forward / forward_original   openaimodel.py:808-884 / :753-806
  * Adapter / ResnetBlock     ldm/modules/encoders/adapter.py:334-346 / :301-313
  * ControlNet.forward        controlnet/cldm/cldm.py:792-813
  * ControlledUnetModel.forward / ControlLDM.apply_model   cldm.py:27-50 / :836-849
"""
import math
import torch
import torch.nn.functional as F

from . import arch, autocast
from . import precision as P
from .precision import st, wt, like


def timestep_embedding(t, dim, max_period=10000):
    # util.py:171-175: cos first, then sin
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _gn(x, p, name, eps):
    # GroupNorm32.forward (util.py:223-225): super().forward(x.float()).type(x.dtype)
    return like(F.group_norm(x.float(), 32, p[name + '.weight'], p[name + '.bias'], eps), x)


def _ln(x, p, name):
    # nn.LayerNorm: fp32 output under autocast whatever the input dtype; the engine stores it as fp16
    return st(F.layer_norm(x, x.shape[-1:], p[name + '.weight'], p[name + '.bias'], 1e-5), 'norm')


# Test-only knob (tests/test_oracle_autocast.py): evaluate every convolution on a channels-last copy of its input.
# In exact arithmetic that is a no-op; on the CPU it only changes the ORDER of the fp32 sums inside the kernel.
REORDER_FP32_SUMS = False


def _conv(x, p, name, stride=1, padding=1):
    if REORDER_FP32_SUMS and x.dim() == 4:
        x = x.contiguous(memory_format=torch.channels_last)
    return F.conv2d(x, wt(p[name + '.weight']), p[name + '.bias'], stride=stride, padding=padding)


def _lin(x, p, name, bias=True, wscale=None):
    return F.linear(x, wt(p[name + '.weight'], wscale), p[name + '.bias'] if bias else None)


def time_embed(p, prefix, t, mc):
    # engine: timestep_embed writes fp16; time_embed.0 and .2 carry SiLU in their epilogues (the only use of `emb`
    # is SiLU(emb) in every ResBlock's emb_layers, openaimodel.py:238-244), so `emb` itself is never stored
    e = st(timestep_embedding(t, mc), 'misc')
    e = _lin(e, p, prefix + 'time_embed.0')
    e = st(F.silu(e), 'misc')
    return _lin(e, p, prefix + 'time_embed.2')


def resblock(p, pre, x, emb, down=False):
    # openaimodel.py:275-301 (no scale-shift norm, dropout p=0); down=True: AvgPool2d(2) on h and x (:276-282, use_conv=False)
    h = st(F.silu(_gn(x, p, pre + 'in_layers.0', 1e-5)), 'norm')
    if down:
        h = st(F.avg_pool2d(h, 2, 2), 'misc')
        x = st(F.avg_pool2d(x, 2, 2), 'misc')
    h = _conv(h, p, pre + 'in_layers.2')
    e = like(_lin(st(F.silu(emb), 'misc'), p, pre + 'emb_layers.1'), h)      # emb_out = self.emb_layers(emb).type(h.dtype)
    h = st(h + e[:, :, None, None])                                   # engine: emb row added in the conv's fp32 epilogue
    h = st(_conv(st(F.silu(_gn(h, p, pre + 'out_layers.0', 1e-5)), 'norm'), p, pre + 'out_layers.3'))
    if (pre + 'skip_connection.weight') in p:
        x = st(_conv(x, p, pre + 'skip_connection', padding=0))
    return st(x + h, 'resid')


def attention(p, pre, x, ctx, heads, fp32_sim=False):
    """attention.py:177-202 ; softmax(q k^T d^-1/2) v ; to_out has a bias, q/k/v do not.
    fp32_sim: the ControlNet side's copy computes q k^T in fp32 outside autocast
    (controlnet/ldm/modules/attention.py:174-177); only the 'autocast' mode can tell the difference."""
    ctx = x if ctx is None else ctx
    c = p[pre + 'to_q.weight'].shape[0]
    d = c // heads
    if P.MODE == 'engine':
        raise RuntimeError("'engine' mode folds the LayerNorm into the projections: go through transformer_block")
    q = _lin(x, p, pre + 'to_q', bias=False)
    k = _lin(ctx, p, pre + 'to_k', bias=False)
    v = _lin(ctx, p, pre + 'to_v', bias=False)
    b, n, _ = q.shape

    def split(t):        # 'b n (h d) -> (b h) n d', and the same einsum strings as the reference: the CPU then runs the
        return t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)   # same kernels
    q, k, v = split(q), split(k), split(v)
    if fp32_sim:
        with autocast.fp32_island():
            sim = torch.einsum('b i d, b j d -> b i j', q.float(), k.float()) * (d ** -0.5)
    else:
        sim = torch.einsum('b i d, b j d -> b i j', q, k) * (d ** -0.5)
    attn = sim.softmax(dim=-1)
    o = torch.einsum('b i j, b j d -> b i d', attn, v)
    o = o.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, c)
    return _lin(o, p, pre + 'to_out.0')


def _ln_lin(p, x, ln, lin, bias=True, wscale=None):
    """'engine' mode: a LayerNorm folded into the Linear it feeds (fgdm_amd/csrc/common.h IgemmArgs::ln_stats).  The packed
    weights are fp16(gamma_k W_nk [* wscale]), the bias picks up sum_k beta_k W_nk, and the GEMM runs on the RAW fp16 row with
    (mean, rstd) applied to the fp32 accumulator: rstd (x W'^T - mean sum_k W') + c == Linear'(normalise(x)) with the normalised
    row never rounded or stored."""
    key = ('ln', ln, lin, wscale)
    hit = P._wcache.get(key)
    if hit is None:
        W = p[lin + '.weight'] if wscale is None else p[lin + '.weight'] * wscale
        c = W @ p[ln + '.bias']
        if bias:
            c = c + p[lin + '.bias']
        hit = ((W * p[ln + '.weight'][None, :]).half().float(), c)
        P._wcache[key] = hit
    xn = F.layer_norm(x, x.shape[-1:], None, None, 1e-5)
    return F.linear(xn, hit[0], hit[1])


def _attention_engine(p, pre, x, ctx, heads, d, ln):
    """The engine's attention numerics (fgdm_amd/csrc/attention.hip): x is the RAW token matrix, `ln` the LayerNorm that is
    folded into to_q (and, for self-attention, to_k / to_v); log2(e) d^-1/2 folded into the packed to_q weights; fp32 scores
    in the log2 domain, fp16 probabilities, normaliser = sum of the fp16 probabilities where the padded PV tile has a spare
    row (d = 40, 80), of the fp32 ones otherwise (d = 160)."""
    import math
    q = st(_ln_lin(p, x, ln, pre + 'to_q', bias=False, wscale=math.log2(math.e) * d ** -0.5))
    if ctx is None:
        k = st(_ln_lin(p, x, ln, pre + 'to_k', bias=False))
        v = st(_ln_lin(p, x, ln, pre + 'to_v', bias=False))
    else:
        k = st(_lin(ctx, p, pre + 'to_k', bias=False))
        v = st(_lin(ctx, p, pre + 'to_v', bias=False))
    b, n, c = q.shape

    def split(t):
        return t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    s = torch.matmul(q, k.transpose(-1, -2))
    pr = torch.exp2(s - s.amax(dim=-1, keepdim=True))
    p16 = st(pr, 'attn')
    spare_row = (d % 32) != 0
    den = (p16 if spare_row else pr).sum(dim=-1, keepdim=True)
    o = st(torch.matmul(p16, v) / den, 'attn').permute(0, 2, 1, 3).reshape(b, n, c)
    return _lin(o, p, pre + 'to_out.0')


def transformer_block(p, pre, x, ctx, heads, fp32_sim=False):
    # attention.py:234-240
    if P.MODE == 'engine':      # norm1 / norm2 / norm3 live inside the GEMMs they feed
        d = x.shape[-1] // heads
        x = st(st(_attention_engine(p, pre + 'attn1.', x, None, heads, d, pre + 'norm1')) + x, 'resid')
        x = st(st(_attention_engine(p, pre + 'attn2.', x, ctx, heads, d, pre + 'norm2')) + x, 'resid')
        a, g = _ln_lin(p, x, pre + 'norm3', pre + 'ff.net.0.proj').chunk(2, dim=-1)
        return st(st(_lin(st(a * F.gelu(g)), p, pre + 'ff.net.2')) + x, 'resid')
    x = st(st(attention(p, pre + 'attn1.', _ln(x, p, pre + 'norm1'), None, heads, fp32_sim)) + x, 'resid')
    x = st(st(attention(p, pre + 'attn2.', _ln(x, p, pre + 'norm2'), ctx, heads, fp32_sim)) + x, 'resid')
    h = _ln(x, p, pre + 'norm3')
    h = _lin(h, p, pre + 'ff.net.0.proj')
    a, g = h.chunk(2, dim=-1)
    h = st(a * F.gelu(g))                               # exact erf GELU, attention.py:43-44; engine: GEGLU epilogue
    return st(st(_lin(h, p, pre + 'ff.net.2')) + x, 'resid')


def spatial_transformer(p, pre, x, ctx, heads, fp32_sim=False):
    # attention.py:275-292 ; GroupNorm eps 1e-6 (attention.py:76-77)
    b, c, hh, ww = x.shape
    x_in = x
    x = st(_gn(x, p, pre + 'norm', 1e-6), 'norm')
    x = st(_conv(x, p, pre + 'proj_in', padding=0))
    x = x.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    x = transformer_block(p, pre + 'transformer_blocks.0.', x, ctx, heads, fp32_sim)
    x = x.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    if fp32_sim:
        x = x.contiguous()        # the ControlNet-side copy does (controlnet/ldm/modules/attention.py:331); numerically a no-op
    x = st(_conv(x, p, pre + 'proj_out', padding=0))
    return st(x + x_in, 'resid')


def run_block(p, pre, layers, h, emb, ctx, fp32_sim=False):
    for j, l in enumerate(layers):
        lp = f'{pre}{j}.'
        if l[0] == 'conv':
            h = st(F.conv2d(h, wt(p[lp + 'weight']), p[lp + 'bias'], padding=1))
        elif l[0] == 'res':
            h = resblock(p, lp, h, emb)
        elif l[0] == 'attn':
            h = spatial_transformer(p, lp, h, ctx, l[2], fp32_sim)
        elif l[0] == 'down':
            h = st(F.conv2d(h, wt(p[lp + 'op.weight']), p[lp + 'op.bias'], stride=2, padding=1))
        elif l[0] == 'up':
            h = F.interpolate(h, scale_factor=2, mode='nearest')
            h = st(F.conv2d(h, wt(p[lp + 'conv.weight']), p[lp + 'conv.bias'], padding=1))
    return h


def adapter_forward(p, pre, x, cin=4):
    # adapter.py:334-346 + ResnetBlock.forward :301-313 (ksize=1, sk=True, use_conv=False)
    feats = []
    x = st(F.conv2d(x, wt(p[pre + 'conv_in.weight']), p[pre + 'conv_in.bias'], padding=1))
    body = arch.adapter_blocks(cin)
    nlev = len(arch.ADAPTER_CHANNELS)
    nrb = len(body) // nlev
    for i in range(nlev):
        for j in range(nrb):
            k = i * nrb + j
            ic, oc, down = body[k]
            b = f'{pre}body.{k}.'
            if down:
                x = st(F.avg_pool2d(x, kernel_size=2, stride=2), 'misc')
            if ic != oc:
                x = st(F.conv2d(x, wt(p[b + 'in_conv.weight']), p[b + 'in_conv.bias']))
            h = F.conv2d(x, wt(p[b + 'block1.weight']), p[b + 'block1.bias'], padding=1)
            h = st(F.relu(h))
            h = st(F.conv2d(h, wt(p[b + 'block2.weight']), p[b + 'block2.bias']))
            x = st(h + x, 'resid')
        feats.append(x)
    return feats


def time_adapter_forward(p, pre, x, emb, cin=4):
    # adapter.py:405-417
    feats = []
    x = st(F.conv2d(x, wt(p[pre + 'conv_in.weight']), p[pre + 'conv_in.bias'], padding=1))
    body = arch.adapter_blocks(cin)
    for k, (ic, oc, down) in enumerate(body):
        x = resblock(p, f'{pre}body.{k}.', x, emb, down=down)
        if k % 2 == 1:
            feats.append(x)
    return feats


def unet_forward(p, cfg, x, t, ctx, prefix='', use_adapter=False, pcond=None,
                 control=None, only_mid_control=False, conds=None, fp32_sim=False):
    """eps = UNet(x, t, ctx).

    use_adapter=False, control=None : UNetModel.forward_original (openaimodel.py:753-806)
    use_adapter=True                : UNetModel.forward with FG-DM adapter (openaimodel.py:808-884);
                                      feature k is added after input block 3k+2 *before* the skip push
    control=[13 tensors]            : ControlledUnetModel.forward (cldm.py:27-50); list is consumed from the end
    conds=[tensors]                 : AdaptUNetModel.forward (openaimodel.py:1263-1320, num_prompts = len(conds) + 1):
                                      `adapters.{k}(conds[k])` features are summed onto the `adapter(prompt)` features
                                      (prompt = pcond, the reference's `control` argument, or x)
    fp32_sim=True                   : the model is built from contronet/ldm (cldm.py's UNet and ControlNet): QK^T in fp32
    """
    inp, mid, out = arch.unet_blocks(cfg)
    emb = time_embed(p, prefix, t, cfg['model_channels'])
    ctx = st(ctx, 'misc')
    h = st(x.float(), 'misc')                        # h = x.type(self.dtype)
    fa, fks = None, []
    prompt = h if pcond is None else st(pcond, 'misc')
    if use_adapter == 'time':      # use_time_adapter=True: fa = self.adapter(prompt, emb)  (openaimodel.py:843-844)
        fa = time_adapter_forward(p, prefix + 'adapter.', prompt, emb, cfg['in_channels'])
    elif use_adapter:
        fa = adapter_forward(p, prefix + 'adapter.', prompt, cfg['in_channels'])
    if conds is not None:
        fks = [adapter_forward(p, f'{prefix}adapters.{kdx}.', st(cond, 'misc'), cfg['in_channels']) for kdx, cond in enumerate(conds)]
    controls = _control_lists(control)
    hs = []
    k = 0
    for i, layers in enumerate(inp):
        h = run_block(p, f'{prefix}input_blocks.{i}.', layers, h, emb, ctx, fp32_sim)
        if fa is not None and (i + 1) % 3 == 0:
            if fks:                                  # fk = sum_k fas_list[k][idx]; h = h + fk + fa[idx]  (openaimodel.py:1301-1305)
                fk = fks[0][k]
                for f in fks[1:]:
                    fk = st(fk + f[k], 'resid')
                h = st(h + fk, 'resid')
            h = st(h + fa[k], 'resid')
            k += 1
        hs.append(h)
    if fa is not None:
        assert k == len(fa)
    h = run_block(p, f'{prefix}middle_block.', mid, h, emb, ctx, fp32_sim)
    for c in controls:
        h = st(h + c.pop(), 'resid')                          # h += control.pop()  (cldm.py:40)
    for i, layers in enumerate(out):
        skip = hs.pop()
        if not only_mid_control:
            for c in controls:
                skip = st(skip + c.pop(), 'resid')            # hs.pop() + control.pop()  (cldm.py:46)
        h = torch.cat([h, skip], dim=1)
        h = run_block(p, f'{prefix}output_blocks.{i}.', layers, h, emb, ctx, fp32_sim)
    h = like(h, x.float())                           # h = h.type(x.dtype)
    h = st(F.silu(_gn(h, p, prefix + 'out.0', 1e-5)), 'norm')
    return F.conv2d(h, wt(p[prefix + 'out.2.weight']), p[prefix + 'out.2.bias'], padding=1)


def _control_lists(control):
    """control: None, one list of 13 residuals (the reference), or a list of such lists (several ControlNets: applied
    one after the other, which is how the engine accumulates them in place)."""
    if control is None:
        return []
    if len(control) and isinstance(control[0], (list, tuple)):
        return [list(c) for c in control]
    return [list(control)]


def hint_block(p, prefix, hint):
    # cldm.py:655-671: 8 conv3x3, SiLU between, stride 2 at convs 2,4,6 (0-based)
    h = st(hint.float(), 'misc')
    for k in range(8):
        stride = 2 if k in (2, 4, 6) else 1
        h = F.conv2d(h, wt(p[f'{prefix}input_hint_block.{2 * k}.weight']),
                     p[f'{prefix}input_hint_block.{2 * k}.bias'], stride=stride, padding=1)
        if k != 7:
            h = F.silu(h)
        h = st(h)
    return h


def controlnet_forward(p, cfg, x, hint, t, ctx, prefix=''):
    """13 control residuals; cldm.py:792-813."""
    inp, mid, _ = arch.unet_blocks(cfg)
    emb = time_embed(p, prefix, t, cfg['model_channels'])
    guided = hint_block(p, prefix, hint)
    outs = []
    ctx = st(ctx, 'misc')
    h = st(x.float(), 'misc')
    for i, layers in enumerate(inp):
        h = run_block(p, f'{prefix}input_blocks.{i}.', layers, h, emb, ctx, fp32_sim=True)
        if guided is not None:
            h = st(h + guided, 'resid')
            guided = None
        # zero convs: raw (unscaled, and in 'engine' mode unrounded: the engine applies scale, rounding and the add into
        # the UNet's skip tensor in the same epilogue)
        outs.append(F.conv2d(h, wt(p[f'{prefix}zero_convs.{i}.0.weight']), p[f'{prefix}zero_convs.{i}.0.bias']))
    h = run_block(p, f'{prefix}middle_block.', mid, h, emb, ctx, fp32_sim=True)
    outs.append(F.conv2d(h, wt(p[prefix + 'middle_block_out.0.weight']), p[prefix + 'middle_block_out.0.bias']))
    return outs


def control_ldm_apply(p, cfg, x, t, ctx, hints, scales=None,
                      unet_prefix='model.diffusion_model.', cn_prefixes=('control_model.',)):
    """ControlLDM.apply_model (cldm.py:836-849).

    ``hints`` is a list with one hint per ControlNet; with several ControlNets
    (BASELINE configs 4/5 -- not in the reference) the 13-tensor
    residual lists are all added to the ControlledUnetModel's skips, one ControlNet after the other.
    hints=None -> plain UNet (c_concat None branch, cldm.py:842-843).
    """
    if hints is None:
        return unet_forward(p, cfg, x, t, ctx, prefix=unet_prefix, fp32_sim=True)
    controls = []
    for hint, cp in zip(hints, cn_prefixes):
        ctrl = controlnet_forward(p, cfg, x, hint, t, ctx, prefix=cp)
        sc = scales if scales is not None else [1.0] * 13
        controls.append([st(c * s) for c, s in zip(ctrl, sc)])       # control = [c * scale ...]  (cldm.py:846)
    return unet_forward(p, cfg, x, t, ctx, prefix=unet_prefix, control=controls, fp32_sim=True)

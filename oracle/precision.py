"""Precision modes of the CPU oracle.  TEST INFRASTRUCTURE.

  'fp32'      the reference's PyTorch-CPU path: everything fp32.  This is the mode the golden vectors from the imported
              reference pin (tests/test_oracle_golden.py); every helper below is the identity in it.
  'autocast'  the reference's CUDA path: ``torch.autocast("cuda")`` fp16 (scripts/txt2img_fgdm_inference.py:212-217),
              emulated op by op by oracle/autocast.py.  oracle.nn carries the reference's explicit dtype casts
              (``GroupNorm32``'s ``.type(x.dtype)``, ``h.type(x.dtype)`` ...), which are no-ops in fp32.
  'engine'    the numerics policy of the HIP engine, stated once here and mirrored by ``st`` / ``wt`` calls in oracle.nn:
                * weights and every GEMM / conv / attention operand are fp16 (round-to-nearest-even), products are
                  accumulated in fp32 (MFMA);
                * a GEMM's fused epilogue -- bias, timestep-embedding row, SiLU / ReLU / GEGLU, ControlNet scale --
                  is evaluated in fp32 on the accumulator and the result is rounded ONCE to fp16;
                * a residual / skip / adapter-feature add reads two fp16 tensors, adds in fp32 and rounds to fp16;
                * GroupNorm(+SiLU): fp32 statistics and affine, one rounding to fp16 at the end;
                * the three LayerNorms of a transformer block are FOLDED into the Linears they feed (to_q/to_k/to_v, attn2.to_q,
                  ff.net.0.proj): weights fp16(gamma_k W_nk), bias + sum_k beta_k W_nk, the GEMM runs on the raw fp16 tokens and
                  (mean, rstd) -- fp32 sums of the stored tokens -- are applied to the fp32 accumulator; the normalised tokens
                  are never rounded or stored (CLIP's LayerNorms stay separate kernels: fp32 statistics, one rounding);
                * attention: log2(e) d^-1/2 is folded into the to_q weights before their fp16 rounding; scores and the
                  running max stay fp32; probabilities are rounded to fp16 for the PV product, the normaliser is the
                  sum of those fp16 probabilities (head dims with a spare MFMA row: 40, 80) or of the fp32 ones (160);
                * the stacked emb_layers output, the final conv's eps and all sampler state are fp32.
  'w16'       fp16 WEIGHTS (``wt``, as 'engine' rounds them), every activation exact fp32 (``st`` is the identity, LayerNorms stay
              separate): the smallest change any engine that hands fp16 weights to the MFMA makes to the reference's CPU path.
              tests/test_oracle_autocast.py::test_fp16_weights_alone_cost_1e_3 holds its distance from 'fp32' on a whole
              network as a number: the literal 1e-3 of the north star is out of reach before a single activation is rounded.
              (the next line is about 'engine':)
              Against this mode the engine differs only by fp32 summation order: blocks agree to < 1e-3 (tests/test_gpu_blocks.py);
              whole networks do not -- fp16 storage amplifies ANY perturbation to the 1e-3 level (tests/test_oracle_autocast.py).
"""
import contextlib

import torch

MODE = 'fp32'
_wcache = {}
# Round 4: which CLASS of stored tensors carries the activation share of the engine policy's distance from the
# fp32 path?  Every st() call in oracle.nn names its class; classes listed in EXACT stay exact fp32 in 'engine' mode
# (precision.exact('resid') ...).  tools/parity_decompose.py prints the table; no product code reads this.
#   'resid'  the residual stream: results of residual / skip / control / adapter-feature adds (what the next block starts from)
#   'norm'   GroupNorm(+SiLU) / LayerNorm outputs
#   'gemm'   convolution / Linear outputs (fused epilogues included: + emb row, GEGLU, ControlNet scale, q / k / v)
#   'attn'   attention probabilities and the attention output before to_out
#   'misc'   inputs (x, context, hint), pooled tensors, the timestep embedding chain
CLASSES = ('resid', 'norm', 'gemm', 'attn', 'misc')
EXACT = frozenset()


def st(x, cls='gemm'):
    """A tensor the engine stores to HBM between kernels: fp16 rounding in 'engine' mode (unless its class is in EXACT), identity
    otherwise."""
    if MODE == 'engine' and cls not in EXACT:
        return x.half().float()
    return x


@contextlib.contextmanager
def exact(*classes):
    """Inside ``with precision.mode('engine'), precision.exact('resid'):`` tensors of the named classes are not rounded."""
    global EXACT
    for c in classes:
        if c not in CLASSES:
            raise ValueError(c)
    prev = EXACT
    EXACT = frozenset(classes)
    try:
        yield
    finally:
        EXACT = prev


def wt(t, scale=None):
    """A weight as the engine's packed copy holds it (fp16, optionally pre-multiplied in fp32); cached."""
    if MODE not in ('engine', 'w16'):
        return t if scale is None else t * scale
    key = (t.data_ptr(), t.numel(), scale)
    hit = _wcache.get(key)
    if hit is None:
        v = t if scale is None else t * scale
        hit = (t, v.half().float())
        _wcache[key] = hit
    return hit[1]


def like(y, x):
    """``y.type(x.dtype)`` of the reference (GroupNorm32.forward util.py:223-225, UNetModel.forward
    openaimodel.py:803,880): a no-op unless the autocast emulation gave x a lower-precision dtype."""
    return y if y.dtype == x.dtype else y.to(x.dtype)


@contextlib.contextmanager
def mode(name):
    """``with precision.mode('engine'):`` / ``'autocast'`` / ``'fp32'`` / ``'w16'``."""
    global MODE
    if name not in ('fp32', 'autocast', 'engine', 'w16'):
        raise ValueError(name)
    prev = MODE
    MODE = name
    try:
        if name == 'autocast':
            from . import autocast
            with autocast.emulate() as m:
                yield m
        else:
            yield None
    finally:
        MODE = prev
        if name in ('engine', 'w16'):
            _wcache.clear()

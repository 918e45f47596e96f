"""CPU emulation of ``torch.autocast("cuda")`` (fp16) for the oracle.  TEST INFRASTRUCTURE.

The reference runs the whole sampling loop under ``with precision_scope("cuda")``
(scripts/txt2img_fgdm_inference.py:212-217, precision "autocast"): its CUDA path is an fp16 path whose
rounding points are decided op by op by the autocast dispatcher, not by the model code.  There is no CUDA
device here or on the MI355X box, so this module restates the dispatcher's *policy* as a
``TorchFunctionMode`` that works on CPU tensors (real ``torch.float16`` tensors carry the dtype):

  * "lower_precision_fp" ops (conv*, linear, matmul, bmm, mm, addmm, baddbmm, einsum): every floating
    tensor argument is cast to fp16; the op is evaluated on those fp16 values with fp32 accumulation
    (what cuDNN / cuBLAS / MFMA do) and the result is rounded ONCE to fp16.
  * "fp32" ops (group_norm, layer_norm, softmax, log_softmax, exp, log, pow, rsqrt, reciprocal, sum,
    prod, cumsum, norm, softplus ...): floating arguments are cast to fp32, the result is fp32.
  * "promote" ops (cat, stack, addcmul, ...): arguments are cast to the widest floating type among them.
  * everything else (add, mul, silu, gelu, interpolate, avg_pool2d, type casts ...) runs in the dtype of
    its inputs with ordinary type promotion; CPU fp16 kernels compute in fp32 and round once, like the GPU's.

The lists are those of aten/src/ATen/autocast_mode.cpp for the torch 2.0.0 the reference pins (fgdm.yml:169).
Nested ``torch.autocast(enabled=False, device_type='cuda')`` regions of the reference
(controlnet/ldm/modules/attention.py:174-177: fp32 QK^T on the ControlNet side) are honoured by patching
``torch.autocast`` while the mode is active.

Used two ways, both on CPU:
  * tools/make_goldens.py runs the REFERENCE's own modules under this mode -> tests/golden/*_ac.npz
    ("the reference under its own autocast policy");
  * oracle.nn run under this mode (``with autocast.emulate():``) is the oracle's 'autocast' precision mode;
    tests/test_oracle_autocast.py holds it to those goldens, and tests/test_oracle_golden.py keeps holding the
    SAME code without the mode to the fp32 goldens.
What this cannot pin: vendor-kernel internals (e.g. whether a bias is added before or after the fp16
rounding of a convolution's output, split-K summation order).  Those move results by O(2^-12) per op.
"""
import contextlib

import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode

T = torch.Tensor

_LOWER = {
    torch.conv1d, torch.conv2d, torch.conv3d, torch.conv_transpose2d, torch.convolution,
    F.conv1d, F.conv2d, F.conv3d, F.conv_transpose2d, F.linear, torch._C._nn.linear,
    torch.matmul, T.matmul, T.__matmul__, torch.bmm, T.bmm, torch.mm, T.mm, torch.mv, T.mv,
    torch.addmm, T.addmm, torch.baddbmm, T.baddbmm, torch.addbmm, torch.einsum, torch.prelu, F.prelu,
    F.scaled_dot_product_attention,
}
# ops whose argument 1 is a parameter (cached cast)
_WEIGHTED = {torch.conv1d, torch.conv2d, torch.conv3d, F.conv1d, F.conv2d, F.conv3d, F.linear, torch._C._nn.linear}
_FP32 = {
    F.group_norm, torch.group_norm, F.layer_norm, torch.layer_norm, torch.native_layer_norm,
    torch.exp, T.exp, torch.expm1, torch.log, T.log, torch.log2, torch.log10, torch.log1p,
    torch.pow, T.pow, T.__pow__, T.__rpow__, torch.rsqrt, T.rsqrt, torch.reciprocal, T.reciprocal,
    T.__rtruediv__,
    torch.acos, torch.asin, torch.cosh, torch.sinh, torch.tan, torch.erfinv, F.softplus,
    torch.norm, T.norm, torch.linalg.norm, F.normalize, torch.dist, torch.cdist, torch.renorm, torch.logsumexp,
    F.mse_loss, F.l1_loss, F.smooth_l1_loss, F.cosine_similarity,
}
# fp32_set_opt_dtype: run with dtype=float32 unless the caller passed a dtype
_FP32_OPT_DTYPE = {
    F.softmax, torch.softmax, T.softmax, F.log_softmax, torch.log_softmax, T.log_softmax,
    torch.sum, T.sum, torch.prod, T.prod, torch.cumsum, T.cumsum, torch.cumprod, T.cumprod,
}
_PROMOTE = {torch.cat, torch.concat, torch.stack, torch.addcmul, torch.addcdiv, torch.atan2, torch.cross, torch.dot,
            torch.tensordot, torch.bilinear, F.bilinear, F.grid_sample}


def _map(x, fn):
    if isinstance(x, torch.Tensor):
        return fn(x)
    if isinstance(x, (list, tuple)):
        return type(x)(_map(v, fn) for v in x)
    if isinstance(x, dict):
        return {k: _map(v, fn) for k, v in x.items()}
    return x


def _floats(x, acc):
    if isinstance(x, torch.Tensor):
        if x.is_floating_point():
            acc.append(x)
    elif isinstance(x, (list, tuple)):
        for v in x:
            _floats(v, acc)
    elif isinstance(x, dict):
        for v in x.values():
            _floats(v, acc)
    return acc


class Emulation(TorchFunctionMode):
    """The mode; see the module docstring.  ``stats`` counts how many calls each class handled."""

    def __init__(self):
        super().__init__()
        self._wcache = {}
        self.enabled = True
        self.stats = {'lower': 0, 'fp32': 0, 'promote': 0}

    def _grid16(self, t, weight=False):
        """fp32 tensor whose values lie on the fp16 grid (the cast autocast performs).  WEIGHTS (argument 1 of conv / linear)
        are cached like autocast's own weight-cast cache; activations never are (a cache entry keeps its tensor alive)."""
        if t.dtype == torch.float16:
            return t.float()
        if not weight or t.dtype != torch.float32 or t.numel() < 4096:
            return t.half().float()
        key = (t.data_ptr(), t.numel(), t._version)
        hit = self._wcache.get(key)
        if hit is None:
            hit = (t, t.half().float())          # keep `t` alive so the key stays unique
            self._wcache[key] = hit
        return hit[1]

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if not self.enabled:
            return func(*args, **kwargs)
        if func in _LOWER:
            self.stats['lower'] += 1
            f = lambda t: self._grid16(t) if t.is_floating_point() else t
            if func in _WEIGHTED and len(args) >= 2 and isinstance(args[1], torch.Tensor) and args[1].is_floating_point():
                args = (args[0], self._grid16(args[1], weight=True)) + tuple(args[2:])
            out = func(*_map(args, f), **_map(kwargs, f))
            return _map(out, lambda t: t.half() if t.is_floating_point() else t)
        if func in _FP32:
            self.stats['fp32'] += 1
            f = lambda t: t.float() if t.is_floating_point() else t
            return func(*_map(args, f), **_map(kwargs, f))
        if func in _FP32_OPT_DTYPE:
            self.stats['fp32'] += 1
            if kwargs.get('dtype') is None and not any(isinstance(a, torch.dtype) for a in args):
                f = lambda t: t.float() if t.is_floating_point() else t
                return func(*_map(args, f), **_map(kwargs, f))
            return func(*args, **kwargs)
        if func in _PROMOTE:
            fl = _floats([args, kwargs], [])
            if fl and len({t.dtype for t in fl}) > 1:
                self.stats['promote'] += 1
                wide = torch.float32 if any(t.dtype == torch.float32 for t in fl) else fl[0].dtype
                if any(t.dtype == torch.float64 for t in fl):
                    wide = torch.float64
                f = lambda t: t.to(wide) if t.is_floating_point() else t
                return func(*_map(args, f), **_map(kwargs, f))
        return func(*args, **kwargs)


class _NestedAutocast:
    """Stand-in for ``torch.autocast(...)`` context managers entered INSIDE an emulated region:
    ``enabled=False`` switches the emulation off for the block (the reference's fp32 QK^T island)."""

    def __init__(self, mode, *args, enabled=True, **kw):
        self.mode, self.want = mode, bool(enabled)

    def __enter__(self):
        self.prev = self.mode.enabled
        self.mode.enabled = self.want
        return self

    def __exit__(self, *exc):
        self.mode.enabled = self.prev
        return False


_ACTIVE = []      # stack of modes entered through emulate()


@contextlib.contextmanager
def fp32_island():
    """The oracle's counterpart of the reference's ``with torch.autocast(enabled=False, device_type='cuda'):``
    (controlnet/ldm/modules/attention.py:174-177): switches the innermost active emulation off for the block."""
    if not _ACTIVE:
        yield
        return
    m = _ACTIVE[-1]
    prev, m.enabled = m.enabled, False
    try:
        yield
    finally:
        m.enabled = prev


@contextlib.contextmanager
def emulate(active=True):
    """``with emulate():`` evaluate torch code under the CUDA-autocast fp16 policy on CPU.
    ``active=False`` is a no-op (plain fp32), so callers can switch precision with one flag."""
    if not active:
        yield None
        return
    mode = Emulation()
    real = torch.autocast

    def nested(*a, **k):
        return _NestedAutocast(mode, *a, **k)
    torch.autocast = nested
    _ACTIVE.append(mode)
    try:
        with mode:
            yield mode
    finally:
        _ACTIVE.pop()
        torch.autocast = real

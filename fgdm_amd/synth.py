"""Deterministic synthetic weights and inputs.

There are no checkpoints in the build/bench environment, so
weights are generated from a counter-based generator keyed by the reference's
state-dict key name; the same bytes are produced on every rank / machine, so the
HIP engine, the CPU oracle and the golden-vector script all see identical
parameters.  Zero-initialised reference modules (zero_module: openaimodel.py:249,
attention.py:269, cldm.py:670,790) get the same non-zero rule, otherwise a
random-init network would output exactly 0 (SURVEY hazard H1).
"""
import hashlib
import numpy as np

DEFAULT_SEED = 1234


def _rng(name, seed):
    h = hashlib.sha256(f'{seed}:{name}'.encode()).digest()
    return np.random.Generator(np.random.Philox(key=int.from_bytes(h[:8], 'little')))


def make_tensor(name, shape, seed=DEFAULT_SEED):
    """float32 ndarray for state-dict key ``name``.

    >=2-D (conv / linear weights): U(-a, a), a = sqrt(3 / fan_in)  (variance 1/fan_in)
    1-D '*.weight' (GroupNorm / LayerNorm gain): 1 + 0.2 U(-1, 1)
    1-D '*.bias': 0.1 U(-1, 1)
    """
    shape = tuple(int(s) for s in shape)
    u = _rng(name, seed).random(shape, dtype=np.float32) * 2.0 - 1.0
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return (u * np.float32(np.sqrt(3.0 / fan_in))).astype(np.float32)
    if name.endswith('weight'):
        return (1.0 + 0.2 * u).astype(np.float32)
    return (0.1 * u).astype(np.float32)


def make_state_dict(shapes, seed=DEFAULT_SEED):
    """{key: float32 ndarray} for an ordered mapping key -> shape."""
    return {k: make_tensor(k, s, seed) for k, s in shapes.items()}


def latents(n, h=64, w=64, c=4, seed=42):
    """x_T ~ N(0,1) for the GLOBAL batch [n, c, h, w]; ranks slice it."""
    return _rng('x_T', seed).standard_normal((n, c, h, w), dtype=np.float32)


def context(n, seed=43, tokens=77, dim=768):
    """Stand-in for CLIP text embeddings [n, 77, 768], O(1) entries."""
    return _rng('context', seed).standard_normal((n, tokens, dim), dtype=np.float32)


def hint(n, res=512, seed=45, grid=8):
    """Segmentation-like RGB hint in [0,1]: grid x grid random palette colours, nearest-upsampled."""
    g = _rng('hint', seed)
    cells = g.integers(0, 256, size=(n, 3, grid, grid)).astype(np.float32) / 255.0
    rep = res // grid
    return np.repeat(np.repeat(cells, rep, axis=2), rep, axis=3).astype(np.float32)

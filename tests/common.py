"""Helpers shared by the tests."""
import os

import numpy as np
import torch

from fgdm_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def gold(name):
    return np.load(os.path.join(GOLD, name + '.npz'))


def params(shapes, prefix=''):
    """Synthetic parameters as torch CPU tensors: {prefix+key: tensor}; names hashed WITH the prefix,
    exactly as tools/make_goldens.py load_synth() does."""
    return {prefix + k: torch.from_numpy(synth.make_tensor(prefix + k, s)) for k, s in shapes.items()}


def relerr(a, b):
    """normwise relative error ||a-b|| / ||b||"""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def report(name, err, tol):
    """Record a measured parity error (normwise relative) so it can be quoted; returns err."""
    line = f'{name}: rel_err={err:.3e} (tol {tol:.1e})'
    print(line)
    d = os.path.join(os.path.dirname(GOLD), '..', 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, 'parity_errors.txt'), 'a') as f:
            f.write(line + '\n')
    return err

#!/usr/bin/env python3
"""Where the epilogue of the short-K linears goes: the L0 shapes with the ablation bits of IgemmArgs::debug.
Usage (GPU box): python tools/bench_epilogue.py [--iters 20]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgdm_amd import _lib

SHAPES = [
    ('L0 lin 320->320 +res', 32, 64, 64, 320, 0, 320, 1, 1, 0, 0, 1),
    ('L0 lin 320->320', 32, 64, 64, 320, 0, 320, 1, 1, 0, 0, 0),
    ('L0 lin 1280->320 +res', 32, 64, 64, 1280, 0, 320, 1, 1, 0, 0, 1),
    ('L0 geglu 320->2560', 32, 64, 64, 320, 0, 2560, 1, 1, 0, 3, 0),
]
BITS = [(0, 'full'), (4, 'no epilogue'), (1, 'no K-loop loads'), (2, 'no MFMAs')]
if os.environ.get('CFG'):
    BITS = [(0, 'auto')] + [(int(c) / 256.0, 'cfg %s' % c) for c in os.environ['CFG'].split(',')]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    lib = _lib.load()
    print('shape'.ljust(24) + ''.join(n.rjust(22) for _, n in BITS))
    for name, B, H, W, C0, C1, Co, ks, st, up, act, res in SHAPES:
        row = name.ljust(24)
        for bits, _ in BITS:
            ms = C.c_float()
            rc = lib.fgdm_bench_igemm(B, H, W, C0, C1, Co, ks, st, up, act, res, int(bits * 256), a.iters, C.byref(ms))
            row += (f'{ms.value * 1e3:7.0f} us' if rc == 0 else '-').rjust(22)
        print(row, flush=True)


if __name__ == '__main__':
    main()

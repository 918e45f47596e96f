#!/usr/bin/env python3
"""Time GroupNorm / LayerNorm on the shapes of the C3 workload (B = 32 rows of a CFG batch); GB/s are ALGORITHMIC
(read once + write once, fp16).  Usage (GPU box): python tools/bench_norm.py [--iters 50]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgdm_amd import _lib

# kind (0 GroupNorm+SiLU, 1 LayerNorm), B, HW, C0, C1
SHAPES = [(0, 32, 4096, 320, 0), (0, 32, 4096, 320, 320), (0, 32, 4096, 640, 320), (0, 32, 1024, 640, 0), (0, 32, 1024, 640, 640),
          (0, 32, 1024, 1280, 640), (0, 32, 256, 1280, 0), (0, 32, 256, 1280, 1280), (0, 32, 64, 1280, 0), (0, 32, 64, 1280, 1280),
          (0, 16, 4096, 320, 0), (0, 16, 1024, 640, 0), (0, 16, 1024, 640, 640), (0, 16, 64, 1280, 0), (0, 2, 1024, 640, 0),
          (1, 32, 4096, 320, 0), (1, 32, 1024, 640, 0), (1, 32, 256, 1280, 0), (1, 32, 64, 1280, 0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=50)
    a = ap.parse_args()
    lib = _lib.load()
    for kind, B, HW, C0, C1 in SHAPES:
        ms = C.c_float()
        rc = lib.fgdm_bench_norm(kind, B, HW, C0, C1, 1, a.iters, C.byref(ms))
        nbytes = 4.0 * B * HW * (C0 + C1)
        name = ('groupnorm' if kind == 0 else 'layernorm') + f' B{B} HW{HW} C{C0}' + (f'+{C1}' if C1 else '')
        print(f'{name:36s} rc={rc} {ms.value * 1e3:8.1f} us  {nbytes / (ms.value * 1e-3) / 1e9:8.0f} GB/s', flush=True)


if __name__ == '__main__':
    main()

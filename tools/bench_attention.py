#!/usr/bin/env python3
"""Time the fused attention kernel on the shapes of the C3 workload (B = 32 rows of a CFG batch, 8 heads); TF/s are
ALGORITHMIC (4 B T Tk C).  Usage (GPU box): python tools/bench_attention.py [--iters 20]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgdm_amd import _lib

SHAPES = [(32, 8, 4096, 4096, 40), (32, 8, 4096, 77, 40), (32, 8, 1024, 1024, 80), (32, 8, 1024, 77, 80),
          (32, 8, 256, 256, 160), (32, 8, 256, 77, 160), (32, 8, 64, 64, 160), (16, 8, 4096, 4096, 40)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--only', type=int, default=-1, help='index into SHAPES (profiling runs)')
    a = ap.parse_args()
    lib = _lib.load()
    for B, H, T, Tk, d in (SHAPES if a.only < 0 else [SHAPES[a.only]]):
        ms = C.c_float()
        rc = lib.fgdm_bench_attention(B, H, T, Tk, d, a.iters, C.byref(ms))
        fl = 4.0 * B * T * Tk * H * d
        print(f'attn B{B} T{T} Tk{Tk} d{d:<4d} rc={rc} {ms.value * 1e3:9.1f} us {fl / (ms.value * 1e-3) / 1e12:8.1f} TF/s', flush=True)


if __name__ == '__main__':
    main()

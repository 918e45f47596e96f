#!/usr/bin/env python3
"""Feed-forward pair (GEGLU projection + output projection, ldm/modules/attention.py:37-64) in row chunks that share one 4C
intermediate buffer: does a chunk's intermediate stay on chip between producer and consumer?
Usage (GPU box): python tools/bench_ff.py [--iters 10]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgdm_amd import _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=10)
    a = ap.parse_args()
    lib = _lib.load()
    for M, Cw in ((131072, 320), (32768, 640), (8192, 1280)):
        flops = 2.0 * M * Cw * 8 * Cw + 2.0 * M * 4 * Cw * Cw
        row = f'FF M{M} C{Cw}:'
        ch = M
        while ch >= max(M // 16, 2048):
            ms = C.c_float()
            rc = lib.fgdm_bench_ff(M, Cw, ch, a.iters, C.byref(ms))
            row += f'  chunk {ch}: {ms.value * 1e3:7.0f} us {flops / (ms.value * 1e-3) / 1e12:5.0f} TF/s' if rc == 0 else f'  chunk {ch}: rc={rc}'
            ch //= 2
        print(row, flush=True)


if __name__ == '__main__':
    main()

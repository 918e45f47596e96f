#!/usr/bin/env python3
"""Time the implicit-GEMM tile configurations on the layer shapes of the C3 workload (B = 32 rows of CFG batch).
Usage (GPU box): python tools/bench_igemm.py [--iters 20]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgdm_amd import _lib

# name, B, H, W, C0, C1, Cout, ksize, stride, up, act, resid
SHAPES = [
    ('L0 conv 320->320', 32, 64, 64, 320, 0, 320, 3, 1, 0, 0, 1),
    ('L0 conv 640->320 cat', 32, 64, 64, 320, 320, 320, 3, 1, 0, 0, 0),
    ('L0 lin 320->320', 32, 64, 64, 320, 0, 320, 1, 1, 0, 0, 1),
    ('L0 lin 320->640 (qk)', 32, 64, 64, 320, 0, 640, 1, 1, 0, 0, 0),
    ('L0 geglu 320->2560', 32, 64, 64, 320, 0, 2560, 1, 1, 0, 3, 0),
    ('L0 lin 1280->320 (ffo)', 32, 64, 64, 1280, 0, 320, 1, 1, 0, 0, 1),
    ('L1 conv 640->640', 32, 32, 32, 640, 0, 640, 3, 1, 0, 0, 1),
    ('L1 lin 640->640', 32, 32, 32, 640, 0, 640, 1, 1, 0, 0, 1),
    ('L1 geglu 640->5120', 32, 32, 32, 640, 0, 5120, 1, 1, 0, 3, 0),
    ('L2 conv 1280->1280', 32, 16, 16, 1280, 0, 1280, 3, 1, 0, 0, 1),
    ('L2 conv 2560->1280 cat', 32, 16, 16, 1280, 1280, 1280, 3, 1, 0, 0, 0),
    ('L2 lin 1280->1280', 32, 16, 16, 1280, 0, 1280, 1, 1, 0, 0, 1),
    ('L2 geglu 1280->10240', 32, 16, 16, 1280, 0, 10240, 1, 1, 0, 3, 0),
    ('L2 lin 1280->3840 (qkv)', 32, 16, 16, 1280, 0, 3840, 1, 1, 0, 0, 0),
    ('L2 lin 5120->1280 (ffo)', 32, 16, 16, 5120, 0, 1280, 1, 1, 0, 0, 1),
    ('L3 conv 1280->1280', 32, 8, 8, 1280, 0, 1280, 3, 1, 0, 0, 1),
    ('L3 conv 2560->1280 cat', 32, 8, 8, 1280, 1280, 1280, 3, 1, 0, 0, 0),
    ('L3 lin 1280->1280', 32, 8, 8, 1280, 0, 1280, 1, 1, 0, 0, 1),
    ('L3 lin 1280->3840 (qkv)', 32, 8, 8, 1280, 0, 3840, 1, 1, 0, 0, 0),
    ('L3 lin 5120->1280 (ffo)', 32, 8, 8, 5120, 0, 1280, 1, 1, 0, 0, 1),
    ('L3 geglu 1280->10240', 32, 8, 8, 1280, 0, 10240, 1, 1, 0, 3, 0),
    ('L1 down s2 320->320', 32, 64, 64, 320, 0, 320, 3, 2, 0, 0, 0),
    ('L0 up 640->640', 32, 32, 32, 640, 0, 640, 3, 1, 1, 0, 0),
    # the same GEMMs as the L0 / L1 / L2 convolutions without the 3x3 gather (what the addressing and the tap re-reads cost)
    ('L0 lin 2880->320', 32, 64, 64, 2880, 0, 320, 1, 1, 0, 0, 1),
    ('L1 lin 5760->640', 32, 32, 32, 5760, 0, 640, 1, 1, 0, 0, 1),
    ('L2 lin 11520->1280', 32, 16, 16, 11520, 0, 1280, 1, 1, 0, 0, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--cfgs', default='0,4,5,6,7,8,9', help='force_cfg + 256 * ablation bits (IgemmArgs::debug), comma separated')
    ap.add_argument('--shapes', default='', help='only shapes whose name contains one of these comma-separated substrings')
    ap.add_argument('--batch', type=int, default=0, help='override the batch (rows of the CFG batch) of every shape: how does the time scale with tiles per CU?')
    a = ap.parse_args()
    want = [w for w in a.shapes.split(',') if w]
    lib = _lib.load()
    cfgs = [int(c) for c in a.cfgs.split(',')]
    print('shape'.ljust(26) + ''.join(f'cfg{c:>4}(TF/s us)'.rjust(18) for c in cfgs))
    for name, B, H, W, C0, C1, Co, ks, st, up, act, res in SHAPES:
        if want and not any(w in name for w in want):
            continue
        if a.batch:
            B = a.batch
        Ho, Wo = (2 * H, 2 * W) if up else (((H - 1) // 2 + 1, (W - 1) // 2 + 1) if st == 2 else (H, W))
        flops = 2.0 * B * Ho * Wo * Co * ks * ks * (C0 + C1)
        row = name.ljust(26)
        for c in cfgs:
            ms = C.c_float()
            rc = lib.fgdm_bench_igemm(B, H, W, C0, C1, Co, ks, st, up, act, res, c, a.iters, C.byref(ms))
            row += (f'{flops / (ms.value * 1e-3) / 1e12:8.0f} {ms.value * 1e3:7.0f}' if rc == 0 else '       -       -').rjust(18)
        print(row, flush=True)


if __name__ == '__main__':
    main()

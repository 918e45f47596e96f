#!/usr/bin/env python3
"""Calibration of the MFMA roof on THIS device with the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS, fp16, fp32
accumulate) on the GEMM shapes the implicit-GEMM convolutions reduce to, with random and with all-zero operands (the
gap is the chip lowering its clock under load).  Not part of the product: a yardstick for tools/bench_igemm.py."""
import sys
import time

import torch

SHAPES = [('L0 conv 320->320   M131072 N320  K2880 ', 131072, 320, 2880),
          ('L1 conv 640->640   M32768  N640  K5760 ', 32768, 640, 5760),
          ('L2 conv 1280->1280 M8192   N1280 K11520', 8192, 1280, 11520),
          ('L3 conv 1280->1280 M2048   N1280 K11520', 2048, 1280, 11520),
          ('L0 lin  320->320   M131072 N320  K320  ', 131072, 320, 320),
          ('L0 geglu-sized     M131072 N2560 K320  ', 131072, 2560, 320),
          ('square             M8192   N8192 K8192 ', 8192, 8192, 8192)]


def bench(m, n, k, zero, iters=20):
    a = torch.zeros(m, k, dtype=torch.half, device='cuda') if zero else (torch.rand(m, k, device='cuda') * 2 - 1).half()
    b = torch.zeros(n, k, dtype=torch.half, device='cuda') if zero else ((torch.rand(n, k, device='cuda') * 2 - 1) / k ** 0.5).half()
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        c = a @ b.t()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * m * n * k / (ms * 1e-3) / 1e12, ms * 1e3


def main():
    print('shape'.ljust(42) + 'random: TF/s     us      zeros: TF/s     us')
    for name, m, n, k in SHAPES:
        r, z = bench(m, n, k, False), bench(m, n, k, True)
        print(f'{name}  {r[0]:10.0f} {r[1]:8.0f}   {z[0]:10.0f} {z[1]:8.0f}', flush=True)


if __name__ == '__main__':
    main()

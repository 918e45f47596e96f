#!/usr/bin/env python3
"""Which class of stored tensors carries the ACTIVATION share of the engine policy's distance from the reference's fp32 path?
(CPU only; test infrastructure: imports oracle/.)

The full-width ControlNet + UNet evaluation at 8x8 (the case of tests/test_oracle_autocast.py) is run by the CPU oracle in:
  fp32, w16 (fp16 weights, every activation exact), engine (the HIP engine's policy), and engine with ONE class of st()
  roundings left exact at a time (oracle/precision.py: resid / norm / gemm / attn / misc), plus engine with ONLY one class rounded.
All distances are normwise relative errors against the reference's fp32 golden (tests/golden/controlnet_full.npz).
Usage: python tools/parity_decompose.py [out.txt]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import golden_inputs as gi                      # noqa: E402
from common import gold, params, relerr         # noqa: E402
from oracle import arch, nn as onn, precision   # noqa: E402


def main():
    out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else None

    def say(line):
        print(line, flush=True)
        if out:
            out.write(line + '\n')
            out.flush()

    p = params(arch.unet_param_shapes(gi.SD_CFG, adapter=False), 'model.diffusion_model.')
    p.update(params(arch.controlnet_param_shapes(gi.SD_CFG), 'control_model.'))
    run = lambda: onn.control_ldm_apply(p, gi.SD_CFG, gi.get('cn/x'), torch.tensor([981, 21]), gi.get('cn/ctx'),
                                        [gi.hint(2, 64, 45)], scales=gi.CTRL_SCALES).float()
    g32 = gold('controlnet_full')['eps_ctrl']
    gac = gold('controlnet_full_ac')['eps_ctrl'].astype(np.float32)
    say('ControlLDM.apply_model, full width, 8x8 latent, t = (981, 21): normwise relative error vs the reference fp32 golden')
    say(f'  floor: reference under its own autocast policy           {relerr(gac, g32):.3e}')
    res = {}
    with torch.no_grad():
        with precision.mode('fp32'):
            res['fp32'] = relerr(run(), g32)
        say(f'  oracle fp32 (restatement check)                          {res["fp32"]:.3e}')
        with precision.mode('w16'):
            res['w16'] = relerr(run(), g32)
        say(f'  w16: fp16 weights, every activation exact                {res["w16"]:.3e}')
        with precision.mode('engine'):
            res['engine'] = relerr(run(), g32)
        say(f'  engine policy (all classes rounded)                      {res["engine"]:.3e}')
        say('  engine policy with ONE class of stored tensors left exact (what removing that class buys):')
        for c in precision.CLASSES:
            with precision.mode('engine'), precision.exact(c):
                res['no_' + c] = relerr(run(), g32)
            say(f'    {c:6s} exact                                           {res["no_" + c]:.3e}')
        say('  engine policy with ONLY one class rounded (what that class costs on top of the fp16 weights):')
        for c in precision.CLASSES:
            others = [o for o in precision.CLASSES if o != c]
            with precision.mode('engine'), precision.exact(*others):
                res['only_' + c] = relerr(run(), g32)
            say(f'    only {c:6s} rounded                                   {res["only_" + c]:.3e}')
        with precision.mode('engine'), precision.exact(*precision.CLASSES):
            res['none'] = relerr(run(), g32)
        say(f'  engine policy, no activation rounded (LayerNorm folds only) {res["none"]:.3e}')
    act = (max(res['engine'] ** 2 - res['w16'] ** 2, 0.0)) ** 0.5
    say(f'  activation share of the engine policy, sqrt(engine^2 - w16^2) = {act:.3e}')
    if out:
        out.close()


if __name__ == '__main__':
    main()

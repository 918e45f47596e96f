"""Build hygiene (no GPU needed): every gfx950 kernel must compile without scratch (register spills).
The 256x320 implicit-GEMM tile sits close to the 256-VGPR budget; a spill there costs 2x in run time
(measured when the split-K path was first added to the same instantiation)."""
import os
import re
import subprocess

import pytest

from fgdm_amd import build

CSRC = os.path.join(os.path.dirname(os.path.abspath(build.__file__)), 'csrc')


@pytest.mark.parametrize('src', ['igemm.hip', 'igemm2.hip', 'attention.hip', 'norm.hip', 'elementwise.hip', 'boundary.hip', 'text.hip'])
def test_no_scratch(src, tmp_path):
    cmd = [build._hipcc(), *build.FLAGS, '-Rpass-analysis=kernel-resource-usage', '-c', os.path.join(CSRC, src),
           '-o', str(tmp_path / 'x.o')]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    scratch = [int(v) for v in re.findall(r'ScratchSize \[bytes/lane\]: (\d+)', out.stderr)]
    assert scratch, 'no kernels reported'
    assert max(scratch) == 0, f'register spills in {src}: {scratch}'

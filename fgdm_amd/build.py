"""Build libfgdm_hip.so in-tree with hipcc for gfx950 (no torch, no cmake: one hipcc invocation per source)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libfgdm_hip.so')
SOURCES = ['igemm.hip', 'igemm2.hip', 'norm.hip', 'attention.hip', 'elementwise.hip', 'boundary.hip', 'text.hip', 'engine.hip']
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
         '-Wno-unused-variable', '-ffp-contract=fast', '-mllvm', '-amdgpu-mfma-vgpr-form=1', '-fno-honor-nans']


# bit-exact byte kernels: no FMA contraction (the flag comes last, so it overrides -ffp-contract=fast)
EXTRA_FLAGS = {'boundary.hip': ['-ffp-contract=off']}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'fgdm.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        cmd = [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(src, []), '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {src}')
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)

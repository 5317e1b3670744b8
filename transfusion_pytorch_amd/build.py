"""Build libtfx_hip.so (the C-ABI HIP library) for gfx950 with hipcc, in-tree.

    python -m transfusion_pytorch_amd.build          # build if sources are newer than the .so
    python -m transfusion_pytorch_amd.build --force
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libtfx_hip.so')
SOURCES = ['gemm.hip', 'attention.hip', 'tokenwise.hip', 'decode.hip', 'collective.hip', 'runner.hip']
HEADERS = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.h', '.inc'))] + [os.path.join(os.path.dirname(HERE), 'include', 'tfx.h')]   # incl. the generated asm loops
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-Wno-unused-result'] + os.environ.get('TFX_HIPCC_EXTRA', '').split()   # e.g. -DTFX_PP_TIMING (tools/pp_timing.py)

# per-file flags.  attention.hip: hipcc's SLP vectoriser packs adjacent fp32 adds / multiplies into v_pk_*_f32, which issue in 6.5 clocks per
# wave against 2 x 2.7 for the scalar pair on gfx950 (tools/valu_probe.hip): the attention kernels are VALU-bound, keep them scalar
FILE_FLAGS = {'attention.hip': ['-fno-slp-vectorize']}


def _hipcc() -> str:
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(LIBDIR, s.replace('.hip', '.o'))
        cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(s, []), '-c', os.path.join(CSRC, s), '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), s))
        objs.append(o)
    for pr, s in procs:
        if pr.wait() != 0:
            raise RuntimeError(f'hipcc failed on {s}')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-ldl', '-o', LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))

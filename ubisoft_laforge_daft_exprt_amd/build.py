"""Builds libdaft_exprt_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libdaft_exprt_hip.so')
SOURCES = ['dx_runtime.hip', 'dx_gemm.hip', 'dx_ffpair.hip', 'dx_attention.hip', 'dx_rows.hip', 'dx_upsample.hip', 'dx_loss.hip', 'dx_optim.hip', 'dx_pitch.hip']
F16_SOURCES = ['dx_gemm.hip', 'dx_ffpair.hip', 'dx_attention.hip', 'dx_rows.hip', 'dx_pitch.hip']    # compiled a second time with -DDX_F16 (fp16 operand mode)
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wno-unused-value', '-Wno-unused-result'] + os.environ.get('DX_EXTRA_HIPCC_FLAGS', '').split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(job):
    src, f16 = job
    obj = os.path.join(CSRC, os.path.splitext(src)[0] + ('_f16.o' if f16 else '.o'))
    deps = [os.path.join(CSRC, src), os.path.join(CSRC, 'dx_common.h'), os.path.join(CSRC, 'dx_f16_names.h'), os.path.join(CSRC, 'dx_rowvec.h')]
    if _stale(obj, deps):
        subprocess.run(['hipcc', *FLAGS, *(['-DDX_F16'] if f16 else []), '-c', os.path.join(CSRC, src), '-o', obj], check=True)
    return obj


STAMP = os.path.join(CSRC, '.build_flags')     # the flag string the objects in the tree were compiled with


def _flags_changed() -> bool:
    """Objects built with other flags (an ablation build via DX_EXTRA_HIPCC_FLAGS) are newer than their sources, so mtimes alone would
    keep them: the flag string is recorded next to the objects and any difference rebuilds everything."""
    want = ' '.join(FLAGS)
    try:
        with open(STAMP) as f:
            return f.read() != want
    except OSError:
        return any(f.endswith('.o') for f in os.listdir(CSRC))     # objects of unknown provenance


def build(force: bool = False, verbose: bool = False) -> str:
    jobs = [(src, False) for src in SOURCES] + [(src, True) for src in F16_SOURCES]
    force = force or _flags_changed()
    if force:
        for src, f16 in jobs:
            obj = os.path.join(CSRC, os.path.splitext(src)[0] + ('_f16.o' if f16 else '.o'))
            if os.path.exists(obj):
                os.remove(obj)
    with ThreadPoolExecutor(max_workers=6) as pool:
        objs = list(pool.map(_compile, jobs))
    with open(STAMP, 'w') as f:
        f.write(' '.join(FLAGS))
    if _stale(LIB, objs):
        subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs], check=True)
    if verbose:
        print('built', LIB, file=sys.stderr)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)

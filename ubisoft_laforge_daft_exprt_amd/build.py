"""Builds libdaft_exprt_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libdaft_exprt_hip.so')
SOURCES = ['dx_runtime.hip', 'dx_gemm.hip', 'dx_ffpair.hip', 'dx_attention.hip', 'dx_rows.hip', 'dx_upsample.hip', 'dx_loss.hip', 'dx_optim.hip']
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wno-unused-value', '-Wno-unused-result']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(CSRC, os.path.splitext(src)[0] + '.o')
    deps = [os.path.join(CSRC, src), os.path.join(CSRC, 'dx_common.h')]
    if _stale(obj, deps):
        subprocess.run(['hipcc', *FLAGS, '-c', os.path.join(CSRC, src), '-o', obj], check=True)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    if force:
        for src in SOURCES:
            obj = os.path.join(CSRC, os.path.splitext(src)[0] + '.o')
            if os.path.exists(obj):
                os.remove(obj)
    with ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as pool:
        objs = list(pool.map(_compile, SOURCES))
    if _stale(LIB, objs):
        subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs], check=True)
    if verbose:
        print('built', LIB, file=sys.stderr)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)

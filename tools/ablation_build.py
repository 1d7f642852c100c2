"""Diagnostic builds for timing ablations: ONE source of the library recompiled with extra flags and linked with the other, unchanged
objects into tools/ab/lib_<name>.so (git-ignored; travels to the GPU box).  Select it with DX_LIB_PATH=tools/ab/lib_<name>.so.

    python tools/ablation_build.py <name> <source.hip> <flags...>        e.g.  attn_noexp dx_attention.hip -DDX_ATTN_ABL=1
"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from ubisoft_laforge_daft_exprt_amd import build as B   # noqa: E402


def main():
    name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build()
    out_dir = os.path.join(REPO, 'tools', 'ab')
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for s in B.SOURCES:
        stem = os.path.splitext(s)[0]
        if s == src:
            o = os.path.join(out_dir, f'{stem}_{name}.o')
            subprocess.run(['hipcc', *B.FLAGS, *flags, '-c', os.path.join(B.CSRC, s), '-o', o], check=True)
            objs.append(o)
        else:
            objs.append(os.path.join(B.CSRC, stem + '.o'))
    for s in B.F16_SOURCES:
        objs.append(os.path.join(B.CSRC, os.path.splitext(s)[0] + '_f16.o'))
    lib = os.path.join(out_dir, f'lib_{name}.so')
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib, *objs], check=True)
    print(lib)


if __name__ == '__main__':
    main()

"""ctypes binding of libdaft_exprt_hip.so, generated from include/daft_exprt_hip.h.

There is no CPU or PyTorch fallback: if the shared library is missing or an entry point fails, the product path raises.
"""
from __future__ import annotations

import ctypes
import os
import re

PKG = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(PKG), 'include', 'daft_exprt_hip.h')
# DX_LIB_PATH: a diagnostic build of the SAME library (tools/ablation_build.py: timing ablations of one source file); never a fallback
LIB_PATH = os.environ.get('DX_LIB_PATH') or os.path.join(PKG, 'libdaft_exprt_hip.so')

_SCALARS = {'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float, 'double': ctypes.c_double,
            'uint64_t': ctypes.c_uint64, 'uint32_t': ctypes.c_uint32}


def parse_header(path: str = HEADER, with_names: bool = False):
    """-> {name: (restype, [argtypes])} for every function declared in the header (``with_names``: + [argument names])."""
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r'\b(int|const char\*)\s+(dx_\w+)\s*\(([^)]*)\)\s*;', text):
        argtypes, argnames = [], []
        args = args.strip()
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                argnames.append(re.split(r'[\s*]+', a)[-1])
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = a.replace('const ', '').split()[0]
                    argtypes.append(_SCALARS[base])
        restype = ctypes.c_int if ret == 'int' else ctypes.c_char_p
        protos[name] = (restype, argtypes, argnames) if with_names else (restype, argtypes)
    return protos


class DxError(RuntimeError):
    pass


# Diagnostic hook (bench.py / tools only; never read by any computation): when set to a list, every C-ABI launch is bracketed by
# two events on torch's current stream -- the stream the kernels are launched on -- and appended as
# (entry point, {argument name: value}, start event, stop event).  profiling.py prices the records.
TIMER = None


def set_timer(records):
    global TIMER
    old, TIMER = TIMER, records
    return old


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f'{LIB_PATH} is missing: build it with `python -m ubisoft_laforge_daft_exprt_amd.build` '
                '(hipcc --offload-arch=gfx950). There is no fallback path.')
        self._dll = ctypes.CDLL(LIB_PATH)
        self._last_error = None
        protos = parse_header(with_names=True)
        # fp16 twins (csrc/dx_f16_names.h): the same entry points compiled with fp16 in place of bf16, suffix _f16
        f16 = os.path.join(PKG, 'csrc', 'dx_f16_names.h')
        if os.path.exists(f16):
            for base in re.findall(r'#define (dx_\w+) \1_f16', open(f16).read()):
                if base in protos and hasattr(self._dll, base + '_f16'):
                    protos[base + '_f16'] = protos[base]
        for name, (restype, argtypes, argnames) in protos.items():
            fn = getattr(self._dll, name)  # AttributeError if the header and the library disagree
            fn.restype = restype
            fn.argtypes = argtypes
            if name == 'dx_last_error':
                self._last_error = fn
                setattr(self, name, fn)
            elif restype is ctypes.c_int and name != 'dx_version':
                setattr(self, name, self._checked(name, fn, argnames))
            else:
                setattr(self, name, fn)

    def _checked(self, name, fn, argnames):
        launches = 'stream' in argnames

        def call(*args):
            if TIMER is not None and launches:
                import torch
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                rc = fn(*args)
                b.record()
                TIMER.append((name, dict(zip(argnames, args)), a, b))
            else:
                rc = fn(*args)
            if rc != 0:
                msg = self._last_error().decode(errors='replace') if self._last_error else ''
                raise DxError(f'{name} failed (code {rc}): {msg}')
        call.__name__ = name
        return call


_LIB = None


def lib() -> _Lib:
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB

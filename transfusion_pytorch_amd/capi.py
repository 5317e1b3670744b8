"""ctypes binding of the C ABI in include/tfx.h (libtfx_hip.so).

The ctypes `Structure`s are GENERATED from the header text, so the Python side can never drift from
the C side.  There is no fallback: if the library is missing the import of anything that needs it raises.
"""
from __future__ import annotations

import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'tfx.h')
LIB_PATH = os.environ.get('TFX_LIB') or os.path.join(HERE, 'lib', 'libtfx_hip.so')     # TFX_LIB: A/B a second build (tools/ab.sh)

_SCALARS = {'int32_t': ctypes.c_int32, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float, 'int': ctypes.c_int}


def _strip_comments(src: str) -> str:
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return re.sub(r'//[^\n]*', '', src)


def parse_header(path: str = HEADER):
    """returns (structs: name -> [(field, ctype)], functions: name -> [arg ctype], enums: name -> int)"""
    src = _strip_comments(open(path).read())
    structs, funcs, enums = {}, {}, {}
    for body in re.findall(r'enum\s*\{(.*?)\}', src, flags=re.S):
        for item in body.split(','):
            if '=' in item:
                k, v = item.split('=')
                enums[k.strip()] = int(v.strip())
    for body, name in re.findall(r'typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;', src, flags=re.S):
        fields = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            decl = decl.replace('const ', '')
            m = re.match(r'(\w+)\s*(.*)', decl, flags=re.S)
            base, rest = m.group(1), m.group(2)
            for d in rest.split(','):
                d = d.strip()
                is_ptr = d.startswith('*')
                fname = d.lstrip('* ').strip()
                if is_ptr:
                    fields.append((fname, ctypes.c_void_p))
                else:
                    fields.append((fname, _SCALARS[base]))
        structs[name] = fields
    for ret, name, args in re.findall(r'\b(int|const char\s*\*)\s+(tfx_\w+)\s*\(([^)]*)\)\s*;', src):
        argt = []
        for a in args.split(','):
            a = a.strip().replace('const ', '')
            if a in ('void', ''):
                continue
            if '*' in a:
                argt.append(ctypes.c_void_p)
            else:
                argt.append(_SCALARS[a.split()[0]])
        funcs[name] = (ctypes.c_char_p if 'char' in ret else ctypes.c_int, argt)
    return structs, funcs, enums


STRUCT_FIELDS, FUNCTIONS, ENUMS = parse_header()


def _make_struct(name, fields):
    return type(name, (ctypes.Structure,), {'_fields_': fields})


STRUCTS = {n: _make_struct(n, f) for n, f in STRUCT_FIELDS.items()}
_lib = None


class TfxError(RuntimeError):
    pass


def lib():
    """load libtfx_hip.so (fail loudly - there is no CPU / PyTorch fallback for the product path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TfxError(f'{LIB_PATH} is missing: build it with `python -m transfusion_pytorch_amd.build` '
                           '(hipcc --offload-arch=gfx950); the Transfusion hot path has no fallback')
        L = ctypes.CDLL(LIB_PATH)
        for fname, (ret, argt) in FUNCTIONS.items():
            fn = getattr(L, fname)          # AttributeError = header / library drift
            fn.restype = ret
            fn.argtypes = argt
        _lib = L
    return _lib


def ptr(t):
    """raw device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def make_args(struct_name: str, **kw):
    S = STRUCTS[struct_name]
    a = S()
    names = {f for f, _ in STRUCT_FIELDS[struct_name]}
    for k, v in kw.items():
        if k not in names:
            raise KeyError(f'{struct_name} has no field {k}')
        if hasattr(v, 'data_ptr'):
            v = v.data_ptr()
        setattr(a, k, v)
    return a


def check(rc: int, what: str):
    if rc != 0:
        raise TfxError(f'{what} failed with code {rc}')


def call(fn_name: str, args, stream: int):
    """call a `int tfx_xxx(const args*, void* stream)` entry point."""
    rc = getattr(lib(), fn_name)(ctypes.byref(args), ctypes.c_void_p(stream))
    if rc != 0:
        raise TfxError(f'{fn_name} failed with code {rc}')

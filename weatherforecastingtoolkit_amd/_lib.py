"""ctypes binding of libwfae.so, generated from include/wfae.h.

The argument types of every entry point are parsed from the C header so the
Python side cannot drift from the ABI.  There is NO fallback: if the shared
library is missing or a symbol cannot be bound, importing the compute path
raises — the product never runs on a CPU/ATen substitute.
"""
from __future__ import annotations

import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(_ROOT, "include", "wfae.h")
LIB_PATH = os.path.join(_PKG, "libwfae.so")

_CT = {
    "int": ctypes.c_int,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "size_t": ctypes.c_size_t,
    "float": ctypes.c_float,
    "wfae_stream_t": ctypes.c_void_p,
}


def _ctype(decl: str):
    decl = decl.strip()
    if "*" in decl:
        return ctypes.c_void_p
    ty = decl.replace("const", "").split()
    # last token is the parameter name
    base = ty[0] if len(ty) > 1 else ty[0]
    if base not in _CT:
        raise ValueError(f"wfae.h: unknown parameter type in '{decl}'")
    return _CT[base]


def parse_header(path: str = HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every declaration in wfae.h"""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out = {}
    for m in re.finditer(r"\b(int64_t|int|size_t|const char\s*\*)\s+(wfae_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        if ret == "int":
            restype = ctypes.c_int
        elif ret == "int64_t":
            restype = ctypes.c_int64
        elif ret == "size_t":
            restype = ctypes.c_size_t
        else:
            restype = ctypes.c_char_p
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                argtypes.append(_ctype(a))
                argnames.append(a.replace("*", " ").split()[-1])
        out[name] = (restype, argtypes, argnames)
    return out


class WfaeError(RuntimeError):
    pass


_lib = None
_decls = None


def load(path: str = LIB_PATH):
    """Load libwfae.so and bind every symbol the header declares."""
    global _lib, _decls
    if _lib is not None:
        return _lib
    # torch's bundled HIP runtime must be in the process first: libwfae.so then binds
    # to that same libamdhip64 (one runtime, one set of streams) instead of a second copy.
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise WfaeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(weatherforecastingtoolkit_amd/csrc/build.sh). There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    decls = parse_header()
    for name, (restype, argtypes, _) in decls.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise WfaeError(f"libwfae.so does not export {name} declared in include/wfae.h") from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib, _decls = lib, decls
    return lib


def call(name: str, *args):
    """Invoke an int-returning entry point; raise WfaeError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.wfae_last_error_string()
        raise WfaeError(f"{name} failed ({rc}): {msg.decode() if msg else '?'}")

"""ctypes binding of liburse_hip.so (the C ABI declared in include/urse.h).

The product path has no CPU or PyTorch fallback: if the library is missing or a
call fails, an exception is raised.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("URSE_LIB_PATH") or os.path.join(_HERE, "liburse_hip.so")   # (override: A/B of diagnostic builds)
HEADER_PATH = os.path.join(_HERE, "..", "include", "urse.h")

_lib = None


class UrseError(RuntimeError):
    pass


def declared_symbols():
    """Names of every function include/urse.h declares."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(urse_[a-z0-9_]+)\s*\(", src)))


def _ctype(decl):
    decl = decl.strip()
    if "*" in decl:
        return ctypes.c_void_p
    base = decl.replace("const", "").split()
    ty = base[0] if base else "int"
    return {"int": ctypes.c_int, "int32_t": ctypes.c_int, "int64_t": ctypes.c_int64, "long": ctypes.c_int64,
            "float": ctypes.c_float, "double": ctypes.c_double, "unsigned": ctypes.c_uint,
            "uint64_t": ctypes.c_uint64}[ty]


def prototypes():
    """{name: [ctypes of each parameter]} parsed from include/urse.h."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(urse_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        params = m.group(2).strip()
        protos[m.group(1)] = [] if params in ("", "void") else [_ctype(x) for x in params.split(",")]
    return protos


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UrseError(
                "liburse_hip.so not found at %s - run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the hot path)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in prototypes().items():
            fn = getattr(lib, name)  # AttributeError if the header declares something the library lacks
            fn.argtypes = argtypes
            fn.restype = ctypes.c_char_p if name == "urse_last_error" else ctypes.c_int
        _lib = lib
    return _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Call `urse_<name>` with tensors converted to device pointers, ints/floats passed through."""
    lib = load()
    fn = getattr(lib, "urse_" + name)
    cargs = [_ptr(a) if (a is None or isinstance(a, torch.Tensor)) else a for a in args]
    rc = fn(*cargs)
    if rc != 0:
        raise UrseError("urse_%s failed (%d): %s" % (name, rc, lib.urse_last_error().decode()))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise UrseError("the URSE hot path runs on the GPU only (got a %s tensor); there is no CPU fallback"
                            % t.device)

"""ctypes binding of libscat_hip.so. Prototypes are parsed from include/scat_hip.h, so the
header is the single source of truth for the C ABI. The product path has NO fallback: if the
library is missing, importing a kernel raises."""
from __future__ import annotations

import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "scat_hip.h")
LIBPATH = os.path.join(HERE, "libscat_hip.so")
if os.environ.get("SCAT_LIBPATH"):
    # measurement tools only (tools/pw_stamp.py, rows_stamp.py): the -DSCAT_DIAG build whose kernels can overwrite their
    # outputs with time stamps.  Said loudly, because results from that library are not the product's.
    import warnings

    LIBPATH = os.path.abspath(os.environ["SCAT_LIBPATH"])
    warnings.warn(f"scat_amd: SCAT_LIBPATH set, loading {LIBPATH} instead of the shipped libscat_hip.so", RuntimeWarning)

_CT = {
    "int": ctypes.c_int,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "float": ctypes.c_float,
}


def parse_header(path: str = HEADER):
    """-> {name: (restype, [(ctype, argname)])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"(?m)^\s*(const char\*|int64_t|int)\s+(scat_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        at = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    at.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    ty, an = a.rsplit(" ", 1)
                    at.append((_CT[ty.replace("const ", "").strip()], an))
        rt = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "const char*": ctypes.c_char_p}[ret]
        protos[name] = (rt, at)
    return protos


class ScatError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIBPATH):
            raise ScatError(
                f"{LIBPATH} not found: build it with `python -m scat_amd.build` (hipcc, gfx950). "
                "There is no CPU fallback on the product path.")
        # torch bundles its own HIP runtime and publishes it RTLD_GLOBAL; let it initialise the device
        # first so this library's hip* symbols resolve to that one runtime (two runtimes racing for
        # /dev/kfd in one process end in "No HIP GPUs are available").
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
        self.cdll = ctypes.CDLL(LIBPATH)
        self.protos = parse_header()
        for name, (rt, at) in self.protos.items():
            fn = getattr(self.cdll, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = rt
            fn.argtypes = [t for t, _ in at]
            if rt is ctypes.c_int and name not in ("scat_version", "scat_get_math_mode", "scat_epilogue_stats_groups", "scat_splitk_reduce_pending",
                                                      "scat_epilogue_bnb_groups"):
                setattr(self, name, self._checked(fn, name))
            else:
                setattr(self, name, fn)

    def _checked(self, fn, name):
        last = self.cdll.scat_last_error
        last.restype = ctypes.c_char_p

        def call(*a):
            rc = fn(*a)
            if rc != 0:
                raise ScatError(f"{name} failed ({rc}): {last().decode(errors='replace')}")
            return rc

        call.__name__ = name
        return call


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib

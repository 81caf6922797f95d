"""ctypes binding of ``libpathpyg_amd.so``; prototypes are read from ``include/pathpyg_amd.h``.

There is no fallback: if the library has not been built (``python -m pathpyg_amd._build``) loading
fails with a RuntimeError, and every compute entry point of the package goes through here.
"""
from __future__ import annotations

import ctypes
import pathlib
import re

PKG = pathlib.Path(__file__).resolve().parent
HEADER = PKG.parent / "include" / "pathpyg_amd.h"
LIB_PATH = PKG / "lib" / "libpathpyg_amd.so"

_SCALARS = {
    "int": ctypes.c_int,
    "int64_t": ctypes.c_int64,
    "int32_t": ctypes.c_int32,
    "size_t": ctypes.c_size_t,
    "double": ctypes.c_double,
    "float": ctypes.c_float,
    "pp_stream_t": ctypes.c_void_p,
}

_lib = None
_protos: dict[str, tuple] | None = None


def _ctype(decl: str):
    decl = decl.replace("const", " ").strip()
    if "*" in decl:
        return ctypes.c_char_p if decl.replace(" ", "") == "char*" else ctypes.c_void_p
    return _SCALARS[decl.split()[0]]


def declared_functions() -> dict[str, tuple]:
    """{name: (restype, [argtypes])} for every function the C header declares."""
    global _protos
    if _protos is None:
        text = re.sub(r"/\*.*?\*/", " ", HEADER.read_text(), flags=re.S)
        text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
        protos = {}
        for ret, name, args in re.findall(r"([A-Za-z_][\w\s\*]*?)\b(pp_\w+)\s*\(([^;{}]*?)\)\s*;", text):
            ret = ret.strip()
            if not ret or ret.startswith(("typedef", "enum")):
                continue
            argtypes = []
            if args.strip() and args.strip() != "void":
                for a in args.split(","):
                    a = a.strip()
                    # drop the parameter name (last identifier) unless the declaration is a bare type
                    m = re.match(r"(.*?[\*\s])(\w+)$", a)
                    argtypes.append(_ctype(m.group(1) if m else a))
            protos[name] = (_ctype(ret), argtypes)
        _protos = protos
    return _protos


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP library first (python -m pathpyg_amd._build, or "
                "__graft_entry__.build()). pathpyg_amd has no CPU fallback."
            )
        handle = ctypes.CDLL(str(LIB_PATH))
        for name, (restype, argtypes) in declared_functions().items():
            fn = getattr(handle, name)      # AttributeError here = header/library mismatch
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.pp_version() != 100:
            raise RuntimeError("libpathpyg_amd.so version does not match the Python package")
        _lib = handle
    return _lib


class HipError(RuntimeError):
    pass


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().pp_last_error().decode(errors="replace")
        if status == -2:
            raise ValueError(f"{what}: {msg}")
        raise HipError(f"{what} failed with status {status}: {msg}")

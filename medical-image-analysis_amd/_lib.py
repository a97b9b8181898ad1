"""ctypes binding of libmia_hip.so. Prototypes are read from include/mia_hip.h, so the header is the
single source of truth for the C ABI. Importing this module never falls back to anything: a missing
library or symbol raises."""
import ctypes
import os
import re

# torch first: it carries its own HIP runtime, and libmia_hip.so must bind to THAT copy (same streams,
# same device context). Loading the library before torch would pull in a second runtime.
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "mia_hip.h")
LIB_PATH = os.path.join(HERE, "libmia_hip.so")

_CTYPES = {
    "int": ctypes.c_int,
    "long long": ctypes.c_longlong,
    "unsigned long long": ctypes.c_ulonglong,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
}


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|long\s+long|int)\s+(mia_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else ctypes.c_longlong if "long" in ret else ctypes.c_int
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = " ".join(a.split(" ")[:-1]).replace("const ", "").strip()
                    argtypes.append(_CTYPES[base])
        protos[name] = (restype, argtypes)
    return protos


# int-returning entries whose value is a count or a yes/no, not an error code
_COUNTS = ("mia_version", "mia_cbam_ws_floats", "mia_dwconv_part_rows", "mia_dwconv_bwd_data_lin_supported",
           "mia_dwconv_bwd_data_lin_rows")


class MiaError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python {os.path.join(HERE, 'build.py')}` "
                "(there is no fallback path)")
        self._dll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (restype, argtypes) in self.protos.items():
            fn = getattr(self._dll, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
            if restype is ctypes.c_int and name not in _COUNTS:
                setattr(self, name, self._checked(name, fn))
            else:
                setattr(self, name, fn)

    def _checked(self, name, fn):
        last = self._dll.mia_last_error
        last.restype = ctypes.c_char_p

        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise MiaError(f"{name} failed ({rc}): {last().decode()}")
            return rc
        call.__name__ = name
        return call


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib

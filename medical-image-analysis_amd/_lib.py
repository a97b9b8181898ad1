"""Binding of libmia_hip.so. Prototypes are read from include/mia_hip.h, so the header is the single source of truth
for the C ABI. Importing this module never falls back to anything: a missing library or symbol raises.

Two ways to make the same calls: ctypes (always set up), and the generated fast-call module `_mia_fast` (build.py),
which converts arguments straight from the call's argument vector (~0.6 us a call instead of ~5). The fast module is
used when it is present and was generated from these prototypes; it calls through function pointers taken from the
library ctypes loaded, so there is one copy of the library either way. MIA_NO_FASTCALL=1 keeps ctypes (A/B runs)."""
import ctypes
import os

# torch first: it carries its own HIP runtime, and libmia_hip.so must bind to THAT copy (same streams,
# same device context). Loading the library before torch would pull in a second runtime.
import torch  # noqa: F401

from . import _proto

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = _proto.HEADER
LIB_PATH = os.path.join(HERE, "libmia_hip.so")

_CTYPES = {
    "int": ctypes.c_int,
    "longlong": ctypes.c_longlong,
    "ulonglong": ctypes.c_ulonglong,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "ptr": ctypes.c_void_p,
}
_COUNTS = _proto.COUNTS


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    protos = {}
    for name, (ret, kinds) in _proto.parse_protos(path).items():
        restype = ctypes.c_char_p if ret == "str" else ctypes.c_longlong if ret == "longlong" else ctypes.c_int
        protos[name] = (restype, [_CTYPES[k] for k in kinds])
    return protos


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
        self.fastcall = False
        for name, (restype, argtypes) in self.protos.items():
            fn = getattr(self._dll, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
            if restype is ctypes.c_int and name not in _COUNTS:
                setattr(self, name, self._checked(name, fn))
            else:
                setattr(self, name, fn)
        if not os.environ.get("MIA_NO_FASTCALL"):
            self._bind_fast()

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

    def _bind_fast(self):
        try:
            from . import _mia_fast
        except ImportError:
            return
        if _mia_fast.digest() != _proto.header_digest():
            return      # generated from other prototypes: calling through it could pass a wrong argument list
        addr = lambda f: ctypes.cast(f, ctypes.c_void_p).value     # noqa: E731
        names = [n for n, (restype, _) in self.protos.items() if restype is not ctypes.c_char_p]
        _mia_fast.bind({n: addr(getattr(self._dll, n)) for n in names}, MiaError, addr(self._dll.mia_last_error))
        for n in names:
            setattr(self, n, getattr(_mia_fast, n))
        self.fastcall = True


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib

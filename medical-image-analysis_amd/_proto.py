"""Prototypes of the C ABI, read from include/mia_hip.h (the single source of truth for the boundary). Pure Python,
no torch: used by the ctypes binding (_lib.py), by the generator of the fast-call binding (build.py) and by the ABI tests."""
import hashlib
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "mia_hip.h")

# int-returning entries whose value is a count or a yes/no, not an error code
COUNTS = ("mia_version", "mia_cbam_ws_floats", "mia_dwconv_part_rows", "mia_dwconv_bwd_data_lin_supported",
          "mia_dwconv_bwd_data_lin_rows")


def parse_protos(path=HEADER):
    """-> {name: (ret, [kinds])}; ret in {"int", "longlong", "str"}, kinds in {"int", "longlong", "ulonglong", "float",
    "double", "ptr"}, for every function declared in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|long\s+long|int)\s+(mia_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        rkind = "str" if "char" in ret else "longlong" if "long" in ret else "int"
        kinds = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    kinds.append("ptr")
                else:
                    base = " ".join(a.split(" ")[:-1]).replace("const ", "").strip()
                    kinds.append({"int": "int", "long long": "longlong", "unsigned long long": "ulonglong",
                                  "float": "float", "double": "double"}[base])
        protos[name] = (rkind, kinds)
    return protos


def header_digest(path=HEADER) -> str:
    """Digest of the prototypes (not of the comments): a binding generated from another set of prototypes is refused."""
    protos = parse_protos(path)
    h = hashlib.sha256()
    for name in sorted(protos):
        h.update(f"{name}:{protos[name][0]}:{','.join(protos[name][1])};".encode())
    return h.hexdigest()[:16]

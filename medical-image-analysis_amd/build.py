"""Builds libmia_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the .so travels to
the GPU box with the source tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmia_hip.so")
SOURCES = ["runtime.hip", "gemm.hip", "dwconv.hip", "pointwise.hip", "head.hip", "preprocess.hip", "optim.hip", "seg.hip", "metrics.hip", "augment.hip"]


def kernel_tree_hash() -> str:
    """Digest of everything that decides which kernels a training step launches and what they move: the kernel
    sources and the Python modules that sequence them. Measurements kept on file (HBM traffic per kernel from the
    counter passes) carry the digest of the tree they were taken on, so a reader can tell whether they still apply."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files += [os.path.join(HERE, f) for f in ("ops.py", "trunk.py", "head_ops.py", "heads.py", "optim.py", "ddp.py")]
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    # the public header too: _lib.py derives every ctypes prototype from it, so a library older than the header
    # could be called with the wrong argument list
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "mia_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        objs.append(obj)
        if (not force) and os.path.exists(obj) and os.path.getmtime(obj) > max(
                os.path.getmtime(path), os.path.getmtime(os.path.join(CSRC, "common.h"))):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

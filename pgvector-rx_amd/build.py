"""Builds libhnswrx.so (HIP kernels + C ABI + host graph driver) for gfx950, in-tree.

Each translation unit is compiled to its own object (in parallel, only when it or a header changed) and the
objects are linked into the shared library: a one-line kernel edit costs one TU, not the whole engine."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.environ.get("HX_LIB") or os.path.join(HERE, "libhnswrx.so")   # HX_LIB: load a prebuilt variant as is (kernel tuning experiments)
# translation unit -> the non-shared headers it includes (every TU also depends on hx_internal.h and include/*.h)
SOURCES = {
    "hx_engine.hip": ["hx_ops.h", "hx_fused_core.h", "hx_fused.inc.h"],
    "hx_fused_f32.hip": ["hx_ops.h", "hx_fused_core.h", "hx_fused_kernel.h"],
    "hx_fused_f16.hip": ["hx_ops.h", "hx_fused_core.h", "hx_fused_kernel.h"],
    "hx_fused_bit.hip": ["hx_ops.h", "hx_fused_core.h", "hx_fused_kernel.h"],
    "hx_fused_sparse.hip": ["hx_ops.h", "hx_fused_core.h", "hx_fused_kernel.h"],
    "hx_links.hip": ["hx_ops.h", "hx_fused_core.h"],
    "hx_biglist.hip": ["hx_ops.h", "hx_fused_core.h"],
    "hx_mfma.hip": ["hx_ops.h"],
    "hx_sparse.hip": ["hx_ops.h"],
    "hx_group.hip": [],
    "hx_batch.hip": ["hx_ops.h"],
    "hx_index.cpp": [],
}
# -ffp-contract=off: mul and add are rounded separately, as in the reference's unfused Rust
# (and as the oracle's ORC_ORDER_W64 emulation assumes).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-x", "hip"]


def _sources():
    return {s: d for s, d in SOURCES.items() if os.path.exists(os.path.join(CSRC, s))}


def _shared_deps():
    inc = os.path.join(HERE, "..", "include")
    return [os.path.join(CSRC, "hx_internal.h")] + [os.path.join(inc, f) for f in os.listdir(inc)]


def _obj_path(src):
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _tu_stale(src, extra):
    o = _obj_path(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in extra if os.path.exists(os.path.join(CSRC, h))] + _shared_deps()
    return any(os.path.getmtime(f) > t for f in deps)


def stale():
    """True when the library is missing or older than ANY file under csrc/ or include/."""
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + _shared_deps()
    return any(os.path.getmtime(f) > t for f in deps)


def _flags_changed():
    """The objects were compiled with other extra flags (HX_CFLAGS, e.g. -DFUSED_TIMERS or -DHX_EXPERIMENTS) than this call asks for."""
    try:
        return open(os.path.join(OBJ, ".flags")).read() != os.environ.get("HX_CFLAGS", "")
    except OSError:
        return os.path.isdir(OBJ) and any(f.endswith(".o") for f in os.listdir(OBJ)) and bool(os.environ.get("HX_CFLAGS", ""))


def build(force=False, verbose=False):
    if os.environ.get("HX_LIB") or (not force and not stale() and not _flags_changed()):
        return LIB
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    # several rank processes import the package at once (bench.py --gpus N, tests/test_dist_gpu.py): one of them builds, the rest wait
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale() and not _flags_changed():
                return LIB
            return _build_locked(force or _flags_changed(), verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("HX_CFLAGS", "").split()
    todo = [s for s, d in _sources().items() if force or _tu_stale(s, d)]

    def compile_one(src):
        cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", _obj_path(src)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), 6))) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj_path(s) for s in _sources()] + ["-o", LIB, "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(os.path.join(OBJ, ".flags"), "w") as f:
        f.write(os.environ.get("HX_CFLAGS", ""))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

"""Builds libhnswrx.so (HIP kernels + C ABI + host graph driver) for gfx950, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("HX_LIB") or os.path.join(HERE, "libhnswrx.so")   # HX_LIB: load a prebuilt variant as is (kernel tuning experiments)
SOURCES = ["hx_engine.hip", "hx_group.hip", "hx_index.cpp"]
HEADERS = ["hx_internal.h", os.path.join("..", "..", "include", "hnswrx.h")]
# -ffp-contract=off: mul and add are rounded separately, as in the reference's unfused Rust
# (and as the oracle's ORC_ORDER_W64 emulation assumes).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-x", "hip"]


def stale():
    """True when the library is missing or older than ANY file under csrc/ or include/."""
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    inc = os.path.join(HERE, "..", "include")
    deps += [os.path.join(inc, f) for f in os.listdir(inc)]
    return any(os.path.getmtime(f) > t for f in deps)


def build(force=False, verbose=False):
    if os.environ.get("HX_LIB") or (not force and not stale()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB, "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

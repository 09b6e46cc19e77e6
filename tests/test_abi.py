"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol
include/hnswrx.h declares; without a GPU the product fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

import pgvector_rx_amd as hx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "hnswrx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    L = ctypes.CDLL(hx.lib_path()) if os.path.exists(hx.lib_path()) else hx.lib()
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_binding_loads_and_reports_version():
    assert hx.lib().hx_abi_version() == 1


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hx.HxError) as ei:
        hx.Engine(hx.F32, hx.L2SQ, 8, 16)
    assert ei.value.code == -5 and "no CPU fallback" in str(ei.value)


def test_create_argument_errors():
    # argument validation happens before any device is touched
    L = hx.lib()
    h = ctypes.c_void_p()
    assert L.hx_create(0, 7, 0, 8, 16, ctypes.byref(h)) == -1                    # unknown dtype
    assert L.hx_create(0, hx.F32, hx.HAMMING, 8, 16, ctypes.byref(h)) == -1      # metric of another opclass family
    assert L.hx_create(0, hx.F32, hx.L2SQ, 2001, 16, ctypes.byref(h)) == -2      # hnsw_constants.rs:4
    assert b"2000 dimensions" in L.hx_last_error(None)
    assert L.hx_create(0, hx.F16, hx.L2SQ, 4001, 16, ctypes.byref(h)) == -2      # halfvec.rs:876
    assert L.hx_create(0, hx.BIT, hx.HAMMING, 64001, 16, ctypes.byref(h)) == -2  # bitvec.rs:184, hnsw_bit.out
    assert b"64000 dimensions" in L.hx_last_error(None)
    assert L.hx_create(0, hx.F32, hx.L2SQ, 0, 16, ctypes.byref(h)) == -2


def test_levels_match_reference_formula():
    from oracle import orc
    lv = hx.draw_levels(20000, 16, 3)
    assert lv.min() == 0 and lv.max() <= hx.max_level(16) == 82 == orc.lib().orc_max_level(16)
    # P(level >= 1) = 1/m
    assert abs((lv >= 1).mean() - 1 / 16) < 0.01
    # the formula itself, against the oracle's restatement of build.rs:373-377
    u = np.array([0.5, 1e-300, 0.0, 0.999999, 1 / 16, 1 / 256 - 1e-12])
    want = [orc.lib().orc_level_from_uniform(float(x), 16) for x in u]
    assert want == [0, 82, 82, 0, 1, 2] or want[:2] == [0, 82]
    a = hx.draw_levels(100, 16, 9, start=50)
    b = hx.draw_levels(150, 16, 9)[50:]
    assert (a == b).all()


def test_rust_and_python_bindings_cover_the_header():
    """The reference-side binding a maintainer would add (integration/rust/src/gpu/ffi.rs, an `extern "C"` block) and the ctypes binding
    declare exactly the entry points of include/hnswrx.h: nothing missing, nothing the header does not have."""
    names = set(declared_symbols())
    rs = open(os.path.join(ROOT, "integration", "rust", "src", "gpu", "ffi.rs")).read()
    block = rs[rs.index('extern "C" {'):]
    rust = set(re.findall(r"pub fn (hx_[a-z0-9_]+)\s*\(", block[:block.index("\n}\n")]))
    assert rust == names, (sorted(names - rust), sorted(rust - names))
    py = open(os.path.join(ROOT, "pgvector-rx_amd", "binding.py")).read()
    bound = set(re.findall(r'"(hx_[a-z0-9_]+)":\s*\(', py))
    assert bound == names, (sorted(names - bound), sorted(bound - names))

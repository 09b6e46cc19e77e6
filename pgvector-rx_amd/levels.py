"""Level draws for index build.

The reference draws `floor(-ln(rand) * mL)` from an unseeded rand::random (src/index/build.rs:373-377),
so its graphs are not reproducible; builds here take an explicit level array.  This generator is
counter-based (splitmix64) so any rank can produce the levels of any row range without communication.
"""
import numpy as np


def max_level(m):
    """hnsw_get_max_level, src/types/hnsw.rs:337-349 at BLCKSZ 8192."""
    return min((8192 - 24 - 8 - 4 - 4) // 6 // m - 2, 255)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def draw_levels(n, m, seed, start=0):
    """Levels of rows [start, start+n): min(floor(-ln(max(U, MIN_POSITIVE)) / ln m), max_level)."""
    with np.errstate(over="ignore"):
        ctr = np.arange(start, start + n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
    u = (_splitmix64(ctr) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    u = np.maximum(u, np.finfo(np.float64).tiny)
    lv = np.floor(-np.log(u) * (1.0 / np.log(float(m))))
    return np.minimum(lv, max_level(m)).astype(np.int32)


def batch_schedule(size0, n, batch):
    """Batch sizes hx_index_insert uses for n rows appended to an index of size0 elements with batch cap
    `batch`: each batch is at most 1/8 of the current graph (hx_index.cpp, ramp-up rule)."""
    out, size, left = [], size0, n
    while left > 0:
        b = min(batch, left, max(1, size // 8))
        out.append(b)
        size += b
        left -= b
    return out

"""pgvector-rx_amd: MI355X-native HNSW distance engine for pgvector-rx's hot path.

The product is the C-ABI shared library built from csrc/ (include/hnswrx.h).  This Python package is
thin ctypes plumbing over that ABI for tests, bench.py and the multi-GPU (torch.distributed / RCCL)
driver; it contains no arithmetic of its own and has no CPU fallback.
"""
from .binding import (  # noqa: F401
    BIT, F16, F32, HAMMING, JACCARD, L1, L2SQ, NEG_IP, QUERY_SLOT, SPARSE,
    Engine, HxError, Index, lib, lib_path, pack_sparse, sparse_record_bytes,
)
from .levels import batch_schedule, draw_levels, max_level  # noqa: F401

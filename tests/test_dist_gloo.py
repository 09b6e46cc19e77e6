"""N>1 path on CPU: world_size-2 and -3 gloo runs of the sharded batch build (pgvector-rx_amd/dist_build.py)
must leave every rank with the same replicated state as the single-rank run, with the search work actually
split between ranks."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

import pgvector_rx_amd as hx

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(world, n, batch, fmt="host"):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), str(n), str(batch), fmt],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o
        outs.append(re.search(r"DIGEST (\w+) size=(\d+) search=(\d+) owned=(\d+) replicated=(\d+)", o).groups())
    return outs


@pytest.mark.parametrize("fmt", ["host", "device"])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_build_replicas_agree(world, fmt):
    """fmt = the exchange format of dist_build.py: serialized host buffers (hx_index_batch_*) or device records (hx_index_dbatch_*; here the
    "device" buffers are host memory behind raw pointers, the stand-in index checks that every record arrives where the protocol says)."""
    n, batch = 3000, 128
    single = run_world(1, n, batch, fmt)[0]
    outs = run_world(world, n, batch, fmt)
    assert all(o[0] == single[0] for o in outs), (single, outs)          # same state on every rank == single-rank state
    assert all(int(o[1]) == n for o in outs)
    searched = [int(o[2]) for o in outs]
    assert sum(searched) == int(single[2]) + int(single[4]) - int(outs[0][4])   # every sharded member searched exactly once
    assert max(searched) - min(searched) <= len(hx.batch_schedule(0, n, batch)) + 128   # and the slices are balanced
    assert all(int(o[3]) > 0 for o in outs)


def test_slice_bounds_and_schedule():
    import importlib
    db = importlib.import_module("pgvector-rx_amd.dist_build")
    for b in [1, 7, 8, 100, 8192]:
        for w in [1, 2, 3, 8]:
            lo, hi = db.slice_bounds(b, w)
            assert lo[0] == 0 and hi[-1] == b and all(hi[r] == lo[r + 1] for r in range(w - 1))
    s = hx.batch_schedule(0, 100000, 8192)
    assert sum(s) == 100000 and s[0] == 1 and max(s) == 8192
    size = 0
    for b in s:
        assert b <= max(1, size // 8)
        size += b

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


def test_bench_launcher_takes_the_other_ranks_down_when_one_dies():
    """`python bench.py --gpus N` started without a launcher watches the rank processes it starts: the first one that exits non-zero ends the others and
    the launcher exits non-zero within seconds -- nobody is left inside a collective waiting for a dead peer (here: no GPU, so every rank dies at
    torch.cuda.set_device; on a GPU box a rank that faults mid-build takes the same path)."""
    import time
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check needs a box without a GPU (the ranks must fail)")
    root = os.path.dirname(HERE)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--rows", "1000", "--queries", "10", "--no-cpu"],
                         cwd=root, capture_output=True, text=True, timeout=240)
    assert out.returncode != 0
    assert "the other ranks were stopped" in out.stderr, out.stderr[-1500:]
    assert out.stdout.strip() == ""                                     # no JSON line from a failed job
    assert time.time() - t0 < 200


def test_bring_up_agrees_on_gloo_when_rccl_is_not_asked_for():
    """dist_build.bring_up(want="gloo") with two ranks on CPU: rendezvous with an explicit timeout, a Comm over the world group, collectives work."""
    port = free_port()
    code = ("import os, sys, importlib, torch; sys.path.insert(0, %r); db = importlib.import_module('pgvector-rx_amd.dist_build'); "
            "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']); c, dev = db.bring_up(r, w, torch.device('cpu'), want='gloo', gloo_timeout_s=60); "
            "t = torch.tensor([r + 1]); c.all_reduce(t); out = torch.empty(w, dtype=torch.int64); c.all_gather_into_tensor(out, torch.tensor([r])); "
            "print('OK', c.backend, dev.type, int(t.item()), out.tolist(), c.get_world_size(), c.get_rank())") % os.path.dirname(HERE)
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for r, p in enumerate(procs):
        o, _ = p.communicate(timeout=120)
        assert p.returncode == 0, o
        assert "OK gloo cpu 3 [0, 1] 2 %d" % r in o, o

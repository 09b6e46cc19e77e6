"""The multi-process build with the real engine: N rank processes (all on cuda:0, collectives over gloo) through
pgvector-rx_amd/dist_build.insert_sharded must each end with the graph of the single-process build -- element for element, distance
bits, heap TIDs and entry point -- for both exchange formats (device records; serialized host buffers)."""
import os
import re
import socket
import subprocess
import sys

import pytest

import pgvector_rx_amd as hx

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(world, args):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_gpu_worker.py")] + [str(a) for a in args],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=150)
        assert p.returncode == 0, o
        outs.append(re.search(r"DIGEST (\w+) size=(\d+) elems=(\w+) device_batches=(\d+) fused_redone=(\d+)", o).groups())
    return outs


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("fmt", [1, 0], ids=["device-records", "host-buffers"])
@pytest.mark.parametrize("shape", [(4000, 48, 16, 64, 512, hx.F32, hx.L2SQ), (3000, 256, 8, 32, 300, hx.BIT, hx.HAMMING)], ids=["f32-l2-m16", "bit-hamming-m8"])
def test_multi_process_build_equals_single_process(world, fmt, shape):
    n = shape[0]
    single = run_world(1, list(shape) + [fmt])[0]
    outs = run_world(world, list(shape) + [fmt])
    assert all(int(o[1]) == n for o in outs)
    assert all(o[0] == single[0] and o[2] == single[2] for o in outs), (single, outs)
    if fmt:
        assert all(int(o[3]) > 0 for o in outs)          # the device-record exchange was the one exercised
    assert all(int(o[4]) == 0 for o in outs)

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
        outs.append(re.search(r"DIGEST (\w+) size=(\d+) elems=(\w+) device_batches=(\d+) fused_redone=(\d+) backend=(\w+)", o).groups())
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


def test_multi_process_build_with_lists_of_33_to_64_slots():
    """m = 20 (lists of 40): the device batch pipeline prunes these lists with k_list_ops' pair memo on the mirror (hx_biglist.hip) and hands the pruned lists
    to the other ranks as device records -- same graph as one process builds."""
    shape = (2500, 32, 20, 48, 400, hx.F32, hx.L2SQ)
    single = run_world(1, list(shape) + [1])[0]
    outs = run_world(2, list(shape) + [1])
    assert all(int(o[1]) == shape[0] for o in outs)
    assert all(o[0] == single[0] and o[2] == single[2] for o in outs), (single, outs)
    assert all(int(o[3]) > 0 and int(o[4]) == 0 for o in outs)


def test_multi_process_build_with_lists_of_more_than_64_slots():
    """m = 40: the device batch pipeline does not serve the shape, so the ranks exchange serialized lists (hx_index_batch_*); searches, select_neighbors and
    the owned lists' back-links still run in device kernels on every rank (hx_biglist.hip) -- same graph as one process builds."""
    shape = (1800, 16, 40, 96, 300, hx.F32, hx.L2SQ)
    single = run_world(1, list(shape) + [1])[0]
    outs = run_world(2, list(shape) + [1])
    assert all(int(o[1]) == shape[0] for o in outs)
    assert all(o[0] == single[0] and o[2] == single[2] for o in outs), (single, outs)
    assert all(int(o[3]) == 0 and int(o[4]) == 0 for o in outs)


@pytest.mark.parametrize("world", [2, 3])
def test_replicated_step_after_sharded_batches(world):
    """A schedule that ENDS in replicated steps after sharded device batches (a later small insert on the same index): lists that other ranks
    pruned arrived through k_import_recs, so this rank's resident pair matrices of them are stale and must have been invalidated -- otherwise
    the replicated prunes read wrong pair distances, differently on every rank, and the replicas diverge silently.  m = 16 is the shape that
    keeps pair matrices resident; the hubs of the ramp-up are the lists every later step prunes again."""
    shape = [6000, 24, 16, 64, 512, hx.F32, hx.L2SQ, 1]
    single = run_world(1, shape + [700])[0]
    outs = run_world(world, shape + [700])
    assert all(int(o[1]) == 6000 for o in outs)
    assert all(o[0] == single[0] and o[2] == single[2] for o in outs), (single, outs)
    assert all(int(o[3]) > 0 for o in outs)


def test_rccl_path_with_one_rank():
    """The RCCL branch end to end on the one GPU of this box: an nccl process group of world size 1 brought up exactly as bench.py does
    (dist_build.bring_up: gloo rendezvous, RCCL group, agreement, probe all_reduce), and insert_sharded forced through the device-batch
    stages -- all_gather_into_tensor on device buffers over RCCL, the torch-stream <-> engine-stream hand-offs, the pruned-list export and
    import calls -- so that none of it runs for the first time in the driver's 8-GPU job.  The graph must equal the plain single-process build."""
    shape = [4000, 48, 16, 64, 512, hx.F32, hx.L2SQ, 1]
    single = run_world(1, shape)[0]
    got = run_world(1, shape + [0, "nccl", 1])[0]
    assert got[5] == "nccl", got                      # RCCL came up; a silent fall-back to gloo would hide the branch again
    assert int(got[3]) > 0                            # device batches took the sharded stages
    assert got[0] == single[0] and got[2] == single[2] and int(got[1]) == 4000

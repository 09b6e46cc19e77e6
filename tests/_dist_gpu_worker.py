"""Worker for test_dist_gpu.py: one rank of a multi-process build with the REAL engine (every rank on cuda:0, collectives over
gloo), through pgvector-rx_amd/dist_build.insert_sharded.  Prints a digest of the rank's final graph."""
import hashlib
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault("HX_DIST_WTABS", "2")      # exchange the W tables also over the gloo staging of these one-GPU rehearsals
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgvector_rx_amd as hx  # noqa: E402

db = importlib.import_module("pgvector-rx_amd.dist_build")


def graph_digest(ix, n):
    h = hashlib.sha256()
    lv = ix.export_levels()
    h.update(lv.tobytes())
    h.update(np.int64(ix.entry).tobytes())
    for layer in range(int(max(lv.max(), 0)) + 1):
        ids, d, cnt = ix.export_layer(layer)
        lm = ids.shape[1]
        valid = np.arange(lm)[None, :] < cnt[:, None]
        h.update(cnt.tobytes())
        h.update(np.where(valid, ids, 0).tobytes())
        h.update(np.where(valid, d.view(np.uint32), 0).tobytes())
    for i in range(n):
        h.update(np.asarray(ix.heaptids(i), np.int64).tobytes())
    return h.hexdigest()


class OneRank:
    @staticmethod
    def get_world_size():
        return 1

    @staticmethod
    def get_rank():
        return 0


def main():
    n, dim, m, efc, batch, dt, metric, fmt = (int(x) for x in sys.argv[1:9])
    tail = int(sys.argv[9]) if len(sys.argv) > 9 else 0          # rows inserted by a second, REPLICATED insert_sharded call after the sharded batches
    backend = sys.argv[10] if len(sys.argv) > 10 else "gloo"     # "nccl": RCCL data path brought up exactly as bench.py does (dist_build.bring_up)
    shard_one = len(sys.argv) > 11 and sys.argv[11] == "1"       # a world of one rank still runs the sharded stages
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    rng = np.random.default_rng(123)
    if dt == hx.F32:
        rows = rng.random((n, dim)).astype(np.float32)
    elif dt == hx.F16:
        rows = rng.random((n, dim)).astype(np.float16).view(np.uint16)
    else:
        rows = np.packbits(rng.integers(0, 2, (n, dim)).astype(np.uint8), axis=1, bitorder="big")
    rows[n // 2] = rows[7]                 # a duplicate of an old row, and a pair of identical rows inside one batch
    rows[n // 2 + 5] = rows[n // 2 + 3]
    levels = hx.draw_levels(n, m, seed=5)
    gpu = torch.device("cuda", 0)
    xdev = torch.device("cpu")
    used = "none"
    if world > 1 or backend == "nccl":
        torch.cuda.set_device(gpu)
        d, xdev = db.bring_up(rank, world, gpu, want=backend, log=lambda msg: print(msg, flush=True))
        used = d.backend
    else:
        d = OneRank
    e = hx.Engine(dt, metric, dim, n, device=0)
    e.append(rows)
    ix = hx.Index(e, m, efc)
    head = n - tail
    elems = db.insert_sharded(ix, 0, levels[:head], batch, d, xdev, min_shard=16, gpu=gpu if fmt else None, shard_single_rank=shard_one)
    stages = dict(db.STAGE_SECONDS)
    if tail:   # a small later insert on the same index: below min_shard, so every rank performs it redundantly ON ITS OWN replica of the graph
        elems = np.concatenate([elems, db.insert_sharded(ix, head, levels[head:], batch, d, xdev, min_shard=1 << 30, gpu=gpu if fmt else None)])
    print("DIGEST %s size=%d elems=%s device_batches=%d fused_redone=%d backend=%s" % (
        graph_digest(ix, n), ix.size, hashlib.sha256(elems.tobytes()).hexdigest()[:16], int(stages.get("device_batches", 0)), ix.fused_stats()["redone"], used), flush=True)
    if world > 1 or backend == "nccl":
        dist.barrier()
        dist.destroy_process_group()
    ix.close()
    e.close()


if __name__ == "__main__":
    main()

"""Multi-GPU index build: one lock-step batch of inserts shared by the ranks of a node.

The reference builds strictly sequentially on one core (amcanbuildparallel = false, src/index/handler.rs:153-154).
Here every rank (one process per GPU, torch.distributed, backend "nccl" = RCCL over xGMI) keeps a replica of the
rows (288 GB of HBM per GPU makes that free) and of the graph, and each batch is split two ways:

  search stage   rank r runs find_element_neighbors for a contiguous slice of the batch; the members' new
                 neighbour lists are exchanged with ONE all_gather (fixed-size records, padded per rank)
  links stage    every rank applies duplicate merge / entry-point update identically, then groups and prunes only the
                 back-link lists it owns (owner = target row id % world); the updated lists travel as self-describing
                 records (target, layer, list) in ONE all_gather (plus a tiny all_gather of the payload sizes)

so the graph after every batch is bit-identical on all ranks and identical to the single-GPU build with the same
batch schedule.  The only data-path collectives are those two all_gathers per batch.  Batches smaller than
`min_shard` members are built redundantly on every rank (a collective would cost more than it saves).

Two exchange formats.  Device-resident (default whenever hx_index_dbatch_supported: m <= 32, rows <= 8 KiB, no level draw beyond
the traversal kernel's layers): k_fused<insert> writes the members' lists into a torch-allocated DEVICE buffer, RCCL all-gathers
that buffer, every rank scatters it into its graph copy on the device, the back-link kernels write the lists they pruned as
device records, RCCL all-gathers those, and a kernel scatters the other ranks' records -- no list ever visits the host.  Host
format (hx_index_batch_*: serialized host buffers) remains for the batches the device kernels do not serve.
"""
import time

import numpy as np
import torch

import os

STAGE_SECONDS = {}      # wall time per stage of the last insert_sharded call (rank-local, for bench.py's report)
WTAB_EXCHANGE = int(os.environ.get("HX_DIST_WTABS", "1"))   # 1: exchange the W tables when the data path is RCCL (over gloo's host staging they cost more than they save); 2: always (tests); 0: never


class Comm:
    """The process group the build's collectives run on (insert_sharded only needs these calls)."""

    def __init__(self, dist, group, backend):
        self.dist, self.group, self.backend = dist, group, backend

    def get_world_size(self):
        return self.dist.get_world_size(self.group)

    def get_rank(self):
        return self.dist.get_rank(self.group)

    def all_gather_into_tensor(self, out, inp):
        return self.dist.all_gather_into_tensor(out, inp, group=self.group)

    def all_reduce(self, t, op=None):
        return self.dist.all_reduce(t, op=op or self.dist.ReduceOp.SUM, group=self.group)

    def barrier(self):
        return self.dist.barrier(group=self.group)


def bring_up(rank, world, dev, want="nccl", gloo_timeout_s=300, nccl_timeout_s=120, log=None):
    """Process groups of a multi-rank build: gloo first (rendezvous, timing reductions, and the place where the ranks AGREE on the
    data-path backend), then -- want == "nccl" -- an RCCL group for the data path.  The ranks agree over gloo that every one of them
    CREATED its RCCL group before any of them issues an RCCL collective, and agree again on the probe's result, so a rank that cannot
    bring RCCL up never leaves its peers inside an RCCL collective: all of them fall back to gloo together.  Every group has an explicit
    timeout (a rank that died takes its peers out in minutes, not in gloo's default half hour; the launcher's watchdog is faster still).
    Returns (Comm, device the data-path collectives want their tensors on)."""
    import datetime
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=gloo_timeout_s))
    comm = Comm(dist, dist.group.WORLD, "gloo")
    if want != "nccl":
        return comm, torch.device("cpu")

    def agree(ok):
        flag = torch.tensor([int(ok)], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    grp, why = None, ""
    try:   # RCCL over xGMI
        try:
            grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=nccl_timeout_s), device_id=dev)
        except TypeError:
            grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=nccl_timeout_s))
    except Exception as ex:   # noqa: BLE001
        why = str(ex).splitlines()[0] if str(ex) else type(ex).__name__
    created = agree(grp is not None)
    ok = False
    if created:
        try:
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe, group=grp)
            torch.cuda.synchronize(dev)
            ok = int(probe.item()) == world
        except Exception as ex:   # noqa: BLE001
            why = str(ex).splitlines()[0] if str(ex) else type(ex).__name__
        ok = agree(ok)
    if ok:
        return Comm(dist, grp, "nccl"), dev
    if log and why:
        log("rank %d could not bring up the nccl (RCCL) backend (%s); every rank uses gloo" % (rank, why))
    return comm, torch.device("cpu")


def _t(name, t0):
    STAGE_SECONDS[name] = STAGE_SECONDS.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()


def slice_bounds(b, world):
    """Contiguous member slices: rank r searches [lo[r], hi[r]) of a batch of b members."""
    per = -(-b // world)
    lo = [min(b, r * per) for r in range(world)]
    hi = [min(b, r * per + per) for r in range(world)]
    return lo, hi


def exchange(payload, sizes, rank, dist, device):
    """All-gathers one variable-size byte payload per rank (sizes known to every rank without communication).
    Returns the list of payloads (numpy uint8), own entry included."""
    world = len(sizes)
    cap = max(max(sizes), 1)
    send = torch.zeros(cap, dtype=torch.uint8, device=device)
    if sizes[rank]:
        send[:sizes[rank]] = torch.from_numpy(payload).to(device)
    recv = torch.empty(world * cap, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(recv, send)
    host = recv.cpu().numpy()
    return [host[r * cap:r * cap + sizes[r]] for r in range(world)]


def gather_sizes(n, dist, device):
    """Payload sizes of all ranks (the links stage groups only the ops a rank owns, so sizes are not known elsewhere)."""
    world = dist.get_world_size()
    out = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, torch.tensor([n], dtype=torch.int64, device=device))
    return [int(x) for x in out.cpu().tolist()]


def all_gather_device(send, world, dist, device):
    """All-gathers equal-sized uint8 device tensors.  `device` is where the backend wants its tensors (the GPU for nccl = RCCL over
    xGMI; the CPU for gloo rehearsals, which stage through host memory)."""
    recv = torch.empty(world * send.numel(), dtype=torch.uint8, device=send.device)
    if torch.device(device).type == "cuda":
        dist.all_gather_into_tensor(recv, send)
        torch.cuda.current_stream(send.device).synchronize()      # the engine reads the buffer on its own stream
    else:
        host = torch.empty(world * send.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(host, send.cpu())
        recv.copy_(host)
        if send.is_cuda:
            torch.cuda.synchronize(send.device)
    return recv


def _device_batch(ix, first_row, levels, tids, dist, device, gpu, world, rank):
    """One batch through the device-resident stages; returns the elements holding the batch's tids."""
    b = len(levels)
    per = -(-b // world)                                          # member i lives in record i: rank r fills records [r*per, r*per+per)
    lo, hi = min(b, rank * per), min(b, rank * per + per)
    rb, lb = ix.dbatch_record_bytes, ix.dbatch_list_record_bytes
    t0 = time.perf_counter()
    ix.dbatch_begin(first_row, levels, tids)
    t0 = _t("begin", t0)
    on_gpu = torch.device(gpu).type == "cuda"                      # (a CPU "device" only in the CPU-side test of this exchange)
    send = torch.zeros(per * rb, dtype=torch.uint8, device=gpu)
    if on_gpu:
        torch.cuda.current_stream(gpu).synchronize()              # the engine runs on its own (non-blocking) stream
    ix.dbatch_search(lo, hi, send.data_ptr())
    t0 = _t("search", t0)
    recs = all_gather_device(send, world, dist, device)
    t0 = _t("allgather_new", t0)
    # the members' W tables travel too (4 KB each at ef_construction 200): every rank then prunes the lists it owns with the look-ups a single GPU
    # has for ALL members instead of streaming the rows of the members other ranks searched (HX_DIST_WTABS: see above)
    wb = getattr(ix, "dbatch_wtab_bytes", 0) if world > 1 and (WTAB_EXCHANGE == 2 or (WTAB_EXCHANGE == 1 and torch.device(device).type == "cuda")) else 0
    if wb:
        wsend = torch.zeros(per * wb, dtype=torch.uint8, device=gpu)
        if on_gpu:
            torch.cuda.current_stream(gpu).synchronize()
        ix.dbatch_export_wtabs(lo, hi, wsend.data_ptr())
        wrecv = all_gather_device(wsend, world, dist, device)
        for r in range(world):
            rlo, rhi = min(b, r * per), min(b, r * per + per)
            if r != rank and rhi > rlo:
                ix.dbatch_import_wtabs(rlo, rhi, wrecv.data_ptr() + r * per * wb)
        t0 = _t("allgather_wtabs", t0)
    mine = ix.dbatch_links(rank, world, recs.data_ptr())
    t0 = _t("links", t0)
    sizes = gather_sizes(mine, dist, device)
    cap = max(max(sizes), 1)
    lsend = torch.zeros(cap * lb, dtype=torch.uint8, device=gpu)
    if on_gpu:
        torch.cuda.current_stream(gpu).synchronize()
    ix.dbatch_export_links(lsend.data_ptr())
    t0 = _t("export_links", t0)
    lrecv = all_gather_device(lsend, world, dist, device)
    t0 = _t("allgather_links", t0)
    for r in range(world):
        if r != rank and sizes[r]:
            ix.dbatch_import_links(lrecv.data_ptr() + r * cap * lb, sizes[r])
    out = ix.dbatch_end(b)                                        # synchronises the engine's stream: the buffers may go
    _t("import_links", t0)
    return out


def insert_sharded(ix, first_row, levels, batch, dist, device, tids=None, min_shard=256, size_fn=None, gpu=None, shard_single_rank=False):
    """hx_index_insert for rows [first_row, first_row + len(levels)) with every batch shared by the ranks.
    `ix` exposes the staged batch API of binding.Index (tests drive this with a stand-in object and gloo).
    gpu: torch device of this rank's engine; given, batches the device kernels serve exchange device buffers.
    shard_single_rank: a world of ONE rank still takes the sharded stages (the one-GPU rehearsal of the RCCL path: group bring-up,
    all_gather_into_tensor on device buffers and the stream hand-offs run exactly as they do at 8 ranks)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    levels = np.ascontiguousarray(levels, np.int32)
    n = len(levels)
    tids = np.arange(first_row, first_row + n, dtype=np.int64) if tids is None else np.ascontiguousarray(tids, np.int64)
    elems = np.empty(n, np.uint32)
    done = 0
    STAGE_SECONDS.clear()
    while done < n:
        size = ix.size
        b = min(batch, n - done, max(1, size // 8))           # same ramp-up rule as hx_index_insert
        if ix.entry < 0 or b < min_shard or (world == 1 and not shard_single_rank):
            # replicated: every rank performs the identical single-GPU step
            t0 = time.perf_counter()
            elems[done:done + b] = ix.insert(first_row + done, levels[done:done + b], tids[done:done + b], batch=b)
            _t("replicated_small_batches", t0)
            done += b
            continue
        if gpu is not None and ix.dbatch_supported(levels[done:done + b]):
            elems[done:done + b] = _device_batch(ix, first_row + done, levels[done:done + b], tids[done:done + b], dist, device, gpu, world, rank)
            STAGE_SECONDS["device_batches"] = STAGE_SECONDS.get("device_batches", 0) + 1
            done += b
            continue
        lo, hi = slice_bounds(b, world)
        t0 = time.perf_counter()
        ix.batch_begin(first_row + done, levels[done:done + b], tids[done:done + b])
        t0 = _t("begin", t0)
        ix.batch_search(lo[rank], hi[rank])
        t0 = _t("search", t0)
        sizes = [ix.batch_new_bytes(lo[r], hi[r]) for r in range(world)]
        mine = ix.batch_export_new(lo[rank], hi[rank])
        t0 = _t("export_new", t0)
        bufs = exchange(mine, sizes, rank, dist, device)
        t0 = _t("allgather_new", t0)
        for r, buf in enumerate(bufs):
            if r != rank and sizes[r]:
                ix.batch_import_new(lo[r], hi[r], buf)
        t0 = _t("import_new", t0)
        ix.batch_links(rank, world)
        t0 = _t("links", t0)
        mine = ix.batch_export_links()
        sizes = gather_sizes(len(mine), dist, device)
        t0 = _t("export_links", t0)
        bufs = exchange(mine, sizes, rank, dist, device)
        t0 = _t("allgather_links", t0)
        for r, buf in enumerate(bufs):
            if r != rank and sizes[r]:
                ix.batch_import_links(buf)
        t0 = _t("import_links", t0)
        elems[done:done + b] = ix.batch_end(b)
        done += b
    return elems

"""Worker for test_dist_gloo.py: runs pgvector-rx_amd/dist_build.insert_sharded over gloo (CPU) against a
stand-in for the staged batch API, and prints a digest of the final replicated state."""
import hashlib
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
db = importlib.import_module("pgvector-rx_amd.dist_build")

M = 4


def lm(layer):
    return 2 * M if layer == 0 else M


def mix(*xs):
    h = hashlib.blake2b(np.asarray(xs, np.uint64).tobytes(), digest_size=8).digest()
    return int.from_bytes(h, "little") | 1


class FakeIndex:
    """Same call protocol and payload sizes as binding.Index's staged batch API; the 'lists' are 8-byte tokens that
    depend on everything the real stage would depend on (member id, its level, the previous state of the target)."""

    def __init__(self, levels_all):
        self.levels_all = levels_all
        self.size = 0
        self.entry = -1
        self.new_tok = {}      # element -> token of its neighbour lists
        self.link_tok = {}     # target -> token of its (pruned) list
        self.calls = {"search": 0, "links_owned": 0, "replicated": 0}
        self.open = None
        self.use_dbatch = False

    # --- single-GPU path: begin, search(0,b), links(0,1), end -- exactly what hx_index_insert does
    def insert(self, first_row, levels, tids, batch):
        levels = list(levels)
        out = []
        if self.entry < 0:
            self.entry = first_row
            self.size += 1
            out.append(first_row)
            first_row, levels = first_row + 1, levels[1:]
        if levels:
            self.batch_begin(first_row, levels, None)
            self.batch_search(0, len(levels))
            self.calls["search"] -= len(levels)
            self.batch_links(0, 1)
            out.extend(self.batch_end(len(levels)).tolist())
        self.calls["replicated"] += len(out)
        return np.asarray(out, np.uint32)

    def _targets(self, e):
        return sorted({(e * 7 + k * 13) % self.batch_base for k in range(3)})

    # --- staged path
    def batch_begin(self, first_row, levels, tids):
        assert self.open is None and first_row == self.size
        self.open = {"first": first_row, "levels": list(levels), "got": set()}
        self.batch_base = first_row
        self.size += len(levels)

    def _bytes(self, lv):
        return sum(4 + lm(l) * 8 for l in range(lv + 1))

    def batch_search(self, lo, hi):
        for i in range(lo, hi):
            e = self.open["first"] + i
            self.new_tok[e] = mix(e, self.open["levels"][i])
            self.open["got"].add(i)
        self.calls["search"] += hi - lo

    def batch_new_bytes(self, lo, hi):
        return sum(self._bytes(self.open["levels"][i]) for i in range(lo, hi))

    def batch_export_new(self, lo, hi):
        parts = []
        for i in range(lo, hi):
            assert i in self.open["got"]
            n = self._bytes(self.open["levels"][i])
            parts.append(np.resize(np.frombuffer(self.new_tok[self.open["first"] + i].to_bytes(8, "little"), np.uint8), n))
        return np.concatenate(parts) if parts else np.empty(0, np.uint8)

    def batch_import_new(self, lo, hi, buf):
        assert len(buf) == self.batch_new_bytes(lo, hi)
        off = 0
        for i in range(lo, hi):
            n = self._bytes(self.open["levels"][i])
            rec = bytes(buf[off:off + n])
            assert rec == bytes(np.resize(np.frombuffer(rec[:8], np.uint8), n))      # padding/slicing intact
            self.new_tok[self.open["first"] + i] = int.from_bytes(rec[:8], "little")
            self.open["got"].add(i)
            off += n

    def _groups(self):
        ops = {}
        for i in range(len(self.open["levels"])):
            e = self.open["first"] + i
            for t in self._targets(e):
                ops.setdefault(t, []).append(e)
        return sorted(ops.items())

    def batch_links(self, rank, world):
        assert self.open["got"] == set(range(len(self.open["levels"])))
        self.open["groups"] = self._groups()
        self.open["rank"], self.open["world"] = rank, world
        for t, es in self.open["groups"]:
            if t % world == rank:
                tok = self.link_tok.get(t, 0)
                for e in es:
                    tok = mix(t, tok, e)
                self.link_tok[t] = tok
                self.calls["links_owned"] += 1

    def batch_links_bytes(self):
        rank, world = self.open["rank"], self.open["world"]
        return sum(8 + 4 + lm(0) * 8 for t, _ in self.open["groups"] if t % world == rank)

    def batch_export_links(self):
        rank, world = self.open["rank"], self.open["world"]
        n = 4 + lm(0) * 8
        parts = []
        for t, _ in self.open["groups"]:
            if t % world == rank:
                parts.append(np.frombuffer(np.asarray([t, 0], np.uint32).tobytes(), np.uint8))
                parts.append(np.resize(np.frombuffer(self.link_tok[t].to_bytes(8, "little"), np.uint8), n))
        return np.concatenate(parts) if parts else np.empty(0, np.uint8)

    def batch_import_links(self, buf):
        n, off = 4 + lm(0) * 8, 0
        while off < len(buf):
            t = int(np.frombuffer(bytes(buf[off:off + 4]), np.uint32)[0])
            assert t % self.open["world"] != self.open["rank"]                 # only other ranks' lists arrive
            self.link_tok[t] = int.from_bytes(bytes(buf[off + 8:off + 16]), "little")
            off += 8 + n
        assert off == len(buf)

    def batch_end(self, b):
        out = np.arange(self.open["first"], self.open["first"] + b, dtype=np.uint32)
        self.open = None
        return out

    # --- device-resident staged path (hx_index_dbatch_*): same protocol, the "device" buffers are host memory reached through raw pointers
    dbatch_record_bytes = 24
    dbatch_list_record_bytes = 16

    def dbatch_supported(self, levels):
        return self.use_dbatch

    @staticmethod
    def _view(ptr, nbytes):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(ptr))

    def dbatch_begin(self, first_row, levels, tids):
        self.batch_begin(first_row, levels, tids)

    def dbatch_search(self, lo, hi, d_records):
        self.batch_search(lo, hi)
        v = self._view(d_records, (hi - lo) * self.dbatch_record_bytes).view(np.uint64).reshape(hi - lo, 3)
        for i in range(lo, hi):
            e = self.open["first"] + i
            v[i - lo] = (e, self.open["levels"][i], self.new_tok[e])

    def dbatch_links(self, rank, world, d_records):
        b = len(self.open["levels"])
        v = self._view(d_records, b * self.dbatch_record_bytes).view(np.uint64).reshape(b, 3)
        for i in range(b):                                   # every member's record arrived, in member order, intact
            e = self.open["first"] + i
            assert int(v[i, 0]) == e and int(v[i, 1]) == self.open["levels"][i], (i, v[i])
            self.new_tok[e] = int(v[i, 2])
            self.open["got"].add(i)
        self.batch_links(rank, world)
        self.open["mine"] = [(t, self.link_tok[t]) for t, _ in self.open["groups"] if t % world == rank]
        return len(self.open["mine"])

    def dbatch_export_links(self, d_out):
        m = self.open["mine"]
        if m:
            self._view(d_out, len(m) * self.dbatch_list_record_bytes).view(np.uint64).reshape(len(m), 2)[:] = np.asarray(m, np.uint64)

    def dbatch_import_links(self, d_list_records, n):
        v = self._view(d_list_records, n * self.dbatch_list_record_bytes).view(np.uint64).reshape(n, 2)
        for t, tok in v.tolist():
            assert t % self.open["world"] != self.open["rank"]
            self.link_tok[int(t)] = int(tok)

    def dbatch_end(self, b):
        return self.batch_end(b)

    def digest(self):
        h = hashlib.sha256()
        for k in sorted(self.new_tok):
            h.update(np.asarray([k, self.new_tok[k]], np.uint64).tobytes())
        for k in sorted(self.link_tok):
            h.update(np.asarray([k, self.link_tok[k]], np.uint64).tobytes())
        return h.hexdigest()


class OneRank:
    @staticmethod
    def get_world_size():
        return 1

    @staticmethod
    def get_rank():
        return 0


def main():
    n, batch = int(sys.argv[1]), int(sys.argv[2])
    device_format = len(sys.argv) > 3 and sys.argv[3] == "device"
    rng = np.random.default_rng(5)
    levels = np.minimum(rng.geometric(0.75, n) - 1, 4).astype(np.int32)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=world)
        d = dist
    else:
        d = OneRank
    ix = FakeIndex(levels)
    ix.use_dbatch = device_format
    elems = db.insert_sharded(ix, 0, levels, batch, d, torch.device("cpu"), min_shard=16, gpu=torch.device("cpu") if device_format else None)
    assert elems.tolist() == list(range(n))
    print("DIGEST %s size=%d search=%d owned=%d replicated=%d" % (ix.digest(), ix.size, ix.calls["search"], ix.calls["links_owned"], ix.calls["replicated"]), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

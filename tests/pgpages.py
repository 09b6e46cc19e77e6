"""Reader for the HNSW index page image, written the way the reference's SCAN path reads pages (not the way the
writers under test produce them): element payload at etup + 72 (scan.rs:188), neighbour TIDs at
ntup + 4 + ((level - layer) * m + i) * 6 (scan.rs:268-279), page chain through the special area (types/hnsw.rs:17-27),
meta page right after the 24-byte page header (build.rs:82-86).  Test infrastructure only."""
import struct

import numpy as np

BLCKSZ, HDR, INVALID_BLK = 8192, 24, 0xFFFFFFFF


def tid(buf, o):
    hi, lo, pos = struct.unpack_from("<HHH", buf, o)
    return (hi << 16) | lo, pos


def page_header(pg):
    lsn_hi, lsn_lo, checksum, flags, lower, upper, special, psv, prune = struct.unpack_from("<IIHHHHHHI", pg, 0)
    return dict(lsn=(lsn_hi, lsn_lo), checksum=checksum, flags=flags, lower=lower, upper=upper, special=special,
                page_size=psv & 0xFF00, layout_version=psv & 0xFF, prune_xid=prune)


def items(pg):
    h = page_header(pg)
    out = []
    for i in range((h["lower"] - HDR) // 4):
        lp, = struct.unpack_from("<I", pg, HDR + 4 * i)
        out.append((lp & 0x7FFF, (lp >> 15) & 3, lp >> 17))      # lp_off, lp_flags, lp_len
    return out


def decode(pages, m):
    """-> meta dict, elements {(blk, off): dict(level, heaptids, neighbortid, value bytes)}, neighbours {(blk, off): [tid...]}, chain [blk...]"""
    pages = [bytes(p) for p in np.asarray(pages, np.uint8)]
    mh = page_header(pages[0])
    magic, version, dims, mm, efc, eblk, eoff, elevel, ins = struct.unpack_from("<IIIHHIHhI", pages[0], HDR)
    meta = dict(magic=magic, version=version, dimensions=dims, m=mm, ef_construction=efc, entry=(eblk, eoff), entry_level=elevel,
                insert_page=ins, pd_lower=mh["lower"])
    elements, neigh, chain = {}, {}, []
    blk = 1
    while blk != INVALID_BLK:
        chain.append(blk)
        pg = pages[blk]
        h = page_header(pg)
        assert h["page_size"] == BLCKSZ and h["layout_version"] == 4 and h["special"] == BLCKSZ - 8 and h["lsn"] == (0, 0) and h["checksum"] == 0 and h["flags"] == 0
        nxt, unused, page_id = struct.unpack_from("<IHH", pg, h["special"])
        assert page_id == 0xFF90 and unused == 0
        its = items(pg)
        # line pointers: tuples packed downwards from the special area, MAXALIGNed, no overlap
        expect = h["special"]
        for (lo, fl, ln) in its:
            assert fl == 1 and lo % 8 == 0
            expect -= (ln + 7) & ~7
            assert lo == expect
        assert h["upper"] == expect and h["lower"] == HDR + 4 * len(its) and h["lower"] <= h["upper"]
        for i, (lo, fl, ln) in enumerate(its):
            t = pg[lo:lo + ln]
            if t[0] == 1:
                level, deleted, ver = t[1], t[2], t[3]
                hts = [tid(t, 4 + 6 * k) for k in range(10)]
                vl, = struct.unpack_from("<I", t, 72)
                assert vl & 3 == 0 and (vl >> 2) <= ln - 72
                elements[(blk, i + 1)] = dict(level=level, deleted=deleted, version=ver, heaptids=hts, neighbortid=tid(t, 64),
                                              unused=struct.unpack_from("<H", t, 70)[0], value=t[72:72 + (vl >> 2)], tuple_len=ln)
            elif t[0] == 2:
                count, = struct.unpack_from("<H", t, 2)
                neigh[(blk, i + 1)] = dict(version=t[1], count=count, raw=t, tuple_len=ln)
            else:
                raise AssertionError("unknown tuple type %d" % t[0])
        blk = nxt
    return meta, elements, neigh, chain


def neighbour_tids(ntup, level, layer, m):
    """load_neighbor_tids, scan.rs:236-283: the lm slots of `layer` (invalid TIDs = padding are dropped)."""
    lm = 2 * m if layer == 0 else m
    start = (level - layer) * m
    out = []
    for i in range(lm):
        b, o = tid(ntup["raw"], 4 + (start + i) * 6)
        if b != INVALID_BLK:
            out.append((b, o))
    return out

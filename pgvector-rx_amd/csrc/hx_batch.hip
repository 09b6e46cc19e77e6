// hx_batch.hip -- the device-resident half of one insert batch (build_callback, src/index/build.rs:400-535, for many rows at once).
//
// k_fused<insert> leaves every batch member's new neighbour lists in DEVICE records (one record per member:
// cnt[HX_FUSED_MAXL] | ids[HX_FUSED_MAXL][2m] | d[HX_FUSED_MAXL][2m]).  Everything between that launch and the back-link kernels used
// to pass through the host (lists copied out, re-uploaded into the graph mirror, back-link ops listed by host threads and
// uploaded).  Here it stays on the device:
//   * duplicate candidates (build.rs:482-512): the leading zero-distance layer-0 neighbours of every member, and members with
//     byte-identical rows inside the batch (row hashes, radix sort, equal neighbours in the sorted order), each confirmed by a
//     byte comparison of the two rows; only the (rare) confirmed pairs travel to the host, which decides merges in row order;
//   * k_apply_new scatters the members' lists into the graph mirror (a merged member becomes a tombstone) and counts the back-link
//     ops each member emits for the lists THIS rank owns (owner = target % world);
//   * k_emit_ops writes those ops (key = target << 7 | layer, new element, distance) in update_neighbor_connections order
//     (graph/mod.rs:451-458: member, layer descending, slot ascending) straight into the grouping arrays of hx_group.hip;
//   * k_import_recs scatters list records pruned by other ranks (multi-GPU exchange) into the mirror.
// A multi-GPU build all-gathers the member records and the pruned-list records as device buffers (pgvector-rx_amd/dist_build.py):
// no list ever visits the host during a build.
#include "hx_ops.h"

#include <hipcub/hipcub.hpp>

namespace {

__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

// 64-bit hash of a row's pitch bytes (padding is zero): one wave per row, position-salted fragment hashes summed across the wave
__global__ void k_row_hash(const uint8_t *__restrict__ rows, uint32_t pitch, uint32_t base, uint32_t b, unsigned long long *__restrict__ hash, uint32_t *__restrict__ idx)
{
    const uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (i >= b) return;
    const uint8_t *r = rows + (size_t)(base + i) * pitch;
    unsigned long long h = 0;
    for (uint32_t c = lane * 16u; c < pitch; c += 1024u) {
        const u4 v = *(const u4 *)(r + c);
        const unsigned long long pos = (unsigned long long)(c >> 4) * 0x9E3779B97F4A7C15ull;
        h += mix64(((unsigned long long)v[0] | ((unsigned long long)v[1] << 32)) ^ pos);
        h += mix64(((unsigned long long)v[2] | ((unsigned long long)v[3] << 32)) ^ (pos + 0x632BE59BD9B4E019ull));
    }
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned int lo = (unsigned int)__shfl_xor((int)(unsigned int)h, o, 64), hi = (unsigned int)__shfl_xor((int)(unsigned int)(h >> 32), o, 64);
        h += (unsigned long long)lo | ((unsigned long long)hi << 32);
    }
    if (lane == 0) { hash[i] = mix64(h); idx[i] = i; }
}

// members whose row hash equals their predecessor's in the (stable) sorted order: candidate pair (later member, earlier member)
__global__ void k_adjacent(const unsigned long long *__restrict__ hs, const uint32_t *__restrict__ is, uint32_t b, uint32_t base,
                           uint32_t *__restrict__ pa, uint32_t *__restrict__ pb, uint32_t *__restrict__ counter)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0 || p >= b || hs[p] != hs[p - 1]) return;
    const uint32_t s = atomicAdd(counter, 1u);
    pa[s] = base + is[p]; pb[s] = base + is[p - 1];
}

// leading zero-distance layer-0 neighbours of every member (build.rs:484-490 breaks at the first non-zero distance)
__global__ void k_zero_prefix(const uint32_t *__restrict__ rec, uint32_t rw, uint32_t m, uint32_t base, uint32_t b,
                              uint32_t *__restrict__ pa, uint32_t *__restrict__ pb, uint32_t *__restrict__ counter)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b) return;
    const uint32_t *r = rec + (size_t)i * rw;
    const uint32_t c = r[0], lm0 = 2u * m;
    const uint32_t *ids = r + HX_FUSED_MAXL; const float *d = (const float *)(r + HX_FUSED_MAXL + HX_FUSED_MAXL * lm0);
    uint32_t z = 0;
    while (z < c && !(d[z] != 0.0f)) z++;
    if (!z) return;
    const uint32_t s = atomicAdd(counter, z);                  // one contiguous range per member, list order inside it
    for (uint32_t k = 0; k < z; k++) { pa[s + k] = base + i; pb[s + k] = ids[k]; }
}

__global__ void k_rows_equal_b(const uint8_t *__restrict__ rows, uint32_t pitch, uint32_t n_pairs,
                               const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint8_t *__restrict__ eq)
{
    const uint32_t p = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (p >= n_pairs) return;
    const uint8_t *ra = rows + (size_t)a[p] * pitch, *rb = rows + (size_t)b[p] * pitch;
    bool same = true;
    for (uint32_t c = lane * 16u; c < pitch; c += 1024u) {
        const u4 x = *(const u4 *)(ra + c), y = *(const u4 *)(rb + c);
        same = same && x[0] == y[0] && x[1] == y[1] && x[2] == y[2] && x[3] == y[3];
    }
    const unsigned long long diff = __ballot(!same);
    if (lane == 0) eq[p] = diff == 0ull;
}

struct MirrorPtrs {
    uint32_t *l0_ids; float *l0_d; uint16_t *l0_cnt; int32_t *level; const uint32_t *up_block; uint32_t *up_ids; float *up_d; uint16_t *up_cnt; uint8_t *pm_valid; uint32_t m;
};

// one wave per member: lists into the mirror (elements[new].neighbors[lc].items = neighbors, mod.rs:422), tombstone for a merged row,
// and the number of back-link ops it emits for lists this rank owns
__global__ void k_apply_new(const uint32_t *__restrict__ rec, uint32_t rw, uint32_t base, uint32_t b, const uint8_t *__restrict__ dup,
                            MirrorPtrs mp, uint32_t rank, uint32_t world, uint32_t *__restrict__ nops)
{
    const uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (i >= b) return;
    const uint32_t id = base + i, lm0 = 2u * mp.m;
    const uint32_t *r = rec + (size_t)i * rw;
    int lv = mp.level[id];
    if (lv < 0) lv = -1 - lv;
    if (lv >= HX_FUSED_MAXL) { if (lane == 0) nops[i] = 0; return; }     // never produced here: the host path owns such members
    const bool merged = dup && dup[i];
    uint32_t total = 0;
    for (int lc = 0; lc <= lv; lc++) {
        const uint32_t c = merged ? 0u : r[lc];
        const uint32_t *ids = r + HX_FUSED_MAXL + (uint32_t)lc * lm0; const uint32_t *d = r + HX_FUSED_MAXL + HX_FUSED_MAXL * lm0 + (uint32_t)lc * lm0;
        uint32_t nb = 0;
        if (lane < c) nb = ids[lane];
        if (lc == 0) {
            if (lane < c) { mp.l0_ids[(size_t)id * lm0 + lane] = nb; mp.l0_d[(size_t)id * lm0 + lane] = __builtin_bit_cast(float, d[lane]); }
            if (lane == 0) { mp.l0_cnt[id] = (uint16_t)c; if (mp.pm_valid) mp.pm_valid[id] = 0; }
        } else {
            const uint32_t blk = mp.up_block[id] + (uint32_t)(lc - 1);
            if (lane < c) { mp.up_ids[(size_t)blk * mp.m + lane] = nb; mp.up_d[(size_t)blk * mp.m + lane] = __builtin_bit_cast(float, d[lane]); }
            if (lane == 0) mp.up_cnt[blk] = (uint16_t)c;
        }
        total += (uint32_t)__popcll(__ballot(lane < c && nb % world == rank));
    }
    if (lane == 0) { nops[i] = total; mp.level[id] = merged ? -1 - lv : lv; }
}

// one wave per member: its back-link ops in update_neighbor_connections order (mod.rs:451-458), owned targets only
__global__ void k_emit_ops(const uint32_t *__restrict__ rec, uint32_t rw, uint32_t base, uint32_t b, const int32_t *__restrict__ level, uint32_t m,
                           const uint32_t *__restrict__ opstart, uint32_t rank, uint32_t world,
                           unsigned long long *__restrict__ keys, uint32_t *__restrict__ op_new, float *__restrict__ op_d)
{
    const uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (i >= b) return;
    const int lv = level[base + i];
    if (lv < 0 || lv >= HX_FUSED_MAXL) return;                  // tombstone: no links
    const uint32_t lm0 = 2u * m;
    const uint32_t *r = rec + (size_t)i * rw;
    uint32_t o = opstart[i];
    for (int lc = lv; lc >= 0; lc--) {
        const uint32_t c = r[lc];
        const uint32_t *ids = r + HX_FUSED_MAXL + (uint32_t)lc * lm0; const uint32_t *d = r + HX_FUSED_MAXL + HX_FUSED_MAXL * lm0 + (uint32_t)lc * lm0;
        uint32_t nb = 0; bool own = false;
        if (lane < c) { nb = ids[lane]; own = nb % world == rank; }
        const unsigned long long mk = __ballot(own);
        if (own) {
            const uint32_t s = o + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
            keys[s] = ((unsigned long long)nb << 7) | (unsigned long long)lc; op_new[s] = base + i; op_d[s] = __builtin_bit_cast(float, d[lane]);
        }
        o += (uint32_t)__popcll(mk);
    }
}

__global__ void k_total(const uint32_t *__restrict__ nops, const uint32_t *__restrict__ opstart, uint32_t b, uint32_t *__restrict__ out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = b ? opstart[b - 1] + nops[b - 1] : 0u;
}

// one wave per list record {target, layer, cnt, ids[2m], d[2m]} (a list another rank pruned): into the mirror
__global__ void k_import_recs(const uint32_t *__restrict__ xr, uint32_t xw, uint32_t n, MirrorPtrs mp)
{
    const uint32_t g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (g >= n) return;
    const uint32_t *r = xr + (size_t)g * xw;
    const uint32_t target = r[0], layer = r[1], c = r[2], lm0 = 2u * mp.m;
    if (layer == 0) {
        if (lane < c) { mp.l0_ids[(size_t)target * lm0 + lane] = r[3 + lane]; mp.l0_d[(size_t)target * lm0 + lane] = __builtin_bit_cast(float, r[3 + lm0 + lane]); }
        if (lane == 0) { mp.l0_cnt[target] = (uint16_t)c; if (mp.pm_valid) mp.pm_valid[target] = 0; }   // a list another rank pruned: this rank's cached pair matrix of it is stale (as k_mirror_lists does for host-written lists)
    } else {
        const uint32_t blk = mp.up_block[target] + layer - 1u;
        if (lane < c) { mp.up_ids[(size_t)blk * mp.m + lane] = r[3 + lane]; mp.up_d[(size_t)blk * mp.m + lane] = __builtin_bit_cast(float, r[3 + lm0 + lane]); }
        if (lane == 0) mp.up_cnt[blk] = (uint16_t)c;
    }
}

// a member's lists computed on the host (lock-step path: levels beyond the kernel's layers) written into its record
__global__ void k_fill_rec(uint32_t *__restrict__ rec, uint32_t rw, uint32_t slot, uint32_t m, const uint32_t *__restrict__ src)
{   // src: cnt[MAXL] | ids[MAXL][2m] | d[MAXL][2m], the record layout itself
    for (uint32_t w = threadIdx.x; w < HX_FUSED_MAXL + 2u * HX_FUSED_MAXL * 2u * m; w += blockDim.x) rec[(size_t)slot * rw + w] = src[w];
}

static MirrorPtrs mirror_ptrs(hx_engine *e)
{
    HxMirror &mr = e->mirror;
    return MirrorPtrs{mr.d_l0_ids, mr.d_l0_d, mr.d_l0_cnt, mr.d_level, mr.d_up_block, mr.d_up_ids, mr.d_up_d, mr.d_up_cnt, mr.d_pm_valid, mr.m};
}

static int bw_reserve(hx_engine *e, size_t dev_bytes, size_t host_bytes)
{
    HxBatchWork &w = e->bw;
    if (dev_bytes > w.cap) {
        HX_HIP(e, hipStreamSynchronize(e->stream));
        if (w.d) (void)hipFree(w.d);
        w.d = nullptr; w.cap = 0;
        HX_HIP(e, hipMalloc((void **)&w.d, dev_bytes * 2));
        w.cap = dev_bytes * 2;
    }
    if (host_bytes > w.cap_h) {
        HX_HIP(e, hipStreamSynchronize(e->stream));
        if (w.h) (void)hipHostFree(w.h);
        w.h = nullptr; w.cap_h = 0;
        HX_HIP(e, hipHostMalloc((void **)&w.h, host_bytes * 2, hipHostMallocDefault));
        w.cap_h = host_bytes * 2;
    }
    if (!w.h_ctr) HX_HIP(e, hipHostMalloc((void **)&w.h_ctr, 256, hipHostMallocDefault));
    return HX_OK;
}
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

uint32_t hx_rec_words(uint32_t m) { return (HX_FUSED_MAXL + 2u * HX_FUSED_MAXL * 2u * m + 3u) & ~3u; }

int hx_engine::db_reserve_records(uint64_t n_records)
{
    const size_t need = (size_t)n_records * hx_rec_words(mirror.m) * 4;
    if (need > bw.cap_rec) {
        HX_HIP(this, hipSetDevice(device));
        HX_HIP(this, hipStreamSynchronize(stream));
        if (bw.d_rec) (void)hipFree(bw.d_rec);
        bw.d_rec = nullptr; bw.cap_rec = 0;
        HX_HIP(this, hipMalloc((void **)&bw.d_rec, need + need / 2));
        bw.cap_rec = need + need / 2;
    }
    return HX_OK;
}

// W tables of the batch's members: 2^k >= 2 * ef_construction entries each (off above 512: the table is built in the traversal kernel's LDS)
int hx_engine::db_begin_wtabs(uint32_t base, uint32_t b, uint32_t ef_construction)
{
    static const bool off = getenv("HX_WTAB") && atoi(getenv("HX_WTAB")) == 0;
    bw.wt_size = 0; bw.wt_base = base; bw.wt_n = b;
    if (off || ef_construction > 512 || b == 0) return HX_OK;
    uint32_t T = 64; while (T < 2 * ef_construction) T <<= 1;
    HX_HIP(this, hipSetDevice(device));
    const size_t need = (size_t)b * T * 8;
    if (need > bw.cap_wtab) {
        HX_HIP(this, hipStreamSynchronize(stream));
        if (bw.d_wtab) (void)hipFree(bw.d_wtab);
        bw.d_wtab = nullptr; bw.cap_wtab = 0;
        HX_HIP(this, hipMalloc(&bw.d_wtab, need + need / 2));
        bw.cap_wtab = need + need / 2;
    }
    if (b > bw.cap_wtv) {
        HX_HIP(this, hipStreamSynchronize(stream));
        if (bw.d_wt_valid) (void)hipFree(bw.d_wt_valid);
        bw.d_wt_valid = nullptr; bw.cap_wtv = 0;
        HX_HIP(this, hipMalloc((void **)&bw.d_wt_valid, (size_t)b * 2));
        bw.cap_wtv = (size_t)b * 2;
    }
    HX_HIP(this, hipMemsetAsync(bw.d_wt_valid, 0, b, stream));
    bw.wt_size = T;
    return HX_OK;
}

int hx_engine::db_fill_record(uint32_t *d_rec, uint32_t slot, const uint32_t *h_src)
{
    const uint32_t m = mirror.m, words = HX_FUSED_MAXL + 2u * HX_FUSED_MAXL * 2u * m;
    HX_HIP(this, hipSetDevice(device));
    int rc = bw_reserve(this, 4096, (size_t)words * 4); if (rc) return rc;
    uint32_t *d_tmp = nullptr;
    HX_HIP(this, hipMalloc((void **)&d_tmp, (size_t)words * 4));
    memcpy(bw.h, h_src, (size_t)words * 4);
    hipError_t s = hipMemcpyAsync(d_tmp, bw.h, (size_t)words * 4, hipMemcpyHostToDevice, stream);
    if (s == hipSuccess) { hipLaunchKernelGGL(k_fill_rec, dim3(1), dim3(256), 0, stream, d_rec, hx_rec_words(m), slot, m, (const uint32_t *)d_tmp); s = hipGetLastError(); }
    if (s == hipSuccess) s = hipStreamSynchronize(stream);
    (void)hipFree(d_tmp);
    if (s != hipSuccess) return fail(HX_E_HIP, std::string("db_fill_record: ") + hipGetErrorString(s));
    return HX_OK;
}

// Confirmed duplicate candidates of the batch rows [base, base + b): (za, zb) = (member, zero-distance layer-0 neighbour) pairs, grouped per
// member in list order (d_rec == nullptr: skipped, the caller has the lists on the host); (ha, hb) = (later member, earlier member of the same
// batch with identical bytes), ascending in ha.  All ids are element ids.
int hx_engine::db_dup_candidates(uint32_t base, uint32_t b, const uint32_t *d_rec, std::vector<uint32_t> &za, std::vector<uint32_t> &zb,
                                 std::vector<uint32_t> &ha, std::vector<uint32_t> &hb)
{
    za.clear(); zb.clear(); ha.clear(); hb.clear();
    if (b == 0) return HX_OK;
    HX_HIP(this, hipSetDevice(device));
    const uint32_t m = mirror.m, lm0 = 2u * (m ? m : 1u), rw = hx_rec_words(m ? m : 1u);
    size_t tmp_sort = 0;
    HX_HIP(this, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                                    (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)b, 0, 64, stream));
    const size_t zcap = d_rec ? (size_t)b * lm0 : 0, pcap = zcap + b;
    size_t o = 0;
    const size_t o_h = o; o += al256((size_t)b * 8); const size_t o_hs = o; o += al256((size_t)b * 8);
    const size_t o_i = o; o += al256((size_t)b * 4); const size_t o_is = o; o += al256((size_t)b * 4);
    const size_t o_pa = o; o += al256(pcap * 4); const size_t o_pb = o; o += al256(pcap * 4); const size_t o_eq = o; o += al256(pcap);
    const size_t o_ctr = o; o += 256; const size_t o_tmp = o; o += al256(tmp_sort);
    int rc = bw_reserve(this, o, pcap * 9 + 256); if (rc) return rc;
    uint8_t *d = bw.d;
    uint32_t *ctr = (uint32_t *)(d + o_ctr), *pa = (uint32_t *)(d + o_pa), *pb = (uint32_t *)(d + o_pb);
    HX_HIP(this, hipMemsetAsync(ctr, 0, 256, stream));
    if (d_rec) hipLaunchKernelGGL(k_zero_prefix, dim3((b + 255) / 256), dim3(256), 0, stream, d_rec, rw, m, base, b, pa, pb, ctr);
    if (b > 1) {
        hipLaunchKernelGGL(k_row_hash, dim3((b + 3) / 4), dim3(256), 0, stream, d_rows, (uint32_t)pitch, base, b, (unsigned long long *)(d + o_h), (uint32_t *)(d + o_i));
        size_t ts = tmp_sort;
        HX_HIP(this, hipcub::DeviceRadixSort::SortPairs(d + o_tmp, ts, (const unsigned long long *)(d + o_h), (unsigned long long *)(d + o_hs),
                                                        (const uint32_t *)(d + o_i), (uint32_t *)(d + o_is), (int)b, 0, 64, stream));
        hipLaunchKernelGGL(k_adjacent, dim3((b + 255) / 256), dim3(256), 0, stream, (const unsigned long long *)(d + o_hs), (const uint32_t *)(d + o_is), b, base,
                           pa + zcap, pb + zcap, ctr + 1);
    }
    HX_HIP(this, hipGetLastError());
    HX_HIP(this, hipMemcpyAsync(bw.h_ctr, ctr, 16, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipStreamSynchronize(stream));
    const uint32_t nz = bw.h_ctr[0], nh = bw.h_ctr[1];
    if (nz == 0 && nh == 0) return HX_OK;
    uint8_t *eq = d + o_eq;
    if (nz) hipLaunchKernelGGL(k_rows_equal_b, dim3((nz + 3) / 4), dim3(256), 0, stream, d_rows, (uint32_t)pitch, nz, pa, pb, eq);
    if (nh) hipLaunchKernelGGL(k_rows_equal_b, dim3((nh + 3) / 4), dim3(256), 0, stream, d_rows, (uint32_t)pitch, nh, pa + zcap, pb + zcap, eq + zcap);
    HX_HIP(this, hipGetLastError());
    uint32_t *h_pa = (uint32_t *)bw.h, *h_pb = h_pa + (nz + nh); uint8_t *h_eq = (uint8_t *)(h_pb + (nz + nh));
    if (nz) {
        HX_HIP(this, hipMemcpyAsync(h_pa, pa, (size_t)nz * 4, hipMemcpyDeviceToHost, stream));
        HX_HIP(this, hipMemcpyAsync(h_pb, pb, (size_t)nz * 4, hipMemcpyDeviceToHost, stream));
        HX_HIP(this, hipMemcpyAsync(h_eq, eq, nz, hipMemcpyDeviceToHost, stream));
    }
    if (nh) {
        HX_HIP(this, hipMemcpyAsync(h_pa + nz, pa + zcap, (size_t)nh * 4, hipMemcpyDeviceToHost, stream));
        HX_HIP(this, hipMemcpyAsync(h_pb + nz, pb + zcap, (size_t)nh * 4, hipMemcpyDeviceToHost, stream));
        HX_HIP(this, hipMemcpyAsync(h_eq + nz, eq + zcap, nh, hipMemcpyDeviceToHost, stream));
    }
    HX_HIP(this, hipStreamSynchronize(stream));
    // zero-distance pairs: ranges arrive in arbitrary member order; a stable sort by member keeps the list order inside a member
    std::vector<uint32_t> ord;
    for (uint32_t k = 0; k < nz; k++) if (h_eq[k]) ord.push_back(k);
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return h_pa[x] < h_pa[y]; });
    for (uint32_t k : ord) { za.push_back(h_pa[k]); zb.push_back(h_pb[k]); }
    ord.clear();
    for (uint32_t k = nz; k < nz + nh; k++) if (h_eq[k]) ord.push_back(k);
    std::sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return h_pa[x] < h_pa[y]; });
    for (uint32_t k : ord) { ha.push_back(h_pa[k]); hb.push_back(h_pb[k]); }
    return HX_OK;
}

// Members' lists into the mirror, then this rank's back-link ops into the grouping arrays; *n_ops_out ops are ready for
// links_run_grouped(..., on_device = true).  h_dup (nullable): one byte per member, non-zero = merged as a duplicate.
int hx_engine::db_apply(uint32_t base, uint32_t b, const uint32_t *d_rec, const uint8_t *h_dup, uint32_t rank, uint32_t world, uint32_t *n_ops_out)
{
    *n_ops_out = 0;
    if (b == 0) return HX_OK;
    HX_HIP(this, hipSetDevice(device));
    const uint32_t m = mirror.m, rw = hx_rec_words(m);
    size_t tmp_scan = 0;
    HX_HIP(this, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_scan, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)b, stream));
    size_t o = 0;
    const size_t o_dup = o; o += al256(b); const size_t o_n = o; o += al256((size_t)b * 4); const size_t o_s = o; o += al256((size_t)b * 4);
    const size_t o_ctr = o; o += 256; const size_t o_tmp = o; o += al256(tmp_scan);
    int rc = bw_reserve(this, o, (size_t)b + 256); if (rc) return rc;
    uint8_t *d = bw.d;
    const uint8_t *d_dup = nullptr;
    if (h_dup) { memcpy(bw.h, h_dup, b); HX_HIP(this, hipMemcpyAsync(d + o_dup, bw.h, b, hipMemcpyHostToDevice, stream)); d_dup = d + o_dup; }
    uint32_t *nops = (uint32_t *)(d + o_n), *opstart = (uint32_t *)(d + o_s), *ctr = (uint32_t *)(d + o_ctr);
    const MirrorPtrs mp = mirror_ptrs(this);
    hipLaunchKernelGGL(k_apply_new, dim3((b + 3) / 4), dim3(256), 0, stream, d_rec, rw, base, b, d_dup, mp, rank, world, nops);
    size_t ts = tmp_scan;
    HX_HIP(this, hipcub::DeviceScan::ExclusiveSum(d + o_tmp, ts, (const uint32_t *)nops, opstart, (int)b, stream));
    hipLaunchKernelGGL(k_total, dim3(1), dim3(64), 0, stream, (const uint32_t *)nops, (const uint32_t *)opstart, b, ctr);
    HX_HIP(this, hipGetLastError());
    HX_HIP(this, hipMemcpyAsync(bw.h_ctr, ctr, 4, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipStreamSynchronize(stream));
    const uint32_t n_ops = bw.h_ctr[0];
    *n_ops_out = n_ops;
    if (n_ops == 0) return HX_OK;
    if ((rc = hx_group_reserve(this, n_ops, grp))) return rc;
    hipLaunchKernelGGL(k_emit_ops, dim3((b + 3) / 4), dim3(256), 0, stream, d_rec, rw, base, b, (const int32_t *)mirror.d_level, m, (const uint32_t *)opstart, rank, world,
                       grp.d_keys, grp.d_new, grp.d_d);
    HX_HIP(this, hipGetLastError());
    return HX_OK;
}

int hx_engine::db_import_lists(const uint32_t *d_xrec, uint32_t n_records)
{
    if (n_records == 0) return HX_OK;
    HX_HIP(this, hipSetDevice(device));
    hipLaunchKernelGGL(k_import_recs, dim3((n_records + 3) / 4), dim3(256), 0, stream, d_xrec, hx_xrec_words(mirror.m), n_records, mirror_ptrs(this));
    HX_HIP(this, hipGetLastError());
    return HX_OK;
}

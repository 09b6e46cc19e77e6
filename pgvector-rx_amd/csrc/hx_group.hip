// hx_group.hip -- groups a batch's back-link ops per (target, layer) list ON THE DEVICE (update_neighbor_connections order is kept
// inside a group: the sort is a stable radix sort of the ops' keys).  Replaces the host-side bucket sort of hx_index_batch_links
// for single-process builds, where that stage was 9.5 s of a 33 s build of 20M x bit(1024) (short rows: the kernels are fast, the
// host is not).  Kept in its own translation unit because of hipcub's compile time.
#include "hx_internal.h"

#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace {
__global__ void k_iota(uint32_t *v, uint32_t n) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] = i; }

// sorted position i: gather the op's payload, flag the first op of every (target, layer) run
__global__ void k_flags(const unsigned long long *keys, const uint32_t *idx, const uint32_t *new_in, const float *d_in,
                        uint32_t *new_s, float *d_s, uint32_t *flag, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = idx[i];
    new_s[i] = new_in[src]; d_s[i] = d_in[src];
    flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// group g = exclusive-scan(flag): its list, its first op; counters[0] = number of groups
__global__ void k_groups(const unsigned long long *keys, const uint32_t *flag, const uint32_t *gid, uint32_t *tg, uint32_t *ly, uint32_t *off,
                         uint32_t n, uint32_t *counters)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) { const uint32_t g = gid[i]; tg[g] = (uint32_t)(keys[i] >> 7); ly[g] = (uint32_t)(keys[i] & 127ull); off[g] = i; }
    if (i == n - 1) { const uint32_t ng = gid[i] + flag[i]; counters[0] = ng; off[ng] = n; }
}

// launch order: lists with a long chain of ops (>= hub_min) for k_links_hub, the others for k_links_cached; counters[1] = hubs, [2] = others, [3] = longest chain
__global__ void k_split(const uint32_t *off, const uint32_t *counters_in, uint32_t hub_min, uint32_t *gmap_hub, uint32_t *gmap_norm, uint32_t *counters,
                        const uint32_t *tg, const uint32_t *ly, const uint16_t *l0_cnt, const uint8_t *pm_valid, uint32_t lm0, uint32_t *gmap_fill)
{   // one atomic per wave and class (every thread hitting the same three counters serialises: measured 1.2 ms per batch)
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    const bool in = g < counters_in[0];
    const uint32_t c = in ? off[g + 1] - off[g] : 0u;
    const bool hub = in && hub_min && c >= hub_min, norm = in && !hub;
    const unsigned long long mh = __ballot(hub), mn = __ballot(norm), below = (1ull << lane) - 1ull;
    uint32_t bh = 0, bn = 0;
    if (lane == 0) { if (mh) bh = atomicAdd(&counters[1], (uint32_t)__popcll(mh)); if (mn) bn = atomicAdd(&counters[2], (uint32_t)__popcll(mn)); }
    bh = __shfl(bh, 0, 64); bn = __shfl(bn, 0, 64);
    if (hub) gmap_hub[bh + (uint32_t)__popcll(mh & below)] = g;
    if (norm) gmap_norm[bn + (uint32_t)__popcll(mn & below)] = g;
    // full layer-0 lists whose resident pair matrix is missing: k_pm_fill computes it before the back-link kernels run (counters[4])
    if (pm_valid) {
        bool fill = false;
        if (in && ly[g] == 0u) { const uint32_t t = tg[g]; fill = l0_cnt[t] == lm0 && pm_valid[t] < lm0; }
        const unsigned long long mf = __ballot(fill);
        uint32_t bf = 0;
        if (lane == 0 && mf) bf = atomicAdd(&counters[4], (uint32_t)__popcll(mf));
        bf = __shfl(bf, 0, 64);
        if (fill) gmap_fill[bf + (uint32_t)__popcll(mf & below)] = g;
    }
    uint32_t mx = c;
    for (int o = 32; o >= 1; o >>= 1) { const uint32_t other = (uint32_t)__shfl_xor((int)mx, o, 64); mx = other > mx ? other : mx; }
    if (lane == 0 && mx) atomicMax(&counters[3], mx);
}
}  // namespace

// pinned host staging for n_ops ops: the caller writes keys / new ids / distances (op order) straight into it
int hx_group_stage(hx_engine *e, uint32_t n_ops, HxGroupWork &w, unsigned long long **keys, uint32_t **op_new, float **op_d)
{
    const size_t need = (size_t)n_ops * 16 + 64;
    if (need > w.cap_h) {
        HX_HIP(e, hipSetDevice(e->device));
        if (w.h) (void)hipHostFree(w.h);
        w.h = nullptr; w.cap_h = 0;
        HX_HIP(e, hipHostMalloc((void **)&w.h, need * 2, hipHostMallocDefault));
        w.cap_h = need * 2;
    }
    *keys = (unsigned long long *)w.h;
    *op_new = (uint32_t *)(w.h + (size_t)n_ops * 8);
    *op_d = (float *)(w.h + (size_t)n_ops * 12);
    return HX_OK;
}

// Device arrays of HxGroupWork are (re)allocated here for n_ops ops; afterwards w.d_keys / w.d_new / w.d_d are where the ops
// (op order) must be put -- uploaded by hx_group_ops, or written in place by the batch pipeline's emission kernel (hx_batch.hip).
namespace {
struct GroupLayout { size_t kin, kout, iin, iout, nin, din, ns, ds, flag, gid, tg, ly, off, gh, gn, gf, ctr, tmp, total, tmp_bytes; };
int group_layout(hx_engine *e, uint32_t n_ops, GroupLayout &L)
{
    hipStream_t st = e->stream;
    size_t tmp_sort = 0, tmp_scan = 0;
    HX_HIP(e, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                                 (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n_ops, 0, 39, st));
    HX_HIP(e, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_scan, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n_ops, st));
    L.tmp_bytes = std::max(tmp_sort, tmp_scan);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t n = n_ops;
    size_t o = 0;
    L.kin = o; o += al(n * 8); L.kout = o; o += al(n * 8);
    L.iin = o; o += al(n * 4); L.iout = o; o += al(n * 4);
    L.nin = o; o += al(n * 4); L.din = o; o += al(n * 4);
    L.ns = o; o += al(n * 4); L.ds = o; o += al(n * 4);
    L.flag = o; o += al(n * 4); L.gid = o; o += al(n * 4);
    L.tg = o; o += al(n * 4); L.ly = o; o += al(n * 4); L.off = o; o += al((n + 1) * 4);
    L.gh = o; o += al(n * 4); L.gn = o; o += al(n * 4); L.gf = o; o += al(n * 4);
    L.ctr = o; o += 256;
    L.tmp = o; o += al(L.tmp_bytes);
    L.total = o;
    return HX_OK;
}
}  // namespace

int hx_group_reserve(hx_engine *e, uint32_t n_ops, HxGroupWork &w)
{
    HX_HIP(e, hipSetDevice(e->device));
    GroupLayout L; int rc = group_layout(e, n_ops ? n_ops : 1, L); if (rc) return rc;
    if (L.total > w.cap) {
        HX_HIP(e, hipStreamSynchronize(e->stream));
        if (w.d) (void)hipFree(w.d);
        w.d = nullptr; w.cap = 0;
        HX_HIP(e, hipMalloc((void **)&w.d, L.total * 2));
        w.cap = L.total * 2;
    }
    if (!w.h_ctr) HX_HIP(e, hipHostMalloc((void **)&w.h_ctr, 64, hipHostMallocDefault));
    w.d_keys = (unsigned long long *)(w.d + L.kin); w.d_new = (uint32_t *)(w.d + L.nin); w.d_d = (float *)(w.d + L.din);
    return HX_OK;
}

// groups the n_ops ops already present in w.d_keys / w.d_new / w.d_d (hx_group_reserve(n_ops) placed those arrays)
int hx_group_run(hx_engine *e, uint32_t n_ops, uint32_t hub_min, HxGroupWork &w, uint32_t counters_out[5], bool want_fill)
{
    for (int i = 0; i < 5; i++) counters_out[i] = 0;
    if (n_ops == 0) return HX_OK;
    HX_HIP(e, hipSetDevice(e->device));
    hipStream_t st = e->stream;
    GroupLayout L; int rc = group_layout(e, n_ops, L); if (rc) return rc;
    if (L.total > w.cap || !w.d) return e->fail(HX_E_STATE, "hx_group_run without hx_group_reserve");
    uint8_t *b = w.d;
    HX_HIP(e, hipMemsetAsync(b + L.ctr, 0, 256, st));
    const uint32_t tb = 256, gb = (n_ops + tb - 1) / tb;
    hipLaunchKernelGGL(k_iota, dim3(gb), dim3(tb), 0, st, (uint32_t *)(b + L.iin), n_ops);
    size_t tb_sort = L.tmp_bytes;
    HX_HIP(e, hipcub::DeviceRadixSort::SortPairs(b + L.tmp, tb_sort, (const unsigned long long *)(b + L.kin), (unsigned long long *)(b + L.kout),
                                                 (const uint32_t *)(b + L.iin), (uint32_t *)(b + L.iout), (int)n_ops, 0, 39, st));
    hipLaunchKernelGGL(k_flags, dim3(gb), dim3(tb), 0, st, (const unsigned long long *)(b + L.kout), (const uint32_t *)(b + L.iout),
                       (const uint32_t *)(b + L.nin), (const float *)(b + L.din), (uint32_t *)(b + L.ns), (float *)(b + L.ds), (uint32_t *)(b + L.flag), n_ops);
    size_t tb_scan = L.tmp_bytes;
    HX_HIP(e, hipcub::DeviceScan::ExclusiveSum(b + L.tmp, tb_scan, (const uint32_t *)(b + L.flag), (uint32_t *)(b + L.gid), (int)n_ops, st));
    hipLaunchKernelGGL(k_groups, dim3(gb), dim3(tb), 0, st, (const unsigned long long *)(b + L.kout), (const uint32_t *)(b + L.flag), (const uint32_t *)(b + L.gid),
                       (uint32_t *)(b + L.tg), (uint32_t *)(b + L.ly), (uint32_t *)(b + L.off), n_ops, (uint32_t *)(b + L.ctr));
    hipLaunchKernelGGL(k_split, dim3(gb), dim3(tb), 0, st, (const uint32_t *)(b + L.off), (const uint32_t *)(b + L.ctr), hub_min,
                       (uint32_t *)(b + L.gh), (uint32_t *)(b + L.gn), (uint32_t *)(b + L.ctr),
                       (const uint32_t *)(b + L.tg), (const uint32_t *)(b + L.ly), (const uint16_t *)e->mirror.d_l0_cnt, want_fill ? (const uint8_t *)e->mirror.d_pm_valid : nullptr, 2u * e->mirror.m,
                       (uint32_t *)(b + L.gf));
    HX_HIP(e, hipGetLastError());
    HX_HIP(e, hipMemcpyAsync(w.h_ctr, b + L.ctr, 20, hipMemcpyDeviceToHost, st));
    HX_HIP(e, hipStreamSynchronize(st));
    for (int i = 0; i < 5; i++) counters_out[i] = w.h_ctr[i];
    w.gmap_fill = (const uint32_t *)(b + L.gf);
    w.tg = (const uint32_t *)(b + L.tg); w.ly = (const uint32_t *)(b + L.ly); w.off = (const uint32_t *)(b + L.off);
    w.op_new = (const uint32_t *)(b + L.ns); w.op_d = (const float *)(b + L.ds);
    w.gmap_hub = (const uint32_t *)(b + L.gh); w.gmap_norm = (const uint32_t *)(b + L.gn);
    return HX_OK;
}

// keys / new ids / distances arrive in host memory (op order)
int hx_group_ops(hx_engine *e, uint32_t n_ops, const unsigned long long *h_keys, const uint32_t *h_new, const float *h_d, uint32_t hub_min,
                 HxGroupWork &w, uint32_t counters_out[5], bool want_fill)
{
    for (int i = 0; i < 5; i++) counters_out[i] = 0;
    if (n_ops == 0) return HX_OK;
    int rc = hx_group_reserve(e, n_ops, w); if (rc) return rc;
    HX_HIP(e, hipMemcpyAsync(w.d_keys, h_keys, (size_t)n_ops * 8, hipMemcpyHostToDevice, e->stream));
    HX_HIP(e, hipMemcpyAsync(w.d_new, h_new, (size_t)n_ops * 4, hipMemcpyHostToDevice, e->stream));
    HX_HIP(e, hipMemcpyAsync(w.d_d, h_d, (size_t)n_ops * 4, hipMemcpyHostToDevice, e->stream));
    return hx_group_run(e, n_ops, hub_min, w, counters_out, want_fill);
}

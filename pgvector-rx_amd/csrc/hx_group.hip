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
__global__ void k_split(const uint32_t *off, const uint32_t *counters_in, uint32_t hub_min, uint32_t *gmap_hub, uint32_t *gmap_norm, uint32_t *counters)
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

// Device arrays of HxGroupWork are (re)allocated here for n_ops ops.  keys/new/d arrive in host memory in op order.
int hx_group_ops(hx_engine *e, uint32_t n_ops, const unsigned long long *h_keys, const uint32_t *h_new, const float *h_d, uint32_t hub_min,
                 HxGroupWork &w, uint32_t counters_out[4])
{
    if (n_ops == 0) { counters_out[0] = counters_out[1] = counters_out[2] = counters_out[3] = 0; return HX_OK; }
    HX_HIP(e, hipSetDevice(e->device));
    hipStream_t st = e->stream;
    size_t tmp_sort = 0, tmp_scan = 0;
    HX_HIP(e, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                                 (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n_ops, 0, 39, st));
    HX_HIP(e, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_scan, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n_ops, st));
    const size_t tmp_bytes = std::max(tmp_sort, tmp_scan);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t n = n_ops;
    size_t o = 0;
    const size_t o_kin = o; o += al(n * 8); const size_t o_kout = o; o += al(n * 8);
    const size_t o_iin = o; o += al(n * 4); const size_t o_iout = o; o += al(n * 4);
    const size_t o_nin = o; o += al(n * 4); const size_t o_din = o; o += al(n * 4);
    const size_t o_ns = o; o += al(n * 4); const size_t o_ds = o; o += al(n * 4);
    const size_t o_flag = o; o += al(n * 4); const size_t o_gid = o; o += al(n * 4);
    const size_t o_tg = o; o += al(n * 4); const size_t o_ly = o; o += al(n * 4); const size_t o_off = o; o += al((n + 1) * 4);
    const size_t o_gh = o; o += al(n * 4); const size_t o_gn = o; o += al(n * 4);
    const size_t o_ctr = o; o += 256;
    const size_t o_tmp = o; o += al(tmp_bytes);
    if (o > w.cap) {
        if (w.d) (void)hipFree(w.d);
        w.d = nullptr; w.cap = 0;
        HX_HIP(e, hipMalloc((void **)&w.d, o * 2));
        w.cap = o * 2;
    }
    if (!w.h_ctr) HX_HIP(e, hipHostMalloc((void **)&w.h_ctr, 64, hipHostMallocDefault));
    uint8_t *b = w.d;
    static const bool dbg = getenv("HX_DEBUG") != nullptr; static double acc[4] = {0, 0, 0, 0}; static int calls = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = dbg ? now() : 0.0;
    HX_HIP(e, hipMemcpyAsync(b + o_kin, h_keys, n * 8, hipMemcpyHostToDevice, st));
    HX_HIP(e, hipMemcpyAsync(b + o_nin, h_new, n * 4, hipMemcpyHostToDevice, st));
    HX_HIP(e, hipMemcpyAsync(b + o_din, h_d, n * 4, hipMemcpyHostToDevice, st));
    HX_HIP(e, hipMemsetAsync(b + o_ctr, 0, 256, st));
    const uint32_t tb = 256, gb = (n_ops + tb - 1) / tb;
    if (dbg) { (void)hipStreamSynchronize(st); const double t = now(); acc[0] += t - t0; t0 = t; }
    hipLaunchKernelGGL(k_iota, dim3(gb), dim3(tb), 0, st, (uint32_t *)(b + o_iin), n_ops);
    size_t tb_sort = tmp_bytes;
    HX_HIP(e, hipcub::DeviceRadixSort::SortPairs(b + o_tmp, tb_sort, (const unsigned long long *)(b + o_kin), (unsigned long long *)(b + o_kout),
                                                 (const uint32_t *)(b + o_iin), (uint32_t *)(b + o_iout), (int)n_ops, 0, 39, st));
    if (dbg) { (void)hipStreamSynchronize(st); const double t = now(); acc[1] += t - t0; t0 = t; }
    hipLaunchKernelGGL(k_flags, dim3(gb), dim3(tb), 0, st, (const unsigned long long *)(b + o_kout), (const uint32_t *)(b + o_iout),
                       (const uint32_t *)(b + o_nin), (const float *)(b + o_din), (uint32_t *)(b + o_ns), (float *)(b + o_ds), (uint32_t *)(b + o_flag), n_ops);
    size_t tb_scan = tmp_bytes;
    HX_HIP(e, hipcub::DeviceScan::ExclusiveSum(b + o_tmp, tb_scan, (const uint32_t *)(b + o_flag), (uint32_t *)(b + o_gid), (int)n_ops, st));
    hipLaunchKernelGGL(k_groups, dim3(gb), dim3(tb), 0, st, (const unsigned long long *)(b + o_kout), (const uint32_t *)(b + o_flag), (const uint32_t *)(b + o_gid),
                       (uint32_t *)(b + o_tg), (uint32_t *)(b + o_ly), (uint32_t *)(b + o_off), n_ops, (uint32_t *)(b + o_ctr));
    hipLaunchKernelGGL(k_split, dim3(gb), dim3(tb), 0, st, (const uint32_t *)(b + o_off), (const uint32_t *)(b + o_ctr), hub_min,
                       (uint32_t *)(b + o_gh), (uint32_t *)(b + o_gn), (uint32_t *)(b + o_ctr));
    HX_HIP(e, hipGetLastError());
    HX_HIP(e, hipMemcpyAsync(w.h_ctr, b + o_ctr, 16, hipMemcpyDeviceToHost, st));
    HX_HIP(e, hipStreamSynchronize(st));
    if (dbg) { const double t = now(); acc[2] += t - t0; if (++calls % 50 == 0) fprintf(stderr, "[hx] hx_group_ops x%d: upload %.1f ms, iota+sort %.1f ms, flags+scan+groups+split+readback %.1f ms (n_ops %u)\n", calls, acc[0] * 1e3, acc[1] * 1e3, acc[2] * 1e3, n_ops); }
    for (int i = 0; i < 4; i++) counters_out[i] = w.h_ctr[i];
    w.tg = (const uint32_t *)(b + o_tg); w.ly = (const uint32_t *)(b + o_ly); w.off = (const uint32_t *)(b + o_off);
    w.op_new = (const uint32_t *)(b + o_ns); w.op_d = (const float *)(b + o_ds);
    w.gmap_hub = (const uint32_t *)(b + o_gh); w.gmap_norm = (const uint32_t *)(b + o_gn);
    return HX_OK;
}

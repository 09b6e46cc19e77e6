// hx_internal.h -- shared between the HIP engine (hx_engine.hip) and the host graph driver (hx_index.cpp).
// Not part of the ABI; the ABI is include/hnswrx.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/hnswrx.h"

// One request channel = pinned host staging + device mirrors for the id/offset/result arrays of a launch.
// The lock-step driver writes its requests straight into the pinned arrays (no extra copy).
struct HxChannel {
    // distance groups (hx_distances_batch shape)
    uint32_t *h_grp_q = nullptr, *h_grp_off = nullptr, *h_ids = nullptr;
    float *h_out = nullptr;
    uint32_t *d_grp_q = nullptr, *d_grp_off = nullptr, *d_ids = nullptr;
    float *d_out = nullptr;
    size_t cap_groups = 0, cap_ids = 0;
    // pair groups (hx_pairwise_many shape)
    uint32_t *h_pg_off = nullptr, *h_pids = nullptr, *h_wg_tab = nullptr;   // wg_tab: {group, first pair} per workgroup
    uint16_t *h_pg_na = nullptr, *h_pg_nb = nullptr;
    uint64_t *h_pg_out_off = nullptr;
    float *h_pout = nullptr;
    uint32_t *d_pg_off = nullptr, *d_pids = nullptr, *d_wg_tab = nullptr;
    uint16_t *d_pg_na = nullptr, *d_pg_nb = nullptr;
    uint64_t *d_pg_out_off = nullptr;
    float *d_pout = nullptr;
    size_t cap_pgroups = 0, cap_pids = 0, cap_pout = 0, cap_wg = 0;
};

struct HxKernelStat { uint64_t launches = 0, units = 0; double ms = 0.0; };

struct hx_engine {
    int device = 0, dtype = 0, metric = 0, dim = 0;
    uint64_t row_bytes = 0, pitch = 0, capacity = 0, n_rows = 0;
    uint8_t *d_rows = nullptr;
    uint8_t *d_queries = nullptr; uint32_t cap_queries = 0, n_queries = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing = false; float last_ms = 0.f;
    HxKernelStat stat_dist, stat_pair;
    HxChannel ch;
    std::string err;

    // internal (driver-facing) entry points; arrays live in ch.h_* and results land in ch.h_out / ch.h_pout
    int ensure_dist_capacity(size_t groups, size_t ids);
    int ensure_pair_capacity(size_t groups, size_t ids, size_t outs);
    int run_dist(uint32_t n_groups, uint32_t n_ids);                 // blocking: H2D, kernel, D2H, sync
    int run_pair(uint32_t n_groups, uint32_t n_ids, uint64_t n_out); // blocking
    int fail(int code, const std::string &msg) { err = msg; return code; }
};

#define HX_HIP(e, call)                                                                             \
    do {                                                                                            \
        hipError_t _s = (call);                                                                     \
        if (_s != hipSuccess) return (e)->fail(HX_E_HIP, std::string(#call) + ": " + hipGetErrorString(_s)); \
    } while (0)

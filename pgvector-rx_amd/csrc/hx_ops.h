// hx_ops.h -- device helpers shared by the engine's translation units: 16-byte fragment type, LDS-free wave reductions,
// the per-(dtype, metric) distance operators in the canonical summation order (see hx_engine.hip), pair-slab constants and
// the dtype/metric dispatch macro.  Not part of the ABI.
#pragma once
#include "hx_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// =================================================================================================
// device helpers
// =================================================================================================
__device__ __forceinline__ float bits2f(unsigned int u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ float half2f(unsigned int h16)
{   // exact widening, subnormals included (half_to_f32, halfvec.rs:54-87)
    return (float)__builtin_bit_cast(_Float16, (unsigned short)h16);
}

// xor-butterfly exchange v[lane ^ OFF] without touching LDS: __shfl_xor compiles to ds_bpermute_b32 (an LDS-crossbar round
// trip, ~100+ cycles each, six in a row per reduced value -- measured as half of a row batch's time in the traversal
// kernel).  gfx950 has the half/row swaps as VALU ops, and the steps inside a 16-lane row are DPP moves:
//   32: v_permlane32_swap(v, v) leaves {lo,lo} and {hi,hi}; 16: v_permlane16_swap likewise on row pairs;
//   8 = row_half_mirror . row_mirror, 4 = quad_perm[3,2,1,0] . row_half_mirror, 2 / 1 = quad_perm.
// The sums are the butterfly's own (a + b is the same value in both partners), so the canonical order is unchanged.
template <int CTRL> __device__ __forceinline__ unsigned int dpp_mov(unsigned int v)
{
    return (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
template <int OFF> __device__ __forceinline__ void xor_pair(unsigned int v, unsigned int &a, unsigned int &b)
{   // a (+) b == v[lane] (+) v[lane ^ OFF] for a commutative (+)
    if constexpr (OFF == 32) { auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); a = r[0]; b = r[1]; }
    else if constexpr (OFF == 16) { auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); a = r[0]; b = r[1]; }
    else if constexpr (OFF == 8) { a = v; b = dpp_mov<0x141>(dpp_mov<0x140>(v)); }
    else if constexpr (OFF == 4) { a = v; b = dpp_mov<0x1B>(dpp_mov<0x141>(v)); }
    else if constexpr (OFF == 2) { a = v; b = dpp_mov<0x4E>(v); }
    else { a = v; b = dpp_mov<0xB1>(v); }
}
template <int LPR, int OFF> __device__ __forceinline__ float lanes_sum_f_step(float v)
{
    if constexpr (OFF < LPR) {
        unsigned int a, b; xor_pair<OFF>(__builtin_bit_cast(unsigned int, v), a, b);
        v = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    }
    if constexpr (OFF > 1) return lanes_sum_f_step<LPR, OFF / 2>(v); else return v;
}
template <int LPR, int OFF> __device__ __forceinline__ int lanes_sum_i_step(int v)
{
    if constexpr (OFF < LPR) { unsigned int a, b; xor_pair<OFF>((unsigned int)v, a, b); v = (int)a + (int)b; }
    if constexpr (OFF > 1) return lanes_sum_i_step<LPR, OFF / 2>(v); else return v;
}
template <int LPR> __device__ __forceinline__ float lanes_sum_f(float v) { return lanes_sum_f_step<LPR, 32>(v); }
// Four 64-lane sums at once, each by the very same butterfly tree: at the xor-32 step the two halves of a wave can work on different values
// (v_permlane32_swap(a, b) hands lanes 0-31 both halves of a and lanes 32-63 both halves of b), at the xor-16 step the four 16-lane rows likewise,
// and the steps inside a row then reduce all four together: 3 swaps + 7 adds + 4 row steps instead of 4 x (6 exchanges + 6 adds).
// Returns z with sum(a0) in lanes 0-15, sum(a2) in 16-31, sum(a1) in 32-47, sum(a3) in 48-63 (lanes_sum4_lane(r) = a lane that holds sum(a_r)).
__device__ __forceinline__ float lanes_sum4_f(float a0, float a1, float a2, float a3)
{
    auto f = [](unsigned int u) { return __builtin_bit_cast(float, u); };
    auto u = [](float x) { return __builtin_bit_cast(unsigned int, x); };
    const auto r01 = __builtin_amdgcn_permlane32_swap(u(a0), u(a1), false, false);      // {a0.lo | a1.lo}, {a0.hi | a1.hi}
    const auto r23 = __builtin_amdgcn_permlane32_swap(u(a2), u(a3), false, false);
    const float p = f(r01[0]) + f(r01[1]), q = f(r23[0]) + f(r23[1]);                    // lanes 0-31: a0 (a2) folded once, lanes 32-63: a1 (a3)
    const auto rr = __builtin_amdgcn_permlane16_swap(u(p), u(q), false, false);          // rows {p0 q0 p2 q2}, {p1 q1 p3 q3}
    const float z = f(rr[0]) + f(rr[1]);                                                 // rows: a0, a2, a1, a3 folded twice
    return lanes_sum_f_step<64, 8>(z);
}
__device__ __forceinline__ constexpr int lanes_sum4_lane(int r) { return r == 0 ? 0 : r == 1 ? 32 : r == 2 ? 16 : 48; }
template <int LPR> __device__ __forceinline__ int lanes_sum_i(int v) { return lanes_sum_i_step<LPR, 32>(v); }

// ---- per-(dtype, metric) operators: add() consumes one 16-byte fragment pair, finish() reduces ----
enum { K_L2 = 0, K_IP = 1, K_L1 = 2 };

template <int KIND> __device__ __forceinline__ void fterm(float &acc, float x, float y)
{
    if (KIND == K_L2) { float d = x - y; acc = acc + d * d; }
    else if (KIND == K_IP) { acc = acc + x * y; }
    else { acc = acc + __builtin_fabsf(x - y); }
}

template <int KIND> struct OpF32 {
    static constexpr bool kSparse = false;
    static constexpr int query_minw = 0;              // k_fused<query>: waves per SIMD the register budget is cut for (0: FUSED_MINW)
    static constexpr bool mfma_split_ok = false;
    static constexpr bool sorted_array_ok = true;    // real-valued distances: ties are rare enough for k_fused's sorted-array searches
    typedef float acc_t;
    static __device__ __forceinline__ void init(acc_t &a) { a = 0.0f; }
    static __device__ __forceinline__ void add(acc_t &acc, const u4 &a, const u4 &b)
    {
#pragma unroll
        for (int t = 0; t < 4; t++) fterm<KIND>(acc, bits2f(a[t]), bits2f(b[t]));
    }
    template <int LPR> static __device__ __forceinline__ float finish(acc_t acc)
    {
        float s = lanes_sum_f<LPR>(acc);
        return KIND == K_IP ? -s : s;      // vector_negative_inner_product, vector.rs:631
    }
    static constexpr bool kFloatAcc = true;
    static __device__ __forceinline__ float post(float s) { return KIND == K_IP ? -s : s; }
};

template <int KIND> struct OpF16 {
    static constexpr bool kSparse = false;
    static constexpr int query_minw = KIND == K_L1 ? 0 : 3;   // the f16 -> f32 widening of L2 / inner product needs 141 VGPRs: at 4 waves per SIMD (128) the query kernel spilled 25 of them (56 B scratch); 3 waves, no spills (C4 shape +3.5 % queries/s)
    static constexpr bool mfma_split_ok = KIND == K_IP;   // k_fused MODE 3: select_neighbors on the matrix cores (halfvec inner product is a true f16 GEMM)
    static constexpr bool sorted_array_ok = true;
    typedef float acc_t;
    static __device__ __forceinline__ void init(acc_t &a) { a = 0.0f; }
    static __device__ __forceinline__ void add(acc_t &acc, const u4 &a, const u4 &b)
    {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            fterm<KIND>(acc, half2f(a[t] & 0xffffu), half2f(b[t] & 0xffffu));
            fterm<KIND>(acc, half2f(a[t] >> 16), half2f(b[t] >> 16));
        }
    }
    template <int LPR> static __device__ __forceinline__ float finish(acc_t acc)
    {
        float s = lanes_sum_f<LPR>(acc);
        return KIND == K_IP ? -s : s;
    }
    static constexpr bool kFloatAcc = true;
    static __device__ __forceinline__ float post(float s) { return KIND == K_IP ? -s : s; }
};

struct OpHamming {   // bitvec.rs:97-106: popcount(a ^ b); integer, order-free
    static constexpr bool kSparse = false;
    static constexpr int query_minw = 0;
    static constexpr bool mfma_split_ok = false;
    static constexpr bool sorted_array_ok = false;   // integer-valued: ties everywhere, the heap kernels only
    typedef int acc_t;
    static __device__ __forceinline__ void init(acc_t &a) { a = 0; }
    static __device__ __forceinline__ void add(acc_t &acc, const u4 &a, const u4 &b)
    {
#pragma unroll
        for (int t = 0; t < 4; t++) acc += __popc(a[t] ^ b[t]);
    }
    template <int LPR> static __device__ __forceinline__ float finish(acc_t acc) { return (float)lanes_sum_i<LPR>(acc); }
    static constexpr bool kFloatAcc = false;
    static __device__ __forceinline__ float post(float s) { return s; }
};

struct JacAcc { int ab, aa, bb; };
struct OpJaccard {   // bitvec.rs:113-132
    static constexpr bool kSparse = false;
    static constexpr int query_minw = 0;
    static constexpr bool mfma_split_ok = false;
    static constexpr bool sorted_array_ok = false;
    typedef JacAcc acc_t;
    static __device__ __forceinline__ void init(acc_t &a) { a.ab = a.aa = a.bb = 0; }
    static __device__ __forceinline__ void add(acc_t &acc, const u4 &a, const u4 &b)
    {
#pragma unroll
        for (int t = 0; t < 4; t++) { acc.ab += __popc(a[t] & b[t]); acc.aa += __popc(a[t]); acc.bb += __popc(b[t]); }
    }
    template <int LPR> static __device__ __forceinline__ float finish(acc_t acc)
    {
        int ab = lanes_sum_i<LPR>(acc.ab), aa = lanes_sum_i<LPR>(acc.aa), bb = lanes_sum_i<LPR>(acc.bb);
        double d = ab == 0 ? 1.0 : 1.0 - ((double)ab / (double)(aa + bb - ab));
        return (float)d;                   // f64 result, `as f32` on the build path (build.rs:367)
    }
    static constexpr bool kFloatAcc = false;
    static __device__ __forceinline__ float post(float s) { return s; }
};

// 64 butterflies at once: a[s] is this lane's partial of pair s (s < 64).  Stage k exchanges half of the live values
// across lanes l <-> l^off and adds, halving the number of live registers; every add has exactly the two operands the
// plain xor butterfly of that pair has at that stage (p[l] + p[l^off], commutative), so the sums are the same bits.
// Returns, in lane l, the full sum of pair l.  63 cross-lane moves instead of 64*6.
template <int N> __device__ __forceinline__ void xstage(const float (&in)[2 * N], float (&out)[N], uint32_t lane, int off)
{
    const bool up = (lane & (uint32_t)off) != 0;
#pragma unroll
    for (int s = 0; s < N; s++) {
        const float lo = in[s], hi = in[s + N];
        const float send = up ? lo : hi, keep = up ? hi : lo;
        out[s] = keep + __shfl_xor(send, off, 64);
    }
}
__device__ __forceinline__ float reduce64_transposed(const float (&a)[64], uint32_t lane)
{
    float v32[32], v16[16], v8[8], v4[4], v2[2], v1[1];
    xstage<32>(a, v32, lane, 32); xstage<16>(v32, v16, lane, 16); xstage<8>(v16, v8, lane, 8);
    xstage<4>(v8, v4, lane, 4); xstage<2>(v4, v2, lane, 2); xstage<1>(v2, v1, lane, 1);
    return v1[0];
}
// results of a wave's HX_PAIRS_PER_WAVE accumulators: res0 = pair `lane` (< 64), res1 = pair 64+lane (lanes 0,1)
template <class OP, int NP>
__device__ __forceinline__ void reduce_pairs(typename OP::acc_t (&acc)[NP], uint32_t lane, float &res0, float &res1)
{
    res0 = 0.0f; res1 = 0.0f;
    if constexpr (OP::kFloatAcc && NP >= 64) {
        float a[64];
#pragma unroll
        for (int s = 0; s < 64; s++) a[s] = acc[s];
        res0 = OP::post(reduce64_transposed(a, lane));
#pragma unroll
        for (int s = 64; s < NP; s++) { const float d = OP::template finish<64>(acc[s]); if (lane == (uint32_t)(s - 64)) res1 = d; }
    } else {
#pragma unroll
        for (int s = 0; s < NP; s++) {
            const float d = OP::template finish<64>(acc[s]);
            if (s < 64) { if (lane == (uint32_t)s) res0 = d; } else { if (lane == (uint32_t)(s - 64)) res1 = d; }
        }
    }
}

#define HX_PAIR_WG 512
#define HX_PAIR_WAVES 8
#define HX_PAIRS_PER_WAVE 66
#define HX_PAIR_SLAB (HX_PAIR_WAVES * HX_PAIRS_PER_WAVE)   /* 528 = 33*32/2 */


// ---- sparsevec rows (hx_sparse.hip, and the traversal kernel's sparse distance path) ----------------------------------------------------
// Row record: { int32 nnz; int32 pad[3]; int32 index[cap]; float value[cap] }, cap = min(dim, 1000).  A distance is a merge join of two ascending
// index lists (src/types/sparsevec.rs:873-950, 1038-1088), its f32 terms added in merged-index order: ONE LANE walks one pair with the reference's own loop.
struct SpRow { const int32_t *idx; const float *val; int nnz; };
template <class P> __device__ __forceinline__ SpRow sp_row(P rec, uint32_t cap)
{
    SpRow r; r.nnz = *(const int32_t *)rec; r.idx = (const int32_t *)(rec + 16); r.val = (const float *)(rec + 16 + (size_t)cap * 4); return r;
}
// KIND: K_L2 sparse_l2_squared_distance (sparsevec.rs:873-918), K_IP sparse_inner_product (:921-950), K_L1 sparsevec_l1_distance (:1038-1088)
template <int KIND>
__device__ float sp_merge(const SpRow a, const SpRow b)
{
    float distance = 0.0f;
    int bpos = 0;
    for (int i = 0; i < a.nnz; i++) {
        const int32_t ai = a.idx[i];
        int32_t bi = -1;
        for (int j = bpos; j < b.nnz; j++) {
            bi = b.idx[j];
            if (ai == bi) {
                if (KIND == K_L2) { const float diff = a.val[i] - b.val[j]; distance += diff * diff; }
                else if (KIND == K_IP) distance += a.val[i] * b.val[j];
                else distance += __builtin_fabsf(a.val[i] - b.val[j]);
            } else if (ai > bi) {
                if (KIND == K_L2) distance += b.val[j] * b.val[j];
                else if (KIND == K_L1) distance += __builtin_fabsf(b.val[j]);
            }
            if (ai >= bi) bpos = j + 1;
            if (bi >= ai) break;
        }
        if (ai != bi) {
            if (KIND == K_L2) distance += a.val[i] * a.val[i];
            else if (KIND == K_L1) distance += __builtin_fabsf(a.val[i]);
        }
    }
    if (KIND != K_IP)
        for (int j = bpos; j < b.nnz; j++) {
            if (KIND == K_L2) distance += b.val[j] * b.val[j]; else distance += __builtin_fabsf(b.val[j]);
        }
    return KIND == K_IP ? -distance : distance;           // sparsevec_negative_inner_product, sparsevec.rs:993-1003
}
// the traversal kernel's view of the type: no chunk-wise accumulation -- f_dist_batch hands a row batch to f_dist_sparse (one lane per row)
template <int KIND> struct OpSparse {
    static constexpr bool kSparse = true, mfma_split_ok = false, sorted_array_ok = false, kFloatAcc = false;
    static constexpr int query_minw = 0;
    static constexpr int kind = KIND;
    typedef float acc_t;
    static __device__ __forceinline__ void init(acc_t &a) { a = 0.0f; }
    static __device__ __forceinline__ void add(acc_t &, const u4 &, const u4 &) {}
    template <int LPR> static __device__ __forceinline__ float finish(acc_t a) { return a; }
    static __device__ __forceinline__ float post(float s) { return s; }
};
__device__ __forceinline__ void tri_decode(uint32_t p, uint32_t &i, uint32_t &j)
{   // p = i*(i-1)/2 + j, j < i
    uint32_t ii = (uint32_t)((1.0f + __builtin_sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
    while (ii * (ii - 1) / 2 > p) ii--;
    while ((ii + 1) * ii / 2 <= p) ii++;
    i = ii; j = p - ii * (ii - 1) / 2;
}

#define HX_DISPATCH(e, CALL_F32, CALL_F16, CALL_HAM, CALL_JAC)                               \
    do {                                                                                      \
        if ((e)->dtype == HX_F32) {                                                           \
            if ((e)->metric == HX_L2SQ) { CALL_F32(K_L2); } else if ((e)->metric == HX_NEG_IP) { CALL_F32(K_IP); } else { CALL_F32(K_L1); } \
        } else if ((e)->dtype == HX_F16) {                                                    \
            if ((e)->metric == HX_L2SQ) { CALL_F16(K_L2); } else if ((e)->metric == HX_NEG_IP) { CALL_F16(K_IP); } else { CALL_F16(K_L1); } \
        } else {                                                                              \
            if ((e)->metric == HX_HAMMING) { CALL_HAM; } else { CALL_JAC; }                   \
        }                                                                                     \
    } while (0)

static inline size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }

// hx_engine.hip -- device row store + batched distance kernels for gfx950 (MI355X), and the engine half
// of the C ABI declared in include/hnswrx.h.
//
// Replaces, for the HNSW hot path of pgvector-rx, the per-pair scalar loops
//   compute_l2_squared / compute_inner_product / compute_l1_distance   src/types/vector.rs:516-567
//   the halfvec equivalents                                          src/types/halfvec.rs:685-733
//   compute_hamming_distance / compute_jaccard_distance              src/types/bitvec.rs:93-132
//   l2_normalize_raw                                                 src/types/vector.rs:106-126, halfvec.rs:204-233
// reached through graph::DistanceFn (src/graph/mod.rs:144-145) and load_element (src/index/scan.rs:155-228).
//
// Canonical summation order (every float kernel in this file, and oracle ORC_ORDER_W64):
//   a row is cut into 1 KiB chunks; lane l of a 64-lane wavefront owns the 16 bytes at chunk*1024 + 16*l
//   (4 f32 or 8 f16 elements) and adds their terms, in element order, chunk after chunk, into ONE f32
//   accumulator (mul and add rounded separately: compiled with -ffp-contract=off, like the reference's
//   unfused Rust).  The 64 lane partials are then combined by an xor butterfly 32,16,8,4,2,1.
//   Kernels that give a row fewer than 64 lanes (rows <= 512 B) produce the same bits: the missing
//   lanes' partials are +0.0 and x + 0.0 == x for every accumulator value reachable here.
//
// There is no CPU fallback anywhere in this file: without a GPU hx_create fails with HX_E_NODEVICE.
#include "hx_ops.h"

static thread_local std::string g_create_err;

// =================================================================================================
// K1: query-vs-rows, one workgroup per expansion group.
//   query parked in LDS once per group; each wavefront streams RIF row-slots at a time with 16-byte
//   coalesced loads (a row-slot is 64/LPR rows; LPR lanes cover one row); per-lane f32 partials;
//   xor-butterfly reduction; one f32 store per row.
//   HBM-bound: algorithmic bytes per distance = row payload (dim * elem size).
// =================================================================================================
template <class OP, int LPR, int WAVES, int RIF>
__global__ void __launch_bounds__(WAVES * 64)
k_dist_groups(const uint8_t *__restrict__ rows, const uint8_t *__restrict__ queries, uint32_t pitch,
              const uint32_t *__restrict__ grp_q, const uint32_t *__restrict__ grp_off,
              const uint32_t *__restrict__ ids, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t q_lds[];
    const uint32_t g = blockIdx.x;
    const uint32_t beg = grp_off[g], n = grp_off[g + 1] - beg;
    if (n == 0) return;
    const uint32_t qsel = grp_q[g];
    const uint8_t *qsrc = (qsel & HX_QUERY_SLOT) ? queries + (size_t)(qsel & 0x7fffffffu) * pitch
                                                 : rows + (size_t)qsel * pitch;
    for (uint32_t o = threadIdx.x * 16u; o < pitch; o += WAVES * 64u * 16u)
        *(u4 *)(q_lds + o) = *(const u4 *)(qsrc + o);
    __syncthreads();

    constexpr int RPS = 64 / LPR;                       // rows per wave step
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t sub = lane / LPR, sl = lane % LPR;
    const uint32_t nslots = (n + RPS - 1) / RPS;

    for (uint32_t s0 = wave * RIF; s0 < nslots; s0 += WAVES * RIF) {
        typename OP::acc_t acc[RIF];
        const uint8_t *rp[RIF];
        uint32_t ridx[RIF];
#pragma unroll
        for (int k = 0; k < RIF; k++) {
            uint32_t r = (s0 + k) * RPS + sub;
            ridx[k] = r;
            uint32_t id = ids[beg + (r < n ? r : 0u)];   // out-of-range slots re-read row 0 of the group; never stored
            rp[k] = rows + (size_t)id * pitch;
            OP::init(acc[k]);
        }
        for (uint32_t c = sl * 16u; c < pitch; c += LPR * 16u) {
            const u4 qv = *(const u4 *)(q_lds + c);
            u4 rv[RIF];
#pragma unroll
            for (int k = 0; k < RIF; k++) rv[k] = *(const u4 *)(rp[k] + c);
#pragma unroll
            for (int k = 0; k < RIF; k++) OP::add(acc[k], qv, rv[k]);
        }
#pragma unroll
        for (int k = 0; k < RIF; k++) {
            float d = OP::template finish<LPR>(acc[k]);
            if (sl == 0 && ridx[k] < n) out[beg + ridx[k]] = d;
        }
    }
}

// =================================================================================================
// K2: many small pair blocks (select_neighbors / back-link pruning operands).
//   One 512-thread workgroup per slab of <= HX_PAIR_SLAB pairs of one group (a 33-row back-link block is
//   exactly one slab).  The group's <= 64 rows move HBM -> registers -> LDS one 1 KiB chunk at a time,
//   double-buffered: while the waves compute on chunk c out of LDS buffer c&1, their global loads for
//   chunk c+1 are already in flight (one barrier per chunk).  Each wavefront owns HX_PAIRS_PER_WAVE
//   consecutive pairs, one f32 accumulator per pair in registers, walks (i,j) with scalar arithmetic and
//   reads both 16-byte fragments of a pair from LDS (contiguous, conflict-free).
//   Same canonical order as K1, so d(a,b) is the same bits whichever kernel produced it.
//   LDS/VALU-bound, not HBM-bound: each row is fetched once per slab and reused for up to 63 pairs.
// =================================================================================================
// STAGE = rows a wave stages per chunk (>= ceil(rows of the largest group / 8)); MINW = waves per SIMD asked of
// the register allocator (4 = two workgroups per CU).
template <class OP, int STAGE, int MINW>
__global__ void __launch_bounds__(HX_PAIR_WG, MINW)
k_pair_groups(const uint8_t *__restrict__ rows, uint32_t pitch,
              const uint32_t *__restrict__ pg_off, const uint16_t *__restrict__ pg_na,
              const uint16_t *__restrict__ pg_nb, const uint32_t *__restrict__ pids,
              const uint64_t *__restrict__ pg_out_off, const uint32_t *__restrict__ wg_tab,
              float *__restrict__ out, uint32_t lds_rows)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t g = wg_tab[2 * blockIdx.x], p0 = wg_tab[2 * blockIdx.x + 1];
    const uint32_t na = pg_na[g], nb = pg_nb[g], R = na + nb;
    const uint32_t *ids = pids + pg_off[g];
    const uint32_t P = nb ? na * nb : na * (na - 1) / 2;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t buf_bytes = lds_rows * 1024u;

    // first pair of this wave (wave-uniform)
    const uint32_t pw0 = p0 + wave * HX_PAIRS_PER_WAVE;
    uint32_t i0 = 0, j0 = 0;
    if (pw0 < P) { if (nb) { i0 = pw0 / nb; j0 = pw0 % nb; } else tri_decode(pw0, i0, j0); }
    i0 = __builtin_amdgcn_readfirstlane(i0); j0 = __builtin_amdgcn_readfirstlane(j0);
    const bool active = pw0 < P;

    // rows this wave stages: wave, wave+8, ...
    uint32_t rid[STAGE];
#pragma unroll
    for (int t = 0; t < STAGE; t++) { const uint32_t r = wave + t * HX_PAIR_WAVES; rid[t] = r < R ? ids[r] : 0u; }

    typename OP::acc_t acc[HX_PAIRS_PER_WAVE];
#pragma unroll
    for (int s = 0; s < HX_PAIRS_PER_WAVE; s++) OP::init(acc[s]);

    u4 pre[STAGE];
    auto prefetch = [&](uint32_t c0) {
        const uint32_t off = c0 + lane * 16u;
#pragma unroll
        for (int t = 0; t < STAGE; t++) {
            u4 v = {0u, 0u, 0u, 0u};
            if (wave + t * HX_PAIR_WAVES < R && off < pitch) v = *(const u4 *)(rows + (size_t)rid[t] * pitch + off);
            pre[t] = v;
        }
    };
    prefetch(0);
    uint32_t bufsel = 0;
    for (uint32_t c0 = 0; c0 < pitch; c0 += 1024u, bufsel ^= 1u) {
        uint8_t *buf = lds + bufsel * buf_bytes;
#pragma unroll
        for (int t = 0; t < STAGE; t++) {
            const uint32_t r = wave + t * HX_PAIR_WAVES;
            if (r < R) *(u4 *)(buf + r * 1024u + lane * 16u) = pre[t];
        }
        __syncthreads();
        if (c0 + 1024u < pitch) prefetch(c0 + 1024u);
        if (active) {
            uint32_t i = i0, j = j0;
#pragma unroll
            for (int s = 0; s < HX_PAIRS_PER_WAVE; s++) {
                const uint32_t ra = i < na ? i : na - 1u;            // pairs past P re-read a valid row; never stored
                const uint32_t rb = nb ? na + j : j;
                const u4 a = *(const u4 *)(buf + ra * 1024u + lane * 16u);
                const u4 b = *(const u4 *)(buf + rb * 1024u + lane * 16u);
                OP::add(acc[s], a, b);
                j++;
                if (j == (nb ? nb : i)) { i++; j = 0; }
            }
        }
    }
    if (!active) return;
    float res0, res1;
    reduce_pairs<OP, HX_PAIRS_PER_WAVE>(acc, lane, res0, res1);
    const uint64_t ob = pg_out_off[g];
    if (pw0 + lane < P) out[ob + pw0 + lane] = res0;
    if (lane < HX_PAIRS_PER_WAVE - 64 && pw0 + 64 + lane < P) out[ob + pw0 + 64 + lane] = res1;
}

// =================================================================================================
// normalisation (a4): f64 norm accumulated in index order by ONE thread per row, exactly as
// l2_normalize_raw (vector.rs:106-126): v*v is exact in f64 for f32-origin v, the sum is sequential.
// =================================================================================================
__device__ __forceinline__ unsigned short f32_to_half_ref(float f)
{   // f32_to_half, halfvec.rs:92-143, INCLUDING its flush of |x| < 2^-24 to zero
    unsigned int bits = __builtin_bit_cast(unsigned int, f);
    unsigned int sign = (bits >> 31) & 1u;
    int exp = (int)((bits >> 23) & 0xffu);
    unsigned int mant = bits & 0x7fffffu;
    if (exp == 0xff) {
        if (mant == 0) return (unsigned short)((sign << 15) | (0x1fu << 10));
        unsigned int m = mant >> 13; if (m < 1) m = 1;
        return (unsigned short)((sign << 15) | (0x1fu << 10) | m);
    }
    if (exp > 142) return (unsigned short)((sign << 15) | (0x1fu << 10));
    if (exp < 103) return (unsigned short)(sign << 15);
    if (exp < 113) {
        int shift = 113 - exp;
        unsigned int full = mant | 0x800000u;
        unsigned int m = full >> (shift + 13);
        unsigned int round_bit = (full >> (shift + 12)) & 1u;
        bool sticky = (full & ((1u << (shift + 12)) - 1u)) != 0;
        unsigned int r = (sign << 15) | m;
        if (round_bit && (sticky || (m & 1u))) r += 1;
        return (unsigned short)r;
    }
    unsigned int half_exp = (unsigned int)(exp - 127 + 15) & 0x1fu;
    unsigned int half_mant = mant >> 13;
    unsigned int round_bit = (mant >> 12) & 1u, sticky = mant & 0xfffu;
    unsigned int r = (sign << 15) | (half_exp << 10) | half_mant;
    if (round_bit && (sticky || (half_mant & 1u))) r += 1;
    return (unsigned short)r;
}

template <int DT>
__global__ void k_normalize(uint8_t *rows, uint32_t pitch, int dim, uint64_t n, double *norms)
{
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    uint8_t *row = rows + r * pitch;
    double norm = 0.0;
    for (int i = 0; i < dim; i++) {
        double v = DT == HX_F32 ? (double)((const float *)row)[i] : (double)half2f(((const unsigned short *)row)[i]);
        norm += v * v;
    }
    norm = sqrt(norm);
    if (norms) norms[r] = norm;
    if (norm > 0.0) {
        for (int i = 0; i < dim; i++) {
            if (DT == HX_F32) { float *x = (float *)row; x[i] = (float)((double)x[i] / norm); }
            else { unsigned short *x = (unsigned short *)row; x[i] = f32_to_half_ref((float)((double)half2f(x[i]) / norm)); }
        }
    }
}

__global__ void k_rows_equal(const uint8_t *__restrict__ rows, uint32_t pitch, uint32_t n_pairs,
                             const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint8_t *__restrict__ eq)
{
    uint32_t p = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (p >= n_pairs) return;
    const uint8_t *ra = rows + (size_t)a[p] * pitch, *rb = rows + (size_t)b[p] * pitch;
    bool same = true;
    for (uint32_t c = lane * 16u; c < pitch; c += 1024u) {
        u4 x = *(const u4 *)(ra + c), y = *(const u4 *)(rb + c);
        same = same && x[0] == y[0] && x[1] == y[1] && x[2] == y[2] && x[3] == y[3];
    }
    unsigned long long diff = __ballot(!same);
    if (lane == 0) eq[p] = diff == 0ull;
}

// =================================================================================================
// host side: dispatch
// =================================================================================================
static inline int lanes_per_row(uint64_t pitch)
{
    if (pitch > 512) return 64;
    if (pitch > 256) return 32;
    if (pitch > 128) return 16;
    return 8;
}

template <class OP>
static void launch_dist(hx_engine *e, uint32_t n_groups)
{
    const HxChannel &c = e->ch;
    const uint32_t pitch = (uint32_t)e->pitch;
    const size_t lds = e->pitch;
    switch (lanes_per_row(e->pitch)) {
    case 64: hipLaunchKernelGGL((k_dist_groups<OP, 64, 4, 4>), dim3(n_groups), dim3(256), lds, e->stream, e->d_rows, e->d_queries, pitch, c.d_grp_q, c.d_grp_off, c.d_ids, c.d_out); break;
    case 32: hipLaunchKernelGGL((k_dist_groups<OP, 32, 4, 2>), dim3(n_groups), dim3(256), lds, e->stream, e->d_rows, e->d_queries, pitch, c.d_grp_q, c.d_grp_off, c.d_ids, c.d_out); break;
    case 16: hipLaunchKernelGGL((k_dist_groups<OP, 16, 2, 2>), dim3(n_groups), dim3(128), lds, e->stream, e->d_rows, e->d_queries, pitch, c.d_grp_q, c.d_grp_off, c.d_ids, c.d_out); break;
    default: hipLaunchKernelGGL((k_dist_groups<OP, 8, 1, 2>), dim3(n_groups), dim3(64), lds, e->stream, e->d_rows, e->d_queries, pitch, c.d_grp_q, c.d_grp_off, c.d_ids, c.d_out); break;
    }
}

template <class OP, int STAGE, int MINW>
static hipError_t launch_pair_v(hx_engine *e, uint32_t n_wgs, uint32_t lds_rows)
{
    const HxChannel &c = e->ch;
    const size_t lds = 2 * (size_t)lds_rows * 1024u;
    static thread_local size_t attr_set = 0;
    if (lds > 65536 && attr_set < lds) {
        hipError_t s = hipFuncSetAttribute((const void *)k_pair_groups<OP, STAGE, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        if (s != hipSuccess) return s;
        attr_set = 160 * 1024;
    }
    hipLaunchKernelGGL((k_pair_groups<OP, STAGE, MINW>), dim3(n_wgs), dim3(HX_PAIR_WG), lds, e->stream, e->d_rows, (uint32_t)e->pitch,
                       c.d_pg_off, c.d_pg_na, c.d_pg_nb, c.d_pids, c.d_pg_out_off, c.d_wg_tab, c.d_pout, lds_rows);
    return hipGetLastError();
}
template <class OP>
static hipError_t launch_pair(hx_engine *e, uint32_t n_wgs, uint32_t lds_rows)
{
    // <= 40 rows per group (the (2m+1)-row back-link blocks at m <= 19): 5 staged rows per wave
    if (lds_rows <= 40) return launch_pair_v<OP, 5, 2>(e, n_wgs, lds_rows);
    return launch_pair_v<OP, 8, 2>(e, n_wgs, lds_rows);
}

int hx_engine::layout_round(const HxRound &r)
{
    HxChannel &c = ch;
    c.round = r;
    c.max_wgs = (uint32_t)(r.n_pout / HX_PAIR_SLAB + r.n_pgroups + 1);
    size_t o = 0;
    const size_t o_grp_q = o;      o = al16(o + (size_t)r.n_dgroups * 4);
    const size_t o_grp_off = o;    o = al16(o + ((size_t)r.n_dgroups + 1) * 4);
    const size_t o_ids = o;        o = al16(o + (size_t)r.n_dids * 4);
    const size_t o_pg_off = o;     o = al16(o + ((size_t)r.n_pgroups + 1) * 4);
    const size_t o_pg_out = o;     o = al16(o + (size_t)r.n_pgroups * 8);
    const size_t o_pg_na = o;      o = al16(o + (size_t)r.n_pgroups * 2);
    const size_t o_pg_nb = o;      o = al16(o + (size_t)r.n_pgroups * 2);
    const size_t o_pids = o;       o = al16(o + (size_t)r.n_pids * 4);
    const size_t o_wg = o;         o = al16(o + (size_t)c.max_wgs * 8);
    const size_t o_glist = o;      o = al16(o + (size_t)r.n_pgroups * 4);
    const size_t o_flag = o;       o = al16(o + (size_t)r.n_pgroups);
    c.req_bytes = o;
    const size_t o_out = 0, o_pout = al16((size_t)r.n_dids * 4);
    c.res_bytes = al16(o_pout + (size_t)r.n_pout * 4);
    if (c.req_bytes > c.cap_req) {
        size_t n = std::max(c.req_bytes, c.cap_req * 2) + 4096;
        if (c.h_req) (void)hipHostFree(c.h_req);
        if (c.d_req) (void)hipFree(c.d_req);
        c.h_req = c.d_req = nullptr; c.cap_req = 0;
        HX_HIP(this, hipHostMalloc((void **)&c.h_req, n, hipHostMallocDefault));
        HX_HIP(this, hipMalloc((void **)&c.d_req, n));
        c.cap_req = n;
    }
    if (c.res_bytes > c.cap_res) {
        size_t n = std::max(c.res_bytes, c.cap_res * 2) + 4096;
        if (c.h_res) (void)hipHostFree(c.h_res);
        if (c.d_res) (void)hipFree(c.d_res);
        c.h_res = c.d_res = nullptr; c.cap_res = 0;
        HX_HIP(this, hipHostMalloc((void **)&c.h_res, n, hipHostMallocDefault));
        HX_HIP(this, hipMalloc((void **)&c.d_res, n));
        c.cap_res = n;
    }
#define HX_SUB(T, name, off) c.h_##name = (T *)(c.h_req + (off)); c.d_##name = (T *)(c.d_req + (off))
    HX_SUB(uint32_t, grp_q, o_grp_q); HX_SUB(uint32_t, grp_off, o_grp_off); HX_SUB(uint32_t, ids, o_ids);
    HX_SUB(uint32_t, pg_off, o_pg_off); HX_SUB(uint64_t, pg_out_off, o_pg_out); HX_SUB(uint16_t, pg_na, o_pg_na);
    HX_SUB(uint16_t, pg_nb, o_pg_nb); HX_SUB(uint32_t, pids, o_pids); HX_SUB(uint32_t, wg_tab, o_wg); HX_SUB(uint32_t, glist, o_glist);
    c.h_pg_flag = c.h_req + o_flag;
    if (r.n_pgroups) memset(c.h_pg_flag, 0, r.n_pgroups);
#undef HX_SUB
    c.h_out = (float *)(c.h_res + o_out); c.d_out = (float *)(c.d_res + o_out);
    c.h_pout = (float *)(c.h_res + o_pout); c.d_pout = (float *)(c.d_res + o_pout);
    return HX_OK;
}

int hx_engine::run_round()
{
    HxChannel &c = ch;
    const HxRound &r = c.round;
    const bool do_dist = r.n_dgroups > 0 && r.n_dids > 0;
    // workgroup table + LDS rows of the pair launch; groups flagged for the matrix cores get one workgroup each in their own launch
    uint32_t n_wgs = 0, lds_rows = 1, n_mf = 0;
    if (r.n_pgroups > 0 && r.n_pout > 0) {
        for (uint32_t g = 0; g < r.n_pgroups; g++) {
            const uint32_t na = c.h_pg_na[g], nb = c.h_pg_nb[g];
            if (na + nb > HX_PAIR_MAX_ROWS) return fail(HX_E_ARG, "pair group exceeds HX_PAIR_MAX_ROWS");
            const uint32_t P = nb ? na * nb : na * (na - 1) / 2;
            if (P == 0) continue;
            if (c.h_pg_flag[g]) { c.h_glist[n_mf++] = g; continue; }
            lds_rows = std::max(lds_rows, na + nb);
            for (uint32_t p0 = 0; p0 < P; p0 += HX_PAIR_SLAB) {
                if (n_wgs >= c.max_wgs) return fail(HX_E_STATE, "pair workgroup table overflow");
                c.h_wg_tab[2 * n_wgs] = g; c.h_wg_tab[2 * n_wgs + 1] = p0; n_wgs++;
            }
        }
    }
    if (n_mf && (dtype != HX_F16 || metric != HX_NEG_IP)) return fail(HX_E_ARG, "the MFMA pair path serves halfvec inner product");
    if (!do_dist && n_wgs == 0 && n_mf == 0) return HX_OK;
    HX_HIP(this, hipMemcpyAsync(c.d_req, c.h_req, c.req_bytes, hipMemcpyHostToDevice, stream));
    if (dtype == HX_SPARSE && n_mf) return fail(HX_E_ARG, "the MFMA pair path serves halfvec inner product");
    if (do_dist && dtype == HX_SPARSE) {                        // sparsevec: the merge-join kernels of hx_sparse.hip on the same request arrays
        if (timing) HX_HIP(this, hipEventRecord(ev0, stream));
        HX_HIP(this, hx_launch_sparse_dist(this, r.n_dgroups));
        if (timing) HX_HIP(this, hipEventRecord(ev1, stream));
    } else
    if (do_dist) {
        if (timing) HX_HIP(this, hipEventRecord(ev0, stream));
#define F32C(K) launch_dist<OpF32<K>>(this, r.n_dgroups)
#define F16C(K) launch_dist<OpF16<K>>(this, r.n_dgroups)
        HX_DISPATCH(this, F32C, F16C, launch_dist<OpHamming>(this, r.n_dgroups), launch_dist<OpJaccard>(this, r.n_dgroups));
#undef F32C
#undef F16C
        HX_HIP(this, hipGetLastError());
        if (timing) HX_HIP(this, hipEventRecord(ev1, stream));
    }
    if (n_wgs && dtype == HX_SPARSE) {
        if (timing) HX_HIP(this, hipEventRecord(ev2, stream));
        HX_HIP(this, hx_launch_sparse_pairs(this, n_wgs));
        if (timing) HX_HIP(this, hipEventRecord(ev3, stream));
    } else
    if (n_wgs) {
        if (timing) HX_HIP(this, hipEventRecord(ev2, stream));
        hipError_t ls = hipSuccess;
#define F32C(K) ls = launch_pair<OpF32<K>>(this, n_wgs, lds_rows)
#define F16C(K) ls = launch_pair<OpF16<K>>(this, n_wgs, lds_rows)
        HX_DISPATCH(this, F32C, F16C, ls = launch_pair<OpHamming>(this, n_wgs, lds_rows), ls = launch_pair<OpJaccard>(this, n_wgs, lds_rows));
#undef F32C
#undef F16C
        HX_HIP(this, ls);
        if (timing) HX_HIP(this, hipEventRecord(ev3, stream));
    }
    if (n_mf) {
        if (timing) HX_HIP(this, hipEventRecord(ev4, stream));
        HX_HIP(this, hx_launch_pair_mfma(this, n_mf, c.d_glist));
        if (timing) HX_HIP(this, hipEventRecord(ev5, stream));
    }
    HX_HIP(this, hipMemcpyAsync(c.h_res, c.d_res, c.res_bytes, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipStreamSynchronize(stream));
    if (timing) {
        if (n_mf) { float ms = 0.f; HX_HIP(this, hipEventElapsedTime(&ms, ev4, ev5)); uint64_t np = 0; for (uint32_t k = 0; k < n_mf; k++) { const uint32_t g = c.h_glist[k]; const uint32_t na = c.h_pg_na[g], nb = c.h_pg_nb[g]; np += nb ? (uint64_t)na * nb : (uint64_t)na * (na - 1) / 2; }
                    stat_mfma.launches++; stat_mfma.units += np; stat_mfma.ms += ms; }
        if (do_dist) { HX_HIP(this, hipEventElapsedTime(&last_ms, ev0, ev1)); stat_dist.launches++; stat_dist.units += r.n_dids; stat_dist.ms += last_ms; }
        if (n_wgs) { float ms = 0.f; HX_HIP(this, hipEventElapsedTime(&ms, ev2, ev3)); last_ms = ms; stat_pair.launches++; stat_pair.units += r.n_pout; stat_pair.ms += ms; }
    }
    return HX_OK;
}

#include "hx_fused.inc.h"

// =================================================================================================
// C ABI: engine
// =================================================================================================
extern "C" {

int hx_abi_version(void) { return HX_ABI_VERSION; }

const char *hx_last_error(const hx_engine *e) { return e ? e->err.c_str() : g_create_err.c_str(); }

static int create_fail(int code, const std::string &msg) { g_create_err = msg; return code; }

int hx_create(int device, int dtype, int metric, int dim, uint64_t capacity_rows, hx_engine **out)
{
    if (!out) return create_fail(HX_E_ARG, "out is NULL");
    *out = nullptr;
    if (dtype < HX_F32 || dtype > HX_SPARSE) return create_fail(HX_E_ARG, "unknown dtype");
    const bool bit_metric = metric == HX_HAMMING || metric == HX_JACCARD;
    if (metric < HX_L2SQ || metric > HX_JACCARD || bit_metric != (dtype == HX_BIT))
        return create_fail(HX_E_ARG, "metric does not belong to this dtype's operator classes");
    // index dimension limits: hnsw_constants.rs:4 (vector 2000), halfvec.rs:876 (2x), bitvec.rs:184 (32x); sparsevec: SPARSEVEC_MAX_DIM 1e9,
    // at most 1000 non-zero elements in an indexed value
    const int max_dim = dtype == HX_F32 ? 2000 : dtype == HX_F16 ? 4000 : dtype == HX_BIT ? 64000 : 1000000000;
    if (dim < 1) return create_fail(HX_E_DIM, "column must have at least 1 dimension");
    if (dim > max_dim) return create_fail(HX_E_DIM, "column cannot have more than " + std::to_string(max_dim) + " dimensions for hnsw index");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return create_fail(HX_E_NODEVICE, "no HIP device visible: the engine has no CPU fallback");
    if (device < 0 || device >= ndev) return create_fail(HX_E_ARG, "device index out of range");
    hx_engine *e = new (std::nothrow) hx_engine();
    if (!e) return create_fail(HX_E_NOMEM, "out of host memory");
    e->device = device; e->dtype = dtype; e->metric = metric; e->dim = dim;
    e->row_bytes = dtype == HX_F32 ? (uint64_t)dim * 4 : dtype == HX_F16 ? (uint64_t)dim * 2 : dtype == HX_BIT ? (uint64_t)(dim + 7) / 8
                 : (16 + 8 * (uint64_t)std::min(dim, HX_SPARSE_MAX_NNZ) + 15) & ~(uint64_t)15;      // sparsevec record (include/hnswrx.h)
    e->pitch = (e->row_bytes + 15) & ~(uint64_t)15;
    e->capacity = capacity_rows ? capacity_rows : 1024;
    hipError_t s;
#define CREATE_HIP(call) if ((s = (call)) != hipSuccess) { std::string m = std::string(#call) + ": " + hipGetErrorString(s); hx_destroy(e); return create_fail(HX_E_HIP, m); }
    CREATE_HIP(hipSetDevice(device));
    CREATE_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    CREATE_HIP(hipEventCreate(&e->ev0));
    CREATE_HIP(hipEventCreate(&e->ev1));
    CREATE_HIP(hipEventCreate(&e->ev2));
    CREATE_HIP(hipEventCreate(&e->ev3));
    CREATE_HIP(hipEventCreate(&e->ev4));
    CREATE_HIP(hipEventCreate(&e->ev5));
    s = hipMalloc((void **)&e->d_rows, e->capacity * e->pitch);
    if (s != hipSuccess) { hx_destroy(e); return create_fail(HX_E_NOMEM, "cannot reserve row store in HBM"); }
    e->cap_queries = 64;
    CREATE_HIP(hipMalloc((void **)&e->d_queries, (size_t)(e->cap_queries + 1) * e->pitch));
    CREATE_HIP(hipMemsetAsync(e->d_queries, 0, (size_t)(e->cap_queries + 1) * e->pitch, e->stream));
    CREATE_HIP(hipStreamSynchronize(e->stream));
#undef CREATE_HIP
    *out = e;
    return HX_OK;
}

int hx_destroy(hx_engine *e)
{
    if (!e) return HX_OK;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    HxChannel &c = e->ch;
    HxMirror &mr = e->mirror;
    void *hp[] = {c.h_req, c.h_res, mr.h_stage, mr.io.h_io, mr.h_lk, e->grp.h_ctr, e->grp.h, e->bw.h, e->bw.h_ctr};
    void *dp[] = {c.d_req, c.d_res, e->d_rows, e->d_queries, mr.d_l0_ids, mr.d_l0_d, mr.d_l0_cnt, mr.d_level, mr.d_up_block, mr.d_up_ids, mr.d_up_d, mr.d_up_cnt, mr.io.d_vis, mr.d_stage, mr.io.d_io, mr.d_lk, mr.d_pm, mr.d_pm_valid, mr.io.d_spill,
                  mr.d_disc, mr.d_emask, mr.d_spill_big, mr.d_vis_big, e->grp.d, e->bw.d, e->bw.d_rec, e->d_xl, e->bw.d_wtab, e->bw.d_wt_valid, e->d_mf_norm2, e->d_wg, e->wsel.d_wl, e->wsel.d_cnt, e->wsel.d_slot, e->wsel.d_task, e->wsel.d_layer, e->wsel.d_counters};
    for (void *p : hp) if (p) (void)hipHostFree(p);
    for (void *p : dp) if (p) (void)hipFree(p);
    for (HxFusedIo &io : e->scan_io) {                           // pipelined scan slots: a launch still in flight is drained first
        if (io.stream) (void)hipStreamSynchronize(io.stream);
        if (io.h_io) (void)hipHostFree(io.h_io);
        if (io.d_io) (void)hipFree(io.d_io);
        if (io.d_vis) (void)hipFree(io.d_vis);
        if (io.d_spill) (void)hipFree(io.d_spill);
        if (io.ev0) (void)hipEventDestroy(io.ev0);
        if (io.ev1) (void)hipEventDestroy(io.ev1);
        if (io.ev_dep) (void)hipEventDestroy(io.ev_dep);
        if (io.stream && io.own_stream) (void)hipStreamDestroy(io.stream);
    }
    if (e->ev_scan_epoch) (void)hipEventDestroy(e->ev_scan_epoch);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->ev2) (void)hipEventDestroy(e->ev2);
    if (e->ev3) (void)hipEventDestroy(e->ev3);
    if (e->ev4) (void)hipEventDestroy(e->ev4);
    if (e->ev5) (void)hipEventDestroy(e->ev5);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return HX_OK;
}

int hx_dim(const hx_engine *e) { return e ? e->dim : HX_E_ARG; }
uint64_t hx_row_bytes(const hx_engine *e) { return e ? e->row_bytes : 0; }
uint64_t hx_num_rows(const hx_engine *e) { return e ? e->n_rows : 0; }
void *hx_stream(const hx_engine *e) { return e ? (void *)e->stream : nullptr; }

static int append_impl(hx_engine *e, const void *rows, uint64_t n, uint64_t *first, hipMemcpyKind kind)
{
    if (!e) return HX_E_ARG;
    if (!rows && n) return e->fail(HX_E_ARG, "rows is NULL");
    if (e->n_rows + n > e->capacity) return e->fail(HX_E_NOMEM, "row store capacity exceeded (hx_create capacity_rows)");
    if (e->n_rows + n > 0x7fffffffull) return e->fail(HX_E_NOMEM, "row ids are 31-bit");
    HX_HIP(e, hipSetDevice(e->device));
    uint8_t *dst = e->d_rows + e->n_rows * e->pitch;
    if (n) {
        // a device source was produced on a stream this library does not know (the engine's own is non-blocking): wait for the device, so that rows
        // still being written by the caller's kernels are not copied half-done
        if (kind == hipMemcpyDeviceToDevice) HX_HIP(e, hipDeviceSynchronize());
        if (e->pitch != e->row_bytes) HX_HIP(e, hipMemsetAsync(dst, 0, n * e->pitch, e->stream));
        HX_HIP(e, hipMemcpy2DAsync(dst, e->pitch, rows, e->row_bytes, e->row_bytes, n, kind, e->stream));
        HX_HIP(e, hipStreamSynchronize(e->stream));
    }
    if (first) *first = e->n_rows;
    e->n_rows += n;
    return HX_OK;
}
int hx_append_rows(hx_engine *e, const void *rows_host, uint64_t n, uint64_t *first_row_id) { return append_impl(e, rows_host, n, first_row_id, hipMemcpyHostToDevice); }
int hx_append_rows_device(hx_engine *e, const void *rows_dev, uint64_t n, uint64_t *first_row_id) { return append_impl(e, rows_dev, n, first_row_id, hipMemcpyDeviceToDevice); }

int hx_pop_rows(hx_engine *e, uint64_t n)
{
    if (!e) return HX_E_ARG;
    if (n > e->n_rows) return e->fail(HX_E_ARG, "cannot pop more rows than stored");
    e->n_rows -= n;
    return HX_OK;
}

int hx_read_rows(hx_engine *e, uint64_t first, uint64_t n, void *rows_host)
{
    if (!e || (!rows_host && n)) return HX_E_ARG;
    if (first + n > e->n_rows) return e->fail(HX_E_ARG, "row range out of bounds");
    if (!n) return HX_OK;
    HX_HIP(e, hipSetDevice(e->device));
    HX_HIP(e, hipMemcpy2DAsync(rows_host, e->row_bytes, e->d_rows + first * e->pitch, e->pitch, e->row_bytes, n, hipMemcpyDeviceToHost, e->stream));
    HX_HIP(e, hipStreamSynchronize(e->stream));
    return HX_OK;
}

static int normalize_region(hx_engine *e, uint8_t *base, uint64_t n, double *norms_host)
{
    if (e->dtype == HX_BIT) return e->fail(HX_E_ARG, "bit columns have no norm procedure");
    if (!n) return HX_OK;
    double *d_norms = nullptr;
    HX_HIP(e, hipMalloc((void **)&d_norms, n * sizeof(double)));
    const uint32_t blk = 64, grid = (uint32_t)((n + blk - 1) / blk);
    if (e->dtype == HX_SPARSE) (void)hx_launch_sparse_normalize(e, base, n, d_norms);
    else if (e->dtype == HX_F32) hipLaunchKernelGGL((k_normalize<HX_F32>), dim3(grid), dim3(blk), 0, e->stream, base, (uint32_t)e->pitch, e->dim, n, d_norms);
    else hipLaunchKernelGGL((k_normalize<HX_F16>), dim3(grid), dim3(blk), 0, e->stream, base, (uint32_t)e->pitch, e->dim, n, d_norms);
    hipError_t s = hipGetLastError();
    if (s == hipSuccess && norms_host) s = hipMemcpyAsync(norms_host, d_norms, n * sizeof(double), hipMemcpyDeviceToHost, e->stream);
    if (s == hipSuccess) s = hipStreamSynchronize(e->stream);
    (void)hipFree(d_norms);
    if (s != hipSuccess) return e->fail(HX_E_HIP, std::string("normalize: ") + hipGetErrorString(s));
    return HX_OK;
}

int hx_normalize_rows(hx_engine *e, uint64_t first, uint64_t n, double *norms_host)
{
    if (!e) return HX_E_ARG;
    if (first + n > e->n_rows) return e->fail(HX_E_ARG, "row range out of bounds");
    HX_HIP(e, hipSetDevice(e->device));
    return normalize_region(e, e->d_rows + first * e->pitch, n, norms_host);
}

static int set_queries_impl(hx_engine *e, const void *q, uint32_t nq, int normalize, hipMemcpyKind kind)
{
    if (!e) return HX_E_ARG;
    if (!q && nq) return e->fail(HX_E_ARG, "queries is NULL");
    if (nq >= 0x7fffffffu) return e->fail(HX_E_ARG, "too many queries");
    HX_HIP(e, hipSetDevice(e->device));
    if (nq > e->cap_queries) {
        (void)hipFree(e->d_queries); e->d_queries = nullptr;
        e->cap_queries = nq;
        HX_HIP(e, hipMalloc((void **)&e->d_queries, (size_t)(e->cap_queries + 1) * e->pitch));
    }
    if (nq) {
        if (kind == hipMemcpyDeviceToDevice) HX_HIP(e, hipDeviceSynchronize());      // as hx_append_rows_device: the caller's producer stream is unknown
        if (e->pitch != e->row_bytes) HX_HIP(e, hipMemsetAsync(e->d_queries, 0, (size_t)nq * e->pitch, e->stream));
        HX_HIP(e, hipMemcpy2DAsync(e->d_queries, e->pitch, q, e->row_bytes, e->row_bytes, nq, kind, e->stream));
    }
    e->n_queries = nq;
    if (normalize) return normalize_region(e, e->d_queries, nq, nullptr);
    HX_HIP(e, hipStreamSynchronize(e->stream));
    return HX_OK;
}
int hx_set_queries(hx_engine *e, const void *queries_host, uint32_t nq, int normalize) { return set_queries_impl(e, queries_host, nq, normalize, hipMemcpyHostToDevice); }
int hx_set_queries_device(hx_engine *e, const void *queries_dev, uint32_t nq, int normalize) { return set_queries_impl(e, queries_dev, nq, normalize, hipMemcpyDeviceToDevice); }

static int check_ids(hx_engine *e, const uint32_t *ids, size_t n)
{
    for (size_t i = 0; i < n; i++)
        if (ids[i] >= e->n_rows) return e->fail(HX_E_ARG, "row id out of range");
    return HX_OK;
}

int hx_distances_batch(hx_engine *e, uint32_t n_groups, const uint32_t *group_query,
                       const uint32_t *group_offsets, const uint32_t *row_ids, float *out)
{
    if (!e) return HX_E_ARG;
    if (n_groups == 0) return HX_OK;
    if (!group_query || !group_offsets || !row_ids || !out) return e->fail(HX_E_ARG, "NULL argument");
    const uint32_t n_ids = group_offsets[n_groups];
    int rc;
    if ((rc = check_ids(e, row_ids, n_ids))) return rc;
    for (uint32_t g = 0; g < n_groups; g++) {
        if (group_offsets[g + 1] < group_offsets[g]) return e->fail(HX_E_ARG, "group offsets must be non-decreasing");
        uint32_t q = group_query[g];
        if (q & HX_QUERY_SLOT) { if ((q & 0x7fffffffu) > e->cap_queries) return e->fail(HX_E_ARG, "query slot out of range"); }
        else if (q >= e->n_rows) return e->fail(HX_E_ARG, "query row id out of range");
    }
    HX_HIP(e, hipSetDevice(e->device));
    HxRound r; r.n_dgroups = n_groups; r.n_dids = n_ids;
    if ((rc = e->layout_round(r))) return rc;
    memcpy(e->ch.h_grp_q, group_query, n_groups * sizeof(uint32_t));
    memcpy(e->ch.h_grp_off, group_offsets, (n_groups + 1) * sizeof(uint32_t));
    memcpy(e->ch.h_ids, row_ids, n_ids * sizeof(uint32_t));
    if ((rc = e->run_round())) return rc;
    memcpy(out, e->ch.h_out, n_ids * sizeof(float));
    return HX_OK;
}

int hx_distances(hx_engine *e, const void *query_host, const uint32_t *row_ids, uint32_t n, float *out)
{
    if (!e) return HX_E_ARG;
    if (n == 0) return HX_OK;
    if (!query_host || !row_ids || !out) return e->fail(HX_E_ARG, "NULL argument");
    HX_HIP(e, hipSetDevice(e->device));
    // the scratch slot after the query set holds the single query
    uint8_t *slot = e->d_queries + (size_t)e->cap_queries * e->pitch;
    if (e->pitch != e->row_bytes) HX_HIP(e, hipMemsetAsync(slot, 0, e->pitch, e->stream));
    HX_HIP(e, hipMemcpyAsync(slot, query_host, e->row_bytes, hipMemcpyHostToDevice, e->stream));
    // split into groups of 256 rows so one long list still fills the chip
    const uint32_t G = 256, n_groups = (n + G - 1) / G;
    std::vector<uint32_t> gq(n_groups, HX_QUERY_SLOT | e->cap_queries), go(n_groups + 1);
    for (uint32_t g = 0; g <= n_groups; g++) go[g] = std::min(n, g * G);
    return hx_distances_batch(e, n_groups, gq.data(), go.data(), row_ids, out);
}

int hx_pairwise_many(hx_engine *e, uint32_t n_groups, const uint32_t *group_offsets,
                     const uint16_t *na, const uint16_t *nb, const uint32_t *ids,
                     const uint64_t *out_offsets, float *out)
{
    if (!e) return HX_E_ARG;
    if (n_groups == 0) return HX_OK;
    if (!group_offsets || !na || !nb || !ids || !out_offsets || !out) return e->fail(HX_E_ARG, "NULL argument");
    const uint32_t n_ids = group_offsets[n_groups];
    int rc;
    if ((rc = check_ids(e, ids, n_ids))) return rc;
    uint64_t n_out = 0;
    for (uint32_t g = 0; g < n_groups; g++) {
        if (group_offsets[g + 1] - group_offsets[g] != (uint32_t)na[g] + nb[g]) return e->fail(HX_E_ARG, "group_offsets disagree with na+nb");
        if ((uint32_t)na[g] + nb[g] > HX_PAIR_MAX_ROWS) return e->fail(HX_E_ARG, "pair group exceeds HX_PAIR_MAX_ROWS");
        uint64_t P = nb[g] ? (uint64_t)na[g] * nb[g] : (uint64_t)na[g] * (na[g] ? na[g] - 1 : 0) / 2;
        n_out = std::max(n_out, out_offsets[g] + P);
    }
    HX_HIP(e, hipSetDevice(e->device));
    HxRound r; r.n_pgroups = n_groups; r.n_pids = n_ids; r.n_pout = n_out;
    if ((rc = e->layout_round(r))) return rc;
    memcpy(e->ch.h_pg_off, group_offsets, (n_groups + 1) * sizeof(uint32_t));
    memcpy(e->ch.h_pg_na, na, n_groups * sizeof(uint16_t));
    memcpy(e->ch.h_pg_nb, nb, n_groups * sizeof(uint16_t));
    memcpy(e->ch.h_pg_out_off, out_offsets, n_groups * sizeof(uint64_t));
    memcpy(e->ch.h_pids, ids, n_ids * sizeof(uint32_t));
    if ((rc = e->run_round())) return rc;
    memcpy(out, e->ch.h_pout, n_out * sizeof(float));
    return HX_OK;
}

// The same pair blocks on the matrix cores (halfvec inner product only): values in MFMA summation order, i.e. within
// 2 * dim * 2^-24 * |a| |b| of hx_pairwise_many's; norm2_out (nullable, one per id) receives |row|^2 for that bound.
int hx_pairwise_many_mfma(hx_engine *e, uint32_t n_groups, const uint32_t *group_offsets, const uint16_t *na, const uint16_t *nb, const uint32_t *ids,
                          const uint64_t *out_offsets, float *out, float *norm2_out)
{
    if (!e) return HX_E_ARG;
    if (n_groups == 0) return HX_OK;
    if (!group_offsets || !na || !nb || !ids || !out_offsets || !out) return e->fail(HX_E_ARG, "NULL argument");
    if (e->dtype != HX_F16 || e->metric != HX_NEG_IP) return e->fail(HX_E_ARG, "the MFMA pair path serves halfvec inner product");
    const uint32_t n_ids = group_offsets[n_groups];
    int rc;
    if ((rc = check_ids(e, ids, n_ids))) return rc;
    uint64_t n_out = 0;
    for (uint32_t g = 0; g < n_groups; g++) {
        if (group_offsets[g + 1] - group_offsets[g] != (uint32_t)na[g] + nb[g]) return e->fail(HX_E_ARG, "group_offsets disagree with na+nb");
        if ((uint32_t)na[g] + nb[g] > HX_PAIR_MAX_ROWS) return e->fail(HX_E_ARG, "pair group exceeds HX_PAIR_MAX_ROWS");
        const uint64_t P = nb[g] ? (uint64_t)na[g] * nb[g] : (uint64_t)na[g] * (na[g] ? na[g] - 1 : 0) / 2;
        n_out = std::max(n_out, out_offsets[g] + P);
    }
    HX_HIP(e, hipSetDevice(e->device));
    HxRound r; r.n_pgroups = n_groups; r.n_pids = n_ids; r.n_pout = n_out;
    if ((rc = e->layout_round(r))) return rc;
    memcpy(e->ch.h_pg_off, group_offsets, (n_groups + 1) * sizeof(uint32_t));
    memcpy(e->ch.h_pg_na, na, n_groups * sizeof(uint16_t));
    memcpy(e->ch.h_pg_nb, nb, n_groups * sizeof(uint16_t));
    memcpy(e->ch.h_pg_out_off, out_offsets, n_groups * sizeof(uint64_t));
    memcpy(e->ch.h_pids, ids, n_ids * sizeof(uint32_t));
    memset(e->ch.h_pg_flag, 1, n_groups);
    if ((rc = e->run_round())) return rc;
    memcpy(out, e->ch.h_pout, n_out * sizeof(float));
    if (norm2_out) {
        if ((rc = e->mfma_norms(e->n_rows))) return rc;
        for (uint32_t i = 0; i < n_ids; i++) norm2_out[i] = e->h_mf_norm2[ids[i]];
    }
    return HX_OK;
}

int hx_pairwise(hx_engine *e, const uint32_t *ids, uint32_t w, float *out)
{
    if (!e) return HX_E_ARG;
    if (w == 0) return HX_OK;
    if (!ids || !out) return e->fail(HX_E_ARG, "NULL argument");
    int rc;
    // diagonal through K1 (d(a,a) is not 0 for inner product)
    {
        std::vector<uint32_t> gq(ids, ids + w), go(w + 1);
        for (uint32_t i = 0; i <= w; i++) go[i] = i;
        std::vector<float> diag(w);
        if ((rc = hx_distances_batch(e, w, gq.data(), go.data(), ids, diag.data()))) return rc;
        for (uint32_t i = 0; i < w; i++) out[(size_t)i * w + i] = diag[i];
    }
    if (w == 1) return HX_OK;
    // off-diagonal through K2 in 32-row blocks: triangular groups on the block diagonal, rectangles below it
    const uint32_t B = 32, nblk = (w + B - 1) / B;
    std::vector<uint32_t> goff(1, 0), gids;
    std::vector<uint16_t> gna, gnb;
    std::vector<uint64_t> ooff;
    struct Blk { uint32_t bi, bj; };
    std::vector<Blk> blks;
    uint64_t n_out = 0;
    for (uint32_t bi = 0; bi < nblk; bi++)
        for (uint32_t bj = 0; bj <= bi; bj++) {
            uint32_t ai = std::min(B, w - bi * B), aj = std::min(B, w - bj * B);
            if (bi == bj && ai < 2) continue;
            for (uint32_t i = 0; i < ai; i++) gids.push_back(ids[bi * B + i]);
            if (bi != bj) for (uint32_t j = 0; j < aj; j++) gids.push_back(ids[bj * B + j]);
            gna.push_back((uint16_t)ai); gnb.push_back((uint16_t)(bi == bj ? 0 : aj));
            goff.push_back((uint32_t)gids.size());
            ooff.push_back(n_out);
            n_out += bi == bj ? (uint64_t)ai * (ai - 1) / 2 : (uint64_t)ai * aj;
            blks.push_back({bi, bj});
        }
    std::vector<float> tmp(n_out);
    if ((rc = hx_pairwise_many(e, (uint32_t)blks.size(), goff.data(), gna.data(), gnb.data(), gids.data(), ooff.data(), tmp.data()))) return rc;
    for (size_t g = 0; g < blks.size(); g++) {
        uint32_t bi = blks[g].bi, bj = blks[g].bj, ai = gna[g], aj = gnb[g];
        const float *t = tmp.data() + ooff[g];
        if (bi == bj) {
            for (uint32_t i = 1; i < ai; i++) for (uint32_t j = 0; j < i; j++) {
                float d = t[(size_t)i * (i - 1) / 2 + j];
                out[(size_t)(bi * B + i) * w + bj * B + j] = d; out[(size_t)(bj * B + j) * w + bi * B + i] = d;
            }
        } else {
            for (uint32_t i = 0; i < ai; i++) for (uint32_t j = 0; j < aj; j++) {
                float d = t[(size_t)i * aj + j];
                out[(size_t)(bi * B + i) * w + bj * B + j] = d; out[(size_t)(bj * B + j) * w + bi * B + i] = d;
            }
        }
    }
    return HX_OK;
}

int hx_rows_equal(hx_engine *e, uint32_t n_pairs, const uint32_t *a_ids, const uint32_t *b_ids, uint8_t *equal_out)
{
    if (!e) return HX_E_ARG;
    if (n_pairs == 0) return HX_OK;
    if (!a_ids || !b_ids || !equal_out) return e->fail(HX_E_ARG, "NULL argument");
    int rc;
    if ((rc = check_ids(e, a_ids, n_pairs)) || (rc = check_ids(e, b_ids, n_pairs))) return rc;
    HX_HIP(e, hipSetDevice(e->device));
    uint32_t *d_a = nullptr; uint8_t *d_eq = nullptr;
    HX_HIP(e, hipMalloc((void **)&d_a, 2 * (size_t)n_pairs * sizeof(uint32_t)));
    hipError_t s = hipMalloc((void **)&d_eq, n_pairs);
    if (s == hipSuccess) s = hipMemcpyAsync(d_a, a_ids, n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream);
    if (s == hipSuccess) s = hipMemcpyAsync(d_a + n_pairs, b_ids, n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream);
    if (s == hipSuccess) {
        hipLaunchKernelGGL(k_rows_equal, dim3((n_pairs + 3) / 4), dim3(256), 0, e->stream, e->d_rows, (uint32_t)e->pitch, n_pairs, d_a, d_a + n_pairs, d_eq);
        s = hipGetLastError();
    }
    if (s == hipSuccess) s = hipMemcpyAsync(equal_out, d_eq, n_pairs, hipMemcpyDeviceToHost, e->stream);
    if (s == hipSuccess) s = hipStreamSynchronize(e->stream);
    (void)hipFree(d_a); if (d_eq) (void)hipFree(d_eq);
    if (s != hipSuccess) return e->fail(HX_E_HIP, std::string("rows_equal: ") + hipGetErrorString(s));
    return HX_OK;
}

int hx_set_timing(hx_engine *e, int enabled) { if (!e) return HX_E_ARG; e->timing = enabled != 0; return HX_OK; }
int hx_last_kernel_ms(hx_engine *e, float *ms) { if (!e || !ms) return HX_E_ARG; *ms = e->last_ms; return HX_OK; }

int hx_kernel_stats(hx_engine *e, int kind, uint64_t *launches, uint64_t *units, double *ms, int reset)
{
    if (!e) return HX_E_ARG;
    HxKernelStat &s = kind == 0 ? e->stat_dist : kind == 1 ? e->stat_pair : kind == 2 ? e->stat_fused : kind == 3 ? e->stat_links : kind == 5 ? e->stat_scan : kind == 6 ? e->stat_wsel : e->stat_mfma;
    if (kind == 5 && reset) { e->scan_epoch_set = false; e->scan_last_end = 0.0; }
    if (launches) *launches = s.launches;
    if (units) *units = s.units;
    if (ms) *ms = s.ms;
    if (reset) s = HxKernelStat();
    return HX_OK;
}

} // extern "C"

/*
 * hnsw_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the arithmetic and graph control flow of the
 * HNSW hot path of maropu/pgvector-rx (reference, Rust).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * file's library; the product (pgvector-rx_amd/) never links, imports or
 * falls back to it.
 *
 * Parity pinning: the distance functions and graph functions below are
 * checked in tests/test_oracle_golden.py against every known-answer test,
 * pure-Rust unit test and pg_regress expected ordering the reference holds
 * for this path (tests/golden/reference_known_answers.json lists each with
 * its file:line).  Two inputs are NOT pinned by any reference fixture and
 * are restated from the published Rust std algorithm:
 *   - std::collections::BinaryHeap push/pop sift order (tie order among
 *     equal distances)                                   -> "tie-order parity unpinned"
 *   - rand::random level draws (the oracle takes explicit levels instead)
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 * -ffp-contract=off matters: rustc never fuses a*b+c.
 *
 * Summation orders ("order" argument):
 *   ORC_ORDER_SEQ  = the reference's order: one f32 accumulator, index order
 *                    (src/types/vector.rs:518-567, halfvec.rs:687-733).
 *   ORC_ORDER_W64  = the device's canonical order (DESIGN.md "canonical
 *                    summation order"): 64 lane partials, lane l owning
 *                    elements c*64*V + l*V + t (V = 16 bytes / elem size),
 *                    accumulated in increasing (c,t); then an xor butterfly
 *                    32,16,8,4,2,1.  Lets tests demand BIT-EXACT equality
 *                    between the HIP kernels and this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <float.h>

#define ORC_API __attribute__((visibility("default")))

enum { ORC_F32 = 0, ORC_F16 = 1, ORC_BIT = 2, ORC_SPARSE = 3 };
enum { ORC_L2SQ = 0, ORC_NEG_IP = 1, ORC_L1 = 2, ORC_HAMMING = 3, ORC_JACCARD = 4 };
enum { ORC_ORDER_SEQ = 0, ORC_ORDER_W64 = 1, ORC_ORDER_VEC = 2 };
enum { ORC_ITER_OFF = 0, ORC_ITER_RELAXED = 1, ORC_ITER_STRICT = 2 };

#define HNSW_HEAPTIDS 10 /* src/hnsw_constants.rs:85 */

/* ------------------------------------------------------------------ */
/* half <-> float, restating src/types/halfvec.rs:54-87 and :92-143    */
/* ------------------------------------------------------------------ */
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

ORC_API float orc_half_to_f32(uint16_t h)
{
    uint32_t sign = (h >> 15) & 1u, exp = (h >> 10) & 0x1fu, mant = h & 0x3ffu;
    if (exp == 0) {
        if (mant == 0) return bits_f32(sign << 31);
        /* denormal: shift the mantissa up until the hidden bit appears */
        int e = -14;
        while ((mant & 0x400u) == 0) { mant <<= 1; e -= 1; }
        mant &= 0x3ffu;
        return bits_f32((sign << 31) | ((uint32_t)(e + 127) << 23) | (mant << 13));
    }
    if (exp == 31) return bits_f32((sign << 31) | (0xffu << 23) | (mant << 13));
    return bits_f32((sign << 31) | ((exp - 15 + 127) << 23) | (mant << 13));
}

ORC_API uint16_t orc_f32_to_half(float f)
{
    uint32_t bits = f32_bits(f);
    uint16_t sign = (uint16_t)((bits >> 31) & 1u);
    int exp = (int)((bits >> 23) & 0xffu);
    uint32_t mant = bits & 0x7fffffu;
    if (exp == 0xff) {
        if (mant == 0) return (uint16_t)((sign << 15) | (0x1f << 10));
        uint16_t m = (uint16_t)(mant >> 13);
        if (m < 1) m = 1;
        return (uint16_t)((sign << 15) | (0x1f << 10) | m);
    }
    if (exp > 142) return (uint16_t)((sign << 15) | (0x1f << 10));
    if (exp < 103) return (uint16_t)(sign << 15);
    if (exp < 113) {
        int shift = 113 - exp;
        uint32_t full = mant | 0x800000u;
        uint32_t m = full >> (shift + 13);
        uint32_t round_bit = (full >> (shift + 12)) & 1u;
        int sticky = (full & ((1u << (shift + 12)) - 1u)) != 0;
        uint16_t r = (uint16_t)((sign << 15) | (uint16_t)m);
        if (round_bit && (sticky || (m & 1u))) r = (uint16_t)(r + 1);
        return r;
    }
    {
        uint16_t half_exp = (uint16_t)((exp - 127 + 15) & 0x1f);
        uint16_t half_mant = (uint16_t)(mant >> 13);
        uint32_t round_bit = (mant >> 12) & 1u;
        uint32_t sticky = mant & 0xfffu;
        uint16_t r = (uint16_t)((sign << 15) | (half_exp << 10) | half_mant);
        if (round_bit && (sticky || (half_mant & 1u))) r = (uint16_t)(r + 1);
        return r;
    }
}

/* ------------------------------------------------------------------ */
/* scalar distance loops                                               */
/* ------------------------------------------------------------------ */
enum { T_L2 = 0, T_IP = 1, T_L1 = 2 };

static inline float term(int kind, float a, float b)
{
    if (kind == T_L2) { float d = a - b; return d * d; }   /* vector.rs:522-523 */
    if (kind == T_IP) return a * b;                        /* vector.rs:534     */
    return fabsf(a - b);                                   /* vector.rs:564     */
}

static inline float elem_f32(const void *p, int dtype, int i)
{
    if (dtype == ORC_F32) return ((const float *)p)[i];
    return orc_half_to_f32(((const uint16_t *)p)[i]);      /* halfvec.rs:692 */
}

/* reference order: single f32 accumulator in index order */
static float acc_seq(int kind, int dtype, int dim, const void *a, const void *b)
{
    float acc = 0.0f;
    for (int i = 0; i < dim; i++) acc += term(kind, elem_f32(a, dtype, i), elem_f32(b, dtype, i));
    return acc;
}

/* device canonical order (see header): the plain restatement ... */
static float acc_w64_plain(int kind, int dtype, int dim, const void *a, const void *b)
{
    const int V = (dtype == ORC_F32) ? 4 : 8;
    float p[64], t[64];
    for (int l = 0; l < 64; l++) {
        float acc = 0.0f;
        for (int base = l * V; base < dim; base += 64 * V)
            for (int k = 0; k < V && base + k < dim; k++)
                acc = acc + term(kind, elem_f32(a, dtype, base + k), elem_f32(b, dtype, base + k));
        p[l] = acc;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        for (int l = 0; l < 64; l++) t[l] = p[l] + p[l ^ off];
        memcpy(p, t, sizeof p);
    }
    return p[0];
}
/* ... and the same sums arranged so that the compiler can run the 64 lane accumulators side by side (the GPU tests build graphs of
 * thousands of wide rows through this order): per 64*V-element chunk the terms are formed first, then added to the lane partials for
 * k = 0..V-1 in turn -- each partial still receives its own terms in ascending (chunk, k) order, mul and add rounded separately.
 * tests/test_oracle_golden.py checks it against acc_w64_plain bit for bit. */
static const float *h2f_table(void)
{   /* half_to_f32 (halfvec.rs:54-87) of all 65 536 halves, computed once */
    static float *tab = NULL; static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_mutex_lock(&mu);
    if (!tab) { float *t = malloc(65536 * sizeof(float)); for (int i = 0; i < 65536; i++) t[i] = orc_half_to_f32((uint16_t)i); tab = t; }
    pthread_mutex_unlock(&mu);
    return tab;
}
#define W64_BODY(V, LOADA, LOADB)                                                                                   \
    float p[64] = {0}, t[64 * V], u[64];                                                                            \
    const int CH = 64 * V;                                                                                          \
    int c0 = 0;                                                                                                     \
    for (; c0 + CH <= dim; c0 += CH) {                                                                              \
        if (kind == T_L2) for (int i = 0; i < CH; i++) { const float d = LOADA(c0 + i) - LOADB(c0 + i); t[i] = d * d; } \
        else if (kind == T_IP) for (int i = 0; i < CH; i++) t[i] = LOADA(c0 + i) * LOADB(c0 + i);                   \
        else for (int i = 0; i < CH; i++) t[i] = fabsf(LOADA(c0 + i) - LOADB(c0 + i));                              \
        for (int k = 0; k < V; k++) for (int l = 0; l < 64; l++) p[l] = p[l] + t[l * V + k];                        \
    }                                                                                                               \
    for (int i = c0; i < dim; i++) { const int l = (i - c0) / V; p[l] = p[l] + term(kind, LOADA(i), LOADB(i)); }     \
    for (int off = 32; off >= 1; off >>= 1) { for (int l = 0; l < 64; l++) u[l] = p[l] + p[l ^ off]; memcpy(p, u, sizeof p); } \
    return p[0];
__attribute__((optimize("O3", "fp-contract=off"), target_clones("avx512f", "avx2", "default")))
static float acc_w64_f32(int kind, int dim, const float *a, const float *b)
{
#define LA(i) a[i]
#define LB(i) b[i]
    W64_BODY(4, LA, LB)
#undef LA
#undef LB
}
__attribute__((optimize("O3", "fp-contract=off"), target_clones("avx512f", "avx2", "default")))
static float acc_w64_f16(int kind, int dim, const uint16_t *a, const uint16_t *b, const float *tab)
{
#define LA(i) tab[a[i]]
#define LB(i) tab[b[i]]
    W64_BODY(8, LA, LB)
#undef LA
#undef LB
}
static float acc_w64(int kind, int dtype, int dim, const void *a, const void *b)
{
    if (dtype == ORC_F32) return acc_w64_f32(kind, dim, (const float *)a, (const float *)b);
    return acc_w64_f16(kind, dim, (const uint16_t *)a, (const uint16_t *)b, h2f_table());
}
ORC_API float orc_acc_w64_plain(int kind, int dtype, int dim, const void *a, const void *b) { return acc_w64_plain(kind, dtype, dim, a, b); }
ORC_API float orc_acc_w64_fast(int kind, int dtype, int dim, const void *a, const void *b) { return acc_w64(kind, dtype, dim, a, b); }

/* ORC_ORDER_VEC: a reassociated, compiler-vectorised CPU variant (16 partial sums, AVX-512 / AVX2 picked at load time).
 * It is NOT the reference's arithmetic (the reference is the scalar loop above); it exists only so that bench.py can
 * also quote the GPU against a CPU baseline that is not handicapped by the reference's dependent add chain
 * ("vectorised CPU variant").  f32 rows only; halfvec falls back to the scalar order. */
#define VEC_LOOP(EXPR)                                                                         \
    float acc[16] = {0};                                                                       \
    int i = 0;                                                                                 \
    for (; i + 16 <= dim; i += 16)                                                             \
        for (int j = 0; j < 16; j++) { const float x = a[i + j], y = b[i + j]; acc[j] += (EXPR); } \
    float tail = 0.0f;                                                                         \
    for (; i < dim; i++) { const float x = a[i], y = b[i]; tail += (EXPR); }                   \
    for (int w = 8; w >= 1; w >>= 1) for (int j = 0; j < w; j++) acc[j] += acc[j + w];         \
    return acc[0] + tail;
__attribute__((optimize("O3"), target_clones("avx512f", "avx2", "default")))
static float vec_l2(const float *a, const float *b, int dim) { VEC_LOOP((x - y) * (x - y)) }
__attribute__((optimize("O3"), target_clones("avx512f", "avx2", "default")))
static float vec_ip(const float *a, const float *b, int dim) { VEC_LOOP(x * y) }
__attribute__((optimize("O3"), target_clones("avx512f", "avx2", "default")))
static float vec_l1(const float *a, const float *b, int dim) { VEC_LOOP(fabsf(x - y)) }
static float acc_vec(int kind, int dtype, int dim, const void *a, const void *b)
{
    if (dtype != ORC_F32) return acc_seq(kind, dtype, dim, a, b);
    if (kind == T_L2) return vec_l2((const float *)a, (const float *)b, dim);
    if (kind == T_IP) return vec_ip((const float *)a, (const float *)b, dim);
    return vec_l1((const float *)a, (const float *)b, dim);
}

static inline float acc_f(int kind, int dtype, int dim, const void *a, const void *b, int order)
{
    if (order == ORC_ORDER_VEC) return acc_vec(kind, dtype, dim, a, b);
    return order == ORC_ORDER_SEQ ? acc_seq(kind, dtype, dim, a, b) : acc_w64(kind, dtype, dim, a, b);
}

static const uint8_t *popcnt_table(void)
{   /* PostgreSQL's pg_number_of_ones[256] (used at bitvec.rs:102) is the 8-bit popcount */
    static uint8_t tab[256]; static int init = 0;
    if (!init) { for (int i = 0; i < 256; i++) { int c = 0; for (int b = 0; b < 8; b++) c += (i >> b) & 1; tab[i] = (uint8_t)c; } init = 1; }
    return tab;
}

/* bitvec.rs:97-106 */
ORC_API uint64_t orc_hamming(const uint8_t *a, const uint8_t *b, int nbytes)
{
    const uint8_t *tab = popcnt_table(); uint64_t d = 0;
    for (int i = 0; i < nbytes; i++) d += tab[a[i] ^ b[i]];
    return d;
}

/* bitvec.rs:113-132 */
ORC_API double orc_jaccard(const uint8_t *a, const uint8_t *b, int nbytes)
{
    const uint8_t *tab = popcnt_table(); uint64_t ab = 0, aa = 0, bb = 0;
    for (int i = 0; i < nbytes; i++) { ab += tab[a[i] & b[i]]; aa += tab[a[i]]; bb += tab[b[i]]; }
    if (ab == 0) return 1.0;
    return 1.0 - ((double)ab / (double)(aa + bb - ab));
}

static inline int bit_nbytes(int dim) { return (dim + 7) / 8; }

/* ---- sparsevec (src/types/sparsevec.rs) -------------------------------------------------------------------------
 * A row is the fixed-size record the engine stores: { int32 nnz; int32 pad[3]; int32 index[cap]; float value[cap] } (+ padding to 16 bytes),
 * cap = min(dim, 1000) (an indexed sparsevec has at most 1000 non-zero elements, sparsevec.rs hnsw support), indices
 * ascending, unused slots zero.  The three loops below restate the reference's merge joins statement by statement:
 * the f32 accumulation order is theirs. */
#define ORC_SPARSE_MAX_NNZ 1000
static inline int sparse_cap(int dim) { return dim < ORC_SPARSE_MAX_NNZ ? dim : ORC_SPARSE_MAX_NNZ; }
static inline int sp_nnz(const void *r) { return *(const int32_t *)r; }
static inline const int32_t *sp_idx(const void *r) { return (const int32_t *)((const uint8_t *)r + 16); }
static inline const float *sp_val(const void *r, int cap) { return (const float *)((const uint8_t *)r + 16 + (size_t)cap * 4); }

static float sparse_l2sq(const void *a, const void *b, int cap)      /* sparse_l2_squared_distance, sparsevec.rs:873-918 */
{
    const int32_t *ai_ = sp_idx(a), *bi_ = sp_idx(b); const float *ax = sp_val(a, cap), *bx = sp_val(b, cap);
    const int an = sp_nnz(a), bn = sp_nnz(b);
    float distance = 0.0f; int bpos = 0;
    for (int i = 0; i < an; i++) {
        const int32_t ai = ai_[i]; int32_t bi = -1;
        for (int j = bpos; j < bn; j++) {
            bi = bi_[j];
            if (ai == bi) { float diff = ax[i] - bx[j]; distance += diff * diff; }
            else if (ai > bi) distance += bx[j] * bx[j];
            if (ai >= bi) bpos = j + 1;
            if (bi >= ai) break;
        }
        if (ai != bi) distance += ax[i] * ax[i];
    }
    for (int j = bpos; j < bn; j++) distance += bx[j] * bx[j];
    return distance;
}
static float sparse_ip(const void *a, const void *b, int cap)        /* sparse_inner_product, sparsevec.rs:921-950 */
{
    const int32_t *ai_ = sp_idx(a), *bi_ = sp_idx(b); const float *ax = sp_val(a, cap), *bx = sp_val(b, cap);
    const int an = sp_nnz(a), bn = sp_nnz(b);
    float distance = 0.0f; int bpos = 0;
    for (int i = 0; i < an; i++) {
        const int32_t ai = ai_[i];
        for (int j = bpos; j < bn; j++) {
            const int32_t bi = bi_[j];
            if (ai == bi) distance += ax[i] * bx[j];
            if (ai >= bi) bpos = j + 1;
            if (bi >= ai) break;
        }
    }
    return distance;
}
static float sparse_l1(const void *a, const void *b, int cap)        /* sparsevec_l1_distance, sparsevec.rs:1038-1088 */
{
    const int32_t *ai_ = sp_idx(a), *bi_ = sp_idx(b); const float *ax = sp_val(a, cap), *bx = sp_val(b, cap);
    const int an = sp_nnz(a), bn = sp_nnz(b);
    float distance = 0.0f; int bpos = 0;
    for (int i = 0; i < an; i++) {
        const int32_t ai = ai_[i]; int32_t bi = -1;
        for (int j = bpos; j < bn; j++) {
            bi = bi_[j];
            if (ai == bi) distance += fabsf(ax[i] - bx[j]);
            else if (ai > bi) distance += fabsf(bx[j]);
            if (ai >= bi) bpos = j + 1;
            if (bi >= ai) break;
        }
        if (ai != bi) distance += fabsf(ax[i]);
    }
    for (int j = bpos; j < bn; j++) distance += fabsf(bx[j]);
    return distance;
}

/*
 * The opclass support function 1 ("distance proc"), f64 result:
 *   vector_l2_squared_distance vector.rs:598-607, vector_negative_inner_product :624-633,
 *   l1_distance :652-659; halfvec.rs:765-817; bitvec.rs:144-167.
 */
ORC_API double orc_distance(int dtype, int metric, int dim, const void *a, const void *b, int order)
{
    if (dtype == ORC_SPARSE) {                       /* sparsevec_l2_squared_distance / _negative_inner_product / _l1_distance, sparsevec.rs:970-1003, 1038 */
        const int cap = sparse_cap(dim);
        switch (metric) {
        case ORC_L2SQ:   return (double)sparse_l2sq(a, b, cap);
        case ORC_NEG_IP: return -(double)sparse_ip(a, b, cap);
        case ORC_L1:     return (double)sparse_l1(a, b, cap);
        }
        return NAN;
    }
    switch (metric) {
    case ORC_L2SQ:   return (double)acc_f(T_L2, dtype, dim, a, b, order);
    case ORC_NEG_IP: return -(double)acc_f(T_IP, dtype, dim, a, b, order);
    case ORC_L1:     return (double)acc_f(T_L1, dtype, dim, a, b, order);
    case ORC_HAMMING:return (double)orc_hamming(a, b, bit_nbytes(dim));
    case ORC_JACCARD:return orc_jaccard(a, b, bit_nbytes(dim));
    }
    return NAN;
}

/* The SQL-level helpers that exist only for the known-answer tests. */
ORC_API double orc_l2_distance(int dtype, int dim, const void *a, const void *b)   /* vector.rs:586-593; sparsevec.rs:953-966 */
{ return dtype == ORC_SPARSE ? sqrt((double)sparse_l2sq(a, b, sparse_cap(dim))) : sqrt((double)acc_seq(T_L2, dtype, dim, a, b)); }
ORC_API double orc_inner_product(int dtype, int dim, const void *a, const void *b) /* vector.rs:612-619; sparsevec.rs:980-991 */
{ return dtype == ORC_SPARSE ? (double)sparse_ip(a, b, sparse_cap(dim)) : (double)acc_seq(T_IP, dtype, dim, a, b); }
ORC_API double orc_cosine_distance(int dtype, int dim, const void *a, const void *b)
{   /* vector.rs:541-556 + :638-647 */
    if (dtype == ORC_SPARSE) {                       /* sparsevec_cosine_distance, sparsevec.rs:1005-1036: f32 norms over the stored values */
        const int cap = sparse_cap(dim);
        const float *ax = sp_val(a, cap), *bx = sp_val(b, cap);
        double sim = (double)sparse_ip(a, b, cap);
        float na = 0.0f, nb = 0.0f;
        for (int i = 0; i < sp_nnz(a); i++) na += ax[i] * ax[i];
        for (int i = 0; i < sp_nnz(b); i++) nb += bx[i] * bx[i];
        sim /= sqrt((double)na * (double)nb);
        if (sim < -1.0) sim = -1.0; if (sim > 1.0) sim = 1.0;
        return 1.0 - sim;
    }
    float s = 0.0f, na = 0.0f, nb = 0.0f;
    for (int i = 0; i < dim; i++) {
        float ai = elem_f32(a, dtype, i), bi = elem_f32(b, dtype, i);
        s += ai * bi; na += ai * ai; nb += bi * bi;
    }
    double sim = (double)s / sqrt((double)na * (double)nb);
    if (sim < -1.0) sim = -1.0; if (sim > 1.0) sim = 1.0;   /* f64::clamp; NaN stays NaN */
    return 1.0 - sim;
}

/* vector_norm vector.rs:672-683 (f64 accumulator) */
ORC_API double orc_norm(int dtype, int dim, const void *a)
{
    if (dtype == ORC_SPARSE) {                       /* sparsevec_l2_norm: f64 accumulator over the stored values */
        const float *x = sp_val(a, sparse_cap(dim)); double n = 0.0;
        for (int i = 0; i < sp_nnz(a); i++) n += (double)x[i] * (double)x[i];
        return sqrt(n);
    }
    double n = 0.0;
    for (int i = 0; i < dim; i++) { double v = (double)elem_f32(a, dtype, i); n += v * v; }
    return sqrt(n);
}

/* l2_normalize_raw vector.rs:106-126 / halfvec.rs:204-233.  Returns the norm; out is zero-filled when norm==0. */
ORC_API double orc_l2_normalize(int dtype, int dim, const void *in, void *out)
{
    if (dtype == ORC_SPARSE) {                       /* sparsevec_l2_normalize_raw, sparsevec.rs:1123-1178: f64 norm, zeros dropped */
        const int cap = sparse_cap(dim), nnz = sp_nnz(in);
        const float *x = sp_val(in, cap); const int32_t *xi = sp_idx(in);
        double norm = 0.0;
        for (int i = 0; i < nnz; i++) { double v = (double)x[i]; norm += v * v; }
        norm = sqrt(norm);
        memset(out, 0, (16 + (size_t)cap * 8 + 15) & ~(size_t)15);
        int32_t *oi = (int32_t *)((uint8_t *)out + 16); float *ov = (float *)((uint8_t *)out + 16 + (size_t)cap * 4);
        int k = 0;
        if (norm > 0.0) for (int i = 0; i < nnz; i++) { float v = (float)((double)x[i] / norm); if (v != 0.0f) { oi[k] = xi[i]; ov[k] = v; k++; } }
        *(int32_t *)out = k;
        return norm;
    }
    double norm = orc_norm(dtype, dim, in);
    if (dtype == ORC_F32) {
        float *o = out; const float *x = in;
        for (int i = 0; i < dim; i++) o[i] = norm > 0.0 ? (float)((double)x[i] / norm) : 0.0f;
    } else {
        uint16_t *o = out; const uint16_t *x = in;
        for (int i = 0; i < dim; i++)
            o[i] = norm > 0.0 ? orc_f32_to_half((float)((double)orc_half_to_f32(x[i]) / norm)) : 0;
    }
    return norm;
}

/* ------------------------------------------------------------------ */
/* Rust std BinaryHeap restated (max-heap on a caller-chosen ordering) */
/* "tie-order parity unpinned" -- see header.                          */
/* ------------------------------------------------------------------ */
typedef struct { double dist; int idx; } hitem;
typedef struct { hitem *d; int len, cap; int nearest; } heap_t;

/* Ord::cmp of NearestCandidate / FurthestCandidate: graph/mod.rs:103-112,131-139; scan.rs:86-116.
 * partial_cmp(..).unwrap_or(Equal): NaN compares Equal. */
static inline int hcmp(const heap_t *h, const hitem *a, const hitem *b)
{
    double x = h->nearest ? b->dist : a->dist, y = h->nearest ? a->dist : b->dist;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static inline int hle(const heap_t *h, const hitem *a, const hitem *b) { return hcmp(h, a, b) <= 0; }

static void heap_init(heap_t *h, int nearest) { h->d = NULL; h->len = h->cap = 0; h->nearest = nearest; }
static void heap_free(heap_t *h) { free(h->d); h->d = NULL; h->len = h->cap = 0; }

static int heap_sift_up(heap_t *h, int start, int pos)
{
    hitem e = h->d[pos];
    while (pos > start) {
        int parent = (pos - 1) / 2;
        if (hle(h, &e, &h->d[parent])) break;
        h->d[pos] = h->d[parent]; pos = parent;
    }
    h->d[pos] = e;
    return pos;
}
static void heap_push(heap_t *h, hitem it)
{
    if (h->len == h->cap) { h->cap = h->cap ? h->cap * 2 : 64; h->d = realloc(h->d, (size_t)h->cap * sizeof(hitem)); }
    h->d[h->len] = it; h->len++;
    heap_sift_up(h, 0, h->len - 1);
}
static void heap_sift_down_to_bottom(heap_t *h, int pos)
{
    int end = h->len, start = pos;
    hitem e = h->d[pos];
    int child = 2 * pos + 1;
    int lim = end >= 2 ? end - 2 : 0;               /* end.saturating_sub(2) */
    while (child <= lim && end >= 2) {
        if (hle(h, &h->d[child], &h->d[child + 1])) child += 1;
        h->d[pos] = h->d[child]; pos = child; child = 2 * pos + 1;
    }
    if (child == end - 1) { h->d[pos] = h->d[child]; pos = child; }
    h->d[pos] = e;
    heap_sift_up(h, start, pos);
}
static int heap_pop(heap_t *h, hitem *out)
{
    if (h->len == 0) return 0;
    hitem item = h->d[--h->len];
    if (h->len > 0) { hitem t = h->d[0]; h->d[0] = item; item = t; heap_sift_down_to_bottom(h, 0); }
    *out = item;
    return 1;
}

/* stable merge sort of hitem by distance (Rust slice::sort_by is stable);
 * desc=0: a.partial_cmp(b) ascending (graph/mod.rs:249-253, :478-482)
 * desc=1: b.partial_cmp(a)           (scan.rs:441-446) */
static void stable_sort(hitem *v, int n, int desc)
{
    if (n < 2) return;
    hitem *tmp = malloc((size_t)n * sizeof(hitem));
    for (int w = 1; w < n; w *= 2) {
        for (int lo = 0; lo < n; lo += 2 * w) {
            int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                /* take right only if right is strictly "less" than left */
                int right_less = desc ? (v[j].dist > v[i].dist) : (v[j].dist < v[i].dist);
                tmp[k++] = right_less ? v[j++] : v[i++];
            }
            while (i < mid) tmp[k++] = v[i++];
            while (j < hi) tmp[k++] = v[j++];
        }
        memcpy(v, tmp, (size_t)n * sizeof(hitem));
    }
    free(tmp);
}

/* ------------------------------------------------------------------ */
/* in-memory graph (graph/mod.rs:12-84) + build state (build.rs:238-285) */
/* ------------------------------------------------------------------ */
typedef struct { float distance; int idx; } cand_t;          /* graph/mod.rs:16-22 */
typedef struct {
    int level;
    int *ncnt;         /* [level+1] */
    cand_t **nbr;      /* [level+1][lm] */
    int64_t tids[HNSW_HEAPTIDS]; int ntids;
    int merged;        /* batch schedule only: row tombstoned as a duplicate of another element */
    int deleted;       /* HnswElementTupleData.deleted (types/hnsw.rs:112-137): set by vacuum's mark_deleted, tested by the on-disk paths */
} elem_t;

typedef struct orc_index {
    int dtype, metric, dim, m, efc, order, max_level;
    size_t row_bytes;
    uint8_t *values; elem_t *el; int n, cap;
    int entry;                       /* -1 = none (build.rs:255 Option<usize>) */
    int *vis; int vis_epoch, vis_cap;/* visited "set" */
    uint64_t cnt[8];                 /* distance evaluations: [0] entry, [1] search loop, [2] select in find, [3] back-link prune, [4] scan */
    double ind_tuples;
    int ondisk_tombstones;            /* orc_index_insert_on_disk keeps a tombstone for a merged row (the reference adds nothing) */
} orc_index;

static inline int layer_m(int m, int layer) { return layer == 0 ? 2 * m : m; }   /* hnsw_constants.rs:122-128 */

/* types/hnsw.rs:337-349 with BLCKSZ 8192, MAXALIGN'd page header 24, opaque 8, neighbour tuple header 4, ItemId 4, TID 6 */
ORC_API int orc_max_level(int m)
{
    int available = 8192 - 24 - 8 - 4 - 4;
    int v = available / 6 / m - 2;
    return v < 255 ? v : 255;
}
/* build.rs:373-377 given the uniform draw u */
ORC_API int orc_level_from_uniform(double u, int m)
{
    double r = u > DBL_MIN ? u : DBL_MIN;           /* .max(f64::MIN_POSITIVE) */
    double ml = 1.0 / log((double)m);               /* hnsw_constants.rs:132-134 */
    double lv = floor(-log(r) * ml);
    int mx = orc_max_level(m);
    return lv < (double)mx ? (int)lv : mx;
}

ORC_API size_t orc_row_bytes(int dtype, int dim)
{
    if (dtype == ORC_SPARSE) return (16 + (size_t)sparse_cap(dim) * 8 + 15) & ~(size_t)15;
    return dtype == ORC_F32 ? (size_t)dim * 4 : dtype == ORC_F16 ? (size_t)dim * 2 : (size_t)bit_nbytes(dim);
}

ORC_API orc_index *orc_index_new(int dtype, int metric, int dim, int m, int efc, int order)
{
    orc_index *x = calloc(1, sizeof *x);
    x->dtype = dtype; x->metric = metric; x->dim = dim; x->m = m; x->efc = efc; x->order = order;
    x->max_level = orc_max_level(m);
    x->row_bytes = orc_row_bytes(dtype, dim);
    x->entry = -1;
    return x;
}

static void elem_free(elem_t *e)
{
    if (e->nbr) for (int l = 0; l <= e->level; l++) free(e->nbr[l]);
    free(e->nbr); free(e->ncnt); e->nbr = NULL; e->ncnt = NULL;
}
ORC_API void orc_index_free(orc_index *x)
{
    if (!x) return;
    for (int i = 0; i < x->n; i++) elem_free(&x->el[i]);
    free(x->el); free(x->values); free(x->vis); free(x);
}

static inline const void *rowp(const orc_index *x, int i) { return x->values + (size_t)i * x->row_bytes; }

/* HnswBuildState::distance_fn build.rs:346-369: f64 result truncated to f32 */
static inline float dist_build(orc_index *x, const void *a, const void *b, int counter)
{
    x->cnt[counter]++;
    return (float)orc_distance(x->dtype, x->metric, x->dim, a, b, x->order);
}

static void vis_begin(orc_index *x)
{
    if (x->vis_cap < x->cap) { x->vis = realloc(x->vis, (size_t)x->cap * sizeof(int)); memset(x->vis + x->vis_cap, 0, (size_t)(x->cap - x->vis_cap) * sizeof(int)); x->vis_cap = x->cap; }
    x->vis_epoch++;
    if (x->vis_epoch == 0x7fffffff) { memset(x->vis, 0, (size_t)x->vis_cap * sizeof(int)); x->vis_epoch = 1; }
}
static inline int vis_test_set(orc_index *x, int i) { if (x->vis[i] == x->vis_epoch) return 1; x->vis[i] = x->vis_epoch; return 0; }

/* graph/mod.rs:161-255.  ep/out are (distance f32, idx); returns count, nearest first. */
static int search_layer(orc_index *x, const cand_t *ep, int nep, int ef, int layer, const void *q, cand_t *out)
{
    heap_t C, W; heap_init(&C, 1); heap_init(&W, 0);
    int result_len = 0;
    vis_begin(x);
    for (int i = 0; i < nep; i++) {
        vis_test_set(x, ep[i].idx);
        hitem it = { (double)ep[i].distance, ep[i].idx };
        heap_push(&C, it); heap_push(&W, it); result_len++;
    }
    hitem c;
    while (heap_pop(&C, &c)) {
        float f_dist = W.len ? (float)W.d[0].dist : FLT_MAX;
        if ((float)c.dist > f_dist) break;
        elem_t *ce = &x->el[c.idx];
        if (ce->level < layer) continue;
        int cn = ce->ncnt[layer];
        for (int k = 0; k < cn; k++) {
            int e = ce->nbr[layer][k].idx;
            if (vis_test_set(x, e)) continue;
            if (x->el[e].level < layer) continue;
            float ed = dist_build(x, q, rowp(x, e), 1);
            int always_add = result_len < ef;
            f_dist = W.len ? (float)W.d[0].dist : FLT_MAX;
            if (ed < f_dist || always_add) {
                hitem it = { (double)ed, e };
                heap_push(&C, it); heap_push(&W, it); result_len++;
                if (result_len > ef) { hitem drop; heap_pop(&W, &drop); result_len--; }
            }
        }
    }
    /* results.into_iter() = internal array order, then stable sort ascending */
    stable_sort(W.d, W.len, 0);
    int n = W.len;
    for (int i = 0; i < n; i++) { out[i].distance = (float)W.d[i].dist; out[i].idx = W.d[i].idx; }
    heap_free(&C); heap_free(&W);
    return n;
}

/* graph/mod.rs:315-339 */
static int check_element_closer(orc_index *x, const cand_t *e, const cand_t *r, int nr, int counter)
{
    for (int i = 0; i < nr; i++) {
        float d = dist_build(x, rowp(x, e->idx), rowp(x, r[i].idx), counter);
        if (d <= e->distance) return 0;
    }
    return 1;
}

/* graph/mod.rs:269-308; returns count written to out (capacity >= max(ncand, maxn)) */
static int select_neighbors(orc_index *x, const cand_t *cand, int ncand, int maxn, cand_t *out, int counter)
{
    if (ncand <= maxn) { memcpy(out, cand, (size_t)ncand * sizeof(cand_t)); return ncand; }
    cand_t *disc = malloc((size_t)ncand * sizeof(cand_t)); int nd = 0, nr = 0;
    for (int i = 0; i < ncand; i++) {
        if (nr >= maxn) break;
        if (check_element_closer(x, &cand[i], out, nr, counter)) out[nr++] = cand[i];
        else disc[nd++] = cand[i];
    }
    for (int i = 0; i < nd; i++) { if (nr >= maxn) break; out[nr++] = disc[i]; }
    free(disc);
    return nr;
}

/* graph/mod.rs:355-427 */
static void find_element_neighbors(orc_index *x, int new_idx, int entry_idx)
{
    elem_t *ne = &x->el[new_idx];
    const void *q = rowp(x, new_idx);
    int new_level = ne->level, entry_level = x->el[entry_idx].level;
    int cap = x->efc > 1 ? x->efc : 1;
    cand_t *ep = malloc((size_t)(cap + 1) * sizeof(cand_t)), *w = malloc((size_t)(cap + 1) * sizeof(cand_t));
    cand_t *sel = malloc((size_t)(cap + 2 * x->m + 1) * sizeof(cand_t));
    int nep = 1;
    ep[0].distance = dist_build(x, q, rowp(x, entry_idx), 0); ep[0].idx = entry_idx;

    for (int lc = entry_level; lc >= new_level + 1; lc--) {
        int nw = search_layer(x, ep, nep, 1, lc, q, w);
        if (nw > 0) { ep[0] = w[0]; nep = 1; }
    }
    int start = new_level < entry_level ? new_level : entry_level;
    for (int lc = start; lc >= 0; lc--) {
        int lm = layer_m(x->m, lc);
        int nw = search_layer(x, ep, nep, x->efc, lc, q, w);
        int ns = select_neighbors(x, w, nw, lm, sel, 2);
        memcpy(ne->nbr[lc], sel, (size_t)ns * sizeof(cand_t)); ne->ncnt[lc] = ns;
        memcpy(ep, w, (size_t)nw * sizeof(cand_t)); nep = nw;          /* mod.rs:425: whole W */
    }
    free(ep); free(w); free(sel);
}

/* graph/mod.rs:442-489 */
static void update_neighbor_connections(orc_index *x, int new_idx)
{
    elem_t *ne = &x->el[new_idx];
    for (int lc = ne->level; lc >= 0; lc--) {
        int lm = layer_m(x->m, lc);
        int ns = ne->ncnt[lc];
        cand_t *snap = malloc((size_t)(ns + 1) * sizeof(cand_t));
        memcpy(snap, ne->nbr[lc], (size_t)ns * sizeof(cand_t));
        for (int k = 0; k < ns; k++) {
            elem_t *nb = &x->el[snap[k].idx];
            cand_t nc = { snap[k].distance, new_idx };
            if (nb->ncnt[lc] < lm) { nb->nbr[lc][nb->ncnt[lc]++] = nc; continue; }
            int tot = nb->ncnt[lc] + 1;
            hitem *all = malloc((size_t)tot * sizeof(hitem));
            for (int i = 0; i < tot - 1; i++) { all[i].dist = (double)nb->nbr[lc][i].distance; all[i].idx = nb->nbr[lc][i].idx; }
            all[tot - 1].dist = (double)nc.distance; all[tot - 1].idx = nc.idx;
            stable_sort(all, tot, 0);
            cand_t *allc = malloc((size_t)tot * sizeof(cand_t)), *sel = malloc((size_t)tot * sizeof(cand_t));
            for (int i = 0; i < tot; i++) { allc[i].distance = (float)all[i].dist; allc[i].idx = all[i].idx; }
            int nsel = select_neighbors(x, allc, tot, lm, sel, 3);
            memcpy(nb->nbr[lc], sel, (size_t)nsel * sizeof(cand_t)); nb->ncnt[lc] = nsel;
            free(all); free(allc); free(sel);
        }
        free(snap);
    }
}

static int push_element(orc_index *x, const void *row, int level)
{
    if (x->n == x->cap) {
        x->cap = x->cap ? x->cap * 2 : 1024;
        x->el = realloc(x->el, (size_t)x->cap * sizeof(elem_t));
        x->values = realloc(x->values, (size_t)x->cap * x->row_bytes);
    }
    int idx = x->n++;
    memcpy(x->values + (size_t)idx * x->row_bytes, row, x->row_bytes);
    elem_t *e = &x->el[idx];
    memset(e, 0, sizeof *e);
    e->level = level;
    e->ncnt = calloc((size_t)level + 1, sizeof(int));
    e->nbr = calloc((size_t)level + 1, sizeof(cand_t *));
    for (int l = 0; l <= level; l++) e->nbr[l] = calloc((size_t)layer_m(x->m, l), sizeof(cand_t)); /* mod.rs:71-83 */
    return idx;
}

/* duplicate scan of build.rs:482-512; returns the element the row merges into, or -1 */
static int find_duplicate(orc_index *x, int new_idx)
{
    elem_t *ne = &x->el[new_idx];
    for (int k = 0; k < ne->ncnt[0]; k++) {
        if (ne->nbr[0][k].distance != 0.0f) break;
        int d = ne->nbr[0][k].idx;
        if (memcmp(rowp(x, new_idx), rowp(x, d), x->row_bytes) == 0 && x->el[d].ntids < HNSW_HEAPTIDS) return d;
    }
    return -1;
}

/*
 * build_callback build.rs:400-535 for one row (the reference's sequential schedule).
 * `row` must already be normalised for cosine opclasses (orc_l2_normalize; zero-norm rows are
 * skipped by the caller, build.rs:433-435).  Returns the element index the tid landed in.
 */
ORC_API int orc_index_insert(orc_index *x, const void *row, int level, int64_t tid)
{
    if (level > x->max_level) level = x->max_level;
    int new_idx = push_element(x, row, level);
    if (x->entry >= 0) {
        int entry_idx = x->entry;
        find_element_neighbors(x, new_idx, entry_idx);
        int dup = find_duplicate(x, new_idx);
        if (dup >= 0) {
            x->el[dup].tids[x->el[dup].ntids++] = tid;
            elem_free(&x->el[new_idx]); x->n--;          /* elements.pop(); values.truncate() */
            x->ind_tuples += 1.0;
            return dup;
        }
        update_neighbor_connections(x, new_idx);
        if (x->el[new_idx].level > x->el[entry_idx].level) x->entry = new_idx;
    } else {
        x->entry = new_idx;
    }
    x->el[new_idx].tids[0] = tid; x->el[new_idx].ntids = 1;
    x->ind_tuples += 1.0;
    return new_idx;
}

/*
 * Snapshot ("lock-step") schedule used by the batched device build: every row of the batch runs
 * find_element_neighbors against the graph as it stood when the batch began (rows of one batch do
 * not see each other, and share the entry point of the batch start); then, in row order, the
 * duplicate merge / back-links / entry-point update of build.rs:482-525 are applied.  A row merged
 * as a duplicate stays in the arena as a tombstone (merged=1, no links) because later rows of the
 * batch already hold their indices.  n==1 is exactly orc_index_insert apart from the tombstone.
 * Identical rows INSIDE one batch cannot find each other through the graph (they are not linked yet), so the
 * duplicate test of build.rs:482-512 is completed for them in the order the sequential schedule would meet them:
 * first the zero-distance layer-0 neighbours (find_duplicate), then the earliest byte-identical member of this
 * batch that is still an element of its own and has room for another heap TID (HNSW_HEAPTIDS).  20 identical rows
 * in one batch therefore end as 2 elements of 10 TIDs, as they do one row at a time (tests/t/015).
 * out_idx[i] = element that holds tid i.
 */
static uint64_t row_hash(const uint8_t *p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;                     /* FNV-1a: only used to bucket candidates, equality is memcmp */
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}
ORC_API void orc_index_insert_batch(orc_index *x, const void *rows, const int *levels, const int64_t *tids, int n, int *out_idx)
{
    const uint8_t *r = rows;
    int i0 = 0;
    if (x->entry < 0 && n > 0) { int e = orc_index_insert(x, r, levels[0], tids[0]); if (out_idx) out_idx[0] = e; i0 = 1; }
    int first = x->n, entry_idx = x->entry;
    for (int i = i0; i < n; i++) {
        int lv = levels[i] > x->max_level ? x->max_level : levels[i];
        push_element(x, r + (size_t)i * x->row_bytes, lv);
    }
    for (int i = i0; i < n; i++) find_element_neighbors(x, first + (i - i0), entry_idx);
    /* members bucketed by payload hash, each bucket chained in row order */
    int nb = n - i0, tsize = 16; while (tsize < 2 * nb) tsize <<= 1;
    int *head = malloc((size_t)tsize * sizeof(int)), *tail = malloc((size_t)tsize * sizeof(int)), *next = malloc((size_t)(nb + 1) * sizeof(int));
    int *bucket = malloc((size_t)(nb + 1) * sizeof(int));
    uint64_t *hk = malloc((size_t)tsize * sizeof(uint64_t));
    for (int t = 0; t < tsize; t++) head[t] = tail[t] = -1;
    for (int k = 0; k < nb; k++) {
        uint64_t h = row_hash(rowp(x, first + k), x->row_bytes);
        int t = (int)(h & (uint64_t)(tsize - 1));
        while (head[t] >= 0 && hk[t] != h) t = (t + 1) & (tsize - 1);
        if (head[t] < 0) { head[t] = k; hk[t] = h; } else next[tail[t]] = k;
        tail[t] = k; next[k] = -1; bucket[k] = t;
    }
    for (int i = i0; i < n; i++) {
        int new_idx = first + (i - i0);
        int dup = find_duplicate(x, new_idx);
        if (dup >= 0 && x->el[dup].merged) dup = -1;
        if (dup < 0) {
            int t = bucket[i - i0];
            /* members that are merged or full never become eligible again: the bucket's head moves past them */
            while (head[t] >= 0 && head[t] < i - i0 && (x->el[first + head[t]].merged || x->el[first + head[t]].ntids >= HNSW_HEAPTIDS)) head[t] = next[head[t]];
            for (int k = head[t]; k >= 0 && k < i - i0; k = next[k]) {
                elem_t *c = &x->el[first + k];
                if (c->merged || c->ntids >= HNSW_HEAPTIDS) continue;
                if (memcmp(rowp(x, new_idx), rowp(x, first + k), x->row_bytes) == 0) { dup = first + k; break; }
            }
        }
        if (dup >= 0) {
            x->el[dup].tids[x->el[dup].ntids++] = tids[i];
            elem_t *e = &x->el[new_idx];
            for (int l = 0; l <= e->level; l++) e->ncnt[l] = 0;
            e->merged = 1; e->ntids = 0;
            x->ind_tuples += 1.0;
            if (out_idx) out_idx[i] = dup;
            continue;
        }
        update_neighbor_connections(x, new_idx);
        if (x->el[new_idx].level > x->el[x->entry].level) x->entry = new_idx;
        x->el[new_idx].tids[0] = tids[i]; x->el[new_idx].ntids = 1;
        x->ind_tuples += 1.0;
        if (out_idx) out_idx[i] = new_idx;
    }
    free(head); free(tail); free(next); free(bucket); free(hk);
}

/* ---- accessors used by tests ---- */
ORC_API int orc_index_size(const orc_index *x) { return x->n; }
ORC_API int orc_index_entry(const orc_index *x) { return x->entry; }
ORC_API int orc_index_level(const orc_index *x, int i) { return x->el[i].level; }
ORC_API int orc_index_merged(const orc_index *x, int i) { return x->el[i].merged; }
ORC_API int orc_index_ntids(const orc_index *x, int i) { return x->el[i].ntids; }
ORC_API int64_t orc_index_tid(const orc_index *x, int i, int k) { return x->el[i].tids[k]; }
ORC_API uint64_t orc_index_counter(const orc_index *x, int k) { return x->cnt[k]; }
ORC_API void orc_index_reset_counters(orc_index *x) { memset(x->cnt, 0, sizeof x->cnt); }
ORC_API int orc_index_neighbors(const orc_index *x, int i, int layer, int *ids, float *dist)
{
    if (layer > x->el[i].level) return -1;
    int n = x->el[i].ncnt[layer];
    for (int k = 0; k < n; k++) { if (ids) ids[k] = x->el[i].nbr[layer][k].idx; if (dist) dist[k] = x->el[i].nbr[layer][k].distance; }
    return n;
}
/* bulk load of a finished graph (bench.py's cpu_baseline leg times the scalar scan on the same graph the
 * device built): rows, levels (negative = tombstone), entry; then one call per layer with [n][lm] lists. */
ORC_API void orc_index_load(orc_index *x, const void *rows, int n, const int *levels, int entry)
{
    const uint8_t *r = rows;
    for (int i = 0; i < n; i++) {
        int lv = levels[i] < 0 ? -1 - levels[i] : levels[i];
        int idx = push_element(x, r + (size_t)i * x->row_bytes, lv);
        if (levels[i] < 0) x->el[idx].merged = 1; else { x->el[idx].ntids = 1; x->el[idx].tids[0] = i; }
    }
    x->entry = entry;
}
ORC_API void orc_index_set_layer(orc_index *x, int layer, const uint32_t *ids, const float *dist, const uint16_t *cnt)
{
    int lm = layer_m(x->m, layer);
    for (int i = 0; i < x->n; i++) {
        if (x->el[i].level < layer || x->el[i].merged) continue;
        x->el[i].ncnt[layer] = cnt[i];
        for (int k = 0; k < cnt[i]; k++) { x->el[i].nbr[layer][k].idx = (int)ids[(size_t)i * lm + k]; x->el[i].nbr[layer][k].distance = dist ? dist[(size_t)i * lm + k] : 0.0f; }
    }
}
/* test hook mirroring the hand-built graphs of graph/mod.rs:537-584 */
ORC_API int orc_index_add_raw(orc_index *x, const void *row, int level) { int i = push_element(x, row, level); if (x->entry < 0) x->entry = i; x->el[i].ntids = 1; x->el[i].tids[0] = i; return i; }
ORC_API void orc_index_link_raw(orc_index *x, int i, int layer, int j, float d)
{ elem_t *e = &x->el[i]; e->nbr[layer][e->ncnt[layer]].idx = j; e->nbr[layer][e->ncnt[layer]].distance = d; e->ncnt[layer]++; }
ORC_API int orc_search_layer_raw(orc_index *x, const void *q, const int *ep_idx, int nep, int ef, int layer, int *ids, float *dist)
{
    cand_t *ep = malloc((size_t)nep * sizeof(cand_t)), *out = malloc((size_t)(ef + nep + 1) * sizeof(cand_t));
    /* stash the query as a temporary row so dist_build sees arena memory, as the reference's tests do (mod.rs:564) */
    for (int i = 0; i < nep; i++) { ep[i].idx = ep_idx[i]; ep[i].distance = dist_build(x, q, rowp(x, ep_idx[i]), 0); }
    int n = search_layer(x, ep, nep, ef, layer, q, out);
    for (int i = 0; i < n; i++) { ids[i] = out[i].idx; dist[i] = out[i].distance; }
    free(ep); free(out);
    return n;
}
ORC_API int orc_select_neighbors_raw(orc_index *x, const int *cand_idx, const float *cand_dist, int n, int maxn, int *ids)
{
    cand_t *c = malloc((size_t)n * sizeof(cand_t)), *o = malloc((size_t)(n + maxn) * sizeof(cand_t));
    for (int i = 0; i < n; i++) { c[i].idx = cand_idx[i]; c[i].distance = cand_dist[i]; }
    int k = select_neighbors(x, c, n, maxn, o, 2);
    for (int i = 0; i < k; i++) ids[i] = o[i].idx;
    free(c); free(o);
    return k;
}
ORC_API void orc_find_element_neighbors_raw(orc_index *x, int new_idx, int entry_idx) { find_element_neighbors(x, new_idx, entry_idx); }
ORC_API void orc_update_neighbor_connections_raw(orc_index *x, int new_idx) { update_neighbor_connections(x, new_idx); }

/* ------------------------------------------------------------------ */
/* scan: in-memory mirror of src/index/scan.rs (rows addressed by element
 * index instead of (blkno, offno); no deleted tuples, no stale versions) */
/* ------------------------------------------------------------------ */
static __thread int tl_nocount = 0;   /* set by orc_search_many's workers: the shared counters are not atomic */
typedef struct orc_scan {
    orc_index *x; void *q; int q_null;
    int ef_search, iterative; int64_t max_scan_tuples;
    hitem *results; int nres, rescap;        /* sorted, nearest LAST (scan.rs:441-446) */
    heap_t discarded; uint8_t *visited;      /* iterative-scan state (scan.rs:597-612) */
    int first, iter_init; int64_t tuples; double previous_distance;
    int cur; int cur_tid_left; double cur_dist;
} orc_scan;

static inline double dist_scan(orc_scan *s, int e)
{   /* load_element scan.rs:186-192: NULL query => 0.0 */
    if (s->q_null) return 0.0;
    if (!tl_nocount) s->x->cnt[4]++;
    return orc_distance(s->x->dtype, s->x->metric, s->x->dim, s->q, rowp(s->x, e), s->x->order);
}

/* search_layer_disk scan.rs:302-448.  visited: caller bitmap or NULL (=> local); discarded may be NULL.
 * Entry points carry their distances.  Output nearest-last in *out (malloc'd), returns count. */
static int search_layer_scan_skip(orc_scan *s, const hitem *ep, int nep, int ef, int layer,
                                  uint8_t *visited, heap_t *discarded, int add_entry_to_visited, const uint8_t *skip, hitem **out)
{
    orc_index *x = s->x;
    uint8_t *local = NULL;
    if (!visited) { local = calloc((size_t)x->n + 1, 1); visited = local; }
    heap_t C, W; heap_init(&C, 1); heap_init(&W, 0);
    int w_len = 0;
    for (int i = 0; i < nep; i++) {
        if (add_entry_to_visited) visited[ep[i].idx] = 1;
        heap_push(&C, ep[i]); heap_push(&W, ep[i]);
        if (!skip || !skip[ep[i].idx]) w_len++;                 /* skip_count, scan.rs:331-336: vacuum's skip set is traversed but not counted */
    }
    hitem c;
    while (heap_pop(&C, &c)) {
        double f_dist = W.len ? W.d[0].dist : DBL_MAX;
        if (c.dist > f_dist) { if (discarded) heap_push(discarded, c); break; }
        elem_t *ce = &x->el[c.idx];
        if (ce->level < layer) continue;            /* load_neighbor_tids would read beyond the tuple; cannot happen for valid graphs */
        int cn = ce->ncnt[layer];
        for (int k = 0; k < cn; k++) {
            int e = ce->nbr[layer][k].idx;
            if (visited[e]) continue;
            visited[e] = 1;
            int always_add = w_len < ef;
            f_dist = W.len ? W.d[0].dist : DBL_MAX;
            if (x->el[e].deleted) continue;               /* load_element: deleted tuple -> None (scan.rs:178-181) */
            double d = dist_scan(s, e);
            if (!always_add && d >= f_dist) {       /* load_element returned None (scan.rs:195-200) */
                if (discarded) {                    /* second load, scan.rs:385-404 (costs a 2nd distance) */
                    double d2 = dist_scan(s, e);
                    if (x->el[e].level >= layer) { hitem it = { d2, e }; heap_push(discarded, it); }
                }
                continue;
            }
            if (x->el[e].level < layer) continue;
            hitem it = { d, e };
            heap_push(&C, it); heap_push(&W, it);
            if (!skip || !skip[e]) w_len++;                         /* scan.rs:416-419 */
            if (w_len > ef) { hitem ev; heap_pop(&W, &ev); w_len--; if (discarded) heap_push(discarded, ev); }
        }
    }
    if (discarded) { hitem r; while (heap_pop(&C, &r)) heap_push(discarded, r); }
    stable_sort(W.d, W.len, 1);
    int n = W.len;
    *out = malloc((size_t)(n + 1) * sizeof(hitem));
    memcpy(*out, W.d, (size_t)n * sizeof(hitem));
    heap_free(&C); heap_free(&W); free(local);
    return n;
}

static int search_layer_scan(orc_scan *s, const hitem *ep, int nep, int ef, int layer,
                             uint8_t *visited, heap_t *discarded, int add_entry_to_visited, hitem **out)
{
    return search_layer_scan_skip(s, ep, nep, ef, layer, visited, discarded, add_entry_to_visited, NULL, out);
}

/* get_scan_items scan.rs:458-530 */
static void get_scan_items(orc_scan *s, uint8_t *visited, heap_t *discarded)
{
    orc_index *x = s->x;
    s->nres = 0;
    if (x->entry < 0) return;
    hitem ep = { dist_scan(s, x->entry), x->entry };
    int ep_level = x->el[x->entry].level;
    for (int lc = ep_level; lc >= 1; lc--) {
        hitem *w; int nw = search_layer_scan(s, &ep, 1, 1, lc, NULL, NULL, 1, &w);
        if (nw == 0) { free(w); return; }
        ep = w[nw - 1]; free(w);
    }
    hitem *w; int nw = search_layer_scan(s, &ep, 1, s->ef_search, 0, visited, discarded, 1, &w);
    free(s->results); s->results = w; s->nres = nw; s->rescap = nw + 1;
}

/* resume_scan_items scan.rs:538-577 */
static void resume_scan_items(orc_scan *s)
{
    s->nres = 0;
    if (s->discarded.len == 0) return;
    int bs = s->ef_search, nep = 0;
    hitem *ep = malloc((size_t)bs * sizeof(hitem));
    while (nep < bs && s->discarded.len > 0) heap_pop(&s->discarded, &ep[nep++]);
    hitem *w; int nw = search_layer_scan(s, ep, nep, bs, 0, s->visited, &s->discarded, 0, &w);
    free(ep); free(s->results); s->results = w; s->nres = nw; s->rescap = nw + 1;
}

ORC_API orc_scan *orc_scan_begin(orc_index *x, const void *query, int ef_search, int iterative, int64_t max_scan_tuples)
{
    orc_scan *s = calloc(1, sizeof *s);
    s->x = x; s->ef_search = ef_search; s->iterative = iterative; s->max_scan_tuples = max_scan_tuples;
    s->q_null = query == NULL;
    if (query) { s->q = malloc(x->row_bytes); memcpy(s->q, query, x->row_bytes); }
    heap_init(&s->discarded, 1);
    s->first = 1; s->cur = -1; s->previous_distance = -INFINITY;   /* scan.rs:661 f64::NEG_INFINITY */
    return s;
}
ORC_API void orc_scan_end(orc_scan *s)
{ if (!s) return; free(s->q); free(s->results); heap_free(&s->discarded); free(s->visited); free(s); }

/* amgettuple scan.rs:709-876.  Returns 1 and fills tid/dist/elem, or 0 when exhausted. */
ORC_API int orc_scan_next(orc_scan *s, int64_t *tid, double *dist, int *elem)
{
    orc_index *x = s->x;
    if (s->first) {
        int use_iter = s->iterative != ORC_ITER_OFF;
        if (use_iter) { s->visited = calloc((size_t)x->n + 1, 1); get_scan_items(s, s->visited, &s->discarded); }
        else get_scan_items(s, NULL, NULL);
        s->iter_init = use_iter;                     /* scan.rs:789 */
        s->first = 0;
    }
    for (;;) {
        if (s->cur >= 0) {
            if (s->cur_tid_left > 0) {
                int64_t t = x->el[s->cur].tids[--s->cur_tid_left];    /* heaptids.pop() */
                if (s->iterative == ORC_ITER_STRICT) {
                    if (s->cur_dist < s->previous_distance) continue;
                    s->previous_distance = s->cur_dist;
                }
                if (tid) *tid = t; if (dist) *dist = s->cur_dist; if (elem) *elem = s->cur;
                return 1;
            }
            s->cur = -1;
        }
        if (s->nres == 0) {
            if (s->iterative == ORC_ITER_OFF) return 0;
            if (!s->iter_init) return 0;
            if (s->tuples >= s->max_scan_tuples) {
                if (s->discarded.len == 0) return 0;
                hitem sc; heap_pop(&s->discarded, &sc);
                if (s->rescap < 1) { s->results = realloc(s->results, 4 * sizeof(hitem)); s->rescap = 4; }
                s->results[0] = sc; s->nres = 1;
            } else {
                resume_scan_items(s);
            }
            if (s->nres == 0) return 0;
        }
        hitem sc = s->results[--s->nres];
        if (x->el[sc.idx].ntids == 0) continue;
        s->tuples++;
        s->cur = sc.idx; s->cur_tid_left = x->el[sc.idx].ntids; s->cur_dist = sc.dist;
    }
}


/* ------------------------------------------------------------------ */
/* f3: the on-disk insert (aminsert, src/index/insert.rs) and vacuum's repair search (src/index/vacuum.rs:288-407)
 * on the in-memory mirror: elements addressed by index instead of (blkno, offno), every distance through the scan
 * path's f64 (scan.rs:190-191), no concurrency.                                                              */
/* ------------------------------------------------------------------ */
static inline double dist_elems(orc_index *x, int a, int b)
{   /* FunctionCall2Coll(dist_fmgr, collation, datum(a), datum(b)): insert.rs:606, compute_element_distance :745-781 */
    x->cnt[3]++;
    return orc_distance(x->dtype, x->metric, x->dim, rowp(x, a), rowp(x, b), x->order);
}

/* find_element_neighbors_on_disk insert.rs:1021-1123.  out[lc] (malloc'd, nearest first), out_n[lc] for lc = 0..new_level.
 * skip != NULL is vacuum's repair: ef + 1, skip-set members traversed but neither counted nor selected. */
static void find_element_neighbors_on_disk(orc_scan *s, int new_level, int entry, const uint8_t *skip, hitem **out, int *out_n)
{
    orc_index *x = s->x;
    for (int lc = 0; lc <= new_level; lc++) { out[lc] = NULL; out_n[lc] = 0; }
    if (x->el[entry].deleted) return;                                  /* load_element(entry) -> None, insert.rs:1037-1048 */
    int entry_level = x->el[entry].level, nep = 1;
    hitem *epl = malloc(sizeof(hitem));
    epl[0].dist = dist_scan(s, entry); epl[0].idx = entry;
    for (int lc = entry_level; lc >= new_level + 1; lc--) {            /* phase 1, insert.rs:1053-1074 */
        hitem *w; int nw = search_layer_scan_skip(s, epl, nep, 1, lc, NULL, NULL, 1, skip, &w);
        free(epl);
        if (nw == 0) { free(w); return; }
        epl = malloc(sizeof(hitem)); epl[0] = w[nw - 1]; nep = 1; free(w);   /* w.into_iter().last(): the nearest */
    }
    int start = new_level < entry_level ? new_level : entry_level;
    for (int lc = start; lc >= 0; lc--) {                              /* phase 2, insert.rs:1077-1120 */
        int lm = layer_m(x->m, lc);
        int ef = skip ? x->efc + 1 : x->efc;                           /* insert.rs:1081-1086 */
        hitem *w; int nw = search_layer_scan_skip(s, epl, nep, ef, lc, NULL, NULL, 1, skip, &w);
        out[lc] = malloc((size_t)(lm + 1) * sizeof(hitem));
        int k = 0;
        for (int i = nw - 1; i >= 0 && k < lm; i--) {                  /* filtered.iter().rev().take(lm): the lm nearest, NO heuristic */
            if (skip && skip[w[i].idx]) continue;
            out[lc][k++] = w[i];
        }
        out_n[lc] = k;
        free(epl); epl = w; nep = nw;                                  /* ep_list = w (whole W, nearest last) */
    }
    free(epl);
}

/* get_update_index insert.rs:500-739.  Returns -3 = None, -2 = a free slot exists, >= 0 = slot to overwrite. */
static int get_update_index(orc_index *x, int n, int layer, double new_distance)
{
    elem_t *ne = &x->el[n];
    if (ne->deleted) return -3;                                        /* insert.rs:524-527 */
    int lm = layer_m(x->m, layer), cnt = ne->ncnt[layer];
    if (cnt < lm) return -2;                                           /* insert.rs:556-559 */
    hitem *cand = malloc((size_t)(cnt + 1) * sizeof(hitem)); int nc = 0, pruned_deleted = -1;
    for (int i = 0; i < cnt; i++) {                                    /* insert.rs:566-619 */
        int c = ne->nbr[layer][i].idx;
        if (x->el[c].deleted || x->el[c].ntids == 0) { if (pruned_deleted < 0) pruned_deleted = i; continue; }
        cand[nc].dist = dist_elems(x, n, c); cand[nc].idx = c; nc++;
    }
    if (pruned_deleted >= 0) { free(cand); return pruned_deleted; }    /* insert.rs:622-625 */
    stable_sort(cand, nc, 0);                                          /* insert.rs:630-634 */
    cand[nc].dist = new_distance; cand[nc].idx = -1; nc++;             /* the new element, is_new */
    stable_sort(cand, nc, 0);                                          /* insert.rs:661-665: ties keep existing elements first */
    int *sel = malloc((size_t)nc * sizeof(int)), *pru = malloc((size_t)nc * sizeof(int)); int ns = 0, np = 0;
    for (int h = 0; h < nc; h++) {                                     /* insert.rs:673-704 */
        if (ns >= lm) break;
        int closer = 1;
        for (int k = 0; k < ns; k++) {
            const hitem *hc = &cand[h], *sc = &cand[sel[k]];
            if (hc->idx >= 0 && sc->idx >= 0) {                        /* pairs with the new element are not evaluated, insert.rs:680-693 */
                double d = dist_elems(x, hc->idx, sc->idx);
                if (d <= hc->dist) { closer = 0; break; }
            }
        }
        if (closer) sel[ns++] = h; else pru[np++] = h;
    }
    for (int k = 0; k < np && ns < lm; k++) sel[ns++] = pru[k];        /* insert.rs:707-712 */
    int new_selected = 0;
    for (int k = 0; k < ns; k++) if (cand[sel[k]].idx < 0) new_selected = 1;
    int replace = -3;
    if (new_selected) {                                                /* insert.rs:722-737: first existing neighbour, in list order, that was not selected */
        for (int i = 0; i < cnt && replace < 0; i++) {
            int c = ne->nbr[layer][i].idx, found = 0;
            for (int k = 0; k < ns; k++) if (cand[sel[k]].idx == c) { found = 1; break; }
            if (!found) replace = i;
        }
    }
    free(cand); free(sel); free(pru);
    return replace;
}

/* write_neighbor_update insert.rs:793-871 */
static void write_neighbor_update(orc_index *x, int n, int layer, int new_idx, double new_distance, int update_idx)
{
    elem_t *ne = &x->el[n];
    int lm = layer_m(x->m, layer), cnt = ne->ncnt[layer];
    for (int i = 0; i < cnt; i++) if (ne->nbr[layer][i].idx == new_idx) return;      /* connection already exists */
    cand_t nc = { (float)new_distance, new_idx };
    if (update_idx == -2) { if (cnt < lm) { ne->nbr[layer][cnt] = nc; ne->ncnt[layer] = cnt + 1; } }
    else if (update_idx >= 0 && update_idx < cnt) ne->nbr[layer][update_idx] = nc;
}

/* aminsert insert.rs:1227-1480 for one row.  Returns the element that holds the tid. */
ORC_API int orc_index_insert_on_disk(orc_index *x, const void *row, int level, int64_t tid)
{
    if (level > x->max_level) level = x->max_level;
    if (x->entry < 0) {
        int e = push_element(x, row, level);
        x->el[e].tids[0] = tid; x->el[e].ntids = 1; x->entry = e; x->ind_tuples += 1.0;
        return e;
    }
    orc_scan s; memset(&s, 0, sizeof s);
    s.x = x; s.q = malloc(x->row_bytes); memcpy(s.q, row, x->row_bytes);
    int entry = x->entry, entry_level = x->el[entry].level;
    hitem **nb = malloc((size_t)(level + 1) * sizeof(hitem *)); int *nn = malloc((size_t)(level + 1) * sizeof(int));
    find_element_neighbors_on_disk(&s, level, entry, NULL, nb, nn);
    int result = -1;
    for (int k = 0; k < nn[0]; k++) {                                  /* find_duplicate_on_disk insert.rs:1180-1214 */
        if (nb[0][k].dist != 0.0) break;
        int c = nb[0][k].idx;
        if (memcmp(row, rowp(x, c), x->row_bytes) == 0 && x->el[c].ntids > 0 && x->el[c].ntids < HNSW_HEAPTIDS) {   /* add_duplicate_on_disk :1136-1171 */
            x->el[c].tids[x->el[c].ntids++] = tid; result = c; break;
        }
    }
    if (result >= 0 && x->ondisk_tombstones) {          /* test convention shared with the engine: element index == row index, so a merged row leaves a tombstone */
        int e = push_element(x, row, level); x->el[e].merged = 1; x->el[e].ntids = 0;
    }
    if (result < 0) {
        int e = push_element(x, row, level);
        for (int lc = 0; lc <= level; lc++) {
            int lm = layer_m(x->m, lc), c = nn[lc] < lm ? nn[lc] : lm;
            for (int i = 0; i < c; i++) { x->el[e].nbr[lc][i].distance = (float)nb[lc][i].dist; x->el[e].nbr[lc][i].idx = nb[lc][i].idx; }
            x->el[e].ncnt[lc] = c;
        }
        x->el[e].tids[0] = tid; x->el[e].ntids = 1;
        for (int lc = level; lc >= 0; lc--) {                          /* update_neighbors_on_disk insert.rs:883-958 */
            int lm = layer_m(x->m, lc);
            for (int k = 0; k < nn[lc] && k < lm; k++) {
                int n = nb[lc][k].idx;
                if (x->el[n].deleted) continue;                        /* load_element -> None */
                int ui = get_update_index(x, n, lc, nb[lc][k].dist);
                if (ui == -3) continue;
                write_neighbor_update(x, n, lc, e, nb[lc][k].dist, ui);
            }
        }
        if (level > entry_level) x->entry = e;                         /* insert.rs:1453-1470 */
        result = e;
    }
    x->ind_tuples += 1.0;
    for (int lc = 0; lc <= level; lc++) free(nb[lc]);
    free(nb); free(nn); free(s.q);
    return result;
}

/* repair_graph_element vacuum.rs:288-407: new neighbours of element e found with the deleted set (+ e itself) skipped */
ORC_API void orc_index_repair_element(orc_index *x, int e, const uint8_t *deleted)
{
    if (e == x->entry) return;                                         /* vacuum.rs:300-303 */
    uint8_t *skip = malloc((size_t)x->n);
    for (int i = 0; i < x->n; i++) skip[i] = deleted ? deleted[i] : 0;
    skip[e] = 1;
    orc_scan s; memset(&s, 0, sizeof s);
    s.x = x; s.q = malloc(x->row_bytes); memcpy(s.q, rowp(x, e), x->row_bytes);
    int level = x->el[e].level;
    hitem **nb = malloc((size_t)(level + 1) * sizeof(hitem *)); int *nn = malloc((size_t)(level + 1) * sizeof(int));
    find_element_neighbors_on_disk(&s, level, x->entry, skip, nb, nn);
    for (int lc = 0; lc <= level; lc++) {                              /* vacuum.rs:358-373: the tuple is rebuilt from scratch */
        int lm = layer_m(x->m, lc), c = nn[lc] < lm ? nn[lc] : lm;
        for (int i = 0; i < c; i++) { x->el[e].nbr[lc][i].distance = (float)nb[lc][i].dist; x->el[e].nbr[lc][i].idx = nb[lc][i].idx; }
        x->el[e].ncnt[lc] = c;
        free(nb[lc]);
    }
    free(nb); free(nn); free(s.q); free(skip);
}
ORC_API void orc_index_mark_deleted(orc_index *x, int e, int flag) { x->el[e].deleted = flag; }
ORC_API void orc_index_clear_tids(orc_index *x, int e) { x->el[e].ntids = 0; }      /* remove_heap_tids vacuum.rs:118-217 removed every TID */

/* ---- vacuum (src/index/vacuum.rs): pass 1 remove_heap_tids :118-217, pass 2 repair_graph :411-644 (needs_updated :230-285,
 * repair_graph_entry_point :411-520, repair_graph_element :288-407), pass 3 mark_deleted :655-793 -- on the in-memory mirror.
 * dead: the heap TIDs the bulk-delete callback reports dead (any order). ---- */
static int cmp_i64(const void *a, const void *b) { int64_t x = *(const int64_t *)a, y = *(const int64_t *)b; return x < y ? -1 : x > y; }
static int needs_updated(orc_index *x, int e, const uint8_t *deleted)
{
    elem_t *el = &x->el[e];
    for (int lc = 0; lc <= el->level; lc++)
        for (int k = 0; k < el->ncnt[lc]; k++) if (deleted[el->nbr[lc][k].idx]) return 1;
    return el->ncnt[0] < layer_m(x->m, 0);                              /* "also update if layer 0 is not full", vacuum.rs:268-279 */
}
/* repair_graph_element with an explicit entry (the entry-point repair passes the highest point instead), vacuum.rs:288-407 */
static void repair_element_from(orc_index *x, int e, int entry, const uint8_t *deleted)
{
    if (e == entry || entry < 0) return;
    int saved = x->entry; x->entry = entry;
    orc_index_repair_element(x, e, deleted);
    x->entry = saved;
}
ORC_API void orc_index_vacuum(orc_index *x, const int64_t *dead, int n_dead)
{
    int64_t *ds = malloc((size_t)(n_dead + 1) * sizeof(int64_t));
    memcpy(ds, dead, (size_t)n_dead * sizeof(int64_t)); qsort(ds, (size_t)n_dead, sizeof(int64_t), cmp_i64);
    uint8_t *deleted = calloc((size_t)x->n + 1, 1);
    int highest = -1, highest_level = -1;
    for (int e = 0; e < x->n; e++) {                                     /* pass 1 */
        elem_t *el = &x->el[e];
        if (el->merged) continue;                                        /* batch-schedule tombstone: has no tuple */
        if (el->ntids > 0) {
            int k = 0;
            for (int i = 0; i < el->ntids; i++) if (!bsearch(&el->tids[i], ds, (size_t)n_dead, sizeof(int64_t), cmp_i64)) el->tids[k++] = el->tids[i];
            el->ntids = k;
        }
        if (el->ntids == 0) deleted[e] = 1;
        else if (el->level > highest_level && e != x->entry) { highest = e; highest_level = el->level; }
    }
    /* pass 2: entry point first (vacuum.rs:411-520) */
    if (highest >= 0 && !x->el[highest].deleted && needs_updated(x, highest, deleted)) repair_element_from(x, highest, x->entry, deleted);
    if (x->entry >= 0) {
        if (deleted[x->entry]) x->entry = highest;                       /* -1 when nothing is left */
        else if (!x->el[x->entry].deleted && needs_updated(x, x->entry, deleted)) repair_element_from(x, x->entry, highest >= 0 ? highest : x->entry, deleted);
    }
    for (int e = 0; e < x->n; e++) {                                     /* then every element that still holds a heap TID (vacuum.rs:540-640) */
        elem_t *el = &x->el[e];
        if (el->merged || el->ntids == 0 || el->deleted) continue;
        if (!needs_updated(x, e, deleted)) continue;
        if (x->entry < 0 || el->level > x->el[x->entry].level) {
            repair_element_from(x, e, x->entry, deleted);
            if (x->entry < 0 || el->level > x->el[x->entry].level) x->entry = e;
        } else repair_element_from(x, e, x->entry, deleted);
    }
    for (int e = 0; e < x->n; e++) {                                     /* pass 3 */
        elem_t *el = &x->el[e];
        if (el->merged || el->deleted || el->ntids > 0) continue;
        for (int lc = 0; lc <= el->level; lc++) el->ncnt[lc] = 0;
        el->deleted = 1;
    }
    free(ds); free(deleted);
}
ORC_API int orc_index_deleted(const orc_index *x, int e) { return x->el[e].deleted; }
ORC_API void orc_index_set_ondisk_tombstones(orc_index *x, int on) { x->ondisk_tombstones = on; }

/* convenience: non-iterative top-k of one query -> element ids + f64 distances; returns count */
ORC_API int orc_search_topk(orc_index *x, const void *query, int ef_search, int k, int *ids, double *dist)
{
    orc_scan *s = orc_scan_begin(x, query, ef_search, ORC_ITER_OFF, 0);
    int n = 0, e; double d; int64_t t;
    while (n < k && orc_scan_next(s, &t, &d, &e)) { ids[n] = e; dist[n] = d; n++; }
    orc_scan_end(s);
    return n;
}

/* the same scan for nq queries on n_threads host threads, one query per thread at a time (one backend per connection in
 * the reference's process model); read-only on the index.  ids_out[nq][k], cnt_out[nq]. */
typedef struct { orc_index *x; const uint8_t *q; size_t qb; int nq, ef, k, tid, nthr; int *ids; int *cnt; } many_arg;
static void *many_worker(void *vp)
{
    many_arg *a = (many_arg *)vp;
    tl_nocount = 1;
    double *d = malloc((size_t)a->k * sizeof(double));
    for (int q = a->tid; q < a->nq; q += a->nthr)
        a->cnt[q] = orc_search_topk(a->x, a->q + (size_t)q * a->qb, a->ef, a->k, a->ids + (size_t)q * a->k, d);
    free(d);
    return NULL;
}
ORC_API void orc_search_many(orc_index *x, const void *queries, int nq, int ef_search, int k, int n_threads, int *ids_out, int *cnt_out)
{
    if (n_threads < 1) n_threads = 1;
    (void)popcnt_table();
    pthread_t *th = malloc((size_t)n_threads * sizeof(pthread_t));
    many_arg *args = malloc((size_t)n_threads * sizeof(many_arg));
    for (int t = 0; t < n_threads; t++) {
        args[t] = (many_arg){ x, (const uint8_t *)queries, x->row_bytes, nq, ef_search, k, t, n_threads, ids_out, cnt_out };
        pthread_create(&th[t], NULL, many_worker, &args[t]);
    }
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(th); free(args);
}

/* ------------------------------------------------------------------ */
/* the index as PostgreSQL pages, restated from       */
/* build.rs:545-821 step by step (create_meta_page, create_graph_pages  */
/* with placeholder neighbour tuples, write_neighbor_tuples,            */
/* update_meta_page) on top of the four bufpage.c primitives the        */
/* reference calls.  Pages live in one caller buffer; block b is at     */
/* b * 8192.                                                            */
/* ------------------------------------------------------------------ */
#define PG_BLCKSZ 8192u
typedef struct { uint32_t pd_lsn_hi, pd_lsn_lo; uint16_t pd_checksum, pd_flags, pd_lower, pd_upper, pd_special, pd_pagesize_version; uint32_t pd_prune_xid; } pg_page_header;   /* 24 bytes */
typedef struct { uint16_t bi_hi, bi_lo, ip_posid; } pg_tid;                                  /* ItemPointerData */
typedef struct { uint32_t nextblkno; uint16_t unused, page_id; } hnsw_opaque;                /* types/hnsw.rs:17-27 */
typedef struct { uint32_t magic_number, version, dimensions; uint16_t m, ef_construction; uint32_t entry_blkno; uint16_t entry_offno; int16_t entry_level; uint32_t insert_page; } hnsw_meta;   /* types/hnsw.rs:52-72 */
typedef struct { uint8_t type_, level, deleted, version; pg_tid heaptids[HNSW_HEAPTIDS]; pg_tid neighbortid; uint16_t unused; } hnsw_etup;   /* types/hnsw.rs:110-126 */
typedef struct { uint8_t type_, version; uint16_t count; } hnsw_ntup;                        /* types/hnsw.rs:149-157 */
static size_t pg_maxalign(size_t x) { return (x + 7u) & ~(size_t)7u; }                       /* types/hnsw.rs:316-319 */
static void pg_tid_set(pg_tid *t, uint32_t blk, uint16_t off) { t->bi_hi = (uint16_t)(blk >> 16); t->bi_lo = (uint16_t)(blk & 0xffff); t->ip_posid = off; }
static void pg_tid_invalid(pg_tid *t) { pg_tid_set(t, 0xFFFFFFFFu, 0); }
static void pg_page_init(uint8_t *page, size_t special)
{   /* PageInit */
    memset(page, 0, PG_BLCKSZ);
    pg_page_header *h = (pg_page_header *)page;
    h->pd_lower = (uint16_t)sizeof(pg_page_header);
    h->pd_upper = (uint16_t)(PG_BLCKSZ - pg_maxalign(special));
    h->pd_special = h->pd_upper;
    h->pd_pagesize_version = (uint16_t)(PG_BLCKSZ | 4u);
}
static size_t pg_page_free_space(const uint8_t *page)
{   /* PageGetFreeSpace */
    const pg_page_header *h = (const pg_page_header *)page;
    int space = (int)h->pd_upper - (int)h->pd_lower;
    return space < 4 ? 0 : (size_t)(space - 4);
}
static uint16_t pg_page_max_offset(const uint8_t *page)
{   /* PageGetMaxOffsetNumber */
    const pg_page_header *h = (const pg_page_header *)page;
    return h->pd_lower <= sizeof(pg_page_header) ? 0 : (uint16_t)((h->pd_lower - sizeof(pg_page_header)) / 4);
}
static uint16_t pg_page_add_item(uint8_t *page, const void *item, size_t size)
{   /* PageAddItemExtended(page, item, size, InvalidOffsetNumber, 0) */
    pg_page_header *h = (pg_page_header *)page;
    uint16_t off = (uint16_t)(pg_page_max_offset(page) + 1);
    int lower = h->pd_lower + 4, upper = (int)h->pd_upper - (int)pg_maxalign(size);
    if (lower > upper) return 0;
    uint32_t lp = (uint32_t)upper | (1u << 15) | ((uint32_t)size << 17);    /* ItemIdSetNormal: lp_off:15, lp_flags:2 = LP_NORMAL, lp_len:15 */
    memcpy(page + h->pd_lower, &lp, 4);
    memcpy(page + upper, item, size);
    h->pd_lower = (uint16_t)lower; h->pd_upper = (uint16_t)upper;
    return off;
}
static int pg_page_overwrite(uint8_t *page, uint16_t off, const void *item, size_t size)
{   /* PageIndexTupleOverwrite with an unchanged MAXALIGNed size */
    uint32_t lp; memcpy(&lp, page + sizeof(pg_page_header) + (size_t)(off - 1) * 4, 4);
    uint32_t lp_off = lp & 0x7fff, lp_len = lp >> 17;
    if (pg_maxalign(lp_len) != pg_maxalign(size)) return 0;
    memcpy(page + lp_off, item, size);
    lp = lp_off | (1u << 15) | ((uint32_t)size << 17);
    memcpy(page + sizeof(pg_page_header) + (size_t)(off - 1) * 4, &lp, 4);
    return 1;
}
static void hnsw_init_page(uint8_t *page)
{   /* build.rs:59-65 */
    pg_page_init(page, sizeof(hnsw_opaque));
    hnsw_opaque *o = (hnsw_opaque *)(page + ((pg_page_header *)page)->pd_special);
    o->nextblkno = 0xFFFFFFFFu; o->page_id = 0xFF90;
}
typedef struct { uint32_t blkno; uint16_t offno; uint32_t neighbor_page; uint16_t neighbor_offno; } disk_loc;   /* build.rs:283-285 */

/* Returns the number of pages (0 on error: tuple too large, cap too small, add failure).  blk_out/off_out per element (nullable). */
ORC_API uint64_t orc_index_write_pages(const orc_index *x, uint8_t *pages, uint64_t cap_pages, uint32_t *blk_out, uint16_t *off_out)
{
    const size_t max_size = PG_BLCKSZ - pg_maxalign(sizeof(pg_page_header)) - pg_maxalign(sizeof(hnsw_opaque)) - 4;   /* types/hnsw.rs:325-331 */
    const size_t value_size = 8 + x->row_bytes;                                  /* varlena header + payload */
    if (cap_pages < 2) return 0;
    /* create_meta_page */
    uint8_t *mp = pages;
    hnsw_init_page(mp);
    hnsw_meta *meta = (hnsw_meta *)(mp + sizeof(pg_page_header));
    meta->magic_number = 0xA953A953u; meta->version = 1; meta->dimensions = (uint32_t)x->dim;
    meta->m = (uint16_t)x->m; meta->ef_construction = (uint16_t)x->efc;
    meta->entry_blkno = 0xFFFFFFFFu; meta->entry_offno = 0; meta->entry_level = -1; meta->insert_page = 0xFFFFFFFFu;
    ((pg_page_header *)mp)->pd_lower = (uint16_t)(sizeof(pg_page_header) + sizeof(hnsw_meta));
    /* create_graph_pages */
    uint32_t blk = 1;
    uint8_t *page = pages + (size_t)blk * PG_BLCKSZ;
    hnsw_init_page(page);
    disk_loc *locs = calloc((size_t)x->n + 1, sizeof(disk_loc));
    uint8_t *etup_buf = calloc(1, PG_BLCKSZ), *ntup_buf = calloc(1, PG_BLCKSZ);
    uint64_t result = 0;
    for (int idx = 0; idx < x->n; idx++) {
        const elem_t *el = &x->el[idx];
        locs[idx].blkno = 0xFFFFFFFFu;
        if (el->merged) continue;                                                /* the reference pops a merged duplicate from `elements` (build.rs:507-509) */
        size_t etup_size = pg_maxalign(sizeof(hnsw_etup) + value_size);
        size_t ntup_size = pg_maxalign(sizeof(hnsw_ntup) + (size_t)(el->level + 2) * x->m * sizeof(pg_tid));
        size_t combined = etup_size + ntup_size + 4;
        if (etup_size > max_size) goto done;
        memset(etup_buf, 0, etup_size);
        hnsw_etup *et = (hnsw_etup *)etup_buf;
        et->type_ = 1; et->level = (uint8_t)el->level; et->deleted = 0; et->version = 0;
        for (int i = 0; i < HNSW_HEAPTIDS; i++) {
            if (i < el->ntids) pg_tid_set(&et->heaptids[i], (uint32_t)((uint64_t)el->tids[i] >> 16), (uint16_t)(el->tids[i] & 0xffff));
            else pg_tid_invalid(&et->heaptids[i]);
        }
        uint8_t *v = etup_buf + sizeof(hnsw_etup);
        uint32_t vl = (uint32_t)value_size << 2; memcpy(v, &vl, 4);
        if (x->dtype == ORC_BIT) { int32_t bl = x->dim; memcpy(v + 4, &bl, 4); } else { int16_t d16 = (int16_t)x->dim, z = 0; memcpy(v + 4, &d16, 2); memcpy(v + 6, &z, 2); }
        memcpy(v + 8, rowp((orc_index *)x, idx), x->row_bytes);
        size_t fs = pg_page_free_space(page);
        if (fs < etup_size || (combined <= max_size && fs < combined)) {         /* hnsw_build_append_page */
            if ((uint64_t)blk + 2 > cap_pages) goto done;
            ((hnsw_opaque *)(page + ((pg_page_header *)page)->pd_special))->nextblkno = blk + 1;
            blk++; page = pages + (size_t)blk * PG_BLCKSZ; hnsw_init_page(page);
        }
        uint32_t eblk = blk; uint16_t eoff = (uint16_t)(pg_page_max_offset(page) + 1);
        uint32_t nblk; uint16_t noff;
        if (combined <= max_size) { nblk = eblk; noff = (uint16_t)(eoff + 1); } else { nblk = eblk + 1; noff = 1; }
        locs[idx] = (disk_loc){ eblk, eoff, nblk, noff };
        pg_tid_set(&et->neighbortid, nblk, noff);
        if (pg_page_add_item(page, etup_buf, etup_size) != eoff) goto done;
        if (pg_page_free_space(page) < ntup_size) {
            if ((uint64_t)blk + 2 > cap_pages) goto done;
            ((hnsw_opaque *)(page + ((pg_page_header *)page)->pd_special))->nextblkno = blk + 1;
            blk++; page = pages + (size_t)blk * PG_BLCKSZ; hnsw_init_page(page);
        }
        memset(ntup_buf, 0, ntup_size);
        ((hnsw_ntup *)ntup_buf)->type_ = 2;
        if (blk != nblk || pg_page_add_item(page, ntup_buf, ntup_size) != noff) goto done;
    }
    uint32_t insert_page = blk;
    /* write_neighbor_tuples */
    for (int idx = 0; idx < x->n; idx++) {
        const elem_t *el = &x->el[idx];
        if (el->merged) continue;
        size_t ntup_size = pg_maxalign(sizeof(hnsw_ntup) + (size_t)(el->level + 2) * x->m * sizeof(pg_tid));
        memset(ntup_buf, 0, ntup_size);
        hnsw_ntup *nt = (hnsw_ntup *)ntup_buf; nt->type_ = 2; nt->version = 0;
        pg_tid *tids = (pg_tid *)(ntup_buf + sizeof(hnsw_ntup));
        int k = 0;
        for (int lc = el->level; lc >= 0; lc--) {
            int lm = layer_m(x->m, lc);
            for (int i = 0; i < lm; i++, k++) {
                if (i < el->ncnt[lc]) { const disk_loc *nl = &locs[el->nbr[lc][i].idx]; pg_tid_set(&tids[k], nl->blkno, nl->offno); }
                else pg_tid_invalid(&tids[k]);
            }
        }
        nt->count = (uint16_t)k;
        if (!pg_page_overwrite(pages + (size_t)locs[idx].neighbor_page * PG_BLCKSZ, locs[idx].neighbor_offno, ntup_buf, ntup_size)) goto done;
    }
    /* update_meta_page */
    if (x->entry >= 0) { meta->entry_blkno = locs[x->entry].blkno; meta->entry_offno = locs[x->entry].offno; meta->entry_level = (int16_t)x->el[x->entry].level; }
    meta->insert_page = insert_page;
    for (int i = 0; i < x->n; i++) { if (blk_out) blk_out[i] = locs[i].blkno; if (off_out) off_out[i] = locs[i].offno; }
    result = (uint64_t)blk + 1;
done:
    free(locs); free(etup_buf); free(ntup_buf);
    return result;
}

/* exact brute force top-k (ground truth for recall; distances in the index's order) */
ORC_API int orc_bruteforce_topk(orc_index *x, const void *query, int k, int *ids, double *dist)
{
    int n = 0;
    for (int i = 0; i < x->n; i++) {
        if (x->el[i].merged) continue;
        double d = orc_distance(x->dtype, x->metric, x->dim, query, rowp(x, i), x->order);
        if (n == k && !(d < dist[k - 1])) continue;
        int p = n < k ? n++ : k - 1;
        while (p > 0 && d < dist[p - 1]) { dist[p] = dist[p - 1]; ids[p] = ids[p - 1]; p--; }
        dist[p] = d; ids[p] = i;
    }
    return n;
}

/* bulk helpers so Python tests stay fast */
ORC_API void orc_distances_many(int dtype, int metric, int dim, const void *q, const void *rows, const int32_t *ids, int n, int order, double *out)
{
    size_t rb = orc_row_bytes(dtype, dim);
    for (int i = 0; i < n; i++) {
        const uint8_t *r = (const uint8_t *)rows + (size_t)(ids ? ids[i] : i) * rb;
        out[i] = orc_distance(dtype, metric, dim, q, r, order);
    }
}
ORC_API void orc_pairwise(int dtype, int metric, int dim, const void *rows, const int32_t *ids, int w, int order, double *out)
{
    size_t rb = orc_row_bytes(dtype, dim);
    for (int i = 0; i < w; i++) for (int j = 0; j < w; j++)
        out[(size_t)i * w + j] = orc_distance(dtype, metric, dim, (const uint8_t *)rows + (size_t)ids[i] * rb,
                                              (const uint8_t *)rows + (size_t)ids[j] * rb, order);
}

// The "affine in the candidate" distance scorers:
// PairRE, TripleRE, InterHT, TranS  (reference scoring.py:465-743, 1418-1750).
//
// All four score a candidate entity e against a query (the kept entity and the
// relation) as
//
//     score(q, e) = - || U_q * c1(e) + V_q * c2(e) + R_q ||_p          (elementwise over d)
//
// where c1, c2 are the one or two d-wide parts of the entity row (PairRE /
// TripleRE: one part; InterHT: main | aux; TranS: main | tilde), optionally
// L2-normalised per part (`normalize_entities`, F.normalize eps 1e-12), and
// U, V, R are d-wide vectors that depend on the query only.  The query matrix
// handed to these kernels is [n_query, (n_part + 1) * d] = [U | V | R] (V absent
// for one part); `bess_model_desc.width` = n_part * d, reserved[0] = n_part,
// reserved[1] bit 0 = normalise.
//
//   per-triple (HBM bound): same mapping as neg_pertriple.hip - a 16-lane DPP row
//     streams one candidate row, both parts of a d-chunk land in the same lane;
//     the norm of each part is one more DPP reduction over registers.
//   shared (VALU bound): 64 x 64 LDS tile kernel over *dense fp32, already
//     normalised* candidates (bess_normalize_rows gathers + converts + normalises the
//     N candidate rows first; its backward maps the gradient back).
#include <algorithm>
#include <string.h>

#include "common.h"

namespace bess {

constexpr float NORM_EPS = 1e-12f;  // torch.nn.functional.normalize default

// ---------------------------------------------------------------------------
// normalise rows: out[i, p*d + w] = e[i, p*d + w] / max(||e[i, p*d : (p+1)*d]||, eps)
// one wave per row
template <typename T>
__global__ __launch_bounds__(256) void k_normalize_rows(const T* __restrict__ base, const int32_t* __restrict__ idx,
                                                        int64_t n, int d, int n_part, int normalize,
                                                        float* __restrict__ out, float* __restrict__ inv) {
    const int lane = threadIdx.x & 63;
    const int64_t i = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t r = idx ? static_cast<int64_t>(idx[i]) : i;
    const T* rp = base + r * n_part * d;
    float* op = out + i * n_part * d;
    for (int p = 0; p < n_part; ++p) {
        float s = 1.f;
        if (normalize) {
            float ss = 0.f;
            for (int w = lane; w < d; w += 64) {
                const float v = to_f32(rp[p * d + w]);
                ss = fmaf(v, v, ss);
            }
            ss = wave_allreduce_sum(ss);
            s = 1.f / fmaxf(sqrtf(ss), NORM_EPS);
        }
        for (int w = lane; w < d; w += 64) op[p * d + w] = to_f32(rp[p * d + w]) * s;
        if (inv && lane == 0) inv[i * n_part + p] = s;
    }
}

// d_e = inv * (d_hat - hat * <hat, d_hat>)   per part   (inv * d_hat where the norm was clamped)
__global__ __launch_bounds__(256) void k_normalize_rows_bwd(const float* __restrict__ hat, const float* __restrict__ inv,
                                                            const float* __restrict__ d_hat, int64_t n, int d,
                                                            int n_part, float* __restrict__ d_e) {
    const int lane = threadIdx.x & 63;
    const int64_t i = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (i >= n) return;
    for (int p = 0; p < n_part; ++p) {
        const float* hp = hat + (i * n_part + p) * d;
        const float* gp = d_hat + (i * n_part + p) * d;
        float* op = d_e + (i * n_part + p) * d;
        const float s = inv[i * n_part + p];
        float dot = 0.f;
        for (int w = lane; w < d; w += 64) dot = fmaf(hp[w], gp[w], dot);
        dot = wave_allreduce_sum(dot);
        if (s >= 1.f / NORM_EPS) dot = 0.f;  // clamped norm: x / eps is linear
        for (int w = lane; w < d; w += 64) op[w] = s * (gp[w] - hp[w] * dot);
    }
}

// ---------------------------------------------------------------------------
// Query transform: [U | V | R] from the kept entity (head when tails are corrupted, tail when
// heads are) and the relation row, and its backward.  k0, k1 = (normalised) parts of the kept
// entity, o = the model's offset constant (TripleRE: u, added to r_h and r_t).
//
//   PairRE   rel [r_h | r_t]          tails: U = r_t            R = -k0 r_h
//                                     heads: U = r_h            R = -k0 r_t
//   TripleRE rel [r_h | r_m | r_t]    tails: U = r_t + o        R = -(k0 (r_h + o) + r_m)
//                                     heads: U = r_h + o        R = -(k0 (r_t + o) - r_m)
//   InterHT  rel [r]                  tails: U = -(k1 + o)      V =  k0   R = r + k0 o
//                                     heads: U =   k1 + o       V = -k0   R = r - k0 o
//   TranS    rel [r | r_bar | r_hat]  tails: U = -(k1 + o - r_hat)  V =  k0   R = r + k0 (o + r_bar)
//                                     heads: U =   k1 + o + r_bar   V = -k0   R = r - k0 (o - r_hat)
// (reference scoring.py:540-593, 681-743, 1499-1572, 1661-1750, rearranged so that the candidate
// enters as U c1 + V c2 + R.)  One wave per row.
enum AffKind : int { AFF_PAIRRE = 0, AFF_TRIPLERE = 1, AFF_INTERHT = 2, AFF_TRANS = 3 };

struct AffQueryArgs {
    const void* ent_base;
    const int32_t* ent_idx;
    const void* rel_table;
    const int32_t* rel_idx;
    int64_t n;
    int d, n_part, rel_width, kind, tails, normalize;
    float o;
};

template <typename T>
__device__ __forceinline__ void aff_kept_inv(const T* x, int d, int n_part, int normalize, int lane, float (&inv)[2]) {
    inv[0] = inv[1] = 1.f;
    if (!normalize) return;
    for (int p = 0; p < n_part; ++p) {
        float ss = 0.f;
        for (int w = lane; w < d; w += 64) {
            const float v = to_f32(x[p * d + w]);
            ss = fmaf(v, v, ss);
        }
        ss = wave_allreduce_sum(ss);
        inv[p] = 1.f / fmaxf(sqrtf(ss), NORM_EPS);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_aff_query_fwd(AffQueryArgs a, float* __restrict__ query) {
    const int lane = threadIdx.x & 63;
    const int64_t q = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (q >= a.n) return;
    const int d = a.d;
    const T* x = row_ptr(static_cast<const T*>(a.ent_base), a.ent_idx, q, a.n_part * d);
    const T* r = static_cast<const T*>(a.rel_table) + static_cast<int64_t>(a.rel_idx[q]) * a.rel_width;
    float inv[2];
    aff_kept_inv<T>(x, d, a.n_part, a.normalize, lane, inv);
    float* out = query + q * (a.n_part + 1) * d;
    const float o = a.o, sg = a.tails ? 1.f : -1.f;
    for (int w = lane; w < d; w += 64) {
        const float k0 = to_f32(x[w]) * inv[0];
        if (a.kind == AFF_PAIRRE) {
            const float r_h = to_f32(r[w]), r_t = to_f32(r[d + w]);
            out[w] = a.tails ? r_t : r_h;
            out[d + w] = -(k0 * (a.tails ? r_h : r_t));
        } else if (a.kind == AFF_TRIPLERE) {
            const float r_h = to_f32(r[w]) + o, r_m = to_f32(r[d + w]), r_t = to_f32(r[2 * d + w]) + o;
            out[w] = a.tails ? r_t : r_h;
            out[d + w] = a.tails ? -(k0 * r_h + r_m) : -(k0 * r_t - r_m);
        } else {
            const float k1 = to_f32(x[d + w]) * inv[1];
            if (a.kind == AFF_INTERHT) {
                out[w] = -sg * (k1 + o);
                out[d + w] = sg * k0;
                out[2 * d + w] = to_f32(r[w]) + sg * k0 * o;
            } else {
                const float r_bar = to_f32(r[d + w]), r_hat = to_f32(r[2 * d + w]);
                out[w] = a.tails ? -(k1 + o - r_hat) : (k1 + o + r_bar);
                out[d + w] = sg * k0;
                out[2 * d + w] = to_f32(r[w]) + (a.tails ? k0 * (o + r_bar) : -k0 * (o - r_hat));
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_aff_query_bwd(AffQueryArgs a, const float* __restrict__ d_query,
                                                       float* __restrict__ d_ent, float* __restrict__ d_rel) {
    const int lane = threadIdx.x & 63;
    const int64_t q = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (q >= a.n) return;
    const int d = a.d;
    const T* x = row_ptr(static_cast<const T*>(a.ent_base), a.ent_idx, q, a.n_part * d);
    const int64_t rid = a.rel_idx[q];
    const T* r = static_cast<const T*>(a.rel_table) + rid * a.rel_width;
    float* dr = d_rel + rid * a.rel_width;
    float inv[2];
    aff_kept_inv<T>(x, d, a.n_part, a.normalize, lane, inv);
    const float* g = d_query + q * (a.n_part + 1) * d;
    float* de = d_ent + q * a.n_part * d;
    const float o = a.o, sg = a.tails ? 1.f : -1.f;
    // pass 1: gradient wrt the normalised kept parts (staged in d_ent), relation gradient, <k, dk>
    float dot[2] = {0.f, 0.f};
    for (int w = lane; w < d; w += 64) {
        const float k0 = to_f32(x[w]) * inv[0];
        float dk0, dk1 = 0.f;
        if (a.kind == AFF_PAIRRE) {
            const float dU = g[w], dR = g[d + w];
            const float r_h = to_f32(r[w]), r_t = to_f32(r[d + w]);
            dk0 = -dR * (a.tails ? r_h : r_t);
            unsafeAtomicAdd(dr + (a.tails ? d : 0) + w, dU);            // U is r_t (tails) / r_h (heads)
            unsafeAtomicAdd(dr + (a.tails ? 0 : d) + w, -dR * k0);      // the other projection
        } else if (a.kind == AFF_TRIPLERE) {
            const float dU = g[w], dR = g[d + w];
            const float r_h = to_f32(r[w]) + o, r_t = to_f32(r[2 * d + w]) + o;
            dk0 = -dR * (a.tails ? r_h : r_t);
            unsafeAtomicAdd(dr + (a.tails ? 2 * d : 0) + w, dU);
            unsafeAtomicAdd(dr + (a.tails ? 0 : 2 * d) + w, -dR * k0);
            unsafeAtomicAdd(dr + d + w, a.tails ? -dR : dR);
        } else {
            const float k1 = to_f32(x[d + w]) * inv[1];
            const float dU = g[w], dV = g[d + w], dR = g[2 * d + w];
            if (a.kind == AFF_INTERHT) {
                dk1 = -sg * dU;
                dk0 = sg * (dV + dR * o);
                unsafeAtomicAdd(dr + w, dR);
            } else {
                const float r_bar = to_f32(r[d + w]), r_hat = to_f32(r[2 * d + w]);
                dk1 = -sg * dU;
                unsafeAtomicAdd(dr + w, dR);
                if (a.tails) {
                    dk0 = dV + dR * (o + r_bar);
                    unsafeAtomicAdd(dr + 2 * d + w, dU);        // U = -(k1 + o - r_hat)
                    unsafeAtomicAdd(dr + d + w, dR * k0);       // R = r + k0 (o + r_bar)
                } else {
                    dk0 = -dV - dR * (o - r_hat);
                    unsafeAtomicAdd(dr + d + w, dU);            // U = k1 + o + r_bar
                    unsafeAtomicAdd(dr + 2 * d + w, dR * k0);   // R = r - k0 (o - r_hat)
                }
            }
            de[d + w] = dk1;
            dot[1] = fmaf(k1, dk1, dot[1]);
        }
        de[w] = dk0;
        dot[0] = fmaf(k0, dk0, dot[0]);
    }
    // pass 2: through the normalisation, d x = inv (d k - k <k, d k>)
    for (int p = 0; p < a.n_part; ++p) {
        float dt = a.normalize ? wave_allreduce_sum(dot[p]) : 0.f;
        if (inv[p] >= 1.f / NORM_EPS) dt = 0.f;
        if (!a.normalize) continue;
        for (int w = lane; w < d; w += 64) {
            const float k = to_f32(x[p * d + w]) * inv[p];
            de[p * d + w] = inv[p] * (de[p * d + w] - k * dt);
        }
    }
}

static int aff_query_args(const bess_model_desc* d, int32_t side, const void* ent_base, const int32_t* ent_idx,
                          const void* rel_table, const int32_t* rel_idx, int64_t n, AffQueryArgs* a) {
    BESS_REQUIRE(side == BESS_CORRUPT_HEAD || side == BESS_CORRUPT_TAIL, "query (affine): bad side %d", side);
    BESS_REQUIRE(n >= 0, "query (affine): negative size");
    if (n == 0) return BESS_OK;
    BESS_REQUIRE(ent_base && rel_table && rel_idx, "query (affine): NULL pointer");
    const int n_part = d->reserved[0];
    const int dd = d->width / n_part;
    const int kind = (d->reserved[1] >> 8) & 0xff;
    BESS_REQUIRE(kind >= AFF_PAIRRE && kind <= AFF_TRANS, "query (affine): unknown family member %d", kind);
    const int want_rel = (kind == AFF_PAIRRE ? 2 : kind == AFF_INTERHT ? 1 : 3) * dd;
    const int want_part = (kind == AFF_PAIRRE || kind == AFF_TRIPLERE) ? 1 : 2;
    BESS_REQUIRE(d->rel_width == want_rel && n_part == want_part,
                 "query (affine): member %d needs %d entity part(s) and relation rows of %d scalars (got %d, %d)", kind,
                 want_part, want_rel, n_part, d->rel_width);
    float o;
    static_assert(sizeof(o) == sizeof(d->reserved[2]), "constant travels bit-cast in reserved[2]");
    memcpy(&o, &d->reserved[2], sizeof(o));
    *a = AffQueryArgs{ent_base, ent_idx, rel_table, rel_idx, n, dd, n_part, d->rel_width, kind,
                      side == BESS_CORRUPT_TAIL, d->reserved[1] & 1, o};
    return BESS_OK;
}

int affine_query_fwd(const bess_model_desc* d, int32_t side, const void* ent_base, const int32_t* ent_idx,
                     const void* rel_table, const int32_t* rel_idx, int64_t n, float* query, hipStream_t st) {
    AffQueryArgs a;
    if (int e = aff_query_args(d, side, ent_base, ent_idx, rel_table, rel_idx, n, &a)) return e;
    if (n == 0) return BESS_OK;
    BESS_REQUIRE(query, "query_fwd (affine): NULL out");
    const unsigned blocks = static_cast<unsigned>(ceil_div(n, 4));
    if (d->dtype == BESS_F32) k_aff_query_fwd<float><<<blocks, 256, 0, st>>>(a, query);
    else k_aff_query_fwd<half_t><<<blocks, 256, 0, st>>>(a, query);
    return check_launch("query_fwd (affine)");
}

int affine_query_bwd(const bess_model_desc* d, int32_t side, const void* ent_base, const int32_t* ent_idx,
                     const void* rel_table, const int32_t* rel_idx, int64_t n, const float* d_query, float* d_ent,
                     float* d_rel, hipStream_t st) {
    AffQueryArgs a;
    if (int e = aff_query_args(d, side, ent_base, ent_idx, rel_table, rel_idx, n, &a)) return e;
    if (n == 0) return BESS_OK;
    BESS_REQUIRE(d_query && d_ent && d_rel, "query_bwd (affine): NULL pointer");
    const unsigned blocks = static_cast<unsigned>(ceil_div(n, 4));
    if (d->dtype == BESS_F32) k_aff_query_bwd<float><<<blocks, 256, 0, st>>>(a, d_query, d_ent, d_rel);
    else k_aff_query_bwd<half_t><<<blocks, 256, 0, st>>>(a, d_query, d_ent, d_rel);
    return check_launch("query_bwd (affine)");
}

// ---------------------------------------------------------------------------
// per-triple negatives
struct AffArgs {
    const float* query;   // [n_query, (NPART + 1) * d]
    const void* base;
    const int32_t* idx;   // [n_query * n_neg]
    int64_t n_query;
    int n_neg;
    int d;
    int nch;              // chunks of VEC scalars per part
    int nb, items_per_query;
    int normalize;
    float p;  // the norm of the P == 2 kernels: any p != 1 (2: multiply / sqrt; others through powf - common.h lp_*)
};

template <typename T, int VEC, int IT, int NPART>
__device__ __forceinline__ void aff_load_row(const T* rp, int g, int d, int nch, float (&ev)[NPART][IT][VEC]) {
#pragma unroll
    for (int p = 0; p < NPART; ++p)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            load_chunk<T, VEC>(rp + p * d, g + 16 * it, nch, ev[p][it]);
        }
}

template <int VEC, int IT, int NPART>
__device__ __forceinline__ void aff_load_query(const float* qp, int g, int d, int nch, float (&qv)[NPART + 1][IT][VEC]) {
#pragma unroll
    for (int p = 0; p <= NPART; ++p)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            load_chunk<float, VEC>(qp + p * d, g + 16 * it, nch, qv[p][it]);
        }
}

// 1 / max(||part||, eps) for every part (identical in the 16 lanes of the DPP row)
template <int VEC, int IT, int NPART>
__device__ __forceinline__ void aff_inv_norm(const float (&ev)[NPART][IT][VEC], int normalize, float (&inv)[NPART]) {
#pragma unroll
    for (int p = 0; p < NPART; ++p) {
        inv[p] = 1.f;
        if (normalize) {
            float ss = 0.f;
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int v = 0; v < VEC; ++v) ss = fmaf(ev[p][it][v], ev[p][it][v], ss);
            ss = row16_allreduce_sum(ss);
            inv[p] = 1.f / fmaxf(sqrtf(ss), NORM_EPS);
        }
    }
}

// d_w = U_w c1_w + V_w c2_w + R_w   (c already scaled by inv)
template <int VEC, int IT, int NPART>
__device__ __forceinline__ float aff_delta(const float (&qv)[NPART + 1][IT][VEC], const float (&ev)[NPART][IT][VEC],
                                           const float (&inv)[NPART], int it, int v) {
    float dlt = qv[NPART][it][v];
#pragma unroll
    for (int p = NPART - 1; p >= 0; --p) dlt = fmaf(qv[p][it][v], ev[p][it][v] * inv[p], dlt);
    return dlt;
}

template <typename T, int VEC, int IT, int NPART, int P>
__global__ __launch_bounds__(256) void k_aff_pertriple_fwd(AffArgs a, float* __restrict__ out, int64_t ld_out) {
    const int lane = threadIdx.x & 63;
    const int g = lane & 15, sub = lane >> 4;
    const int64_t item = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (item >= a.n_query * a.items_per_query) return;
    const int64_t q = item / a.items_per_query;
    const int k0 = static_cast<int>(item - q * a.items_per_query) * a.nb;
    const int k1 = min(k0 + a.nb, a.n_neg);
    float qv[NPART + 1][IT][VEC];
    aff_load_query<VEC, IT, NPART>(a.query + q * (NPART + 1) * a.d, g, a.d, a.nch, qv);
    const T* base = static_cast<const T*>(a.base);
    const int32_t* idx = a.idx + q * a.n_neg;
    float* orow = out + q * ld_out;
    for (int kb = k0; kb < k1; kb += 4) {  // wave-uniform
        const int k = kb + sub;
        const bool valid = k < k1;
        const int ks = valid ? k : (k1 - 1);
        float ev[NPART][IT][VEC], inv[NPART];
        aff_load_row<T, VEC, IT, NPART>(base + static_cast<int64_t>(idx[ks]) * NPART * a.d, g, a.d, a.nch, ev);
        aff_inv_norm<VEC, IT, NPART>(ev, a.normalize, inv);
        float acc = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float dlt = aff_delta<VEC, IT, NPART>(qv, ev, inv, it, v);
                if (P == 1) acc += fabsf(dlt);
                else acc += lp_term(dlt, a.p);
            }
        acc = row16_allreduce_sum(acc);
        if (P == 2) acc = lp_root(acc, a.p);
        if (g == 0 && valid) orow[k] = -acc;
    }
}

// backward: d_query[q] = [dU | dV | dR] summed over the negatives, d_neg[(q, k), :] optional
template <typename T, int VEC, int IT, int NPART, int P>
__global__ __launch_bounds__(256) void k_aff_pertriple_bwd(AffArgs a, const float* __restrict__ d_out, int64_t ld_dout,
                                                           float* __restrict__ d_query, float* __restrict__ d_neg) {
    const int lane = threadIdx.x & 63;
    const int g = lane & 15, sub = lane >> 4;
    const int64_t item = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (item >= a.n_query * a.items_per_query) return;
    const int64_t q = item / a.items_per_query;
    const int k0 = static_cast<int>(item - q * a.items_per_query) * a.nb;
    const int k1 = min(k0 + a.nb, a.n_neg);
    float qv[NPART + 1][IT][VEC], dq[NPART + 1][IT][VEC];
    aff_load_query<VEC, IT, NPART>(a.query + q * (NPART + 1) * a.d, g, a.d, a.nch, qv);
#pragma unroll
    for (int p = 0; p <= NPART; ++p)
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int v = 0; v < VEC; ++v) dq[p][it][v] = 0.f;
    const T* base = static_cast<const T*>(a.base);
    const int32_t* idx = a.idx + q * a.n_neg;
    for (int kb = k0; kb < k1; kb += 4) {
        const int k = kb + sub;
        const bool valid = k < k1;
        const int ks = valid ? k : (k1 - 1);
        float ev[NPART][IT][VEC], inv[NPART];
        aff_load_row<T, VEC, IT, NPART>(base + static_cast<int64_t>(idx[ks]) * NPART * a.d, g, a.d, a.nch, ev);
        aff_inv_norm<VEC, IT, NPART>(ev, a.normalize, inv);
        // score = -||delta||_p  ->  d score / d delta_w = -sgn(delta_w)  |  -delta_w / ||delta||
        float go = valid ? -d_out[q * ld_dout + ks] : 0.f;
        if (P == 2) {
            float ss = 0.f;
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float dlt = aff_delta<VEC, IT, NPART>(qv, ev, inv, it, v);
                    ss += lp_term(dlt, a.p);
                }
            ss = row16_allreduce_sum(ss);
            go *= lp_inv(lp_root(ss, a.p), a.p);  // norm^(1 - p); 0 at the kink
        }
        float dc[NPART][IT][VEC];   // gradient wrt the normalised parts
        float dots[NPART];
#pragma unroll
        for (int p = 0; p < NPART; ++p) dots[p] = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float dlt = aff_delta<VEC, IT, NPART>(qv, ev, inv, it, v);
                const float s = (P == 1) ? go * sgnf(dlt) : go * lp_dterm(dlt, a.p);
                dq[NPART][it][v] += s;
#pragma unroll
                for (int p = 0; p < NPART; ++p) {
                    const float hat = ev[p][it][v] * inv[p];
                    dq[p][it][v] = fmaf(s, hat, dq[p][it][v]);
                    dc[p][it][v] = s * qv[p][it][v];
                    dots[p] = fmaf(hat, dc[p][it][v], dots[p]);
                }
            }
        if (d_neg) {
            float* dn = d_neg + (q * a.n_neg + ks) * NPART * a.d;
#pragma unroll
            for (int p = 0; p < NPART; ++p) {
                float dot = 0.f;
                if (a.normalize) {
                    dot = row16_allreduce_sum(dots[p]);
                    if (inv[p] >= 1.f / NORM_EPS) dot = 0.f;
                }
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int c = g + 16 * it;
                    if (valid && c < a.nch) {
                        float o[VEC];
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const float hat = ev[p][it][v] * inv[p];
                            o[v] = a.normalize ? inv[p] * (dc[p][it][v] - hat * dot) : dc[p][it][v];
                        }
                        // (fp32 tables: one 16-byte piece per lane, streamed past the caches - as in neg_pertriple.hip)
                        if constexpr (VEC == 4) {
                            typedef float f4 __attribute__((ext_vector_type(4)));
                            f4 o4 = {o[0], o[1], o[2], o[3]};
                            __builtin_nontemporal_store(o4, reinterpret_cast<f4*>(dn + p * a.d + c * VEC));
                        } else {
#pragma unroll
                            for (int v = 0; v < VEC; ++v) dn[p * a.d + c * VEC + v] = o[v];
                        }
                    }
                }
            }
        }
    }
    float* dqp = d_query + q * (NPART + 1) * a.d;
#pragma unroll
    for (int p = 0; p <= NPART; ++p)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = g + 16 * it;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float x = dq[p][it][v];
                x += __shfl_xor(x, 16, 64);
                x += __shfl_xor(x, 32, 64);
                if (sub == 0 && c < a.nch) {
                    if (a.items_per_query == 1) dqp[p * a.d + c * VEC + v] = x;
                    else unsafeAtomicAdd(dqp + p * a.d + c * VEC + v, x);
                }
            }
        }
}

template <typename T, int VEC, int IT, int NPART>
static void aff_pt_launch(int p, bool fwd, const AffArgs& a, float* out, const float* d_out, int64_t ld, float* dq,
                          float* dn, hipStream_t st) {
    const unsigned blocks = static_cast<unsigned>(ceil_div(a.n_query * a.items_per_query, 4));
    if (fwd) {
        if (p == 1) k_aff_pertriple_fwd<T, VEC, IT, NPART, 1><<<blocks, 256, 0, st>>>(a, out, ld);
        else k_aff_pertriple_fwd<T, VEC, IT, NPART, 2><<<blocks, 256, 0, st>>>(a, out, ld);
    } else {
        if (p == 1) k_aff_pertriple_bwd<T, VEC, IT, NPART, 1><<<blocks, 256, 0, st>>>(a, d_out, ld, dq, dn);
        else k_aff_pertriple_bwd<T, VEC, IT, NPART, 2><<<blocks, 256, 0, st>>>(a, d_out, ld, dq, dn);
    }
}

template <typename T, int VEC, int NPART>
static int aff_pt_by_it(int it, int p, bool fwd, const AffArgs& a, float* out, const float* d_out, int64_t ld,
                        float* dq, float* dn, hipStream_t st) {
    if (it <= 1) aff_pt_launch<T, VEC, 1, NPART>(p, fwd, a, out, d_out, ld, dq, dn, st);
    else if (it <= 2) aff_pt_launch<T, VEC, 2, NPART>(p, fwd, a, out, d_out, ld, dq, dn, st);
    else if (it <= 4) aff_pt_launch<T, VEC, 4, NPART>(p, fwd, a, out, d_out, ld, dq, dn, st);
    else if (it <= 8) aff_pt_launch<T, VEC, 8, NPART>(p, fwd, a, out, d_out, ld, dq, dn, st);
    else return fail(BESS_EUNSUPPORTED, "affine scorers: part of %d scalars too wide", a.d);
    return BESS_OK;
}

int affine_pertriple(const bess_model_desc* d, bool fwd, const float* query, int64_t n_query, const void* neg_base,
                     const int32_t* neg_idx, int64_t n_neg, float* out, const float* d_out, int64_t ld, float* dq,
                     float* dn, hipStream_t st) {
    const int n_part = d->reserved[0];
    const int dd = d->width / n_part;
    const int vec = (dd % 4 == 0) ? 4 : 1;
    AffArgs a;
    a.query = query;
    a.base = neg_base;
    a.idx = neg_idx;
    a.n_query = n_query;
    a.n_neg = static_cast<int>(n_neg);
    a.d = dd;
    a.nch = dd / vec;
    int nb = 64;
    while (nb > 8 && n_query * ceil_div(n_neg, nb) < 256 * 16 * 2) nb >>= 1;
    a.nb = nb;
    a.items_per_query = static_cast<int>(ceil_div(n_neg, nb));
    a.normalize = d->reserved[1] & 1;
    a.p = static_cast<float>(d->norm_p);
    const int it = static_cast<int>(ceil_div(a.nch, 16));
    if (!fwd && a.items_per_query > 1) {
        hipError_t e = fill_words_async(dq, 0u, n_query * (n_part + 1) * dd, st);
        if (e != hipSuccess) return fail(static_cast<int>(e), "memset d_query: %s", hipGetErrorString(e));
    }
    int rc;
#define BESS_AFF(T, V)                                                                                   \
    (n_part == 1 ? aff_pt_by_it<T, V, 1>(it, d->norm_p, fwd, a, out, d_out, ld, dq, dn, st)              \
                 : aff_pt_by_it<T, V, 2>(it, d->norm_p, fwd, a, out, d_out, ld, dq, dn, st))
    if (d->dtype == BESS_F32) rc = vec == 4 ? BESS_AFF(float, 4) : BESS_AFF(float, 1);
    else rc = vec == 4 ? BESS_AFF(half_t, 4) : BESS_AFF(half_t, 1);
#undef BESS_AFF
    if (rc) return rc;
    return check_launch(fwd ? "neg_score_pertriple_fwd (affine)" : "neg_score_pertriple_bwd (affine)");
}

// K9 for the per-triple negatives of the own shard (the affine counterpart of
// k_pertriple_grad_segments): one 16-lane group per unique destination row.  The row and
// its part norms are loaded once; every reference (q, k) of the segment contributes
// d score / d c_p = s * U_p with s recomputed from query[q] and d_out[q, k]; the
// normalisation backward is linear, so it is applied once to the summed gradient.
struct AffSegArgs {
    const float* query;          // [n_query, (NPART + 1) * d]
    const void* table;
    const float* d_out;
    int64_t ld_dout;
    const int32_t* refs;
    const int32_t* seg_rows;
    const int32_t* seg_offsets;
    const int32_t* n_seg;
    int n_neg;
    int d;
    int nch;
    int normalize;
    const int32_t* long_segs;    // optional: rows with more than BESS_SEGMENT_CAP references (segments.hip)
    float p;                     // the norm of the P == 2 kernels (any p != 1)
};

// dc += contributions of references [r0, r1) of one destination row (ev, inv: the row and its part norms)
template <int VEC, int IT, int NPART, int P>
__device__ __forceinline__ void aff_seg_accumulate(const AffSegArgs& a, int g, const float (&ev)[NPART][IT][VEC],
                                                   const float (&inv)[NPART], int r0, int r1,
                                                   float (&dc)[NPART][IT][VEC]) {
    for (int r = r0; r < r1; ++r) {
            const int ref = a.refs[r];
            const int q = ref / a.n_neg;
            const int k = ref - q * a.n_neg;
            float go = -a.d_out[q * a.ld_dout + k];
            float qv[NPART + 1][IT][VEC];
            aff_load_query<VEC, IT, NPART>(a.query + static_cast<int64_t>(q) * (NPART + 1) * a.d, g, a.d, a.nch, qv);
            if (P == 2) {
                float ss = 0.f;
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const float dlt = aff_delta<VEC, IT, NPART>(qv, ev, inv, it, v);
                        ss += lp_term(dlt, a.p);
                    }
                ss = row16_allreduce_sum(ss);
                go *= lp_inv(lp_root(ss, a.p), a.p);
            }
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float dlt = aff_delta<VEC, IT, NPART>(qv, ev, inv, it, v);
                    const float sc = (P == 1) ? go * sgnf(dlt) : go * lp_dterm(dlt, a.p);
#pragma unroll
                    for (int p = 0; p < NPART; ++p) dc[p][it][v] = fmaf(sc, qv[p][it][v], dc[p][it][v]);
                }
        }
}

// summed d score / d c_p of a row -> gradient of the stored row (normalisation backward, once) -> output
template <typename T, int VEC, int IT, int NPART>
__device__ __forceinline__ void aff_seg_finish(const AffSegArgs& a, int g, const float (&ev)[NPART][IT][VEC],
                                               const float (&inv)[NPART], const float (&dc)[NPART][IT][VEC],
                                               int64_t seg, int64_t row, float* __restrict__ grad_seg, T* table_rw,
                                               float lr) {
    const int W = NPART * a.d;
#pragma unroll
        for (int p = 0; p < NPART; ++p) {
            float dot = 0.f;
            if (a.normalize) {
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) dot = fmaf(ev[p][it][v] * inv[p], dc[p][it][v], dot);
                dot = row16_allreduce_sum(dot);
                if (inv[p] >= 1.f / NORM_EPS) dot = 0.f;
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = g + 16 * it;
                if (c < a.nch) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const float hat = ev[p][it][v] * inv[p];
                        const float de = a.normalize ? inv[p] * (dc[p][it][v] - hat * dot) : dc[p][it][v];
                        const int64_t o = p * a.d + c * VEC + v;
                        if (grad_seg) grad_seg[seg * W + o] = de;
                        else table_rw[row * W + o] = static_cast<T>(ev[p][it][v] - lr * de);
                    }
                }
            }
        }
}

template <typename T, int VEC, int IT, int NPART, int P>
__global__ __launch_bounds__(256) void k_aff_grad_segments(AffSegArgs a, float* __restrict__ grad_seg, T* table_rw,
                                                           float lr) {
    const int lane = threadIdx.x & 63;
    const int g = lane & 15;
    const int n_seg = *a.n_seg;
    const int64_t group0 = (blockIdx.x * 256ll + threadIdx.x) >> 4;
    const int64_t n_group = (gridDim.x * 256ll) >> 4;
    const T* table = static_cast<const T*>(a.table);
    const int W = NPART * a.d;
    for (int64_t seg = group0; seg < n_seg; seg += n_group) {
        const int64_t row = a.seg_rows[seg];
        const int r0 = a.seg_offsets[seg], r1 = a.seg_offsets[seg + 1];
        if (a.long_segs && r1 - r0 > BESS_SEGMENT_CAP) continue;  // left to k_aff_long_segments
        float ev[NPART][IT][VEC], inv[NPART], dc[NPART][IT][VEC];
        aff_load_row<T, VEC, IT, NPART>(table + row * W, g, a.d, a.nch, ev);
        aff_inv_norm<VEC, IT, NPART>(ev, a.normalize, inv);
#pragma unroll
        for (int p = 0; p < NPART; ++p)
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int v = 0; v < VEC; ++v) dc[p][it][v] = 0.f;
        aff_seg_accumulate<VEC, IT, NPART, P>(a, g, ev, inv, r0, r1, dc);
        aff_seg_finish<T, VEC, IT, NPART>(a, g, ev, inv, dc, seg, row, grad_seg, table_rw, lr);
    }
}

// Rows with more than BESS_SEGMENT_CAP references, as k_long_segments (segments.hip): slices of the
// references are shared by all groups, the partial d score / d c_p meet in long_grad[li, :] through float
// atomics, and the group that adds a row's last slice reads the sum back, applies the normalisation
// backward, writes the row out and leaves the scratch row zero.
template <typename T, int VEC, int IT, int NPART, int P>
__global__ __launch_bounds__(256) void k_aff_long_segments(AffSegArgs a, float* __restrict__ long_grad,
                                                           int32_t* __restrict__ long_cnt, int32_t capacity,
                                                           float* __restrict__ grad_seg, T* table_rw, float lr) {
    const int lane = threadIdx.x & 63, g = lane & 15;
    const int64_t group0 = (blockIdx.x * 256ll + threadIdx.x) >> 4;
    const int64_t n_group = (gridDim.x * 256ll) >> 4;
    const int n_long = min(a.long_segs[0], capacity);
    const T* table = static_cast<const T*>(a.table);
    const int W = NPART * a.d;
    for (int li = 0; li < n_long; ++li) {
        const int seg = a.long_segs[1 + li];
        const int64_t row = a.seg_rows[seg];
        const int r0 = a.seg_offsets[seg], r1 = a.seg_offsets[seg + 1];
        const int parts = (r1 - r0 + BESS_SEGMENT_CAP - 1) / BESS_SEGMENT_CAP;
        if (group0 >= parts) continue;
        float ev[NPART][IT][VEC], inv[NPART];
        aff_load_row<T, VEC, IT, NPART>(table + row * W, g, a.d, a.nch, ev);
        aff_inv_norm<VEC, IT, NPART>(ev, a.normalize, inv);
        float* sum = long_grad + static_cast<int64_t>(li) * W;
        for (int64_t pt = group0; pt < parts; pt += n_group) {
            float dc[NPART][IT][VEC];
#pragma unroll
            for (int p = 0; p < NPART; ++p)
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) dc[p][it][v] = 0.f;
            const int rb = r0 + static_cast<int>(pt) * BESS_SEGMENT_CAP;
            aff_seg_accumulate<VEC, IT, NPART, P>(a, g, ev, inv, rb, min(r1, rb + BESS_SEGMENT_CAP), dc);
#pragma unroll
            for (int p = 0; p < NPART; ++p)
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int c = g + 16 * it;
                    if (c < a.nch) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) atomicAdd(sum + p * a.d + c * VEC + v, dc[p][it][v]);
                    }
                }
            __threadfence();
            int old = 0;
            if (g == 0) old = atomicAdd(long_cnt + li, 1);
            old = __shfl(old, lane & 48, 64);
            if ((old + 1) % parts != 0) continue;
            // the last arriver puts the counter back to zero: the scratch can serve the next launch as it is
            if (g == 0) __hip_atomic_store(long_cnt + li, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
#pragma unroll
            for (int p = 0; p < NPART; ++p)
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int c = g + 16 * it;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        dc[p][it][v] = 0.f;
                        if (c < a.nch) {
                            float* sp = sum + p * a.d + c * VEC + v;
                            dc[p][it][v] = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(sp, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            aff_seg_finish<T, VEC, IT, NPART>(a, g, ev, inv, dc, seg, row, grad_seg, table_rw, lr);
        }
    }
}

template <typename T, int VEC, int NPART>
static int aff_seg_by_it(int it, int p, const AffSegArgs& a, float* grad_seg, void* rw, float lr, unsigned grid,
                         hipStream_t st, float* long_grad = nullptr, int32_t* long_cnt = nullptr, int32_t cap = 0) {
    T* t = static_cast<T*>(rw);
#define BESS_AFS(ITV)                                                                                              \
    (long_grad ? (p == 1 ? k_aff_long_segments<T, VEC, ITV, NPART, 1><<<grid, 256, 0, st>>>(a, long_grad, long_cnt, cap, grad_seg, t, lr) \
                         : k_aff_long_segments<T, VEC, ITV, NPART, 2><<<grid, 256, 0, st>>>(a, long_grad, long_cnt, cap, grad_seg, t, lr)) \
               : (p == 1 ? k_aff_grad_segments<T, VEC, ITV, NPART, 1><<<grid, 256, 0, st>>>(a, grad_seg, t, lr)     \
                         : k_aff_grad_segments<T, VEC, ITV, NPART, 2><<<grid, 256, 0, st>>>(a, grad_seg, t, lr)))
    if (it <= 1) BESS_AFS(1);
    else if (it <= 2) BESS_AFS(2);
    else if (it <= 4) BESS_AFS(4);
    else if (it <= 8) BESS_AFS(8);
    else return fail(BESS_EUNSUPPORTED, "affine scorers: part of %d scalars too wide", a.d);
#undef BESS_AFS
    return BESS_OK;
}

int affine_grad_segments(const bess_model_desc* d, const float* query, void* table, int64_t n_neg,
                         const float* d_out, int64_t ld_dout, const int32_t* refs_sorted, const int32_t* seg_rows,
                         const int32_t* seg_offsets, const int32_t* n_seg, int64_t max_seg, float* grad_seg,
                         float fused_sgd_lr, const int32_t* long_segs, int64_t long_cap, float* long_grad,
                         int32_t* long_count, hipStream_t st) {
    const int n_part = d->reserved[0];
    const int dd = d->width / n_part;
    const int vec = (dd % 4 == 0) ? 4 : 1;
    AffSegArgs a{query, table, d_out, ld_dout, refs_sorted, seg_rows, seg_offsets, n_seg, static_cast<int>(n_neg),
                 dd, dd / vec, d->reserved[1] & 1, long_segs, static_cast<float>(d->norm_p)};
    const int it = static_cast<int>(ceil_div(a.nch, 16));
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(ceil_div(max_seg, 16), 256 * 16));
    int rc;
#define BESS_AFSD(T, V)                                                                                       \
    (n_part == 1 ? aff_seg_by_it<T, V, 1>(it, d->norm_p, a, grad_seg, table, fused_sgd_lr, grid, st)          \
                 : aff_seg_by_it<T, V, 2>(it, d->norm_p, a, grad_seg, table, fused_sgd_lr, grid, st))
    if (d->dtype == BESS_F32) rc = vec == 4 ? BESS_AFSD(float, 4) : BESS_AFSD(float, 1);
    else rc = vec == 4 ? BESS_AFSD(half_t, 4) : BESS_AFSD(half_t, 1);
#undef BESS_AFSD
    if (rc) return rc;
    if (long_segs) {  // the rows left out above, by all groups together (usually none)
        const int32_t cap = static_cast<int32_t>(long_cap);
#define BESS_AFSL(T, V)                                                                                                  \
    (n_part == 1 ? aff_seg_by_it<T, V, 1>(it, d->norm_p, a, grad_seg, table, fused_sgd_lr, 1024, st, long_grad, long_count, cap) \
                 : aff_seg_by_it<T, V, 2>(it, d->norm_p, a, grad_seg, table, fused_sgd_lr, 1024, st, long_grad, long_count, cap))
        if (d->dtype == BESS_F32) rc = vec == 4 ? BESS_AFSL(float, 4) : BESS_AFSL(float, 1);
        else rc = vec == 4 ? BESS_AFSL(half_t, 4) : BESS_AFSL(half_t, 1);
#undef BESS_AFSL
        if (rc) return rc;
    }
    return check_launch("neg_pertriple_grad_segments (affine)");
}

// ---------------------------------------------------------------------------
// shared negatives: 64 x 64 tile, dense fp32 candidates C [N, NPART * d] (already normalised)
constexpr int AKT = 16;
constexpr int ALDP = 68;

// tile[v][k][m] = src[(m0 + m) * ld + v * d + k0 + k], rows clamped, k beyond d zero-filled
template <int NV>
__device__ __forceinline__ void aff_stage_fetch(const float* __restrict__ src, int64_t n, int64_t ld, int64_t m0, int d,
                                                int k0, float (&v)[NV][4]) {
    const int m = threadIdx.x >> 2, kc = (threadIdx.x & 3) * 4;
    const float* rp = src + min(m0 + m, n - 1) * ld + k0 + kc;
    const bool vec_ok = (d & 3) == 0 && k0 + kc + 3 < d;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        if (vec_ok) {
            VecLoad<float, 4>::load(rp + p * d, v[p]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[p][i] = (k0 + kc + i < d) ? rp[p * d + i] : 0.f;
        }
    }
}
template <int NV>
__device__ __forceinline__ void aff_stage_store(float (*tile)[AKT][ALDP], const float (&v)[NV][4]) {
    const int m = threadIdx.x >> 2, kc = (threadIdx.x & 3) * 4;
#pragma unroll
    for (int p = 0; p < NV; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) tile[p][kc + i][m] = v[p][i];
}

template <int NPART, int P>
__global__ __launch_bounds__(256) void k_aff_shared_fwd(const float* __restrict__ Q, int64_t S, const float* __restrict__ C,
                                                        int64_t N, int d, float* __restrict__ out, int64_t ld_out, float pf) {
    __shared__ __attribute__((aligned(16))) float Qs[NPART + 1][AKT][ALDP];
    __shared__ __attribute__((aligned(16))) float Cs[NPART][AKT][ALDP];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t q0 = static_cast<int64_t>(blockIdx.y) * 64, j0 = static_cast<int64_t>(blockIdx.x) * 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    float qv[NPART + 1][4], cv[NPART][4];
    aff_stage_fetch<NPART + 1>(Q, S, static_cast<int64_t>(NPART + 1) * d, q0, d, 0, qv);
    aff_stage_fetch<NPART>(C, N, static_cast<int64_t>(NPART) * d, j0, d, 0, cv);
    for (int k0 = 0; k0 < d; k0 += AKT) {
        aff_stage_store<NPART + 1>(Qs, qv);
        aff_stage_store<NPART>(Cs, cv);
        __syncthreads();
        if (k0 + AKT < d) {
            aff_stage_fetch<NPART + 1>(Q, S, static_cast<int64_t>(NPART + 1) * d, q0, d, k0 + AKT, qv);
            aff_stage_fetch<NPART>(C, N, static_cast<int64_t>(NPART) * d, j0, d, k0 + AKT, cv);
        }
#pragma unroll
        for (int k = 0; k < AKT; ++k) {
            float u[NPART + 1][4], c[NPART][4];
#pragma unroll
            for (int p = 0; p <= NPART; ++p) {
                const float4 t = *reinterpret_cast<const float4*>(&Qs[p][k][ty * 4]);
                u[p][0] = t.x; u[p][1] = t.y; u[p][2] = t.z; u[p][3] = t.w;
            }
#pragma unroll
            for (int p = 0; p < NPART; ++p) {
                const float4 t = *reinterpret_cast<const float4*>(&Cs[p][k][tx * 4]);
                c[p][0] = t.x; c[p][1] = t.y; c[p][2] = t.z; c[p][3] = t.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float dlt = u[NPART][i];
#pragma unroll
                    for (int p = NPART - 1; p >= 0; --p) dlt = fmaf(u[p][i], c[p][j], dlt);
                    if (P == 1) acc[i][j] += fabsf(dlt);
                    else acc[i][j] += lp_term(dlt, pf);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t q = q0 + ty * 4 + i;
        if (q >= S) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t jj = j0 + tx * 4 + j;
            if (jj < N) out[q * ld_out + jj] = -(P == 2 ? lp_root(acc[i][j], pf) : acc[i][j]);
        }
    }
}

// Backward tile kernel.  X = the side whose gradient is produced (rows a), Y = the other side (rows b,
// the reduction).  XQ: X is the query side ([U | V | R], NPART + 1 vectors, gradient for all of them),
// otherwise X is the candidate side (NPART vectors).  Output tile: 64 rows a x 64 columns w of d.
//   t(a, b, w) = coef(a, b) * h(delta(a, b, w)),  h = sgn (p = 1) | identity (p = 2, coef = g / out)
//   XQ : dU += t * c1, dV += t * c2, dR += t          !XQ: dc1 += t * U, dc2 += t * V
template <int NPART, int P, bool XQ>
__global__ __launch_bounds__(256) void k_aff_shared_bwd(const float* __restrict__ X, int64_t nx, const float* __restrict__ Y,
                                                        int64_t ny, int d, const float* __restrict__ d_out, int64_t sa,
                                                        int64_t sb, const float* __restrict__ out, int64_t oa, int64_t ob,
                                                        float* __restrict__ dX, int64_t b_chunk, float pf) {
    constexpr int NVX = XQ ? NPART + 1 : NPART;
    constexpr int NVY = XQ ? NPART : NPART + 1;
    __shared__ __attribute__((aligned(16))) float Ks[AKT][ALDP];        // coefficient [b][a]
    __shared__ __attribute__((aligned(16))) float Ys[NVY][AKT][ALDP];   // [v][b][w]
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int t = threadIdx.x;
    const int64_t a0 = static_cast<int64_t>(blockIdx.y) * 64;
    const int w0 = blockIdx.x * 64;
    const int64_t ldx = static_cast<int64_t>(NVX) * d, ldy = static_cast<int64_t>(NVY) * d;
    float xv[NVX][4][4], acc[NVX][4][4];
#pragma unroll
    for (int p = 0; p < NVX; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t a = a0 + ty * 4 + i;
                const int w = w0 + tx * 4 + j;
                acc[p][i][j] = 0.f;
                xv[p][i][j] = (a < nx && w < d) ? X[a * ldx + p * d + w] : 0.f;
            }
    const int64_t b_lo = static_cast<int64_t>(blockIdx.z) * b_chunk;
    const int64_t b_hi = min(b_lo + b_chunk, ny);
    if (b_lo >= b_hi) return;
    const int yb = t >> 4, ywc = (t & 15) * 4;
    float kv[4], yv[NVY][4];
    auto fetch = [&](int64_t b0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int bb, al;
            if (sb == 1) { bb = t & 15; al = (t >> 4) + 16 * i; }
            else { al = t & 63; bb = (t >> 6) + 4 * i; }
            const int64_t a = a0 + al;
            const bool ok = a < nx && b0 + bb < b_hi;
            const int64_t ac = min(a, nx - 1), bc = min(b0 + bb, b_hi - 1);
            const float g = d_out[ac * sa + bc * sb];
            float c;
            if (P == 2) {
                const float o = out[ac * oa + bc * ob];
                c = -g * lp_inv(-o, pf);          // -g * ||delta||^(1 - p),  out = -||delta||   (p = 2: g / out)
            } else {
                c = -g;
            }
            kv[i] = ok ? c : 0.f;
        }
        const float* rp = Y + min(b0 + yb, b_hi - 1) * ldy;
#pragma unroll
        for (int p = 0; p < NVY; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int w = w0 + ywc + i;
                yv[p][i] = (w < d) ? rp[p * d + w] : 0.f;
            }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (sb == 1) Ks[t & 15][(t >> 4) + 16 * i] = kv[i];
            else Ks[(t >> 6) + 4 * i][t & 63] = kv[i];
        }
#pragma unroll
        for (int p = 0; p < NVY; ++p)
            *reinterpret_cast<float4*>(&Ys[p][yb][ywc]) = make_float4(yv[p][0], yv[p][1], yv[p][2], yv[p][3]);
    };
    fetch(b_lo);
    for (int64_t b0 = b_lo; b0 < b_hi; b0 += AKT) {
        stash();
        __syncthreads();
        if (b0 + AKT < b_hi) fetch(b0 + AKT);
#pragma unroll 4
        for (int k = 0; k < AKT; ++k) {
            const float4 c4 = *reinterpret_cast<const float4*>(&Ks[k][ty * 4]);
            const float c[4] = {c4.x, c4.y, c4.z, c4.w};
            float y[NVY][4];
#pragma unroll
            for (int p = 0; p < NVY; ++p) {
                const float4 t4 = *reinterpret_cast<const float4*>(&Ys[p][k][tx * 4]);
                y[p][0] = t4.x; y[p][1] = t4.y; y[p][2] = t4.z; y[p][3] = t4.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // delta = U c1 + V c2 + R with (U, V, R) on the query side, (c1, c2) on the candidate side
                    float dlt;
                    if (XQ) {
                        dlt = xv[NPART][i][j];
#pragma unroll
                        for (int p = NPART - 1; p >= 0; --p) dlt = fmaf(xv[p][i][j], y[p][j], dlt);
                    } else {
                        dlt = y[NPART][j];
#pragma unroll
                        for (int p = NPART - 1; p >= 0; --p) dlt = fmaf(y[p][j], xv[p][i][j], dlt);
                    }
                    const float tt = (P == 1) ? c[i] * sgn_prescaled(dlt * SGN_PRESCALE) : c[i] * lp_dterm(dlt, pf);
#pragma unroll
                    for (int p = 0; p < NPART; ++p) acc[p][i][j] = fmaf(tt, y[p][j], acc[p][i][j]);
                    if (XQ) acc[NPART][i][j] += tt;
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < NVX; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t a = a0 + ty * 4 + i;
            if (a >= nx) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int w = w0 + tx * 4 + j;
                if (w < d) {
                    float* o = dX + a * ldx + p * d + w;
                    if (gridDim.z == 1) *o = acc[p][i][j];
                    else unsafeAtomicAdd(o, acc[p][i][j]);
                }
            }
        }
}

template <int NPART, int P, bool XQ>
static int aff_bwd_launch(const float* X, int64_t nx, const float* Y, int64_t ny, int d, const float* d_out, int64_t sa,
                          int64_t sb, const float* out, int64_t oa, int64_t ob, float* dX, hipStream_t st, float pf) {
    constexpr int NVX = XQ ? NPART + 1 : NPART;
    const int64_t tiles = ceil_div(d, 64) * ceil_div(nx, 64);
    int64_t split = 1;
    while (tiles * split < 1024 && ceil_div(ny, split * 2) >= 4 * AKT) split *= 2;
    int64_t chunk = ceil_div(ceil_div(ny, split), AKT) * AKT;
    split = ceil_div(ny, chunk);
    if (split > 1) {
        hipError_t e = fill_words_async(dX, 0u, nx * NVX * d, st);
        if (e != hipSuccess) return fail(static_cast<int>(e), "memset: %s", hipGetErrorString(e));
    }
    const dim3 grid(static_cast<unsigned>(ceil_div(d, 64)), static_cast<unsigned>(ceil_div(nx, 64)),
                    static_cast<unsigned>(split));
    k_aff_shared_bwd<NPART, P, XQ><<<grid, 256, 0, st>>>(X, nx, Y, ny, d, d_out, sa, sb, out, oa, ob, dX, chunk, pf);
    return BESS_OK;
}

int affine_shared_fwd(const bess_model_desc* d, const float* query, int64_t S, const float* cand, int64_t N, float* out,
                      int64_t ld, hipStream_t st) {
    const int n_part = d->reserved[0];
    const int dd = d->width / n_part;
    const float pf = static_cast<float>(d->norm_p);
    const dim3 grid(static_cast<unsigned>(ceil_div(N, 64)), static_cast<unsigned>(ceil_div(S, 64)));
    if (n_part == 1) {
        if (d->norm_p == 1) k_aff_shared_fwd<1, 1><<<grid, 256, 0, st>>>(query, S, cand, N, dd, out, ld, pf);
        else k_aff_shared_fwd<1, 2><<<grid, 256, 0, st>>>(query, S, cand, N, dd, out, ld, pf);
    } else {
        if (d->norm_p == 1) k_aff_shared_fwd<2, 1><<<grid, 256, 0, st>>>(query, S, cand, N, dd, out, ld, pf);
        else k_aff_shared_fwd<2, 2><<<grid, 256, 0, st>>>(query, S, cand, N, dd, out, ld, pf);
    }
    return check_launch("neg_score_shared_fwd (affine)");
}

int affine_shared_bwd(const bess_model_desc* d, const float* query, int64_t S, const float* cand, int64_t N,
                      const float* out, int64_t ld_out, const float* d_out, int64_t ld_dout, float* d_query,
                      float* d_cand, hipStream_t st) {
    const int n_part = d->reserved[0];
    const int dd = d->width / n_part;
    const float pf = static_cast<float>(d->norm_p);
    int rc;
#define BESS_AFFB(NP, PP)                                                                                          \
    do {                                                                                                           \
        rc = aff_bwd_launch<NP, PP, true>(query, S, cand, N, dd, d_out, ld_dout, 1, out, ld_out, 1, d_query, st, pf);  \
        if (!rc) rc = aff_bwd_launch<NP, PP, false>(cand, N, query, S, dd, d_out, 1, ld_dout, out, 1, ld_out,      \
                                                    d_cand, st, pf);                                               \
    } while (0)
    if (n_part == 1) {
        if (d->norm_p == 1) BESS_AFFB(1, 1); else BESS_AFFB(1, 2);
    } else {
        if (d->norm_p == 1) BESS_AFFB(2, 1); else BESS_AFFB(2, 2);
    }
#undef BESS_AFFB
    if (rc) return rc;
    return check_launch("neg_score_shared_bwd (affine)");
}

}  // namespace bess

using namespace bess;

extern "C" int bess_normalize_rows(int32_t dtype, const void* base, const int32_t* idx, int64_t n_rows, int32_t width,
                                   int32_t n_part, int32_t normalize, float* out, float* inv_norm, void* stream) {
    BESS_REQUIRE(dtype == BESS_F32 || dtype == BESS_F16, "normalize_rows: unknown dtype %d", dtype);
    BESS_REQUIRE(n_rows >= 0 && width > 0 && n_part > 0 && width % n_part == 0, "normalize_rows: bad shape");
    if (n_rows == 0) return BESS_OK;
    BESS_REQUIRE(base && out, "normalize_rows: NULL pointer");
    const unsigned blocks = static_cast<unsigned>(ceil_div(n_rows, 4));
    if (dtype == BESS_F32)
        k_normalize_rows<float><<<blocks, 256, 0, as_stream(stream)>>>(static_cast<const float*>(base), idx, n_rows,
                                                                       width / n_part, n_part, normalize, out, inv_norm);
    else
        k_normalize_rows<half_t><<<blocks, 256, 0, as_stream(stream)>>>(static_cast<const half_t*>(base), idx, n_rows,
                                                                        width / n_part, n_part, normalize, out, inv_norm);
    return check_launch("normalize_rows");
}

extern "C" int bess_normalize_rows_bwd(const float* hat, const float* inv_norm, const float* d_hat, int64_t n_rows,
                                       int32_t width, int32_t n_part, float* d_rows, void* stream) {
    BESS_REQUIRE(n_rows >= 0 && width > 0 && n_part > 0 && width % n_part == 0, "normalize_rows_bwd: bad shape");
    if (n_rows == 0) return BESS_OK;
    BESS_REQUIRE(hat && inv_norm && d_hat && d_rows, "normalize_rows_bwd: NULL pointer");
    k_normalize_rows_bwd<<<static_cast<unsigned>(ceil_div(n_rows, 4)), 256, 0, as_stream(stream)>>>(
        hat, inv_norm, d_hat, n_rows, width / n_part, n_part, d_rows);
    return check_launch("normalize_rows_bwd");
}

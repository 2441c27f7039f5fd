// K5 on gfx950: per-triple negatives, the HBM-bound hot kernel of the BESS step.
//
//   out[q, k] = sign * reduce_w f(query[q, w], E[neg_idx[q, k], w])       k < n_neg
//
// (reference `reduce_embedding(q.unsqueeze(1) o N)`, scoring.py:199 / 254, fed by
//  the gather `self.entity_embedding[gather_idx]`, bess.py:332-337 - here the
//  gather is fused: negative rows are read straight from the shard, exactly
//  once, and nothing of shape [S, N, W] is ever materialised.)
//
// Roofline: one gathered row (W*sz bytes: 2 KiB for ComplEx d=256 fp32) per
// scored triple against 2-3 flops per scalar -> HBM bound; algorithmic bytes =
// n_query*n_neg*(W*sz + 4 idx + 4 out) + n_query*W*4.
//
// Mapping (CDNA4, wave64): a wave owns one (query, block of NB negatives) work
// item.  Its four 16-lane DPP rows each stream a different negative row, lane g
// of a row reading the 16-byte chunks g, g+16, g+32, ... of that table row
// (256 contiguous bytes per DPP row per load instruction -> two full 128 B
// lines).  The query lives in registers in the same chunk layout, so the inner
// loop is loads + FMAs only; the cross-lane reduction is 4 DPP row-rotate adds
// shared by the four rows in flight.  UNROLL row-groups are issued back to back
// to keep >= 8 x 16 B loads per lane in flight (the guide's recipe for ~5.7 TB/s
// random-row gathers: "4 rows in flight per wave, 16 waves per CU").
#include "common.h"

namespace bess {

struct NegPtArgs {
    const float* query;
    const void* base;
    const int32_t* idx;
    int64_t n_query;
    int n_neg;
    int W;
    int nch;  // chunks (of VEC scalars) per row
    int nb;   // negatives per work item
    int items_per_query;
    float sign;
};

template <typename T, int VEC, int IT, int RED, int UNROLL>
__global__ __launch_bounds__(256) void k_neg_pertriple_fwd(NegPtArgs a, float* __restrict__ out,
                                                           int64_t ld_out) {
    const int lane = threadIdx.x & 63;
    const int g = lane & 15;
    const int sub = lane >> 4;
    const int64_t item = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (item >= a.n_query * a.items_per_query) return;
    const int64_t q = item / a.items_per_query;
    const int k0 = static_cast<int>(item - q * a.items_per_query) * a.nb;
    const int k1 = min(k0 + a.nb, a.n_neg);

    // query chunks -> registers (f32, chunk c covers scalars [c*VEC, c*VEC+VEC))
    float qv[IT][VEC];
    const float* qp = a.query + q * a.W;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = g + 16 * it;
        if (c < a.nch) {
            VecLoad<float, VEC>::load(qp + c * VEC, qv[it]);
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) qv[it][v] = 0.f;
        }
    }

    const T* base = static_cast<const T*>(a.base);
    const int32_t* idx = a.idx + q * a.n_neg;
    float* orow = out + q * ld_out;

    for (int kb = k0; kb < k1; kb += 4 * UNROLL) {  // kb is wave-uniform
        float ev[UNROLL][IT][VEC];
        bool valid[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = kb + sub + 4 * u;
            valid[u] = k < k1;
            const int ks = valid[u] ? k : (k1 - 1);  // keep the wave converged
            const T* rp = base + static_cast<int64_t>(idx[ks]) * a.W;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = g + 16 * it;
                if (c < a.nch) {
                    VecLoad<T, VEC>::load(rp + c * VEC, ev[u][it]);
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) ev[u][it][v] = 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            float acc = 0.f;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    if (RED == RED_DOT) {
                        acc = fmaf(qv[it][v], ev[u][it][v], acc);
                    } else if (RED == RED_L1) {
                        acc += fabsf(qv[it][v] - ev[u][it][v]);
                    } else {
                        const float dlt = qv[it][v] - ev[u][it][v];
                        acc = fmaf(dlt, dlt, acc);
                    }
                }
            }
            acc = row16_allreduce_sum(acc);
            if (RED == RED_L2) acc = sqrtf(acc);
            if (g == 0 && valid[u]) orow[kb + sub + 4 * u] = a.sign * acc;
        }
    }
}

// Backward: d_neg[(q, k), :] = g * df/de ; d_query[q, :] += sum_k g * df/dq  with
// g = d_out[q, k].  f = q.e (DOT) | -||q-e||_1 | -||q-e||_2 (sign folded in).
// Same mapping as forward; the per-item partial of d_query is added atomically
// (items_per_query partials per query).
template <typename T, int VEC, int IT, int RED>
__global__ __launch_bounds__(256) void k_neg_pertriple_bwd(NegPtArgs a,
                                                           const float* __restrict__ d_out,
                                                           int64_t ld_dout,
                                                           float* __restrict__ d_query,
                                                           float* __restrict__ d_neg) {
    const int lane = threadIdx.x & 63;
    const int g = lane & 15;
    const int sub = lane >> 4;
    const int64_t item = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (item >= a.n_query * a.items_per_query) return;
    const int64_t q = item / a.items_per_query;
    const int k0 = static_cast<int>(item - q * a.items_per_query) * a.nb;
    const int k1 = min(k0 + a.nb, a.n_neg);

    float qv[IT][VEC], dq[IT][VEC];
    const float* qp = a.query + q * a.W;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = g + 16 * it;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            qv[it][v] = 0.f;
            dq[it][v] = 0.f;
        }
        if (c < a.nch) VecLoad<float, VEC>::load(qp + c * VEC, qv[it]);
    }
    const T* base = static_cast<const T*>(a.base);
    const int32_t* idx = a.idx + q * a.n_neg;

    for (int kb = k0; kb < k1; kb += 4) {
        const int k = kb + sub;
        const bool valid = k < k1;
        const int ks = valid ? k : (k1 - 1);
        const T* rp = base + static_cast<int64_t>(idx[ks]) * a.W;
        float ev[IT][VEC];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = g + 16 * it;
#pragma unroll
            for (int v = 0; v < VEC; ++v) ev[it][v] = 0.f;
            if (c < a.nch) VecLoad<T, VEC>::load(rp + c * VEC, ev[it]);
        }
        float go = valid ? a.sign * d_out[q * ld_dout + ks] : 0.f;
        if (RED == RED_L2) {
            float ss = 0.f;
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float dlt = qv[it][v] - ev[it][v];
                    ss = fmaf(dlt, dlt, ss);
                }
            ss = row16_allreduce_sum(ss);
            go = ss > 0.f ? go / sqrtf(ss) : 0.f;
        }
        float* dn = d_neg + (q * a.n_neg + ks) * a.W;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = g + 16 * it;
            float de[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float dqe;  // d f / d q_w  (d f / d e_w = -dqe for distances)
                if (RED == RED_DOT) {
                    dqe = go * ev[it][v];
                    de[v] = go * qv[it][v];
                } else if (RED == RED_L1) {
                    dqe = go * sgnf(qv[it][v] - ev[it][v]);
                    de[v] = -dqe;
                } else {
                    dqe = go * (qv[it][v] - ev[it][v]);
                    de[v] = -dqe;
                }
                dq[it][v] += dqe;
            }
            if (d_neg && valid && c < a.nch) {  // d_neg == NULL: only d_query is wanted
#pragma unroll
                for (int v = 0; v < VEC; ++v) dn[c * VEC + v] = de[v];
            }
        }
    }
    // combine the four DPP rows, then one atomic per scalar per work item
    float* dqp = d_query + q * a.W;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = g + 16 * it;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float x = dq[it][v];
            x += __shfl_xor(x, 16, 64);
            x += __shfl_xor(x, 32, 64);
            if (sub == 0 && c < a.nch) {
                if (a.items_per_query == 1) dqp[c * VEC + v] = x;
                else unsafeAtomicAdd(dqp + c * VEC + v, x);
            }
        }
    }
}

// ---- host dispatch ---------------------------------------------------------
template <typename T, int VEC, int IT, int RED>
static void launch_fwd(const NegPtArgs& a, float* out, int64_t ld, hipStream_t st) {
    const int64_t items = a.n_query * a.items_per_query;
    constexpr int EPL = IT * VEC;  // scalars per lane per row
    constexpr int UNROLL = EPL <= 16 ? 4 : (EPL <= 32 ? 2 : 1);
    k_neg_pertriple_fwd<T, VEC, IT, RED, UNROLL><<<ceil_div(items, 4), 256, 0, st>>>(a, out, ld);
}
template <typename T, int VEC, int IT, int RED>
static void launch_bwd(const NegPtArgs& a, const float* d_out, int64_t ld, float* dq, float* dn,
                       hipStream_t st) {
    const int64_t items = a.n_query * a.items_per_query;
    k_neg_pertriple_bwd<T, VEC, IT, RED><<<ceil_div(items, 4), 256, 0, st>>>(a, d_out, ld, dq, dn);
}

template <typename T, int VEC, int IT>
static void by_red(int red, bool fwd, const NegPtArgs& a, float* out, const float* d_out, int64_t ld,
                   float* dq, float* dn, hipStream_t st) {
    switch (red) {
        case RED_DOT:
            fwd ? launch_fwd<T, VEC, IT, RED_DOT>(a, out, ld, st)
                : launch_bwd<T, VEC, IT, RED_DOT>(a, d_out, ld, dq, dn, st);
            break;
        case RED_L1:
            fwd ? launch_fwd<T, VEC, IT, RED_L1>(a, out, ld, st)
                : launch_bwd<T, VEC, IT, RED_L1>(a, d_out, ld, dq, dn, st);
            break;
        default:
            fwd ? launch_fwd<T, VEC, IT, RED_L2>(a, out, ld, st)
                : launch_bwd<T, VEC, IT, RED_L2>(a, d_out, ld, dq, dn, st);
    }
}

template <typename T, int VEC>
static int by_it(int it, int red, bool fwd, const NegPtArgs& a, float* out, const float* d_out,
                 int64_t ld, float* dq, float* dn, hipStream_t st) {
    if (it <= 1) by_red<T, VEC, 1>(red, fwd, a, out, d_out, ld, dq, dn, st);
    else if (it <= 2) by_red<T, VEC, 2>(red, fwd, a, out, d_out, ld, dq, dn, st);
    else if (it <= 4) by_red<T, VEC, 4>(red, fwd, a, out, d_out, ld, dq, dn, st);
    else if (it <= 8) by_red<T, VEC, 8>(red, fwd, a, out, d_out, ld, dq, dn, st);
    else if (it <= 16) by_red<T, VEC, 16>(red, fwd, a, out, d_out, ld, dq, dn, st);
    else return fail(BESS_EUNSUPPORTED, "neg_score_pertriple: row of %d scalars too wide", a.W);
    return BESS_OK;
}

static int run(const bess_model_desc* d, bool fwd, const float* query, int64_t n_query,
               const void* neg_base, const int32_t* neg_idx, int64_t n_neg, float* out,
               const float* d_out, int64_t ld, float* dq, float* dn, void* stream) {
    if (int e = check_desc(d)) return e;
    BESS_REQUIRE(n_query >= 0 && n_neg >= 0 && n_neg < (1ll << 31), "neg_score_pertriple: bad sizes");
    if (n_query == 0 || n_neg == 0) return BESS_OK;
    BESS_REQUIRE(query && neg_base && neg_idx, "neg_score_pertriple: NULL pointer");
    BESS_REQUIRE(ld >= n_neg, "neg_score_pertriple: leading dimension %lld < n_neg %lld", (long long)ld,
                 (long long)n_neg);
    if (d->scorer == BESS_BOXE)
        return boxe_negatives(d, fwd, false, query, n_query, neg_base, neg_idx, n_neg, out, d_out, ld, dq, dn,
                              as_stream(stream));
    if (d->scorer == BESS_AFFINE)
        return affine_pertriple(d, fwd, query, n_query, neg_base, neg_idx, n_neg, out, d_out, ld, dq, dn,
                                as_stream(stream));
    const int W = d->width;
    const int maxvec = d->dtype == BESS_F32 ? 4 : 8;
    // widest vector that divides the row: f32 {4,1}, f16 {8,2,1} scalars per lane load
    int vec = maxvec;
    if (W % vec) vec = (d->dtype == BESS_F16 && W % 2 == 0) ? 2 : 1;
    NegPtArgs a;
    a.query = query;
    a.base = neg_base;
    a.idx = neg_idx;
    a.n_query = n_query;
    a.n_neg = static_cast<int>(n_neg);
    a.W = W;
    a.nch = W / vec;
    // 64 negatives per work item: ~128 KiB of rows per wave at 2 KiB rows; shrink the
    // item when the launch would not fill 256 CUs x 16 waves
    int nb = 64;
    while (nb > 8 && n_query * ceil_div(n_neg, nb) < 256 * 16 * 2) nb >>= 1;
    a.nb = nb;
    a.items_per_query = static_cast<int>(ceil_div(n_neg, nb));
    a.sign = is_distance(d->scorer) ? -1.f : 1.f;
    const int it = static_cast<int>(ceil_div(a.nch, 16));
    const int red = reduce_of(d);
    hipStream_t st = as_stream(stream);
    if (!fwd && a.items_per_query > 1) {
        hipError_t e = hipMemsetAsync(dq, 0, sizeof(float) * n_query * W, st);
        if (e != hipSuccess) return fail(static_cast<int>(e), "memset d_query: %s", hipGetErrorString(e));
    }
    int rc = BESS_OK;
    if (d->dtype == BESS_F32) {
        if (vec == 4) rc = by_it<float, 4>(it, red, fwd, a, out, d_out, ld, dq, dn, st);
        else rc = by_it<float, 1>(it, red, fwd, a, out, d_out, ld, dq, dn, st);
    } else {
        if (vec == 8) rc = by_it<half_t, 8>(it, red, fwd, a, out, d_out, ld, dq, dn, st);
        else if (vec == 2) rc = by_it<half_t, 2>(it, red, fwd, a, out, d_out, ld, dq, dn, st);
        else rc = by_it<half_t, 1>(it, red, fwd, a, out, d_out, ld, dq, dn, st);
    }
    if (rc) return rc;
    return check_launch(fwd ? "neg_score_pertriple_fwd" : "neg_score_pertriple_bwd");
}

}  // namespace bess

extern "C" int bess_neg_score_pertriple_fwd(const bess_model_desc* d, const float* query,
                                            int64_t n_query, const void* neg_base,
                                            const int32_t* neg_idx, int64_t n_neg, float* out,
                                            int64_t ld_out, void* stream) {
    if (n_query > 0 && n_neg > 0 && !out) return bess::fail(BESS_EINVAL, "neg_score_pertriple_fwd: NULL out");
    return bess::run(d, true, query, n_query, neg_base, neg_idx, n_neg, out, nullptr, ld_out, nullptr,
                     nullptr, stream);
}

extern "C" int bess_neg_score_pertriple_bwd(const bess_model_desc* d, const float* query,
                                            int64_t n_query, const void* neg_base,
                                            const int32_t* neg_idx, int64_t n_neg,
                                            const float* d_out, int64_t ld_dout, float* d_query,
                                            float* d_neg, void* stream) {
    if (n_query > 0 && n_neg > 0 && !(d_out && d_query))
        return bess::fail(BESS_EINVAL, "neg_score_pertriple_bwd: NULL pointer");
    return bess::run(d, false, query, n_query, neg_base, neg_idx, n_neg, nullptr, d_out, ld_dout,
                     d_query, d_neg, stream);
}

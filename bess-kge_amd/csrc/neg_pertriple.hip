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
#include "loss_rows.h"

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
    int accum = 0;  // forward: add to the scores already there (a later column window of a wide row)
    float p = 2.f;  // the norm of the RED_L2 kernels (any p != 1)
    int dn_by_row = 0;  // backward: d_neg row of reference k is neg_idx[k] (BESS_FLAG_DNEG_BY_ROW), not q * n_neg + k
};

// Fused training forward (FUSE): besides the scores, accumulate the loss gradient wrt the query,
//     d_query[q] = sum_k dL/ds(q, k) * ds(q, k)/dq,
// in the same pass over the rows, so the backward never re-reads them.  dL/ds of all three
// losses has the form  C_q * g(s_k) * exp(beta * s_k) / sum_k' exp(beta * s_k')  (self-adversarial
// softmax weights, or the softmax of the sampled-softmax cross entropy; beta = 0 gives the
// uniform 1/N): an online-softmax accumulation (running max m, normaliser l, weighted sum
// acc - as in flash attention) needs no second pass.  Every work item writes its partial
// (m, l, acc[W]); k_combine_dq merges the items of a query and applies C_q / l.
struct FuseArgs {
    const float* pos;   // [n_query] positive scores (margin ranking); may be NULL otherwise
    int kind;           // BESS_LOSS_*
    float beta;         // adversarial_scale | 1 (ssce) | 0 (uniform weights)
    float margin;
    float shift;        // ssce: log(n_entity - 1) - log(N), added to the negative scores
    float* st_ml;       // [n_query, items, 2]
    float* st_acc;      // [n_query, items, W]
    const uint8_t* mask = nullptr;  // [mask_rows (1 | n_query), mask_cols] over the last mask_cols columns, or NULL
    int64_t mask_rows = 0;
    int mask_cols = 0, mask_from = 0;
};

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

template <typename T, int VEC, int IT, int RED, int UNROLL, bool FUSE>
__global__ __launch_bounds__(256) void k_neg_pertriple_fwd(NegPtArgs a, float* __restrict__ out,
                                                           int64_t ld_out, FuseArgs f) {
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
        load_chunk<float, VEC>(qp, g + 16 * it, a.nch, qv[it]);
    }

    const T* base = static_cast<const T*>(a.base);
    const int32_t* idx = a.idx + q * a.n_neg;
    float* orow = out + q * ld_out;
    // online-softmax state of this 16-lane row group (FUSE)
    float fm = -INFINITY, fl = 0.f, facc[IT][VEC];
    float fpos = 0.f;
    if (FUSE) {
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int v = 0; v < VEC; ++v) facc[it][v] = 0.f;
        if (f.kind == BESS_LOSS_MARGIN) fpos = f.pos[q];
    }

    // the row index of the *next* group of rows is fetched while the current rows are in
    // flight: no row load waits behind its own index load (matters most for short rows)
    int32_t nrow[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) nrow[u] = idx[min(k0 + sub + 4 * u, k1 - 1)];
    for (int kb = k0; kb < k1; kb += 4 * UNROLL) {  // kb is wave-uniform
        float ev[UNROLL][IT][VEC];
        bool valid[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = kb + sub + 4 * u;
            valid[u] = k < k1;
            const T* rp = base + static_cast<int64_t>(nrow[u]) * a.W;  // clamped: keeps the wave converged
            nrow[u] = idx[min(k + 4 * UNROLL, k1 - 1)];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                load_chunk<T, VEC>(rp, g + 16 * it, a.nch, ev[u][it]);
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
                        acc += lp_term(qv[it][v] - ev[u][it][v], a.p);
                    }
                }
            }
            acc = row16_allreduce_sum(acc);
            if (RED == RED_L2) acc = lp_root(acc, a.p);
            float sc = a.sign * acc;
            if (FUSE && f.mask && valid[u]) {
                // K7 inside the pass (the padding mask of triple-specific negatives): a masked-out candidate gets
                // BESS_BAD_NEGATIVE_SCORE added before it is stored and before it enters the softmax
                const int kcol = kb + sub + 4 * u - f.mask_from;
                const int64_t mrow = f.mask_rows == 1 ? 0 : q;
                if (kcol >= 0 && f.mask[mrow * f.mask_cols + kcol] == 0) sc += BESS_BAD_NEGATIVE_SCORE;
            }
            if (g == 0 && valid[u]) orow[kb + sub + 4 * u] = a.accum ? orow[kb + sub + 4 * u] + sc : sc;
            if (FUSE && valid[u]) {  // uniform within the 16-lane group
                const float z = f.beta * (sc + f.shift);
                const float m_new = fmaxf(fm, z);
                const float corr = expf(fm - m_new);
                const float e = expf(z - m_new);
                float gs = 1.f;
                if (f.kind == BESS_LOSS_LOGSIGMOID) gs = sigmoid_f(sc + f.margin);
                else if (f.kind == BESS_LOSS_MARGIN) gs = (sc - fpos + f.margin > 0.f) ? 1.f : 0.f;
                const float p = e * gs;
                fl = fmaf(fl, corr, e);
                fm = m_new;
                const float inv_norm = RED == RED_L2 ? lp_inv(acc, a.p) : 0.f;
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        float dsdq;  // d score / d query_w
                        if (RED == RED_DOT) dsdq = ev[u][it][v];
                        else if (RED == RED_L1) dsdq = -sgnf(qv[it][v] - ev[u][it][v]);
                        else dsdq = -lp_dterm(qv[it][v] - ev[u][it][v], a.p) * inv_norm;
                        facc[it][v] = fmaf(facc[it][v], corr, p * dsdq);
                    }
            }
        }
    }
    if (FUSE) {
        // merge the four row groups of the wave, then one partial per work item
        float m_all = fmaxf(fm, __shfl_xor(fm, 16, 64));
        m_all = fmaxf(m_all, __shfl_xor(m_all, 32, 64));
        const float sc = (fm == -INFINITY) ? 0.f : expf(fm - m_all);
        float l = fl * sc;
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int64_t slot = q * a.items_per_query + (item - q * a.items_per_query);
        if (lane == 0) {
            f.st_ml[slot * 2 + 0] = m_all;
            f.st_ml[slot * 2 + 1] = l;
        }
        float* ap = f.st_acc + slot * a.W;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = g + 16 * it;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float x = facc[it][v] * sc;
                x += __shfl_xor(x, 16, 64);
                x += __shfl_xor(x, 32, 64);
                if (sub == 0 && c < a.nch) ap[c * VEC + v] = x;
            }
        }
    }
}

// d loss / d query from the items' partials, one wave per query (loss_rows.h: combine_dq_row)
__global__ __launch_bounds__(256) void k_combine_dq(const float* __restrict__ st_ml, const float* __restrict__ st_acc,
                                                    int64_t n_query, int items, int W, int kind, float loss_scale,
                                                    const float* __restrict__ pos, const float* __restrict__ weight,
                                                    int64_t weight_len, const float* __restrict__ norm,
                                                    float* __restrict__ d_query) {
    const int64_t q = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (q >= n_query) return;
    combine_dq_row(st_ml, st_acc, q, items, W, kind, loss_scale, pos, weight, weight_len, norm, d_query + q * W, nullptr);
}

// Backward: d_neg[(q, k), :] = g * df/de ; d_query[q, :] += sum_k g * df/dq  with
// g = d_out[q, k].  f = q.e (DOT) | -||q-e||_1 | -||q-e||_2 (sign folded in).
// Same mapping and the same load discipline as the forward: UNROLL row groups issued back to back,
// unconditional loads at clamped rows, the row ids and score gradients of the next groups fetched
// while the current rows are in flight (the first version of this kernel loaded one row group per
// iteration behind its own index load: 675 us for the launch the forward does in 293).  The per-item
// partial of d_query is added atomically (items_per_query partials per query).
template <typename T, int VEC, int IT, int RED, int UNROLL>
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
#pragma unroll
        for (int v = 0; v < VEC; ++v) dq[it][v] = 0.f;
        load_chunk<float, VEC>(qp, g + 16 * it, a.nch, qv[it]);
    }
    const T* base = static_cast<const T*>(a.base);
    const int32_t* idx = a.idx + q * a.n_neg;
    const float* grow = d_out + q * ld_dout;

    int32_t nrow[UNROLL];
    float ngo[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int kk = min(k0 + sub + 4 * u, k1 - 1);
        nrow[u] = idx[kk];
        ngo[u] = grow[kk];
    }
    // d_query == NULL (it came out of the fused forward) and a bilinear scorer: d f / d e = g * q does not involve
    // the candidate rows - they are not read at all, the kernel only writes d_neg
    const bool need_rows = !(RED == RED_DOT && d_query == nullptr);
    for (int kb = k0; kb < k1; kb += 4 * UNROLL) {  // kb is wave-uniform
        float ev[UNROLL][IT][VEC], go[UNROLL];
        int ks[UNROLL];
        int32_t row_now[UNROLL];
        bool valid[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = kb + sub + 4 * u;
            valid[u] = k < k1;
            ks[u] = min(k, k1 - 1);
            const T* rp = base + static_cast<int64_t>(nrow[u]) * a.W;
            row_now[u] = nrow[u];
            go[u] = valid[u] ? a.sign * ngo[u] : 0.f;
            const int kn = min(k + 4 * UNROLL, k1 - 1);
            nrow[u] = idx[kn];
            ngo[u] = grow[kn];
            if (need_rows) {
#pragma unroll
                for (int it = 0; it < IT; ++it) load_chunk<T, VEC>(rp, g + 16 * it, a.nch, ev[u][it]);
            } else {
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) ev[u][it][v] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            float gg = go[u];
            if (RED == RED_L2) {
                float ss = 0.f;
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        ss += lp_term(qv[it][v] - ev[u][it][v], a.p);
                    }
                ss = row16_allreduce_sum(ss);
                gg *= lp_inv(lp_root(ss, a.p), a.p);
            }
            float* dn = d_neg + (a.dn_by_row ? static_cast<int64_t>(row_now[u]) : q * a.n_neg + ks[u]) * a.W;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = g + 16 * it;
                float de[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float dqe;  // d f / d q_w  (d f / d e_w = -dqe for distances)
                    if (RED == RED_DOT) {
                        dqe = gg * ev[u][it][v];
                        de[v] = gg * qv[it][v];
                    } else if (RED == RED_L1) {
                        dqe = gg * sgnf(qv[it][v] - ev[u][it][v]);
                        de[v] = -dqe;
                    } else {
                        dqe = gg * lp_dterm(qv[it][v] - ev[u][it][v], a.p);
                        de[v] = -dqe;
                    }
                    dq[it][v] += dqe;
                }
                if (d_neg && valid[u] && c < a.nch) {  // d_neg == NULL: only d_query is wanted
                    // written once, read much later (by C8 / the update): streamed past the caches - for fp32 tables
                    // (one 16-byte piece per lane and chunk: 645 vs 665 us at C2, 533 vs 658 at C3, 119 vs 155 at C1;
                    // the two pieces per lane of fp16 rows came out 12-16 % slower non-temporal)
                    if constexpr (VEC == 4) {
                        typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
                        for (int v = 0; v < VEC; v += 4) {
                            f4 o = {de[v], de[v + 1], de[v + 2], de[v + 3]};
                            __builtin_nontemporal_store(o, reinterpret_cast<f4*>(dn + c * VEC + v));
                        }
                    } else {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) dn[c * VEC + v] = de[v];
                    }
                }
            }
        }
    }
    if (!d_query) return;
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
static void launch_fwd(const NegPtArgs& a, float* out, int64_t ld, const FuseArgs* fuse, hipStream_t st) {
    const int64_t items = a.n_query * a.items_per_query;
    constexpr int EPL = IT * VEC;  // scalars per lane per row
    constexpr int UNROLL = EPL <= 16 ? 4 : (EPL <= 32 ? 2 : 1);
    if (fuse) {
        constexpr int FU = EPL <= 16 ? 2 : 1;  // the running d_query sum takes EPL more registers
        k_neg_pertriple_fwd<T, VEC, IT, RED, FU, true><<<ceil_div(items, 4), 256, 0, st>>>(a, out, ld, *fuse);
    } else {
        k_neg_pertriple_fwd<T, VEC, IT, RED, UNROLL, false><<<ceil_div(items, 4), 256, 0, st>>>(a, out, ld, FuseArgs{});
    }
}
template <typename T, int VEC, int IT, int RED>
static void launch_bwd(const NegPtArgs& a, const float* d_out, int64_t ld, float* dq, float* dn,
                       hipStream_t st) {
    const int64_t items = a.n_query * a.items_per_query;
    constexpr int EPL = IT * VEC;          // scalars per lane per row; the running d_query sum takes EPL registers too
    constexpr int BU = EPL <= 16 ? 2 : 1;  // (as the fused forward)
    k_neg_pertriple_bwd<T, VEC, IT, RED, BU><<<ceil_div(items, 4), 256, 0, st>>>(a, d_out, ld, dq, dn);
}

template <typename T, int VEC, int IT>
static void by_red(int red, bool fwd, const NegPtArgs& a, float* out, const float* d_out, int64_t ld,
                   float* dq, float* dn, const FuseArgs* fuse, hipStream_t st) {
    switch (red) {
        case RED_DOT:
            fwd ? launch_fwd<T, VEC, IT, RED_DOT>(a, out, ld, fuse, st)
                : launch_bwd<T, VEC, IT, RED_DOT>(a, d_out, ld, dq, dn, st);
            break;
        case RED_L1:
            fwd ? launch_fwd<T, VEC, IT, RED_L1>(a, out, ld, fuse, st)
                : launch_bwd<T, VEC, IT, RED_L1>(a, d_out, ld, dq, dn, st);
            break;
        default:
            fwd ? launch_fwd<T, VEC, IT, RED_L2>(a, out, ld, fuse, st)
                : launch_bwd<T, VEC, IT, RED_L2>(a, d_out, ld, dq, dn, st);
    }
}

template <typename T, int VEC>
static int by_it(int it, int red, bool fwd, const NegPtArgs& a, float* out, const float* d_out,
                 int64_t ld, float* dq, float* dn, const FuseArgs* fuse, hipStream_t st) {
    if (it <= 1) by_red<T, VEC, 1>(red, fwd, a, out, d_out, ld, dq, dn, fuse, st);
    else if (it <= 2) by_red<T, VEC, 2>(red, fwd, a, out, d_out, ld, dq, dn, fuse, st);
    else if (it <= 4) by_red<T, VEC, 4>(red, fwd, a, out, d_out, ld, dq, dn, fuse, st);
    else if (it <= 8) by_red<T, VEC, 8>(red, fwd, a, out, d_out, ld, dq, dn, fuse, st);
    else if (it <= 16) by_red<T, VEC, 16>(red, fwd, a, out, d_out, ld, dq, dn, fuse, st);
    else return fail(BESS_EUNSUPPORTED, "neg_score_pertriple: row of %d scalars too wide", a.W);
    return BESS_OK;
}

// ~128 KiB of rows per work item (64 negatives of 2 KiB, up to 256 of 512 B: a wave then runs
// enough iterations to amortise its start-up); shrink the item when the launch would not fill
// 256 CUs x 16 waves
static int negatives_per_item(int64_t n_query, int64_t n_neg, int64_t row_bytes) {
    int nb = 64;
    while (nb < 256 && nb * row_bytes < 131072) nb <<= 1;
    while (nb > 8 && n_query * ceil_div(n_neg, nb) < 256 * 16 * 2) nb >>= 1;
    return nb;
}

static int run(const bess_model_desc* d, bool fwd, const float* query, int64_t n_query,
               const void* neg_base, const int32_t* neg_idx, int64_t n_neg, float* out,
               const float* d_out, int64_t ld, float* dq, float* dn, void* stream,
               const FuseArgs* fuse = nullptr) {
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
    a.nb = negatives_per_item(n_query, n_neg, static_cast<int64_t>(W) * (d->dtype == BESS_F32 ? 4 : 2));  // == row_bytes_of(d)
    a.items_per_query = static_cast<int>(ceil_div(n_neg, a.nb));
    a.sign = is_distance(d->scorer) ? -1.f : 1.f;
    a.p = static_cast<float>(d->norm_p);
    a.dn_by_row = (!fwd && (d->reserved[0] & BESS_FLAG_DNEG_BY_ROW)) ? 1 : 0;
    const int red = reduce_of(d);
    hipStream_t st = as_stream(stream);
    if (!fwd && a.items_per_query > 1 && dq) {
        hipError_t e = fill_words_async(dq, 0u, n_query * W, st);
        if (e != hipSuccess) return fail(static_cast<int>(e), "memset d_query: %s", hipGetErrorString(e));
    }
    // A 16-lane group keeps 16 x 16 chunks of a row in registers (1024 f32 / 2048 f16 scalars at full vector
    // width).  Wider rows are processed in column windows of that size: the dot product and the p = 1 distance are
    // sums over columns (forward: later windows add to the scores; backward: every window writes its own columns).
    // Not for the p = 2 distance (its square root and its gradient need the whole row) nor for the fused training
    // forward (the softmax needs the finished score): BESS_EUNSUPPORTED, the callers use the shared / two-pass forms.
    const int max_cols = 16 * 16 * vec;
    if (W > max_cols) {
        if (red == RED_L2)
            return fail(BESS_EUNSUPPORTED, "neg_score_pertriple: p = 2 on rows of %d scalars (more than %d)", W, max_cols);
        if (fuse)
            return fail(BESS_EUNSUPPORTED, "neg_score_pertriple: fused training forward on rows of %d scalars (more than %d)",
                        W, max_cols);
    }
    const int64_t sz = d->dtype == BESS_F32 ? 4 : 2;
    for (int col0 = 0; col0 < W; col0 += max_cols) {
        const int cols = W - col0 < max_cols ? W - col0 : max_cols;
        NegPtArgs w = a;
        w.query = query + col0;
        w.base = static_cast<const char*>(neg_base) + col0 * sz;
        w.nch = cols / vec;
        w.accum = col0 > 0;
        const int it = static_cast<int>(ceil_div(w.nch, 16));
        float* dqw = dq ? dq + col0 : nullptr;
        float* dnw = dn ? dn + col0 : nullptr;
        int rc = BESS_OK;
        if (d->dtype == BESS_F32) {
            if (vec == 4) rc = by_it<float, 4>(it, red, fwd, w, out, d_out, ld, dqw, dnw, fuse, st);
            else rc = by_it<float, 1>(it, red, fwd, w, out, d_out, ld, dqw, dnw, fuse, st);
        } else {
            if (vec == 8) rc = by_it<half_t, 8>(it, red, fwd, w, out, d_out, ld, dqw, dnw, fuse, st);
            else if (vec == 2) rc = by_it<half_t, 2>(it, red, fwd, w, out, d_out, ld, dqw, dnw, fuse, st);
            else rc = by_it<half_t, 1>(it, red, fwd, w, out, d_out, ld, dqw, dnw, fuse, st);
        }
        if (rc) return rc;
    }
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
    if (n_query > 0 && n_neg > 0 && !(d_out && (d_query || d_neg)))
        return bess::fail(BESS_EINVAL, "neg_score_pertriple_bwd: NULL pointer");
    // (d_query == NULL: only d_neg is wanted - d_query came out of bess_neg_score_pertriple_fwd_dq; DistMult /
    // ComplEx then never read the candidate rows)
    return bess::run(d, false, query, n_query, neg_base, neg_idx, n_neg, nullptr, d_out, ld_dout,
                     d_query, d_neg, stream);
}

static int64_t row_bytes_of(const bess_model_desc* d) {
    return static_cast<int64_t>(d->width) * (d->dtype == BESS_F32 ? 4 : 2);
}

extern "C" int bess_neg_pertriple_items(const bess_model_desc* d, int64_t n_query, int64_t n_neg, int32_t* items) {
    if (!d || !items || n_query < 0 || n_neg < 0) return bess::fail(BESS_EINVAL, "neg_pertriple_items: bad argument");
    *items = n_neg > 0 ? static_cast<int32_t>(bess::ceil_div(n_neg, bess::negatives_per_item(n_query, n_neg, row_bytes_of(d))))
                       : 0;
    return BESS_OK;
}

extern "C" int bess_neg_score_pertriple_fwd_dq(const bess_model_desc* d, const bess_loss_desc* l, const float* query,
                                               int64_t n_query, const void* neg_base, const int32_t* neg_idx,
                                               int64_t n_neg, const float* pos, const float* weight,
                                               int64_t weight_len, float* out, int64_t ld_out, float* d_query,
                                               float* state_ml, float* state_acc, void* stream) {
    return bess_neg_score_pertriple_fwd_dq_masked(d, l, query, n_query, neg_base, neg_idx, n_neg, pos, weight, weight_len,
                                                  nullptr, 0, 0, out, ld_out, d_query, state_ml, state_acc, stream);
}

extern "C" int bess_neg_score_pertriple_fwd_dq_masked(const bess_model_desc* d, const bess_loss_desc* l,
                                                      const float* query, int64_t n_query, const void* neg_base,
                                                      const int32_t* neg_idx, int64_t n_neg, const float* pos,
                                                      const float* weight, int64_t weight_len, const uint8_t* mask,
                                                      int64_t mask_rows, int64_t mask_cols, float* out,
                                                      int64_t ld_out, float* d_query, float* state_ml,
                                                      float* state_acc, void* stream) {
    using namespace bess;
    if (int e = check_desc(d)) return e;
    BESS_REQUIRE(l, "neg_score_pertriple_fwd_dq: NULL loss descriptor");
    BESS_REQUIRE(d->scorer <= BESS_COMPLEX, "neg_score_pertriple_fwd_dq: scorer %d has no fused form", d->scorer);
    BESS_REQUIRE(l->kind == BESS_LOSS_LOGSIGMOID || l->kind == BESS_LOSS_MARGIN || l->kind == BESS_LOSS_SSCE,
                 "neg_score_pertriple_fwd_dq: unknown loss %d", l->kind);
    if (n_query <= 0 || n_neg <= 0) return BESS_OK;
    // (d_query may be NULL: the partials are then left for bess_pertriple_tail, which combines them where it uses them)
    BESS_REQUIRE(out && state_ml && state_acc && weight && (weight_len == 1 || weight_len == n_query),
                 "neg_score_pertriple_fwd_dq: NULL pointer or bad weight length");
    BESS_REQUIRE(pos || l->kind == BESS_LOSS_LOGSIGMOID, "neg_score_pertriple_fwd_dq: this loss needs the positive scores");
    FuseArgs f;
    f.pos = pos;
    f.kind = l->kind;
    f.beta = l->kind == BESS_LOSS_SSCE ? 1.f : (l->adversarial ? l->adversarial_scale : 0.f);
    f.margin = l->margin;
    f.shift = l->kind == BESS_LOSS_SSCE ? l->ssce_shift : 0.f;
    f.st_ml = state_ml;
    f.st_acc = state_acc;
    if (mask) {
        BESS_REQUIRE(mask_cols > 0 && mask_cols <= n_neg && (mask_rows == 1 || mask_rows == n_query),
                     "neg_score_pertriple_fwd_dq_masked: mask [%lld, %lld] for %lld queries x %lld negatives",
                     (long long)mask_rows, (long long)mask_cols, (long long)n_query, (long long)n_neg);
        f.mask = mask;
        f.mask_rows = mask_rows;
        f.mask_cols = static_cast<int>(mask_cols);
        f.mask_from = static_cast<int>(n_neg - mask_cols);
    }
    const int rc = run(d, true, query, n_query, neg_base, neg_idx, n_neg, out, nullptr, ld_out, nullptr, nullptr, stream,
                       &f);
    if (rc) return rc;
    const int items = static_cast<int>(ceil_div(n_neg, negatives_per_item(n_query, n_neg, row_bytes_of(d))));
    if (!d_query) return BESS_OK;  // the partials stay as they are: bess_pertriple_tail combines them where it uses them
    k_combine_dq<<<static_cast<unsigned>(ceil_div(n_query, 4)), 256, 0, as_stream(stream)>>>(
        state_ml, state_acc, n_query, items, d->width, l->kind, l->loss_scale, pos, weight, weight_len, nullptr, d_query);
    return check_launch("neg_score_pertriple_fwd_dq");
}

extern "C" int bess_neg_score_pertriple_fwd_partials(const bess_model_desc* d, const bess_loss_desc* l,
                                                     const float* query, int64_t n_query, const void* neg_base,
                                                     const int32_t* neg_idx, int64_t n_neg, float* out,
                                                     int64_t ld_out, float* state_ml, float* state_acc, void* stream) {
    using namespace bess;
    if (int e = check_desc(d)) return e;
    BESS_REQUIRE(l, "neg_score_pertriple_fwd_partials: NULL loss descriptor");
    BESS_REQUIRE(d->scorer <= BESS_COMPLEX, "neg_score_pertriple_fwd_partials: scorer %d has no fused form", d->scorer);
    if (l->kind != BESS_LOSS_LOGSIGMOID && l->kind != BESS_LOSS_SSCE)
        return fail(BESS_EUNSUPPORTED, "neg_score_pertriple_fwd_partials: loss %d weighs a negative by the positive score, "
                                       "which the scoring shard does not have", l->kind);
    if (n_query <= 0 || n_neg <= 0) return BESS_OK;
    BESS_REQUIRE(out && state_ml && state_acc, "neg_score_pertriple_fwd_partials: NULL pointer");
    FuseArgs f;
    f.pos = nullptr;
    f.kind = l->kind;
    f.beta = l->kind == BESS_LOSS_SSCE ? 1.f : (l->adversarial ? l->adversarial_scale : 0.f);
    f.margin = l->margin;
    f.shift = l->kind == BESS_LOSS_SSCE ? l->ssce_shift : 0.f;
    f.st_ml = state_ml;
    f.st_acc = state_acc;
    return run(d, true, query, n_query, neg_base, neg_idx, n_neg, out, nullptr, ld_out, nullptr, nullptr, stream, &f);
}

extern "C" int bess_combine_dq_partials(const float* state_ml, const float* state_acc, int64_t n_query, int32_t items,
                                        int32_t width, const float* norm, float* d_query, void* stream) {
    using namespace bess;
    BESS_REQUIRE(n_query >= 0 && items > 0 && width > 0, "combine_dq_partials: bad sizes");
    if (n_query == 0) return BESS_OK;
    BESS_REQUIRE(state_ml && state_acc && norm && d_query, "combine_dq_partials: NULL pointer");
    k_combine_dq<<<static_cast<unsigned>(ceil_div(n_query, 4)), 256, 0, as_stream(stream)>>>(
        state_ml, state_acc, n_query, items, width, 0, 1.f, nullptr, nullptr, 1, norm, d_query);
    return check_launch("combine_dq_partials");
}

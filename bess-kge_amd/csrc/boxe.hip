// BoxE negative scoring (reference scoring.py:1149-1415).
//
// An entity row is [base | bump] (2 x d).  For a query (kept entity + relation) and a
// candidate e, each of the two candidate parts p ("slots") meets one relation box:
//
//     x      = e_p + S_p                    (S: the kept entity's contribution to the bumped point)
//     x'     = tanh(x)                      (apply_tanh)
//     dist   = |x' - C_p|                   (C: box centre)
//     inside = dist <= H_p                  (H: half width; per dimension, or all(d) per slot)
//     B = 1 + 2 H,  A = 1 / B
//     f      = inside ? dist * A : dist * B - H * (B - A)
//     score  = -( ||f_0||_p + ||f_1||_p )
//
// (`boxe_score`, scoring.py:1250-1340, with width = 2 H and k = H (B - A).)  The query matrix is
// [n_query, 6 d] = [S_0 | C_0 | H_0 | S_1 | C_1 | H_1]; which box (head / tail) a slot meets and all
// relation-side preprocessing (geometric-mean width normalisation, elu size, tanh of the box) is
// query-only work done by the host side (S rows).  desc.width = 2 d, desc.reserved[0]: bit 0 =
// tanh, bit 1 = per-dimension choice.
//
// ~15 VALU ops per element and per-query parameters: there is no useful LDS tile form, so one
// kernel family serves both regimes - a 16-lane DPP row streams one candidate row (mapping of
// neg_pertriple.hip); shared negatives are the same kernel with a zero index stride (the N
// candidate rows stay cache resident) and an atomic d_neg.
#include <algorithm>

#include "common.h"

namespace bess {

struct BoxArgs {
    const float* query;  // [n_query, 6 d]
    const void* base;
    const int32_t* idx;  // [n_query * n_neg] (idx_stride = n_neg) or [n_neg] (idx_stride = 0) or NULL (row k)
    int64_t idx_stride;
    int64_t n_query;
    int n_neg;
    int d;
    int nch;
    int nb, items_per_query;
    float p;  // the norm of the P == 2 kernels: any p != 1 (common.h lp_*)
};

template <int VEC, int IT>
struct BoxQuery {
    float s[2][IT][VEC], c[2][IT][VEC], h[2][IT][VEC];
    __device__ __forceinline__ void load(const float* qp, int g, int d, int nch) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int ch = g + 16 * it;
                load_chunk<float, VEC>(qp + (3 * p + 0) * d, ch, nch, s[p][it]);
                load_chunk<float, VEC>(qp + (3 * p + 1) * d, ch, nch, c[p][it]);
                load_chunk<float, VEC>(qp + (3 * p + 2) * d, ch, nch, h[p][it]);
            }
    }
};

template <typename T, int VEC, int IT>
__device__ __forceinline__ void box_load_row(const T* rp, int g, int d, int nch, float (&ev)[2][IT][VEC]) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            load_chunk<T, VEC>(rp + p * d, g + 16 * it, nch, ev[p][it]);
        }
}

__device__ __forceinline__ const int32_t* box_idx(const BoxArgs& a, int64_t q) {
    return a.idx ? a.idx + q * a.idx_stride : nullptr;
}

// per element: bumped point (after tanh), distance to the centre, inside flag
template <bool TANH>
__device__ __forceinline__ void box_point(float e, float s, float c, float h, float& xp, float& dist, bool& inside) {
    xp = e + s;
    if (TANH) xp = tanhf(xp);
    dist = fabsf(xp - c);
    inside = dist <= h;
}

__device__ __forceinline__ float box_final(float dist, float h, bool inside) {
    const float B = 1.f + 2.f * h;
    const float A = 1.f / B;
    return inside ? dist * A : dist * B - h * (B - A);
}

template <typename T, int VEC, int IT, int P, bool TANH, bool PERDIM>
__global__ __launch_bounds__(256) void k_box_fwd(BoxArgs a, float* __restrict__ out, int64_t ld_out) {
    const int lane = threadIdx.x & 63;
    const int g = lane & 15, sub = lane >> 4;
    const int64_t item = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (item >= a.n_query * a.items_per_query) return;
    const int64_t q = item / a.items_per_query;
    const int k0 = static_cast<int>(item - q * a.items_per_query) * a.nb;
    const int k1 = min(k0 + a.nb, a.n_neg);
    BoxQuery<VEC, IT> qv;
    qv.load(a.query + q * 6 * a.d, g, a.d, a.nch);
    const T* base = static_cast<const T*>(a.base);
    const int32_t* idx = box_idx(a, q);
    float* orow = out + q * ld_out;
    for (int kb = k0; kb < k1; kb += 4) {
        const int k = kb + sub;
        const bool valid = k < k1;
        const int ks = valid ? k : (k1 - 1);
        const int64_t row = idx ? static_cast<int64_t>(idx[ks]) : ks;
        float ev[2][IT][VEC];
        box_load_row<T, VEC, IT>(base + row * 2 * a.d, g, a.d, a.nch, ev);
        float total = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float outside = 0.f;
            if (!PERDIM) {  // one choice per slot: inside only if every dimension is
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        float xp, dist;
                        bool in;
                        box_point<TANH>(ev[p][it][v], qv.s[p][it][v], qv.c[p][it][v], qv.h[p][it][v], xp, dist, in);
                        // padded chunks (all zero) are "inside": dist = |tanh(0) - 0| = 0 <= 0
                        outside += in ? 0.f : 1.f;
                    }
                outside = row16_allreduce_sum(outside);
            }
            float acc = 0.f;
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float xp, dist;
                    bool in;
                    box_point<TANH>(ev[p][it][v], qv.s[p][it][v], qv.c[p][it][v], qv.h[p][it][v], xp, dist, in);
                    if (!PERDIM) in = outside == 0.f;
                    const float f = box_final(dist, qv.h[p][it][v], in);
                    if (P == 1) acc += fabsf(f);
                    else acc += lp_term(f, a.p);
                }
            acc = row16_allreduce_sum(acc);
            total += (P == 2) ? lp_root(acc, a.p) : acc;
        }
        if (g == 0 && valid) orow[k] = -total;
    }
}

template <typename T, int VEC, int IT, int P, bool TANH, bool PERDIM>
__global__ __launch_bounds__(256) void k_box_bwd(BoxArgs a, const float* __restrict__ d_out, int64_t ld_dout,
                                                 float* __restrict__ d_query, float* __restrict__ d_neg,
                                                 int d_neg_atomic) {
    const int lane = threadIdx.x & 63;
    const int g = lane & 15, sub = lane >> 4;
    const int64_t item = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (item >= a.n_query * a.items_per_query) return;
    const int64_t q = item / a.items_per_query;
    const int k0 = static_cast<int>(item - q * a.items_per_query) * a.nb;
    const int k1 = min(k0 + a.nb, a.n_neg);
    BoxQuery<VEC, IT> qv;
    qv.load(a.query + q * 6 * a.d, g, a.d, a.nch);
    float ds[2][IT][VEC], dc[2][IT][VEC], dh[2][IT][VEC];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int v = 0; v < VEC; ++v) ds[p][it][v] = dc[p][it][v] = dh[p][it][v] = 0.f;
    const T* base = static_cast<const T*>(a.base);
    const int32_t* idx = box_idx(a, q);
    for (int kb = k0; kb < k1; kb += 4) {
        const int k = kb + sub;
        const bool valid = k < k1;
        const int ks = valid ? k : (k1 - 1);
        const int64_t row = idx ? static_cast<int64_t>(idx[ks]) : ks;
        float ev[2][IT][VEC];
        box_load_row<T, VEC, IT>(base + row * 2 * a.d, g, a.d, a.nch, ev);
        const float go = valid ? -d_out[q * ld_dout + ks] : 0.f;  // d score / d (||f_0|| + ||f_1||)
        float* dn = nullptr;
        if (d_neg) dn = d_neg + (d_neg_atomic ? static_cast<int64_t>(ks) : q * a.n_neg + ks) * 2 * a.d;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float outside = 0.f, ss = 0.f;
            if (!PERDIM || P == 2) {
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        float xp, dist;
                        bool in;
                        box_point<TANH>(ev[p][it][v], qv.s[p][it][v], qv.c[p][it][v], qv.h[p][it][v], xp, dist, in);
                        outside += in ? 0.f : 1.f;
                    }
                if (!PERDIM) outside = row16_allreduce_sum(outside);
            }
            if (P == 2) {
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        float xp, dist;
                        bool in;
                        box_point<TANH>(ev[p][it][v], qv.s[p][it][v], qv.c[p][it][v], qv.h[p][it][v], xp, dist, in);
                        if (!PERDIM) in = outside == 0.f;
                        const float f = box_final(dist, qv.h[p][it][v], in);
                        ss += lp_term(f, a.p);
                    }
                ss = row16_allreduce_sum(ss);
            }
            const float gp = (P == 2) ? go * lp_inv(lp_root(ss, a.p), a.p) : go;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int ch = g + 16 * it;
                float de[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float h = qv.h[p][it][v], c = qv.c[p][it][v];
                    float xp, dist;
                    bool in;
                    box_point<TANH>(ev[p][it][v], qv.s[p][it][v], c, h, xp, dist, in);
                    if (!PERDIM) in = outside == 0.f;
                    const float B = 1.f + 2.f * h, A = 1.f / B;
                    const float f = in ? dist * A : dist * B - h * (B - A);
                    const float df = (P == 1) ? gp * sgnf(f) : gp * lp_dterm(f, a.p);
                    const float ddist = df * (in ? A : B);
                    const float dhh = df * (in ? -2.f * dist * A * A : 2.f * dist - (B - A) - h * (2.f + 2.f * A * A));
                    const float sg = sgnf(xp - c);
                    float dx = ddist * sg;
                    dc[p][it][v] -= dx;
                    if (TANH) dx *= (1.f - xp * xp);
                    ds[p][it][v] += dx;
                    dh[p][it][v] += dhh;
                    de[v] = dx;
                }
                if (dn && valid && ch < a.nch) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        if (d_neg_atomic) unsafeAtomicAdd(dn + p * a.d + ch * VEC + v, de[v]);
                        else dn[p * a.d + ch * VEC + v] = de[v];
                    }
                }
            }
        }
    }
    float* dqp = d_query + q * 6 * a.d;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int ch = g + 16 * it;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float x[3] = {ds[p][it][v], dc[p][it][v], dh[p][it][v]};
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    x[j] += __shfl_xor(x[j], 16, 64);
                    x[j] += __shfl_xor(x[j], 32, 64);
                    if (sub == 0 && ch < a.nch) {
                        float* o = dqp + (3 * p + j) * a.d + ch * VEC + v;
                        if (a.items_per_query == 1) *o = x[j];
                        else unsafeAtomicAdd(o, x[j]);
                    }
                }
            }
        }
}

template <typename T, int VEC, int IT, int P>
static void box_launch(int flags, bool fwd, const BoxArgs& a, float* out, const float* d_out, int64_t ld, float* dq,
                       float* dn, int dn_atomic, hipStream_t st) {
    const unsigned blocks = static_cast<unsigned>(ceil_div(a.n_query * a.items_per_query, 4));
    const bool th = flags & 1, pd = flags & 2;
#define BESS_BOX(TH, PD)                                                                                   \
    (fwd ? k_box_fwd<T, VEC, IT, P, TH, PD><<<blocks, 256, 0, st>>>(a, out, ld)                            \
         : k_box_bwd<T, VEC, IT, P, TH, PD><<<blocks, 256, 0, st>>>(a, d_out, ld, dq, dn, dn_atomic))
    if (th && pd) BESS_BOX(true, true);
    else if (th) BESS_BOX(true, false);
    else if (pd) BESS_BOX(false, true);
    else BESS_BOX(false, false);
#undef BESS_BOX
}

template <typename T, int VEC>
static int box_by_it(int it, int p, int flags, bool fwd, const BoxArgs& a, float* out, const float* d_out, int64_t ld,
                     float* dq, float* dn, int dn_atomic, hipStream_t st) {
#define BESS_BOXP(ITV)                                                                        \
    (p == 1 ? box_launch<T, VEC, ITV, 1>(flags, fwd, a, out, d_out, ld, dq, dn, dn_atomic, st) \
            : box_launch<T, VEC, ITV, 2>(flags, fwd, a, out, d_out, ld, dq, dn, dn_atomic, st))
    if (it <= 1) BESS_BOXP(1);
    else if (it <= 2) BESS_BOXP(2);
    else if (it <= 4) BESS_BOXP(4);
    else if (it <= 8) BESS_BOXP(8);
    else return fail(BESS_EUNSUPPORTED, "BoxE: embedding size %d too wide for the kernels (max 512; 128 when not a multiple of 4)", a.d);
#undef BESS_BOXP
    return BESS_OK;
}

// shared == true: idx is one list of n_neg rows for every query (or NULL = rows 0..n_neg-1), d_neg is [n_neg, 2d]
int boxe_negatives(const bess_model_desc* d, bool fwd, bool shared, const float* query, int64_t n_query,
                   const void* neg_base, const int32_t* neg_idx, int64_t n_neg, float* out, const float* d_out,
                   int64_t ld, float* dq, float* dn, hipStream_t st) {
    const int dd = d->width / 2;
    const int vec = (dd % 4 == 0) ? 4 : 1;
    BoxArgs a;
    a.query = query;
    a.base = neg_base;
    a.idx = neg_idx;
    a.idx_stride = shared ? 0 : n_neg;
    a.n_query = n_query;
    a.n_neg = static_cast<int>(n_neg);
    a.d = dd;
    a.nch = dd / vec;
    int nb = 64;
    while (nb > 8 && n_query * ceil_div(n_neg, nb) < 256 * 16 * 2) nb >>= 1;
    a.nb = nb;
    a.items_per_query = static_cast<int>(ceil_div(n_neg, nb));
    a.p = static_cast<float>(d->norm_p);
    const int it = static_cast<int>(ceil_div(a.nch, 16));
    if (!fwd) {
        hipError_t e = hipSuccess;
        if (a.items_per_query > 1) e = fill_words_async(dq, 0u, n_query * 6 * dd, st);
        if (e == hipSuccess && shared && dn) e = fill_words_async(dn, 0u, n_neg * 2 * dd, st);
        if (e != hipSuccess) return fail(static_cast<int>(e), "memset: %s", hipGetErrorString(e));
    }
    int rc;
    const int flags = d->reserved[0];
    if (d->dtype == BESS_F32)
        rc = vec == 4 ? box_by_it<float, 4>(it, d->norm_p, flags, fwd, a, out, d_out, ld, dq, dn, shared, st)
                      : box_by_it<float, 1>(it, d->norm_p, flags, fwd, a, out, d_out, ld, dq, dn, shared, st);
    else
        rc = vec == 4 ? box_by_it<half_t, 4>(it, d->norm_p, flags, fwd, a, out, d_out, ld, dq, dn, shared, st)
                      : box_by_it<half_t, 1>(it, d->norm_p, flags, fwd, a, out, d_out, ld, dq, dn, shared, st);
    if (rc) return rc;
    return check_launch(fwd ? "neg_score fwd (BoxE)" : "neg_score bwd (BoxE)");
}

// ---------------------------------------------------------------------------
// K9 for the per-triple negatives of the own shard (BoxE counterpart of k_pertriple_grad_segments,
// segments.hip): one 16-lane group per unique destination row; the row is loaded once and every
// reference (q, k) of the segment contributes d score / d e recomputed from query[q] (its six
// vectors) and d_out[q, k] - the arithmetic of k_box_bwd's d_neg, summed on chip instead of through
// an [n_query * n_neg, 2 d] tensor and atomics.  Rows with more than BESS_SEGMENT_CAP references go
// to the whole grid in slices (k_box_long_segments), as in segments.hip.
struct BoxSegArgs {
    const float* query;          // [n_query, 6 d]
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
    const int32_t* long_segs;
    float p;
};

// acc += d score / d e over references [r0, r1) of one row (ev)
template <int VEC, int IT, int P, bool TANH, bool PERDIM>
__device__ __forceinline__ void box_seg_accumulate(const BoxSegArgs& a, int g, const float (&ev)[2][IT][VEC], int r0,
                                                   int r1, float (&acc)[2][IT][VEC]) {
    for (int r = r0; r < r1; ++r) {
        const int ref = a.refs[r];
        const int q = ref / a.n_neg;
        const int k = ref - q * a.n_neg;
        const float go = -a.d_out[q * a.ld_dout + k];
        BoxQuery<VEC, IT> qv;
        qv.load(a.query + static_cast<int64_t>(q) * 6 * a.d, g, a.d, a.nch);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float outside = 0.f, ss = 0.f;
            if (!PERDIM) {
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        float xp, dist;
                        bool in;
                        box_point<TANH>(ev[p][it][v], qv.s[p][it][v], qv.c[p][it][v], qv.h[p][it][v], xp, dist, in);
                        outside += in ? 0.f : 1.f;
                    }
                outside = row16_allreduce_sum(outside);
            }
            if (P == 2) {
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        float xp, dist;
                        bool in;
                        box_point<TANH>(ev[p][it][v], qv.s[p][it][v], qv.c[p][it][v], qv.h[p][it][v], xp, dist, in);
                        if (!PERDIM) in = outside == 0.f;
                        const float f = box_final(dist, qv.h[p][it][v], in);
                        ss += lp_term(f, a.p);
                    }
                ss = row16_allreduce_sum(ss);
            }
            const float gp = (P == 2) ? go * lp_inv(lp_root(ss, a.p), a.p) : go;
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float h = qv.h[p][it][v], c = qv.c[p][it][v];
                    float xp, dist;
                    bool in;
                    box_point<TANH>(ev[p][it][v], qv.s[p][it][v], c, h, xp, dist, in);
                    if (!PERDIM) in = outside == 0.f;
                    const float B = 1.f + 2.f * h, A = 1.f / B;
                    const float f = in ? dist * A : dist * B - h * (B - A);
                    const float df = (P == 1) ? gp * sgnf(f) : gp * lp_dterm(f, a.p);
                    float dx = df * (in ? A : B) * sgnf(xp - c);
                    if (TANH) dx *= (1.f - xp * xp);
                    acc[p][it][v] += dx;
                }
        }
    }
}

template <typename T, int VEC, int IT>
__device__ __forceinline__ void box_seg_store(const BoxSegArgs& a, int g, const float (&ev)[2][IT][VEC],
                                              const float (&acc)[2][IT][VEC], int64_t seg, int64_t row,
                                              float* __restrict__ grad_seg, T* table_rw, float lr) {
    const int W = 2 * a.d;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = g + 16 * it;
            if (c >= a.nch) continue;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const int64_t o = p * a.d + c * VEC + v;
                if (grad_seg) grad_seg[seg * W + o] = acc[p][it][v];
                else table_rw[row * W + o] = static_cast<T>(ev[p][it][v] - lr * acc[p][it][v]);
            }
        }
}

template <typename T, int VEC, int IT, int P, bool TANH, bool PERDIM>
__global__ __launch_bounds__(256) void k_box_grad_segments(BoxSegArgs a, float* __restrict__ grad_seg, T* table_rw,
                                                           float lr) {
    const int g = threadIdx.x & 15;
    const int n_seg = *a.n_seg;
    const int64_t group0 = (blockIdx.x * 256ll + threadIdx.x) >> 4;
    const int64_t n_group = (gridDim.x * 256ll) >> 4;
    const T* table = static_cast<const T*>(a.table);
    for (int64_t seg = group0; seg < n_seg; seg += n_group) {
        const int64_t row = a.seg_rows[seg];
        const int r0 = a.seg_offsets[seg], r1 = a.seg_offsets[seg + 1];
        if (a.long_segs && r1 - r0 > BESS_SEGMENT_CAP) continue;  // left to k_box_long_segments
        float ev[2][IT][VEC], acc[2][IT][VEC];
        box_load_row<T, VEC, IT>(table + row * 2 * a.d, g, a.d, a.nch, ev);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[p][it][v] = 0.f;
        box_seg_accumulate<VEC, IT, P, TANH, PERDIM>(a, g, ev, r0, r1, acc);
        box_seg_store<T, VEC, IT>(a, g, ev, acc, seg, row, grad_seg, table_rw, lr);
    }
}

template <typename T, int VEC, int IT, int P, bool TANH, bool PERDIM>
__global__ __launch_bounds__(256) void k_box_long_segments(BoxSegArgs a, float* __restrict__ long_grad,
                                                           int32_t* __restrict__ long_cnt, int32_t capacity,
                                                           float* __restrict__ grad_seg, T* table_rw, float lr) {
    const int lane = threadIdx.x & 63, g = lane & 15;
    const int64_t group0 = (blockIdx.x * 256ll + threadIdx.x) >> 4;
    const int64_t n_group = (gridDim.x * 256ll) >> 4;
    const int n_long = min(a.long_segs[0], capacity);
    const T* table = static_cast<const T*>(a.table);
    const int W = 2 * a.d;
    for (int li = 0; li < n_long; ++li) {
        const int seg = a.long_segs[1 + li];
        const int64_t row = a.seg_rows[seg];
        const int r0 = a.seg_offsets[seg], r1 = a.seg_offsets[seg + 1];
        const int parts = (r1 - r0 + BESS_SEGMENT_CAP - 1) / BESS_SEGMENT_CAP;
        if (group0 >= parts) continue;
        float ev[2][IT][VEC];
        box_load_row<T, VEC, IT>(table + row * W, g, a.d, a.nch, ev);
        float* sum = long_grad + static_cast<int64_t>(li) * W;
        for (int64_t pt = group0; pt < parts; pt += n_group) {
            float acc[2][IT][VEC];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[p][it][v] = 0.f;
            const int rb = r0 + static_cast<int>(pt) * BESS_SEGMENT_CAP;
            box_seg_accumulate<VEC, IT, P, TANH, PERDIM>(a, g, ev, rb, min(r1, rb + BESS_SEGMENT_CAP), acc);
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int c = g + 16 * it;
                    if (c < a.nch) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) atomicAdd(sum + p * a.d + c * VEC + v, acc[p][it][v]);
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
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int c = g + 16 * it;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        acc[p][it][v] = 0.f;
                        if (c < a.nch) {
                            float* sp = sum + p * a.d + c * VEC + v;
                            acc[p][it][v] = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(sp, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            box_seg_store<T, VEC, IT>(a, g, ev, acc, seg, row, grad_seg, table_rw, lr);
        }
    }
}

template <typename T, int VEC, int IT, int P>
static void box_seg_launch(int flags, const BoxSegArgs& a, float* grad_seg, void* rw, float lr, unsigned grid,
                           hipStream_t st, float* long_grad, int32_t* long_cnt, int32_t cap) {
    T* t = static_cast<T*>(rw);
    const bool th = flags & 1, pd = flags & 2;
#define BESS_BOXS(TH, PD)                                                                                             \
    (long_grad ? k_box_long_segments<T, VEC, IT, P, TH, PD><<<grid, 256, 0, st>>>(a, long_grad, long_cnt, cap, grad_seg, t, lr) \
               : k_box_grad_segments<T, VEC, IT, P, TH, PD><<<grid, 256, 0, st>>>(a, grad_seg, t, lr))
    if (th && pd) BESS_BOXS(true, true);
    else if (th) BESS_BOXS(true, false);
    else if (pd) BESS_BOXS(false, true);
    else BESS_BOXS(false, false);
#undef BESS_BOXS
}

template <typename T, int VEC>
static int box_seg_by_it(int it, int p, int flags, const BoxSegArgs& a, float* grad_seg, void* rw, float lr,
                         unsigned grid, hipStream_t st, float* long_grad = nullptr, int32_t* long_cnt = nullptr,
                         int32_t cap = 0) {
#define BESS_BOXSP(ITV)                                                                                    \
    (p == 1 ? box_seg_launch<T, VEC, ITV, 1>(flags, a, grad_seg, rw, lr, grid, st, long_grad, long_cnt, cap) \
            : box_seg_launch<T, VEC, ITV, 2>(flags, a, grad_seg, rw, lr, grid, st, long_grad, long_cnt, cap))
    if (it <= 1) BESS_BOXSP(1);
    else if (it <= 2) BESS_BOXSP(2);
    else if (it <= 4) BESS_BOXSP(4);
    else if (it <= 8) BESS_BOXSP(8);
    else return fail(BESS_EUNSUPPORTED, "BoxE: embedding size %d too wide for the kernels (max 512; 128 when not a multiple of 4)", a.d);
#undef BESS_BOXSP
    return BESS_OK;
}

int boxe_grad_segments(const bess_model_desc* d, const float* query, void* table, int64_t n_neg, const float* d_out,
                       int64_t ld_dout, const int32_t* refs_sorted, const int32_t* seg_rows,
                       const int32_t* seg_offsets, const int32_t* n_seg, int64_t max_seg, float* grad_seg,
                       float fused_sgd_lr, const int32_t* long_segs, int64_t long_cap, float* long_grad,
                       int32_t* long_count, hipStream_t st) {
    const int dd = d->width / 2;
    const int vec = (dd % 4 == 0) ? 4 : 1;
    BoxSegArgs a{query, table, d_out, ld_dout, refs_sorted, seg_rows, seg_offsets, n_seg, static_cast<int>(n_neg),
                 dd, dd / vec, long_segs, static_cast<float>(d->norm_p)};
    const int it = static_cast<int>(ceil_div(a.nch, 16));
    const int flags = d->reserved[0];
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(ceil_div(max_seg, 16), 256 * 16));
    int rc;
#define BESS_BOXSD(G, LG, LC, CAP)                                                                                     \
    (d->dtype == BESS_F32                                                                                              \
         ? (vec == 4 ? box_seg_by_it<float, 4>(it, d->norm_p, flags, a, grad_seg, table, fused_sgd_lr, G, st, LG, LC, CAP)  \
                     : box_seg_by_it<float, 1>(it, d->norm_p, flags, a, grad_seg, table, fused_sgd_lr, G, st, LG, LC, CAP)) \
         : (vec == 4 ? box_seg_by_it<half_t, 4>(it, d->norm_p, flags, a, grad_seg, table, fused_sgd_lr, G, st, LG, LC, CAP) \
                     : box_seg_by_it<half_t, 1>(it, d->norm_p, flags, a, grad_seg, table, fused_sgd_lr, G, st, LG, LC, CAP)))
    rc = BESS_BOXSD(grid, nullptr, nullptr, 0);
    if (rc) return rc;
    if (long_segs) {
        rc = BESS_BOXSD(1024u, long_grad, long_count, static_cast<int32_t>(long_cap));
        if (rc) return rc;
    }
#undef BESS_BOXSD
    return check_launch("neg_pertriple_grad_segments (BoxE)");
}

}  // namespace bess

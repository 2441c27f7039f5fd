// Per-triple kernels of the BESS hot path on gfx950:
//   K2+K3  score_triple            (reference scoring.py:321-330,423-434,804-813,905-916)
//   K2+K6  query transform          (scoring.py:342,354,446-448,460-462,825,837,928-946;
//                                    utils.py:72-112 complex_multiplication / complex_rotation)
// and their backward passes.
//
// Work unit: one 64-lane wavefront per triple; lanes stride over the embedding
// (consecutive lanes -> consecutive scalars, 256 B per wave load), partial sums
// are combined with DPP / cross-row shuffles.  These kernels touch 2-3 rows per
// triple (S rows in total) - they are the small side of the step; the rows that
// dominate HBM traffic are the negatives (neg_pertriple.hip / neg_shared.hip).
//
// Entity rows of RotatE / ComplEx are stored [re(d) | im(d)]
// (scoring.py:404-408, 881-885); RotatE relation rows are phases [d].
#include "common.h"
#include "loss_rows.h"

namespace bess {

struct TripleArgs {
    const void* head_base;
    const int32_t* head_idx;
    const void* tail_base;
    const int32_t* tail_idx;
    const void* rel_table;
    const int32_t* rel_idx;
    int64_t n;
    int W;
    int Wr;
    int norm_p;
};

// ------------------------------------------------------------------ forward
template <typename T, int SCORER>
__device__ __forceinline__ void triple_fwd_body(const TripleArgs& a, int64_t s, int lane, float* __restrict__ out) {
    const T* h = row_ptr(static_cast<const T*>(a.head_base), a.head_idx, s, a.W);
    const T* t = row_ptr(static_cast<const T*>(a.tail_base), a.tail_idx, s, a.W);
    const T* r = static_cast<const T*>(a.rel_table) + static_cast<int64_t>(a.rel_idx[s]) * a.Wr;
    float acc = 0.f;
    if (SCORER == BESS_TRANSE) {
        for (int e = lane; e < a.W; e += 64) {
            const float x = to_f32(h[e]) + to_f32(r[e]) - to_f32(t[e]);
            acc += (a.norm_p == 1) ? fabsf(x) : lp_term(x, static_cast<float>(a.norm_p));
        }
    } else if (SCORER == BESS_DISTMULT) {
        for (int e = lane; e < a.W; e += 64) acc += to_f32(h[e]) * to_f32(r[e]) * to_f32(t[e]);
    } else {
        const int d = a.W / 2;
        for (int e = lane; e < d; e += 64) {
            const float hr = to_f32(h[e]), hi = to_f32(h[d + e]);
            const float tr = to_f32(t[e]), ti = to_f32(t[d + e]);
            float rr, ri;
            if (SCORER == BESS_ROTATE) {
                const float ph = to_f32(r[e]);
                rr = cosf(ph);
                ri = sinf(ph);
            } else {
                rr = to_f32(r[e]);
                ri = to_f32(r[d + e]);
            }
            const float qr = hr * rr - hi * ri;
            const float qi = hr * ri + hi * rr;
            if (SCORER == BESS_ROTATE) {
                const float xr = qr - tr, xi = qi - ti;
                acc += (a.norm_p == 1) ? (fabsf(xr) + fabsf(xi))
                                       : (lp_term(xr, static_cast<float>(a.norm_p)) + lp_term(xi, static_cast<float>(a.norm_p)));
            } else {
                acc += qr * tr + qi * ti;
            }
        }
    }
    acc = wave_allreduce_sum(acc);
    if (SCORER == BESS_TRANSE || SCORER == BESS_ROTATE)
        acc = -((a.norm_p == 1) ? acc : lp_root(acc, static_cast<float>(a.norm_p)));
    if (lane == 0) out[s] = acc;
}

template <typename T, int SCORER>
__global__ __launch_bounds__(256) void k_score_triple_fwd(TripleArgs a, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (s >= a.n) return;
    triple_fwd_body<T, SCORER>(a, s, lane, out);
}

struct QueryArgs {
    const void* ent_base;
    const int32_t* ent_idx;
    const void* rel_table;
    const int32_t* rel_idx;
    int64_t n;
    int W;
    int Wr;
    int side;
};

template <typename T, int SCORER>
__device__ __forceinline__ void query_fwd_body(const QueryArgs& a, int64_t q, int lane, float* __restrict__ query) {
    const T* x = row_ptr(static_cast<const T*>(a.ent_base), a.ent_idx, q, a.W);
    const T* r = static_cast<const T*>(a.rel_table) + static_cast<int64_t>(a.rel_idx[q]) * a.Wr;
    float* o = query + q * a.W;
    const bool tail = a.side == BESS_CORRUPT_TAIL;
    if (SCORER == BESS_TRANSE) {
        for (int e = lane; e < a.W; e += 64)
            o[e] = tail ? (to_f32(x[e]) + to_f32(r[e])) : (to_f32(x[e]) - to_f32(r[e]));
    } else if (SCORER == BESS_DISTMULT) {
        for (int e = lane; e < a.W; e += 64) o[e] = to_f32(x[e]) * to_f32(r[e]);
    } else {
        const int d = a.W / 2;
        for (int e = lane; e < d; e += 64) {
            const float xr = to_f32(x[e]), xi = to_f32(x[d + e]);
            float rr, ri;
            if (SCORER == BESS_ROTATE) {
                // heads: rotate the tail by -r (scoring.py:446-448)
                const float ph = tail ? to_f32(r[e]) : -to_f32(r[e]);
                rr = cosf(ph);
                ri = sinf(ph);
            } else {
                // heads: conjugate relation times tail (scoring.py:928-932)
                rr = to_f32(r[e]);
                ri = tail ? to_f32(r[d + e]) : -to_f32(r[d + e]);
            }
            o[e] = xr * rr - xi * ri;
            o[d + e] = xr * ri + xi * rr;
        }
    }
}

template <typename T, int SCORER>
__global__ __launch_bounds__(256) void k_query_fwd(QueryArgs a, float* __restrict__ query) {
    const int lane = threadIdx.x & 63;
    const int64_t q = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (q >= a.n) return;
    query_fwd_body<T, SCORER>(a, q, lane, query);
}

// K2 + K3 + K6 of one replica's micro-batch in one launch: the positive score of every triple and the query
// row in front of its negatives (the same wave reads h, r, t once from HBM; the second use hits L1 / L2)
template <typename T, int SCORER>
__global__ __launch_bounds__(256) void k_query_triple_fwd(TripleArgs a, QueryArgs qa, float* __restrict__ query,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (s >= a.n) return;
    query_fwd_body<T, SCORER>(qa, s, lane, query);
    triple_fwd_body<T, SCORER>(a, s, lane, out);
}

// the same launch with copy / fill jobs in spare workgroups behind the triples' (a training step's prologue - the
// concatenated candidate list, cleared gradient targets, the update's generation counter - without a launch of
// its own: nothing of this launch reads what the jobs write)
template <typename T, int SCORER>
__global__ __launch_bounds__(256) void k_query_triple_fwd_jobs(TripleArgs a, QueryArgs qa, float* __restrict__ query,
                                                               float* __restrict__ out, WordJobs J, int triple_blocks) {
    if (static_cast<int>(blockIdx.x) >= triple_blocks) {
        run_word_jobs(J, static_cast<int>(blockIdx.x) - triple_blocks, static_cast<int>(gridDim.x) - triple_blocks, 256);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (s >= a.n) return;
    query_fwd_body<T, SCORER>(qa, s, lane, query);
    triple_fwd_body<T, SCORER>(a, s, lane, out);
}

// ----------------------------------------------------------------- backward
template <bool ACCUM>
__device__ __forceinline__ void put(float* p, float v) {
    if (ACCUM) *p += v;
    else *p = v;
}
// BYROW (k_query_triple_bwd_parts): the entity gradients are ADDED (fp32 atomics) into matrices over the row space of
// the entities' tables, at the row ids the triple names - where bess_direct_update picks them up - instead of being
// stored as row s of dense [n_triple, W] arrays
template <bool BYROW>
__device__ __forceinline__ void emit(float* p, float v) {
    if (BYROW) {
        if (v != 0.f) unsafeAtomicAdd(p, v);
    } else {
        *p = v;
    }
}
template <bool BYROW, bool ACCUM>
__device__ __forceinline__ void emit2(float* p, float v) {
    if (BYROW) {
        if (v != 0.f) unsafeAtomicAdd(p, v);
    } else {
        put<ACCUM>(p, v);
    }
}

template <typename T, int SCORER, bool BYROW = false>
__device__ __forceinline__ void triple_bwd_body(const TripleArgs& a, int64_t s, int lane,
                                                const float g, float* __restrict__ d_head,
                                                float* __restrict__ d_tail, float* __restrict__ d_rel) {
    const T* h = row_ptr(static_cast<const T*>(a.head_base), a.head_idx, s, a.W);
    const T* t = row_ptr(static_cast<const T*>(a.tail_base), a.tail_idx, s, a.W);
    const int64_t rid = a.rel_idx[s];
    const T* r = static_cast<const T*>(a.rel_table) + rid * a.Wr;
    float* dh = d_head + (BYROW ? (a.head_idx ? static_cast<int64_t>(a.head_idx[s]) : s) : s) * a.W;
    float* dt = d_tail + (BYROW ? (a.tail_idx ? static_cast<int64_t>(a.tail_idx[s]) : s) : s) * a.W;
    float* dr = d_rel + rid * a.Wr;

    if (SCORER == BESS_TRANSE) {
        float inv = 0.f;
        const float pf = static_cast<float>(a.norm_p);
        if (a.norm_p != 1) {
            float ss = 0.f;
            for (int e = lane; e < a.W; e += 64) {
                const float x = to_f32(h[e]) + to_f32(r[e]) - to_f32(t[e]);
                ss += lp_term(x, pf);
            }
            ss = wave_allreduce_sum(ss);
            inv = lp_inv(lp_root(ss, pf), pf);
        }
        for (int e = lane; e < a.W; e += 64) {
            const float x = to_f32(h[e]) + to_f32(r[e]) - to_f32(t[e]);
            const float dx = -g * ((a.norm_p == 1) ? sgnf(x) : lp_dterm(x, pf) * inv);
            emit<BYROW>(dh + e, dx);
            emit<BYROW>(dt + e, -dx);
            if (dx != 0.f) unsafeAtomicAdd(dr + e, dx);
        }
    } else if (SCORER == BESS_DISTMULT) {
        for (int e = lane; e < a.W; e += 64) {
            const float hv = to_f32(h[e]), rv = to_f32(r[e]), tv = to_f32(t[e]);
            emit<BYROW>(dh + e, g * rv * tv);
            emit<BYROW>(dt + e, g * hv * rv);
            unsafeAtomicAdd(dr + e, g * hv * tv);
        }
    } else {
        const int d = a.W / 2;
        float inv = 0.f;
        const float pf = static_cast<float>(a.norm_p);
        if (SCORER == BESS_ROTATE && a.norm_p != 1) {
            float ss = 0.f;
            for (int e = lane; e < d; e += 64) {
                const float hr = to_f32(h[e]), hi = to_f32(h[d + e]);
                const float ph = to_f32(r[e]);
                const float c = cosf(ph), sn = sinf(ph);
                const float xr = hr * c - hi * sn - to_f32(t[e]);
                const float xi = hr * sn + hi * c - to_f32(t[d + e]);
                ss += lp_term(xr, pf) + lp_term(xi, pf);
            }
            ss = wave_allreduce_sum(ss);
            inv = lp_inv(lp_root(ss, pf), pf);
        }
        for (int e = lane; e < d; e += 64) {
            const float hr = to_f32(h[e]), hi = to_f32(h[d + e]);
            const float tr = to_f32(t[e]), ti = to_f32(t[d + e]);
            if (SCORER == BESS_ROTATE) {
                const float ph = to_f32(r[e]);
                const float c = cosf(ph), sn = sinf(ph);
                const float xr = hr * c - hi * sn - tr;
                const float xi = hr * sn + hi * c - ti;
                const float dxr = -g * ((a.norm_p == 1) ? sgnf(xr) : lp_dterm(xr, pf) * inv);
                const float dxi = -g * ((a.norm_p == 1) ? sgnf(xi) : lp_dterm(xi, pf) * inv);
                emit<BYROW>(dh + e, dxr * c + dxi * sn);
                emit<BYROW>(dh + d + e, -dxr * sn + dxi * c);
                emit<BYROW>(dt + e, -dxr);
                emit<BYROW>(dt + d + e, -dxi);
                const float dph = dxr * (-hr * sn - hi * c) + dxi * (hr * c - hi * sn);
                if (dph != 0.f) unsafeAtomicAdd(dr + e, dph);
            } else {
                const float rr = to_f32(r[e]), ri = to_f32(r[d + e]);
                emit<BYROW>(dh + e, g * (rr * tr + ri * ti));
                emit<BYROW>(dh + d + e, g * (-ri * tr + rr * ti));
                emit<BYROW>(dt + e, g * (hr * rr - hi * ri));
                emit<BYROW>(dt + d + e, g * (hr * ri + hi * rr));
                unsafeAtomicAdd(dr + e, g * (hr * tr + hi * ti));
                unsafeAtomicAdd(dr + d + e, g * (-hi * tr + hr * ti));
            }
        }
    }
}

template <typename T, int SCORER>
__global__ __launch_bounds__(256) void k_score_triple_bwd(TripleArgs a,
                                                          const float* __restrict__ d_out,
                                                          float* __restrict__ d_head,
                                                          float* __restrict__ d_tail,
                                                          float* __restrict__ d_rel) {
    const int lane = threadIdx.x & 63;
    const int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (s >= a.n) return;
    triple_bwd_body<T, SCORER>(a, s, lane, d_out[s], d_head, d_tail, d_rel);
}

// ACCUM: d_ent already holds this lane's contribution of the positive score (written by the same lane
// just before: same element -> lane mapping in both bodies); add instead of overwrite
template <typename T, int SCORER, bool ACCUM, bool BYROW = false>
__device__ __forceinline__ void query_bwd_body(const QueryArgs& a, int64_t q, int lane,
                                               const float* dq_row, float* __restrict__ d_ent,
                                               float* __restrict__ d_rel) {
    const T* x = row_ptr(static_cast<const T*>(a.ent_base), a.ent_idx, q, a.W);
    const int64_t rid = a.rel_idx[q];
    const T* r = static_cast<const T*>(a.rel_table) + rid * a.Wr;
    const float* dq = dq_row;
    float* dx = d_ent + (BYROW ? (a.ent_idx ? static_cast<int64_t>(a.ent_idx[q]) : q) : q) * a.W;
    float* dr = d_rel + rid * a.Wr;
    const bool tail = a.side == BESS_CORRUPT_TAIL;
    if (SCORER == BESS_TRANSE) {
        for (int e = lane; e < a.W; e += 64) {
            const float g = dq[e];
            emit2<BYROW, ACCUM>(dx + e, g);
            if (g != 0.f) unsafeAtomicAdd(dr + e, tail ? g : -g);
        }
    } else if (SCORER == BESS_DISTMULT) {
        for (int e = lane; e < a.W; e += 64) {
            const float g = dq[e];
            emit2<BYROW, ACCUM>(dx + e, g * to_f32(r[e]));
            unsafeAtomicAdd(dr + e, g * to_f32(x[e]));
        }
    } else {
        const int d = a.W / 2;
        for (int e = lane; e < d; e += 64) {
            const float xr = to_f32(x[e]), xi = to_f32(x[d + e]);
            const float gr = dq[e], gi = dq[d + e];
            if (SCORER == BESS_ROTATE) {
                const float sg = tail ? 1.f : -1.f;
                const float ph = sg * to_f32(r[e]);
                const float c = cosf(ph), sn = sinf(ph);
                emit2<BYROW, ACCUM>(dx + e, gr * c + gi * sn);
                emit2<BYROW, ACCUM>(dx + d + e, -gr * sn + gi * c);
                const float dph = gr * (-xr * sn - xi * c) + gi * (xr * c - xi * sn);
                unsafeAtomicAdd(dr + e, sg * dph);
            } else {
                const float sg = tail ? 1.f : -1.f;  // conj r for heads
                const float rr = to_f32(r[e]), ri = sg * to_f32(r[d + e]);
                emit2<BYROW, ACCUM>(dx + e, gr * rr + gi * ri);
                emit2<BYROW, ACCUM>(dx + d + e, -gr * ri + gi * rr);
                unsafeAtomicAdd(dr + e, gr * xr + gi * xi);
                unsafeAtomicAdd(dr + d + e, sg * (-gr * xi + gi * xr));
            }
        }
    }
}

// K3' + K6' of ONE triple in one pass over its rows: the gradient of the positive score w.r.t. head / tail / relation
// and of the query w.r.t. the entity it was built from (the head for tail corruption, else the tail), summed where
// they meet - every gradient element is written once and the relation row takes ONE atomic per element (the two
// bodies above, run one after the other, read the rows twice, read-modify-write the entity's gradient and send two
// atomics per relation element: 4 M instead of 2 M at the C2 micro-batch).
template <typename T, int SCORER, bool BYROW = false>
__device__ __forceinline__ void query_triple_bwd_body(const TripleArgs& a, int side, int64_t s, int lane, const float g,
                                                      const float* dq, float* __restrict__ d_head,
                                                      float* __restrict__ d_tail, float* __restrict__ d_rel) {
    const T* h = row_ptr(static_cast<const T*>(a.head_base), a.head_idx, s, a.W);
    const T* t = row_ptr(static_cast<const T*>(a.tail_base), a.tail_idx, s, a.W);
    const int64_t rid = a.rel_idx[s];
    const T* r = static_cast<const T*>(a.rel_table) + rid * a.Wr;
    float* dh = d_head + (BYROW ? (a.head_idx ? static_cast<int64_t>(a.head_idx[s]) : s) : s) * a.W;
    float* dt = d_tail + (BYROW ? (a.tail_idx ? static_cast<int64_t>(a.tail_idx[s]) : s) : s) * a.W;
    float* dr = d_rel + rid * a.Wr;
    const bool tl = side == BESS_CORRUPT_TAIL;  // the query was built from the head
    const float sg = tl ? 1.f : -1.f;
    auto add_rel = [&](float* p, float v) {
        if (v != 0.f) unsafeAtomicAdd(p, v);
    };
    if (SCORER == BESS_TRANSE) {
        float inv = 0.f;
        const float pf = static_cast<float>(a.norm_p);
        if (a.norm_p != 1) {
            float ss = 0.f;
            for (int e = lane; e < a.W; e += 64) ss += lp_term(to_f32(h[e]) + to_f32(r[e]) - to_f32(t[e]), pf);
            ss = wave_allreduce_sum(ss);
            inv = lp_inv(lp_root(ss, pf), pf);
        }
        for (int e = lane; e < a.W; e += 64) {
            const float x = to_f32(h[e]) + to_f32(r[e]) - to_f32(t[e]);
            const float dx = -g * ((a.norm_p == 1) ? sgnf(x) : lp_dterm(x, pf) * inv);
            const float gq = dq[e];
            emit<BYROW>(dh + e, tl ? dx + gq : dx);
            emit<BYROW>(dt + e, tl ? -dx : -dx + gq);
            add_rel(dr + e, dx + sg * gq);
        }
    } else if (SCORER == BESS_DISTMULT) {
        for (int e = lane; e < a.W; e += 64) {
            const float hv = to_f32(h[e]), rv = to_f32(r[e]), tv = to_f32(t[e]);
            const float gq = dq[e];
            float vh = g * rv * tv, vt = g * hv * rv;
            if (tl) vh = fmaf(gq, rv, vh);
            else vt = fmaf(gq, rv, vt);
            emit<BYROW>(dh + e, vh);
            emit<BYROW>(dt + e, vt);
            add_rel(dr + e, fmaf(gq, tl ? hv : tv, g * hv * tv));
        }
    } else {
        const int d = a.W / 2;
        float inv = 0.f;
        const float pf = static_cast<float>(a.norm_p);
        if (SCORER == BESS_ROTATE && a.norm_p != 1) {
            float ss = 0.f;
            for (int e = lane; e < d; e += 64) {
                const float hr = to_f32(h[e]), hi = to_f32(h[d + e]);
                const float ph = to_f32(r[e]);
                const float c = cosf(ph), sn = sinf(ph);
                ss += lp_term(hr * c - hi * sn - to_f32(t[e]), pf) + lp_term(hr * sn + hi * c - to_f32(t[d + e]), pf);
            }
            ss = wave_allreduce_sum(ss);
            inv = lp_inv(lp_root(ss, pf), pf);
        }
        for (int e = lane; e < d; e += 64) {
            const float hr = to_f32(h[e]), hi = to_f32(h[d + e]);
            const float tr = to_f32(t[e]), ti = to_f32(t[d + e]);
            const float gr = dq[e], gi = dq[d + e];
            const float xr = tl ? hr : tr, xi = tl ? hi : ti;  // the entity the query was built from
            float vhr, vhi, vtr, vti;
            if (SCORER == BESS_ROTATE) {
                const float ph = to_f32(r[e]);
                const float c = cosf(ph), sn = sinf(ph);
                const float yr = hr * c - hi * sn - tr;
                const float yi = hr * sn + hi * c - ti;
                const float dxr = -g * ((a.norm_p == 1) ? sgnf(yr) : lp_dterm(yr, pf) * inv);
                const float dxi = -g * ((a.norm_p == 1) ? sgnf(yi) : lp_dterm(yi, pf) * inv);
                vhr = dxr * c + dxi * sn;
                vhi = -dxr * sn + dxi * c;
                vtr = -dxr;
                vti = -dxi;
                // the query rotates by sg * phase: cos is even, sin odd
                const float snq = sg * sn;
                const float qr = gr * c + gi * snq, qi = -gr * snq + gi * c;
                if (tl) vhr += qr, vhi += qi;
                else vtr += qr, vti += qi;
                const float dph = dxr * (-hr * sn - hi * c) + dxi * (hr * c - hi * sn);
                const float dphq = gr * (-xr * snq - xi * c) + gi * (xr * c - xi * snq);
                add_rel(dr + e, dph + sg * dphq);
            } else {
                const float rr = to_f32(r[e]), ri = to_f32(r[d + e]);
                vhr = g * (rr * tr + ri * ti);
                vhi = g * (-ri * tr + rr * ti);
                vtr = g * (hr * rr - hi * ri);
                vti = g * (hr * ri + hi * rr);
                const float riq = sg * ri;  // conj r for heads
                const float qr = gr * rr + gi * riq, qi = -gr * riq + gi * rr;
                if (tl) vhr += qr, vhi += qi;
                else vtr += qr, vti += qi;
                unsafeAtomicAdd(dr + e, g * (hr * tr + hi * ti) + (gr * xr + gi * xi));
                unsafeAtomicAdd(dr + d + e, g * (-hi * tr + hr * ti) + sg * (-gr * xi + gi * xr));
            }
            emit<BYROW>(dh + e, vhr);
            emit<BYROW>(dh + d + e, vhi);
            emit<BYROW>(dt + e, vtr);
            emit<BYROW>(dt + d + e, vti);
        }
    }
}

template <typename T, int SCORER>
__global__ __launch_bounds__(256) void k_query_bwd(QueryArgs a, const float* __restrict__ d_query,
                                                   float* __restrict__ d_ent,
                                                   float* __restrict__ d_rel) {
    const int lane = threadIdx.x & 63;
    const int64_t q = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (q >= a.n) return;
    query_bwd_body<T, SCORER, false>(a, q, lane, d_query + q * a.W, d_ent, d_rel);
}

// K3' + K6' in one launch: gradients of the positive score w.r.t. head / tail rows and of the query w.r.t. the
// entity it was built from, summed where both hit the same row (the query of a tail-corruption step is built
// from the head: d_head = d pos / d h + d query / d h), relation gradients accumulated once
template <typename T, int SCORER>
__global__ __launch_bounds__(256) void k_query_triple_bwd(TripleArgs a, QueryArgs qa, const float* __restrict__ d_out,
                                                          const float* __restrict__ d_query,
                                                          float* __restrict__ d_head, float* __restrict__ d_tail,
                                                          float* __restrict__ d_rel) {
    const int lane = threadIdx.x & 63;
    const int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (s >= a.n) return;
    query_triple_bwd_body<T, SCORER>(a, qa.side, s, lane, d_out[s], d_query + s * qa.W, d_head, d_tail, d_rel);
}

// The S-row tail of a training step with per-triple negatives scored by the fused forward, in ONE launch
// (was k_combine_dq + k_loss_rows + k_sum_rows + k_query_triple_bwd: 47 us of launches for ~25 MB at the C2
// micro-batch - latency, not bandwidth).  All four are per triple; a wave takes one triple through them:
//   d loss / d query from the forward's partials -> a row in LDS (never written to memory unless asked for),
//   K8: the triple's loss term, d loss / d positive score (kept in a register), the gradient row of its scores,
//   K3' + K6': the gradients of head, tail and relation rows from the two.
// The workgroup that finishes last sums the row terms in a fixed order (16 waves per workgroup at large
// micro-batches; tickets in two levels, common.h: last_workgroup_ticket).
struct TailArgs {
    const float* st_ml;
    const float* st_acc;
    int items;
    bess_loss_desc l;
    const float* pos;
    const float* neg;
    int64_t n_neg, ld_neg;
    const float* weight;
    int64_t weight_len;
    float* row_loss;
    float* loss;
    float* d_pos;
    float* d_neg;
    int64_t ld_dneg;
    float* d_query;  // [n, W] or NULL
    int32_t* counter;
};

template <typename T, int SCORER, int CH>
__global__ __launch_bounds__(1024) void k_pertriple_tail(TripleArgs a, QueryArgs qa, TailArgs t, float* __restrict__ d_head,
                                                         float* __restrict__ d_tail, float* __restrict__ d_rel) {
    extern __shared__ float tail_lds[];  // [waves][W] rows of d loss / d query; then the partial sums of the loss
    __shared__ int last;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int64_t s = static_cast<int64_t>(blockIdx.x) * nw + wv;
    if (s < a.n) {  // (whole waves)
        float* row = tail_lds + wv * a.W;
        combine_dq_row(t.st_ml, t.st_acc, s, t.items, a.W, t.l.kind, t.l.loss_scale, t.pos, t.weight, t.weight_len, nullptr,
                       row, t.d_query ? t.d_query + s * a.W : nullptr);
        float g;
#define BESS_TAIL_LOSS(KIND, ADV)                                                                                        \
    g = loss_row_impl<KIND, ADV, true, CH>(t.l, t.pos, t.neg, s, t.n_neg, t.ld_neg, t.weight, t.weight_len, t.row_loss, \
                                           t.d_pos, t.d_neg, t.ld_dneg, nullptr)
        if (t.l.kind == BESS_LOSS_SSCE) BESS_TAIL_LOSS(BESS_LOSS_SSCE, false);
        else if (t.l.kind == BESS_LOSS_LOGSIGMOID && t.l.adversarial) BESS_TAIL_LOSS(BESS_LOSS_LOGSIGMOID, true);
        else if (t.l.kind == BESS_LOSS_LOGSIGMOID) BESS_TAIL_LOSS(BESS_LOSS_LOGSIGMOID, false);
        else if (t.l.adversarial) BESS_TAIL_LOSS(BESS_LOSS_MARGIN, true);
        else BESS_TAIL_LOSS(BESS_LOSS_MARGIN, false);
#undef BESS_TAIL_LOSS
        // (the row in LDS was written by lanes of this wave, whose LDS accesses execute in order)
        query_triple_bwd_body<T, SCORER>(a, qa.side, s, lane, g, row, d_head, d_tail, d_rel);
    }
    // the sum of the row terms: as the one-launch form of k_loss_rows (loss.hip) - same order as k_sum_rows when the
    // workgroup has 1024 threads
    release_to_agent();
    __syncthreads();
    if (threadIdx.x == 0) last = last_workgroup_ticket(t.counter);
    __syncthreads();
    if (!last) return;
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < a.n; i += blockDim.x) acc += load_agent(t.row_loss + i);
    tail_lds[threadIdx.x] = acc;
    __syncthreads();
    for (int h = blockDim.x >> 1; h > 0; h >>= 1) {
        if (static_cast<int>(threadIdx.x) < h) tail_lds[threadIdx.x] += tail_lds[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) t.loss[0] = tail_lds[0];
}

// The same behind bess_neg_score_shared_bwd_parts, by row: d_query arrives as
// n_dq slabs [n_triple, W] that are summed as the triple's wave reads them (into LDS: the body then reads the row
// from there), and spare workgroups behind the triples' add the candidates' gradient rows - the sum of n_de slabs
// [n_neg, W] - into the accumulator at the candidates' row ids: one atomic per (candidate, column) instead of one
// per query slice.  LDS: 4 waves x W floats (dynamic).
template <typename T, int SCORER>
__global__ __launch_bounds__(256) void k_query_triple_bwd_parts(TripleArgs a, QueryArgs qa, const float* __restrict__ d_out,
                                                                const float* __restrict__ dq_parts, int n_dq,
                                                                const float* __restrict__ de_parts, int n_de, int64_t n_neg,
                                                                const int32_t* __restrict__ neg_idx,
                                                                float* __restrict__ acc_head, float* __restrict__ acc_tail,
                                                                float* __restrict__ acc_neg, float* __restrict__ d_rel,
                                                                int triple_blocks) {
    extern __shared__ float dq_rows[];  // [4][W]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (static_cast<int>(blockIdx.x) >= triple_blocks) {
        const int64_t n_wave = (static_cast<int64_t>(gridDim.x) - triple_blocks) * 4;
        for (int64_t j = (static_cast<int64_t>(blockIdx.x) - triple_blocks) * 4 + wv; j < n_neg; j += n_wave) {
            float* dst = acc_neg + static_cast<int64_t>(neg_idx[j]) * a.W;
            if ((a.W & 3) == 0) {  // 16 bytes per lane, the slabs' loads in flight together
                for (int e = lane * 4; e < a.W; e += 256) {
                    const float* src = de_parts + j * a.W + e;
                    const int64_t slab = n_neg * a.W;
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 16
                    for (int p = 0; p < n_de; ++p) {
                        const float4 v = *reinterpret_cast<const float4*>(src + p * slab);
                        s0 += v.x, s1 += v.y, s2 += v.z, s3 += v.w;
                    }
                    if (s0 != 0.f) unsafeAtomicAdd(dst + e, s0);
                    if (s1 != 0.f) unsafeAtomicAdd(dst + e + 1, s1);
                    if (s2 != 0.f) unsafeAtomicAdd(dst + e + 2, s2);
                    if (s3 != 0.f) unsafeAtomicAdd(dst + e + 3, s3);
                }
                continue;
            }
            for (int e = lane; e < a.W; e += 64) {
                float sum = 0.f;
                for (int p = 0; p < n_de; ++p) sum += de_parts[(static_cast<int64_t>(p) * n_neg + j) * a.W + e];
                if (sum != 0.f) unsafeAtomicAdd(dst + e, sum);
            }
        }
        return;
    }
    const int64_t s = blockIdx.x * 4ll + wv;
    if (s >= a.n) return;  // (whole waves; no barrier in this kernel)
    float* row = dq_rows + wv * a.W;  // (read back below by the lanes of this wave, whose LDS accesses execute in order)
    if ((a.W & 3) == 0) {
        const int64_t slab = a.n * a.W;
        for (int e = lane * 4; e < a.W; e += 256) {
            const float* src = dq_parts + s * a.W + e;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 16
            for (int p = 0; p < n_dq; ++p) {
                const float4 v = *reinterpret_cast<const float4*>(src + p * slab);
                s0 += v.x, s1 += v.y, s2 += v.z, s3 += v.w;
            }
            *reinterpret_cast<float4*>(row + e) = make_float4(s0, s1, s2, s3);
        }
    } else {
        for (int e = lane; e < a.W; e += 64) {
            float sum = 0.f;
            for (int p = 0; p < n_dq; ++p) sum += dq_parts[(static_cast<int64_t>(p) * a.n + s) * a.W + e];
            row[e] = sum;
        }
    }
    query_triple_bwd_body<T, SCORER, true>(a, qa.side, s, lane, d_out[s], row, acc_head, acc_tail, d_rel);
}

template <template <typename, int> class Launcher, typename... Args>
static int dispatch(const bess_model_desc* d, Args... args) {
#define BESS_CASE(SC)                                                   \
    case SC:                                                            \
        if (d->dtype == BESS_F32) Launcher<float, SC>::run(args...);    \
        else Launcher<half_t, SC>::run(args...);                        \
        break;
    switch (d->scorer) {
        BESS_CASE(BESS_TRANSE)
        BESS_CASE(BESS_ROTATE)
        BESS_CASE(BESS_DISTMULT)
        BESS_CASE(BESS_COMPLEX)
    }
#undef BESS_CASE
    return BESS_OK;
}

template <typename T, int SC>
struct LTripleFwd {
    static void run(TripleArgs a, float* out, hipStream_t st) {
        k_score_triple_fwd<T, SC><<<ceil_div(a.n, 4), 256, 0, st>>>(a, out);
    }
};
template <typename T, int SC>
struct LTripleBwd {
    static void run(TripleArgs a, const float* d_out, float* dh, float* dt, float* dr, hipStream_t st) {
        k_score_triple_bwd<T, SC><<<ceil_div(a.n, 4), 256, 0, st>>>(a, d_out, dh, dt, dr);
    }
};
template <typename T, int SC>
struct LQueryFwd {
    static void run(QueryArgs a, float* q, hipStream_t st) {
        k_query_fwd<T, SC><<<ceil_div(a.n, 4), 256, 0, st>>>(a, q);
    }
};
template <typename T, int SC>
struct LQueryBwd {
    static void run(QueryArgs a, const float* dq, float* dx, float* dr, hipStream_t st) {
        k_query_bwd<T, SC><<<ceil_div(a.n, 4), 256, 0, st>>>(a, dq, dx, dr);
    }
};

template <typename T, int SC>
struct LQueryTripleFwd {
    static void run(TripleArgs a, QueryArgs qa, float* q, float* out, hipStream_t st) {
        k_query_triple_fwd<T, SC><<<ceil_div(a.n, 4), 256, 0, st>>>(a, qa, q, out);
    }
};
template <typename T, int SC>
struct LQueryTripleFwdJobs {
    static void run(TripleArgs a, QueryArgs qa, float* q, float* out, WordJobs J, int job_blocks, hipStream_t st) {
        const int tb = static_cast<int>(ceil_div(a.n, 4));
        k_query_triple_fwd_jobs<T, SC><<<tb + job_blocks, 256, 0, st>>>(a, qa, q, out, J, tb);
    }
};
template <typename T, int SC>
struct LQueryTripleBwdParts {
    static void run(TripleArgs a, QueryArgs qa, const float* d_out, const float* dqp, int n_dq, const float* dep, int n_de,
                    int64_t n_neg, const int32_t* neg_idx, float* ah, float* at, float* an, float* dr, hipStream_t st) {
        const int tb = static_cast<int>(ceil_div(a.n, 4));
        const int nb = static_cast<int>(std::min<int64_t>(ceil_div(n_neg, 4), 1024));
        k_query_triple_bwd_parts<T, SC><<<tb + nb, 256, 4 * a.W * sizeof(float), st>>>(a, qa, d_out, dqp, n_dq, dep, n_de,
                                                                                      n_neg, neg_idx, ah, at, an, dr, tb);
    }
};
template <typename T, int SC>
struct LQueryTripleBwd {
    static void run(TripleArgs a, QueryArgs qa, const float* d_out, const float* dq, float* dh, float* dt, float* dr,
                    hipStream_t st) {
        k_query_triple_bwd<T, SC><<<ceil_div(a.n, 4), 256, 0, st>>>(a, qa, d_out, dq, dh, dt, dr);
    }
};

template <typename T, int SC>
struct LPertripleTail {
    static void run(TripleArgs a, QueryArgs qa, TailArgs t, float* dh, float* dt, float* dr, int ch, hipStream_t st) {
        // 16 waves (one triple each) per workgroup once that still fills the chip, fewer when their rows would not fit
        // 32 KB of LDS (W = 4096: two waves)
        int nw = a.n > 1024 ? 16 : 4;
        while (nw > 1 && static_cast<size_t>(nw) * a.W * sizeof(float) > (32u << 10)) nw >>= 1;
        const unsigned grid = static_cast<unsigned>(ceil_div(a.n, nw));
        const size_t lds = sizeof(float) * static_cast<size_t>(std::max(nw * a.W, 64 * nw));
        if (ch == 4) k_pertriple_tail<T, SC, 4><<<grid, 64 * nw, lds, st>>>(a, qa, t, dh, dt, dr);
        else k_pertriple_tail<T, SC, 12><<<grid, 64 * nw, lds, st>>>(a, qa, t, dh, dt, dr);
    }
};

}  // namespace bess

using namespace bess;

static int triple_args(const bess_model_desc* d, const void* hb, const int32_t* hi, const void* tb,
                       const int32_t* ti, const void* rt, const int32_t* ri, int64_t n,
                       TripleArgs* a) {
    if (int e = check_desc(d)) return e;
    BESS_REQUIRE(n >= 0, "score_triple: negative n_triple");
    BESS_REQUIRE(n == 0 || (hb && tb && rt && ri), "score_triple: NULL pointer");
    *a = TripleArgs{hb, hi, tb, ti, rt, ri, n, d->width, d->rel_width, d->norm_p};
    return BESS_OK;
}

extern "C" int bess_score_triple_fwd(const bess_model_desc* d, const void* head_base,
                                     const int32_t* head_idx, const void* tail_base,
                                     const int32_t* tail_idx, const void* rel_table,
                                     const int32_t* rel_idx, int64_t n_triple, float* out,
                                     void* stream) {
    TripleArgs a;
    if (int e = triple_args(d, head_base, head_idx, tail_base, tail_idx, rel_table, rel_idx, n_triple, &a))
        return e;
    if (n_triple == 0) return BESS_OK;
    BESS_REQUIRE(out, "score_triple_fwd: NULL out");
    dispatch<LTripleFwd>(d, a, out, as_stream(stream));
    return check_launch("score_triple_fwd");
}

extern "C" int bess_score_triple_bwd(const bess_model_desc* d, const void* head_base,
                                     const int32_t* head_idx, const void* tail_base,
                                     const int32_t* tail_idx, const void* rel_table,
                                     const int32_t* rel_idx, int64_t n_triple, const float* d_out,
                                     float* d_head, float* d_tail, float* d_rel_table,
                                     void* stream) {
    TripleArgs a;
    if (int e = triple_args(d, head_base, head_idx, tail_base, tail_idx, rel_table, rel_idx, n_triple, &a))
        return e;
    if (n_triple == 0) return BESS_OK;
    BESS_REQUIRE(d_out && d_head && d_tail && d_rel_table, "score_triple_bwd: NULL pointer");
    dispatch<LTripleBwd>(d, a, d_out, d_head, d_tail, d_rel_table, as_stream(stream));
    return check_launch("score_triple_bwd");
}

static int query_args(const bess_model_desc* d, int32_t side, const void* eb, const int32_t* ei,
                      const void* rt, const int32_t* ri, int64_t n, QueryArgs* a) {
    if (int e = check_desc(d)) return e;
    BESS_REQUIRE(side == BESS_CORRUPT_HEAD || side == BESS_CORRUPT_TAIL, "query: bad side %d", side);
    BESS_REQUIRE(n >= 0, "query: negative n_query");
    BESS_REQUIRE(n == 0 || (eb && rt && ri), "query: NULL pointer");
    *a = QueryArgs{eb, ei, rt, ri, n, d->width, d->rel_width, side};
    return BESS_OK;
}

extern "C" int bess_query_fwd(const bess_model_desc* d, int32_t side, const void* ent_base,
                              const int32_t* ent_idx, const void* rel_table,
                              const int32_t* rel_idx, int64_t n_query, float* query, void* stream) {
    if (d && d->scorer == BESS_AFFINE) {
        if (int e = check_desc(d)) return e;
        return affine_query_fwd(d, side, ent_base, ent_idx, rel_table, rel_idx, n_query, query, as_stream(stream));
    }
    QueryArgs a;
    if (int e = query_args(d, side, ent_base, ent_idx, rel_table, rel_idx, n_query, &a)) return e;
    if (n_query == 0) return BESS_OK;
    BESS_REQUIRE(query, "query_fwd: NULL out");
    dispatch<LQueryFwd>(d, a, query, as_stream(stream));
    return check_launch("query_fwd");
}

extern "C" int bess_query_bwd(const bess_model_desc* d, int32_t side, const void* ent_base,
                              const int32_t* ent_idx, const void* rel_table,
                              const int32_t* rel_idx, int64_t n_query, const float* d_query,
                              float* d_ent, float* d_rel_table, void* stream) {
    if (d && d->scorer == BESS_AFFINE) {
        if (int e = check_desc(d)) return e;
        return affine_query_bwd(d, side, ent_base, ent_idx, rel_table, rel_idx, n_query, d_query, d_ent, d_rel_table,
                                as_stream(stream));
    }
    QueryArgs a;
    if (int e = query_args(d, side, ent_base, ent_idx, rel_table, rel_idx, n_query, &a)) return e;
    if (n_query == 0) return BESS_OK;
    BESS_REQUIRE(d_query && d_ent && d_rel_table, "query_bwd: NULL pointer");
    dispatch<LQueryBwd>(d, a, d_query, d_ent, d_rel_table, as_stream(stream));
    return check_launch("query_bwd");
}

extern "C" int bess_query_triple_fwd(const bess_model_desc* d, int32_t side, const void* head_base,
                                     const int32_t* head_idx, const void* tail_base, const int32_t* tail_idx,
                                     const void* rel_table, const int32_t* rel_idx, int64_t n_triple, float* query,
                                     float* out, void* stream) {
    TripleArgs a;
    if (int e = triple_args(d, head_base, head_idx, tail_base, tail_idx, rel_table, rel_idx, n_triple, &a))
        return e;
    BESS_REQUIRE(d->scorer <= BESS_COMPLEX, "query_triple_fwd: TransE / RotatE / DistMult / ComplEx only");
    QueryArgs qa;
    const bool tail = side == BESS_CORRUPT_TAIL;
    if (int e = query_args(d, side, tail ? head_base : tail_base, tail ? head_idx : tail_idx, rel_table, rel_idx,
                           n_triple, &qa))
        return e;
    if (n_triple == 0) return BESS_OK;
    BESS_REQUIRE(query && out, "query_triple_fwd: NULL out");
    dispatch<LQueryTripleFwd>(d, a, qa, query, out, as_stream(stream));
    return check_launch("query_triple_fwd");
}

extern "C" int bess_query_triple_fwd_jobs(const bess_model_desc* d, int32_t side, const void* head_base,
                                          const int32_t* head_idx, const void* tail_base, const int32_t* tail_idx,
                                          const void* rel_table, const int32_t* rel_idx, int64_t n_triple, float* query,
                                          float* out, int32_t n_jobs, void* const* job_dst, const void* const* job_src,
                                          const uint32_t* job_value, const int64_t* job_words, void* stream) {
    TripleArgs a;
    if (int e = triple_args(d, head_base, head_idx, tail_base, tail_idx, rel_table, rel_idx, n_triple, &a))
        return e;
    BESS_REQUIRE(d->scorer <= BESS_COMPLEX, "query_triple_fwd_jobs: TransE / RotatE / DistMult / ComplEx only");
    QueryArgs qa;
    const bool tail = side == BESS_CORRUPT_TAIL;
    if (int e = query_args(d, side, tail ? head_base : tail_base, tail ? head_idx : tail_idx, rel_table, rel_idx,
                           n_triple, &qa))
        return e;
    WordJobs J{};
    int64_t words = 0;
    if (int e = make_word_jobs(n_jobs, job_dst, job_src, job_value, job_words, &J, &words, "query_triple_fwd_jobs")) return e;
    if (n_triple == 0 && words == 0) return BESS_OK;
    BESS_REQUIRE(n_triple == 0 || (query && out), "query_triple_fwd_jobs: NULL out");
    // ~8 words per thread of the job workgroups, at most one workgroup per CU
    const int job_blocks = static_cast<int>(std::min<int64_t>(ceil_div(words, 8 * 256), 256));
    dispatch<LQueryTripleFwdJobs>(d, a, qa, query, out, J, job_blocks, as_stream(stream));
    return check_launch("query_triple_fwd_jobs");
}

extern "C" int bess_query_triple_bwd_parts(const bess_model_desc* d, int32_t side, const void* head_base,
                                           const int32_t* head_idx, const void* tail_base, const int32_t* tail_idx,
                                           const void* rel_table, const int32_t* rel_idx, int64_t n_triple,
                                           const float* d_out, const float* dq_parts, int32_t n_dq_parts,
                                           const float* dneg_parts, int32_t n_dneg_parts, int64_t n_neg,
                                           const int32_t* neg_idx, float* acc_head, float* acc_tail, float* acc_neg,
                                           float* d_rel_table, void* stream) {
    TripleArgs a;
    if (int e = triple_args(d, head_base, head_idx, tail_base, tail_idx, rel_table, rel_idx, n_triple, &a))
        return e;
    BESS_REQUIRE(d->scorer <= BESS_COMPLEX, "query_triple_bwd_parts: TransE / RotatE / DistMult / ComplEx only");
    QueryArgs qa;
    const bool tail = side == BESS_CORRUPT_TAIL;
    if (int e = query_args(d, side, tail ? head_base : tail_base, tail ? head_idx : tail_idx, rel_table, rel_idx,
                           n_triple, &qa))
        return e;
    BESS_REQUIRE(n_triple > 0 && n_neg > 0 && n_dq_parts >= 1 && n_dneg_parts >= 1, "query_triple_bwd_parts: bad sizes");
    BESS_REQUIRE(d_out && dq_parts && dneg_parts && neg_idx && head_idx && tail_idx && acc_head && acc_tail && acc_neg &&
                     d_rel_table, "query_triple_bwd_parts: NULL pointer (rows are named by index: by-row accumulators)");
    BESS_REQUIRE(d->width <= 4096, "query_triple_bwd_parts: rows of %d scalars (at most 4096)", d->width);
    dispatch<LQueryTripleBwdParts>(d, a, qa, d_out, dq_parts, n_dq_parts, dneg_parts, n_dneg_parts, n_neg, neg_idx, acc_head,
                                   acc_tail, acc_neg, d_rel_table, as_stream(stream));
    return check_launch("query_triple_bwd_parts");
}

extern "C" int bess_query_triple_bwd(const bess_model_desc* d, int32_t side, const void* head_base,
                                     const int32_t* head_idx, const void* tail_base, const int32_t* tail_idx,
                                     const void* rel_table, const int32_t* rel_idx, int64_t n_triple,
                                     const float* d_out, const float* d_query, float* d_head, float* d_tail,
                                     float* d_rel_table, void* stream) {
    TripleArgs a;
    if (int e = triple_args(d, head_base, head_idx, tail_base, tail_idx, rel_table, rel_idx, n_triple, &a))
        return e;
    BESS_REQUIRE(d->scorer <= BESS_COMPLEX, "query_triple_bwd: TransE / RotatE / DistMult / ComplEx only");
    QueryArgs qa;
    const bool tail = side == BESS_CORRUPT_TAIL;
    if (int e = query_args(d, side, tail ? head_base : tail_base, tail ? head_idx : tail_idx, rel_table, rel_idx,
                           n_triple, &qa))
        return e;
    if (n_triple == 0) return BESS_OK;
    BESS_REQUIRE(d_out && d_query && d_head && d_tail && d_rel_table, "query_triple_bwd: NULL pointer");
    dispatch<LQueryTripleBwd>(d, a, qa, d_out, d_query, d_head, d_tail, d_rel_table, as_stream(stream));
    return check_launch("query_triple_bwd");
}

extern "C" int bess_pertriple_tail_supported(const bess_model_desc* d, int64_t n_neg) {
    return d && d->scorer <= BESS_COMPLEX && d->width <= 4096 && n_neg > 0 && n_neg % 4 == 0 && n_neg <= 256 * 12;
}

extern "C" int bess_pertriple_tail(const bess_model_desc* d, const bess_loss_desc* l, int32_t side, const void* head_base,
                                   const int32_t* head_idx, const void* tail_base, const int32_t* tail_idx,
                                   const void* rel_table, const int32_t* rel_idx, int64_t n_triple,
                                   const float* state_ml, const float* state_acc, int32_t items, const float* pos,
                                   const float* neg, int64_t n_neg, int64_t ld_neg, const float* weight,
                                   int64_t weight_len, float* row_loss, float* loss, float* d_pos, float* d_neg,
                                   int64_t ld_dneg, float* d_query, float* d_head, float* d_tail, float* d_rel_table,
                                   int32_t* counter, void* stream) {
    TripleArgs a;
    if (int e = triple_args(d, head_base, head_idx, tail_base, tail_idx, rel_table, rel_idx, n_triple, &a)) return e;
    BESS_REQUIRE(d->scorer <= BESS_COMPLEX, "pertriple_tail: TransE / RotatE / DistMult / ComplEx only");
    QueryArgs qa;
    const bool tail = side == BESS_CORRUPT_TAIL;
    if (int e = query_args(d, side, tail ? head_base : tail_base, tail ? head_idx : tail_idx, rel_table, rel_idx, n_triple,
                           &qa))
        return e;
    BESS_REQUIRE(l, "pertriple_tail: NULL loss descriptor");
    BESS_REQUIRE(l->kind >= BESS_LOSS_LOGSIGMOID && l->kind <= BESS_LOSS_SSCE, "pertriple_tail: unknown loss %d", l->kind);
    BESS_REQUIRE(n_triple > 0 && items > 0, "pertriple_tail: bad sizes");
    if (!bess_pertriple_tail_supported(d, n_neg))
        return fail(BESS_EUNSUPPORTED, "pertriple_tail: rows of %lld scores (a multiple of 4, at most 3072) of %d scalars "
                                       "(at most 4096)", (long long)n_neg, d->width);
    BESS_REQUIRE(state_ml && state_acc && pos && neg && weight && row_loss && loss && d_pos && d_neg && d_head && d_tail &&
                     d_rel_table && counter, "pertriple_tail: NULL pointer");
    BESS_REQUIRE(weight_len == 1 || weight_len == n_triple, "pertriple_tail: weight_len must be 1 or n_triple");
    BESS_REQUIRE(ld_neg >= n_neg && ld_dneg >= n_neg && ld_neg % 4 == 0 && ld_dneg % 4 == 0 &&
                     reinterpret_cast<uintptr_t>(neg) % 16 == 0 && reinterpret_cast<uintptr_t>(d_neg) % 16 == 0,
                 "pertriple_tail: score rows must be 16-byte aligned");
    TailArgs t{state_ml, state_acc, items, *l, pos, neg, n_neg, ld_neg, weight, weight_len, row_loss, loss, d_pos, d_neg,
               ld_dneg, d_query, counter};
    dispatch<LPertripleTail>(d, a, qa, t, d_head, d_tail, d_rel_table, n_neg <= 1024 ? 4 : 12, as_stream(stream));
    return check_launch("pertriple_tail");
}

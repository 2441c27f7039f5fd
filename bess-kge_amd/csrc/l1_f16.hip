// K4 for TransE / RotatE with p = 1 on fp16 tables (BASELINE configs[3]: ogbl-wikikg2 TransE
// d=256 fp16, flat shared negatives): packed-fp16 forms of the L1 distance matrix and of its
// two backward products.
//
// The reference's fp16 mode is `model.half()`: the query `h + r` is an fp16 tensor and the
// distance matrix is computed from fp16 operands (reference scoring.py:194-197, 342, 354;
// notebooks/3_wikikg2_fp16.ipynb:300-392).  Here the query is rounded to fp16 (round to nearest
// even) when a tile is staged, candidates are the fp16 table rows as they are; everything
// after the first subtraction is exact or fp32:
//
//   forward   sum_w |q - e| = 2 sum_w max(q, e) - sum_w q - sum_w e.  max of two fp16 numbers is
//             exact, v_dot2_f32_f16 adds two of them into an fp32 accumulator: ONE lane-op per
//             element (v_pk_max_f16 + v_dot2_f32_f16 per packed pair) instead of two
//             (v_sub_f32, v_add_f32 |.|) - there is no rounded difference at all.
//   backward  dX[a, w] = sum_b c[a, b] sgn(x[a, w] - y[b, w]).  Two b per lane-op: d =
//             v_pk_add_f16(x, -y) (its sign and zero-ness are exact: fp16 keeps subnormals),
//             the bits of d read as int16 have the sign of d, so clamping them to [-1, 1]
//             (v_pk_min_i16, v_pk_max_i16) is sgn(d) with sgn(0) = 0 - the value torch's
//             autograd of the p-norm uses at a tie - and v_dot2_i32_i16 multiplies by the
//             coefficients quantised to int16 with one fp32 scale per output row
//             (c = scale * k / 32767, |k| <= 32767; the rounding remainder is diffused into the
//             next coefficient of the row, so a row keeps its sum: measured relative error of the
//             gradients 1e-5 .. 1.4e-4, at or below what fp16 coefficients would give)
//             and accumulates in int32: exact, order independent - the reduction can be split
//             over workgroups with integer atomics and stay bitwise reproducible.
//             4 lane-ops per 2 elements instead of 3 per element.
//
// Tiles: 64 x 64 outputs per 256-thread workgroup, 4 x 4 per lane, operands staged through LDS
// as packed dwords ([pair][row] images, one ds_read_b128 per operand and pair step), rows
// gathered by index straight into the image, next stage prefetched into registers.
#include "common.h"

namespace bess {

typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef short s2v __attribute__((ext_vector_type(2)));

constexpr int PT = 64;    // tile edge
constexpr int PLD = 68;   // padded leading dimension of the LDS images (dwords)
constexpr int FKH = 32;   // forward: halfs of W per stage (16 packed pairs)
constexpr int BKY = 16;   // backward: reduction rows per stage (8 packed pairs)

// single-instruction helpers (hipcc scalarises packed int16 min / max into v_cmp + v_cndmask per
// half, and puts a canonicalising v_pk_max_f16 x, x in front of a packed max of loaded bits)
__device__ __forceinline__ uint32_t pk_max_f16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_sub_f16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pack_rn(float a, float b) {
    const h2v v = {static_cast<_Float16>(a), static_cast<_Float16>(b)};  // v_cvt_f16_f32: round to nearest even
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float dot2_ones(uint32_t v, float acc) {
    const h2v ones = {static_cast<_Float16>(1.f), static_cast<_Float16>(1.f)};
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2v, v), ones, acc, false);
}

// K7 inside the forward epilogue (same rule as k_mask_scores, loss.hip)
struct KillArgs {
    int diag_step, ht, ppp;
    const uint8_t* mask;
    int64_t mask_rows, mask_cols, n_neg;
};
// what K7 does to row s: the column it kills (augmentation; -1: none) and the row of the mask that applies
__device__ __forceinline__ void kill_row(const KillArgs& k, int64_t s, int64_t* kcol, const uint8_t** mrow) {
    const int cut = k.ppp / 2;
    const int64_t blk = k.ppp > 0 ? s / k.ppp : 0;
    const int p = k.ppp > 0 ? static_cast<int>(s - blk * k.ppp) : 0;
    *kcol = -1;
    if (k.diag_step > 0) *kcol = static_cast<int64_t>(k.diag_step) * (k.ht ? (blk * cut + (p % cut)) : s);
    *mrow = nullptr;
    if (k.mask) {
        int64_t r = s;
        if (k.mask_rows == 1) r = 0;
        else if (k.mask_rows == 2) r = (p >= cut) ? 1 : 0;
        *mrow = k.mask + r * k.mask_cols;
    }
}

// out[q, j] = -sum_w |fp16(Q[q, w]) - E[idx[j], w]|        W % 32 == 0
__global__ __launch_bounds__(256) void k_l1_fwd_pk(const float* __restrict__ Q, int64_t nq,
                                                   const half_t* __restrict__ E, const int32_t* __restrict__ eidx,
                                                   int64_t ne, int W, float* __restrict__ out, int64_t ld,
                                                   KillArgs kill) {
    __shared__ __attribute__((aligned(16))) uint32_t Qs[FKH / 2][PLD];
    __shared__ __attribute__((aligned(16))) uint32_t Es[FKH / 2][PLD];
    __shared__ float rsq[PT], rse[PT];
    const int t = threadIdx.x;
    const int tx = t & 15, ty = t >> 4;
    const int64_t q0 = static_cast<int64_t>(blockIdx.y) * PT;
    const int64_t j0 = static_cast<int64_t>(blockIdx.x) * PT;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    // staging: thread t brings 8 scalars of row m of each operand per stage (rows past the end are
    // clamped, their results dropped at the store)
    const int m = t >> 2, kc = t & 3;
    const float* qp = Q + min(q0 + m, nq - 1) * W + kc * 8;
    const int64_t er = min(j0 + m, ne - 1);
    const half_t* ep = E + (eidx ? static_cast<int64_t>(eidx[er]) : er) * W + kc * 8;
    float4 qa = *reinterpret_cast<const float4*>(qp), qb = *reinterpret_cast<const float4*>(qp + 4);
    uint4 ev = *reinterpret_cast<const uint4*>(ep);
    float sq = 0.f, se = 0.f;  // sums of the staged (rounded) values of row m
    for (int k0 = 0; k0 < W; k0 += FKH) {
        const uint32_t qd[4] = {pack_rn(qa.x, qa.y), pack_rn(qa.z, qa.w), pack_rn(qb.x, qb.y), pack_rn(qb.z, qb.w)};
        const uint32_t ed[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Qs[kc * 4 + i][m] = qd[i];
            Es[kc * 4 + i][m] = ed[i];
            sq = dot2_ones(qd[i], sq);
            se = dot2_ones(ed[i], se);
        }
        __syncthreads();
        if (k0 + FKH < W) {
            qa = *reinterpret_cast<const float4*>(qp + k0 + FKH);
            qb = *reinterpret_cast<const float4*>(qp + k0 + FKH + 4);
            ev = *reinterpret_cast<const uint4*>(ep + k0 + FKH);
        }
#pragma unroll
        for (int k = 0; k < FKH / 2; ++k) {
            const uint4 a4 = *reinterpret_cast<const uint4*>(&Qs[k][ty * 4]);
            const uint4 b4 = *reinterpret_cast<const uint4*>(&Es[k][tx * 4]);
            const uint32_t a[4] = {a4.x, a4.y, a4.z, a4.w};
            const uint32_t b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = dot2_ones(pk_max_f16(a[i], b[j]), acc[i][j]);
        }
        __syncthreads();
    }
    sq += __shfl_xor(sq, 1, 64);
    sq += __shfl_xor(sq, 2, 64);
    se += __shfl_xor(se, 1, 64);
    se += __shfl_xor(se, 2, 64);
    if (kc == 0) {
        rsq[m] = sq;
        rse[m] = se;
    }
    __syncthreads();
    const bool any_kill = kill.diag_step > 0 || kill.mask;  // wave-uniform
    const int64_t mask_from = kill.n_neg - kill.mask_cols;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t q = q0 + ty * 4 + i;
        if (q >= nq) continue;
        float* o = out + q * ld;
        int64_t kcol = -1;
        const uint8_t* mrow = nullptr;
        if (any_kill) kill_row(kill, q, &kcol, &mrow);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t jj = j0 + tx * 4 + j;
            if (jj < ne) {
                float v = -((2.f * acc[i][j] - rsq[ty * 4 + i]) - rse[tx * 4 + j]);
                if (any_kill) {
                    bool kl = jj == kcol;
                    // the mask overrides the diagonal on its columns (bess.py:227-228)
                    if (mrow && jj >= mask_from) kl = mrow[jj - mask_from] == 0;
                    if (kl) v += BESS_BAD_NEGATIVE_SCORE;
                }
                o[jj] = v;
            }
        }
    }
}

// one operand of the backward: rows of W scalars, f32 (the query matrix: rounded to fp16 here, as in
// the forward) or f16 (table rows), optionally gathered
template <typename T>
struct PkRows {
    const T* base;
    const int32_t* idx;
    int64_t n;
    __device__ __forceinline__ const T* row(int64_t i, int W) const {
        return base + (idx ? static_cast<int64_t>(idx[i]) : i) * W;
    }
};
__device__ __forceinline__ _Float16 to_h(float v) { return static_cast<_Float16>(v); }
__device__ __forceinline__ _Float16 to_h(half_t v) { return v; }

// dX[x, w] = sign * sum_y coef(x, y) * sgn(fp16(X[x, w]) - fp16(Y[y, w])),   coef(x, y) = coef[x * sx + y * sy]
// xscale[x] >= max_y |coef(x, y)|.  iacc != NULL: this workgroup's slice of the reduction is added
// to the int32 image iacc[x, w] (k_l1_bwd_finish scales it); else the full sum is written to dX.
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void k_l1_bwd_pk(PkRows<TX> X, PkRows<TY> Y, int W, float sign,
                                                   const float* __restrict__ coef, int64_t sx, int64_t sy,
                                                   const float* __restrict__ xscale, float* __restrict__ dX,
                                                   int* __restrict__ iacc, int64_t y_chunk) {
    __shared__ __attribute__((aligned(16))) uint32_t Cs[BKY / 2][PLD];  // [y pair][x]: int16 pair
    __shared__ __attribute__((aligned(16))) uint32_t Ys[BKY / 2][PLD];  // [y pair][w]: fp16 pair
    const int t = threadIdx.x;
    const int tx = t & 15, ty = t >> 4;
    const int64_t x0 = static_cast<int64_t>(blockIdx.y) * PT;
    const int w0 = blockIdx.x * PT;
    const int64_t y_lo = static_cast<int64_t>(blockIdx.z) * y_chunk;
    const int64_t y_hi = min(y_lo + y_chunk, Y.n);
    if (y_lo >= y_hi) return;

    uint32_t xv[4][4];
    int acc[4][4];
    const _Float16 hz = static_cast<_Float16>(0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const TX* xr = X.row(min(x0 + ty * 4 + i, X.n - 1), W);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // + 0: a -0 becomes +0, so that x - y is +0 (int16 0) whenever x == y
            const _Float16 h = to_h(xr[min(w0 + tx * 4 + j, W - 1)]) + hz;
            const h2v d = {h, h};
            xv[i][j] = __builtin_bit_cast(uint32_t, d);
            acc[i][j] = 0;
        }
    }
    uint32_t one2 = 0x00010001u, mone2 = 0xffffffffu;
    asm volatile("" : "+v"(one2), "+v"(mone2));  // keep them in registers (no literal per instruction)

    // coefficients staged by this thread: 4 consecutive y of row cxr
    const bool along_x = sx == 1;  // lanes run along the unit-stride dimension of coef
    const int cxr = along_x ? (t & 63) : (t >> 2);
    const int cy = along_x ? (t >> 6) * 4 : (t & 3) * 4;
    const int64_t cxg = min(x0 + cxr, X.n - 1);
    const bool x_ok = x0 + cxr < X.n;
    const float sc = xscale[cxg];
    const float inv = (x_ok && sc > 0.f) ? 32767.f / sc : 0.f;
    // candidate / query rows staged by this thread: 2 columns of the rows of one y pair
    const int yp = t >> 5;
    const int ywc = min(w0 + (t & 31) * 2, W - 2);
    float cv[4];
    uint32_t yv[2];
    auto fetch = [&](int64_t yb) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t y = yb + cy + k;
            const float c = coef[cxg * sx + min(y, y_hi - 1) * sy];
            cv[k] = y < y_hi ? c : 0.f;
        }
        const TY* r0 = Y.row(min(yb + 2 * yp, y_hi - 1), W) + ywc;
        const TY* r1 = Y.row(min(yb + 2 * yp + 1, y_hi - 1), W) + ywc;
        const _Float16 a0 = to_h(r0[0]), a1 = to_h(r0[1]), b0 = to_h(r1[0]), b1 = to_h(r1[1]);
        const h2v lo = {a0, b0}, hi = {a1, b1};
        yv[0] = __builtin_bit_cast(uint32_t, lo);
        yv[1] = __builtin_bit_cast(uint32_t, hi);
    };
    // Quantisation with error diffusion along the reduction: the rounding remainder of a coefficient is
    // carried into the next one this thread stages for the same output row (a fixed order), so the
    // coefficients of a row keep their sum - a tail of many coefficients below half a unit is not lost
    // (plain rounding would drop it coherently wherever sgn(x - y) has the same sign for most y).
    float carry = 0.f;
    auto stash = [&]() {
        int kq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v = fmaf(cv[k], inv, carry);
            const float r = rintf(v);
            carry = v - r;
            kq[k] = static_cast<int>(r);
        }
        Cs[cy / 2][cxr] = (static_cast<uint32_t>(kq[0]) & 0xffffu) | (static_cast<uint32_t>(kq[1]) << 16);
        Cs[cy / 2 + 1][cxr] = (static_cast<uint32_t>(kq[2]) & 0xffffu) | (static_cast<uint32_t>(kq[3]) << 16);
        *reinterpret_cast<uint2*>(&Ys[yp][(t & 31) * 2]) = make_uint2(yv[0], yv[1]);
    };

    fetch(y_lo);
    for (int64_t yb = y_lo; yb < y_hi; yb += BKY) {
        stash();
        __syncthreads();
        if (yb + BKY < y_hi) fetch(yb + BKY);
#pragma unroll
        for (int k = 0; k < BKY / 2; ++k) {
            const uint4 c4 = *reinterpret_cast<const uint4*>(&Cs[k][ty * 4]);
            const uint4 y4 = *reinterpret_cast<const uint4*>(&Ys[k][tx * 4]);
            const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
            const uint32_t y[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t d = pk_sub_f16(xv[i][j], y[j]);
                    const uint32_t sg = pk_max_i16(pk_min_i16(d, one2), mone2);
                    acc[i][j] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2v, sg), __builtin_bit_cast(s2v, c[i]),
                                                       acc[i][j], false);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t x = x0 + ty * 4 + i;
        if (x >= X.n) continue;
        const float f = sign * xscale[x] * (1.f / 32767.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int w = w0 + tx * 4 + j;
            if (w >= W) continue;
            if (iacc) atomicAdd(iacc + x * W + w, acc[i][j]);
            else dX[x * W + w] = f * static_cast<float>(acc[i][j]);
        }
    }
}

// dX = sign * xscale / 32767 * iacc, in place over the same buffer (int32 -> f32)
__global__ __launch_bounds__(256) void k_l1_bwd_finish(float* __restrict__ dX, const float* __restrict__ xscale,
                                                       int64_t nx, int W, float sign) {
    const int64_t total = nx * W;
    for (int64_t t = blockIdx.x * 256ll + threadIdx.x; t < total; t += 256ll * gridDim.x) {
        const int v = reinterpret_cast<const int*>(dX)[t];
        dX[t] = sign * xscale[t / W] * (1.f / 32767.f) * static_cast<float>(v);
    }
}

// rowmax[a] = max_b |g[a, b]|, colmax[b] = max_a |g[a, b]| (colmax zeroed by the host; non-negative
// floats order like their bit patterns, so the column maximum is an integer atomic max)
constexpr int AM_ROWS = 16;
__global__ __launch_bounds__(256) void k_absmax_rows_cols(const float* __restrict__ g, int64_t S, int64_t N, int64_t ld,
                                                          float* __restrict__ rowmax, uint32_t* __restrict__ colmax) {
    __shared__ float part[4][AM_ROWS];
    const int t = threadIdx.x;
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * AM_ROWS;
    float rm[AM_ROWS];
#pragma unroll
    for (int r = 0; r < AM_ROWS; ++r) rm[r] = 0.f;
    for (int64_t c0 = 0; c0 < N; c0 += 256) {
        const int64_t col = c0 + t;
        float cm = 0.f;
#pragma unroll
        for (int r = 0; r < AM_ROWS; ++r) {
            const float v = (col < N && r0 + r < S) ? fabsf(g[(r0 + r) * ld + col]) : 0.f;
            cm = fmaxf(cm, v);
            rm[r] = fmaxf(rm[r], v);
        }
        if (col < N && cm > 0.f) atomicMax(colmax + col, __float_as_uint(cm));
    }
#pragma unroll
    for (int r = 0; r < AM_ROWS; ++r) {
        const float v = wave_allreduce_max(rm[r]);
        if ((t & 63) == 0) part[t >> 6][r] = v;
    }
    __syncthreads();
    if (t < AM_ROWS && r0 + t < S) rowmax[r0 + t] = fmaxf(fmaxf(part[0][t], part[1][t]), fmaxf(part[2][t], part[3][t]));
}

bool l1_pk_eligible(const bess_model_desc* d) {
    return d->dtype == BESS_F16 && is_distance(d->scorer) && d->scorer <= BESS_ROTATE && d->norm_p == 1 &&
           d->width % FKH == 0;
}

int l1_pk_fwd(const bess_model_desc* d, const float* query, int64_t n_query, const void* neg_base,
              const int32_t* neg_idx, int64_t n_neg, float* out, int64_t ld_out, const bess_kill_desc* k,
              hipStream_t st) {
    KillArgs ka{};
    if (k) ka = KillArgs{k->diag_step, k->ht, k->ppp, k->mask, k->mask_rows, k->mask ? k->mask_cols : 0, n_neg};
    const dim3 grid(static_cast<unsigned>(ceil_div(n_neg, PT)), static_cast<unsigned>(ceil_div(n_query, PT)));
    k_l1_fwd_pk<<<grid, 256, 0, st>>>(query, n_query, static_cast<const half_t*>(neg_base), neg_idx, n_neg, d->width,
                                      out, ld_out, ka);
    return check_launch("neg_score_shared_fwd (packed f16 L1)");
}

int64_t l1_pk_bwd_workspace(int64_t n_query, int64_t n_neg) {
    return static_cast<int64_t>(sizeof(float)) * (n_query + n_neg);
}

template <typename TX, typename TY>
static int l1_bwd_one(PkRows<TX> X, PkRows<TY> Y, int W, const float* coef, int64_t sx, int64_t sy,
                      const float* xscale, float* dX, hipStream_t st) {
    const int64_t tiles = ceil_div(W, PT) * ceil_div(X.n, PT);
    // split the reduction over Y until ~4 workgroups per CU are in flight; the slices meet in an int32
    // image (integer atomics: the sum does not depend on their order)
    int64_t split = 1;
    while (tiles * split < 1024 && ceil_div(Y.n, split * 2) >= 4 * BKY) split *= 2;
    int64_t chunk = ceil_div(ceil_div(Y.n, split), BKY) * BKY;
    split = ceil_div(Y.n, chunk);
    const dim3 grid(static_cast<unsigned>(ceil_div(W, PT)), static_cast<unsigned>(ceil_div(X.n, PT)),
                    static_cast<unsigned>(split));
    if (split > 1) {
        hipError_t e = hipMemsetAsync(dX, 0, sizeof(float) * X.n * W, st);
        if (e != hipSuccess) return fail(static_cast<int>(e), "memset: %s", hipGetErrorString(e));
        k_l1_bwd_pk<TX, TY><<<grid, 256, 0, st>>>(X, Y, W, -1.f, coef, sx, sy, xscale, dX, reinterpret_cast<int*>(dX),
                                                  chunk);
        int64_t blocks = ceil_div(X.n * W, 256);
        k_l1_bwd_finish<<<static_cast<unsigned>(blocks > 4096 ? 4096 : blocks), 256, 0, st>>>(dX, xscale, X.n, W,
                                                                                               -1.f);
    } else {
        k_l1_bwd_pk<TX, TY><<<grid, 256, 0, st>>>(X, Y, W, -1.f, coef, sx, sy, xscale, dX, nullptr, chunk);
    }
    return BESS_OK;
}

int l1_pk_bwd(const bess_model_desc* d, const float* query, int64_t n_query, const void* neg_base,
              const int32_t* neg_idx, int64_t n_neg, const float* d_out, int64_t ld_dout, float* d_query,
              float* d_neg, void* workspace, hipStream_t st) {
    float* rowmax = static_cast<float*>(workspace);
    float* colmax = rowmax + n_query;
    hipError_t e = hipMemsetAsync(colmax, 0, sizeof(float) * n_neg, st);
    if (e != hipSuccess) return fail(static_cast<int>(e), "memset: %s", hipGetErrorString(e));
    k_absmax_rows_cols<<<static_cast<unsigned>(ceil_div(n_query, AM_ROWS)), 256, 0, st>>>(
        d_out, n_query, n_neg, ld_dout, rowmax, reinterpret_cast<uint32_t*>(colmax));
    PkRows<float> Q{query, nullptr, n_query};
    PkRows<half_t> E{static_cast<const half_t*>(neg_base), neg_idx, n_neg};
    if (int er = l1_bwd_one<float, half_t>(Q, E, d->width, d_out, ld_dout, 1, rowmax, d_query, st)) return er;
    if (int er = l1_bwd_one<half_t, float>(E, Q, d->width, d_out, 1, ld_dout, colmax, d_neg, st)) return er;
    return check_launch("neg_score_shared_bwd (packed f16 L1)");
}

}  // namespace bess

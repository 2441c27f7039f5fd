// K4 on gfx950: negatives shared by the whole micro-batch.
//
//   out[q, j] = sign * reduce_w f(query[q, w], E[neg_idx[j], w])      q < n_query, j < n_neg
//
// (reference: `pea.distance_matrix(q, N.view(-1, W), p)` for TransE / RotatE,
//  scoring.py:194-197, and `torch.matmul(q, N.reshape(-1, W).T)` for DistMult /
//  ComplEx, scoring.py:251-252.)
//
// Every gathered row is reused by all n_query queries, so this regime is
// compute bound (2-3 flops per (q, j, w)), not HBM bound.  Structure: LDS-tiled
// 64 x 64 output tile per 256-thread workgroup, 4 x 4 register micro-tile per
// lane, 16-deep K stages; negative rows are gathered by index straight into the
// LDS tile (never materialised in HBM).  The p-norm forms have no matrix-core
// formulation (|a - b| is not bilinear) and run on the VALU; the dot form uses
// the same tiling here and is the candidate for the f32 MFMA path.
//
// Backward reuses one tiled kernel:
//   dX[a, w] = sum_b coef[a, b] * f'(X[a, w], Y[b, w])
// called as (X, Y) = (query, negatives) for d_query and (negatives, query) with
// the coefficient matrix read transposed for d_neg - the sum over the
// micro-batch happens on chip, so d_neg needs no atomics.
#include "common.h"

namespace bess {

constexpr int TM = 64;   // tile rows (queries / "a")
constexpr int TN = 64;   // tile cols (negatives / "w")
constexpr int KT = 16;   // stage depth
constexpr int LDP = 68;  // padded LDS leading dimension (floats): 272 B rows, 16 B aligned

// One operand of the tile kernels: rows of `width` scalars, either f32 with
// identity indexing (the query matrix) or table dtype T gathered through idx.
template <typename T>
struct RowSrc {
    const T* base;
    const int32_t* idx;
    int64_t n;
    __device__ __forceinline__ const T* row(int64_t i, int width) const {
        const int64_t r = idx ? static_cast<int64_t>(idx[i]) : i;
        return base + r * width;
    }
};

// stage loader: tile[k][m] = src.row(m0 + m)[k0 + k]   (zero outside)
template <typename T>
__device__ __forceinline__ void load_stage(const RowSrc<T>& src, int64_t m0, int k0, int W,
                                           float (*tile)[LDP]) {
    const int t = threadIdx.x;
    const int m = t >> 2;         // 0..63
    const int kc = (t & 3) * 4;   // 0,4,8,12
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (m0 + m < src.n) {
        const T* rp = src.row(m0 + m, W) + k0 + kc;
        if ((W & 3) == 0 && k0 + kc + 3 < W) {
            VecLoad<T, 4>::load(rp, v);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (k0 + kc + i < W) v[i] = to_f32(rp[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[kc + i][m] = v[i];
}

// 16 stage values of one thread: 4 scalars of row m at columns kc..kc+3
template <typename T>
__device__ __forceinline__ void stage_store(float (*tile)[LDP], int kc, int m, const float (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[kc + i][m] = v[i];
}

// ALIGNED: W % 16 == 0 (every stage is full and rows are 16-B aligned).  The row a
// thread stages never changes (m = t / 4), so its pointer - and the index load
// behind it - is hoisted out of the K loop; rows past the end are clamped to the
// last valid row (their results are dropped at the store) so the loop has no
// bounds branches; the next stage is fetched into registers before the current
// one is computed.  (profiles/ubench/l1_tile.hip: 35 -> 48 T lane-ops/s.)
// MI: query rows per thread (4: 64 x 64 tile; 2: 32 x 64 tile for launches whose 64-row grid leaves CUs idle - the
// stage still brings 64 query rows, the upper half unused).
template <typename T, int RED, bool ALIGNED, int MI>
__global__ __launch_bounds__(256) void k_neg_shared_fwd(RowSrc<float> Q, RowSrc<T> E, int W,
                                                        float sign, float* __restrict__ out,
                                                        int64_t ld_out, float p) {
    __shared__ __attribute__((aligned(16))) float Qs[KT][LDP];
    __shared__ __attribute__((aligned(16))) float Es[KT][LDP];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t q0 = static_cast<int64_t>(blockIdx.y) * (16 * MI);
    const int64_t j0 = static_cast<int64_t>(blockIdx.x) * TN;
    float acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    auto compute = [&]() {
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            float a[MI];
            if constexpr (MI == 4) {
                const float4 a4 = *reinterpret_cast<const float4*>(&Qs[k][ty * 4]);
                a[0] = a4.x, a[1] = a4.y, a[2] = a4.z, a[3] = a4.w;
            } else {
                const float2 a2 = *reinterpret_cast<const float2*>(&Qs[k][ty * 2]);
                a[0] = a2.x, a[1] = a2.y;
            }
            const float4 b4 = *reinterpret_cast<const float4*>(&Es[k][tx * 4]);
            const float b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (RED == RED_DOT) {
                        acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
                    } else if (RED == RED_L1) {
                        acc[i][j] += fabsf(a[i] - b[j]);
                    } else {
                        acc[i][j] += lp_term(a[i] - b[j], p);
                    }
                }
        }
    };

    if (ALIGNED) {
        const int m = threadIdx.x >> 2, kc = (threadIdx.x & 3) * 4;
        const float* qp = Q.row(min(q0 + m, Q.n - 1), W) + kc;
        const T* ep = E.row(min(j0 + m, E.n - 1), W) + kc;
        float qv[4], ev[4];
        VecLoad<float, 4>::load(qp, qv);
        VecLoad<T, 4>::load(ep, ev);
        for (int k0 = 0; k0 < W; k0 += KT) {
            stage_store<float>(Qs, kc, m, qv);
            stage_store<float>(Es, kc, m, ev);
            __syncthreads();
            if (k0 + KT < W) {
                VecLoad<float, 4>::load(qp + k0 + KT, qv);
                VecLoad<T, 4>::load(ep + k0 + KT, ev);
            }
            compute();
            __syncthreads();
        }
    } else {
        for (int k0 = 0; k0 < W; k0 += KT) {
            load_stage<float>(Q, q0, k0, W, Qs);
            load_stage<T>(E, j0, k0, W, Es);
            __syncthreads();
            compute();
            __syncthreads();
        }
    }
    const bool row16 = (ld_out & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int64_t q = q0 + ty * MI + i;
        if (q >= Q.n) continue;
        float* o = out + q * ld_out;
        float v4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[i][j];
            if (RED == RED_L2) v = lp_root(v, p);
            v4[j] = sign * v;
        }
        const int64_t jj0 = j0 + tx * 4;
        if (row16 && jj0 + 3 < E.n) {  // 16-byte aligned rows: one store for the thread's four scores
            *reinterpret_cast<float4*>(o + jj0) = make_float4(v4[0], v4[1], v4[2], v4[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (jj0 + j < E.n) o[jj0 + j] = v4[j];
        }
    }
}

// dX[a, w] = sum_b coef(a, b) * f'(X[a, w], Y[b, w]);  coef from d_out (and out for p=2)
//   DOT: coef = g            f' = y
//   L1 : coef = sign*g       f' = sgn(x - y)
//   L2 : coef = g / out      f' = x - y        (out = -dist; 0 where dist == 0)
// g = d_out[a*sa + b*sb], out likewise with (oa, ob) strides.
// ROUND16: f32 operands (the query matrix) are rounded to fp16 (nearest even) as they are loaded - the
// backward of the packed-fp16 forward (l1_f16.hip), whose scores are a function of the rounded query;
// differences of fp16 values are exact in fp32, so sgn(x - y) (0 at a tie) is exact.
// MI: rows of X per thread (4: 64-row tile; 2: 32-row tile for problems whose 64-row grid cannot fill the chip).
// One of the two products of a backward call: its operands, the strides of d_out / out seen from it, its
// output and its grid (gx column tiles x gy row tiles x gz slices of the reduction, b_chunk rows each).
template <typename TX, typename TY>
struct BwdSide {
    RowSrc<TX> X;
    RowSrc<TY> Y;
    int64_t sa, sb, oa, ob;
    float* dX;
    int64_t b_chunk;
    int gx, gy, gz;
};

template <typename TX, typename TY, int RED, bool VEC4, bool ROUND16, int MI>
__device__ __forceinline__ void neg_shared_bwd_tile(const BwdSide<TX, TY>& S, int W, float sign,
                                                    const float* __restrict__ d_out,
                                                    const float* __restrict__ out, int block,
                                                    float (*Cs)[LDP], float (*Ys)[LDP], float p) {
    const RowSrc<TX> X = S.X;
    const RowSrc<TY> Y = S.Y;
    const int64_t sa = S.sa, sb = S.sb, oa = S.oa, ob = S.ob, b_chunk = S.b_chunk;
    float* __restrict__ dX = S.dX;
    const int bx = block % S.gx, by = (block / S.gx) % S.gy, bz = block / (S.gx * S.gy);
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    constexpr int TMB = 16 * MI;  // rows of X in this workgroup's tile
    const int64_t a0 = static_cast<int64_t>(by) * TMB;
    const int w0 = bx * TN;
    float acc[MI][4], xv[MI][4];
    // this thread's 4 x 4 values of X: rows and columns past the end are clamped (their results are dropped
    // at the store), so the 4 row loads are issued together - a per-element `in range ? load : 0` made the
    // compiler wait for each of the 16 loads (and the index load in front of it) in turn: 16 round trips
    // to L2 / HBM at the start of every workgroup, ~8 us of a ~50 us workgroup
    {
        const TX* xr[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) xr[i] = X.row(min(a0 + ty * MI + i, X.n - 1), W);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (VEC4) {
                VecLoad<TX, 4>::load(xr[i] + min(w0 + tx * 4, W - 4), xv[i]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[i][j] = to_f32(xr[i][min(w0 + tx * 4 + j, W - 1)]);
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = 0.f;
                if (ROUND16) xv[i][j] = static_cast<float>(static_cast<_Float16>(xv[i][j]));
                if (RED == RED_L1) xv[i][j] *= SGN_PRESCALE;
            }
    }
    // blockIdx.z owns the slice [b_lo, b_hi) of the reduction; partial sums of
    // several slices are combined with fp32 atomics (dX zeroed by the host)
    const int64_t b_lo = static_cast<int64_t>(bz) * b_chunk;
    const int64_t b_hi = min(b_lo + b_chunk, Y.n);
    if (b_lo >= b_hi) return;

    // what this thread stages: MI coefficients Cs[bb][al] and 4 scalars Ys[yb][ywc..+3].
    // Lanes run along the unit-stride dimension of d_out.  Out-of-range rows are clamped
    // (their coefficient is forced to 0), so the stage has no divergent branches.
    const int t = threadIdx.x;
    const int yb = t >> 4, ywc = (t & 15) * 4;
    const int ycol = VEC4 ? min(w0 + ywc, W - 4) : w0 + ywc;  // clamped columns are dropped at the store
    float cv[MI], yv[4];
    auto fetch = [&](int64_t b0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            int bb, al;
            if (sb == 1) {  // d_out[a, b] contiguous in b
                bb = t & 15;
                al = (t >> 4) + 16 * i;
            } else {        // read transposed: contiguous in a
                al = t % TMB;
                bb = t / TMB + (256 / TMB) * i;
            }
            const int64_t a = a0 + al;
            const bool ok = a < X.n && b0 + bb < b_hi;
            const int64_t ac = min(a, X.n - 1), bc = min(b0 + bb, b_hi - 1);
            const float g = d_out[ac * sa + bc * sb];
            float c;
            if (RED == RED_L2) {
                const float o = out[ac * oa + bc * ob];  // o = -norm: d(-norm) / d x_w = -sgn |x_w - y_w|^(p-1) norm^(1-p)
                c = -g * lp_inv(-o, p);
            } else {
                c = sign * g;
            }
            cv[i] = ok ? c : 0.f;
        }
        const TY* rp = Y.row(min(b0 + yb, b_hi - 1), W);
        if (VEC4) {
            VecLoad<TY, 4>::load(rp + ycol, yv);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) yv[i] = (ycol + i < W) ? to_f32(rp[ycol + i]) : 0.f;
        }
        if (ROUND16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) yv[i] = static_cast<float>(static_cast<_Float16>(yv[i]));
        }
        if (RED == RED_L1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) yv[i] *= SGN_PRESCALE;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (sb == 1) Cs[t & 15][(t >> 4) + 16 * i] = cv[i];
            else Cs[t / TMB + (256 / TMB) * i][t % TMB] = cv[i];
        }
        *reinterpret_cast<float4*>(&Ys[yb][ywc]) = make_float4(yv[0], yv[1], yv[2], yv[3]);
    };

    fetch(b_lo);
    for (int64_t b0 = b_lo; b0 < b_hi; b0 += KT) {
        stash();
        __syncthreads();
        if (b0 + KT < b_hi) fetch(b0 + KT);  // next stage in flight under the compute
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            float c[MI];
            if (MI == 4) {
                const float4 c4 = *reinterpret_cast<const float4*>(&Cs[k][ty * 4]);
                c[0] = c4.x, c[1] = c4.y, c[MI - 2] = c4.z, c[MI - 1] = c4.w;
            } else {
                const float2 c2 = *reinterpret_cast<const float2*>(&Cs[k][ty * 2]);
                c[0] = c2.x, c[1] = c2.y;
            }
            const float4 y4 = *reinterpret_cast<const float4*>(&Ys[k][tx * 4]);
            const float y[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (RED == RED_DOT) acc[i][j] = fmaf(c[i], y[j], acc[i][j]);
                    else if (RED == RED_L1) acc[i][j] = fmaf(c[i], sgn_prescaled(xv[i][j] - y[j]), acc[i][j]);
                    else acc[i][j] = fmaf(c[i], lp_dterm(xv[i][j] - y[j], p), acc[i][j]);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int64_t a = a0 + ty * MI + i;
        if (a >= X.n) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int w = w0 + tx * 4 + j;
            if (w < W) {
                if (S.gz == 1) dX[a * W + w] = acc[i][j];
                else unsafeAtomicAdd(dX + a * W + w, acc[i][j]);
            }
        }
    }
}

// Both products of a backward call in ONE launch: workgroups [0, blocks_a) compute d_query tiles, the rest d_neg
// tiles.  The products are independent, so at notebook sizes (a few hundred workgroups each) they run side by
// side instead of one after the other, and at large sizes the second fills the chip as the first drains.
template <typename TE, int RED, bool VEC4, bool ROUND16, int MIA, int MIB>
__global__ __launch_bounds__(256) void k_neg_shared_bwd(BwdSide<float, TE> A, BwdSide<TE, float> B, int W,
                                                        float sign, const float* __restrict__ d_out,
                                                        const float* __restrict__ out, int blocks_a, float p) {
    __shared__ __attribute__((aligned(16))) float Cs[KT][LDP];  // [b][a]
    __shared__ __attribute__((aligned(16))) float Ys[KT][LDP];  // [b][w]
    const int block = blockIdx.x;
    if (block < blocks_a) neg_shared_bwd_tile<float, TE, RED, VEC4, ROUND16, MIA>(A, W, sign, d_out, out, block, Cs, Ys, p);
    else neg_shared_bwd_tile<TE, float, RED, VEC4, ROUND16, MIB>(B, W, sign, d_out, out, block - blocks_a, Cs, Ys, p);
}

// ---- p = 1, both products from ONE evaluation of sgn(q - e) -----------------------------------------------
// The two products of the L1 backward,
//     d_query[i, w] =  sum_j c[i, j] sgn(q[i, w] - e[j, w]),      d_neg[j, w] = -sum_i c[i, j] sgn(q[i, w] - e[j, w]),
// share their expensive part: the subtraction and the sign (v_sub + v_med3, the half-rate one) of every
// (i, j, w).  The tile kernel above evaluates it once per product - 3 VALU instructions per element and product,
// 9.5 issue cycles.  Here a thread owns one column w and 32 candidates j (their e[j, w] and their d_neg
// accumulators stay in registers for the workgroup's whole range of queries) and walks over the queries: per
// (i, j, w) one v_sub, one v_med3 and two v_fma (one into the query's partial, one into the candidate's
// accumulator) - 4 instructions, 12.1 issue cycles for BOTH products (36 % fewer).  Workgroup: 256 candidates x
// 32 columns x a slice of the queries; the coefficients of 8 queries x 256 candidates are staged through LDS
// (read as broadcast b128: the 32 lanes of a column group read the same 16 bytes); the queries' partials of the
// workgroup's 8 candidate groups meet in LDS and leave as one fp32 atomic per (i, w) and step; the candidates'
// accumulators leave as atomics at the end (one per (j, w) and query slice).  Both outputs are zero on entry.
constexpr int FB_TW = 64, FB_IS = 8;  // (FB_NW waves = groups of 32 candidates per workgroup: template parameter)

template <typename TE, bool ROUND16, int FB_NW>
__global__ __launch_bounds__(64 * FB_NW, 4) void k_l1_bwd_both(RowSrc<float> Q, RowSrc<TE> E, int W, float sign,
                                                        const float* __restrict__ d_out, int64_t ld,
                                                        float* __restrict__ dq, float* __restrict__ de, int i_chunk) {
    // A wave = 64 columns x ONE group of 32 candidates: the coefficient of (query, candidate) is the same for all
    // its lanes - a wave-uniform address, i.e. scalar loads (s_load_dwordx8 into SGPRs that the v_fma reads
    // directly): no LDS staging of the coefficients, no barrier for them.  (Round 4, on the counters of
    // profiles/r04/pmc_l1_kernels.txt - 0.26 of a wave's life in s_waitcnt: a variant that fetches coefficients and
    // query values a step ahead through vector loads + LDS, as k_l1_bwd_parts does, 108 VGPRs, still four waves per
    // SIMD: 407.2 vs 409.2 us at 4096 x 4352, 1388 vs 1427 at 8192 x 8448, 26.4 vs 24.5 at 512 x 544 - with four waves
    // per SIMD the waits were already covered; not kept.)  Eight waves = 256 candidates; their
    // partial sums of d_query meet in LDS every FB_IS queries.
    __shared__ float Rs[FB_NW][FB_IS][FB_TW];
    const int t = threadIdx.x;
    const int w = t & 63;
    const int jg = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t j0 = static_cast<int64_t>(blockIdx.x) * (32 * FB_NW);
    const int w0 = blockIdx.y * FB_TW;
    const int64_t i_lo = static_cast<int64_t>(blockIdx.z) * i_chunk;
    const int64_t i_hi = min(i_lo + i_chunk, Q.n);
    if (i_lo >= i_hi) return;
    const int wc = min(w0 + w, W - 1);  // (columns past the end: clamped, dropped at the stores)
    const bool w_ok = w0 + w < W;
    const int64_t jbase = j0 + jg * 32;  // wave-uniform
    // (n_neg % 32 == 0: a wave's 32 candidates are all there or none is - waves of the last tile past the end
    // load and compute nothing, they only keep the workgroup's barriers company)
    const bool active = jbase < E.n;

    // this wave's 32 candidates at the lane's column, pre-scaled for the one-instruction sign
    // (row ids first, all 32 loads together, then the 32 row loads together: written with E.row() the compiler
    // put every load behind its own branch on `idx` - 32 round trips in a row at the start of every workgroup)
    float e[32], acc[32];
    {
        int32_t rows[32];  // (row ids are int32 everywhere in the library)
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) rows[jj] = static_cast<int32_t>(min(jbase + jj, E.n - 1));
        if (E.idx) {
#pragma unroll
            for (int jj = 0; jj < 32; ++jj) rows[jj] = E.idx[rows[jj]];
        }
        const TE* col = E.base + wc;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
            e[jj] = to_f32(col[static_cast<int64_t>(rows[jj]) * W]) * SGN_PRESCALE;
            acc[jj] = 0.f;
        }
        // the scaled f32 value is what the loop subtracts (left to itself the compiler keeps the f16 value and folds
        // conversion and scale into a v_fma_mix_f32 per element: a VOP3P instruction, 0.6 of the plain issue rate)
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) asm volatile("" : "+v"(e[jj]));
    }
    // (launched only with n_neg % 32 == 0 and whole steps of queries: no masks in the inner loop; `sign` is
    // applied to the sums at the stores)
    const float* qcol = Q.base + wc;  // (the query matrix is dense f32: Q.idx == NULL by construction)
    float qnext = qcol[i_lo * W];    // the query value is fetched one query ahead of its use
    // The coefficients too are fetched one query ahead: two sets of 32 SGPRs take turns.  (Fetched where they are
    // used, a wave waits ~0.4 us for its scalar loads on every query: hidden when four waves share the SIMD, but
    // the whole kernel time of a notebook-size micro-batch whose grid is one wave per SIMD.)  Scalar loads return
    // out of order, so a wait on them is a wait for all: the next set is requested AFTER the first use of the
    // current one (the only place the compiler has to wait), not before.
    const float* __restrict__ cbase = d_out + (active ? jbase : 0);  // wave-uniform
    float cA[32], cB[32];
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) cA[jj] = cbase[i_lo * ld + jj];
    auto one_query = [&](const float (&cur)[32], float (&nxt)[32], int64_t i, int ii) {
        float qv = qnext;
        qnext = qcol[min(i + 1, i_hi - 1) * W];
        if (ROUND16) qv = static_cast<float>(static_cast<_Float16>(qv));
        qv *= SGN_PRESCALE;
        asm volatile("" : "+v"(qv));  // (else the rounding and the scale are folded into 32 v_fma_mix_f32: VOP3P rate)
        float pq4[4] = {0.f, 0.f, 0.f, 0.f};  // (four chains: one accumulator would serialise 32 dependent v_fma)
        {
            const float sg = sgn_prescaled(qv - e[0]);
            pq4[0] = fmaf(cur[0], sg, pq4[0]);
            acc[0] = fmaf(cur[0], sg, acc[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float* __restrict__ crow = cbase + min(i + 1, i_hi - 1) * ld;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) nxt[jj] = crow[jj];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 1; jj < 32; ++jj) {
            const float c = cur[jj];
            const float sg = sgn_prescaled(qv - e[jj]);
            pq4[jj & 3] = fmaf(c, sg, pq4[jj & 3]);
            acc[jj] = fmaf(c, sg, acc[jj]);  // (d_neg is minus this sum: the sign is flipped once, at the store)
        }
        Rs[jg][ii][w] = active ? (pq4[0] + pq4[1]) + (pq4[2] + pq4[3]) : 0.f;
    };
    for (int64_t i0 = i_lo; i0 < i_hi; i0 += FB_IS) {
        // (inactive waves - candidates past the end in the last tile - run the same loop on the first candidates'
        // coefficients and drop everything at the stores: no branch around 128 instructions per query)
#pragma unroll 1
        for (int ii = 0; ii < FB_IS; ii += 2) {
            one_query(cA, cB, i0 + ii, ii);
            one_query(cB, cA, i0 + ii + 1, ii + 1);
        }
        __syncthreads();
        // FB_NW waves x 64 columns: sum the candidate groups, one atomic per (i, w)
#pragma unroll
        for (int ii = t >> 6; ii < FB_IS; ii += FB_NW) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < FB_NW; ++g) sum += Rs[g][ii][w];
            const int64_t i = i0 + ii;
            if (i < i_hi && w_ok && sum != 0.f) unsafeAtomicAdd(dq + i * W + w0 + w, sign * sum);
        }
        __syncthreads();  // Rs is rewritten by the next step
    }
    if (w_ok) {
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
            const int64_t j = jbase + jj;
            if (j < E.n && acc[jj] != 0.f) unsafeAtomicAdd(de + j * W + w0 + w, -sign * acc[jj]);
        }
    }
}

// ---- the same two products as PARTIAL SUMS: no atomics, no barrier ------------------------------------------------
// A wave = 64 columns x one group of 32 candidates x one slice of the queries, exactly as above - but it shares
// nothing with other waves: the query's partial sum over the wave's 32 candidates is STORED, as row i of slab
// `candidate group` of dq_parts [groups, S, W], the candidates' sums over the slice as rows of slab `slice` of
// de_parts [slices, N, W].  Their consumer (k_query_triple_bwd_parts) adds the slabs up where it reads them.
// At the notebook micro-batch (S = 512, N = 544) the 3.7 M fp32 atomics of the sums - 15 MB at the ~1.3 TB/s
// memory-side atomics run at - the barriers of the LDS meeting every 8 queries and the idle waves of the last
// candidate tile (3 of 20) were most of the kernel's 24 us.  Workgroups are four independent waves.
template <typename TE, bool ROUND16>
__global__ __launch_bounds__(256, 2) void k_l1_bwd_parts(RowSrc<float> Q, RowSrc<TE> E, int W, float sign,
                                                     const float* __restrict__ d_out, int64_t ld,
                                                     float* __restrict__ dq_parts, float* __restrict__ de_parts,
                                                     int i_chunk, int n_group, int n_task) {
    const int w = threadIdx.x & 63;
    // task = (slice, candidate group): consecutive waves take consecutive groups of the same slice
    const int task = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + (threadIdx.x >> 6));
    if (task >= n_task) return;
    const int jg = task % n_group, sl = task / n_group;
    const int w0 = blockIdx.y * FB_TW;
    const int i_lo = sl * i_chunk;
    const int n_i = min(i_chunk, static_cast<int>(Q.n) - i_lo);  // even: S % 8 == 0, slices of multiples of 8
    const int wc = min(w0 + w, W - 1);  // (columns past the end: clamped, dropped at the stores)
    const bool w_ok = w0 + w < W;
    const int64_t jbase = static_cast<int64_t>(jg) * 32;  // (n_neg % 32 == 0: all 32 candidates exist)
    // Coefficients and query values arrive a BLOCK of FB_IS queries ahead, through vector loads: the block's 8 x 32
    // coefficients as four coalesced loads per lane, staged in a wave-private piece of LDS and read back as broadcast
    // b128 (a query's 32 coefficients: 8 reads, fetched one query ahead of their use), the lane's 8 query values
    // straight into registers.  (Round 3 / early round 4 fetched them with scalar loads one query ahead: every
    // s_load of the 128 B of a query is a miss of the scalar cache - ~0.7 us a query with two waves per SIMD where the
    // 128 VALU instructions of a query need 0.2: 20 us for the kernel at the notebook micro-batch.)  The waves of a
    // workgroup still share nothing: no barrier.
    __shared__ __attribute__((aligned(16))) float cs[4][2][FB_IS][32];  // [wave][buffer][query][candidate]
    __shared__ float qs[4][2][FB_IS][64];                               // [wave][buffer][query][column]
    float(*mine)[FB_IS][32] = cs[threadIdx.x >> 6];
    float(*myq)[FB_IS][64] = qs[threadIdx.x >> 6];
    const int lq = w >> 5, lj = w & 31;
    const int n_blk = n_i / FB_IS;  // (S % 8 == 0 and slices of multiples of 8 queries: whole blocks)
    float rc[4], rq[FB_IS];
    auto load_block = [&](int blk) {
        const float* cp = d_out + static_cast<int64_t>(i_lo + blk * FB_IS + lq) * ld + jbase + lj;
#pragma unroll
        for (int k = 0; k < 4; ++k) rc[k] = cp[static_cast<int64_t>(2 * k) * ld];
        const float* qp = Q.base + static_cast<int64_t>(i_lo + blk * FB_IS) * W + wc;
#pragma unroll
        for (int k = 0; k < FB_IS; ++k) rq[k] = qp[static_cast<int64_t>(k) * W];
    };
    float* __restrict__ dq_out = dq_parts + (static_cast<int64_t>(jg) * Q.n + i_lo) * W + w0 + w;
    load_block(0);  // (requested before the candidates' values below: one wait for both)
    float e[32], acc[32];
    {
        int32_t rows[32];
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) rows[jj] = static_cast<int32_t>(jbase + jj);
        if (E.idx) {
#pragma unroll
            for (int jj = 0; jj < 32; ++jj) rows[jj] = E.idx[rows[jj]];
        }
        const TE* col = E.base + wc;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
            e[jj] = to_f32(col[static_cast<int64_t>(rows[jj]) * W]) * SGN_PRESCALE;
            acc[jj] = 0.f;
        }
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) asm volatile("" : "+v"(e[jj]));  // (see k_l1_bwd_both)
    }
    float order = 0.f;  // (the previous query's sum: see the asm below)
    auto one_query = [&](const float* crow, float qv) {
        if (ROUND16) qv = static_cast<float>(static_cast<_Float16>(qv));
        qv *= SGN_PRESCALE;
        // the scaled value is what the loop subtracts (k_l1_bwd_both), and it "depends" on the previous query's sum:
        // left alone the compiler evaluates sgn(q - e) of several queries' pairs up front and spills them
        asm volatile("" : "+v"(qv) : "v"(order));
        float pq4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 c4 = *reinterpret_cast<const float4*>(crow + 4 * k);  // broadcast: the wave reads one address
            const float c[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float sg = sgn_prescaled(qv - e[4 * k + u]);
                pq4[u] = fmaf(c[u], sg, pq4[u]);
                acc[4 * k + u] = fmaf(c[u], sg, acc[4 * k + u]);  // (d_neg is minus this sum: flipped at the store)
            }
        }
        order = sign * ((pq4[0] + pq4[1]) + (pq4[2] + pq4[3]));
        if (w_ok) *dq_out = order;
        dq_out += W;
    };
#pragma unroll 1
    for (int blk = 0; blk < n_blk; ++blk) {
        float(*buf)[32] = mine[blk & 1];
        float(*qbuf)[64] = myq[blk & 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) buf[lq + 2 * k][lj] = rc[k];
#pragma unroll
        for (int k = 0; k < FB_IS; ++k) qbuf[k][w] = rq[k];
        load_block(min(blk + 1, n_blk - 1));  // in flight while this block is worked on (after the last: a surplus copy)
#pragma unroll 1
        for (int ii = 0; ii < FB_IS; ++ii) one_query(buf[ii], qbuf[ii][w]);
    }
    if (w_ok) {
        float* __restrict__ de_out = de_parts + (static_cast<int64_t>(sl) * E.n + jbase) * W + w0 + w;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) de_out[static_cast<int64_t>(jj) * W] = -sign * acc[jj];
    }
}

// Worth it when the 256-candidate x 64-column tiles are mostly full, and up to micro-batches of 8192 queries: the
// candidates' sums leave as one atomic per (j, w) and QUERY SLICE, and filling the chip in whole rounds takes
// more slices the larger the problem - at S = 16,384 / 65,536 (63+ slices: 1 GB of atomics per step and more) the
// two-product tile kernel is 8 - 14 % faster again (c4 sweep: 11.6 vs 12.6 ms, 175 vs 201 ms).
static bool use_l1_bwd_both(const bess_model_desc* d, int64_t S, int64_t N) {
    return reduce_of(d) == RED_L1 && S >= 256 && S <= 8192 && N >= 256 && N % 32 == 0 && S % FB_IS == 0 &&
           d->width >= FB_TW;
}

// The partial-sum form pays for having no atomics with the slabs of d_query it stores - one per group of 32
// candidates, S x W each: up to notebook sizes (512 x 544: 8.9 MB, 21.6 -> ~19 us against 24.1 with atomics;
// 256 x 288: 9.8 against 14.8) a gain, from 1024 x 1088 on (36 MB: 50.4 against 48.5, then 2x slower) a loss.
static bool use_l1_bwd_parts(const bess_model_desc* d, int64_t S, int64_t N) {
    return use_l1_bwd_both(d, S, N) && S * (N / 32) * d->width * 4 <= (24ll << 20);
}

// launch plan of k_l1_bwd_both: waves per workgroup, queries per slice, slices, candidate tiles
struct L1BothPlan {
    int nw;
    int64_t i_chunk, slices, jt, wt;
};
static L1BothPlan plan_l1_bwd_both(int64_t S, int64_t N, int W, bool /*unused*/) {
    L1BothPlan p;
    // eight waves (256 candidates) per workgroup; four where the candidates are few (finer tiles: 544 candidates
    // are 4.25 tiles of 128 but 2.1 of 256).  Measured (profiles/sweep_l1_bwd.py, us, 4 / 8 waves): 512 x 544
    // 25.2 / 29.5, 1024 x 1088 49.3 / 52.8, 2048 x 2176 123.7 / 123.4, 4096 x 4352 451 / 409
    p.nw = N < 2048 ? 4 : 8;
    p.jt = ceil_div(N, 32 * p.nw);
    p.wt = ceil_div(W, FB_TW);
    // Query slices.  16 waves per CU are resident (103 VGPRs: 4 per SIMD): the grid should fill them in whole
    // rounds - 1564 eight-wave workgroups are 4 rounds of 512 of which the last is almost empty, 1496 are 3.  Cost of
    // a plan = rounds x (queries per slice + ~24 queries' worth of loading the candidates' registers and storing
    // their sums) + 1 per slice (every slice adds one atomic per element of d_neg: at 512 x 544, 22 slices of 24
    // queries take 23.8 us, 32 of 16 take 27.5).
    const int64_t resident = 256 * 16 / p.nw, tiles = p.jt * p.wt;
    p.i_chunk = S;
    p.slices = 1;
    double best = 1e300;
    const double per_slice = 1.0;
    for (int64_t sl = 1; sl <= ceil_div(S, 16); ++sl) {
        const int64_t ch = ceil_div(ceil_div(S, sl), FB_IS) * FB_IS;
        const int64_t actual = ceil_div(S, ch);
        const double cost = static_cast<double>(ceil_div(tiles * actual, resident)) * (ch + 24) + actual * per_slice;
        if (cost < best) best = cost, p.i_chunk = ch, p.slices = actual;
    }
    return p;
}

// plan of k_l1_bwd_parts: candidate groups (= slabs of dq_parts), query slices (= slabs of de_parts).  The waves are
// independent: (groups x column tiles x slices) of them should fill the chip's 1024 SIMDs a whole number of times,
// 2 - 4 waves each, without making a slice shorter than 8 queries (every wave pays ~24 queries' worth of loading its
// candidates and storing their sums).
struct L1PartsPlan {
    int64_t groups, slices, i_chunk;
};
static L1PartsPlan plan_l1_bwd_parts(int64_t S, int64_t N, int W) {
    L1PartsPlan p;
    p.groups = N / 32;
    const int64_t per_slice = p.groups * ceil_div(W, FB_TW);  // waves per slice
    p.i_chunk = S;
    p.slices = 1;
    double best = 1e300;
    for (int64_t sl = 1; sl <= S / FB_IS; ++sl) {
        const int64_t ch = ceil_div(ceil_div(S, sl), FB_IS) * FB_IS;
        const int64_t actual = ceil_div(S, ch);
        const int64_t waves = per_slice * actual;
        // rounds over the 4096 wave slots of the chip at 4 per SIMD; a SIMD's waves share its issue slots
        const double per_simd = static_cast<double>(ceil_div(waves, 1024));
        // (measured, round 4 kernel - coefficients through LDS -, us for 4 / 8 / 11 / 16 / 22 / 32 / 64 slices:
        //  512 x 544: 53.5 / 30.0 / 24.1 / 18.7 / 16.3 / 16.2 / 16.8; 256 x 288: 28.2 / 16.4 / 13.3 / 10.5 / - / 8.3 / -;
        //  1024 x 544: 102.6 / 54.8 / 42.6 / 31.7 / 25.7 / 24.9 / 23.0 - a query costs a wave ~0.4 us alone on its
        //  SIMD; beyond ~2 waves per SIMD the 27 MB of slabs the launch stores are what is left)
        const double cost = per_simd * (ch + 12.0) / std::min(per_simd, 2.5) + 0.1 * actual;
        if (cost < best) best = cost, p.i_chunk = ch, p.slices = actual;
    }
    return p;
}

template <typename TE>
static int run_l1_bwd_parts(const bess_model_desc* d, RowSrc<float> Q, RowSrc<TE> E, const float* d_out,
                            int64_t ld_dout, float* dq_parts, float* de_parts, hipStream_t st, bool round16) {
    const int W = d->width;
    const L1PartsPlan p = plan_l1_bwd_parts(Q.n, E.n, W);
    const int64_t tasks = p.groups * p.slices;
    BESS_REQUIRE(tasks < (1ll << 31) && ceil_div(W, FB_TW) < 65536, "neg_score_shared_bwd_parts: problem too large");
    const dim3 grid(static_cast<unsigned>(ceil_div(tasks, 4)), static_cast<unsigned>(ceil_div(W, FB_TW)));
    const float sign = is_distance(d->scorer) ? -1.f : 1.f;
    if (round16)
        k_l1_bwd_parts<TE, true><<<grid, 256, 0, st>>>(Q, E, W, sign, d_out, ld_dout, dq_parts, de_parts,
                                                       static_cast<int>(p.i_chunk), static_cast<int>(p.groups),
                                                       static_cast<int>(tasks));
    else
        k_l1_bwd_parts<TE, false><<<grid, 256, 0, st>>>(Q, E, W, sign, d_out, ld_dout, dq_parts, de_parts,
                                                        static_cast<int>(p.i_chunk), static_cast<int>(p.groups),
                                                        static_cast<int>(tasks));
    return check_launch("neg_score_shared_bwd_parts");
}

template <typename TE>
static int run_l1_bwd_both(const bess_model_desc* d, RowSrc<float> Q, RowSrc<TE> E, const float* d_out,
                           int64_t ld_dout, float* d_query, float* d_neg, hipStream_t st, bool round16) {
    const int W = d->width;
    const L1BothPlan p = plan_l1_bwd_both(Q.n, E.n, W, false);
    const int nw = p.nw;
    const int64_t jt = p.jt, wt = p.wt, slices = p.slices;
    BESS_REQUIRE(jt < (1ll << 31) && wt < 65536 && slices < 65536, "neg_score_shared_bwd: problem too large for one launch");
    const dim3 grid(static_cast<unsigned>(jt), static_cast<unsigned>(wt), static_cast<unsigned>(slices));
    const float sign = is_distance(d->scorer) ? -1.f : 1.f;
    const int ic = static_cast<int>(p.i_chunk);
#define BESS_FB(R16, NW) \
    k_l1_bwd_both<TE, R16, NW><<<grid, 64 * NW, 0, st>>>(Q, E, W, sign, d_out, ld_dout, d_query, d_neg, ic)
    if (nw == 8) {
        if (round16) BESS_FB(true, 8);
        else BESS_FB(false, 8);
    } else {
        if (round16) BESS_FB(true, 4);
        else BESS_FB(false, 4);
    }
#undef BESS_FB
    return check_launch("neg_score_shared_bwd (p = 1, both products)");
}

template <typename T>
static int run_fwd(const bess_model_desc* d, RowSrc<float> Q, RowSrc<T> E, float* out, int64_t ld,
                   hipStream_t st) {
    // 32-row tiles when the 64-row grid would leave most CUs without a workgroup
    const bool small = ceil_div(E.n, TN) * ceil_div(Q.n, TM) < 192;
    const dim3 grid(static_cast<unsigned>(ceil_div(E.n, TN)), static_cast<unsigned>(ceil_div(Q.n, small ? TM / 2 : TM)));
    const float sign = is_distance(d->scorer) ? -1.f : 1.f;
    const bool aligned = d->width % KT == 0;
#define BESS_FWD_ARGS <<<grid, 256, 0, st>>>(Q, E, d->width, sign, out, ld, static_cast<float>(d->norm_p))
#define BESS_FWD(RED)                                                         \
    do {                                                                      \
        if (aligned && small) k_neg_shared_fwd<T, RED, true, 2> BESS_FWD_ARGS;  \
        else if (aligned) k_neg_shared_fwd<T, RED, true, 4> BESS_FWD_ARGS;      \
        else if (small) k_neg_shared_fwd<T, RED, false, 2> BESS_FWD_ARGS;       \
        else k_neg_shared_fwd<T, RED, false, 4> BESS_FWD_ARGS;                  \
    } while (0)
    switch (reduce_of(d)) {
        case RED_DOT: BESS_FWD(RED_DOT); break;
        case RED_L1: BESS_FWD(RED_L1); break;
        default: BESS_FWD(RED_L2);
    }
#undef BESS_FWD
#undef BESS_FWD_ARGS
    return check_launch("neg_score_shared_fwd");
}

// Launch plan of one product.  Workgroups resident at once: 5 per CU is what the kernel's 94 VGPRs admit
// (rocprofv3 SQ counters at 4 per CU: VALU issue 61 % of the cycles, waves parked at waits / barriers 27 % of
// theirs - a wave alone issues at half rate, so resident waves are what fills the pipe).  The reduction over Y is
// cut into `split` slices of at least 4 stages so that the grid fills those slots in whole rounds:
// cost(split) = rounds of the grid over the slots / split, a little extra per slice for its prologue and the atomics.
template <typename TX, typename TY>
static bool plan_bwd_side(int W, BwdSide<TX, TY>& S) {
    const int64_t slots = 256 * 5;
    auto plan = [&](int64_t tile_rows, int64_t* best_split) {
        const int64_t tiles = ceil_div(W, TN) * ceil_div(S.X.n, tile_rows);
        double best = 1e30;
        for (int64_t s = 1; s <= 32; ++s) {
            if (s > 1 && ceil_div(S.Y.n, s) < 4 * KT) break;
            const double cost = static_cast<double>(ceil_div(tiles * s, slots)) / s * (1.0 + 0.01 * s);
            if (cost < best) best = cost, *best_split = s;
        }
        return tiles * *best_split;
    };
    int64_t split = 1, split32 = 1;
    const int64_t blocks64 = plan(TM, &split);
    // 32-row tiles when the 64-row grid leaves most of the chip idle (under two workgroups per CU)
    const bool small = blocks64 < 2 * 256 && plan(TM / 2, &split32) > blocks64;
    if (small) split = split32;
    S.b_chunk = ceil_div(ceil_div(S.Y.n, split), KT) * KT;
    S.gx = static_cast<int>(ceil_div(W, TN));
    S.gy = static_cast<int>(ceil_div(S.X.n, small ? TM / 2 : TM));
    S.gz = static_cast<int>(ceil_div(S.Y.n, S.b_chunk));
    return small;
}

template <typename TE>
static int run_bwd(const bess_model_desc* d, RowSrc<float> Q, RowSrc<TE> E, const float* d_out, int64_t ld_dout,
                   const float* out, int64_t ld_out, float* d_query, float* d_neg, hipStream_t st, bool round16) {
    const int W = d->width;
    if (use_l1_bwd_both(d, Q.n, E.n) && Q.idx == nullptr) {  // (the kernel reads the dense f32 query matrix)
        if (!(d->reserved[0] & BESS_FLAG_PREZEROED)) {  // its outputs are sums of atomics
            hipError_t e = hipSuccess;
            if (d_neg == d_query + Q.n * W) e = fill_words_async(d_query, 0u, (Q.n + E.n) * W, st);
            else {
                e = fill_words_async(d_query, 0u, Q.n * W, st);
                if (e == hipSuccess) e = fill_words_async(d_neg, 0u, E.n * W, st);
            }
            if (e != hipSuccess) return fail(static_cast<int>(e), "fill: %s", hipGetErrorString(e));
        }
        return run_l1_bwd_both<TE>(d, Q, E, d_out, ld_dout, d_query, d_neg, st, round16);
    }
    BwdSide<float, TE> A{Q, E, ld_dout, 1, ld_out, 1, d_query, 0, 0, 0, 0};
    BwdSide<TE, float> B{E, Q, 1, ld_dout, 1, ld_out, d_neg, 0, 0, 0, 0};
    const bool small_a = plan_bwd_side(W, A), small_b = plan_bwd_side(W, B);
    const int64_t blocks_a = static_cast<int64_t>(A.gx) * A.gy * A.gz, blocks_b = static_cast<int64_t>(B.gx) * B.gy * B.gz;
    BESS_REQUIRE(blocks_a + blocks_b < (1ll << 31), "neg_score_shared_bwd: problem too large for one launch");
    // partial sums of several slices are combined with fp32 atomics: zero their targets - with one memset
    // when both gradients share an allocation (d_neg right behind d_query)
    auto zero = [&](float* p, int64_t n) -> int {
        hipError_t e = fill_words_async(p, 0u, n, st);
        return e == hipSuccess ? BESS_OK : fail(static_cast<int>(e), "memset: %s", hipGetErrorString(e));
    };
    if (d->reserved[0] & BESS_FLAG_PREZEROED) {
        // the caller cleared both targets (one launch for all fills of a step: bess_step_prologue)
    } else if (A.gz > 1 && B.gz > 1 && d_neg == d_query + Q.n * W) {
        if (int e = zero(d_query, (Q.n + E.n) * W)) return e;
    } else {
        if (A.gz > 1)
            if (int e = zero(d_query, Q.n * W)) return e;
        if (B.gz > 1)
            if (int e = zero(d_neg, E.n * W)) return e;
    }
    const unsigned grid = static_cast<unsigned>(blocks_a + blocks_b);
    const int ba = static_cast<int>(blocks_a);
    const float sign = is_distance(d->scorer) ? -1.f : 1.f;
    const bool vec4 = W % 4 == 0;
#define BESS_BWD_ARGS <<<grid, 256, 0, st>>>(A, B, W, sign, d_out, out, ba, static_cast<float>(d->norm_p))
#define BESS_BWD_MI(RED, V, R16)                                                   \
    do {                                                                           \
        if (small_a && small_b) k_neg_shared_bwd<TE, RED, V, R16, 2, 2> BESS_BWD_ARGS; \
        else if (small_a) k_neg_shared_bwd<TE, RED, V, R16, 2, 4> BESS_BWD_ARGS;       \
        else if (small_b) k_neg_shared_bwd<TE, RED, V, R16, 4, 2> BESS_BWD_ARGS;       \
        else k_neg_shared_bwd<TE, RED, V, R16, 4, 4> BESS_BWD_ARGS;                    \
    } while (0)
    const int red = reduce_of(d);  // RED_L1 or RED_L2: the bilinear scorers' products run on the matrix cores
    if (red == RED_L1 && vec4 && round16) BESS_BWD_MI(RED_L1, true, true);
    else if (red == RED_L1 && vec4) BESS_BWD_MI(RED_L1, true, false);
    else if (red == RED_L1) BESS_BWD_MI(RED_L1, false, false);
    else if (vec4) BESS_BWD_MI(RED_L2, true, false);
    else BESS_BWD_MI(RED_L2, false, false);
#undef BESS_BWD_MI
#undef BESS_BWD_ARGS
    return BESS_OK;
}

// packed-fp16 L1 path (l1_f16.hip): scorer / dtype / width fit, not switched off, 16-B aligned operands
static bool use_l1_pk(const bess_model_desc* d, const void* query, const void* neg_base) {
    return l1_pk_eligible(d) && !(d->reserved[0] & BESS_FLAG_FP32_MATH) &&
           reinterpret_cast<uintptr_t>(query) % 16 == 0 && reinterpret_cast<uintptr_t>(neg_base) % 16 == 0;
}

}  // namespace bess

using namespace bess;

extern "C" int bess_neg_score_shared_fwd_masked(const bess_model_desc* d, const float* query, int64_t n_query,
                                                const void* neg_base, const int32_t* neg_idx, int64_t n_neg,
                                                float* out, int64_t ld_out, const bess_kill_desc* kill,
                                                void* workspace, int64_t workspace_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    if (kill && d->scorer <= BESS_COMPLEX && n_query > 0 && n_neg > 0 && query && neg_base && out &&
        ld_out >= n_neg && use_l1_pk(d, query, neg_base)) {
        BESS_REQUIRE(kill->diag_step >= 0, "neg_score_shared_fwd_masked: negative diag_step");
        if (kill->mask) {
            BESS_REQUIRE(kill->mask_cols > 0 && kill->mask_cols <= n_neg, "neg_score_shared_fwd_masked: mask_cols");
            BESS_REQUIRE(kill->mask_rows == 1 || kill->mask_rows == 2 || kill->mask_rows == n_query,
                         "neg_score_shared_fwd_masked: mask_rows %lld not 1, 2 or n_query", (long long)kill->mask_rows);
        }
        if (kill->ht || (kill->mask && kill->mask_rows == 2))
            BESS_REQUIRE(kill->ppp >= 2 && (kill->ppp % 2) == 0 && (n_query % kill->ppp) == 0,
                         "neg_score_shared_fwd_masked: 'ht' needs an even block size dividing n_query");
        return l1_pk_fwd(d, query, n_query, neg_base, neg_idx, n_neg, out, ld_out, kill, as_stream(stream));
    }
    if (int e = bess_neg_score_shared_fwd_ws(d, query, n_query, neg_base, neg_idx, n_neg, out, ld_out, workspace,
                                             workspace_bytes, stream))
        return e;
    if (!kill || n_query == 0 || n_neg == 0) return BESS_OK;
    return bess_mask_scores(out, n_query, n_neg, ld_out, kill->diag_step, kill->ht, kill->ppp, kill->mask,
                            kill->mask_rows, kill->mask_cols, stream);
}

extern "C" int bess_neg_score_shared_fwd_loss(const bess_model_desc* d, const float* query, int64_t n_query,
                                              const void* neg_base, const int32_t* neg_idx, int64_t n_neg,
                                              float* out, int64_t ld_out, const bess_kill_desc* kill,
                                              const bess_loss_desc* l, const float* pos, const float* weight,
                                              int64_t weight_len, float* row_loss, float* loss, float* d_pos,
                                              float* d_neg, int64_t ld_dneg, int32_t* counters, void* workspace,
                                              int64_t workspace_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    BESS_REQUIRE(l && pos && weight && row_loss && loss && d_pos && d_neg && counters, "neg_score_shared_fwd_loss: NULL pointer");
    BESS_REQUIRE(n_query > 0 && n_neg > 0 && query && neg_base && out && ld_out >= n_neg && ld_dneg >= n_neg,
                 "neg_score_shared_fwd_loss: bad sizes / NULL pointer");
    BESS_REQUIRE(l->kind >= BESS_LOSS_LOGSIGMOID && l->kind <= BESS_LOSS_SSCE, "neg_score_shared_fwd_loss: unknown loss %d", l->kind);
    BESS_REQUIRE(weight_len == 1 || weight_len == n_query, "neg_score_shared_fwd_loss: weight_len must be 1 or n_query");
    // the scoring launch (K7 in its epilogue where the kernel has one), then the loss launch(es)
    if (int e = bess_neg_score_shared_fwd_masked(d, query, n_query, neg_base, neg_idx, n_neg, out, ld_out, kill, workspace,
                                                 workspace_bytes, stream))
        return e;
    return bess_loss_fwd_bwd_one_launch(l, pos, out, n_query, n_neg, ld_out, weight, weight_len, row_loss, loss, d_pos,
                                        d_neg, ld_dneg, nullptr, counters, stream);
}

extern "C" int64_t bess_neg_score_shared_workspace(const bess_model_desc* d, int64_t n_query, int64_t n_neg) {
    if (!d || check_desc(d) || n_query <= 0 || n_neg <= 0) return 0;
    if (d->scorer == BESS_BOXE || d->scorer == BESS_AFFINE || reduce_of(d) != RED_DOT) return 0;
    if (d->reserved[0] & BESS_FLAG_FP32_MATH) return 0;  // the exact fp32 MFMA kernels are asked for
    return gemm_split_workspace(n_query, n_neg, d->width);
}

extern "C" int bess_neg_score_shared_fwd_ws(const bess_model_desc* d, const float* query, int64_t n_query,
                                            const void* neg_base, const int32_t* neg_idx, int64_t n_neg,
                                            float* out, int64_t ld_out, void* workspace,
                                            int64_t workspace_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    BESS_REQUIRE(n_query >= 0 && n_neg >= 0, "neg_score_shared_fwd: bad sizes");
    if (n_query == 0 || n_neg == 0) return BESS_OK;
    BESS_REQUIRE(query && neg_base && out, "neg_score_shared_fwd: NULL pointer");
    BESS_REQUIRE(ld_out >= n_neg, "neg_score_shared_fwd: leading dimension < n_neg");
    if (d->scorer == BESS_BOXE)
        return boxe_negatives(d, true, true, query, n_query, neg_base, neg_idx, n_neg, out, nullptr, ld_out, nullptr,
                              nullptr, as_stream(stream));
    if (d->scorer == BESS_AFFINE) {
        BESS_REQUIRE(!neg_idx, "neg_score_shared_fwd: affine scorers take dense f32 candidates (bess_normalize_rows)");
        return affine_shared_fwd(d, query, n_query, static_cast<const float*>(neg_base), n_neg, out, ld_out,
                                 as_stream(stream));
    }
    if (reduce_of(d) == RED_DOT) {  // bilinear scorers: matrix cores
        const bool fp32 = d->reserved[0] & BESS_FLAG_FP32_MATH;
        const int64_t want = workspace && !fp32 ? gemm_split_workspace(n_query, n_neg, d->width) : 0;
        if (want > 0 && workspace_bytes >= want)
            return gemm_split_fwd(d->dtype, query, n_query, neg_base, neg_idx, n_neg, d->width, out, ld_out,
                                  workspace, workspace_bytes, as_stream(stream));
        return gemm_dot_fwd(d->dtype, query, n_query, neg_base, neg_idx, n_neg, d->width, out, ld_out,
                            as_stream(stream));
    }
    if (use_l1_pk(d, query, neg_base))
        return l1_pk_fwd(d, query, n_query, neg_base, neg_idx, n_neg, out, ld_out, nullptr, as_stream(stream));
    RowSrc<float> Q{query, nullptr, n_query};
    if (d->dtype == BESS_F32)
        return run_fwd<float>(d, Q, RowSrc<float>{static_cast<const float*>(neg_base), neg_idx, n_neg}, out,
                              ld_out, as_stream(stream));
    return run_fwd<half_t>(d, Q, RowSrc<half_t>{static_cast<const half_t*>(neg_base), neg_idx, n_neg}, out,
                           ld_out, as_stream(stream));
}

extern "C" int bess_neg_score_shared_fwd_pruned(const bess_model_desc* d, const float* query, int64_t n_query,
                                                const void* neg_base, const int32_t* neg_idx, int64_t n_neg,
                                                float* out, int64_t ld_out, const float* thr, uint8_t* flags,
                                                int64_t ld_flags, void* workspace, int64_t workspace_bytes,
                                                void* stream) {
    if (int e = check_desc(d)) return e;
    BESS_REQUIRE(n_query >= 0 && n_neg >= 0, "neg_score_shared_fwd_pruned: bad sizes");
    if (n_query == 0 || n_neg == 0) return BESS_OK;
    BESS_REQUIRE(query && neg_base && out && thr && flags, "neg_score_shared_fwd_pruned: NULL pointer");
    BESS_REQUIRE(ld_out >= n_neg && ld_flags >= ceil_div(n_neg, 64), "neg_score_shared_fwd_pruned: leading dimensions");
    hipStream_t st = as_stream(stream);
    // every block starts out flagged: kernels without a pruning epilogue (and the fp32 fallback of the split
    // product) write all scores, which is what a set flag promises
    BESS_REQUIRE(ld_flags % 4 == 0 && reinterpret_cast<uintptr_t>(flags) % 4 == 0,
                 "neg_score_shared_fwd_pruned: flag rows must be 4-byte aligned");
    {
        hipError_t e = fill_words_async(flags, 0x01010101u, n_query * ld_flags / 4, st);
        if (e != hipSuccess) return fail(static_cast<int>(e), "fill: %s", hipGetErrorString(e));
    }
    if (d->scorer <= BESS_COMPLEX) {
        if (reduce_of(d) == RED_DOT) {
            const bool fp32 = d->reserved[0] & BESS_FLAG_FP32_MATH;
            const int64_t want = workspace && !fp32 ? gemm_split_workspace(n_query, n_neg, d->width) : 0;
            if (want > 0 && workspace_bytes >= want)
                return gemm_split_fwd(d->dtype, query, n_query, neg_base, neg_idx, n_neg, d->width, out, ld_out,
                                      workspace, workspace_bytes, st, thr, flags, ld_flags);
        } else if (use_l1_pk(d, query, neg_base)) {
            return l1_pk_fwd(d, query, n_query, neg_base, neg_idx, n_neg, out, ld_out, nullptr, st, thr, flags, ld_flags);
        }
    }
    return bess_neg_score_shared_fwd_ws(d, query, n_query, neg_base, neg_idx, n_neg, out, ld_out, workspace,
                                        workspace_bytes, stream);
}

// ---- ranks without the score matrix ---------------------------------------------------------------------
// counts of a stored score tile (scorers / shapes whose kernel has no counting epilogue): one wave per row
__global__ __launch_bounds__(256) void k_count_scores(const float* __restrict__ sc, int64_t ld, int64_t n_row,
                                                      int64_t n_col, const float* __restrict__ thr,
                                                      const int32_t* __restrict__ excl, int64_t col0,
                                                      int32_t* __restrict__ counts, int round16) {
    const int lane = threadIdx.x & 63;
    const int64_t r = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (r >= n_row) return;
    const float th = thr[r];
    const int64_t ex = static_cast<int64_t>(excl[r]) - col0;
    const float* row = sc + r * ld;
    int cg = 0, ce = 0;
    for (int64_t j = lane; j < n_col; j += 64) {
        const float v = count_value(row[j], round16);
        const bool in = j != ex;
        cg += (in && v > th) ? 1 : 0;
        ce += (in && v == th) ? 1 : 0;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        cg += __shfl_xor(cg, o, 64);
        ce += __shfl_xor(ce, o, 64);
    }
    if (lane == 0) {
        if (cg) atomicAdd(counts + 2 * r, cg);
        if (ce) atomicAdd(counts + 2 * r + 1, ce);
    }
}

constexpr int64_t COUNT_TILE_BYTES = 64ll << 20;  // score tile of the scorers without a counting epilogue
static int64_t count_tile_cols(int64_t n_query, int64_t n_neg) {
    int64_t c = COUNT_TILE_BYTES / 4 / n_query / 64 * 64;
    if (c < 64) c = 64;
    return c < n_neg ? c : (n_neg + 3) / 4 * 4;
}
static bool counts_in_epilogue(const bess_model_desc* d, const float* query, const void* neg_base, int64_t n_query,
                               int64_t n_neg, int64_t* ws_want) {
    *ws_want = 0;
    if (d->scorer > BESS_COMPLEX) return false;
    if (reduce_of(d) == RED_DOT) {
        if (d->reserved[0] & BESS_FLAG_FP32_MATH) return false;
        *ws_want = gemm_split_workspace(n_query, n_neg, d->width);
        return *ws_want > 0;
    }
    return !query || use_l1_pk(d, query, neg_base);
}

extern "C" int64_t bess_neg_score_shared_fwd_counts_workspace(const bess_model_desc* d, int64_t n_query,
                                                              int64_t n_neg) {
    if (!d || check_desc(d) || n_query <= 0 || n_neg <= 0) return 0;
    int64_t want = 0;
    if (counts_in_epilogue(d, nullptr, nullptr, n_query, n_neg, &want) && want > 0) return want;
    // (the packed L1 kernel needs none, but whether it applies depends on the pointers' alignment: the tile is
    // what the call falls back on)
    return n_query * count_tile_cols(n_query, n_neg) * 4;
}

extern "C" int bess_neg_score_shared_fwd_counts(const bess_model_desc* d, const float* query, int64_t n_query,
                                                const void* neg_base, const int32_t* neg_idx, int64_t n_neg,
                                                const float* thr, const int32_t* excl, int32_t* counts,
                                                int32_t round_f16, void* workspace, int64_t workspace_bytes,
                                                void* stream) {
    if (int e = check_desc(d)) return e;
    BESS_REQUIRE(n_query >= 0 && n_neg >= 0, "neg_score_shared_fwd_counts: bad sizes");
    if (n_query == 0 || n_neg == 0) return BESS_OK;
    BESS_REQUIRE(query && neg_base && thr && excl && counts, "neg_score_shared_fwd_counts: NULL pointer");
    BESS_REQUIRE(n_neg < (1ll << 31), "neg_score_shared_fwd_counts: candidate ids are int32");
    hipStream_t st = as_stream(stream);
    const CountArgs cnt{excl, counts, 0, round_f16 ? 1 : 0};
    int64_t want = 0;
    if (counts_in_epilogue(d, query, neg_base, n_query, n_neg, &want)) {
        if (reduce_of(d) == RED_DOT) {
            if (workspace && workspace_bytes >= want)
                return gemm_split_fwd(d->dtype, query, n_query, neg_base, neg_idx, n_neg, d->width, nullptr, 0,
                                      workspace, workspace_bytes, st, thr, nullptr, 0, &cnt);
        } else {
            // one launch per 65,536 candidates (32 MiB of fp16 rows at W = 256): the launch's rows stay in the
            // Infinity Cache while its query tiles pass over them - over all 2.5 M rows of a wikikg2 shard in one
            // launch every row tile would stream the table from HBM again
            constexpr int64_t L1_COUNT_CHUNK = 65536;
            for (int64_t j0 = 0; j0 < n_neg; j0 += L1_COUNT_CHUNK) {
                const int64_t nc = n_neg - j0 < L1_COUNT_CHUNK ? n_neg - j0 : L1_COUNT_CHUNK;
                const CountArgs cj{excl, counts, j0, cnt.round16};
                const void* base = neg_idx ? neg_base : static_cast<const char*>(neg_base) + j0 * d->width * 2;
                if (int e = l1_pk_fwd(d, query, n_query, base, neg_idx ? neg_idx + j0 : nullptr, nc, nullptr, 0, nullptr,
                                      st, thr, nullptr, 0, &cj))
                    return e;
            }
            return BESS_OK;
        }
    }
    // no counting epilogue for this scorer / shape: score tiles through the workspace, counted by a second kernel
    BESS_REQUIRE(d->scorer != BESS_AFFINE || !neg_idx,
                 "neg_score_shared_fwd_counts: affine scorers take dense f32 candidates (bess_normalize_rows)");
    const int64_t cols = count_tile_cols(n_query, n_neg);
    BESS_REQUIRE(workspace && workspace_bytes >= n_query * cols * 4 && reinterpret_cast<uintptr_t>(workspace) % 16 == 0,
                 "neg_score_shared_fwd_counts: workspace too small (bess_neg_score_shared_fwd_counts_workspace) or misaligned");
    float* tile = static_cast<float*>(workspace);
    const int64_t row_bytes = static_cast<int64_t>(d->width) * ((d->scorer == BESS_AFFINE || d->dtype == BESS_F32) ? 4 : 2);
    for (int64_t j0 = 0; j0 < n_neg; j0 += cols) {
        const int64_t nc = n_neg - j0 < cols ? n_neg - j0 : cols;
        const void* base = neg_idx ? neg_base : static_cast<const char*>(neg_base) + j0 * row_bytes;
        if (int e = bess_neg_score_shared_fwd_ws(d, query, n_query, base, neg_idx ? neg_idx + j0 : nullptr, nc, tile, cols,
                                                 nullptr, 0, stream))
            return e;
        k_count_scores<<<static_cast<unsigned>(ceil_div(n_query, 4)), 256, 0, st>>>(tile, cols, n_query, nc, thr, excl,
                                                                                    j0, counts, cnt.round16);
        if (int e = check_launch("count_scores")) return e;
    }
    return BESS_OK;
}

// ---- single (query, candidate) scores in the arithmetic of the all-entity kernels -------------------------
// out[i] = score(query[i], candidate idx[i]) as bess_neg_score_shared_fwd_counts on a (like_n_query x like_n_neg)
// problem computes it: the SAME kernel on the diagonal tiles of the (pairs x pairs) problem - 128 x 128 tiles of the
// split-fp16 product, 64 x 64 of the packed L1 kernel - or, for the kernels without a diagonal form, on
// 1024 x 1024 blocks whose diagonal is kept (the per-element arithmetic of these kernels does not depend on where
// in the matrix an element sits).
constexpr int64_t PAIR_CHUNK = 1024, PAIR_DIAG_CHUNK = 8192;
__global__ void k_take_diagonal(const float* __restrict__ tile, int64_t ld, int64_t n, int64_t block,
                                float* __restrict__ out) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) out[i] = tile[i * ld + i % block];  // (block >= n: the plain diagonal; else tiles stored side by side)
}

extern "C" int64_t bess_neg_score_shared_fwd_pairs_workspace(const bess_model_desc* d, int64_t like_n_query,
                                                             int64_t like_n_neg) {
    if (!d || check_desc(d) || like_n_query <= 0 || like_n_neg <= 0) return 0;
    int64_t want = 0;
    if (counts_in_epilogue(d, nullptr, nullptr, like_n_query, like_n_neg, &want) && want > 0)
        return PAIR_DIAG_CHUNK * 128 * 4 + gemm_split_workspace_any(PAIR_DIAG_CHUNK, PAIR_DIAG_CHUNK, d->width);
    const int64_t blocks = PAIR_CHUNK * PAIR_CHUNK * 4, diag = PAIR_DIAG_CHUNK * 64 * 4;
    return blocks > diag ? blocks : diag;
}

extern "C" int bess_neg_score_shared_fwd_pairs(const bess_model_desc* d, const float* query, const void* neg_base,
                                               const int32_t* neg_idx, int64_t n_pair, int64_t like_n_query,
                                               int64_t like_n_neg, float* out, void* workspace,
                                               int64_t workspace_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    BESS_REQUIRE(n_pair >= 0 && like_n_query > 0 && like_n_neg > 0, "neg_score_shared_fwd_pairs: bad sizes");
    if (n_pair == 0) return BESS_OK;
    BESS_REQUIRE(query && neg_base && neg_idx && out, "neg_score_shared_fwd_pairs: NULL pointer");
    BESS_REQUIRE(d->scorer <= BESS_COMPLEX, "neg_score_shared_fwd_pairs: TransE / RotatE / DistMult / ComplEx only");
    BESS_REQUIRE(workspace && workspace_bytes >= bess_neg_score_shared_fwd_pairs_workspace(d, like_n_query, like_n_neg) &&
                     reinterpret_cast<uintptr_t>(workspace) % 16 == 0,
                 "neg_score_shared_fwd_pairs: workspace too small (bess_neg_score_shared_fwd_pairs_workspace) or misaligned");
    hipStream_t st = as_stream(stream);
    float* tile = static_cast<float*>(workspace);
    int64_t want = 0;
    const bool epi = counts_in_epilogue(d, query, neg_base, like_n_query, like_n_neg, &want);
    const bool gemm = epi && reduce_of(d) == RED_DOT;
    const int64_t qw = d->width;  // (the four native scorers: query rows as wide as entity rows)
    const int64_t chunk = epi ? PAIR_DIAG_CHUNK : PAIR_CHUNK;
    const int64_t block = gemm ? 128 : (epi ? 64 : PAIR_CHUNK);  // columns of `tile`
    void* gws = static_cast<char*>(workspace) + PAIR_DIAG_CHUNK * 128 * 4;
    const int64_t gws_bytes = workspace_bytes - PAIR_DIAG_CHUNK * 128 * 4;
    for (int64_t p0 = 0; p0 < n_pair; p0 += chunk) {
        const int64_t np = n_pair - p0 < chunk ? n_pair - p0 : chunk;
        const float* q = query + p0 * qw;
        int rc;
        if (gemm)
            rc = gemm_split_fwd(d->dtype, q, np, neg_base, neg_idx + p0, np, d->width, tile, 128, gws, gws_bytes, st,
                                nullptr, nullptr, 0, nullptr, true);
        else if (epi)
            rc = l1_pk_fwd(d, q, np, neg_base, neg_idx + p0, np, tile, 64, nullptr, st, nullptr, nullptr, 0, nullptr, true);
        else
            rc = bess_neg_score_shared_fwd_ws(d, q, np, neg_base, neg_idx + p0, np, tile, PAIR_CHUNK, nullptr, 0, stream);
        if (rc) return rc;
        k_take_diagonal<<<static_cast<unsigned>(ceil_div(np, 256)), 256, 0, st>>>(tile, block, np, block, out + p0);
        if (int e = check_launch("take_diagonal")) return e;
    }
    return BESS_OK;
}

extern "C" int bess_neg_score_shared_fwd(const bess_model_desc* d, const float* query,
                                         int64_t n_query, const void* neg_base,
                                         const int32_t* neg_idx, int64_t n_neg, float* out,
                                         int64_t ld_out, void* stream) {
    return bess_neg_score_shared_fwd_ws(d, query, n_query, neg_base, neg_idx, n_neg, out, ld_out, nullptr, 0, stream);
}

extern "C" int64_t bess_neg_score_shared_bwd_workspace(const bess_model_desc* d, int64_t n_query, int64_t n_neg) {
    if (!d || check_desc(d) || n_query <= 0 || n_neg <= 0) return 0;
    if (d->scorer == BESS_BOXE || d->scorer == BESS_AFFINE || reduce_of(d) != RED_DOT) return 0;
    if (d->reserved[0] & BESS_FLAG_FP32_MATH) return 0;
    return gemm_split_bwd_workspace(n_query, n_neg, d->width);
}

extern "C" int bess_neg_score_shared_bwd_parts_plan(const bess_model_desc* d, int64_t n_query, int64_t n_neg,
                                                    int32_t* n_dq_parts, int32_t* n_dneg_parts) {
    BESS_REQUIRE(n_dq_parts && n_dneg_parts, "neg_score_shared_bwd_parts_plan: NULL out");
    *n_dq_parts = *n_dneg_parts = 0;
    if (!d || check_desc(d) || d->scorer > BESS_COMPLEX || !use_l1_bwd_parts(d, n_query, n_neg)) return BESS_OK;
    const L1PartsPlan p = plan_l1_bwd_parts(n_query, n_neg, d->width);
    *n_dq_parts = static_cast<int32_t>(p.groups);
    *n_dneg_parts = static_cast<int32_t>(p.slices);
    return BESS_OK;
}

extern "C" int bess_neg_score_shared_bwd_parts(const bess_model_desc* d, const float* query, int64_t n_query,
                                               const void* neg_base, const int32_t* neg_idx, int64_t n_neg,
                                               const float* d_out, int64_t ld_dout, float* dq_parts,
                                               float* dneg_parts, void* stream) {
    if (int e = check_desc(d)) return e;
    BESS_REQUIRE(n_query > 0 && n_neg > 0 && query && neg_base && d_out && dq_parts && dneg_parts && ld_dout >= n_neg,
                 "neg_score_shared_bwd_parts: NULL pointer / bad sizes");
    if (d->scorer > BESS_COMPLEX || !use_l1_bwd_parts(d, n_query, n_neg))
        return fail(BESS_EUNSUPPORTED, "neg_score_shared_bwd_parts: this scorer / shape has no partial-sum form "
                                       "(bess_neg_score_shared_bwd_parts_plan says 0 parts)");
    const bool r16 = use_l1_pk(d, query, neg_base);
    RowSrc<float> Q{query, nullptr, n_query};
    hipStream_t st = as_stream(stream);
    if (d->dtype == BESS_F32)
        return run_l1_bwd_parts<float>(d, Q, RowSrc<float>{static_cast<const float*>(neg_base), neg_idx, n_neg}, d_out,
                                       ld_dout, dq_parts, dneg_parts, st, false);
    return run_l1_bwd_parts<half_t>(d, Q, RowSrc<half_t>{static_cast<const half_t*>(neg_base), neg_idx, n_neg}, d_out,
                                    ld_dout, dq_parts, dneg_parts, st, r16);
}

extern "C" int bess_neg_score_shared_bwd(const bess_model_desc* d, const float* query,
                                         int64_t n_query, const void* neg_base,
                                         const int32_t* neg_idx, int64_t n_neg, const float* out,
                                         int64_t ld_out, const float* d_out, int64_t ld_dout,
                                         float* d_query, float* d_neg, void* stream) {
    return bess_neg_score_shared_bwd_ws(d, query, n_query, neg_base, neg_idx, n_neg, out, ld_out, d_out, ld_dout,
                                        d_query, d_neg, nullptr, 0, stream);
}

extern "C" int bess_neg_score_shared_bwd_ws(const bess_model_desc* d, const float* query,
                                            int64_t n_query, const void* neg_base,
                                            const int32_t* neg_idx, int64_t n_neg, const float* out,
                                            int64_t ld_out, const float* d_out, int64_t ld_dout,
                                            float* d_query, float* d_neg, void* workspace,
                                            int64_t workspace_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    BESS_REQUIRE(n_query >= 0 && n_neg >= 0, "neg_score_shared_bwd: bad sizes");
    if (n_query == 0 || n_neg == 0) return BESS_OK;
    BESS_REQUIRE(query && neg_base && d_out && d_query && d_neg, "neg_score_shared_bwd: NULL pointer");
    BESS_REQUIRE(reduce_of(d) != RED_L2 || out, "neg_score_shared_bwd: p=2 needs the forward scores");
    BESS_REQUIRE(ld_dout >= n_neg && (!out || ld_out >= n_neg), "neg_score_shared_bwd: leading dimension < n_neg");
    hipStream_t st = as_stream(stream);
    if (d->scorer == BESS_BOXE)
        return boxe_negatives(d, false, true, query, n_query, neg_base, neg_idx, n_neg, nullptr, d_out, ld_dout,
                              d_query, d_neg, st);
    if (d->scorer == BESS_AFFINE) {
        BESS_REQUIRE(!neg_idx, "neg_score_shared_bwd: affine scorers take dense f32 candidates (bess_normalize_rows)");
        BESS_REQUIRE(d->norm_p == 1 || out, "neg_score_shared_bwd: p != 1 needs the forward scores");
        return affine_shared_bwd(d, query, n_query, static_cast<const float*>(neg_base), n_neg, out, ld_out, d_out,
                                 ld_dout, d_query, d_neg, st);
    }
    if (reduce_of(d) == RED_DOT) {
        const bool fp32 = d->reserved[0] & BESS_FLAG_FP32_MATH;
        const int64_t want = workspace && !fp32 ? gemm_split_bwd_workspace(n_query, n_neg, d->width) : 0;
        if (want > 0 && workspace_bytes >= want)
            return gemm_split_bwd(d->dtype, d_out, ld_dout, n_query, query, neg_base, neg_idx, n_neg, d->width,
                                  d_query, d_neg, workspace, workspace_bytes, st);
        if (int e = gemm_dot_dq(d->dtype, d_out, ld_dout, n_query, neg_base, neg_idx, n_neg, d->width, d_query, st))
            return e;
        return gemm_dot_de(d_out, ld_dout, n_query, query, n_neg, d->width, d_neg, st);
    }
    // the packed-fp16 forward scores the query rounded to fp16: differentiate that function
    const bool r16 = use_l1_pk(d, query, neg_base);
    RowSrc<float> Q{query, nullptr, n_query};
    if (d->dtype == BESS_F32) {
        RowSrc<float> E{static_cast<const float*>(neg_base), neg_idx, n_neg};
        if (int e = run_bwd<float>(d, Q, E, d_out, ld_dout, out, ld_out, d_query, d_neg, st, false)) return e;
    } else {
        RowSrc<half_t> E{static_cast<const half_t*>(neg_base), neg_idx, n_neg};
        if (int e = run_bwd<half_t>(d, Q, E, d_out, ld_dout, out, ld_out, d_query, d_neg, st, r16)) return e;
    }
    return check_launch("neg_score_shared_bwd");
}

// K4 for TransE / RotatE with p = 1 on fp16 tables (BASELINE configs[3]: ogbl-wikikg2 TransE
// d=256 fp16, flat shared negatives): packed-fp16 form of the L1 distance matrix (forward), with
// K7 (mask / augment kill) in its epilogue.
//
// The reference's fp16 mode is `model.half()`: the query `h + r` is an fp16 tensor and the
// distance matrix is computed from fp16 operands (reference scoring.py:194-197, 342, 354;
// notebooks/3_wikikg2_fp16.ipynb:300-392).  Here the query is rounded to fp16 (round to nearest
// even) when a tile is staged, candidates are the fp16 table rows as they are; everything
// after the first subtraction is exact or fp32:
//
//   forward   sum_w |q - e| = 2 sum_w max(q, e) - sum_w q - sum_w e.  max of two fp16 numbers is
//             exact, v_dot2_f32_f16 adds two of them into an fp32 accumulator: two packed
//             instructions per PAIR of elements (v_pk_max_f16, v_dot2_f32_f16) instead of two scalar
//             ones per element (v_sub_f32, v_add_f32 |.|) - and there is no rounded difference at all.
//   backward  k_neg_shared_bwd<ROUND16> (neg_shared.hip): the fp32 kernel with the query rounded to
//             fp16 as it is loaded, so that it differentiates exactly this function (sgn(q16 - e) is
//             exact in fp32, 0 at a tie).  A packed form of it was built and measured (v_pk_add_f16
//             for the difference, its int16 bits clamped to [-1, 1] by v_pk_min_i16 / v_pk_max_i16 =
//             sgn with sgn(0) = 0, v_dot2_i32_i16 against int16-quantised coefficients, exact int32
//             accumulation): 4 packed instructions per 2 elements - and SLOWER than the 3 fp32
//             instructions per element (799 vs 637 us at S = N = 4096, W = 256), because on gfx950
//             every VOP3P / dot2 instruction issues at 0.6 of the plain VALU rate
//             (profiles/ubench/valu_pk.hip: v_pk_max_f16, v_pk_add_f16, v_pk_min_i16, v_pk_fma_f16,
//             v_dot2c_f32_f16, v_dot2c_i32_i16 all 36.7 T lane-instructions/s against 60.8 for
//             v_add_f32 / v_xor_b32).  Two elements per packed instruction are therefore worth 1.2
//             scalar instructions, not 2: the forward (2 packed per pair against 2 scalar per element)
//             gains 1.25x, a backward needs <= 3 packed per pair to gain anything.
//
// Tiles: 64 x 64 outputs per 256-thread workgroup, 4 x 4 per lane, operands staged through LDS
// as packed dwords ([pair][row] images, one ds_read_b128 per operand and pair step), rows
// gathered by index straight into the image, next stage prefetched into registers.
#include "common.h"

namespace bess {

typedef _Float16 h2v __attribute__((ext_vector_type(2)));

constexpr int PT = 64;    // tile edge
constexpr int PLD = 68;   // padded leading dimension of the LDS images (dwords)
constexpr int FKH = 32;   // forward: halfs of W per stage (16 packed pairs)

// single-instruction helper (hipcc puts a canonicalising v_pk_max_f16 x, x in front of a packed max of loaded bits)
__device__ __forceinline__ uint32_t pk_max_f16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
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

// Pruned scoring (top-k over all entities, bess_neg_score_shared_fwd_pruned): a row's 64-column block is stored
// only when one of its scores is above the row's threshold; flags[q, block] says which were.
struct PruneArgs {
    const float* thr;  // [nq] or NULL: store everything
    uint8_t* flags;    // [nq, ldf]
    int64_t ldf;
    CountArgs cnt;     // cnt.counts != NULL: count the scores above / equal to thr instead, store nothing
    int diag;          // only the diagonal 64 x 64 tiles (nq == ne), stored side by side: out is [nq, 64]
};

// out[q, j] = -sum_w |fp16(Q[q, w]) - E[idx[j], w]|        W % 32 == 0
// MI: query rows per thread (4: 64 x 64 tile; 2: 32 x 64 and 1: 16 x 64 tiles for launches whose grid of larger tiles
// leaves CUs idle - at the notebook micro-batch, 512 x 544, a 32-row tile's wave runs ~2 k packed instructions alone
// on its SIMD: the launch is issue-bound per wave while three quarters of the chip's issue slots are empty)
// NS: 32-half slices of W per stage (per pair of barriers).  (NS = 2 for the 16-row tiles - half the stages, the same
// order of sums - was measured at the notebook micro-batch: 10.7 against 10.6 us; the launch is not a chain of stage
// latencies.  Every launch uses NS = 1.)
template <int MI, int NS = 1>
__global__ __launch_bounds__(256) void k_l1_fwd_pk(const float* __restrict__ Q, int64_t nq,
                                                   const half_t* __restrict__ E, const int32_t* __restrict__ eidx,
                                                   int64_t ne, int W, float* __restrict__ out, int64_t ld,
                                                   KillArgs kill, PruneArgs prune) {
    __shared__ __attribute__((aligned(16))) uint32_t Qs[NS * FKH / 2][PLD];
    __shared__ __attribute__((aligned(16))) uint32_t Es[NS * FKH / 2][PLD];
    __shared__ float rsq[PT], rse[PT];
    const int t = threadIdx.x;
    const int tx = t & 15, ty = t >> 4;
    constexpr int QT = 16 * MI;  // query rows of the tile
    const int64_t q0 = static_cast<int64_t>(prune.diag ? blockIdx.x : blockIdx.y) * QT;
    const int64_t j0 = static_cast<int64_t>(blockIdx.x) * PT;
    float acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    // staging: thread t brings 8 scalars of row m of each operand per stage (rows past the end are
    // clamped, their results dropped at the store)
    const int m = t >> 2, kc = t & 3;
    const bool stage_q = m < QT;  // wave-uniform (16 rows per wave)
    const float* qp = Q + min(q0 + (stage_q ? m : 0), nq - 1) * W + kc * 8;
    const int64_t er = min(j0 + m, ne - 1);
    const half_t* ep = E + (eidx ? static_cast<int64_t>(eidx[er]) : er) * W + kc * 8;
    float4 qa[NS], qb[NS];
    uint4 ev[NS];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            qa[sl] = *reinterpret_cast<const float4*>(qp + k0 + sl * FKH);
            qb[sl] = *reinterpret_cast<const float4*>(qp + k0 + sl * FKH + 4);
            ev[sl] = *reinterpret_cast<const uint4*>(ep + k0 + sl * FKH);
        }
    };
    fetch(0);
    float sq = 0.f, se = 0.f;  // sums of the staged (rounded) values of row m
    for (int k0 = 0; k0 < W; k0 += NS * FKH) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const uint32_t qd[4] = {pack_rn(qa[sl].x, qa[sl].y), pack_rn(qa[sl].z, qa[sl].w), pack_rn(qb[sl].x, qb[sl].y),
                                    pack_rn(qb[sl].z, qb[sl].w)};
            const uint32_t ed[4] = {ev[sl].x, ev[sl].y, ev[sl].z, ev[sl].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (stage_q) Qs[sl * (FKH / 2) + kc * 4 + i][m] = qd[i];
                Es[sl * (FKH / 2) + kc * 4 + i][m] = ed[i];
                sq = dot2_ones(qd[i], sq);
                se = dot2_ones(ed[i], se);
            }
        }
        __syncthreads();
        if (k0 + NS * FKH < W) fetch(k0 + NS * FKH);
#pragma unroll
        for (int k = 0; k < NS * FKH / 2; ++k) {
            uint32_t a[MI];
            if constexpr (MI == 4) {
                const uint4 a4 = *reinterpret_cast<const uint4*>(&Qs[k][ty * 4]);
                a[0] = a4.x, a[1] = a4.y, a[2] = a4.z, a[3] = a4.w;
            } else if constexpr (MI == 2) {
                const uint2 a2 = *reinterpret_cast<const uint2*>(&Qs[k][ty * 2]);
                a[0] = a2.x, a[1] = a2.y;
            } else {
                a[0] = Qs[k][ty];
            }
            const uint4 b4 = *reinterpret_cast<const uint4*>(&Es[k][tx * 4]);
            const uint32_t b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < MI; ++i)
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
        if (stage_q) rsq[m] = sq;
        rse[m] = se;
    }
    __syncthreads();
    const bool any_kill = kill.diag_step > 0 || kill.mask;  // wave-uniform
    const bool row16 = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const int64_t mask_from = kill.n_neg - kill.mask_cols;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int64_t q = q0 + ty * MI + i;
        if (q >= nq) continue;
        float* o = out + q * ld;
        int64_t kcol = -1;
        const uint8_t* mrow = nullptr;
        if (any_kill) kill_row(kill, q, &kcol, &mrow);
        float v4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t jj = j0 + tx * 4 + j;
            float v = -((2.f * acc[i][j] - rsq[ty * MI + i]) - rse[tx * 4 + j]);
            if (any_kill && jj < ne) {
                bool kl = jj == kcol;
                // the mask overrides the diagonal on its columns (bess.py:227-228)
                if (mrow && jj >= mask_from) kl = mrow[jj - mask_from] == 0;
                if (kl) v += BESS_BAD_NEGATIVE_SCORE;
            }
            v4[j] = v;
        }
        const int64_t jj0 = j0 + tx * 4;
        const int64_t oc = jj0 - (prune.diag ? j0 : 0);  // column of the thread's first score in `out`
        if (prune.cnt.counts) {  // (wave-uniform) ranks: the row's counts over these 64 columns, nothing stored
            const float th = prune.thr[q];
            const int64_t ex = static_cast<int64_t>(prune.cnt.excl[q]) - prune.cnt.col0;  // the row's excluded column
            int cg = 0, ce = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = jj0 + j < ne && jj0 + j != ex;
                const float v = count_value(v4[j], prune.cnt.round16);
                cg += (in && v > th) ? 1 : 0;
                ce += (in && v == th) ? 1 : 0;
            }
            // the 16 threads with this ty hold the row: one DPP row reduction for both counts (each <= 64)
            const int both = static_cast<int>(row16_allreduce_sum(static_cast<float>(cg + (ce << 12))));
            cg = both & 4095, ce = both >> 12;
            if (tx == 0) {
                if (cg) atomicAdd(prune.cnt.counts + 2 * q, cg);
                if (ce) atomicAdd(prune.cnt.counts + 2 * q + 1, ce);
            }
            continue;
        }
        if (prune.thr) {  // (wave-uniform; the 16 threads with this ty hold the row's 64 columns: one 16-lane group)
            const float th = prune.thr[q];
            bool hit = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) hit = hit || (jj0 + j < ne && v4[j] > th);
            const unsigned long long m = __ballot(hit);
            const bool any = ((m >> (16 * ((t & 63) >> 4))) & 0xffffull) != 0;
            if (tx == 0) prune.flags[q * prune.ldf + blockIdx.x] = any ? 1 : 0;
            if (!any) continue;
        }
        if (row16 && jj0 + 3 < ne) {  // 16-byte aligned rows: one store for the thread's four scores
            *reinterpret_cast<float4*>(o + oc) = make_float4(v4[0], v4[1], v4[2], v4[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (jj0 + j < ne) o[oc + j] = v4[j];
        }
    }
}

bool l1_pk_eligible(const bess_model_desc* d) {
    return d->dtype == BESS_F16 && is_distance(d->scorer) && d->scorer <= BESS_ROTATE && d->norm_p == 1 &&
           d->width % FKH == 0;
}

int l1_pk_fwd(const bess_model_desc* d, const float* query, int64_t n_query, const void* neg_base,
              const int32_t* neg_idx, int64_t n_neg, float* out, int64_t ld_out, const bess_kill_desc* k,
              hipStream_t st, const float* thr, uint8_t* flags, int64_t ld_flags, const CountArgs* count, bool diag) {
    const PruneArgs pr{thr, flags, ld_flags, count ? *count : CountArgs{nullptr, nullptr, 0, 0}, diag ? 1 : 0};
    if (diag) {  // pair scores: 64 x 64 tiles on the diagonal, out [n, 64]
        BESS_REQUIRE(n_query == n_neg && ld_out == PT && !thr && !count && !k, "l1_pk_fwd: bad diagonal problem");
        k_l1_fwd_pk<4><<<dim3(static_cast<unsigned>(ceil_div(n_neg, PT)), 1), 256, 0, st>>>(
            query, n_query, static_cast<const half_t*>(neg_base), neg_idx, n_neg, d->width, out, ld_out, KillArgs{}, pr);
        return check_launch("neg_score_shared_fwd (packed f16 L1, diagonal tiles)");
    }
    KillArgs ka{};
    if (k) ka = KillArgs{k->diag_step, k->ht, k->ppp, k->mask, k->mask_rows, k->mask ? k->mask_cols : 0, n_neg};
    // 32-row tiles when the 64-row grid would leave most CUs without a workgroup
    const bool small = ceil_div(n_neg, PT) * ceil_div(n_query, PT) < 192;
    // (16-row tiles, kernel alone: 10.6 vs 12.1 us at 512 x 544, 9.5 vs 11.7 at 256 x 288, equal from 1024 x 544 on)
    const bool tiny = ceil_div(n_neg, PT) * ceil_div(n_query, PT / 2) < 256;  // not even one 32-row tile per CU
    const dim3 grid(static_cast<unsigned>(ceil_div(n_neg, PT)),
                    static_cast<unsigned>(ceil_div(n_query, tiny ? PT / 4 : small ? PT / 2 : PT)));
    if (tiny)
        k_l1_fwd_pk<1><<<grid, 256, 0, st>>>(query, n_query, static_cast<const half_t*>(neg_base), neg_idx, n_neg,
                                                    d->width, out, ld_out, ka, pr);
    else if (small)
        k_l1_fwd_pk<2><<<grid, 256, 0, st>>>(query, n_query, static_cast<const half_t*>(neg_base), neg_idx, n_neg,
                                                    d->width, out, ld_out, ka, pr);
    else
        k_l1_fwd_pk<4><<<grid, 256, 0, st>>>(query, n_query, static_cast<const half_t*>(neg_base), neg_idx, n_neg,
                                                    d->width, out, ld_out, ka, pr);
    return check_launch("neg_score_shared_fwd (packed f16 L1)");
}

}  // namespace bess

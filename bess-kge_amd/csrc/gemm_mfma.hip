// fp32 GEMM on the CDNA4 matrix cores for the bilinear scorers (DistMult /
// ComplEx) with shared negatives:
//
//     C[m, n] = sum_k A(m, k) * B(k, n)            (exact fp32, v_mfma_f32_32x32x2_f32)
//
//   forward   out[q, j]  = Q[q, :] . E[idx[j], :]        (reference scoring.py:251-252)
//   backward  dQ[q, w]   = sum_j G[q, j] E[idx[j], w]
//             dE[j, w]   = sum_q G[q, j] Q[q, w]          (reduced over the micro-batch on chip)
//
// The f32-input MFMA is bit-for-bit a k-ordered fmaf chain (no reduced precision:
// gfx950 has no xf32), so parity with the fp32 reference is unaffected; it runs
// at the fp32 vector peak but leaves the VALU free and needs one VGPR per
// operand per lane.
//
// Tiling: 128 x 128 output tile per 256-thread workgroup (4 waves, 64 x 64 per
// wave = 2 x 2 MFMA tiles of 32 x 32, 64 accumulator registers), K staged
// through LDS in slices of 32 as [k][m] / [k][n] images so that a fragment read
// is one ds_read_b32 per lane (lanes 0-31 consecutive m, lanes 32-63 the next
// k).  Global -> register -> LDS staging is split around the MFMA phase
// (loads for slice s+1 are issued before the MFMAs of slice s, written to the
// other LDS buffer after them).
//
// Operands are "row sources" (base, optional int32 row index, leading dim):
//   ROWS_MN: row r holds A(m = r, k = 0..K-1)   contiguous along k  (Q, E for forward; G for dQ)
//   ROWS_K : row r holds A(k = r, m = 0..M-1)   contiguous along m  (E for dQ; G^T and Q for dE)
// so negative rows are gathered by index straight into LDS.
#include "common.h"

namespace bess {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GK = 32;  // k slice
// TILE (edge of the square output tile, 128 or 64) is a template parameter: 64 is
// used when a 128-tiling would leave CUs idle (the backward GEMMs have only W
// output columns).  LDS images: LD_T = TILE + 1 for transposing stores (ROWS_MN,
// odd stride), LD_D = TILE + 4 for direct 16-B stores (ROWS_K, 16-B aligned rows).

struct GemmOperand {
    const void* base;
    const int32_t* idx;  // optional row index
    int64_t rows;        // number of rows (M|N for ROWS_MN, K for ROWS_K)
    int64_t ld;          // elements between consecutive rows
};

// registers holding one thread's share of a TILE x 32 slice: TILE/32 x float4
template <int TILE>
struct Stage {
    float v[TILE / 32][4];
};

template <typename T, bool ROWS_K, int TILE>
__device__ __forceinline__ void stage_load(const GemmOperand& op, int64_t mn0, int64_t mn_end, int k0,
                                           int64_t k_end, Stage<TILE>& st) {
    constexpr int P = TILE / 32;       // passes
    constexpr int CPR = TILE / 4;      // 16-B chunks per k-row (ROWS_K)
    constexpr int KPP = 256 / CPR;     // k-rows per pass (ROWS_K)
    const T* base = static_cast<const T*>(op.base);
    const int t = threadIdx.x;
    if (!ROWS_K) {
        // thread -> (row = p*32 + t/8, k chunk = (t%8)*4); 16 B (f32) along k
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int64_t r = mn0 + p * 32 + (t >> 3);
            const int kk = k0 + (t & 7) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) st.v[p][i] = 0.f;
            if (r < mn_end) {
                const int64_t row = op.idx ? static_cast<int64_t>(op.idx[r]) : r;
                const T* rp = base + row * op.ld + kk;
                if (kk + 3 < k_end && (op.ld & 3) == 0) {
                    VecLoad<T, 4>::load(rp, st.v[p]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (kk + i < k_end) st.v[p][i] = to_f32(rp[i]);
                }
            }
        }
    } else {
        // thread -> (k = p*KPP + t/CPR, mn chunk = (t%CPR)*4); 16 B along m/n
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int64_t k = k0 + p * KPP + t / CPR;
            const int64_t c = mn0 + (t % CPR) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) st.v[p][i] = 0.f;
            if (k < k_end) {
                const int64_t row = op.idx ? static_cast<int64_t>(op.idx[k]) : k;
                const T* rp = base + row * op.ld + c;
                if (c + 3 < mn_end && (op.ld & 3) == 0) {
                    VecLoad<T, 4>::load(rp, st.v[p]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (c + i < mn_end) st.v[p][i] = to_f32(rp[i]);
                }
            }
        }
    }
}

template <bool ROWS_K, int TILE>
__device__ __forceinline__ void stage_store(float* img, const Stage<TILE>& st) {
    constexpr int P = TILE / 32;
    constexpr int CPR = TILE / 4;
    constexpr int KPP = 256 / CPR;
    constexpr int LD_T = TILE + 1, LD_D = TILE + 4;
    const int t = threadIdx.x;
    if (!ROWS_K) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int m = p * 32 + (t >> 3);
            const int kc = (t & 7) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) img[(kc + i) * LD_T + m] = st.v[p][i];
        }
    } else {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int k = p * KPP + t / CPR;
            const int c = (t % CPR) * 4;
            *reinterpret_cast<float4*>(&img[k * LD_D + c]) =
                make_float4(st.v[p][0], st.v[p][1], st.v[p][2], st.v[p][3]);
        }
    }
}

template <typename TA, bool A_ROWS_K, typename TB, bool B_ROWS_K, int TILE>
__global__ __launch_bounds__(256) void k_gemm_f32_mfma(GemmOperand A, GemmOperand B, int64_t M, int64_t N,
                                                       int64_t K, float* __restrict__ C, int64_t ldc) {
    constexpr int GT = TILE;
    constexpr int WT = TILE / 2;   // per-wave tile edge (64 or 32)
    constexpr int MT = WT / 32;    // 32x32 MFMA tiles per wave and dimension
    constexpr int LD_T = TILE + 1, LD_D = TILE + 4;
    constexpr int LDA = A_ROWS_K ? LD_D : LD_T;
    constexpr int LDB = B_ROWS_K ? LD_D : LD_T;
    __shared__ __attribute__((aligned(16))) float As[2][GK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK * LDB];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * WT, wn = (wave & 1) * WT;
    const int64_t m0 = static_cast<int64_t>(blockIdx.y) * GT;
    const int64_t n0 = static_cast<int64_t>(blockIdx.x) * GT;
    const int l31 = lane & 31, lk = lane >> 5;

    f32x16 acc[MT][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Stage<TILE> sa, sb;
    stage_load<TA, A_ROWS_K, TILE>(A, m0, M, 0, K, sa);
    stage_load<TB, B_ROWS_K, TILE>(B, n0, N, 0, K, sb);
    stage_store<A_ROWS_K, TILE>(As[0], sa);
    stage_store<B_ROWS_K, TILE>(Bs[0], sb);
    __syncthreads();

    const int n_slice = static_cast<int>((K + GK - 1) / GK);
    for (int s = 0; s < n_slice; ++s) {
        const int cur = s & 1;
        const bool more = s + 1 < n_slice;
        if (more) {  // issue next slice's global loads ahead of this slice's MFMAs
            stage_load<TA, A_ROWS_K, TILE>(A, m0, M, (s + 1) * GK, K, sa);
            stage_load<TB, B_ROWS_K, TILE>(B, n0, N, (s + 1) * GK, K, sb);
        }
        const float* a_img = As[cur];
        const float* b_img = Bs[cur];
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2) {
            float a[MT], b[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                a[i] = a_img[(kk + lk) * LDA + wm + 32 * i + l31];
                b[i] = b_img[(kk + lk) * LDB + wn + 32 * i + l31];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            stage_store<A_ROWS_K, TILE>(As[cur ^ 1], sa);
            stage_store<B_ROWS_K, TILE>(Bs[cur ^ 1], sb);
        }
        __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int64_t col = n0 + wn + j * 32 + l31;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < M) C[row * ldc + col] = acc[i][j][r];
            }
        }
}

template <typename TA, bool AK, typename TB, bool BK>
static int launch(const GemmOperand& A, const GemmOperand& B, int64_t M, int64_t N, int64_t K, float* C,
                  int64_t ldc, hipStream_t st) {
    // 128-tiles unless they would leave most of the 256 CUs without a workgroup
    if (ceil_div(N, 128) * ceil_div(M, 128) >= 384) {
        const dim3 grid(static_cast<unsigned>(ceil_div(N, 128)), static_cast<unsigned>(ceil_div(M, 128)));
        k_gemm_f32_mfma<TA, AK, TB, BK, 128><<<grid, 256, 0, st>>>(A, B, M, N, K, C, ldc);
    } else {
        const dim3 grid(static_cast<unsigned>(ceil_div(N, 64)), static_cast<unsigned>(ceil_div(M, 64)));
        k_gemm_f32_mfma<TA, AK, TB, BK, 64><<<grid, 256, 0, st>>>(A, B, M, N, K, C, ldc);
    }
    return check_launch("gemm_f32_mfma");
}

// out[q, j] = Q[q] . E[idx[j]]
int gemm_dot_fwd(int dtype, const float* Q, int64_t S, const void* E, const int32_t* idx, int64_t N, int W,
                 float* out, int64_t ld, hipStream_t st) {
    GemmOperand A{Q, nullptr, S, W};
    GemmOperand B{E, idx, N, W};
    if (dtype == BESS_F32) return launch<float, false, float, false>(A, B, S, N, W, out, ld, st);
    return launch<float, false, half_t, false>(A, B, S, N, W, out, ld, st);
}

// dQ[q, w] = sum_j G[q, j] E[idx[j], w]
int gemm_dot_dq(int dtype, const float* G, int64_t ldg, int64_t S, const void* E, const int32_t* idx, int64_t N,
                int W, float* dQ, hipStream_t st) {
    GemmOperand A{G, nullptr, S, ldg};
    GemmOperand B{E, idx, N, W};
    if (dtype == BESS_F32) return launch<float, false, float, true>(A, B, S, W, N, dQ, W, st);
    return launch<float, false, half_t, true>(A, B, S, W, N, dQ, W, st);
}

// dE[j, w] = sum_q G[q, j] Q[q, w]
int gemm_dot_de(const float* G, int64_t ldg, int64_t S, const float* Q, int64_t N, int W, float* dE,
                hipStream_t st) {
    GemmOperand A{G, nullptr, S, ldg};  // rows are k = q, contiguous along m = j
    GemmOperand B{Q, nullptr, S, W};    // rows are k = q, contiguous along n = w
    return launch<float, true, float, true>(A, B, N, W, S, dE, W, st);
}

}  // namespace bess

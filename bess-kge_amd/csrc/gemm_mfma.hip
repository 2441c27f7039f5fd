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
    const int32_t* run_if;  // optional gate (device word): the kernel returns at once while it is 0
};

// registers holding one thread's share of a TILE x 32 slice: TILE/32 x float4
template <int TILE>
struct Stage {
    float v[TILE / 32][4];
};

// Per-thread loader of one operand.  What a thread stages never changes shape across
// the K loop, so everything that does not depend on the slice is computed once:
//   ROWS_MN: the P row pointers (row index loaded once, rows past the end clamped - their
//            products land in output rows / columns that are never stored);
//   ROWS_K : the column chunk (clamped likewise); the row of slice s + 1 (one index load
//            per pass) is fetched one slice ahead, so no load in the loop waits on another.
// Before this, every pass loaded its row index and waited for it (s_waitcnt vmcnt(0))
// before issuing the row load: 8 dependent round trips per slice in front of the MFMAs.
// FAST (chosen by the host when every leading dimension and K are multiples of 4): all loads
// are unconditional 16-B loads at clamped addresses and the k tail is zeroed with selects, so
// the 2 x P loads of a slice are issued back to back.  (With per-lane bounds branches the
// compiler separates the passes with s_waitcnt vmcnt(0): 8 serialised round trips per slice.)
template <typename T, bool ROWS_K, int TILE, bool FAST>
struct Loader {
    static constexpr int P = TILE / 32;
    static constexpr int CPR = TILE / 4;
    static constexpr int KPP = 256 / CPR;
    const T* ptr[P];     // ROWS_MN: row base + k chunk;  ROWS_K: unused
    int64_t row[P];      // ROWS_K: table row of the slice that will be loaded next
    const T* base;
    const int32_t* idx;
    int64_t ld, mn_end, k_end, col;
    bool vec;            // 16-B loads allowed (rows are 16-B aligned and the chunk is whole)

    __device__ __forceinline__ void init(const GemmOperand& op, int64_t mn0, int64_t mn_end_, int64_t k_end_) {
        const int t = threadIdx.x & 255;
        base = static_cast<const T*>(op.base);
        idx = op.idx;
        ld = op.ld;
        mn_end = mn_end_;
        k_end = k_end_;
        vec = (ld & 3) == 0;
        if (!ROWS_K) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int64_t r = min(mn0 + p * 32 + (t >> 3), mn_end - 1);
                const int64_t rr = idx ? static_cast<int64_t>(idx[r]) : r;
                ptr[p] = base + rr * ld + (t & 7) * 4;
            }
        } else {
            col = mn0 + (t % CPR) * 4;
            if (vec && mn_end >= 4) col = min(col, mn_end - 4);  // clamped columns are never stored
            else vec = false;
            next_rows(0);
        }
    }
    // ROWS_K: fetch the row indices of the slice starting at k0 (consumed by the next load())
    __device__ __forceinline__ void next_rows(int k0) {
        if (ROWS_K) {
            const int t = threadIdx.x & 255;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int64_t k = min(static_cast<int64_t>(k0) + p * KPP + t / CPR, k_end - 1);
                row[p] = idx ? static_cast<int64_t>(idx[k]) : k;
            }
        }
    }
    __device__ __forceinline__ void load(int k0, Stage<TILE>& st) const {
        const int t = threadIdx.x & 255;
        if (FAST) {
            if (!ROWS_K) {
                const int64_t kk = k0 + (t & 7) * 4;
                const int64_t kc = min(kk, k_end - 4) - (t & 7) * 4;  // ptr already holds the chunk offset
#pragma unroll
                for (int p = 0; p < P; ++p) VecLoad<T, 4>::load(ptr[p] + kc, st.v[p]);
            } else {
#pragma unroll
                for (int p = 0; p < P; ++p) VecLoad<T, 4>::load(base + row[p] * ld + col, st.v[p]);
            }
            return;
        }
        if (!ROWS_K) {
            const int kk = k0 + (t & 7) * 4;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (vec && kk + 3 < k_end) {
                    VecLoad<T, 4>::load(ptr[p] + k0, st.v[p]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) st.v[p][i] = (kk + i < k_end) ? to_f32(ptr[p][k0 + i]) : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int64_t k = static_cast<int64_t>(k0) + p * KPP + t / CPR;
                const T* rp = base + row[p] * ld + col;
                if (vec) {
                    VecLoad<T, 4>::load(rp, st.v[p]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) st.v[p][i] = (col + i < mn_end) ? to_f32(rp[i]) : 0.f;
                }
                if (k >= k_end) {  // rows past the end of the reduction contribute nothing
#pragma unroll
                    for (int i = 0; i < 4; ++i) st.v[p][i] = 0.f;
                }
            }
        }
    }
    // FAST: zero what lies past the end of the reduction.  Called after the MFMAs of the
    // current slice, so that nothing between the load issue and the MFMAs waits on the loads.
    __device__ __forceinline__ void fix(int k0, Stage<TILE>& st) const {
        if (!FAST) return;
        const int t = threadIdx.x & 255;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const bool live = ROWS_K ? (static_cast<int64_t>(k0) + p * KPP + t / CPR < k_end)
                                     : (static_cast<int64_t>(k0) + (t & 7) * 4 < k_end);
#pragma unroll
            for (int i = 0; i < 4; ++i) st.v[p][i] = live ? st.v[p][i] : 0.f;
        }
    }
};

template <bool ROWS_K, int TILE>
__device__ __forceinline__ void stage_store(float* img, const Stage<TILE>& st) {
    constexpr int P = TILE / 32;
    constexpr int CPR = TILE / 4;
    constexpr int KPP = 256 / CPR;
    constexpr int LD_T = TILE + 1, LD_D = TILE + 4;
    const int t = threadIdx.x & 255;
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

template <typename TA, bool A_ROWS_K, typename TB, bool B_ROWS_K, int TILE, bool FAST>
__global__ __launch_bounds__(256) void k_gemm_f32_mfma(GemmOperand A, GemmOperand B, int64_t M, int64_t N,
                                                       int64_t K, float* __restrict__ C, int64_t ldc) {
    if (A.run_if && *A.run_if == 0) return;  // fallback launch of the split-fp16 path that is not needed
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

    // Two workgroups share a CU (one wave each per SIMD) and run the same code at the same rate:
    // started together they reach their load / LDS-store phases together and the matrix pipe
    // idles.  Workgroups are dispatched round-robin over 256 CUs, so (i, i + 256) are the
    // co-resident pairs: delay every other group of 256 once, by about half a store phase.
    if (((blockIdx.y * gridDim.x + blockIdx.x) >> 8) & 1) __builtin_amdgcn_s_sleep(24);
    Stage<TILE> sa, sb;
    Loader<TA, A_ROWS_K, TILE, FAST> la;
    Loader<TB, B_ROWS_K, TILE, FAST> lb;
    la.init(A, m0, M, K);
    lb.init(B, n0, N, K);
    la.load(0, sa);
    lb.load(0, sb);
    la.next_rows(GK);
    lb.next_rows(GK);
    la.fix(0, sa);
    lb.fix(0, sb);
    stage_store<A_ROWS_K, TILE>(As[0], sa);
    stage_store<B_ROWS_K, TILE>(Bs[0], sb);
    __syncthreads();

    const int n_slice = static_cast<int>((K + GK - 1) / GK);
    for (int s = 0; s < n_slice; ++s) {
        const int cur = s & 1;
        const bool more = s + 1 < n_slice;
        if (more) {  // issue next slice's global loads ahead of this slice's MFMAs
            la.load((s + 1) * GK, sa);
            lb.load((s + 1) * GK, sb);
            la.next_rows((s + 2) * GK);  // and the row indices of the slice after that
            lb.next_rows((s + 2) * GK);
        }
        const float* a_img = As[cur];
        const float* b_img = Bs[cur];
        // fragments of step kk + 2 are read while the MFMAs of step kk issue (two register
        // sets): the LDS latency is off the MFMA issue path
        float a[2][MT], b[2][MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            a[0][i] = a_img[lk * LDA + wm + 32 * i + l31];
            b[0][i] = b_img[lk * LDB + wn + 32 * i + l31];
        }
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2) {
            const int c = (kk >> 1) & 1;
            if (kk + 2 < GK) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    a[c ^ 1][i] = a_img[(kk + 2 + lk) * LDA + wm + 32 * i + l31];
                    b[c ^ 1][i] = b_img[(kk + 2 + lk) * LDB + wn + 32 * i + l31];
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the reads above the MFMAs (the scheduler sinks them)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
            la.fix((s + 1) * GK, sa);
            lb.fix((s + 1) * GK, sb);
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

// Producer / consumer variant of the 128 x 128 kernel: 512 threads, waves 0-3 only issue
// MFMAs (+ the LDS fragment reads), waves 4-7 only move the next K slice global -> registers
// -> LDS.  A workgroup's waves are dealt to the SIMDs cyclically, so every SIMD hosts one
// consumer and one producer and its matrix pipe is fed by a wave that does nothing else
// (profiles/ubench/mfma_f32.hip: this inner loop alone sustains 142-149 TFLOP/s with one wave
// per SIMD).  In the symmetric kernel the two co-resident workgroups run in lock step and reach
// their load / store phases together; rocprofv3 shows the pipe 69 % busy there.
template <typename TA, bool A_ROWS_K, typename TB, bool B_ROWS_K>
__global__ __launch_bounds__(512) void k_gemm_f32_mfma_ws(GemmOperand A, GemmOperand B, int64_t M, int64_t N,
                                                          int64_t K, float* __restrict__ C, int64_t ldc,
                                                          int tiles_x, int n_tiles) {
    if (A.run_if && *A.run_if == 0) return;
    constexpr int TILE = 128, WT = 64, MT = 2;
    constexpr int LD_T = TILE + 1, LD_D = TILE + 4;
    constexpr int LDA = A_ROWS_K ? LD_D : LD_T;
    constexpr int LDB = B_ROWS_K ? LD_D : LD_T;
    __shared__ __attribute__((aligned(16))) float As[2][GK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK * LDB];
    const bool producer = threadIdx.x >= 256;  // wave-uniform
    const int n_slice = static_cast<int>((K + GK - 1) / GK);
    // One tile per workgroup, row-major (the hardware's dispatch order then gives every XCD the
    // tile columns c = x (mod 8) of each row: its B panels stay in that XCD's L2).  The loop form
    // also runs as a persistent grid (fewer workgroups than tiles); on this shape that measured
    // 93 vs 101 TFLOP/s, with either tile order and with one or two slices in flight.
    const int slot = blockIdx.x, slots = gridDim.x;
    const int t_begin = 0, t_end = n_tiles;
    auto tile_origin = [&](int li, int64_t& m0, int64_t& n0) {
        m0 = static_cast<int64_t>(li / tiles_x) * TILE;
        n0 = static_cast<int64_t>(li % tiles_x) * TILE;
    };

    if (producer) {
        Stage<TILE> sa, sb;
        Loader<TA, A_ROWS_K, TILE, true> la;
        Loader<TB, B_ROWS_K, TILE, true> lb;
        for (int tile = t_begin + slot; tile < t_end; tile += slots) {
            int64_t m0, n0;
            tile_origin(tile, m0, n0);
            la.init(A, m0, M, K);
            lb.init(B, n0, N, K);
            la.load(0, sa);
            lb.load(0, sb);
            la.next_rows(GK);
            lb.next_rows(GK);
            for (int s = 0; s < n_slice; ++s) {
                la.fix(s * GK, sa);
                lb.fix(s * GK, sb);
                stage_store<A_ROWS_K, TILE>(As[s & 1], sa);
                stage_store<B_ROWS_K, TILE>(Bs[s & 1], sb);
                if (s + 1 < n_slice) {  // in flight while the consumers work on slice s
                    la.load((s + 1) * GK, sa);
                    lb.load((s + 1) * GK, sb);
                    la.next_rows((s + 2) * GK);
                    lb.next_rows((s + 2) * GK);
                }
                __syncthreads();  // slice s is in LDS; buffer (s + 1) & 1 was released one barrier ago
            }
            __syncthreads();      // the consumers are done with the last slice of this tile
        }
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * WT, wn = (wave & 1) * WT;
    const int l31 = lane & 31, lk = lane >> 5;
    for (int tile = t_begin + slot; tile < t_end; tile += slots) {
        int64_t m0, n0;
        tile_origin(tile, m0, n0);
        f32x16 acc[MT][MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        __syncthreads();  // slice 0 stored
        for (int s = 0; s < n_slice; ++s) {
            const float* a_img = As[s & 1];
            const float* b_img = Bs[s & 1];
            float a[2][MT], b[2][MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                a[0][i] = a_img[lk * LDA + wm + 32 * i + l31];
                b[0][i] = b_img[lk * LDB + wn + 32 * i + l31];
            }
#pragma unroll
            for (int kk = 0; kk < GK; kk += 2) {
                const int c = (kk >> 1) & 1;
                if (kk + 2 < GK) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        a[c ^ 1][i] = a_img[(kk + 2 + lk) * LDA + wm + 32 * i + l31];
                        b[c ^ 1][i] = b_img[(kk + 2 + lk) * LDB + wn + 32 * i + l31];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();  // done with slice s: its buffer may be refilled
        }
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
}

template <typename TA, bool AK, typename TB, bool BK>
static int launch(const GemmOperand& A, const GemmOperand& B, int64_t M, int64_t N, int64_t K, float* C,
                  int64_t ldc, hipStream_t st) {
    const bool fast = A.ld % 4 == 0 && B.ld % 4 == 0 && K % 4 == 0 && K >= 4 && M >= 4 && N >= 4;
    // 128-tiles unless they would leave most of the 256 CUs without a workgroup
    if (ceil_div(N, 128) * ceil_div(M, 128) >= 384) {
        const dim3 grid(static_cast<unsigned>(ceil_div(N, 128)), static_cast<unsigned>(ceil_div(M, 128)));
        if (fast) {
            const int tiles_x = static_cast<int>(ceil_div(N, 128));
            const int n_tiles = tiles_x * static_cast<int>(ceil_div(M, 128));
            k_gemm_f32_mfma_ws<TA, AK, TB, BK><<<n_tiles, 512, 0, st>>>(A, B, M, N, K, C, ldc, tiles_x,
                                                                                       n_tiles);
        }
        else k_gemm_f32_mfma<TA, AK, TB, BK, 128, false><<<grid, 256, 0, st>>>(A, B, M, N, K, C, ldc);
    } else {
        const dim3 grid(static_cast<unsigned>(ceil_div(N, 64)), static_cast<unsigned>(ceil_div(M, 64)));
        if (fast) k_gemm_f32_mfma<TA, AK, TB, BK, 64, true><<<grid, 256, 0, st>>>(A, B, M, N, K, C, ldc);
        else k_gemm_f32_mfma<TA, AK, TB, BK, 64, false><<<grid, 256, 0, st>>>(A, B, M, N, K, C, ldc);
    }
    return check_launch("gemm_f32_mfma");
}

// out[q, j] = Q[q] . E[idx[j]]
int gemm_dot_fwd(int dtype, const float* Q, int64_t S, const void* E, const int32_t* idx, int64_t N, int W,
                 float* out, int64_t ld, hipStream_t st, const int32_t* run_if) {
    GemmOperand A{Q, nullptr, S, W, run_if};
    GemmOperand B{E, idx, N, W, nullptr};
    if (dtype == BESS_F32) return launch<float, false, float, false>(A, B, S, N, W, out, ld, st);
    return launch<float, false, half_t, false>(A, B, S, N, W, out, ld, st);
}

// dQ[q, w] = sum_j G[q, j] E[idx[j], w]
int gemm_dot_dq(int dtype, const float* G, int64_t ldg, int64_t S, const void* E, const int32_t* idx, int64_t N,
                int W, float* dQ, hipStream_t st, const int32_t* run_if) {
    GemmOperand A{G, nullptr, S, ldg, run_if};
    GemmOperand B{E, idx, N, W, nullptr};
    if (dtype == BESS_F32) return launch<float, false, float, true>(A, B, S, W, N, dQ, W, st);
    return launch<float, false, half_t, true>(A, B, S, W, N, dQ, W, st);
}

// dE[j, w] = sum_q G[q, j] Q[q, w]
int gemm_dot_de(const float* G, int64_t ldg, int64_t S, const float* Q, int64_t N, int W, float* dE,
                hipStream_t st, const int32_t* run_if) {
    GemmOperand A{G, nullptr, S, ldg, run_if};  // rows are k = q, contiguous along m = j
    GemmOperand B{Q, nullptr, S, W, nullptr};   // rows are k = q, contiguous along n = w
    return launch<float, true, float, true>(A, B, N, W, S, dE, W, st);
}

}  // namespace bess

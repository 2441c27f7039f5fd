// Forward GEMM of the bilinear scorers (DistMult / ComplEx, shared negatives and the
// all-entities scoring of TopK / AllScores) on the fp16 matrix cores at fp32 accuracy:
//
//     out[q, j] = Q[q, :] . E[idx[j], :]                 (reference scoring.py:251-252)
//
// gfx950 has no fp32-input MFMA faster than the vector unit (v_mfma_f32_32x32x2_f32 = 157 TFLOP/s,
// gemm_mfma.hip), but v_mfma_f32_32x32x16_f16 runs 16 x faster per product.  Every fp32 operand
// is therefore split into two fp16 numbers
//
//     x = hi + lo / 2048,   hi = fp16(x),   lo = fp16((x - hi) * 2048)       (x - hi is exact)
//
// which together carry 22 significand bits (lo is kept scaled so that it is a normal fp16
// number whenever hi is), and the product is evaluated as
//
//     Q . E = Qhi.Ehi + (Qhi.Elo + Qlo.Ehi) / 2048          (fp32 accumulation inside the MFMA)
//
// in two fp32 accumulators (main, correction), summed once at the end.  The dropped lo.lo term
// and the rounding of lo are each <= 2^-22 of |x||y| per product - below the fp32 rounding of
// the reference's own accumulation over W (measured against a float64 product: 3-4e-7 of
// max|out| vs 1e-6 for an fp32 fma chain, profiles/bench_gemm_split.py) - so the parity
// tolerance (rtol 1e-4 / atol 1e-5, test_bess.py:245-246) is untouched; the cost is 3 fp16
// MFMAs per 16 k instead of 8 fp32 MFMAs of twice their length.  fp16 tables have lo == 0
// exactly: 2 MFMAs.  Operands must be finite and |x| < 65504 (fp16 range; KGE embeddings are
// O(1)): the pre-pass raises a flag in the workspace when it meets an element that is not, the
// split kernels then return at once and the exact fp32 MFMA kernels - queued behind them, gated on
// the same flag - compute the product (no host synchronisation); desc.reserved[0] &
// BESS_FLAG_FP32_MATH keeps the fp32 kernels from the start.
//
// Two kernels, with a caller-provided workspace between them:
//   k_split_rows   gathers the rows (by index), splits them and writes, per row and per block
//                  of 32 k, one 128-B line [32 hi | 32 lo] (k padded with zeros to a multiple
//                  of 32).  Splitting inside the GEMM instead costs 6 VALU operations per
//                  element for every tile that reads the element (32 x for a 4096 x 4096
//                  problem) and made the loader waves, not the matrix pipe, the bound
//                  (184 TFLOP/s; profiles/ubench/gemm_split_probe.hip).
//   k_gemm_split_f16   one 512-thread workgroup per CU, persistent over its 128 x 128 tiles:
//     waves 4-7 (producers) copy lines global -> registers (3 K slices in flight) -> LDS, over
//               a flat sequence of (tile, slice) pairs, so the loads of the next tile are in
//               flight while the consumers finish the current one.  The LDS image keeps the
//               line per row, its 16-B slots XOR-swizzled by the row so that the ds_read_b128
//               fragment reads are conflict free.
//     waves 0-3 (consumers): 64 x 64 of the tile each = 2 x 2 MFMA tiles x {main, corr}; per 16 k:
//               8 ds_read_b128, 12 MFMAs.  The barrier that releases a slice sits between its
//               two k steps (after both have been read into registers), so the fragment reads of
//               the next slice are issued under the MFMAs of this one.
#include "common.h"

namespace bess {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int SK = 32;            // k slice held by one LDS image
constexpr int ROW_B = 128;        // bytes per image row: 32 hi | 32 lo
constexpr int IMG_B = 128 * ROW_B;
#ifndef BESS_SPLIT_RING
#define BESS_SPLIT_RING 3
#endif
constexpr int RING = BESS_SPLIT_RING;  // K slices a producer keeps in flight (registers)

struct SplitSrc {
    const void* base;
    const int32_t* idx;  // optional row index
    int64_t rows;
    int64_t ld;
    int32_t* range_flag;  // set to 1 when an element cannot be split: not finite, or |x| >= 65504 (fp16 range)
};

// an element the split cannot represent (hi would be inf / nan): the product then falls back to the fp32 kernels
__device__ __forceinline__ bool unsplittable(float x) { return !(fabsf(x) < 65504.f); }

// 16-B slot `c` (0..7) of image row `row`, swizzled: rows r, r+2, .. r+14 of one parity land
// on the 8 slots of their half of the 256-B bank row
__device__ __forceinline__ int slot_off(int row, int c) { return row * ROW_B + ((c ^ ((row >> 1) & 7)) << 4); }

// ---- pre-pass: rows (gathered by index) -> split lines --------------------------------------
// One thread per 8 consecutive k of one row.  dst row r, block b (32 k): 128 B at
// dst + (r * n_blk + b) * 128 = [32 hi | 32 lo]; k >= W is written as zero.
template <typename T, bool VEC>
__device__ __forceinline__ void split_rows_body(const SplitSrc& src, int W, int n_blk, char* __restrict__ dst,
                                                int64_t gid) {
    const int chunks = n_blk * 4;  // 8-k chunks per row
    const int64_t r = gid / chunks;
    if (r >= src.rows) return;
    const int c = static_cast<int>(gid - r * chunks);
    const int k = c * 8;
    const int64_t rr = src.idx ? static_cast<int64_t>(src.idx[r]) : r;
    const T* row = static_cast<const T*>(src.base) + rr * src.ld;
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (VEC) {  // W % 8 == 0: a chunk is inside or outside
        if (k < W) VecLoad<T, 8>::load(row + k, x);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (k + i < W) x[i] = static_cast<float>(row[k + i]);
    }
    h8 hi, lo;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hi[i] = static_cast<_Float16>(x[i]);
        lo[i] = static_cast<_Float16>((x[i] - static_cast<float>(hi[i])) * 2048.f);
        bad |= unsplittable(x[i]);
    }
    if (bad && src.range_flag) atomicOr(src.range_flag, 1);
    char* line = dst + (r * n_blk + (c >> 2)) * ROW_B + (c & 3) * 16;
    *reinterpret_cast<h8*>(line) = hi;
    *reinterpret_cast<h8*>(line + 64) = lo;
}

// both operands of a product in one launch: blocks [0, blocks_a) split `a`, the rest `b`
template <typename TB, bool VEC>
__global__ __launch_bounds__(256) void k_split_rows(SplitSrc a, char* __restrict__ dst_a, int blocks_a, SplitSrc b,
                                                    char* __restrict__ dst_b, int W, int n_blk) {
    if (static_cast<int>(blockIdx.x) < blocks_a)
        split_rows_body<float, VEC>(a, W, n_blk, dst_a, static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x);
    else
        split_rows_body<TB, VEC>(b, W, n_blk, dst_b,
                                 static_cast<int64_t>(blockIdx.x - blocks_a) * 256 + threadIdx.x);
}

// Pre-pass over 32 x 32 tiles of a row source [R rows (by index)][C columns], through LDS:
//   PLAIN: image row r, k = columns   (line (r, bx)), as k_split_rows
//   TRANS: image row c, k = rows      (line (c, by)): the transposed operand of a backward product
// grid = (k blocks of the plain image, k blocks of the transposed image); blocks past the data
// write zero lines (the k padding that makes the slice count a multiple of the k split).
template <typename T, bool PLAIN, bool TRANS, bool VEC>
__global__ __launch_bounds__(256) void k_split_tile32(SplitSrc src, int64_t C, char* __restrict__ dst_plain,
                                                      int nblk_plain, char* __restrict__ dst_trans,
                                                      int nblk_trans) {
    __shared__ float tile[32][33];
    const int t = threadIdx.x;
    const int64_t r0 = static_cast<int64_t>(blockIdx.y) * 32, c0 = static_cast<int64_t>(blockIdx.x) * 32;
    const T* base = static_cast<const T*>(src.base);
    if (VEC) {  // C % 4 == 0, 16-B aligned rows: one 4-element load per thread covers the tile
        const int64_t r = r0 + (t >> 3), c = c0 + (t & 7) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < src.rows && c < C) {
            const int64_t rr = src.idx ? static_cast<int64_t>(src.idx[r]) : r;
            VecLoad<T, 4>::load(base + rr * src.ld + c, v);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) tile[t >> 3][(t & 7) * 4 + i] = v[i];
        if ((unsplittable(v[0]) || unsplittable(v[1]) || unsplittable(v[2]) || unsplittable(v[3])) && src.range_flag)
            atomicOr(src.range_flag, 1);
    } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int64_t r = r0 + p * 8 + (t >> 5), c = c0 + (t & 31);
            float v = 0.f;
            if (r < src.rows && c < C) {
                const int64_t rr = src.idx ? static_cast<int64_t>(src.idx[r]) : r;
                v = static_cast<float>(base[rr * src.ld + c]);
            }
            if (unsplittable(v) && src.range_flag) atomicOr(src.range_flag, 1);
            tile[p * 8 + (t >> 5)][t & 31] = v;
        }
    }
    __syncthreads();
    // 8 threads write one 128-B line with one 16-B store each: pieces 0-3 = hi of k 8j .. 8j + 7,
    // pieces 4-7 = lo of the same k (every thread computes both halves of its 8 numbers)
    const int row = t >> 3, piece = t & 7, k0 = (piece & 3) * 8;
    auto put = [&](const float (&x)[8], char* line) {
        h8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const _Float16 hi = static_cast<_Float16>(x[i]);
            const _Float16 lo = static_cast<_Float16>((x[i] - static_cast<float>(hi)) * 2048.f);
            o[i] = piece < 4 ? hi : lo;
        }
        *reinterpret_cast<h8*>(line + piece * 16) = o;
    };
    if (PLAIN && r0 + row < src.rows) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = tile[row][k0 + i];
        put(x, dst_plain + ((r0 + row) * nblk_plain + blockIdx.x) * ROW_B);
    }
    if (TRANS && c0 + row < C) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = tile[k0 + i][row];
        put(x, dst_trans + ((c0 + row) * nblk_trans + blockIdx.y) * ROW_B);
    }
}

// out[i] = part[0][i] + part[1][i] + ... (fixed order), 4 elements per thread
__global__ __launch_bounds__(256) void k_sum_parts(const float* __restrict__ part, int64_t stride, int n_part,
                                                   float* __restrict__ out, int64_t n4) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 a = reinterpret_cast<const float4*>(part)[i];
    for (int p = 1; p < n_part; ++p) {
        const float4 b = reinterpret_cast<const float4*>(part + p * stride)[i];
        a.x += b.x;
        a.y += b.y;
        a.z += b.z;
        a.w += b.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// one producer thread's share of a slice: 16-B piece (t & 7) of the lines of rows p * 32 + (t >> 3)
struct LineLoader {
    const char* ptr[4];
    __device__ __forceinline__ void init(const char* img, int64_t rows, int64_t pitch, int64_t r0) {
        const int t = threadIdx.x & 255;
#pragma unroll
        for (int p = 0; p < 4; ++p)  // rows past the end are clamped: their products are never stored
            ptr[p] = img + min(r0 + p * 32 + (t >> 3), rows - 1) * pitch + (t & 7) * 16;
    }
    __device__ __forceinline__ void load(int slice, u32x4 (&v)[4]) const {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#ifdef BESS_PROBE_NO_LOAD  // profiles/ubench/gemm_split_probe.hip: producers without global traffic
            v[p] = u32x4{static_cast<unsigned>(slice), 1u, 2u, 3u};
#else
            v[p] = *reinterpret_cast<const u32x4*>(ptr[p] + static_cast<int64_t>(slice) * ROW_B);
#endif
        }
    }
};

__device__ __forceinline__ void line_store(char* img, const u32x4 (&v)[4]) {
    const int t = threadIdx.x & 255;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(img + slot_off(p * 32 + (t >> 3), t & 7)) = v[p];
}

__device__ __forceinline__ f32x16 mma16(h8 a, h8 b, f32x16 c) {
#ifdef BESS_PROBE_NO_MFMA  // consumers without the matrix pipe
    c[0] += static_cast<float>(a[0]) * static_cast<float>(b[0]);
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

// profiles/ubench/gemm_split_probe.hip -DBESS_PROBE_TICKS: clocks the two roles spend waiting at the
// per-slice barrier (workgroup 0 overwrites C[0..3] with {producer wait, consumer wait, total, slices})
#ifdef BESS_PROBE_TICKS
#define BESS_TICK_BARRIER(acc)                                      \
    {                                                               \
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); \
        __syncthreads();                                            \
        acc += __builtin_amdgcn_s_memtime() - t0_;                  \
    }
#else
#define BESS_TICK_BARRIER(acc) __syncthreads()
#endif

// A, B: split images of M and N rows (n_slice lines each); C[m, n] = A[m] . B[n]
template <bool B_LO, int EPI>  // EPI: 0 stores, 1 pruned stores (top-k passes), 2 counts (ranks; nothing is stored)
// Split K (the backward products have few output tiles and a long k): a "tile" index t stands for
// output tile t / ksplit and the k range [t % ksplit, +1) * n_slice lines; part p of an output
// tile goes to C + p * part_stride (summed in a fixed order by k_sum_parts: deterministic).
__global__ __launch_bounds__(512) void k_gemm_split_f16(const char* __restrict__ A, const char* __restrict__ B,
                                                        int64_t M, int64_t N, int n_slice,
                                                        float* __restrict__ C, int64_t ldc, int tiles_x,
                                                        int n_tiles, int ksplit, int64_t part_stride,
                                                        const int32_t* __restrict__ range_flag, const float* __restrict__ thr,
                                                        uint8_t* __restrict__ pflags, int64_t ldf, CountArgs cnt) {
    __shared__ __attribute__((aligned(16))) char lds[2][2][IMG_B];  // [buffer][operand]
    if (range_flag && *range_flag) return;  // operands out of the fp16 range: the fp32 kernels take over
    const int slot = blockIdx.x, slots = gridDim.x;
    const int my_tiles = (n_tiles - slot + slots - 1) / slots;
    const int total = my_tiles * n_slice;  // (tile, slice) pairs of this workgroup, in order
    const int64_t pitch = static_cast<int64_t>(n_slice) * ksplit * ROW_B;

    unsigned long long pw = 0, cw = 0;  // BESS_PROBE_TICKS
    (void)pw;
    (void)cw;
#ifdef BESS_PROBE_TICKS
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
    if (threadIdx.x >= 256) {  // ---- producers
        u32x4 va[RING][4], vb[RING][4];  // register ring: slices g .. g + RING - 1
        LineLoader la, lb;
        // load cursor: runs RING - 1 slices ahead of the store cursor and stops on the last slice
        // (the surplus loads at the end re-read it and are never stored).  Every call issues the
        // same 8 loads unconditionally, so the compiler's vmcnt waits stay exact: the stores of
        // slice g wait for its loads only, those of the later slices stay in flight.
        int gl = 0, sl = 0, tl = slot;
        auto issue = [&](u32x4 (&xa)[4], u32x4 (&xb)[4]) {
            if (sl == 0) {
                const int ot = tl / ksplit;
                const int64_t k_off = static_cast<int64_t>(tl % ksplit) * n_slice * ROW_B;
                // (tiles_x == 0: only the diagonal tiles ot x ot, stored side by side as a [M, 128] matrix -
                // single (query, candidate) scores, bess_neg_score_shared_fwd_pairs)
                la.init(A + k_off, M, pitch, static_cast<int64_t>(tiles_x ? ot / tiles_x : ot) * 128);
                lb.init(B + k_off, N, pitch, static_cast<int64_t>(tiles_x ? ot % tiles_x : ot) * 128);
            }
            la.load(sl, xa);
            lb.load(sl, xb);
            if (++gl < total && ++sl == n_slice) {
                sl = 0;
                tl += slots;
            }
        };
        int gs = 0;  // store cursor
        auto put = [&](const u32x4 (&xa)[4], const u32x4 (&xb)[4]) {
            line_store(lds[gs & 1][0], xa);
            line_store(lds[gs & 1][1], xb);
            ++gs;
            BESS_TICK_BARRIER(pw);  // slice stored; the consumers have released the other buffer
        };
#pragma unroll
        for (int u = 0; u < RING - 1; ++u) issue(va[u], vb[u]);
        int g = 0;
        for (; g + RING <= total; g += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                issue(va[(u + RING - 1) % RING], vb[(u + RING - 1) % RING]);
                put(va[u], vb[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < RING - 1; ++u)  // the last total % RING slices are already in the ring
            if (g + u < total) put(va[u], vb[u]);
        __syncthreads();  // pairs with the consumers' barrier inside the last slice
#ifdef BESS_PROBE_TICKS
        if (blockIdx.x == 0 && threadIdx.x == 256) C[0] = static_cast<float>(pw);
#endif
        return;
    }

    // ---- consumers
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int l31 = lane & 31, lk = lane >> 5;
    // byte offsets of this lane's fragments inside an image: [k step][hi | lo], row block i adds 32 rows
    int off_a[2][2], off_b[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            off_a[ks][part] = slot_off(wm + l31, ks * 2 + lk + 4 * part);
            off_b[ks][part] = slot_off(wn + l31, ks * 2 + lk + 4 * part);
        }

    f32x16 accm[2][2], accc[2][2];
    auto zero = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    accm[i][j][r] = 0.f;
                    accc[i][j][r] = 0.f;
                }
    };
    zero();

    h8 fa[2][2][2], fb[2][2][2];  // [register set][row block][hi | lo]
#define BESS_RD(set, g, ks)                                                                          \
    {                                                                                                \
        const char* ia = lds[(g) & 1][0];                                                            \
        const char* ib = lds[(g) & 1][1];                                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                              \
            fa[set][i][0] = *reinterpret_cast<const h8*>(ia + off_a[ks][0] + i * 32 * ROW_B);        \
            fa[set][i][1] = *reinterpret_cast<const h8*>(ia + off_a[ks][1] + i * 32 * ROW_B);        \
            fb[set][i][0] = *reinterpret_cast<const h8*>(ib + off_b[ks][0] + i * 32 * ROW_B);        \
            if (B_LO) fb[set][i][1] = *reinterpret_cast<const h8*>(ib + off_b[ks][1] + i * 32 * ROW_B); \
        }                                                                                            \
    }
#define BESS_MM_MAIN(set)                                                                                   \
    {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)         \
            accm[i][j] = mma16(fa[set][i][0], fb[set][j][0], accm[i][j]); \
    }
#define BESS_MM_CORR(set)                                                                                   \
    {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)         \
            accc[i][j] = mma16(fa[set][i][1], fb[set][j][0], accc[i][j]); \
        if (B_LO) {                                                                                         \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)     \
                accc[i][j] = mma16(fa[set][i][0], fb[set][j][1], accc[i][j]); \
        }                                                                                                   \
    }

    // Per slice: [12 MFMAs of k-step 0, the 8 fragment reads of k-step 1 slotted one per MFMA gap]
    // barrier [12 MFMAs of k-step 1, with the reads of k-step 0 of the next slice].  A burst of
    // reads between two MFMAs would leave the pipe idle while they issue; one per gap is hidden
    // (profiles/ubench/mfma_f16.hip).  Every read has >= 4 MFMAs (128 cycles) before its first
    // use and the barrier finds the slice already in registers.
    constexpr int N_RD = B_LO ? 8 : 6, N_MM = B_LO ? 12 : 8;
#define BESS_INTERLEAVE()                                                  \
    {                                                                      \
        _Pragma("unroll") for (int q = 0; q < N_RD; ++q) {                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             \
        }                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, N_MM - N_RD, 0);       \
    }
    int s = 0, tile = slot;
    __syncthreads();  // slice 0 stored
    BESS_RD(0, 0, 0);
    for (int g = 0; g < total; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        BESS_RD(1, g, 1);
        BESS_MM_MAIN(0);
        BESS_MM_CORR(0);
        BESS_INTERLEAVE();
        __builtin_amdgcn_sched_barrier(0);
        BESS_TICK_BARRIER(cw);  // slice g is in registers (buffer released); slice g + 1 is stored
        __builtin_amdgcn_sched_barrier(0);
        BESS_RD(0, g + 1, 0);  // after the last slice: a stale image, never used
        BESS_MM_MAIN(1);
        BESS_MM_CORR(1);
        BESS_INTERLEAVE();
        __builtin_amdgcn_sched_barrier(0);
        if (++s == n_slice) {  // tile done: main + corr / 2048 -> C
            const int ot = tile / ksplit;
            const int64_t m0 = static_cast<int64_t>(tiles_x ? ot / tiles_x : ot) * 128 + wm;
            const int64_t n0 = static_cast<int64_t>(tiles_x ? ot % tiles_x : ot) * 128 + wn;
            // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
            float* c0 = C + (tile % ksplit) * part_stride + (m0 + 4 * lk) * ldc + (tiles_x ? n0 : wn) + l31;
            if constexpr (EPI == 2) {
                // counting epilogue (ranks): nothing is stored.  Lane l of the wave keeps the two counts of row
                // m0 + l: a row's 64 scores sit in the 32 lanes of one lk, so one ballot per comparison holds two
                // rows' verdicts (low half: row rr, high half: rr + 4), counted with s_bcnt1 and kept by the
                // row's lane; one pair of atomics per row and tile at the end
                const int l64 = threadIdx.x & 63;
                const bool rok = m0 + l64 < M;
                const int tv = __builtin_bit_cast(int, rok ? thr[m0 + l64] : INFINITY);
                // the row's excluded column, relative to this wave's first one (anything outside 0 .. 63: none here)
                int64_t exl = rok ? static_cast<int64_t>(cnt.excl[m0 + l64]) - (cnt.col0 + n0) : -1;
                const int ex = (exl >= 0 && exl < 64) ? static_cast<int>(exl) : -1;
                int cgt = 0, ceq = 0;
                const bool ok0 = n0 + l31 < N, ok1 = n0 + l31 + 32 < N;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = i * 32 + (r & 3) + 8 * (r >> 2);  // rows rr (lk = 0) and rr + 4 (lk = 1)
                        const float th0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tv, rr));
                        const float th1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tv, rr + 4));
                        const int e0 = __builtin_amdgcn_readlane(ex, rr), e1 = __builtin_amdgcn_readlane(ex, rr + 4);
                        const float th = lk ? th1 : th0;
                        const int e = lk ? e1 : e0;
                        const float v0 = count_value(accm[i][0][r] + accc[i][0][r] * (1.f / 2048.f), cnt.round16);
                        const float v1 = count_value(accm[i][1][r] + accc[i][1][r] * (1.f / 2048.f), cnt.round16);
                        const bool in0 = ok0 && l31 != e, in1 = ok1 && l31 + 32 != e;
                        const unsigned long long g0 = __ballot(in0 && v0 > th), g1 = __ballot(in1 && v1 > th);
                        const unsigned long long q0 = __ballot(in0 && v0 == th), q1 = __ballot(in1 && v1 == th);
                        const int gt_lo = __builtin_popcount(static_cast<unsigned>(g0)) + __builtin_popcount(static_cast<unsigned>(g1));
                        const int gt_hi = __builtin_popcount(static_cast<unsigned>(g0 >> 32)) + __builtin_popcount(static_cast<unsigned>(g1 >> 32));
                        const int eq_lo = __builtin_popcount(static_cast<unsigned>(q0)) + __builtin_popcount(static_cast<unsigned>(q1));
                        const int eq_hi = __builtin_popcount(static_cast<unsigned>(q0 >> 32)) + __builtin_popcount(static_cast<unsigned>(q1 >> 32));
                        if (l64 == rr) cgt = gt_lo, ceq = eq_lo;  // (every row of the wave is met exactly once)
                        if (l64 == rr + 4) cgt = gt_hi, ceq = eq_hi;
                    }
                if (rok && n0 < N) {
                    if (cgt) atomicAdd(cnt.counts + 2 * (m0 + l64), cgt);
                    if (ceq) atomicAdd(cnt.counts + 2 * (m0 + l64) + 1, ceq);
                }
            } else if constexpr (EPI == 1) {
                // pruned stores (top-k passes): the 32 lanes with this lk hold a row's 64 columns - the row's block
                // is written only when one of them is above the row's threshold, and flagged.  The thresholds of
                // the wave's 64 rows come in with ONE load (lane l: row m0 + l) and are read out per row with
                // v_readlane; the rows' verdicts are collected in a scalar mask and leave as one byte store per lane
                const int l64 = threadIdx.x & 63;
                const int tv = __builtin_bit_cast(int, (m0 + l64 < M) ? thr[m0 + l64] : INFINITY);
                unsigned long long rows_hit = 0;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = i * 32 + (r & 3) + 8 * (r >> 2);  // rows rr (lk = 0) and rr + 4 (lk = 1)
                        const float th0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tv, rr));
                        const float th1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tv, rr + 4));
                        const float th = lk ? th1 : th0;
                        const bool ok0 = n0 + l31 < N, ok1 = n0 + l31 + 32 < N;
                        const float v0 = accm[i][0][r] + accc[i][0][r] * (1.f / 2048.f);
                        const float v1 = accm[i][1][r] + accc[i][1][r] * (1.f / 2048.f);
                        const unsigned long long hits = __ballot((ok0 && v0 > th) || (ok1 && v1 > th));
                        const bool any0 = (hits & 0xffffffffull) != 0, any1 = (hits >> 32) != 0;
                        rows_hit |= (any0 ? 1ull << rr : 0ull) | (any1 ? 1ull << (rr + 4) : 0ull);
                        if (lk ? any1 : any0) {  // (rows past M have threshold +inf: never stored)
                            if (ok0) c0[rr * ldc] = v0;
                            if (ok1) c0[rr * ldc + 32] = v1;
                        }
                    }
                if (m0 + l64 < M && n0 < N) pflags[(m0 + l64) * ldf + n0 / 64] = (rows_hit >> l64) & 1ull;
            } else if (m0 + 64 <= M && n0 + 64 <= N) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            c0[(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc + j * 32] =
                                accm[i][j][r] + accc[i][j][r] * (1.f / 2048.f);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rr = i * 32 + (r & 3) + 8 * (r >> 2);
                            if (m0 + 4 * lk + rr < M && n0 + l31 + j * 32 < N)
                                c0[rr * ldc + j * 32] = accm[i][j][r] + accc[i][j][r] * (1.f / 2048.f);
                        }
            }
            zero();
            s = 0;
            tile += slots;
        }
    }
#ifdef BESS_PROBE_TICKS
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        C[1] = static_cast<float>(cw);
        C[2] = static_cast<float>(__builtin_amdgcn_s_memtime() - t_start);
        C[3] = static_cast<float>(total);
    }
#endif
#undef BESS_RD
#undef BESS_MM_MAIN
#undef BESS_MM_CORR
#undef BESS_INTERLEAVE
}

// ---- 256 x 128 tile, eight symmetric waves ------------------------------------------------
// The 128 x 128 kernel above is bound by the operand lines it pulls from L2 (32 KiB per slice
// of 1.05 MFLOP at 38-51 GB/s per CU).  A 256 x 128 tile needs 48 KiB for twice the flops.  Its
// eight 64 x 64 wave tiles leave no room for loader waves at 256 VGPRs (two waves per SIMD), so
// every wave does both jobs in one instruction stream: loads of slice g + 2 are issued into
// registers before the MFMAs of slice g, slice g + 1 is written to LDS after them, one barrier
// per slice.  Fragments are single-buffered (accumulators 128 + fragments 32 + two load stages
// 48 VGPRs); the LDS latency of one wave is covered by the MFMAs of the other wave of its SIMD.
// Images must be padded to whole tiles (256 / 128 rows; pad rows may hold anything - their
// products are never stored), so a thread's six line pointers are one base plus constants.
constexpr int W8_A = 256, W8_B = 128, W8_IMG = (W8_A + W8_B) * ROW_B;  // 48 KiB per buffer

template <bool B_LO, int EPI>  // EPI: 0 stores, 1 pruned stores (top-k passes), 2 counts (ranks; nothing is stored)
__global__ __launch_bounds__(512) void k_gemm_split_w8(const char* __restrict__ A, const char* __restrict__ B,
                                                       int64_t M, int64_t N, int n_slice,
                                                       float* __restrict__ C, int64_t ldc, int tiles_x,
                                                       int n_tiles, int ksplit, int64_t part_stride,
                                                       const int32_t* __restrict__ range_flag, const float* __restrict__ thr,
                                                        uint8_t* __restrict__ pflags, int64_t ldf, CountArgs cnt) {
    extern __shared__ __attribute__((aligned(16))) char lds8[];  // [2][A rows 0-255 | B rows 0-127]
    if (range_flag && *range_flag) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int slot = blockIdx.x, slots = gridDim.x;
    const int my_tiles = (n_tiles - slot + slots - 1) / slots;
    const int total = my_tiles * n_slice;
    const int64_t pitch = static_cast<int64_t>(n_slice) * ksplit * ROW_B;

    // loader role: piece (t & 7) of the lines of image rows (t >> 3) + 64 p, p = 0..3 (A) and 0..1 (B)
    const int lrow = t >> 3, piece = t & 7;
    const char *pa = A, *pb = B;
    int gl = 0, sl = 0, tl = slot;  // load cursor, two slices ahead; stops on the last slice
    auto issue = [&](u32x4 (&v)[6]) {
        if (sl == 0) {
            const int ot = tl / ksplit;
            const int64_t k_off = static_cast<int64_t>(tl % ksplit) * n_slice * ROW_B + piece * 16;
            pa = A + (static_cast<int64_t>(ot / tiles_x) * W8_A + lrow) * pitch + k_off;
            pb = B + (static_cast<int64_t>(ot % tiles_x) * W8_B + lrow) * pitch + k_off;
        }
        const int64_t so = static_cast<int64_t>(sl) * ROW_B;
#pragma unroll
        for (int p = 0; p < 4; ++p) v[p] = *reinterpret_cast<const u32x4*>(pa + p * 64 * pitch + so);
#pragma unroll
        for (int p = 0; p < 2; ++p) v[4 + p] = *reinterpret_cast<const u32x4*>(pb + p * 64 * pitch + so);
        if (++gl < total && ++sl == n_slice) {
            sl = 0;
            tl += slots;
        }
    };
    auto put = [&](char* img, const u32x4 (&v)[6]) {
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(img + slot_off(p * 64 + lrow, piece)) = v[p];
#pragma unroll
        for (int p = 0; p < 2; ++p)
            *reinterpret_cast<u32x4*>(img + W8_A * ROW_B + slot_off(p * 64 + lrow, piece)) = v[4 + p];
    };

    // MFMA role: wave tile 64 x 64 at (wm, wn) of the 256 x 128 tile
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int l31 = lane & 31, lk = lane >> 5;
    int off_a[2][2], off_b[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            off_a[ks][part] = slot_off(wm + l31, ks * 2 + lk + 4 * part);
            off_b[ks][part] = W8_A * ROW_B + slot_off(wn + l31, ks * 2 + lk + 4 * part);
        }
    f32x16 accm[2][2], accc[2][2];
    auto zero = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    accm[i][j][r] = 0.f;
                    accc[i][j][r] = 0.f;
                }
    };
    zero();
    auto kstep = [&](const char* img, int ks) {
        h8 fa[2][2], fb[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa[i][0] = *reinterpret_cast<const h8*>(img + off_a[ks][0] + i * 32 * ROW_B);
            fa[i][1] = *reinterpret_cast<const h8*>(img + off_a[ks][1] + i * 32 * ROW_B);
            fb[i][0] = *reinterpret_cast<const h8*>(img + off_b[ks][0] + i * 32 * ROW_B);
            if (B_LO) fb[i][1] = *reinterpret_cast<const h8*>(img + off_b[ks][1] + i * 32 * ROW_B);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) accm[i][j] = mma16(fa[i][0], fb[j][0], accm[i][j]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) accc[i][j] = mma16(fa[i][1], fb[j][0], accc[i][j]);
        if (B_LO) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) accc[i][j] = mma16(fa[i][0], fb[j][1], accc[i][j]);
        }
    };

    u32x4 r0[6], r1[6];  // slices g (even) / g (odd) on their way to LDS
    issue(r0);
    issue(r1);
    put(lds8, r0);
    __syncthreads();
    int s = 0, tile = slot;
    // two slices per trip so that the register stages have fixed names
    auto slice = [&](int g, u32x4 (&mine)[6], u32x4 (&next)[6]) {
        issue(mine);  // slice g + 2 into the stage slice g has left
        const char* img = lds8 + (g & 1) * W8_IMG;
        kstep(img, 0);
        kstep(img, 1);
        put(lds8 + ((g + 1) & 1) * W8_IMG, next);  // slice g + 1 (after the last slice: a surplus copy)
        __syncthreads();
        if (++s == n_slice) {  // tile done: main + corr / 2048 -> C
            const int ot = tile / ksplit;
            const int64_t m0 = static_cast<int64_t>(ot / tiles_x) * W8_A + wm;
            const int64_t n0 = static_cast<int64_t>(ot % tiles_x) * W8_B + wn;
            float* c0 = C + (tile % ksplit) * part_stride + (m0 + 4 * lk) * ldc + n0 + l31;
            if constexpr (EPI == 2) {
                // counting epilogue (ranks): nothing is stored.  Lane l of the wave keeps the two counts of row
                // m0 + l: a row's 64 scores sit in the 32 lanes of one lk, so one ballot per comparison holds two
                // rows' verdicts (low half: row rr, high half: rr + 4), counted with s_bcnt1 and kept by the
                // row's lane; one pair of atomics per row and tile at the end
                const int l64 = threadIdx.x & 63;
                const bool rok = m0 + l64 < M;
                const int tv = __builtin_bit_cast(int, rok ? thr[m0 + l64] : INFINITY);
                // the row's excluded column, relative to this wave's first one (anything outside 0 .. 63: none here)
                int64_t exl = rok ? static_cast<int64_t>(cnt.excl[m0 + l64]) - (cnt.col0 + n0) : -1;
                const int ex = (exl >= 0 && exl < 64) ? static_cast<int>(exl) : -1;
                int cgt = 0, ceq = 0;
                const bool ok0 = n0 + l31 < N, ok1 = n0 + l31 + 32 < N;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = i * 32 + (r & 3) + 8 * (r >> 2);  // rows rr (lk = 0) and rr + 4 (lk = 1)
                        const float th0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tv, rr));
                        const float th1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tv, rr + 4));
                        const int e0 = __builtin_amdgcn_readlane(ex, rr), e1 = __builtin_amdgcn_readlane(ex, rr + 4);
                        const float th = lk ? th1 : th0;
                        const int e = lk ? e1 : e0;
                        const float v0 = count_value(accm[i][0][r] + accc[i][0][r] * (1.f / 2048.f), cnt.round16);
                        const float v1 = count_value(accm[i][1][r] + accc[i][1][r] * (1.f / 2048.f), cnt.round16);
                        const bool in0 = ok0 && l31 != e, in1 = ok1 && l31 + 32 != e;
                        const unsigned long long g0 = __ballot(in0 && v0 > th), g1 = __ballot(in1 && v1 > th);
                        const unsigned long long q0 = __ballot(in0 && v0 == th), q1 = __ballot(in1 && v1 == th);
                        const int gt_lo = __builtin_popcount(static_cast<unsigned>(g0)) + __builtin_popcount(static_cast<unsigned>(g1));
                        const int gt_hi = __builtin_popcount(static_cast<unsigned>(g0 >> 32)) + __builtin_popcount(static_cast<unsigned>(g1 >> 32));
                        const int eq_lo = __builtin_popcount(static_cast<unsigned>(q0)) + __builtin_popcount(static_cast<unsigned>(q1));
                        const int eq_hi = __builtin_popcount(static_cast<unsigned>(q0 >> 32)) + __builtin_popcount(static_cast<unsigned>(q1 >> 32));
                        if (l64 == rr) cgt = gt_lo, ceq = eq_lo;  // (every row of the wave is met exactly once)
                        if (l64 == rr + 4) cgt = gt_hi, ceq = eq_hi;
                    }
                if (rok && n0 < N) {
                    if (cgt) atomicAdd(cnt.counts + 2 * (m0 + l64), cgt);
                    if (ceq) atomicAdd(cnt.counts + 2 * (m0 + l64) + 1, ceq);
                }
            } else if constexpr (EPI == 1) {
                // pruned stores (top-k passes): the 32 lanes with this lk hold a row's 64 columns - the row's block
                // is written only when one of them is above the row's threshold, and flagged.  The thresholds of
                // the wave's 64 rows come in with ONE load (lane l: row m0 + l) and are read out per row with
                // v_readlane; the rows' verdicts are collected in a scalar mask and leave as one byte store per lane
                const int l64 = threadIdx.x & 63;
                const int tv = __builtin_bit_cast(int, (m0 + l64 < M) ? thr[m0 + l64] : INFINITY);
                unsigned long long rows_hit = 0;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = i * 32 + (r & 3) + 8 * (r >> 2);  // rows rr (lk = 0) and rr + 4 (lk = 1)
                        const float th0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tv, rr));
                        const float th1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tv, rr + 4));
                        const float th = lk ? th1 : th0;
                        const bool ok0 = n0 + l31 < N, ok1 = n0 + l31 + 32 < N;
                        const float v0 = accm[i][0][r] + accc[i][0][r] * (1.f / 2048.f);
                        const float v1 = accm[i][1][r] + accc[i][1][r] * (1.f / 2048.f);
                        const unsigned long long hits = __ballot((ok0 && v0 > th) || (ok1 && v1 > th));
                        const bool any0 = (hits & 0xffffffffull) != 0, any1 = (hits >> 32) != 0;
                        rows_hit |= (any0 ? 1ull << rr : 0ull) | (any1 ? 1ull << (rr + 4) : 0ull);
                        if (lk ? any1 : any0) {  // (rows past M have threshold +inf: never stored)
                            if (ok0) c0[rr * ldc] = v0;
                            if (ok1) c0[rr * ldc + 32] = v1;
                        }
                    }
                if (m0 + l64 < M && n0 < N) pflags[(m0 + l64) * ldf + n0 / 64] = (rows_hit >> l64) & 1ull;
            } else if (m0 + 64 <= M && n0 + 64 <= N) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            c0[(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc + j * 32] =
                                accm[i][j][r] + accc[i][j][r] * (1.f / 2048.f);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rr = i * 32 + (r & 3) + 8 * (r >> 2);
                            if (m0 + 4 * lk + rr < M && n0 + l31 + j * 32 < N)
                                c0[rr * ldc + j * 32] = accm[i][j][r] + accc[i][j][r] * (1.f / 2048.f);
                        }
            }
            zero();
            s = 0;
            tile += slots;
        }
    };
    int g = 0;
    for (; g + 2 <= total; g += 2) {
        slice(g, r0, r1);
        slice(g + 1, r1, r0);
    }
    if (g < total) slice(g, r0, r1);
}

// (the switch that keeps the exact fp32 MFMA kernels is a descriptor flag, BESS_FLAG_FP32_MATH: the callers in
// neg_shared.hip do not ask for this path then; the library reads no environment variables for dispatch)
static bool split_disabled() { return false; }

static int n_compute_units() {
    static const int n = [] {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) return 256;
        return cu;
    }();
    return n;
}

// images are allocated with their rows rounded up to whole tiles of the 256 x 128 kernel
static int64_t pad_a(int64_t rows) { return ceil_div(rows, W8_A) * W8_A; }
static int64_t pad_b(int64_t rows) { return ceil_div(rows, W8_B) * W8_B; }

// C[M, N] (+ k parts) = A . B^T from split images; picks the 256 x 128 kernel when the images are padded
// to whole tiles (forward) and its tiles fill the chip: measured 3-8 % faster there, no more - the two
// kernels meet the same wall (profiles/ubench/mfma_f16.hip: the clock the chip holds under MFMA + LDS load)
static int launch_product(const char* A, const char* B, int64_t M, int64_t N, int n_slice, float* C, int64_t ldc,
                          int ksplit, int64_t part_stride, bool b_lo, bool padded, const int32_t* flag,
                          hipStream_t st, const float* thr = nullptr, uint8_t* pflags = nullptr, int64_t ldf = 0,
                          const CountArgs* count = nullptr, bool diag = false) {
    BESS_REQUIRE(!thr || (ksplit == 1 && (pflags || count)),
                 "gemm_split: pruned stores / counts need an unsplit product and a flag or count array");
    const CountArgs cnt = count ? *count : CountArgs{nullptr, nullptr, 0, 0};
    const int epi = count ? 2 : (thr ? 1 : 0);
    const int cus = n_compute_units();
    const int64_t tx = ceil_div(N, 128);
    const int64_t t8 = tx * ceil_div(M, W8_A) * ksplit, t4 = tx * ceil_div(M, 128) * ksplit;
    BESS_REQUIRE(t4 < (1ll << 31), "gemm_split: too many tiles");
    if (diag) {  // the diagonal 128 x 128 tiles only (M == N), C is [M, 128]
        BESS_REQUIRE(M == N && ksplit == 1 && !thr && !count && ldc == 128, "gemm_split: bad diagonal product");
        const int64_t td = ceil_div(M, 128);
        const int grid = static_cast<int>(td < cus ? td : cus);
        if (b_lo)
            k_gemm_split_f16<true, 0><<<grid, 512, 0, st>>>(A, B, M, N, n_slice, C, ldc, 0, static_cast<int>(td), 1, 0, flag,
                                                            nullptr, nullptr, 0, cnt);
        else
            k_gemm_split_f16<false, 0><<<grid, 512, 0, st>>>(A, B, M, N, n_slice, C, ldc, 0, static_cast<int>(td), 1, 0, flag,
                                                             nullptr, nullptr, 0, cnt);
        return check_launch("gemm_split_f16 (diagonal tiles)");
    }
    const bool wide = padded && t8 >= cus;
    if (wide) {
        static const bool attr = [] {
            const int bytes = 2 * W8_IMG;
            bool ok = true;
            for (const void* f : {reinterpret_cast<const void*>(&k_gemm_split_w8<true, 0>),
                                  reinterpret_cast<const void*>(&k_gemm_split_w8<false, 0>),
                                  reinterpret_cast<const void*>(&k_gemm_split_w8<true, 1>),
                                  reinterpret_cast<const void*>(&k_gemm_split_w8<false, 1>),
                                  reinterpret_cast<const void*>(&k_gemm_split_w8<true, 2>),
                                  reinterpret_cast<const void*>(&k_gemm_split_w8<false, 2>)})
                ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
            return ok;
        }();
        BESS_REQUIRE(attr, "gemm_split: cannot reserve %d bytes of LDS", 2 * W8_IMG);
        const int grid = static_cast<int>(t8 < cus ? t8 : cus);
#define BESS_W8(LO, PR)                                                                                          \
    k_gemm_split_w8<LO, PR><<<grid, 512, 2 * W8_IMG, st>>>(A, B, M, N, n_slice, C, ldc, static_cast<int>(tx),     \
                                                           static_cast<int>(t8), ksplit, part_stride, flag, thr, \
                                                           pflags, ldf, cnt)
        if (epi == 2) {
            if (b_lo) BESS_W8(true, 2);
            else BESS_W8(false, 2);
        } else if (epi == 1) {
            if (b_lo) BESS_W8(true, 1);
            else BESS_W8(false, 1);
        } else {
            if (b_lo) BESS_W8(true, 0);
            else BESS_W8(false, 0);
        }
#undef BESS_W8
        return check_launch("gemm_split_w8");
    }
    const int grid = static_cast<int>(t4 < cus ? t4 : cus);
#define BESS_T4(LO, PR)                                                                                    \
    k_gemm_split_f16<LO, PR><<<grid, 512, 0, st>>>(A, B, M, N, n_slice, C, ldc, static_cast<int>(tx),        \
                                                   static_cast<int>(t4), ksplit, part_stride, flag, thr, pflags, ldf, \
                                                   cnt)
    if (epi == 2) {
        if (b_lo) BESS_T4(true, 2);
        else BESS_T4(false, 2);
    } else if (epi == 1) {
        if (b_lo) BESS_T4(true, 1);
        else BESS_T4(false, 1);
    } else {
        if (b_lo) BESS_T4(true, 0);
        else BESS_T4(false, 0);
    }
#undef BESS_T4
    return check_launch("gemm_split_f16");
}

static int64_t split_pitch(int W) { return ceil_div(W, SK) * ROW_B; }
constexpr int64_t FLAG_BYTES = 256;  // tail of every workspace: the range flag (one int32, zeroed per call)
constexpr int64_t SPLIT_CHUNK = 65536;  // candidate rows split per pass at most (128 MiB of lines at W = 512)

// Workspace the split path wants for this shape; 0 = the shape is left to the fp32 kernels
// (too few 128-tiles to occupy the chip, or the path is switched off).
int64_t gemm_split_workspace(int64_t S, int64_t N, int W) {
    if (split_disabled() || S < 1 || N < 1 || W < 1) return 0;
    if (ceil_div(N, 128) * ceil_div(S, 128) < 256) return 0;
    return (pad_a(S) + pad_b(N < SPLIT_CHUNK ? N : SPLIT_CHUNK)) * split_pitch(W) + FLAG_BYTES;
}

// the same for a caller that takes the split path whatever the shape (pair scores in the arithmetic of a larger product)
int64_t gemm_split_workspace_any(int64_t S, int64_t N, int W) {
    return (pad_a(S) + pad_b(N < SPLIT_CHUNK ? N : SPLIT_CHUNK)) * split_pitch(W) + FLAG_BYTES;
}

// splits `a` (f32 rows; skipped when a.rows == 0) and `b` (table dtype) in one launch
static int split_rows(const SplitSrc& a, char* dst_a, int dtype_b, const SplitSrc& b, char* dst_b, int W,
                      hipStream_t st) {
    const int n_blk = static_cast<int>(ceil_div(W, SK));
    const int64_t blocks_a = ceil_div(a.rows * n_blk * 4, 256), blocks_b = ceil_div(b.rows * n_blk * 4, 256);
    BESS_REQUIRE(blocks_a + blocks_b < (1ll << 31), "gemm_split: too many rows");
    const unsigned grid = static_cast<unsigned>(blocks_a + blocks_b);
    const auto al = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    const bool vec = W % 8 == 0 && a.ld % 8 == 0 && b.ld % 8 == 0 && al(a.base) && al(b.base);
    const int ba = static_cast<int>(blocks_a);
    if (dtype_b == BESS_F32) {
        if (vec) k_split_rows<float, true><<<grid, 256, 0, st>>>(a, dst_a, ba, b, dst_b, W, n_blk);
        else k_split_rows<float, false><<<grid, 256, 0, st>>>(a, dst_a, ba, b, dst_b, W, n_blk);
    } else {
        if (vec) k_split_rows<half_t, true><<<grid, 256, 0, st>>>(a, dst_a, ba, b, dst_b, W, n_blk);
        else k_split_rows<half_t, false><<<grid, 256, 0, st>>>(a, dst_a, ba, b, dst_b, W, n_blk);
    }
    return check_launch("split_rows");
}

__global__ void k_poison_counts(const int32_t* __restrict__ flag, int32_t* __restrict__ counts, int64_t n) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (*flag && i < n) counts[i] = INT32_MIN;
}

// out[q, j] = Q[q] . E[idx[j]] through the workspace (>= gemm_split_workspace bytes, 16-B aligned)
int gemm_split_fwd(int dtype, const float* Q, int64_t S, const void* E, const int32_t* idx, int64_t N, int W,
                   float* out, int64_t ld, void* ws, int64_t ws_bytes, hipStream_t st, const float* thr,
                   uint8_t* pflags, int64_t ldf, const CountArgs* count, bool diag) {
    const int64_t pitch = split_pitch(W);
    BESS_REQUIRE(!diag || (S == N && idx && !thr && !count && ld == 128), "gemm_split: bad diagonal product");
    BESS_REQUIRE(reinterpret_cast<uintptr_t>(ws) % 16 == 0, "gemm_split: workspace must be 16-B aligned");
    BESS_REQUIRE(ws_bytes >= FLAG_BYTES + 256 * pitch, "gemm_split: workspace too small");
    ws_bytes -= FLAG_BYTES;
    int32_t* flag = reinterpret_cast<int32_t*>(static_cast<char*>(ws) + ws_bytes / 16 * 16);
    {
        hipError_t e = fill_words_async(flag, 0u, 1, st);
        if (e != hipSuccess) return fail(static_cast<int>(e), "memset: %s", hipGetErrorString(e));
    }
    int64_t chunk = (ws_bytes / pitch - pad_a(S)) / 128 * 128;  // candidate rows per pass
    BESS_REQUIRE(chunk >= 128, "gemm_split: workspace too small");
    if (chunk >= N) chunk = N;
    char* qa = static_cast<char*>(ws);
    char* eb = qa + pad_a(S) * pitch;
    const int n_slice = static_cast<int>(ceil_div(W, SK));
    for (int64_t j0 = 0; j0 < N; j0 += chunk) {
        const int64_t nc = N - j0 < chunk ? N - j0 : chunk;
        // rows j0 .. j0 + nc: through the index if there is one, else consecutive rows of the table
        const int64_t sz = dtype == BESS_F32 ? 4 : 2;
        SplitSrc src{idx ? E : static_cast<const char*>(E) + j0 * W * sz, idx ? idx + j0 : nullptr, nc, W, flag};
        // the query rows ride along with the first chunk
        if (int e = split_rows(SplitSrc{Q, nullptr, j0 == 0 ? S : 0, W, flag}, qa, dtype, src, eb, W, st)) return e;
        // (pruned stores: the chunk's flags start at block j0 / 64 - chunks are multiples of 128 rows)
        CountArgs cj{nullptr, nullptr, 0, 0};
        if (count) cj = CountArgs{count->excl, count->counts, count->col0 + j0, count->round16};
        BESS_REQUIRE(!diag || nc == N, "gemm_split: the diagonal product wants its operands in one pass");
        if (int e = launch_product(qa, eb, S, nc, n_slice, count ? nullptr : out + (diag ? 0 : j0), ld, 1, 0,
                                   dtype == BESS_F32, true, flag, st, thr, pflags ? pflags + j0 / 64 : nullptr, ldf,
                                   count ? &cj : nullptr, diag))
            return e;
    }
    if (count) {
        // (no score matrix for the fp32 kernels to fill) an operand outside the fp16 range: every count becomes
        // negative - the caller sees it where it reads the counts and scores that batch through the matrix
        k_poison_counts<<<static_cast<unsigned>(ceil_div(2 * S, 256)), 256, 0, st>>>(flag, count->counts, 2 * S);
        return check_launch("poison_counts");
    }
    // an operand outside the fp16 range (or not finite) raised the flag: the split kernels returned at once and
    // the exact fp32 kernels compute the whole product now; else they are the ones that return at once
    // (diagonal form: no fp32 fallback - an operand outside the fp16 range leaves `out` unwritten; the counting pass
    // over the same rows meets the same operand and poisons its counts, which is what the caller acts on)
    if (diag) return BESS_OK;
    return gemm_dot_fwd(dtype, Q, S, E, idx, N, W, out, ld, st, flag);
}

// ---- backward products -----------------------------------------------------------------
//   dQ[q, w] = sum_j G[q, j] E[idx[j], w]     A = G (plain image),  B = E^T   k = j
//   dE[j, w] = sum_q G[q, j] Q[q, w]          A = G^T,              B = Q^T   k = q
// One pass over G writes both of its images.
struct BwdPlan {
    int ks;          // k split of a product with `tiles` output tiles and `k` reduction length
    int nblk;        // k blocks per image row, a multiple of ks
};
static BwdPlan plan_k(int64_t tiles, int64_t k) {
    const int blk = static_cast<int>(ceil_div(k, SK));
    int ks = 1;
    while (ks < 8 && tiles * ks < 192 && blk / (ks * 2) >= 8) ks *= 2;
    return BwdPlan{ks, static_cast<int>(ceil_div(blk, ks)) * ks};
}

struct BwdLayout {
    BwdPlan pq, pe;  // plans of the dQ and dE products
    int64_t g_plain, g_trans, e_trans, q_trans, parts, total;
};
static bool bwd_layout(int64_t S, int64_t N, int W, BwdLayout& L) {
    if (split_disabled() || S < 128 || N < 128 || W < 64 || W % 4) return false;
    const int64_t tw = ceil_div(W, 128);
    L.pq = plan_k(ceil_div(S, 128) * tw, N);
    L.pe = plan_k(ceil_div(N, 128) * tw, S);
    // worth it only when both products can occupy at least half of the chip
    if (ceil_div(S, 128) * tw * L.pq.ks < 128 || ceil_div(N, 128) * tw * L.pe.ks < 128) return false;
    if (L.pq.nblk > 65535 || L.pe.nblk > 65535) return false;  // k blocks ride on gridDim.y of the pre-pass
    L.g_plain = S * L.pq.nblk * ROW_B;
    L.g_trans = N * L.pe.nblk * ROW_B;
    L.e_trans = static_cast<int64_t>(W) * L.pq.nblk * ROW_B;
    L.q_trans = static_cast<int64_t>(W) * L.pe.nblk * ROW_B;
    const int64_t p1 = L.pq.ks > 1 ? L.pq.ks * S * W * 4 : 0, p2 = L.pe.ks > 1 ? L.pe.ks * N * W * 4 : 0;
    L.parts = p1 > p2 ? p1 : p2;
    L.total = L.g_plain + L.g_trans + L.e_trans + L.q_trans + L.parts + FLAG_BYTES;
    return true;
}

int64_t gemm_split_bwd_workspace(int64_t S, int64_t N, int W) {
    BwdLayout L;
    return bwd_layout(S, N, W, L) ? L.total : 0;
}

static int product(const char* A, const char* B, int64_t M, int64_t N, const BwdPlan& p, float* C, int64_t ldc,
                   float* parts, const int32_t* flag, hipStream_t st) {
    float* dst = p.ks > 1 ? parts : C;
    if (int e = launch_product(A, B, M, N, p.nblk / p.ks, dst, ldc, p.ks, M * ldc, true, false, flag, st)) return e;
    if (p.ks > 1) {
        const int64_t n4 = M * ldc / 4;
        k_sum_parts<<<static_cast<unsigned>(ceil_div(n4, 256)), 256, 0, st>>>(parts, M * ldc, p.ks, C, n4);
        return check_launch("sum_parts");
    }
    return BESS_OK;
}

// d_query [S, W] and d_neg [N, W] (both dense, leading dimension W) through the workspace
int gemm_split_bwd(int dtype, const float* G, int64_t ldg, int64_t S, const float* Q, const void* E,
                   const int32_t* idx, int64_t N, int W, float* dQ, float* dE, void* ws, int64_t ws_bytes,
                   hipStream_t st) {
    BwdLayout L;
    BESS_REQUIRE(bwd_layout(S, N, W, L) && ws_bytes >= L.total, "gemm_split_bwd: workspace too small");
    BESS_REQUIRE(reinterpret_cast<uintptr_t>(ws) % 16 == 0, "gemm_split: workspace must be 16-B aligned");
    char* gp = static_cast<char*>(ws);
    char* gt = gp + L.g_plain;
    char* et = gt + L.g_trans;
    char* qt = et + L.e_trans;
    float* parts = reinterpret_cast<float*>(qt + L.q_trans);
    int32_t* flag = reinterpret_cast<int32_t*>(static_cast<char*>(ws) + L.total - FLAG_BYTES);
    {
        hipError_t e = fill_words_async(flag, 0u, 1, st);
        if (e != hipSuccess) return fail(static_cast<int>(e), "memset: %s", hipGetErrorString(e));
    }
    const auto al = [](const void* p, int64_t ld, int64_t sz) {
        return reinterpret_cast<uintptr_t>(p) % 16 == 0 && ld * sz % 16 == 0;
    };
    // G [S, N]: plain image with k = j (nblk of dQ), transposed image with k = q (nblk of dE)
    const dim3 gg(L.pq.nblk, L.pe.nblk);
    const SplitSrc sg{G, nullptr, S, ldg, flag};
    if (N % 4 == 0 && al(G, ldg, 4))
        k_split_tile32<float, true, true, true><<<gg, 256, 0, st>>>(sg, N, gp, L.pq.nblk, gt, L.pe.nblk);
    else
        k_split_tile32<float, true, true, false><<<gg, 256, 0, st>>>(sg, N, gp, L.pq.nblk, gt, L.pe.nblk);
    // E rows [N (by index), W] -> E^T: image rows w, k = j       (W % 4 == 0 checked by bwd_layout)
    const dim3 ge(static_cast<unsigned>(ceil_div(W, 32)), L.pq.nblk);
    const SplitSrc se{E, idx, N, W, flag};
    if (dtype == BESS_F32) {
        if (al(E, W, 4)) k_split_tile32<float, false, true, true><<<ge, 256, 0, st>>>(se, W, nullptr, 0, et, L.pq.nblk);
        else k_split_tile32<float, false, true, false><<<ge, 256, 0, st>>>(se, W, nullptr, 0, et, L.pq.nblk);
    } else {
        if (al(E, W, 2) && W % 8 == 0)
            k_split_tile32<half_t, false, true, true><<<ge, 256, 0, st>>>(se, W, nullptr, 0, et, L.pq.nblk);
        else
            k_split_tile32<half_t, false, true, false><<<ge, 256, 0, st>>>(se, W, nullptr, 0, et, L.pq.nblk);
    }
    // Q [S, W] -> Q^T: image rows w, k = q
    const dim3 gq(static_cast<unsigned>(ceil_div(W, 32)), L.pe.nblk);
    const SplitSrc sq{Q, nullptr, S, W, flag};
    if (al(Q, W, 4)) k_split_tile32<float, false, true, true><<<gq, 256, 0, st>>>(sq, W, nullptr, 0, qt, L.pe.nblk);
    else k_split_tile32<float, false, true, false><<<gq, 256, 0, st>>>(sq, W, nullptr, 0, qt, L.pe.nblk);
    if (int e = check_launch("split_tile32")) return e;
    if (int e = product(gp, et, S, W, L.pq, dQ, W, parts, flag, st)) return e;
    if (int e = product(gt, qt, N, W, L.pe, dE, W, parts, flag, st)) return e;
    // (k_sum_parts has then summed garbage into dQ / dE: overwritten by the fp32 kernels, which run only if the flag is up)
    if (int e = gemm_dot_dq(dtype, G, ldg, S, E, idx, N, W, dQ, st, flag)) return e;
    return gemm_dot_de(G, ldg, S, Q, N, W, dE, st, flag);
}

}  // namespace bess

// Shared device/host helpers of the gfx950 BESS kernels.
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/besskge_hip.h"

namespace bess {

// ---- host side: error reporting (thread local text + return codes) --------
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

#define BESS_REQUIRE(cond, ...)                          \
    do {                                                 \
        if (!(cond)) return ::bess::fail(BESS_EINVAL, __VA_ARGS__); \
    } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// scorer properties
inline bool is_complex_entity(int scorer) { return scorer == BESS_ROTATE || scorer == BESS_COMPLEX; }
inline bool is_distance(int scorer) {
    return scorer == BESS_TRANSE || scorer == BESS_ROTATE || scorer == BESS_AFFINE || scorer == BESS_BOXE;
}
int check_desc(const bess_model_desc* d);

// reductions used by the negative-scoring kernels
enum Reduce : int { RED_DOT = 0, RED_L1 = 1, RED_L2 = 2 };  // RED_L2: every p != 1 (p is a run-time argument)
inline int reduce_of(const bess_model_desc* d) {
    if (!is_distance(d->scorer)) return RED_DOT;
    return d->norm_p == 1 ? RED_L1 : RED_L2;
}

// fp32 MFMA GEMMs of the bilinear scorers (gemm_mfma.hip)
// (run_if: optional device word; the kernels return at once while it is 0 - the conditional fallback of the split path)
int gemm_dot_fwd(int dtype, const float* Q, int64_t S, const void* E, const int32_t* idx, int64_t N, int W,
                 float* out, int64_t ld, hipStream_t st, const int32_t* run_if = nullptr);
// Counting epilogue of the all-entity scoring kernels (ranks without the score matrix: bess_neg_score_shared_fwd_counts):
// counts[row, 0] += #{columns: score > thr[row]}, counts[row, 1] += #{score == thr[row]}; column excl[row] (an index
// into the whole candidate list, -1: none) is left out; col0 = list index of the launch's first column.
// round16: compare the scores as rounded to fp16 (what a `model.half()` reference hands to its ranking).
struct CountArgs {
    const int32_t* excl;
    int32_t* counts;
    int64_t col0;
    int round16;
};
__device__ __forceinline__ float count_value(float v, int round16) {
    return round16 ? static_cast<float>(static_cast<_Float16>(v)) : v;
}
// split-fp16 MFMA variant of gemm_dot_fwd (gemm_split.hip): workspace it wants for a shape
// (0 = leave the shape to the fp32 kernels) and the product through that workspace
int64_t gemm_split_workspace(int64_t S, int64_t N, int W);
int64_t gemm_split_workspace_any(int64_t S, int64_t N, int W);  // (bytes for a shape, eligible or not)
// (thr / flags: pruned stores for the top-k passes - a row's 64-column block is written only when one of its
// scores is above thr[row]; flags[row, ld_flags] (one byte per block) says which were)
int gemm_split_fwd(int dtype, const float* Q, int64_t S, const void* E, const int32_t* idx, int64_t N, int W,
                   float* out, int64_t ld, void* ws, int64_t ws_bytes, hipStream_t st, const float* thr = nullptr,
                   uint8_t* flags = nullptr, int64_t ld_flags = 0, const CountArgs* count = nullptr,
                   bool diag = false);  // diag: S == N, out [S, 128] receives the diagonal 128 x 128 tiles only
int64_t gemm_split_bwd_workspace(int64_t S, int64_t N, int W);
int gemm_split_bwd(int dtype, const float* G, int64_t ldg, int64_t S, const float* Q, const void* E,
                   const int32_t* idx, int64_t N, int W, float* dQ, float* dE, void* ws, int64_t ws_bytes,
                   hipStream_t st);
int gemm_dot_dq(int dtype, const float* G, int64_t ldg, int64_t S, const void* E, const int32_t* idx, int64_t N,
                int W, float* dQ, hipStream_t st, const int32_t* run_if = nullptr);
int gemm_dot_de(const float* G, int64_t ldg, int64_t S, const float* Q, int64_t N, int W, float* dE,
                hipStream_t st, const int32_t* run_if = nullptr);

// packed-fp16 L1 distance matrix and its backward products (l1_f16.hip): TransE / RotatE, p = 1, f16 tables
bool l1_pk_eligible(const bess_model_desc* d);
int l1_pk_fwd(const bess_model_desc* d, const float* query, int64_t n_query, const void* neg_base,
              const int32_t* neg_idx, int64_t n_neg, float* out, int64_t ld_out, const bess_kill_desc* kill,
              hipStream_t st, const float* thr = nullptr, uint8_t* flags = nullptr, int64_t ld_flags = 0,
              const CountArgs* count = nullptr, bool diag = false);  // diag: out [n, 64] = the diagonal 64 x 64 tiles

// affine-in-the-candidate distance scorers (affine.hip)
int affine_pertriple(const bess_model_desc* d, bool fwd, const float* query, int64_t n_query, const void* neg_base,
                     const int32_t* neg_idx, int64_t n_neg, float* out, const float* d_out, int64_t ld, float* dq,
                     float* dn, hipStream_t st);
int affine_query_fwd(const bess_model_desc* d, int32_t side, const void* ent_base, const int32_t* ent_idx,
                     const void* rel_table, const int32_t* rel_idx, int64_t n, float* query, hipStream_t st);
int affine_query_bwd(const bess_model_desc* d, int32_t side, const void* ent_base, const int32_t* ent_idx,
                     const void* rel_table, const int32_t* rel_idx, int64_t n, const float* d_query, float* d_ent,
                     float* d_rel, hipStream_t st);
int boxe_grad_segments(const bess_model_desc* d, const float* query, void* table, int64_t n_neg, const float* d_out,
                       int64_t ld_dout, const int32_t* refs_sorted, const int32_t* seg_rows,
                       const int32_t* seg_offsets, const int32_t* n_seg, int64_t max_seg, float* grad_seg,
                       float fused_sgd_lr, const int32_t* long_segs, int64_t long_cap, float* long_grad,
                       int32_t* long_count, hipStream_t st);
int affine_grad_segments(const bess_model_desc* d, const float* query, void* table, int64_t n_neg,
                         const float* d_out, int64_t ld_dout, const int32_t* refs_sorted, const int32_t* seg_rows,
                         const int32_t* seg_offsets, const int32_t* n_seg, int64_t max_seg, float* grad_seg,
                         float fused_sgd_lr, const int32_t* long_segs, int64_t long_cap, float* long_grad,
                         int32_t* long_count, hipStream_t st);
int affine_shared_fwd(const bess_model_desc* d, const float* query, int64_t S, const float* cand, int64_t N, float* out,
                      int64_t ld, hipStream_t st);
int affine_shared_bwd(const bess_model_desc* d, const float* query, int64_t S, const float* cand, int64_t N,
                      const float* out, int64_t ld_out, const float* d_out, int64_t ld_dout, float* d_query,
                      float* d_cand, hipStream_t st);

// BoxE (boxe.hip): one kernel family for per-triple and shared negatives
int boxe_negatives(const bess_model_desc* d, bool fwd, bool shared, const float* query, int64_t n_query,
                   const void* neg_base, const int32_t* neg_idx, int64_t n_neg, float* out, const float* d_out,
                   int64_t ld, float* dq, float* dn, hipStream_t st);

// ---- device side ------------------------------------------------------------
typedef _Float16 half_t;

template <typename T>
__device__ __forceinline__ float to_f32(T v) {
    return static_cast<float>(v);
}

// VEC contiguous elements of T -> float registers, one vector load
template <typename T, int VEC>
struct VecLoad {
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    __device__ __forceinline__ static void load(const T* __restrict__ p, float (&out)[VEC]) {
        vec_t v = *reinterpret_cast<const vec_t*>(p);
#pragma unroll
        for (int i = 0; i < VEC; ++i) out[i] = static_cast<float>(v[i]);
    }
};
template <typename T>
struct VecLoad<T, 1> {
    __device__ __forceinline__ static void load(const T* __restrict__ p, float (&out)[1]) {
        out[0] = static_cast<float>(*p);
    }
};

// Chunk c (VEC scalars) of a row of nch chunks, zeros past the end - without a branch: the load
// is unconditional at a clamped (valid) address and the result is selected.  A per-lane
// `if (c < nch) load` makes the compiler emit exec-masked blocks separated by
// s_waitcnt vmcnt(0), which serialises the row loads of an iteration.
template <typename T, int VEC>
__device__ __forceinline__ void load_chunk(const T* __restrict__ row, int c, int nch, float (&out)[VEC]) {
    VecLoad<T, VEC>::load(row + min(c, nch - 1) * VEC, out);
    const bool live = c < nch;
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = live ? out[i] : 0.f;
}

// The same without the select: lanes past the end of the row read the row's last chunk.  For elementwise uses
// whose results for those lanes are never stored (one v_cndmask per scalar less in the inner loops).
template <typename T, int VEC>
__device__ __forceinline__ void load_chunk_clamped(const T* __restrict__ row, int c, int nch, float (&out)[VEC]) {
    VecLoad<T, VEC>::load(row + min(c, nch - 1) * VEC, out);
}

// sum over the 16 lanes of a DPP row; every lane of the row gets the total
__device__ __forceinline__ float row16_allreduce_sum(float v) {
#define BESS_DPP_ROR(x, n)                                                                     \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)),     \
                                                           0x120 | (n), 0xf, 0xf, false))
    v += BESS_DPP_ROR(v, 8);
    v += BESS_DPP_ROR(v, 4);
    v += BESS_DPP_ROR(v, 2);
    v += BESS_DPP_ROR(v, 1);
#undef BESS_DPP_ROR
    return v;
}

// sum over the 64 lanes of a wave; every lane gets the total
__device__ __forceinline__ float wave_allreduce_sum(float v) {
    v = row16_allreduce_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_allreduce_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

__device__ __forceinline__ float sgnf(float x) {
    return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
}

// "Is this the last workgroup of the launch to get here?" - called by ONE thread of every workgroup, after the
// workgroup's results are written and a release_to_agent() + __syncthreads().
// No device-scope fence: on this chip (one L2 per XCD) a __threadfence() is a write-back of the XCD's whole L2
// (buffer_wbl2) - with megabytes of freshly written score gradients in it, 256 workgroups fencing cost ~10 us of a
// 40 us launch.  Instead, what the last workgroup reads from the others (a few floats per workgroup) is written
// with agent-scope atomic stores (store_agent: write-through, complete once s_waitcnt has seen them) and read with
// agent-scope loads (load_agent); everything else becomes visible at the kernel boundary as usual.
// Tickets are taken in two levels: the workgroup's group (blockIdx.x % 16) has a counter of its own, the last arriver
// of each group takes a ticket at the root - 16 + 16 same-address atomics in a row at most instead of gridDim.x.
// counters: int32 [BESS_TICKET_INTS], zero on entry, left zero (every counter on a 128-byte line of its own).
__device__ __forceinline__ void store_agent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every store_agent of this wave has completed: the wait of a release fence without its cache write-back (a
// workgroup-scope fence emits no wait at all here - the waves of a workgroup share their CU's cache)
__device__ __forceinline__ void release_to_agent() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ bool last_workgroup_ticket(int32_t* counters) {
    const int grid = static_cast<int>(gridDim.x), sub = static_cast<int>(blockIdx.x) & 15;
    const int members = (grid - sub + 15) >> 4;
    int32_t* mine = counters + 32 * (1 + sub);
    if (__hip_atomic_fetch_add(mine, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != members - 1) return false;
    __hip_atomic_store(mine, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (nobody else comes here in this launch)
    if (__hip_atomic_fetch_add(counters, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != min(grid, 16) - 1) return false;
    __hip_atomic_store(counters, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}


// sgn(x - y) for operands pre-scaled by 2^100 (exact: power of two): the
// difference d' = (x - y) * 2^100 is clamped to [-1, 1] by one v_med3.  Exact for
// x == y and whenever |x - y| >= 2^-100; a smaller non-zero difference needs
// |x|, |y| < 2^-76, which embeddings never are.  No overflow below 2^27.
constexpr float SGN_PRESCALE = 1.2676506002282294e30f;  // 2^100
__device__ __forceinline__ float sgn_prescaled(float d_scaled) {
    return __builtin_amdgcn_fmed3f(d_scaled, -1.f, 1.f);
}

// Copy / fill jobs over 32-bit words that ride in the spare workgroups of a step's early launches
// (bess_step_prologue, bess_query_triple_fwd_jobs): the concatenated candidate list of an augmented step, cleared
// gradient targets, the generation counter of bess_direct_update ...
struct WordJobs {
    uint32_t* dst[BESS_MAX_WORD_JOBS];
    const uint32_t* src[BESS_MAX_WORD_JOBS];  // NULL: fill with value
    uint32_t value[BESS_MAX_WORD_JOBS];
    int64_t first[BESS_MAX_WORD_JOBS + 1];    // prefix sums of the jobs' word counts
    int n;
};
// workgroup `block` of `n_blocks` (of `threads` threads each) takes its share of the jobs' words
__device__ __forceinline__ void run_word_jobs(const WordJobs& J, int block, int n_blocks, int threads) {
    const int64_t total = J.first[J.n];
    const int64_t stride = static_cast<int64_t>(n_blocks) * threads;
    for (int64_t i = static_cast<int64_t>(block) * threads + threadIdx.x; i < total; i += stride) {
        int j = 0;
#pragma unroll
        for (int k = 1; k < BESS_MAX_WORD_JOBS; ++k) j += (k < J.n && i >= J.first[k]) ? 1 : 0;
        const int64_t off = i - J.first[j];
        // (copy jobs add their value to what they copy: 0 for a plain copy; src == dst with value 1 is the
        // increment of a device-side counter - the generation of bess_direct_update)
        J.dst[j][off] = J.src[j] ? J.src[j][off] + J.value[j] : J.value[j];
    }
}
// host side: checked jobs from the C ABI's arrays; *words = their total
int make_word_jobs(int32_t n_jobs, void* const* job_dst, const void* const* job_src, const uint32_t* job_value,
                   const int64_t* job_words, WordJobs* J, int64_t* words, const char* who);

// Fill `n` 32-bit words with `v` on `st` - a kernel, not hipMemsetAsync: memset nodes recorded into a hipGraph
// (torch.cuda.graph capture of a training step) did not reliably re-run in order on replay - a buffer zeroed
// this way kept what an earlier replay had left in its (pool-recycled) memory.  A kernel node does.
static __global__ __launch_bounds__(256) void k_fill_words(uint32_t* __restrict__ p, uint32_t v, int64_t n) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) p[i] = v;
}
inline hipError_t fill_words_async(void* p, uint32_t v, int64_t n_words, hipStream_t st) {
    if (n_words <= 0) return hipSuccess;
#ifdef BESS_PROBE_MEMSET_NODES
    // probe build only (make PROBE_MEMSET=1 -> profiles/ubench/bin/libbesskge_hip_memset.so, read by
    // profiles/graph_memset_probe.py): the round-2 form of the fills, to settle what a recorded memset node does
    if (v == 0u || v == 0xffffffffu) return hipMemsetAsync(p, static_cast<int>(v & 0xffu), sizeof(uint32_t) * n_words, st);
#endif
    int64_t blocks = (n_words + 255) / 256;
    blocks = blocks > 2048 ? 2048 : blocks;
    k_fill_words<<<static_cast<unsigned>(blocks), 256, 0, st>>>(static_cast<uint32_t*>(p), v, n_words);
    return hipGetLastError();
}

// Pieces of the p-norm of the distance scorers for any p >= 1 (reference scoring.py:174: `torch.norm(x,
// p=scoring_norm)`).  The kernels' RED_L2 branch carries p at run time: p = 2 keeps its multiply / sqrt (a
// wave-uniform test), every other p goes through powf.   norm = (sum |d|^p)^(1/p);   d norm / d d_w =
// sgn(d_w) |d_w|^(p-1) * norm^(1-p).
__device__ __forceinline__ float lp_term(float d, float p) { return p == 2.f ? d * d : powf(fabsf(d), p); }
__device__ __forceinline__ float lp_root(float acc, float p) { return p == 2.f ? sqrtf(acc) : powf(acc, 1.f / p); }
__device__ __forceinline__ float lp_dterm(float d, float p) {
    return p == 2.f ? d : copysignf(powf(fabsf(d), p - 1.f), d);
}
__device__ __forceinline__ float lp_inv(float norm, float p) {  // norm^(1-p); 0 at the kink norm == 0
    return norm > 0.f ? (p == 2.f ? 1.f / norm : powf(norm, 1.f - p)) : 0.f;
}

template <typename T>
__device__ __forceinline__ const T* row_ptr(const T* base, const int32_t* idx, int64_t i, int width) {
    const int64_t r = idx ? static_cast<int64_t>(idx[i]) : i;
    return base + r * width;
}

}  // namespace bess

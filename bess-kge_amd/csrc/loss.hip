// K7 (mask / augment) and K8 (loss + score gradients) of the BESS step on gfx950.
//
//   K7: reference bess.py:182-245 - additive BAD_NEGATIVE_SCORE on padding
//       negatives and on the true head/tail among augmented negatives.
//   K8: reference loss.py:28-51 (self-adversarial weights = detached softmax),
//       loss.py:115-134 (log-sigmoid), 179-195 (margin ranking),
//       224-251 (sampled-softmax cross entropy); always fp32, summed (not
//       averaged) over the micro-batch (bess.py:254-260).
//
// One wavefront per triple row: the row of n_neg scores is streamed 2-3 times
// (max, normaliser, terms) - it is L2 resident right after the scoring kernel.
// The scalar loss is reduced in a fixed order (per-row terms, then one
// single-workgroup tree) so it is bitwise reproducible run to run.
#include "common.h"
#include "loss_rows.h"

namespace bess {

__global__ __launch_bounds__(256) void k_mask_scores(float* __restrict__ neg, int64_t n_triple,
                                                     int64_t n_neg, int64_t ld, int diag_step, int ht,
                                                     int ppp, const uint8_t* __restrict__ mask,
                                                     int64_t mask_rows, int64_t mask_cols) {
    const int64_t total = n_triple * n_neg;
    const int cut = ppp / 2;
    for (int64_t t = blockIdx.x * 256ll + threadIdx.x; t < total; t += 256ll * gridDim.x) {
        const int64_t s = t / n_neg;
        const int64_t j = t - s * n_neg;
        const int64_t blk = ppp > 0 ? s / ppp : 0;
        const int p = ppp > 0 ? static_cast<int>(s - blk * ppp) : 0;
        bool kill = false;
        if (diag_step > 0) {
            const int64_t qpos = ht ? (blk * cut + (p % cut)) : s;
            kill = (j == static_cast<int64_t>(diag_step) * qpos);
        }
        const int64_t mj = j - (n_neg - mask_cols);
        if (mask && mj >= 0) {
            int64_t mrow = s;
            if (mask_rows == 1) mrow = 0;
            else if (mask_rows == 2) mrow = (p >= cut) ? 1 : 0;
            // the mask overrides the diagonal on its columns (bess.py:227-228)
            kill = mask[mrow * mask_cols + mj] == 0;
        }
        if (kill) neg[s * ld + j] += BESS_BAD_NEGATIVE_SCORE;
    }
}

template <int KIND, bool ADV, bool GRAD, int CH>
__global__ __launch_bounds__(256) void k_loss_rows(bess_loss_desc l, const float* __restrict__ pos,
                                                   const float* __restrict__ neg, int64_t n_triple,
                                                   int64_t n_neg, int64_t ld_neg,
                                                   const float* __restrict__ weight,
                                                   int64_t weight_len, float* __restrict__ row_loss,
                                                   float* __restrict__ d_pos,
                                                   float* __restrict__ d_neg, int64_t ld_dneg,
                                                   float* __restrict__ row_norm, int32_t* __restrict__ counter,
                                                   float* __restrict__ loss) {
    const int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (s < n_triple)
        loss_row<KIND, ADV, GRAD, CH>(l, pos, neg, s, n_neg, ld_neg, weight, weight_len, row_loss, d_pos, d_neg, ld_dneg,
                                      row_norm);
    if (!counter) return;  // (uniform: the sum is a second launch, k_sum_rows)
    // One launch: the workgroup that finishes last sums the row terms, in a fixed order (bitwise reproducible
    // whatever the order the workgroups ran in), and leaves the counter at zero for the next call.
    __shared__ float part[256];
    __shared__ int last;
    release_to_agent();
    __syncthreads();
    if (threadIdx.x == 0) last = last_workgroup_ticket(counter);
    __syncthreads();
    if (!last) return;
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n_triple; i += 256) acc += load_agent(row_loss + i);
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (threadIdx.x < h) part[threadIdx.x] += part[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = part[0];
}

// Prediction rank of the positive among its candidates (reference metric.py:129-182):
// 1 + #{candidates better}; "better" = '>' (optimistic), '>=' (pessimistic) or the
// mean of the two (average).  NaN positives count as -inf.  With worst_inf a
// positive beaten by every candidate gets rank +inf.
__global__ __launch_bounds__(256) void k_ranks_from_scores(const float* __restrict__ pos,
                                                           const float* __restrict__ cand, int64_t n_row,
                                                           int64_t n_cand, int64_t ld, int mode, int worst_inf,
                                                           float* __restrict__ rank) {
    const int lane = threadIdx.x & 63;
    const int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (s >= n_row) return;
    float p = pos[s];
    if (p != p) p = -INFINITY;
    const float* row = cand + s * ld;
    float gt = 0.f, ge = 0.f;
    for (int64_t j = lane; j < n_cand; j += 64) {
        const float c = row[j];
        gt += (c > p) ? 1.f : 0.f;
        ge += (c >= p) ? 1.f : 0.f;
    }
    gt = wave_allreduce_sum(gt);
    ge = wave_allreduce_sum(ge);
    if (lane == 0) {
        const float n = static_cast<float>(n_cand);
        float better;
        bool worst;
        if (mode == 0) { better = gt; worst = gt == n; }
        else if (mode == 1) { better = ge; worst = ge == n; }
        else { better = 0.5f * (gt + ge); worst = (gt == n) || (ge == n); }
        rank[s] = (worst_inf && worst) ? INFINITY : 1.f + better;
    }
}

// Rank from ordered candidate ids (reference metric.py:184-217): position (1-based)
// of the ground truth in the list, else n + 1 (or +inf).
__global__ __launch_bounds__(256) void k_ranks_from_indices(const int64_t* __restrict__ truth,
                                                            const int64_t* __restrict__ cand, int64_t n_row,
                                                            int64_t n_cand, int worst_inf,
                                                            float* __restrict__ rank) {
    const int64_t s = blockIdx.x * 256ll + threadIdx.x;
    if (s >= n_row) return;
    const int64_t t = truth[s];
    float r = worst_inf ? INFINITY : static_cast<float>(n_cand + 1);
    for (int64_t j = n_cand - 1; j >= 0; --j)
        if (cand[s * n_cand + j] == t) r = static_cast<float>(j + 1);
    rank[s] = r;
}

// fixed-order sum of row_loss -> loss[0]
__global__ __launch_bounds__(1024) void k_sum_rows(const float* __restrict__ row_loss, int64_t n,
                                                   float* __restrict__ loss) {
    __shared__ float part[1024];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) acc += row_loss[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = part[0];
}

template <int KIND, bool ADV, int CH>
static void launch_loss_ch(bool grad, const bess_loss_desc& l, const float* pos, const float* neg,
                           int64_t S, int64_t N, int64_t ld, const float* w, int64_t wl, float* rl,
                           float* dp, float* dn, int64_t ldd, float* rn, int32_t* cnt, float* loss, hipStream_t st) {
    const unsigned grid = static_cast<unsigned>(ceil_div(S, 4));
    if (grad) k_loss_rows<KIND, ADV, true, CH><<<grid, 256, 0, st>>>(l, pos, neg, S, N, ld, w, wl, rl, dp, dn, ldd, rn, cnt, loss);
    else k_loss_rows<KIND, ADV, false, CH><<<grid, 256, 0, st>>>(l, pos, neg, S, N, ld, w, wl, rl, dp, dn, ldd, rn, cnt, loss);
}

template <int KIND, bool ADV>
static void launch_loss(bool grad, const bess_loss_desc& l, const float* pos, const float* neg,
                        int64_t S, int64_t N, int64_t ld, const float* w, int64_t wl, float* rl,
                        float* dp, float* dn, int64_t ldd, float* rn, int32_t* cnt, float* loss, hipStream_t st) {
    // rows in registers when their shape allows 16-byte accesses (see k_loss_rows)
    const bool vec = N % 4 == 0 && ld % 4 == 0 && reinterpret_cast<uintptr_t>(neg) % 16 == 0 &&
                     (!grad || (ldd % 4 == 0 && reinterpret_cast<uintptr_t>(dn) % 16 == 0));
#define BESS_LOSS_CH(CH) launch_loss_ch<KIND, ADV, CH>(grad, l, pos, neg, S, N, ld, w, wl, rl, dp, dn, ldd, rn, cnt, loss, st)
    if (!vec) BESS_LOSS_CH(0);
    else if (N > 256 * 24) BESS_LOSS_CH(-1);
    else if (N <= 256 * 4) BESS_LOSS_CH(4);
    else if (N <= 256 * 12) BESS_LOSS_CH(12);
    else BESS_LOSS_CH(24);
#undef BESS_LOSS_CH
}

}  // namespace bess

using namespace bess;

extern "C" int bess_mask_scores(float* neg, int64_t n_triple, int64_t n_neg, int64_t ld,
                                int32_t diag_step, int32_t ht, int32_t ppp, const uint8_t* mask,
                                int64_t mask_rows, int64_t mask_cols, void* stream) {
    BESS_REQUIRE(n_triple >= 0 && n_neg >= 0 && ld >= n_neg, "mask_scores: bad sizes");
    if (n_triple == 0 || n_neg == 0) return BESS_OK;
    BESS_REQUIRE(neg, "mask_scores: NULL scores");
    BESS_REQUIRE(diag_step >= 0, "mask_scores: negative diag_step");
    if (mask) {
        BESS_REQUIRE(mask_cols > 0 && mask_cols <= n_neg, "mask_scores: mask_cols %lld not in (0, n_neg]", (long long)mask_cols);
        BESS_REQUIRE(mask_rows == 1 || mask_rows == 2 || mask_rows == n_triple,
                     "mask_scores: mask_rows %lld not 1, 2 or n_triple", (long long)mask_rows);
    } else {
        mask_cols = 0;
    }
    if (ht || mask_rows == 2)
        BESS_REQUIRE(ppp >= 2 && (ppp % 2) == 0 && (n_triple % ppp) == 0, "mask_scores: 'ht' needs an even block size dividing n_triple");
    const int64_t total = n_triple * n_neg;
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 4096) blocks = 4096;
    k_mask_scores<<<static_cast<unsigned>(blocks), 256, 0, as_stream(stream)>>>(
        neg, n_triple, n_neg, ld, diag_step, ht, ppp, mask, mask_rows, mask_cols);
    return check_launch("mask_scores");
}

extern "C" int bess_loss_fwd_bwd(const bess_loss_desc* l, const float* pos, const float* neg,
                                 int64_t n_triple, int64_t n_neg, int64_t ld_neg,
                                 const float* weight, int64_t weight_len, float* row_loss,
                                 float* loss, float* d_pos, float* d_neg, int64_t ld_dneg,
                                 void* stream) {
    return bess_loss_fwd_bwd_norm(l, pos, neg, n_triple, n_neg, ld_neg, weight, weight_len, row_loss, loss, d_pos, d_neg,
                                  ld_dneg, nullptr, stream);
}

static int loss_impl(const bess_loss_desc* l, const float* pos, const float* neg, int64_t n_triple, int64_t n_neg,
                     int64_t ld_neg, const float* weight, int64_t weight_len, float* row_loss, float* loss,
                     float* d_pos, float* d_neg, int64_t ld_dneg, float* row_norm, int32_t* counter, void* stream) {
    BESS_REQUIRE(l, "loss: NULL descriptor");
    BESS_REQUIRE(l->kind >= BESS_LOSS_LOGSIGMOID && l->kind <= BESS_LOSS_SSCE, "loss: unknown kind %d", l->kind);
    BESS_REQUIRE(n_triple > 0 && n_neg > 0 && n_neg < (1ll << 31), "loss: bad sizes");
    BESS_REQUIRE(pos && neg && weight && row_loss && loss, "loss: NULL pointer");
    BESS_REQUIRE(weight_len == 1 || weight_len == n_triple, "loss: weight_len must be 1 or n_triple");
    BESS_REQUIRE(ld_neg >= n_neg, "loss: leading dimension < n_neg");
    const bool grad = d_pos || d_neg;
    if (grad) BESS_REQUIRE(d_pos && d_neg && ld_dneg >= n_neg, "loss: gradients need d_pos, d_neg and ld_dneg >= n_neg");
    hipStream_t st = as_stream(stream);
    const bool adv = l->adversarial != 0;
    // the one-launch form costs one atomic per workgroup (4 rows each), taken in two levels (last_workgroup_ticket:
    // same-address atomics serialise at ~40 ns a piece).  Very large grids keep the second launch.
    if (n_triple > 4 * 4096) counter = nullptr;
#define BESS_LOSS(KIND, ADV) \
    launch_loss<KIND, ADV>(grad, *l, pos, neg, n_triple, n_neg, ld_neg, weight, weight_len, row_loss, d_pos, d_neg, ld_dneg, \
                           row_norm, counter, loss, st)
    switch (l->kind) {
        case BESS_LOSS_LOGSIGMOID: adv ? BESS_LOSS(BESS_LOSS_LOGSIGMOID, true) : BESS_LOSS(BESS_LOSS_LOGSIGMOID, false); break;
        case BESS_LOSS_MARGIN: adv ? BESS_LOSS(BESS_LOSS_MARGIN, true) : BESS_LOSS(BESS_LOSS_MARGIN, false); break;
        default: BESS_LOSS(BESS_LOSS_SSCE, false);
    }
#undef BESS_LOSS
    if (int e = check_launch("loss rows")) return e;
    if (counter) return BESS_OK;  // summed by the launch's last workgroup
    k_sum_rows<<<1, 1024, 0, st>>>(row_loss, n_triple, loss);
    return check_launch("loss sum");
}

extern "C" int bess_loss_fwd_bwd_norm(const bess_loss_desc* l, const float* pos, const float* neg,
                                      int64_t n_triple, int64_t n_neg, int64_t ld_neg,
                                      const float* weight, int64_t weight_len, float* row_loss,
                                      float* loss, float* d_pos, float* d_neg, int64_t ld_dneg,
                                      float* row_norm, void* stream) {
    return loss_impl(l, pos, neg, n_triple, n_neg, ld_neg, weight, weight_len, row_loss, loss, d_pos, d_neg, ld_dneg,
                     row_norm, nullptr, stream);
}

extern "C" int bess_loss_fwd_bwd_one_launch(const bess_loss_desc* l, const float* pos, const float* neg,
                                            int64_t n_triple, int64_t n_neg, int64_t ld_neg,
                                            const float* weight, int64_t weight_len, float* row_loss,
                                            float* loss, float* d_pos, float* d_neg, int64_t ld_dneg,
                                            float* row_norm, int32_t* counter, void* stream) {
    BESS_REQUIRE(counter, "loss_one_launch: NULL counter");
    return loss_impl(l, pos, neg, n_triple, n_neg, ld_neg, weight, weight_len, row_loss, loss, d_pos, d_neg, ld_dneg,
                     row_norm, counter, stream);
}

extern "C" int bess_ranks_from_scores(const float* pos, const float* cand, int64_t n_row, int64_t n_cand,
                                      int64_t ld, int32_t mode, int32_t worst_rank_infty, float* rank,
                                      void* stream) {
    BESS_REQUIRE(n_row >= 0 && n_cand > 0 && ld >= n_cand, "ranks_from_scores: bad sizes");
    BESS_REQUIRE(mode >= 0 && mode <= 2, "ranks_from_scores: mode must be 0 (optimistic), 1 (pessimistic), 2 (average)");
    if (n_row == 0) return BESS_OK;
    BESS_REQUIRE(pos && cand && rank, "ranks_from_scores: NULL pointer");
    k_ranks_from_scores<<<static_cast<unsigned>(ceil_div(n_row, 4)), 256, 0, as_stream(stream)>>>(
        pos, cand, n_row, n_cand, ld, mode, worst_rank_infty, rank);
    return check_launch("ranks_from_scores");
}

extern "C" int bess_ranks_from_indices(const int64_t* ground_truth, const int64_t* candidates, int64_t n_row,
                                       int64_t n_cand, int32_t worst_rank_infty, float* rank, void* stream) {
    BESS_REQUIRE(n_row >= 0 && n_cand > 0, "ranks_from_indices: bad sizes");
    if (n_row == 0) return BESS_OK;
    BESS_REQUIRE(ground_truth && candidates && rank, "ranks_from_indices: NULL pointer");
    k_ranks_from_indices<<<static_cast<unsigned>(ceil_div(n_row, 256)), 256, 0, as_stream(stream)>>>(
        ground_truth, candidates, n_row, n_cand, worst_rank_infty, rank);
    return check_launch("ranks_from_indices");
}

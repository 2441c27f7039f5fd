// K11 on gfx950: streaming top-k of score rows (TopKQueryBessKGE, next-1).
//
// Replaces `torch.topk(torch.concat([window_scores, running_best]))` +
// `gather_indices` of the reference's sliding-window loop (bess.py:771-822) and
// the final `torch.topk` over the shards' lists (bess.py:889-894).
//
// One wavefront per query row.  The running list (kk <= 64 entries, sorted by
// descending score) lives in registers, entry j in lane j.  The window is
// streamed 64 candidates at a time; a candidate enters only if it beats the
// current kk-th score tau (`__ballot(x > tau)`), so after the first few chunks
// almost every chunk costs one load, one compare and one ballot: the expected
// number of insertions over a stream of L random scores is ~kk * ln(L / kk).
// An insertion is O(1) wave operations: rank by ballot + popcount, shift by one
// lane (DPP-style __shfl_up), write the new entry.  Ties keep the earlier entry
// first (torch.topk leaves the order of equal scores unspecified).
#include "common.h"

namespace bess {

__global__ __launch_bounds__(256) void k_topk_update(const float* __restrict__ scores, int64_t n_row,
                                                     int64_t n_col, int64_t ld, const int32_t* __restrict__ ids,
                                                     int64_t ids_rows, int32_t id_base,
                                                     const uint8_t* __restrict__ mask, int64_t mask_rows,
                                                     float* __restrict__ best_score,
                                                     int32_t* __restrict__ best_id, int kk) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (row >= n_row) return;
    float bs = -INFINITY;
    int32_t bi = 0;
    if (lane < kk) {
        bs = best_score[row * kk + lane];
        bi = best_id[row * kk + lane];
    }
    float tau = __shfl(bs, kk - 1, 64);
    const float* srow = scores + row * ld;
    const int32_t* irow = ids ? ids + (ids_rows == 1 ? 0 : row) * n_col : nullptr;
    const uint8_t* mrow = mask ? mask + (mask_rows == 1 ? 0 : row) * n_col : nullptr;
    // U chunks of 64 candidates are loaded back to back (U independent loads in
    // flight per lane: the row is streamed, not pointer-chased), then examined
    constexpr int U = 8;
    for (int64_t c0 = 0; c0 < n_col; c0 += 64 * U) {
        float xs[U];
        int32_t xis[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = c0 + 64 * u + lane;
            xs[u] = -INFINITY;
            xis[u] = 0;
            if (j < n_col) {
                xs[u] = srow[j];
                if (mrow && mrow[j] == 0) xs[u] += BESS_BAD_NEGATIVE_SCORE;
                xis[u] = irow ? irow[j] : id_base + static_cast<int32_t>(j);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float x = xs[u];
            const int32_t xi = xis[u];
            unsigned long long m = __ballot(x > tau);
            while (m) {
                const int l = __ffsll(static_cast<long long>(m)) - 1;
                const float xv = __shfl(x, l, 64);
                const int32_t iv = __shfl(xi, l, 64);
                // entries that stay ahead of the newcomer (>=: earlier entries win ties)
                const int pos = __popcll(__ballot(lane < kk && bs >= xv));
                const float up_s = __shfl_up(bs, 1, 64);
                const int32_t up_i = __shfl_up(bi, 1, 64);
                if (lane < kk) {
                    if (lane > pos) {
                        bs = up_s;
                        bi = up_i;
                    } else if (lane == pos) {
                        bs = xv;
                        bi = iv;
                    }
                }
                tau = __shfl(bs, kk - 1, 64);
                m &= ~(1ull << l);
                m &= __ballot(x > tau);
            }
        }
    }
    if (lane < kk) {
        best_score[row * kk + lane] = bs;
        best_id[row * kk + lane] = bi;
    }
}

}  // namespace bess

using namespace bess;

extern "C" int bess_topk_update(const float* scores, int64_t n_row, int64_t n_col, int64_t ld,
                                const int32_t* ids, int64_t ids_rows, int32_t id_base, const uint8_t* mask,
                                int64_t mask_rows, float* best_score, int32_t* best_id, int32_t kk,
                                void* stream) {
    BESS_REQUIRE(n_row >= 0 && n_col >= 0 && ld >= n_col, "topk_update: bad sizes");
    BESS_REQUIRE(kk >= 1 && kk <= 64, "topk_update: list length %d not in [1, 64]", kk);
    if (n_row == 0 || n_col == 0) return BESS_OK;
    BESS_REQUIRE(scores && best_score && best_id, "topk_update: NULL pointer");
    BESS_REQUIRE(!ids || ids_rows == 1 || ids_rows == n_row, "topk_update: ids_rows must be 1 or n_row");
    BESS_REQUIRE(!mask || mask_rows == 1 || mask_rows == n_row, "topk_update: mask_rows must be 1 or n_row");
    k_topk_update<<<static_cast<unsigned>(ceil_div(n_row, 4)), 256, 0, as_stream(stream)>>>(
        scores, n_row, n_col, ld, ids, ids_rows, id_base, mask, mask_rows, best_score, best_id, kk);
    return check_launch("topk_update");
}

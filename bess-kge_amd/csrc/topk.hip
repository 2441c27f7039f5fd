// K11 on gfx950: streaming top-k of score rows (TopKQueryBessKGE, next-1).
//
// Replaces `torch.topk(torch.concat([window_scores, running_best]))` +
// `gather_indices` of the reference's sliding-window loop (bess.py:771-822) and
// the final `torch.topk` over the shards' lists (bess.py:889-894).
//
// One workgroup (four wavefronts) per query row.  A wave's list (kk <= 64 entries, sorted by
// descending score) lives in registers, entry j in lane j (65 .. 128 entries: two registers per lane).  The window is
// streamed 64 candidates at a time; a candidate enters only if it beats the
// current kk-th score tau (`__ballot(x > tau)`), so after the first few chunks
// almost every chunk costs one load, one compare and one ballot: the expected
// number of insertions over a stream of L random scores is ~kk * ln(L / kk).
// An insertion is O(1) wave operations: rank by ballot + popcount, shift by one
// lane (DPP-style __shfl_up), write the new entry.  Ties keep the earlier entry
// first (torch.topk leaves the order of equal scores unspecified).
#include "common.h"

namespace bess {

// WPR waves per query row (4 rows per workgroup, or 1 row whose four waves each stream a
// contiguous quarter of the columns - more bytes in flight when there are few rows, at the price
// of one list warm-up per wave).  With WPR = 4, wave 0 starts from the running list, waves 1-3 from
// empty lists; at the end wave 0 examines the other three lists as three more chunks, in wave
// order - quarters are in column order, so ties resolve exactly as in a single left-to-right pass.
// VEC: rows are 16-B aligned (ld % 4 == 0), a lane loads 4 consecutive columns with one instruction.
// TWO: lists of 65 .. 128 entries - entry j < 64 in lane j of the first register pair, entry 64 + j in lane j of
// the second; an insertion shifts the first into the second through lane 63 -> lane 0.
template <int WPR, bool VEC, bool TWO>
__global__ __launch_bounds__(256) void k_topk_update(const float* __restrict__ scores, int64_t n_row,
                                                     int64_t n_col, int64_t ld, const int32_t* __restrict__ ids,
                                                     int64_t ids_rows, int32_t id_base,
                                                     const uint8_t* __restrict__ mask, int64_t mask_rows,
                                                     float* __restrict__ best_score,
                                                     int32_t* __restrict__ best_id, int kk) {
    __shared__ float l_s[3][TWO ? 128 : 64];
    __shared__ int32_t l_i[3][TWO ? 128 : 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = WPR == 4 ? wave : 0;  // which quarter of the columns
    const int64_t row = WPR == 4 ? static_cast<int64_t>(blockIdx.x) : blockIdx.x * 4ll + wave;
    if (row >= n_row) return;  // WPR == 1 only (whole waves; no barrier on that path)
    float bs = -INFINITY, bs1 = -INFINITY;  // (bs1, bi1): entries 64 .. 127 (TWO)
    int32_t bi = 0, bi1 = 0;
    const int k0 = TWO ? 64 : kk;  // entries held by the first pair
    if (part == 0 && lane < k0) {
        bs = best_score[row * kk + lane];
        bi = best_id[row * kk + lane];
    }
    if (TWO && part == 0 && 64 + lane < kk) {
        bs1 = best_score[row * kk + 64 + lane];
        bi1 = best_id[row * kk + 64 + lane];
    }
    float tau = TWO ? __shfl(bs1, kk - 65, 64) : __shfl(bs, kk - 1, 64);
    // candidate (xv, iv) enters the list if it beats the kk-th entry
    auto insert = [&](float xv, int32_t iv) {
        // entries that stay ahead of the newcomer (>=: earlier entries win ties)
        const int pos0 = __popcll(__ballot(lane < k0 && bs >= xv));
        const float up_s = __shfl_up(bs, 1, 64);
        const int32_t up_i = __shfl_up(bi, 1, 64);
        if (TWO) {
            // the second half: shifted as a whole when the newcomer lands in the first (whose last entry
            // moves over), from the newcomer's place on when it lands here
            const int pos1 = __popcll(__ballot(64 + lane < kk && bs1 >= xv));
            const float last_s = __shfl(bs, 63, 64);
            const int32_t last_i = __shfl(bi, 63, 64);
            float up1_s = __shfl_up(bs1, 1, 64);
            int32_t up1_i = __shfl_up(bi1, 1, 64);
            if (lane == 0) {
                up1_s = last_s;
                up1_i = last_i;
            }
            if (64 + lane < kk) {
                if (pos0 < 64) {  // newcomer in the first half: everything here moves up by one
                    bs1 = up1_s;
                    bi1 = up1_i;
                } else if (lane > pos1) {
                    bs1 = up1_s;
                    bi1 = up1_i;
                } else if (lane == pos1) {
                    bs1 = xv;
                    bi1 = iv;
                }
            }
        }
        if (lane < k0) {
            if (lane > pos0) {
                bs = up_s;
                bi = up_i;
            } else if (lane == pos0) {
                bs = xv;
                bi = iv;
            }
        }
        tau = TWO ? __shfl(bs1, kk - 65, 64) : __shfl(bs, kk - 1, 64);
    };
    // a chunk of 64 candidates (one per lane), examined in lane order
    auto examine = [&](float x, int32_t xi) {
        unsigned long long m = __ballot(x > tau);
        while (m) {
            const int l = __ffsll(static_cast<long long>(m)) - 1;
            insert(__shfl(x, l, 64), __shfl(xi, l, 64));
            m &= ~(1ull << l);
            m &= __ballot(x > tau);
        }
    };
    const float* srow = scores + row * ld;
    const int32_t* irow = ids ? ids + (ids_rows == 1 ? 0 : row) * n_col : nullptr;
    const uint8_t* mrow = mask ? mask + (mask_rows == 1 ? 0 : row) * n_col : nullptr;
    // U groups are loaded back to back (U independent loads in flight per lane: the row is
    // streamed, not pointer-chased), then examined
    constexpr int U = VEC ? 4 : 8, PER = VEC ? 4 : 1, STEP = 64 * U * PER;
    const int64_t seg = WPR == 4 ? (n_col + 4 * STEP - 1) / (4 * STEP) * STEP : n_col;  // columns per wave
    const int64_t c_begin = part * seg, c_end = min(n_col, c_begin + seg);
    for (int64_t c0 = c_begin; c0 < c_end; c0 += STEP) {
        float xs[U][PER];
        int32_t xis[U][PER];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = c0 + 64 * PER * u + lane * PER;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                xs[u][i] = -INFINITY;
                xis[u][i] = 0;
            }
            if constexpr (VEC) {
                if (j + 3 < c_end) {
                    VecLoad<float, 4>::load(srow + j, xs[u]);
                } else {  // the last, partial group of the row
#pragma unroll
                    for (int i = 0; i < PER; ++i)
                        if (j + i < c_end) xs[u][i] = srow[j + i];
                }
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    if (j + i < c_end) {
                        if (mrow && mrow[j + i] == 0) xs[u][i] += BESS_BAD_NEGATIVE_SCORE;
                        xis[u][i] = irow ? irow[j + i] : id_base + static_cast<int32_t>(j + i);
                    }
                }
            } else if (j < c_end) {
                xs[u][0] = srow[j];
                if (mrow && mrow[j] == 0) xs[u][0] += BESS_BAD_NEGATIVE_SCORE;
                xis[u][0] = irow ? irow[j] : id_base + static_cast<int32_t>(j);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (!VEC) {
                examine(xs[u][0], xis[u][0]);
            } else {
                // 256 candidates, column = 4 * lane + i: taken in column order (lowest lane first,
                // then lowest component), so that equal scores keep their left-to-right order
                unsigned long long any = __ballot(xs[u][0] > tau || xs[u][1] > tau || xs[u][2] > tau || xs[u][3] > tau);
                while (any) {
                    const int l = __ffsll(static_cast<long long>(any)) - 1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float xv = __shfl(xs[u][i], l, 64);
                        if (xv > tau) insert(xv, __shfl(xis[u][i], l, 64));  // wave-uniform branch
                    }
                    any &= ~(1ull << l);
                    any &= __ballot(xs[u][0] > tau || xs[u][1] > tau || xs[u][2] > tau || xs[u][3] > tau);
                }
            }
        }
    }
    if (WPR == 4) {
        if (wave > 0) {
            l_s[wave - 1][lane] = lane < k0 ? bs : -INFINITY;
            l_i[wave - 1][lane] = bi;
            if (TWO) {
                l_s[wave - 1][64 + lane] = 64 + lane < kk ? bs1 : -INFINITY;
                l_i[wave - 1][64 + lane] = bi1;
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                examine(l_s[w][lane], l_i[w][lane]);
                if (TWO) examine(l_s[w][64 + lane], l_i[w][64 + lane]);
            }
        }
    }
    if (part == 0 && lane < k0) {
        best_score[row * kk + lane] = bs;
        best_id[row * kk + lane] = bi;
    }
    if (TWO && part == 0 && 64 + lane < kk) {
        best_score[row * kk + 64 + lane] = bs1;
        best_id[row * kk + 64 + lane] = bi1;
    }
}

// The same update when the scoring kernel has pruned the tile against the rows' current k-th scores
// (bess_neg_score_shared_fwd_pruned): flags[row, b] != 0 marks the blocks of 64 columns that hold a score above
// the threshold the row had when the tile was scored - only those were written, only those are read.  One wave
// per row: a dword of flags per lane names 256 blocks (16,384 columns) per step, the flagged ones are fetched
// four at a time (four independent loads in flight) and examined in column order, so equal scores keep their
// left-to-right order exactly as in the dense pass.  After the first tiles of a long row almost nothing is
// flagged: the pass costs the flag bytes (1/256 of the scores).
template <bool TWO>
__global__ __launch_bounds__(256) void k_topk_update_flagged(const float* __restrict__ scores, int64_t n_row,
                                                             int64_t n_col, int64_t ld,
                                                             const uint32_t* __restrict__ flags, int64_t ldf32,
                                                             int32_t id_base, float* __restrict__ best_score,
                                                             int32_t* __restrict__ best_id, int kk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = blockIdx.x * 4ll + wave;
    if (row >= n_row) return;
    float bs = -INFINITY, bs1 = -INFINITY;
    int32_t bi = 0, bi1 = 0;
    const int k0 = TWO ? 64 : kk;
    if (lane < k0) {
        bs = best_score[row * kk + lane];
        bi = best_id[row * kk + lane];
    }
    if (TWO && 64 + lane < kk) {
        bs1 = best_score[row * kk + 64 + lane];
        bi1 = best_id[row * kk + 64 + lane];
    }
    float tau = TWO ? __shfl(bs1, kk - 65, 64) : __shfl(bs, kk - 1, 64);
    auto insert = [&](float xv, int32_t iv) {
        const int pos0 = __popcll(__ballot(lane < k0 && bs >= xv));
        const float up_s = __shfl_up(bs, 1, 64);
        const int32_t up_i = __shfl_up(bi, 1, 64);
        if (TWO) {
            const int pos1 = __popcll(__ballot(64 + lane < kk && bs1 >= xv));
            const float last_s = __shfl(bs, 63, 64);
            const int32_t last_i = __shfl(bi, 63, 64);
            float up1_s = __shfl_up(bs1, 1, 64);
            int32_t up1_i = __shfl_up(bi1, 1, 64);
            if (lane == 0) {
                up1_s = last_s;
                up1_i = last_i;
            }
            if (64 + lane < kk) {
                if (pos0 < 64 || lane > pos1) {
                    bs1 = up1_s;
                    bi1 = up1_i;
                } else if (lane == pos1) {
                    bs1 = xv;
                    bi1 = iv;
                }
            }
        }
        if (lane < k0) {
            if (lane > pos0) {
                bs = up_s;
                bi = up_i;
            } else if (lane == pos0) {
                bs = xv;
                bi = iv;
            }
        }
        tau = TWO ? __shfl(bs1, kk - 65, 64) : __shfl(bs, kk - 1, 64);
    };
    auto examine = [&](float x, int32_t xi) {
        unsigned long long m = __ballot(x > tau);
        while (m) {
            const int l = __ffsll(static_cast<long long>(m)) - 1;
            insert(__shfl(x, l, 64), __shfl(xi, l, 64));
            m &= ~(1ull << l);
            m &= __ballot(x > tau);
        }
    };
    const float* srow = scores + row * ld;
    const int64_t n_block = (n_col + 63) / 64;
    const int64_t n_word = (n_block + 3) / 4;
    const uint32_t* frow = flags + row * ldf32;
    for (int64_t w0 = 0; w0 < n_word; w0 += 64) {
        const uint32_t f = w0 + lane < n_word ? frow[w0 + lane] : 0u;
        unsigned long long m = __ballot(f != 0u);
        // flagged blocks of this step, four at a time: (lane of the word, byte in it) in column order
        int64_t blk[4];
        int n_blk = 0;
        auto flush = [&]() {
            float x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t j = blk[i < n_blk ? i : 0] * 64 + lane;
                x[i] = (i < n_blk && j < n_col) ? srow[j] : -INFINITY;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < n_blk) examine(x[i], id_base + static_cast<int32_t>(blk[i] * 64 + lane));
            n_blk = 0;
        };
        while (m) {
            const int l = __ffsll(static_cast<long long>(m)) - 1;
            m &= ~(1ull << l);
            const uint32_t fl = __shfl(f, l, 64);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if ((fl >> (8 * b)) & 0xffu) {  // wave-uniform
                    const int64_t block = (w0 + l) * 4 + b;
                    if (block < n_block) {
                        blk[n_blk++] = block;
                        if (n_blk == 4) flush();
                    }
                }
            }
        }
        if (n_blk) flush();
    }
    if (lane < k0) {
        best_score[row * kk + lane] = bs;
        best_id[row * kk + lane] = bi;
    }
    if (TWO && 64 + lane < kk) {
        best_score[row * kk + 64 + lane] = bs1;
        best_id[row * kk + 64 + lane] = bi1;
    }
}

}  // namespace bess

using namespace bess;

extern "C" int bess_topk_update_flagged(const float* scores, int64_t n_row, int64_t n_col, int64_t ld,
                                        const uint8_t* flags, int64_t ld_flags, int32_t id_base, float* best_score,
                                        int32_t* best_id, int32_t kk, void* stream) {
    BESS_REQUIRE(n_row >= 0 && n_row < (1ll << 31) && n_col >= 0 && ld >= n_col, "topk_update_flagged: bad sizes");
    BESS_REQUIRE(kk >= 1 && kk <= 128, "topk_update_flagged: list length %d not in [1, 128]", kk);
    if (n_row == 0 || n_col == 0) return BESS_OK;
    BESS_REQUIRE(scores && flags && best_score && best_id, "topk_update_flagged: NULL pointer");
    BESS_REQUIRE(ld_flags % 4 == 0 && ld_flags >= (n_col + 63) / 64 && reinterpret_cast<uintptr_t>(flags) % 4 == 0,
                 "topk_update_flagged: flag rows must be 4-byte aligned and hold one byte per 64 columns");
    const unsigned grid = static_cast<unsigned>(ceil_div(n_row, 4));
    const uint32_t* f32 = reinterpret_cast<const uint32_t*>(flags);
    if (kk > 64)
        k_topk_update_flagged<true><<<grid, 256, 0, as_stream(stream)>>>(scores, n_row, n_col, ld, f32, ld_flags / 4,
                                                                         id_base, best_score, best_id, kk);
    else
        k_topk_update_flagged<false><<<grid, 256, 0, as_stream(stream)>>>(scores, n_row, n_col, ld, f32, ld_flags / 4,
                                                                          id_base, best_score, best_id, kk);
    return check_launch("topk_update_flagged");
}

extern "C" int bess_topk_update(const float* scores, int64_t n_row, int64_t n_col, int64_t ld,
                                const int32_t* ids, int64_t ids_rows, int32_t id_base, const uint8_t* mask,
                                int64_t mask_rows, float* best_score, int32_t* best_id, int32_t kk,
                                void* stream) {
    BESS_REQUIRE(n_row >= 0 && n_row < (1ll << 31) && n_col >= 0 && ld >= n_col, "topk_update: bad sizes");
    BESS_REQUIRE(kk >= 1 && kk <= 128, "topk_update: list length %d not in [1, 128]", kk);
    if (n_row == 0 || n_col == 0) return BESS_OK;
    BESS_REQUIRE(scores && best_score && best_id, "topk_update: NULL pointer");
    BESS_REQUIRE(!ids || ids_rows == 1 || ids_rows == n_row, "topk_update: ids_rows must be 1 or n_row");
    BESS_REQUIRE(!mask || mask_rows == 1 || mask_rows == n_row, "topk_update: mask_rows must be 1 or n_row");
    // few rows: four waves per row keep enough loads in flight; many rows: one wave per row
    // (measured crossover on 256 CUs, profiles/bench_topk.py)
    const bool wide = n_row <= 6144;
    const bool vec = ld % 4 == 0 && reinterpret_cast<uintptr_t>(scores) % 16 == 0;
    const unsigned grid = static_cast<unsigned>(wide ? n_row : ceil_div(n_row, 4));
    hipStream_t st = as_stream(stream);
#define BESS_TOPK(WPR, VEC)                                                                                       \
    do {                                                                                                          \
        if (kk > 64)                                                                                              \
            k_topk_update<WPR, VEC, true><<<grid, 256, 0, st>>>(scores, n_row, n_col, ld, ids, ids_rows, id_base, \
                                                                mask, mask_rows, best_score, best_id, kk);        \
        else                                                                                                      \
            k_topk_update<WPR, VEC, false><<<grid, 256, 0, st>>>(scores, n_row, n_col, ld, ids, ids_rows, id_base, \
                                                                 mask, mask_rows, best_score, best_id, kk);       \
    } while (0)
    if (wide) {
        if (vec) BESS_TOPK(4, true);
        else BESS_TOPK(4, false);
    } else {
        if (vec) BESS_TOPK(1, true);
        else BESS_TOPK(1, false);
    }
#undef BESS_TOPK
    return check_launch("topk_update");
}

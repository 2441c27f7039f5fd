// K9 without atomics: segmented reduction of per-reference gradients into one
// gradient row per *unique* destination row of the shard, then K10 on those rows.
//
// The reference's autograd turns the gather `entity_embedding[idx]` into a dense
// zero-filled [M, W] gradient + `index_put_(accumulate)` (48 % + 17 % of its CPU
// step time).  A direct GPU transcription (one fp32 atomic
// per scalar of every gathered row) runs at the memory-side atomic rate
// (~1.3 TB/s), 4-5x below plain stores.  Here the references are grouped by
// destination row (an inverted index built with a stable radix sort, so the
// summation order - and therefore the result - is bitwise reproducible), each
// destination is summed in registers by one 16-lane DPP row, and written once.
//
// Per-triple negatives: the contribution of reference (q, k) to row
// e = neg_idx[q, k] is recomputed on the fly from query[q] (L2 / Infinity-Cache
// resident, S rows), d_out[q, k] and the row itself, so the [S*N, W] gradient of
// the gathered rows is never materialised:
//     DOT:  d e += g * query[q]
//     L1 :  d e += g * sgn(query[q] - e)        (score = -||q - e||_1)
//     L2 :  d e += g * (query[q] - e) / ||q - e||_2
#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>

#include "common.h"

namespace bess {

static inline size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

// ---- device-wide stable LSD radix sort of (row id, reference) pairs: kernels only ------------------------------
// rocPRIM's radix sort clears its histograms / look-back state with hipMemsetAsync.  Recorded into a hipGraph those
// become MEMSET nodes, and on this ROCm a recorded training step whose side branch interleaves memset nodes with
// the sort's kernels is not safe to replay: round 3 saw buffers that a recorded clear should have zeroed keep what
// the previous replay left (Adam moments of 1e20), round 4 a memory-aperture fault inside the Onesweep kernel
// that reads the look-back state those nodes clear (profiles/r04/graph_memset_probe.md).  Simple graphs replay
// their memset nodes correctly (same probe, experiment A) - the failure needs the step's topology - so nothing of
// this library puts a memset node into a graph any more: every clear is a kernel (common.h: fill_words_async) and
// the sort below is our own.  tests/test_graph_nodes.py pins "no memset node in any recorded step".
//
// One pass per RB-bit place (RB = ceil(row_bits / passes) <= 6: three passes for the 17 bits of BASELINE configs[1]'s
// shard, four for the 19 of configs[3]'s).  A WAVE owns a contiguous tile of up to 2048 keys.  The lanes that share
// a digit find each other with RB ballots (wave multisplit), so ranks inside a wave's 64 keys cost no atomics and
// keep input order - the sort is stable, equal rows keep reference order, sums stay bitwise reproducible.  The
// wave first sorts its tile by digit INTO LDS and then writes it out in that order: keys of one digit leave as
// runs of ~32 consecutive words (64 bins over 2048 keys).  A first version scattered every key straight to its
// place with 9-bit digits - 4 M isolated 4-byte stores per index of C2's million references, ~270 MB of write
// traffic that the HBM-bound forward kernel running beside the index build paid for with 30 us.
//   k_rsort_hist     per-wave digit histograms                                  H[digit][wave]
//   k_rsort_rowscan  one workgroup per digit: exclusive scan over the waves (kept next to the counts), the total
//   k_rsort_scatter  tile sorted by digit in LDS, written out run by run
constexpr int RS_WAVES = 4;   // waves per workgroup (each works alone)
constexpr int RS_MAX_BITS = 6;
constexpr int RS_TILE = 2048;  // keys per wave at most (2 x 8 KiB of LDS per wave)

struct RSortArgs {
    const int32_t* keys_in;
    const int32_t* vals_in;  // NULL: the value is the position (first pass)
    int32_t* keys_out;
    int32_t* vals_out;
    int32_t* hist;     // [1 << rb][n_wave] counts
    int32_t* scanned;  // [1 << rb][n_wave] their exclusive scan over the waves
    int32_t* totals;   // [1 << rb]
    int64_t n;
    int64_t tile;  // keys per wave (multiple of 64, <= RS_TILE)
    int n_wave;
    int shift, rb;
};

__device__ __forceinline__ uint64_t rsort_peers(int digit, int rb, bool valid) {
    uint64_t peers = __ballot(valid);
    for (int b = 0; b < rb; ++b) {
        const bool bit = (digit >> b) & 1;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return valid ? peers : 0ull;
}

__global__ __launch_bounds__(64 * RS_WAVES) void k_rsort_hist(RSortArgs a) {
    __shared__ int32_t cnt[RS_WAVES][1 << RS_MAX_BITS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int w = blockIdx.x * RS_WAVES + wv;
    const int nd = 1 << a.rb;
    if (w >= a.n_wave) return;  // (whole waves, no barrier in this kernel)
    cnt[wv][lane] = 0;
    const int64_t lo = w * a.tile, hi = min(lo + a.tile, a.n);
    const uint64_t below = (1ull << lane) - 1ull;
    int32_t knext = lo + lane < hi ? a.keys_in[lo + lane] : 0;  // one chunk of keys ahead
    for (int64_t i0 = lo; i0 < hi; i0 += 64) {
        const bool valid = i0 + lane < hi;
        const int32_t key = knext;
        if (i0 + 64 + lane < hi) knext = a.keys_in[i0 + 64 + lane];
        const int digit = (static_cast<uint32_t>(key) >> a.shift) & (nd - 1);
        const uint64_t peers = rsort_peers(digit, a.rb, valid);
        if (valid && (peers & below) == 0) cnt[wv][digit] += __popcll(peers);  // the lowest lane of each digit
    }
    if (lane < nd) a.hist[static_cast<int64_t>(lane) * a.n_wave + w] = cnt[wv][lane];
}

// workgroup d: scanned[d][0 .. n_wave) = exclusive scan of hist[d][.], totals[d] its sum
__global__ __launch_bounds__(256) void k_rsort_rowscan(const int32_t* __restrict__ hist, int32_t* __restrict__ scanned,
                                                       int32_t* __restrict__ totals, int n_wave) {
    __shared__ int32_t wsum[4];
    __shared__ int32_t carry_s;
    const int32_t* row = hist + static_cast<int64_t>(blockIdx.x) * n_wave;
    int32_t* out = scanned + static_cast<int64_t>(blockIdx.x) * n_wave;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n_wave; i0 += 256) {
        const int i = i0 + t;
        const int32_t v = i < n_wave ? row[i] : 0;
        int32_t inc = v;  // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t u = __shfl_up(inc, off, 64);
            if (lane >= off) inc += u;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int32_t before = carry_s;
        for (int k = 0; k < wv; ++k) before += wsum[k];
        if (i < n_wave) out[i] = before + inc - v;
        __syncthreads();
        if (t == 255) carry_s = before + inc;
        __syncthreads();
    }
    if (t == 0) totals[blockIdx.x] = carry_s;
}

__global__ __launch_bounds__(64 * RS_WAVES) void k_rsort_scatter(RSortArgs a) {
    __shared__ int32_t lk[RS_WAVES][RS_TILE], lv[RS_WAVES][RS_TILE];
    __shared__ int32_t off[RS_WAVES][1 << RS_MAX_BITS];     // running place of every digit inside the tile
    __shared__ int32_t delta[RS_WAVES][1 << RS_MAX_BITS];   // global place of a digit's first key - its place in the tile
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int w = blockIdx.x * RS_WAVES + wv;
    const int nd = 1 << a.rb;
    if (w >= a.n_wave) return;  // (whole waves: no barrier in this kernel; a wave's LDS accesses execute in order)
    {
        // lane d: digit d.  Place of the digit's run inside the tile (exclusive scan of the wave's counts over the
        // digits) and in the output (scan of the totals over the digits + scan of the counts over the waves)
        const int32_t c = lane < nd ? a.hist[static_cast<int64_t>(lane) * a.n_wave + w] : 0;
        const int32_t tot = lane < nd ? a.totals[lane] : 0;
        int32_t ci = c, ti = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t u = __shfl_up(ci, o, 64), v = __shfl_up(ti, o, 64);
            if (lane >= o) ci += u, ti += v;
        }
        const int32_t local = ci - c, base = ti - tot;
        off[wv][lane] = local;
        delta[wv][lane] = lane < nd ? base + a.scanned[static_cast<int64_t>(lane) * a.n_wave + w] - local : 0;
    }
    const int64_t lo = w * a.tile, hi = min(lo + a.tile, a.n);
    const int n_tile = static_cast<int>(hi - lo);
    const uint64_t below = (1ull << lane) - 1ull;
    int32_t knext = lo + lane < hi ? a.keys_in[lo + lane] : 0;
    int32_t vnext = (a.vals_in && lo + lane < hi) ? a.vals_in[lo + lane] : 0;
    for (int64_t i0 = lo; i0 < hi; i0 += 64) {
        const bool valid = i0 + lane < hi;
        const int32_t key = knext;
        const int32_t val = a.vals_in ? vnext : static_cast<int32_t>(i0 + lane);
        if (i0 + 64 + lane < hi) {
            knext = a.keys_in[i0 + 64 + lane];
            if (a.vals_in) vnext = a.vals_in[i0 + 64 + lane];
        }
        const int digit = (static_cast<uint32_t>(key) >> a.shift) & (nd - 1);
        const uint64_t peers = rsort_peers(digit, a.rb, valid);
        const int32_t at = valid ? off[wv][digit] : 0;  // (read by every lane of the digit before its lowest lane advances it)
        const int rank = __popcll(peers & below);
        if (valid) {
            lk[wv][at + rank] = key;
            lv[wv][at + rank] = val;
            if (rank == 0) off[wv][digit] = at + __popcll(peers);
        }
    }
    for (int i = lane; i < n_tile; i += 64) {  // the tile in digit order: runs of one digit go to consecutive places
        const int32_t key = lk[wv][i];
        const int digit = (static_cast<uint32_t>(key) >> a.shift) & (nd - 1);
        const int64_t at = static_cast<int64_t>(delta[wv][digit]) + i;
        a.keys_out[at] = key;
        a.vals_out[at] = lv[wv][i];
    }
}

// waves (= tiles) of a sort of n keys: tiles of RS_TILE keys (runs of ~32 keys per digit at the write-out), shorter
// ones (>= 512) for short lists
static void rsort_plan(int64_t n, int* n_wave, int64_t* tile) {
    int64_t t = std::min<int64_t>(RS_TILE, std::max<int64_t>(512, (n + 511) / 512));
    t = (t + 63) / 64 * 64;
    *tile = t;
    *n_wave = static_cast<int>((n + t - 1) / t);
}
static int rsort_passes(int bits) { return (bits + RS_MAX_BITS - 1) / RS_MAX_BITS; }
static size_t rsort_hist_bytes(int64_t n) {  // counts + scanned counts + totals
    int nw;
    int64_t tile;
    rsort_plan(n, &nw, &tile);
    return sizeof(int32_t) * (size_t(1) << RS_MAX_BITS) * (2 * static_cast<size_t>(nw) + 1);
}

// keys[n] (row ids < 2^bits) -> keys_sorted, refs_sorted (positions 0 .. n-1 in sorted order, stable).
// tmp_k / tmp_v: n int32 each; hist: rsort_hist_bytes(n).
static int radix_sort_refs(const int32_t* keys, int64_t n, int bits, int32_t* keys_sorted, int32_t* refs_sorted,
                           int32_t* tmp_k, int32_t* tmp_v, int32_t* hist, hipStream_t st) {
    const int passes = rsort_passes(bits);
    const int rb = (bits + passes - 1) / passes;
    RSortArgs a{};
    a.n = n;
    a.hist = hist;
    a.rb = rb;
    rsort_plan(n, &a.n_wave, &a.tile);
    a.scanned = hist + (size_t(1) << RS_MAX_BITS) * static_cast<size_t>(a.n_wave);
    a.totals = a.scanned + (size_t(1) << RS_MAX_BITS) * static_cast<size_t>(a.n_wave);
    const unsigned grid = static_cast<unsigned>((a.n_wave + RS_WAVES - 1) / RS_WAVES);
    // the last pass writes (keys_sorted, refs_sorted); the passes before alternate so that it does
    for (int p = 0; p < passes; ++p) {
        const bool to_final = ((passes - 1 - p) % 2) == 0;
        a.keys_in = p == 0 ? keys : (to_final ? tmp_k : keys_sorted);
        a.vals_in = p == 0 ? nullptr : (to_final ? tmp_v : refs_sorted);
        a.keys_out = to_final ? keys_sorted : tmp_k;
        a.vals_out = to_final ? refs_sorted : tmp_v;
        a.shift = p * rb;
        k_rsort_hist<<<grid, 64 * RS_WAVES, 0, st>>>(a);
        k_rsort_rowscan<<<1u << rb, 256, 0, st>>>(hist, a.scanned, a.totals, a.n_wave);
        k_rsort_scatter<<<grid, 64 * RS_WAVES, 0, st>>>(a);
    }
    return check_launch("radix sort of the references");
}

struct CubSizes {
    size_t sort, rle, scan, total_cub;
};

static hipError_t cub_sizes(int64_t n, CubSizes* cs) {
    int32_t* p = nullptr;
    size_t a = 0, b = 0, c = 0;
    hipError_t e = hipcub::DeviceRunLengthEncode::Encode(nullptr, b, p, p, p, p, static_cast<int>(n), 0);
    if (e != hipSuccess) return e;
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, c, p, p, static_cast<int>(n), 0);
    if (e != hipSuccess) return e;
    cs->sort = a;
    cs->rle = b;
    cs->scan = c;
    cs->total_cub = std::max(a, std::max(b, c));
    return hipSuccess;
}

// Small lists (heads, tails, shared negatives of a step: a few thousand references): the whole
// index - stable sort by row, unique rows, offsets, long-row list - in ONE workgroup and one
// launch.  The device-wide pipeline above is ~17 launches of 4-5 us each, which is all latency at
// this size (and sits between backward and update of a step with a stateful optimiser).
constexpr int SEG_CAP_FOR_SMALL = BESS_SEGMENT_CAP;
constexpr int SMALL_T = 1024, SMALL_N = SMALL_T * 15;  // up to 15,360 references (2, 4 or 15 per thread)

// The row ids to index may sit in several lists (heads, tails, negatives as the sampler handed them over):
// reference `at` of their concatenation is entry at - first[l] of list l - no concatenated copy is made.
struct IdLists {
    const int32_t* p[BESS_MAX_ROW_LISTS];
    int32_t first[BESS_MAX_ROW_LISTS + 1];  // first[l] = ids in lists 0 .. l-1; first[n ..] = total
    int n;
};
__device__ __forceinline__ int32_t id_at(const IdLists& L, int at) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < BESS_MAX_ROW_LISTS; ++k) l += (k < L.n && at >= L.first[k]) ? 1 : 0;
    return L.p[l][at - L.first[l]];
}

template <int SMALL_I>
struct SmallIndexTemp {
    typedef hipcub::BlockRadixSort<int32_t, SMALL_T, SMALL_I, int32_t> Sort;
    typedef hipcub::BlockDiscontinuity<int32_t, SMALL_T> Disc;
    typedef hipcub::BlockScan<int32_t, SMALL_T> Scan;
    union {
        typename Sort::TempStorage sort;
        typename Disc::TempStorage disc;
        typename Scan::TempStorage scan;
    } tmp;
    int32_t n_long;
};

// body of the one-workgroup index (SMALL_T threads, all of them call it)
template <int SMALL_I>
__device__ __forceinline__ void small_segment_index(SmallIndexTemp<SMALL_I>& sh, const IdLists& ids, int n,
                                                    int row_bits, int32_t* __restrict__ refs_sorted,
                                                    int32_t* __restrict__ seg_rows,
                                                    int32_t* __restrict__ seg_offsets, int32_t* __restrict__ n_seg,
                                                    int32_t* __restrict__ long_segs, int32_t long_cap) {
    typedef typename SmallIndexTemp<SMALL_I>::Sort Sort;
    typedef typename SmallIndexTemp<SMALL_I>::Disc Disc;
    typedef typename SmallIndexTemp<SMALL_I>::Scan Scan;
    const int t = threadIdx.x;
    int32_t key[SMALL_I], val[SMALL_I];
#pragma unroll
    for (int i = 0; i < SMALL_I; ++i) {
        const int at = t * SMALL_I + i;  // blocked arrangement; the tail sorts behind every real row
        key[i] = at < n ? id_at(ids, at) : static_cast<int32_t>(1u << row_bits);  // row ids are < 2^row_bits
        val[i] = at;
    }
    Sort(sh.tmp.sort).Sort(key, val, 0, row_bits + 1);  // LSD radix sort: stable
    __syncthreads();
    int32_t head[SMALL_I];
    Disc(sh.tmp.disc).FlagHeads(head, key, hipcub::Inequality());  // first item of the block is a head
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SMALL_I; ++i)
        if (t * SMALL_I + i >= n) head[i] = 0;
    int32_t sid[SMALL_I], total = 0;
    Scan(sh.tmp.scan).ExclusiveSum(head, sid, total);
#pragma unroll
    for (int i = 0; i < SMALL_I; ++i) {
        const int at = t * SMALL_I + i;
        if (at < n) {
            refs_sorted[at] = val[i];
            if (head[i]) {
                seg_rows[sid[i]] = key[i];
                seg_offsets[sid[i]] = at;
            }
        }
    }
    if (t == 0) {
        *n_seg = total;
        seg_offsets[total] = n;
        sh.n_long = 0;
    }
    if (!long_segs) return;
    __syncthreads();  // offsets written by this workgroup are visible to it
    for (int s2 = t; s2 < total; s2 += SMALL_T) {
        if (seg_offsets[s2 + 1] - seg_offsets[s2] > SEG_CAP_FOR_SMALL) {
            const int li = atomicAdd(&sh.n_long, 1);
            if (li < long_cap) long_segs[1 + li] = s2;
        }
    }
    __syncthreads();
    if (t == 0) long_segs[0] = sh.n_long;
}

template <int SMALL_I>
__global__ __launch_bounds__(SMALL_T) void k_small_segment_index(IdLists ids, int n, int row_bits,
                                                                 int32_t* __restrict__ refs_sorted,
                                                                 int32_t* __restrict__ seg_rows,
                                                                 int32_t* __restrict__ seg_offsets,
                                                                 int32_t* __restrict__ n_seg,
                                                                 int32_t* __restrict__ long_segs, int32_t long_cap) {
    __shared__ SmallIndexTemp<SMALL_I> sh;
    small_segment_index<SMALL_I>(sh, ids, n, row_bits, refs_sorted, seg_rows, seg_offsets, n_seg, long_segs, long_cap);
}

// Prologue of a notebook-size training step, ONE launch for what used to be a handful of 2-5 us dispatches
// in front of and between its kernels: workgroup 0 builds the index of the step's small update lists
// (their row ids are inputs of the step), the other workgroups run copy / fill jobs on 32-bit words - the
// concatenated candidate list of an augmented step, the zeroed relation gradient, the zeroed targets of the
// backward's atomics, ...
// (struct WordJobs, run_word_jobs, make_word_jobs: common.h - bess_query_triple_fwd_jobs runs the same jobs in the
// spare workgroups of the query / positive-score launch)

template <int SMALL_I>
__global__ __launch_bounds__(SMALL_T) void k_step_prologue(WordJobs J, IdLists ids, int n_ids, int row_bits,
                                                           int32_t* __restrict__ refs_sorted,
                                                           int32_t* __restrict__ seg_rows,
                                                           int32_t* __restrict__ seg_offsets,
                                                           int32_t* __restrict__ n_seg,
                                                           int32_t* __restrict__ long_segs, int32_t long_cap) {
    __shared__ SmallIndexTemp<SMALL_I> sh;
    const int index_blocks = n_ids > 0 ? 1 : 0;
    if (static_cast<int>(blockIdx.x) < index_blocks) {
        small_segment_index<SMALL_I>(sh, ids, n_ids, row_bits, refs_sorted, seg_rows, seg_offsets, n_seg, long_segs,
                                     long_cap);
        return;
    }
    run_word_jobs(J, static_cast<int>(blockIdx.x) - index_blocks, static_cast<int>(gridDim.x) - index_blocks, SMALL_T);
}

// seg_offsets[n_seg] = n_refs (close the last segment); ExclusiveSum wrote only n entries
// (and start the list of long segments that k_find_long_segments fills next at zero entries)
__global__ void k_close_offsets(int32_t* __restrict__ seg_offsets, const int32_t* __restrict__ n_seg,
                                int32_t n_refs, int32_t* __restrict__ long_segs) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        seg_offsets[*n_seg] = n_refs;
        if (long_segs) long_segs[0] = 0;
    }
}

// Row-sparse optimisers on the unique rows of a step ("lazy" semantics: state of
// untouched rows is left alone, like torch.optim.SparseAdam / Adagrad on sparse
// gradients).  state1/state2 are f32 [M, W] (momentum | sum of squares | exp_avg, exp_avg_sq).
struct OptArgs {
    int kind;
    float lr, momentum, beta1, beta2, eps, weight_decay;
    float bias1, bias2;           // 1 - beta^t for Adam (from the host's step count)
    const int32_t* step_ptr;      // optional: step count on the device, overrides bias1 / bias2
    const int32_t* slot_map;      // optional: state row of every table row (paged state), -1 = none
};

// Row of the state tables that belongs to table row `row`: the row itself, or - paged state - the slot
// bess_assign_state_rows gave it (-1 when the pool is exhausted: the row is then stepped from zero state
// and its state is not kept).
__device__ __forceinline__ int64_t state_row(const OptArgs& o, int64_t row) {
    return o.slot_map ? static_cast<int64_t>(o.slot_map[row]) : row;
}

// bias corrections from the device-side step count, when there is one
__device__ __forceinline__ OptArgs opt_resolve(OptArgs o) {
    if (o.kind == BESS_OPT_ADAM && o.step_ptr) {
        const float t = static_cast<float>(*o.step_ptr);
        o.bias1 = 1.f - powf(o.beta1, t);
        o.bias2 = 1.f - powf(o.beta2, t);
    }
    return o;
}

// one optimiser step on one scalar (p: parameter, g: summed gradient, s1 / s2: its state)
__device__ __forceinline__ void opt_step(const OptArgs& o, float& p, float g, float& s1, float& s2) {
    if (o.kind == BESS_OPT_SGD) {
        g += o.weight_decay * p;
        if (o.momentum != 0.f) {
            s1 = o.momentum * s1 + g;
            g = s1;
        }
        p -= o.lr * g;
    } else if (o.kind == BESS_OPT_ADAGRAD) {
        g += o.weight_decay * p;
        s1 += g * g;
        p -= o.lr * g / (sqrtf(s1) + o.eps);
    } else {  // BESS_OPT_ADAM (decoupled weight decay when weight_decay != 0: AdamW)
        p -= o.lr * o.weight_decay * p;
        s1 = o.beta1 * s1 + (1.f - o.beta1) * g;
        s2 = o.beta2 * s2 + (1.f - o.beta2) * g * g;
        p -= o.lr * sqrtf(o.bias2) / o.bias1 * s1 / (sqrtf(s2) + o.eps);
    }
}

// What the per-row pass does with the summed gradient of a row beyond "write it" / "plain SGD":
// a stateful optimiser applied in the same pass (kind >= 0), with the other contributions to the
// row (heads, tails, shared negatives ..., already summed per unique row in xsum) added first -
// xmap[seg] is the row of xsum that belongs to segment seg, or -1.
struct SegOpt {
    OptArgs o;        // o.kind < 0: off
    float* state1;
    float* state2;
    const int32_t* xmap;
    const float* xsum;
};

struct SegArgs {
    const float* query;
    const void* table;
    const float* d_out;
    int64_t ld_dout;
    const int32_t* refs;         // sorted reference ids (q * n_neg + k)
    const int32_t* seg_rows;     // destination row of each segment
    const int32_t* seg_offsets;  // [n_seg + 1]
    const int32_t* n_seg;        // device scalar
    int n_neg;
    int W;
    int nch;
    float sign;
    const int32_t* long_segs;    // optional: [0] = number of segments longer than SEG_CAP, [1..] = their ids
    SegOpt opt;
    float p;  // the norm of the RED_L2 kernels (any p != 1)
    // column windows that run side by side in ONE launch: workgroup b works on window b % n_win (workgroups go
    // to the XCDs round-robin, so with n_win dividing 8 an XCD only ever sees the windows b % n_win == xcd %
    // n_win and its L2 keeps just their slices of the query matrix), on segment groups b / n_win
    int n_win;
    int win_cols;  // scalars per window (nch * VEC)
};

// A row that collects very many references (padded candidate lists, a hot entity) would keep one
// 16-lane group busy for its whole length - 1 % of a million references on one row made the pass 30 x
// slower.  Segments longer than SEG_CAP are therefore left out of the per-row pass and handled by the
// whole grid: every group takes slices of SEG_CAP references, partial sums meet in a scratch row through
// float atomics, a last small kernel applies / stores the row.  (Rows below the cap stay on the
// atomic-free, bitwise reproducible path.)
constexpr int SEG_CAP = BESS_SEGMENT_CAP;

__global__ void k_find_long_segments(const int32_t* __restrict__ seg_offsets, const int32_t* __restrict__ n_seg,
                                     int32_t* __restrict__ long_segs, int32_t capacity) {
    const int n = *n_seg;
    for (int s = blockIdx.x * 256 + threadIdx.x; s < n; s += 256 * gridDim.x) {
        if (seg_offsets[s + 1] - seg_offsets[s] > SEG_CAP) {
            const int li = atomicAdd(&long_segs[0], 1);
            if (li < capacity) long_segs[1 + li] = s;
        }
    }
}

// acc += gradient contributions of references [r0, r1) (one destination row, held in `ev` for the
// distance scorers) - lanes of a 16-lane group stride the row.
// One reference = its query row (W floats) and one score gradient.  The references are visited two
// at a time with the loads of the next one issued before the arithmetic of the current one, and their
// ids are fetched two references ahead: before, every reference cost two dependent round trips (id,
// then query row) in front of its FMAs.
template <int VEC, int IT, int RED>
__device__ __forceinline__ void seg_accumulate(const SegArgs& a, int g, const float (&ev)[IT][VEC], int r0, int r1,
                                               float (&acc)[IT][VEC]) {
    auto fetch = [&](int ref, float (&qv)[IT][VEC], float& go) {
        // (references are non-negative: the unsigned division is half the instructions of the signed one)
        const unsigned q = static_cast<unsigned>(ref) / static_cast<unsigned>(a.n_neg);
        const unsigned k = static_cast<unsigned>(ref) - q * static_cast<unsigned>(a.n_neg);
        go = a.sign * a.d_out[q * a.ld_dout + k];
        const float* qp = a.query + static_cast<int64_t>(q) * a.W;
        // the p = 2 norm sums over the whole row (lanes past its end must hold zeros); the dot product and the
        // p = 1 distance are elementwise and the results of those lanes are never stored: no select needed
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            if (RED == RED_L2) load_chunk<float, VEC>(qp, g + 16 * it, a.nch, qv[it]);
            else load_chunk_clamped<float, VEC>(qp, g + 16 * it, a.nch, qv[it]);
        }
    };
    auto accumulate = [&](const float (&qv)[IT][VEC], float go) {
        if (RED == RED_L2) {
            float ss = 0.f;
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    ss += lp_term(qv[it][v] - ev[it][v], a.p);
                }
            ss = row16_allreduce_sum(ss);
            go *= lp_inv(lp_root(ss, a.p), a.p);
        }
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                if (RED == RED_DOT) acc[it][v] = fmaf(go, qv[it][v], acc[it][v]);
                else if (RED == RED_L1) acc[it][v] -= go * sgnf(qv[it][v] - ev[it][v]);
                else acc[it][v] = fmaf(-go, lp_dterm(qv[it][v] - ev[it][v], a.p), acc[it][v]);
            }
    };
    if (r0 >= r1) return;
    const int last = r1 - 1;
    float qa[IT][VEC], qb[IT][VEC], ga, gb;
    int ref_b = a.refs[min(r0 + 1, last)];
    fetch(a.refs[r0], qa, ga);
    int r = r0;
    for (; r + 2 <= r1; r += 2) {
        const int ref_a2 = a.refs[min(r + 2, last)];
        fetch(ref_b, qb, gb);
        accumulate(qa, ga);
        const int ref_b2 = a.refs[min(r + 3, last)];
        fetch(ref_a2, qa, ga);  // past the end: the last reference again, not accumulated
        accumulate(qb, gb);
        ref_b = ref_b2;
    }
    if (r < r1) accumulate(qa, ga);
}

// The summed gradient `acc` of a row (value `ev`) goes out: as a gradient row, as the fused plain
// SGD step, or through a stateful optimiser together with the row's other contributions.
// (Round 4 experiment, C2 + AdamW, where this pass takes 347 us against 177 us with plain SGD: a variant of the
// kernel that requests the row and its two state rows BEFORE the reduction over the row's references - 168 VGPRs, three
// waves per SIMD instead of four - took 353 us, the replayed step 0.831 ms against 0.809: the occupancy costs more than
// the hidden round trip gains; not kept.)
template <typename T, int VEC, int IT>
__device__ __forceinline__ void seg_finish(const SegArgs& a, int g, int64_t seg, int64_t row,
                                           const float (&ev)[IT][VEC], const float (&acc)[IT][VEC],
                                           float* __restrict__ grad_seg, T* table_rw, float lr) {
    if (a.opt.o.kind >= 0) {
        const OptArgs oo = opt_resolve(a.opt.o);
        // state rows are touched once per step: streamed past the caches (non-temporal), so that they do
        // not evict the query slice the column window keeps in L2
        const int x = a.opt.xmap ? a.opt.xmap[seg] : -1;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = g + 16 * it;
            if (c >= a.nch) continue;
            float xs[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) xs[v] = 0.f;
            if (x >= 0) load_chunk<float, VEC>(a.opt.xsum + static_cast<int64_t>(x) * a.W, c, a.nch, xs);
            typedef float fvec __attribute__((ext_vector_type(VEC)));
            const int64_t at = row * a.W + c * VEC;
            const int64_t srow = state_row(oo, row);
            const int64_t sat = srow * a.W + c * VEC;
            const bool has1 = a.opt.state1 && srow >= 0, has2 = a.opt.state2 && srow >= 0;
            fvec s1v = 0.f, s2v = 0.f;
            if (has1) s1v = __builtin_nontemporal_load(reinterpret_cast<const fvec*>(a.opt.state1 + sat));
            if (has2) s2v = __builtin_nontemporal_load(reinterpret_cast<const fvec*>(a.opt.state2 + sat));
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float p = ev[it][v], s1 = s1v[v], s2 = s2v[v];
                opt_step(oo, p, acc[it][v] + xs[v], s1, s2);
                table_rw[at + v] = static_cast<T>(p);
                s1v[v] = s1;
                s2v[v] = s2;
            }
            if (has1) __builtin_nontemporal_store(s1v, reinterpret_cast<fvec*>(a.opt.state1 + sat));
            if (has2) __builtin_nontemporal_store(s2v, reinterpret_cast<fvec*>(a.opt.state2 + sat));
        }
        return;
    }
    if (grad_seg) {
        float* out = grad_seg + seg * a.W;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = g + 16 * it;
            if (c < a.nch) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) out[c * VEC + v] = acc[it][v];
            }
        }
    } else {
        // each row is written once per step: streamed past the caches, like the optimiser state
        typedef T tvec __attribute__((ext_vector_type(VEC)));
        T* out = table_rw + row * a.W;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = g + 16 * it;
            if (c < a.nch) {
                tvec o;
#pragma unroll
                for (int v = 0; v < VEC; ++v) o[v] = static_cast<T>(ev[it][v] - lr * acc[it][v]);
                __builtin_nontemporal_store(o, reinterpret_cast<tvec*>(out + c * VEC));
            }
        }
    }
}

// grad_seg != NULL: write the per-row gradient.  grad_seg == NULL: apply SGD in
// place, table[row] -= lr * grad (each row is owned by exactly one 16-lane
// group, which read the old value before writing the new one).
template <typename T, int VEC, int IT, int RED>
__global__ __launch_bounds__(256) void k_pertriple_grad_segments(SegArgs a, float* __restrict__ grad_seg,
                                                                 T* table_rw, float lr) {
    const int lane = threadIdx.x & 63;
    const int g = lane & 15;
    const int n_seg = *a.n_seg;
    int64_t block = blockIdx.x, n_block = gridDim.x;
    if (a.n_win > 1) {
        const int64_t col0 = static_cast<int64_t>(block % a.n_win) * a.win_cols;
        block /= a.n_win;
        n_block /= a.n_win;
        a.query += col0;
        a.table = static_cast<const T*>(a.table) + col0;
        table_rw += col0;
        if (grad_seg) grad_seg += col0;
        if (a.opt.state1) a.opt.state1 += col0;
        if (a.opt.state2) a.opt.state2 += col0;
        if (a.opt.xsum) a.opt.xsum += col0;
    }
    const int64_t group0 = (block * 256ll + threadIdx.x) >> 4;
    const int64_t n_group = (n_block * 256ll) >> 4;
    const T* table = static_cast<const T*>(a.table);
    for (int64_t seg = group0; seg < n_seg; seg += n_group) {
        const int64_t row = a.seg_rows[seg];
        const int r0 = a.seg_offsets[seg], r1 = a.seg_offsets[seg + 1];
        if (a.long_segs && r1 - r0 > SEG_CAP) continue;  // left to k_long_segments_*
        float ev[IT][VEC], acc[IT][VEC];
        // the row itself: needed inside the loop by the distance scorers, only for the final
        // read-modify-write by the dot-product scorers (loaded late there: 32 registers less in the loop)
        constexpr bool EV_EARLY = RED != RED_DOT;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = g + 16 * it;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                ev[it][v] = 0.f;
                acc[it][v] = 0.f;
            }
            if (EV_EARLY) load_chunk<T, VEC>(table + row * a.W, c, a.nch, ev[it]);
        }
        seg_accumulate<VEC, IT, RED>(a, g, ev, r0, r1, acc);
        if (!EV_EARLY && grad_seg == nullptr) {
#pragma unroll
            for (int it = 0; it < IT; ++it) load_chunk<T, VEC>(table + row * a.W, g + 16 * it, a.nch, ev[it]);
        }
        seg_finish<T, VEC, IT>(a, g, seg, row, ev, acc, grad_seg, table_rw, lr);
    }
}

// Long segments: all groups share the slices of SEG_CAP references of every long segment; partial sums
// are added to long_grad[li, :] (all zero on entry).  The group that completes the last slice of a row
// (a counter per row, never reset: it grows by `parts` per pass) reads the sum back, puts it where the
// per-row pass would have (gradient row, or the fused SGD step) and leaves the scratch row zero again.
template <typename T, int VEC, int IT, int RED>
__global__ __launch_bounds__(256) void k_long_segments(SegArgs a, float* __restrict__ long_grad,
                                                       int32_t* __restrict__ long_cnt, int32_t capacity,
                                                       float* __restrict__ grad_seg, T* __restrict__ table_rw,
                                                       float lr) {
    const int lane = threadIdx.x & 63, g = lane & 15;
    const int64_t group0 = (blockIdx.x * 256ll + threadIdx.x) >> 4;
    const int64_t n_group = (gridDim.x * 256ll) >> 4;
    const int n_long = min(a.long_segs[0], capacity);
    const T* table = static_cast<const T*>(a.table);
    for (int li = 0; li < n_long; ++li) {
        const int seg = a.long_segs[1 + li];
        const int64_t row = a.seg_rows[seg];
        const int r0 = a.seg_offsets[seg], r1 = a.seg_offsets[seg + 1];
        const int parts = (r1 - r0 + SEG_CAP - 1) / SEG_CAP;
        if (group0 >= parts) continue;
        float ev[IT][VEC];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) ev[it][v] = 0.f;
            if (RED != RED_DOT || grad_seg == nullptr) load_chunk<T, VEC>(table + row * a.W, g + 16 * it, a.nch, ev[it]);
        }
        float* sum = long_grad + static_cast<int64_t>(li) * a.W;
        for (int64_t p = group0; p < parts; p += n_group) {
            float acc[IT][VEC];
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[it][v] = 0.f;
            const int rb = r0 + static_cast<int>(p) * SEG_CAP;
            seg_accumulate<VEC, IT, RED>(a, g, ev, rb, min(r1, rb + SEG_CAP), acc);
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = g + 16 * it;
                if (c < a.nch) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) atomicAdd(sum + c * VEC + v, acc[it][v]);
                }
            }
            __threadfence();  // this group's adds are performed before its slice is counted
            int old = 0;
            if (g == 0) old = atomicAdd(long_cnt + li, 1);
            old = __shfl(old, lane & 48, 64);  // from the first lane of the 16-lane group
            if ((old + 1) % parts != 0) continue;
            // the last arriver puts the counter back to zero: the scratch can serve the next launch as it is
            if (g == 0) __hip_atomic_store(long_cnt + li, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
            // last slice of the row: every partial sum is in (device-scope loads: past the L1)
            float tot[IT][VEC];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = g + 16 * it;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    tot[it][v] = 0.f;
                    if (c < a.nch) {
                        float* sp = sum + c * VEC + v;
                        tot[it][v] = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(sp, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            seg_finish<T, VEC, IT>(a, g, seg, row, ev, tot, grad_seg, table_rw, lr);
        }
    }
}

// table[seg_rows[s], :] -= lr * grad_seg[s, :]    (unique rows: plain read-modify-write)
template <typename T>
__global__ __launch_bounds__(256) void k_apply_segments(T* __restrict__ table, int W,
                                                        const int32_t* __restrict__ seg_rows,
                                                        const int32_t* __restrict__ n_seg,
                                                        const float* __restrict__ grad_seg, float lr) {
    const int64_t total = static_cast<int64_t>(*n_seg) * W;
    for (int64_t t = blockIdx.x * 256ll + threadIdx.x; t < total; t += 256ll * gridDim.x) {
        const int64_t s = t / W;
        const int c = static_cast<int>(t - s * W);
        T* p = table + static_cast<int64_t>(seg_rows[s]) * W + c;
        *p = static_cast<T>(static_cast<float>(*p) - lr * grad_seg[t]);
    }
}

// out[s, :] = sum over the references r of segment s of src[refs[r], :]   (generic K9:
// any list of (row, gradient row) contributions, e.g. heads + tails + shared negatives)
__global__ __launch_bounds__(256) void k_segment_sum_rows(const float* __restrict__ src, int W,
                                                          const int32_t* __restrict__ refs,
                                                          const int32_t* __restrict__ seg_offsets,
                                                          const int32_t* __restrict__ n_seg,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (blockIdx.x * 256ll + threadIdx.x) >> 6;
    const int64_t n_wave = (gridDim.x * 256ll) >> 6;
    const int ns = *n_seg;
    for (int64_t s = wave0; s < ns; s += n_wave) {
        const int r0 = seg_offsets[s], r1 = seg_offsets[s + 1];
        if ((W & 3) == 0) {  // 16 B per lane and load, four references in flight; summed in reference order
            for (int c = lane * 4; c < W; c += 256) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                for (int r = r0; r < r1; r += 4) {
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        v[u] = *reinterpret_cast<const float4*>(src + static_cast<int64_t>(refs[min(r + u, r1 - 1)]) * W + c);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (r + u < r1) {
                            acc[0] += v[u].x;
                            acc[1] += v[u].y;
                            acc[2] += v[u].z;
                            acc[3] += v[u].w;
                        }
                }
                *reinterpret_cast<float4*>(out + s * W + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            }
            continue;
        }
        for (int c = lane; c < W; c += 64) {
            float acc = 0.f;
            for (int r = r0; r < r1; ++r) acc += src[static_cast<int64_t>(refs[r]) * W + c];
            out[s * W + c] = acc;
        }
    }
}

// K9 + K10 for the small lists of a step (heads, tails, shared negatives, rows returned by C8 ...):
// the references of the lists' virtual concatenation are grouped by destination row
// (bess_build_segment_index over the concatenated row ids); one wave per unique row sums its
// contributions in reference order (fixed: bitwise reproducible) straight from the lists - no
// concatenated gradient, no [unique rows, W] buffer - and applies the optimiser: every touched
// row is read, updated and written ONCE (one rounding per step, also for f16 tables, where a
// packed-f16 atomic would round per contribution).
struct RowLists {
    const float* grad[BESS_MAX_ROW_LISTS];
    int32_t first[BESS_MAX_ROW_LISTS + 1];  // first[l] = references in lists 0 .. l-1
    int n;
};

// A dense `table2 += alpha * grad2` on a small replicated table (the relation table's plain SGD step) rides along
// in the last workgroups of the launch: one dispatch less per notebook-size step.
struct AxpyJob {
    void* table;
    const float* grad;
    int64_t n;
    float alpha;
    int blocks;  // workgroups at the end of the grid that do it (0: none)
};

template <typename T>
__global__ __launch_bounds__(256) void k_coalesced_update(OptArgs o, RowLists L, T* __restrict__ table, int W,
                                                          const int32_t* __restrict__ refs,
                                                          const int32_t* __restrict__ seg_rows,
                                                          const int32_t* __restrict__ seg_offsets,
                                                          const int32_t* __restrict__ n_seg,
                                                          float* __restrict__ state1, float* __restrict__ state2,
                                                          const int32_t* __restrict__ keep, float* __restrict__ sum_out,
                                                          AxpyJob x) {
    const int main_blocks = static_cast<int>(gridDim.x) - x.blocks;
    if (static_cast<int>(blockIdx.x) >= main_blocks) {
        T* t2 = static_cast<T*>(x.table);
        for (int64_t t = (blockIdx.x - main_blocks) * 256ll + threadIdx.x; t < x.n; t += 256ll * x.blocks)
            t2[t] = static_cast<T>(static_cast<float>(t2[t]) + x.alpha * x.grad[t]);
        return;
    }
    o = opt_resolve(o);
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (blockIdx.x * 256ll + threadIdx.x) >> 6;
    const int64_t n_wave = (main_blocks * 256ll) >> 6;
    const int ns = *n_seg;
    for (int64_t s = wave0; s < ns; s += n_wave) {
        if (keep && keep[s] == 0) continue;
        const int r0 = seg_offsets[s], r1 = seg_offsets[s + 1];
        const int64_t row = seg_rows[s];
        if ((W & 3) == 0) {
            // four scalars per lane and load, the rows of four references in flight (their pointers are wave-uniform:
            // resolved once per reference, not once per column pass); summed in reference order as before
            for (int c = lane * 4; c < W; c += 256) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                for (int r = r0; r < r1; r += 4) {
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ref = refs[min(r + u, r1 - 1)];
                        int l = 0;
#pragma unroll
                        for (int k = 1; k < BESS_MAX_ROW_LISTS; ++k) l += (k < L.n && ref >= L.first[k]) ? 1 : 0;
                        v[u] = *reinterpret_cast<const float4*>(L.grad[l] + static_cast<int64_t>(ref - L.first[l]) * W + c);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (r + u < r1) {
                            acc[0] += v[u].x;
                            acc[1] += v[u].y;
                            acc[2] += v[u].z;
                            acc[3] += v[u].w;
                        }
                }
                if (sum_out) {
                    *reinterpret_cast<float4*>(sum_out + s * W + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                    continue;
                }
                const int64_t srow = state_row(o, row);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t at = row * W + c + j;
                    const int64_t sat = srow * W + c + j;
                    float p = static_cast<float>(table[at]);
                    float s1 = (state1 && srow >= 0) ? state1[sat] : 0.f, s2 = (state2 && srow >= 0) ? state2[sat] : 0.f;
                    opt_step(o, p, acc[j], s1, s2);
                    if (state1 && srow >= 0) state1[sat] = s1;
                    if (state2 && srow >= 0) state2[sat] = s2;
                    table[at] = static_cast<T>(p);
                }
            }
            continue;
        }
        for (int c = lane; c < W; c += 64) {
            float acc = 0.f;
            for (int r = r0; r < r1; ++r) {
                const int ref = refs[r];
                int l = 0;
#pragma unroll
                for (int k = 1; k < BESS_MAX_ROW_LISTS; ++k) l += (k < L.n && ref >= L.first[k]) ? 1 : 0;
                acc += L.grad[l][static_cast<int64_t>(ref - L.first[l]) * W + c];
            }
            if (sum_out) {
                sum_out[s * W + c] = acc;
                continue;
            }
            const int64_t at = row * W + c;
            const int64_t srow = state_row(o, row);
            const int64_t sat = srow * W + c;
            float p = static_cast<float>(table[at]);
            float s1 = (state1 && srow >= 0) ? state1[sat] : 0.f, s2 = (state2 && srow >= 0) ? state2[sat] : 0.f;
            opt_step(o, p, acc, s1, s2);
            if (state1 && srow >= 0) state1[sat] = s1;
            if (state2 && srow >= 0) state2[sat] = s2;
            table[at] = static_cast<T>(p);
        }
    }
}

// Paged optimiser state: the state tables hold `capacity` rows; a table row gets one when it is first
// stepped (slot_map[row] = its state row, -1 before).  One thread per unique row of the step; rows are
// unique, so no two threads assign the same row.  counter[0] counts the assignments asked for - beyond
// `capacity` a row stays at -1 (the host reads counter[0] > capacity to notice an exhausted pool; the count
// saturates a little above capacity: at most one launch's worth of threads beyond it).
__global__ __launch_bounds__(256) void k_assign_state_rows(const int32_t* __restrict__ seg_rows,
                                                           const int32_t* __restrict__ n_seg,
                                                           const int32_t* __restrict__ keep,
                                                           int32_t* __restrict__ slot_map, int32_t* __restrict__ counter,
                                                           int32_t capacity) {
    const int ns = *n_seg;
    for (int64_t s = blockIdx.x * 256ll + threadIdx.x; s < ns; s += 256ll * gridDim.x) {
        if (keep && keep[s] == 0) continue;
        const int64_t row = seg_rows[s];
        if (slot_map[row] >= 0) continue;
        // saturating: once the pool is exhausted (counter > capacity tells the host so) rows without state
        // stop counting - they ask again every step, and a counter that kept growing would wrap after 2^31
        // requests and hand the slots out a second time
        if (__atomic_load_n(counter, __ATOMIC_RELAXED) > capacity) continue;
        const int slot = atomicAdd(counter, 1);
        if (slot >= 0 && slot < capacity) slot_map[row] = slot;
    }
}

// keep (optional): only segments with keep[s] != 0 are updated
template <typename T>
__global__ __launch_bounds__(256) void k_apply_segments_opt(OptArgs o, T* __restrict__ table, int W,
                                                            const int32_t* __restrict__ seg_rows,
                                                            const int32_t* __restrict__ n_seg,
                                                            const float* __restrict__ grad_seg,
                                                            float* __restrict__ state1,
                                                            float* __restrict__ state2,
                                                            const int32_t* __restrict__ keep) {
    o = opt_resolve(o);
    const int64_t total = static_cast<int64_t>(*n_seg) * W;
    for (int64_t t = blockIdx.x * 256ll + threadIdx.x; t < total; t += 256ll * gridDim.x) {
        const int64_t s = t / W;
        if (keep && keep[s] == 0) continue;
        const int c = static_cast<int>(t - s * W);
        const int64_t row = seg_rows[s];
        const int64_t at = row * W + c;
        const int64_t srow = state_row(o, row);
        const int64_t sat = srow * W + c;
        float p = static_cast<float>(table[at]);
        float s1 = (state1 && srow >= 0) ? state1[sat] : 0.f, s2 = (state2 && srow >= 0) ? state2[sat] : 0.f;
        opt_step(o, p, grad_seg[t], s1, s2);
        if (state1 && srow >= 0) state1[sat] = s1;
        if (state2 && srow >= 0) state2[sat] = s2;
        table[at] = static_cast<T>(p);
    }
}

// xmap[s] = x for every unique extra row x that is also a segment s of the big index (binary search in
// its ascending seg_rows), keep[x] = 1 for the extra rows that are not (they get their own update)
__global__ __launch_bounds__(256) void k_map_extra_rows(const int32_t* __restrict__ seg_rows,
                                                        const int32_t* __restrict__ n_seg,
                                                        const int32_t* __restrict__ xrows,
                                                        const int32_t* __restrict__ n_x,
                                                        int32_t* __restrict__ xmap, int32_t* __restrict__ keep) {
    const int nx = *n_x, ns = *n_seg;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < nx; x += 256 * gridDim.x) {
        const int r = xrows[x];
        int lo = 0, hi = ns;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (seg_rows[mid] < r) lo = mid + 1;
            else hi = mid;
        }
        const bool found = lo < ns && seg_rows[lo] == r;
        if (found) xmap[lo] = x;
        keep[x] = found ? 0 : 1;
    }
}

template <typename T, int VEC, int IT>
static void seg_by_red(int red, const SegArgs& a, float* grad_seg, void* table_rw, float lr, unsigned grid,
                       hipStream_t st, float* long_grad = nullptr, int32_t* long_cnt = nullptr, int32_t long_cap = 0) {
    T* rw = static_cast<T*>(table_rw);
    if (long_grad) {  // the slices of the long segments, all groups together
        if (red == RED_DOT)
            k_long_segments<T, VEC, IT, RED_DOT><<<grid, 256, 0, st>>>(a, long_grad, long_cnt, long_cap, grad_seg, rw, lr);
        else if (red == RED_L1)
            k_long_segments<T, VEC, IT, RED_L1><<<grid, 256, 0, st>>>(a, long_grad, long_cnt, long_cap, grad_seg, rw, lr);
        else
            k_long_segments<T, VEC, IT, RED_L2><<<grid, 256, 0, st>>>(a, long_grad, long_cnt, long_cap, grad_seg, rw, lr);
        return;
    }
    if (red == RED_DOT) k_pertriple_grad_segments<T, VEC, IT, RED_DOT><<<grid, 256, 0, st>>>(a, grad_seg, rw, lr);
    else if (red == RED_L1) k_pertriple_grad_segments<T, VEC, IT, RED_L1><<<grid, 256, 0, st>>>(a, grad_seg, rw, lr);
    else k_pertriple_grad_segments<T, VEC, IT, RED_L2><<<grid, 256, 0, st>>>(a, grad_seg, rw, lr);
}

template <typename T, int VEC>
static int seg_by_it(int it, int red, const SegArgs& a, float* grad_seg, void* rw, float lr, unsigned grid,
                     hipStream_t st, float* long_grad = nullptr, int32_t* long_cnt = nullptr, int32_t long_cap = 0) {
    if (it <= 1) seg_by_red<T, VEC, 1>(red, a, grad_seg, rw, lr, grid, st, long_grad, long_cnt, long_cap);
    else if (it <= 2) seg_by_red<T, VEC, 2>(red, a, grad_seg, rw, lr, grid, st, long_grad, long_cnt, long_cap);
    else if (it <= 4) seg_by_red<T, VEC, 4>(red, a, grad_seg, rw, lr, grid, st, long_grad, long_cnt, long_cap);
    else if (it <= 8) seg_by_red<T, VEC, 8>(red, a, grad_seg, rw, lr, grid, st, long_grad, long_cnt, long_cap);
    else if (it <= 16) seg_by_red<T, VEC, 16>(red, a, grad_seg, rw, lr, grid, st, long_grad, long_cnt, long_cap);
    else return fail(BESS_EUNSUPPORTED, "grad_segments: row of %d scalars too wide", a.W);
    return BESS_OK;
}

// ---- K9 + K10 without an index: direct-addressed accumulation --------------------------------------------------
// For shards whose fp32 image fits a scratch budget, the backward kernels ADD their gradient rows straight into
// acc[row id] (an [M, W] fp32 matrix that is zero between steps; fp32 atomics), so nothing has to be sorted to
// find the rows of a step: one wave per REFERENCE of the step's row-id lists claims its row (atomic exchange of
// the step's generation number into claim[row]: the first reference of a row wins, the others see the generation
// already there and leave), reads the summed gradient, applies the optimiser - one read-modify-write and ONE
// rounding of the row per step, whatever the optimiser - and leaves the acc row zero again.  What the notebook-size
// steps paid for the sorted index (12 of 75 us, profiles/r03/step_r03_c4g.txt) is gone, and so are the dense
// [n, W] gradient arrays of heads / tails / shared negatives and their clears.
template <typename T>
__global__ __launch_bounds__(256) void k_direct_update(OptArgs o, IdLists ids, int n_ids, T* __restrict__ table, int W,
                                                       float* __restrict__ acc, int32_t* __restrict__ claim,
                                                       const int32_t* __restrict__ generation,
                                                       float* __restrict__ state1, float* __restrict__ state2,
                                                       AxpyJob x) {
    const int main_blocks = static_cast<int>(gridDim.x) - x.blocks;
    if (static_cast<int>(blockIdx.x) >= main_blocks) {
        T* t2 = static_cast<T*>(x.table);
        for (int64_t t = (blockIdx.x - main_blocks) * 256ll + threadIdx.x; t < x.n; t += 256ll * x.blocks)
            t2[t] = static_cast<T>(static_cast<float>(t2[t]) + x.alpha * x.grad[t]);
        return;
    }
    o = opt_resolve(o);
    const int lane = threadIdx.x & 63;
    const int gen = *generation;
    const int64_t wave0 = (blockIdx.x * 256ll + threadIdx.x) >> 6;
    const int64_t n_wave = (main_blocks * 256ll) >> 6;
    for (int64_t ref = wave0; ref < n_ids; ref += n_wave) {
        const int64_t row = id_at(ids, static_cast<int>(ref));
        int seen = 0;
        if (lane == 0) seen = atomicExch(claim + row, gen);
        seen = __shfl(seen, 0, 64);
        if (seen == gen) continue;  // an earlier reference of this row has taken it
        float* a = acc + row * W;
        T* p = table + row * W;
        float* s1 = state1 ? state1 + row * W : nullptr;
        float* s2 = state2 ? state2 + row * W : nullptr;
        if ((W & 3) == 0) {
            for (int c = lane * 4; c < W; c += 256) {
                const float4 g4 = *reinterpret_cast<const float4*>(a + c);
                *reinterpret_cast<float4*>(a + c) = make_float4(0.f, 0.f, 0.f, 0.f);
                const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = static_cast<float>(p[c + j]);
                    float m1 = s1 ? s1[c + j] : 0.f, m2 = s2 ? s2[c + j] : 0.f;
                    opt_step(o, v, g[j], m1, m2);
                    if (s1) s1[c + j] = m1;
                    if (s2) s2[c + j] = m2;
                    p[c + j] = static_cast<T>(v);
                }
            }
        } else {
            for (int c = lane; c < W; c += 64) {
                const float g = a[c];
                a[c] = 0.f;
                float v = static_cast<float>(p[c]);
                float m1 = s1 ? s1[c] : 0.f, m2 = s2 ? s2[c] : 0.f;
                opt_step(o, v, g, m1, m2);
                if (s1) s1[c] = m1;
                if (s2) s2[c] = m2;
                p[c] = static_cast<T>(v);
            }
        }
    }
}

int make_word_jobs(int32_t n_jobs, void* const* job_dst, const void* const* job_src, const uint32_t* job_value,
                   const int64_t* job_words, WordJobs* J, int64_t* words, const char* who) {
    BESS_REQUIRE(n_jobs >= 0 && n_jobs <= BESS_MAX_WORD_JOBS, "%s: %d jobs (0 .. %d)", who, n_jobs, BESS_MAX_WORD_JOBS);
    BESS_REQUIRE(n_jobs == 0 || (job_dst && job_src && job_value && job_words), "%s: NULL job arrays", who);
    *J = WordJobs{};
    J->n = n_jobs;
    int64_t w = 0;
    for (int j = 0; j < n_jobs; ++j) {
        BESS_REQUIRE(job_words[j] >= 0 && (job_words[j] == 0 || job_dst[j]), "%s: job %d", who, j);
        BESS_REQUIRE(reinterpret_cast<uintptr_t>(job_dst[j]) % 4 == 0 && reinterpret_cast<uintptr_t>(job_src[j]) % 4 == 0,
                     "%s: job %d is not 4-byte aligned", who, j);
        J->dst[j] = static_cast<uint32_t*>(job_dst[j]);
        J->src[j] = static_cast<const uint32_t*>(job_src[j]);
        J->value[j] = job_value[j];
        J->first[j] = w;
        w += job_words[j];
    }
    for (int j = n_jobs; j <= BESS_MAX_WORD_JOBS; ++j) J->first[j] = w;
    *words = w;
    return BESS_OK;
}

}  // namespace bess

using namespace bess;

extern "C" int bess_segment_index_workspace(int64_t n_refs, size_t* bytes) {
    BESS_REQUIRE(bytes, "segment_index_workspace: NULL out");
    BESS_REQUIRE(n_refs >= 0 && n_refs < (1ll << 31), "segment_index_workspace: n_refs out of range");
    CubSizes cs;
    hipError_t e = cub_sizes(n_refs > 0 ? n_refs : 1, &cs);
    if (e != hipSuccess) return fail(static_cast<int>(e), "hipcub size query: %s", hipGetErrorString(e));
    // [sorted keys][sort scratch: keys][sort scratch: references][counts][digit histograms][cub temp]
    *bytes = 4 * align_up(sizeof(int32_t) * static_cast<size_t>(n_refs)) + align_up(rsort_hist_bytes(n_refs > 0 ? n_refs : 1)) +
             align_up(cs.total_cub) + 256;
    return BESS_OK;
}

extern "C" int bess_build_segment_index(const int32_t* idx, int64_t n_refs, int32_t row_bits,
                                        int32_t* refs_sorted, int32_t* seg_rows, int32_t* seg_offsets,
                                        int32_t* n_seg, int32_t* long_segs, int64_t long_cap, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    BESS_REQUIRE(!long_segs || long_cap >= n_refs / SEG_CAP + 1, "build_segment_index: long_cap < n_refs / %d + 1",
                 SEG_CAP);
    BESS_REQUIRE(n_refs > 0 && n_refs < (1ll << 31), "build_segment_index: n_refs out of range");
    BESS_REQUIRE(idx && refs_sorted && seg_rows && seg_offsets && n_seg && workspace, "build_segment_index: NULL pointer");
    BESS_REQUIRE(row_bits >= 1 && row_bits <= 31, "build_segment_index: row_bits out of range");
    size_t need = 0;
    if (int e = bess_segment_index_workspace(n_refs, &need)) return e;
    BESS_REQUIRE(workspace_bytes >= need, "build_segment_index: workspace of %zu bytes, need %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    const int n = static_cast<int>(n_refs);
    if (n_refs <= SMALL_N && row_bits <= 30) {
        const int32_t lc = static_cast<int32_t>(long_cap);
        IdLists ids{};
        ids.n = 1;
        ids.p[0] = idx;
        for (int l = 1; l <= BESS_MAX_ROW_LISTS; ++l) ids.first[l] = n;
        if (n <= SMALL_T * 2)
            k_small_segment_index<2><<<1, SMALL_T, 0, st>>>(ids, n, row_bits, refs_sorted, seg_rows, seg_offsets, n_seg, long_segs, lc);
        else if (n <= SMALL_T * 4)
            k_small_segment_index<4><<<1, SMALL_T, 0, st>>>(ids, n, row_bits, refs_sorted, seg_rows, seg_offsets, n_seg, long_segs, lc);
        else
            k_small_segment_index<15><<<1, SMALL_T, 0, st>>>(ids, n, row_bits, refs_sorted, seg_rows, seg_offsets, n_seg, long_segs, lc);
        return check_launch("build_segment_index (small)");
    }
    char* ws = static_cast<char*>(workspace);
    const size_t blk = align_up(sizeof(int32_t) * static_cast<size_t>(n_refs));
    int32_t* keys_sorted = reinterpret_cast<int32_t*>(ws);
    int32_t* tmp_k = reinterpret_cast<int32_t*>(ws + blk);
    int32_t* tmp_v = reinterpret_cast<int32_t*>(ws + 2 * blk);
    int32_t* counts = reinterpret_cast<int32_t*>(ws + 3 * blk);
    int32_t* hist = reinterpret_cast<int32_t*>(ws + 4 * blk);
    const size_t hist_bytes = align_up(rsort_hist_bytes(n_refs));
    void* cub_tmp = ws + 4 * blk + hist_bytes;
    size_t cub_bytes = workspace_bytes - 4 * blk - hist_bytes;
    // stable LSD radix sort on the significant bits only: equal rows keep reference order (own kernels: no memset
    // node when the step is recorded into a hipGraph)
    if (int rc = radix_sort_refs(idx, n_refs, row_bits, keys_sorted, refs_sorted, tmp_k, tmp_v, hist, st)) return rc;
    hipError_t e;
    e = hipcub::DeviceRunLengthEncode::Encode(cub_tmp, cub_bytes, keys_sorted, seg_rows, counts, n_seg, n, st);
    if (e != hipSuccess) return fail(static_cast<int>(e), "run-length encode: %s", hipGetErrorString(e));
    // counts beyond n_seg are undefined but never read: offsets are consumed up to n_seg only
    e = hipcub::DeviceScan::ExclusiveSum(cub_tmp, cub_bytes, counts, seg_offsets, n, st);
    if (e != hipSuccess) return fail(static_cast<int>(e), "exclusive scan: %s", hipGetErrorString(e));
    k_close_offsets<<<1, 64, 0, st>>>(seg_offsets, n_seg, static_cast<int32_t>(n_refs), long_segs);
    if (long_segs) {
        k_find_long_segments<<<static_cast<unsigned>(std::min<int64_t>(ceil_div(n_refs, 256), 1024)), 256, 0, st>>>(
            seg_offsets, n_seg, long_segs, static_cast<int32_t>(long_cap));
    }
    return check_launch("build_segment_index");
}

// Entries [n_seg, max_seg) of a segment list become copies of its last real row with a zero gradient: the list
// then has a length the host knows (max_seg) and can be handed on as an ordinary (row ids, gradient rows) list -
// a duplicate with a zero contribution changes no row's total - without reading n_seg back (no host
// synchronisation, recordable into a hipGraph).
__global__ __launch_bounds__(256) void k_pad_segments(int32_t* __restrict__ seg_rows, const int32_t* __restrict__ n_seg,
                                                      int64_t max_seg, float* __restrict__ grad, int W) {
    const int64_t ns = *n_seg;
    if (ns >= max_seg || ns <= 0) return;
    const int32_t last = seg_rows[ns - 1];
    const int64_t total = (max_seg - ns) * W;
    for (int64_t t = blockIdx.x * 256ll + threadIdx.x; t < total; t += 256ll * gridDim.x) {
        const int64_t s = ns + t / W;
        grad[s * W + t % W] = 0.f;
        if (t % W == 0) seg_rows[s] = last;
    }
}

extern "C" int bess_pad_segments(int32_t* seg_rows, const int32_t* n_seg, int64_t max_seg, float* grad_seg,
                                 int32_t width, void* stream) {
    BESS_REQUIRE(seg_rows && n_seg && grad_seg, "pad_segments: NULL pointer");
    BESS_REQUIRE(max_seg > 0 && width > 0, "pad_segments: bad sizes");
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(ceil_div(max_seg * width, 256 * 8), 2048));
    k_pad_segments<<<grid < 1 ? 1 : grid, 256, 0, as_stream(stream)>>>(seg_rows, n_seg, max_seg, grad_seg, width);
    return check_launch("pad_segments");
}

extern "C" int bess_step_prologue(int32_t n_jobs, void* const* job_dst, const void* const* job_src,
                                  const uint32_t* job_value, const int64_t* job_words, int32_t n_lists,
                                  const int32_t* const* id_lists, const int64_t* id_lens, int32_t row_bits,
                                  int32_t* refs_sorted, int32_t* seg_rows, int32_t* seg_offsets, int32_t* n_seg,
                                  int32_t* long_segs, int64_t long_cap, void* stream) {
    BESS_REQUIRE(n_lists >= 0 && n_lists <= BESS_MAX_ROW_LISTS, "step_prologue: %d id lists (0 .. %d)", n_lists,
                 BESS_MAX_ROW_LISTS);
    WordJobs J{};
    int64_t words = 0;
    if (int e = make_word_jobs(n_jobs, job_dst, job_src, job_value, job_words, &J, &words, "step_prologue")) return e;
    IdLists ids{};
    ids.n = n_lists;
    int64_t n_ids = 0;
    if (n_lists > 0) {
        BESS_REQUIRE(id_lists && id_lens && refs_sorted && seg_rows && seg_offsets && n_seg, "step_prologue: NULL index pointer");
        for (int l = 0; l < n_lists; ++l) {
            BESS_REQUIRE(id_lens[l] >= 0 && (id_lens[l] == 0 || id_lists[l]), "step_prologue: id list %d", l);
            ids.p[l] = id_lists[l];
            ids.first[l] = static_cast<int32_t>(n_ids);
            n_ids += id_lens[l];
        }
        BESS_REQUIRE(n_ids > 0 && n_ids <= SMALL_N, "step_prologue: %lld row ids (1 .. %d: larger lists go through "
                     "bess_build_segment_index)", static_cast<long long>(n_ids), SMALL_N);
        for (int l = n_lists; l <= BESS_MAX_ROW_LISTS; ++l) ids.first[l] = static_cast<int32_t>(n_ids);
        BESS_REQUIRE(row_bits >= 1 && row_bits <= 30, "step_prologue: row_bits out of range");
        BESS_REQUIRE(!long_segs || long_cap >= n_ids / SEG_CAP + 1, "step_prologue: long_cap < n_ids / %d + 1", SEG_CAP);
    }
    if (words == 0 && n_ids == 0) return BESS_OK;
    const int n = static_cast<int>(n_ids);
    const int32_t lc = static_cast<int32_t>(long_cap);
    // ~4 words per thread of the job workgroups, at most one workgroup per CU
    const unsigned grid = static_cast<unsigned>((n > 0 ? 1 : 0) + std::min<int64_t>(ceil_div(words, 4 * SMALL_T), 256));
    hipStream_t st = as_stream(stream);
    if (n <= SMALL_T * 2)
        k_step_prologue<2><<<grid, SMALL_T, 0, st>>>(J, ids, n, row_bits, refs_sorted, seg_rows, seg_offsets, n_seg, long_segs, lc);
    else if (n <= SMALL_T * 4)
        k_step_prologue<4><<<grid, SMALL_T, 0, st>>>(J, ids, n, row_bits, refs_sorted, seg_rows, seg_offsets, n_seg, long_segs, lc);
    else
        k_step_prologue<15><<<grid, SMALL_T, 0, st>>>(J, ids, n, row_bits, refs_sorted, seg_rows, seg_offsets, n_seg, long_segs, lc);
    return check_launch("step_prologue");
}

static int grad_segments_impl(const bess_model_desc* d, const float* query, int64_t n_query, void* table,
                              int64_t n_neg, const float* d_out, int64_t ld_dout, const int32_t* refs_sorted,
                              const int32_t* seg_rows, const int32_t* seg_offsets, const int32_t* n_seg,
                              int64_t max_seg, float* grad_seg, float fused_sgd_lr, const int32_t* long_segs,
                              int64_t long_cap, float* long_grad, int32_t* long_count, const SegOpt& opt,
                              void* stream) {
    if (int e = check_desc(d)) return e;
    BESS_REQUIRE(!long_segs || (long_grad && long_count && long_cap >= n_query * n_neg / SEG_CAP + 1),
                 "grad_segments: long_segs needs long_grad, long_count and long_cap >= n_refs / %d + 1", SEG_CAP);
    BESS_REQUIRE(n_query > 0 && n_neg > 0 && n_query * n_neg < (1ll << 31), "grad_segments: bad sizes");
    BESS_REQUIRE(query && table && d_out && refs_sorted && seg_rows && seg_offsets && n_seg,
                 "grad_segments: NULL pointer");
    BESS_REQUIRE(ld_dout >= n_neg && max_seg > 0, "grad_segments: bad leading dimension / max_seg");
    BESS_REQUIRE(opt.o.kind < 0 || d->scorer <= BESS_COMPLEX,
                 "step_segments: the fused optimiser step exists for TransE / RotatE / DistMult / ComplEx");
    if (d->scorer == BESS_AFFINE)
        return affine_grad_segments(d, query, table, n_neg, d_out, ld_dout, refs_sorted, seg_rows, seg_offsets, n_seg,
                                    max_seg, grad_seg, fused_sgd_lr, long_segs, long_cap, long_grad, long_count,
                                    as_stream(stream));
    if (d->scorer == BESS_BOXE)
        return boxe_grad_segments(d, query, table, n_neg, d_out, ld_dout, refs_sorted, seg_rows, seg_offsets, n_seg,
                                  max_seg, grad_seg, fused_sgd_lr, long_segs, long_cap, long_grad, long_count,
                                  as_stream(stream));
    BESS_REQUIRE(d->scorer <= BESS_COMPLEX, "grad_segments: scorer %d has no segmented form", d->scorer);
    const int W = d->width;
    const int maxvec = d->dtype == BESS_F32 ? 4 : 8;
    int vec = maxvec;
    if (W % vec) vec = (d->dtype == BESS_F16 && W % 2 == 0) ? 2 : 1;
    const int red = reduce_of(d);
    // 16 segments per 256-thread workgroup; grid-stride over the (device-side) segment count
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(ceil_div(max_seg, 16), 256 * 16));
    hipStream_t st = as_stream(stream);
    // Every reference re-reads its query row, so the pass streams n_ref * W floats out of an
    // [n_query, W] matrix: 8 MiB for 4096 x 512, twice an XCD's L2 - the rows then come from the
    // Infinity Cache at its ~9 TB/s.  The gradient is elementwise in the column (except the p = 2
    // norm), so the pass is cut into column windows whose slice of the query matrix is at most the
    // 4 MiB of an L2; each window re-reads only the 8 bytes of ids / score gradients per reference.
    // Measured on 4096 x 512 (training step, ms): one pass 0.734, 2 x 256 columns 0.690, 3 windows
    // 0.701, 4 windows 0.707, 8 windows 0.806 - the per-window fixed cost sets the optimum.
    const int unit = vec * 16;  // columns one 16-lane group covers per iteration
    int win = W;
    int n_conc = 1;  // windows of one launch (side by side, mapped to XCDs: SegArgs.n_win)
    if (red != RED_L2 && n_query * static_cast<int64_t>(W) * 4 > (4ll << 20)) {
        const int64_t fit = (4ll << 20) / (n_query * 4);
        if (fit >= unit) win = static_cast<int>(fit / unit) * unit;
        // two windows: side by side in ONE launch (workgroup b works on window b % 2: an XCD's L2 then holds only
        // its own window's query slice) - 1 % faster than one after the other (0.637 vs 0.644 ms per C2 training
        // step); four or eight concurrent windows are no better / slower (0.643 / 0.752: the ids and score
        // gradients are re-read per window, the rows per reference get short)
        if (win * 2 == W && W % (2 * unit) == 0) {
            n_conc = 2;
            win = W;
        }
    }
    // ... and a window is at most what a 16-lane group keeps in registers (16 iterations): rows wider than that
    // (1024 f32 / 2048 f16 scalars) are windowed for this reason alone; the p = 2 norm needs the whole row
    if (red != RED_L2 && win > 16 * unit) win = 16 * unit;
    const int64_t sz = d->dtype == BESS_F32 ? 4 : 2;
    for (int col0 = 0; col0 < W; col0 += win) {
        const int cols = W - col0 < win ? W - col0 : win;
        char* tab = static_cast<char*>(table) + col0 * sz;
        SegOpt wopt = opt;  // the window's columns of the state tables and of the extra gradients
        if (wopt.state1) wopt.state1 += col0;
        if (wopt.state2) wopt.state2 += col0;
        if (wopt.xsum) wopt.xsum += col0;
        SegArgs a{query + col0, tab, d_out, ld_dout, refs_sorted, seg_rows, seg_offsets, n_seg,
                  static_cast<int>(n_neg), W, cols / vec / n_conc, is_distance(d->scorer) ? -1.f : 1.f, long_segs, wopt,
                  static_cast<float>(d->norm_p), n_conc, cols / n_conc};
        const int it = static_cast<int>(ceil_div(a.nch, 16));
        float* gs = grad_seg ? grad_seg + col0 : nullptr;
        const unsigned grid = static_cast<unsigned>(n_conc * std::min<int64_t>(ceil_div(max_seg, 16), 256 * 16 / n_conc));
        int rc;
        if (d->dtype == BESS_F32) {
            rc = (vec == 4) ? seg_by_it<float, 4>(it, red, a, gs, tab, fused_sgd_lr, grid, st)
                            : seg_by_it<float, 1>(it, red, a, gs, tab, fused_sgd_lr, grid, st);
        } else {
            rc = (vec == 8) ? seg_by_it<half_t, 8>(it, red, a, gs, tab, fused_sgd_lr, grid, st)
                 : (vec == 2) ? seg_by_it<half_t, 2>(it, red, a, gs, tab, fused_sgd_lr, grid, st)
                              : seg_by_it<half_t, 1>(it, red, a, gs, tab, fused_sgd_lr, grid, st);
        }
        if (rc) return rc;
    }
    if (long_segs) {  // the rows left out above (usually none: one launch that finds nothing to do)
        const int32_t cap = static_cast<int32_t>(long_cap);
        // whole rows, unless they are wider than a group's registers (windows as above; the scratch rows and the
        // counters are left zero by every launch, so the windows can share them)
        const int lwin = (red != RED_L2 && W > 16 * unit) ? 16 * unit : W;
        for (int col0 = 0; col0 < W; col0 += lwin) {
            const int cols = W - col0 < lwin ? W - col0 : lwin;
            SegOpt wopt = opt;
            if (wopt.state1) wopt.state1 += col0;
            if (wopt.state2) wopt.state2 += col0;
            if (wopt.xsum) wopt.xsum += col0;
            char* tab = static_cast<char*>(table) + col0 * sz;
            SegArgs a{query + col0, tab, d_out, ld_dout, refs_sorted, seg_rows, seg_offsets, n_seg,
                      static_cast<int>(n_neg), W, cols / vec, is_distance(d->scorer) ? -1.f : 1.f, long_segs, wopt,
                      static_cast<float>(d->norm_p), 1, cols};
            const int it = static_cast<int>(ceil_div(a.nch, 16));
            const unsigned lgrid = 1024;  // 16 K groups share the slices
            float* gs = grad_seg ? grad_seg + col0 : nullptr;
            float* lg = long_grad + col0;
            int rc;
            if (d->dtype == BESS_F32) {
                rc = (vec == 4) ? seg_by_it<float, 4>(it, red, a, gs, tab, fused_sgd_lr, lgrid, st, lg, long_count, cap)
                                : seg_by_it<float, 1>(it, red, a, gs, tab, fused_sgd_lr, lgrid, st, lg, long_count, cap);
            } else {
                rc = (vec == 8) ? seg_by_it<half_t, 8>(it, red, a, gs, tab, fused_sgd_lr, lgrid, st, lg, long_count, cap)
                     : (vec == 2) ? seg_by_it<half_t, 2>(it, red, a, gs, tab, fused_sgd_lr, lgrid, st, lg, long_count, cap)
                                  : seg_by_it<half_t, 1>(it, red, a, gs, tab, fused_sgd_lr, lgrid, st, lg, long_count, cap);
            }
            if (rc) return rc;
        }
    }
    return check_launch("neg_pertriple_grad_segments");
}

static OptArgs opt_args(const bess_opt_desc* o) {
    OptArgs a{o->kind, o->lr, o->momentum, o->beta1, o->beta2, o->eps, o->weight_decay, 1.f, 1.f,
              reinterpret_cast<const int32_t*>(static_cast<uintptr_t>(o->step_ptr)),
              reinterpret_cast<const int32_t*>(static_cast<uintptr_t>(o->slot_map))};
    if (o->kind == BESS_OPT_ADAM && !o->step_ptr) {
        a.bias1 = 1.f - powf(o->beta1, static_cast<float>(o->step));
        a.bias2 = 1.f - powf(o->beta2, static_cast<float>(o->step));
    }
    return a;
}

static int check_opt(const bess_opt_desc* o, const float* state1, const float* state2, const char* who) {
    BESS_REQUIRE(o, "%s: NULL descriptor", who);
    BESS_REQUIRE(o->kind >= BESS_OPT_SGD && o->kind <= BESS_OPT_ADAM, "%s: unknown optimiser %d", who, o->kind);
    if (o->kind == BESS_OPT_SGD && o->momentum != 0.f) BESS_REQUIRE(state1, "%s: SGD with momentum needs state1", who);
    if (o->kind == BESS_OPT_ADAGRAD) BESS_REQUIRE(state1, "%s: Adagrad needs state1", who);
    if (o->kind == BESS_OPT_ADAM) {
        BESS_REQUIRE(state1 && state2, "%s: Adam needs state1 and state2", who);
        BESS_REQUIRE(o->step >= 1 || o->step_ptr, "%s: Adam needs step >= 1", who);
    }
    return BESS_OK;
}

extern "C" int bess_neg_pertriple_grad_segments(const bess_model_desc* d, const float* query, int64_t n_query,
                                                void* table, int64_t n_neg, const float* d_out,
                                                int64_t ld_dout, const int32_t* refs_sorted,
                                                const int32_t* seg_rows, const int32_t* seg_offsets,
                                                const int32_t* n_seg, int64_t max_seg, float* grad_seg,
                                                float fused_sgd_lr, const int32_t* long_segs,
                                                int64_t long_cap, float* long_grad, int32_t* long_count,
                                                void* stream) {
    SegOpt off{};
    off.o.kind = -1;
    return grad_segments_impl(d, query, n_query, table, n_neg, d_out, ld_dout, refs_sorted, seg_rows, seg_offsets,
                              n_seg, max_seg, grad_seg, fused_sgd_lr, long_segs, long_cap, long_grad, long_count, off,
                              stream);
}

extern "C" int bess_neg_pertriple_step_segments(const bess_model_desc* d, const float* query, int64_t n_query,
                                                void* table, int64_t n_neg, const float* d_out,
                                                int64_t ld_dout, const int32_t* refs_sorted,
                                                const int32_t* seg_rows, const int32_t* seg_offsets,
                                                const int32_t* n_seg, int64_t max_seg,
                                                const int32_t* long_segs, int64_t long_cap, float* long_grad,
                                                int32_t* long_count, const bess_opt_desc* o, float* state1,
                                                float* state2, const int32_t* extra_map,
                                                const float* extra_sum, void* stream) {
    if (int e = check_opt(o, state1, state2, "step_segments")) return e;
    BESS_REQUIRE(!extra_map == !extra_sum, "step_segments: extra_map and extra_sum come together");
    SegOpt opt{opt_args(o), o->kind == BESS_OPT_SGD && o->momentum == 0.f ? nullptr : state1,
               o->kind == BESS_OPT_ADAM ? state2 : nullptr, extra_map, extra_sum};
    return grad_segments_impl(d, query, n_query, table, n_neg, d_out, ld_dout, refs_sorted, seg_rows, seg_offsets,
                              n_seg, max_seg, nullptr, 0.f, long_segs, long_cap, long_grad, long_count, opt, stream);
}

extern "C" int bess_map_extra_rows(const int32_t* seg_rows, const int32_t* n_seg, int64_t max_seg,
                                   const int32_t* extra_rows, const int32_t* n_extra, int64_t max_extra,
                                   int32_t* extra_map, int32_t* keep, void* stream) {
    BESS_REQUIRE(seg_rows && n_seg && extra_rows && n_extra && extra_map && keep, "map_extra_rows: NULL pointer");
    BESS_REQUIRE(max_seg > 0 && max_extra > 0, "map_extra_rows: bad sizes");
    hipStream_t st = as_stream(stream);
    hipError_t e = fill_words_async(extra_map, 0xffffffffu, max_seg, st);  // -1 everywhere
    if (e != hipSuccess) return fail(static_cast<int>(e), "memset: %s", hipGetErrorString(e));
    k_map_extra_rows<<<static_cast<unsigned>(std::min<int64_t>(ceil_div(max_extra, 256), 1024)), 256, 0, st>>>(
        seg_rows, n_seg, extra_rows, n_extra, extra_map, keep);
    return check_launch("map_extra_rows");
}

extern "C" int bess_apply_segments_sgd(int32_t dtype, int32_t width, void* table, const int32_t* seg_rows,
                                       const int32_t* n_seg, int64_t max_seg, const float* grad_seg, float lr,
                                       void* stream) {
    BESS_REQUIRE(dtype == BESS_F32 || dtype == BESS_F16, "apply_segments_sgd: unknown dtype %d", dtype);
    BESS_REQUIRE(width > 0 && max_seg > 0, "apply_segments_sgd: bad sizes");
    BESS_REQUIRE(table && seg_rows && n_seg && grad_seg, "apply_segments_sgd: NULL pointer");
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(ceil_div(max_seg * width, 256), 256 * 16));
    if (dtype == BESS_F32)
        k_apply_segments<float><<<grid, 256, 0, as_stream(stream)>>>(static_cast<float*>(table), width, seg_rows,
                                                                     n_seg, grad_seg, lr);
    else
        k_apply_segments<half_t><<<grid, 256, 0, as_stream(stream)>>>(static_cast<half_t*>(table), width, seg_rows,
                                                                      n_seg, grad_seg, lr);
    return check_launch("apply_segments_sgd");
}

extern "C" int bess_segment_sum_rows(int32_t width, const float* src, const int32_t* refs_sorted,
                                     const int32_t* seg_offsets, const int32_t* n_seg, int64_t max_seg,
                                     float* grad_seg, void* stream) {
    BESS_REQUIRE(width > 0 && max_seg > 0, "segment_sum_rows: bad sizes");
    BESS_REQUIRE(src && refs_sorted && seg_offsets && n_seg && grad_seg, "segment_sum_rows: NULL pointer");
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(ceil_div(max_seg, 4), 256 * 16));
    k_segment_sum_rows<<<grid, 256, 0, as_stream(stream)>>>(src, width, refs_sorted, seg_offsets, n_seg, grad_seg);
    return check_launch("segment_sum_rows");
}

extern "C" int bess_apply_segments_opt(const bess_opt_desc* o, int32_t dtype, int32_t width, void* table,
                                       const int32_t* seg_rows, const int32_t* n_seg, int64_t max_seg,
                                       const float* grad_seg, float* state1, float* state2,
                                       const int32_t* keep, void* stream) {
    if (int e = check_opt(o, state1, state2, "apply_segments_opt")) return e;
    BESS_REQUIRE(dtype == BESS_F32 || dtype == BESS_F16, "apply_segments_opt: unknown dtype %d", dtype);
    BESS_REQUIRE(width > 0 && max_seg > 0, "apply_segments_opt: bad sizes");
    BESS_REQUIRE(table && seg_rows && n_seg && grad_seg, "apply_segments_opt: NULL pointer");
    const OptArgs a = opt_args(o);
    float* s1 = o->kind == BESS_OPT_SGD && o->momentum == 0.f ? nullptr : state1;
    float* s2 = o->kind == BESS_OPT_ADAM ? state2 : nullptr;
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(ceil_div(max_seg * width, 256), 256 * 16));
    if (dtype == BESS_F32)
        k_apply_segments_opt<float><<<grid, 256, 0, as_stream(stream)>>>(a, static_cast<float*>(table), width,
                                                                         seg_rows, n_seg, grad_seg, s1, s2, keep);
    else
        k_apply_segments_opt<half_t><<<grid, 256, 0, as_stream(stream)>>>(a, static_cast<half_t*>(table), width,
                                                                          seg_rows, n_seg, grad_seg, s1, s2, keep);
    return check_launch("apply_segments_opt");
}

extern "C" int bess_direct_update(const bess_opt_desc* o, int32_t dtype, int32_t width, void* table, int32_t n_lists,
                                  const int32_t* const* id_lists, const int64_t* id_lens, float* acc, int32_t* claim,
                                  const int32_t* generation, float* state1, float* state2, void* axpy_table,
                                  const float* axpy_grad, int64_t axpy_n, float axpy_alpha, void* stream) {
    if (int e = check_opt(o, state1, state2, "direct_update")) return e;
    BESS_REQUIRE(!o->slot_map, "direct_update: paged optimiser state goes through the indexed update");
    BESS_REQUIRE(dtype == BESS_F32 || dtype == BESS_F16, "direct_update: unknown dtype %d", dtype);
    BESS_REQUIRE(width > 0 && table && acc && claim && generation, "direct_update: NULL pointer / bad width");
    BESS_REQUIRE(n_lists >= 1 && n_lists <= BESS_MAX_ROW_LISTS && id_lists && id_lens, "direct_update: %d lists (1 .. %d)",
                 n_lists, BESS_MAX_ROW_LISTS);
    BESS_REQUIRE(axpy_n >= 0 && (axpy_n == 0 || (axpy_table && axpy_grad)), "direct_update: axpy operands");
    IdLists ids{};
    ids.n = n_lists;
    int64_t total = 0;
    for (int l = 0; l < n_lists; ++l) {
        BESS_REQUIRE(id_lens[l] >= 0 && (id_lens[l] == 0 || id_lists[l]), "direct_update: list %d", l);
        ids.p[l] = id_lists[l];
        ids.first[l] = static_cast<int32_t>(total);
        total += id_lens[l];
    }
    BESS_REQUIRE(total > 0 && total < (1ll << 31), "direct_update: %lld references", static_cast<long long>(total));
    for (int l = n_lists; l <= BESS_MAX_ROW_LISTS; ++l) ids.first[l] = static_cast<int32_t>(total);
    const OptArgs a = opt_args(o);
    float* s1 = o->kind == BESS_OPT_SGD && o->momentum == 0.f ? nullptr : state1;
    float* s2 = o->kind == BESS_OPT_ADAM ? state2 : nullptr;
    AxpyJob x{axpy_table, axpy_grad, axpy_n, axpy_alpha,
              static_cast<int>(std::min<int64_t>(ceil_div(axpy_n, 256 * 4), 256))};
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(ceil_div(total, 4), 256 * 16)) + x.blocks;
    if (dtype == BESS_F32)
        k_direct_update<float><<<grid, 256, 0, as_stream(stream)>>>(a, ids, static_cast<int>(total),
                                                                    static_cast<float*>(table), width, acc, claim,
                                                                    generation, s1, s2, x);
    else
        k_direct_update<half_t><<<grid, 256, 0, as_stream(stream)>>>(a, ids, static_cast<int>(total),
                                                                     static_cast<half_t*>(table), width, acc, claim,
                                                                     generation, s1, s2, x);
    return check_launch("direct_update");
}

extern "C" int bess_coalesced_update(const bess_opt_desc* o, int32_t dtype, int32_t width, void* table,
                                     int32_t n_lists, const float* const* list_grad, const int64_t* list_rows,
                                     const int32_t* refs_sorted, const int32_t* seg_rows,
                                     const int32_t* seg_offsets, const int32_t* n_seg, int64_t max_seg,
                                     float* state1, float* state2, const int32_t* keep, float* sum_out,
                                     void* stream) {
    return bess_coalesced_update_axpy(o, dtype, width, table, n_lists, list_grad, list_rows, refs_sorted, seg_rows,
                                      seg_offsets, n_seg, max_seg, state1, state2, keep, sum_out, nullptr, nullptr, 0, 0.f,
                                      stream);
}

extern "C" int bess_coalesced_update_axpy(const bess_opt_desc* o, int32_t dtype, int32_t width, void* table,
                                          int32_t n_lists, const float* const* list_grad, const int64_t* list_rows,
                                          const int32_t* refs_sorted, const int32_t* seg_rows,
                                          const int32_t* seg_offsets, const int32_t* n_seg, int64_t max_seg,
                                          float* state1, float* state2, const int32_t* keep, float* sum_out,
                                          void* axpy_table, const float* axpy_grad, int64_t axpy_n, float axpy_alpha,
                                          void* stream) {
    BESS_REQUIRE(axpy_n >= 0 && (axpy_n == 0 || (axpy_table && axpy_grad)), "coalesced_update: axpy operands");
    if (!sum_out)
        if (int e = check_opt(o, state1, state2, "coalesced_update")) return e;
    BESS_REQUIRE(dtype == BESS_F32 || dtype == BESS_F16, "coalesced_update: unknown dtype %d", dtype);
    BESS_REQUIRE(width > 0 && max_seg > 0, "coalesced_update: bad sizes");
    BESS_REQUIRE(n_lists >= 1 && n_lists <= BESS_MAX_ROW_LISTS, "coalesced_update: %d lists (1 .. %d)", n_lists,
                 BESS_MAX_ROW_LISTS);
    BESS_REQUIRE((table || sum_out) && list_grad && list_rows && refs_sorted && seg_rows && seg_offsets && n_seg,
                 "coalesced_update: NULL pointer");
    RowLists L{};
    L.n = n_lists;
    int64_t total = 0;
    for (int l = 0; l < n_lists; ++l) {
        BESS_REQUIRE(list_rows[l] >= 0 && (list_rows[l] == 0 || list_grad[l]), "coalesced_update: list %d", l);
        L.grad[l] = list_grad[l];
        L.first[l] = static_cast<int32_t>(total);
        total += list_rows[l];
    }
    BESS_REQUIRE(total > 0 && total < (1ll << 31), "coalesced_update: %lld references", static_cast<long long>(total));
    for (int l = n_lists; l <= BESS_MAX_ROW_LISTS; ++l) L.first[l] = static_cast<int32_t>(total);
    OptArgs a{};
    float *s1 = nullptr, *s2 = nullptr;
    if (!sum_out) {
        a = opt_args(o);
        s1 = o->kind == BESS_OPT_SGD && o->momentum == 0.f ? nullptr : state1;
        s2 = o->kind == BESS_OPT_ADAM ? state2 : nullptr;
    }
    AxpyJob x{axpy_table, axpy_grad, axpy_n, axpy_alpha,
              static_cast<int>(std::min<int64_t>(ceil_div(axpy_n, 256 * 4), 256))};
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(ceil_div(max_seg, 4), 256 * 16)) + x.blocks;
    if (dtype == BESS_F32)
        k_coalesced_update<float><<<grid, 256, 0, as_stream(stream)>>>(a, L, static_cast<float*>(table), width,
                                                                       refs_sorted, seg_rows, seg_offsets, n_seg, s1,
                                                                       s2, keep, sum_out, x);
    else
        k_coalesced_update<half_t><<<grid, 256, 0, as_stream(stream)>>>(a, L, static_cast<half_t*>(table), width,
                                                                        refs_sorted, seg_rows, seg_offsets, n_seg, s1,
                                                                        s2, keep, sum_out, x);
    return check_launch("coalesced_update");
}

extern "C" int bess_assign_state_rows(const int32_t* seg_rows, const int32_t* n_seg, int64_t max_seg,
                                      const int32_t* keep, int32_t* slot_map, int32_t* counter, int64_t capacity,
                                      void* stream) {
    BESS_REQUIRE(seg_rows && n_seg && slot_map && counter, "assign_state_rows: NULL pointer");
    BESS_REQUIRE(max_seg > 0 && capacity > 0 && capacity < (1ll << 31), "assign_state_rows: bad sizes");
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(ceil_div(max_seg, 256), 1024));
    k_assign_state_rows<<<grid, 256, 0, as_stream(stream)>>>(seg_rows, n_seg, keep, slot_map, counter,
                                                            static_cast<int32_t>(capacity));
    return check_launch("assign_state_rows");
}

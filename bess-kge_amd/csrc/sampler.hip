// Device-side index sampling: the random streams of the
// reference's numpy samplers, reproduced bit-for-bit in HBM.
//
//   RandomShardedNegativeSampler.__call__   negative_sampler.py:104-132
//       rng.integers(1 << 31, size=[step, n, n, B, K]).astype(int32) % shard_counts[src]
//   RandomShardedBatchSampler.sample_triples batch_sampler.py:373-399
//       offsets + rng.integers(1 << 63, size=[step, n, (n,) ppp]) % counts
//   ShardedBatchSampler.__getitem__          batch_sampler.py:138-196 (h, r, t lookup,
//       tail block transpose)
//
// numpy's Generator is PCG64 (XSL-RR 128/64, 128-bit LCG state).  Its bounded
// draws for these two ranges reduce to "x >> 1" of the raw output (Lemire's
// method with range 2^31 / 2^63: the rejection threshold is 0):
//   * 32-bit draws take the low half of a 64-bit output first, then the high
//     half (pcg64_next32 keeps the second half buffered across calls);
//   * 64-bit draws use whole outputs and leave that buffer alone.
// An LCG can be advanced by any distance in O(log distance): every thread
// jumps to its own stretch of the stream with a host-provided table of the
// affine maps of 2^j steps, then steps sequentially.  Integer work, no floating
// point: outputs are bit-exact by construction and checked against numpy.
#include "common.h"

namespace bess {

struct U128 {
    uint64_t hi, lo;
};

__device__ __forceinline__ U128 mul128(U128 a, U128 b) {
    U128 r;
    r.lo = a.lo * b.lo;
    r.hi = __umul64hi(a.lo, b.lo) + a.hi * b.lo + a.lo * b.hi;
    return r;
}
__device__ __forceinline__ U128 add128(U128 a, U128 b) {
    U128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull);
    return r;
}

// PCG_DEFAULT_MULTIPLIER_128
__device__ __forceinline__ U128 pcg_mult() { return U128{0x2360ED051FC65DA4ull, 0x4385DF649FCCF645ull}; }

__device__ __forceinline__ uint64_t pcg_output(U128 s) {
    const uint64_t x = s.hi ^ s.lo;
    const unsigned r = static_cast<unsigned>(s.hi >> 58);
    return (x >> r) | (x << ((64u - r) & 63u));
}

// state after `steps` steps; table[j] = {a_hi, a_lo, c_hi, c_lo} of the map s -> a*s + c for 2^j steps
__device__ __forceinline__ U128 pcg_advance(U128 s, uint64_t steps, const uint64_t* __restrict__ table) {
    for (int j = 0; steps; ++j, steps >>= 1) {
        if (steps & 1ull) {
            const U128 a{table[4 * j + 0], table[4 * j + 1]};
            const U128 c{table[4 * j + 2], table[4 * j + 3]};
            s = add128(mul128(a, s), c);
        }
    }
    return s;
}

constexpr int DRAWS_PER_THREAD = 8;

struct NegArgs {
    U128 state, inc;
    const uint64_t* table;
    int32_t lead;         // 1: logical element 0 is the buffered half-word
    uint32_t lead_value;  // raw (before >> 1)
    int64_t chunk_len;    // logical elements per step that are produced (src_count * n * B * K)
    int64_t chunk_stride; // logical elements per step (n * n * B * K)
    int64_t chunk_off;    // src_begin * n * B * K
    int64_t n_chunk;      // steps
    int64_t block;        // n * B * K   (elements per source shard)
    int64_t BK, K;
    int32_t n_shard;
    const int32_t* shard_counts;  // [n_shard]
    // type-based sampling (nullable)
    const int32_t* wanted;        // [step, n_shard, B]
    const int32_t* type_counts;   // [n_shard, n_type]
    const int32_t* type_offsets;  // [n_shard, n_type]
    int32_t n_type;
    int32_t local_sampling;
    int32_t* out;
};

__device__ __forceinline__ void emit_negative(const NegArgs& a, int64_t step, int64_t rel, uint32_t raw) {
    // rel: offset inside the produced part of this step
    const int64_t f_in_step = a.chunk_off + rel;
    const int32_t src = static_cast<int32_t>(f_in_step / a.block);
    const int32_t v = static_cast<int32_t>(raw >> 1);
    const int32_t in_shard = a.shard_counts[src];
    // numpy: x % 0 == 0 for integers
    const int32_t row = in_shard > 0 ? v % in_shard : 0;
    int32_t res = row;
    if (a.wanted) {
        const int64_t in_block = f_in_step - static_cast<int64_t>(src) * a.block;
        const int32_t dst = static_cast<int32_t>(in_block / a.BK);
        const int64_t b = (in_block - static_cast<int64_t>(dst) * a.BK) / a.K;
        const int32_t scorer = a.local_sampling ? src : dst;
        const int32_t ty = a.wanted[(step * a.n_shard + scorer) * (a.BK / a.K) + b];
        const int32_t cnt = a.type_counts[static_cast<int64_t>(src) * a.n_type + ty];
        // TypeBasedShardedNegativeSampler (negative_sampler.py:180-230) reduces the shard-level row again
        res = (cnt > 0 ? row % cnt : 0) + a.type_offsets[static_cast<int64_t>(src) * a.n_type + ty];
    }
    a.out[step * a.chunk_len + rel] = res;
}

// grid.y = step; threads of grid.x cover the 64-bit draws overlapping the step's chunk
__global__ __launch_bounds__(256) void k_sample_negatives(NegArgs a) {
    const int64_t step = blockIdx.y;
    const int64_t f0 = step * a.chunk_stride + a.chunk_off;  // first logical element wanted
    const int64_t f1 = f0 + a.chunk_len;
    // logical element f >= lead sits at stream position p = f - lead
    int64_t p0 = f0 - a.lead;
    if (p0 < 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) emit_negative(a, step, 0, a.lead_value);
        p0 = 0;
    }
    const int64_t p1 = f1 - a.lead;  // exclusive
    if (p1 <= p0) return;
    const int64_t d_first = p0 >> 1;
    const int64_t d_last = (p1 - 1) >> 1;
    const int64_t t = blockIdx.x * 256ll + threadIdx.x;
    const int64_t d0 = d_first + t * DRAWS_PER_THREAD;
    if (d0 > d_last) return;
    U128 s = pcg_advance(a.state, static_cast<uint64_t>(d0), a.table);
    const U128 mult = pcg_mult();
#pragma unroll 1
    for (int i = 0; i < DRAWS_PER_THREAD; ++i) {
        const int64_t d = d0 + i;
        if (d > d_last) break;
        s = add128(mul128(s, mult), a.inc);
        const uint64_t x = pcg_output(s);
        const int64_t pa = 2 * d, pb = 2 * d + 1;
        if (pa >= p0 && pa < p1) emit_negative(a, step, pa + a.lead - f0, static_cast<uint32_t>(x));
        if (pb >= p0 && pb < p1) emit_negative(a, step, pb + a.lead - f0, static_cast<uint32_t>(x >> 32));
    }
}

// out[f] = offsets[b] + (x_f >> 1) % counts[b],  b = (f / inner) % n_bucket
__global__ __launch_bounds__(256) void k_sample_bucket_idx(U128 state, U128 inc, const uint64_t* __restrict__ table,
                                                           int64_t n_out, int64_t inner, int64_t n_bucket,
                                                           const int64_t* __restrict__ counts,
                                                           const int64_t* __restrict__ offsets,
                                                           int64_t* __restrict__ out) {
    const int64_t t = blockIdx.x * 256ll + threadIdx.x;
    const int64_t d0 = t * DRAWS_PER_THREAD;
    if (d0 >= n_out) return;
    U128 s = pcg_advance(state, static_cast<uint64_t>(d0), table);
    const U128 mult = pcg_mult();
#pragma unroll 1
    for (int i = 0; i < DRAWS_PER_THREAD; ++i) {
        const int64_t f = d0 + i;
        if (f >= n_out) break;
        s = add128(mul128(s, mult), inc);
        const uint64_t x = pcg_output(s) >> 1;
        const int64_t b = (f / inner) % n_bucket;
        const uint64_t cnt = static_cast<uint64_t>(counts[b]);
        out[f] = offsets[b] + static_cast<int64_t>(cnt ? x % cnt : 0ull);
    }
}

// head/relation/tail lookup; sample_idx [step, n, n2, ppp]; tail written as [step, n2, n, ppp] when swap
__global__ __launch_bounds__(256) void k_lookup_triples(const int32_t* __restrict__ triples, int64_t n_triple,
                                                        const int64_t* __restrict__ sample_idx, int64_t n_out,
                                                        int64_t n1, int64_t n2, int64_t ppp, int swap_tail,
                                                        int32_t* __restrict__ head, int32_t* __restrict__ relation,
                                                        int32_t* __restrict__ tail) {
    for (int64_t f = blockIdx.x * 256ll + threadIdx.x; f < n_out; f += 256ll * gridDim.x) {
        int64_t id = sample_idx[f];
        id = id < 0 ? 0 : (id >= n_triple ? n_triple - 1 : id);
        const int32_t h = triples[3 * id + 0], r = triples[3 * id + 1], t = triples[3 * id + 2];
        if (head) head[f] = h;
        if (relation) relation[f] = r;
        if (tail) {
            int64_t g = f;
            if (swap_tail) {
                const int64_t j = f % ppp;
                const int64_t b = (f / ppp) % n2;
                const int64_t a = (f / (ppp * n2)) % n1;
                const int64_t st = f / (ppp * n2 * n1);
                g = ((st * n2 + b) * n1 + a) * ppp + j;
            }
            tail[g] = t;
        }
    }
}

static inline U128 u128(const uint64_t* p) { return U128{p[0], p[1]}; }

// `TripleBasedShardedNegativeSampler.__call__` (reference negative_sampler.py:422-477): the candidate lists
// are fixed tables padded per shard ([n_list, n_shard, L], made once on the host and kept in HBM); a step
// looks the list of every sampled triple up and lays entities and padding mask out for the exchange:
//   ent [step, shard_neg, shard, T, L]       = table(t)[lookup[step, shard, t], shard_neg, :]   ("gather" layout)
//   msk gather layout as ent, or score layout [step, shard, T, shard_neg, L]
// T = (mid, triple) folded; with two tables ("ht": heads corrupted in the first `half` triples of every
// block of `per_part`, tails in the rest) table(t) = t % per_part < half ? table_h : table_t.
// One thread per 4 consecutive list entries (L is padded by the caller's tables, any L works).
__global__ __launch_bounds__(256) void k_gather_candidate_lists(const int32_t* __restrict__ tab_h,
                                                                const int32_t* __restrict__ tab_t,
                                                                const uint8_t* __restrict__ msk_h,
                                                                const uint8_t* __restrict__ msk_t,
                                                                const int64_t* __restrict__ lookup, int64_t n_step,
                                                                int64_t n_shard, int64_t T, int64_t per_part,
                                                                int64_t half, int64_t n_neg_shard, int64_t L,
                                                                int32_t mask_gather_layout,
                                                                int32_t* __restrict__ ent, uint8_t* __restrict__ msk) {
    const int64_t total = n_step * n_neg_shard * n_shard * T * L;
    for (int64_t o = blockIdx.x * 256ll + threadIdx.x; o < total; o += 256ll * gridDim.x) {
        // o indexes the gather layout [step, shard_neg, shard, t, l]
        int64_t r = o;
        const int64_t l = r % L; r /= L;
        const int64_t t = r % T; r /= T;
        const int64_t sh = r % n_shard; r /= n_shard;
        const int64_t sn = r % n_neg_shard;
        const int64_t st = r / n_neg_shard;
        const bool tails = tab_t && (t % per_part) >= half;
        const int64_t row = lookup[(st * n_shard + sh) * T + t];
        const int64_t src = (row * n_neg_shard + sn) * L + l;
        if (ent) ent[o] = (tails ? tab_t : tab_h)[src];
        if (msk) {
            const uint8_t m = (tails ? msk_t : msk_h)[src];
            if (mask_gather_layout) msk[o] = m;
            else msk[(((st * n_shard + sh) * T + t) * n_neg_shard + sn) * L + l] = m;
        }
    }
}

}  // namespace bess

using namespace bess;

extern "C" int bess_sample_negatives(const bess_pcg64_state* gen, const uint64_t* jump_table, int64_t n_step,
                                     int32_t n_shard, int32_t src_begin, int32_t src_count, int64_t B, int64_t K,
                                     const int32_t* shard_counts, const int32_t* wanted_type,
                                     const int32_t* type_counts, const int32_t* type_offsets, int32_t n_type,
                                     int32_t local_sampling, int32_t* out, void* stream) {
    BESS_REQUIRE(gen && jump_table && out, "sample_negatives: NULL pointer");
    BESS_REQUIRE(n_step >= 0 && n_shard > 0 && B > 0 && K > 0, "sample_negatives: bad shape");
    BESS_REQUIRE(src_begin >= 0 && src_count > 0 && src_begin + src_count <= n_shard,
                 "sample_negatives: shard range [%d, %d) outside %d shards", src_begin, src_begin + src_count, n_shard);
    BESS_REQUIRE(shard_counts, "sample_negatives: shard_counts is NULL");
    BESS_REQUIRE(!wanted_type || (type_counts && type_offsets && n_type > 0), "sample_negatives: type tables missing");
    BESS_REQUIRE(n_step <= 65535, "sample_negatives: more than 65535 steps per call");
    if (n_step == 0) return BESS_OK;
    NegArgs a;
    a.state = U128{gen->state_hi, gen->state_lo};
    a.inc = U128{gen->inc_hi, gen->inc_lo};
    a.table = jump_table;
    a.lead = gen->has_uint32 ? 1 : 0;
    a.lead_value = gen->uinteger;
    a.K = K;
    a.BK = B * K;
    a.block = static_cast<int64_t>(n_shard) * a.BK;
    a.chunk_stride = static_cast<int64_t>(n_shard) * a.block;
    a.chunk_len = static_cast<int64_t>(src_count) * a.block;
    a.chunk_off = static_cast<int64_t>(src_begin) * a.block;
    a.n_chunk = n_step;
    a.n_shard = n_shard;
    a.shard_counts = shard_counts;
    a.wanted = wanted_type;
    a.type_counts = type_counts;
    a.type_offsets = type_offsets;
    a.n_type = n_type;
    a.local_sampling = local_sampling;
    a.out = out;
    // at most chunk_len / 2 + 1 draws overlap a chunk
    const int64_t draws = a.chunk_len / 2 + 2;
    const int64_t threads = ceil_div(draws, DRAWS_PER_THREAD);
    dim3 grid(static_cast<unsigned>(ceil_div(threads, 256)), static_cast<unsigned>(n_step));
    k_sample_negatives<<<grid, 256, 0, as_stream(stream)>>>(a);
    return check_launch("sample_negatives");
}

extern "C" int bess_sample_bucket_indices(const bess_pcg64_state* gen, const uint64_t* jump_table, int64_t n_out,
                                          int64_t inner, int64_t n_bucket, const int64_t* counts,
                                          const int64_t* offsets, int64_t* out, void* stream) {
    BESS_REQUIRE(gen && jump_table && counts && offsets && out, "sample_bucket_indices: NULL pointer");
    BESS_REQUIRE(n_out >= 0 && inner > 0 && n_bucket > 0, "sample_bucket_indices: bad shape");
    if (n_out == 0) return BESS_OK;
    const int64_t threads = ceil_div(n_out, DRAWS_PER_THREAD);
    k_sample_bucket_idx<<<static_cast<unsigned>(ceil_div(threads, 256)), 256, 0, as_stream(stream)>>>(
        U128{gen->state_hi, gen->state_lo}, U128{gen->inc_hi, gen->inc_lo}, jump_table, n_out, inner, n_bucket,
        counts, offsets, out);
    return check_launch("sample_bucket_indices");
}

extern "C" int bess_lookup_triples(const int32_t* triples, int64_t n_triple, const int64_t* sample_idx,
                                   int64_t n_step, int64_t n1, int64_t n2, int64_t per_part, int32_t swap_tail,
                                   int32_t* head, int32_t* relation, int32_t* tail, void* stream) {
    BESS_REQUIRE(triples && sample_idx, "lookup_triples: NULL pointer");
    BESS_REQUIRE(n_triple > 0 && n_step >= 0 && n1 > 0 && n2 > 0 && per_part > 0, "lookup_triples: bad shape");
    const int64_t n_out = n_step * n1 * n2 * per_part;
    if (n_out == 0) return BESS_OK;
    const unsigned blocks = static_cast<unsigned>(std::min<int64_t>(ceil_div(n_out, 256), 4096));
    k_lookup_triples<<<blocks, 256, 0, as_stream(stream)>>>(triples, n_triple, sample_idx, n_out, n1, n2, per_part,
                                                            swap_tail, head, relation, tail);
    return check_launch("lookup_triples");
}

extern "C" int bess_gather_candidate_lists(const int32_t* table_h, const int32_t* table_t, const uint8_t* mask_h,
                                           const uint8_t* mask_t, int64_t n_list, const int64_t* lookup,
                                           int64_t n_step, int64_t n_shard, int64_t n_triple, int64_t per_part,
                                           int64_t half, int64_t n_neg_shard, int64_t list_len,
                                           int32_t mask_gather_layout, int32_t* entities, uint8_t* mask,
                                           void* stream) {
    BESS_REQUIRE(n_step >= 0 && n_shard > 0 && n_triple >= 0 && n_neg_shard > 0 && list_len > 0 && n_list > 0,
                 "gather_candidate_lists: bad sizes");
    if (n_step == 0 || n_triple == 0) return BESS_OK;
    BESS_REQUIRE(table_h && lookup && (entities || mask), "gather_candidate_lists: NULL pointer");
    BESS_REQUIRE(!mask || mask_h, "gather_candidate_lists: mask output without a mask table");
    BESS_REQUIRE(!table_t || (per_part > 0 && half >= 0 && half <= per_part && n_triple % per_part == 0 &&
                              (!mask || mask_t)),
                 "gather_candidate_lists: the two-table form needs per_part | n_triple, 0 <= half <= per_part");
    if (!table_t) per_part = 1, half = 1;
    const int64_t total = n_step * n_neg_shard * n_shard * n_triple * list_len;
    BESS_REQUIRE(total < (1ll << 40), "gather_candidate_lists: output too large");
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 8192) blocks = 8192;
    k_gather_candidate_lists<<<static_cast<unsigned>(blocks), 256, 0, as_stream(stream)>>>(
        table_h, table_t, mask_h, mask_t, lookup, n_step, n_shard, n_triple, per_part, half, n_neg_shard, list_len,
        mask_gather_layout, entities, mask);
    return check_launch("gather_candidate_lists");
}

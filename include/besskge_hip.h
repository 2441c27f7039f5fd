/*
 * besskge_hip.h - C ABI of the MI355X (gfx950) BESS hot path.
 *
 * One shared library, `libbesskge_hip.so`, loaded with ctypes at
 * `import besskge` - the place where the reference dlopens its PopART
 * custom-op library (reference besskge/__init__.py:10-37).  The reference has
 * no exported C functions of its own (its .so only self-registers a PopART
 * graph pattern, custom_ops/remove_all_reduce_pattern.cpp:45-47); each entry
 * point below therefore cites the reference *Python* expression it replaces.
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = invalid argument (BESS_E*),
 *     >0 = hipError_t of the failing runtime call (BESS_ECOMM_BASE + ncclResult_t
 *     for RCCL); bess_last_error() gives text;
 *   - no ownership transfer: every buffer is allocated by the caller (PyTorch)
 *     and passed as a raw device pointer; the library never allocates,
 *     synchronises or frees;
 *   - calls are asynchronous on the given hipStream_t (passed as void*);
 *   - no global mutable state except communicators (bess_comm_*); safe from one
 *     host thread per device;
 *   - index arrays are int32 (the dtype the samplers emit,
 *     batch_sampler.py:170-178); row offsets are computed in 64 bit, so a
 *     shard may exceed 4 GiB (config 5: 62.5 M rows x 2 KiB);
 *   - "rows" arguments come as (base, idx): row i is base[idx[i]*width ...];
 *     idx == NULL means the identity (row i is base[i*width ...]).  With
 *     n_shard == 1 `base` is the shard itself and idx the sampler's indices
 *     (gather fused into the scoring kernel, nothing materialised); with
 *     n_shard > 1 `base` is the all-to-all receive buffer and idx a static
 *     re-ordering map.
 */
#ifndef BESSKGE_HIP_H
#define BESSKGE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BESS_ABI_VERSION 3  /* 3 (round 4): the "last workgroup" counters are int32 [BESS_TICKET_INTS] (were [1]) */

/* error codes (negative); positive return values are hipError_t, or
 * BESS_ECOMM_BASE + ncclResult_t for a failing RCCL call */
#define BESS_OK 0
#define BESS_EINVAL -1       /* bad enum / size / null pointer            */
#define BESS_EUNSUPPORTED -2 /* legal but not built (width/alignment...)  */
#define BESS_ECOMM_BASE 10000

/* scoring functions (reference besskge/scoring.py:258-462, 746-946) */
#define BESS_TRANSE 0
#define BESS_ROTATE 1
#define BESS_DISTMULT 2
#define BESS_COMPLEX 3
/* PairRE / TripleRE / InterHT / TranS (scoring.py:465-743, 1418-1750), as seen by the
 * negative-scoring kernels: score = -|| U*c1(e) + V*c2(e) + R ||_p with the query matrix
 * [n_query, (n_part + 1) * d] = [U | V | R] (V absent for n_part = 1), entity rows of
 * n_part * d scalars = [c1 | c2], each part optionally L2-normalised.
 * desc.width = n_part * d, desc.reserved[0] = n_part (1 | 2), desc.reserved[1] bit 0 =
 * normalise the parts (per-triple kernels only; the shared kernels take candidates that
 * bess_normalize_rows has already gathered, converted to f32 and normalised).
 * bess_query_fwd / bess_query_bwd build [U | V | R] from the kept entity and the relation row
 * (and back-propagate through it, including the normalisation); for them desc.reserved[1]
 * bits 8-15 name the member (0 PairRE, 1 TripleRE, 2 InterHT, 3 TranS), desc.reserved[2] holds
 * the bits of the float constant (TripleRE u, InterHT / TranS offset) and desc.rel_width the
 * relation row (2d, 3d, d, 3d). */
#define BESS_AFFINE 4
/* BoxE (scoring.py:1149-1415) as seen by the negative-scoring kernels: entity rows
 * [base | bump] (desc.width = 2 d), query matrix [n_query, 6 d] =
 * [S_0 | C_0 | H_0 | S_1 | C_1 | H_1] (shift, box centre, half width met by candidate part
 * 0 / 1); desc.reserved[0] bit 0 = apply_tanh, bit 1 = dist_func_per_dim.  See csrc/boxe.hip. */
#define BESS_BOXE 5

/* table element types */
#define BESS_F32 0
#define BESS_F16 1

/* which entity the negatives replace */
#define BESS_CORRUPT_HEAD 0 /* score_heads: query built from (r, t) */
#define BESS_CORRUPT_TAIL 1 /* score_tails: query built from (h, r) */

/* loss functions (reference besskge/loss.py:109-251) */
#define BESS_LOSS_LOGSIGMOID 0
#define BESS_LOSS_MARGIN 1
#define BESS_LOSS_SSCE 2

/* most (row ids, gradient rows) lists one multi-list update call takes */
#define BESS_MAX_ROW_LISTS 8

/* value added to the score of a masked negative (reference bess.py:31) */
#define BESS_BAD_NEGATIVE_SCORE (-50000.0f)

/* bess_model_desc.reserved[0] of TransE / RotatE / DistMult / ComplEx: flags */
#define BESS_FLAG_PREZEROED 2 /* bess_neg_score_shared_bwd(_ws): d_query and d_neg are zero on entry (e.g. cleared by
                               * bess_step_prologue): the call does not clear them itself */
#define BESS_FLAG_DNEG_BY_ROW 4 /* bess_neg_score_pertriple_bwd: d_neg is a matrix over the ROW SPACE of neg_base and the
                                  * gradient of reference k is stored at row neg_idx[k] (plain stores) - for lists that
                                  * name every row at most once (negatives that arrived through the all-to-all: the
                                  * receive-buffer gradient is written in place, no [n_query * n_neg, W] copy) */
#define BESS_FLAG_FP32_MATH 1 /* shared negatives with the plain fp32 kernels only: no packed-fp16 L1 forward
                                 (TransE / RotatE on f16 tables), no split-fp16 matrix-core products (DistMult /
                                 ComplEx) */

typedef struct bess_model_desc {
    int32_t scorer;    /* BESS_TRANSE ...                                  */
    int32_t norm_p;    /* p >= 1 of the distance scorers (scoring.py:174: any p; 1 and 2 have their own kernels,
                          other p go through powf), ignored otherwise */
    int32_t dtype;     /* element type of entity and relation tables       */
    int32_t width;     /* W : scalars per entity row (2d for RotatE/ComplEx) */
    int32_t rel_width; /* Wr: scalars per relation row (d for RotatE)      */
    int32_t reserved[3];
} bess_model_desc;

/* row-sparse optimisers (K10) */
#define BESS_OPT_SGD 0     /* + momentum, weight decay                    */
#define BESS_OPT_ADAGRAD 1
#define BESS_OPT_ADAM 2    /* lazy Adam; decoupled weight decay = AdamW   */

typedef struct bess_opt_desc {
    int32_t kind; /* BESS_OPT_* */
    int32_t step; /* 1-based step count (Adam bias correction) */
    float lr;
    float momentum;
    float beta1;
    float beta2;
    float eps;
    float weight_decay;
    int64_t step_ptr; /* 0, or a device `const int32_t*`: the step count is read there by the kernels
                         instead of `step` (a hipGraph replays the same launch with a growing count) */
    int64_t slot_map; /* 0: state1 / state2 are [M, W] like the table.  Else a device `const int32_t*` [M] (paged
                         state): the state of table row r is row slot_map[r] of state1 / state2 [capacity, W]
                         (-1: the row has none - it is stepped from zero state and none is kept);
                         bess_assign_state_rows hands the rows of a step their state rows first */
} bess_opt_desc;

typedef struct bess_loss_desc {
    int32_t kind;        /* BESS_LOSS_*                                    */
    int32_t adversarial; /* self-adversarial negative weights (loss.py:39) */
    float margin;
    float adversarial_scale;
    float loss_scale;
    float ssce_shift; /* log(n_entity-1) - log(N)  (loss.py:233-237)        */
    int32_t reserved[2];
} bess_loss_desc;

/* State of a numpy `Generator(PCG64)` (`rng.bit_generator.state`): 128-bit LCG
 * state and increment, plus the half of a 64-bit output that `integers()` with a
 * 32-bit range keeps buffered between calls. */
typedef struct bess_pcg64_state {
    uint64_t state_hi, state_lo;
    uint64_t inc_hi, inc_lo;
    uint32_t has_uint32;
    uint32_t uinteger;
} bess_pcg64_state;

int bess_version(void);
/* copies the calling thread's last error text; returns its length */
int bess_last_error(char* buf, size_t len);

/* K1 - `self.entity_embedding[gather_idx]` (bess.py:332-337, 502-507).
 * out[i, :] = table[idx[i], :], i < n; dtype preserved. */
int bess_gather_rows(int32_t dtype, int32_t width, const void* table,
                     const int32_t* idx, int64_t n, void* out, void* stream);

/* K2+K3 - `score_fn.score_triple(h, r, t)` (scoring.py:321-330, 423-434,
 * 804-813, 905-916).  out[s] (f32), s < n_triple. */
int bess_score_triple_fwd(const bess_model_desc* d, const void* head_base,
                          const int32_t* head_idx, const void* tail_base,
                          const int32_t* tail_idx, const void* rel_table,
                          const int32_t* rel_idx, int64_t n_triple, float* out,
                          void* stream);

/* backward of bess_score_triple_fwd: d_head / d_tail [n_triple, W] f32 are
 * overwritten; d_rel_table [n_rel, Wr] f32 is accumulated (atomic). */
int bess_score_triple_bwd(const bess_model_desc* d, const void* head_base,
                          const int32_t* head_idx, const void* tail_base,
                          const int32_t* tail_idx, const void* rel_table,
                          const int32_t* rel_idx, int64_t n_triple,
                          const float* d_out, float* d_head, float* d_tail,
                          float* d_rel_table, void* stream);

/* K2+K6 - the query transform in front of score_heads / score_tails
 * (scoring.py:342,354,446-448,460-462,825,837,928-932,944-946; utils.py:72-112):
 *   TAIL: h+r | rot(h,r) | h*r | cmul(h,r)     HEAD: t-r | rot(t,-r) | r*t | cmul(conj r,t)
 * query[q, :] (f32 [n_query, W]) for entity row q and relation rel_idx[q]. */
int bess_query_fwd(const bess_model_desc* d, int32_t side, const void* ent_base,
                   const int32_t* ent_idx, const void* rel_table,
                   const int32_t* rel_idx, int64_t n_query, float* query,
                   void* stream);

/* backward of bess_query_fwd: d_ent [n_query, W] f32 overwritten,
 * d_rel_table accumulated (atomic). */
int bess_query_bwd(const bess_model_desc* d, int32_t side, const void* ent_base,
                   const int32_t* ent_idx, const void* rel_table,
                   const int32_t* rel_idx, int64_t n_query, const float* d_query,
                   float* d_ent, float* d_rel_table, void* stream);

/* K2 + K3 + K6 in one launch (TransE / RotatE / DistMult / ComplEx): bess_score_triple_fwd (out
 * [n_triple]) and bess_query_fwd (query [n_triple, W]) of the same triples - the query is built
 * from the head rows when side == BESS_CORRUPT_TAIL, from the tail rows otherwise - and their
 * backward: bess_score_triple_bwd + bess_query_bwd, the two gradients that land on the same
 * entity row (the one the query was built from) summed: d_head / d_tail [n_triple, W] f32
 * overwritten, d_rel_table accumulated. */
int bess_query_triple_fwd(const bess_model_desc* d, int32_t side, const void* head_base,
                          const int32_t* head_idx, const void* tail_base, const int32_t* tail_idx,
                          const void* rel_table, const int32_t* rel_idx, int64_t n_triple,
                          float* query, float* out, void* stream);
int bess_query_triple_bwd(const bess_model_desc* d, int32_t side, const void* head_base,
                          const int32_t* head_idx, const void* tail_base, const int32_t* tail_idx,
                          const void* rel_table, const int32_t* rel_idx, int64_t n_triple,
                          const float* d_out, const float* d_query, float* d_head, float* d_tail,
                          float* d_rel_table, void* stream);

/* K5 - per-triple negatives, `reduce_embedding(q[:,None] o N)` (scoring.py:199,254):
 *   out[q*ld_out + k] = score(query[q], neg_base[neg_idx[q*n_neg + k]]),  k < n_neg
 * HBM-bound: one gathered row per scored triple, read straight from the shard.
 * Row width W: a row lives in registers, 16 lanes x 16 chunks - f32 up to 1024 scalars when
 * W % 4 == 0, else 256; f16 up to 2048 when W % 8 == 0, 512 when W % 2 == 0, else 256.  Wider rows
 * (embedding_size 1000 of RotatE / ComplEx ...) are scored, back-propagated and reduced (K9) in column
 * windows of that size - the dot product and the p = 1 distance are sums over columns.  Refused on such
 * rows (BESS_EUNSUPPORTED): the p = 2 distance (its root and gradient need the whole row) and the fused
 * training forward (the softmax needs the finished score; callers take the two-pass path). */
int bess_neg_score_pertriple_fwd(const bess_model_desc* d, const float* query,
                                 int64_t n_query, const void* neg_base,
                                 const int32_t* neg_idx, int64_t n_neg, float* out,
                                 int64_t ld_out, void* stream);

/* backward of K5.  d_out read with ld_dout.  d_query [n_query, W] f32 is
 * overwritten; d_neg [n_query*n_neg, W] f32 (gradient w.r.t. every gathered
 * row, same order as neg_idx) is overwritten - or pass d_neg = NULL when the
 * row gradients are produced by bess_neg_pertriple_grad_segments instead.
 * d_query = NULL (with d_neg; TransE / RotatE / DistMult / ComplEx): only d_neg is wanted - d_query came out of
 * bess_neg_score_pertriple_fwd_dq.  DistMult / ComplEx then do not read the candidate rows at all (d_neg =
 * coefficient x query): the one pass over rows that arrived through the all-to-all was the fused forward. */
int bess_neg_score_pertriple_bwd(const bess_model_desc* d, const float* query,
                                 int64_t n_query, const void* neg_base,
                                 const int32_t* neg_idx, int64_t n_neg,
                                 const float* d_out, int64_t ld_dout,
                                 float* d_query, float* d_neg, void* stream);

/* K4 - shared negatives, `pea.distance_matrix(q, N)` / `q @ N.T`
 * (scoring.py:194-197, 251-252):
 *   out[q*ld_out + j] = score(query[q], neg_base[neg_idx[j]]),  j < n_neg */
int bess_neg_score_shared_fwd(const bess_model_desc* d, const float* query,
                              int64_t n_query, const void* neg_base,
                              const int32_t* neg_idx, int64_t n_neg, float* out,
                              int64_t ld_out, void* stream);

/* TransE / RotatE with p = 1 on f16 tables and W % 32 == 0 (BASELINE configs[3]) take packed-fp16
 * forms of K4 (csrc/l1_f16.hip) unless desc.reserved[0] has BESS_FLAG_FP32_MATH: as in the
 * reference's fp16 mode (`model.half()`: the query `h + r` is an fp16 tensor) the query is rounded
 * to fp16 (nearest even) when it meets the fp16 candidates; the sum over W is then exact-or-fp32
 * (sum |q - e| = 2 sum max(q, e) - sum q - sum e).  bess_neg_score_shared_bwd(_ws) then
 * differentiates that function: sgn(fp16(q) - e), exact, 0 at a tie. */

/* K4 + K7 in one call: bess_neg_score_shared_fwd_ws followed by bess_mask_scores(out, n_query,
 * n_neg, ld_out, kill->diag_step, kill->ht, kill->ppp, kill->mask, kill->mask_rows,
 * kill->mask_cols); the packed-fp16 L1 kernel applies the kill in its epilogue (one pass over
 * the scores less).  kill == NULL: no masking. */
typedef struct bess_kill_desc {
    int32_t diag_step; /* > 0: augmentation, the true head / tail sits at column diag_step * qpos(s) */
    int32_t ht;
    int32_t ppp;
    int32_t reserved;
    const uint8_t* mask; /* [mask_rows, mask_cols] or NULL */
    int64_t mask_rows;
    int64_t mask_cols;
} bess_kill_desc;
int bess_neg_score_shared_fwd_masked(const bess_model_desc* d, const float* query,
                                     int64_t n_query, const void* neg_base,
                                     const int32_t* neg_idx, int64_t n_neg, float* out,
                                     int64_t ld_out, const bess_kill_desc* kill, void* workspace,
                                     int64_t workspace_bytes, void* stream);

/* K4 with a scratch buffer.  For the bilinear scorers (DistMult, ComplEx) and shapes of at
 * least 256 output tiles of 128 x 128, bess_neg_score_shared_workspace returns the bytes of
 * device scratch (16-B aligned, contents irrelevant, free again when the call has run on
 * `stream`) with which the product runs on the fp16 matrix cores at fp32 accuracy: operands
 * are split into fp16 pairs carrying 22 significand bits, products accumulate in fp32
 * (csrc/gemm_split.hip).  An operand that is not finite or not below 65504 in magnitude (the fp16
 * range) is noticed by the splitting pre-pass: the call then computes the product with the exact
 * fp32 MFMA kernels instead, by itself and without a host synchronisation (they are queued behind
 * the split kernels, each side gated on a flag in the scratch).  desc.reserved[0] &
 * BESS_FLAG_FP32_MATH keeps the fp32 kernels from the start (no scratch is asked for then).  It returns 0 when the shape or the scorer does
 * not use scratch.  With workspace == NULL or fewer bytes than asked for, the call is
 * bess_neg_score_shared_fwd (exact fp32 MFMA). */
int64_t bess_neg_score_shared_workspace(const bess_model_desc* d, int64_t n_query, int64_t n_neg);
int bess_neg_score_shared_fwd_ws(const bess_model_desc* d, const float* query,
                                 int64_t n_query, const void* neg_base,
                                 const int32_t* neg_idx, int64_t n_neg, float* out,
                                 int64_t ld_out, void* workspace, int64_t workspace_bytes,
                                 void* stream);

/* backward of K4.  `out` is the forward result (needed for p = 2).
 * d_query [n_query, W] and d_neg [n_neg, W] (f32) are overwritten.  The distance scorers compute
 * both products in one launch; when d_neg == d_query + n_query * W (one allocation holding both)
 * the partial-sum targets are zeroed with one memset instead of two. */
int bess_neg_score_shared_bwd(const bess_model_desc* d, const float* query,
                              int64_t n_query, const void* neg_base,
                              const int32_t* neg_idx, int64_t n_neg,
                              const float* out, int64_t ld_out, const float* d_out,
                              int64_t ld_dout, float* d_query, float* d_neg,
                              void* stream);

/* backward of K4 with a scratch buffer, as bess_neg_score_shared_fwd_ws: the two products
 * (d_query = d_out . E, d_neg = d_out^T . Q) on the fp16 matrix cores from split operands,
 * k split over workgroups with the partial tiles summed in a fixed order (deterministic).
 * bess_neg_score_shared_bwd_workspace returns the scratch bytes, 0 when the shape or the
 * scorer does not use scratch. */
/* K4 for the top-k passes over all entities (bess.py:771-822): scores against the rows' current k-th best.
 * thr [n_query]: a row's block of 64 consecutive candidates (block b = columns 64 b .. 64 b + 63) is written to
 * `out` only if one of its scores is above thr[row]; flags [n_query, ld_flags] (uint8, ld_flags >= ceil(n_neg /
 * 64), a multiple of 4, 4-byte aligned) receives 1 for every block that was written, 0 for the others (their part of
 * `out` is left as it was).  bess_topk_update_flagged reads only the flagged blocks.  Scorers / shapes whose
 * kernel has no pruning epilogue write everything and flag everything (same results, no saving): the split-fp16
 * matrix-core product (DistMult / ComplEx; workspace as for bess_neg_score_shared_fwd_ws) and the packed-fp16 L1
 * kernel (TransE / RotatE p = 1 on f16 tables) prune. */
int bess_neg_score_shared_fwd_pruned(const bess_model_desc* d, const float* query, int64_t n_query,
                                     const void* neg_base, const int32_t* neg_idx, int64_t n_neg, float* out,
                                     int64_t ld_out, const float* thr, uint8_t* flags, int64_t ld_flags,
                                     void* workspace, int64_t workspace_bytes, void* stream);

/* K4 for ranks over all entities without the score matrix (replaces the reference's AllScoresBESS window loop +
 * Evaluation.ranks_from_scores, bess.py:1005-1062 + metric.py:129-182, when only ranks / metrics are wanted):
 *   counts[q, 0] += #{ j != excl[q] : score(q, j) >  thr[q] }
 *   counts[q, 1] += #{ j != excl[q] : score(q, j) == thr[q] }
 * thr [n_query] f32: the score of the row's true completion; excl [n_query] int32: the position of that completion
 * in the candidate list (0 .. n_neg - 1), or -1 when it is not among these candidates (another shard's entity);
 * counts [n_query, 2] int32 is accumulated into (clear it before the first shard / window).  The counting is the
 * epilogue of the split-fp16 matrix-core product (DistMult / ComplEx) and of the packed-fp16 L1 kernel (TransE /
 * RotatE p = 1 on f16 tables): no score is written.  Other scorers / shapes score tiles of at most 64 MiB into the
 * workspace and count them there.  workspace: bess_neg_score_shared_fwd_counts_workspace bytes, 16-B aligned.
 * round_f16 != 0: every score is rounded to fp16 before it is compared (thr is taken as given) - the ranking a
 * reference whose model is in half precision makes of its fp16 scores, ties included.
 * If an operand of the matrix-core product is outside the fp16 range every count of the call is set to INT32_MIN
 * (there is no score matrix for the fp32 kernels to fall back on): a caller that sees negative counts scores that
 * batch through bess_neg_score_shared_fwd_ws + bess_ranks_from_scores instead. */
int64_t bess_neg_score_shared_fwd_counts_workspace(const bess_model_desc* d, int64_t n_query, int64_t n_neg);
int bess_neg_score_shared_fwd_counts(const bess_model_desc* d, const float* query, int64_t n_query,
                                     const void* neg_base, const int32_t* neg_idx, int64_t n_neg, const float* thr,
                                     const int32_t* excl, int32_t* counts, int32_t round_f16, void* workspace,
                                     int64_t workspace_bytes, void* stream);

/* Single (query, candidate) scores in the arithmetic of the all-entity pass: out[i] = score(query[i], row
 * neg_idx[i] of neg_base) exactly as bess_neg_score_shared_fwd_counts / _fwd_ws computes that element inside a
 * (like_n_query x like_n_neg) problem - the same kernel run on the diagonal tiles of the (pairs x pairs) problem
 * (split-fp16 product, packed L1 kernel) or on 1024 x 1024 blocks whose diagonal is kept (the other kernels); the
 * kernels' per-element arithmetic does not depend on the element's place in the matrix.  For the positive scores
 * and the filtered completions that a rank count is corrected with (pipeline.py:233-271): a per-triple kernel
 * would round differently and move ranks by one at near-ties.  TransE / RotatE / DistMult / ComplEx.
 * workspace: bess_neg_score_shared_fwd_pairs_workspace bytes, 16-B aligned. */
int64_t bess_neg_score_shared_fwd_pairs_workspace(const bess_model_desc* d, int64_t like_n_query, int64_t like_n_neg);
int bess_neg_score_shared_fwd_pairs(const bess_model_desc* d, const float* query, const void* neg_base,
                                    const int32_t* neg_idx, int64_t n_pair, int64_t like_n_query, int64_t like_n_neg,
                                    float* out, void* workspace, int64_t workspace_bytes, void* stream);

int64_t bess_neg_score_shared_bwd_workspace(const bess_model_desc* d, int64_t n_query, int64_t n_neg);
int bess_neg_score_shared_bwd_ws(const bess_model_desc* d, const float* query,
                                 int64_t n_query, const void* neg_base,
                                 const int32_t* neg_idx, int64_t n_neg, const float* out,
                                 int64_t ld_out, const float* d_out, int64_t ld_dout,
                                 float* d_query, float* d_neg, void* workspace,
                                 int64_t workspace_bytes, void* stream);

/* K7 - mask / augment block of BessKGE.forward (bess.py:182-245), in place:
 *   neg[s, j] += BAD_NEGATIVE_SCORE  where
 *     j == diag_step * qpos(s)                       (diag_step > 0: augment)
 *     or, for the last mask_cols columns, !mask[mrow(s), j - (n_neg - mask_cols)]
 * qpos/mrow: block = s / ppp, p = s % ppp, cut = ppp/2
 *   ht == 0: qpos = s                     ht == 1: qpos = block*cut + p % cut
 *   mask_rows == 1: mrow = 0;  mask_rows == 2: mrow = (p >= cut);  else mrow = s
 * mask is uint8 (bool) [mask_rows, mask_cols], 1 = real negative. */
int bess_mask_scores(float* neg, int64_t n_triple, int64_t n_neg, int64_t ld,
                     int32_t diag_step, int32_t ht, int32_t ppp,
                     const uint8_t* mask, int64_t mask_rows, int64_t mask_cols,
                     void* stream);

/* K8 - loss + its gradient w.r.t. the scores (loss.py:28-51,115-134,179-195,
 * 224-251), fp32.  weight has weight_len (1 or n_triple) entries.
 * row_loss [n_triple] (caller-provided scratch) receives the per-triple terms,
 * loss[0] their fixed-order sum (bitwise reproducible); d_pos [n_triple] and
 * d_neg [n_triple, ld_dneg] are overwritten (pass NULL for both to skip
 * gradients). */
int bess_loss_fwd_bwd(const bess_loss_desc* l, const float* pos, const float* neg,
                      int64_t n_triple, int64_t n_neg, int64_t ld_neg,
                      const float* weight, int64_t weight_len, float* row_loss,
                      float* loss, float* d_pos, float* d_neg, int64_t ld_dneg,
                      void* stream);

/* The same, also writing row_norm [n_triple, 2] (optional, may be NULL) = (m, L / C) of every row's
 * softmax over its negative scores - what bess_combine_dq_partials takes on the shards that scored
 * the negatives (ScoreMoving training): m = max of beta * (score + shift) (and the positive score,
 * sampled softmax), L = sum of exp(beta * (score + shift) - m) (+ exp(pos - m)), C = loss_scale * w
 * (x 1/2 for the log-sigmoid loss); beta = adversarial_scale, 1 (sampled softmax) or 0 (uniform
 * weights: m = 0, L = n_neg).  Margin ranking: the negatives' weights do not depend on the positive
 * score, the gradients' gate does - row_norm is written but of no use to the fused form. */
int bess_loss_fwd_bwd_norm(const bess_loss_desc* l, const float* pos, const float* neg,
                           int64_t n_triple, int64_t n_neg, int64_t ld_neg,
                           const float* weight, int64_t weight_len, float* row_loss,
                           float* loss, float* d_pos, float* d_neg, int64_t ld_dneg,
                           float* row_norm, void* stream);

/* The same in ONE launch for micro-batches of up to 16,384 triples (two launches above that): the workgroup that finishes last forms loss[0] from row_loss (fixed
 * order: bitwise reproducible).  counter: int32 [BESS_TICKET_INTS] on the device, zero on entry, left zero (keep one
 * array per stream; the workgroups take their "am I last" tickets in two levels, each counter on its own line). */
#define BESS_TICKET_INTS 544
int bess_loss_fwd_bwd_one_launch(const bess_loss_desc* l, const float* pos, const float* neg,
                                 int64_t n_triple, int64_t n_neg, int64_t ld_neg,
                                 const float* weight, int64_t weight_len, float* row_loss,
                                 float* loss, float* d_pos, float* d_neg, int64_t ld_dneg,
                                 float* row_norm, int32_t* counter, void* stream);

/* next-1 / K11 - streaming top-k (reference bess.py:771-822 loop body, 889-894):
 * merge the n_col candidates of every row into its running list of the kk best
 * (best_score / best_id [n_row, kk], sorted by descending score, in place).
 * Candidate j of row r has score scores[r*ld + j] (+ BAD_NEGATIVE_SCORE where
 * mask says padding) and id ids[(ids_rows == 1 ? 0 : r)*n_col + j], or
 * id_base + j when ids == NULL.  kk <= 128.  Earlier entries win ties. */
int bess_topk_update(const float* scores, int64_t n_row, int64_t n_col, int64_t ld,
                     const int32_t* ids, int64_t ids_rows, int32_t id_base,
                     const uint8_t* mask, int64_t mask_rows, float* best_score,
                     int32_t* best_id, int32_t kk, void* stream);

/* The same update from a pruned score tile (bess_neg_score_shared_fwd_pruned): only the blocks of 64 columns
 * with a non-zero flag are read; candidate ids are id_base + column.  ld_flags: a multiple of 4. */
int bess_topk_update_flagged(const float* scores, int64_t n_row, int64_t n_col, int64_t ld,
                             const uint8_t* flags, int64_t ld_flags, int32_t id_base, float* best_score,
                             int32_t* best_id, int32_t kk, void* stream);

/* next-2 - prediction ranks (reference metric.py:129-217), fp32 ranks:
 * from scores: 1 + #{cand[s, j] better than pos[s]}, mode 0 optimistic ('>'),
 * 1 pessimistic ('>='), 2 average; from ordered int64 candidate ids: 1-based
 * position of ground_truth[s], else n_cand + 1 (or +inf with worst_rank_infty). */
int bess_ranks_from_scores(const float* pos, const float* cand, int64_t n_row,
                           int64_t n_cand, int64_t ld, int32_t mode,
                           int32_t worst_rank_infty, float* rank, void* stream);
int bess_ranks_from_indices(const int64_t* ground_truth, const int64_t* candidates,
                            int64_t n_row, int64_t n_cand, int32_t worst_rank_infty,
                            float* rank, void* stream);

/* K9 - sparse scatter-add, backward of K1 (autograd index_put_(accumulate)):
 *   dst[idx[i], :] += scale * src[i, :]     (f32 atomics, duplicates allowed)
 * dst is f32 [*, width]. */
int bess_scatter_add_rows(float* dst, int32_t width, const int32_t* idx,
                          const float* src, int64_t n, float scale, void* stream);

/* K10 - sparse SGD step on a shard (table dtype f32 or f16):
 *   table[idx[i], :] -= lr * grad[i, :]     (atomic, duplicates accumulate) */
int bess_sparse_sgd(int32_t dtype, int32_t width, void* table, const int32_t* idx,
                    const float* grad, int64_t n, float lr, void* stream);

/* bess_sparse_sgd for up to BESS_MAX_ROW_LISTS (row ids, gradient rows) lists in ONE launch
 * (heads, tails, shared negatives ... of a step): list l has list_rows[l] rows. */
int bess_sparse_sgd_lists(int32_t dtype, int32_t width, void* table, int32_t n_lists,
                          const int32_t* const* list_idx, const float* const* list_grad,
                          const int64_t* list_rows, float lr, void* stream);
/* The same with a dense `axpy_table[i] += axpy_alpha * axpy_grad[i]` (axpy_n elements, axpy_table of the table's dtype:
 * the relation table's plain SGD step) done by spare workgroups of the same launch. */
int bess_sparse_sgd_lists_axpy(int32_t dtype, int32_t width, void* table, int32_t n_lists,
                               const int32_t* const* list_idx, const float* const* list_grad,
                               const int64_t* list_rows, float lr, void* axpy_table, const float* axpy_grad,
                               int64_t axpy_n, float axpy_alpha, void* stream);

/* K9 without atomics (n_shard == 1, per-triple negatives read straight from
 * the shard): group the n_refs references idx[i] by destination row with a
 * stable radix sort (bitwise reproducible sums), reduce each group on chip,
 * write one gradient row per unique destination, apply the optimiser to those.
 *
 * bess_segment_index_workspace: bytes of scratch needed for n_refs references.
 * bess_build_segment_index: row_bits = bits needed for the largest row id;
 *   refs_sorted [n_refs] (reference ids ordered by row, ties in reference order),
 *   seg_rows [n_refs] (row of each segment), seg_offsets [n_refs + 1],
 *   n_seg [1] (device scalar: number of unique rows, read by the consumers so
 *   that no host synchronisation is needed);
 *   long_segs (optional, int32 [long_cap + 1], long_cap >= n_refs / BESS_SEGMENT_CAP + 1):
 *   [0] = number of segments with more than BESS_SEGMENT_CAP references, [1..] their ids
 *   (any order) - rows that very many references point at (padded candidate lists, hot
 *   entities), which bess_neg_pertriple_grad_segments spreads over the whole device. */
#define BESS_SEGMENT_CAP 256
#define BESS_MAX_WORD_JOBS 8
#define BESS_SMALL_INDEX_MAX 15360 /* row ids bess_step_prologue indexes (one workgroup) */
int bess_segment_index_workspace(int64_t n_refs, size_t* bytes);
int bess_build_segment_index(const int32_t* idx, int64_t n_refs, int32_t row_bits,
                             int32_t* refs_sorted, int32_t* seg_rows,
                             int32_t* seg_offsets, int32_t* n_seg, int32_t* long_segs,
                             int64_t long_cap, void* workspace,
                             size_t workspace_bytes, void* stream);

/* Rows [*n_seg, max_seg) of (seg_rows, grad_seg [max_seg, width]) become copies of the last real row with zero
 * gradient rows: the segment list can then be passed on with its host-known length max_seg (e.g. as one of the
 * lists of bess_coalesced_update) without reading *n_seg back. */
int bess_pad_segments(int32_t* seg_rows, const int32_t* n_seg, int64_t max_seg, float* grad_seg, int32_t width,
                      void* stream);

/* Prologue of a notebook-size step in ONE launch (the step is launch-bound there: reference micro-batch
 * S = 512, K = 32 of notebooks/3_wikikg2_fp16.ipynb:251-256):
 *   - up to BESS_MAX_WORD_JOBS copy / fill jobs over 32-bit words: job j writes job_words[j] words at
 *     job_dst[j], copied from job_src[j] PLUS job_value[j] (0 for a plain copy; source == destination with
 *     value 1 increments a device-side counter: the generation of bess_direct_update) or, when job_src[j] is
 *     NULL, set to job_value[j] (the concatenated candidate list of an augmented step = the `torch.concat` of
 *     bess.py:369-393; zeroed gradient buffers);
 *   - the segment index (as bess_build_segment_index builds it) of the concatenation of up to
 *     BESS_MAX_ROW_LISTS row-id lists, read where they are (1 .. BESS_SMALL_INDEX_MAX ids in all; n_lists = 0:
 *     no index).  refs_sorted / seg_rows: int32 [n_ids], seg_offsets: [n_ids + 1], n_seg: [1],
 *     long_segs: optional, [long_cap + 1].
 * Jobs and index are independent of each other (different workgroups of the same launch): no job may
 * produce an id list of the same call. */
int bess_step_prologue(int32_t n_jobs, void* const* job_dst, const void* const* job_src,
                       const uint32_t* job_value, const int64_t* job_words, int32_t n_lists,
                       const int32_t* const* id_lists, const int64_t* id_lens, int32_t row_bits,
                       int32_t* refs_sorted, int32_t* seg_rows, int32_t* seg_offsets, int32_t* n_seg,
                       int32_t* long_segs, int64_t long_cap, void* stream);

/* grad_seg[s, :] (f32 [max_seg, W], rows >= *n_seg untouched) = sum over the
 * references (q, k) of segment s of d score(query[q], e) / d e * d_out[q, k],
 * e = table[seg_rows[s]].  Same arithmetic as bess_neg_score_pertriple_bwd's
 * d_neg, summed per destination row; nothing of size [n_refs, W] is written.
 * With grad_seg == NULL the SGD step is fused: table[seg_rows[s]] -=
 * fused_sgd_lr * (that sum), each row read and written by its one owner (use
 * only when no later gradient computation still needs the old rows).
 * long_segs / long_cap as filled by bess_build_segment_index, plus long_grad (f32
 * [long_cap, W]) and long_count (int32 [long_cap]) - scratch that must be ALL ZERO before
 * the first call and is left ALL ZERO by every call (the group that finishes a row zeroes its
 * sums and its counter), so one allocation serves every later step, whatever its index -
 * or NULL / 0 / NULL / NULL: segments beyond BESS_SEGMENT_CAP references are then
 * summed by all workgroups together (partial sums through float atomics into long_grad,
 * the group that adds the last slice of a row writes it out) instead of by one 16-lane
 * group each. */
int bess_neg_pertriple_grad_segments(const bess_model_desc* d, const float* query,
                                     int64_t n_query, void* table, int64_t n_neg,
                                     const float* d_out, int64_t ld_dout,
                                     const int32_t* refs_sorted, const int32_t* seg_rows,
                                     const int32_t* seg_offsets, const int32_t* n_seg,
                                     int64_t max_seg, float* grad_seg,
                                     float fused_sgd_lr, const int32_t* long_segs,
                                     int64_t long_cap, float* long_grad, int32_t* long_count,
                                     void* stream);

/* K10 on unique rows: table[seg_rows[s], :] -= lr * grad_seg[s, :], s < *n_seg
 * (plain read-modify-write: one rounding per row and step, also for f16). */
int bess_apply_segments_sgd(int32_t dtype, int32_t width, void* table,
                            const int32_t* seg_rows, const int32_t* n_seg,
                            int64_t max_seg, const float* grad_seg, float lr,
                            void* stream);

/* generic K9: grad_seg[s, :] = sum of src[refs_sorted[r], :] over the references r
 * of segment s - coalesces any list of (row, gradient row) contributions
 * (heads + tails + shared negatives ...) indexed by bess_build_segment_index. */
int bess_segment_sum_rows(int32_t width, const float* src, const int32_t* refs_sorted,
                          const int32_t* seg_offsets, const int32_t* n_seg,
                          int64_t max_seg, float* grad_seg, void* stream);

/* K10 on unique rows with per-row state (f32 [M, W]): SGD(+momentum, weight
 * decay), Adagrad, Adam / AdamW with "lazy" semantics - only rows touched by
 * the step move (torch.optim.SparseAdam / sparse Adagrad); replaces the dense
 * poptorch.optim step of the notebooks, which cannot be afforded on a shard of
 * tens of GB.  state1 / state2 may be NULL when the optimiser has no such state.
 * keep (optional, int32 [max_seg]): only segments with keep[s] != 0 are updated. */
int bess_apply_segments_opt(const bess_opt_desc* o, int32_t dtype, int32_t width,
                            void* table, const int32_t* seg_rows, const int32_t* n_seg,
                            int64_t max_seg, const float* grad_seg, float* state1,
                            float* state2, const int32_t* keep, void* stream);

/* K9 + K10 for the SMALL lists of a step (heads, tails, shared negatives, rows returned by the
 * backward all-to-all ...; autograd index_put_(accumulate) + optimiser): the lists' row ids are
 * concatenated and indexed by bess_build_segment_index; reference r of the concatenation is
 * gradient row (r - first[l]) of list l.  Every unique row is summed in reference order (bitwise
 * reproducible) straight from the n_lists gradient arrays list_grad[l] (f32 [list_rows[l], width])
 * and updated in the same pass: one read, one optimiser step, ONE store per touched row - so an
 * f16 table is rounded once per step (a packed-f16 atomic add rounds per contribution).
 * keep as in bess_apply_segments_opt.  With sum_out (f32 [max_seg, width]) != NULL the per-row sums
 * are written there instead and nothing is updated (o, table, state may then be NULL). */
int bess_coalesced_update(const bess_opt_desc* o, int32_t dtype, int32_t width, void* table,
                          int32_t n_lists, const float* const* list_grad, const int64_t* list_rows,
                          const int32_t* refs_sorted, const int32_t* seg_rows,
                          const int32_t* seg_offsets, const int32_t* n_seg, int64_t max_seg,
                          float* state1, float* state2, const int32_t* keep, float* sum_out,
                          void* stream);

/* The same launch, with a dense `axpy_table[i] += axpy_alpha * axpy_grad[i]` (i < axpy_n; axpy_table of the
 * same dtype as `table`) done by extra workgroups: the plain-SGD step of the replicated relation table rides
 * along with the shard's update (one dispatch less per notebook-size step).  The two tables must not overlap. */
int bess_coalesced_update_axpy(const bess_opt_desc* o, int32_t dtype, int32_t width, void* table,
                               int32_t n_lists, const float* const* list_grad, const int64_t* list_rows,
                               const int32_t* refs_sorted, const int32_t* seg_rows,
                               const int32_t* seg_offsets, const int32_t* n_seg, int64_t max_seg,
                               float* state1, float* state2, const int32_t* keep, float* sum_out,
                               void* axpy_table, const float* axpy_grad, int64_t axpy_n, float axpy_alpha,
                               void* stream);

/* Paged optimiser state (a shard of tens of GB cannot carry one or two fp32 state tables of its own size:
 * BASELINE configs[4] is 128 GB per shard): state tables of `capacity` rows, a table row gets one the first
 * time it is stepped.  For every unique row seg_rows[s], s < *n_seg (with keep[s] != 0 where keep is given)
 * that has slot_map[row] < 0: slot_map[row] = counter[0]++ while that is below capacity.  The state pools
 * must be zero when a slot is first used (allocate them zeroed).  counter[0] > capacity afterwards tells
 * the host that the pool is exhausted. */
int bess_assign_state_rows(const int32_t* seg_rows, const int32_t* n_seg, int64_t max_seg,
                           const int32_t* keep, int32_t* slot_map, int32_t* counter, int64_t capacity,
                           void* stream);

/* K9 + K10 in one pass for the stateful optimisers (what bess_neg_pertriple_grad_segments
 * with fused_sgd_lr is for plain SGD): the summed gradient of each unique row is consumed by
 * the optimiser step where it is formed - no [n_seg, W] gradient is written or re-read.
 * The other contributions to the same table (heads, tails, shared negatives ...) come as
 * their own per-unique-row sums extra_sum [n_extra, W] with extra_map [max_seg]
 * (bess_map_extra_rows: extra_map[s] = row of extra_sum that belongs to segment s, or -1),
 * so that a row touched from both sides still gets ONE update with its total gradient;
 * bess_map_extra_rows also returns keep [max_extra] = 1 for the extra rows that are no
 * segment of the big index - update those with bess_apply_segments_opt(..., keep).
 * TransE / RotatE / DistMult / ComplEx; long_* as in bess_neg_pertriple_grad_segments. */
int bess_neg_pertriple_step_segments(const bess_model_desc* d, const float* query,
                                     int64_t n_query, void* table, int64_t n_neg,
                                     const float* d_out, int64_t ld_dout,
                                     const int32_t* refs_sorted, const int32_t* seg_rows,
                                     const int32_t* seg_offsets, const int32_t* n_seg,
                                     int64_t max_seg, const int32_t* long_segs,
                                     int64_t long_cap, float* long_grad, int32_t* long_count,
                                     const bess_opt_desc* o, float* state1, float* state2,
                                     const int32_t* extra_map, const float* extra_sum,
                                     void* stream);
int bess_map_extra_rows(const int32_t* seg_rows, const int32_t* n_seg, int64_t max_seg,
                        const int32_t* extra_rows, const int32_t* n_extra, int64_t max_extra,
                        int32_t* extra_map, int32_t* keep, void* stream);

/* dense axpy on a replicated table: table -= lr * grad (relation table) */
int bess_dense_sgd(int32_t dtype, void* table, const float* grad, int64_t n_elem,
                   float lr, void* stream);

/* Fused training forward of the per-triple regime: the scores of bess_neg_score_pertriple_fwd
 * *and* d loss / d query, in one pass over the negative rows (online-softmax accumulation of the
 * loss weights; see csrc/neg_pertriple.hip).  Valid when nothing is masked out of the scores
 * afterwards (no negative_mask, no augmentation) and the loss is taken over exactly these n_neg
 * scores; TransE / RotatE / DistMult / ComplEx.  pos [n_query] (needed by margin ranking and ssce),
 * weight [1 | n_query] = triple weights.  state_ml [n_query, items, 2] and state_acc
 * [n_query, items, W] are scratch, items from bess_neg_pertriple_items(). */
int bess_neg_pertriple_items(const bess_model_desc* d, int64_t n_query, int64_t n_neg, int32_t* items);
int bess_neg_score_pertriple_fwd_dq(const bess_model_desc* d, const bess_loss_desc* l,
                                    const float* query, int64_t n_query, const void* neg_base,
                                    const int32_t* neg_idx, int64_t n_neg, const float* pos,
                                    const float* weight, int64_t weight_len, float* out,
                                    int64_t ld_out, float* d_query, float* state_ml,
                                    float* state_acc, void* stream);
/* The same with K7's mask inside the pass: mask [mask_rows (1 | n_query), mask_cols] (bytes, 0 = masked
 * out) over the LAST mask_cols columns, as bess_mask_scores applies it - a masked-out candidate gets
 * BESS_BAD_NEGATIVE_SCORE added before the score is stored and before it enters the loss weights
 * (the padding mask of triple-specific negatives, negative_sampler.py:479-540); mask == NULL: none. */
int bess_neg_score_pertriple_fwd_dq_masked(const bess_model_desc* d, const bess_loss_desc* l,
                                           const float* query, int64_t n_query, const void* neg_base,
                                           const int32_t* neg_idx, int64_t n_neg, const float* pos,
                                           const float* weight, int64_t weight_len, const uint8_t* mask,
                                           int64_t mask_rows, int64_t mask_cols, float* out,
                                           int64_t ld_out, float* d_query, float* state_ml,
                                           float* state_acc, void* stream);

/* d_query == NULL (either form): the per-item partials are left in state_ml / state_acc for bess_pertriple_tail.
 *
 * bess_pertriple_tail: the per-triple remainder of such a training step in ONE launch - what the reference's
 * autograd graph does between the negative scores and the embedding gradients for one micro-batch
 * (reference besskge/bess.py:248-262 loss, 322-468 backward of K3 / K6; loss.py:28-251):
 *   d loss / d query from the partials (never stored unless d_query != NULL),
 *   K8 (as bess_loss_fwd_bwd: row_loss [n], loss [1], d_pos [n], d_neg [n, n_neg]),
 *   K3' + K6' (as bess_query_triple_bwd: d_head / d_tail [n, W], relation gradients ADDED into d_rel_table).
 * Results are those of bess_neg_score_pertriple_fwd_dq + bess_loss_fwd_bwd + bess_query_triple_bwd, to the bit
 * (relation gradients: fp32 atomics, order-dependent in both).  n_neg % 4 == 0, n_neg <= 3072, 16-byte aligned
 * score rows, TransE / RotatE / DistMult / ComplEx (bess_pertriple_tail_supported; else BESS_EUNSUPPORTED).
 * counter: int32 [BESS_TICKET_INTS], zero on entry, left zero (the last workgroup sums the row terms in a fixed order). */
int bess_pertriple_tail_supported(const bess_model_desc* d, int64_t n_neg);
int bess_pertriple_tail(const bess_model_desc* d, const bess_loss_desc* l, int32_t side, const void* head_base,
                        const int32_t* head_idx, const void* tail_base, const int32_t* tail_idx,
                        const void* rel_table, const int32_t* rel_idx, int64_t n_triple,
                        const float* state_ml, const float* state_acc, int32_t items, const float* pos,
                        const float* neg, int64_t n_neg, int64_t ld_neg, const float* weight,
                        int64_t weight_len, float* row_loss, float* loss, float* d_pos, float* d_neg,
                        int64_t ld_dneg, float* d_query, float* d_head, float* d_tail, float* d_rel_table,
                        int32_t* counter, void* stream);

/* The same pass for queries whose negatives are spread over several shards (ScoreMoving: each shard
 * scores the gathered queries against its own rows).  bess_neg_score_pertriple_fwd_partials writes the
 * scores and leaves the per-item partials (m_i, l_i, acc_i) of the online softmax in state_ml / state_acc
 * (log-sigmoid and sampled-softmax losses; margin ranking weighs by the positive score, which only the
 * triple's own shard has: BESS_EUNSUPPORTED).  Once the owner of the triples has seen all scores it sends
 * back, per query, norm = (m, L / C_q): the maximum and the normaliser of beta * (score + shift) over ALL
 * negatives of the query (and the positive, sampled softmax), C_q = loss_scale * w_q (x 1/2 log-sigmoid);
 * bess_combine_dq_partials then gives this shard's share of d loss / d query,
 *   d_query[q] = C_q / L * sum_i exp(m_i - m) acc_i,
 * without a second pass over the negative rows. */
int bess_neg_score_pertriple_fwd_partials(const bess_model_desc* d, const bess_loss_desc* l,
                                          const float* query, int64_t n_query, const void* neg_base,
                                          const int32_t* neg_idx, int64_t n_neg, float* out,
                                          int64_t ld_out, float* state_ml, float* state_acc, void* stream);
int bess_combine_dq_partials(const float* state_ml, const float* state_acc, int64_t n_query,
                             int32_t items, int32_t width, const float* norm, float* d_query,
                             void* stream);

/* `torch.nn.functional.normalize(part, p=2, dim=-1)` of every d-wide part of the rows
 * (scoring.py:549-551 and the like), fused with the gather and the f32 conversion:
 *   out[i, p*d + w] = row_i[p*d + w] * inv[i, p],  inv = 1 / max(||part||_2, 1e-12)
 * (inv = 1 when normalize == 0: plain gather + convert).  inv_norm may be NULL. */
int bess_normalize_rows(int32_t dtype, const void* base, const int32_t* idx, int64_t n_rows,
                        int32_t width, int32_t n_part, int32_t normalize, float* out,
                        float* inv_norm, void* stream);
/* its backward: d_rows = inv * (d_hat - hat * <hat, d_hat>) per part */
int bess_normalize_rows_bwd(const float* hat, const float* inv_norm, const float* d_hat,
                            int64_t n_rows, int32_t width, int32_t n_part, float* d_rows,
                            void* stream);

/* ---- device-side index sampling (bit-exact numpy streams) -------------------
 * `jump_table` (device, uint64 [64][4] = {a_hi, a_lo, c_hi, c_lo}): the affine map
 * s -> a*s + c of 2^j generator steps for this generator's increment.  The host
 * passes the generator state *before* the call and advances its own copy; the
 * kernels never write generator state. */

/* `RandomShardedNegativeSampler.__call__` (negative_sampler.py:104-132) and, with
 * `wanted_type`, `TypeBasedShardedNegativeSampler.__call__` (180-230):
 *   out = rng.integers(1 << 31, size=[n_step, n_shard, n_shard, B, K]).astype(int32)
 *         % shard_counts[src]  ( % type_counts[src, ty] + type_offsets[src, ty] )
 * restricted to source shards [src_begin, src_begin + src_count): out is
 * [n_step, src_count, n_shard, B, K].  wanted_type [n_step, n_shard, B] is the type
 * every triple of every scoring shard wants; the scoring shard of block [src, dst]
 * is src when local_sampling, dst otherwise. */
int bess_sample_negatives(const bess_pcg64_state* gen, const uint64_t* jump_table, int64_t n_step,
                          int32_t n_shard, int32_t src_begin, int32_t src_count, int64_t B, int64_t K,
                          const int32_t* shard_counts, const int32_t* wanted_type,
                          const int32_t* type_counts, const int32_t* type_offsets, int32_t n_type,
                          int32_t local_sampling, int32_t* out, void* stream);

/* `TripleBasedShardedNegativeSampler.__call__` (negative_sampler.py:422-477): per-step look-up of the fixed
 * candidate lists and their layout for the exchange.  table_h (int32 [n_list, n_neg_shard, list_len]: local
 * rows per owning shard, padded; mask_h uint8 of the same shape marks the real entries) is made once by the
 * host sampler and kept in HBM; lookup (int64 [n_step, n_shard, n_triple]) names the list of every sampled
 * triple (n_triple = the (block, triple) axes folded).  Outputs (either may be NULL):
 *   entities int32 [n_step, n_neg_shard, n_shard, n_triple, list_len]   (what the gathering shard needs)
 *   mask     uint8, that layout (mask_gather_layout != 0) or [n_step, n_shard, n_triple, n_neg_shard, list_len]
 * With table_t != NULL ("ht"): in every block of per_part triples the first `half` take table_h / mask_h
 * (heads are corrupted), the rest table_t / mask_t. */
int bess_gather_candidate_lists(const int32_t* table_h, const int32_t* table_t, const uint8_t* mask_h,
                                const uint8_t* mask_t, int64_t n_list, const int64_t* lookup,
                                int64_t n_step, int64_t n_shard, int64_t n_triple, int64_t per_part,
                                int64_t half, int64_t n_neg_shard, int64_t list_len,
                                int32_t mask_gather_layout, int32_t* entities, uint8_t* mask, void* stream);

/* `RandomShardedBatchSampler.sample_triples` (batch_sampler.py:373-399):
 *   out[f] = offsets[b] + rng.integers(1 << 63, size=n_out)[f] % counts[b],
 *   b = (f / inner) % n_bucket   (out flat over [n_step, buckets..., inner]) */
int bess_sample_bucket_indices(const bess_pcg64_state* gen, const uint64_t* jump_table, int64_t n_out,
                               int64_t inner, int64_t n_bucket, const int64_t* counts,
                               const int64_t* offsets, int64_t* out, void* stream);

/* `hrt = self.triples[sample_idx]` + the tail block transpose of
 * `ShardedBatchSampler.__getitem__` (batch_sampler.py:150-167).  triples int32
 * [n_triple, 3]; sample_idx int64 [n_step, n1, n2, per_part]; head / relation in
 * that layout, tail as [n_step, n2, n1, per_part] when swap_tail.  Any output may
 * be NULL. */
int bess_lookup_triples(const int32_t* triples, int64_t n_triple, const int64_t* sample_idx,
                        int64_t n_step, int64_t n1, int64_t n2, int64_t per_part, int32_t swap_tail,
                        int32_t* head, int32_t* relation, int32_t* tail, void* stream);

/* ---- collectives between shards (RCCL over xGMI) -----------------------------
 * Replace `poptorch_experimental_addons.collectives` (reference bess.py:14-19):
 * `all_to_all_single_cross_replica` (call sites bess.py:346-350, 583-595),
 * `all_gather_cross_replica` (bess.py:519-545) and PopTorch's implicit sum of the
 * replicated parameters' gradients.  One process per GPU, rank == shard.  A communicator
 * is the library's only global state; it is created and destroyed by the caller and
 * belongs to the device that was current at bess_comm_init_rank.  Every collective is
 * asynchronous on the given hipStream_t - pass the stream the kernels run on: a step is
 * then ONE in-order queue of kernels and collectives, and can be captured into a hipGraph.
 * Every rank must issue the same collectives in the same order.
 *
 * bess_comm_unique_id: 128 opaque bytes made by ONE rank and handed to all others by the
 *   host (torch.distributed store, MPI, a file ...), then bess_comm_init_rank on every rank
 *   (blocks until all `world` ranks have called it).
 * bess_comm_init_all: all n communicators of a single-process job at once (dev_ids NULL:
 *   devices 0..n-1); comms[i] is rank i.
 * bess_comm_destroy: after the stream work that uses the communicator has drained AND after every hipGraph
 *   that recorded its collectives has been destroyed (ncclCommDestroy waits for those graphs: a host hang,
 *   not an error, if one is still alive). */
#define BESS_COMM_ID_BYTES 128
typedef struct bess_comm bess_comm;
int bess_comm_unique_id(uint8_t* id);
int bess_comm_init_rank(int32_t world, int32_t rank, const uint8_t* id, bess_comm** comm);
int bess_comm_init_all(int32_t n, const int32_t* dev_ids, bess_comm** comms);
int bess_comm_destroy(bess_comm* comm);
int bess_comm_info(const bess_comm* comm, int32_t* world, int32_t* rank, int32_t* device);

/* C1 / C4 / C5 / C8 - balanced all-to-all: block p (bytes_per_peer bytes) of `send` goes to
 * rank p, block p of `recv` came from rank p (one grouped send / recv per peer: each
 * block rides its own xGMI link).  send != recv. */
int bess_alltoall(bess_comm* comm, const void* send, void* recv, int64_t bytes_per_peer, void* stream);
/* C2 / C3 - all-gather: recv [world, bytes] in rank order. */
int bess_allgather(bess_comm* comm, const void* send, void* recv, int64_t bytes, void* stream);
/* C9 - sum over ranks of the replicated tables' gradients (in place when send == recv). */
int bess_allreduce_sum_f32(bess_comm* comm, const float* send, float* recv, int64_t n, void* stream);
/* K1 + C1 in one call (bess.py:332-350): rows table[idx[p * rows_per_peer + i]] are packed
 * into send [world, rows_per_peer, width] and exchanged into recv (same shape: block p =
 * the rows rank p packed for this rank).  Row size must be a multiple of 16 bytes. */
int bess_pack_exchange(bess_comm* comm, int32_t dtype, int32_t width, const void* table,
                       const int32_t* idx, int64_t rows_per_peer, void* send, void* recv,
                       void* stream);

/* K4 + K7 + K8 of a training step with shared negatives behind one call (scoring.py:194-197 / 251-252 +
 * bess.py:182-245 + loss.py:28-251): out [n_query, ld_out] receives the (masked) scores as from
 * bess_neg_score_shared_fwd_masked (kill may be NULL), then row_loss / loss / d_pos / d_neg as from
 * bess_loss_fwd_bwd_one_launch with pos = the positive scores: the scoring launch followed by the loss launch(es),
 * issued by one call (a step driven from C or from a recorded plan makes one call less; a variant that finished
 * the loss rows inside the packed L1 kernel - its last workgroup per block of rows - was built and measured in
 * round 4: 49 us against 12.7 + 9.4 at the notebook micro-batch, the rows of a block then run one after the other
 * on four waves).  counters: int32 [BESS_TICKET_INTS] on the device, zero on entry,
 * left zero (keep one array per stream). */
int bess_neg_score_shared_fwd_loss(const bess_model_desc* desc, const float* query, int64_t n_query,
                                   const void* neg_base, const int32_t* neg_idx, int64_t n_neg, float* out,
                                   int64_t ld_out, const bess_kill_desc* kill, const bess_loss_desc* loss_desc,
                                   const float* pos, const float* weight, int64_t weight_len, float* row_loss,
                                   float* loss, float* d_pos, float* d_neg, int64_t ld_dneg, int32_t* counters,
                                   void* workspace, int64_t workspace_bytes, void* stream);

/* bess_query_triple_fwd with the copy / fill jobs of bess_step_prologue (same arrays, same meaning) run by spare
 * workgroups of the SAME launch: the prologue of a training step whose update needs no index costs no launch of
 * its own.  Nothing this launch reads may be written by a job. */
int bess_query_triple_fwd_jobs(const bess_model_desc* desc, int32_t side, const void* head_base,
                               const int32_t* head_idx, const void* tail_base, const int32_t* tail_idx,
                               const void* rel_table, const int32_t* rel_idx, int64_t n_triple, float* query,
                               float* out, int32_t n_jobs, void* const* job_dst, const void* const* job_src,
                               const uint32_t* job_value, const int64_t* job_words, void* stream);

/* The two products of the shared-negative backward WITHOUT atomics, and their consumer (TransE / RotatE with p = 1:
 * the kernel that forms both products from one evaluation of sgn(q - e); bess_neg_score_shared_bwd_parts_plan
 * returns 0 parts for everything else).  bess_neg_score_shared_bwd_parts writes
 *     dq_parts   f32 [n_dq_parts,   n_query, W]   d_query   = their sum over the first axis
 *     dneg_parts f32 [n_dneg_parts, n_neg,   W]   d_neg     = likewise
 * (every element written, nothing to clear; part counts from the _plan call for the same desc and sizes) with plain
 * stores.  bess_query_triple_bwd_parts is bess_query_triple_bwd BY ROW, fed by them: gradient rows are ADDED (fp32
 * atomics) into accumulators over the row spaces of the tables, at the row ids the references name;
 * the triple's wave sums its d_query row from the parts as it reads it; spare workgroups add every candidate's
 * summed gradient row into acc_neg at row neg_idx[j] (one atomic per candidate and column, not one per query
 * slice).  acc_head / acc_tail / acc_neg: accumulators over the row spaces of head_base / tail_base / the candidates'
 * table (bess_direct_update).  Reference: autograd of pea.distance_matrix + index_put_ (scoring.py:194-197). */
int bess_neg_score_shared_bwd_parts_plan(const bess_model_desc* desc, int64_t n_query, int64_t n_neg,
                                         int32_t* n_dq_parts, int32_t* n_dneg_parts);
int bess_neg_score_shared_bwd_parts(const bess_model_desc* desc, const float* query, int64_t n_query,
                                    const void* neg_base, const int32_t* neg_idx, int64_t n_neg, const float* d_out,
                                    int64_t ld_dout, float* dq_parts, float* dneg_parts, void* stream);
int bess_query_triple_bwd_parts(const bess_model_desc* desc, int32_t side, const void* head_base,
                                const int32_t* head_idx, const void* tail_base, const int32_t* tail_idx,
                                const void* rel_table, const int32_t* rel_idx, int64_t n_triple, const float* d_out,
                                const float* dq_parts, int32_t n_dq_parts, const float* dneg_parts,
                                int32_t n_dneg_parts, int64_t n_neg, const int32_t* neg_idx, float* acc_head,
                                float* acc_tail, float* acc_neg, float* d_rel_table, void* stream);

/* K9 + K10 without an index (direct-addressed accumulation): for a table whose fp32 image the caller can afford
 * as scratch.  acc [rows of table, width] f32 is ZERO between steps; the step's backward kernels add their
 * gradient rows into it at the rows' ids (bess_query_triple_bwd_parts; bess_scatter_add_rows / bess_sparse_sgd_lists
 * with lr = -1 for gradients that exist as dense lists).  bess_direct_update then visits the step's row-id lists
 * (n_lists <= BESS_MAX_ROW_LISTS, read where they are): one wave per reference claims its row - claim
 * [rows of table] int32 holds the generation number of the last step that updated the row, *generation is this
 * step's (the caller increments it once per step before the call, e.g. with a copy job of bess_step_prologue
 * whose source is its destination and whose value is 1) - the first reference of a row applies the optimiser
 * to table[row] with acc[row] as the summed gradient (one read-modify-write, one rounding per row and step; state1
 * / state2 as for bess_coalesced_update, full-size state only) and zeroes acc[row]; the others leave.  Replaces
 * bess_build_segment_index + bess_coalesced_update for the small lists of a step (reference: autograd's
 * index_put_(accumulate) + the optimiser step).  axpy_*: as bess_coalesced_update_axpy. */
int bess_direct_update(const bess_opt_desc* opt, int32_t dtype, int32_t width, void* table, int32_t n_lists,
                       const int32_t* const* id_lists, const int64_t* id_lens, float* acc, int32_t* claim,
                       const int32_t* generation, float* state1, float* state2, void* axpy_table,
                       const float* axpy_grad, int64_t axpy_n, float axpy_alpha, void* stream);
/* ---- step plans -----------------------------------------------------------------
 * A step as a recorded list of this library's calls, replayed from C (csrc/plan.hip).  The reference's step is
 * compiled once by PopTorch and then runs without Python (bess.py:322-468 under poptorch.trainingModel,
 * `device_iterations` micro-batches per host call); here the host program runs the step once while noting every call
 * that enqueues work - bess_plan_add_call: the entry point's name and, per argument, its kind and value - and
 * bess_plan_run issues the same calls again on the given stream: kernels and collectives (bess_pack_exchange,
 * bess_alltoall, bess_allreduce_sum_f32 ...) alike, from any host thread (one per device for a process that drives
 * several GPUs).  Argument kinds: INT (values[k] = the integer), FLOAT (values[k] = the bits of a double), PTR
 * (values[k] = an address that stays valid: device buffers, communicators), BLOB (blobs[k] / blob_bytes[k]: host
 * bytes the call reads - descriptors, arrays of pointers / sizes - copied into the plan), STREAM (the last argument
 * of every recordable entry point: replaced by bess_plan_run's).  A plan owns nothing on the device: it is valid
 * while the buffers it names are.  bess_plan_knows: 1 for the names of entry points that can be part of a plan. */
#define BESS_PLAN_ARG_INT 0
#define BESS_PLAN_ARG_FLOAT 1
#define BESS_PLAN_ARG_PTR 2
#define BESS_PLAN_ARG_BLOB 3
#define BESS_PLAN_ARG_STREAM 4
typedef struct bess_plan bess_plan;
int bess_plan_create(bess_plan** plan);
int bess_plan_destroy(bess_plan* plan);
int bess_plan_knows(const char* name);
int bess_plan_length(const bess_plan* plan);
int bess_plan_add_call(bess_plan* plan, const char* name, int32_t n_args, const uint8_t* kinds,
                       const uint64_t* values, const void* const* blobs, const int64_t* blob_bytes);
int bess_plan_run(const bess_plan* plan, void* stream);

/* ---- recorded steps -----------------------------------------------------------
 * A step of this library can be captured into a hipGraph as a whole (every call is asynchronous on the
 * caller's stream, every clear is a kernel).  bess_graph_node_counts tells what a captured graph holds:
 * counts[t] = nodes of hipGraphNodeType t (0 kernel, 1 memcpy, 2 memset, 3 host, 4 child graph, 5 empty ...;
 * t < n_kinds; nodes of child graphs are counted too).  Host-side inspection, no stream work. */
#define BESS_GRAPH_NODE_KERNEL 0
#define BESS_GRAPH_NODE_MEMCPY 1
#define BESS_GRAPH_NODE_MEMSET 2
int bess_graph_node_counts(void* graph, int32_t* counts, int32_t n_kinds);

#ifdef __cplusplus
}
#endif
#endif /* BESSKGE_HIP_H */

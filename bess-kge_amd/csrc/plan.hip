// Step plans: a training / scoring step as a recorded list of this library's own calls, replayed from C.
//
// The reference hands its step to PopTorch, which compiles it once and then runs `device_iterations` of it per
// host call with no Python in between (reference besskge/bess.py:322-468 is traced, not interpreted, per step).
// Here a step is 6 - 25 calls of this C ABI; issued from Python an eager notebook-size step spends 0.25 ms of
// host time on 0.07 ms of kernels (~900 Python-level calls), and two replicas in lock-step 0.7 ms on 0.2.
// hipGraph replay removes that for one process per GPU - but recording RCCL's send / recv into a graph has never
// met a peer, and a graph cannot be driven from several host threads at once.  A plan is the same idea one level
// up: the host program (besskge/_native.py: `record_plan`) runs the step ONCE through the library while every
// call that enqueues work is noted - entry point, argument values, the bytes of the descriptors and pointer
// arrays it was given - and `bess_plan_run` issues the same calls again on the stream it is handed: kernels,
// bess_pack_exchange, bess_alltoall, bess_allreduce_sum_f32 alike, no Python, no GIL.  Like a recorded graph a
// plan owns nothing on the device and is only valid while the buffers it names are (the host program keeps them
// in a private memory pool and copies new inputs into the step's input buffers).
#include <string.h>

#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "common.h"

namespace bess {

union Slot {
    int64_t i;
    double d;
    void* p;
};

constexpr int PLAN_MAX_ARGS = 32;

template <typename T>
static T slot_get(const Slot& s) {
    if constexpr (std::is_pointer_v<T>) return reinterpret_cast<T>(s.p);
    else if constexpr (std::is_floating_point_v<T>) return static_cast<T>(s.d);
    else return static_cast<T>(s.i);
}
template <typename... A, size_t... I>
static int invoke(int (*f)(A...), const Slot* s, std::index_sequence<I...>) {
    return f(slot_get<A>(s[I])...);
}
template <typename... A>
static int invoke(int (*f)(A...), const Slot* s) {
    return invoke(f, s, std::index_sequence_for<A...>{});
}
template <typename... A>
static constexpr int arity(int (*)(A...)) {
    return static_cast<int>(sizeof...(A));
}

struct PlanFn {
    const char* name;
    int n_args;
    int (*thunk)(const Slot*);
};

// every entry point that enqueues work on a stream (their last argument); host-side queries and the communicator's
// life cycle are not part of a step
#define BESS_PLAN_FN(fn) \
    PlanFn { #fn, arity(&fn), [](const Slot* s) -> int { return invoke(&fn, s); } }
static const PlanFn PLAN_FNS[] = {
    BESS_PLAN_FN(bess_gather_rows),
    BESS_PLAN_FN(bess_score_triple_fwd),
    BESS_PLAN_FN(bess_score_triple_bwd),
    BESS_PLAN_FN(bess_query_fwd),
    BESS_PLAN_FN(bess_query_bwd),
    BESS_PLAN_FN(bess_query_triple_fwd),
    BESS_PLAN_FN(bess_query_triple_fwd_jobs),
    BESS_PLAN_FN(bess_query_triple_bwd),
    BESS_PLAN_FN(bess_query_triple_bwd_parts),
    BESS_PLAN_FN(bess_neg_score_pertriple_fwd),
    BESS_PLAN_FN(bess_neg_score_pertriple_bwd),
    BESS_PLAN_FN(bess_neg_score_pertriple_fwd_dq),
    BESS_PLAN_FN(bess_neg_score_pertriple_fwd_dq_masked),
    BESS_PLAN_FN(bess_neg_score_pertriple_fwd_partials),
    BESS_PLAN_FN(bess_combine_dq_partials),
    BESS_PLAN_FN(bess_pertriple_tail),
    BESS_PLAN_FN(bess_neg_score_shared_fwd),
    BESS_PLAN_FN(bess_neg_score_shared_fwd_ws),
    BESS_PLAN_FN(bess_neg_score_shared_fwd_masked),
    BESS_PLAN_FN(bess_neg_score_shared_fwd_loss),
    BESS_PLAN_FN(bess_neg_score_shared_fwd_pruned),
    BESS_PLAN_FN(bess_neg_score_shared_fwd_counts),
    BESS_PLAN_FN(bess_neg_score_shared_fwd_pairs),
    BESS_PLAN_FN(bess_neg_score_shared_bwd),
    BESS_PLAN_FN(bess_neg_score_shared_bwd_ws),
    BESS_PLAN_FN(bess_neg_score_shared_bwd_parts),
    BESS_PLAN_FN(bess_mask_scores),
    BESS_PLAN_FN(bess_loss_fwd_bwd),
    BESS_PLAN_FN(bess_loss_fwd_bwd_norm),
    BESS_PLAN_FN(bess_loss_fwd_bwd_one_launch),
    BESS_PLAN_FN(bess_topk_update),
    BESS_PLAN_FN(bess_topk_update_flagged),
    BESS_PLAN_FN(bess_ranks_from_scores),
    BESS_PLAN_FN(bess_ranks_from_indices),
    BESS_PLAN_FN(bess_scatter_add_rows),
    BESS_PLAN_FN(bess_sparse_sgd),
    BESS_PLAN_FN(bess_sparse_sgd_lists),
    BESS_PLAN_FN(bess_sparse_sgd_lists_axpy),
    BESS_PLAN_FN(bess_dense_sgd),
    BESS_PLAN_FN(bess_build_segment_index),
    BESS_PLAN_FN(bess_pad_segments),
    BESS_PLAN_FN(bess_step_prologue),
    BESS_PLAN_FN(bess_neg_pertriple_grad_segments),
    BESS_PLAN_FN(bess_neg_pertriple_step_segments),
    BESS_PLAN_FN(bess_apply_segments_sgd),
    BESS_PLAN_FN(bess_apply_segments_opt),
    BESS_PLAN_FN(bess_segment_sum_rows),
    BESS_PLAN_FN(bess_coalesced_update),
    BESS_PLAN_FN(bess_coalesced_update_axpy),
    BESS_PLAN_FN(bess_direct_update),
    BESS_PLAN_FN(bess_assign_state_rows),
    BESS_PLAN_FN(bess_map_extra_rows),
    BESS_PLAN_FN(bess_normalize_rows),
    BESS_PLAN_FN(bess_normalize_rows_bwd),
    BESS_PLAN_FN(bess_sample_negatives),
    BESS_PLAN_FN(bess_sample_bucket_indices),
    BESS_PLAN_FN(bess_lookup_triples),
    BESS_PLAN_FN(bess_gather_candidate_lists),
    BESS_PLAN_FN(bess_alltoall),
    BESS_PLAN_FN(bess_allgather),
    BESS_PLAN_FN(bess_allreduce_sum_f32),
    BESS_PLAN_FN(bess_pack_exchange),
};
#undef BESS_PLAN_FN

static const PlanFn* find_fn(const char* name) {
    for (const PlanFn& f : PLAN_FNS)
        if (strcmp(f.name, name) == 0) return &f;
    return nullptr;
}

struct PlanCall {
    const PlanFn* fn;
    uint8_t kind[PLAN_MAX_ARGS];
    Slot value[PLAN_MAX_ARGS];  // BESS_PLAN_ARG_BLOB: value.i = offset into the arena
};

}  // namespace bess

struct bess_plan {
    std::vector<bess::PlanCall> calls;
    std::vector<char> arena;  // the bytes of descriptors / pointer arrays the calls were given (16-byte aligned pieces)
};

using namespace bess;

extern "C" int bess_plan_create(bess_plan** plan) {
    BESS_REQUIRE(plan, "plan_create: NULL result pointer");
    *plan = new bess_plan();
    return BESS_OK;
}

extern "C" int bess_plan_destroy(bess_plan* plan) {
    delete plan;
    return BESS_OK;
}

extern "C" int bess_plan_knows(const char* name) { return name && find_fn(name) ? 1 : 0; }

extern "C" int bess_plan_length(const bess_plan* plan) { return plan ? static_cast<int>(plan->calls.size()) : 0; }

extern "C" int bess_plan_add_call(bess_plan* plan, const char* name, int32_t n_args, const uint8_t* kinds,
                                  const uint64_t* values, const void* const* blobs, const int64_t* blob_bytes) {
    BESS_REQUIRE(plan && name && (n_args == 0 || (kinds && values)), "plan_add_call: NULL pointer");
    const PlanFn* fn = find_fn(name);
    BESS_REQUIRE(fn, "plan_add_call: `%s` is not an entry point that enqueues work", name);
    BESS_REQUIRE(n_args == fn->n_args && n_args <= PLAN_MAX_ARGS, "plan_add_call: %s takes %d arguments, got %d", name,
                 fn->n_args, n_args);
    PlanCall c{};
    c.fn = fn;
    for (int k = 0; k < n_args; ++k) {
        c.kind[k] = kinds[k];
        switch (kinds[k]) {
            case BESS_PLAN_ARG_INT: c.value[k].i = static_cast<int64_t>(values[k]); break;
            case BESS_PLAN_ARG_FLOAT: memcpy(&c.value[k].d, &values[k], sizeof(double)); break;
            case BESS_PLAN_ARG_PTR: c.value[k].p = reinterpret_cast<void*>(static_cast<uintptr_t>(values[k])); break;
            case BESS_PLAN_ARG_STREAM: c.value[k].p = nullptr; break;
            case BESS_PLAN_ARG_BLOB: {
                BESS_REQUIRE(blobs && blob_bytes && blobs[k] && blob_bytes[k] > 0, "plan_add_call: %s argument %d: empty blob",
                             name, k);
                const size_t at = (plan->arena.size() + 15) & ~size_t(15);
                plan->arena.resize(at + static_cast<size_t>(blob_bytes[k]));
                memcpy(plan->arena.data() + at, blobs[k], static_cast<size_t>(blob_bytes[k]));
                c.value[k].i = static_cast<int64_t>(at);
                break;
            }
            default: return fail(BESS_EINVAL, "plan_add_call: %s argument %d: unknown kind %d", name, k, kinds[k]);
        }
    }
    BESS_REQUIRE(n_args > 0 && kinds[n_args - 1] == BESS_PLAN_ARG_STREAM, "plan_add_call: the last argument of %s is its stream",
                 name);
    plan->calls.push_back(c);
    return BESS_OK;
}

extern "C" int bess_plan_run(const bess_plan* plan, void* stream) {
    BESS_REQUIRE(plan, "plan_run: NULL plan");
    const char* base = plan->arena.data();
    for (const PlanCall& c : plan->calls) {
        Slot s[PLAN_MAX_ARGS];
        for (int k = 0; k < c.fn->n_args; ++k) {
            s[k] = c.value[k];
            if (c.kind[k] == BESS_PLAN_ARG_BLOB) s[k].p = const_cast<char*>(base) + c.value[k].i;
            else if (c.kind[k] == BESS_PLAN_ARG_STREAM) s[k].p = stream;
        }
        if (int rc = c.fn->thunk(s)) return rc;  // (the callee has set the error text)
    }
    return BESS_OK;
}

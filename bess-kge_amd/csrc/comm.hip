// C1-C9 of the BESS step on gfx950: the collectives between shards, over RCCL / xGMI.
//
// The reference uses two collectives of poptorch_experimental_addons
// (reference besskge/bess.py:14-19): `all_to_all_single_cross_replica` (equal
// splits along dim 0; call sites bess.py:346-350, 583-595) and
// `all_gather_cross_replica` (bess.py:519-545); replicated parameters are summed
// over replicas by PopTorch itself.  Here they are entry points of the library,
// asynchronous on the caller's hipStream_t - the stream the gather / scoring
// kernels run on - so a step is one in-order queue of kernels and collectives
// (nothing waits on a second, host-managed collective stream) and can be
// captured into a hipGraph as a whole.
//
// xGMI is point-to-point: the all-to-all is one grouped send/recv per peer, each
// block riding its own link; block sizes are equal by construction of the BESS
// samplers ("balanced all-to-all", docs/source/bess.rst:58-73).
#include <rccl/rccl.h>
#include <string.h>

#include <vector>

#include "common.h"

struct bess_comm {
    ncclComm_t nccl;
    int32_t world;
    int32_t rank;
    int32_t device;
};

namespace bess {

constexpr int NCCL_ERROR_BASE = BESS_ECOMM_BASE;

static int nccl_fail(ncclResult_t r, const char* what) {
    fail(NCCL_ERROR_BASE + static_cast<int>(r), "%s: %s", what, ncclGetErrorString(r));
    return NCCL_ERROR_BASE + static_cast<int>(r);
}

#define BESS_NCCL(call, what)                               \
    do {                                                    \
        ncclResult_t r_ = (call);                           \
        if (r_ != ncclSuccess) return nccl_fail(r_, what);  \
    } while (0)

#define BESS_HIP(call, what)                                                                   \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) return fail(static_cast<int>(e_), "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)

// rows of `table` (16-B chunks) gathered straight into the send buffer of the exchange
__global__ __launch_bounds__(256) void k_pack_rows(const uint4* __restrict__ table, const int32_t* __restrict__ idx,
                                                   int64_t n, int cpr, uint4* __restrict__ out) {
    const int64_t total = n * cpr;
    for (int64_t t = blockIdx.x * 256ll + threadIdx.x; t < total; t += 256ll * gridDim.x) {
        const int64_t i = t / cpr;
        const int c = static_cast<int>(t - i * cpr);
        out[t] = table[static_cast<int64_t>(idx[i]) * cpr + c];
    }
}

static int alltoall(bess_comm* c, const void* send, void* recv, int64_t bytes, hipStream_t st) {
    const char* s = static_cast<const char*>(send);
    char* r = static_cast<char*>(recv);
    BESS_NCCL(ncclGroupStart(), "alltoall: ncclGroupStart");
    for (int p = 0; p < c->world; ++p) {
        ncclResult_t a = ncclSend(s + p * bytes, static_cast<size_t>(bytes), ncclInt8, p, c->nccl, st);
        ncclResult_t b = a == ncclSuccess
                             ? ncclRecv(r + p * bytes, static_cast<size_t>(bytes), ncclInt8, p, c->nccl, st)
                             : a;
        if (b != ncclSuccess) {
            ncclGroupEnd();
            return nccl_fail(b, "alltoall: ncclSend / ncclRecv");
        }
    }
    BESS_NCCL(ncclGroupEnd(), "alltoall: ncclGroupEnd");
    return BESS_OK;
}

}  // namespace bess

using namespace bess;

extern "C" int bess_comm_unique_id(uint8_t* id) {
    BESS_REQUIRE(id, "comm_unique_id: NULL pointer");
    static_assert(sizeof(ncclUniqueId) == BESS_COMM_ID_BYTES, "BESS_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
    ncclUniqueId u;
    BESS_NCCL(ncclGetUniqueId(&u), "comm_unique_id: ncclGetUniqueId");
    memcpy(id, u.internal, BESS_COMM_ID_BYTES);
    return BESS_OK;
}

extern "C" int bess_comm_init_rank(int32_t world, int32_t rank, const uint8_t* id, bess_comm** comm) {
    BESS_REQUIRE(comm, "comm_init_rank: NULL result pointer");
    *comm = nullptr;
    BESS_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init_rank: rank %d of %d", rank, world);
    BESS_REQUIRE(id, "comm_init_rank: NULL id");
    int dev = 0;
    BESS_HIP(hipGetDevice(&dev), "comm_init_rank: hipGetDevice");
    ncclUniqueId u;
    memcpy(u.internal, id, BESS_COMM_ID_BYTES);
    ncclComm_t nc;
    BESS_NCCL(ncclCommInitRank(&nc, world, u, rank), "comm_init_rank: ncclCommInitRank");
    *comm = new bess_comm{nc, world, rank, dev};
    return BESS_OK;
}

extern "C" int bess_comm_init_all(int32_t n, const int32_t* dev_ids, bess_comm** comms) {
    BESS_REQUIRE(n >= 1 && comms, "comm_init_all: bad arguments");
    for (int i = 0; i < n; ++i) comms[i] = nullptr;
    std::vector<ncclComm_t> nc(static_cast<size_t>(n));
    std::vector<int> devs(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) devs[i] = dev_ids ? dev_ids[i] : i;
    BESS_NCCL(ncclCommInitAll(nc.data(), n, devs.data()), "comm_init_all: ncclCommInitAll");
    for (int i = 0; i < n; ++i) comms[i] = new bess_comm{nc[i], n, i, devs[i]};
    return BESS_OK;
}

extern "C" int bess_comm_destroy(bess_comm* c) {
    if (!c) return BESS_OK;
    ncclResult_t r = ncclCommDestroy(c->nccl);
    delete c;
    if (r != ncclSuccess) return nccl_fail(r, "comm_destroy: ncclCommDestroy");
    return BESS_OK;
}

extern "C" int bess_comm_info(const bess_comm* c, int32_t* world, int32_t* rank, int32_t* device) {
    BESS_REQUIRE(c, "comm_info: NULL communicator");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    if (device) *device = c->device;
    return BESS_OK;
}

extern "C" int bess_alltoall(bess_comm* c, const void* send, void* recv, int64_t bytes_per_peer, void* stream) {
    BESS_REQUIRE(c, "alltoall: NULL communicator");
    BESS_REQUIRE(bytes_per_peer >= 0, "alltoall: negative size");
    if (bytes_per_peer == 0) return BESS_OK;
    BESS_REQUIRE(send && recv && send != recv, "alltoall: NULL or aliased buffers");
    return alltoall(c, send, recv, bytes_per_peer, as_stream(stream));
}

extern "C" int bess_allgather(bess_comm* c, const void* send, void* recv, int64_t bytes, void* stream) {
    BESS_REQUIRE(c, "allgather: NULL communicator");
    BESS_REQUIRE(bytes >= 0, "allgather: negative size");
    if (bytes == 0) return BESS_OK;
    BESS_REQUIRE(send && recv, "allgather: NULL pointer");
    BESS_NCCL(ncclAllGather(send, recv, static_cast<size_t>(bytes), ncclInt8, c->nccl, as_stream(stream)),
              "allgather: ncclAllGather");
    return BESS_OK;
}

extern "C" int bess_allreduce_sum_f32(bess_comm* c, const float* send, float* recv, int64_t n, void* stream) {
    BESS_REQUIRE(c, "allreduce_sum: NULL communicator");
    BESS_REQUIRE(n >= 0, "allreduce_sum: negative size");
    if (n == 0) return BESS_OK;
    BESS_REQUIRE(send && recv, "allreduce_sum: NULL pointer");
    BESS_NCCL(ncclAllReduce(send, recv, static_cast<size_t>(n), ncclFloat32, ncclSum, c->nccl, as_stream(stream)),
              "allreduce_sum: ncclAllReduce");
    return BESS_OK;
}

extern "C" int bess_pack_exchange(bess_comm* c, int32_t dtype, int32_t width, const void* table,
                                  const int32_t* idx, int64_t rows_per_peer, void* send, void* recv,
                                  void* stream) {
    BESS_REQUIRE(c, "pack_exchange: NULL communicator");
    BESS_REQUIRE(dtype == BESS_F32 || dtype == BESS_F16, "pack_exchange: unknown dtype %d", dtype);
    BESS_REQUIRE(width > 0 && rows_per_peer >= 0, "pack_exchange: bad sizes");
    if (rows_per_peer == 0) return BESS_OK;
    BESS_REQUIRE(table && idx && send && recv && send != recv, "pack_exchange: NULL or aliased buffers");
    const int64_t row_bytes = static_cast<int64_t>(width) * (dtype == BESS_F32 ? 4 : 2);
    if (row_bytes % 16) return fail(BESS_EUNSUPPORTED, "pack_exchange: rows of %lld bytes (not a multiple of 16)",
                                    static_cast<long long>(row_bytes));
    hipStream_t st = as_stream(stream);
    const int64_t n = rows_per_peer * c->world;
    const int cpr = static_cast<int>(row_bytes / 16);
    int64_t blocks = ceil_div(n * cpr, 256);
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    k_pack_rows<<<static_cast<unsigned>(blocks), 256, 0, st>>>(static_cast<const uint4*>(table), idx, n, cpr,
                                                               static_cast<uint4*>(send));
    if (int e = check_launch("pack_exchange")) return e;
    return alltoall(c, send, recv, rows_per_peer * row_bytes, st);
}

// ---- recorded steps ------------------------------------------------------------------------------------------
// Which node types a recorded step holds (child graphs included): the library's fills are kernels, so a step
// recorded over it holds kernel nodes (+ RCCL's) - tests/test_graph_nodes.py asserts that no memset node is left.
static int count_nodes(hipGraph_t g, int32_t* counts, int32_t n_kinds, int depth) {
    size_t n = 0;
    BESS_HIP(hipGraphGetNodes(g, nullptr, &n), "graph_node_counts: hipGraphGetNodes");
    if (n == 0) return BESS_OK;
    std::vector<hipGraphNode_t> nodes(n);
    BESS_HIP(hipGraphGetNodes(g, nodes.data(), &n), "graph_node_counts: hipGraphGetNodes");
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType t;
        BESS_HIP(hipGraphNodeGetType(nodes[i], &t), "graph_node_counts: hipGraphNodeGetType");
        const int k = static_cast<int>(t);
        if (k >= 0 && k < n_kinds) counts[k] += 1;
        if (t == hipGraphNodeTypeGraph && depth < 8) {
            hipGraph_t child = nullptr;
            BESS_HIP(hipGraphChildGraphNodeGetGraph(nodes[i], &child), "graph_node_counts: child graph");
            if (int e = count_nodes(child, counts, n_kinds, depth + 1)) return e;
        }
    }
    return BESS_OK;
}

extern "C" int bess_graph_node_counts(void* graph, int32_t* counts, int32_t n_kinds) {
    BESS_REQUIRE(graph && counts && n_kinds > 0, "graph_node_counts: NULL graph / counts");
    for (int i = 0; i < n_kinds; ++i) counts[i] = 0;
    return count_nodes(static_cast<hipGraph_t>(graph), counts, n_kinds, 0);
}

// Row movers of the BESS hot path on gfx950: K1 gather, K9 scatter-add,
// K10 sparse / dense SGD; plus the library's error plumbing.
//
// All three are pure HBM-bound byte movers: one 16-lane DPP row per table row
// slice, 16 B per lane per access, 64-bit row offsets.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "common.h"

namespace bess {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fail(static_cast<int>(e), "%s: %s", what, hipGetErrorString(e));
        return static_cast<int>(e);
    }
    return BESS_OK;
}

int check_desc(const bess_model_desc* d) {
    if (!d) return fail(BESS_EINVAL, "model descriptor is NULL");
    if (d->scorer < BESS_TRANSE || d->scorer > BESS_BOXE)
        return fail(BESS_EINVAL, "unknown scorer %d", d->scorer);
    if (d->dtype != BESS_F32 && d->dtype != BESS_F16)
        return fail(BESS_EINVAL, "unknown dtype %d", d->dtype);
    if (d->scorer == BESS_BOXE) {
        if (d->width <= 0 || d->width % 2) return fail(BESS_EINVAL, "BoxE: entity width %d is not 2 d", d->width);
        if (d->norm_p < 1) return fail(BESS_EINVAL, "scoring norm %d is not a p >= 1", d->norm_p);
        return BESS_OK;
    }
    if (d->scorer == BESS_AFFINE) {
        const int n_part = d->reserved[0];
        if (n_part != 1 && n_part != 2) return fail(BESS_EINVAL, "affine scorer: n_part %d not in {1, 2}", n_part);
        if (d->width <= 0 || d->width % n_part) return fail(BESS_EINVAL, "affine scorer: width %d", d->width);
        if (d->norm_p < 1) return fail(BESS_EINVAL, "scoring norm %d is not a p >= 1", d->norm_p);
        return BESS_OK;
    }
    if (d->width <= 0 || d->rel_width <= 0)
        return fail(BESS_EINVAL, "non-positive width %d / %d", d->width, d->rel_width);
    if (is_distance(d->scorer) && d->norm_p < 1)
        return fail(BESS_EINVAL, "scoring norm %d is not a p >= 1", d->norm_p);
    const bool cplx = is_complex_entity(d->scorer);
    if (cplx && (d->width % 2)) return fail(BESS_EINVAL, "complex scorer needs even width");
    int want_rel = d->width;
    if (d->scorer == BESS_ROTATE) want_rel = d->width / 2;
    if (d->rel_width != want_rel)
        return fail(BESS_EINVAL, "rel_width %d does not match scorer (want %d)", d->rel_width, want_rel);
    return BESS_OK;
}

// ---------------------------------------------------------------------------
// K1: out[i, :] = table[idx[i], :].  Units of 16 B ("chunks"); thread t of the
// grid copies chunk (t % cpr) of row (t / cpr) -> consecutive lanes read
// consecutive 16 B of one row: fully coalesced for rows >= 256 B.
template <typename CH>
__global__ __launch_bounds__(256) void k_gather_rows(const CH* __restrict__ table,
                                                     const int32_t* __restrict__ idx, int64_t n,
                                                     int cpr, CH* __restrict__ out) {
    const int64_t total = n * cpr;
    for (int64_t t = blockIdx.x * 256ll + threadIdx.x; t < total; t += 256ll * gridDim.x) {
        const int64_t i = t / cpr;
        const int c = static_cast<int>(t - i * cpr);
        out[t] = table[static_cast<int64_t>(idx[i]) * cpr + c];
    }
}

// K9: dst[idx[i], :] += scale * src[i, :]   (f32 atomics; a wave-instruction adds
// 256 contiguous bytes of one row = the shape the memory-side atomic unit
// takes at full rate)
__global__ __launch_bounds__(256) void k_scatter_add_rows(float* __restrict__ dst, int width,
                                                          const int32_t* __restrict__ idx,
                                                          const float* __restrict__ src, int64_t n,
                                                          float scale) {
    const int64_t total = n * width;
    for (int64_t t = blockIdx.x * 256ll + threadIdx.x; t < total; t += 256ll * gridDim.x) {
        const int64_t i = t / width;
        const int c = static_cast<int>(t - i * width);
        const float v = scale * src[t];
        if (v != 0.f) unsafeAtomicAdd(dst + static_cast<int64_t>(idx[i]) * width + c, v);
    }
}

// K10 on an f16 shard: two halves per lane, packed atomic add
__global__ __launch_bounds__(256) void k_scatter_add_rows_f16(__half2* __restrict__ dst, int width2,
                                                              const int32_t* __restrict__ idx,
                                                              const float2* __restrict__ src,
                                                              int64_t n, float scale) {
    const int64_t total = n * width2;
    for (int64_t t = blockIdx.x * 256ll + threadIdx.x; t < total; t += 256ll * gridDim.x) {
        const int64_t i = t / width2;
        const int c = static_cast<int>(t - i * width2);
        const float2 v = src[t];
        unsafeAtomicAdd(dst + static_cast<int64_t>(idx[i]) * width2 + c,
                        __floats2half2_rn(scale * v.x, scale * v.y));
    }
}

// K9 + K10 (atomic form) for several (row ids, gradient rows) lists in one launch
struct SgdLists {
    const int32_t* idx[BESS_MAX_ROW_LISTS];
    const float* grad[BESS_MAX_ROW_LISTS];
    int64_t first[BESS_MAX_ROW_LISTS + 1];  // first[l] = rows in lists 0 .. l-1
    int n;
};
// (x: a dense `table2 += alpha * grad2` on a small replicated table - the relation table's plain SGD step - in the
// last workgroups of the launch: one dispatch less per training step)
struct DenseAxpy {
    void* table;
    const float* grad;
    int64_t n;
    float alpha;
    int blocks;
};
template <bool F16>
__global__ __launch_bounds__(256) void k_scatter_add_lists(void* __restrict__ dst, int width, SgdLists L, float scale,
                                                           DenseAxpy x) {
    const int main_blocks = static_cast<int>(gridDim.x) - x.blocks;
    if (static_cast<int>(blockIdx.x) >= main_blocks) {
        for (int64_t t = (blockIdx.x - main_blocks) * 256ll + threadIdx.x; t < x.n; t += 256ll * x.blocks) {
            if (F16) {
                half_t* t2 = static_cast<half_t*>(x.table);
                t2[t] = static_cast<half_t>(static_cast<float>(t2[t]) + x.alpha * x.grad[t]);
            } else {
                float* t2 = static_cast<float*>(x.table);
                t2[t] += x.alpha * x.grad[t];
            }
        }
        return;
    }
    const int wv = F16 ? width / 2 : width;  // values a thread handles per row position: 1 float or 2 halves
    const int64_t total = L.first[L.n] * wv;
    for (int64_t t = blockIdx.x * 256ll + threadIdx.x; t < total; t += 256ll * main_blocks) {
        const int64_t i = t / wv;
        const int c = static_cast<int>(t - i * wv);
        int l = 0;
#pragma unroll
        for (int k = 1; k < BESS_MAX_ROW_LISTS; ++k) l += (k < L.n && i >= L.first[k]) ? 1 : 0;
        const int64_t r = i - L.first[l];
        const int64_t row = L.idx[l][r];
        if (F16) {
            const float2 v = reinterpret_cast<const float2*>(L.grad[l])[r * wv + c];
            unsafeAtomicAdd(static_cast<__half2*>(dst) + row * wv + c, __floats2half2_rn(scale * v.x, scale * v.y));
        } else {
            const float v = scale * L.grad[l][r * wv + c];
            if (v != 0.f) unsafeAtomicAdd(static_cast<float*>(dst) + row * wv + c, v);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_dense_axpy(T* __restrict__ table,
                                                    const float* __restrict__ grad, int64_t n,
                                                    float alpha) {
    for (int64_t t = blockIdx.x * 256ll + threadIdx.x; t < n; t += 256ll * gridDim.x)
        table[t] = static_cast<T>(static_cast<float>(table[t]) + alpha * grad[t]);
}

static int grid_for(int64_t work_items) {
    int64_t b = ceil_div(work_items, 256);
    const int64_t cap = 256 * 16;  // 16 blocks per CU, grid-stride beyond
    return static_cast<int>(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace bess

using namespace bess;

extern "C" int bess_version(void) { return BESS_ABI_VERSION; }

extern "C" int bess_last_error(char* buf, size_t len) {
    const size_t n = strlen(g_err);
    if (buf && len) {
        const size_t c = n < len - 1 ? n : len - 1;
        memcpy(buf, g_err, c);
        buf[c] = 0;
    }
    return static_cast<int>(n);
}

extern "C" int bess_gather_rows(int32_t dtype, int32_t width, const void* table,
                                const int32_t* idx, int64_t n, void* out, void* stream) {
    BESS_REQUIRE(dtype == BESS_F32 || dtype == BESS_F16, "gather_rows: unknown dtype %d", dtype);
    BESS_REQUIRE(width > 0 && n >= 0, "gather_rows: bad sizes width=%d n=%lld", width, (long long)n);
    if (n == 0) return BESS_OK;
    BESS_REQUIRE(table && idx && out, "gather_rows: NULL pointer");
    const int64_t row_bytes = static_cast<int64_t>(width) * (dtype == BESS_F32 ? 4 : 2);
    hipStream_t st = as_stream(stream);
    if (row_bytes % 16 == 0) {
        const int cpr = static_cast<int>(row_bytes / 16);
        k_gather_rows<uint4><<<grid_for(n * cpr), 256, 0, st>>>(
            static_cast<const uint4*>(table), idx, n, cpr, static_cast<uint4*>(out));
    } else if (row_bytes % 4 == 0) {
        const int cpr = static_cast<int>(row_bytes / 4);
        k_gather_rows<uint32_t><<<grid_for(n * cpr), 256, 0, st>>>(
            static_cast<const uint32_t*>(table), idx, n, cpr, static_cast<uint32_t*>(out));
    } else {
        const int cpr = static_cast<int>(row_bytes / 2);
        k_gather_rows<uint16_t><<<grid_for(n * cpr), 256, 0, st>>>(
            static_cast<const uint16_t*>(table), idx, n, cpr, static_cast<uint16_t*>(out));
    }
    return check_launch("gather_rows");
}

extern "C" int bess_scatter_add_rows(float* dst, int32_t width, const int32_t* idx,
                                     const float* src, int64_t n, float scale, void* stream) {
    BESS_REQUIRE(width > 0 && n >= 0, "scatter_add_rows: bad sizes");
    if (n == 0) return BESS_OK;
    BESS_REQUIRE(dst && idx && src, "scatter_add_rows: NULL pointer");
    k_scatter_add_rows<<<grid_for(n * width), 256, 0, as_stream(stream)>>>(dst, width, idx, src, n,
                                                                           scale);
    return check_launch("scatter_add_rows");
}

extern "C" int bess_sparse_sgd(int32_t dtype, int32_t width, void* table, const int32_t* idx,
                               const float* grad, int64_t n, float lr, void* stream) {
    BESS_REQUIRE(dtype == BESS_F32 || dtype == BESS_F16, "sparse_sgd: unknown dtype %d", dtype);
    BESS_REQUIRE(width > 0 && n >= 0, "sparse_sgd: bad sizes");
    if (n == 0) return BESS_OK;
    BESS_REQUIRE(table && idx && grad, "sparse_sgd: NULL pointer");
    if (dtype == BESS_F32) {
        k_scatter_add_rows<<<grid_for(n * width), 256, 0, as_stream(stream)>>>(
            static_cast<float*>(table), width, idx, grad, n, -lr);
    } else {
        if (width % 2) return fail(BESS_EUNSUPPORTED, "sparse_sgd f16 needs an even width");
        k_scatter_add_rows_f16<<<grid_for(n * (width / 2)), 256, 0, as_stream(stream)>>>(
            static_cast<__half2*>(table), width / 2, idx, reinterpret_cast<const float2*>(grad), n,
            -lr);
    }
    return check_launch("sparse_sgd");
}

extern "C" int bess_dense_sgd(int32_t dtype, void* table, const float* grad, int64_t n_elem,
                              float lr, void* stream) {
    BESS_REQUIRE(dtype == BESS_F32 || dtype == BESS_F16, "dense_sgd: unknown dtype %d", dtype);
    if (n_elem <= 0) return BESS_OK;
    BESS_REQUIRE(table && grad, "dense_sgd: NULL pointer");
    if (dtype == BESS_F32)
        k_dense_axpy<float><<<grid_for(n_elem), 256, 0, as_stream(stream)>>>(
            static_cast<float*>(table), grad, n_elem, -lr);
    else
        k_dense_axpy<half_t><<<grid_for(n_elem), 256, 0, as_stream(stream)>>>(
            static_cast<half_t*>(table), grad, n_elem, -lr);
    return check_launch("dense_sgd");
}

extern "C" int bess_sparse_sgd_lists(int32_t dtype, int32_t width, void* table, int32_t n_lists,
                                     const int32_t* const* list_idx, const float* const* list_grad,
                                     const int64_t* list_rows, float lr, void* stream) {
    return bess_sparse_sgd_lists_axpy(dtype, width, table, n_lists, list_idx, list_grad, list_rows, lr, nullptr, nullptr, 0,
                                      0.f, stream);
}

extern "C" int bess_sparse_sgd_lists_axpy(int32_t dtype, int32_t width, void* table, int32_t n_lists,
                                          const int32_t* const* list_idx, const float* const* list_grad,
                                          const int64_t* list_rows, float lr, void* axpy_table, const float* axpy_grad,
                                          int64_t axpy_n, float axpy_alpha, void* stream) {
    BESS_REQUIRE(axpy_n >= 0 && (axpy_n == 0 || (axpy_table && axpy_grad)), "sparse_sgd_lists: axpy operands");
    BESS_REQUIRE(dtype == BESS_F32 || dtype == BESS_F16, "sparse_sgd_lists: unknown dtype %d", dtype);
    BESS_REQUIRE(width > 0 && n_lists >= 1 && n_lists <= BESS_MAX_ROW_LISTS, "sparse_sgd_lists: bad sizes");
    BESS_REQUIRE(table && list_idx && list_grad && list_rows, "sparse_sgd_lists: NULL pointer");
    if (dtype == BESS_F16 && width % 2) return fail(BESS_EUNSUPPORTED, "sparse_sgd_lists f16 needs an even width");
    SgdLists L{};
    L.n = n_lists;
    int64_t total = 0;
    for (int l = 0; l < n_lists; ++l) {
        BESS_REQUIRE(list_rows[l] >= 0 && (list_rows[l] == 0 || (list_idx[l] && list_grad[l])), "sparse_sgd_lists: list %d",
                     l);
        L.idx[l] = list_idx[l];
        L.grad[l] = list_grad[l];
        L.first[l] = total;
        total += list_rows[l];
    }
    for (int l = n_lists; l <= BESS_MAX_ROW_LISTS; ++l) L.first[l] = total;
    if (total == 0 && axpy_n == 0) return BESS_OK;
    hipStream_t st = as_stream(stream);
    const DenseAxpy x{axpy_table, axpy_grad, axpy_n, axpy_alpha,
                      static_cast<int>(std::min<int64_t>(ceil_div(axpy_n, 256 * 4), 256))};
    if (dtype == BESS_F32)
        k_scatter_add_lists<false><<<grid_for(total * width) + x.blocks, 256, 0, st>>>(table, width, L, -lr, x);
    else
        k_scatter_add_lists<true><<<grid_for(total * (width / 2)) + x.blocks, 256, 0, st>>>(table, width, L, -lr, x);
    return check_launch("sparse_sgd_lists");
}

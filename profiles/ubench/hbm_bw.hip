// Achievable HBM bandwidth on this box (the second roofline denominator):
// streaming read, streaming copy, and random 2 KiB-row reads of an 8 GiB buffer.
//   hipcc -O3 --offload-arch=gfx950 hbm_bw.hip -o hbm_bw && ./hbm_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ p, size_t n, float* out) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += 256ull * gridDim.x) {
        float4 v = p[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ p, float4* __restrict__ q, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += 256ull * gridDim.x) q[i] = p[i];
}
// one 16-lane group per random row of J x 256 B (J x 16 B per lane), rows chosen by a hash
template <int J>
__global__ __launch_bounds__(256) void k_rows(const float4* __restrict__ p, size_t n_rows, size_t n_read, float* out) {
    const size_t grp = (blockIdx.x * 256ull + threadIdx.x) >> 4;
    const int g = threadIdx.x & 15;
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t r = grp; r < n_read; r += (256ull * gridDim.x) >> 4) {
        const size_t row = (r * 2654435761ull + 12345ull) % n_rows;
        const float4* rp = p + row * (16 * J);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            float4 v = rp[g + 16 * j];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

int main() {
    const size_t bytes = 8ull << 30, n = bytes / 16;
    float4 *a, *b;
    float* out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, double moved, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-44s %8.3f ms  %6.2f TB/s\n", name, ms, moved / ms / 1e9);
    };
    for (int blocks : {2048, 8192, 32768}) {
        char nm[64];
        snprintf(nm, sizeof nm, "stream read 8 GiB, %d blocks", blocks);
        run(nm, (double)bytes, [&] { k_read<<<blocks, 256>>>(a, n, out); });
    }
    run("stream copy 8 GiB -> 8 GiB (read + write)", 2.0 * bytes, [&] { k_copy<<<8192, 256>>>(a, b, n); });
    const size_t n_rows = bytes / 2048;
    run("random 2 KiB rows, 1 Mi rows of 4 Mi (HBM)", 2048.0 * (1 << 20), [&] { k_rows<8><<<16384, 256>>>(a, n_rows, 1 << 20, out); });
    run("random 2 KiB rows, 4 Mi rows of 4 Mi (HBM)", 2048.0 * (4 << 20), [&] { k_rows<8><<<16384, 256>>>(a, n_rows, 4 << 20, out); });
    run("random 2 KiB rows, 1 Mi rows of 93,773 (L3)", 2048.0 * (1 << 20), [&] { k_rows<8><<<16384, 256>>>(a, 93773, 1 << 20, out); });
    run("random 1 KiB rows, 4 Mi rows of 8 Mi (HBM)", 1024.0 * (4 << 20), [&] { k_rows<4><<<16384, 256>>>(a, bytes / 1024, 4 << 20, out); });
    run("random 512 B rows, 4 Mi rows of 16 Mi (HBM)", 512.0 * (4 << 20), [&] { k_rows<2><<<16384, 256>>>(a, bytes / 512, 4 << 20, out); });
    run("random 512 B rows, 4 Mi rows of 8 Mi (4 GiB, HBM)", 512.0 * (4 << 20), [&] { k_rows<2><<<16384, 256>>>(a, bytes / 1024, 4 << 20, out); });
    run("random 512 B rows, 1 Mi rows of 312,576 (L3)", 512.0 * (1 << 20), [&] { k_rows<2><<<16384, 256>>>(a, 312576, 1 << 20, out); });
    return 0;
}

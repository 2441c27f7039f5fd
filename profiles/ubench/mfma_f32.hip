// What does the f32 MFMA inner loop of gemm_mfma.hip sustain without the K-slice machinery?
//   mode 0: MFMAs on register operands only          (pipe ceiling)
//   mode 1: + ds_read2_b32 fragments, software pipelined as in the GEMM (LDS image never refreshed)
//   mode 2: as 1 + one __syncthreads() per 64 MFMAs    (the per-slice barrier)
// varying workgroups per CU (1 or 2 waves per SIMD).
//   hipcc -O3 --offload-arch=gfx950 mfma_f32.hip -o mfma_f32 && ./mfma_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int slices, float seed) {
    __shared__ float A[32 * 129], B[32 * 129];
    for (int i = threadIdx.x; i < 32 * 129; i += 256) { A[i] = seed * i; B[i] = seed + i; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lk = lane >> 5;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float a[2][2] = {{seed, seed + 1}, {seed + 2, seed + 3}}, b[2][2] = {{seed, seed * 2}, {seed * 3, seed * 4}};
    for (int s = 0; s < slices; ++s) {
        if (MODE >= 1) {
            for (int i = 0; i < 2; ++i) { a[0][i] = A[lk * 129 + wm + 32 * i + l31]; b[0][i] = B[lk * 129 + wn + 32 * i + l31]; }
        }
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            const int c = (kk >> 1) & 1;
            if (MODE >= 1 && kk + 2 < 32) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[c ^ 1][i] = A[(kk + 2 + lk) * 129 + wm + 32 * i + l31];
                    b[c ^ 1][i] = B[(kk + 2 + lk) * 129 + wn + 32 * i + l31];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[MODE ? c : 0][i], b[MODE ? c : 0][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 2) __syncthreads();
    }
    float t = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

int main() {
    float* out; (void)hipMalloc(&out, 4 * 256 * 2048);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int slices = 400;
    for (int mode = 0; mode < 3; ++mode)
        for (int wg_per_cu : {1, 2, 4}) {
            const int blocks = 256 * wg_per_cu;
            auto launch = [&] {
                if (mode == 0) k<0><<<blocks, 256>>>(out, slices, 0.001f);
                else if (mode == 1) k<1><<<blocks, 256>>>(out, slices, 0.001f);
                else k<2><<<blocks, 256>>>(out, slices, 0.001f);
            };
            launch(); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double flops = double(blocks) * 4 * slices * 64.0 * 4096.0;
            printf("mode %d  %d wave(s)/SIMD: %7.3f ms  %6.1f TFLOP/s\n", mode, wg_per_cu, ms, flops / ms / 1e9);
        }
    return 0;
}

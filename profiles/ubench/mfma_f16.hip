// What does the consumer loop of csrc/gemm_split.hip sustain without the producers?
//   mode 0: 24 x v_mfma_f32_32x32x16_f16 per slice on register operands only   (pipe ceiling at the
//           clock the chip holds under this load)
//   mode 1: + the 16 ds_read_b128 fragment reads per slice, placed as in the GEMM (image never refreshed)
//   mode 2: as 1 + one __syncthreads() per slice
//   mode 3: as 2 + four idle partner waves in the workgroup that only take the barrier (the producers' seat)
//   hipcc -O3 --offload-arch=gfx950 mfma_f16.hip -o mfma_f16 && ./mfma_f16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int slices, float seed, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) char lds[2][2][16384];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = seed * i;
    __syncthreads();
    if (threadIdx.x >= 256) {  // MODE 3 only: partner waves
        for (int s = 0; s < slices; ++s) __syncthreads();
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lk = lane >> 5;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    int off_a[2][2], off_b[2][2];
    for (int ks = 0; ks < 2; ++ks)
        for (int part = 0; part < 2; ++part) {
            const int c = ks * 2 + lk + 4 * part, f = (l31 >> 1) & 7;
            off_a[ks][part] = (wm + l31) * 128 + ((c ^ f) << 4);
            off_b[ks][part] = (wn + l31) * 128 + ((c ^ f) << 4);
        }
    f32x16 accm[2][2], accc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.f; accc[i][j][r] = 0.f; }
    h8 fa[2][2][2], fb[2][2][2];
    for (int s = 0; s < 2; ++s) for (int i = 0; i < 2; ++i) for (int p = 0; p < 2; ++p) for (int e = 0; e < 8; ++e) {
        fa[s][i][p][e] = static_cast<_Float16>(seed * (e + i + p + s));
        fb[s][i][p][e] = static_cast<_Float16>(seed * (e - i + 2 * p + s));
    }
#define RD(set, g, ks)                                                                         \
    if (MODE >= 1) {                                                                           \
        const char* ia = lds[(g) & 1][0];                                                      \
        const char* ib = lds[(g) & 1][1];                                                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                        \
            fa[set][i][0] = *reinterpret_cast<const h8*>(ia + off_a[ks][0] + i * 4096);        \
            fa[set][i][1] = *reinterpret_cast<const h8*>(ia + off_a[ks][1] + i * 4096);        \
            fb[set][i][0] = *reinterpret_cast<const h8*>(ib + off_b[ks][0] + i * 4096);        \
            fb[set][i][1] = *reinterpret_cast<const h8*>(ib + off_b[ks][1] + i * 4096);        \
        }                                                                                      \
    }
#define MAIN(set)                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
        accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i][0], fb[set][j][0], accm[i][j], 0, 0, 0);
#define CORR(set)                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
        accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i][1], fb[set][j][0], accc[i][j], 0, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
        accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i][0], fb[set][j][1], accc[i][j], 0, 0, 0);
    // one k step: 12 MFMAs with the 8 fragment reads of the next k step slotted one per MFMA gap
    // (a burst of 8 ds_read_b128 between two MFMAs leaves the pipe idle while they issue)
#define INTERLEAVE()                                                                           \
    if (MODE >= 1) {                                                                           \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                 \
        }                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                     \
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int g = 0; g < slices; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        RD(1, g, 1);
        MAIN(0);
        CORR(0);
        INTERLEAVE();
        __builtin_amdgcn_sched_barrier(0);
        if (MODE >= 2) __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        RD(0, g + 1, 0);
        MAIN(1);
        CORR(1);
        INTERLEAVE();
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = c1 - c0;
    float t = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) t += accm[i][j][r] + accc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

int main() {
    float* out; (void)hipMalloc(&out, 4 * 256 * 2048);
    unsigned long long* cyc; (void)hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int slices = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        const int blocks = 256, threads = mode == 3 ? 512 : 256;
        auto launch = [&] {
            if (mode == 0) k<0><<<blocks, threads>>>(out, slices, 0.001f, cyc);
            else if (mode == 1) k<1><<<blocks, threads>>>(out, slices, 0.001f, cyc);
            else if (mode == 2) k<2><<<blocks, threads>>>(out, slices, 0.001f, cyc);
            else k<3><<<blocks, threads>>>(out, slices, 0.001f, cyc);
        };
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = double(blocks) * 4 * slices * 24.0 * 32768.0;
        unsigned long long hc = 0; (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        printf("mode %d: %7.3f ms  %7.1f TFLOP/s fp16 MFMA = %6.1f TFLOP/s of split-fp32 product, %6.1f ns = %6.1f s_memtime ticks per slice\n",
               mode, ms, flops / ms / 1e9, flops / 3 / ms / 1e9, ms * 1e6 / slices, double(hc) / slices);
    }
    return 0;
}

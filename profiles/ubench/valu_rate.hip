// VALU issue-rate probe for the p-norm kernels: how many wave64 v_sub_f32 / v_add_f32(|x|)
// per second does an MI355X sustain, as a function of waves per SIMD and with the LDS
// read mix of k_neg_shared_fwd (2 ds_read_b128 per 32 VALU)?
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

// PACKED: the differences are formed two at a time (v_pk_add_f32 with the query element broadcast and
// the candidate pair negated: 8 instead of 16 instructions per k), the |.| accumulation stays scalar
// (VOP3P has no abs modifier): 24 instead of 32 instructions for the same 16 (i, j) pairs.
// Result (profiles/r01/ubench_valu_rate.log): the same time - a v_pk_add_f32 costs two scalar issues,
// fp32 vector work on gfx950 is one lane-op per lane and clock either way.
template <bool WITH_LDS, bool PACKED>
__global__ __launch_bounds__(256) void k_probe(float* out, int iters, float seed) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed * i;
    __syncthreads();
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float a[4] = {seed, seed + 1, seed + 2, seed + 3}, b[4] = {seed * 2, seed * 3, seed * 4, seed * 5};
    const float4* l4 = reinterpret_cast<const float4*>(lds);
    for (int it = 0; it < iters; ++it) {
        if (WITH_LDS) {
            float4 x = l4[(threadIdx.x >> 4) + ((it & 15) << 4)];
            float4 y = l4[(threadIdx.x & 15) + ((it & 15) << 4) + 256];
            a[0] = x.x; a[1] = x.y; a[2] = x.z; a[3] = x.w;
            b[0] = y.x; b[1] = y.y; b[2] = y.z; b[3] = y.w;
        } else {
            // keep the operands loop-variant without extra VALU work inside the 32-op body
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        }
        if (PACKED) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const f32x2 d = f32x2{a[i], a[i]} - f32x2{b[j], b[j + 1]};
                    acc[4 * i + j] += fabsf(d.x);
                    acc[4 * i + j + 1] += fabsf(d.y);
                }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * i + j] += fabsf(a[i] - b[j]);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int packed = 0; packed < 2; ++packed)
    for (int lds = 0; lds < 2; ++lds)
        for (int wg_per_cu : {1, 2, 4, 8}) {
            const int blocks = 256 * wg_per_cu;
            auto launch = [&]() {
                if (packed) {
                    if (lds) k_probe<true, true><<<blocks, 256>>>(out, iters, 0.5f);
                    else k_probe<false, true><<<blocks, 256>>>(out, iters, 0.5f);
                } else {
                    if (lds) k_probe<true, false><<<blocks, 256>>>(out, iters, 0.5f);
                    else k_probe<false, false><<<blocks, 256>>>(out, iters, 0.5f);
                }
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double ops = double(blocks) * 256 * iters * 32.0;  // lane-ops of the scalar form (sub + add|.|)
            printf("packed=%d lds_mix=%d  %d wave(s)/SIMD: %8.3f ms  %6.2f T lane-ops/s  (%.2f cycles per wave-instruction per SIMD at 2.4 GHz)\n",
                   packed, lds, wg_per_cu, ms, ops / ms / 1e9,
                   (ms * 1e-3 * 2.4e9) / (double(wg_per_cu) * iters * 32.0));
        }
    return 0;
}

// Issue-rate probe of the packed forms the fp16 L1 tile kernels are built from (gfx950):
//   mode 0  forward:   m = v_pk_max_f16(q2, e2);  acc = v_dot2_f32_f16(m, {1,1}, acc)        2 ops / 2 elements
//   mode 1  backward:  d = v_pk_add_f16(x2, -y2); t = v_pk_max_i16(v_pk_min_i16(d, 1), -1);
//                      acc = v_dot2_i32_i16(t, c2, acc)                                       4 ops / 2 elements
//   mode 3  forward, hybrid: v_max_f16 (low halves), v_max_f16 op_sel (high halves), v_dot2_f32_f16   3 ops / 2 elements,
//           two of them on the full-rate pipe
//   mode 2  backward, sign by bit transfer (ties -> +c): d; c ^ (d & 0x80008000); v_dot2_f32_f16   4 ops / 2 elements
// with the LDS read mix of a 4 x 4 register tile (2 ds_read_b128 per 16 (i, j) pairs).
//   hipcc -O3 --offload-arch=gfx950 valu_pk.hip -o valu_pk && ./valu_pk
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));

// single-instruction helpers: the compiler scalarises packed int16 min / max (v_cmp + v_cndmask per half)
// and puts a canonicalising v_pk_max_f16 x, x in front of a packed float max of loaded bits
__device__ __forceinline__ uint32_t pk_max_f16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_sub_f16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int MODE, bool WITH_LDS>
__global__ __launch_bounds__(256) void k_probe(float* out, int iters, float seed) {
    __shared__ uint4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = make_uint4(i * 2654435761u, i * 40503u, i + 77u, i * 3u);
    __syncthreads();
    float facc[16];
    int iacc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { facc[i] = 0.f; iacc[i] = 0; }
    uint32_t a[4] = {0x3c003800u, 0x34003c00u, 0x38003400u, 0x3c003c00u}, b[4] = {0x35003900u, 0x3a003b00u, 0x33003100u, 0x3c003200u};
    const h2 ones = {(_Float16)1.f, (_Float16)1.f};
    uint32_t one2 = 0x00010001u, mone2 = 0xffffffffu;
    asm volatile("" : "+v"(one2), "+v"(mone2));
    for (int it = 0; it < iters; ++it) {
        if (WITH_LDS) {
            const uint4 x = lds[(threadIdx.x >> 4) + ((it & 15) << 4)];
            const uint4 y = lds[(threadIdx.x & 15) + ((it & 15) << 4) + 256];
            a[0] = x.x; a[1] = x.y; a[2] = x.z; a[3] = x.w;
            b[0] = y.x; b[1] = y.y; b[2] = y.z; b[3] = y.w;
        } else {
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const h2 x = __builtin_bit_cast(h2, a[i]), y = __builtin_bit_cast(h2, b[j]);
                (void)x; (void)y;
                if (MODE == 0) {
                    const uint32_t m = pk_max_f16(a[i], b[j]);
                    facc[4 * i + j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, m), ones, facc[4 * i + j], false);
                } else if (MODE == 3) {
                    // hybrid: the two maxima on the full-rate pipe (v_max_f16, low and high halves), the sum on the
                    // packed pipe
                    uint32_t m;
                    asm("v_max_f16 %0, %1, %2" : "=v"(m) : "v"(a[i]), "v"(b[j]));
                    asm("v_max_f16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(m) : "v"(a[i]), "v"(b[j]));
                    facc[4 * i + j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, m), ones, facc[4 * i + j], false);
                } else if (MODE == 1) {
                    const uint32_t d = pk_sub_f16(a[i], b[j]);
                    const uint32_t t = pk_max_i16(pk_min_i16(d, one2), mone2);
                    iacc[4 * i + j] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, t), __builtin_bit_cast(s2, b[(j + 1) & 3]), iacc[4 * i + j], false);
                } else {
                    const uint32_t d = pk_sub_f16(a[i], b[j]);
                    const uint32_t sg = (d & 0x80008000u) ^ b[(j + 1) & 3];
                    facc[4 * i + j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, sg), ones, facc[4 * i + j], false);
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += facc[i] + (float)iacc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, bool LDS>
static void bench(float* out, const char* name, int ops_per_pair) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int wg_per_cu : {1, 2, 4, 8}) {
        const int blocks = 256 * wg_per_cu;
        k_probe<MODE, LDS><<<blocks, 256>>>(out, iters, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k_probe<MODE, LDS><<<blocks, 256>>>(out, iters, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double pairs = double(blocks) * 256 * iters * 16.0;  // packed (2-element) steps
        printf("%-28s lds_mix=%d %d wave(s)/SIMD: %8.3f ms  %6.2f T elements/s  %6.2f T lane-ops/s\n", name, (int)LDS,
               wg_per_cu, ms, 2.0 * pairs / ms / 1e9, ops_per_pair * pairs / ms / 1e9);
    }
}

// one instruction at a time: 16 independent chains per lane, 8 waves per SIMD
template <int OP>
__global__ __launch_bounds__(256) void k_single(float* out, int iters) {
    uint32_t a[16];
    float f[16];
    int ii[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = 0x3c003800u + i * 257u + threadIdx.x; f[i] = i; ii[i] = i; }
    uint32_t b = 0x35003900u + threadIdx.x, c2 = 0x00010001u;
    asm volatile("" : "+v"(b), "+v"(c2));
    const h2 ones = {(_Float16)1.f, (_Float16)1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) a[i] = pk_max_f16(a[i], b);
            else if (OP == 1) a[i] = pk_sub_f16(a[i], b);
            else if (OP == 2) a[i] = pk_min_i16(a[i], b);
            else if (OP == 3) f[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a[i]), ones, f[i], false);
            else if (OP == 4) ii[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, a[i]), __builtin_bit_cast(s2, b), ii[i], false);
            else if (OP == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(b));
            else if (OP == 6) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c2));
            else if (OP == 7) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            else if (OP == 8) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            else if (OP == 9) asm volatile("v_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            else if (OP == 10) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c2));
            else if (OP == 11) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c2));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += f[i] + (float)ii[i] + (float)a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
static void single(float* out, const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 8;
    k_single<OP><<<blocks, 256>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_single<OP><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s %8.3f ms  %6.2f T lane-instructions/s\n", name, ms, double(blocks) * 256 * iters * 16.0 / ms / 1e9);
}

// fp32 forms of the L1 backward step acc += c * sgn(x - y) (operands pre-scaled so that med3(d, -1, 1) is the sign):
//   OP 0  v_sub_f32 + v_med3_f32 + v_fmac_f32 per element                                  (csrc/neg_shared.hip)
//   OP 1  v_pk_add_f32 (2 differences) + 2 v_med3_f32 + v_pk_fma_f32 per 2 elements        (packed fp32 candidates)
//   OP 2..6  single instructions: v_pk_fma_f32, v_pk_add_f32, v_med3_f32, v_fma_f32 (VOP3), v_fmac_f32 (VOP2)
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k_f32(float* out, int iters) {
    f2 acc[8], x[8];
    f2 y = {0.25f + threadIdx.x, 0.75f}, c = {1.5f, -0.5f};
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = f2{0.f, 0.f}; x[i] = f2{float(i) + threadIdx.x, float(i) * 0.5f}; }
    asm volatile("" : "+v"(y), "+v"(c));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) {
                float d0, d1;
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(x[i].x), "v"(y.x));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(x[i].y), "v"(y.y));
                asm volatile("v_med3_f32 %0, %0, -1.0, 1.0" : "+v"(d0));
                asm volatile("v_med3_f32 %0, %0, -1.0, 1.0" : "+v"(d1));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].x) : "v"(c.x), "v"(d0));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].y) : "v"(c.x), "v"(d1));
            } else if (OP == 1) {
                f2 d;
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x[i]), "v"(y));
                asm volatile("v_med3_f32 %0, %0, -1.0, 1.0" : "+v"(d.x));
                asm volatile("v_med3_f32 %0, %0, -1.0, 1.0" : "+v"(d.y));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(c), "v"(d));
            } else if (OP == 2) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(c), "v"(y));
            } else if (OP == 3) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
            } else if (OP == 4) {
                asm volatile("v_med3_f32 %0, %0, -1.0, 1.0" : "+v"(acc[i].x));
                asm volatile("v_med3_f32 %0, %0, -1.0, 1.0" : "+v"(acc[i].y));
            } else if (OP == 5) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(c.x), "v"(y.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(c.y), "v"(y.y));
            } else {
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].x) : "v"(c.x), "v"(y.x));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].y) : "v"(c.y), "v"(y.y));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
static void f32(float* out, const char* name, double instr_per_pair) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 8;
    k_f32<OP><<<blocks, 256>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_f32<OP><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double pairs = double(blocks) * 256 * iters * 8.0;
    printf("%-34s %8.3f ms  %6.2f T elements/s  %6.2f T lane-instructions/s\n", name, ms, 2.0 * pairs / ms / 1e9,
           instr_per_pair * pairs / ms / 1e9);
}

// Do the two instruction families overlap when DIFFERENT waves of a SIMD run them?  SPLIT of every 8 waves run the
// packed forward (v_pk_max_f16 + v_dot2: 2 elements per pair of instructions), the others the fp32 forward
// (v_sub_f32 + v_add_f32 |d|: 1 element per pair of instructions); same number of instruction pairs per wave.
template <int SPLIT>
__global__ __launch_bounds__(256) void k_mix(float* out, int iters) {
    float facc[16];
    uint32_t a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { facc[i] = 0.f; a[i] = 0x3c003800u + i * 257u + threadIdx.x; }
    uint32_t b = 0x35003900u + threadIdx.x;
    float bf = 0.37f + threadIdx.x;
    asm volatile("" : "+v"(b), "+v"(bf));
    const h2 ones = {(_Float16)1.f, (_Float16)1.f};
    const int wave = (blockIdx.x * 4 + (threadIdx.x >> 6)) & 7;
    if (wave < SPLIT) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t m = pk_max_f16(a[i], b);
                facc[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, m), ones, facc[i], false);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float d;
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(__builtin_bit_cast(float, a[i])), "v"(bf));
                asm volatile("v_add_f32 %0, %0, |%1|" : "+v"(facc[i]) : "v"(d));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += facc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int SPLIT>
static void mix(float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 8;
    k_mix<SPLIT><<<blocks, 256>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_mix<SPLIT><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double pairs = double(blocks) * 256 * iters * 16.0;  // instruction pairs
    const double elems = pairs * (SPLIT / 8.0) * 2.0 + pairs * (1.0 - SPLIT / 8.0);
    printf("mix: %d of 8 waves packed, %d fp32: %8.3f ms  %6.2f T elements/s  %6.2f T lane-instructions/s\n", SPLIT, 8 - SPLIT,
           ms, elems / ms / 1e9, 2.0 * pairs / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 64);
    single<5>(out, "v_add_f32");
    single<7>(out, "v_xor_b32");
    single<0>(out, "v_pk_max_f16");
    single<1>(out, "v_pk_add_f16");
    single<2>(out, "v_pk_min_i16");
    single<8>(out, "v_pk_add_u16");
    single<6>(out, "v_pk_fma_f16");
    single<3>(out, "v_dot2c_f32_f16");
    single<4>(out, "v_dot2c_i32_i16");
    single<9>(out, "v_max_f16");
    single<10>(out, "v_sad_u16");
    single<11>(out, "v_sad_u8");
    mix<8>(out);
    mix<6>(out);
    mix<4>(out);
    mix<2>(out);
    mix<0>(out);
    f32<2>(out, "v_pk_fma_f32", 1);
    f32<3>(out, "v_pk_add_f32", 1);
    f32<4>(out, "v_med3_f32 (x2)", 2);
    f32<5>(out, "v_fma_f32 (x2)", 2);
    f32<6>(out, "v_fmac_f32 (x2)", 2);
    f32<0>(out, "bwd f32: sub+med3+fmac", 6);
    f32<1>(out, "bwd pk f32: pk_add+2 med3+pk_fma", 4);
    bench<3, false>(out, "fwd 2 v_max_f16 + dot2", 3);
    bench<3, true>(out, "fwd 2 v_max_f16 + dot2", 3);
    bench<0, false>(out, "fwd pk_max+dot2", 2);
    bench<0, true>(out, "fwd pk_max+dot2", 2);
    bench<1, false>(out, "bwd sub+min+max+sdot2", 4);
    bench<1, true>(out, "bwd sub+min+max+sdot2", 4);
    bench<2, false>(out, "bwd sub+and+xor+fdot2", 4);
    bench<2, true>(out, "bwd sub+and+xor+fdot2", 4);
    return 0;
}

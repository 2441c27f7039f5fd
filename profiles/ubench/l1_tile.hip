// Exploration bench for the shared-negative L1 kernel (TransE / RotatE, K4d):
// which part of the 64x64 LDS-tiled kernel keeps it at ~0.45 of the VALU issue rate,
// and does a no-LDS formulation (one operand in SGPRs) do better?
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize l1_tile.hip -o l1_tile && ./l1_tile
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

constexpr int LDP = 68;

// MODE 0: as shipped (load -> barrier -> compute -> barrier)      KT deep stages
// MODE 1: loads only for the first stage (barriers kept)
// MODE 2: loads only for the first stage, no barriers in the loop
// MODE 3: register prefetch of the next stage before computing the current one
template <int KT, int MODE>
__global__ __launch_bounds__(256) void k_tile(const float* __restrict__ Q, const float* __restrict__ E, int S, int N,
                                              int W, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float Qs[KT][LDP];
    __shared__ __attribute__((aligned(16))) float Es[KT][LDP];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int q0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    constexpr int V = KT / 4;  // floats per thread per operand per stage
    const int t = threadIdx.x;
    const int m = t >> 2, kc = (t & 3) * V;
    const float* qp = Q + (size_t)(q0 + m) * W + kc;
    const float* ep = E + (size_t)(j0 + m) * W + kc;
    float qv[V], ev[V];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < V; i += 4) {
            float4 a = *reinterpret_cast<const float4*>(qp + k0 + i);
            float4 b = *reinterpret_cast<const float4*>(ep + k0 + i);
            qv[i] = a.x; qv[i + 1] = a.y; qv[i + 2] = a.z; qv[i + 3] = a.w;
            ev[i] = b.x; ev[i + 1] = b.y; ev[i + 2] = b.z; ev[i + 3] = b.w;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            Qs[kc + i][m] = qv[i];
            Es[kc + i][m] = ev[i];
        }
    };
    auto compute = [&]() {
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const float4 a4 = *reinterpret_cast<const float4*>(&Qs[k][ty * 4]);
            const float4 b4 = *reinterpret_cast<const float4*>(&Es[k][tx * 4]);
            const float a[4] = {a4.x, a4.y, a4.z, a4.w};
            const float b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += fabsf(a[i] - b[j]);
        }
    };
    if (MODE == 3) {
        gload(0);
        for (int k0 = 0; k0 < W; k0 += KT) {
            lstore();
            __syncthreads();
            if (k0 + KT < W) gload(k0 + KT);
            compute();
            __syncthreads();
        }
    } else {
        for (int k0 = 0; k0 < W; k0 += KT) {
            if (MODE == 0 || k0 == 0) {
                gload(k0);
                lstore();
            }
            if (MODE != 2 || k0 == 0) __syncthreads();
            compute();
            if (MODE != 2) __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) out[(size_t)(q0 + ty * 4 + i) * N + j0 + tx * 4 + j] = -acc[i][j];
}

// No LDS: lane l owns QR query rows (q0 + l + 64 r); the negative row values are wave-uniform and
// come through scalar loads (SGPR operand of v_sub_f32).  One wave = 64*QR queries x JB negatives.
template <int QR, int JB, int WB>
__global__ __launch_bounds__(256) void k_sgpr(const float* __restrict__ Q, const float* __restrict__ E, int S, int N,
                                              int W, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q0 = (blockIdx.y * 4 + wave) * 64 * QR;
    const int j0 = blockIdx.x * JB;
    float acc[QR][JB];
#pragma unroll
    for (int r = 0; r < QR; ++r)
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[r][j] = 0.f;
    const float* qp[QR];
#pragma unroll
    for (int r = 0; r < QR; ++r) qp[r] = Q + (size_t)(q0 + lane + 64 * r) * W;
    const float* eb = E + (size_t)j0 * W;  // uniform
    for (int k0 = 0; k0 < W; k0 += WB) {
        float qv[QR][WB];
#pragma unroll
        for (int r = 0; r < QR; ++r)
#pragma unroll
            for (int i = 0; i < WB; i += 4) {
                float4 a = *reinterpret_cast<const float4*>(qp[r] + k0 + i);
                qv[r][i] = a.x; qv[r][i + 1] = a.y; qv[r][i + 2] = a.z; qv[r][i + 3] = a.w;
            }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            float e[WB];
#pragma unroll
            for (int i = 0; i < WB; ++i) e[i] = eb[(size_t)j * W + k0 + i];  // uniform address -> s_load
#pragma unroll
            for (int i = 0; i < WB; ++i)
#pragma unroll
                for (int r = 0; r < QR; ++r) acc[r][j] += fabsf(qv[r][i] - e[i]);
        }
    }
#pragma unroll
    for (int r = 0; r < QR; ++r)
#pragma unroll
        for (int j = 0; j < JB; ++j) out[(size_t)(q0 + lane + 64 * r) * N + j0 + j] = -acc[r][j];
}

static double check(const std::vector<float>& Q, const std::vector<float>& E, const std::vector<float>& o, int S, int N, int W) {
    double worst = 0;
    for (int t = 0; t < 200; ++t) {
        int q = (t * 7919) % S, j = (t * 104729) % N;
        double s = 0;
        for (int w = 0; w < W; ++w) s += fabs((double)Q[(size_t)q * W + w] - E[(size_t)j * W + w]);
        worst = fmax(worst, fabs(-s - o[(size_t)q * N + j]) / fmax(1.0, s));
    }
    return worst;
}

int main() {
    const int S = 4096, N = 4096;
    for (int W : {256, 400}) {
        const int Wp = W;  // W multiple of 16 here
        std::vector<float> hQ((size_t)S * Wp), hE((size_t)N * Wp), ho((size_t)S * N);
        srand(1);
        for (auto& x : hQ) x = rand() / (float)RAND_MAX - 0.5f;
        for (auto& x : hE) x = rand() / (float)RAND_MAX - 0.5f;
        float *Q, *E, *out;
        hipMalloc(&Q, hQ.size() * 4); hipMalloc(&E, hE.size() * 4); hipMalloc(&out, ho.size() * 4);
        hipMemcpy(Q, hQ.data(), hQ.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(E, hE.data(), hE.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto run = [&](const char* name, auto launch, bool verify) {
            launch(); hipDeviceSynchronize();
            hipError_t err = hipGetLastError();
            if (err != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(err)); return; }
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            double worst = -1;
            if (verify) { hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost); worst = check(hQ, hE, ho, S, N, W); }
            printf("W=%d %-44s %8.1f us  %6.2f T lane-ops/s  relerr %.1e\n", W, name, ms * 1e3, 2.0 * S * N * W / ms / 1e9, worst);
        };
        dim3 g(N / 64, S / 64);
        run("tile KT=16 as shipped", [&] { k_tile<16, 0><<<g, 256>>>(Q, E, S, N, W, out); }, true);
        run("tile KT=16 no loads after stage 0", [&] { k_tile<16, 1><<<g, 256>>>(Q, E, S, N, W, out); }, false);
        run("tile KT=16 no loads, no barriers", [&] { k_tile<16, 2><<<g, 256>>>(Q, E, S, N, W, out); }, false);
        run("tile KT=16 register prefetch", [&] { k_tile<16, 3><<<g, 256>>>(Q, E, S, N, W, out); }, true);
        run("tile KT=32 as shipped", [&] { k_tile<32, 0><<<g, 256>>>(Q, E, S, N, W, out); }, true);
        run("tile KT=32 register prefetch", [&] { k_tile<32, 3><<<g, 256>>>(Q, E, S, N, W, out); }, true);
        run("sgpr QR=2 JB=16 WB=4", [&] { k_sgpr<2, 16, 4><<<dim3(N / 16, S / (4 * 64 * 2)), 256>>>(Q, E, S, N, W, out); }, true);
        run("sgpr QR=2 JB=16 WB=8", [&] { k_sgpr<2, 16, 8><<<dim3(N / 16, S / (4 * 64 * 2)), 256>>>(Q, E, S, N, W, out); }, true);
        run("sgpr QR=4 JB=8 WB=8", [&] { k_sgpr<4, 8, 8><<<dim3(N / 8, S / (4 * 64 * 4)), 256>>>(Q, E, S, N, W, out); }, true);
        run("sgpr QR=1 JB=32 WB=4", [&] { k_sgpr<1, 32, 4><<<dim3(N / 32, S / (4 * 64)), 256>>>(Q, E, S, N, W, out); }, true);
        hipFree(Q); hipFree(E); hipFree(out);
    }
    return 0;
}

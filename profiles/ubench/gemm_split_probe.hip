// Where does the time of csrc/gemm_split.hip go?  The kernel is compiled here with one part
// removed at a time (results are then wrong; only the time matters):
//   hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950
//         [-DBESS_PROBE_NO_MFMA] [-DBESS_PROBE_NO_LOAD] gemm_split_probe.hip -o probe && ./probe
#include "../../bess-kge_amd/csrc/gemm_split.hip"

#include <stdio.h>

namespace bess {
int fail(int code, const char*, ...) { return code; }
int check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -1; }
}  // namespace bess

int main() {
    const int64_t S = 4096, N = 4096, M = 93773;
    const int W = 512;
    float *q, *e, *out;
    int32_t* idx;
    hipMalloc(&q, S * W * 4);
    hipMalloc(&e, M * W * 4);
    hipMalloc(&out, S * N * 4);
    hipMalloc(&idx, N * 4);
    hipMemset(q, 0x3c, S * W * 4);
    hipMemset(e, 0x3c, M * W * 4);
    int32_t* h = new int32_t[N];
    for (int64_t i = 0; i < N; ++i) h[i] = static_cast<int32_t>((i * 7919) % M);
    hipMemcpy(idx, h, N * 4, hipMemcpyHostToDevice);
    const int64_t ws_bytes = bess::gemm_split_workspace(S, N, W);
    void* ws;
    hipMalloc(&ws, ws_bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) bess::gemm_split_fwd(BESS_F32, q, S, e, idx, N, W, out, N, ws, ws_bytes, nullptr);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) bess::gemm_split_fwd(BESS_F32, q, S, e, idx, N, W, out, N, ws, ws_bytes, nullptr);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps;
    printf("%8.1f us  %7.1f TFLOP/s (4096 x 4096 x 512, split pre-pass included)\n", us, 2.0 * S * N * W / us / 1e6);
    // the GEMM alone, on the images the last call left in the workspace
    const int n_slice = W / 32;
    char* qa = static_cast<char*>(ws);
    char* eb = qa + S * n_slice * 128;
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i)
        bess::k_gemm_split_f16<true><<<256, 512>>>(qa, eb, S, N, n_slice, out, N, 32, 1024, 1, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("%8.1f us  %7.1f TFLOP/s (GEMM kernel alone)\n", ms * 1e3 / reps, 2.0 * S * N * W / (ms * 1e3 / reps) / 1e6);
#ifdef BESS_PROBE_TICKS
    float tk[4];
    hipMemcpy(tk, out, 16, hipMemcpyDeviceToHost);
    printf("workgroup 0, per slice: %.0f clocks, of which the producers wait %.0f and the consumers %.0f at the barrier (%.0f slices)\n",
           tk[2] / tk[3], tk[0] / tk[3], tk[1] / tk[3], tk[3]);
#endif
    return 0;
}

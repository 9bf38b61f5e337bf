// Developer probe: which event-based timing of one dispatch agrees with rocprofv3's kernel duration?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
int main() {
    const size_t n = (size_t)48 << 20 >> 4;   // 48 MB in, 48 MB out
    uint4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 1, n * 16);
    hipStream_t s; hipStreamCreate(&s);
    const int K = 200;
    std::vector<hipEvent_t> ea(K), eb(K);
    for (int i = 0; i < K; ++i) { hipEventCreate(&ea[i]); hipEventCreate(&eb[i]); }
    dim3 g((n + 255) / 256), blk(256);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_copy, g, blk, 0, s, a, b, n);
    hipStreamSynchronize(s);
    // mode 1: ext launch with start+stop, every 4th launch
    for (int i = 0; i < K; ++i) {
        for (int j = 0; j < 3; ++j) hipLaunchKernelGGL(k_copy, g, blk, 0, s, a, b, n);
        hipExtLaunchKernelGGL(k_copy, g, blk, 0, s, ea[i], eb[i], 0, a, b, n);
    }
    hipStreamSynchronize(s);
    double t1 = 0; for (int i = 0; i < K; ++i) { float ms; hipEventElapsedTime(&ms, ea[i], eb[i]); t1 += ms; }
    // mode 2: ext launch with stop only; elapsed(stop, stop)
    for (int i = 0; i < K; ++i) {
        for (int j = 0; j < 3; ++j) hipLaunchKernelGGL(k_copy, g, blk, 0, s, a, b, n);
        hipExtLaunchKernelGGL(k_copy, g, blk, 0, s, nullptr, eb[i], 0, a, b, n);
    }
    hipStreamSynchronize(s);
    double t2 = 0; int ok2 = 0; for (int i = 0; i < K; ++i) { float ms = 0; if (hipEventElapsedTime(&ms, eb[i], eb[i]) == hipSuccess) { t2 += ms; ++ok2; } }
    // mode 3: ext launch with the SAME event as start and stop
    for (int i = 0; i < K; ++i) {
        for (int j = 0; j < 3; ++j) hipLaunchKernelGGL(k_copy, g, blk, 0, s, a, b, n);
        hipExtLaunchKernelGGL(k_copy, g, blk, 0, s, eb[i], eb[i], 0, a, b, n);
    }
    hipStreamSynchronize(s);
    double t3 = 0; int ok3 = 0; for (int i = 0; i < K; ++i) { float ms = 0; if (hipEventElapsedTime(&ms, eb[i], eb[i]) == hipSuccess) { t3 += ms; ++ok3; } }
    // mode 4: plain wall clock over 4K launches
    hipEvent_t w0, w1; hipEventCreate(&w0); hipEventCreate(&w1);
    hipEventRecord(w0, s);
    for (int i = 0; i < 4 * K; ++i) hipLaunchKernelGGL(k_copy, g, blk, 0, s, a, b, n);
    hipEventRecord(w1, s); hipStreamSynchronize(s);
    float w; hipEventElapsedTime(&w, w0, w1);
    printf("ext(a,b) %.3f us | ext(null,b) elapsed(b,b) %.3f us (ok %d) | ext(b,b) %.3f us (ok %d) | back-to-back %.3f us/launch\n",
           t1 / K * 1e3, ok2 ? t2 / ok2 * 1e3 : -1., ok2, ok3 ? t3 / ok3 * 1e3 : -1., ok3, w / (4 * K) * 1e3);
    return 0;
}

// Developer probe: is a payload written by one stream's kernel visible to ANOTHER stream's kernel when the two streams are ordered
// only by a flag word (set by a tiny kernel behind the producer, polled by a tiny kernel in front of the consumer)?
// HIP knows nothing about that dependency, so whatever acquire / release the runtime attaches to dispatch packets is what
// back-to-back kernels of ONE queue get.  Variants: plain / nt / sc1 payload stores, plain / sc1 consumer loads.
// Build: hipcc --offload-arch=gfx950 -O2 tools/flag_coherence_probe.hip -o tools/flag_coherence_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 plain, 1 nontemporal, 2 sc1 (write-through)
__global__ __launch_bounds__(256) void k_fill(u32x4* p, size_t n16, unsigned v) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    const u32x4 val = {v, v, v, v};
    for (; i < n16; i += stride) {
        if (MODE == 0) p[i] = val;
        else if (MODE == 1) __builtin_nontemporal_store(val, p + i);
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p + i), "v"(val) : "memory");
    }
    if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int MODE>   // 0 plain, 1 sc1 loads
__global__ __launch_bounds__(256) void k_check(const u32x4* p, size_t n16, unsigned v, unsigned* bad, unsigned* oldest) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    unsigned nb = 0, mn = v;
    for (; i < n16; i += stride) {
        u32x4 x;
        if (MODE == 0) x = p[i];
        else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p + i) : "memory");
        for (int k = 0; k < 4; ++k) { if (x[k] != v) { ++nb; if (x[k] < mn) mn = x[k]; } }
    }
    if (nb) { atomicAdd(bad, nb); atomicMin(oldest, mn); }
}
__global__ void k_set(unsigned* flag, unsigned v) { if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_wait(const unsigned* flag, unsigned v) {
    if (threadIdx.x != 0) return;
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - v) < 0) __builtin_amdgcn_s_sleep(2);
}

template <int FM, int CM>
static void run(const char* name, hipStream_t a, hipStream_t b, u32x4* buf, size_t n16, unsigned* flags, int grid_fill, int grid_check) {
    unsigned* ready = flags, *ack = flags + 64, *bad = flags + 128, *oldest = flags + 192;
    CK(hipMemset(flags, 0, 4096));
    unsigned big = 0xffffffffu;
    CK(hipMemcpy(oldest, &big, 4, hipMemcpyHostToDevice));
    CK(hipMemset(buf, 0, n16 * 16));
    CK(hipDeviceSynchronize());
    const int E = 300;
    for (int e = 1; e <= E; ++e) {
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, b, ready, (unsigned)e);
        hipLaunchKernelGGL((k_check<CM>), dim3(grid_check), dim3(256), 0, b, buf, n16, (unsigned)e, bad, oldest);
        hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, b, ack, (unsigned)e);
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, a, ack, (unsigned)(e - 1));
        hipLaunchKernelGGL((k_fill<FM>), dim3(grid_fill), dim3(256), 0, a, buf, n16, (unsigned)e);
        hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, a, ready, (unsigned)e);
    }
    CK(hipDeviceSynchronize());
    unsigned nb = 0;
    CK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
    printf("%-58s stale words %10u of %zu x %d epochs\n", name, nb, n16 * 4, E);
}

int main() {
    CK(hipSetDevice(0));
    hipStream_t a, b, am, bm;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    uint32_t m0[8] = {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}, m1[8] = {0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    CK(hipExtStreamCreateWithCUMask(&am, 8, m0));
    CK(hipExtStreamCreateWithCUMask(&bm, 8, m1));
    unsigned* flags;
    CK(hipMalloc(&flags, 4096));
    for (size_t bytes : {(size_t)64 << 10, (size_t)4 << 20, (size_t)32 << 20}) {
        u32x4* buf;
        CK(hipMalloc(&buf, bytes));
        const size_t n16 = bytes / 16;
        const int gf = (int)((n16 + 255) / 256 < 2048 ? (n16 + 255) / 256 : 2048), gc = gf;
        printf("---- payload %zu KB\n", bytes >> 10);
        run<0, 0>("plain stores, plain loads, ordinary streams", a, b, buf, n16, flags, gf, gc);
        run<1, 0>("nt stores,    plain loads, ordinary streams", a, b, buf, n16, flags, gf, gc);
        run<2, 0>("sc1 stores,   plain loads, ordinary streams", a, b, buf, n16, flags, gf, gc);
        run<0, 1>("plain stores, sc1 loads,   ordinary streams", a, b, buf, n16, flags, gf, gc);
        run<2, 1>("sc1 stores,   sc1 loads,   ordinary streams", a, b, buf, n16, flags, gf, gc);
        run<0, 0>("plain stores, plain loads, CU-masked streams (32 / 224)", am, bm, buf, n16, flags, gf, gc);
        run<1, 0>("nt stores,    plain loads, CU-masked streams (32 / 224)", am, bm, buf, n16, flags, gf, gc);
        run<2, 0>("sc1 stores,   plain loads, CU-masked streams (32 / 224)", am, bm, buf, n16, flags, gf, gc);
        CK(hipFree(buf));
    }
    printf("done\n");
    return 0;
}

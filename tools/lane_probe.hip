// Developer probe for the flag-synchronised exchange lane: what the building blocks cost on MI355X.
//   1. hipExtStreamCreateWithCUMask: which CUs (XCD, SE, CU) a mask's bits select
//   2. cross-stream hand-off ping-pong: events vs hipStreamWriteValue32/WaitValue32 vs in-memory flags written / polled by kernels
//   3. a dependent tiny kernel in a stream (the price of a flag-set / flag-wait kernel)
// Build: hipcc --offload-arch=gfx950 -O2 tools/lane_probe.hip -o tools/lane_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_where(unsigned* out, long long spin) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;      // HW_REG_XCC_ID[3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | 4);              // HW_REG_HW_ID[15:0]
        out[blockIdx.x] = (xcc << 16) | (hw & 0xffffu);
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}

__global__ void k_set(unsigned* flag, unsigned v) { __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_wait(const unsigned* flag, unsigned v, unsigned* err) {
    unsigned n = 0;
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - v) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++n > (1u << 22)) { *err = 1; break; }
    }
}
__global__ void k_wait_set(const unsigned* flag, unsigned v, unsigned* out, unsigned w, unsigned* err) {
    unsigned n = 0;
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - v) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++n > (1u << 22)) { *err = 1; break; }
    }
    __hip_atomic_store(out, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_nop() {}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void where(const char* name, hipStream_t s, unsigned* d_out, int nblk) {
    CK(hipMemsetAsync(d_out, 0xff, nblk * 4, s));
    hipLaunchKernelGGL(k_where, dim3(nblk), dim3(256), 0, s, d_out, 2000LL);      // 20 us per block: blocks spread over every CU the queue may use
    CK(hipStreamSynchronize(s));
    unsigned* h = (unsigned*)malloc(nblk * 4);
    CK(hipMemcpy(h, d_out, nblk * 4, hipMemcpyDeviceToHost));
    // distinct (xcc, se, sh, cu)
    static unsigned char seen[16][8][2][16];
    memset(seen, 0, sizeof(seen));
    int per_xcc[16] = {0};
    for (int i = 0; i < nblk; ++i) {
        const unsigned xcc = h[i] >> 16, hw = h[i] & 0xffff;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        if (!seen[xcc & 15][se][sh][cu]) { seen[xcc & 15][se][sh][cu] = 1; per_xcc[xcc & 15]++; }
    }
    int tot = 0;
    printf("%-28s distinct CUs per XCD:", name);
    for (int x = 0; x < 8; ++x) { printf(" %2d", per_xcc[x]); tot += per_xcc[x]; }
    printf("  total %d\n", tot);
    if (tot <= 64) {
        printf("    (xcc,se,cu):");
        for (int x = 0; x < 8; ++x) for (int se = 0; se < 8; ++se) for (int sh = 0; sh < 2; ++sh) for (int cu = 0; cu < 16; ++cu)
            if (seen[x][se][sh][cu]) printf(" %d.%d.%d", x, se, cu);
        printf("\n");
    }
    free(h);
}

int main() {
    CK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d\n", prop.name, prop.multiProcessorCount);
    int can = 0;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("CanUseStreamWaitValue %d\n", can);
    unsigned* d_out;
    const int nblk = 4096;
    CK(hipMalloc(&d_out, nblk * 4));
    hipStream_t s0;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    where("unmasked", s0, d_out, nblk);
    const int nbits[] = {8, 16, 32, 64, 128, 192};
    for (int t = 0; t < 6; ++t) {
        uint32_t mask[8] = {0};
        for (int b = 0; b < nbits[t]; ++b) mask[b >> 5] |= 1u << (b & 31);
        hipStream_t sm;
        hipError_t e = hipExtStreamCreateWithCUMask(&sm, 8, mask);
        if (e != hipSuccess) { printf("CU mask stream (%d bits): %s\n", nbits[t], hipGetErrorString(e)); continue; }
        char nm[64]; snprintf(nm, sizeof(nm), "mask low %d bits", nbits[t]);
        where(nm, sm, d_out, nblk);
        CK(hipStreamDestroy(sm));
    }
    {   // whole XCDs taken out (bit i = CU i/8 of XCD i%8: XCD x is bit x of every byte), and one single CU taken out
        const uint32_t pats[4] = {0x7f7f7f7fu, 0xfefefefeu, 0x3f3f3f3fu, 0xffffffffu};
        const char* nms[4] = {"all but XCD 7", "all but XCD 0", "all but XCD 6,7", "all but bit 255"};
        for (int t = 0; t < 4; ++t) {
            uint32_t mask[8];
            for (int i = 0; i < 8; ++i) mask[i] = pats[t];
            if (t == 3) mask[7] = 0x7fffffffu;
            hipStream_t sm;
            if (hipExtStreamCreateWithCUMask(&sm, 8, mask) == hipSuccess) { where(nms[t], sm, d_out, nblk); CK(hipStreamDestroy(sm)); }
            else printf("%s: mask rejected\n", nms[t]);
        }
    }
    {   // the complement of the low 64 bits
        uint32_t mask[8];
        for (int i = 0; i < 8; ++i) mask[i] = i < 2 ? 0u : 0xffffffffu;
        hipStream_t sm;
        if (hipExtStreamCreateWithCUMask(&sm, 8, mask) == hipSuccess) { where("mask bits 64..255", sm, d_out, nblk); CK(hipStreamDestroy(sm)); }
    }

    // ---- hand-offs ---------------------------------------------------------------------------------------------------
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    {
        uint32_t mask[8] = {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0};
        CK(hipExtStreamCreateWithCUMask(&b, 8, mask));      // its own hardware queue
    }
    unsigned* flags;
    CK(hipMalloc(&flags, 4096));
    CK(hipMemset(flags, 0, 4096));
    unsigned* f0 = flags, *f1 = flags + 64, *err = flags + 128;
    const int IT = 2000;
    // (i) one stream, dependent tiny kernels
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 0; i < IT; ++i) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, a);
        CK(hipStreamSynchronize(a));
        const double t1 = now();
        if (rep) printf("tiny kernel chain, one stream: %.2f us per launch (host-paired)\n", (t1 - t0) / IT * 1e6);
    }
    // (ii) ping-pong by events: a: nop, record e1 ; b: wait e1, nop, record e2 ; a: wait e2 ...
    {
        hipEvent_t* ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * 2 * IT);
        for (int i = 0; i < 2 * IT; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            for (int i = 0; i < IT; ++i) {
                hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, a);
                CK(hipEventRecord(ev[2 * i], a));
                CK(hipStreamWaitEvent(b, ev[2 * i], 0));
                hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, b);
                CK(hipEventRecord(ev[2 * i + 1], b));
                CK(hipStreamWaitEvent(a, ev[2 * i + 1], 0));
            }
            CK(hipDeviceSynchronize());
            const double t1 = now();
            if (rep) printf("ping-pong by events:           %.2f us per round trip (2 kernels + 2 cross-stream hops)\n", (t1 - t0) / IT * 1e6);
        }
    }
    // (iii) ping-pong by flags written and polled by kernels (monotonic values)
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(flags, 0, 4096));
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 1; i <= IT; ++i) {
            // a: [wait f1 >= i-1] set f0 = i ; b: wait f0 >= i, set f1 = i      (the wait and the set of a side are ONE kernel)
            hipLaunchKernelGGL(k_wait_set, dim3(1), dim3(64), 0, a, f1, (unsigned)(i - 1), f0, (unsigned)i, err);
            hipLaunchKernelGGL(k_wait_set, dim3(1), dim3(64), 0, b, f0, (unsigned)i, f1, (unsigned)i, err);
        }
        CK(hipDeviceSynchronize());
        const double t1 = now();
        unsigned e = 0;
        CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        if (rep) printf("ping-pong by kernel flags:     %.2f us per round trip (2 kernels + 2 flag hops), err %u\n", (t1 - t0) / IT * 1e6, e);
    }
    // (iv) ping-pong by hipStreamWriteValue32 / hipStreamWaitValue32
    if (can) {
        unsigned* sig;
        if (hipExtMallocWithFlags((void**)&sig, 4096, hipMallocSignalMemory) != hipSuccess) { (void)hipGetLastError(); CK(hipMalloc(&sig, 4096)); }
        CK(hipMemset(sig, 0, 4096));
        bool ok = true;
        for (int rep = 0; rep < 2 && ok; ++rep) {
            CK(hipMemset(sig, 0, 8));
            CK(hipDeviceSynchronize());
            const double t0 = now();
            for (int i = 1; i <= IT && ok; ++i) {
                hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, a);
                ok = ok && hipStreamWriteValue32(a, sig, (unsigned)i, 0) == hipSuccess;
                ok = ok && hipStreamWaitValue32(b, sig, (unsigned)i, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
                hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, b);
                ok = ok && hipStreamWriteValue32(b, sig + 1, (unsigned)i, 0) == hipSuccess;
                ok = ok && hipStreamWaitValue32(a, sig + 1, (unsigned)i, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
            }
            CK(hipDeviceSynchronize());
            const double t1 = now();
            if (rep && ok) printf("ping-pong by stream values:    %.2f us per round trip (2 kernels + 2 write/wait pairs)\n", (t1 - t0) / IT * 1e6);
        }
        if (!ok) printf("stream value ops failed: %s\n", hipGetErrorString(hipGetLastError()));
    }
    // (v) one-way: a sets flags as fast as it can; b waits on each (how far behind does the waiter run?)
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(flags, 0, 4096));
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 1; i <= IT; ++i) {
            hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, a, f0, (unsigned)i);
            hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, b, f0, (unsigned)i, err);
        }
        CK(hipDeviceSynchronize());
        const double t1 = now();
        if (rep) printf("one-way set / wait chains:     %.2f us per pair\n", (t1 - t0) / IT * 1e6);
    }
    printf("done\n");
    return 0;
}

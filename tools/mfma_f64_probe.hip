// Developer probe: operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950 (prints, for every lane and result register, which
// (i, j) of D = A B it holds).  hipcc --offload-arch=gfx950 -O2 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out) {
    const int l = threadIdx.x;
    // assumed operand layout: A lane l = (i = l % 16, k = l / 16), B lane l = (j = l % 16, k = l / 16)
    const int i = l % 16, kk = l / 16;
    double a = (kk == 0) ? (double)(i + 1) : 0.0;          // A[i][0] = i + 1
    double b = (kk == 0) ? (double)(100 * (i + 1)) : 0.0;  // B[0][j] = 100 (j + 1)
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
    // second product: distinguishes k placement: A[i][k] = 1 for all, B[k][j] = 10^k -> D = 1111 if each lane group holds a distinct k
    double a2 = 1.0, b2 = kk == 0 ? 1.0 : (kk == 1 ? 10.0 : (kk == 2 ? 100.0 : 1000.0));
    d4 c2 = {0, 0, 0, 0};
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, c2, 0, 0, 0);
    out[256 + l] = c2[0];
}
int main() {
    double* d; hipMalloc(&d, 512 * 8);
    k<<<1, 64>>>(d);
    double h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok_a = 1, ok_b = 1;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        const long val = (long)h[l * 4 + v];      // 100 (i + 1)(j + 1)
        const int j = l % 16;
        const long ij = val / 100;                // (i + 1)(j + 1)
        const int i = (int)(ij / (j + 1)) - 1;
        if (i != 4 * (l / 16) + v) ok_a = 0;
        if (i != (l / 16) + 4 * v) ok_b = 0;
        if (l < 20 || l % 16 == 0) printf("lane %2d v %d: value %ld -> i = %d (j = %d)\n", l, v, val, i, j);
    }
    printf("layout i = 4 (l / 16) + v : %s\nlayout i = (l / 16) + 4 v : %s\nk-sum check (expect 1111): %g\n", ok_a ? "YES" : "no", ok_b ? "YES" : "no", h[256]);
    return 0;
}

// Shared by the translation units of the low-rank codecs (cfx_lowrank.hip: the C-space chain; cfx_lrgram.hip: the N-space chain).
#ifndef CFX_LR_H
#define CFX_LR_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cfx.h"
#include "cfx_internal.h"

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LR_MAXB CFX_MAX_BATCH

struct LrItem {
    const h16* x; const h16* base; h16* new_base; void* packet;
    const float* q0;     // C x RP fp32 start (rank columns used, the rest zero)
    char* ws;            // this tensor's workspace
};
struct LrBatch { LrItem it[LR_MAXB]; };
struct LrDec { const h16* U; const h16* V; const h16* base; h16* out; };
struct LrDecBatch { LrDec it[LR_MAXB]; };

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline int lr_rp(int rank) { return rank <= 8 ? 8 : (rank <= 16 ? 16 : 32); }

// cfx_lowrank.hip
#define CFX_I_FLAG_LR_FACTORS_ONLY 0x100     // cfx_lr_compress_batch (quantized = 1): stop when fp16 U (N x r), V^T (C x r) are in the workspace
#define CFX_I_FLAG_LR_ABS 0x200              // ... of |x - base| instead of x - base
extern "C" CFX_HIDDEN size_t cfx_i_lr_workspace_bytes_any(int N, int C, int rank, int batch);
extern "C" CFX_HIDDEN void cfx_i_lr_factor_offsets(int N, int C, int rank, size_t* offU16, size_t* offV16, size_t* per);
extern "C" CFX_HIDDEN int cfx_i_lr_decode_launch(cfx_ctx* ctx, int N, int C, int rank, int batch, const LrDec* items, bool vt, hipStream_t s);
// cfx_lrgram.hip: the N-space ("Gram") chain
CFX_HIDDEN bool cfx_i_lrg_ok(int N, int C);
CFX_HIDDEN size_t cfx_i_lrg_extra_bytes(int N, int C, int RP);
CFX_HIDDEN int cfx_i_lrg_factors(cfx_ctx* ctx, int quantized, int N, int C, int rank, int batch, const LrBatch& b, size_t offD, size_t offU16,
                                 size_t offV16, size_t extra, int absd, int want_decode, int* decoded, hipStream_t s);

// cfx_lrslab.hip: the slab-resident chain (one persistent launch)
CFX_HIDDEN bool cfx_i_lrs_ok(int N, int C, int RP);
CFX_HIDDEN size_t cfx_i_lrs_extra_bytes(int N, int C, int RP);
CFX_HIDDEN int cfx_i_lrs_fit(cfx_ctx* ctx, int N, int C, int RP, void* stream);
CFX_HIDDEN int cfx_i_lrs_factors(cfx_ctx* ctx, int quantized, int N, int C, int rank, int batch, const LrBatch& b, size_t offU16, size_t offV16,
                                 size_t extra, int absd, int want_decode, int* decoded, hipStream_t s);

// T (RP x RP fp32, upper triangular, row-major at T[i * RP + m]) = chol(G)^-T for the symmetrised G (RP x RP fp64 in LDS); rank = r <= RP.
// A non-positive pivot (rank-deficient residual, e.g. x == base) zeroes that direction instead of producing NaNs.  Called by EVERY
// thread of a workgroup of NT threads (NT >= 64, multiple of 64); the factorisation itself runs in the registers of wave 0.  G and L
// are scratch (RP x (RP + 1) doubles each); the caller synchronises before reading T.
template <int RP, int NT>
__device__ __forceinline__ void lr_chol_T(double (*G)[RP + 1], double (*L)[RP + 1], int r, float* T, double* gmax_s, double* dinv_s,
                                          double pivot_tol = 1e-13) {
    const int tid = threadIdx.x;
    double gsym[(RP * RP + NT - 1) / NT];
#pragma unroll
    for (int q = 0; q < (RP * RP + NT - 1) / NT; ++q) {
        const int e = tid + q * NT;
        gsym[q] = (e < RP * RP) ? 0.5 * (G[e / RP][e % RP] + G[e % RP][e / RP]) : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < (RP * RP + NT - 1) / NT; ++q) {
        const int e = tid + q * NT;
        if (e < RP * RP) { G[e / RP][e % RP] = gsym[q]; L[e / RP][e % RP] = 0.0; }
    }
    __syncthreads();
    // Right-looking Cholesky and the triangular inverse in the REGISTERS of one wave: lane i holds row i (RP doubles); the pivot
    // and the column entries the other lanes need travel by v_readlane (an SGPR broadcast), so a step has no LDS round trip and
    // no barrier - ~16 cycles per trailing-update element instead of two workgroup barriers per column (r = 32: 56 -> ~15 us per
    // call, most of what is left is the reduction of the partial Grams above).  Same operations in the same order as the
    // LDS form it replaces (no contraction): identical factors.
    if (tid == 0) {
        double m = 0.0;
        for (int k = 0; k < r; ++k) m = fmax(m, G[k][k]);
        *gmax_s = m;
    }
    __syncthreads();
    if (tid < 64) {
    const double gmax = *gmax_s;
    const int i = tid;
    auto bcast = [](double v, int lane) -> double {          // value of lane `lane` (wave-uniform index) in every lane
        const int2 q = __builtin_bit_cast(int2, v);
        int2 o;
        o.x = __builtin_amdgcn_readlane(q.x, lane);
        o.y = __builtin_amdgcn_readlane(q.y, lane);
        return __builtin_bit_cast(double, o);
    };
    double g[RP];
#pragma unroll
    for (int k = 0; k < RP; ++k) g[k] = (i < RP) ? G[i][k] : 0.0;
    unsigned deadmask = 0;                                  // wave-uniform
#pragma unroll
    for (int j = 0; j < RP; ++j) {
        const double piv = bcast(g[j], j);
        const bool bad = (j >= r) || !(piv > gmax * pivot_tol);
        // 1 / sqrt(piv) by v_rsq_f64 + two Newton steps (full fp64 accuracy) instead of a correctly rounded sqrt and a division on
        // the critical path of every column; the factor T leaves this kernel as fp32
        double inv = __builtin_amdgcn_rsq(bad ? 1.0 : piv);
        inv = inv * (1.5 - 0.5 * (bad ? 1.0 : piv) * inv * inv);
        inv = inv * (1.5 - 0.5 * (bad ? 1.0 : piv) * inv * inv);
        double l = 0.0;
        if (i >= j && i < r && j < r) l = bad ? (i == j ? 1.0 : 0.0) : (i == j ? piv * inv : g[j] * inv);
        if (bad && j < r) deadmask |= 1u << j;
        if (i == 0) dinv_s[j] = bad ? 1.0 : inv;
        g[j] = l;                                            // column j of L replaces column j of G
#pragma unroll
        for (int k = j + 1; k < RP; ++k) {
            const double lk = bcast(l, k);
            g[k] -= l * lk;      // every lane, every k > j: rows <= j have l = 0 or only touch their unused upper part, and so do the
                                 // entries k > i - a per-(lane, k) condition would keep ~100 exec masks alive in SGPRs (1.8 k spills)
        }
        __builtin_amdgcn_sched_barrier(0);                   // keep a step's broadcasts (SGPRs) from being hoisted across steps
    }
    // X = L^-1: lane i computes column i (x[m] = X[m][i]) by forward substitution; the rows of L go through LDS once (same wave:
    // in order, no barrier) and are read back as broadcasts (one address for all lanes) - as SGPR broadcasts the ~500 entries
    // were all kept alive at once (1.6 k SGPR spills).  Two accumulators halve the dependent chain of a row.
    if (i < RP) {
#pragma unroll
        for (int k = 0; k < RP; ++k) L[i][k] = g[k];
    }
    double x[RP];
#pragma unroll
    for (int m = 0; m < RP; ++m) {
        double s0 = (m == i) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k + 1 < m; k += 2) { s0 -= L[m][k] * x[k]; s1 -= L[m][k + 1] * x[k + 1]; }   // x[k] = 0 for k < i: no mask per (lane, k)
        if (m & 1) s0 -= L[m][m - 1] * x[m - 1];
        x[m] = (m >= i && m < r && i < r) ? (s0 + s1) * dinv_s[m] : 0.0;
    }
    if (i < RP) {
    // T[k][j] = X[j][k] (k <= j): Q = Z T; thread i writes row i of T.  Directions with a vanished pivot are dropped.
#pragma unroll
    for (int m = 0; m < RP; ++m) {
        float v = 0.f;
        if (i < r && m < r && m >= i && !((deadmask >> m) & 1u)) v = (float)x[m];
        T[i * RP + m] = v;
    }
    }
    }
}
// Cholesky factor of the symmetrised G (RP x RP fp64 in LDS) in the registers of ONE wave: lane i holds row i, the pivot and the
// column entries the other lanes need travel by v_readlane.  Out (LDS, fp32): L row-major (zero above the diagonal), 1 / L[j][j], and
// the mask of directions whose pivot is not above tol x the largest diagonal entry (rank-deficient residual, e.g. x == base): their
// column of L is the unit vector, the caller zeroes them.  No triangular inverse: the caller solves Y L^T = W row by row.
template <int RP>
__device__ __forceinline__ unsigned lr_chol_rows(const double (*G)[RP + 1], int r, double tol, bool own_diag, double (&g)[RP], double& myinv) {
    const int i = threadIdx.x & 63, ii = i < RP ? i : 0;
    auto bcast = [](double v, int lane) -> double {          // value of lane `lane` (a constant) in every lane
        const int2 q = __builtin_bit_cast(int2, v);
        int2 o;
        o.x = __builtin_amdgcn_readlane(q.x, lane);
        o.y = __builtin_amdgcn_readlane(q.y, lane);
        return __builtin_bit_cast(double, o);
    };
    double gmax = 0.0;
#pragma unroll
    for (int k = 0; k < RP; ++k) g[k] = 0.5 * (G[ii][k] + G[k][ii]);
#pragma unroll
    for (int k = 0; k < RP; ++k) gmax = (k < r) ? fmax(gmax, G[k][k]) : gmax;
    const double thr = gmax * tol;
    unsigned dead = 0;
    myinv = 1.0;
#pragma unroll
    for (int j = 0; j < RP; ++j) {
        const double piv = bcast(g[j], j);
        // a direction is dropped when its pivot is not above tol x the largest diagonal entry - or, own_diag, x ITS OWN diagonal entry
        // (what is left of the column after the earlier ones, relative to the column: the test for graded columns, whose small
        // diagonal entries are legitimate)
        const bool ok = (j < r) && (piv > (own_diag ? G[j][j] * tol : thr));
        const double pv = ok ? piv : 1.0;
        // 1 / sqrt(piv) by v_rsq_f64 (~26 bits) + one Newton step (~51 bits): no correctly rounded sqrt and no division on the chain
        // from one pivot to the next
        double inv = __builtin_amdgcn_rsq(pv);
        inv = inv * __builtin_fma(pv * inv, -0.5 * inv, 1.5);
        double l = ok ? g[j] * inv : (i == j ? 1.0 : 0.0);           // lane j: piv * inv = sqrt(piv)
        l = (i >= j && i < r) ? l : 0.0;
        myinv = (i == j && ok) ? inv : myinv;
        dead |= (ok || j >= r) ? 0u : (1u << j);
        g[j] = l;
#pragma unroll
        for (int k = j + 1; k < RP; ++k) g[k] = __builtin_fma(-l, bcast(l, k), g[k]);     // every lane, every k > j: rows above the diagonal have l = 0
        __builtin_amdgcn_sched_barrier(0);                           // keep a step's broadcasts (SGPRs) from being hoisted across steps
    }
    return dead;
}

#endif

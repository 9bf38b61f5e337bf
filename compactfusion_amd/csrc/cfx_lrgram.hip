// Low-rank residual codecs, N-space ("Gram") form of the factorisation chain - part of libcfx.so.
//
// The reference's subspace_iter (xfuser/compact/compress_lowrank.py:14-61) is, for A = x - base (N x C, N << C):
//     Q <- orth(A^T (A Q))  twice ;  U = orth(A Q) ;  V = U^T A
// The C-space chain of cfx_lowrank.hip follows it product by product: 13 launches, each 5-35 us for a few microseconds of
// traffic.  Here the same iteration runs in N-space.  With G = A A^T (N x N) and Y0 = A Q0:
//     W1 = G Y0        M1 = Y0^T W1 (= Z1^T Z1, Z1 = A^T Y0)     T1 = chol(M1)^-T     Y1 = W1 T1   (= A orth(Z1))
//     W2 = G Y1        M2 = Y1^T W2 (= Z2^T Z2)                  T2 = chol(M2)^-T     Y2 = W2 T2   (= A orth(Z2))
//     M3 = Y2^T Y2 = T2^T (W2^T W2) T2                           T3 = chol(M3)^-T     U  = Y2 T3   (= orth(A Q2))
//     V  = U^T A
// - identical in exact arithmetic, orthonormalisation after every multiplication included (without it the result drifts from the
// reference by up to 3e-3 on fast-decaying spectra; with it 1e-5 .. 6e-5, measured in numpy).  A is touched by three passes
// instead of seven, two of them pure fp16-input MFMA work with no range issue (A is fp16; Q0 and U are split hi + lo), and
// everything between is N x r sized.  Launches (r x r factorisations run in the LAST-ARRIVING workgroup of the launch that
// produced their input: ticket counter, write-through partials - the in-launch finalize of cfx_kernels.hip):
//     k_lrg_prep   D = x - base                                                      (materialised once, 3.3 MB at the FLUX shard)
//     k_lrg_gram   G = D D^T (64 x 64 tiles, upper triangle + mirror, 2 column slabs), Y0 = D Q0       v_mfma_f32_32x32x16_f16
//     k_lrg_gy<0>  W1 = G Y0, M1 partials; last arriver: T1                                           v_mfma_f32_32x32x2_f32
//     k_lrg_gy<1>  Y1 = W1 T1 (every workgroup, N x r x r), W2 = G Y1, M2 and W2^T W2 partials; last arriver: T2, T3, U
//     k_lrg_v      V = U^T D                                                                           v_mfma_f32_32x32x16_f16
//     (decode / int4 factor quantisation: the kernels the C-space chain ends with)
// All reductions have a fixed order (no float atomics): results are reproducible run to run.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "cfx_lr.h"

// Pivots below this fraction of the largest are dropped.  In N-space the Gram matrix carries its fp32 accumulation error (1e-6 relative)
// into every product, so the null directions of a rank-deficient residual surface with pivots ~1e-10 of the largest instead of ~1e-14;
// kept, they are normalised noise that is not orthogonal to the true directions (numpy replay of this chain: projection error 3e-2 ..
// 5e-1 on a rank-3 residual at 1e-13, 3.5e-4 at 1e-9 .. 1e-5).  1e-10 of a sigma^4-scaled pivot = singular values below 0.3 % of the largest.
#define LRG_PIVOT_TOL 1e-10
#define LRG_GQ 18          // float4 G operands per lane in k_lrg_gy: NP / 32 <= 18, i.e. N <= 576
#define LRG_LD 72          // halves per LDS row of a 64-column chunk: 144 B, 16-byte aligned, rows spread over the banks

__device__ __forceinline__ h16x8 lrg_ld8(const h16* p) { return *reinterpret_cast<const h16x8*>(p); }
__device__ __forceinline__ void lrg_st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float lrg_ld_wt(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
typedef unsigned lrg_u4 __attribute__((ext_vector_type(4)));
// 16-byte write-through store (an agent-scope atomic store lowers to sc1 only up to 8 bytes).  hipcc does not count an asm store: the
// publishing wave drains it with its own s_waitcnt vmcnt(0); the s_nop keeps the data registers alive until the store has read them
__device__ __forceinline__ void lrg_st16_wt(void* p, lrg_u4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void lrg_st_wt(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double lrg_ld_wt(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct LrgArgs {
    int N, C, NP, TN, npair, kslab, KS, r, batch;
    int per_tensor;          // workgroups per tensor of the launch
    int xcd_group;           // 8 / batch when that divides: a tensor's workgroups stay on ITS XCDs (block b runs on XCD b % 8), so its
                             // D (3.3 MB at the FLUX shard) stays resident in their L2s; 0: plain tensor-major order
    size_t offD, offG, offGp, offY0, offY0p, offW1, offW2, offMp, offPp, offT, offUf, offU16, offV16;
    int absd;                // factorise |x - base|
    int u_in_packet;         // LOW_RANK: U (N x r) and V (r x C) straight into the packet; LOW_RANK_Q: fp16 U (N x r), V^T (C x r) to the workspace
    unsigned* tick;
};

__device__ __forceinline__ bool lrg_block(const LrgArgs& a, int& z, int& idx) {
    const int bid = blockIdx.x;
    if (a.xcd_group) { z = (bid & 7) / a.xcd_group; idx = (bid >> 3) * a.xcd_group + (bid & 7) % a.xcd_group; }
    else { z = bid / a.per_tensor; idx = bid - z * a.per_tensor; }
    return z < a.batch && idx < a.per_tensor;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lrg_prep(LrBatch b, size_t n8, size_t offD, int absd) {
    const LrItem it = b.it[blockIdx.y];
    h16* D = (h16*)(it.ws + offD);
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n8; i += stride) {
        h16x8 v = lrg_ld8(it.x + i * 8);
        if (it.base) v = v - lrg_ld8(it.base + i * 8);                 // fp16, one rounding (torch eager: x - base)
        if (absd) {                                                    // |x - base|: the matrix behind the 1-bit codec's rank-K scales
            typedef unsigned short u16x8_ __attribute__((ext_vector_type(8)));
            u16x8_ bb = __builtin_bit_cast(u16x8_, v);
            bb &= (unsigned short)0x7fff;
            v = __builtin_bit_cast(h16x8, bb);
        }
        *reinterpret_cast<h16x8*>(D + i * 8) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// G slab ks (NP x NP fp32) = D[:, slab] D[:, slab]^T ; Y0 slab ks (NP x RP fp32) = D[:, slab] Q0[slab, :]
// Workgroups of a tensor: [pairs (ti <= tj) of 64-row tiles x 2 slabs | 64-row tiles x 2 slabs for Y0].  A 64-column chunk of both
// row tiles goes through LDS with coalesced 16-byte loads (the next chunk is in flight while this one is multiplied); wave w owns
// the 32 x 32 sub-tile (w & 1, w >> 1).  Operand layout of v_mfma_f32_32x32x16_f16: lane l holds row (A) / column (B) l & 31 and
// the 8 consecutive k = 8 (l >> 5) .. + 7 - the same k-set for A and B, which is all a dot product needs.
// ---------------------------------------------------------------------------------------------------------------------
template <int RP>
__global__ __launch_bounds__(256) void k_lrg_gram(LrBatch b, LrgArgs a) {
    int z, idx;
    if (!lrg_block(a, z, idx)) return;
    const LrItem it = b.it[z];
    const h16* D = (const h16*)(it.ws + a.offD);
    const int KS = a.KS;
    const int ks = idx % KS, u = idx / KS;
    const int k0 = ks * a.kslab, k1 = min(a.C, k0 + a.kslab);
    const int N = a.N, C = a.C, NP = a.NP;
    __shared__ h16 As[64 * LRG_LD];
    __shared__ h16 Bs[64 * LRG_LD];
    __shared__ unsigned last_flag;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    unsigned* tick = a.tick + z * 64 + 2 + u;                         // one ticket per tile pair / Y0 row tile (words 0, 1: k_lrg_gy)
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (u < a.npair) {
        int ti = 0, rem = u;
        while (rem >= a.TN - ti) { rem -= a.TN - ti; ++ti; }
        const int tj = ti + rem;
        const int i0 = ti * 64, j0 = tj * 64;
        const bool diag = ti == tj;
        const int si = w & 1, sj = w >> 1;
        // (Measured with early exits, K,V of the FLUX shard, 864 workgroups: the loads alone 10.7 of the kernel's 24.5 us - 70 MB of tile
        // re-reads arrive at ~7 TB/s whether they hit L2 or not - staging + MFMA +2.0, the write-through partial tiles +2.1, the pair
        // reduce +9.7; 4 / 2 column slabs instead of 8: 19.3 / 27.9 us.)
        // Up to 6 chunks (the whole slab of the FLUX / SD3 shards) are requested
        // at once - 24 x 16 bytes per thread in flight - and then staged through LDS chunk by chunk without another global wait
        constexpr int DEPTH = 6;
        h16x8 ra[DEPTH][2], rb[DEPTH][2];
        for (int g0 = k0; g0 < k1; g0 += DEPTH * 64) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int c0 = g0 + d * 64;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int p = tid + 256 * q, row = p >> 3, c8 = (p & 7) * 8;
                    ra[d][q] = (h16x8)(h16)0;
                    rb[d][q] = (h16x8)(h16)0;
                    if (c0 < k1) {
                        if (i0 + row < N) ra[d][q] = lrg_ld8(D + (size_t)(i0 + row) * C + c0 + c8);
                        if (diag) rb[d][q] = ra[d][q];
                        else if (j0 + row < N) rb[d][q] = lrg_ld8(D + (size_t)(j0 + row) * C + c0 + c8);
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (g0 + d * 64 < k1) {                               // uniform
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int p = tid + 256 * q, row = p >> 3, c8 = (p & 7) * 8;
                        *reinterpret_cast<h16x8*>(&As[row * LRG_LD + c8]) = ra[d][q];
                        *reinterpret_cast<h16x8*>(&Bs[row * LRG_LD + c8]) = rb[d][q];
                    }
                    __syncthreads();
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const h16x8 av = *reinterpret_cast<const h16x8*>(&As[(si * 32 + li) * LRG_LD + kk * 16 + lh * 8]);
                        const h16x8 bv = *reinterpret_cast<const h16x8*>(&Bs[(sj * 32 + li) * LRG_LD + kk * 16 + lh * 8]);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
                    }
                }
            }
        }
        // this slab's partial tile, write-through; C/D layout: column j = lane & 31, row i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        float* Gp = (float*)(it.ws + a.offGp) + ((size_t)u * KS + ks) * 4096;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) lrg_st_wt(&Gp[(si * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * lh) * 64 + sj * 32 + li], acc[rg]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = old == (unsigned)(KS - 1);
            if (last_flag) __hip_atomic_store(tick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!last_flag) return;
        // the pair's last slab: sum the KS partial tiles in fixed order, write the tile and its mirror image
        float* G = (float*)(it.ws + a.offG);
        const float* P0 = (const float*)(it.ws + a.offGp) + (size_t)u * KS * 4096;
        // 2048 float2 positions, 8 per thread, KS slabs each: every load is issued before any is consumed (one fabric round trip);
        // 8-byte agent-scope loads = global_load_dwordx2 sc1, which the compiler tracks (an asm dwordx4 load it does not)
        typedef unsigned long long u64_;
        for (int half = 0; half < 2; ++half) {
            u64_ pv[4][8];
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    pv[v][q] = __hip_atomic_load((const u64_*)(P0 + (size_t)min(q, KS - 1) * 4096) + (tid + 256 * (v + 4 * half)), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {                         // fixed order
                    const float2 f = __builtin_bit_cast(float2, pv[v][q]);
                    s0 += (q < KS) ? f.x : 0.f;
                    s1 += (q < KS) ? f.y : 0.f;
                }
                const int e = (tid + 256 * (v + 4 * half)) * 2, gi = i0 + (e >> 6), gj = j0 + (e & 63);
                *reinterpret_cast<float2*>(&G[(size_t)gi * NP + gj]) = make_float2(s0, s1);
                if (!diag) { G[(size_t)gj * NP + gi] = s0; G[(size_t)(gj + 1) * NP + gi] = s1; }
            }
        }
        return;
    }
    // ---- Y0 slab: 64 rows x RP, Q0 as hi + lo fp16 (the product is then exact to 2^-22 of an fp32 product) ----
    const int ti = u - a.npair, i0 = ti * 64;
    const int si = w & 1, part = w >> 1;                              // waves 0, 1: hi ; 2, 3: lo
    h16* Qh = Bs;                                                     // [32][LRG_LD]: row n = column of Q0, k along the row
    h16* Ql = Bs + 32 * LRG_LD;
    for (int i = tid; i < 64 * LRG_LD; i += 256) Bs[i] = (h16)0;      // rows n >= RP stay zero
    constexpr int QV = 64 * RP / 256;                                 // floats of the Q0 chunk per thread
    h16x8 ra[2];
    float rq[QV];
    auto load = [&](int c0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = tid + 256 * q, row = p >> 3, c8 = (p & 7) * 8;
            ra[q] = (h16x8)(h16)0;
            if (i0 + row < N) ra[q] = lrg_ld8(D + (size_t)(i0 + row) * C + c0 + c8);
        }
#pragma unroll
        for (int q = 0; q < QV; ++q) {
            const int e = tid + 256 * q;                             // e = kk * RP + n
            rq[q] = it.q0[(size_t)c0 * RP + e];
        }
    };
    load(k0);
    for (int c0 = k0; c0 < k1; c0 += 64) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = tid + 256 * q, row = p >> 3, c8 = (p & 7) * 8;
            *reinterpret_cast<h16x8*>(&As[row * LRG_LD + c8]) = ra[q];
        }
#pragma unroll
        for (int q = 0; q < QV; ++q) {
            const int e = tid + 256 * q, kk = e / RP, n = e - kk * RP;
            const h16 hi = (h16)rq[q];
            Qh[n * LRG_LD + kk] = hi;
            Ql[n * LRG_LD + kk] = (h16)(rq[q] - (float)hi);
        }
        __syncthreads();
        if (c0 + 64 < k1) load(c0 + 64);
        const h16* Qp = part ? Ql : Qh;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const h16x8 av = *reinterpret_cast<const h16x8*>(&As[(si * 32 + li) * LRG_LD + kk * 16 + lh * 8]);
            const h16x8 bv = *reinterpret_cast<const h16x8*>(&Qp[li * LRG_LD + kk * 16 + lh * 8]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
        }
    }
    __syncthreads();
    float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(As);   // 2 x 32 x 33 floats = 8448 B <= 9216 B
    if (part) {
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) red[si][(rg & 3) + 8 * (rg >> 2) + 4 * lh][li] = acc[rg];
    }
    __syncthreads();
    float* Yp = (float*)(it.ws + a.offY0p) + ((size_t)ti * KS + ks) * 64 * RP;      // this slab's partial rows, write-through
    if (!part) {
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            const int row = (rg & 3) + 8 * (rg >> 2) + 4 * lh;
            if (li < RP) lrg_st_wt(&Yp[(si * 32 + row) * RP + li], acc[rg] + red[si][row][li]);      // hi + lo
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = old == (unsigned)(KS - 1);
        if (last_flag) __hip_atomic_store(tick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last_flag) return;
    float* Y0 = (float*)(it.ws + a.offY0);
    const float* P0 = (const float*)(it.ws + a.offY0p) + (size_t)ti * KS * 64 * RP;
    for (int e = tid; e < 64 * RP; e += 256) {
        float pq[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) pq[q] = lrg_ld_wt(&P0[(size_t)min(q, KS - 1) * 64 * RP + e]);     // unconditional: all in flight together
        float sacc = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) sacc += (q < KS) ? pq[q] : 0.f;
        Y0[(size_t)i0 * RP + e] = sacc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// W (32-row tile) = G Y with Y = Y0 (MODE 0: the two slabs summed) or W1 T1 (MODE 1), G = the two slabs summed.
// fp32-input MFMA (v_mfma_f32_32x32x2_f32: an exact fp32 FMA chain): A[i][k] = G[i0 + i][j], B[k][n] = Y[j][n]; a 64-column chunk
// of the G rows is staged through LDS (coalesced), all of Y sits in LDS; the 4 waves take 16 columns of the chunk each and their
// partial tiles are summed in fixed order.  Epilogue: partial M = Y_tile^T W_tile (fp64; MODE 1 also P = W_tile^T W_tile), then the
// ticket - the last workgroup of the tensor to arrive factorises:
//   MODE 0:  T1 = chol(sum M)^-T                                   -> T[0]
//   MODE 1:  T2 = chol(sum M)^-T ; M3 = T2^T (sum P) T2 ; T3 = chol(M3)^-T ; U = W2 (T2 T3)      -> Uf (fp32), U fp16 (packet / workspace)
// Dynamic LDS: doubles Gd, Ld, Sd [RP][RP+1], misc | Ts, T2s [RP * RP] | red[4][32][33] | T23 [RP * RP] | Ys [NP * RP]
// ---------------------------------------------------------------------------------------------------------------------
// (one wave per SIMD: the single-wave factorisation of the last arriver wants the whole register file at RP = 32)
template <int RP, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_lrg_gy(LrBatch b, LrgArgs a) {
    int z, idx;
    if (!lrg_block(a, z, idx)) return;
    const LrItem it = b.it[z];
    const int N = a.N, NP = a.NP, r = a.r;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int i0 = idx * 32, ntiles = a.per_tensor;
    extern __shared__ double lrg_smem[];
    double (*Gd)[RP + 1] = reinterpret_cast<double (*)[RP + 1]>(lrg_smem);
    double (*Ld)[RP + 1] = Gd + RP;
    double (*Sd)[RP + 1] = Ld + RP;
    double* misc = reinterpret_cast<double*>(Sd + RP);                // [0] gmax, [1 .. RP] dinv
    float* Ts = reinterpret_cast<float*>(misc + RP + 2);             // RP * RP
    float* T2s = Ts + RP * RP;                                        // RP * RP (MODE 1 finalize)
    float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(T2s + RP * RP);
    float* T23 = reinterpret_cast<float*>(red + 4);                   // RP * RP (MODE 1 finalize)
    float* Ys = T23 + RP * RP;                                        // NP x RP
    __shared__ unsigned last_flag;
    float* W = (float*)(it.ws + (MODE ? a.offW2 : a.offW1));
    float* Tg = (float*)(it.ws + a.offT);                             // T1 | T2 | T3, RP * RP each

    // ---- this wave's share of the G rows: straight from L2 into the MFMA operand layout, ALL loads in flight before anything else.
    // Wave w takes columns [w QW, (w + 1) QW), QW = NP / 4; lane (li, lh) holds G[i0 + li][w QW + 8 q + 4 lh .. + 3] (32-byte segments
    // of 32 rows per instruction: every byte is used).  A staged copy through LDS cost 9 dependent round trips per workgroup. ----
    const int QW = NP >> 2, nq = QW >> 3;
    float4 gq[LRG_GQ];
    {
        const float* Gr = (const float*)(it.ws + a.offG) + (size_t)(i0 + li) * NP + w * QW + 4 * lh;
#pragma unroll
        for (int q = 0; q < LRG_GQ; ++q) {
            gq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < nq) gq[q] = *reinterpret_cast<const float4*>(Gr + 8 * q);
        }
    }
    // ---- Y into LDS ----
    if (MODE == 0) {
        const float* Ya = (const float*)(it.ws + a.offY0);
        for (int i = tid; i < NP * RP; i += 256) Ys[i] = (i / RP < N) ? Ya[i] : 0.f;
    } else {
        for (int i = tid; i < RP * RP; i += 256) Ts[i] = Tg[i];
        __syncthreads();
        const float* W1 = (const float*)(it.ws + a.offW1);
        for (int j = tid; j < NP; j += 256) {
            float in[RP], out[RP];
#pragma unroll
            for (int k = 0; k < RP; ++k) in[k] = (j < N) ? W1[(size_t)j * RP + k] : 0.f;
#pragma unroll
            for (int n = 0; n < RP; ++n) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < RP; ++k) s = fmaf(in[k], Ts[k * RP + n], s);
                out[n] = s;
            }
#pragma unroll
            for (int n = 0; n < RP; ++n) Ys[j * RP + n] = out[n];
        }
    }
    // ---- W tile = G rows x Y ----
    __syncthreads();                                                  // Ys complete
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int q = 0; q < LRG_GQ; ++q) {
        if (q < nq) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = w * QW + 8 * q + 4 * lh + e;        // k = lh of this MFMA step: the same column for A and B
                const float av = e == 0 ? gq[q].x : (e == 1 ? gq[q].y : (e == 2 ? gq[q].z : gq[q].w));
                const float bv = (li < RP) ? Ys[col * RP + li] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) red[w][(rg & 3) + 8 * (rg >> 2) + 4 * lh][li] = acc[rg];
    __syncthreads();
    for (int i = tid; i < 32 * RP; i += 256) {
        const int row = i / RP, n = i - row * RP;
        const float s = ((red[0][row][n] + red[1][row][n]) + red[2][row][n]) + red[3][row][n];
        red[0][row][n] = s;                                           // (row, n) is read and written by this thread only
        lrg_st_wt(&W[(size_t)(i0 + row) * RP + n], s);
    }
    __syncthreads();
    {
        double* Mp = (double*)(it.ws + a.offMp) + (size_t)idx * RP * RP;
        double* Pp = (double*)(it.ws + a.offPp) + (size_t)idx * RP * RP;
        for (int i = tid; i < RP * RP; i += 256) {
            const int p = i / RP, q = i - p * RP;
            double m = 0.0, pp = 0.0;
            for (int row = 0; row < 32; ++row) {
                const double wv = (double)red[0][row][q];
                m += (double)Ys[(i0 + row) * RP + p] * wv;
                if (MODE) pp += (double)red[0][row][p] * wv;
            }
            lrg_st_wt(&Mp[i], m);
            if (MODE) lrg_st_wt(&Pp[i], pp);
        }
    }
    // publish: every storing wave drains its write-through stores, then one lane draws the ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.tick + z * 64 + MODE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = old == (unsigned)(ntiles - 1);
        if (last_flag) __hip_atomic_store(a.tick + z * 64 + MODE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-resetting
    }
    __syncthreads();
    if (!last_flag) return;
    // ---- the tensor's last workgroup: the factorisations ----
    {
        const double* Mp = (const double*)(it.ws + a.offMp);
        const double* Pp = (const double*)(it.ws + a.offPp);
        // every load unconditional (clamped tile index, masked value) and issued before any is consumed: one fabric round trip per
        // element instead of one per tile (cdna_hip_programming.md: a branch per load serialises them)
        for (int i = tid; i < RP * RP; i += 256) {
            double mv[LRG_GQ], pv[LRG_GQ];
#pragma unroll
            for (int t = 0; t < LRG_GQ; ++t) {
                mv[t] = lrg_ld_wt(&Mp[(size_t)min(t, ntiles - 1) * RP * RP + i]);
                if (MODE) pv[t] = lrg_ld_wt(&Pp[(size_t)min(t, ntiles - 1) * RP * RP + i]);
            }
            double m = 0.0, pp = 0.0;
#pragma unroll
            for (int t = 0; t < LRG_GQ; ++t) {                       // fixed order
                m += (t < ntiles) ? mv[t] : 0.0;
                if (MODE) pp += (t < ntiles) ? pv[t] : 0.0;
            }
            Gd[i / RP][i % RP] = m;
            if (MODE) Sd[i / RP][i % RP] = pp;
        }
    }
    __syncthreads();
    lr_chol_T<RP, 256>(Gd, Ld, r, MODE ? T2s : Ts, &misc[0], &misc[1], LRG_PIVOT_TOL);
    __syncthreads();
    if (MODE == 0) {
        for (int i = tid; i < RP * RP; i += 256) Tg[i] = Ts[i];
        return;
    }
    // M3 = T2^T S T2 (S = W2^T W2 symmetric): first X = S T2 into Ld, then M3 = T2^T X into Gd
    for (int i = tid; i < RP * RP; i += 256) {
        const int p = i / RP, q = i - p * RP;
        double s = 0.0;
        for (int k = 0; k < RP; ++k) s += 0.5 * (Sd[p][k] + Sd[k][p]) * (double)T2s[k * RP + q];
        Ld[p][q] = s;
    }
    __syncthreads();
    for (int i = tid; i < RP * RP; i += 256) {
        const int p = i / RP, q = i - p * RP;
        double s = 0.0;
        for (int k = 0; k < RP; ++k) s += (double)T2s[k * RP + p] * Ld[k][q];
        Gd[p][q] = s;
    }
    __syncthreads();
    lr_chol_T<RP, 256>(Gd, Ld, r, Ts, &misc[0], &misc[1], LRG_PIVOT_TOL);          // T3 -> Ts
    __syncthreads();
    // T23 = T2 T3, then U = W2 T23 for every row
    for (int i = tid; i < RP * RP; i += 256) {
        const int p = i / RP, q = i - p * RP;
        float s = 0.f;
        for (int k = 0; k < RP; ++k) s = fmaf(T2s[p * RP + k], Ts[k * RP + q], s);
        T23[i] = s;
    }
    __syncthreads();
    float* Uf = (float*)(it.ws + a.offUf);
    h16* U16 = a.u_in_packet ? (h16*)it.packet : (h16*)(it.ws + a.offU16);
    for (int j = tid; j < NP; j += 256) {
        float in[RP], out[RP];
#pragma unroll
        for (int k = 0; k < RP; ++k) in[k] = (j < N) ? lrg_ld_wt(&W[(size_t)j * RP + k]) : 0.f;
#pragma unroll
        for (int n = 0; n < RP; ++n) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < RP; ++k) s = fmaf(in[k], T23[k * RP + n], s);
            out[n] = s;
        }
#pragma unroll
        for (int n = 0; n < RP; ++n) Uf[(size_t)j * RP + n] = out[n];
        if (j < N) {
#pragma unroll
            for (int n = 0; n < RP; ++n)
                if (n < r) U16[(size_t)j * r + n] = (h16)out[n];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// V (r x 64-column block) = U^T D: fp16-input MFMA with U as hi + lo.  Both operands want 8 consecutive k = 8 consecutive ROWS of
// D / U per lane, so a 64-row chunk of the D block and of U is TRANSPOSED into LDS ([column][row], [rank index][row]).
// Waves: column sub-tile (w & 1) x {hi, lo} (w >> 1).  Output: LOW_RANK V (r x C) fp16 at packet + N r halves; LOW_RANK_Q V^T (C x r).
// ---------------------------------------------------------------------------------------------------------------------
template <int RP>
__global__ __launch_bounds__(256) void k_lrg_v(LrBatch b, LrgArgs a) {
    int z, idx;
    if (!lrg_block(a, z, idx)) return;
    const LrItem it = b.it[z];
    const h16* D = (const h16*)(it.ws + a.offD);
    const float* Uf = (const float*)(it.ws + a.offUf);
    const int N = a.N, C = a.C, r = a.r;
    const int c0 = idx * 64;
    __shared__ h16 Dt[64 * LRG_LD];                                   // [column][row]
    __shared__ h16 Uh[32 * LRG_LD], Ul[32 * LRG_LD];                  // [rank index][row]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int sj = w & 1, part = w >> 1;
    for (int i = tid; i < 32 * LRG_LD; i += 256) { Uh[i] = (h16)0; Ul[i] = (h16)0; }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    constexpr int UV = 64 * RP / 256;
    h16x8 rd[2][2];                                                   // two 64-row chunks in flight
    float ru[2][UV];
    auto load = [&](int n0, int slot) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = tid + 256 * q, row = p >> 3, c8 = (p & 7) * 8;
            rd[slot][q] = (h16x8)(h16)0;
            if (n0 + row < N && c0 + c8 < C) rd[slot][q] = lrg_ld8(D + (size_t)(n0 + row) * C + c0 + c8);
        }
#pragma unroll
        for (int q = 0; q < UV; ++q) {
            const int e = tid + 256 * q, row = e / RP;
            ru[slot][q] = (n0 + row < N) ? Uf[(size_t)n0 * RP + e] : 0.f;
        }
    };
    auto stage = [&](const h16x8 (&xd)[2], const float (&xu)[UV]) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = tid + 256 * q, row = p >> 3, c8 = (p & 7) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) Dt[(c8 + e) * LRG_LD + row] = xd[q][e];
        }
#pragma unroll
        for (int q = 0; q < UV; ++q) {
            const int e = tid + 256 * q, row = e / RP, m = e - row * RP;
            const h16 hi = (h16)xu[q];
            Uh[m * LRG_LD + row] = hi;
            Ul[m * LRG_LD + row] = (h16)(xu[q] - (float)hi);
        }
        __syncthreads();
    };
    const h16* Up = part ? Ul : Uh;
    auto mma = [&]() {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const h16x8 av = *reinterpret_cast<const h16x8*>(&Up[li * LRG_LD + kk * 16 + lh * 8]);                 // A[m = li][k = row]
            const h16x8 bv = *reinterpret_cast<const h16x8*>(&Dt[(sj * 32 + li) * LRG_LD + kk * 16 + lh * 8]);     // B[k = row][n = column]
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
        }
    };
    load(0, 0);
    load(64, 1);
    for (int n0 = 0; n0 < N; n0 += 128) {
        stage(rd[0], ru[0]);
        load(n0 + 128, 0);
        mma();
        if (n0 + 64 < N) {
            stage(rd[1], ru[1]);
            load(n0 + 192, 1);
            mma();
        }
    }
    __syncthreads();
    float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(Dt);   // 2 x 32 x 33 floats = 8448 B <= 9216 B
    if (part) {
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) red[sj][(rg & 3) + 8 * (rg >> 2) + 4 * lh][li] = acc[rg];
    }
    __syncthreads();
    if (!part) {
        const int c = c0 + sj * 32 + li;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            const int m = (rg & 3) + 8 * (rg >> 2) + 4 * lh;          // rank index
            if (m < r && c < C) {
                const h16 v = (h16)(acc[rg] + red[sj][m][li]);
                if (a.u_in_packet) ((h16*)it.packet)[(size_t)N * r + (size_t)m * C + c] = v;
                else ((h16*)(it.ws + a.offV16))[(size_t)c * r + m] = v;
            }
        }
    }
}

// =====================================================================================================================
// The whole chain as ONE persistent launch (k_lrp).  The multi-launch form above spends most of its time between kernels: every
// product is followed by a last-arriver factorisation that the rest of the chip waits for, a kernel boundary, and a fresh ramp.
// Here a grid that is co-resident by construction (host: occupancy x the CUs the stream may use) walks through the stages with
// grid-wide barriers per tensor (an arrival counter, "open" words per XCD polled with L1-bypassing loads, agent-scope release /
// acquire fences around them - what a kernel boundary does, minus the boundary):
//   S0  D = x - base                                                                 (all workgroups, grid-stride)
//   S1  G_ks = D[:, slab ks] D[:, slab ks]^T  (64 x 64 tile pairs, mirror written), Y0_ks = D[:, slab ks] Q0[slab ks]
//   S3  row tile t (32 rows): G rows = sum_ks G_ks rows, kept in REGISTERS for S4; W1 = G Y0; partial M1 = Y0^T W1
//   S4  (same workgroups) T1 = chol(sum M1)^-T, Y1 = W1 T1, W2 = G Y1, partial M2 = Y1^T W2, P = W2^T W2
//   S5  every workgroup: T2, T3 (the two r x r factorisations, redundantly - nobody waits for a broadcast), U = W2 T2 T3 into LDS;
//       then per 64-column block: V = U^T D, and - LOW_RANK with error feedback - the state update new_base = base + fp16(U V) from
//       the same U and V, with the arithmetic of k_lr_decode (the receiver's kernel: states stay bit-identical)
// The barrier words are self-resetting (the last workgroup to leave zeroes them): nothing in the launch depends on a host-side
// counter, so it can be captured in a hipGraph.  A wait that never ends (the grid was not co-resident after all) gives up after the
// context's gate timeout and reports through the context's error word, like every other in-kernel wait of this library.
// =====================================================================================================================
#define LRP_MAX_KS 4               // column slabs of the Gram pass, at most
#define LRP_TW 256                 // barrier words per tensor: [0] arrivals, [1] leavers, [32 + 16 x] "open" word of XCD x
__device__ __forceinline__ void lrg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
struct LrpArgs {
    int N, C, NP, TN, npair, KS, kslab, r, batch, nwg_t, nt, zmod;
    int absd, u_in_packet, fuse_decode;
    size_t offD, offG, offY0p, offW1, offW2, offMp, offMp2, offPp, offU16, offV16;
    unsigned* tick;
    unsigned* err;
    long long timeout;
    unsigned long long* stamps;      // developer hook (cfx_debug_stamps): 16 words per workgroup, 100 MHz wall clock
};

// Grid barrier of one tensor's workgroups.  What the stages hand each other is stored WRITE-THROUGH (sc1: lrp_st / lrp_st16), so the
// release side is just "my stores have been acknowledged" - an L2 write-back (buffer_wbl2, what an agent-scope release fence emits)
// costs ~18 us here when 1024 waves each ask for one.  The acquire side invalidates the non-coherent L2 / L1 lines once per
// workgroup (buffer_inv sc1): the consumers read with plain loads so that tile re-reads hit their XCD's L2.
__device__ __forceinline__ void lrp_sync(unsigned* blk, unsigned stage, unsigned nwg, unsigned* err, long long timeout) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == stage * nwg) {
#pragma unroll
            for (int x = 0; x < 8; ++x) __hip_atomic_store(blk + 32 + 16 * x, stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // HW_REG_XCC_ID
            const unsigned* open = blk + 32 + 16 * xcc;
            const long long t0 = wall_clock64();
            while ((int)(__hip_atomic_load(open, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - stage) < 0) {
                __builtin_amdgcn_s_sleep(4);
                if (wall_clock64() - t0 > timeout) {
                    if (err) (void)__hip_atomic_fetch_add(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
        asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}
__device__ __forceinline__ void lrp_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lrp_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lrp_st16(void* p, lrg_u4 v) { lrg_st16_wt(p, v); }

// LDS carve-up (bytes); the stages reuse the same memory
template <int RP> struct LrpLds {
    static constexpr int chol = 3 * RP * (RP + 1) * 8 + (RP + 2) * 8;          // Gd, Ld, Sd, misc
    static constexpr int ts = 3 * RP * RP * 4;                                // Ts, T2s, T23
    static constexpr int head = (chol + ts + 255) / 256 * 256;
    static constexpr int red = 4 * 32 * 33 * 4;
    __host__ __device__ static int s1() { return 4 * 64 * LRG_LD * 2; }
    __host__ __device__ static int s34(int NP) { return head + red + NP * RP * 4; }
    __host__ __device__ static int ut(int NP) { return RP * (NP + 8) * 2; }    // one transposed half of U
    __host__ __device__ static int s5(int NP) { return head + 2 * ut(NP) + NP * RP * 2 + 64 * LRG_LD * 2 + RP * 64 * 2; }
    __host__ __device__ static int total(int NP) {
        int m = s1();
        if (s34(NP) > m) m = s34(NP);
        if (s5(NP) > m) m = s5(NP);
        return m;
    }
};

// One tile pair x one column slab of the Gram pass: acc = D[i0 .. +63, slab] D[j0 .. +63, slab]^T, 64-column chunks through a
// double-buffered LDS tile, DEPTH chunks of both row tiles in flight in registers (every load unconditional, one LDS-only barrier
// per chunk).  Writes the tile and, off the diagonal, its mirror image, write-through.
template <bool DIAG>
__device__ __forceinline__ void lrp_gram_pair(const h16* D, int C, int NP, int i0, int j0, int k0, int nch, h16* As, h16* Bs, float* Gk) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int si = w & 1, sj = w >> 1;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    constexpr int DEPTH = 6;
    h16x8 ra[DEPTH][2], rb[DEPTH][2];
    const int r0 = tid >> 3, c8 = (tid & 7) * 8;                      // this thread's rows r0, r0 + 32 of a tile, 8 columns at c8
    const h16* Da = D + (size_t)(i0 + r0) * C + k0 + c8;
    const h16* Db = D + (size_t)(j0 + r0) * C + k0 + c8;
    const size_t half = (size_t)32 * C;
    auto issue = [&](int ch, h16x8 (&xa)[2], h16x8 (&xb)[2]) {
        const int co = min(ch, nch - 1) * 64;                         // past the slab: a redundant load instead of a branch
        xa[0] = lrg_ld8(Da + co);
        xa[1] = lrg_ld8(Da + half + co);
        if (!DIAG) { xb[0] = lrg_ld8(Db + co); xb[1] = lrg_ld8(Db + half + co); }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d, ra[d], rb[d]);
    for (int cb = 0; cb < nch; cb += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int ch = cb + d;
            if (ch < nch) {                                           // uniform
                h16* Ab = As + (ch & 1) * 64 * LRG_LD;
                h16* Bb = Bs + (ch & 1) * 64 * LRG_LD;
                *reinterpret_cast<h16x8*>(&Ab[r0 * LRG_LD + c8]) = ra[d][0];
                *reinterpret_cast<h16x8*>(&Ab[(r0 + 32) * LRG_LD + c8]) = ra[d][1];
                if (!DIAG) {
                    *reinterpret_cast<h16x8*>(&Bb[r0 * LRG_LD + c8]) = rb[d][0];
                    *reinterpret_cast<h16x8*>(&Bb[(r0 + 32) * LRG_LD + c8]) = rb[d][1];
                }
                issue(ch + DEPTH, ra[d], rb[d]);
                lrg_lds_barrier();                                    // LDS hand-off only: the prefetched chunks stay in flight
                const h16* Bp = DIAG ? Ab : Bb;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const h16x8 av = *reinterpret_cast<const h16x8*>(&Ab[(si * 32 + li) * LRG_LD + kk * 16 + lh * 8]);
                    const h16x8 bv = *reinterpret_cast<const h16x8*>(&Bp[(sj * 32 + li) * LRG_LD + kk * 16 + lh * 8]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
                }
            }
        }
    }
    // C/D layout: column j = lane & 31, row i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) lrp_st(&Gk[(size_t)(i0 + si * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * lh) * NP + j0 + sj * 32 + li], acc[rg]);
    if (!DIAG) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            lrp_st16(&Gk[(size_t)(j0 + sj * 32 + li) * NP + i0 + si * 32 + 8 * q + 4 * lh], __builtin_bit_cast(lrg_u4, v));
        }
    }
}

template <int RP>
__global__ __launch_bounds__(256) void k_lrp(LrBatch b, LrpArgs a) {
    const int bid = blockIdx.x;
    int z, idx;
    if (a.zmod) { z = bid % a.batch; idx = bid / a.batch; } else { z = bid / a.nwg_t; idx = bid - z * a.nwg_t; }
    const LrItem it = b.it[z];
    const int N = a.N, C = a.C, NP = a.NP, r = a.r, KS = a.KS, nwg = a.nwg_t;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    unsigned* bar = a.tick + z * LRP_TW;
    extern __shared__ double lrp_smem[];
    char* sm = reinterpret_cast<char*>(lrp_smem);
    h16* D = (h16*)(it.ws + a.offD);
#define LSTAMP(k) do { if (a.stamps && tid == 0) a.stamps[(size_t)bid * 16 + (k)] = wall_clock64(); } while (0)
    LSTAMP(0);

    // ---------------- S0: D = x - base ----------------
    {
        // D has NP rows (rows >= N zero): every later tile load is unconditional, which is what lets the compiler count the loads in
        // flight (a guarded load is a branch, and behind a branch it waits for vmcnt(0): no prefetch)
        const size_t n8 = (size_t)N * C / 8, np8 = (size_t)NP * C / 8, stride = (size_t)nwg * 256;
        constexpr int PU = 4;                                         // 2 x 4 loads in flight per thread
        for (size_t i0 = (size_t)idx * 256 + tid; i0 < np8; i0 += stride * PU) {
            h16x8 xv[PU], bv[PU];
#pragma unroll
            for (int q = 0; q < PU; ++q) {
                const size_t i = i0 + stride * q;
                xv[q] = (h16x8)(h16)0; bv[q] = (h16x8)(h16)0;
                if (i < n8) {
                    xv[q] = __builtin_nontemporal_load(reinterpret_cast<const h16x8*>(it.x + i * 8));
                    if (it.base) bv[q] = lrg_ld8(it.base + i * 8);
                }
            }
#pragma unroll
            for (int q = 0; q < PU; ++q) {
                const size_t i = i0 + stride * q;
                h16x8 v = xv[q] - bv[q];                              // fp16, one rounding (torch eager: x - base); base absent: x - 0 = x
                if (a.absd) {
                    typedef unsigned short u16x8_ __attribute__((ext_vector_type(8)));
                    u16x8_ bb = __builtin_bit_cast(u16x8_, v);
                    bb &= (unsigned short)0x7fff;
                    v = __builtin_bit_cast(h16x8, bb);
                }
                if (i < np8) lrp_st16(D + i * 8, __builtin_bit_cast(lrg_u4, v));
            }
        }
    }
    LSTAMP(1);
    lrp_sync(bar, 1, nwg, a.err, a.timeout);
    LSTAMP(2);

    // ---------------- S1: Gram slabs and Y0 slabs ----------------
    {
        h16* As = reinterpret_cast<h16*>(sm);                          // [2][64 * LRG_LD]
        h16* Bs = As + 2 * 64 * LRG_LD;
        const int nitems = (a.npair + a.TN) * KS;
        for (int item = idx; item < nitems; item += nwg) {
            const int ks = item % KS, u = item / KS;
            const int k0 = ks * a.kslab, k1 = min(C, k0 + a.kslab);
            const int nch = (k1 - k0 + 63) / 64;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            __syncthreads();                                          // the previous item's LDS reads are over
            if (u < a.npair) {
                int ti = 0, rem = u;
                while (rem >= a.TN - ti) { rem -= a.TN - ti; ++ti; }
                const int tj = ti + rem;
                float* Gk = (float*)(it.ws + a.offG) + (size_t)ks * NP * NP;
                if (ti == tj) lrp_gram_pair<true>(D, C, NP, ti * 64, tj * 64, k0, nch, As, Bs, Gk);
                else lrp_gram_pair<false>(D, C, NP, ti * 64, tj * 64, k0, nch, As, Bs, Gk);
            } else {
                // Y0 slab of row tile ti: 64 rows x RP, Q0 as hi + lo fp16
                const int ti = u - a.npair, i0 = ti * 64;
                const int si = w & 1, part = w >> 1;                  // waves 0, 1: hi ; 2, 3: lo
                h16* A0 = As;
                h16* Qh = Bs;                                         // [32][LRG_LD]
                h16* Ql = Bs + 32 * LRG_LD;
                for (int i = tid; i < 64 * LRG_LD; i += 256) Bs[i] = (h16)0;
                constexpr int QV = 64 * RP / 256;
                h16x8 ra[2];
                float rq[QV];
                auto load = [&](int c0) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int p = tid + 256 * q, row = p >> 3, c8 = (p & 7) * 8;
                        ra[q] = lrg_ld8(D + (size_t)(i0 + row) * C + c0 + c8);
                    }
#pragma unroll
                    for (int q = 0; q < QV; ++q) rq[q] = it.q0[(size_t)c0 * RP + tid + 256 * q];
                };
                load(k0);
                for (int c0 = k0; c0 < k1; c0 += 64) {
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int p = tid + 256 * q, row = p >> 3, c8 = (p & 7) * 8;
                        *reinterpret_cast<h16x8*>(&A0[row * LRG_LD + c8]) = ra[q];
                    }
#pragma unroll
                    for (int q = 0; q < QV; ++q) {
                        const int e = tid + 256 * q, kk = e / RP, n = e - kk * RP;
                        const h16 hi = (h16)rq[q];
                        Qh[n * LRG_LD + kk] = hi;
                        Ql[n * LRG_LD + kk] = (h16)(rq[q] - (float)hi);
                    }
                    __syncthreads();
                    if (c0 + 64 < k1) load(c0 + 64);
                    const h16* Qp = part ? Ql : Qh;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const h16x8 av = *reinterpret_cast<const h16x8*>(&A0[(si * 32 + li) * LRG_LD + kk * 16 + lh * 8]);
                        const h16x8 bv = *reinterpret_cast<const h16x8*>(&Qp[li * LRG_LD + kk * 16 + lh * 8]);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
                    }
                }
                __syncthreads();
                float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(A0);
                if (part) {
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) red[si][(rg & 3) + 8 * (rg >> 2) + 4 * lh][li] = acc[rg];
                }
                __syncthreads();
                float* Yp = (float*)(it.ws + a.offY0p) + (size_t)ks * NP * RP;
                if (!part) {
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const int row = (rg & 3) + 8 * (rg >> 2) + 4 * lh;
                        if (li < RP) lrp_st(&Yp[(size_t)(i0 + si * 32 + row) * RP + li], acc[rg] + red[si][row][li]);      // hi + lo
                    }
                }
            }
        }
    }
    LSTAMP(3);
    lrp_sync(bar, 2, nwg, a.err, a.timeout);
    LSTAMP(4);

    // ---------------- S3 / S4: the two products with G, N x r sized ----------------
    double (*Gd)[RP + 1] = reinterpret_cast<double (*)[RP + 1]>(sm);
    double (*Ld)[RP + 1] = Gd + RP;
    double (*Sd)[RP + 1] = Ld + RP;
    double* misc = reinterpret_cast<double*>(Sd + RP);                // [0] gmax, [1 .. RP] dinv
    float* Ts = reinterpret_cast<float*>(misc + RP + 2);
    float* T2s = Ts + RP * RP;
    float* T23 = T2s + RP * RP;
    float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(sm + LrpLds<RP>::head);
    float* Ys = reinterpret_cast<float*>(sm + LrpLds<RP>::head + LrpLds<RP>::red);      // NP x RP
    float* W1 = (float*)(it.ws + a.offW1);
    float* W2 = (float*)(it.ws + a.offW2);
    const int QW = NP >> 2, nq = QW >> 3;
    const bool single = nwg >= a.nt;                                  // one row tile per workgroup: its G rows stay in registers
    float4 gq[LRG_GQ];
    auto load_gq = [&](int t) {
        // one round of LRG_GQ unconditional 16-byte loads per column slab (clamped q, masked value): a guarded load is a branch, and the
        // loads behind a branch are waited for one by one (measured: 11 us for these 140 KB instead of 2)
        const float* Gr = (const float*)(it.ws + a.offG) + (size_t)(t * 32 + li) * NP + w * QW + 4 * lh;
#pragma unroll
        for (int q = 0; q < LRG_GQ; ++q) gq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ks = 0; ks < KS; ++ks) {                             // fixed order
            float4 pq[LRG_GQ];
#pragma unroll
            for (int q = 0; q < LRG_GQ; ++q) pq[q] = *reinterpret_cast<const float4*>(Gr + (size_t)ks * NP * NP + 8 * min(q, nq - 1));
#pragma unroll
            for (int q = 0; q < LRG_GQ; ++q) {
                const float m = q < nq ? 1.f : 0.f;
                gq[q].x += m * pq[q].x; gq[q].y += m * pq[q].y; gq[q].z += m * pq[q].z; gq[q].w += m * pq[q].w;
            }
        }
    };
    auto product = [&](int t, float* Wout) {                          // W tile t = G rows x Ys; leaves the tile in red[0]
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int q = 0; q < LRG_GQ; ++q) {
            if (q < nq) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int col = w * QW + 8 * q + 4 * lh + e;
                    const float av = e == 0 ? gq[q].x : (e == 1 ? gq[q].y : (e == 2 ? gq[q].z : gq[q].w));
                    const float bv = (li < RP) ? Ys[col * RP + li] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) red[w][(rg & 3) + 8 * (rg >> 2) + 4 * lh][li] = acc[rg];
        __syncthreads();
        for (int i = tid; i < 32 * RP; i += 256) {
            const int row = i / RP, n = i - row * RP;
            const float s = ((red[0][row][n] + red[1][row][n]) + red[2][row][n]) + red[3][row][n];
            red[0][row][n] = s;
            lrp_st(&Wout[(size_t)(t * 32 + row) * RP + n], s);
        }
        __syncthreads();
    };
    if (idx < a.nt) {
        if (single) load_gq(idx);
        for (int i = tid; i < NP * RP; i += 256) {
            const float* Yp = (const float*)(it.ws + a.offY0p) + i;
            float pq[LRP_MAX_KS];
#pragma unroll
            for (int ks = 0; ks < LRP_MAX_KS; ++ks) pq[ks] = Yp[(size_t)min(ks, KS - 1) * NP * RP];
            float s = pq[0];
#pragma unroll
            for (int ks = 1; ks < LRP_MAX_KS; ++ks) s += (ks < KS) ? pq[ks] : 0.f;
            Ys[i] = (i / RP < N) ? s : 0.f;
        }
        __syncthreads();
        LSTAMP(15);
        for (int t = idx; t < a.nt; t += nwg) {
            if (!single) load_gq(t);
            product(t, W1);
            double* Mp = (double*)(it.ws + a.offMp) + (size_t)t * RP * RP;
            for (int i = tid; i < RP * RP; i += 256) {
                const int p = i / RP, q = i - p * RP;
                double m = 0.0;
                for (int row = 0; row < 32; ++row) m += (double)Ys[(t * 32 + row) * RP + p] * (double)red[0][row][q];
                lrp_st(&Mp[i], m);
            }
            __syncthreads();
        }
    }
    LSTAMP(5);
    lrp_sync(bar, 3, nwg, a.err, a.timeout);
    LSTAMP(6);
    if (idx < a.nt) {
        const double* Mp = (const double*)(it.ws + a.offMp);
        for (int i = tid; i < RP * RP; i += 256) {
            double mv[LRG_GQ];
#pragma unroll
            for (int t = 0; t < LRG_GQ; ++t) mv[t] = Mp[(size_t)min(t, a.nt - 1) * RP * RP + i];     // unconditional: one round trip
            double m = 0.0;
#pragma unroll
            for (int t = 0; t < LRG_GQ; ++t) m += (t < a.nt) ? mv[t] : 0.0;      // fixed order
            Gd[i / RP][i % RP] = m;
        }
        __syncthreads();
        lr_chol_T<RP, 256>(Gd, Ld, r, Ts, &misc[0], &misc[1], LRG_PIVOT_TOL);       // T1
        __syncthreads();
        for (int j = tid; j < NP; j += 256) {                         // Y1 = W1 T1
            float in[RP], out[RP];
#pragma unroll
            for (int k = 0; k < RP; ++k) in[k] = (j < N) ? W1[(size_t)j * RP + k] : 0.f;
#pragma unroll
            for (int n = 0; n < RP; ++n) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < RP; ++k) s = fmaf(in[k], Ts[k * RP + n], s);
                out[n] = s;
            }
#pragma unroll
            for (int n = 0; n < RP; ++n) Ys[j * RP + n] = out[n];
        }
        __syncthreads();
        for (int t = idx; t < a.nt; t += nwg) {
            if (!single) load_gq(t);
            product(t, W2);
            double* Mp2 = (double*)(it.ws + a.offMp2) + (size_t)t * RP * RP;
            double* Pp = (double*)(it.ws + a.offPp) + (size_t)t * RP * RP;
            for (int i = tid; i < RP * RP; i += 256) {
                const int p = i / RP, q = i - p * RP;
                double m = 0.0, pp = 0.0;
                for (int row = 0; row < 32; ++row) {
                    const double wv = (double)red[0][row][q];
                    m += (double)Ys[(t * 32 + row) * RP + p] * wv;
                    pp += (double)red[0][row][p] * wv;
                }
                lrp_st(&Mp2[i], m);
                lrp_st(&Pp[i], pp);
            }
            __syncthreads();
        }
    }
    LSTAMP(7);
    lrp_sync(bar, 4, nwg, a.err, a.timeout);
    LSTAMP(8);

    // ---------------- S5: T2, T3, U (every workgroup), then V and the state update per column block ----------------
    {
        const double* Mp2 = (const double*)(it.ws + a.offMp2);
        const double* Pp = (const double*)(it.ws + a.offPp);
        for (int i = tid; i < RP * RP; i += 256) {
            double mv[LRG_GQ], pv[LRG_GQ];
#pragma unroll
            for (int t = 0; t < LRG_GQ; ++t) {
                mv[t] = Mp2[(size_t)min(t, a.nt - 1) * RP * RP + i];
                pv[t] = Pp[(size_t)min(t, a.nt - 1) * RP * RP + i];
            }
            double m = 0.0, pp = 0.0;
#pragma unroll
            for (int t = 0; t < LRG_GQ; ++t) { m += (t < a.nt) ? mv[t] : 0.0; pp += (t < a.nt) ? pv[t] : 0.0; }
            Gd[i / RP][i % RP] = m;
            Sd[i / RP][i % RP] = pp;
        }
        __syncthreads();
        LSTAMP(12);
        lr_chol_T<RP, 256>(Gd, Ld, r, T2s, &misc[0], &misc[1], LRG_PIVOT_TOL);
        __syncthreads();
        LSTAMP(13);
        for (int i = tid; i < RP * RP; i += 256) {                    // X = S T2
            const int p = i / RP, q = i - p * RP;
            double s = 0.0;
            for (int k = 0; k < RP; ++k) s += 0.5 * (Sd[p][k] + Sd[k][p]) * (double)T2s[k * RP + q];
            Ld[p][q] = s;
        }
        __syncthreads();
        for (int i = tid; i < RP * RP; i += 256) {                    // M3 = T2^T X
            const int p = i / RP, q = i - p * RP;
            double s = 0.0;
            for (int k = 0; k < RP; ++k) s += (double)T2s[k * RP + p] * Ld[k][q];
            Gd[p][q] = s;
        }
        __syncthreads();
        LSTAMP(14);
        lr_chol_T<RP, 256>(Gd, Ld, r, Ts, &misc[0], &misc[1], LRG_PIVOT_TOL);          // T3
        __syncthreads();
        for (int i = tid; i < RP * RP; i += 256) {
            const int p = i / RP, q = i - p * RP;
            float s = 0.f;
            for (int k = 0; k < RP; ++k) s = fmaf(T2s[p * RP + k], Ts[k * RP + q], s);
            T23[i] = s;
        }
        __syncthreads();
    }
    LSTAMP(9);                                                        // factorisations done
    const int UST = NP + 8;                                           // halves per row of the transposed U halves
    h16* Uh = reinterpret_cast<h16*>(sm + LrpLds<RP>::head);          // [RP][UST]: hi
    h16* Ul = Uh + RP * UST;                                          //            lo
    h16* U16s = Ul + RP * UST;                                        // [NP][RP] row-major (the fp16 U of the packet)
    h16* Dt = U16s + NP * RP;                                         // [64][LRG_LD]: [column][row] of a 64-row chunk
    h16* v16s = Dt + 64 * LRG_LD;                                     // [RP][64]
    {
        h16* U16g = a.u_in_packet ? (h16*)it.packet : (h16*)(it.ws + a.offU16);
        for (int j = tid; j < NP; j += 256) {
            float in[RP], out[RP];
#pragma unroll
            for (int k = 0; k < RP; ++k) in[k] = (j < N) ? W2[(size_t)j * RP + k] : 0.f;
#pragma unroll
            for (int n = 0; n < RP; ++n) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < RP; ++k) s = fmaf(in[k], T23[k * RP + n], s);
                out[n] = s;
            }
#pragma unroll
            for (int n = 0; n < RP; ++n) {
                const h16 hi = (h16)out[n];
                Uh[n * UST + j] = hi;
                Ul[n * UST + j] = (h16)(out[n] - (float)hi);
                U16s[j * RP + n] = hi;
                if (idx == 0 && j < N && n < r) U16g[(size_t)j * r + n] = hi;
            }
        }
        for (int i = tid; i < RP * 8; i += 256) { Uh[(i >> 3) * UST + NP + (i & 7)] = (h16)0; Ul[(i >> 3) * UST + NP + (i & 7)] = (h16)0; }
    }
    LSTAMP(10);                                                       // U in LDS
    {
        const int nblk = (C + 63) / 64;
        const int sj = w & 1, part = w >> 1;
        const h16* Up = part ? Ul : Uh;
        const int nch = NP / 64;
        for (int blk = idx; blk < nblk; blk += nwg) {
            const int c0 = blk * 64;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            constexpr int VD = 3;                                     // 64-row chunks in flight
            h16x8 rd[VD][2];
            auto issue = [&](int ch, h16x8 (&x)[2]) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int p = tid + 256 * q, row = min(ch, nch - 1) * 64 + (p >> 3), c8 = (p & 7) * 8;      // C % 64 == 0, D has NP rows
                    x[q] = lrg_ld8(D + (size_t)row * C + c0 + c8);
                }
            };
#pragma unroll
            for (int d = 0; d < VD; ++d) issue(d, rd[d]);
            for (int cb = 0; cb < nch; cb += VD) {
#pragma unroll
                for (int d = 0; d < VD; ++d) {
                    const int ch = cb + d;
                    if (ch < nch) {
                        lrg_lds_barrier();                            // U is complete (first pass) / the previous chunk's reads are over
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int p = tid + 256 * q, row = p >> 3, c8 = (p & 7) * 8;
#pragma unroll
                            for (int e = 0; e < 8; ++e) Dt[(c8 + e) * LRG_LD + row] = rd[d][q][e];
                        }
                        issue(ch + VD, rd[d]);
                        lrg_lds_barrier();
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            h16x8 av = (h16x8)(h16)0;
                            if (li < RP) av = *reinterpret_cast<const h16x8*>(&Up[li * UST + ch * 64 + kk * 16 + lh * 8]);      // A[m = li][k = row]
                            const h16x8 bv = *reinterpret_cast<const h16x8*>(&Dt[(sj * 32 + li) * LRG_LD + kk * 16 + lh * 8]); // B[k = row][n = column]
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();
            float (*red2)[32][33] = reinterpret_cast<float (*)[32][33]>(Dt);     // 2 x 32 x 33 floats = 8448 B <= 9216 B
            if (part) {
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) red2[sj][(rg & 3) + 8 * (rg >> 2) + 4 * lh][li] = acc[rg];
            }
            __syncthreads();
            if (!part) {
                const int c = c0 + sj * 32 + li;
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int m = (rg & 3) + 8 * (rg >> 2) + 4 * lh;  // rank index
                    if (m < RP) {
                        const h16 v = (m < r && c < C) ? (h16)(acc[rg] + red2[sj][m][li]) : (h16)0;
                        v16s[m * 64 + sj * 32 + li] = v;
                        if (m < r && c < C) {
                            if (a.u_in_packet) ((h16*)it.packet)[(size_t)N * r + (size_t)m * C + c] = v;
                            else ((h16*)(it.ws + a.offV16))[(size_t)c * r + m] = v;
                        }
                    }
                }
            }
            if (a.fuse_decode) {
                __syncthreads();
                // new_base[:, block] = base + fp16(U V): the arithmetic of k_lr_decode (v_dot2 chain over the k-pairs in order, one rounding to
                // fp16, one fp16 add).  Thread: 8 columns (cg), rows slot, slot + 32, ...
                const int cg = tid & 7, slot = tid >> 3;
                const int c = c0 + cg * 8;
                if (c < C) {
                    h16x2 vp[RP / 2][8];
#pragma unroll
                    for (int kk = 0; kk < RP / 2; ++kk)
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            vp[kk][i][0] = v16s[(2 * kk) * 64 + cg * 8 + i];
                            vp[kk][i][1] = v16s[(2 * kk + 1) * 64 + cg * 8 + i];
                        }
                    for (int n0 = slot; n0 < N; n0 += 64) {           // two rows in flight
                        const int n1 = n0 + 32;
                        const bool h1 = n1 < N;
                        h16x8 ba = (h16x8)(h16)0, bb = (h16x8)(h16)0;
                        if (it.base) {
                            ba = __builtin_nontemporal_load(reinterpret_cast<const h16x8*>(it.base + (size_t)n0 * C + c));
                            if (h1) bb = __builtin_nontemporal_load(reinterpret_cast<const h16x8*>(it.base + (size_t)n1 * C + c));
                        }
                        float acca[8], accb[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) { acca[i] = 0.f; accb[i] = 0.f; }
#pragma unroll
                        for (int kk = 0; kk < RP / 2; ++kk) {
                            if (2 * kk < r) {
                                const h16x2 ua = *reinterpret_cast<const h16x2*>(&U16s[n0 * RP + 2 * kk]);
                                const h16x2 ub = *reinterpret_cast<const h16x2*>(&U16s[(h1 ? n1 : n0) * RP + 2 * kk]);
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    acca[i] = __builtin_amdgcn_fdot2(ua, vp[kk][i], acca[i], false);
                                    accb[i] = __builtin_amdgcn_fdot2(ub, vp[kk][i], accb[i], false);
                                }
                            }
                        }
                        h16x8 oa, ob;
#pragma unroll
                        for (int i = 0; i < 8; ++i) { oa[i] = (h16)acca[i]; ob[i] = (h16)accb[i]; }
                        if (it.base) { oa = ba + oa; ob = bb + ob; }
                        __builtin_nontemporal_store(oa, reinterpret_cast<h16x8*>(it.new_base + (size_t)n0 * C + c));
                        if (h1) __builtin_nontemporal_store(ob, reinterpret_cast<h16x8*>(it.new_base + (size_t)n1 * C + c));
                    }
                }
            }
        }
    }
    // ---------------- leave: the last workgroup of the tensor resets the barrier words ----------------
    LSTAMP(11);
#undef LSTAMP
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)nwg - 1) {
            __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(bar + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int x = 0; x < 8; ++x) __hip_atomic_store(bar + 32 + 16 * x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static inline int lrg_np(int N) { return (N + 63) / 64 * 64; }

// shapes the N-space chain covers: the Gram matrix must stay small (N <= 640: 1.6 MB per slab) and the column slabs whole chunks
bool cfx_i_lrg_ok(int N, int C) { return N >= 32 && N <= 576 && (C % 128) == 0 && C >= 128; }
// column slabs per tile pair: as many as divide the columns into whole 64-column chunks, up to 8 (more, shorter workgroups: the
// kernel is bound by load latency, not by bytes)
static inline int lrg_ks(int C) { return (C % 512) == 0 ? 8 : ((C % 256) == 0 ? 4 : 2); }

// bytes the chain needs behind the C-space chain's own per-tensor layout (which provides D, U16, V16)
static size_t lrg5_extra_bytes(int N, int C, int RP) {      // the multi-launch form (CFX_LR_CHAIN=gram5)
    const size_t NP = lrg_np(N), nt = (N + 31) / 32;
    const size_t KS = lrg_ks(C), TN = NP / 64, npair = TN * (TN + 1) / 2;
    size_t o = 0;
    o += al256(NP * NP * 4);              // G
    o += al256(npair * KS * 4096 * 4);    // G partial tiles, one per column slab
    o += al256(NP * RP * 4);              // Y0
    o += al256(TN * KS * 64 * RP * 4);    // Y0 partial rows
    o += al256(NP * RP * 4) * 2;          // W1, W2
    o += al256(nt * RP * RP * 8) * 2;     // M partials, W^T W partials
    o += al256(3 * RP * RP * 4);          // T1, T2, T3
    o += al256(NP * RP * 4);              // U fp32
    return o;
}
static size_t lrp_extra_bytes(int N, int C, int RP) {       // the single-launch form
    const size_t NP = lrg_np(N), nt = (N + 31) / 32;
    size_t o = 0;
    o += al256(LRP_MAX_KS * NP * NP * 4); // G, one matrix per column slab
    o += al256(LRP_MAX_KS * NP * RP * 4); // Y0, one per column slab
    o += al256(NP * RP * 4) * 2;          // W1, W2
    o += al256(nt * RP * RP * 8) * 3;     // partial M1 | M2 | W2^T W2
    return o;
}
size_t cfx_i_lrg_extra_bytes(int N, int C, int RP) {
    if (!cfx_i_lrg_ok(N, C)) return 0;
    const size_t a = lrg5_extra_bytes(N, C, RP), b = lrp_extra_bytes(N, C, RP), c = cfx_i_lrs_extra_bytes(N, C, RP);
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

template <int RP>
static int lrg_run(cfx_ctx* ctx, const LrBatch& b, LrgArgs a, hipStream_t s) {
    const int N = a.N, C = a.C;
    const size_t n8 = (size_t)N * C / 8;
    LAUNCH(ctx, KID_LR_PREP, s, k_lrg_prep, dim3((unsigned)((n8 + 255) / 256 < 1024 ? (n8 + 255) / 256 : 1024), a.batch), dim3(256), 0, s, b, n8, a.offD, a.absd);
    auto grid_of = [&](int per_tensor) {
        a.per_tensor = per_tensor;
        if (a.xcd_group) return dim3((unsigned)((per_tensor + a.xcd_group - 1) / a.xcd_group * 8));
        return dim3((unsigned)(per_tensor * a.batch));
    };
    dim3 g = grid_of((a.npair + a.TN) * a.KS);
    LAUNCH(ctx, KID_LR_AQ, s, (k_lrg_gram<RP>), g, dim3(256), 0, s, b, a);
    const int nt = (N + 31) / 32;
    const size_t lds = (size_t)3 * RP * (RP + 1) * 8 + (RP + 2) * 8 + (size_t)3 * RP * RP * 4 + (size_t)4 * 32 * 33 * 4 + (size_t)a.NP * RP * 4;
    g = grid_of(nt);
    static size_t attr_bytes[3] = {0, 0, 0};                          // dynamic LDS the two kernels have been allowed so far
    const int ai = RP == 8 ? 0 : (RP == 16 ? 1 : 2);
    if (lds > 64 * 1024 && lds > attr_bytes[ai]) {
        if (hipFuncSetAttribute((const void*)k_lrg_gy<RP, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_lrg_gy<RP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return fail(ctx, CFX_ERR_LAUNCH, "low-rank: the device does not grant the LDS the N-space chain needs");
        }
        attr_bytes[ai] = lds;
    }
    LAUNCH(ctx, KID_LR_ATY, s, (k_lrg_gy<RP, 0>), g, dim3(256), lds, s, b, a);
    LAUNCH(ctx, KID_LR_CHOL, s, (k_lrg_gy<RP, 1>), g, dim3(256), lds, s, b, a);
    g = grid_of((C + 63) / 64);
    LAUNCH(ctx, KID_LR_APPLY, s, (k_lrg_v<RP>), g, dim3(256), 0, s, b, a);
    return check_launch(ctx, "low-rank (N-space chain) launch");
}

template <int RP>
static int lrp_run(cfx_ctx* ctx, const LrBatch& b, LrpArgs a, hipStream_t s) {
    const size_t lds = (size_t)LrpLds<RP>::total(a.NP);
    static size_t attr_bytes = 0, occ_lds = 0;
    static int per_cu = 0;
    if (lds > 64 * 1024 && lds > attr_bytes) {
        if (hipFuncSetAttribute((const void*)k_lrp<RP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return fail(ctx, CFX_ERR_LAUNCH, "low-rank: the device does not grant the LDS the single-launch chain needs");
        }
        attr_bytes = lds;
    }
    if (!per_cu || occ_lds != lds) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lrp<RP>, 256, lds) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            per_cu = 1;
        }
        occ_lds = lds;
    }
    // every workgroup of the launch must be resident at once (they wait for each other): at most occupancy x the CUs this stream's
    // queue may use, split evenly over the tensors
    static const char* wg_env = getenv("CFX_LRP_WGS");
    const int cap = per_cu * cfx_i_stream_cus(ctx, (void*)s);
    int want = wg_env ? atoi(wg_env) : 128;
    if (want < 1) want = 1;
    int nwg_t = cap / a.batch;
    if (nwg_t > want) nwg_t = want;
    if (nwg_t < 1) return fail(ctx, CFX_ERR_LAUNCH, "low-rank: the stream has fewer CUs than the batch has tensors");
    a.nwg_t = nwg_t;
    // column slabs: the split with the fewest (rounds of tile-pair items per workgroup) / (slabs), ties to fewer slabs
    static const char* ks_env = getenv("CFX_LRP_KS");
    int best = 1;
    double best_cost = 1e30;
    for (int ks = 1; ks <= LRP_MAX_KS; ks *= 2) {
        if (a.C % (64 * ks)) continue;
        const int items = (a.npair + a.TN) * ks;
        const double cost = (double)((items + nwg_t - 1) / nwg_t) / ks;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = ks; }
    }
    if (ks_env && atoi(ks_env) >= 1 && atoi(ks_env) <= LRP_MAX_KS && a.C % (64 * atoi(ks_env)) == 0) best = atoi(ks_env);
    a.KS = best;
    a.kslab = a.C / a.KS;
    a.zmod = (a.batch <= 8 && 8 % a.batch == 0) ? 1 : 0;
    LAUNCH(ctx, KID_LR_CHAIN, s, (k_lrp<RP>), dim3((unsigned)(nwg_t * a.batch)), dim3(256), lds, s, b, a);
    return check_launch(ctx, "low-rank (single-launch chain)");
}

// Factors of every tensor of the batch: LOW_RANK -> U, V straight into the packets; LOW_RANK_Q -> fp16 U (N x r) at offU16 and V^T
// (C x r) at offV16 of each tensor's workspace (what the int4 factor quantiser of cfx_lowrank.hip takes).  `extra` = offset of
// cfx_i_lrg_extra_bytes() bytes inside each tensor's workspace.  want_decode: also new_base = base + fp16(U V) (LOW_RANK with error
// feedback); *decoded says whether the launch did it (the single-launch form does, the multi-launch form leaves it to the caller).
int cfx_i_lrg_factors(cfx_ctx* ctx, int quantized, int N, int C, int rank, int batch, const LrBatch& b, size_t offD, size_t offU16, size_t offV16,
                      size_t extra, int absd, int want_decode, int* decoded, hipStream_t s) {
    const int RPv = lr_rp(rank);
    if (decoded) *decoded = 0;
    static const char* chain_env = getenv("CFX_LR_CHAIN");
    // the slab-resident chain when its workgroups are co-resident on this stream's CUs (not on a CU-masked lane)
    if (!(chain_env && (!strcmp(chain_env, "gram5") || !strcmp(chain_env, "gram1"))) && cfx_i_lrs_fit(ctx, N, C, RPv, (void*)s) >= 1)
        return cfx_i_lrs_factors(ctx, quantized, N, C, rank, batch, b, offU16, offV16, extra, absd, want_decode, decoded, s);
    if (!(chain_env && !strcmp(chain_env, "gram5"))) {
        const size_t NP = lrg_np(N), nt = (N + 31) / 32;
        for (int first = 0; first < batch; first += 4) {               // LRP_TW words of the launch's ticket block per tensor
            const int nb = batch - first < 4 ? batch - first : 4;
            LrBatch bb;
            memset(&bb, 0, sizeof(bb));
            for (int i = 0; i < nb; ++i) bb.it[i] = b.it[first + i];
            LrpArgs a;
            memset(&a, 0, sizeof(a));
            a.N = N; a.C = C; a.NP = (int)NP; a.TN = (int)(NP / 64); a.npair = a.TN * (a.TN + 1) / 2; a.r = rank; a.batch = nb; a.nt = (int)nt;
            a.absd = absd; a.u_in_packet = quantized ? 0 : 1;
            a.fuse_decode = (want_decode && !quantized) ? 1 : 0;
            size_t o = extra;
            a.offD = offD; a.offU16 = offU16; a.offV16 = offV16;
            a.offG = o;   o += al256(LRP_MAX_KS * NP * NP * 4);
            a.offY0p = o; o += al256(LRP_MAX_KS * NP * RPv * 4);
            a.offW1 = o;  o += al256(NP * RPv * 4);
            a.offW2 = o;  o += al256(NP * RPv * 4);
            a.offMp = o;  o += al256(nt * RPv * RPv * 8);
            a.offMp2 = o; o += al256(nt * RPv * RPv * 8);
            a.offPp = o;  o += al256(nt * RPv * RPv * 8);
            a.tick = cfx_i_ticket_block(ctx, (void*)s);
            if (!a.tick) return fail(ctx, CFX_ERR_LAUNCH, "low-rank: no ticket block");
            a.err = ctx->gate_err;
            a.timeout = ctx->gate_timeout;
            a.stamps = (unsigned long long*)ctx->dbg_stamps;
            const int rc = RPv == 8 ? lrp_run<8>(ctx, bb, a, s) : lrp_run<16>(ctx, bb, a, s);
            if (rc != CFX_OK) return rc;
        }
        if (decoded) *decoded = (want_decode && !quantized) ? 1 : 0;
        return CFX_OK;
    }
    LrgArgs a;
    memset(&a, 0, sizeof(a));
    a.N = N; a.C = C; a.NP = lrg_np(N); a.TN = a.NP / 64; a.npair = a.TN * (a.TN + 1) / 2; a.r = rank; a.batch = batch;
    a.KS = lrg_ks(C);
    a.kslab = C / a.KS;
    a.xcd_group = (batch <= 8 && 8 % batch == 0) ? 8 / batch : 0;
    const size_t NP = a.NP, nt = (N + 31) / 32, KS = a.KS;
    size_t o = extra;
    a.offD = offD; a.offU16 = offU16; a.offV16 = offV16;
    a.offG = o;   o += al256(NP * NP * 4);
    a.offGp = o;  o += al256((size_t)a.npair * KS * 4096 * 4);
    a.offY0 = o;  o += al256(NP * RPv * 4);
    a.offY0p = o; o += al256((size_t)a.TN * KS * 64 * RPv * 4);
    a.offW1 = o;  o += al256(NP * RPv * 4);
    a.offW2 = o;  o += al256(NP * RPv * 4);
    a.offMp = o;  o += al256(nt * RPv * RPv * 8);
    a.offPp = o;  o += al256(nt * RPv * RPv * 8);
    a.offT = o;   o += al256(3 * (size_t)RPv * RPv * 4);
    a.offUf = o;  o += al256(NP * RPv * 4);
    a.u_in_packet = quantized ? 0 : 1;
    a.absd = absd;
    a.tick = cfx_i_ticket_block(ctx, (void*)s);
    if (!a.tick) return fail(ctx, CFX_ERR_LAUNCH, "low-rank: no ticket block");
    if (RPv == 8) return lrg_run<8>(ctx, b, a, s);
    if (RPv == 16) return lrg_run<16>(ctx, b, a, s);
    return lrg_run<32>(ctx, b, a, s);
}

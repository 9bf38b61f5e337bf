// Low-rank residual codecs, N-space ("Gram") form of the factorisation chain - part of libcfx.so.
//
// The reference's subspace_iter (xfuser/compact/compress_lowrank.py:14-61) is, for A = x - base (N x C, N << C):
//     Q <- orth(A^T (A Q))  twice ;  U = orth(A Q) ;  V = U^T A
// The C-space chain of cfx_lowrank.hip follows it product by product: 8-10 launches (13 when this file was written), each 5-35 us for
// a few microseconds of traffic.  Here the same iteration runs in N-space.  With G = A A^T (N x N) and Y0 = A Q0:
//     W1 = G Y0        M1 = Y0^T W1 (= Z1^T Z1, Z1 = A^T Y0)     T1 = chol(M1)^-T     Y1 = W1 T1   (= A orth(Z1))
//     W2 = G Y1        M2 = Y1^T W2 (= Z2^T Z2)                  T2 = chol(M2)^-T     Y2 = W2 T2   (= A orth(Z2))
//     M3 = Y2^T Y2 = T2^T (W2^T W2) T2                           T3 = chol(M3)^-T     U  = Y2 T3   (= orth(A Q2))
//     V  = U^T A
// - identical in exact arithmetic, orthonormalisation after every multiplication included (without it the result drifts from the
// reference by up to 3e-3 on fast-decaying spectra; with it 1e-5 .. 6e-5, measured in numpy).  A is touched by three passes
// instead of seven, two of them pure fp16-input MFMA work with no range issue (A is fp16; Q0 and U are split hi + lo), and
// everything between is N x r sized.  Launches (r x r factorisations run in the LAST-ARRIVING workgroup of the launch that
// produced their input: ticket counter, write-through partials - the in-launch finalize of cfx_absmean.hip):
//     k_lrg_prep   D = x - base                                                      (materialised once, 3.3 MB at the FLUX shard)
//     k_lrg_gram   G = D D^T (64 x 64 tiles, upper triangle + mirror, 2 column slabs), Y0 = D Q0       v_mfma_f32_32x32x16_f16
//     k_lrg_gy<0>  W1 = G Y0, M1 partials; last arriver: T1                                           v_mfma_f32_32x32x2_f32
//     k_lrg_gy<1>  Y1 = W1 T1 (every workgroup, N x r x r), W2 = G Y1, M2 and W2^T W2 partials; last arriver: T2, T3, U
//     k_lrg_v      V = U^T D                                                                           v_mfma_f32_32x32x16_f16
//     (decode / int4 factor quantisation: the kernels the C-space chain ends with)
// All reductions have a fixed order (no float atomics): results are reproducible run to run.
//
// Since round 3 this six-launch form is the FALLBACK: where C / 32 workgroups per tensor are co-resident on the stream's CUs the
// slab-resident chain of cfx_lrslab.hip runs instead (one persistent launch, neither A nor A A^T ever written: 41 / 53 us per K,V
// pair at the FLUX shard against 90 / 128 here); this one remains for CU-masked lanes and for CFX_LR_CHAIN=gram5.
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
static size_t lrg5_extra_bytes(int N, int C, int RP) {
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
size_t cfx_i_lrg_extra_bytes(int N, int C, int RP) {
    if (!cfx_i_lrg_ok(N, C)) return 0;
    return lrg5_extra_bytes(N, C, RP);      // (the slab-resident chain keeps what it hands over in the context's arena)
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

// Factors of every tensor of the batch: LOW_RANK -> U, V straight into the packets; LOW_RANK_Q -> fp16 U (N x r) at offU16 and V^T
// (C x r) at offV16 of each tensor's workspace (what the int4 factor quantiser of cfx_lowrank.hip takes).  `extra` = offset of
// cfx_i_lrg_extra_bytes() bytes inside each tensor's workspace.  want_decode: also new_base = base + fp16(U V) (LOW_RANK with error
// feedback); *decoded says whether the launch did it (the single-launch form does, the multi-launch form leaves it to the caller).
int cfx_i_lrg_factors(cfx_ctx* ctx, int quantized, int N, int C, int rank, int batch, const LrBatch& b, size_t offD, size_t offU16, size_t offV16,
                      size_t extra, int absd, int want_decode, int* decoded, hipStream_t s) {
    const int RPv = lr_rp(rank);
    if (decoded) *decoded = 0;
    // the slab-resident chain (cfx_lrslab.hip: one persistent launch) when its workgroups are co-resident on this stream's CUs - not on
    // a CU-masked lane, whose 32 CUs cannot hold C / 32 workgroups: there the six launches below run
    if (ctx->lr_chain == 0 && cfx_i_lrs_fit(ctx, N, C, RPv, (void*)s) >= 1)
        return cfx_i_lrs_factors(ctx, quantized, N, C, rank, batch, b, offU16, offV16, extra, absd, want_decode, decoded, s);
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
    // rank 32 is only ever taken by the slab-resident launch (the six-launch form's two back-to-back factorisations spilled at RP = 32
    // and measured slower than the C-space chain: not built)
    return fail(ctx, CFX_ERR_SHAPE, "low-rank: the six-launch N-space chain is built for ranks up to 16");
}

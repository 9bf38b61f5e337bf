// Low-rank residual codecs (LOW_RANK, LOW_RANK_Q) on gfx950 - part of libcfx.so.
//
// Replaces the reference's `subspace_iter` (xfuser/compact/compress_lowrank.py:14-61: randomised subspace iteration with
// Householder QR, run through torch.compile + cuBLAS/cuSOLVER: ~10 library calls, A read 6 times) and the LOW_RANK /
// LOW_RANK_Q branches of slowpath.py (:54-75 encode, :120-131 + :151-164 decode) with a fixed chain of small kernels:
//
//   D  = x - base ; Y = D Q0                        k_lr_aq<FROMX>  (fp16 residual, formed on the fly and materialised once)
//   2x { Z = D^T Y ; Y = D orth(Z), orth(Z) = Z chol(Z^T Z)^-T }   k_lr_aty (+ partial Gram), k_lr_aq<ORTH> (factor + substitution + product)
//   Z' = D^T Y ; G = Q^T Z' (= Y^T Y)                 k_lr_aty (+ partial Q^T Z')
//   T  = chol(G)^-T ; U = Y T ; V = (Z' T)^T          k_lr_apply2                 (U = orth(Y), V = U^T D without re-reading D)
//   new_base = base + fp16(U16 V16)                   k_lr_decode   (the receiver's kernel, run on the sender's packet)
//
// orth() is Cholesky-QR with the r x r Gram matrix accumulated and factorised in fp64 (Z = Q R, R = chol(Z^T Z)^T): the
// subspace - hence U V, the projection of D onto it - is the one Householder QR gives; only the signs/rotations a QR leaves
// free can differ, and fp32 rounding.  The random start is NOT orthonormalised (compress_lowrank.py:41-42 does QR(randn)):
// span(D^T D Q0) does not depend on the basis chosen for span(Q0).
//
// This is the chain for shards of more than 576 rows (below that: the one-launch slab-resident chain, cfx_lrslab.hip; on CU-masked lanes
// the N-space chain, cfx_lrgram.hip).  Both products are v_mfma_f32_32x32x16_f16 with D as it is (fp16: exact) and the fp32 operand split
// into fp16 hi + lo under an exact power-of-two scale (an MFMA tile is 32 wide whatever the rank: fp32-input MFMA cost 8 us a pass at
// (4096, 1152)); 8 launches up to rank 16, 10 at rank 32.
// All reductions have a fixed order (no float atomics) that does not depend on the batch: results are reproducible run to run.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include "cfx.h"
#include "cfx_internal.h"
#include "cfx_lr.h"

// workspace carve-up (per tensor), all offsets 256-byte aligned
struct LrWs {
    size_t D, Qa, Zb, Y, Gp, T, U16, V16, Uq, Vq, Vsec, Zp, tick, Ymax, gram, total;
};
#define LR_NS_MAX 8            // row splits of Z = D^T Y (k_lr_aty): partial tiles of up to 8 workgroups, summed by the last to arrive
#define LR_NS_TARGET 768       //   ... as many as make ~768 workgroups (measured at (4096, 1152): 4 splits 98.5 us, 8: 92.2, 11: 98.8, 16: 100.6)
// Row splits of Z = D^T Y: enough workgroups to fill the machine (a column tile x all N is C / 32 workgroups a tensor - 72 for K,V of
// (4096, 1152), each walking 4096 rows: 50 us), at least two 128-row chunks each.
static int lr_aty_splits(int N, int C, int /*batch*/) {
    // (as if for a K,V pair whatever the batch: the order of a sum - hence the bits - must not depend on what else is in the launch)
    const int tiles = ((C + 31) / 32) * 2;
    int ns = (LR_NS_TARGET + tiles - 1) / tiles;
    ns = std::min(ns, std::min(LR_NS_MAX, N / 256));
    return std::max(ns, 1);
}
static LrWs lr_layout(int N, int C, int RP) {
    LrWs w;
    size_t o = 0;
    w.D = o;   o += al256((size_t)((N + 63) / 64 * 64) * C * 2);     // 64-row padded (the N-space chain reads whole tiles)
    w.Qa = o;  o += al256((size_t)C * RP * 4);
    w.Zb = o;  o += al256((size_t)C * RP * 4);
    w.Y = o;   o += al256((size_t)4 * N * RP * 4);        // up to 4 column-group partials of Y
    w.Gp = o;  o += al256((size_t)((C + 31) / 32) * RP * RP * 8);
    w.T = o;   o += al256((size_t)RP * RP * 4);
    w.U16 = o; o += al256((size_t)N * RP * 2);
    w.V16 = o; o += al256((size_t)C * RP * 2);
    w.Uq = o;  o += al256((size_t)N * RP * 2 + 256);
    w.Vq = o;  o += al256((size_t)C * RP * 2 + 256);
    w.Vsec = o; o += al256((size_t)C * RP / 2 + 4 * RP + 256);
    w.Zp = o;  o += al256((size_t)lr_aty_splits(N, C, 1) * C * RP * 4);     // (batch 1: the most splits a tensor can get)
    w.tick = o; o += al256((size_t)((C + 31) / 32) * 4);
    w.Ymax = o; o += al256((size_t)((N + 31) / 32) * 4 * 4);               // one float per workgroup of k_lr_aq
    w.gram = o; o += cfx_i_lrg_extra_bytes(N, C, RP);       // the N-space chain's Gram matrix, N x r intermediates (0 when it does not apply)
    w.total = o;
    return w;
}

// ---------------------------------------------------------------------------------------------------------------------
// Ypart[g] (N x RP) = D[:, cols of group g] . Q[cols of group g, :]      Y = sum of the group partials (summed, in fixed
// order, by whoever reads Y).  grid (ceil(N/32), g, batch), g = 1, 2 or 4 column groups: a workgroup owns 32 rows x every g-th 256-column chunk.
// v_mfma_f32_32x32x16_f16: A = D (fp16 as it is: exact), B = Q as fp16 hi + lo (two instructions; Q scaled by 16 so that the lo halves
// of an orthonormal basis's entries stay normal numbers - Q0 ~ randn and |Q| <= 1 afterwards are far from fp16's range), fp32 sums: 22
// bits of Q instead of the 24 an fp32-input MFMA keeps, at an eighth of its issue time - the pass was bound by v_mfma_f32_32x32x2_f32
// (2 N C 32 flops whatever the rank: 8 us of MFMA for K,V of (4096, 1152)).  The D and Q chunks go through LDS (D rows as they are, Q
// transposed: an operand is 8 consecutive k of one row / column); the next chunk's loads are in flight while this one is multiplied;
// each of the 4 waves takes 64 of the chunk's 256 columns and the 4 partial tiles are summed in fixed order.
// Also written: the largest |entry| of this workgroup's partial (Ymax, one slot a workgroup) - k_lr_aty scales Y into fp16's range by it.
// ---------------------------------------------------------------------------------------------------------------------
// FROMX (the first product of a chain): D = x - base is formed on the fly (fp16, one rounding, as torch eager) and written to the
//   workspace for the later passes - every element of D is read by exactly one workgroup here, so k_lr_prep is not needed.
// ORTH (the second and third product): the operand is Q = orth(Z) = Z chol(Z^T Z)^-T, formed HERE - every workgroup sums the column tiles'
//   partial Grams and factorises for itself (lr_factor_to_lds), then a thread turns the row of Z of its column of the chunk into a row of Q
//   by forward substitution on the way into LDS: no launch for it, no round trip of Q through memory.  The row tile 0 workgroups also write
//   Q out (k_lr_aty's Gram part of the last pass wants it).
#define LR_Q_SCALE 16.f
template <int RP> __device__ __forceinline__ void lr_factor_to_lds(const double* Gp, int nparts, int r, float* ts);
template <int RP, bool FROMX, bool ORTH>
__global__ __launch_bounds__(256) void k_lr_aq(LrBatch b, int N, int C, size_t offD, size_t offQ, size_t offY, int use_q0, int absd, size_t offTick,
                                               size_t offYmax, size_t offZ, size_t offG, int nparts, int r) {
    constexpr int CK = 256, LDD = CK + 8;            // 528-byte rows: 16-byte aligned operands
    constexpr int QI = (CK / 2) * (RP / 4) / 256;    // (column pair, 4 ranks) items of the Q chunk per thread (4 at RP = 32)
    const LrItem it = b.it[blockIdx.z];
    h16* D = (h16*)(it.ws + offD);
    const float* Q = use_q0 ? it.q0 : (const float*)(it.ws + offQ);
    const float* Zin = (const float*)(it.ws + offZ);
    float* Qout = (float*)(it.ws + offQ);
    const float qscale = use_q0 ? 1.f : LR_Q_SCALE;  // the caller's start matrix goes in as it is (randn: no small entries to protect; |q0| < 65504)
    __shared__ __attribute__((aligned(16))) float ts[ORTH ? RP * RP : 4];
    float* Y = (float*)(it.ws + offY) + (size_t)blockIdx.y * N * RP;
    __shared__ __attribute__((aligned(16))) h16 dsm[32 * LDD];                    // 16.5 KB
    __shared__ __attribute__((aligned(16))) h16 qsm[2 * 32 * LDD];                // Q^T hi | lo: [rank][column], 33 KB; reused for the wave partials
    float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(qsm);              // 4 x 32 x 33 floats = 16.5 KB
    __shared__ float wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int n0 = blockIdx.x * 32;
    if (FROMX && blockIdx.x == 0 && blockIdx.y == 0) {      // the chain's first launch: k_lr_aty's tickets start at zero
        unsigned* tick = (unsigned*)(it.ws + offTick);
        for (int i = tid; i < (C + 31) / 32; i += 256) tick[i] = 0u;
    }
    for (int i = tid; i < 2 * 32 * LDD / 8; i += 256) reinterpret_cast<h16x8*>(qsm)[i] = (h16x8)(h16)0;     // ranks >= RP: zero columns of B
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    h16x8 dreg[4];
    float4 qreg[ORTH ? 1 : QI][2];
    float4 zreg[ORTH ? RP / 4 : 1];                  // ORTH: the row of Z of this thread's column of the chunk
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {                // 32 rows x 32 sixteen-byte pieces
            const int i = tid + 256 * u, rr = i >> 5, pc = (i & 31) * 8;
            dreg[u] = (h16x8)(h16)0;
            if (n0 + rr < N && c0 + pc < C) {
                const size_t o = (size_t)(n0 + rr) * C + c0 + pc;
                if (FROMX) {
                    h16x8 xv = *reinterpret_cast<const h16x8*>(it.x + o);
                    if (it.base) xv = xv - *reinterpret_cast<const h16x8*>(it.base + o);
                    if (absd) {                          // the matrix to factorise is |x - base| (rank-K scales of the 1-bit codec)
                        typedef unsigned short u16x8_ __attribute__((ext_vector_type(8)));
                        u16x8_ bb = __builtin_bit_cast(u16x8_, xv);
                        bb &= (unsigned short)0x7fff;
                        xv = __builtin_bit_cast(h16x8, bb);
                    }
                    *reinterpret_cast<h16x8*>(D + o) = xv;
                    dreg[u] = xv;
                } else dreg[u] = *reinterpret_cast<const h16x8*>(D + o);
            }
        }
        if constexpr (ORTH) {
#pragma unroll
            for (int k4 = 0; k4 < RP / 4; ++k4) {
                zreg[k4] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c0 + tid < C) zreg[k4] = *reinterpret_cast<const float4*>(Zin + (size_t)(c0 + tid) * RP + 4 * k4);
            }
        } else {
#pragma unroll
            for (int u = 0; u < QI; ++u) {           // columns 2 cp, 2 cp + 1 (C is even), ranks 4 k4 .. + 3
                const int i = tid + 256 * u, cp = i / (RP / 4), k4 = i - cp * (RP / 4);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    qreg[u][h] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (c0 + 2 * cp + h < C) qreg[u][h] = *reinterpret_cast<const float4*>(Q + (size_t)(c0 + 2 * cp + h) * RP + 4 * k4);
                }
            }
        }
    };
    const int cstep = gridDim.y * CK;                // (1, 2 or 4 column groups: the host's choice - enough workgroups, as few partials as that allows)
    int c0 = blockIdx.y * CK;
    if (c0 < C) load_chunk(c0);
    if constexpr (ORTH) lr_factor_to_lds<RP>((const double*)(it.ws + offG), nparts, r, ts);      // (the first chunk's loads are in flight under it)
    for (; c0 < C; c0 += cstep) {
        __syncthreads();                             // previous chunk fully consumed
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u, rr = i >> 5, pc = (i & 31) * 8;
            *reinterpret_cast<h16x8*>(&dsm[rr * LDD + pc]) = dreg[u];
        }
        if constexpr (ORTH) {
            // q = z L^-T by forward substitution (ts: L row-major with 1 / diagonal on the diagonal, 0 for a dropped direction), as k_lr_apply2
            const float* zin = reinterpret_cast<const float*>(zreg);
            float out[RP];
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                float lrow[RP];
#pragma unroll
                for (int c = 0; c < (j + 4) / 4; ++c) *reinterpret_cast<float4*>(&lrow[4 * c]) = *reinterpret_cast<const float4*>(&ts[j * RP + 4 * c]);
                float sacc = zin[j];
#pragma unroll
                for (int k = 0; k < j; ++k) sacc = fmaf(-out[k], lrow[k], sacc);
                out[j] = sacc * lrow[j];
                if (RP > 16 && (j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const float a0 = out[j] * LR_Q_SCALE;
                const h16 h0 = (h16)a0;
                qsm[j * LDD + tid] = h0;
                qsm[(32 + j) * LDD + tid] = (h16)(a0 - (float)h0);
            }
            if (blockIdx.x == 0 && c0 + tid < C) {
#pragma unroll
                for (int k4 = 0; k4 < RP / 4; ++k4)
                    *reinterpret_cast<float4*>(Qout + (size_t)(c0 + tid) * RP + 4 * k4) = make_float4(out[4 * k4], out[4 * k4 + 1], out[4 * k4 + 2], out[4 * k4 + 3]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < QI; ++u) {
                const int i = tid + 256 * u, cp = i / (RP / 4), k4 = i - cp * (RP / 4);
                const float* q0p = reinterpret_cast<const float*>(&qreg[u][0]);
                const float* q1p = reinterpret_cast<const float*>(&qreg[u][1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a0 = q0p[e] * qscale, a1 = q1p[e] * qscale;
                    const h16 h0 = (h16)a0, h1 = (h16)a1;
                    h16x2 hi, lo;
                    hi[0] = h0; hi[1] = h1;
                    lo[0] = (h16)(a0 - (float)h0); lo[1] = (h16)(a1 - (float)h1);
                    *reinterpret_cast<h16x2*>(&qsm[(4 * k4 + e) * LDD + 2 * cp]) = hi;
                    *reinterpret_cast<h16x2*>(&qsm[(32 + 4 * k4 + e) * LDD + 2 * cp]) = lo;
                }
            }
        }
        __syncthreads();
        if (c0 + cstep < C) load_chunk(c0 + cstep);  // in flight while this chunk is multiplied
        const int cw = w * 64;                       // this wave's 64 columns of the chunk
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const h16x8 av = *reinterpret_cast<const h16x8*>(&dsm[li * LDD + cw + kk * 16 + lk * 8]);
            const h16x8 bh = *reinterpret_cast<const h16x8*>(&qsm[li * LDD + cw + kk * 16 + lk * 8]);
            const h16x8 bl = *reinterpret_cast<const h16x8*>(&qsm[(32 + li) * LDD + cw + kk * 16 + lk * 8]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bl, acc, 0, 0, 0);
        }
    }
    __syncthreads();                                 // every wave is done reading qsm before it becomes `red`
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) red[w][(rg & 3) + 8 * (rg >> 2) + 4 * lk][li] = acc[rg];
    __syncthreads();
    float mx = 0.f;
    for (int i = tid; i < 32 * RP; i += 256) {
        const int rr = i / RP, k = i - rr * RP;
        const float sum = (((red[0][rr][k] + red[1][rr][k]) + red[2][rr][k]) + red[3][rr][k]) * (1.f / qscale);
        if (n0 + rr < N) { Y[(size_t)(n0 + rr) * RP + k] = sum; mx = fmaxf(mx, fabsf(sum)); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) wmax[w] = mx;
    __syncthreads();
    if (tid == 0) ((float*)(it.ws + offYmax))[blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}

// ---------------------------------------------------------------------------------------------------------------------
// Z (C x RP) = D^T (C x N) . Y (N x RP)      the one GEMM-shaped step with a long inner dimension (K = N).
//   v_mfma_f32_32x32x16_f16: A[i][k] = D[n + k][c0 + i] (fp16 as it is), B[k][j] = Y[n + k][j] as fp16 hi + lo of Y scaled by a power of
//   two that brings its largest entry below 1 (from k_lr_aq's per-workgroup maxima; exact, undone on the result): 22 bits of Y instead
//   of an fp32-input MFMA's 24, at an eighth of its issue time.  An operand is 8 consecutive k - rows of D and Y - of one column: both
//   chunks are staged TRANSPOSED in LDS (two rows a thread, one 4-byte write per column).
//   workgroup = one 32-column tile of D x its share of N (blockIdx.z); its 4 waves split a 128-row chunk, partial 32x32 tiles are summed
//   in fixed order.  (Y = fixed-order sum of the `ny` column-group slabs k_lr_aq wrote.)  Epilogue: this tile's part of the r x r Gram matrix
//   in fp64.  gram_mode 0: Ztile^T Ztile ; 1: Qtile^T Ztile (Q = the orthonormal basis Y was formed with).
// ---------------------------------------------------------------------------------------------------------------------
template <int RP>
__global__ __launch_bounds__(256) void k_lr_aty(LrBatch b, int N, int C, size_t offD, size_t offY, size_t offZ, size_t offQ, size_t offG, int gram_mode,
                                                size_t offZp, size_t offTick, int rows_per_split, size_t offYmax, int nymax, int ny) {
    constexpr int NCH = 128, LDT = NCH + 8;          // rows of D / Y per chunk: each wave multiplies 32 of them; 272-byte LDS rows
    constexpr int YI = (NCH / 2) * (RP / 4) / 256 > 0 ? (NCH / 2) * (RP / 4) / 256 : 1;     // (row pair, 4 ranks) items per thread (2 at RP = 32)
    constexpr int YN = (NCH / 2) * (RP / 4);         // items of a chunk (128 at RP = 8: half the threads)
    const LrItem it = b.it[blockIdx.y];
    const h16* D = (const h16*)(it.ws + offD);
    const float* Y = (const float*)(it.ws + offY);
    float* Z = (float*)(it.ws + offZ);
    const float* Qb = (const float*)(it.ws + offQ);
    // column tiles 2j and 2j + 1 share every 128-byte line of D: workgroups i and i + 8 of a run of 16 take them - the same XCD, whose L2
    // then serves the line's second half (workgroups go to the XCDs round-robin)
    int tx = blockIdx.x;
    if ((tx | 15) < (int)gridDim.x) tx = (tx & ~15) + 2 * (tx & 7) + ((tx >> 3) & 1);
    double* Gp = (double*)(it.ws + offG) + (size_t)tx * RP * RP;
    __shared__ __attribute__((aligned(16))) h16 dT[32 * LDT];                      // D chunk transposed [column][row]: 8.5 KB
    __shared__ __attribute__((aligned(16))) h16 yT[2 * 32 * LDT];                  // Y chunk transposed, hi | lo: [rank][row], 17 KB
    __shared__ float red[4][32][33];
    __shared__ float wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c0 = tx * 32;
    const int li = lane & 31, lk = lane >> 5;
    const size_t slab = (size_t)N * RP;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    h16x8 dreg[2];
    float4 yreg[YI][2];
    const int nlim = min(N, (int)(blockIdx.z + 1) * rows_per_split);
    auto load_chunk = [&](int nb) {
        {                                            // rows 2 rp, 2 rp + 1 of the chunk, 8 columns
            const int rp = tid >> 2, pc = (tid & 3) * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                dreg[h] = (h16x8)(h16)0;
                if (nb + 2 * rp + h < nlim && c0 + pc < C) dreg[h] = *reinterpret_cast<const h16x8*>(D + (size_t)(nb + 2 * rp + h) * C + c0 + pc);
            }
        }
#pragma unroll
        for (int u = 0; u < YI; ++u) {
            const int i = tid + 256 * u, rp = i / (RP / 4), k4 = i - rp * (RP / 4);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                yreg[u][h] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < YN && nb + 2 * rp + h < nlim) {
                    const float* yp = Y + (size_t)(nb + 2 * rp + h) * RP + 4 * k4;
                    float4 sv = *reinterpret_cast<const float4*>(yp);
                    if (ny > 1) {                    // (uniform) the column groups' partials, in fixed order
                        const float4 s1 = *reinterpret_cast<const float4*>(yp + slab);
                        sv = make_float4(sv.x + s1.x, sv.y + s1.y, sv.z + s1.z, sv.w + s1.w);
                        if (ny > 2) {
                            const float4 s2 = *reinterpret_cast<const float4*>(yp + 2 * slab), s3 = *reinterpret_cast<const float4*>(yp + 3 * slab);
                            sv = make_float4((sv.x + s2.x) + s3.x, (sv.y + s2.y) + s3.y, (sv.z + s2.z) + s3.z, (sv.w + s2.w) + s3.w);
                        }
                    }
                    yreg[u][h] = sv;
                }
            }
        }
    };
    // blockIdx.z: this workgroup's share of the rows (a multiple of NCH); the last of a column tile's workgroups to arrive sums the
    // partial tiles in split order (bits do not depend on who is last) and goes on to the tile's Gram part
    const int NS = gridDim.z, nbeg = blockIdx.z * rows_per_split, nend = min(N, nbeg + rows_per_split);
    load_chunk(nbeg);
    // the power of two that brings |Y| below 1: Y is the sum of the column groups' partials, each workgroup of k_lr_aq left the largest
    // |entry| of its own
    float ysc, yinv;
    {
        const float* ym = (const float*)(it.ws + offYmax);
        float mx = 0.f;
        for (int i = tid; i < nymax; i += 256) mx = fmaxf(mx, ym[i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if (lane == 0) wmax[w] = mx;
        for (int i = tid; i < 2 * 32 * LDT / 8; i += 256) reinterpret_cast<h16x8*>(yT)[i] = (h16x8)(h16)0;        // ranks >= RP: zero columns of B
        __syncthreads();
        const float bound = (float)ny * fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        int e = 0;
        if (bound > 0.f && bound < 3.0e38f) (void)frexpf(bound, &e);
        e = max(-100, min(100, e));
        ysc = ldexpf(1.f, -e); yinv = ldexpf(1.f, e);
    }
    for (int nb = nbeg; nb < nend; nb += NCH) {
        __syncthreads();
        {
            const int rp = tid >> 2, pc = (tid & 3) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                h16x2 v; v[0] = dreg[0][e]; v[1] = dreg[1][e];
                *reinterpret_cast<h16x2*>(&dT[(pc + e) * LDT + 2 * rp]) = v;
            }
        }
#pragma unroll
        for (int u = 0; u < YI; ++u) {
            const int i = tid + 256 * u, rp = i / (RP / 4), k4 = i - rp * (RP / 4);
            if (i < YN) {
                const float* y0p = reinterpret_cast<const float*>(&yreg[u][0]);
                const float* y1p = reinterpret_cast<const float*>(&yreg[u][1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a0 = y0p[e] * ysc, a1 = y1p[e] * ysc;
                    const h16 h0 = (h16)a0, h1 = (h16)a1;
                    h16x2 hi, lo;
                    hi[0] = h0; hi[1] = h1;
                    lo[0] = (h16)(a0 - (float)h0); lo[1] = (h16)(a1 - (float)h1);
                    *reinterpret_cast<h16x2*>(&yT[(4 * k4 + e) * LDT + 2 * rp]) = hi;
                    *reinterpret_cast<h16x2*>(&yT[(32 + 4 * k4 + e) * LDT + 2 * rp]) = lo;
                }
            }
        }
        __syncthreads();
        if (nb + NCH < nend) load_chunk(nb + NCH);   // in flight while this chunk is multiplied
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {             // this wave's 32 rows of the chunk (rows beyond the share were staged as zeros)
            const int r8 = w * 32 + kk * 16 + lk * 8;
            const h16x8 av = *reinterpret_cast<const h16x8*>(&dT[li * LDT + r8]);
            const h16x8 bh = *reinterpret_cast<const h16x8*>(&yT[li * LDT + r8]);
            const h16x8 bl = *reinterpret_cast<const h16x8*>(&yT[(32 + li) * LDT + r8]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bl, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] *= yinv;
    // C/D layout: column j = lane & 31, row i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) red[w][(rg & 3) + 8 * (rg >> 2) + 4 * lk][li] = acc[rg];
    __syncthreads();
    if (NS == 1) {
        for (int i = tid; i < 32 * 32; i += 256) {
            const int cc = i >> 5, k = i & 31;
            const float s = ((red[0][cc][k] + red[1][cc][k]) + red[2][cc][k]) + red[3][cc][k];
            red[0][cc][k] = s;          // (cc, k) is read and written by this thread only
            if (c0 + cc < C && k < RP) Z[(size_t)(c0 + cc) * RP + k] = s;
        }
    } else {
        float* Zp = (float*)(it.ws + offZp);
        for (int i = tid; i < 32 * RP; i += 256) {
            const int cc = i / RP, k = i - cc * RP;
            const float s = ((red[0][cc][k] + red[1][cc][k]) + red[2][cc][k]) + red[3][cc][k];
            if (c0 + cc < C) __hip_atomic_store(&Zp[((size_t)blockIdx.z * C + c0 + cc) * RP + k], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // publish: every storing wave drains its write-through stores, then one lane draws the tile's ticket
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __shared__ int last_flag;
        if (tid == 0) {
            unsigned* tick = (unsigned*)(it.ws + offTick) + tx;
            const unsigned old = __hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = old == (unsigned)(NS - 1);
            if (last_flag) __hip_atomic_store(tick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // self-resetting
        }
        __syncthreads();
        if (!last_flag) return;
        for (int i = tid; i < 32 * RP; i += 256) {
            const int cc = i / RP, k = i - cc * RP;
            float pv[LR_NS_MAX];
#pragma unroll
            for (int t = 0; t < LR_NS_MAX; ++t)       // every load unconditional and in flight together (clamped split, masked value)
                pv[t] = __hip_atomic_load(&Zp[((size_t)min(t, NS - 1) * C + min(c0 + cc, C - 1)) * RP + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < LR_NS_MAX; ++t) s += (t < NS) ? pv[t] : 0.f;                              // fixed order
            if (c0 + cc >= C) s = 0.f;
            red[0][cc][k] = s;
            if (c0 + cc < C) Z[(size_t)(c0 + cc) * RP + k] = s;
        }
    }
    float* qt = reinterpret_cast<float*>(yT);        // the tile's 32 rows of Q (gram_mode 1): one load a thread instead of 32 dependent ones
    if (gram_mode)
        for (int i = tid; i < 32 * RP; i += 256) qt[i] = (c0 + i / RP < C) ? Qb[(size_t)c0 * RP + i] : 0.f;
    __syncthreads();
    for (int i = tid; i < RP * RP; i += 256) {
        const int a = i / RP, bb = i - a * RP;
        double g = 0.0;
        if (gram_mode) {
            for (int cc = 0; cc < 32; ++cc) g += (double)qt[cc * RP + a] * (double)red[0][cc][bb];     // (rows beyond C: zeros)
        } else {
            for (int cc = 0; cc < 32; ++cc)
                if (c0 + cc < C) g += (double)red[0][cc][a] * (double)red[0][cc][bb];
        }
        Gp[i] = g;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The factor every k_lr_apply2 workgroup needs, formed BY every workgroup (no launch of its own, nobody waits for a broadcast - what the
// slab-resident chain does with everything r x r sized): G = sum of the column tiles' partial Grams (fp64, fixed order: the same bits in
// every workgroup), symmetrised, factorised in one wave's registers (cfx_lr.h); ts (LDS, fp32) = L row-major with 1 / L[j][j] ON the
// diagonal (0 for a dropped direction - a non-positive pivot of a rank-deficient residual, e.g. x == base: its column of the result is
// zero instead of NaNs).  No triangular inverse: the caller solves Out L^T = In row by row.  256 threads.
// ---------------------------------------------------------------------------------------------------------------------
template <int RP>
__device__ __forceinline__ void lr_factor_to_lds(const double* Gp, int nparts, int r, float* ts) {
    constexpr int E = RP * RP, GRP = (256 / E) > 0 ? (256 / E) : 1, EPT = (E + 255) / 256, U = 8;   // (24 parts in flight: slower, registers)
    __shared__ double G[RP][RP + 1];
    __shared__ double part[GRP][E];
    const int tid = threadIdx.x;
    {
        const int e0 = tid % (E < 256 ? E : 256), g = (E < 256) ? tid / E : 0;
        double a[EPT][U];
#pragma unroll
        for (int q = 0; q < EPT; ++q)
#pragma unroll
            for (int u = 0; u < U; ++u) a[q][u] = 0.0;
        if (g < GRP) {
            int p = g;
            for (; p + (U - 1) * GRP < nparts; p += U * GRP) {
#pragma unroll
                for (int q = 0; q < EPT; ++q)
#pragma unroll
                    for (int u = 0; u < U; ++u) a[q][u] += Gp[(size_t)(p + u * GRP) * E + e0 + q * 256];
            }
            for (; p < nparts; p += GRP) {
#pragma unroll
                for (int q = 0; q < EPT; ++q) a[q][0] += Gp[(size_t)p * E + e0 + q * 256];
            }
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                double t = 0.0;
#pragma unroll
                for (int u = 0; u < U; ++u) t += a[q][u];
                part[g][e0 + q * 256] = t;
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < E; e += 256) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < GRP; ++q) t += part[q][e];
        G[e / RP][e % RP] = t;
    }
    __syncthreads();
    if (tid < 64) {
        double g[RP], myinv;
        const unsigned dead = lr_chol_rows<RP>(G, r, 1e-13, false, g, myinv);
        if (tid < RP) {
#pragma unroll
            for (int k = 0; k < RP; ++k) ts[tid * RP + k] = (k == tid) ? (((dead >> tid) & 1u) ? 0.f : (float)myinv) : (float)g[k];
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// Out = In (rows x RP fp32) . chol(G)^-T   one thread per row (the factor: lr_factor_to_lds, by the workgroup itself).
//   mode 0: fp32 rows x RP (the next Q) ; mode 1: fp16 rows x r, row-major (U, or V^T for LOW_RANK_Q) ;
//   mode 2: fp16 r x rows, i.e. transposed (V in the LOW_RANK wire layout)
// ---------------------------------------------------------------------------------------------------------------------
struct LrApply { int rows, r, in_slabs, mode, out_in_packet; size_t offIn, offG, offOut, pkt_off_halves; };
template <int RP>
__device__ __forceinline__ void lr_apply_body(const LrItem& it, int bx, int rows, int r, size_t offIn, int in_slabs, size_t offG, int nparts, int mode,
                                              size_t offOut, int out_in_packet, size_t pkt_off_halves, float* ts) {
    const float* In = (const float*)(it.ws + offIn);
    lr_factor_to_lds<RP>((const double*)(it.ws + offG), nparts, r, ts);
    const int row = bx * 256 + threadIdx.x;
    if (row >= rows) return;
    float in[RP], out[RP];
    const size_t slab = (size_t)rows * RP;
#pragma unroll
    for (int k = 0; k < RP; ++k) {
        float v = In[(size_t)row * RP + k];
        for (int p = 1; p < in_slabs; ++p) v += In[(size_t)p * slab + (size_t)row * RP + k];     // fixed order
        in[k] = v;
    }
    // Out = In L^-T by forward substitution (ts: L with 1 / diagonal on the diagonal); a row of L is read as 16-byte broadcasts
#pragma unroll
    for (int j = 0; j < RP; ++j) {
        float lrow[RP];
#pragma unroll
        for (int c = 0; c < (j + 4) / 4; ++c) *reinterpret_cast<float4*>(&lrow[4 * c]) = *reinterpret_cast<const float4*>(&ts[j * RP + 4 * c]);
        float s = in[j];
#pragma unroll
        for (int k = 0; k < j; ++k) s = fmaf(-out[k], lrow[k], s);
        out[j] = s * lrow[j];
        if (RP > 16 && (j & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // four rows of the factor in flight at a time (registers)
    }
    if (mode == 0) {
        float* O = (float*)(it.ws + offOut);
#pragma unroll
        for (int j = 0; j < RP; ++j) O[(size_t)row * RP + j] = out[j];
    } else {
        h16* O = out_in_packet ? ((h16*)it.packet + pkt_off_halves) : (h16*)(it.ws + offOut);
#pragma unroll
        for (int j = 0; j < RP; ++j) {
            if (j < r) {
                if (mode == 1) O[(size_t)row * r + j] = (h16)out[j];
                else O[(size_t)j * rows + row] = (h16)out[j];
            }
        }
    }
}
// ---------------------------------------------------------------------------------------------------------------------
// LOW_RANK_Q: the int4 quantiser of BOTH factors in one launch (quantize_int4 of u and of v.t(), slowpath.py:62-67, i.e.
// compress_quantize.py:552-573: per column min / max over the rows, scale = fp16(fp16(max - min) / 15.000001), codes
// round((x - min) / scale) clamped to 0 .. 15, two ROWS per byte) and - want_dq - the factors the receiver will see
// (q * scale + min, two fp16 roundings: compress_quantize.py:626-636) for the error-feedback decode.  The arithmetic of
// k_minmax_compress / k_int4_quant / k_int4_dequant applied to a rows x r matrix (bit-identical: tests), without their nine launches and
// copies for matrices of 17 K and 98 K elements.  grid.x: blocks [0, nsu) = U (N x r) in row shares (one for a shard of up to ~1000 rows),
// the next LRQ_VS = V^T (C x r) in row shares; every block finds the column statistics of its whole matrix itself (32 columns: cheaper
// than waiting for each other).
// ---------------------------------------------------------------------------------------------------------------------
#define LRQ_VS 6
static inline int lrq_u_shares(int N) { return std::max(1, std::min(8, N / 512)); }
struct LrQ4 { const h16* U; const h16* V; unsigned char* secU; unsigned char* secV; h16* Uq; h16* Vq; };
struct LrQ4Batch { LrQ4 it[LR_MAXB]; };
__global__ __launch_bounds__(1024) void k_lr_q4(LrQ4Batch b, int N, int C, int r, int want_dq, int nsu) {
    const LrQ4 it = b.it[blockIdx.y];
    const bool isu = (int)blockIdx.x < nsu;
    const int R = isu ? N : C, share = isu ? (int)blockIdx.x : (int)blockIdx.x - nsu, ns = isu ? nsu : LRQ_VS;
    const h16* X = isu ? it.U : it.V;
    unsigned char* sec = isu ? it.secU : it.secV;
    h16* Xq = isu ? it.Uq : it.Vq;
    // a thread: 8 consecutive columns (16 bytes) of a row; r / 8 threads a row, 1024 / (r / 8) rows a pass
    const int tid = threadIdx.x, oc = r >> 3, cq = tid % oc, rw = tid / oc, rpass = 1024 / oc;
    __shared__ h16x8 smn[1024], smx[1024];                    // [row of the pass][column octet]
    __shared__ h16 s2mn[32][33], s2mx[32][33];
    __shared__ h16 scs[32], mns[32];
    const h16 pinf = __builtin_bit_cast(h16, (unsigned short)0x7c00), ninf = __builtin_bit_cast(h16, (unsigned short)0xfc00);
    h16x8 mn = (h16x8)pinf, mx = (h16x8)ninf;
    if (rw < rpass) {
        for (int row0 = rw; row0 < R; row0 += 4 * rpass) {           // four rows in flight (clamped: a repeated row changes nothing)
            h16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const h16x8*>(X + (size_t)min(row0 + u * rpass, R - 1) * r + 8 * cq);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) { mn[e] = v[u][e] < mn[e] ? v[u][e] : mn[e]; mx[e] = v[u][e] > mx[e] ? v[u][e] : mx[e]; }
        }
    }
    smn[tid] = mn; smx[tid] = mx;
    __syncthreads();
    {
        // columns c < r, 32 row groups: group g takes rows g, g + 32, .. of the pass
        const int c = tid & 31, g = tid >> 5;
        h16 a = pinf, bb = ninf;
        if (c < r) {
            const h16* pm = reinterpret_cast<const h16*>(smn);
            const h16* px = reinterpret_cast<const h16*>(smx);
            for (int k = g; k < rpass; k += 32) {
                const h16 u = pm[(k * oc + (c >> 3)) * 8 + (c & 7)], v = px[(k * oc + (c >> 3)) * 8 + (c & 7)];
                a = u < a ? u : a;
                bb = v > bb ? v : bb;
            }
        }
        s2mn[g][c] = a; s2mx[g][c] = bb;
    }
    __syncthreads();
    if (tid < 32) {
        h16 a = pinf, bb = ninf;
        for (int k = 0; k < 32; ++k) {
            const h16 u = s2mn[k][tid], v = s2mx[k][tid];
            a = u < a ? u : a;
            bb = v > bb ? v : bb;
        }
        const h16 rng = bb - a;
        const h16 sc = (h16)((float)rng / 15.000001f);
        scs[tid] = sc; mns[tid] = a;
        if (share == 0 && tid < r) {
            h16* S = (h16*)(sec + (size_t)(R / 2) * r);
            S[tid] = sc;
            S[r + tid] = a;
        }
    }
    __syncthreads();
    h16x8 sc8, m8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc8[e] = scs[8 * cq + e]; m8[e] = mns[8 * cq + e]; }
    const int pairs = R / 2, p0 = (int)((long)pairs * share / ns), p1 = (int)((long)pairs * (share + 1) / ns);
    if (rw < rpass) {
        for (int kk = p0 + rw; kk < p1; kk += rpass) {
            h16x8 x[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) x[h] = *reinterpret_cast<const h16x8*>(X + (size_t)(2 * kk + h) * r + 8 * cq);
            unsigned long long outb = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const h16x8 dm = x[h] - m8;
                h16x8 dq;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    h16 v = __builtin_rintf16((h16)((float)dm[e] / (float)sc8[e]));      // round half to even (torch.round) of the fp16 quotient
                    if (v != v) v = (h16)0;
                    v = v < (h16)0 ? (h16)0 : v;
                    v = v > (h16)15.0f ? (h16)15.0f : v;
                    const unsigned qi = (unsigned)(float)v & 15u;
                    outb |= (unsigned long long)qi << (8 * e + 4 * h);
                    dq[e] = (h16)(float)qi;
                }
                if (want_dq) *reinterpret_cast<h16x8*>(Xq + (size_t)(2 * kk + h) * r + 8 * cq) = dq * sc8 + m8;      // two roundings (contraction is off)
            }
            *reinterpret_cast<unsigned long long*>(sec + (size_t)kk * r + 8 * cq) = outb;
        }
    }
}

// The receiver's side of k_lr_q4: both int4 factors of every tensor back to fp16 (q * scale + min, two roundings:
// compress_quantize.py:626-636; the arithmetic of k_int4_dequant) in one launch.  Sections may start at addresses that are only 8-byte
// aligned: the codes are read 8 bytes at a time.  grid.x as k_lr_q4's.
struct LrDq4 { const unsigned char* secU; const unsigned char* secV; h16* Uq; h16* Vq; };
struct LrDq4Batch { LrDq4 it[LR_MAXB]; };
__global__ __launch_bounds__(1024) void k_lr_dq4(LrDq4Batch b, int N, int C, int r, int nsu) {
    const LrDq4 it = b.it[blockIdx.y];
    const bool isu = (int)blockIdx.x < nsu;
    const int R = isu ? N : C, share = isu ? (int)blockIdx.x : (int)blockIdx.x - nsu, ns = isu ? nsu : LRQ_VS;
    const unsigned char* sec = isu ? it.secU : it.secV;
    h16* Xq = isu ? it.Uq : it.Vq;
    const int tid = threadIdx.x, oc = r >> 3, cq = tid % oc, rw = tid / oc, rpass = 1024 / oc;
    const h16* S = (const h16*)(sec + (size_t)(R / 2) * r);
    h16x8 sc8, m8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc8[e] = S[8 * cq + e]; m8[e] = S[r + 8 * cq + e]; }
    const int pairs = R / 2, p0 = (int)((long)pairs * share / ns), p1 = (int)((long)pairs * (share + 1) / ns);
    if (rw < rpass) {
        for (int kk = p0 + rw; kk < p1; kk += rpass) {
            const unsigned long long by = *reinterpret_cast<const unsigned long long*>(sec + (size_t)kk * r + 8 * cq);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                h16x8 dq;
#pragma unroll
                for (int e = 0; e < 8; ++e) dq[e] = (h16)(float)((unsigned)(by >> (8 * e + 4 * h)) & 15u);
                *reinterpret_cast<h16x8*>(Xq + (size_t)(2 * kk + h) * r + 8 * cq) = dq * sc8 + m8;      // two roundings (contraction is off)
            }
        }
    }
}

// two products by the same factor in one launch (U = Y T and V = Z' T at the end of the chain): blocks [0, nb0) do `a`, the rest `c`
template <int RP>
__global__ __launch_bounds__(256) void k_lr_apply2(LrBatch b, LrApply a, LrApply c, int nb0, int nparts) {
    __shared__ __attribute__((aligned(16))) float ts[RP * RP];
    const bool first = (int)blockIdx.x < nb0;
    const LrApply& p = first ? a : c;
    lr_apply_body<RP>(b.it[blockIdx.y], first ? blockIdx.x : blockIdx.x - nb0, p.rows, p.r, p.offIn, p.in_slabs, p.offG, nparts, p.mode, p.offOut,
                      p.out_in_packet, p.pkt_off_halves, ts);
}

// ---------------------------------------------------------------------------------------------------------------------
// out = base + fp16( sum_k U16[n][k] * V16[k][c] )      decode of LOW_RANK (slowpath.py:151-154: torch.matmul(u, v), fp32
// accumulation, fp16 result) fused with the residual add (main.py:232, :376).  VT: V given transposed (C x r) as LOW_RANK_Q
// stores it.  Tile = 8 rows x 512 channels like the other dequant kernels; V lives in registers, U rows are broadcast loads.
// ---------------------------------------------------------------------------------------------------------------------

template <int RP, bool VT>
__global__ __launch_bounds__(256) void k_lr_decode(LrDecBatch b, int N, int C, int r) {
    const LrDec it = b.it[blockIdx.z];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x * 512 + lane * 8;
    const int r0 = blockIdx.y * 32, r1 = min(N, r0 + 32);
    // The tile's U rows (32 x r halves, one contiguous block) go through LDS once: every lane of a wave needs the same
    // u pair at the same time, which LDS serves as a broadcast read instead of a dependent global load per k-pair and row.
    __shared__ h16x2 us[32][RP / 2];
    {
        const int hp = r >> 1, npairs = (r1 - r0) * hp;
        const h16x2* src = reinterpret_cast<const h16x2*>(it.U + (size_t)r0 * r);
        for (int i = threadIdx.x; i < npairs; i += 256) us[i / hp][i % hp] = src[i];
    }
    const bool act = c < C;
    // V as k-pairs: vp[kk][i] = (V[2kk][c+i], V[2kk+1][c+i]) so one v_dot2_f32_f16 does two MACs with an fp32 accumulator
    h16x2 vp[RP / 2][8];
#pragma unroll
    for (int kk = 0; kk < RP / 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 8; ++i) vp[kk][i] = (h16x2)(h16)0;
        if (act && 2 * kk < r) {
            if (VT) {
                if (r != RP) {      // generic: 4-byte gathers
#pragma unroll
                    for (int i = 0; i < 8; ++i) vp[kk][i] = *reinterpret_cast<const h16x2*>(it.V + (size_t)(c + i) * r + 2 * kk);
                }
            } else {
                const h16x8 lo = *reinterpret_cast<const h16x8*>(it.V + (size_t)(2 * kk) * C + c);
                const h16x8 hi = *reinterpret_cast<const h16x8*>(it.V + (size_t)(2 * kk + 1) * C + c);
#pragma unroll
                for (int i = 0; i < 8; ++i) { vp[kk][i][0] = lo[i]; vp[kk][i][1] = hi[i]; }
            }
        }
    }
    if (VT && r == RP && act) {
        // V^T rows c .. c+7 are one contiguous block of 8 * RP halves: coalesced 16-byte loads, then pick the pairs
        const h16x8* blk = reinterpret_cast<const h16x8*>(it.V + (size_t)c * RP);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int q = 0; q < RP / 8; ++q) {
                const h16x8 t = blk[i * (RP / 8) + q];
#pragma unroll
                for (int e = 0; e < 4; ++e) { vp[q * 4 + e][i][0] = t[2 * e]; vp[q * 4 + e][i][1] = t[2 * e + 1]; }
            }
        }
    }
    __syncthreads();
    if (!act) return;
    // two rows in flight per wave: both state loads are issued before the dot products
    for (int lr = w; lr < r1 - r0; lr += 8) {
        const int ra = r0 + lr, rb = ra + 4;
        const bool hb = rb < r1;
        h16x8 ba = (h16x8)(h16)0, bb = (h16x8)(h16)0;
        if (it.base) {
            ba = __builtin_nontemporal_load(reinterpret_cast<const h16x8*>(it.base + (size_t)ra * C + c));
            if (hb) bb = __builtin_nontemporal_load(reinterpret_cast<const h16x8*>(it.base + (size_t)rb * C + c));
        }
        float acca[8], accb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { acca[i] = 0.f; accb[i] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < RP / 2; ++kk) {
            if (2 * kk < r) {
                const h16x2 ua = us[lr][kk];
                const h16x2 ub = us[hb ? lr + 4 : lr][kk];
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
        __builtin_nontemporal_store(oa, reinterpret_cast<h16x8*>(it.out + (size_t)ra * C + c));
        if (hb) __builtin_nontemporal_store(ob, reinterpret_cast<h16x8*>(it.out + (size_t)rb * C + c));
    }
}

// MFMA form of the same product for the larger ranks, where the VALU form above runs out of issue slots and registers.
// The product is computed transposed,
//   out^T tile (32 cols x 32 rows) = V^T tile (32 cols x r)  @  U^T tile (r x 32 rows)      v_mfma_f32_32x32x8_f16, r/8 steps
// (A operand = V^T: lane l holds column c + (l & 31), k = 8 ks + 4 (l >> 5) + 0..3; B operand = U^T: row r0 + (l & 31), same
// k; D: lane l holds output row r0 + (l & 31), columns c + 8 g + 4 (l >> 5) + 0..3 in registers 4g .. 4g+3), so a lane ends
// up with 4 consecutive columns of one row = one 8-byte LDS write.  A workgroup covers 32 rows x 512 columns (wave w: columns
// 128 w .. 128 w + 127, four MFMA column tiles), parks the fp16 product in LDS and then runs the SAME epilogue access
// pattern as the VALU kernel: a wave reads one full 512-column row segment (16 B per lane), adds the state, streams it out.
// (Writing the MFMA registers straight to memory - 32 rows x 16 B per instruction - measured 5x slower than that.)
// The V^T fragments stay in registers while the workgroup walks down rows_per_wg rows.  fp32 accumulation in MFMA order;
// sender (error feedback) and receiver run this same kernel, so their states stay bit-identical.
#define LR_DEC_LDS_STRIDE (512 + 4)          // halves per LDS row: +8 bytes so the 32 row-lanes of a write spread over banks

template <int RP, bool VT>
__global__ __launch_bounds__(256) void k_lr_decode_mfma(LrDecBatch b, int N, int C, int r, int rows_per_wg) {
    const LrDec it = b.it[blockIdx.z];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int cx = blockIdx.x * 512;
    __shared__ h16 tile[32 * LR_DEC_LDS_STRIDE];
    const bool exact = r == RP;                    // factor rows are whole 8-byte aligned k-chunks
    h16x4 a[4][RP / 8];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int col = cx + w * 128 + ct * 32 + li;
#pragma unroll
        for (int ks = 0; ks < RP / 8; ++ks) {
            a[ct][ks] = (h16x4)(h16)0;
            const int k0 = ks * 8 + 4 * hi;
            if (col < C) {
                if (VT && exact) {
                    a[ct][ks] = *reinterpret_cast<const h16x4*>(it.V + (size_t)col * r + k0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k0 + e < r) a[ct][ks][e] = VT ? it.V[(size_t)col * r + k0 + e] : it.V[(size_t)(k0 + e) * C + col];
                }
            }
        }
    }
    const int c = cx + lane * 8;                   // epilogue: this lane's 8 columns
    const int rbeg = blockIdx.y * rows_per_wg, rend = min(N, rbeg + rows_per_wg);
    for (int row0 = rbeg; row0 < rend; row0 += 32) {
        const int row = row0 + li;
        h16x4 bq[RP / 8];
#pragma unroll
        for (int ks = 0; ks < RP / 8; ++ks) {
            bq[ks] = (h16x4)(h16)0;
            const int k0 = ks * 8 + 4 * hi;
            if (row < rend) {
                if (exact) {
                    bq[ks] = *reinterpret_cast<const h16x4*>(it.U + (size_t)row * r + k0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k0 + e < r) bq[ks][e] = it.U[(size_t)row * r + k0 + e];
                }
            }
        }
        // the state rows of the epilogue are requested before the matrix work
        h16x8 bs[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int rr = row0 + w + 4 * q;
            bs[q] = (h16x8)(h16)0;
            if (it.base && rr < rend && c < C) bs[q] = __builtin_nontemporal_load(reinterpret_cast<const h16x8*>(it.base + (size_t)rr * C + c));
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            f32x16 acc = (f32x16)0.f;
#pragma unroll
            for (int ks = 0; ks < RP / 8; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a[ct][ks], bq[ks], acc, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (h16)acc[4 * g + e];
                *reinterpret_cast<h16x4*>(&tile[li * LR_DEC_LDS_STRIDE + w * 128 + ct * 32 + 8 * g + 4 * hi]) = o;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int lr = w + 4 * q, rr = row0 + lr;
            if (rr < rend && c < C) {
                // LDS rows are 8-byte aligned only (stride 1032 B): two 8-byte reads
                const h16x4 lo = *reinterpret_cast<const h16x4*>(&tile[lr * LR_DEC_LDS_STRIDE + lane * 8]);
                const h16x4 hi4 = *reinterpret_cast<const h16x4*>(&tile[lr * LR_DEC_LDS_STRIDE + lane * 8 + 4]);
                h16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi4[e]; }
                if (it.base) o = bs[q] + o;
                __builtin_nontemporal_store(o, reinterpret_cast<h16x8*>(it.out + (size_t)rr * C + c));
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------

static bool lr_shape_ok(int quantized, int N, int C, int rank) {
    if (N <= 0 || C <= 0 || (C % 8) != 0 || rank < 2 || rank > 32 || (rank & 1)) return false;   // k-pairs: even rank
    if (quantized && ((N % 2) || (C % 2) || (rank % 8))) return false;
    return true;
}

#define LR_DISPATCH(RP_, CALL) do { if (RP_ == 8) { constexpr int RP = 8; CALL; } else if (RP_ == 16) { constexpr int RP = 16; CALL; } else { constexpr int RP = 32; CALL; } } while (0)

extern "C" {

size_t cfx_lr_packet_bytes(int quantized, int N, int C, int rank) {
    if (!lr_shape_ok(quantized, N, C, rank)) return 0;
    if (!quantized) return (size_t)(N + C) * rank * 2;
    return (size_t)N * rank / 2 + 4 * rank + (size_t)C * rank / 2 + 4 * rank;
}

size_t cfx_lr_workspace_bytes(int quantized, int N, int C, int rank, int batch) {
    if (!lr_shape_ok(quantized, N, C, rank) || batch < 1 || batch > LR_MAXB) return 0;
    return cfx_i_lr_workspace_bytes_any(N, C, rank, batch);
}

size_t cfx_i_lr_workspace_bytes_any(int N, int C, int rank, int batch) {
    size_t per = lr_layout(N, C, lr_rp(rank)).total;
    // the int4 factor quantiser's own scratch (min/max partials), for the larger factor
    per += al256(cfx_workspace_bytes(CFX_CODEC_INT4, (N > C ? N : C) + ((N > C ? N : C) & 1), 32, 0, 1));
    return per * batch;
}

void cfx_i_lr_factor_offsets(int N, int C, int rank, size_t* offU16, size_t* offV16, size_t* per) {
    const LrWs w = lr_layout(N, C, lr_rp(rank));
    *offU16 = w.U16; *offV16 = w.V16;
    *per = cfx_i_lr_workspace_bytes_any(N, C, rank, 1);
}

int cfx_i_lr_decode_launch(cfx_ctx* ctx, int N, int C, int rank, int batch, const LrDec* items, bool vt, hipStream_t s) {
    LrDecBatch db;
    memset(&db, 0, sizeof(db));
    for (int i = 0; i < batch; ++i) db.it[i] = items[i];
    const int RPv = lr_rp(rank);
    // rank <= 16: the VALU form is at the HBM roofline (tools/lowrank_bench.py); rank 32: it runs out of issue slots and
    // registers (2.8x over the roofline), the MFMA form is back at it.  cfx_set_lr_decode forces one (measurements).
    const bool mfma = ctx->lr_decode ? ctx->lr_decode == 2 : RPv == 32;
    if (mfma) {
        const int CBk = (C + 511) / 512;
        int rows = 32;
        for (int cand = 128; cand >= 64; cand >>= 1)
            if ((long)CBk * ((N + cand - 1) / cand) * batch >= 768) { rows = cand; break; }
        const dim3 grid(CBk, (N + rows - 1) / rows, batch);
        if (vt) LR_DISPATCH(RPv, LAUNCH(ctx, KID_LR_DECODE, s, (k_lr_decode_mfma<RP, true>), grid, dim3(256), 0, s, db, N, C, rank, rows));
        else LR_DISPATCH(RPv, LAUNCH(ctx, KID_LR_DECODE, s, (k_lr_decode_mfma<RP, false>), grid, dim3(256), 0, s, db, N, C, rank, rows));
        return check_launch(ctx, "lr decode launch");
    }
    const dim3 grid((C + 511) / 512, (N + 31) / 32, batch);
    if (vt) LR_DISPATCH(RPv, LAUNCH(ctx, KID_LR_DECODE, s, (k_lr_decode<RP, true>), grid, dim3(256), 0, s, db, N, C, rank));
    else LR_DISPATCH(RPv, LAUNCH(ctx, KID_LR_DECODE, s, (k_lr_decode<RP, false>), grid, dim3(256), 0, s, db, N, C, rank));
    return check_launch(ctx, "lr decode launch");
}

// init_q[i]: device pointer to a C x RP fp32 matrix (RP = 8/16/32 >= rank; columns >= rank zero) - the random start.
int cfx_lr_compress_batch(cfx_ctx* ctx, int quantized, int N, int C, int rank, int flags, int batch, const cfx_comp_item* items,
                          const void* const* init_q, void* workspace, size_t workspace_bytes, void* stream) {
    if (!ctx || !items || !init_q) return fail(ctx, CFX_ERR_NULL, "lr compress: null ctx/items/init_q");
    if (batch < 1 || batch > LR_MAXB) return fail(ctx, CFX_ERR_BATCH, "lr compress: batch out of range");
    // internal callers (the rank-K scales of the 1-bit codec, cfx_binary_rank_*): factorise |x - base| and stop once the fp16 factors
    // U (N x r) and V^T (C x r) are in the workspace; any rank 1 .. 32
    const bool factors_only = flags & CFX_I_FLAG_LR_FACTORS_ONLY;
    const int absd = (flags & CFX_I_FLAG_LR_ABS) ? 1 : 0;
    if (factors_only ? !(N > 0 && C > 0 && C % 8 == 0 && rank >= 1 && rank <= 32 && quantized) : !lr_shape_ok(quantized, N, C, rank))
        return fail(ctx, CFX_ERR_SHAPE, "lr compress: bad shape/rank");
    const size_t need = cfx_i_lr_workspace_bytes_any(N, C, rank, batch);
    if (!workspace || workspace_bytes < need) return fail(ctx, CFX_ERR_WORKSPACE, "lr compress: workspace too small");
    const bool upd = (flags & CFX_FLAG_UPDATE_CACHE) && !factors_only;
    const int RPv = lr_rp(rank);
    const LrWs w = lr_layout(N, C, RPv);
    const size_t per = need / batch;
    LrBatch b;
    memset(&b, 0, sizeof(b));
    for (int i = 0; i < batch; ++i) {
        if (!items[i].x || !items[i].packet || !init_q[i]) return fail(ctx, CFX_ERR_NULL, "lr compress: null x/packet/init_q");
        if (upd && !items[i].new_base) return fail(ctx, CFX_ERR_NULL, "lr compress: UPDATE_CACHE needs new_base");
        if (!AL16(items[i].x) || !AL16(items[i].base) || !AL16(items[i].new_base) || !AL16(items[i].packet) || !AL16(init_q[i]))
            return fail(ctx, CFX_ERR_ALIGN, "lr compress: pointers must be 16-byte aligned");
        b.it[i].x = (const h16*)items[i].x; b.it[i].base = (const h16*)items[i].base; b.it[i].new_base = (h16*)items[i].new_base;
        b.it[i].packet = items[i].packet; b.it[i].q0 = (const float*)init_q[i]; b.it[i].ws = (char*)workspace + per * i;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t E = (size_t)N * C;
    const int nparts = (C + 31) / 32;
    const int ns_ = lr_aty_splits(N, C, batch);
    const int rps = ((N + ns_ - 1) / ns_ + 127) / 128 * 128;          // rows per split: whole 128-row chunks
    // column groups of Y = D Q (each writes a partial of Y every reader sums): as few as still give the machine a workgroup per CU
    const int rt_ = ((N + 31) / 32) * 2, gy_ = rt_ >= 256 ? 1 : (rt_ >= 128 ? 2 : 4);       // (a K,V pair's worth whatever the batch: same bits)
#define LR_GY0_MUL 2          // (measured at (4096, 1152): x1 20.6 + 11.9 us for the first two launches, x2 16.5 + 13.7, x4 15.0 + 15.8)
    const int gy0_ = std::min(4, gy_ * LR_GY0_MUL);      // the first product also forms D = x - base: three times the bytes
    const dim3 g_aq((N + 31) / 32, gy_, batch), g_aq0((N + 31) / 32, gy0_, batch), g_aty(nparts, batch, (N + rps - 1) / rps), g_apc((C + 255) / 256, batch), g_apn((N + 255) / 256, batch);
    // The N-space chain (cfx_lrgram.hip: 5 launches up to the factors) for shards whose Gram matrix is small, else the C-space chain.
    // (rank > 16: the two factorisations the chain's last launch runs back to back in one wave spill at RP = 32 - measured slower than the
    // C-space chain's separate launches)
    // rank 32: only the slab-resident form of the N-space chain takes it (where its workgroups fit the stream's CUs)
    const bool slab32 = RPv == 32 && ctx->lr_chain == 0 && cfx_i_lrs_fit(ctx, N, C, RPv, (void*)s) >= 1;
    const bool gram = cfx_i_lrg_ok(N, C) && (RPv <= 16 || slab32) && ctx->lr_chain != 2;
    int decoded = 0;                     // the single-launch chain also does the error-feedback update of LOW_RANK
    if (gram) {
        const int rg = cfx_i_lrg_factors(ctx, quantized, N, C, rank, batch, b, w.D, w.U16, w.V16, w.gram, absd,
                                         (upd && !(flags & CFX_FLAG_NO_EF) && !factors_only) ? 1 : 0, &decoded, s);
        if (rg != CFX_OK) return rg;
    } else {
        // D = x - base is formed (and stored) by the first product.
        // Y = D Q0 ; Z = D^T Y ; then twice: Y = D orth(Z) (the orthonormalisation inside the product) ; Z = D^T Y.  The last Z comes with
        // Q^T Z = Y^T Y for the factor U = Y chol(.)^-T and V = (Z chol(.)^-T)^T (k_lr_apply2 below): 7 launches up to the factors.
        const LrApply aq_ = {C, rank, 1, 0, 0, w.Zb, w.Gp, w.Qa, 0};
        LR_DISPATCH(RPv, LAUNCH(ctx, KID_LR_AQ, s, (k_lr_aq<RP, true, false>), g_aq0, dim3(256), 0, s, b, N, C, w.D, w.Qa, w.Y, 1, absd, w.tick, w.Ymax, w.Zb, w.Gp, nparts, rank));
        for (int iter = 0; iter < 3; ++iter) {
            const int gy_i = iter == 0 ? gy0_ : gy_;
            LR_DISPATCH(RPv, LAUNCH(ctx, KID_LR_ATY, s, (k_lr_aty<RP>), g_aty, dim3(256), 0, s, b, N, C, w.D, w.Y, w.Zb, w.Qa, w.Gp, iter == 2 ? 1 : 0, w.Zp, w.tick, rps, w.Ymax,
                                    (int)(g_aq.x * gy_i), gy_i));
            if (iter < 2) {
                // rank <= 16: the orthonormalisation inside the product (measured at (4096, 1152): r = 8 92 -> 87 us; at rank 32 the 32-step
                // factorisation and substitution in 256 workgroups of one wave a SIMD cost more than the launch they save: 202 -> 215)
                if (RPv <= 16) LR_DISPATCH(RPv, LAUNCH(ctx, KID_LR_AQ, s, (k_lr_aq<(RP <= 16 ? RP : 16), false, true>), g_aq, dim3(256), 0, s, b, N, C, w.D, w.Qa, w.Y, 0, 0, w.tick, w.Ymax, w.Zb, w.Gp, nparts, rank));
                else {
                    LR_DISPATCH(RPv, LAUNCH(ctx, KID_LR_APPLY, s, (k_lr_apply2<RP>), g_apc, dim3(256), 0, s, b, aq_, aq_, (int)g_apc.x, nparts));
                    LR_DISPATCH(RPv, LAUNCH(ctx, KID_LR_AQ, s, (k_lr_aq<RP, false, false>), g_aq, dim3(256), 0, s, b, N, C, w.D, w.Qa, w.Y, 0, 0, w.tick, w.Ymax, w.Zb, w.Gp, nparts, rank));
                }
            }
        }
    }
    int rc = CFX_OK;
    LrDec dec[LR_MAXB];
    if (!quantized) {
        // U (N x r) and V (r x C) straight into the packet: [U | V]
        const LrApply au = {N, rank, gy_, 1, 1, w.Y, w.Gp, 0, 0}, av = {C, rank, 1, 2, 1, w.Zb, w.Gp, 0, (size_t)N * rank};
        if (!gram) LR_DISPATCH(RPv, LAUNCH(ctx, KID_LR_APPLY, s, (k_lr_apply2<RP>), dim3(g_apn.x + g_apc.x, batch), dim3(256), 0, s, b, au, av, (int)g_apn.x, nparts));
        for (int i = 0; i < batch; ++i) {
            dec[i].U = (const h16*)items[i].packet; dec[i].V = (const h16*)items[i].packet + (size_t)N * rank;
            dec[i].base = (const h16*)items[i].base; dec[i].out = (h16*)items[i].new_base;
        }
    } else {
        // U16 (N x r), V^T16 (C x r) -> int4 factor quantiser (the native int4 kernel) -> packet sections; then the dequantised
        // factors (what the receiver will see) feed the error-feedback decode
        const LrApply au = {N, rank, gy_, 1, 0, w.Y, w.Gp, w.U16, 0}, av = {C, rank, 1, 1, 0, w.Zb, w.Gp, w.V16, 0};
        if (!gram) LR_DISPATCH(RPv, LAUNCH(ctx, KID_LR_APPLY, s, (k_lr_apply2<RP>), dim3(g_apn.x + g_apc.x, batch), dim3(256), 0, s, b, au, av, (int)g_apn.x, nparts));
        if (factors_only) return check_launch(ctx, "lr factor launch");
        const size_t secU = (size_t)N * rank / 2 + 4 * rank, secV = (size_t)C * rank / 2 + 4 * rank;       // bytes
        const size_t i4ws_off = w.total;
        // both factors of every tensor through the int4 quantiser in ONE launch, straight into the packet sections; with error
        // feedback also the dequantised factors (what the receiver will see) for the decode below
        LrQ4Batch qb;
        memset(&qb, 0, sizeof(qb));
        for (int i = 0; i < batch; ++i) {
            char* wsi = (char*)workspace + per * i;
            char* pk = (char*)items[i].packet;
            qb.it[i] = {(const h16*)(wsi + w.U16), (const h16*)(wsi + w.V16), (unsigned char*)pk, (unsigned char*)pk + secU, (h16*)(wsi + w.Uq), (h16*)(wsi + w.Vq)};
            dec[i].U = (const h16*)(wsi + w.Uq);
            dec[i].V = (const h16*)(wsi + w.Vq);
            dec[i].base = (const h16*)items[i].base; dec[i].out = (h16*)items[i].new_base;
        }
        (void)secV; (void)i4ws_off;
        LAUNCH(ctx, KID_INT4_QUANT, s, k_lr_q4, dim3(lrq_u_shares(N) + LRQ_VS, batch), dim3(1024), 0, s, qb, N, C, rank, (upd && !(flags & CFX_FLAG_NO_EF)) ? 1 : 0, lrq_u_shares(N));
    }
    if (upd) {
        if (flags & CFX_FLAG_NO_EF) {
            for (int i = 0; i < batch; ++i)
                if (items[i].new_base != items[i].x) (void)hipMemcpyAsync(items[i].new_base, items[i].x, E * 2, hipMemcpyDeviceToDevice, s);
        } else if (!decoded) {
            rc = cfx_i_lr_decode_launch(ctx, N, C, rank, batch, dec, quantized != 0, s);
            if (rc != CFX_OK) return rc;
        }
    }
    return check_launch(ctx, "lr compress launch");
}

int cfx_lr_decompress_batch(cfx_ctx* ctx, int quantized, int N, int C, int rank, int batch, const cfx_decomp_item* items,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!ctx || !items) return fail(ctx, CFX_ERR_NULL, "lr decompress: null ctx/items");
    if (batch < 1 || batch > LR_MAXB) return fail(ctx, CFX_ERR_BATCH, "lr decompress: batch out of range");
    if (!lr_shape_ok(quantized, N, C, rank)) return fail(ctx, CFX_ERR_SHAPE, "lr decompress: bad shape/rank");
    hipStream_t s = (hipStream_t)stream;
    LrDec dec[LR_MAXB];
    for (int i = 0; i < batch; ++i) {
        if (!items[i].packet || !items[i].recon) return fail(ctx, CFX_ERR_NULL, "lr decompress: null packet/recon");
        if (!AL16(items[i].packet) || !AL16(items[i].recon) || !AL16(items[i].base)) return fail(ctx, CFX_ERR_ALIGN, "lr decompress: pointers must be 16-byte aligned");
    }
    if (!quantized) {
        for (int i = 0; i < batch; ++i) {
            dec[i].U = (const h16*)items[i].packet; dec[i].V = (const h16*)items[i].packet + (size_t)N * rank;
            dec[i].base = (const h16*)items[i].base; dec[i].out = (h16*)items[i].recon;
        }
        return cfx_i_lr_decode_launch(ctx, N, C, rank, batch, dec, false, s);
    }
    const size_t need = cfx_lr_workspace_bytes(quantized, N, C, rank, batch);
    if (!workspace || workspace_bytes < need) return fail(ctx, CFX_ERR_WORKSPACE, "lr decompress: workspace too small");
    const LrWs w = lr_layout(N, C, lr_rp(rank));
    const size_t per = need / batch;
    const size_t secU = (size_t)N * rank / 2 + 4 * rank, secV = (size_t)C * rank / 2 + 4 * rank;
    LrDq4Batch qb;
    memset(&qb, 0, sizeof(qb));
    (void)secV;
    for (int i = 0; i < batch; ++i) {
        char* wsi = (char*)workspace + per * i;
        const unsigned char* pk = (const unsigned char*)items[i].packet;
        qb.it[i] = {pk, pk + secU, (h16*)(wsi + w.Uq), (h16*)(wsi + w.Vq)};
        dec[i].U = (const h16*)(wsi + w.Uq); dec[i].V = (const h16*)(wsi + w.Vq);
        dec[i].base = (const h16*)items[i].base; dec[i].out = (h16*)items[i].recon;
    }
    LAUNCH(ctx, KID_INT4_DEQUANT, s, k_lr_dq4, dim3(lrq_u_shares(N) + LRQ_VS, batch), dim3(1024), 0, s, qb, N, C, rank, lrq_u_shares(N));
    return cfx_i_lr_decode_launch(ctx, N, C, rank, batch, dec, true, s);
}

}  // extern "C"

// Low-rank residual codecs, slab-resident form of the factorisation chain - part of libcfx.so.
//
// The reference's subspace_iter (xfuser/compact/compress_lowrank.py:14-61) for A = x - base (N x C, N << C), written as in
// cfx_lrgram.hip (orthonormalisation after every multiplication, the r x r factors by Cholesky of the Gram matrices):
//     Y0 = A Q0
//     W1 = A (A^T Y0)      M1 = Y0^T W1      T1 = chol(M1)^-T      Y1 = W1 T1
//     W2 = A (A^T Y1)      U = W2 chol(W2^T W2)^-T        (= W2 T2 T3 of the N-space chain, whose T2 T3 is this inverse factor)
//     V  = U^T A           new_base = base + fp16(U V)
// Here A is never written anywhere and A A^T is never formed.  ONE persistent launch; workgroup j of a tensor owns the 32-column
// slab j of A for the whole chain: the slab is read from x and base exactly once (HBM: x + base in, new_base out - 6 bytes an
// element, what the 1-bit codec moves), kept in REGISTERS in the B-operand layout of v_mfma_f32_16x16x32_f16 (a lane: 8 consecutive
// columns of one row) and, transposed, in LDS (35 KB).  Against the slab both halves of a product are local:
//     Z  = slab^T Y   (32 x r,  K = N)     A operand = transposed slab, B operand = Y^T as fp16 hi + lo (LDS)
//     Wp = slab Z     (N x r,   K = 32)    A operand = Z^T as fp16 hi + lo, B operand = the slab registers
// and what remains between two products is the sum of the N x r partials over the slabs - three such sums in the chain (Y0, W1,
// W2).  No barrier anywhere: every fp32 word that changes hands carries a 2-bit SEQUENCE TAG in its two lowest mantissa bits (the
// value is rounded to 22 bits of mantissa - what the fp16 hi + lo operands of the products keep anyway), the n-th sum since the
// hand-over arena was zeroed writes tag (n + 1) mod 4, and a reader polls the words it needs (16-byte L2-bypassing loads) until all
// carry this sum's tag: a word holds either what the previous sum left there or the new value, never anything else, because the
// arena belongs to the context and nothing but this kernel writes to it.  A workgroup sums its 1 / nwg share of the cells over all
// partials in fixed order (reproducible run to run), publishes the share the same way, and every workgroup polls the N x r result:
// two memory hops per sum (~1.7 us each) instead of two grid barriers and two hops.  The count of sums lives in the arena (the last
// workgroup of a launch moves it on): nothing depends on a host-side counter, the launch can be captured in a hipGraph.
// Everything r x r sized - the fp64 Gram
// matrices (v_mfma_f64_16x16x4_f64 over the fp32 values: exact products), the factorisations, Y = W T (v_mfma_f32_16x16x4_f32) - is
// done by every workgroup redundantly: nobody waits for a broadcast.  V = U^T slab and the state update of the slab's columns come
// from the registers the slab and base were loaded into at the start, with the arithmetic of k_lr_decode (the receiver's kernel).
//
// The workgroups of a launch wait for each other: the host only takes this form when C / 32 workgroups per tensor are co-resident
// on the CUs the stream may use (one workgroup per CU: ~150 KB of LDS); otherwise the multi-launch chains run.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "cfx_lr.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef unsigned lrs_u4 __attribute__((ext_vector_type(4)));

#define LRS_PIVOT_TOL 1e-10   // first factorisation, as the N-space chain (cfx_lrgram.hip): pivots below this fraction of the largest are dropped
#define LRS_PIVOT_REL 1e-9    // last factorisation: pivots below this fraction of their own diagonal entry are dropped
#define LRS_SW 32             // columns of a slab
#define LRS_NW 8              // waves of a workgroup (two per SIMD: one hides the other's LDS / MFMA latencies)
#define LRS_NT (64 * LRS_NW)
#define LRS_TQ 5              // 16-row tiles per wave: 8 waves x 5 x 16 = 640 rows at most (the LDS allows ~576)
#define LRS_ZH 40             // halves per LDS row of Z^T (32 + 8: 16-byte aligned rows spread over the banks)
#define LRS_J 6               // partials / cells (16 bytes each) a thread polls at once

struct LrsArgs {
    int N, C, NPK, r, batch, nwg_t, zmod;
    int absd, u_in_packet, fuse_decode;
    // the geometry of a sum, worked out by the host (an integer division by a run-time value is ~40 instructions, a vector one more;
    // round 6: the index arithmetic of a sum cost ~0.4 us of the chain three times): cells a workgroup sums and threads / sub-lists
    // per cell, [0] without / [1] with the r x r matrix behind the values; 2^20 / cw rounded up (tid / cw by multiplication: exact for tid < 512);
    // log2 of the batch when the workgroup -> (tensor, index) map is tensors-interleaved
    int cpw[2], cw[2], subs[2], cwinv[2], batch_log2, nxs_log2;
    size_t offU16, offV16;
    char* arena;                     // [256 B: word 0 = launches since the arena was zeroed] then per tensor [partials of every slab | the sum]
    size_t arena_stride, offFull;    // bytes per tensor; offset of the sum inside a tensor's part
    unsigned* tick;                  // a zeroed ticket word: workgroups that have left
    unsigned* err;
    long long timeout;
    Probe probe;                     // developer build (cfx_dev.h): 16 words per workgroup, 100 MHz wall clock
};

// A word that changes hands: the fp32 value rounded to 22 bits of mantissa, its two lowest bits the sequence tag of the sum it belongs to.
__device__ __forceinline__ unsigned lrs_pack(float v, unsigned seq) { return ((__builtin_bit_cast(unsigned, v) + 2u) & ~3u) | seq; }
__device__ __forceinline__ float lrs_val(unsigned q) { return __builtin_bit_cast(float, q & ~3u); }
// 16-byte write-through store / L2-bypassing load (sc1).  hipcc does not count asm memory operations: nothing ever waits for the
// stores (the s_nop keeps the data registers alive until the store has read them), and a batch of loads is followed by lrs_wait
__device__ __forceinline__ void lrs_st16(void* p, lrs_u4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
typedef unsigned lrs_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lrs_st8(void* p, lrs_u2 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
// an fp64 value as two tagged words (hi, lo: ~44 bits of mantissa between them)
__device__ __forceinline__ lrs_u2 lrs_pack64(double d, unsigned seq) {
    lrs_u2 o;
    o[0] = lrs_pack((float)d, seq);
    o[1] = lrs_pack((float)(d - (double)__builtin_bit_cast(float, o[0] & ~3u)), seq);
    return o;
}
__device__ __forceinline__ double lrs_val64(unsigned hi, unsigned lo) { return (double)__builtin_bit_cast(float, hi & ~3u) + (double)__builtin_bit_cast(float, lo & ~3u); }
#define LRS_LD16(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(ptr) : "memory")
template <int J> __device__ __forceinline__ void lrs_wait(lrs_u4 (&q)[J]) {
    static_assert(J == 6 || J == 10, "operand lists below");
    if constexpr (J == 6)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5])::"memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]), "+v"(q[8]),
                     "+v"(q[9])::"memory");
}
__device__ __forceinline__ bool lrs_tagged(lrs_u4 q, unsigned seq) { return ((q[0] & q[1] & q[2] & q[3] & 3u) == seq) && (((q[0] | q[1] | q[2] | q[3]) & 3u) == seq); }

// chol(G) for the caller's substitution: L (fp32, row-major), 1 / diagonal, dead directions.  One wave.
template <int RP>
__device__ __forceinline__ void lrs_chol_L(const double (*G)[RP + 1], int r, float* Lf, float* dinvf, unsigned* deadw, double tol, bool own_diag) {
    const int i = threadIdx.x & 63;
    double g[RP], myinv;
    const unsigned dead = lr_chol_rows<RP>(G, r, tol, own_diag, g, myinv);
    if (i < RP) {
#pragma unroll
        for (int k = 0; k < RP; ++k) Lf[i * RP + k] = (float)g[k];
        dinvf[i] = ((dead >> i) & 1u) ? 0.f : (float)myinv;         // a dropped direction: zero column of the result
    }
    if (i == 0) *deadw = dead;
}

// Rank 32: the same factorisation BLOCKED (round 5).  lr_chol_rows updates the whole trailing matrix after every pivot: ~500 dependent
// v_readlane + v_fma_f64 pairs in one wave's registers, 10-12 us, twice a launch.  Here a panel of 4 columns is factorised in the lanes'
// registers (lane i = row i; the same step, the same masks, the same dead-direction rule as lr_chol_rows, restricted to the panel), and the
// rest of the matrix takes the panel's rank-4 update as ONE v_mfma_f64_16x16x4_f64 per 16 x 16 tile (D = C + (-Lp) Lp^T: the four
// products of an entry in the order the four steps would have applied them).  The matrix lives in LDS (`M`, 32 x 33 doubles) between panels;
// a wave's LDS operations execute in order, so the only waits are for data.  8 panels x (4 short steps + 3 tile updates).  One wave calls it.
__device__ __forceinline__ void lrs_chol_L_blocked32(const double (*G)[33], int r, float* Lf, float* dinvf, unsigned* deadw, double tol, bool own_diag,
                                                     double* M /* LDS: 32 x 33 */) {
    constexpr int RP = 32, MS = 33;
    const int lane = threadIdx.x & 63, i = lane & 31, l16 = lane & 15, lq = lane >> 4;
    auto bcast = [](double v, int src) -> double {
        const int2 q = __builtin_bit_cast(int2, v);
        int2 o;
        o.x = __builtin_amdgcn_readlane(q.x, src);
        o.y = __builtin_amdgcn_readlane(q.y, src);
        return __builtin_bit_cast(double, o);
    };
    // symmetrised copy (lanes 0 - 31: a row each), the original diagonal kept for the dead-direction test
    double gmax = 0.0;
#pragma unroll
    for (int k = 0; k < RP; ++k) gmax = (k < r) ? fmax(gmax, G[k][k]) : gmax;
    const double thr = gmax * tol;
    if (lane < RP) {
#pragma unroll
        for (int k = 0; k < RP; ++k) M[i * MS + k] = 0.5 * (G[i][k] + G[k][i]);
    }
    unsigned dead = 0;
    double myinv = 1.0;
#pragma unroll 1
    for (int p = 0; p < RP / 4; ++p) {
        const int j0 = 4 * p;
        double c[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) c[t] = M[i * MS + j0 + t];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = j0 + t;
            const double piv = bcast(c[t], j);
            const bool ok = (j < r) && (piv > (own_diag ? G[j][j] * tol : thr));
            const double pv = ok ? piv : 1.0;
            double inv = __builtin_amdgcn_rsq(pv);
            inv = inv * __builtin_fma(pv * inv, -0.5 * inv, 1.5);
            double l = ok ? c[t] * inv : (i == j ? 1.0 : 0.0);
            l = (i >= j && i < r) ? l : 0.0;
            myinv = (i == j && ok) ? inv : myinv;
            dead |= (ok || j >= r) ? 0u : (1u << j);
            c[t] = l;
#pragma unroll
            for (int t2 = t + 1; t2 < 4; ++t2) c[t2] = __builtin_fma(-l, bcast(l, j0 + t2), c[t2]);
        }
        if (lane < RP) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { Lf[i * RP + j0 + t] = (float)c[t]; M[i * MS + j0 + t] = c[t]; }     // (M's panel columns: the MFMA operands)
        }
        if (j0 + 4 >= RP) break;
        // trailing update, columns >= j0 + 4: tiles (a, b) = (rows 16 a .., columns 16 b ..) of the lower triangle that still have such columns
#pragma unroll
        for (int tl = 0; tl < 3; ++tl) {
            const int a = tl == 0 ? 0 : 1, b = tl == 2 ? 1 : 0;
            if (16 * b + 15 < j0 + 4) continue;                        // (uniform) every column of the tile is final already
            const double av = -M[(16 * a + l16) * MS + j0 + lq];       // A[i = l16][k = lq] = -Lp
            const double bv = M[(16 * b + l16) * MS + j0 + lq];        // B[k = lq][j = l16] = Lp^T
            f64x4 cv;
#pragma unroll
            for (int v = 0; v < 4; ++v) cv[v] = M[(16 * a + lq + 4 * v) * MS + 16 * b + l16];        // D[i = lq + 4 v][j = l16]
            cv = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, cv, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int col = 16 * b + l16;
                if (col >= j0 + 4) M[(16 * a + lq + 4 * v) * MS + col] = cv[v];                      // (the panel's own columns stay what the panel made them)
            }
        }
    }
    // rows above the diagonal of L: zero (lr_chol_rows writes the masked l there: 0)
    if (lane < RP) dinvf[i] = ((dead >> i) & 1u) ? 0.f : (float)myinv;
    if (lane == 0) *deadw = dead;
}

// LDS carve-up (bytes).  NPK = N rounded up to 32 (the K step of the fp16 MFMA); RR = RP rounded up to 16 (the rank groups of the MFMA
// tiles).  ONE N x r matrix lives in LDS at a time (Y0, W1, Y1, W2, U in turn, in place), fp32, transposed [rank][row]: the operand of
// the slab product reads 8 consecutive rows of a rank and splits them into fp16 hi + lo on the way; the fp64 Gram matrix, the
// substitution and the state update read it as it is.
template <int RP> struct LrsLds {
    static constexpr int RR = (RP + 15) / 16 * 16;
    __host__ __device__ static int nps(int NPK) { return NPK + 4; }                             // floats per row of the N x r matrix (banks)
    __host__ __device__ static int dt(int) { return 0; }                                        // slab^T  [32][NPK + 8] fp16
    __host__ __device__ static int yt(int NPK) { return 32 * (NPK + 8) * 2; }                   // the N x r matrix [RR][NPS] fp32
    __host__ __device__ static int sc(int NPK) { return yt(NPK) + RR * nps(NPK) * 4; }          // 16 KB of scratch: wave partials of the products
    __host__ __device__ static int zt(int NPK) { return sc(NPK) + 16384; }                      // Z^T hi | lo, [RR][LRS_ZH] fp16 each
    __host__ __device__ static int zf(int NPK) { return zt(NPK) + 2 * RR * LRS_ZH * 2; }        // Z [32][RR] fp32 (for its Gram matrix)
    __host__ __device__ static int ch(int NPK) { return zf(NPK) + 32 * RR * 4; }                // factorisation: G fp64, L fp32, 1 / diagonal, dead mask
    static constexpr int ch_bytes = (RP * (RP + 1) * 8 + RP * RP * 4 + RP * 4 + 16 + 15) / 16 * 16 + 64;    // (+ 2 x 8 wave maxima, norm_scale)
    __host__ __device__ static int total(int NPK) { return ch(NPK) + ch_bytes; }
};

template <int RP>
__global__ __launch_bounds__(LRS_NT) void k_lrs(LrBatch b, LrsArgs a) {
    typedef LrsLds<RP> L;
    constexpr int RR = L::RR, RG = RR / 16;                           // rank groups of 16 (the N dimension of an MFMA tile)
    const int bid = blockIdx.x;
    int z, idx;
    if (a.zmod) { z = bid & (a.batch - 1); idx = bid >> a.batch_log2; } else { z = bid / a.nwg_t; idx = bid - z * a.nwg_t; }
    const LrItem it = b.it[z];
    const int N = a.N, C = a.C, NPK = a.NPK, r = a.r, nwg = a.nwg_t;
    const int NPH = NPK + 8, NPS = NPK + 4, nmt = NPK / 16;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), l16 = lane & 15, lq = lane >> 4;     // w: scalar (uniform branches)
    extern __shared__ double lrs_smem[];
    char* sm = reinterpret_cast<char*>(lrs_smem);
    h16* Dt = reinterpret_cast<h16*>(sm + L::dt(NPK));
    float* Yt = reinterpret_cast<float*>(sm + L::yt(NPK));            // [RR][NPS]
    f32x4* red4 = reinterpret_cast<f32x4*>(sm + L::sc(NPK));
    double* scr64 = reinterpret_cast<double*>(sm + L::sc(NPK));
    h16* Zth = reinterpret_cast<h16*>(sm + L::zt(NPK));
    h16* Ztl = Zth + RR * LRS_ZH;
    float* Zf = reinterpret_cast<float*>(sm + L::zf(NPK));
    double (*Gd)[RP + 1] = reinterpret_cast<double (*)[RP + 1]>(sm + L::ch(NPK));
    float* Lf = reinterpret_cast<float*>(Gd + RP);                    // Cholesky factor (fp32, row-major), 1 / diagonal, dead directions
    float* dinvf = Lf + RP * RP;
    unsigned* deadw = reinterpret_cast<unsigned*>(dinvf + RP);
    lrs_u4* part = reinterpret_cast<lrs_u4*>(a.arena + 256 + (size_t)z * a.arena_stride);     // [slab][pcells] cells of 4 tagged words
    lrs_u4* full = reinterpret_cast<lrs_u4*>(a.arena + 256 + (size_t)z * a.arena_stride + a.offFull);      // [pcells]
    const unsigned launches = __hip_atomic_load(reinterpret_cast<const unsigned*>(a.arena), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned tag0 = launches * 3u + 1u;                         // sum n since the arena was zeroed carries tag (n + 1) mod 4
    // cells (4 words) of a partial: the N x r values row by row, then an r x r fp64 matrix as (hi, lo) word pairs
    const int cells = NPK * RP / 4, gcells = RP * RP / 2, pcells = cells + gcells;
    // the slab of this workgroup: with the round-robin dispatch of workgroups over the 8 XCDs, the workgroups of tensor z sit on the
    // XCDs = z mod batch; neighbouring slabs go to ONE of them, so that the two 64-byte halves of a 128-byte line of x and base are
    // asked for by the same L2 (a hint only: nothing depends on where a workgroup really runs)
    int slab = idx;
    if (a.nxs_log2 >= 0) { const int nx = 1 << a.nxs_log2; slab = (idx & (nx - 1)) * (nwg >> a.nxs_log2) + (idx >> a.nxs_log2); }      // (host: nwg is a multiple of nx)
    const int c0 = slab * LRS_SW;
    const Probe probe = a.probe.of(bid);
#define LSTAMP(k) probe.at(k)
    LSTAMP(0);

    // ---------------- the slab: registers (rows t * 16 + l16 of tile t = w + LRS_NW q, columns c0 + 8 lq .. + 7) and LDS (transposed) ----------------
    h16x8 ds[LRS_TQ], bs[LRS_TQ];
    {
        h16x8 xv[LRS_TQ];
#pragma unroll
        for (int q = 0; q < LRS_TQ; ++q) {                              // every load unconditional (clamped row): all in flight at once
            const int row = min((w + LRS_NW * q) * 16 + l16, N - 1);
            const size_t o = (size_t)row * C + c0 + 8 * lq;
            xv[q] = __builtin_nontemporal_load(reinterpret_cast<const h16x8*>(it.x + o));
            bs[q] = (h16x8)(h16)0;
            if (it.base) bs[q] = __builtin_nontemporal_load(reinterpret_cast<const h16x8*>(it.base + o));
        }
        // meanwhile: zero what must read as zero (ranks >= RP of the N x r matrix and of Z^T, rows >= N of the transposed arrays)
        for (int i = tid; i < (L::sc(NPK) - L::dt(NPK)) / 16; i += LRS_NT) reinterpret_cast<lrs_u4*>(sm)[i] = (lrs_u4)0u;
        for (int i = tid; i < 2 * RR * LRS_ZH * 2 / 16; i += LRS_NT) reinterpret_cast<lrs_u4*>(Zth)[i] = (lrs_u4)0u;
        __syncthreads();
        for (int e = tid; e < LRS_SW * RP; e += LRS_NT) {               // Q0 rows of the slab -> Z^T as hi + lo
            const int wc = e / RP, k = e - wc * RP;
            const float v = it.q0[(size_t)(c0 + wc) * RP + k];
            const h16 hi = (h16)v;
            Zth[k * LRS_ZH + wc] = hi;
            Ztl[k * LRS_ZH + wc] = (h16)(v - (float)hi);
        }
#pragma unroll
        for (int q = 0; q < LRS_TQ; ++q) {
            const int t = w + LRS_NW * q, row = t * 16 + l16;
            h16x8 d = xv[q] - bs[q];                                  // fp16, one rounding (torch eager: x - base); base absent: x - 0 = x
            if (a.absd) {
                typedef unsigned short u16x8_ __attribute__((ext_vector_type(8)));
                u16x8_ bb = __builtin_bit_cast(u16x8_, d);
                bb &= (unsigned short)0x7fff;
                d = __builtin_bit_cast(h16x8, bb);
            }
            if (row >= N) d = (h16x8)(h16)0;
            ds[q] = d;
            if (t < nmt) {
#pragma unroll
                for (int e = 0; e < 8; ++e) Dt[(8 * lq + e) * NPH + row] = d[e];
            }
        }
    }
    __syncthreads();
    LSTAMP(1);

    // Wp = slab Z as [n][RP] fp32 partial of this workgroup (write-through): D[i = rank][j = row], a lane holds 4 consecutive ranks
    auto product_b = [&](unsigned tag) {
        h16x8 zh[RG], zl[RG];
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            zh[g] = *reinterpret_cast<const h16x8*>(&Zth[(16 * g + l16) * LRS_ZH + 8 * lq]);
            zl[g] = *reinterpret_cast<const h16x8*>(&Ztl[(16 * g + l16) * LRS_ZH + 8 * lq]);
        }
        lrs_u4* P = part + (size_t)slab * pcells;
        const unsigned seq = tag & 3u;
#pragma unroll
        for (int q = 0; q < LRS_TQ; ++q) {
            const int t = w + LRS_NW * q;
            if (t < nmt) {
#pragma unroll
                for (int g = 0; g < RG; ++g) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(zh[g], ds[q], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(zl[g], ds[q], acc, 0, 0, 0);
                    if (16 * g + 4 * lq < RP) {
                        lrs_u4 o;
#pragma unroll
                        for (int v = 0; v < 4; ++v) o[v] = lrs_pack(acc[v], seq);
                        lrs_st16(&P[(size_t)(t * 16 + l16) * (RP / 4) + 4 * g + lq], o);
                    }
                }
            }
        }
    };

    // The iteration does not care about the scale of Y (every factor renormalises), but the fp16 operands do: Y0 = A Q0 grows with
    // sigma, Z = A^T Y with sigma^2 - a residual with entries of a few units at the FLUX shard puts Z past 65504.  Before a product the
    // N x r matrix is therefore scaled by the power of two that brings its largest entry into [0.5, 1) - exact, the same in every
    // workgroup (all hold the same matrix), and folded into the operand conversion below.  |Z| <= N max|A| then: in range for any
    // residual with entries below ~100.
    // The largest entry is tracked by whoever writes the matrix (the gather of a sum, the substitution): run_max is each thread's share.
    float ysc = 1.f, run_max = 0.f;
    float* mxs = reinterpret_cast<float*>(sm + L::ch(NPK) + L::ch_bytes - 64);
    auto norm_scale = [&](int which) {
        float m = run_max;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float* mx = mxs + which * LRS_NW;                             // its own words per call site: nothing to wait for afterwards
        if (lane == 0) mx[w] = m;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LRS_NW; ++k) m = fmaxf(m, mx[k]);
        int e = 0;
        (void)frexpf(m, &e);
        ysc = (m > 0.f && m < 3.0e38f) ? ldexpf(1.f, -e) : 1.f;
    };

    // Z = slab^T Y (32 x RP): 2 RG output tiles (16 columns of the slab x 16 ranks), each wave one tile over its share of K = N; the
    // shares are summed through LDS in wave order.  The B operand - 8 consecutive rows of a rank - is split into fp16 hi + lo on the
    // way from the fp32 matrix.  final_v: the result is V (fp16, into LDS [rank][column] and into the packet / the workspace);
    // otherwise Z^T as hi + lo for product_b, and - want_gram - this slab's share of Z^T Z behind its partial
    auto product_a = [&](bool final_v, bool want_gram, unsigned tag) {
        constexpr int T = 2 * RG, KS = LRS_NW / T;                     // tiles, K shares per tile
        const int tile = w % T, mt = tile & 1, g = tile >> 1, ksh = w / T;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int nk32 = NPK / 32;
#pragma unroll
        for (int i = 0; i < (LRS_TQ * LRS_NW / 2 + KS - 1) / KS; ++i) {      // K steps of this wave, operands loaded unconditionally
            const int ks = ksh + KS * i;
            const int n0 = min(ks, nk32 - 1) * 32 + 8 * lq;
            const h16x8 av = *reinterpret_cast<const h16x8*>(&Dt[(16 * mt + l16) * NPH + n0]);
            const f32x4 y0 = *reinterpret_cast<const f32x4*>(&Yt[(16 * g + l16) * NPS + n0]) * ysc;
            const f32x4 y1 = *reinterpret_cast<const f32x4*>(&Yt[(16 * g + l16) * NPS + n0 + 4]) * ysc;
            h16x8 bh, bl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bh[e] = (h16)y0[e]; bl[e] = (h16)(y0[e] - (float)bh[e]);
                bh[4 + e] = (h16)y1[e]; bl[4 + e] = (h16)(y1[e] - (float)bh[4 + e]);
            }
            if (ks >= nk32) { bh = (h16x8)(h16)0; bl = (h16x8)(h16)0; }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bl, acc, 0, 0, 0);
        }
        red4[w * 64 + lane] = acc;
        __syncthreads();
        if (tid < 64 * T) {
            const int tl = tid >> 6, ln = tid & 63, mt2 = tl & 1, g2 = tl >> 1;
            f32x4 s = red4[tl * 64 + ln];
#pragma unroll
            for (int k = 1; k < KS; ++k) s += red4[(k * T + tl) * 64 + ln];
            const int rr = 16 * g2 + (ln & 15), w0 = 16 * mt2 + 4 * (ln >> 4);      // D[i = column of the slab][j = rank]
            if (!final_v) {
                h16x4 hi, lo;
#pragma unroll
                for (int v = 0; v < 4; ++v) { hi[v] = (h16)s[v]; lo[v] = (h16)(s[v] - (float)hi[v]); }
                *reinterpret_cast<h16x4*>(&Zth[rr * LRS_ZH + w0]) = hi;
                *reinterpret_cast<h16x4*>(&Ztl[rr * LRS_ZH + w0]) = lo;
                if (want_gram) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) Zf[(w0 + v) * RR + rr] = s[v];
                }
            } else {
                h16x4 v16;
#pragma unroll
                for (int v = 0; v < 4; ++v) v16[v] = (rr < r) ? (h16)s[v] : (h16)0;
                *reinterpret_cast<h16x4*>(&Zth[rr * LRS_ZH + w0]) = v16;
                if (rr < r) {
                    if (a.u_in_packet && ((N * r) & 3) == 0) *reinterpret_cast<h16x4*>(&((h16*)it.packet)[(size_t)N * r + (size_t)rr * C + c0 + w0]) = v16;
                    else if (a.u_in_packet) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) ((h16*)it.packet)[(size_t)N * r + (size_t)rr * C + c0 + w0 + v] = v16[v];
                    } else {
#pragma unroll
                        for (int v = 0; v < 4; ++v) ((h16*)(it.ws + a.offV16))[(size_t)(c0 + w0 + v) * r + rr] = v16[v];
                    }
                }
            }
        }
        __syncthreads();
        if (want_gram && w >= LRS_NW - RG * RG) {
            // this slab's share of Z^T Z (= Y^T W of the sum that follows: W = A Z, Z = A^T Y), fp64 from the fp32 Z: exact products; one
            // 16 x 16 tile a wave.  It travels behind the partial, every entry as a (hi, lo) pair of tagged words
            const int tg = LRS_NW - 1 - w, gi = tg / RG, gj = tg - gi * RG;
            f64x4 acc2 = {0.0, 0.0, 0.0, 0.0};
            double za[8], zb[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) { za[ks] = (double)Zf[(4 * ks + lq) * RR + 16 * gi + l16]; zb[ks] = (double)Zf[(4 * ks + lq) * RR + 16 * gj + l16]; }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(za[ks], zb[ks], acc2, 0, 0, 0);
            unsigned* G2 = reinterpret_cast<unsigned*>(part + (size_t)slab * pcells + cells);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ri = 16 * gi + lq + 4 * v, rj = 16 * gj + l16;      // D[i = lq + 4 v][j = l16]
                if (ri < RP && rj < RP) lrs_st8(&G2[2 * (ri * RP + rj)], lrs_pack64(acc2[v], tag & 3u));
            }
        }
    };

    // Sum of the partials over the slabs into the N x r matrix (LDS, [rank][row] fp32): this workgroup's share of the cells over all
    // partials (fixed order), published; then everybody polls the whole result
    // A wait that gives up (a.timeout ticks of the 100 MHz wall clock after its first failed poll) ends the chain for this workgroup: the
    // phases that follow are skipped, nothing more is stored - no factors from sums that never completed, the state stays as it was.
    bool failed = false;
    // (rank 32, 10 cells a thread in flight instead of 6 - one round trip instead of two for the gather and for a workgroup's share of the
    // reduction: 95.7 vs 95.2 us, nothing; a sum there moves 210 KB per workgroup through sc1 loads and stores, that is what it takes)
    constexpr int JJ = LRS_J;
    auto allreduce = [&](unsigned tag, bool with_gram) {
        const unsigned seq = tag & 3u;
        const int tcells = with_gram ? pcells : cells;                // with_gram: the r x r fp64 matrix behind the values is summed (in fp64) too -> Gd
        const int gi = with_gram ? 1 : 0;
        const int cpw = a.cpw[gi], cw = a.cw[gi], subs = a.subs[gi];
        const int sub = (int)(((unsigned)tid * (unsigned)a.cwinv[gi]) >> 20), ci = tid - sub * cw;
        for (int cb = 0; cb < cpw; cb += cw) {
            const int cl = cb + ci, cell = idx * cpw + cl;
            const bool act = sub < subs && cl < cpw && cell < tcells;
            const bool dbl = cell >= cells;                           // a cell of the fp64 matrix: two (hi, lo) pairs
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            double d0 = 0.0, d1 = 0.0;
            if (act) {
                const lrs_u4* src = part + cell;
                for (int p0 = sub; p0 < nwg; p0 += subs * JJ) {    // slab order: the result does not depend on the batch
                    lrs_u4 q[JJ];
                    long long t0 = 0;
                    int nfail = 0;
                    for (;;) {
#pragma unroll
                        for (int j = 0; j < JJ; ++j) LRS_LD16(q[j], src + (size_t)min(p0 + subs * j, nwg - 1) * pcells);     // unconditional: one round trip
                        lrs_wait(q);
                        bool ok = true;
#pragma unroll
                        for (int j = 0; j < JJ; ++j) ok = ok && lrs_tagged(q[j], seq);
                        if (ok) break;
                        if (failed) break;
                        if ((++nfail & 7) == 0) {                     // (the clock is a scalar memory read: not after every failed poll)
                            const long long now = wall_clock64();
                            if (!t0) t0 = now;
                            else if (now - t0 > a.timeout) { failed = true; break; }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < JJ; ++j)
                        if (p0 + subs * j < nwg) {
                            if (dbl) { d0 += lrs_val64(q[j][0], q[j][1]); d1 += lrs_val64(q[j][2], q[j][3]); }
                            else {
#pragma unroll
                                for (int k = 0; k < 4; ++k) s[k] += lrs_val(q[j][k]);
                            }
                        }
                }
            }
            if (dbl) { const double dd[2] = {d0, d1}; s = __builtin_bit_cast(f32x4, dd); }
            if (sub < subs) red4[sub * cw + ci] = s;
            __syncthreads();
            if (sub == 0 && act) {
                lrs_u4 o;
                if (dbl) {
                    double t0d = 0.0, t1d = 0.0;
                    for (int s2 = 0; s2 < subs; ++s2) {
                        const f32x4 rv = red4[s2 * cw + ci];
                        double dd[2];
                        __builtin_memcpy(dd, &rv, 16);
                        t0d += dd[0]; t1d += dd[1];
                    }
                    const lrs_u2 a0 = lrs_pack64(t0d, seq), a1 = lrs_pack64(t1d, seq);
                    o[0] = a0[0]; o[1] = a0[1]; o[2] = a1[0]; o[3] = a1[1];
                } else {
                    f32x4 t = red4[ci];
                    for (int s2 = 1; s2 < subs; ++s2) t += red4[s2 * cw + ci];
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = lrs_pack(t[k], seq);
                }
                lrs_st16(&full[cell], o);
            }
            __syncthreads();
        }
        if (with_gram) LSTAMP(12);
        for (int i0 = tid; i0 < tcells; i0 += LRS_NT * JJ) {
            lrs_u4 q[JJ];
            long long t0 = 0;
            int nfail = 0;
            for (;;) {
#pragma unroll
                for (int j = 0; j < JJ; ++j) LRS_LD16(q[j], full + min(i0 + LRS_NT * j, tcells - 1));
                lrs_wait(q);
                bool ok = true;
#pragma unroll
                for (int j = 0; j < JJ; ++j) ok = ok && lrs_tagged(q[j], seq);
                if (ok) break;
                if (failed) break;
                if ((++nfail & 7) == 0) {                     // (the clock is a scalar memory read: not after every failed poll)
                    const long long now = wall_clock64();
                    if (!t0) t0 = now;
                    else if (now - t0 > a.timeout) { failed = true; break; }
                }
            }
#pragma unroll
            for (int j = 0; j < JJ; ++j) {
                const int c = i0 + LRS_NT * j;
                if (c < cells) {
                    const int n = c / (RP / 4), rq = c - n * (RP / 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const float v = lrs_val(q[j][k]); Yt[(4 * rq + k) * NPS + n] = v; run_max = fmaxf(run_max, fabsf(v)); }
                } else if (c < tcells) {
                    const int e = 2 * (c - cells);
                    Gd[e / RP][e % RP] = lrs_val64(q[j][0], q[j][1]);
                    Gd[(e + 1) / RP][(e + 1) % RP] = lrs_val64(q[j][2], q[j][3]);
                }
            }
        }
        failed = __syncthreads_or(failed ? 1 : 0) != 0;              // (uniform from here on)
    };

    // P = W^T W into Gd: fp64 from the fp32 values (exact products, fp64 sums).  The waves split K = N; a wave loads its rows of the
    // RG rank groups once and feeds the RG (RG + 1) / 2 tiles of the upper triangle (16 x 16 each) from them; the waves' shares are
    // summed through LDS in wave order, a tile at a time (the scratch holds one tile of every wave), the lower triangle is the mirror
    auto gram64 = [&]() {
        constexpr int NTL = RG * (RG + 1) / 2;
        f64x4 am[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t) am[t] = f64x4{0.0, 0.0, 0.0, 0.0};
        const int nks = NPK / 4;
        constexpr int UN = RG == 1 ? 10 : 6;                          // K steps whose operands are in flight at once
        for (int i0 = 0; w + LRS_NW * i0 < nks; i0 += UN) {
            float av[UN][RG];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int ks = w + LRS_NW * (i0 + u);
                const int n = min(ks, nks - 1) * 4 + lq;
#pragma unroll
                for (int g = 0; g < RG; ++g) av[u][g] = (ks < nks) ? Yt[(16 * g + l16) * NPS + n] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                int t = 0;
#pragma unroll
                for (int gi = 0; gi < RG; ++gi)
#pragma unroll
                    for (int gj = gi; gj < RG; ++gj, ++t)
                        am[t] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[u][gi], (double)av[u][gj], am[t], 0, 0, 0);
            }
        }
        {
            int t = 0;
#pragma unroll
            for (int gi = 0; gi < RG; ++gi)
#pragma unroll
                for (int gj = gi; gj < RG; ++gj, ++t) {
                    // D[i][j]: j = lane & 15, i = (lane >> 4) + 4 v  (the fp64 form's own map)
#pragma unroll
                    for (int v = 0; v < 4; ++v) scr64[(w * 64 + lane) * 4 + v] = am[t][v];
                    __syncthreads();
                    if (tid < 256) {
                        const int ln = tid & 63, v = tid >> 6;
                        const int i = 16 * gi + (ln >> 4) + 4 * v, jj = 16 * gj + (ln & 15);
                        double m = 0.0;
#pragma unroll
                        for (int k = 0; k < LRS_NW; ++k) m += scr64[(k * 64 + ln) * 4 + v];
                        if (i < RP && jj < RP) { Gd[i][jj] = m; if (gi != gj) Gd[jj][i] = m; }
                    }
                    __syncthreads();
                }
        }
    };

    // Y = W L^-T in place, one row per thread by forward substitution (fp32; the factor's entries are LDS broadcasts, a row of the
    // factor read once as 16-byte pieces)
    auto apply_l = [&]() {
        run_max = 0.f;
        if (RP <= 8) {
            // rows tid and tid + LRS_NT in ONE pass (NPK <= 2 LRS_NT): the factor's rows are read once for both
            const int n0 = tid, n1 = tid + LRS_NT;
            const bool h0 = n0 < NPK, h1 = n1 < NPK;
            float wa[RP], wb[RP], ya[RP], yb[RP];
#pragma unroll
            for (int k = 0; k < RP; ++k) { wa[k] = Yt[k * NPS + (h0 ? n0 : 0)]; wb[k] = Yt[k * NPS + (h1 ? n1 : 0)]; }
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                float lrow[(RP + 3) / 4 * 4];
#pragma unroll
                for (int c = 0; c < (j + 3) / 4; ++c) *reinterpret_cast<f32x4*>(&lrow[4 * c]) = *reinterpret_cast<const f32x4*>(&Lf[j * RP + 4 * c]);
                const float dj = dinvf[j];                                // (0 for a dropped direction: zero column)
                float sa = wa[j], sb = wb[j];
#pragma unroll
                for (int k = 0; k < j; ++k) { sa = fmaf(-ya[k], lrow[k], sa); sb = fmaf(-yb[k], lrow[k], sb); }
                ya[j] = sa * dj;
                yb[j] = sb * dj;
            }
#pragma unroll
            for (int k = 0; k < RP; ++k) {
                if (h0) { Yt[k * NPS + n0] = ya[k]; run_max = fmaxf(run_max, fabsf(ya[k])); }
                if (h1) { Yt[k * NPS + n1] = yb[k]; run_max = fmaxf(run_max, fabsf(yb[k])); }
            }
        } else {
            // rank 16 / 32: one row at a time (two rows' values and the factor's entries do not fit the registers of 8 waves); the second
            // pass - rows >= LRS_NT - only exists in wave 0
            for (int n = tid; n < NPK; n += LRS_NT) {
                float y[RP];                                          // W's row, replaced entry by entry
                // (the row stride is a run-time value: walking a pointer keeps ONE address in registers - computed up front, the RP
                // addresses were spilled at rank 32)
                const float* rp = Yt + n;
#pragma unroll
                for (int k = 0; k < RP; ++k) { y[k] = *rp; rp += NPS; asm volatile("" : "+v"(rp)); }
#pragma unroll
                for (int j = 0; j < RP; ++j) {
                    float lrow[(RP + 3) / 4 * 4];
#pragma unroll
                    for (int c = 0; c < (j + 3) / 4; ++c) *reinterpret_cast<f32x4*>(&lrow[4 * c]) = *reinterpret_cast<const f32x4*>(&Lf[j * RP + 4 * c]);
                    float sa = y[j];
#pragma unroll
                    for (int k = 0; k < j; ++k) sa = fmaf(-y[k], lrow[k], sa);
                    y[j] = sa * dinvf[j];
                    if (RP > 16 || (j & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // the factor's rows in flight: four (rank 16), one (rank 32)
                }
                float* wp = Yt + n;
#pragma unroll
                for (int k = 0; k < RP; ++k) { *wp = y[k]; run_max = fmaxf(run_max, fabsf(y[k])); wp += NPS; asm volatile("" : "+v"(wp)); }
            }
        }
        __syncthreads();
    };

    // ---------------- Y0 = A Q0 ----------------
    product_b(tag0);
    LSTAMP(2);
    run_max = 0.f;
    allreduce(tag0, false);
    if (!failed) {
    norm_scale(0);
    LSTAMP(3);

    // ---------------- W1 = A (A^T Y0), Y1 = W1 chol(M1)^-T ----------------
    product_a(false, true, tag0 + 1);                                 // also: this slab's share of M1 = Z1^T Z1 (= Y0^T W1), behind the partial
    product_b(tag0 + 1);
    LSTAMP(4);
    allreduce(tag0 + 1, true);                                        // W1 and M1
    LSTAMP(5);
    }
    if (!failed) {
    // (Round 5 tried the factorisation over the WHOLE workgroup - thread t holding entries t and t + 512, column j through LDS, one barrier a
    // step, bit-identical results: k_lrs<32> went from 238 to 145 VGPRs and from 99 to 117 us, rank 16 from 53 to 61 - a barrier of eight
    // waves per step costs more than one wave's v_readlane chain.  Out again; so is two rows per lane in wave 0's substitution, which
    // spilled 2 KB a lane.)
    if (w == 0) {
        if constexpr (RP == 32) lrs_chol_L_blocked32(Gd, r, Lf, dinvf, deadw, LRS_PIVOT_TOL, false, scr64);
        else lrs_chol_L<RP>(Gd, r, Lf, dinvf, deadw, LRS_PIVOT_TOL, false);     // chol(M1)
    }
    __syncthreads();
    LSTAMP(14);
    apply_l();                                                        // Y1 = W1 chol(M1)^-T
    norm_scale(1);
    LSTAMP(6);

    // ---------------- W2 = A (A^T Y1), U = W2 chol(W2^T W2)^-T ----------------
    product_a(false, false, 0u);
    product_b(tag0 + 2);
    LSTAMP(7);
    allreduce(tag0 + 2, false);
    LSTAMP(8);
    }
    if (!failed) {
    // The N-space chain's T2 T3 (T2 = chol(Y1^T W2)^-T, T3 = chol(T2^T P T2)^-T, P = W2^T W2) is upper triangular and makes W2
    // orthonormal, so it IS the inverse Cholesky factor of P: one factorisation.  W2's columns are graded (norms ~ sigma^3), P's
    // pivots span ~ sigma^6 - Cholesky without pivoting does not mind (its error goes with the condition of the matrix scaled to unit
    // diagonal: orthogonality of U 1e-9 .. 1e-5 for sigma_r / sigma_1 down to 3e-5, checked in fp64), but "small against the largest
    // diagonal entry" is the wrong test for a dead direction here: a pivot is compared with its OWN diagonal entry (legitimate
    // directions: > 1e-8 of it; the null directions of a rank-deficient residual: < 1e-10)
    gram64();
    LSTAMP(13);
    if (w == 0) {
        if constexpr (RP == 32) lrs_chol_L_blocked32(Gd, r, Lf, dinvf, deadw, LRS_PIVOT_REL, true, scr64);
        else lrs_chol_L<RP>(Gd, r, Lf, dinvf, deadw, LRS_PIVOT_REL, true);
    }
    __syncthreads();
    LSTAMP(9);
    apply_l();                                                        // U, fp32 in place: its fp16 rounding is what the packet carries
    ysc = 1.f;                                                        // (orthonormal columns: no scale - V = U^T A is wanted as it is)
    if (idx == 0) {
        h16* U16g = a.u_in_packet ? (h16*)it.packet : (h16*)(it.ws + a.offU16);
        for (int n = tid; n < N; n += LRS_NT)                           // (a row a thread: no division per element)
            for (int m = 0; m < r; ++m) U16g[(size_t)n * r + m] = (h16)Yt[m * NPS + n];
    }
    LSTAMP(10);

    // ---------------- V = U^T A for the slab's columns, state update of the slab's columns ----------------
    product_a(true, false, 0u);
    if (a.fuse_decode) {
        // new_base[:, slab] = base + fp16(U V): the arithmetic of k_lr_decode (v_dot2 chain over the k-pairs in order, one rounding
        // to fp16, one fp16 add), from the registers base was loaded into at the start; 8 k-pairs at a time (registers)
        float acc[LRS_TQ][8];
#pragma unroll
        for (int q = 0; q < LRS_TQ; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[q][i] = 0.f;
#pragma unroll
        for (int kh = 0; kh < (RP / 2 + 7) / 8; ++kh) {
            h16x2 vp[8][8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k2 = 8 * kh + kk;
                    vp[kk][i][0] = (2 * k2 < RR) ? Zth[(2 * k2) * LRS_ZH + 8 * lq + i] : (h16)0;
                    vp[kk][i][1] = (2 * k2 + 1 < RR) ? Zth[(2 * k2 + 1) * LRS_ZH + 8 * lq + i] : (h16)0;
                }
#pragma unroll
            for (int q = 0; q < LRS_TQ; ++q) {
                const int row = (w + LRS_NW * q) * 16 + l16;
                if (row < N) {
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        const int k2 = 8 * kh + kk;
                        if (2 * k2 < r && 2 * k2 < RP) {
                            h16x2 ua;
                            ua[0] = (h16)Yt[(2 * k2) * NPS + row];
                            ua[1] = (h16)Yt[(2 * k2 + 1) * NPS + row];
#pragma unroll
                            for (int i = 0; i < 8; ++i) acc[q][i] = __builtin_amdgcn_fdot2(ua, vp[kk][i], acc[q][i], false);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < LRS_TQ; ++q) {
            const int row = (w + LRS_NW * q) * 16 + l16;
            if (row < N) {
                h16x8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (h16)acc[q][i];
                if (it.base) o = bs[q] + o;
                __builtin_nontemporal_store(o, reinterpret_cast<h16x8*>(it.new_base + (size_t)row * C + c0 + 8 * lq));
            }
        }
    }
    LSTAMP(11);
    }
    if (failed && tid == 0 && a.err) (void)__hip_atomic_fetch_add(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#undef LSTAMP
    // ---------------- leave: the last workgroup of the launch resets the ticket and moves the arena's launch count on ----------------
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)(nwg * a.batch) - 1) {
            __hip_atomic_store(a.tick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned*>(a.arena), launches + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static inline int lrs_npk(int N) { return (N + 31) / 32 * 32; }

bool cfx_i_lrs_ok(int N, int C, int RP) {
    if (N < 32 || N > 16 * LRS_NW * LRS_TQ || (C % 128) != 0 || C < 512 || RP > 32) return false;
    const int npk = lrs_npk(N);
    return (RP == 8 ? LrsLds<8>::total(npk) : (RP == 16 ? LrsLds<16>::total(npk) : LrsLds<32>::total(npk))) <= 160 * 1024;
}

size_t cfx_i_lrs_extra_bytes(int, int, int) { return 0; }       // nothing in the caller's workspace: the hand-over arena is the context's

// bytes of one tensor's part of the arena: the partials of every slab, then the sum (4-byte tagged words)
static size_t lrs_tensor_bytes(int N, int C, int RP, size_t* off_full) {
    const size_t npk = lrs_npk(N);
    const size_t pbytes = npk * RP * 4 + (size_t)RP * RP * 8;           // a partial: the N x r values, then an r x r fp64 matrix as word pairs
    const size_t part = al256((size_t)(C / LRS_SW) * pbytes);
    if (off_full) *off_full = part;
    return part + al256(pbytes);
}

// The arena of `stream` laid out for this shape (allocated / zeroed as needed; the zeroing is stream-ordered before the launch).
static char* lrs_arena(cfx_ctx* ctx, void* stream, int N, int C, int RP, int nb, size_t* stride, size_t* off_full) {
    *stride = lrs_tensor_bytes(N, C, RP, off_full);
    const size_t need = 256 + *stride * nb;
    const unsigned long long key = ((unsigned long long)N << 40) ^ ((unsigned long long)C << 16) ^ ((unsigned long long)RP << 8) ^ (unsigned long long)nb;
    int slot = -1;
    for (int i = 0; i < ctx->lrs_n; ++i)
        if (ctx->lrs_stream[i] == stream) { slot = i; break; }
    if (slot < 0) {
        if (ctx->lrs_n < 8) slot = ctx->lrs_n++;
        else {
            slot = (int)(ctx->lrs_next++ % 8);                         // every slot taken: the oldest changes hands - once whatever its
            (void)hipDeviceSynchronize();                              // previous owner had in flight on it has finished
        }
        ctx->lrs_stream[slot] = stream;
        ctx->lrs_key[slot] = ~0ull;
    }
    if (ctx->lrs_bytes[slot] < need) {
        if (ctx->lrs_arena[slot]) { (void)hipDeviceSynchronize(); (void)hipFree(ctx->lrs_arena[slot]); }
        ctx->lrs_arena[slot] = nullptr;
        ctx->lrs_bytes[slot] = 0;
        void* p = nullptr;
        if (hipMalloc(&p, need) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        ctx->lrs_arena[slot] = (char*)p;
        ctx->lrs_bytes[slot] = need;
        ctx->lrs_key[slot] = ~0ull;
    }
    if (ctx->lrs_key[slot] != key) {
        // another layout: what the words hold no longer says anything about the tags - start again from zero (tag 0, count 0)
        if (hipMemsetAsync(ctx->lrs_arena[slot], 0, need, (hipStream_t)stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        ctx->lrs_key[slot] = key;
    }
    return ctx->lrs_arena[slot];
}

template <int RP>
static int lrs_run(cfx_ctx* ctx, const LrBatch& b, LrsArgs a, hipStream_t s) {
    const size_t lds = (size_t)LrsLds<RP>::total(a.NPK);
    static size_t attr_bytes = 0;
    if (lds > attr_bytes) {
        if (hipFuncSetAttribute((const void*)k_lrs<RP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return fail(ctx, CFX_ERR_LAUNCH, "low-rank: the device does not grant the LDS the slab-resident chain needs");
        }
        attr_bytes = lds;
    }
    a.zmod = (a.batch <= 8 && 8 % a.batch == 0) ? 1 : 0;
    a.batch_log2 = a.batch == 8 ? 3 : (a.batch == 4 ? 2 : (a.batch == 2 ? 1 : 0));
    a.nxs_log2 = (a.zmod && a.nwg_t % (8 / a.batch) == 0) ? 3 - a.batch_log2 : -1;
    {
        const int cells = a.NPK * RP / 4, pcells = cells + RP * RP / 2;
        for (int gi = 0; gi < 2; ++gi) {
            const int tcells = gi ? pcells : cells;
            a.cpw[gi] = (tcells + a.nwg_t - 1) / a.nwg_t;
            a.cw[gi] = a.cpw[gi] < LRS_NT ? a.cpw[gi] : LRS_NT;
            a.subs[gi] = LRS_NT / a.cw[gi];
            a.cwinv[gi] = ((1 << 20) + a.cw[gi] - 1) / a.cw[gi];     // floor(t / cw) = (t * cwinv) >> 20: exact for every cw <= 512, t < 512
        }
    }
    // Two of these launches in flight at once - from two streams - could each hold only a part of the CUs and wait for workgroups
    // that find no room.  Launches of ONE stream are in order; when the stream changes, the new one first waits for everything the
    // previous one has been given so far (an event recorded there now: nothing is paid while a context keeps to one stream).
    if (ctx->lrs_last_stream != (void*)s) {
        if (ctx->lrs_last_stream || ctx->lrs_ev) {
            if (!ctx->lrs_ev && hipEventCreateWithFlags(&ctx->lrs_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx->lrs_ev = nullptr; }
            if (ctx->lrs_ev && (hipEventRecord(ctx->lrs_ev, (hipStream_t)ctx->lrs_last_stream) != hipSuccess ||
                                hipStreamWaitEvent(s, ctx->lrs_ev, 0) != hipSuccess))
                (void)hipGetLastError();       // (a capturing stream cannot wait for an event outside its graph: the caller serialises)
        } else if (!ctx->lrs_ev && hipEventCreateWithFlags(&ctx->lrs_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx->lrs_ev = nullptr; }
        ctx->lrs_last_stream = (void*)s;
    }
    LAUNCH(ctx, KID_LR_CHAIN, s, (k_lrs<RP>), dim3((unsigned)(a.nwg_t * a.batch)), dim3(LRS_NT), lds, s, b, a);
    return check_launch(ctx, "low-rank (slab-resident chain)");
}

// Tensors of a batch one launch can take: every workgroup must be resident at once (one per CU: the LDS), so C / 32 per tensor
// out of the CUs the stream's queue may use; 0: this stream cannot run the form at all.
int cfx_i_lrs_fit(cfx_ctx* ctx, int N, int C, int RP, void* stream) {
    if (!cfx_i_lrs_ok(N, C, RP)) return 0;
    const int cap = cfx_i_stream_cus(ctx, stream) - 2;                // a little room for whatever else is on the device
    const int nb = cap / (C / LRS_SW);
    return nb > LR_MAXB ? LR_MAXB : nb;
}

// Factors of every tensor of the batch (see cfx_i_lrg_factors) by the slab-resident chain.  `extra` = offset of
// cfx_i_lrs_extra_bytes() bytes inside each tensor's workspace.
int cfx_i_lrs_factors(cfx_ctx* ctx, int quantized, int N, int C, int rank, int batch, const LrBatch& b, size_t offU16, size_t offV16,
                      size_t extra, int absd, int want_decode, int* decoded, hipStream_t s) {
    const int RPv = lr_rp(rank);
    if (decoded) *decoded = 0;
    const int fit = cfx_i_lrs_fit(ctx, N, C, RPv, (void*)s);
    if (fit < 1) return fail(ctx, CFX_ERR_LAUNCH, "low-rank: the slab-resident chain does not fit the stream's CUs");
    const size_t npk = lrs_npk(N);
    (void)extra;
    for (int first = 0; first < batch; first += fit) {
        const int nb = batch - first < fit ? batch - first : fit;
        LrBatch bb;
        memset(&bb, 0, sizeof(bb));
        for (int i = 0; i < nb; ++i) bb.it[i] = b.it[first + i];
        LrsArgs a;
        memset(&a, 0, sizeof(a));
        a.N = N; a.C = C; a.NPK = (int)npk; a.r = rank; a.batch = nb; a.nwg_t = C / LRS_SW;
        a.absd = absd; a.u_in_packet = quantized ? 0 : 1;
        a.fuse_decode = (want_decode && !quantized && RPv <= 16) ? 1 : 0;      // (rank 32: the receiver's kernel is the MFMA form - the caller runs it)
        a.offU16 = offU16; a.offV16 = offV16;
        a.arena = lrs_arena(ctx, (void*)s, N, C, RPv, nb, &a.arena_stride, &a.offFull);
        a.tick = cfx_i_ticket_block(ctx, (void*)s);
        if (!a.arena || !a.tick) return fail(ctx, CFX_ERR_LAUNCH, "low-rank: cannot set up the hand-over arena");
        a.err = ctx->gate_err;
        a.timeout = ctx->gate_timeout;
        a.probe = cfx_i_probe(ctx);
        const int rc = RPv == 8 ? lrs_run<8>(ctx, bb, a, s) : (RPv == 16 ? lrs_run<16>(ctx, bb, a, s) : lrs_run<32>(ctx, bb, a, s));
        if (rc != CFX_OK) return rc;
    }
    if (decoded) *decoded = (want_decode && !quantized && RPv <= 16) ? 1 : 0;
    return CFX_OK;
}

// libcfx.so - the abs-mean codec family: the 1-bit and the 2-bit residual codec (fastpath.py), stand-alone kernels, the one-launch compress,
// the layer launches (k_absmean_compress<bits, gated>, k_int2_compress_gated), the cross-layer pipeline (k_binary_pipe), the 1-bit codec with
// rank-K scales.  Shared device code: cfx_device.h; the C-ABI and the dispatch: cfx_api.hip.  Reference citations are relative to
// /root/reference/xfuser/compact/.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "cfx.h"
#include "cfx_internal.h"
#include "cfx_device.h"
#include "cfx_host.h"

// cfx_lowrank.hip (the rank-K scales of the 1-bit codec are low-rank factors of |x - base|)
#define CFX_I_FLAG_LR_FACTORS_ONLY 0x100
#define CFX_I_FLAG_LR_ABS 0x200
extern "C" CFX_HIDDEN size_t cfx_i_lr_workspace_bytes_any(int N, int C, int rank, int batch);
extern "C" CFX_HIDDEN void cfx_i_lr_factor_offsets(int N, int C, int rank, size_t* offU16, size_t* offV16, size_t* per);

// ---------------------------------------------------------------------------------------------------
// abs-mean statistics pass (1-bit and 2-bit codecs)      replaces fastpath.py:150-166 / :614-625 (E1/E2)
//   rowpart[n][cb]  = sum over the tile's 512 channels of |x-base| (units of 2^-24)
//   colpart[p][c]   = sum over the tile's R rows
//   EMIT_BITS: also write bit i of byte j = (x-base)[n,8j+i] >= 0        (fastpath.py:58-85)
// ---------------------------------------------------------------------------------------------------
// The body is shared by the stand-alone kernel, the single-launch compress kernel (k_absmean_compress) and the fused pipeline
// kernel (k_binary_pipe): (bx, by) = tile index, rowpart = this tensor's workspace, NW = waves per workgroup, sm = NW x TILE_C
// words of LDS.  Workspace: rowpart[cb][n] (a row's partials are CB strided words: the finalize reads them coalesced over n),
// then colpart[p][c].
// PUB (needs C % 128 == 0): the sign bits are consumed by other workgroups of the SAME launch (gated reconstruction), so they are
// published write-through; a lane's byte per row would be one fabric write each, so a wave transposes its US rows through LDS
// and 4 lanes per row store 16 bytes.
// KEEP (needs R == NW * US: one trip of the row loop): the tile of x and of the state stays in the caller's registers (xk, bk) -
// the workgroup finishes its own tile from them once the scales exist (absmean_fused_body).
// TAG (the layer launches): the partials go out as tagged words (put_tagged).  The sign bits a reconstruction workgroup will read are
// drained BEFORE the tile's column partials are stored: the gate only opens once every column block's V job has read every tile's
// column partials, so a tile's bits are in memory by then; its row partials need not wait for anything.
template <bool EMIT_BITS, int US, bool WT = false, int NW = WAVES, bool PUB = false, bool KEEP = false, bool TAG = false>
__device__ __forceinline__ void absmean_stats_body(const cfx_comp_item& it, int N, int C, int R, int CB, int bx, int by,
                                                   u64* rowpart, u64 (*sm)[TILE_C], Probe probe = Probe(),
                                                   h16x8* xk = nullptr, h16x8* bk = nullptr, TagArena ta = TagArena()) {
#define SSTAMP(k) probe.at(k)
    const TileCoord t = tile_coord_at(bx, by, N, C, R);
    const int cb = bx;
    u64* colpart = rowpart + (size_t)N * CB;
    const h16* x = (const h16*)it.x;
    const h16* base = (const h16*)it.base;
    unsigned char* bitsout = (unsigned char*)it.packet;
    const int C8 = C >> 3;

    // Per-thread partial sums are kept in fp64: every |d| is a multiple of 2^-24 below 2^16 and a thread adds at most a few
    // hundred of them, so the fp64 sums are EXACT (< 2^53 units) and convert losslessly to the integer unit counts below;
    // two conversions + two fp64 adds per element cost about half the VALU time of the integer formulation.
    double col[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) col[i] = 0.0;

    for (int r = t.r0 + t.w; r < t.r1; r += NW * US) {
        h16x8 xv[US], bv[US];
#pragma unroll
        for (int j = 0; j < US; ++j) {
            const int rr = r + NW * j;
            xv[j] = (h16x8)(h16)0;
            bv[j] = (h16x8)(h16)0;
            if (rr < t.r1 && t.act) {
                // 1-bit: x is not needed again (the EF pass works from the packed bits) -> streaming load
                xv[j] = EMIT_BITS ? ld8nt(x + (size_t)rr * C + t.c) : ld8(x + (size_t)rr * C + t.c);
                if (base) bv[j] = ld8(base + (size_t)rr * C + t.c);
            }
        }
        if (KEEP) {
#pragma unroll
            for (int j = 0; j < US; ++j) { xk[j] = xv[j]; bk[j] = bv[j]; }
        }
        u64 rs[4] = {0, 0, 0, 0};           // the butterfly reduces 4 rows; unused ones stay 0
        if (probe.on()) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SSTAMP(8); }
#pragma unroll
        for (int j = 0; j < US; ++j) {
            const int rr = r + NW * j;
            if (rr < t.r1 && t.act) {
                const h16x8 d = xv[j] - bv[j];
                const h16x8 a = habs8(d);
                unsigned byte = 0;
                double rsum = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    byte |= (d[i] >= (h16)0 ? 1u : 0u) << i;
                    const double v = (double)(float)a[i];
                    col[i] += v;
                    rsum += v;
                }
                rs[j] = (u64)(rsum * 16777216.0);
                if (EMIT_BITS && !PUB) bitsout[(size_t)rr * C8 + (t.c >> 3)] = (unsigned char)byte;
                if (EMIT_BITS && PUB) ((unsigned char*)&sm[t.w][0])[j * 64 + t.lane] = (unsigned char)byte;
            }
        }
        if (EMIT_BITS && PUB) {
            // same wave wrote the bytes: LDS operations of a wave execute in order, no barrier
            const int j = t.lane >> 2, seg = t.lane & 3;
            const int rr = r + NW * j;
            if (t.lane < 4 * US && rr < t.r1 && bx * TILE_C + seg * 128 < C) {
                const u32x4 v = *(const u32x4*)((const unsigned char*)&sm[t.w][0] + j * 64 + seg * 16);
                st16_wt(bitsout + (size_t)rr * C8 + (bx * (TILE_C >> 3)) + seg * 16, v);
            }
        }
        SSTAMP(9);
        // wave-wide row sums by DPP adds (no LDS crossbar round trips): a lane's row sum is below 8 x 2^40 units, split at bit 24
        // so that both halves of the wave sum fit 32 bits (64 x 2^24, 64 x 2^19); the totals are wave-uniform
        u64 tot = 0;
#pragma unroll
        for (int j = 0; j < US; ++j) {
            const unsigned lo = wave_sum_u32_dpp((unsigned)(rs[j] & 0xffffffu));
            const unsigned hi = wave_sum_u32_dpp((unsigned)(rs[j] >> 24));
            const u64 tj = ((u64)hi << 24) + lo;
            if ((t.lane >> 4) == j) tot = tj;
        }
        SSTAMP(10);
        if ((t.lane & 15) == 0 && (t.lane >> 4) < US) {
            const int rr = r + NW * (t.lane >> 4);
            if (rr < t.r1) {
                if (TAG) put_tagged(ta.trow, rowpart, (size_t)cb * N + rr, tot, ta.tagbits);
                else if (WT) put_part(part32_of(rowpart, N, C, CB), rowpart, (size_t)cb * N + rr, tot);
                else rowpart[(size_t)cb * N + rr] = tot;
            }
        }
    }
#pragma unroll
    // exact; [i][lane ^ 8i]: conflict-free here AND in the column-order read below ([i][lane] made that one 8-way conflicted)
    for (int i = 0; i < 8; ++i) sm[t.w][i * 64 + (t.lane ^ (i << 3))] = (u64)(col[i] * 16777216.0);
    SSTAMP(11);
    if (TAG && EMIT_BITS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's sign bits are in memory (see above)
    if (WT) lds_barrier(); else __syncthreads();
    SSTAMP(12);
    for (int k = threadIdx.x; k < TILE_C; k += NW * 64) {   // k = channel within the tile: coalesced global writes
        const int s = (k & 7) * 64 + ((k >> 3) ^ ((k & 7) << 3));
        const int cc = bx * TILE_C + k;
        if (cc < C) {
            u64 v = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += sm[w][s];
            if (TAG) put_tagged(ta.tcol, colpart, (size_t)by * C + cc, v, ta.tagbits);
            else if (WT) put_part(part32_of(rowpart, N, C, CB) + (size_t)N * CB, colpart, (size_t)by * C + cc, v);
            else colpart[(size_t)by * C + cc] = v;
        }
    }
#undef SSTAMP
}

template <bool EMIT_BITS>
__global__ __launch_bounds__(NTHR) void k_absmean_stats(BatchC batch, int N, int C, int R, u64* ws, size_t ws_stride) {
    __shared__ u64 sm[WAVES][TILE_C];
    absmean_stats_body<EMIT_BITS, UNROLL_S>(batch.it[blockIdx.z], N, C, R, gridDim.x, blockIdx.x, blockIdx.y, ws + (size_t)blockIdx.z * ws_stride, sm);
}

// finalize: U[n] = rowmean/mean(rowmean) (1-bit, fastpath.py:164-165) or rowmean/(mean+1e-6) (2-bit, :619-622);
//           V[c] = colmean (fastpath.py:160,166 / :618).  Written straight into the packet tail (replaces the
//           torch.cat of main.py:149-152).  grid = (1 + ceil(C/256), batch).
// Body shared with k_binary_pipe; NT = threads per block, block bx = 0 does the rows, blocks 1.. do NT/4 columns each;
// smem = NT + 1 words of LDS.  All sums are exact integers, so the result does not depend on NT.
template <int NT>
__device__ __forceinline__ void absmean_finalize_body(const cfx_comp_item& it, int N, int C, int CB, int P, int per_byte, int eps_mode,
                                                      const u64* rowpart, int bx, u64* smem) {
    const u64* colpart = rowpart + (size_t)N * CB;
    h16* U = (h16*)((char*)it.packet + (size_t)N * (C / per_byte));
    h16* V = U + N;
    const int tid = threadIdx.x;
    if (bx == 0) {
        // rows: one thread per row (all CB partial loads independent); the row sums of a thread's first KEEP rows stay in
        // registers for the second pass (N <= KEEP * NT: no reload at all); exact sum of the fp16 row means: wave shuffles,
        // then one LDS round over the NT / 64 waves
        constexpr int KEEP = 3;
        u64 srow[KEEP];
        u64 acc = 0;
        int it_n = 0;
        for (int n = tid; n < N; n += NT, ++it_n) {
            u64 s = 0;
#pragma unroll 8
            for (int k = 0; k < CB; ++k) s += rowpart[(size_t)k * N + n];
#pragma unroll
            for (int q = 0; q < KEEP; ++q)
                if (it_n == q) srow[q] = s;
            acc += habs_units(hbits(mean16(s, C)));
        }
        acc = wave_sum_u64(acc);
        if ((tid & 63) == 0) smem[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) {
            u64 tot = 0;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) tot += smem[w];
            smem[NT] = hbits(mean16(tot, N));
        }
        __syncthreads();
        const h16 mu = hfrom((u16)smem[NT]);
        const float den = eps_mode ? (float)(h16)((float)mu + 1e-6f) : (float)mu;
        it_n = 0;
        for (int n = tid; n < N; n += NT, ++it_n) {
            u64 s = 0;
            if (it_n < KEEP) {
#pragma unroll
                for (int q = 0; q < KEEP; ++q)
                    if (it_n == q) s = srow[q];
            } else {
#pragma unroll 8
                for (int k = 0; k < CB; ++k) s += rowpart[(size_t)k * N + n];
            }
            U[n] = (h16)((float)mean16(s, C) / den);
        }
    } else {
        // columns: NT/4 columns per block, 4 threads per column split the P partials; two accumulators and a deep unroll keep
        // all of a thread's loads in flight together
        constexpr int COLS = NT / 4;
        u64* cs = smem;                       // [4][COLS]
        const int cl = tid % COLS, q = tid / COLS;
        const int c = (bx - 1) * COLS + cl;
        u64 s0 = 0, s1 = 0;
        if (c < C) {
            int p = q;
#pragma unroll 6
            for (; p + 4 < P; p += 8) { s0 += colpart[(size_t)p * C + c]; s1 += colpart[(size_t)(p + 4) * C + c]; }
            if (p < P) s0 += colpart[(size_t)p * C + c];
        }
        cs[q * COLS + cl] = s0 + s1;
        __syncthreads();
        if (q == 0 && c < C) V[c] = mean16(cs[cl] + cs[COLS + cl] + cs[2 * COLS + cl] + cs[3 * COLS + cl], N);
    }
}

__global__ __launch_bounds__(1024) void k_absmean_finalize(BatchC batch, int N, int C, int CB, int P, int per_byte,
                                                           int eps_mode, const u64* ws, size_t ws_stride) {
    __shared__ u64 smem[1024 + 8];
    absmean_finalize_body<1024>(batch.it[blockIdx.y], N, C, CB, P, per_byte, eps_mode, ws + (size_t)blockIdx.y * ws_stride, blockIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------------
// 1-bit dequant + base add        replaces _binary_dequant_fastpath (fastpath.py:277-367) AND the
// UPDATE_CACHE branch of _binary_quant_fastpath (fastpath.py:88-120): out = base + (2b-1)*fp16(u[n]*v[c])
// ---------------------------------------------------------------------------------------------------
// UN = rows a wave keeps in flight (2 on the whole chip; 4 on a CU-masked lane, where bytes in flight per CU bound the rate)
template <int NW = WAVES, int UN = UNROLL>
__device__ __forceinline__ void binary_dequant_body(const cfx_decomp_item& it, int N, int C, int R, int tile_x, int tile_y) {
    const TileCoord t = tile_coord_at(tile_x, tile_y, N, C, R);
    const unsigned char* pk = (const unsigned char*)it.packet;
    const int C8 = C >> 3;
    const h16* U = (const h16*)(pk + (size_t)N * C8);
    const h16* V = U + N;
    const h16* base = (const h16*)it.base;
    h16* out = (h16*)it.recon;
    const bool val16 = (((uintptr_t)V) & 15) == 0;
    h16x8 v8 = (h16x8)(h16)0;
    if (t.act) v8 = ld8_tail(V + t.c, val16);

    for (int r = t.r0 + t.w; r < t.r1; r += NW * UN) {
        h16x8 bv[UN];
        unsigned by[UN];
        h16 u[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int rr = r + NW * j;
            bv[j] = (h16x8)(h16)0;
            by[j] = 0;
            u[j] = (h16)0;
            if (rr < t.r1 && t.act) {
                if (base) bv[j] = ld8nt(base + (size_t)rr * C + t.c);
                by[j] = pk[(size_t)rr * C8 + (t.c >> 3)];
                u[j] = U[rr];
            }
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int rr = r + NW * j;
            if (rr < t.r1 && t.act) {
                const h16x8 s = v8 * u[j];                       // fp16(u*v), one rounding (fastpath.py:109,328)
                u16x8 sb = __builtin_bit_cast(u16x8, s);
#pragma unroll
                for (int i = 0; i < 8; ++i) sb[i] ^= ((by[j] >> i) & 1u) ? (u16)0 : (u16)0x8000;   // (2b-1)*s
                const h16x8 recv = __builtin_bit_cast(h16x8, sb);
                st8nt(out + (size_t)rr * C + t.c, base ? (bv[j] + recv) : recv);
            }
        }
    }
}

template <int UN>
__global__ __launch_bounds__(NTHR) void k_binary_dequant(BatchD batch, int N, int C, int R, unsigned* pre, unsigned pre_val) {
    // lane: publish `pre` first - the launch in front of this one in the stream (the previous peer's reconstruction) has finished
    if (pre && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0) st_wt(pre, pre_val);
    binary_dequant_body<WAVES, UN>(batch.it[blockIdx.z], N, C, R, blockIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------
// 1-bit codec with rank-K scales      replaces the K-loop of _binary_quant_fastpath / _binary_dequant_fastpath (fastpath.py:88-120, :330-360)
// and quantize_1bit(rank >= 1) / dequantize_1bit (compress_quantize.py:37-49, :154-225); deprecated in the reference (main.py:188-189).
//   scale[n, c] = fp16( sum_k fp32( fp16(U[n,k] * V[c,k]) ) )     U (N, K), V (C, K) fp16 = the rank-K factors of |x - base| (cfx_lowrank)
//   out = base + (2 b - 1) * scale                                 wire [ bits N*C/8 | U N*K | V C*K ]   (main.py:149-152)
// (Triton's tl.sum adds the fp16 products in fp16 in an unspecified tree order; here they are added in fp32 in index order and
// rounded once - equal for K = 1, within an ulp otherwise; sender and receiver run this same arithmetic on the same fp16 factors.)
// QUANT: x and the factor workspace in, bits + factors + new state out; else packet in, reconstruction out.  K <= 32 (the factor chain's
// limit), eight factors at a time: the partial sums of a thread's rows x 8 channels stay in registers between the chunks.
// ---------------------------------------------------------------------------------------------------
struct RankFac { const h16* U[CFX_MAX_BATCH]; const h16* VT[CFX_MAX_BATCH]; };
template <bool QUANT>
__global__ __launch_bounds__(NTHR) void k_binary_rank(BatchC bc, BatchD bd, RankFac fac, int N, int C, int R, int K, int flags) {
    const TileCoord t = tile_coord(N, C, R);
    const int z = blockIdx.z;
    const int C8 = C >> 3;
    unsigned char* pk = QUANT ? (unsigned char*)bc.it[z].packet : (unsigned char*)bd.it[z].packet;
    h16* Up = (h16*)(pk + (size_t)N * C8);                       // packet sections
    h16* Vp = Up + (size_t)N * K;
    const h16* U = QUANT ? fac.U[z] : Up;
    const h16* VT = QUANT ? fac.VT[z] : Vp;
    const h16* x = QUANT ? (const h16*)bc.it[z].x : nullptr;
    const h16* base = QUANT ? (const h16*)bc.it[z].base : (const h16*)bd.it[z].base;
    h16* out = QUANT ? (h16*)bc.it[z].new_base : (h16*)bd.it[z].recon;
    const bool upd = QUANT ? ((flags & CFX_FLAG_UPDATE_CACHE) && out) : true;
    const bool ef = !(flags & CFX_FLAG_NO_EF);
    if (QUANT) {
        // the factors go into the packet as they are: V by the first row block, U by the first column block
        if (blockIdx.y == 0 && t.act)
            for (int e = 0; e < 8; ++e)
                for (int k = 0; k < K; ++k) Vp[(size_t)(t.c + e) * K + k] = VT[(size_t)(t.c + e) * K + k];
        if (blockIdx.x == 0)
            for (int i = threadIdx.x; i < (t.r1 - t.r0) * K; i += NTHR) Up[(size_t)t.r0 * K + i] = U[(size_t)t.r0 * K + i];
    }
    if (!t.act) return;
    // scale[r][e] = fp16( sum_k fp32( fp16(U[r,k] * V[c+e,k]) ) ), k in index order: chunks of 8 factors, the sums carried across
    constexpr int RPT = UNROLL;                                     // rows of the tile a wave visits (the launches use R = WAVES * UNROLL)
    float acc[RPT][8];
#pragma unroll
    for (int j = 0; j < RPT; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 8) {
        h16 v[8][8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int k = 0; k < 8; ++k) v[e][k] = (k0 + k < K) ? VT[(size_t)(t.c + e) * K + k0 + k] : (h16)0;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int r = t.r0 + t.w + j * WAVES;
            if (r >= t.r1) continue;
            h16 u[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) u[k] = (k0 + k < K) ? U[(size_t)r * K + k0 + k] : (h16)0;
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[j][e] += (float)(h16)(u[k] * v[e][k]);   // fp16 product (one rounding), fp32 sum in index order
        }
    }
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int r = t.r0 + t.w + j * WAVES;
        if (r >= t.r1) continue;
        h16x8 bv = (h16x8)(h16)0, xv = (h16x8)(h16)0;
        if (base) bv = ld8(base + (size_t)r * C + t.c);
        unsigned byte;
        if (QUANT) {
            xv = ld8(x + (size_t)r * C + t.c);
            const h16x8 d = xv - bv;
            byte = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) byte |= (d[i] >= (h16)0 ? 1u : 0u) << i;
            pk[(size_t)r * C8 + (t.c >> 3)] = (unsigned char)byte;
        } else byte = pk[(size_t)r * C8 + (t.c >> 3)];
        if (!upd) continue;
        h16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const h16 sc = (h16)acc[j][e];
            const h16 recv = ((byte >> e) & 1u) ? sc : -sc;
            o[e] = base ? (h16)(bv[e] + recv) : recv;
        }
        if (QUANT && !ef) o = xv;                                   // error feedback off: the state becomes the activation (main.py:233)
        st8(out + (size_t)r * C + t.c, o);
    }
}

// A workgroup's scales out of the launch's TAGGED copies (absmean_tagged_jobs): wave 0 polls the 512 channel words of its column block
// and the token words of its rows - the data is its own "published" mark - and hands them to the other waves through LDS:
// s16[0 .. 512) channel scales, s16[512 + i] token scale of row r0 + i.  Who sees a column block's channel scales tagged knows the
// sign bits / codes prerequisites of that block are in memory too: the V job read every tile's column partials, and a tile stores
// those only after its own packet stores have drained.  Returns false when the wait gave up (the caller stores nothing).
// WATCH: two lanes poll one word each - a channel word of this workgroup's column block and a token word of its rows (other workgroups
// watch other lines) - until they carry the tag; only then does everybody load.  Who has seen the channel word tagged may read the
// block's sign bits / nothing else is needed for them (see above): the caller issues those loads BEFORE scales_from_tagged<.., false>,
// one round trip for both.  Hundreds of workgroups polling ALL their words from the moment their state tile has landed measured 9 %
// slower than the arrival gate they replaced (1.56 vs 1.43 ms per step): the polls load the fabric the statistics chain needs.
#ifndef WATCH_SLEEP
#define WATCH_SLEEP 4
#endif
#ifndef INT2_FLAG_SLEEP
#define INT2_FLAG_SLEEP 2
#endif
__device__ __forceinline__ void scales_watch(const TagArena& ta, int N, int C, int c0, int r0, long long timeout) {
    const int tid = threadIdx.x;
    if (tid < 2) {
        const u64* w = tid ? ta.tU + min(r0, N - 1) : ta.tV + min(c0 + ((r0 >> 1) & (TILE_C - 1)), C - 1);
        SpinClock clk;
        while (!tag_is(ld_wt(w), ta.tagbits)) {
            __builtin_amdgcn_s_sleep(WATCH_SLEEP);
            if (clk.expired(timeout)) break;              // (scales_from_tagged then gives up for everybody)
        }
    }
    lds_barrier();
}
template <int ROWS, bool WATCH = true>
__device__ __forceinline__ bool scales_from_tagged(const TagArena& ta, int N, int C, int c0, int r0, u16* s16, long long timeout, unsigned* err) {
    // every thread polls ONE channel word (512 threads = the column block) and, the first ROWS of them, one token word: two registers a
    // thread instead of a wave's worth of words in one wave's registers (the reconstruction tiles hold 14 - 23 rows of state meanwhile)
    static_assert(ROWS <= TILE_C, "one token word per thread (512 threads = the column block's channels)");
    const int tid = threadIdx.x;
    const u64* pv = ta.tV + min(c0 + tid, C - 1);
    const u64* pu = ta.tU + min(r0 + tid, N - 1);
    SpinClock clk;
    // WATCH first: two lanes poll one word each - a channel word and a token word of this workgroup's own (other workgroups watch other
    // lines) - until they carry the tag; only then does everybody load.  Hundreds of workgroups polling all their words from the moment
    // their state tile has landed measured 9 % slower than the arrival gate they replaced (1.56 vs 1.43 ms per step): the polls load the
    // fabric the statistics chain needs.
    if (WATCH) scales_watch(ta, N, C, c0, r0, timeout);
    for (;;) {
        const u64 v = ld_wt(pv);
        const u64 u = tid < ROWS ? ld_wt(pu) : ta.tagbits;
        const bool ok = tag_is(v, ta.tagbits) && tag_is(u, ta.tagbits);
        s16[tid] = (u16)v;
        if (tid < ROWS) s16[TILE_C + tid] = (u16)u;
        if (!__syncthreads_or(ok ? 0 : 1)) return true;
        __builtin_amdgcn_s_sleep(2);
        if (__syncthreads_or(!ok && clk.expired(timeout) ? 1 : 0)) {
            if (tid == 0) gate_fail(err);
            return false;
        }
    }
}

// The 1-bit launch's reconstruction tiles could take their scales from the tagged copies as well (own packets: no gate, no relay).
// Measured, twice, against the arrival gate on one box: 1.59 vs 1.43-1.45 ms per step - 480 workgroups each polling two fabric words lose
// more than the gate's drain + atomic + relay hop cost (the relays poll through the fabric, everybody else an XCD-local word in L2).  The
// path stays compiled out; the 2-bit launch, whose tiles wait for 4-6 tile flags instead of the slowest of 204 arrivals, keeps it.
#ifndef ONEBIT_D_TAGGED
#define ONEBIT_D_TAGGED 0
#endif
template <int NW, int KR, int KL, bool ST>
__device__ __forceinline__ void binary_dequant_gated_body(const cfx_decomp_item& it, int N, int C, int R, int tile_x, int tile_y,
                                                         unsigned* gate, unsigned expect, unsigned* err, long long timeout, u32x4* lds,
                                                         Probe probe = Probe(), bool remote = false, bool tagged = false, TagArena ta = TagArena(),
                                                         u16* s16 = nullptr) {
    constexpr int K = KR + KL;
#define GSTAMP(k) do { if (ST && probe.on()) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); probe.at(k); } } while (0)
    const TileCoord t = tile_coord_at(tile_x, tile_y, N, C, R);
    if (ST) { probe.at(0); probe.set(7, 4); }
    const unsigned char* pk = (const unsigned char*)it.packet;
    const int C8 = C >> 3;
    const u16* U = (const u16*)(pk + (size_t)N * C8);
    const u16* V = U + N;
    const h16* base = (const h16*)it.base;
    h16* out = (h16*)it.recon;
    const int cc = min(t.c, C - 8);                       // clamped: every load unconditional
    h16x8 bv[KR];
    if (base) {
        // all K rows at once: holding the burst back, or thinning it to a few rows in flight, only moves the contention from the
        // compress group's tile loads to its reduction tail (measured: no gain)
#pragma unroll
        for (int j = 0; j < KR; ++j) bv[j] = ld8nt(base + (size_t)min(t.r0 + t.w + NW * j, t.r1 - 1) * C + cc);
        if constexpr (KL > 0) {
            h16x8 tl[KL > 0 ? KL : 1];
#pragma unroll
            for (int j = 0; j < KL; ++j) tl[j] = ld8nt(base + (size_t)min(t.r0 + t.w + NW * (KR + j), t.r1 - 1) * C + cc);
#pragma unroll
            for (int j = 0; j < KL; ++j) lds[j * (NW * 64) + threadIdx.x] = __builtin_bit_cast(u32x4, tl[j]);   // read back by the same thread
        }
    } else {
#pragma unroll
        for (int j = 0; j < KR; ++j) bv[j] = (h16x8)(h16)0;
    }
    GSTAMP(1);
    h16x8 v8;
    u16 ul;
    unsigned by[K];
    if (tagged) {
        // the packet is one of THIS launch's: its scales come as tagged words (no gate); the sign bits' loads go out with the scale words'
        scales_watch(ta, N, C, tile_x * TILE_C, t.r0, timeout);
        if (ST) probe.at(2);
#pragma unroll
        for (int j = 0; j < K; ++j) by[j] = ld_wt(pk + (size_t)min(t.r0 + t.w + NW * j, t.r1 - 1) * C8 + (cc >> 3));
        if (!scales_from_tagged<NW * K, false>(ta, N, C, tile_x * TILE_C, t.r0, s16, timeout, err)) return;
        v8 = __builtin_bit_cast(h16x8, *(const u16x8*)(s16 + 8 * t.lane));
        ul = s16[TILE_C + min(t.w + NW * min(t.lane, K - 1), t.r1 - 1 - t.r0)];
    } else if (!gate_wait<true>(gate, expect, err, timeout)) return;
    else if (remote) {
        if (ST) probe.at(2);                                            // (uniform) the packet sits in a peer GPU's memory: system-scope loads
        u16x8 vb;
#pragma unroll
        for (int i = 0; i < 8; ++i) vb[i] = ld_sys(V + cc + i);
        v8 = __builtin_bit_cast(h16x8, vb);
        ul = ld_sys(U + min(t.r0 + t.w + NW * min(t.lane, K - 1), t.r1 - 1));
#pragma unroll
        for (int j = 0; j < K; ++j) by[j] = ld_sys(pk + (size_t)min(t.r0 + t.w + NW * j, t.r1 - 1) * C8 + (cc >> 3));
    } else {
        if (ST) probe.at(2);
        v8 = ld8_wt(V + cc);
        // a row's token scale is wave-uniform: lane j fetches row j's, broadcast by readlane below
        ul = ld_wt(U + min(t.r0 + t.w + NW * min(t.lane, K - 1), t.r1 - 1));
#pragma unroll
        for (int j = 0; j < K; ++j) by[j] = ld_wt(pk + (size_t)min(t.r0 + t.w + NW * j, t.r1 - 1) * C8 + (cc >> 3));
    }
    GSTAMP(3);
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int rr = t.r0 + t.w + NW * j;
        const h16 uj = hfrom((u16)__builtin_amdgcn_readlane((int)ul, j));
        if (rr < t.r1 && t.act) {
            const h16x8 s = v8 * uj;                         // fp16(u*v), one rounding (fastpath.py:109,328)
            u16x8 sb = __builtin_bit_cast(u16x8, s);
#pragma unroll
            for (int i = 0; i < 8; ++i) sb[i] ^= ((by[j] >> i) & 1u) ? (u16)0 : (u16)0x8000;   // (2b-1)*s
            const h16x8 recv = __builtin_bit_cast(h16x8, sb);
            h16x8 bj = (h16x8)(h16)0;
            if (j < KR) bj = bv[j < KR ? j : 0];
            else if (base) bj = __builtin_bit_cast(h16x8, lds[(j - KR) * (NW * 64) + threadIdx.x]);
            st8nt(out + (size_t)rr * C + t.c, base ? (bj + recv) : recv);
        }
    }
    GSTAMP(4);
#undef GSTAMP
}

// ---------------------------------------------------------------------------------------------------
// Compress in ONE launch: statistics pass + in-launch finalize by the last-arriving workgroups (no finalize kernel, no
// grid barrier).  Every tile workgroup publishes its partial sums write-through, drains them, and draws two tickets:
//   tick[1 + cb] counts the P row-tiles of column block cb   -> the workgroup that draws P - 1 reduces colpart[0..P)[cb]
//                                                                and writes V for those 512 channels
//   tick[0]      counts all CB x P tiles of the tensor         -> the workgroup that draws CB * P - 1 reduces the row sums,
//                                                                the grand mean and writes U
// All sums are exact integers, so the result is bit-identical to k_absmean_stats + k_absmean_finalize for any arrival
// order.  A ticket word is reset by the workgroup that drew its last value (nobody touches it afterwards in this launch),
// so the ticket block - owned by the cfx_ctx, zeroed once - is reusable by the next launch.  Results never depend on
// dispatch order or workgroup -> XCD placement: the last arriver is whoever happens to arrive last.
// The launch can also carry `ride` reconstruction items (1-bit only): bandwidth work that does not depend on this
// launch's statistics - the previous layer's deferred error-feedback update - streams while the reduction tail, which
// is pure latency, completes (cfx_compress_batch_ex).
// ---------------------------------------------------------------------------------------------------
// Row sums of rows m0 and m1 from the transposed partials rowpart[k][n]: 8 column blocks (16 loads) per batch, every load
// unconditional (clamped block index, masked value) - a remainder loop would be CB dependent round trips.
__device__ __forceinline__ void row_sums2_wt(const u64* rowpart, const unsigned* row32, int N, int CB, int m0, int m1, bool two, u64& s0, u64& s1) {
    // `two` is wave-uniform: a wave whose second rows all lie beyond N skips that half of the batch (one branch around the
    // batch, not one per load)
    s0 = 0; s1 = 0;
    for (int k0 = 0; k0 < CB; k0 += FUSED_RCH) {
        unsigned a[FUSED_RCH], b[FUSED_RCH];
#pragma unroll
        for (int j = 0; j < FUSED_RCH; ++j) a[j] = ld_wt(&row32[(size_t)min(k0 + j, CB - 1) * N + m0]);
        if (two) {
#pragma unroll
            for (int j = 0; j < FUSED_RCH; ++j) b[j] = ld_wt(&row32[(size_t)min(k0 + j, CB - 1) * N + m1]);
        } else {
#pragma unroll
            for (int j = 0; j < FUSED_RCH; ++j) b[j] = 0;
        }
        bool sat = false;
#pragma unroll
        for (int j = 0; j < FUSED_RCH; ++j) {
            s0 += (k0 + j < CB) ? a[j] : 0; s1 += (k0 + j < CB) ? b[j] : 0;
            sat |= (a[j] == PART_SAT) | (b[j] == PART_SAT);
        }
        if (sat) {                               // rare: a partial that did not fit 32 bits - its exact value is in the 64-bit array
            for (int j = 0; j < FUSED_RCH && k0 + j < CB; ++j) {
                if (a[j] == PART_SAT) s0 += ld_wt(&rowpart[(size_t)(k0 + j) * N + m0]) - (u64)PART_SAT;
                if (two && b[j] == PART_SAT) s1 += ld_wt(&rowpart[(size_t)(k0 + j) * N + m1]) - (u64)PART_SAT;
            }
        }
    }
}

// GATED: workgroups of the SAME launch reconstruct from this packet (binary_dequant_gated_body), so everything that goes into
// it is published write-through, and the gate counts the last-arriver JOBS: one arrival per job once its U / V stores have drained,
// batch * (CB + 1) in all.  The tiles need no arrival of their own: a column block's V job only starts after the block's last
// ticket was drawn, and every tile drains its sign bits before it draws its tickets - all jobs in implies all tiles out.
//
// The last arrivers' jobs: V of a column block (last_col), U of the tensor (last_all).
template <bool GATED>
__device__ __forceinline__ void absmean_last_arriver_jobs(const cfx_comp_item& it, int N, int C, int CB, int P, int bx, u64* rowpart,
                                                          unsigned* tick, int per_byte, int eps_mode, u64 (*sm)[TILE_C], bool last_col,
                                                          bool last_all, Probe probe, unsigned* gate, unsigned gate_expect) {
    constexpr int NT = FUSED_NT;
#define STAMP(k) probe.at(k)
    lds_barrier();                             // the flags have been read: sm may be reused
    // The last arrivers' reductions are ONE fabric round trip when N <= 2 NT and P <= FUSED_CH: every load is unconditional
    // (clamped index, masked value) and the loads of BOTH jobs - a workgroup is often last of its column block and of the
    // tensor - are issued before anything is consumed; a wave-uniform branch per load would serialise them into dependent
    // round trips (cdna_hip_programming.md, ".s-level traps" (c)).
    const int tid = threadIdx.x;
    const u64* colpart = rowpart + (size_t)N * CB;
    const unsigned* row32 = part32_of(rowpart, N, C, CB);
    const unsigned* col32 = row32 + (size_t)N * CB;
    h16* U = (h16*)((char*)it.packet + (size_t)N * (C / per_byte));
    h16* V = U + N;
    const int c = bx * TILE_C + tid;
    const int cc = min(c, C - 1);
    unsigned v[FUSED_CH];
    if (last_col) {
#pragma unroll
        for (int j = 0; j < FUSED_CH; ++j) v[j] = ld_wt(&col32[(size_t)min(j, P - 1) * C + cc]);
    }
    u64 keep0 = 0, keep1 = 0;                  // row sums of rows tid, tid + NT
    if (last_all) row_sums2_wt(rowpart, row32, N, CB, min(tid, N - 1), min(tid + NT, N - 1), (tid & ~63) + NT < N, keep0, keep1);
    asm volatile("" ::: "memory");
    STAMP(4);
    // reductions first, every global store last: a barrier must not sit behind an outstanding store.  The tensor-wide job (U) is
    // the longer chain, so it goes first and the column job (V) fills the wait for the other waves.
    u64* smem = &sm[1][0];
    float m0 = 0.f, m1 = 0.f;
    if (last_all) {
        // U: one thread per row, two rows per trip; the first trip's sums are already in registers
        const h16 h0 = mean16(keep0, C), h1 = mean16(keep1, C);
        m0 = (float)h0; m1 = (float)h1;
        u64 acc = 0;
        if (tid < N) acc += habs_units(hbits(h0));
        if (tid + NT < N) acc += habs_units(hbits(h1));
        for (int n0 = tid + 2 * NT; n0 - tid < N; n0 += 2 * NT) {
            const int n1 = n0 + NT;
            u64 s0, s1;
            row_sums2_wt(rowpart, row32, N, CB, min(n0, N - 1), min(n1, N - 1), (n1 & ~63) < N, s0, s1);
            if (n0 < N) acc += habs_units(hbits(mean16(s0, C)));
            if (n1 < N) acc += habs_units(hbits(mean16(s1, C)));
        }
        STAMP(13);
        // a thread's acc is below 2^17 rows x 2^40 units; split at bit 24 so that both halves of the wave sum fit 32 bits
        u64 wtot;
        if (N <= 2 * NT) {
            const unsigned lo = wave_sum_u32_dpp((unsigned)(acc & 0xffffffu));       // 64 x 2^24
            const unsigned hi = wave_sum_u32_dpp((unsigned)(acc >> 24));             // 64 x 2^17 (two rows per thread)
            wtot = ((u64)hi << 24) + lo;
        } else wtot = wave_sum_u64(acc);
        STAMP(14);
        if ((tid & 63) == 0) smem[tid >> 6] = wtot;
    }
    h16 vmean = (h16)0;
    if (last_col) {
        // V of column block bx: one column per thread
        u64 a = 0;
        for (int p0 = 0; p0 < P; p0 += FUSED_CH) {
            if (p0) {
#pragma unroll
                for (int j = 0; j < FUSED_CH; ++j) v[j] = ld_wt(&col32[(size_t)min(p0 + j, P - 1) * C + cc]);
            }
            bool sat = false;
#pragma unroll
            for (int j = 0; j < FUSED_CH; ++j) { a += (p0 + j < P) ? v[j] : 0; sat |= v[j] == PART_SAT; }
            if (sat) {                           // rare: see row_sums2_wt
                for (int j = 0; j < FUSED_CH && p0 + j < P; ++j)
                    if (v[j] == PART_SAT) a += ld_wt(&colpart[(size_t)(p0 + j) * C + cc]) - (u64)PART_SAT;
            }
        }
        vmean = mean16(a, N);
    }
    STAMP(5);
    if (last_all) {
        lds_barrier();
        STAMP(15);
        u64 tot = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) tot += smem[w];           // every thread: no second barrier
        const h16 mu = mean16(tot, N);
        const float den = eps_mode ? (float)(h16)((float)mu + 1e-6f) : (float)mu;
#define PUT16(ptr, val) do { const h16 _v = (val); if (GATED) st_wt((u16*)(ptr), hbits(_v)); else *(ptr) = _v; } while (0)
        if (tid < N) PUT16(&U[tid], (h16)(m0 / den));
        if (tid + NT < N) PUT16(&U[tid + NT], (h16)(m1 / den));
        for (int n0 = tid + 2 * NT; n0 - tid < N; n0 += 2 * NT) {
            const int n1 = n0 + NT;
            u64 s0, s1;
            row_sums2_wt(rowpart, row32, N, CB, min(n0, N - 1), min(n1, N - 1), (n1 & ~63) < N, s0, s1);
            if (n0 < N) PUT16(&U[n0], (h16)((float)mean16(s0, C) / den));
            if (n1 < N) PUT16(&U[n1], (h16)((float)mean16(s1, C) / den));
        }
        if (tid == 0) st_wt(tick + TICK_ALL, 0u);
    }
    if (last_col) {
        if (c < C) PUT16(&V[c], vmean);
        if (tid == 0) st_wt(tick + 1 + bx, 0u);
    }
#undef PUT16
    if (GATED) {
        // the scales are out once every wave's stores have drained: one arrival per finished job
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        if (tid == 0) gate_arrive_few(gate, (last_all ? 1u : 0u) + (last_col ? 1u : 0u));
    }
    STAMP(6);
#undef STAMP
}

// The layer launches' jobs on TAGGED partials (put_tagged): no last arriver - a FIXED workgroup per job, the last-dispatched tiles of the
// tensor (V of column block bx: tile (bx, P - 1); U: tile (CB - 1, P - 2)), polls the very words it reduces until all carry the
// launch's tag.  Fourteen workgroups polling ~70 KB a round is nothing beside the launch's traffic (every TILE polling was: the min/max
// layer's first form).  A thread's loads of a round are issued together, then the tags compared (a test per load serialises them).
// Arithmetic = absmean_last_arriver_jobs (exact integer sums: bit-identical for any order).  do_col / do_row: uniform per workgroup.
__device__ __forceinline__ void absmean_tagged_jobs(const cfx_comp_item& it, int N, int C, int CB, int P, int bx, u64* rowpart, const TagArena& ta,
                                                    int per_byte, int eps_mode, u64 (*sm)[TILE_C], bool do_col, bool do_row, Probe probe,
                                                    unsigned* gate, unsigned* err, long long timeout) {
    constexpr int NT = FUSED_NT;
#define STAMP(k) probe.at(k)
    lds_barrier();                             // sm may be reused
    const int tid = threadIdx.x;
    const u64* colpart = rowpart + (size_t)N * CB;
    h16* U = (h16*)((char*)it.packet + (size_t)N * (C / per_byte));
    h16* V = U + N;
    SpinClock clk;
    bool failed = false;
#define PUT16(ptr, val) st_wt((u16*)(ptr), hbits(val))
    if (do_col) {
        // V of column block bx: one column per thread, FUSED_CH partials a round
        const int c = bx * TILE_C + tid, cc = min(c, C - 1);
        u64 a = 0;
        for (int p0 = 0; p0 < P && !failed; p0 += FUSED_CH) {
            u64 v[FUSED_CH];
            for (;;) {
#pragma unroll
                for (int j = 0; j < FUSED_CH; ++j) v[j] = ld_wt(&ta.tcol[(size_t)min(p0 + j, P - 1) * C + cc]);
                bool ok = true;
#pragma unroll
                for (int j = 0; j < FUSED_CH; ++j) ok = ok && tag_is(v[j], ta.tagbits);
                if (ok) break;
                __builtin_amdgcn_s_sleep(2);
                if (clk.expired(timeout)) { failed = true; break; }
            }
#pragma unroll
            for (int j = 0; j < FUSED_CH; ++j) {
                const u64 w = v[j] & TAG_SAT;
                if (p0 + j < P) a += (w == TAG_SAT && !failed) ? ld_wt(&colpart[(size_t)(p0 + j) * C + cc]) : w;      // rare: the exact sum beside it
            }
        }
        STAMP(4);
        failed = __syncthreads_or(failed ? 1 : 0) != 0;
        if (!failed) {
            if (c < C) {
                const h16 vm = mean16(a, N);
                st_wt(&ta.tV[c], ta.tagbits | (u64)hbits(vm));      // first: the copy the launch's own reconstruction workgroups poll
                PUT16(&V[c], vm);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            if (tid == 0) gate_arrive_few(gate, 1u);
        }
        STAMP(5);
    }
    if (do_row && !failed) {
        // U: one row per thread and pass; its CB partials in one round
        u64* smem = &sm[1][0];
        auto row_sum = [&](int n) {
            const int nc = min(n, N - 1);
            u64 s_ = 0;
            for (int k0 = 0; k0 < CB && !failed; k0 += FUSED_RCH) {
                u64 q[FUSED_RCH];
                for (;;) {
#pragma unroll
                    for (int j = 0; j < FUSED_RCH; ++j) q[j] = ld_wt(&ta.trow[(size_t)min(k0 + j, CB - 1) * N + nc]);
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < FUSED_RCH; ++j) ok = ok && tag_is(q[j], ta.tagbits);
                    if (ok) break;
                    __builtin_amdgcn_s_sleep(2);
                    if (clk.expired(timeout)) { failed = true; break; }
                }
#pragma unroll
                for (int j = 0; j < FUSED_RCH; ++j) {
                    const u64 w = q[j] & TAG_SAT;
                    if (k0 + j < CB) s_ += (w == TAG_SAT && !failed) ? ld_wt(&rowpart[(size_t)(k0 + j) * N + nc]) : w;
                }
            }
            return s_;
        };
        // pass 1: every row's fp16 mean -> the sum of the means (the tensor's grand mean); a thread keeps its first two rows' means
        h16 h0 = (h16)0, h1 = (h16)0;
        u64 acc = 0;
        for (int n0 = tid, i = 0; n0 - tid < N; n0 += NT, ++i) {
            const h16 h = mean16(row_sum(n0), C);
            if (i == 0) h0 = h;
            if (i == 1) h1 = h;
            if (n0 < N) acc += habs_units(hbits(h));
        }
        STAMP(13);
        const u64 wtot = wave_sum_u64(acc);
        if ((tid & 63) == 0) smem[tid >> 6] = wtot;
        failed = __syncthreads_or(failed ? 1 : 0) != 0;
        STAMP(15);
        if (!failed) {
            u64 tot = 0;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) tot += smem[w];
            const h16 mu = mean16(tot, N);
            const float den = eps_mode ? (float)(h16)((float)mu + 1e-6f) : (float)mu;
            for (int n0 = tid, i = 0; n0 < N; n0 += NT, ++i) {
                const h16 h = i == 0 ? h0 : (i == 1 ? h1 : mean16(row_sum(n0), C));      // (beyond 2 NT rows: the partials are read again)
                const h16 un = (h16)((float)h / den);
                st_wt(&ta.tU[n0], ta.tagbits | (u64)hbits(un));
                PUT16(&U[n0], un);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            if (tid == 0) gate_arrive_few(gate, 1u);
        }
        STAMP(6);
    }
#undef PUT16
#undef STAMP
    if (failed && tid == 0) gate_fail(err);
}

// What a statistics workgroup of the 2-bit layer launch does with its own tile once the scales exist (R == FUSED_NW * US, the tile
// of x and of the state still in registers): wait for gate 1, quantise (the codes depend on the scales), publish the codes
// write-through, error feedback, one arrival on gate 2 - the arithmetic of k_int2_quant without reading x and the state again.
template <int US>
__device__ __forceinline__ void own_tile_finish(const cfx_comp_item& it, int N, int C, int R, int bx, int by, int flags, const h16x8* xk,
                                                const h16x8* bk, unsigned* gate1, unsigned expect1, unsigned* gate2, unsigned expect2,
                                                unsigned* err, long long timeout, unsigned char* smw, const TagArena& ta, u16* s16) {
    constexpr int NW = FUSED_NW;
    static_assert(US * 8 <= 64, "a wave publishes its US rows of codes with 8 lanes a row");
    const TileCoord t = tile_coord_at(bx, by, N, C, R);
    h16* nb = (h16*)it.new_base;
    const bool upd = (flags & CFX_FLAG_UPDATE_CACHE) && nb;
    const bool ef = !(flags & CFX_FLAG_NO_EF);
    // the scales: the launch's tagged copies, polled (no gate 1: scales_from_tagged)
    unsigned char* pk = (unsigned char*)it.packet;
    (void)gate1; (void)expect1;
    if (!scales_from_tagged<NW * US>(ta, N, C, bx * TILE_C, t.r0, s16, timeout, err)) return;   // (no codes, no arrival on gate 2: the reconstruction group gives up as well)
    const h16x8 ch8 = __builtin_bit_cast(h16x8, *(const u16x8*)(s16 + 8 * t.lane));
    const u16 ul = s16[TILE_C + min(t.w + NW * min(t.lane, US - 1), t.r1 - 1 - t.r0)];
    const bool has_base = it.base != nullptr;
    u16 codes[US];
#pragma unroll
    for (int j = 0; j < US; ++j) {
        const int rr = t.r0 + t.w + NW * j;
        const h16 tk = hfrom((u16)__builtin_amdgcn_readlane((int)ul, j));
        codes[j] = 0;
        if (rr < t.r1 && t.act) {
            const h16x8 d = xk[j] - bk[j];
            const h16x8 thr = ch8 * tk;                                      // fastpath.py:536
            const h16x8 a = habs8(d);
            unsigned code = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned sg = d[i] >= (h16)0 ? 1u : 0u;                // fastpath.py:539
                const unsigned m = a[i] > thr[i] ? 1u : 0u;                  // fastpath.py:540
                code |= ((sg << 1) | m) << (2 * i);
            }
            codes[j] = (u16)code;
            ((u16*)smw)[j * 64 + t.lane] = (u16)code;
        }
    }
    // publish the codes FIRST (the peers' workgroups wait for them; the state update below is nobody's dependency): a row of the
    // tile is 128 bytes = 8 lanes x 16 bytes (same wave wrote the LDS words: in order)
    {
        const int j = t.lane >> 3, seg = t.lane & 7;
        const int rr = t.r0 + t.w + NW * j;
        if (t.lane < 8 * US && rr < t.r1 && bx * TILE_C + seg * 64 < C)
            st16_wt(pk + (size_t)rr * (C >> 2) + bx * (TILE_C >> 2) + seg * 16, *(const u32x4*)(smw + j * 128 + seg * 16));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (threadIdx.x == 0) {
        st_wt(&ta.tdone[(size_t)bx * ((N + R - 1) / R) + by], ta.tagbits);      // for the launch's own reconstruction tiles: they wait for the tiles whose codes they read
        gate_arrive(gate2, 1u, expect2);                                        // for whoever ships the packets: all tiles in
    }
    if (upd) {
#pragma unroll
        for (int j = 0; j < US; ++j) {
            const int rr = t.r0 + t.w + NW * j;
            const h16 tk = hfrom((u16)__builtin_amdgcn_readlane((int)ul, j));
            if (rr < t.r1 && t.act) {
                const h16x8 recv = int2_recv(codes[j], ch8 * tk);
                st8nt(nb + (size_t)rr * C + t.c, ef ? (has_base ? (bk[j] + recv) : recv) : xk[j]);
            }
        }
    }
}

// KEEP (the 2-bit layer launch): after the statistics and - for a last arriver - its jobs, the workgroup stays and quantises its own
// tile from registers (own_tile_finish).
template <bool EMIT_BITS, int US, bool GATED = false, bool KEEP = false>
__device__ __forceinline__ void absmean_fused_body(const cfx_comp_item& it, int N, int C, int R, int CB, int P, int bx, int by,
                                                   u64* rowpart, unsigned* tick, int per_byte, int eps_mode, u64 (*sm)[TILE_C], int dbg,
                                                   Probe probe, unsigned* gate = nullptr, unsigned gate_expect = 0, int flags = 0,
                                                   unsigned* gate2 = nullptr, unsigned expect2 = 0, unsigned* err = nullptr, long long timeout = 0,
                                                   TagArena ta = TagArena()) {
    // developer probes (cfx_dev.h): per-workgroup phase times, 100 MHz wall clock
#define STAMP(k) probe.at(k)
    STAMP(0);
#ifdef CFX_DEV_PROBES                          // experiment early exits: only in a developer build (python -m compactfusion_amd.build --dev-probes)
    if (dbg == 3) return;                      // experiments: launch cost of the empty grid
    if (dbg == 4) {                            // experiments: the loads alone (no arithmetic, no partial sums)
        const TileCoord t = tile_coord_at(bx, by, N, C, R);
        h16x8 acc = (h16x8)(h16)0;
        for (int r = t.r0 + t.w; r < t.r1; r += FUSED_NW * US) {
            h16x8 xv[US], bv[US];
#pragma unroll
            for (int j = 0; j < US; ++j) {
                const int rr = min(r + FUSED_NW * j, t.r1 - 1);
                xv[j] = ld8nt((const h16*)it.x + (size_t)rr * C + min(t.c, C - 8));
                bv[j] = ld8((const h16*)it.base + (size_t)rr * C + min(t.c, C - 8));
            }
#pragma unroll
            for (int j = 0; j < US; ++j) acc += xv[j] - bv[j];
        }
        if (acc[0] == (h16)12345.0f) ((h16*)it.packet)[threadIdx.x] = acc[1];
        return;
    }
#endif
    h16x8 xk[KEEP ? US : 1], bk[KEEP ? US : 1];
    absmean_stats_body<EMIT_BITS, US, true, FUSED_NW, GATED, KEEP, GATED>(it, N, C, R, CB, bx, by, rowpart, sm, probe, xk, bk, ta);
    STAMP(1);
    if constexpr (GATED) {
        // the layer launches: tagged partials, fixed reducers (absmean_tagged_jobs) - nothing to drain, no ticket to draw
        const bool v_wg = by == P - 1;
        const bool u_wg = P >= 2 ? (by == P - 2 && bx == CB - 1) : (bx == CB - 1);
        probe.copy(2, 1); probe.copy(3, 1); probe.set(7, (v_wg ? 1 : 0) | (u_wg ? 2 : 0));
        if (v_wg || u_wg) {
            // KEEP: the jobs' loads in flight beside the whole tile do not fit 128 registers (tools/resource_usage.py, tests/test_resource_usage.py)
            // - the tile's last two rows of x and of the state sit out the jobs in the LDS rows the statistics do not use (sm[FUSED_NW ..]:
            // same thread writes and reads, no barrier)
            u32x4* park = (u32x4*)&sm[FUSED_NW][0];
            if constexpr (KEEP) {
                park[threadIdx.x] = __builtin_bit_cast(u32x4, xk[US - 1]);
                park[FUSED_NT + threadIdx.x] = __builtin_bit_cast(u32x4, bk[US - 1]);
            }
            absmean_tagged_jobs(it, N, C, CB, P, bx, rowpart, ta, per_byte, eps_mode, sm, v_wg, u_wg, probe, gate, err, timeout);
            if constexpr (KEEP) {
                xk[US - 1] = __builtin_bit_cast(h16x8, park[threadIdx.x]);
                bk[US - 1] = __builtin_bit_cast(h16x8, park[FUSED_NT + threadIdx.x]);
            }
        }
    } else {
    // publish: EVERY storing wave drains its write-through stores, then one lane pair draws the two tickets
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(2);
#ifdef CFX_DEV_PROBES
    if (dbg == 1) return;
#endif
    // Who does what: the column ticket (a returned atomic) elects the workgroup that reduces column block bx (V).  The tensor-wide job (U)
    // goes to a FIXED workgroup, tile (0, 0): the last tile of all is always also the last of its column block and would read both jobs'
    // partials at the ~65 GB/s a single workgroup gets from other CUs; every tile counts itself on tick[TICK_ALL] without waiting for the
    // result, and tile (0, 0), once done with its own work, polls that word (one reader on a line of its own) and reads only the row
    // partials: -0.9 us on the launch.
    unsigned* flag = (unsigned*)&sm[0][0];
    if (threadIdx.x == 0) flag[0] = __hip_atomic_fetch_add(tick + 1 + bx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 64) (void)__hip_atomic_fetch_add(tick + TICK_ALL, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lds_barrier();
    const bool last_col = flag[0] == (unsigned)(P - 1);
    const bool u_wg = bx == 0 && by == 0;
    STAMP(3);
    probe.set(7, (last_col ? 1 : 0) | (u_wg ? 2 : 0));
#ifdef CFX_DEV_PROBES
    if (dbg == 2) {
        if (last_col && threadIdx.x == 0) st_wt(tick + 1 + bx, 0u);
        if (u_wg && threadIdx.x == 0) {
            while (ld_wt(tick + TICK_ALL) != (unsigned)(CB * P)) __builtin_amdgcn_s_sleep(1);
            st_wt(tick + TICK_ALL, 0u);
        }
        return;
    }
#endif
    if (last_col)                          // uniform per workgroup
        absmean_last_arriver_jobs<GATED>(it, N, C, CB, P, bx, rowpart, tick, per_byte, eps_mode, sm, true, false, probe, gate, gate_expect);
    if (u_wg) {
        bool failed = false;
        if (threadIdx.x == 0) {
            SpinClock clk;
            while (ld_wt(tick + TICK_ALL) != (unsigned)(CB * P)) {
                __builtin_amdgcn_s_sleep(1);
                if (clk.expired(timeout)) { failed = true; gate_fail(err); break; }
            }
        }
        if (__syncthreads_or(failed ? 1 : 0)) return;           // (tiles that never arrived: no row scales - the error word says so)
        absmean_last_arriver_jobs<GATED>(it, N, C, CB, P, bx, rowpart, tick, per_byte, eps_mode, sm, false, true, probe, gate, gate_expect);
    }
    }
    if constexpr (KEEP)
        own_tile_finish<US>(it, N, C, R, bx, by, flags, xk, bk, gate, gate_expect, gate2, expect2, err, timeout,
                                       (unsigned char*)&sm[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)][0], ta, (u16*)&sm[FUSED_NW + 4][0]);
#undef STAMP
}

struct FusedArgs {
    int N, C, CB, R, P;      // statistics tiles: CB x P per tensor, R rows each
    int n_st;                // workgroups of the statistics group (CB * P * batch); the rest reconstruct `ride`
    int dq_R, dq_rb;         // ride items: tile height, row blocks per tensor
    int per_byte, eps_mode;
    int dbg;                 // developer builds only (CFX_DEV_PROBES): 1 = stop after publishing, 2 = after the tickets, 3 = empty grid, 4 = loads only
    u64* ws;
    size_t ws_stride;
    unsigned* tick;
    Probe probe;             // developer build: 16 words per workgroup (empty in the product build)
    // gated reconstruction group (GATED kernels): workgroups [n_st, n_st + n_g), tiles of g_R rows, g_rb per tensor; n_gt tiles in all:
    // n_g == n_gt, one tile per workgroup, or n_g < n_gt: a PERSISTENT group, workgroup g takes tiles g, g + n_g, ...
    int n_g, g_R, g_rb, n_gt;
    unsigned* gate;
    unsigned gate_expect;
    unsigned* gate_err;
    long long timeout;       // every in-launch wait gives up after this many ticks of the 100 MHz wall clock (cfx_set_gate_timeout_ms)
    u64* tarena; size_t tarena_stride; unsigned tag;      // GATED: the context's tagged-words arena (per tensor: tag_arena_words), the launch's 24-bit tag
    signed char src[CFX_MAX_BATCH];                        // gated item -> the own tensor of THIS launch whose packet it reads, or -1 (somebody else's packet: gate)
    // external gate (exchange-layer op, cfx_plan_add_exchange_layer): the gated group waits for this word instead of the arrival
    // counter - whoever moves the packets (a collective on the exchange stream) sets it once they have arrived.  NULL: wait on `gate`.
    unsigned* xgate;
    unsigned xexpect;
    int remote;              // the gated items' packets may sit in a peer GPU's memory (read with system-scope loads)
    P2PInline p2p;           // own != NULL: workgroup 0 runs the peer-to-peer exchange and opens xgate itself
};
#ifndef GATE_WPE
#define GATE_WPE 4               // waves per SIMD the single-launch compress kernels are compiled for (2 workgroups / CU)
#endif
// Register budget: 104 VGPRs.  A collective KERNEL (RCCL: 256 threads x ~280 VGPRs) has to find room beside the waiting reconstruction
// group when the collective sits in the path (cfx_plan_add_exchange_layer, needs_room in compress_impl): on a CU that holds one of these
// workgroups - two waves a SIMD - 512 - 2 x 104 = 304 registers stay free, with the launch bound's 128 only 256.  The kernel fits 101
// by itself when the developer probes' branches are compiled in and took 122 without them (same code, other schedule), so the budget is
// stated: amdgpu_num_vgpr counts HALF registers on gfx90a+ (unified 512-entry file: LLVM doubles the request), 52 -> 104.  No spills
// (tests/test_resource_usage.py reads the compiler's remarks: ScratchSize 0, VGPRs <= 104).
template <bool EMIT_BITS, int US, bool GATED = false, bool ST = false>
__global__ __launch_bounds__(FUSED_NT, GATE_WPE) __attribute__((amdgpu_num_vgpr(52))) void k_absmean_compress(BatchC batch, BatchD ride, BatchD gated, FusedArgs a) {
    __shared__ u64 sm[FUSED_NW][TILE_C];
    int b = blockIdx.x;
    if (b < a.n_st) {
        const int per = a.CB * a.P;
        const int z = b / per, rem = b - z * per;
        const int by = rem / a.CB;
        absmean_fused_body<EMIT_BITS, US, GATED>(batch.it[z], a.N, a.C, a.R, a.CB, a.P, rem - by * a.CB, by, a.ws + (size_t)z * a.ws_stride,
                                                 a.tick + z * TICK_WORDS, a.per_byte, a.eps_mode, sm, a.dbg,
                                                 a.probe.of(b), a.gate, a.gate_expect, 0, nullptr, 0u, a.gate_err, a.timeout,
                                                 tag_arena_of(a.tarena, a.tarena_stride, z, a.N, a.C, a.CB, a.P, a.tag));
        if constexpr (GATED) {
            if (b == 0 && a.p2p.own) p2p_exchange_inline(a.gate, a.gate_expect, 1, a.p2p, a.xgate, a.xexpect, a.gate_err);
        }
        return;
    }
    if constexpr (EMIT_BITS) {
        b -= a.n_st;
        if constexpr (GATED) {
            if (b < a.n_g) {
                const int per = a.CB * a.g_rb;
                for (int t = b; t < a.n_gt; t += a.n_g) {             // (one trip unless the group is persistent)
                    const int item = t / per, rem = t - item * per;
                    const int ty = rem / a.CB;
                    const int sz = a.src[item];
                    const TagArena ta = tag_arena_of(a.tarena, a.tarena_stride, sz >= 0 ? sz : 0, a.N, a.C, a.CB, a.P, a.tag);
                    binary_dequant_gated_body<FUSED_NW, GATE_KR, 0, ST>(gated.it[item], a.N, a.C, a.g_R, rem - ty * a.CB, ty, a.xgate ? a.xgate : a.gate,
                                                                a.xgate ? a.xexpect : a.gate_expect, a.gate_err, a.timeout,
                                                                nullptr,
                                                                a.probe.of(blockIdx.x), a.remote != 0 && sz < 0,
                                                                ONEBIT_D_TAGGED && sz >= 0, ta, (u16*)&sm[0][0]);
                }
                return;
            }
            b -= a.n_g;
        }
        const int per = a.CB * a.dq_rb;
        const int item = b / per, rem = b - item * per;
        const int ty = rem / a.CB;
        binary_dequant_body<FUSED_NW>(ride.it[item], a.N, a.C, a.dq_R, rem - ty * a.CB, ty);
    }
}

// ---------------------------------------------------------------------------------------------------
// Software-pipelined 1-bit exchange step: ONE launch carries three independent groups of workgroups,
//   finalize(layer j+1)  |  stats + sign bits(layer j+2)  |  dequant + add(layer j)
// so the two small latency-bound kernels of the compress sequence run underneath the bandwidth-bound reconstruction of
// an earlier layer instead of in front of it (cfx_plan_run_pipelined builds the schedule; each group runs exactly the
// code of its stand-alone kernel, so results are bit-identical).  The latency-critical finalize blocks come first in
// dispatch order.  Any group may be empty (pipeline prologue / epilogue).
// ---------------------------------------------------------------------------------------------------
#ifndef PIPE_US
#define PIPE_US 2      // rows in flight per wave in the fused kernel's stats group (register budget of 8 waves / SIMD)
#endif
#define PIPE_MAX_DQ CFX_PIPE_MAX_DQ
struct BatchDX { cfx_decomp_item it[PIPE_MAX_DQ]; };
struct PipeArgs {
    int N, C, CB;
    int n_fin, fin_bpi, fin_P;          // finalize: blocks in the group, blocks per tensor, partials per column to reduce
    int n_st, st_R, st_P;               // stats: blocks in the group, tile height, row blocks per tensor (grid CB x st_P)
    int dq_R, dq_rb;                    // dequant: tile height, row blocks per tensor
    const u64* ws_fin;
    u64* ws_st;
    size_t ws_stride;
};
// STEADY only names the launch for profilers: steady-state launches (three equally sized units) and the pipeline's
// prologue / epilogue / ragged-unit launches show up as two kernels in rocprofv3 --stats, with separate averages.
template <bool STEADY>
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_binary_pipe(BatchDX dq, BatchC fin, BatchC st, PipeArgs a) {
    __shared__ u64 sm[WAVES][TILE_C];
    int b = blockIdx.x;
    if (b < a.n_fin) {
        const int item = b / a.fin_bpi;
        absmean_finalize_body<NTHR>(fin.it[item], a.N, a.C, a.CB, a.fin_P, 8, 0, a.ws_fin + (size_t)item * a.ws_stride, b - item * a.fin_bpi, &sm[0][0]);
        return;
    }
    b -= a.n_fin;
    if (b < a.n_st) {
        const int per = a.CB * a.st_P;
        const int item = b / per, rem = b - item * per;
        const int ty = rem / a.CB;
        absmean_stats_body<true, PIPE_US>(st.it[item], a.N, a.C, a.st_R, a.CB, rem - ty * a.CB, ty, a.ws_st + (size_t)item * a.ws_stride, sm);
        return;
    }
    b -= a.n_st;
    const int per = a.CB * a.dq_rb;
    const int item = b / per, rem = b - item * per;
    const int ty = rem / a.CB;
    binary_dequant_body<WAVES>(dq.it[item], a.N, a.C, a.dq_R, rem - ty * a.CB, ty);
}

// ---------------------------------------------------------------------------------------------------
// 2-bit quantise (+EF)            replaces _int2_quant_fastpath (fastpath.py:486-580)
// ---------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(NTHR) void k_int2_quant(BatchC batch, int N, int C, int R, int flags) {
    const cfx_comp_item it = batch.it[blockIdx.z];
    const TileCoord t = tile_coord(N, C, R);
    const int C4 = C >> 2;
    unsigned char* pk = (unsigned char*)it.packet;
    const h16* TOK = (const h16*)(pk + (size_t)N * C4);
    const h16* CH = TOK + N;
    const h16* x = (const h16*)it.x;
    const h16* base = (const h16*)it.base;
    h16* nb = (h16*)it.new_base;
    const bool upd = (flags & CFX_FLAG_UPDATE_CACHE) && nb;
    const bool ef = !(flags & CFX_FLAG_NO_EF);
    const bool al16 = (((uintptr_t)CH) & 15) == 0;
    h16x8 ch8 = (h16x8)(h16)0;
    if (t.act) ch8 = ld8_tail(CH + t.c, al16);

    for (int r = t.r0 + t.w; r < t.r1; r += WAVES * UNROLL) {
        h16x8 xv[UNROLL], bv[UNROLL];
        h16 tk[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const int rr = r + WAVES * j;
            xv[j] = (h16x8)(h16)0; bv[j] = (h16x8)(h16)0; tk[j] = (h16)0;
            if (rr < t.r1 && t.act) {
                xv[j] = ld8nt(x + (size_t)rr * C + t.c);
                if (base) bv[j] = ld8nt(base + (size_t)rr * C + t.c);
                tk[j] = TOK[rr];
            }
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const int rr = r + WAVES * j;
            if (rr < t.r1 && t.act) {
                const h16x8 d = xv[j] - bv[j];
                const h16x8 thr = ch8 * tk[j];                                   // fastpath.py:536
                const h16x8 a = habs8(d);
                unsigned code = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned s = d[i] >= (h16)0 ? 1u : 0u;                 // fastpath.py:539
                    const unsigned m = a[i] > thr[i] ? 1u : 0u;                  // fastpath.py:540
                    code |= ((s << 1) | m) << (2 * i);
                }
                *reinterpret_cast<u16*>(pk + (size_t)rr * C4 + (t.c >> 2)) = (u16)code;
                if (upd) {
                    h16x8 o;
                    if (ef) {
                        const h16x8 recv = int2_recv((u16)code, thr);
                        o = base ? (bv[j] + recv) : recv;
                    } else {
                        o = xv[j];
                    }
                    st8nt(nb + (size_t)rr * C + t.c, o);
                }
            }
        }
    }
}

// 2-bit dequant + base add        replaces _int2_dequant_fastpath (fastpath.py:672-741)
__global__ __launch_bounds__(NTHR) void k_int2_dequant(BatchD batch, int N, int C, int R, unsigned* pre, unsigned pre_val) {
    // lane: publish `pre` first - the launch in front of this one in the stream (the previous peer's reconstruction) has finished
    if (pre && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0) st_wt(pre, pre_val);
    const cfx_decomp_item it = batch.it[blockIdx.z];
    const TileCoord t = tile_coord(N, C, R);
    const int C4 = C >> 2;
    const unsigned char* pk = (const unsigned char*)it.packet;
    const h16* TOK = (const h16*)(pk + (size_t)N * C4);
    const h16* CH = TOK + N;
    const h16* base = (const h16*)it.base;
    h16* out = (h16*)it.recon;
    const bool al16 = (((uintptr_t)CH) & 15) == 0;
    h16x8 ch8 = (h16x8)(h16)0;
    if (t.act) ch8 = ld8_tail(CH + t.c, al16);

    for (int r = t.r0 + t.w; r < t.r1; r += WAVES * UNROLL) {
        h16x8 bv[UNROLL];
        u16 cd[UNROLL];
        h16 tk[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const int rr = r + WAVES * j;
            bv[j] = (h16x8)(h16)0; cd[j] = 0; tk[j] = (h16)0;
            if (rr < t.r1 && t.act) {
                if (base) bv[j] = ld8nt(base + (size_t)rr * C + t.c);
                cd[j] = *reinterpret_cast<const u16*>(pk + (size_t)rr * C4 + (t.c >> 2));
                tk[j] = TOK[rr];
            }
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const int rr = r + WAVES * j;
            if (rr < t.r1 && t.act) {
                const h16x8 thr = ch8 * tk[j];
                const h16x8 recv = int2_recv(cd[j], thr);
                st8nt(out + (size_t)rr * C + t.c, base ? (bv[j] + recv) : recv);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 2-bit layer in ONE launch (cfx_compress_batch_gated, codec INT2): two groups of workgroups, two arrival gates
//   S  statistics + in-launch finalize of the rank's own tensors (absmean_fused_body; scales published write-through -> gate 1);
//      every S workgroup then waits for gate 1 itself - the codes depend on the scales - and quantises ITS tile from the registers
//      it loaded for the statistics (own_tile_finish): codes published write-through, error feedback                  -> gate 2
//   D  reconstruction of the looped-back peers: state tiles pulled into registers, wait for gate 2, finish from registers
// Dispatch order S, D: the S workgroups are all resident before any D workgroup and wait only on each other's arrivals, which
// never block; D waits only on S.  Arithmetic = k_int2_quant / k_int2_dequant.  (A separate quantise group re-reading x and the
// state was slower than three launches: its preload and the late D workgroups' burst landed on the reduction tail.  Two launches -
// statistics + finalize alone, then quantise + gated reconstruction - measured 2.08 vs 2.04 ms per step for this form.)
// ---------------------------------------------------------------------------------------------------
template <int NW, int KR, int KL>
__device__ __forceinline__ void int2_dequant_gated_body(const cfx_decomp_item& it, int N, int C, int R, int tile_x, int tile_y,
                                                       unsigned* gate, unsigned expect, unsigned* err, long long timeout, u32x4* lds,
                                                       unsigned* xgate = nullptr, unsigned xexpect = 0, bool remote = false,
                                                       bool tagged = false, TagArena ta = TagArena(), int Rs = 0, u16* s16 = nullptr) {
    constexpr int K = KR + KL;
    const TileCoord t = tile_coord_at(tile_x, tile_y, N, C, R);
    const int C4 = C >> 2;
    const unsigned char* pk = (const unsigned char*)it.packet;
    const u16* TOK = (const u16*)(pk + (size_t)N * C4);
    const u16* CH = TOK + N;
    const h16* base = (const h16*)it.base;
    h16* out = (h16*)it.recon;
    const int cc = min(t.c, C - 8);
    h16x8 bv[KR];
    if (base) {
#pragma unroll
        for (int j = 0; j < KR; ++j) bv[j] = ld8nt(base + (size_t)min(t.r0 + t.w + NW * j, t.r1 - 1) * C + cc);
        if constexpr (KL > 0) {
            h16x8 tl[KL > 0 ? KL : 1];
#pragma unroll
            for (int j = 0; j < KL; ++j) tl[j] = ld8nt(base + (size_t)min(t.r0 + t.w + NW * (KR + j), t.r1 - 1) * C + cc);
#pragma unroll
            for (int j = 0; j < KL; ++j) lds[j * (NW * 64) + threadIdx.x] = __builtin_bit_cast(u32x4, tl[j]);   // read back by the same thread
        }
    } else {
#pragma unroll
        for (int j = 0; j < KR; ++j) bv[j] = (h16x8)(h16)0;
    }
    h16x8 ch8;
    u16 ul;
    u16 cd[K];
    if (tagged) {
        // the packet is one of THIS launch's: wait for the statistics tiles whose codes this tile reads (their flags carry the launch's
        // tag once the codes are in memory) - not for the slowest tile of the launch -, then the scales' tagged copies (complete by then:
        // a tile quantises only after it has seen them)
        bool failed = false;
        if (t.w == 0) {
            const int P = (N + Rs - 1) / Rs, by0 = t.r0 / Rs, by1 = (t.r1 - 1) / Rs;
            const u64* f = ta.tdone + (size_t)tile_x * P;
            SpinClock clk;
            for (;;) {
                const u64 v = by0 + t.lane <= by1 ? ld_wt(f + by0 + t.lane) : ta.tagbits;
                if (__builtin_amdgcn_ballot_w64(!tag_is(v, ta.tagbits)) == 0) break;
                __builtin_amdgcn_s_sleep(INT2_FLAG_SLEEP);
                if (clk.expired(timeout)) { failed = true; if (t.lane == 0) gate_fail(err); break; }
            }
        }
        if (__syncthreads_or(failed ? 1 : 0)) return;
#pragma unroll
        for (int j = 0; j < K; ++j) cd[j] = ld_wt((const u16*)(pk + (size_t)min(t.r0 + t.w + NW * j, t.r1 - 1) * C4) + (cc >> 3));
        if (!scales_from_tagged<NW * K, false>(ta, N, C, tile_x * TILE_C, t.r0, s16, timeout, err)) return;     // (one round trip with the codes')
        ch8 = __builtin_bit_cast(h16x8, *(const u16x8*)(s16 + 8 * t.lane));
        ul = s16[TILE_C + min(t.w + NW * min(t.lane, K - 1), t.r1 - 1 - t.r0)];
    } else if (!(xgate ? gate_wait<true>(xgate, xexpect, err, timeout) : gate_wait(gate, expect, err, timeout))) return;
    else if (remote) {                                            // (uniform) the packet sits in a peer GPU's memory: system-scope loads
        u16x8 vb;
#pragma unroll
        for (int i = 0; i < 8; ++i) vb[i] = ld_sys(CH + cc + i);
        ch8 = __builtin_bit_cast(h16x8, vb);
        ul = ld_sys(TOK + min(t.r0 + t.w + NW * min(t.lane, K - 1), t.r1 - 1));
#pragma unroll
        for (int j = 0; j < K; ++j) cd[j] = ld_sys((const u16*)(pk + (size_t)min(t.r0 + t.w + NW * j, t.r1 - 1) * C4) + (cc >> 3));
    } else {
        ch8 = ld8_wt(CH + cc);
        ul = ld_wt(TOK + min(t.r0 + t.w + NW * min(t.lane, K - 1), t.r1 - 1));
#pragma unroll
        for (int j = 0; j < K; ++j) cd[j] = ld_wt((const u16*)(pk + (size_t)min(t.r0 + t.w + NW * j, t.r1 - 1) * C4) + (cc >> 3));
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int rr = t.r0 + t.w + NW * j;
        const h16 tk = hfrom((u16)__builtin_amdgcn_readlane((int)ul, j));
        if (rr < t.r1 && t.act) {
            const h16x8 thr = ch8 * tk;
            const h16x8 recv = int2_recv(cd[j], thr);
            h16x8 bj = (h16x8)(h16)0;
            if (j < KR) bj = bv[j < KR ? j : 0];
            else if (base) bj = __builtin_bit_cast(h16x8, lds[(j - KR) * (NW * 64) + threadIdx.x]);
            st8nt(out + (size_t)rr * C + t.c, base ? (bj + recv) : recv);
        }
    }
}

#ifndef INT2_D_TAGGED
#define INT2_D_TAGGED 1
#endif
struct Int2LayerArgs {
    int N, C, CB, R, P, n_st;           // group S: CB x P tiles of R rows per own tensor
    int g_R, g_rb, n_g;                 // group D
    int flags;
    u64* ws;
    size_t ws_stride;
    unsigned* tick;
    unsigned* gate1; unsigned expect1;
    unsigned* gate2; unsigned expect2;
    unsigned* err;
    long long timeout;                     // in-launch waits: ticks of the 100 MHz wall clock
    u64* tarena; size_t tarena_stride; unsigned tag;      // the context's tagged-words arena, the launch's 24-bit tag
    signed char src[CFX_MAX_BATCH];                        // gated item -> the own tensor of this launch whose packet it reads, or -1
    unsigned* xgate; unsigned xexpect;     // external gate for group D (exchange-layer op): NULL = group D waits on gate2
    int remote;                            // group D's packets may sit in a peer GPU's memory
    P2PInline p2p;                         // own != NULL: workgroup 0 runs the peer-to-peer exchange and opens xgate itself
};
template <int US>
__global__ __launch_bounds__(FUSED_NT, 4) void k_int2_compress_gated(BatchC batch, BatchD gated, Int2LayerArgs a) {
    // the statistics group: FUSED_NW rows + 4 (a parked row of x, of the state); the reconstruction group: GATE_LDS_ROWS of parked state + 1 of scales
    __shared__ u64 sm[(GATE_LDS_ROWS > FUSED_NW + 4 ? GATE_LDS_ROWS : FUSED_NW + 4) + 1][TILE_C];
    int b = blockIdx.x;
    if (b < a.n_st) {
        const int per = a.CB * a.P;
        const int z = b / per, rem = b - z * per;
        const int by = rem / a.CB;
        absmean_fused_body<false, US, true, true>(batch.it[z], a.N, a.C, a.R, a.CB, a.P, rem - by * a.CB, by, a.ws + (size_t)z * a.ws_stride,
                                                  a.tick + z * TICK_WORDS, 4, 1, sm, 0, Probe(), a.gate1, a.expect1, a.flags, a.gate2, a.expect2, a.err, a.timeout,
                                                  tag_arena_of(a.tarena, a.tarena_stride, z, a.N, a.C, a.CB, a.P, a.tag));
        // (packets complete = the codes gate's last arriver has written the "open" words: XCD 0's)
        if (b == 0 && a.p2p.own) p2p_exchange_inline(a.gate2 + 1 * GATE_LINE, a.expect2, 1, a.p2p, a.xgate, a.xexpect, a.err);
        return;
    }
    b -= a.n_st;
    const int per = a.CB * a.g_rb;
    const int item = b / per, rem = b - item * per;
    const int ty = rem / a.CB;
    const int sz = a.src[item];
    const TagArena ta = tag_arena_of(a.tarena, a.tarena_stride, sz >= 0 ? sz : 0, a.N, a.C, a.CB, a.P, a.tag);
    // (s16: behind the KL rows of state the workgroup parks in LDS)
    int2_dequant_gated_body<FUSED_NW, GATE_KR2, GATE_KL>(gated.it[item], a.N, a.C, a.g_R, rem - ty * a.CB, ty, a.gate2, a.expect2, a.err, a.timeout,
                                                        (u32x4*)&sm[0][0], a.xgate, a.xexpect, a.remote != 0 && sz < 0, INT2_D_TAGGED && sz >= 0, ta, a.R,
                                                        (u16*)&sm[GATE_LDS_ROWS][0]);
}

// ---------------------------------------------------------------------------------------------------
// host side: this family's launches (validated and dispatched by cfx_api.hip)
// ---------------------------------------------------------------------------------------------------

// The gated form can run as one launch when: 1-bit codec, in-launch finalize on, rows of sign bits 16-byte aligned (C % 128 == 0),
// tiles of at most FUSED_NW * GATE_KR (1-bit) / FUSED_NW * (GATE_KR2 + GATE_KL) (2-bit) rows cover the tensor with few enough workgroups to matter.  Otherwise the same work runs as
// compress + one reconstruction launch (identical results).
static bool gated_one_launch(cfx_ctx* ctx, int codec, int C, int CB) {
    return (codec == CFX_CODEC_BINARY || codec == CFX_CODEC_INT2) && ctx->fused && CB <= TICK_MAX_CB && C % 128 == 0 && !ctx->dev_probe && ctx->gated_on;
}

// The tagged-partials arena of a ring (= a stream) for the abs-mean layer launches: context-owned because its words are TAGGED - a stale
// word must never carry a tag a later launch expects, so it starts zeroed and only ever takes this context's tags, handed out in
// sequence (24 bits, never 0; when they wrap the arenas are zeroed again, in stream order).  Returns the launch's tag, 0 on failure.
static unsigned abs_arena_for(cfx_ctx* ctx, unsigned ring, void* stream, size_t need_bytes, u64** arena) {
    hipStream_t s = (hipStream_t)stream;
    if (!ctx->mml_arena_owned[ring] || ctx->mml_arena_owner[ring] != stream) {
        // the ring - and with it its arenas - changed hands: whatever its previous owner still has in flight reads them.  Rare; wait for it
        if (ctx->mml_arena_owned[ring]) (void)hipDeviceSynchronize();
        ctx->mml_arena_owner[ring] = stream;
        ctx->mml_arena_owned[ring] = true;
    }
    if (ctx->abs_arena_bytes[ring] < need_bytes) {
        if (ctx->abs_arena[ring]) (void)hipFree(ctx->abs_arena[ring]);      // (synchronises the device: no launch still reads it)
        ctx->abs_arena[ring] = nullptr;
        ctx->abs_arena_bytes[ring] = 0;
        const size_t cap = (std::max(need_bytes, (size_t)2 << 20) + 4095) & ~(size_t)4095;
        void* m = nullptr;
        if (hipMalloc(&m, cap) != hipSuccess || hipMemsetAsync(m, 0, cap, s) != hipSuccess) {
            (void)hipGetLastError();
            if (m) (void)hipFree(m);
            return 0;
        }
        ctx->abs_arena[ring] = (u64*)m;
        ctx->abs_arena_bytes[ring] = cap;
    }
    unsigned tag = ++ctx->abs_seq & 0xFFFFFFu;
    if (tag == 0) {
        // 16.7 million launches later: a word a smaller layout has not touched since could carry a tag again - start over
        (void)hipDeviceSynchronize();
        for (int i = 0; i < CFX_RING_STREAMS; ++i)
            if (ctx->abs_arena[i]) (void)hipMemset(ctx->abs_arena[i], 0, ctx->abs_arena_bytes[i]);
        (void)hipDeviceSynchronize();
        tag = ++ctx->abs_seq & 0xFFFFFFu;
    }
    *arena = ctx->abs_arena[ring];
    return tag;
}

int cfx_i_absmean_compress(CompressCall& cc) {
    cfx_ctx* ctx = cc.ctx;
    const int codec = cc.codec, N = cc.N, C = cc.C, param = cc.param, flags = cc.flags, batch = cc.batch, n_ride = cc.n_ride, CB = cc.CB;
    int n_gated = cc.n_gated;
    const cfx_comp_item* items = cc.items;
    const cfx_decomp_item* gated = cc.gated;
    void* stream = cc.stream;
    hipStream_t s = (hipStream_t)stream;
    CfxXGate* xg = cc.xg;
    BatchC b = cc.b;
    BatchD rd = cc.rd, gd = cc.gd;
    u64* ws = cc.ws;
    const size_t wstride = cc.wstride;
    const bool upd = cc.upd, capturing = cc.capturing;
    (void)param; (void)n_ride; (void)items; (void)gated; (void)rd; (void)ws; (void)wstride; (void)upd; (void)capturing; (void)xg; (void)gd;
    const bool fused = cc.fused;
    unsigned* tick = cc.tick;
    const unsigned slot = cc.slot;
    const int stream_cus = cc.stream_cus, R = cc.R, P = cc.P;
    (void)tick; (void)slot;
    // one launch only for explicit gated items: folding the error-feedback update of a PLAIN compress call into the launch the same
    // way was measured slower (K,V of the FLUX shard: 20.8 vs 19.2 us 1-bit, 19.9 vs 18.6 us 2-bit) - the gate hop and the write tail
    // cost more than the kernel boundary they replace when only two tensors wait behind the gate
    // (2-bit: the tile stays in registers - exactly one trip of the row loop; 1-bit: any whole number of trips)
    bool one_launch = n_gated && !capturing && gated_one_launch(ctx, codec, C, CB) &&
                      (R == FUSED_NW * 4 || (codec == CFX_CODEC_BINARY && R % (FUSED_NW * 4) == 0));
    if (one_launch && codec == CFX_CODEC_INT2) {
        // the 2-bit layer launch needs every statistics workgroup CO-RESIDENT (each waits at gate 1 for all the others' partial sums
        // while holding its tile in registers): only when they fit the CUs this stream may use, otherwise the multi-launch form
        static int per_cu = 0;
        if (!per_cu && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_int2_compress_gated<4>, FUSED_NT, 0) != hipSuccess || per_cu < 1)) {
            (void)hipGetLastError();
            per_cu = 1;
        }
        if ((long)CB * P * batch > (long)per_cu * stream_cus - 8) one_launch = false;
    }
    // the 1-bit layer launch needs no co-residency (its statistics workgroups never wait), but its gated workgroups spin on slots the
    // statistics group needs when the stream has few CUs: a CU-masked lane runs the multi-launch form
    if (one_launch && stream_cus < 128) one_launch = false;
    const bool absmean_codec = codec == CFX_CODEC_BINARY || codec == CFX_CODEC_INT2;     // (the min/max codecs decide in their own branch below)
    if (xg && absmean_codec && !(one_launch && fused && (codec == CFX_CODEC_INT2 || !upd || (!(flags & CFX_FLAG_NO_EF) && n_gated + batch <= CFX_MAX_BATCH)))) {
        n_gated = 0;            // compress only: the caller runs its collective and the reconstruction behind this launch
        one_launch = false;
    }
    if (xg && xg->needs_room && one_launch) {
        // A collective KERNEL has to be placed while the reconstruction group waits for it.  Once the compress group has gone, the group's
        // n_g workgroups are all that is left of this launch; if they leave 32 workgroup slots of the stream's CUs free, at least 16 CUs
        // hold at most one of them - 320 free VGPRs per SIMD and 128 KB of LDS there, room for a workgroup of RCCL's kernel (256 threads x
        // 280 VGPRs, 20 KB) - and nothing of the launch is pending that could take those slots.  Otherwise: two launches.
        static int per_cu = 0;
        if (!per_cu && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_absmean_compress<true, 4, true>, FUSED_NT, 0) != hipSuccess || per_cu < 1)) {
            (void)hipGetLastError();
            per_cu = 1;
        }
        const long n_g_all = (long)CB * ((N + FUSED_NW * GATE_KR - 1) / (FUSED_NW * GATE_KR)) * (n_gated + (upd ? batch : 0));
        if (n_g_all + 32 > (long)per_cu * stream_cus) {
            n_gated = 0;
            one_launch = false;
        }
    }
    const dim3 grid(CB, P, batch);
    const int Rq = auto_rows(ctx, N, C, batch, true);       // apply passes: same tile map as the (unfused) statistics pass
    const dim3 gridq(CB, (N + Rq - 1) / Rq, batch);
    (void)gridq;
    const int per_byte = codec == CFX_CODEC_BINARY ? 8 : 4;
    if (one_launch && codec == CFX_CODEC_INT2) {
        Int2LayerArgs a;
        memset(&a, 0, sizeof(a));
        a.N = N; a.C = C; a.CB = CB; a.R = R; a.P = P; a.n_st = CB * P * batch;
        a.g_rb = (N + FUSED_NW * (GATE_KR2 + GATE_KL) - 1) / (FUSED_NW * (GATE_KR2 + GATE_KL));
        a.g_R = ((N + a.g_rb - 1) / a.g_rb + FUSED_NW - 1) / FUSED_NW * FUSED_NW;
        a.n_g = CB * a.g_rb * n_gated;
        a.flags = flags; a.ws = ws; a.ws_stride = wstride; a.tick = tick;
        a.gate1 = ctx->gate + (size_t)slot * GATE_STRIDE;
        a.gate2 = a.gate1 + GATE_BLOCK;
        ctx->gate_expect[3 * slot] += (unsigned)batch * (unsigned)(CB + 1);
        ctx->gate_expect[3 * slot + 1] += (unsigned)a.n_st;
        a.expect1 = ctx->gate_expect[3 * slot]; a.expect2 = ctx->gate_expect[3 * slot + 1];
        a.err = ctx->gate_err;
        a.timeout = ctx->gate_timeout;
        a.tarena_stride = tag_arena_words(N, C, CB, P);
        a.tag = abs_arena_for(ctx, slot / TICK_RING, stream, a.tarena_stride * batch * sizeof(u64), &a.tarena);
        if (!a.tag) return fail(ctx, CFX_ERR_LAUNCH, "2-bit layer launch: cannot allocate the partials' arena");
        for (int g_ = 0; g_ < n_gated; ++g_) {
            a.src[g_] = -1;
            for (int i = 0; i < batch; ++i)
                if (gd.it[g_].packet == items[i].packet) a.src[g_] = (signed char)i;
        }
        if (xg) {
            a.xgate = a.gate1 + 2 * GATE_BLOCK;
            a.xexpect = ++ctx->gate_expect[3 * slot + 2];
            a.remote = xg->remote;
            fill_p2p(ctx, xg, a.p2p);
            xg->taken = 1;
            xg->p_gate = a.gate2 + 1 * GATE_LINE;      // the word the codes gate's last arriver writes for XCD 0
            xg->p_expect = a.expect2;
            xg->f_gate = a.xgate; xg->f_expect = a.xexpect;
        }
        const dim3 g(a.n_st + a.n_g);
        LAUNCH(ctx, KID_ABSMEAN_COMPRESS_GATED, s, (k_int2_compress_gated<4>), g, dim3(FUSED_NT), 0, s, b, gd, a);
        return check_launch(ctx, "2-bit layer launch");
    }
    // 1-bit, one launch, CFX_FLAG_UPDATE_CACHE: the error-feedback update is the receiver's reconstruction of our own packet onto our
    // own state - it joins the gated items (when they all fit one launch)
    const bool ef_gated = one_launch && codec == CFX_CODEC_BINARY && upd && !(flags & CFX_FLAG_NO_EF) && n_gated + batch <= CFX_MAX_BATCH;
    const bool one_launch_1bit = one_launch && codec == CFX_CODEC_BINARY && (!upd || ef_gated);
    int n_gated_k = n_gated;
    if (ef_gated) {
        // own tensors FIRST: their workgroups are resident from the start and pull the state tiles the statistics group is reading
        // anyway (measured: 1.60 vs 1.71 ms per step with them last)
        for (int i = n_gated - 1; i >= 0; --i) gd.it[i + batch] = gd.it[i];
        for (int i = 0; i < batch; ++i) { gd.it[i].packet = items[i].packet; gd.it[i].base = items[i].base; gd.it[i].recon = items[i].new_base; }
        n_gated_k = n_gated + batch;
    }
    if (fused) {
        FusedArgs a;
        memset(&a, 0, sizeof(a));
        a.N = N; a.C = C; a.CB = CB; a.R = R; a.P = P;
        a.n_st = CB * P * batch;
        a.dq_R = FUSED_NW * UNROLL;
        a.dq_rb = (N + a.dq_R - 1) / a.dq_R;
        a.per_byte = per_byte; a.eps_mode = codec == CFX_CODEC_INT2 ? 1 : 0;
        a.ws = ws; a.ws_stride = wstride; a.tick = tick;
        a.dbg = ctx->dev_probe;
        a.probe = cfx_i_probe(ctx);
        if (one_launch_1bit) {
            // tiles of the gated group: as few row blocks as GATE_KR rows per wave allow, heights a multiple of FUSED_NW
            a.g_rb = (N + FUSED_NW * GATE_KR - 1) / (FUSED_NW * GATE_KR);
            a.g_R = ((N + a.g_rb - 1) / a.g_rb + FUSED_NW - 1) / FUSED_NW * FUSED_NW;
            a.n_gt = a.n_g = CB * a.g_rb * n_gated_k;
            a.gate = ctx->gate + (size_t)slot * GATE_STRIDE;
            ctx->gate_expect[3 * slot] += (unsigned)batch * (unsigned)(CB + 1);
            a.gate_expect = ctx->gate_expect[3 * slot];
            a.tarena_stride = tag_arena_words(N, C, CB, P);
            a.tag = abs_arena_for(ctx, slot / TICK_RING, stream, a.tarena_stride * batch * sizeof(u64), &a.tarena);
            if (!a.tag) return fail(ctx, CFX_ERR_LAUNCH, "1-bit layer launch: cannot allocate the partials' arena");
            // a gated item whose packet is one of this launch's own takes its scales from the tagged words (no gate, with live peers or
            // without: the own error-feedback update never waits for a peer); anybody else's packet waits for the gate
            for (int g_ = 0; g_ < n_gated_k; ++g_) {
                a.src[g_] = -1;
                for (int i = 0; i < batch; ++i)
                    if (gd.it[g_].packet == items[i].packet) a.src[g_] = (signed char)i;
            }
            if (xg) {
                a.xgate = a.gate + GATE_BLOCK;                    // the slot's second gate block (the 2-bit layer launch's gate 2)
                a.xexpect = ++ctx->gate_expect[3 * slot + 1];
                a.remote = xg->remote;
                fill_p2p(ctx, xg, a.p2p);
                xg->taken = 1;
                xg->p_gate = a.gate; xg->p_expect = a.gate_expect;
                xg->f_gate = a.xgate; xg->f_expect = a.xexpect;
            }
        }
        a.gate_err = ctx->gate_err;
        a.timeout = ctx->gate_timeout;
        const dim3 g(a.n_st + a.n_g + CB * a.dq_rb * n_ride);
#ifdef CFX_DEV_PROBES
        if (one_launch_1bit && ctx->dev_buf && R % 32 == 0) {          // (the instantiation whose reconstruction tiles drain before they stamp)
            LAUNCH(ctx, KID_ABSMEAN_COMPRESS_GATED, s, (k_absmean_compress<true, 4, true, true>), g, dim3(FUSED_NT), 0, s, b, rd, gd, a);
        } else
#endif
        if (one_launch_1bit) {
            LAUNCH(ctx, KID_ABSMEAN_COMPRESS_GATED, s, (k_absmean_compress<true, 4, true>), g, dim3(FUSED_NT), 0, s, b, rd, gd, a);
        } else if (codec == CFX_CODEC_BINARY) {
            if (R % 32 == 0) LAUNCH(ctx, KID_ABSMEAN_COMPRESS_BITS, s, (k_absmean_compress<true, 4>), g, dim3(FUSED_NT), 0, s, b, rd, gd, a);
            else LAUNCH(ctx, KID_ABSMEAN_COMPRESS_BITS, s, (k_absmean_compress<true, 2>), g, dim3(FUSED_NT), 0, s, b, rd, gd, a);
        } else {
            if (R % 32 == 0) LAUNCH(ctx, KID_ABSMEAN_COMPRESS, s, (k_absmean_compress<false, 4>), g, dim3(FUSED_NT), 0, s, b, rd, gd, a);
            else LAUNCH(ctx, KID_ABSMEAN_COMPRESS, s, (k_absmean_compress<false, 2>), g, dim3(FUSED_NT), 0, s, b, rd, gd, a);
        }
    } else {
        if (codec == CFX_CODEC_BINARY) LAUNCH(ctx, KID_ABSMEAN_STATS_BITS, s, k_absmean_stats<true>, grid, dim3(NTHR), 0, s, b, N, C, R, ws, wstride);
        else LAUNCH(ctx, KID_ABSMEAN_STATS, s, k_absmean_stats<false>, grid, dim3(NTHR), 0, s, b, N, C, R, ws, wstride);
        LAUNCH(ctx, KID_ABSMEAN_FINALIZE, s, k_absmean_finalize, dim3(1 + (C + 255) / 256, batch), dim3(1024), 0, s, b, N, C, CB, P, per_byte,
                           codec == CFX_CODEC_INT2 ? 1 : 0, (const u64*)ws, wstride);
        if (n_ride) {
            const int Rr = auto_rows(ctx, N, C, n_ride, false);
            LAUNCH(ctx, KID_BINARY_DEQUANT, s, k_binary_dequant<UNROLL>, dim3(CB, (N + Rr - 1) / Rr, n_ride), dim3(NTHR), 0, s, rd, N, C, Rr, (unsigned*)nullptr, 0u);
        }
    }
    if (codec == CFX_CODEC_INT2) {
        LAUNCH(ctx, KID_INT2_QUANT, s, k_int2_quant, gridq, dim3(NTHR), 0, s, b, N, C, Rq, flags);
    } else if (upd && !one_launch_1bit) {       // (one launch: the statistics workgroups did it from registers)
        if (flags & CFX_FLAG_NO_EF) {
            for (int i = 0; i < batch; ++i)
                if (items[i].new_base != items[i].x)
                    // a copy KERNEL, not hipMemcpyAsync: x may have been produced on another stream and handed over by a flag (exchange
                    // lane), and only kernels take the acquire that makes such a hand-off visible (tools/flag_coherence_probe.hip)
                    hipLaunchKernelGGL(k_copy_probe, dim3(2048), dim3(256), 0, s, (uint4*)items[i].new_base, (const uint4*)items[i].x, (size_t)N * C * 2 / 16);
        } else {
            // error-feedback update == the receiver's dequant+add on our own packet
            BatchD d;
            memset(&d, 0, sizeof(d));
            for (int i = 0; i < batch; ++i) { d.it[i].packet = items[i].packet; d.it[i].base = items[i].base; d.it[i].recon = items[i].new_base; }
            LAUNCH(ctx, KID_BINARY_EF, s, k_binary_dequant<UNROLL>, gridq, dim3(NTHR), 0, s, d, N, C, Rq, (unsigned*)nullptr, 0u);
        }
    }
    if (n_gated && !(one_launch_1bit || (one_launch && codec == CFX_CODEC_INT2))) {
        const int Rg = auto_rows(ctx, N, C, n_gated, false);
        if (codec == CFX_CODEC_BINARY) LAUNCH(ctx, KID_BINARY_DEQUANT, s, k_binary_dequant<UNROLL>, dim3(CB, (N + Rg - 1) / Rg, n_gated), dim3(NTHR), 0, s, gd, N, C, Rg, (unsigned*)nullptr, 0u);
        else LAUNCH(ctx, KID_INT2_DEQUANT, s, k_int2_dequant, dim3(CB, (N + Rg - 1) / Rg, n_gated), dim3(NTHR), 0, s, gd, N, C, Rg, (unsigned*)nullptr, 0u);
    }
    return check_launch(ctx, "compress launch");
}

int cfx_i_absmean_decompress(cfx_ctx* ctx, int codec, int N, int C, int batch, const BatchD& b, int R, void* stream, unsigned* pre, unsigned pre_val) {
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((C + TILE_C - 1) / TILE_C, (N + R - 1) / R, batch);
    if (codec == CFX_CODEC_BINARY) {
        if (stream_cu_count(ctx, stream) < 128) {
            // a CU-masked lane is bound by the bytes each CU keeps in flight: 4 rows per wave instead of 2
            const int R4 = WAVES * 4;
            LAUNCH(ctx, KID_BINARY_DEQUANT, s, k_binary_dequant<4>, dim3(grid.x, (N + R4 - 1) / R4, batch), dim3(NTHR), 0, s, b, N, C, R4, pre, pre_val);
        } else LAUNCH(ctx, KID_BINARY_DEQUANT, s, k_binary_dequant<UNROLL>, grid, dim3(NTHR), 0, s, b, N, C, R, pre, pre_val);
    } else LAUNCH(ctx, KID_INT2_DEQUANT, s, k_int2_dequant, grid, dim3(NTHR), 0, s, b, N, C, R, pre, pre_val);
    return check_launch(ctx, "decompress launch");
}

extern "C" {

// The 2-bit quantise kernel ALONE, scales given: the packet tail already holds tok (N halves) and chan (C halves) - the Triton kernel
// _int2_quant_fastpath (fastpath.py:486-580) as the reference launches it after its eager scale prologue.
int cfx_int2_quantize(cfx_ctx* ctx, int N, int C, int flags, int batch, const cfx_comp_item* items, void* stream) {
    if (!ctx || !items) return fail(ctx, CFX_ERR_NULL, "int2_quantize: null ctx/items");
    if (batch < 1 || batch > CFX_MAX_BATCH) return fail(ctx, CFX_ERR_BATCH, "int2_quantize: batch out of range");
    if (!shape_ok(CFX_CODEC_INT2, N, C, 0)) return fail(ctx, CFX_ERR_SHAPE, "int2_quantize: bad shape");
    const bool upd = flags & CFX_FLAG_UPDATE_CACHE;
    BatchC b;
    memset(&b, 0, sizeof(b));
    for (int i = 0; i < batch; ++i) {
        if (!items[i].x || !items[i].packet) return fail(ctx, CFX_ERR_NULL, "int2_quantize: null x/packet");
        if (upd && !items[i].new_base) return fail(ctx, CFX_ERR_NULL, "int2_quantize: UPDATE_CACHE needs new_base");
        if (!AL16(items[i].x) || !AL16(items[i].base) || !AL16(items[i].new_base) || !AL16(items[i].packet))
            return fail(ctx, CFX_ERR_ALIGN, "int2_quantize: pointers must be 16-byte aligned");
        b.it[i] = items[i];
    }
    hipStream_t s = (hipStream_t)stream;
    const int Rq = auto_rows(ctx, N, C, batch, true);
    LAUNCH(ctx, KID_INT2_QUANT, s, k_int2_quant, dim3((C + TILE_C - 1) / TILE_C, (N + Rq - 1) / Rq, batch), dim3(NTHR), 0, s, b, N, C, Rq, flags);
    return check_launch(ctx, "int2 quantise launch");
}
}  // extern "C"

// One fused launch of the software-pipelined replay (cfx_plan_run_pipelined, cfx_plan.hip): [dequant(dq) | finalize(fin) | stats(st)]
int cfx_i_launch_pipe(cfx_plan* p, hipStream_t s, int N, int C, const int* comp_op, const int* deq_op,
                       const PipeUnit* dq, const PipeUnit* fin, const PipeUnit* st, int fin_parity, int st_parity,
                       hipEvent_t done_ev) {
    cfx_ctx* ctx = p->ctx;
    BatchDX bd; BatchC bf, bs;
    PipeArgs a;
    memset(&a, 0, sizeof(a));
    a.N = N; a.C = C;
    a.CB = (C + TILE_C - 1) / TILE_C;
    a.ws_stride = ws_words(CFX_CODEC_BINARY, N, C);
    u64* ws0 = (u64*)p->pipe_ws;
    const size_t half = (size_t)CFX_MAX_BATCH * a.ws_stride;
    auto rows_of = [&](const PipeUnit* u) { return auto_rows(ctx, N, C, u->n_comp_items, true); };
    if (fin) {
        int n = 0;
        for (int l = 0; l < fin->n_layers; ++l) {
            const PlanOp* o = &p->ops[comp_op[fin->first_layer + l]];
            for (int i = 0; i < o->batch; ++i) bf.it[n++] = o->c[i];
        }
        const int R = rows_of(fin);
        a.fin_P = (N + R - 1) / R;
        a.fin_bpi = 1 + (C + NTHR / 4 - 1) / (NTHR / 4);
        a.n_fin = a.fin_bpi * n;
        a.ws_fin = ws0 + (size_t)fin_parity * half;
    }
    if (st) {
        int n = 0;
        for (int l = 0; l < st->n_layers; ++l) {
            const PlanOp* o = &p->ops[comp_op[st->first_layer + l]];
            for (int i = 0; i < o->batch; ++i) bs.it[n++] = o->c[i];
        }
        a.st_R = rows_of(st);
        a.st_P = (N + a.st_R - 1) / a.st_R;
        a.n_st = a.CB * a.st_P * n;
        a.ws_st = ws0 + (size_t)st_parity * half;
    }
    int n_dq = 0;
    if (dq) {
        int n = 0;
        for (int l = 0; l < dq->n_layers; ++l) {
            const PlanOp* o = &p->ops[deq_op[dq->first_layer + l]];
            for (int i = 0; i < o->batch; ++i) bd.it[n++] = o->d[i];
        }
        a.dq_R = auto_rows(ctx, N, C, n, false);
        a.dq_rb = (N + a.dq_R - 1) / a.dq_R;
        n_dq = a.CB * a.dq_rb * n;
    }
    // profiled as "the" pipeline kernel only when all three groups carry equally sized units (steady state)
    const bool steady = dq && fin && st && dq->n_layers == fin->n_layers && fin->n_layers == st->n_layers;
    if (steady) LAUNCH_DONE(ctx, KID_BINARY_PIPE, s, done_ev, k_binary_pipe<true>, dim3(a.n_fin + a.n_st + n_dq), dim3(NTHR), 0, s, bd, bf, bs, a);
    else LAUNCH_DONE(ctx, KID_BINARY_PIPE_EDGE, s, done_ev, k_binary_pipe<false>, dim3(a.n_fin + a.n_st + n_dq), dim3(NTHR), 0, s, bd, bf, bs, a);
    return check_launch(ctx, "pipelined launch");
}

extern "C" {

size_t cfx_binary_rank_packet_bytes(int N, int C, int rank) {
    if (N <= 0 || C <= 0 || (C % 8) || rank < 1 || rank > 32 || (((size_t)N * (C / 8)) % 2)) return 0;
    return (size_t)N * C / 8 + 2 * ((size_t)N + C) * rank;
}

size_t cfx_binary_rank_workspace_bytes(int N, int C, int rank, int batch) {
    if (!cfx_binary_rank_packet_bytes(N, C, rank) || batch < 1 || batch > CFX_MAX_BATCH) return 0;
    return cfx_i_lr_workspace_bytes_any(N, C, rank, batch);
}

int cfx_binary_rank_compress_batch(cfx_ctx* ctx, int N, int C, int rank, int flags, int batch, const cfx_comp_item* items,
                                   const void* const* init_q, void* workspace, size_t workspace_bytes, void* stream) {
    if (!ctx || !items || !init_q) return fail(ctx, CFX_ERR_NULL, "binary rank-K compress: null ctx/items/init_q");
    if (batch < 1 || batch > CFX_MAX_BATCH) return fail(ctx, CFX_ERR_BATCH, "binary rank-K compress: batch out of range");
    if (!cfx_binary_rank_packet_bytes(N, C, rank)) return fail(ctx, CFX_ERR_SHAPE, "binary rank-K compress: bad shape / rank (1 .. 32)");
    const size_t need = cfx_binary_rank_workspace_bytes(N, C, rank, batch);
    if (!workspace || workspace_bytes < need) return fail(ctx, CFX_ERR_WORKSPACE, "binary rank-K compress: workspace too small");
    const bool upd = flags & CFX_FLAG_UPDATE_CACHE;
    BatchC b;
    memset(&b, 0, sizeof(b));
    for (int i = 0; i < batch; ++i) {
        if (!items[i].x || !items[i].packet) return fail(ctx, CFX_ERR_NULL, "binary rank-K compress: null x/packet");
        if (upd && !items[i].new_base) return fail(ctx, CFX_ERR_NULL, "binary rank-K compress: UPDATE_CACHE needs new_base");
        if (!AL16(items[i].x) || !AL16(items[i].base) || !AL16(items[i].new_base) || !AL16(items[i].packet))
            return fail(ctx, CFX_ERR_ALIGN, "binary rank-K compress: pointers must be 16-byte aligned");
        b.it[i] = items[i];
    }
    // 1. rank-K factors of |x - base| (the low-rank chain, stopped at the fp16 factors in the workspace)
    const int rf = cfx_lr_compress_batch(ctx, 1, N, C, rank, CFX_I_FLAG_LR_FACTORS_ONLY | CFX_I_FLAG_LR_ABS, batch, items, init_q, workspace,
                                         workspace_bytes, stream);
    if (rf != CFX_OK) return rf;
    size_t offU = 0, offV = 0, per1 = 0;
    cfx_i_lr_factor_offsets(N, C, rank, &offU, &offV, &per1);
    const size_t per = need / batch;
    RankFac fac;
    memset(&fac, 0, sizeof(fac));
    for (int i = 0; i < batch; ++i) {
        fac.U[i] = (const h16*)((char*)workspace + per * i + offU);
        fac.VT[i] = (const h16*)((char*)workspace + per * i + offV);
    }
    // 2. sign bits, factors into the packet, error-feedback state
    hipStream_t s = (hipStream_t)stream;
    const int R = WAVES * UNROLL;
    BatchD dummy;
    memset(&dummy, 0, sizeof(dummy));
    LAUNCH(ctx, KID_BINARY_EF, s, k_binary_rank<true>, dim3((C + TILE_C - 1) / TILE_C, (N + R - 1) / R, batch), dim3(NTHR), 0, s, b, dummy, fac, N, C, R, rank, flags);
    return check_launch(ctx, "binary rank-K compress launch");
}

int cfx_binary_rank_decompress_batch(cfx_ctx* ctx, int N, int C, int rank, int batch, const cfx_decomp_item* items, void* stream) {
    if (!ctx || !items) return fail(ctx, CFX_ERR_NULL, "binary rank-K decompress: null ctx/items");
    if (batch < 1 || batch > CFX_MAX_BATCH) return fail(ctx, CFX_ERR_BATCH, "binary rank-K decompress: batch out of range");
    if (!cfx_binary_rank_packet_bytes(N, C, rank)) return fail(ctx, CFX_ERR_SHAPE, "binary rank-K decompress: bad shape / rank (1 .. 32)");
    BatchD b;
    memset(&b, 0, sizeof(b));
    for (int i = 0; i < batch; ++i) {
        if (!items[i].packet || !items[i].recon) return fail(ctx, CFX_ERR_NULL, "binary rank-K decompress: null packet/recon");
        if (!AL16(items[i].packet) || !AL16(items[i].recon) || !AL16(items[i].base)) return fail(ctx, CFX_ERR_ALIGN, "binary rank-K decompress: pointers must be 16-byte aligned");
        b.it[i] = items[i];
    }
    hipStream_t s = (hipStream_t)stream;
    const int R = WAVES * UNROLL;
    BatchC dummy;
    RankFac fac;
    memset(&dummy, 0, sizeof(dummy));
    memset(&fac, 0, sizeof(fac));
    LAUNCH(ctx, KID_BINARY_DEQUANT, s, k_binary_rank<false>, dim3((C + TILE_C - 1) / TILE_C, (N + R - 1) / R, batch), dim3(NTHR), 0, s, dummy, b, fac, N, C, R, rank, 0);
    return check_launch(ctx, "binary rank-K decompress launch");
}

}  // extern "C"

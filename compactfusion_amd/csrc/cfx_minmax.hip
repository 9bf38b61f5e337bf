// libcfx.so - the min/max codec family: residual int8 and int4 (compress_quantize.py:428-484, :522-640): stand-alone kernels, the one-launch
// compress, the layer launch (k_minmax_layer).  Shared device code: cfx_device.h; the C-ABI and the dispatch: cfx_api.hip.
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

// ---------------------------------------------------------------------------------------------------
// per-channel min/max statistics pass (int4 / int8)    compress_quantize.py:452-453, :552-553
//   part[p][c] = {min, max} of (x-base) over the tile's rows (fp16 compares are exact)
// ---------------------------------------------------------------------------------------------------
template <bool WT>
__device__ __forceinline__ void minmax_stats_body(const cfx_comp_item& it, int N, int C, int R, int bx, int by, unsigned* part) {
    const TileCoord t = tile_coord_at(bx, by, N, C, R);                 // part: [P][C] of {min16 | max16<<16}
    const h16* x = (const h16*)it.x;
    const h16* base = (const h16*)it.base;
    h16x8 mn = (h16x8)(h16)65504.0f, mx = (h16x8)(h16)-65504.0f;
    mn = (h16x8)hfrom(0x7c00);   // +inf
    mx = (h16x8)hfrom(0xfc00);   // -inf
    for (int r = t.r0 + t.w; r < t.r1; r += WAVES * UNROLL_S) {
        h16x8 xv[UNROLL_S], bv[UNROLL_S];
#pragma unroll
        for (int j = 0; j < UNROLL_S; ++j) {
            const int rr = r + WAVES * j;
            xv[j] = (h16x8)(h16)0; bv[j] = (h16x8)(h16)0;
            if (rr < t.r1 && t.act) {
                xv[j] = ld8(x + (size_t)rr * C + t.c);
                if (base) bv[j] = ld8(base + (size_t)rr * C + t.c);
            }
        }
#pragma unroll
        for (int j = 0; j < UNROLL_S; ++j) {
            const int rr = r + WAVES * j;
            if (rr < t.r1 && t.act) {
                const h16x8 d = xv[j] - bv[j];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    mn[i] = d[i] < mn[i] ? d[i] : mn[i];
                    mx[i] = d[i] > mx[i] ? d[i] : mx[i];
                }
            }
        }
    }
    __shared__ unsigned sm[WAVES][TILE_C];
#pragma unroll
    // [i][lane ^ 8i]: conflict-free here and in the column-order read below (see absmean_stats_body)
    for (int i = 0; i < 8; ++i) sm[t.w][i * 64 + (t.lane ^ (i << 3))] = (unsigned)hbits(mn[i]) | ((unsigned)hbits(mx[i]) << 16);
    __syncthreads();
    for (int k = threadIdx.x; k < TILE_C; k += NTHR) {
        const int s = (k & 7) * 64 + ((k >> 3) ^ ((k & 7) << 3));
        const int cc = bx * TILE_C + k;
        if (cc < C) {
            h16 a = hfrom((u16)(sm[0][s] & 0xffff)), b = hfrom((u16)(sm[0][s] >> 16));
#pragma unroll
            for (int w = 1; w < WAVES; ++w) {
                const h16 a2 = hfrom((u16)(sm[w][s] & 0xffff)), b2 = hfrom((u16)(sm[w][s] >> 16));
                a = a2 < a ? a2 : a;
                b = b2 > b ? b2 : b;
            }
            const unsigned v = (unsigned)hbits(a) | ((unsigned)hbits(b) << 16);
            if (WT) st_wt(&part[(size_t)by * C + cc], v); else part[(size_t)by * C + cc] = v;
        }
    }
}

__global__ __launch_bounds__(NTHR) void k_minmax_stats(BatchC batch, int N, int C, int R, u64* ws, size_t ws_stride) {
    minmax_stats_body<false>(batch.it[blockIdx.z], N, C, R, blockIdx.x, blockIdx.y, (unsigned*)(ws + (size_t)blockIdx.z * ws_stride));
}

__device__ __forceinline__ h16 hdiv(h16 a, h16 b) { return (h16)((float)a / (float)b); }   // correctly rounded fp16 quotient
__device__ __forceinline__ h16 hrint(h16 a) { return __builtin_rintf16(a); }                // round half to even (torch.round)
// hdiv(a, b) given bf = (float)b and rb = v_rcp_f32(bf) (1 ulp), 4 instructions instead of the 10 of an IEEE fp32 division: one Newton step
// (t = a * rb; q = t + (a - t * b) * rb) leaves the fp32 quotient within 0.5 ulp for every pair of fp16 operands whatever the rcp's last bit
// (exhaustive over the significands: tests/test_fastdiv.py), so its rounding to fp16 is the correctly rounded quotient; v_div_fixup_f32 puts
// IEEE's results for zero / infinite / NaN operands back.  The codecs divide every element by its channel's scale: rb is per channel.
__device__ __forceinline__ h16 hdiv_r(h16 a, float bf, float rb) {
    const float af = (float)a;
    const float t = af * rb;
    const float r = __builtin_fmaf(-t, bf, af);
    const float q = __builtin_fmaf(r, rb, t);
    return (h16)__builtin_amdgcn_div_fixupf(q, bf, af);
}
__device__ __forceinline__ bool hisnan(h16 a) { return a != a; }

// int4 : scale = fp16(fp16(max-min)/15.000001f), min                              compress_quantize.py:556-558
// int8 : scale = fp16(fp16(max-min)/255.0f), zp = clamp(-128 - round(min/scale)) -> int16          :455-463
__device__ __forceinline__ void minmax_write_scales(const cfx_comp_item& it, int N, int C, int codec, int c, h16 mn, h16 mx) {
    const h16 rng = mx - mn;
    if (codec == CFX_CODEC_INT4) {
        h16* S = (h16*)((char*)it.packet + (size_t)(N / 2) * C);
        S[c] = (h16)((float)rng / 15.000001f);
        S[C + c] = mn;
    } else {
        h16* S = (h16*)((char*)it.packet + (size_t)N * C);
        short* Z = (short*)(S + C);
        const h16 scale = (h16)((float)rng / 255.000001f);
        const h16 r = hrint(hdiv(mn, scale));
        h16 z = (h16)-128.0f - r;
        short zi;
        if (hisnan(z)) zi = 0;
        else {
            z = z < (h16)-128.0f ? (h16)-128.0f : z;
            z = z > (h16)127.0f ? (h16)127.0f : z;
            zi = (short)(float)z;
        }
        S[c] = scale;
        Z[c] = zi;
    }
}

// finalize int4 : scale = fp16(fp16(max-min)/15.000001f), min                     compress_quantize.py:556-558
//          int8 : scale = fp16(fp16(max-min)/255.0f), zp = clamp(-128 - round(min/scale)) -> int16   :455-463
__global__ __launch_bounds__(1024) void k_minmax_finalize(BatchC batch, int N, int C, int P, int codec, const u64* ws, size_t ws_stride) {
    const cfx_comp_item it = batch.it[blockIdx.y];
    const unsigned* part = (const unsigned*)(ws + (size_t)blockIdx.y * ws_stride);
    // 256 channels per block; 4 threads per channel split the P partials so their loads are in flight together
    __shared__ unsigned red[4][256];
    const int cl = threadIdx.x & 255, q = threadIdx.x >> 8;
    const int c = blockIdx.x * 256 + cl;
    h16 mn = hfrom(0x7c00), mx = hfrom(0xfc00);
    if (c < C) {
#pragma unroll 4
        for (int p = q; p < P; p += 4) {
            const unsigned v = part[(size_t)p * C + c];
            const h16 a = hfrom((u16)(v & 0xffff)), b = hfrom((u16)(v >> 16));
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
    }
    red[q][cl] = (unsigned)hbits(mn) | ((unsigned)hbits(mx) << 16);
    __syncthreads();
    if (q != 0 || c >= C) return;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const h16 a = hfrom((u16)(red[k][cl] & 0xffff)), b = hfrom((u16)(red[k][cl] >> 16));
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    minmax_write_scales(it, N, C, codec, c, mn, mx);
}

// Compress statistics + in-launch finalize for the per-channel min/max codecs (same ticket scheme as k_absmean_compress;
// only the column-block tickets exist here: there is no row statistic).
__global__ __launch_bounds__(NTHR) void k_minmax_compress(BatchC batch, int N, int C, int R, int CB, int P, int codec, u64* ws, size_t ws_stride,
                                                          unsigned* tick0) {
    const int per = CB * P;
    const int z = blockIdx.x / per, rem = blockIdx.x - z * per;
    const int by = rem / CB, bx = rem - by * CB;
    const cfx_comp_item& it = batch.it[z];
    unsigned* part = (unsigned*)(ws + (size_t)z * ws_stride);
    unsigned* tick = tick0 + z * TICK_WORDS;
    minmax_stats_body<true>(it, N, C, R, bx, by, part);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned flag;
    if (threadIdx.x == 0) flag = __hip_atomic_fetch_add(tick + 1 + bx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (flag != (unsigned)(P - 1)) return;
    // last arriver of column block bx: two columns per thread, FUSED_CH partials of both in flight per batch (unconditional
    // loads with clamped indices: see fused_rows_finalize)
    const int c0 = bx * TILE_C + threadIdx.x, c1 = c0 + NTHR;
    const int cc0 = min(c0, C - 1), cc1 = min(c1, C - 1);
    h16 mn0 = hfrom(0x7c00), mx0 = hfrom(0xfc00), mn1 = mn0, mx1 = mx0;
    for (int p0 = 0; p0 < P; p0 += FUSED_CH) {
        unsigned v0[FUSED_CH], v1[FUSED_CH];
#pragma unroll
        for (int j = 0; j < FUSED_CH; ++j) {
            const size_t row = (size_t)min(p0 + j, P - 1) * C;      // a repeated partial does not change a min / max
            v0[j] = ld_wt(&part[row + cc0]);
            v1[j] = ld_wt(&part[row + cc1]);
        }
#pragma unroll
        for (int j = 0; j < FUSED_CH; ++j) {
            const h16 a0 = hfrom((u16)(v0[j] & 0xffff)), b0 = hfrom((u16)(v0[j] >> 16));
            const h16 a1 = hfrom((u16)(v1[j] & 0xffff)), b1 = hfrom((u16)(v1[j] >> 16));
            mn0 = a0 < mn0 ? a0 : mn0; mx0 = b0 > mx0 ? b0 : mx0;
            mn1 = a1 < mn1 ? a1 : mn1; mx1 = b1 > mx1 ? b1 : mx1;
        }
    }
    if (c0 < C) minmax_write_scales(it, N, C, codec, c0, mn0, mx0);
    if (c1 < C) minmax_write_scales(it, N, C, codec, c1, mn1, mx1);
    if (threadIdx.x == 0) st_wt(tick + 1 + bx, 0u);
}

// ---------------------------------------------------------------------------------------------------
// The min/max codecs' layer in ONE launch (cfx_compress_batch_gated / the exchange-layer ops, codecs INT4 and INT8) - what the 1-bit and
// 2-bit codecs have had: the statistics tile stays in REGISTERS, the scales are finalised inside the launch, every statistics workgroup
// then quantises its own tile from those registers (x and the state are read ONCE: 6.5 / 7.0 B per element is what moves), and the
// reconstruction of the peers' tensors waits in the same launch, state tiles preloaded, for the packets.
//   S  tile (32 rows x 512 channels, 8 waves): load x, state -> d = x - state -> per-channel {min, max} partial of the tile, published
//      write-through -> ticket of the column block; the block's last arriver reduces the P partials, writes scale / min (int4) or scale /
//      zero point (int8) into the packet (compress_quantize.py:452-463, :552-558) and raises the block's COLUMN GATE.  The scales of a
//      tile depend on its column block only (there is no tensor-wide statistic), so a tile waits for the P tiles of its own block, not
//      for the launch.  Then: codes from registers (arithmetic of k_int4_quant / k_int8_quant), published as 16-byte write-through
//      stores through an LDS transpose, one arrival on the codes gate, error-feedback state last (nobody waits for it).
//   D  tile (112 rows x 512 channels): state rows into registers, wait for the gate (the launch's own codes gate, or the external word an
//      exchange stream sets once the peers' packets have arrived), codes + scales through L2-bypassing loads, finish from registers.
// S workgroups precede D in dispatch order and wait only for each other: all of them must be CO-RESIDENT (the host checks; otherwise the
// multi-launch forms run).  Column gates hold a per-stream launch sequence number (monotonic, raised with atomic max: never reset).
// ---------------------------------------------------------------------------------------------------
#define MML_NW FUSED_NW
#define MML_KC 14              // rows of a D tile a wave holds in registers (int4: 7 row pairs)
#define MML_MAX_P 64           // row tiles per column block (one poll load per lane of a wave)
#define MML_MAX_P_TALL 128     // ... of the tall form (two poll loads per lane)
#define MML_NRED 8             // tall form: tiles 0 .. 7 of a column block reduce 64 of its 512 channels each
struct MinMaxLayerArgs {
    int N, C, CB, P, R, n_st;         // group S: CB x P tiles of R rows (32 or 64) per own tensor
    int g_R, g_rb, n_g;               // group D: tiles of g_R rows, g_rb per tensor
    int codec, flags;
    u64* part; size_t part_stride;    // context-owned arena (zeroed once), per own tensor [P][C] partials + [C] scales as TAGGED words:
                                      // {fp16 pair, seq} in one 8-byte store - a reader polls the data itself, no flag, no store fence
    unsigned* codedone; unsigned seq; // one flag word per S tile, index (z * CB + bx) * P + by: "codes (and, the tiles that computed them,
                                      // the scales) published" = the launch's sequence number (context-wide, never reused)
    unsigned* xgate; unsigned xexpect;     // external gate for group D (NULL: a D tile waits for the S tiles whose codes it reads)
    unsigned* err;
    long long timeout;                // in-launch waits: ticks of the 100 MHz wall clock
    int remote;
    signed char src[CFX_MAX_BATCH];   // gated item -> the own tensor whose packet it reads (loop-back forms)
    P2PInline p2p;                    // own != NULL: workgroup 0 runs the peer-to-peer exchange and opens xgate itself
    // tall != 0 (tensors whose S tiles do not fit the chip at once): S tiles ordered column block by column block (row tile fastest), so that a
    // column block's P tiles - the only workgroups a tile waits for - are dispatched together and ahead of every later block's; tiles
    // 0 .. MML_NRED - 1 of the block reduce 64 channels' P partials each (a wave takes every 8th partial) and publish the scales as tagged
    // words; every tile polls the 512 scales of its block
    int tall;
    int coop;                         // the reduce by tiles 0 .. MML_NRED - 1 (always in the tall form; otherwise wherever a channel has more than
                                      // 32 partials: every tile reducing all of them itself would take P / 8 dependent rounds of loads)
    Probe probe;                      // developer build: 16 words per workgroup (100 MHz wall clock per phase; word 7: 1 = S tile, 4 = D tile)
};
#define MML_STAMP(i) st.at(i)
// received values of 8 channels of row h of a code row (int8: h = 0): k_int8_dequant / k_int4_dequant arithmetic
template <bool INT4>
__device__ __forceinline__ h16x8 minmax_recv(u64 codes, int h, h16x8 sc, h16x8 mz) {
    h16x8 qh;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        qh[i] = INT4 ? (h16)(float)((codes >> (8 * i + 4 * h)) & 15u) : (h16)(float)(int)(signed char)(codes >> (8 * i));
    return INT4 ? (qh * sc + mz) : ((qh - mz) * sc);
}
__device__ __forceinline__ u64 ld_wt_or_sys(const u64* p, bool remote) { return remote ? ld_sys(p) : ld_wt(p); }
__device__ __forceinline__ h16x8 ld8_pub(const u16* p, bool remote) {
    if (!remote) return ld8_wt(p);
    u16x8 vb;
#pragma unroll
    for (int i = 0; i < 8; ++i) vb[i] = ld_sys(p + i);
    return __builtin_bit_cast(h16x8, vb);
}
// scale vectors of 8 channels out of a packet other workgroups (or another GPU) published; int8: zp as fp16 values
template <bool INT4>
__device__ __forceinline__ void minmax_ld_scales(const unsigned char* pk, int N, int C, int cc, bool remote, h16x8& sc, h16x8& mz) {
    if (INT4) {
        const u16* S = (const u16*)(pk + (size_t)(N / 2) * C);
        sc = ld8_pub(S + cc, remote);
        mz = ld8_pub(S + C + cc, remote);
    } else {
        const u16* S = (const u16*)(pk + (size_t)N * C);
        sc = ld8_pub(S + cc, remote);
        const u16x8 zb = __builtin_bit_cast(u16x8, ld8_pub(S + C + cc, remote));
#pragma unroll
        for (int i = 0; i < 8; ++i) mz[i] = (h16)(float)(short)zb[i];
    }
}
// scale and min (int4) / scale and zero point (int8) of one channel from its {min, max}: compress_quantize.py:556-558 / :455-463
template <bool INT4>
__device__ __forceinline__ void minmax_scale_of(h16 mn, h16 mx, h16& scale, u16& second) {
    const h16 rng = mx - mn;
    if (INT4) {
        scale = (h16)((float)rng / 15.000001f);
        second = hbits(mn);
    } else {
        scale = (h16)((float)rng / 255.000001f);
        const h16 r = hrint(hdiv(mn, scale));
        h16 z = (h16)-128.0f - r;
        short zi;
        if (hisnan(z)) zi = 0;
        else {
            z = z < (h16)-128.0f ? (h16)-128.0f : z;
            z = z > (h16)127.0f ? (h16)127.0f : z;
            zi = (short)(float)z;
        }
        second = (u16)zi;
    }
}
#ifndef MML_POLL_SLEEP
#define MML_POLL_SLEEP 2
#endif
// RW = rows a wave holds: 4 (tiles of 32 rows) or 8 (tiles of 64 rows: tall tensors, fewer partials per channel)
// Registers: the tile is held as d = x - state (RW rows) plus the state rows the error-feedback pass adds the received values to; with
// RW = 8 the upper MML_PARK rows of the state wait in LDS (`park`, 16 bytes per thread and row, written and read by the same thread) - the
// statistics and the codes only need d, and 64 rows of x AND state beside the reduction's words in flight did not fit 128 registers
// (the compiler spilled 12 / 80 bytes a lane to scratch: tools/resource_usage.py).  x itself is dead once d exists; without error
// feedback (the state becomes x) the last pass reads the tile of x again.
#define MML_PARK 4
template <bool INT4, int RW>
__device__ __forceinline__ void minmax_layer_s_tile(const cfx_comp_item& it, const MinMaxLayerArgs& a, int z, int bx, int by, u64 (*sm)[TILE_C],
                                                    u32x4* park) {
    constexpr int NW = MML_NW;
    constexpr int RPC = INT4 ? 2 : 1;          // rows per code row
    constexpr int CR = RW / RPC;               // code rows a wave holds
    const int N = a.N, C = a.C;
    const TileCoord t = tile_coord_at(bx, by, N, C, a.R);
    const h16* x = (const h16*)it.x;
    const h16* base = (const h16*)it.base;
    const int cc = min(t.c, C - 8);
    const Probe st = a.probe.of(blockIdx.x);
    st.set(7, 1);
    MML_STAMP(0);
    // every wait of this tile gives up a.timeout after the tile started (one time base, no cascade of waits); a tile that gave up stores
    // neither codes nor state nor its flag - whoever waits for it gives up in turn, and the context's error word says so
    SpinClock clk;
    clk.t0 = wall_clock64();
    bool failed = false;
    // ---- the tile into registers (every load unconditional: clamped row, masked use) ----
    constexpr int RREG = RW > 4 ? RW - MML_PARK : RW;       // state rows that stay in registers
    h16x8 dk[RW], bk[RREG];
    bool rv[RW];
    {
        h16x8 xk[RW], bt[RW > RREG ? RW - RREG : 1];
#pragma unroll
        for (int j = 0; j < CR; ++j)
#pragma unroll
            for (int h = 0; h < RPC; ++h) {
                const int q = j * RPC + h;
                const int row = t.r0 + (t.w + NW * j) * RPC + h;
                rv[q] = row < t.r1 && t.act;
                const size_t off = (size_t)min(row, N - 1) * C + cc;
                xk[q] = ld8nt(x + off);
                const h16x8 b = base ? ld8nt(base + off) : (h16x8)(h16)0;
                if (q < RREG) bk[q < RREG ? q : 0] = b; else bt[q >= RREG ? q - RREG : 0] = b;
            }
#pragma unroll
        for (int q = 0; q < RW; ++q) {
            if (q < RREG) dk[q] = xk[q] - bk[q < RREG ? q : 0];
            else {
                dk[q] = xk[q] - bt[q >= RREG ? q - RREG : 0];
                park[(q - RREG) * (NW * 64) + threadIdx.x] = __builtin_bit_cast(u32x4, bt[q >= RREG ? q - RREG : 0]);
            }
        }
    }
    h16x8 mn = (h16x8)hfrom(0x7c00), mx = (h16x8)hfrom(0xfc00);
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        const h16x8 d = dk[q];
        if (rv[q]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                mn[i] = d[i] < mn[i] ? d[i] : mn[i];
                mx[i] = d[i] > mx[i] ? d[i] : mx[i];
            }
        }
    }
    unsigned* sm32 = (unsigned*)&sm[0][0];                 // [NW][TILE_C] words
#pragma unroll
    for (int i = 0; i < 8; ++i) sm32[t.w * TILE_C + i * 64 + (t.lane ^ (i << 3))] = (unsigned)hbits(mn[i]) | ((unsigned)hbits(mx[i]) << 16);
    lds_barrier();
    u64* part = a.part + (size_t)z * a.part_stride;         // [P][C] tagged partials, then [C] tagged scales
    u64* sca = part + (size_t)a.P * C;
    const u64 tag = (u64)a.seq << 32;
    const int k = threadIdx.x;                              // 512 threads: one channel of the tile each
    const int ch = bx * TILE_C + k, chc = min(ch, C - 1);
    {
        const int sidx = (k & 7) * 64 + ((k >> 3) ^ ((k & 7) << 3));
        h16 lo = hfrom((u16)(sm32[sidx] & 0xffff)), hi = hfrom((u16)(sm32[sidx] >> 16));
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const unsigned v = sm32[w * TILE_C + sidx];
            const h16 a2 = hfrom((u16)(v & 0xffff)), b2 = hfrom((u16)(v >> 16));
            lo = a2 < lo ? a2 : lo;
            hi = b2 > hi ? b2 : hi;
        }
        // the partial AND its "published" mark in one 8-byte store: nobody waits for a store to be acknowledged before a flag can follow
        if (ch < C) st_wt(&part[(size_t)by * C + ch], tag | (unsigned)hbits(lo) | ((unsigned)hbits(hi) << 16));
    }
    MML_STAMP(1);                                           // tile loaded, partial issued
    const size_t fbase = ((size_t)z * a.CB + bx) * a.P;
    unsigned char* pk = (unsigned char*)it.packet;
    u16* S = (u16*)(pk + (INT4 ? (size_t)(N / 2) * C : (size_t)N * C));
    h16 scale;
    u16 second;
    constexpr int NB = RW == 4 ? 16 : 8;                    // partials in flight per thread (registers: the tile stays live)
    if (!a.coop || by < MML_NRED) {
        // one wave watches ONE word per tile of the block (lane i: tile i's first channel) until all carry the tag; only then does every
        // thread load its channel's P words (and checks their tags: a tile's 512 stores are not ordered among themselves).  Every thread
        // polling its own words from the start is a hop shorter on an idle chip - and a storm of 8192 loads per tile and round that starves
        // whatever shares the chip, including the tiles being waited for (measured: waits of seconds beside a copy stream)
        if (t.w == 0) {
            const u64* w0 = part + (size_t)bx * TILE_C;
            for (;;) {
                const u64 v0 = t.lane < a.P ? ld_wt(w0 + (size_t)t.lane * C) : tag;
                const u64 v1 = t.lane + 64 < a.P ? ld_wt(w0 + (size_t)(t.lane + 64) * C) : tag;
                if (__builtin_amdgcn_ballot_w64((unsigned)(v0 >> 32) != a.seq || (unsigned)(v1 >> 32) != a.seq) == 0) break;
                __builtin_amdgcn_s_sleep(MML_POLL_SLEEP);
                if (clk.expired(a.timeout)) { failed = true; break; }
            }
        }
        __syncthreads();
    }
    if (!a.coop) {
        // ---- every tile of the column block reduces the block's P partials itself: no last arriver, no second hand-over ----
        h16 lo = hfrom(0x7c00), hi = hfrom(0xfc00);
        for (int p0 = 0; p0 < a.P; p0 += NB) {
            u64 v[NB];
            for (;;) {
                // (all loads issued, THEN the tags compared: a test per load makes the compiler wait for each load in turn)
#pragma unroll
                for (int j = 0; j < NB; ++j) v[j] = ld_wt(&part[(size_t)min(p0 + j, a.P - 1) * C + chc]);   // a repeated partial does not change a min / max
                unsigned bad = 0;
#pragma unroll
                for (int j = 0; j < NB; ++j) bad |= (unsigned)(v[j] >> 32) ^ a.seq;
                if (!bad) break;
                __builtin_amdgcn_s_sleep(MML_POLL_SLEEP);
                if (failed || clk.expired(a.timeout)) { failed = true; break; }
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const h16 a0 = hfrom((u16)(v[j] & 0xffff)), b0 = hfrom((u16)((unsigned)v[j] >> 16));
                lo = a0 < lo ? a0 : lo;
                hi = b0 > hi ? b0 : hi;
            }
        }
        minmax_scale_of<INT4>(lo, hi, scale, second);
        if (by == 0 && ch < C && !failed) {                    // the block's scales into the packet: once
            st_wt(S + ch, hbits(scale));
            st_wt(S + C + ch, second);
        }
    } else {
        // ---- tall form: P x 2 KB per tile would be a second pass over a good part of the tensor - tiles 0 .. 7 reduce 64 channels each ----
        if (by < MML_NRED) {
            const int chr = bx * TILE_C + by * 64 + t.lane, chrc = min(chr, C - 1);
            h16 lo = hfrom(0x7c00), hi = hfrom(0xfc00);
            for (int p0 = t.w; p0 < a.P; p0 += NW * 8) {
                u64 v[8];
                for (;;) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = ld_wt(&part[(size_t)min(p0 + NW * j, a.P - 1) * C + chrc]);
                    unsigned bad = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) bad |= (unsigned)(v[j] >> 32) ^ a.seq;
                    if (!bad) break;
                    __builtin_amdgcn_s_sleep(MML_POLL_SLEEP);
                    if (failed || clk.expired(a.timeout)) { failed = true; break; }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const h16 a0 = hfrom((u16)(v[j] & 0xffff)), b0 = hfrom((u16)((unsigned)v[j] >> 16));
                    lo = a0 < lo ? a0 : lo;
                    hi = b0 > hi ? b0 : hi;
                }
            }
            failed = __syncthreads_or(failed ? 1 : 0) != 0;     // (sm32: the publish above has read it) - a wave that gave up: no scales from this tile
            sm32[t.w * 64 + t.lane] = (unsigned)hbits(lo) | ((unsigned)hbits(hi) << 16);
            __syncthreads();
            if (t.w == 0) {
#pragma unroll
                for (int w = 1; w < NW; ++w) {
                    const unsigned u = sm32[w * 64 + t.lane];
                    const h16 a0 = hfrom((u16)(u & 0xffff)), b0 = hfrom((u16)(u >> 16));
                    lo = a0 < lo ? a0 : lo;
                    hi = b0 > hi ? b0 : hi;
                }
                h16 sc1;
                u16 sec1;
                minmax_scale_of<INT4>(lo, hi, sc1, sec1);
                if (chr < C && !failed) {
                    st_wt(&sca[chr], tag | (unsigned)hbits(sc1) | ((unsigned)sec1 << 16));
                    st_wt(S + chr, hbits(sc1));                 // (the packet's copy: for the receivers, behind this tile's codes flag)
                    st_wt(S + C + chr, sec1);
                }
            }
        }
        if (t.w == 0) {                                         // (one wave watches one word per reducer tile first: see above)
            for (;;) {
                const u64 v0 = t.lane < MML_NRED ? ld_wt(&sca[min(bx * TILE_C + t.lane * 64, C - 1)]) : tag;
                if (__builtin_amdgcn_ballot_w64((unsigned)(v0 >> 32) != a.seq) == 0) break;
                __builtin_amdgcn_s_sleep(MML_POLL_SLEEP);
                if (failed || clk.expired(a.timeout)) { failed = true; break; }
            }
        }
        __syncthreads();
        u64 v;
        for (;;) {
            v = ld_wt(&sca[chc]);
            if ((unsigned)(v >> 32) == a.seq) break;
            __builtin_amdgcn_s_sleep(16);
            if (failed || clk.expired(a.timeout)) { failed = true; break; }
        }
        scale = hfrom((u16)(v & 0xffff));
        second = (u16)((unsigned)v >> 16);
    }
    MML_STAMP(2);                                           // scales known
    // a lane's 8 channels from the 512 per-thread values: through LDS
    u16* sl = (u16*)&sm[0][0];                              // [2][TILE_C] halves (the min / max words are consumed)
    if (__syncthreads_or(failed ? 1 : 0)) {                 // somebody's wait gave up: no codes, no state, no flag from this tile
        if (k == 0) gate_fail(a.err);
        return;
    }
    sl[k] = hbits(scale);
    sl[TILE_C + k] = second;
    __syncthreads();
    h16x8 sc, mz;
    {
        const u16x8 s8 = *(const u16x8*)(sl + t.lane * 8), m8 = *(const u16x8*)(sl + TILE_C + t.lane * 8);
        sc = __builtin_bit_cast(h16x8, s8);
        if (INT4) mz = __builtin_bit_cast(h16x8, m8);
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) mz[i] = (h16)(float)(short)m8[i];
        }
    }
    __syncthreads();
    // ---- own tile: codes from registers ----
    u64* stage = &sm[0][0] + (size_t)t.w * CR * 64;         // this wave's CR code rows x 64 lanes x 8 bytes (same wave writes and reads: in order)
    {
        // codes exactly as k_int4_quant / k_int8_quant compute them, channel by channel: the division by the channel's scale as
        // hdiv_r with one reciprocal per channel - this loop is the kernel's instruction count (tall tensors: it ran at the VALU's pace)
        u64 cj[CR];
#pragma unroll
        for (int j = 0; j < CR; ++j) cj[j] = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float bf = (float)sc[i], rb = __builtin_amdgcn_rcpf(bf);
#pragma unroll
            for (int j = 0; j < CR; ++j) {
                if (INT4) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const h16 d = dk[2 * j + h][i];
                        h16 v = hrint(hdiv_r(d - mz[i], bf, rb));
                        v = __builtin_fmaxf16(v, (h16)0);              // (NaN -> 0, as k_int4_quant's explicit test)
                        v = __builtin_fminf16(v, (h16)15.0f);
                        cj[j] |= (u64)((unsigned)(unsigned short)v & 15u) << (8 * i + 4 * h);
                    }
                } else {
                    const h16 d = dk[j][i];
                    h16 v = hrint(hdiv_r(d, bf, rb) + mz[i]);
                    if (hisnan(v)) v = (h16)0;
                    v = __builtin_fmaxf16(v, (h16)-128.0f);
                    v = __builtin_fminf16(v, (h16)127.0f);
                    cj[j] |= (u64)(unsigned char)(signed char)(short)v << (8 * i);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CR; ++j) stage[j * 64 + t.lane] = cj[j];   // (kept there for the error-feedback pass too: nothing else uses the LDS afterwards)
    }
    {
        // a code row of the tile is 512 bytes = 32 lanes x 16 bytes; lanes [0, 32) take the even code rows of the wave, [32, 64) the odd ones
        const int crows = INT4 ? N / 2 : N;
#pragma unroll
        for (int jj = 0; jj < CR; jj += 2) {
            const int j = jj + (t.lane >> 5), seg = t.lane & 31;
            const int cr = (t.r0 / RPC) + t.w + NW * j;
            if (j < CR && cr < crows && cr * RPC < t.r1 && bx * TILE_C + seg * 16 < C)
                st16_wt(pk + (size_t)cr * C + (size_t)bx * TILE_C + seg * 16, *(const u32x4*)((const unsigned char*)(stage + j * 64) + seg * 16));
        }
    }
    MML_STAMP(3);                                           // codes issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (k == 0) st_wt(a.codedone + fbase + by, a.seq);
    MML_STAMP(4);                                           // codes acknowledged, flag issued
    h16* nb = (h16*)it.new_base;
    if ((a.flags & CFX_FLAG_UPDATE_CACHE) && nb) {
        const bool ef = !(a.flags & CFX_FLAG_NO_EF);
#pragma unroll
        for (int j = 0; j < CR; ++j)
#pragma unroll
            for (int h = 0; h < RPC; ++h) {
                const int q = j * RPC + h;
                const int row = t.r0 + (t.w + NW * j) * RPC + h;
                if (rv[q]) {
                    h16x8 o;
                    if (ef) {
                        const h16x8 recv = minmax_recv<INT4>(stage[j * 64 + t.lane], h, sc, mz);
                        const h16x8 b = q < RREG ? bk[q < RREG ? q : 0] : __builtin_bit_cast(h16x8, park[(q - RREG) * (NW * 64) + threadIdx.x]);
                        o = base ? (b + recv) : recv;
                    } else o = ld8nt(x + (size_t)row * C + cc);
                    st8nt(nb + (size_t)row * C + t.c, o);
                }
            }
    }
    if (st.on()) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MML_STAMP(5);                                       // state stores acknowledged
    }
}

template <bool INT4>
__device__ __forceinline__ void minmax_layer_d_tile(const cfx_decomp_item& it, const MinMaxLayerArgs& a, int item, int bx, int by) {
    constexpr int NW = MML_NW;
    constexpr int RPC = INT4 ? 2 : 1;
    constexpr int KC = MML_KC / RPC;           // code rows a wave holds
    const int N = a.N, C = a.C;
    const TileCoord t = tile_coord_at(bx, by, N, C, a.g_R);
    const unsigned char* pk = (const unsigned char*)it.packet;
    const h16* base = (const h16*)it.base;
    h16* out = (h16*)it.recon;
    const int cc = min(t.c, C - 8);
    const Probe st = a.probe.of(blockIdx.x);
    st.set(7, 4);
    MML_STAMP(0);
    const int kc = a.g_R / (NW * RPC);         // code rows per wave of THIS launch's tiles (<= KC; uniform)
    bool failed = false;
    h16x8 bv[MML_KC];
#pragma unroll
    for (int j = 0; j < KC; ++j)
        if (j < kc) {
#pragma unroll
            for (int h = 0; h < RPC; ++h) {
                const int row = t.r0 + (t.w + NW * j) * RPC + h;
                bv[j * RPC + h] = base ? ld8nt(base + (size_t)min(row, N - 1) * C + cc) : (h16x8)(h16)0;
            }
        }
    if (a.xgate) { if (!gate_wait<true>(a.xgate, a.xexpect, a.err, a.timeout)) return; }
    else {
        // the S tiles whose codes this tile reads (same column block, the row tiles its rows fall into) - and the tiles that wrote the
        // scales: tile 0, tall form tiles 0 .. MML_NRED - 1
        if (t.w == 0) {
            const unsigned* f = a.codedone + ((size_t)a.src[item] * a.CB + bx) * a.P;
            const int by0 = t.r0 / a.R, by1 = (t.r1 - 1) / a.R;
            const int lane = threadIdx.x & 63;
            const int nsc = a.coop ? MML_NRED : 1;
            SpinClock clk;
            for (;;) {
                const int idx = lane < nsc ? lane : by0 + lane - nsc;
                const unsigned v = (lane < nsc || idx <= by1) ? ld_wt(f + min(idx, a.P - 1)) : a.seq;
                if (__builtin_amdgcn_ballot_w64((int)(v - a.seq) < 0) == 0) break;
                __builtin_amdgcn_s_sleep(2);
                if (clk.expired(a.timeout)) { failed = true; if (lane == 0) gate_fail(a.err); break; }
            }
        }
        if (__syncthreads_or(failed ? 1 : 0)) return;       // the codes never came: the state stays as it was
    }
    if (st.on()) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MML_STAMP(1);                                       // state tile in registers AND gate seen
    }
    const bool remote = a.remote != 0;
    h16x8 sc, mz;
    minmax_ld_scales<INT4>(pk, N, C, cc, remote, sc, mz);
    const int crows = INT4 ? N / 2 : N;
    u64 qb[KC];
#pragma unroll
    for (int j = 0; j < KC; ++j)
        if (j < kc) {
            const int cr = min((t.r0 / RPC) + t.w + NW * j, crows - 1);
            qb[j] = ld_wt_or_sys((const u64*)(pk + (size_t)cr * C + cc), remote);
        }
#pragma unroll
    for (int j = 0; j < KC; ++j)
        if (j < kc) {
#pragma unroll
            for (int h = 0; h < RPC; ++h) {
                const int row = t.r0 + (t.w + NW * j) * RPC + h;
                if (row < t.r1 && t.act) {
                    const h16x8 recv = minmax_recv<INT4>(qb[j], h, sc, mz);
                    st8nt(out + (size_t)row * C + t.c, base ? (bv[j * RPC + h] + recv) : recv);
                }
            }
        }
    if (st.on()) {
        MML_STAMP(2);                                       // codes landed, stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MML_STAMP(3);
    }
}

template <bool INT4, int RW>
__global__ __launch_bounds__(FUSED_NT, 4) void k_minmax_layer(BatchC batch, BatchD gated, MinMaxLayerArgs a) {
    __shared__ u64 sm[MML_NW][TILE_C];
    __shared__ u32x4 park[RW > 4 ? MML_PARK * FUSED_NT : 1];           // RW = 8: 32 KB more, still two workgroups a CU
    int b = blockIdx.x;
    // (Tried in round 5 for tall tensors: S and D interleaved column block by column block - S(0) S(1) D(0) S(2) D(1) ... - so that
    // reconstruction tiles stream their state in while statistics tiles sit out their scales' hops.  Config 4: 4.78 ms per step either
    // way, the stamped launch 121 instead of 106 us - the D tiles take the slots the NEXT block's S tiles need; what bounds the S phase is
    // a tile's lifetime in its slot, ~20 us of which ~8 move bytes.)
    if (b < a.n_st) {
        const int per = a.CB * a.P;
        const int z = b / per, rem = b - z * per;
        int bx, by;
        if (a.tall) { bx = rem / a.P; by = rem - bx * a.P; }
        else { by = rem / a.CB; bx = rem - by * a.CB; }
        minmax_layer_s_tile<INT4, RW>(batch.it[z], a, z, bx, by, sm, park);
        if (b == 0 && a.p2p.own) p2p_exchange_inline(a.codedone, a.seq, a.n_st, a.p2p, a.xgate, a.xexpect, a.err);     // packets complete = every S tile's codes flag
        return;
    }
    b -= a.n_st;
    const int per = a.CB * a.g_rb;
    const int item = b / per, rem = b - item * per;
    const int ty = rem / a.CB;
    minmax_layer_d_tile<INT4>(gated.it[item], a, item, rem - ty * a.CB, ty);
}

// int8 quantise (+EF)      compress_quantize.py:465-467 ; EF = dequantize_int8 :482 + main.py:232
__global__ __launch_bounds__(NTHR) void k_int8_quant(BatchC batch, int N, int C, int R, int flags) {
    const cfx_comp_item it = batch.it[blockIdx.z];
    const TileCoord t = tile_coord(N, C, R);
    signed char* q = (signed char*)it.packet;
    const h16* S = (const h16*)(q + (size_t)N * C);
    const short* Z = (const short*)(S + C);
    const h16* x = (const h16*)it.x;
    const h16* base = (const h16*)it.base;
    h16* nb = (h16*)it.new_base;
    const bool upd = (flags & CFX_FLAG_UPDATE_CACHE) && nb;
    const bool ef = !(flags & CFX_FLAG_NO_EF);
    h16x8 sc = (h16x8)(h16)1.0f, zp = (h16x8)(h16)0;
    if (t.act) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc[i] = S[t.c + i]; zp[i] = (h16)(float)Z[t.c + i]; }
    }
    float scf[8], scr[8];                                    // the channel's scale and its reciprocal: hdiv_r
#pragma unroll
    for (int i = 0; i < 8; ++i) { scf[i] = (float)sc[i]; scr[i] = __builtin_amdgcn_rcpf(scf[i]); }
    for (int r = t.r0 + t.w; r < t.r1; r += WAVES * UNROLL) {
        h16x8 xv[UNROLL], bv[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const int rr = r + WAVES * j;
            xv[j] = (h16x8)(h16)0; bv[j] = (h16x8)(h16)0;
            if (rr < t.r1 && t.act) {
                xv[j] = ld8nt(x + (size_t)rr * C + t.c);
                if (base) bv[j] = ld8nt(base + (size_t)rr * C + t.c);
            }
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const int rr = r + WAVES * j;
            if (rr < t.r1 && t.act) {
                const h16x8 d = xv[j] - bv[j];
                u64 outb = 0;
                h16x8 qh;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    h16 v = hrint(hdiv_r(d[i], scf[i], scr[i]) + zp[i]);   // round(x/scale + zp), fp16 after each op
                    if (hisnan(v)) v = (h16)0;
                    v = v < (h16)-128.0f ? (h16)-128.0f : v;
                    v = v > (h16)127.0f ? (h16)127.0f : v;
                    const int qi = (int)(float)v;
                    qh[i] = (h16)(float)qi;                                 // via int: rint(-0.3) = -0 must dequantise as +0
                    outb |= (u64)(unsigned char)(signed char)qi << (8 * i);
                }
                *reinterpret_cast<u64*>(q + (size_t)rr * C + t.c) = outb;
                if (upd) {
                    h16x8 o;
                    if (ef) {
                        const h16x8 recv = (qh - zp) * sc;                // (q - zp) * scale
                        o = base ? (bv[j] + recv) : recv;
                    } else o = xv[j];
                    st8nt(nb + (size_t)rr * C + t.c, o);
                }
            }
        }
    }
}

__global__ __launch_bounds__(NTHR) void k_int8_dequant(BatchD batch, int N, int C, int R, unsigned* pre, unsigned pre_val) {
    // lane: publish `pre` first - the launch in front of this one in the stream (the previous peer's reconstruction) has finished
    if (pre && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0) st_wt(pre, pre_val);
    const cfx_decomp_item it = batch.it[blockIdx.z];
    const TileCoord t = tile_coord(N, C, R);
    const signed char* q = (const signed char*)it.packet;
    const h16* S = (const h16*)(q + (size_t)N * C);
    const short* Z = (const short*)(S + C);
    const h16* base = (const h16*)it.base;
    h16* out = (h16*)it.recon;
    h16x8 sc = (h16x8)(h16)1.0f, zp = (h16x8)(h16)0;
    if (t.act) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc[i] = S[t.c + i]; zp[i] = (h16)(float)Z[t.c + i]; }
    }
    for (int r = t.r0 + t.w; r < t.r1; r += WAVES * UNROLL) {
        h16x8 bv[UNROLL];
        u64 qb[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const int rr = r + WAVES * j;
            bv[j] = (h16x8)(h16)0; qb[j] = 0;
            if (rr < t.r1 && t.act) {
                if (base) bv[j] = ld8nt(base + (size_t)rr * C + t.c);
                qb[j] = *reinterpret_cast<const u64*>(q + (size_t)rr * C + t.c);
            }
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const int rr = r + WAVES * j;
            if (rr < t.r1 && t.act) {
                h16x8 qh;
#pragma unroll
                for (int i = 0; i < 8; ++i) qh[i] = (h16)(float)(int)(signed char)(qb[j] >> (8 * i));
                const h16x8 recv = (qh - zp) * sc;
                st8nt(out + (size_t)rr * C + t.c, base ? (bv[j] + recv) : recv);
            }
        }
    }
}

// int4 quantise (+EF): one wave step handles the row PAIR (2k, 2k+1) because the reference packs two rows per
// byte along N (compress_quantize.py:566-573): byte[k][c] = q[2k][c] | q[2k+1][c] << 4.
// R (rows per tile) is even; pair index space = rows/2.
__global__ __launch_bounds__(NTHR) void k_int4_quant(BatchC batch, int N, int C, int R, int flags) {
    const cfx_comp_item it = batch.it[blockIdx.z];
    const TileCoord t = tile_coord(N, C, R);
    unsigned char* q = (unsigned char*)it.packet;
    const h16* S = (const h16*)(q + (size_t)(N / 2) * C);
    const h16* M = S + C;
    const h16* x = (const h16*)it.x;
    const h16* base = (const h16*)it.base;
    h16* nb = (h16*)it.new_base;
    const bool upd = (flags & CFX_FLAG_UPDATE_CACHE) && nb;
    const bool ef = !(flags & CFX_FLAG_NO_EF);
    const bool al16 = ((((uintptr_t)S) | ((uintptr_t)M)) & 15) == 0;
    h16x8 sc = (h16x8)(h16)1.0f, mn = (h16x8)(h16)0;
    if (t.act) { sc = ld8_tail(S + t.c, al16); mn = ld8_tail(M + t.c, al16); }
    float scf[8], scr[8];                                    // the channel's scale and its reciprocal: hdiv_r
#pragma unroll
    for (int i = 0; i < 8; ++i) { scf[i] = (float)sc[i]; scr[i] = __builtin_amdgcn_rcpf(scf[i]); }
    const int k0 = t.r0 >> 1, k1 = t.r1 >> 1;
    constexpr int U2 = 1;   // one row PAIR per wave step
    for (int k = k0 + t.w; k < k1; k += WAVES * U2) {
        h16x8 xv[U2][2], bv[U2][2];
#pragma unroll
        for (int j = 0; j < U2; ++j) {
            const int kk = k + WAVES * j;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                xv[j][h] = (h16x8)(h16)0; bv[j][h] = (h16x8)(h16)0;
                if (kk < k1 && t.act) {
                    xv[j][h] = ld8nt(x + (size_t)(2 * kk + h) * C + t.c);
                    if (base) bv[j][h] = ld8nt(base + (size_t)(2 * kk + h) * C + t.c);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < U2; ++j) {
            const int kk = k + WAVES * j;
            if (kk < k1 && t.act) {
                u64 outb = 0;
                h16x8 qh[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const h16x8 d = xv[j][h] - bv[j][h];
                    const h16x8 dm = d - mn;                               // (r - min)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        h16 v = hrint(hdiv_r(dm[i], scf[i], scr[i]));
                        if (hisnan(v)) v = (h16)0;
                        v = v < (h16)0 ? (h16)0 : v;
                        v = v > (h16)15.0f ? (h16)15.0f : v;
                        const unsigned qi = (unsigned)(float)v & 15u;
                        qh[h][i] = (h16)(float)qi;
                        outb |= (u64)qi << (8 * i + 4 * h);
                    }
                }
                *reinterpret_cast<u64*>(q + (size_t)kk * C + t.c) = outb;
                if (upd) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        h16x8 o;
                        if (ef) {
                            const h16x8 recv = qh[h] * sc + mn;            // q*scale + min (two roundings; contraction is off)
                            o = base ? (bv[j][h] + recv) : recv;
                        } else o = xv[j][h];
                        st8nt(nb + (size_t)(2 * kk + h) * C + t.c, o);
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(NTHR) void k_int4_dequant(BatchD batch, int N, int C, int R, unsigned* pre, unsigned pre_val) {
    // lane: publish `pre` first - the launch in front of this one in the stream (the previous peer's reconstruction) has finished
    if (pre && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0) st_wt(pre, pre_val);
    const cfx_decomp_item it = batch.it[blockIdx.z];
    const TileCoord t = tile_coord(N, C, R);
    const unsigned char* q = (const unsigned char*)it.packet;
    const h16* S = (const h16*)(q + (size_t)(N / 2) * C);
    const h16* M = S + C;
    const h16* base = (const h16*)it.base;
    h16* out = (h16*)it.recon;
    const bool al16 = ((((uintptr_t)S) | ((uintptr_t)M)) & 15) == 0;
    h16x8 sc = (h16x8)(h16)1.0f, mn = (h16x8)(h16)0;
    if (t.act) { sc = ld8_tail(S + t.c, al16); mn = ld8_tail(M + t.c, al16); }
    const int k0 = t.r0 >> 1, k1 = t.r1 >> 1;
    constexpr int U2 = 1;   // one row PAIR per wave step
    for (int k = k0 + t.w; k < k1; k += WAVES * U2) {
        h16x8 bv[U2][2];
        u64 qb[U2];
#pragma unroll
        for (int j = 0; j < U2; ++j) {
            const int kk = k + WAVES * j;
            qb[j] = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bv[j][h] = (h16x8)(h16)0;
                if (kk < k1 && t.act && base) bv[j][h] = ld8nt(base + (size_t)(2 * kk + h) * C + t.c);
            }
            if (kk < k1 && t.act) qb[j] = *reinterpret_cast<const u64*>(q + (size_t)kk * C + t.c);
        }
#pragma unroll
        for (int j = 0; j < U2; ++j) {
            const int kk = k + WAVES * j;
            if (kk < k1 && t.act) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    h16x8 qh;
#pragma unroll
                    for (int i = 0; i < 8; ++i) qh[i] = (h16)(float)((qb[j] >> (8 * i + 4 * h)) & 15u);
                    const h16x8 recv = qh * sc + mn;
                    st8nt(out + (size_t)(2 * kk + h) * C + t.c, base ? (bv[j][h] + recv) : recv);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host side: this family's launches (validated and dispatched by cfx_api.hip)
// ---------------------------------------------------------------------------------------------------
int cfx_i_minmax_compress(CompressCall& cc) {
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
    const dim3 grid(CB, P, batch);
    const int Rq = auto_rows(ctx, N, C, batch, true);       // apply passes: same tile map as the (unfused) statistics pass
    const dim3 gridq(CB, (N + Rq - 1) / Rq, batch);
    (void)grid;
    // ---- the min/max codecs' layer in ONE launch (k_minmax_layer): statistics tile in registers, in-launch scales, codes from registers,
    // gated reconstruction.  Needs every statistics workgroup CO-RESIDENT on the stream's CUs (each waits for its column block's scales
    // holding its tile); otherwise - tall tensors - the multi-launch forms below run (identical results). ----
    const bool int4 = codec == CFX_CODEC_INT4;
    const int RL = (N + 31) / 32 <= MML_MAX_P / 2 ? 32 : 64;       // S tile height: 32 rows; 64 where that keeps the partials per channel <= MML_MAX_P
    const int PL = (N + RL - 1) / RL;
    bool tall = PL > MML_MAX_P;
    int g_rb = (N + FUSED_NW * MML_KC - 1) / (FUSED_NW * MML_KC);
    int g_R = ((N + g_rb - 1) / g_rb + 15) / 16 * 16;
    const long n_st = (long)CB * PL * batch;
    long n_g = (long)CB * g_rb * n_gated;
    bool layer = fused && ctx->gated_on && !ctx->dev_probe && C % 16 == 0 && stream_cus >= 128 && ctx->stats_rows == 0 && PL <= MML_MAX_P_TALL &&
                 n_st <= MML_MAX_TILES && !capturing;
    if (layer && !n_gated && !xg) {
        // a plain compress call may be under stream capture (the ungated launches are capturable: include/cfx.h); the layer launch is not -
        // its tags and flags are launch arguments that advance with every launch, a replayed node would meet its own old tags
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs != hipStreamCaptureStatusNone) layer = false;
    }
    signed char src[CFX_MAX_BATCH];
    memset(src, 0, sizeof(src));
    if (layer && n_gated && !xg) {
        // loop-back: a gated item waits for the S tiles of the own tensor whose packet it reads
        for (int g_ = 0; g_ < n_gated && layer; ++g_) {
            int m = -1;
            for (int i = 0; i < batch; ++i)
                if (gated[g_].packet == items[i].packet) m = i;
            if (m < 0) layer = false;
            src[g_] = (signed char)m;
        }
    }
    if (layer) {
        static int per_cu4 = 0, per_cu8 = 0;
        int& per_cu = int4 ? per_cu4 : per_cu8;
        if (!per_cu) {
            const hipError_t oe = int4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_minmax_layer<true, 8>, FUSED_NT, 0)
                                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_minmax_layer<false, 8>, FUSED_NT, 0);
            if (oe != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
        }
        const long slots = (long)per_cu * stream_cus;
        // every S tile co-resident; or - tall form - a column block's P tiles at a time, which in-order dispatch (workgroup i on XCD
        // i % 8, every XCD walking its share in order) guarantees as long as they fit a fraction of the slots
        if (n_st > slots - 8) tall = true;
        if (tall && (RL != 64 || PL * 4 > slots)) layer = false;
        // group D's tile height: as low as keeps the group within the slots group S leaves free (small tensors: 60 workgroups of 112
        // rows would leave most of the chip idle behind the gate - the tile's arithmetic, not HBM, is what takes the time there)
        if (n_gated && n_st < slots) {
            const long per_rb = (long)CB * n_gated, fit = (slots - n_st) / per_rb;      // row tiles per tensor that fit
            if (fit > g_rb) {
                const int rows = (int)((N + fit - 1) / fit);
                g_R = std::max(16, (rows + 15) / 16 * 16);
                g_rb = (N + g_R - 1) / g_R;
                n_g = (long)CB * g_rb * n_gated;
            }
        }
        // a collective KERNEL has to find CUs while group D waits (see the 1-bit exchange layer): 32 workgroup slots left free
        if (layer && xg && xg->needs_room && n_g + 32 > slots) layer = false;
    }
    if (xg && !layer) n_gated = 0;      // compress only: the caller runs its exchange and the reconstruction behind this call
    if (layer) {
        MinMaxLayerArgs a;
        memset(&a, 0, sizeof(a));
        a.N = N; a.C = C; a.CB = CB; a.P = PL; a.R = RL; a.n_st = (int)n_st;
        a.g_R = g_R; a.g_rb = g_rb; a.n_g = (int)n_g;
        a.codec = codec; a.flags = flags;
        const unsigned ring = slot / TICK_RING;
        // the partials' arena of this ring (stream): context-owned because its words are TAGGED - a stale word must never carry a tag a
        // later launch expects, so it starts zeroed and only ever takes this context's sequence numbers (which are not reused)
        a.part_stride = (size_t)(PL + 1) * C;
        const size_t need = a.part_stride * batch * sizeof(u64);
        if (!ctx->mml_arena_owned[ring] || ctx->mml_arena_owner[ring] != stream) {
            // the ring - and with it the arena - changed hands (more than CFX_RING_STREAMS streams issue compress launches): whatever its
            // previous owner still has in flight reads this arena.  Rare by construction; wait for it
            if (ctx->mml_arena_owned[ring]) (void)hipDeviceSynchronize();
            ctx->mml_arena_owner[ring] = stream;
            ctx->mml_arena_owned[ring] = true;
        }
        if (ctx->mml_arena_bytes[ring] < need) {
            if (ctx->mml_arena[ring]) (void)hipFree(ctx->mml_arena[ring]);      // (synchronises the device: no launch still reads it)
            ctx->mml_arena[ring] = nullptr;
            ctx->mml_arena_bytes[ring] = 0;
            const size_t cap = (std::max(need, (size_t)4 << 20) + 4095) & ~(size_t)4095;
            void* m = nullptr;
            // (zeroed IN the launch stream: a plain hipMemset runs on the NULL stream, which a non-blocking stream does not wait for)
            if (hipMalloc(&m, cap) != hipSuccess || hipMemsetAsync(m, 0, cap, s) != hipSuccess) {
                (void)hipGetLastError();
                if (m) (void)hipFree(m);
                return fail(ctx, CFX_ERR_LAUNCH, "min/max layer launch: cannot allocate the partials' arena");
            }
            ctx->mml_arena[ring] = (u64*)m;
            ctx->mml_arena_bytes[ring] = cap;
        }
        a.part = ctx->mml_arena[ring];
        a.probe = cfx_i_probe(ctx);
        a.codedone = ctx->colgate + (size_t)ring * MML_MAX_TILES;
        a.tall = tall ? 1 : 0;
        a.coop = (tall || PL > 32) ? 1 : 0;
        a.seq = ++ctx->mml_seq;
        if (a.seq >= 0x7FFFFFFFu) {
            // 2.1 billion launches later: the tiles' "codes published" flags are compared as signed distances (a flag from an earlier
            // launch - or a word never written - must read as BEHIND this launch's number), and a tagged word a smaller layout has not
            // rewritten since could carry a number again - start over at 1 (arenas and flags zeroed, nothing in flight)
            (void)hipDeviceSynchronize();
            for (int i = 0; i < CFX_RING_STREAMS; ++i)
                if (ctx->mml_arena[i]) (void)hipMemset(ctx->mml_arena[i], 0, ctx->mml_arena_bytes[i]);
            (void)hipMemset(ctx->colgate, 0, (size_t)CFX_RING_STREAMS * MML_MAX_TILES * sizeof(unsigned));
            (void)hipDeviceSynchronize();
            ctx->mml_seq = 0;
            a.seq = ++ctx->mml_seq;
        }
        a.err = ctx->gate_err;
        a.timeout = ctx->gate_timeout;
        memcpy(a.src, src, sizeof(src));
        if (xg && n_gated) {
            a.xgate = ctx->gate + (size_t)slot * GATE_STRIDE + GATE_BLOCK;
            a.xexpect = ++ctx->gate_expect[3 * slot + 1];
            a.remote = xg->remote;
            fill_p2p(ctx, xg, a.p2p);
            xg->taken = 1;
            xg->p_gate = a.codedone; xg->p_expect = a.seq; xg->p_count = (int)n_st;      // "packets complete" = every S tile's codes flag
            xg->f_gate = a.xgate; xg->f_expect = a.xexpect;
        }
        const dim3 g((unsigned)(n_st + n_g));
        if (RL == 32) {
            if (int4) LAUNCH(ctx, KID_ABSMEAN_COMPRESS_GATED, s, (k_minmax_layer<true, 4>), g, dim3(FUSED_NT), 0, s, b, gd, a);
            else LAUNCH(ctx, KID_ABSMEAN_COMPRESS_GATED, s, (k_minmax_layer<false, 4>), g, dim3(FUSED_NT), 0, s, b, gd, a);
        } else {
            if (int4) LAUNCH(ctx, KID_ABSMEAN_COMPRESS_GATED, s, (k_minmax_layer<true, 8>), g, dim3(FUSED_NT), 0, s, b, gd, a);
            else LAUNCH(ctx, KID_ABSMEAN_COMPRESS_GATED, s, (k_minmax_layer<false, 8>), g, dim3(FUSED_NT), 0, s, b, gd, a);
        }
        return check_launch(ctx, "min/max layer launch");
    }
    if (fused) {
        LAUNCH(ctx, KID_MINMAX_COMPRESS, s, k_minmax_compress, dim3(CB * P * batch), dim3(NTHR), 0, s, b, N, C, R, CB, P, codec, ws, wstride, tick);
    } else {
        LAUNCH(ctx, KID_MINMAX_STATS, s, k_minmax_stats, grid, dim3(NTHR), 0, s, b, N, C, R, ws, wstride);
        LAUNCH(ctx, KID_MINMAX_FINALIZE, s, k_minmax_finalize, dim3((C + 255) / 256, batch), dim3(1024), 0, s, b, N, C, P, codec, (const u64*)ws, wstride);
    }
    if (codec == CFX_CODEC_INT4) LAUNCH(ctx, KID_INT4_QUANT, s, k_int4_quant, gridq, dim3(NTHR), 0, s, b, N, C, Rq, flags);
    else LAUNCH(ctx, KID_INT8_QUANT, s, k_int8_quant, gridq, dim3(NTHR), 0, s, b, N, C, Rq, flags);
    if (n_gated) {
        const int rcg = decompress_impl(ctx, codec, N, C, param, n_gated, gated, stream, nullptr, 0u);
        if (rcg != CFX_OK) return rcg;
    }
    return check_launch(ctx, "compress launch");
}

int cfx_i_minmax_decompress(cfx_ctx* ctx, int codec, int N, int C, int batch, const BatchD& b, int R, void* stream, unsigned* pre, unsigned pre_val) {
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((C + TILE_C - 1) / TILE_C, (N + R - 1) / R, batch);
    if (codec == CFX_CODEC_INT4) LAUNCH(ctx, KID_INT4_DEQUANT, s, k_int4_dequant, grid, dim3(NTHR), 0, s, b, N, C, R, pre, pre_val);
    else LAUNCH(ctx, KID_INT8_DEQUANT, s, k_int8_dequant, grid, dim3(NTHR), 0, s, b, N, C, R, pre, pre_val);
    return check_launch(ctx, "decompress launch");
}

// libcfx.so - hand-written gfx950 (MI355X / CDNA4) kernels for CompactFusion's residual-compressed
// activation exchange, behind the C-ABI of include/cfx.h.
//
// Design:
//   * Every kernel is an HBM-bound streaming pass over (N, C) fp16 tensors.  A wavefront (64 lanes)
//     owns 512 contiguous channels of one row: each lane moves 16 B (8 halves) per access, so one
//     wave instruction covers 1 KiB contiguous - the coalescing sweet spot on CDNA4.
//   * A workgroup is 4 waves = a tile of R rows x 512 channels; waves interleave over the rows and keep
//     several rows of loads in flight (the tiles are too small for occupancy alone to hide HBM latency).
//   * The scale prologue of the reference (5 eager full-tensor passes, fastpath.py:150-166) is a global
//     reduction, so compress is stats-pass -> tiny finalize -> apply-pass.  The stats pass accumulates
//     |x-base| EXACTLY as 64-bit integers in units of 2^-24 (fp16 values are multiples of 2^-24): the
//     scales are therefore independent of tiling, reduction order and run - bit-reproducible - and equal
//     to oracle/ref_np.py bit for bit.  Partial sums go to a caller-provided workspace (no atomics).
//   * For the 1-bit codec the packed signs do not depend on the scales, so the stats pass already emits
//     them and the error-feedback pass is literally the receiver's dequant+add kernel run on the sender's
//     own packet: sender and receiver state cannot diverge.
//   * Tile -> workgroup mapping is identical in the stats and apply passes, so a tile is re-read by a
//     workgroup with the same index, i.e. (as dispatched on gfx950, block b -> XCD b % 8) from the same
//     XCD's L2 where the first pass left it.
//   * fp16 arithmetic is done with native correctly-rounded fp16 instructions, one rounding per reference
//     op (compile with -ffp-contract=off: an fma would skip the rounding of u*v that fastpath.py:109 has).
//
// Reference citations are relative to /root/reference/xfuser/compact/.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dlfcn.h>
#include "cfx.h"
#include "cfx_internal.h"

// cfx_lowrank.hip (the rank-K scales of the 1-bit codec are low-rank factors of |x - base|)
#define CFX_I_FLAG_LR_FACTORS_ONLY 0x100
#define CFX_I_FLAG_LR_ABS 0x200
extern "C" CFX_HIDDEN size_t cfx_i_lr_workspace_bytes_any(int N, int C, int rank, int batch);
extern "C" CFX_HIDDEN void cfx_i_lr_factor_offsets(int N, int C, int rank, size_t* offU16, size_t* offV16, size_t* per);

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64;

#define TILE_C 512      // channels per wave-row = 64 lanes x 8 halves
#define WAVES 4         // waves per workgroup
#define NTHR (WAVES * 64)
#define UNROLL 2        // rows in flight per wave, apply / dequant kernels (measured best with R = 8: tools/kbench.hip)
#define UNROLL_S 4      // rows in flight per wave, statistics kernels (one exposure of HBM latency per tile of 16 rows)

struct BatchC { cfx_comp_item it[CFX_MAX_BATCH]; };
struct BatchD { cfx_decomp_item it[CFX_MAX_BATCH]; };

// ---------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ u16 hbits(h16 v) { return __builtin_bit_cast(u16, v); }
__device__ __forceinline__ h16 hfrom(u16 v) { return __builtin_bit_cast(h16, v); }

__device__ __forceinline__ h16x8 ld8(const h16* p) { return *reinterpret_cast<const h16x8*>(p); }
__device__ __forceinline__ void st8(h16* p, h16x8 v) { *reinterpret_cast<h16x8*>(p) = v; }

// Streaming (non-temporal) forms for data touched once per launch: measured +10 % on the dequant stream (tools/kbench.hip).
__device__ __forceinline__ h16x8 ld8nt(const h16* p) { return __builtin_nontemporal_load(reinterpret_cast<const h16x8*>(p)); }
__device__ __forceinline__ void st8nt(h16* p, h16x8 v) { __builtin_nontemporal_store(v, reinterpret_cast<h16x8*>(p)); }

// 8 halves from an address that is only guaranteed 2-byte aligned (packet tail sections).
__device__ __forceinline__ h16x8 ld8_tail(const h16* p, bool al16) {
    if (al16) return ld8(p);
    h16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = p[i];
    return r;
}

// |h| as an integer count of 2^-24 (exact for finite fp16; garbage-but-finite for inf/nan).
__device__ __forceinline__ u64 habs_units(u16 b) {
    const unsigned e = (b >> 10) & 31u, m = b & 1023u;
    const unsigned t = e ? (m | 1024u) : m;
    const unsigned sh = e ? e - 1u : 0u;
    return (u64)t << sh;
}

// fp16( fp32(exact_sum * 2^-24) / fp32(n) ) - oracle/ref_np.py mean16_exact
__device__ __forceinline__ h16 mean16(u64 units, int n) {
    // u64 -> fp32 through fp64: exact below 2^53, then ONE rounding to fp32 = the direct conversion, in three instructions
    // instead of the emulated 64-bit integer conversion
    const float s = (float)(double)units * 0x1p-24f;
    return (h16)(s / (float)n);
}

// Wave-wide sum of a u32 with DPP adds (no LDS crossbar round trips): quad, half-row, row, then the two row broadcasts;
// the total is in lane 63.  Caller guarantees the total fits 32 bits.
__device__ __forceinline__ unsigned wave_sum_u32_dpp(unsigned v) {
    v += __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0u, v, 0x141, 0xf, 0xf, true);    // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0u, v, 0x140, 0xf, 0xf, true);    // row_mirror: every lane holds its row's total
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, true);    // row_bcast15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, true);    // row_bcast31 into rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Inter-workgroup hand-off inside ONE launch (the in-launch finalize below).  Per-XCD L2s are not coherent with each other
// and a CU's L1 is never refreshed by other CUs' stores, so partial sums that another workgroup will read in this launch
// are stored WRITE-THROUGH (relaxed agent-scope atomic store = `global_store ... sc1`: the line leaves the XCD) and read
// back with relaxed agent-scope loads (`global_load ... sc1`: bypasses the reader's L1) - no release / acquire fences,
// which would write back / invalidate whole caches per workgroup (MI355X_MICROARCH.md, "Workgroup dispatch ... visibility").
__device__ __forceinline__ void st_wt(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_wt(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_wt(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_wt(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_wt(u16* p, u16 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u16 ld_wt(const u16* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned char ld_wt(const unsigned char* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_wt(unsigned char* p, unsigned char v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// system scope: the word may live in ANOTHER GPU's memory (packets read in place through an IPC mapping, cfx_plan_add_exchange_layer_p2p)
__device__ __forceinline__ u64 ld_sys(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ u16 ld_sys(const u16* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned char ld_sys(const unsigned char* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// 16-byte write-through store (an agent-scope atomic store lowers to `sc1` only up to 8 bytes).  hipcc does not count an asm
// store: the publishing wave drains it with its own `s_waitcnt vmcnt(0)`; the trailing s_nop keeps the data registers alive
// until the store has read them (cdna_hip_programming.md 5.7).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_wt(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic, NOT for its outstanding global stores
// (__syncthreads() also drains vmcnt: behind write-through stores that is a fabric round trip, ~1 us, per barrier).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A failed wait (bounded spin) is counted in the context's error word, which lives in pinned HOST memory: the next native call
// on the context reports it without a device synchronisation (cfx_gate_errors, CFX_ERR_GATE).
__device__ __forceinline__ void gate_fail(unsigned* err) {
    if (err) (void)__hip_atomic_fetch_add(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// One lane waits until *flag has reached `value` (monotonic epochs: signed distance), polling with L1-bypassing loads; gives
// up after `timeout` ticks of the 100 MHz wall clock.
__device__ __forceinline__ void flag_spin(const unsigned* flag, unsigned value, unsigned* err, long long timeout) {
    const long long t0 = wall_clock64();
    while ((int)(ld_wt(flag) - value) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > timeout) { gate_fail(err); break; }
    }
}

// Partial sums that another workgroup of the same launch reduces (WT paths) travel as 32-bit words: a last arriver pulls fresh
// cross-CU data at only ~65 GB/s, and for real activations every partial fits (a tile's sum of |d| would have to reach 256).
// A sum that does not fit leaves the sentinel in the 32-bit word and the exact value in the 64-bit array the non-fused kernels
// use; the reader follows the sentinel.  Layout per tensor: [rowpart u64 N x CB][colpart u64 ceil(N/16) x C][rowpart u32][colpart u32].
#define PART_SAT 0xFFFFFFFFu
__device__ __forceinline__ unsigned* part32_of(const u64* rowpart, int N, int C, int CB) {
    return (unsigned*)(rowpart + (size_t)N * CB + (size_t)((N + 15) / 16) * C);
}
__device__ __forceinline__ void put_part(unsigned* p32, u64* p64, size_t i, u64 v) {
    const bool big = v >= (u64)PART_SAT;
    st_wt(p32 + i, big ? PART_SAT : (unsigned)v);
    if (big) st_wt(p64 + i, v);
}

// The LAYER launches (GATED) hand their partial sums over as TAGGED 8-byte words in an arena the context owns (zeroed once, like the
// min/max layer's): {24-bit launch tag | 40-bit value} in ONE store - the reader polls the data itself: no drain of the stores, no
// ticket, no second round trip for the data once a counter says it is there.  40 bits of 2^-24 units hold a partial up to 65536 (a
// 512-channel row partial at an average |d| of 128: round 4's 32-bit words gave out at 0.5 and cost the launch a second round trip);
// beyond that the word carries TAG_SAT and the exact sum is in the 64-bit side array, stored AND drained first.
#define TAG_SAT ((u64)0xFFFFFFFFFFull)
__device__ __forceinline__ void put_tagged(u64* t, u64* side, size_t i, u64 v, u64 tagbits) {
    if (v >= TAG_SAT) {
        st_wt(side + i, v);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        v = TAG_SAT;
    }
    st_wt(t + i, tagbits | v);
}
__device__ __forceinline__ bool tag_is(u64 w, u64 tagbits) { return ((w ^ tagbits) >> 40) == 0; }
struct TagArena {
    u64* trow;             // [CB][N] tagged row partials of this tensor
    u64* tcol;             // [P][C] tagged column partials
    u64* tU;               // [N] the finished token scales as tagged words {tag | fp16 bits}: a reconstruction workgroup of the same launch
    u64* tV;               // [C] ... and the channel scales       polls THESE - no drain of the packet's copy, no arrival counter, no relay
    u64* tdone;            // [CB][P] 2-bit layer: "this tile's codes are in memory" (the launch's tag), written by the tile once its stores have drained
    u64 tagbits;           // the launch's tag << 40
};

__device__ __forceinline__ TagArena tag_arena_of(u64* arena, size_t stride, int z, int N, int C, int CB, int P, unsigned tag) {
    TagArena ta;
    ta.trow = arena ? arena + (size_t)z * stride : nullptr;
    ta.tcol = arena ? ta.trow + (size_t)N * CB : nullptr;
    ta.tU = arena ? ta.tcol + (size_t)P * C : nullptr;
    ta.tV = arena ? ta.tU + N : nullptr;
    ta.tdone = arena ? ta.tV + C : nullptr;
    ta.tagbits = (u64)tag << 40;
    return ta;
}
__host__ __device__ inline size_t tag_arena_words(int N, int C, int CB, int P) { return (size_t)N * CB + (size_t)P * C + (size_t)N + (size_t)C + (size_t)CB * P; }

__device__ __forceinline__ h16x8 habs8(h16x8 v) {
    u16x8 b = __builtin_bit_cast(u16x8, v);
    b &= (u16)0x7fff;
    return __builtin_bit_cast(h16x8, b);
}

// 2-bit codes -> received values (levels +-0.5 thr, +-2 thr)
__device__ __forceinline__ h16x8 int2_recv(u16 code, h16x8 thr) {
    h16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned idx = (code >> (2 * i)) & 3u;
        const h16 lvl = (idx & 1u) ? (h16)2.0 * thr[i] : (h16)0.5 * thr[i];   // fastpath.py:565-568
        r[i] = (idx & 2u) ? lvl : -lvl;                                          // (+-1) * lvl
    }
    return r;
}

struct TileCoord {
    int lane, w, c, r0, r1;
    bool act;
};
__device__ __forceinline__ TileCoord tile_coord_at(int bx, int by, int N, int C, int R) {
    TileCoord t;
    t.lane = threadIdx.x & 63;
    t.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    t.c = bx * TILE_C + t.lane * 8;
    t.act = t.c < C;
    t.r0 = by * R;
    t.r1 = min(N, t.r0 + R);
    return t;
}
__device__ __forceinline__ TileCoord tile_coord(int N, int C, int R) { return tile_coord_at(blockIdx.x, blockIdx.y, N, C, R); }

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
// QUANT: x and the factor workspace in, bits + factors + new state out; else packet in, reconstruction out.  K <= 8.
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
    h16 v[8][8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[e][k] = (t.act && k < K) ? VT[(size_t)(t.c + e) * K + k] : (h16)0;
    if (QUANT) {
        // the factors go into the packet as they are: V by the first row block, U by the first column block
        if (blockIdx.y == 0 && t.act)
            for (int e = 0; e < 8; ++e)
                for (int k = 0; k < K; ++k) Vp[(size_t)(t.c + e) * K + k] = v[e][k];
        if (blockIdx.x == 0)
            for (int i = threadIdx.x; i < (t.r1 - t.r0) * K; i += NTHR) Up[(size_t)t.r0 * K + i] = U[(size_t)t.r0 * K + i];
    }
    for (int r = t.r0 + t.w; r < t.r1; r += WAVES) {
        if (!t.act) continue;
        h16 u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k] = (k < K) ? U[(size_t)r * K + k] : (h16)0;
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
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += (float)(h16)(u[k] * v[e][k]);       // fp16 product (one rounding), fp32 sum in index order
            const h16 sc = (h16)acc;
            const h16 recv = ((byte >> e) & 1u) ? sc : -sc;
            o[e] = base ? (h16)(bv[e] + recv) : recv;
        }
        if (QUANT && !ef) o = xv;                                   // error feedback off: the state becomes the activation (main.py:233)
        st8(out + (size_t)r * C + t.c, o);
    }
}

// Gated reconstruction: the same arithmetic for a packet that workgroups of THIS launch are still producing (the compress group
// of k_absmean_compress).  A workgroup first pulls its whole tile of the state into registers - K rows per wave, bandwidth work
// that does not depend on the packet and overlaps the compress group's reduction tail, which is pure latency - then one lane
// polls the gate (relaxed, s_sleep), and the tile is finished from registers: sign bits and scales are read with agent-scope
// loads (the producers stored them write-through), so no acquire fence.  The compress workgroups precede the gated ones in
// dispatch order and never wait on anything, so the wait always ends; a bounded spin turns a lost arrival into an error word
// (cfx_gate_errors) instead of a hung GPU.
#ifndef GATE_KR
#define GATE_KR 14
#endif                         // rows of its tile a wave holds in registers, 1-bit launch: tiles of up to FUSED_NW * 14 rows - a (544, C)
                               // tensor is 5 row blocks of 112, 480 gated workgroups for 16 tensors: 308 resident from the start,
                               // the rest take the slots the statistics workgroups leave at ~10 us, well before the gate.  Measured
                               // on one box: 17 rows (4 blocks, 384 workgroups) 1.53 ms per step, 14-16 rows 1.50, 12 rows and fewer
                               // (>= 576 workgroups: some only start after the gate) 1.78-1.85
#ifndef GATE_KR2
#define GATE_KR2 17
#endif
//            // 2-bit launch: 17 rows in registers ...
#ifndef GATE_KL
#define GATE_KL 6
#endif                         // ... plus, in the 2-bit layer launch, 6 rows in LDS (16 bytes per lane and row, 48 KB a workgroup):
                               // there the statistics workgroups stay resident until they have quantised their tiles, so a gated
                               // workgroup that is not resident from the start only gets a slot - and pulls its tile - after the
                               // gate; with 23 rows a wave a (544, C) tensor is 3 row blocks and 204 + 14 x 6 x 3 = 456 workgroups all
                               // fit (2 / CU).  In the 1-bit launch the statistics workgroups retire early, the gated workgroups that
                               // take over their slots spread the preload burst, and that measured faster than the all-resident
                               // forms (1.60 vs 1.77 - 1.91 ms per step)
#define GATE_LDS_ROWS ((GATE_KL * FUSED_NT * 16 + TILE_C * 8 - 1) / (TILE_C * 8))   // rows of the u64[..][TILE_C] LDS array of the 2-bit layer kernel (>= FUSED_NW)
#ifndef GATE_LOCAL_SLEEP
#define GATE_LOCAL_SLEEP 1        // s_sleep units between two polls of the XCD-local word (L2 hits)
#endif
#define GATE_LINE 16           // u32 words per 64-byte line
#define GATE_BLOCK (25 * GATE_LINE)   // a gate block: the arrival counter's line, then per XCD an "open" word, a local word, a relay claim word, a line each.
#define GATE_STRIDE (3 * GATE_BLOCK)  // three gate blocks per ticket-ring slot (the 2-bit exchange layer: scales gate, codes gate, external gate)
                               // Pollers never touch the counter's line: one line serves ~90 accesses per us, and a few hundred
                               // pollers on it queue every arrival behind them (measured: the compress tail went from 12 to 24 us)
// arrival of `inc` units; whoever completes the count opens the gate for every XCD's pollers
// FEW arrivals (the last-arriver jobs of a launch: a dozen): nothing is returned, nobody writes "open" words - the 8 relays poll the
// counter itself (a dozen atomics are not held up by 8 readers; hundreds of tile arrivals were, see gate_wait)
__device__ __forceinline__ void gate_arrive_few(unsigned* gate, unsigned inc) {
    (void)__hip_atomic_fetch_add(gate, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gate_arrive(unsigned* gate, unsigned inc, unsigned expect) {
    const unsigned old = __hip_atomic_fetch_add(gate, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + inc == expect) {
#pragma unroll
        for (int x = 0; x < 8; ++x) st_wt(gate + (1 + x) * GATE_LINE, expect);
    }
}
// one lane polls this XCD's "open" word (relaxed, s_sleep), then the workgroup barrier releases everybody
// Waiting for a gate: a few hundred workgroups polling through the fabric slow the compress group's reduction chain down (every
// poll of a remotely written word is a fabric read; measured on some boxes: 1.68 -> 1.53 ms per step when the pollers merely
// start 2 us later).  So only ONE workgroup per XCD - the first to claim the XCD's relay word for this launch - polls the word the
// gate's last arriver writes for that XCD; when it opens, the relay stores a second, XCD-LOCAL word with a plain store (the line
// stays in that XCD's L2) and everybody else on the XCD polls that one with L1-bypassing loads that the XCD's L2 serves - no
// fabric traffic.  Every 16th poll a waiter looks at the fabric word itself, so nothing depends on the relay or on the XCD
// number being right (a workgroup that mis-identifies its XCD just waits ~2 us longer).  One lane polls, the workgroup barrier
// releases everybody.  (Letting the 8 relays poll the arrival COUNTER instead - one hop less - was far worse, 2.11 vs 1.58 ms per
// step: reads of a line that is receiving atomics stall the arrivals, however few the readers.)
__device__ __forceinline__ unsigned ld_l2(const unsigned* p) {          // L1-bypassing, L2-served load the compiler cannot hoist
    unsigned v;
    asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
// FEW: the gate counts a dozen arrivals (gate_arrive_few) and the relay polls the counter directly - one hop less than waiting for
// the last arriver to learn that it was last (a returned atomic) and to write the "open" words.
// Every in-launch wait gives up on ONE time base: the 100 MHz wall clock against the context's gate_timeout (cfx_set_gate_timeout_ms) -
// never on an iteration count, whose length in seconds depends on what else loads the fabric.  Only a FAILED poll reads the clock.
struct SpinClock {
    long long t0 = 0;
    __device__ __forceinline__ bool expired(long long timeout) {
        const long long now = wall_clock64();
        if (!t0) { t0 = now; return false; }
        return now - t0 > timeout;
    }
};
// Returns whether the gate opened.  A workgroup whose wait gave up must NOT store: what it would reconstruct from has not arrived; the
// states it owns stay as they were and the context's error word (pinned host memory) says so to the host.
template <bool FEW = false>
__device__ __forceinline__ bool gate_wait(unsigned* gate, unsigned expect, unsigned* err, long long timeout) {
    bool failed = false;
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // HW_REG_XCC_ID[3:0]
        unsigned* open = FEW ? gate : gate + (1 + xcc) * GATE_LINE;   // the counter, or the word the gate's last arriver writes (write-through)
        unsigned* local = gate + (9 + xcc) * GATE_LINE;           // written by this XCD's relay (plain store)
        unsigned* claim = gate + (17 + xcc) * GATE_LINE;
        const bool relay = __hip_atomic_exchange(claim, expect, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != expect;
        SpinClock clk;
        unsigned n = 0;
        if (relay) {
            while (ld_wt(open) != expect) {
                __builtin_amdgcn_s_sleep(1);
                if (clk.expired(timeout)) { failed = true; break; }
            }
            if (!failed) *(volatile unsigned*)local = expect;
        } else {
            while (ld_l2(local) != expect) {
                if (GATE_LOCAL_SLEEP) __builtin_amdgcn_s_sleep(GATE_LOCAL_SLEEP);
                ++n;
                if ((n & (FEW ? 255u : 15u)) == 0) {
                    if (ld_wt(open) == expect) break;
                    if (clk.expired(timeout)) { failed = true; break; }
                }
            }
        }
        if (failed) gate_fail(err);
    }
    return __syncthreads_or(failed ? 1 : 0) == 0;
}
// 8 channel scales of a packet another workgroup of this launch published
__device__ __forceinline__ h16x8 ld8_wt(const u16* p) {
    if ((((uintptr_t)p) & 7) == 0) {                        // uniform
        struct { u64 a, b; } q = {ld_wt((const u64*)p), ld_wt((const u64*)p + 1)};
        return __builtin_bit_cast(h16x8, q);
    }
    u16x8 vb;
#pragma unroll
    for (int i = 0; i < 8; ++i) vb[i] = ld_wt(p + i);
    return __builtin_bit_cast(h16x8, vb);
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
#define TICK_WORDS 64          // u32 ticket words per tensor: [1 + cb] column block cb (CB <= 46), [TICK_ALL] all tiles - on a line of its
#define TICK_ALL 48            //   own: in the stand-alone launch that word is POLLED, and polls of a line stall the atomics arriving on it
#define TICK_MAX_CB 46
#define TICK_RING 256          // ticket blocks (CFX_MAX_BATCH tensors each) a context cycles through, one per launch

#define FUSED_NW 8             // waves per workgroup of the single-launch compress kernel (512 threads: the last arriver of a
#define FUSED_NT (FUSED_NW * 64)   //   column block owns one column per thread, of a tensor one row per thread)
#ifndef FUSED_CH
#define FUSED_CH 18
#endif
//            // partial sums a last-arriver thread keeps in flight per batch (one fabric round trip each batch;
                               //   a last arriver reads fresh cross-CU data at ~65 GB/s, so every redundant load counts)
#define FUSED_RCH 6            // column blocks of a row's partials per batch

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

// The peer-to-peer exchange INSIDE the layer launch (cfx_plan_add_exchange_layer_p2p, one-launch form): workgroup 0 of the launch, once its own
// tile work is done, waits until the launch's packets are complete, publishes this rank's word for the layer (own word + 1, taken on the
// device), waits for the peers' words and opens the launch's external gate - what a one-wave kernel on an exchange stream did before.  No
// second launch, no second stream, no hardware-queue requirement; and a resident polling kernel on another queue - harmless to the 1-bit
// launch - cost the 2-bit layer launch 4 us per layer (tools/xgate_probe.py: gated 2.02 ms per step, the same beside a poller 2.27).
struct P2PInline {
    unsigned* own;                               // NULL: no in-launch exchange
    const unsigned* peer[CFX_P2P_MAX_PEERS];
    int n_peers;
    long long timeout;
};
__device__ __forceinline__ unsigned ld_sys32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// p_count consecutive words at p_gate must all have reached p_expect (1: one counter / "open" word)
__device__ __forceinline__ void p2p_exchange_inline(const unsigned* p_gate, unsigned p_expect, int p_count, const P2PInline& p,
                                                    unsigned* f_gate, unsigned f_expect, unsigned* err) {
    if ((threadIdx.x >> 6) != 0) return;        // one wave
    const int lane = threadIdx.x & 63;
    const long long t0 = wall_clock64();
    unsigned epoch = 0;
    if (lane == 0) epoch = ld_sys32(p.own) + 1u;                 // (read before the wait: only this launch writes it)
    bool gave_up = false;
    for (;;) {
        bool behind = false;
        for (int i = lane; i < p_count; i += 64) behind |= (int)(ld_wt(p_gate + i) - p_expect) < 0;
        if (__builtin_amdgcn_ballot_w64(behind) == 0) break;
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > p.timeout) { gave_up = true; break; }
    }
    if (__builtin_amdgcn_ballot_w64(gave_up) != 0) { if (lane == 0) gate_fail(err); return; }      // own packets incomplete: nothing to announce
    // the packets were stored write-through and drained before they were counted complete: publishing after having SEEN that orders them
    // before the word for anybody who reads the word first
    if (lane == 0) __hip_atomic_store(p.own, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    epoch = (unsigned)__builtin_amdgcn_readfirstlane((int)epoch);
    if (lane < p.n_peers) {
        while ((int)(ld_sys32(p.peer[lane]) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > p.timeout) { gave_up = true; break; }
        }
    }
    // a wait that gave up leaves the gate SHUT: the reconstruction groups started with this wave, give up on the same clock a moment later
    // and store nothing - nobody reconstructs from packets that have not arrived
    if (__builtin_amdgcn_ballot_w64(gave_up) != 0) { if (lane == 0) gate_fail(err); return; }
    if (lane == 0) st_wt(f_gate, f_expect);
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
#define MML_MAX_TILES 2048     // statistics tiles of one launch = flag words per ring and kind
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
// 1:m block top-1 sparsifier on the flat (-1, 1024) view       compress_topk.py:44-105, :128-163
// One lane owns 8 consecutive flat elements.  Half-blocks of m <= 8 live inside a lane; m = 16 spans two lanes.
// ---------------------------------------------------------------------------------------------------
// The 8 elements at flat offset e of one tensor (whole waves call it together: the 16-wide blocks talk to their neighbour lanes).
// WT: the packet goes out write-through - workgroups of the same launch read it (k_topk_layer).
#define TOPK_PUT(ptr, v) do { if (WT) st_wt(ptr, v); else *(ptr) = (v); } while (0)
template <int M, bool WT>
__device__ __forceinline__ void topk_compress_unit(const cfx_comp_item& it, size_t e, size_t E, int flags, h16x8 xv, h16x8 bv) {
    const h16* x = (const h16*)it.x;
    const h16* base = (const h16*)it.base;
    h16* nb = (h16*)it.new_base;
    u16* val = (u16*)it.packet;
    unsigned char* idx = (unsigned char*)(val + E / M);
    const bool upd = (flags & CFX_FLAG_UPDATE_CACHE) && nb;
    const bool ef = !(flags & CFX_FLAG_NO_EF);
    const h16x8 d = xv - bv;                          // (x and base loaded by the caller: several units' loads in flight at once)
    const h16x8 a = habs8(d);
    unsigned keep = 0;   // bit i set = element i survives
    if constexpr (M <= 8) {
        constexpr int HB = 8 / M;   // half-blocks per lane
        unsigned sel[HB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
            int best = 0;
            h16 bestv = a[hb * M];
#pragma unroll
            for (int i = 1; i < M; ++i) {
                const h16 v = a[hb * M + i];
                if (v > bestv) { bestv = v; best = i; }     // strict: first maximum wins (tl.argmax)
            }
            sel[hb] = best;
            keep |= 1u << (hb * M + best);
            TOPK_PUT(&val[e / M + hb], hbits(d[hb * M + best]));
        }
        if constexpr (M == 8) {
            const unsigned other = __shfl_xor(sel[0], 1, 64);
            if ((threadIdx.x & 1) == 0) TOPK_PUT(&idx[e / 16], (unsigned char)((sel[0] << 4) | other));
        } else {
#pragma unroll
            for (int bk = 0; bk < HB / 2; ++bk) TOPK_PUT(&idx[e / (2 * M) + bk], (unsigned char)((sel[2 * bk] << 4) | sel[2 * bk + 1]));
        }
    } else {   // M == 16: half-block = lanes (2k, 2k+1); block = 4 lanes
        int best = 0;
        h16 bestv = a[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) if (a[i] > bestv) { bestv = a[i]; best = i; }
        const int odd = threadIdx.x & 1;
        const unsigned pb = hbits(bestv);
        const unsigned ob = __shfl_xor(pb, 1, 64);
        const int oi = __shfl_xor(best, 1, 64);
        // lower lane wins ties (its elements come first)
        const bool mine = odd ? (hfrom((u16)pb) > hfrom((u16)ob)) : !(hfrom((u16)ob) > hfrom((u16)pb));
        const int selidx = mine ? (best + 8 * odd) : (oi + 8 * (1 - odd));   // index within the 16-wide half-block
        if (mine) { keep |= 1u << best; TOPK_PUT(&val[e / 16], hbits(d[best])); }
        const int other = __shfl_xor(selidx, 2, 64);
        if ((threadIdx.x & 3) == 0) TOPK_PUT(&idx[e / 32], (unsigned char)((selidx << 4) | other));
    }
    if (upd) {
        h16x8 o;
        if (ef) {
            h16x8 recv;
#pragma unroll
            for (int i = 0; i < 8; ++i) recv[i] = ((keep >> i) & 1u) ? d[i] : (h16)0;
            o = base ? (bv + recv) : recv;
        } else o = xv;
        st8nt(nb + e, o);
    }
}

template <int M>
__global__ __launch_bounds__(256) void k_topk_compress(BatchC batch, size_t E, int flags) {
    const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (e >= E) return;   // E % 1024 == 0 and 256*8 = 2048: whole waves exit together
    const cfx_comp_item it = batch.it[blockIdx.y];
    const h16x8 xv = ld8nt((const h16*)it.x + e);
    h16x8 bv = (h16x8)(h16)0;
    if (it.base) bv = ld8nt((const h16*)it.base + e);
    topk_compress_unit<M, false>(it, e, E, flags, xv, bv);
}

// What a receiver adds for the 8 elements at flat offset e, from a packet read with plain loads (MODE 0), with L2-bypassing loads (1: another
// workgroup of this launch wrote it) or with system-scope loads (2: another GPU did).  Every value / index byte the lane needs is loaded once;
// load and use are apart so that a caller can put several units' loads in flight (the compiler keeps atomic loads in program order: a use
// between two of them is a round trip each).
template <int M> struct TopkRecv {
    static constexpr int HB = M <= 8 ? 8 / M : 1;         // half-blocks the lane's 8 elements touch
    static constexpr int NB = HB >= 2 ? HB / 2 : 1;       // index bytes (two half-blocks a byte)
    u16 v[HB];
    unsigned char by[NB];
};
template <int M, int MODE>
__device__ __forceinline__ void topk_recv_load(TopkRecv<M>& r, const u16* val, const unsigned char* idx, size_t e) {
    const size_t hb0 = e / M;
#pragma unroll
    for (int k = 0; k < TopkRecv<M>::HB; ++k) r.v[k] = MODE == 0 ? val[hb0 + k] : (MODE == 1 ? ld_wt(val + hb0 + k) : ld_sys(val + hb0 + k));
#pragma unroll
    for (int k = 0; k < TopkRecv<M>::NB; ++k)
        r.by[k] = MODE == 0 ? idx[(hb0 >> 1) + k] : (MODE == 1 ? ld_wt(idx + (hb0 >> 1) + k) : ld_sys(idx + (hb0 >> 1) + k));
}
template <int M>
__device__ __forceinline__ h16x8 topk_recv_make(const TopkRecv<M>& r, size_t e) {
    const size_t hb0 = e / M;
    h16x8 recv;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = M <= 8 ? i / M : 0;
        const size_t hb = hb0 + k;
        const unsigned b = r.by[TopkRecv<M>::HB >= 2 ? k / 2 : 0];
        const unsigned sel = (hb & 1) ? (b & 15u) : (b >> 4);
        recv[i] = ((unsigned)((e + i) % M) == sel) ? hfrom(r.v[k]) : (h16)0;
    }
    return recv;
}

// ---- the top-k layer in ONE launch (cfx_compress_batch_gated / the exchange-layer ops): group S compresses the own tensors (nothing global
// to wait for: a block's survivor is local) and counts itself on the gate; group D - launched with it - holds the peers' state rows in
// registers until the gate (or the external gate: the packets of the other ranks) opens, then reads values + indices and stores.
#define TKL_SU 4                // units (8 elements a thread) of an S workgroup: 8192 elements, their loads in flight together
#define TKL_DU 8                // ... of a D workgroup: 16384 elements, 128 bytes of state a thread held across the wait
struct TopkLayerArgs {
    size_t E;
    int n_sw, n_st;             // S workgroups per own tensor / in all
    int n_dw;                   // D workgroups per reconstruction item
    int flags;
    unsigned* gate; unsigned gate_expect;
    unsigned* xgate; unsigned xexpect;
    unsigned* err;
    long long timeout;
    int remote;
    P2PInline p2p;
};
template <int M>
__global__ __launch_bounds__(256) void k_topk_layer(BatchC batch, BatchD gated, TopkLayerArgs a) {
    int b = blockIdx.x;
    if (b < a.n_st) {
        const int z = b / a.n_sw, sw = b - z * a.n_sw;
        const cfx_comp_item it = batch.it[z];
        h16x8 xv[TKL_SU], xb[TKL_SU];
#pragma unroll
        for (int u = 0; u < TKL_SU; ++u) {                  // every unit's loads first (clamped offset: unconditional)
            const size_t e = (((size_t)sw * TKL_SU + u) * 256 + threadIdx.x) * 8, ec = e < a.E ? e : 0;
            xv[u] = ld8nt((const h16*)it.x + ec);
            xb[u] = it.base ? ld8nt((const h16*)it.base + ec) : (h16x8)(h16)0;
        }
#pragma unroll
        for (int u = 0; u < TKL_SU; ++u) {
            const size_t e = (((size_t)sw * TKL_SU + u) * 256 + threadIdx.x) * 8;
            if (e < a.E) topk_compress_unit<M, true>(it, e, a.E, a.flags, xv[u], xb[u]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) gate_arrive(a.gate, 1u, a.gate_expect);
        // (packets complete = the word the gate's last arriver writes for XCD 0)
        if (b == 0 && a.p2p.own) p2p_exchange_inline(a.gate + GATE_LINE, a.gate_expect, 1, a.p2p, a.xgate, a.xexpect, a.err);
        return;
    }
    b -= a.n_st;
    const int item = b / a.n_dw, dw = b - item * a.n_dw;
    const cfx_decomp_item it = gated.it[item];
    const h16* base = (const h16*)it.base;
    h16* out = (h16*)it.recon;
    h16x8 bv[TKL_DU];
#pragma unroll
    for (int u = 0; u < TKL_DU; ++u) {
        const size_t e = (((size_t)dw * TKL_DU + u) * 256 + threadIdx.x) * 8;
        bv[u] = (base && e < a.E) ? ld8nt(base + e) : (h16x8)(h16)0;
    }
    if (!(a.xgate ? gate_wait<true>(a.xgate, a.xexpect, a.err, a.timeout) : gate_wait<false>(a.gate, a.gate_expect, a.err, a.timeout))) return;
    const u16* val = (const u16*)it.packet;
    const unsigned char* idx = (const unsigned char*)(val + a.E / M);
    // the packet words of G units in flight at once (as many as 16 small registers hold), then their stores
    constexpr int G = (TopkRecv<M>::HB + TopkRecv<M>::NB) <= 2 ? TKL_DU : ((TopkRecv<M>::HB + TopkRecv<M>::NB) <= 4 ? 4 : ((TopkRecv<M>::HB + TopkRecv<M>::NB) <= 8 ? 2 : 1));
#pragma unroll
    for (int u0 = 0; u0 < TKL_DU; u0 += G) {
        TopkRecv<M> rr[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const size_t e = (((size_t)dw * TKL_DU + u0 + g) * 256 + threadIdx.x) * 8, ec = e < a.E ? e : 0;
            if (a.remote) topk_recv_load<M, 2>(rr[g], val, idx, ec);
            else topk_recv_load<M, 1>(rr[g], val, idx, ec);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const size_t e = (((size_t)dw * TKL_DU + u0 + g) * 256 + threadIdx.x) * 8;
            if (e < a.E) {
                const h16x8 rv = topk_recv_make<M>(rr[g], e);
                st8nt(out + e, base ? (bv[u0 + g] + rv) : rv);
            }
        }
    }
}

template <int M>
__global__ __launch_bounds__(256) void k_topk_decompress(BatchD batch, size_t E, unsigned* pre, unsigned pre_val) {
    // lane: publish `pre` first - the launch in front of this one in the stream (the previous peer's reconstruction) has finished
    if (pre && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0) st_wt(pre, pre_val);
    const cfx_decomp_item it = batch.it[blockIdx.y];
    const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (e >= E) return;
    const u16* val = (const u16*)it.packet;
    const unsigned char* idx = (const unsigned char*)(val + E / M);
    const h16* base = (const h16*)it.base;
    h16* out = (h16*)it.recon;
    TopkRecv<M> rr;
    topk_recv_load<M, 0>(rr, val, idx, e);
    const h16x8 recv = topk_recv_make<M>(rr, e);
    h16x8 bv = (h16x8)(h16)0;
    if (base) bv = ld8nt(base + e);
    st8nt(out + e, base ? (bv + recv) : recv);
}

// ---------------------------------------------------------------------------------------------------
// second-order residual (residual = 2): the predictor arithmetic around the codec       main.py:244-266, 378-384
//   k_residual2_delta :  dd = (x - base) - delta_base                       (what gets compressed)
//   k_residual2_update:  new_base = (base + delta_base) + recv ; new_delta_base = fp16(fp32(fp16(delta_base + recv)) * decay)
// one fp16 rounding per reference operation; in-place allowed (new_base == base, new_delta_base == delta_base)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_residual2_delta(const h16* __restrict__ x, const h16* base, const h16* dbase, h16* dd, size_t n8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const h16x8 d = ld8nt(x + i * 8) - ld8(base + i * 8);
    st8(dd + i * 8, d - ld8(dbase + i * 8));
}
__global__ __launch_bounds__(256) void k_residual2_update(const h16* base, const h16* dbase, const h16* __restrict__ recv, h16* nb, h16* ndb,
                                                          float decay, size_t n8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const h16x8 b = ld8(base + i * 8), d = ld8(dbase + i * 8), r = ld8nt(recv + i * 8);
    const h16x8 pred = b + d;
    const h16x8 s = d + r;
    h16x8 nd;
#pragma unroll
    for (int k = 0; k < 8; ++k) nd[k] = (h16)((float)s[k] * decay);
    st8(nb + i * 8, pred + r);
    st8(ndb + i * 8, nd);
}

// ---------------------------------------------------------------------------------------------------
// Ring-attention block merge (the consumer of the reconstructed K,V; reference ring.py:263 update_out_and_lse, taken there
// from the un-vendored yunchang package; published formula):
//     out <- out - sigmoid(lse_b - lse) * (out - out_b) ;  lse <- lse - logsigmoid(lse - lse_b)
// One launch instead of ~10 eager elementwise kernels per block.  out fp32 [B][S][H][D], lse fp32 [B][S][H];
// block_out fp16 [B][H][S][D] or [B][S][H][D] (strides) and block_lse fp32 [B][H][S] as the fused SDPA kernel leaves them.
// first != 0: out = block_out, lse = block_lse.  One thread = 8 consecutive d of one (b, s, h); the D/8 threads of a row all
// read lse[row] and one of them rewrites it in place, so a row's threads must sit in ONE wave (its load instruction then
// precedes its store instruction for every lane): a row takes G = the power of two >= D/8 lanes (G <= 64), lanes d8 >= D/8 idle.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void attn_merge_body(float* __restrict__ out, float* __restrict__ lse, const h16* __restrict__ bo,
                                                const float* __restrict__ bl, int B, int S, int H, int D, int first,
                                                size_t bo_sb, size_t bo_ss, size_t bo_sh, int lg) {
    const int D8 = D >> 3;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t rows = (size_t)B * S * H;
    const int d8 = (int)(i & ((1u << lg) - 1));
    size_t r = i >> lg;
    if (r >= rows || d8 >= D8) return;
    const int h = (int)(r % H); r /= H;
    const int s_ = (int)(r % S);
    const int b = (int)(r / S);
    const size_t o_idx = (((size_t)b * S + s_) * H + h) * D + (size_t)d8 * 8;
    const size_t l_idx = ((size_t)b * S + s_) * H + h;
    const size_t bo_idx = (size_t)b * bo_sb + (size_t)s_ * bo_ss + (size_t)h * bo_sh + (size_t)d8 * 8;
    const float lb = bl[((size_t)b * H + h) * S + s_];
    const h16x8 ob = ld8nt(bo + bo_idx);
    float4* op = reinterpret_cast<float4*>(out + o_idx);
    if (first) {
        op[0] = make_float4((float)ob[0], (float)ob[1], (float)ob[2], (float)ob[3]);
        op[1] = make_float4((float)ob[4], (float)ob[5], (float)ob[6], (float)ob[7]);
        if (d8 == 0) lse[l_idx] = lb;
        return;
    }
    // every lane of the row's group has its lse before lane d8 == 0 of the same wave stores the new one (program order of a wave;
    // the asm statement keeps the compiler from sinking the load below the store)
    const float l = __hip_atomic_load(lse + l_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float x = lb - l;
    const float sg = 1.0f / (1.0f + __expf(-x));                                    // sigmoid(lse_b - lse)
    float4 a = op[0], c = op[1];
    a.x -= sg * (a.x - (float)ob[0]); a.y -= sg * (a.y - (float)ob[1]); a.z -= sg * (a.z - (float)ob[2]); a.w -= sg * (a.w - (float)ob[3]);
    c.x -= sg * (c.x - (float)ob[4]); c.y -= sg * (c.y - (float)ob[5]); c.z -= sg * (c.z - (float)ob[6]); c.w -= sg * (c.w - (float)ob[7]);
    op[0] = a; op[1] = c;
    // logsigmoid(-x) = -softplus(x) = -(max(x,0) + log1p(exp(-|x|)));  lse - logsigmoid(lse - lse_b) = lse + softplus(x)
    if (d8 == 0) lse[l_idx] = l + (fmaxf(x, 0.0f) + log1pf(__expf(-fabsf(x))));
}

// wflag != NULL: the launch ALSO waits (one lane of workgroup 0, after its own merge work) until *wflag has reached wval - the
// exchange lane's "peer r reconstructed" flag (cfx_plan_run_lane) - so that the next attention block, which follows this launch
// in the compute stream, finds the peer's K,V complete without any cross-stream event.
__global__ __launch_bounds__(256) void k_attn_merge(float* __restrict__ out, float* __restrict__ lse, const h16* __restrict__ bo,
                                                    const float* __restrict__ bl, int B, int S, int H, int D, int first,
                                                    size_t bo_sb, size_t bo_ss, size_t bo_sh, int lg,
                                                    const unsigned* wflag, unsigned wval, unsigned* err, long long timeout) {
    attn_merge_body(out, lse, bo, bl, B, S, H, D, first, bo_sb, bo_ss, bo_sh, lg);
    if (wflag && blockIdx.x == 0 && threadIdx.x == 0) flag_spin(wflag, wval, err, timeout);
}

// The plain copy the roofline's `achievable` is measured with (bench.py) and the PMC counters are calibrated on (tools/pmc_summary.py): 16 bytes
// a lane, four loads in flight per thread before the first store, non-temporal both ways (a one-touch stream).
__global__ __launch_bounds__(256) void k_copy_probe(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_* s4 = reinterpret_cast<const u32x4_*>(src);
    u32x4_* d4 = reinterpret_cast<u32x4_*>(dst);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4_ a0 = __builtin_nontemporal_load(s4 + i), a1 = __builtin_nontemporal_load(s4 + i + stride);
        const u32x4_ a2 = __builtin_nontemporal_load(s4 + i + 2 * stride), a3 = __builtin_nontemporal_load(s4 + i + 3 * stride);
        __builtin_nontemporal_store(a0, d4 + i);
        __builtin_nontemporal_store(a1, d4 + i + stride);
        __builtin_nontemporal_store(a2, d4 + i + 2 * stride);
        __builtin_nontemporal_store(a3, d4 + i + 3 * stride);
    }
    for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(s4 + i), d4 + i);
}

// ---------------------------------------------------------------------------------------------------
// host side: C-ABI
// ---------------------------------------------------------------------------------------------------
static bool shape_ok(int codec, int N, int C, int param) {
    if (N <= 0 || C <= 0 || (C % 8) != 0) return false;
    switch (codec) {
        case CFX_CODEC_BINARY: return ((size_t)N * (C / 8)) % 2 == 0;
        case CFX_CODEC_INT2: return true;
        case CFX_CODEC_INT4: return N % 2 == 0;
        case CFX_CODEC_INT8: return true;
        case CFX_CODEC_TOPK:
            return ((size_t)N * C) % 1024 == 0 && (param == 1 || param == 2 || param == 4 || param == 8 || param == 16);
        default: return false;
    }
}

static int auto_rows(const cfx_ctx* ctx, int N, int C, int batch, bool stats) {
    if (ctx && ctx->rows_per_tile > 0) {
        int r = ctx->rows_per_tile;
        if (stats && r < 16) r = 16;
        return (r + 1) & ~1;
    }
    // Measured on MI355X (tools/kbench.hip, tools/microbench.py): short tiles win - one or two wave steps per
    // workgroup, thousands of workgroups - because these launches last 5-20 us and ramp/tail dominate long tiles.
    if (!stats) return WAVES * UNROLL;
    // statistics pass: every 16 rows of tile height cost one more partial per column for the finalize kernel to reduce,
    // so tall tensors take taller tiles as long as >= 768 workgroups remain (S4 (4448,3072): R = 64, P = 70 instead of 278)
    const int CB = (C + TILE_C - 1) / TILE_C;
    const int cands[3] = {128, 64, 32};
    for (int i = 0; i < 3; ++i)
        if ((long)CB * ((N + cands[i] - 1) / cands[i]) * batch >= 768) return cands[i];
    return WAVES * UNROLL_S;
}

extern "C" {

int cfx_abi_version(void) { return CFX_ABI_VERSION; }

cfx_ctx* cfx_create(int device) {
    cfx_ctx* c = new cfx_ctx();
    c->device = device;
    c->rows_per_tile = 0;
    c->prof = nullptr;
    c->prof_cap = c->prof_n = 0;
    c->prof_stride = 1;
    memset(c->prof_seen, 0, sizeof(c->prof_seen));
    c->prof_mask = 0;
    c->tick = nullptr;
    memset(c->tick_next, 0, sizeof(c->tick_next));
    c->n_ring_streams = 0;
    c->n_cu_cache = 0;
    c->cu_cache_next = 0;
    c->ring_clock = 0;
    memset(c->ring_used, 0, sizeof(c->ring_used));
    c->dev_buf = nullptr;
    c->gate = nullptr;
    c->gate_err = nullptr;
    c->gate_timeout = 500000000LL;     // 5 s of the 100 MHz wall clock
    c->fused = 1;
    c->stats_rows = 0;
    c->gated_on = 1;
    c->lr_chain = c->lr_decode = 0;
    c->dev_probe = 0;
    c->allow_shared_queues = 0;
    c->ipc_kind = 0;
    c->ipc_want = 2;
    c->err[0] = 0;
    return c;
}

// Ticket blocks of the in-launch finalize: device memory owned by the context, zeroed ONCE here; every ticket word is
// reset by the workgroup that draws its final value, so a block is clean again when its launch retires.
int cfx_prepare(cfx_ctx* ctx) {
    if (!ctx) return CFX_ERR_NULL;
    if (ctx->tick) return CFX_OK;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != ctx->device && hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, CFX_ERR_LAUNCH, "prepare: hipSetDevice failed");
    static_assert(3 * TICK_RING * CFX_RING_STREAMS == sizeof(((cfx_ctx*)0)->gate_expect) / sizeof(unsigned), "gate_expect has three entries per ring slot");
    const size_t tick_words = (size_t)CFX_RING_STREAMS * TICK_RING * CFX_MAX_BATCH * TICK_WORDS;
    const size_t gate_words = (size_t)(CFX_RING_STREAMS * TICK_RING + 1) * GATE_STRIDE;
    const size_t colgate_words = (size_t)CFX_RING_STREAMS * MML_MAX_TILES;     // tile flags of the min/max layer launch ("codes published"), per ring
    const size_t bytes = (tick_words + gate_words + colgate_words) * sizeof(unsigned);
    void* p = nullptr;
    int rc = CFX_OK;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        if (p) (void)hipFree(p);
        rc = fail(ctx, CFX_ERR_LAUNCH, "prepare: cannot allocate the ticket blocks");
    } else {
        ctx->tick = (unsigned*)p;
        ctx->gate = ctx->tick + tick_words;
        ctx->colgate = ctx->gate + gate_words;
        ctx->mml_seq = 0;
        // the error word: pinned, device-visible HOST memory - a timed-out wait is reported by the next native call, no device sync
        void* e = nullptr;
        if (hipHostMalloc(&e, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); ctx->tick = nullptr; rc = fail(ctx, CFX_ERR_LAUNCH, "prepare: cannot allocate the error word"); }
        else { memset(e, 0, 64); ctx->gate_err = (unsigned*)e; }
        memset(ctx->gate_expect, 0, sizeof(ctx->gate_expect));
    }
    if (cur >= 0 && cur != ctx->device) (void)hipSetDevice(cur);
    return rc;
}

#ifdef CFX_DEV_PROBES      // ---- the developer library only (include/cfx_dev.h) ----
int cfx_dev_stamps(cfx_ctx* ctx, void* buf) {
    if (!ctx) return CFX_ERR_NULL;
    ctx->dev_buf = buf;
    return CFX_OK;
}

int cfx_dev_set_launch_tags(cfx_ctx* ctx, unsigned abs_seq, unsigned mml_seq) {
    if (!ctx) return CFX_ERR_NULL;
    if (!ctx->tick && cfx_prepare(ctx) != CFX_OK) return CFX_ERR_LAUNCH;
    (void)hipDeviceSynchronize();                     // (nothing in flight carries the old numbers)
    ctx->abs_seq = abs_seq;
    ctx->mml_seq = mml_seq;
    return CFX_OK;
}

int cfx_dev_set_probe(cfx_ctx* ctx, int mode) {
    if (!ctx) return CFX_ERR_NULL;
    if (mode < 0 || mode > 4) return fail(ctx, CFX_ERR_BATCH, "dev probe must be 0..4");
    ctx->dev_probe = mode;
    return CFX_OK;
}
#endif

int cfx_set_fused_finalize(cfx_ctx* ctx, int on) {
    if (!ctx) return CFX_ERR_NULL;
    ctx->fused = on != 0;
    return CFX_OK;
}

int cfx_set_stats_rows(cfx_ctx* ctx, int rows) {
    if (!ctx) return CFX_ERR_NULL;
    if (rows < 0 || rows > 4096) return fail(ctx, CFX_ERR_BATCH, "stats rows must be 0 (automatic) .. 4096");
    ctx->stats_rows = rows;
    return CFX_OK;
}

int cfx_set_gated_launch(cfx_ctx* ctx, int on) {
    if (!ctx) return CFX_ERR_NULL;
    ctx->gated_on = on != 0;
    return CFX_OK;
}

int cfx_set_lr_chain(cfx_ctx* ctx, int chain) {
    if (!ctx) return CFX_ERR_NULL;
    if (chain < 0 || chain > 2) return fail(ctx, CFX_ERR_BATCH, "lr chain must be 0 (automatic), 1 (no single launch) or 2 (C-space chain)");
    ctx->lr_chain = chain;
    return CFX_OK;
}

int cfx_set_lr_decode(cfx_ctx* ctx, int mode) {
    if (!ctx) return CFX_ERR_NULL;
    if (mode < 0 || mode > 2) return fail(ctx, CFX_ERR_BATCH, "lr decode must be 0 (automatic), 1 (VALU) or 2 (MFMA)");
    ctx->lr_decode = mode;
    return CFX_OK;
}

int cfx_set_allow_shared_queues(cfx_ctx* ctx, int on) {
    if (!ctx) return CFX_ERR_NULL;
    ctx->allow_shared_queues = on != 0;
    return CFX_OK;
}

// The one environment variable the library looks at - and it is the HIP runtime's, not ours: see cfx.h.
int cfx_hw_queues_ok(void) {
    static int ok = -1;
    if (ok < 0) {
        const char* v = getenv("GPU_MAX_HW_QUEUES");
        ok = (v && atoi(v) >= 2) ? 1 : 0;
    }
    return ok;
}

static void prof_free(cfx_ctx* ctx) {
    for (int i = 0; i < ctx->prof_cap; ++i) { (void)hipEventDestroy(ctx->prof[i].a); (void)hipEventDestroy(ctx->prof[i].b); }
    delete[] ctx->prof;
    ctx->prof = nullptr;
    ctx->prof_cap = ctx->prof_n = 0;
}

void cfx_destroy(cfx_ctx* ctx) {
    if (!ctx) return;
    prof_free(ctx);
    if (ctx->tick) (void)hipFree(ctx->tick);
    for (int i = 0; i < ctx->lrs_n; ++i)
        if (ctx->lrs_arena[i]) (void)hipFree(ctx->lrs_arena[i]);
    for (int i = 0; i < CFX_RING_STREAMS; ++i)
        if (ctx->mml_arena[i]) (void)hipFree(ctx->mml_arena[i]);
    for (int i = 0; i < CFX_RING_STREAMS; ++i)
        if (ctx->abs_arena[i]) (void)hipFree(ctx->abs_arena[i]);
    if (ctx->lrs_ev) (void)hipEventDestroy(ctx->lrs_ev);
    if (ctx->gate_err) (void)hipHostFree(ctx->gate_err);
    delete ctx;
}

int cfx_profile_enable(cfx_ctx* ctx, int capacity, unsigned kernel_mask, int stride) {
    if (!ctx) return CFX_ERR_NULL;
    ctx->prof_stride = stride > 0 ? stride : 1;
    memset(ctx->prof_seen, 0, sizeof(ctx->prof_seen));
    if (capacity > ctx->prof_cap) {
        prof_free(ctx);
        ctx->prof = new ProfRec[capacity];
        for (int i = 0; i < capacity; ++i) {
            if (hipEventCreate(&ctx->prof[i].a) != hipSuccess || hipEventCreate(&ctx->prof[i].b) != hipSuccess)
                return fail(ctx, CFX_ERR_LAUNCH, "profile: hipEventCreate failed");
        }
        ctx->prof_cap = capacity;
    }
    ctx->prof_n = 0;
    ctx->prof_mask = capacity > 0 ? kernel_mask : 0;
    return CFX_OK;
}

int cfx_profile_read(cfx_ctx* ctx, int* kernel_ids, float* ms, int cap) {
    if (!ctx || !kernel_ids || !ms) return CFX_ERR_NULL;
    const int n = ctx->prof_n < cap ? ctx->prof_n : cap;
    for (int i = 0; i < n; ++i) {
        (void)hipEventSynchronize(ctx->prof[i].b);
        float t = 0.f;
        if (hipEventElapsedTime(&t, ctx->prof[i].a, ctx->prof[i].b) != hipSuccess) t = -1.f;
        kernel_ids[i] = ctx->prof[i].kid;
        ms[i] = t;
    }
    ctx->prof_n = 0;
    return n;
}

const char* cfx_kernel_name(int kernel_id) { return (kernel_id > 0 && kernel_id < KID_MAX) ? kid_names[kernel_id] : ""; }

const char* cfx_last_error_string(cfx_ctx* ctx) { return ctx ? ctx->err : "null ctx"; }

int cfx_set_rows_per_tile(cfx_ctx* ctx, int rows) {
    if (!ctx) return CFX_ERR_NULL;
    ctx->rows_per_tile = rows < 0 ? 0 : rows;
    return CFX_OK;
}

size_t cfx_packet_bytes(int codec, int N, int C, int param) {
    if (!shape_ok(codec, N, C, param)) return 0;
    const size_t n = N, c = C;
    switch (codec) {
        case CFX_CODEC_BINARY: return n * c / 8 + 2 * (n + c);
        case CFX_CODEC_INT2: return n * c / 4 + 2 * (n + c);
        case CFX_CODEC_INT4: return n * c / 2 + 4 * c;
        case CFX_CODEC_INT8: return n * c + 4 * c;
        case CFX_CODEC_TOPK: return 2 * (n * c / param) + n * c / (2 * param);
    }
    return 0;
}

// per-tensor workspace in u64 words (worst case R = 16)
static size_t ws_words(int codec, int N, int C) {
    const size_t CB = (C + TILE_C - 1) / TILE_C, P = (N + 15) / 16;
    switch (codec) {
        case CFX_CODEC_BINARY:
        case CFX_CODEC_INT2: return (size_t)N * CB + P * C + ((size_t)N * CB + P * C + 1) / 2;      // + the 32-bit partials of the fused path
        case CFX_CODEC_INT4:
        case CFX_CODEC_INT8: return (P * C + 1) / 2;
        default: return 0;
    }
}

size_t cfx_workspace_bytes(int codec, int N, int C, int param, int batch) {
    if (!shape_ok(codec, N, C, param) || batch < 1 || batch > CFX_MAX_BATCH) return 0;
    return ws_words(codec, N, C) * 8 * batch;
}


// CUs the queue of `stream` may use (hipExtStreamCreateWithCUMask; an ordinary stream has them all).  Cached per stream handle:
// tile shapes are chosen for the CUs a launch will actually get (an exchange lane has 32, not 256).
static int stream_cu_count(cfx_ctx* ctx, void* stream) {
    for (int i = 0; i < ctx->n_cu_cache; ++i)
        if (ctx->cu_cache_stream[i] == stream) return ctx->cu_cache_n[i];
    int total = 0;
    (void)hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, ctx->device);
    int cus = 0;
    uint32_t m[16] = {0};
    if (stream && hipExtStreamGetCUMask((hipStream_t)stream, 16, m) == hipSuccess)
        for (int i = 0; i < 16; ++i) cus += __builtin_popcount(m[i]);
    else (void)hipGetLastError();
    const int n = (cus > 0 && cus < total) ? cus : total;
    const int slot = ctx->n_cu_cache < 8 ? ctx->n_cu_cache++ : (int)(ctx->cu_cache_next++ % 8);
    ctx->cu_cache_stream[slot] = stream;
    ctx->cu_cache_n[slot] = n;
    return n;
}

// pre != NULL: the launch first publishes pre_val at *pre (exchange lane: "the reconstruction in front of this one is complete")
static int decompress_impl(cfx_ctx* ctx, int codec, int N, int C, int param, int batch, const cfx_decomp_item* items, void* stream,
                           unsigned* pre, unsigned pre_val) {
    if (!ctx || !items) return fail(ctx, CFX_ERR_NULL, "decompress: null ctx/items");
    if (batch < 1 || batch > CFX_MAX_BATCH) return fail(ctx, CFX_ERR_BATCH, "decompress: batch out of range");
    if (!shape_ok(codec, N, C, param)) return fail(ctx, codec >= 1 && codec <= 5 ? CFX_ERR_SHAPE : CFX_ERR_CODEC, "decompress: bad codec/shape");
    BatchD b;
    memset(&b, 0, sizeof(b));
    for (int i = 0; i < batch; ++i) {
        if (!items[i].packet || !items[i].recon) return fail(ctx, CFX_ERR_NULL, "decompress: null packet/recon");
        if (!AL16(items[i].packet) || !AL16(items[i].recon) || !AL16(items[i].base)) return fail(ctx, CFX_ERR_ALIGN, "decompress: pointers must be 16-byte aligned");
        b.it[i] = items[i];
    }
    hipStream_t s = (hipStream_t)stream;
    const int R = auto_rows(ctx, N, C, batch, false);
    const dim3 grid((C + TILE_C - 1) / TILE_C, (N + R - 1) / R, batch);
    switch (codec) {
        case CFX_CODEC_BINARY:
            if (stream_cu_count(ctx, stream) < 128) {
                // a CU-masked lane is bound by the bytes each CU keeps in flight: 4 rows per wave instead of 2
                const int R4 = WAVES * 4;
                LAUNCH(ctx, KID_BINARY_DEQUANT, s, k_binary_dequant<4>, dim3(grid.x, (N + R4 - 1) / R4, batch), dim3(NTHR), 0, s, b, N, C, R4, pre, pre_val);
            } else LAUNCH(ctx, KID_BINARY_DEQUANT, s, k_binary_dequant<UNROLL>, grid, dim3(NTHR), 0, s, b, N, C, R, pre, pre_val);
            break;
        case CFX_CODEC_INT2: LAUNCH(ctx, KID_INT2_DEQUANT, s, k_int2_dequant, grid, dim3(NTHR), 0, s, b, N, C, R, pre, pre_val); break;
        case CFX_CODEC_INT4: LAUNCH(ctx, KID_INT4_DEQUANT, s, k_int4_dequant, grid, dim3(NTHR), 0, s, b, N, C, R, pre, pre_val); break;
        case CFX_CODEC_INT8: LAUNCH(ctx, KID_INT8_DEQUANT, s, k_int8_dequant, grid, dim3(NTHR), 0, s, b, N, C, R, pre, pre_val); break;
        case CFX_CODEC_TOPK: {
            const size_t E = (size_t)N * C;
            const dim3 g((unsigned)((E / 8 + 255) / 256), batch);
            switch (param) {
                case 1: LAUNCH(ctx, KID_TOPK_DECOMPRESS, s, k_topk_decompress<1>, g, dim3(256), 0, s, b, E, pre, pre_val); break;
                case 2: LAUNCH(ctx, KID_TOPK_DECOMPRESS, s, k_topk_decompress<2>, g, dim3(256), 0, s, b, E, pre, pre_val); break;
                case 4: LAUNCH(ctx, KID_TOPK_DECOMPRESS, s, k_topk_decompress<4>, g, dim3(256), 0, s, b, E, pre, pre_val); break;
                case 8: LAUNCH(ctx, KID_TOPK_DECOMPRESS, s, k_topk_decompress<8>, g, dim3(256), 0, s, b, E, pre, pre_val); break;
                default: LAUNCH(ctx, KID_TOPK_DECOMPRESS, s, k_topk_decompress<16>, g, dim3(256), 0, s, b, E, pre, pre_val); break;
            }
        } break;
    }
    return check_launch(ctx, "decompress launch");
}

int cfx_decompress_batch(cfx_ctx* ctx, int codec, int N, int C, int param, int batch, const cfx_decomp_item* items, void* stream) {
    return decompress_impl(ctx, codec, N, C, param, batch, items, stream, nullptr, 0u);
}

// statistics tile height of the fused compress launch
static int fused_rows(const cfx_ctx* ctx, int N, int C, int batch, int cus) {
    if (ctx->stats_rows > 0) return (ctx->stats_rows + 15) & ~15;
    // 8 waves x 4 rows in flight = 32 rows per wave step; taller tiles (fewer partials per column for the last arriver to
    // reduce) as long as >= 768 workgroups remain, as in auto_rows
    const int CB = (C + TILE_C - 1) / TILE_C;
    const int cands[2] = {128, 64};
    for (int i = 0; i < 2; ++i)
        if ((long)CB * ((N + cands[i] - 1) / cands[i]) * batch >= 768) return cands[i];
    if (cus < 128) {
        // a CU-masked lane: as many tiles as fit the lane in ONE round (3 workgroups of this kernel per CU), each a few trips of the
        // row loop - measured on 32 CUs, K,V of the FLUX shard: 204 tiles of 32 rows = 3 rounds of latency-bound workgroups 25.7 us
        for (int R = FUSED_NW * UNROLL_S; R <= 512; R += FUSED_NW * UNROLL_S)
            if ((long)CB * ((N + R - 1) / R) * batch <= 3L * cus) return R;
    }
    return FUSED_NW * UNROLL_S;
}

// The gated form can run as one launch when: 1-bit codec, in-launch finalize on, rows of sign bits 16-byte aligned (C % 128 == 0),
// tiles of at most FUSED_NW * GATE_KR (1-bit) / FUSED_NW * (GATE_KR2 + GATE_KL) (2-bit) rows cover the tensor with few enough workgroups to matter.  Otherwise the same work runs as
// compress + one reconstruction launch (identical results).
static bool gated_one_launch(cfx_ctx* ctx, int codec, int C, int CB) {
    return (codec == CFX_CODEC_BINARY || codec == CFX_CODEC_INT2) && ctx->fused && CB <= TICK_MAX_CB && C % 128 == 0 && !ctx->dev_probe && ctx->gated_on;
}

// Ticket / gate blocks are handed out round-robin from a ring PER STREAM (launches of one stream are in order, so a ring slot is never
// shared by two launches in flight; one ring for every stream would let a stalled stream's launch meet a slot that another stream has
// cycled back to).  Needs cfx_prepare.
static unsigned ticket_slot(cfx_ctx* ctx, void* stream) {
    unsigned slot;
    int ring = -1;
    for (int i = 0; i < ctx->n_ring_streams; ++i)
        if (ctx->ring_stream[i] == stream) { ring = i; break; }
    if (ring < 0) {
        if (ctx->n_ring_streams < CFX_RING_STREAMS) ring = ctx->n_ring_streams++;
        else {
            // every ring is taken: the least recently used one changes hands.  The new owner continues at the ring's next slot, a
            // full turn (256 launches) away from whatever its previous owner may still have in flight
            ring = 0;
            for (int i = 1; i < CFX_RING_STREAMS; ++i)
                if (ctx->ring_used[i] < ctx->ring_used[ring]) ring = i;
        }
        ctx->ring_stream[ring] = stream;
    }
    ctx->ring_used[ring] = ++ctx->ring_clock;
    slot = (unsigned)ring * TICK_RING + (ctx->tick_next[ring]++ % TICK_RING);
    return slot;
}

// `xg` (exchange-layer op): the gated items' packets are NOT produced by this call but delivered by somebody else (a collective) once
// this call's packets are complete.  If the one-launch form is possible, the gated group waits on an external gate word and *xg
// says what to wait for (packets complete: counter p_gate has reached p_expect) and what to set afterwards (f_gate = f_expect);
// otherwise only the compress part is launched, xg->taken stays false and the caller reconstructs after its collective.
// peer-to-peer exchange layer: the launch itself publishes / awaits the flag words (P2PInline) - the caller launches nothing else
static void fill_p2p(cfx_ctx* ctx, CfxXGate* xg, P2PInline& p) {
    memset(&p, 0, sizeof(p));
    if (!xg->p2p_own) return;
    p.own = xg->p2p_own;
    p.n_peers = xg->p2p_n;
    for (int i = 0; i < xg->p2p_n; ++i) p.peer[i] = xg->p2p_peer[i];
    p.timeout = ctx->gate_timeout;
    xg->inline_done = 1;
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

static int compress_impl(cfx_ctx* ctx, int codec, int N, int C, int param, int flags, int batch, const cfx_comp_item* items,
                         int n_ride, const cfx_decomp_item* ride, int n_gated, const cfx_decomp_item* gated,
                         void* workspace, size_t workspace_bytes, void* stream, CfxXGate* xg = nullptr) {
    if (xg) { xg->taken = 0; xg->inline_done = 0; xg->p_gate = xg->f_gate = nullptr; xg->p_expect = xg->f_expect = 0; xg->p_count = 1; }
    if (!ctx || !items) return fail(ctx, CFX_ERR_NULL, "compress: null ctx/items");
    if (n_gated < 0 || n_gated > CFX_MAX_BATCH || (n_gated && !gated)) return fail(ctx, CFX_ERR_BATCH, "compress: gated batch out of range");
    if (batch < 1 || batch > CFX_MAX_BATCH) return fail(ctx, CFX_ERR_BATCH, "compress: batch out of range");
    if (!shape_ok(codec, N, C, param)) return fail(ctx, codec >= 1 && codec <= 5 ? CFX_ERR_SHAPE : CFX_ERR_CODEC, "compress: bad codec/shape");
    if (n_ride < 0 || n_ride > CFX_MAX_BATCH || (n_ride && !ride)) return fail(ctx, CFX_ERR_BATCH, "compress: ride-along batch out of range");
    if (n_ride && codec != CFX_CODEC_BINARY) return fail(ctx, CFX_ERR_CODEC, "compress: ride-along reconstruction items need the 1-bit codec");
    const bool upd = flags & CFX_FLAG_UPDATE_CACHE;
    BatchC b;
    memset(&b, 0, sizeof(b));
    for (int i = 0; i < batch; ++i) {
        if (!items[i].x || !items[i].packet) return fail(ctx, CFX_ERR_NULL, "compress: null x/packet");
        if (upd && !items[i].new_base) return fail(ctx, CFX_ERR_NULL, "compress: UPDATE_CACHE needs new_base");
        if (!AL16(items[i].x) || !AL16(items[i].base) || !AL16(items[i].new_base) || !AL16(items[i].packet))
            return fail(ctx, CFX_ERR_ALIGN, "compress: pointers must be 16-byte aligned");
        b.it[i] = items[i];
    }
    BatchD rd;
    memset(&rd, 0, sizeof(rd));
    for (int i = 0; i < n_ride; ++i) {
        if (!ride[i].packet || !ride[i].recon) return fail(ctx, CFX_ERR_NULL, "compress: null ride-along packet/recon");
        if (!AL16(ride[i].packet) || !AL16(ride[i].recon) || !AL16(ride[i].base)) return fail(ctx, CFX_ERR_ALIGN, "compress: pointers must be 16-byte aligned");
        rd.it[i] = ride[i];
    }
    BatchD gd;
    memset(&gd, 0, sizeof(gd));
    for (int i = 0; i < n_gated; ++i) {
        if (!gated[i].packet || !gated[i].recon) return fail(ctx, CFX_ERR_NULL, "compress: null gated packet/recon");
        if (!AL16(gated[i].packet) || !AL16(gated[i].recon) || !AL16(gated[i].base)) return fail(ctx, CFX_ERR_ALIGN, "compress: pointers must be 16-byte aligned");
        gd.it[i] = gated[i];
    }
    const size_t need = cfx_workspace_bytes(codec, N, C, param, batch);
    if (need && (!workspace || workspace_bytes < need)) return fail(ctx, CFX_ERR_WORKSPACE, "compress: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t wstride = ws_words(codec, N, C);
    u64* ws = (u64*)workspace;
    const int CB = (C + TILE_C - 1) / TILE_C;
    // Under stream capture NO layer form is taken (round 6).  The one-launch forms take the value their gates open at, their ticket-ring
    // slot and their launch tags as launch ARGUMENTS that the host advances with every launch - a replayed graph node would wait for
    // numbers that have gone by - so a capturing stream gets the capturable sequence instead, from this very call: compress (tickets that
    // reset themselves) ; reconstruct the gated items in stream order (an exchange-layer op: the caller's exchange in between, xg->taken
    // stays 0).  Bit-identical results; two (int4 / int8: three) launches per layer instead of one, and no host call per replay.
    // (Device-side counters would keep the one-launch form capturable: every workgroup of a launch has to read the launch's number and
    // exactly one has to advance it once ALL have read it - the last workgroup to leave, an exit ticket per workgroup - and the gate
    // blocks have to be reset by it as well: ~0.5 us on every launch of the headline path for a mode the plan replay does not need.)
    bool capturing = false;
    if (n_gated || xg) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        capturing = cs != hipStreamCaptureStatusNone;
    }

    if (codec == CFX_CODEC_TOPK) {
        const size_t E = (size_t)N * C;
        // ---- the layer in ONE launch (k_topk_layer): the reconstruction group launched with the compress group, gated on the packets ----
        const int stream_cus_t = n_gated ? stream_cu_count(ctx, stream) : 0;
        bool layer = n_gated && ctx->gated_on && !ctx->dev_probe && stream_cus_t >= 128 && !capturing;
        if (layer && !xg) {
            // loop-back: every reconstruction item reads one of this launch's packets
            for (int g_ = 0; g_ < n_gated && layer; ++g_) {
                bool mine = false;
                for (int i = 0; i < batch; ++i) mine = mine || gated[g_].packet == items[i].packet;
                layer = mine;
            }
        }
        if (layer && !ctx->tick && cfx_prepare(ctx) != CFX_OK) return CFX_ERR_LAUNCH;
        if (layer) {
            if (ctx->gate_err && *(volatile unsigned*)ctx->gate_err)
                return fail(ctx, CFX_ERR_GATE, "compress: an earlier gate / flag wait on this context timed out (cfx_gate_errors reads and clears the count)");
            const unsigned slot = ticket_slot(ctx, stream);
            TopkLayerArgs a;
            memset(&a, 0, sizeof(a));
            a.E = E;
            a.n_sw = (int)((E / 8 + 256 * TKL_SU - 1) / (256 * TKL_SU));
            a.n_st = a.n_sw * batch;
            a.n_dw = (int)((E / 8 + 256 * TKL_DU - 1) / (256 * TKL_DU));
            a.flags = flags;
            a.gate = ctx->gate + (size_t)slot * GATE_STRIDE;
            ctx->gate_expect[3 * slot] += (unsigned)a.n_st;
            a.gate_expect = ctx->gate_expect[3 * slot];
            a.err = ctx->gate_err;
            a.timeout = ctx->gate_timeout;
            if (xg) {
                a.xgate = a.gate + GATE_BLOCK;
                a.xexpect = ++ctx->gate_expect[3 * slot + 1];
                a.remote = xg->remote;
                fill_p2p(ctx, xg, a.p2p);
                xg->taken = 1;
                xg->p_gate = a.gate + GATE_LINE; xg->p_expect = a.gate_expect;      // the word the gate's last arriver writes for XCD 0
                xg->f_gate = a.xgate; xg->f_expect = a.xexpect;
            }
            const dim3 g((unsigned)(a.n_st + a.n_dw * n_gated));
            switch (param) {
                case 1: LAUNCH(ctx, KID_ABSMEAN_COMPRESS_GATED, s, k_topk_layer<1>, g, dim3(256), 0, s, b, gd, a); break;
                case 2: LAUNCH(ctx, KID_ABSMEAN_COMPRESS_GATED, s, k_topk_layer<2>, g, dim3(256), 0, s, b, gd, a); break;
                case 4: LAUNCH(ctx, KID_ABSMEAN_COMPRESS_GATED, s, k_topk_layer<4>, g, dim3(256), 0, s, b, gd, a); break;
                case 8: LAUNCH(ctx, KID_ABSMEAN_COMPRESS_GATED, s, k_topk_layer<8>, g, dim3(256), 0, s, b, gd, a); break;
                default: LAUNCH(ctx, KID_ABSMEAN_COMPRESS_GATED, s, k_topk_layer<16>, g, dim3(256), 0, s, b, gd, a); break;
            }
            return check_launch(ctx, "topk layer launch");
        }
        const dim3 g((unsigned)((E / 8 + 255) / 256), batch);
        switch (param) {
            case 1: LAUNCH(ctx, KID_TOPK_COMPRESS, s, k_topk_compress<1>, g, dim3(256), 0, s, b, E, flags); break;
            case 2: LAUNCH(ctx, KID_TOPK_COMPRESS, s, k_topk_compress<2>, g, dim3(256), 0, s, b, E, flags); break;
            case 4: LAUNCH(ctx, KID_TOPK_COMPRESS, s, k_topk_compress<4>, g, dim3(256), 0, s, b, E, flags); break;
            case 8: LAUNCH(ctx, KID_TOPK_COMPRESS, s, k_topk_compress<8>, g, dim3(256), 0, s, b, E, flags); break;
            default: LAUNCH(ctx, KID_TOPK_COMPRESS, s, k_topk_compress<16>, g, dim3(256), 0, s, b, E, flags); break;
        }
        const int rc_t = check_launch(ctx, "topk compress launch");
        // no layer form here: an exchange-layer op runs its exchange and the reconstruction behind this call; a plain gated call gets the
        // reconstruction in stream order
        if (rc_t != CFX_OK || xg || !n_gated) return rc_t;
        return cfx_i_decompress_impl(ctx, codec, N, C, param, n_gated, gated, stream, nullptr, 0u);
    }

    // statistics + finalize: ONE launch with the in-launch finalize (default), or the two-kernel sequence
    const bool fused = ctx->fused && CB <= TICK_MAX_CB;
    // Ticket / gate blocks are handed out round-robin from a ring PER STREAM (launches of one stream are in order, so a ring slot is
    // never shared by two launches in flight; one ring for every stream would let a stalled stream's launch meet a slot that another
    // stream has cycled back to)
    unsigned* tick = nullptr;
    unsigned slot = 0;
    const int stream_cus = stream_cu_count(ctx, stream);
    if (fused) {
        if (!ctx->tick && cfx_prepare(ctx) != CFX_OK) return CFX_ERR_LAUNCH;
        slot = ticket_slot(ctx, stream);
        tick = ctx->tick + (size_t)slot * CFX_MAX_BATCH * TICK_WORDS;
    }
    if (ctx->gate_err && *(volatile unsigned*)ctx->gate_err)
        return fail(ctx, CFX_ERR_GATE, "compress: an earlier gate / flag wait on this context timed out (cfx_gate_errors reads and clears the count)");
    const int R = fused ? fused_rows(ctx, N, C, batch, stream_cus) : auto_rows(ctx, N, C, batch, true);
    const int P = (N + R - 1) / R;
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
    if (codec == CFX_CODEC_BINARY || codec == CFX_CODEC_INT2) {
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
    } else {
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
    }
    return check_launch(ctx, "compress launch");
}

int cfx_compress_batch_ex(cfx_ctx* ctx, int codec, int N, int C, int param, int flags, int batch, const cfx_comp_item* items,
                          int n_ride, const cfx_decomp_item* ride, void* workspace, size_t workspace_bytes, void* stream) {
    return compress_impl(ctx, codec, N, C, param, flags, batch, items, n_ride, ride, 0, nullptr, workspace, workspace_bytes, stream);
}

int cfx_compress_batch_gated(cfx_ctx* ctx, int codec, int N, int C, int param, int flags, int batch, const cfx_comp_item* items,
                             int n_ride, const cfx_decomp_item* ride, int n_gated, const cfx_decomp_item* gated,
                             void* workspace, size_t workspace_bytes, void* stream) {
    return compress_impl(ctx, codec, N, C, param, flags, batch, items, n_ride, ride, n_gated, gated, workspace, workspace_bytes, stream);
}

int cfx_gate_errors(cfx_ctx* ctx) {
    if (!ctx) return CFX_ERR_NULL;
    if (!ctx->gate_err) return 0;
    const unsigned v = __atomic_exchange_n(ctx->gate_err, 0u, __ATOMIC_RELAXED);      // pinned host memory: no device synchronisation
    return (int)v;
}

// After a wait gave up: the launch it belonged to left arrival counters short of what the host expects of the ring slot, ticket words
// undrawn, possibly a low-rank hand-over arena mid-sum.  Drain the device, zero the counters and what the host expects of them, have the
// low-rank arenas re-zeroed at their next use, clear the error word.  Sequence-tagged words (min/max and abs-mean partials, tile flags)
// need nothing: their sequence numbers are never reused.  Returns the number of failed waits that were pending, or < 0.
int cfx_gate_recover(cfx_ctx* ctx) {
    if (!ctx) return CFX_ERR_NULL;
    if (!ctx->tick) return 0;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != ctx->device && hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, CFX_ERR_LAUNCH, "gate_recover: hipSetDevice failed");
    int rc = 0;
    const size_t words = (size_t)((char*)ctx->colgate - (char*)ctx->tick) / sizeof(unsigned);      // ticket blocks + gate blocks
    if (hipDeviceSynchronize() != hipSuccess || hipMemset(ctx->tick, 0, words * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        rc = fail(ctx, CFX_ERR_LAUNCH, "gate_recover: cannot reset the ticket / gate blocks");
    } else {
        memset(ctx->gate_expect, 0, sizeof(ctx->gate_expect));
        for (int i = 0; i < ctx->lrs_n; ++i) ctx->lrs_key[i] = ~0ull;
        rc = ctx->gate_err ? (int)__atomic_exchange_n(ctx->gate_err, 0u, __ATOMIC_RELAXED) : 0;
    }
    if (cur >= 0 && cur != ctx->device) (void)hipSetDevice(cur);
    return rc;
}

int cfx_set_gate_timeout_ms(cfx_ctx* ctx, int ms) {
    if (!ctx) return CFX_ERR_NULL;
    if (ms <= 0) return fail(ctx, CFX_ERR_BATCH, "gate timeout must be positive");
    ctx->gate_timeout = (long long)ms * 100000LL;
    return CFX_OK;
}

int cfx_compress_batch(cfx_ctx* ctx, int codec, int N, int C, int param, int flags, int batch, const cfx_comp_item* items,
                       void* workspace, size_t workspace_bytes, void* stream) {
    return cfx_compress_batch_ex(ctx, codec, N, C, param, flags, batch, items, 0, nullptr, workspace, workspace_bytes, stream);
}

int cfx_compress(cfx_ctx* ctx, int codec, const void* x, const void* base, void* new_base, void* packet, int N, int C, int param,
                 int flags, void* workspace, size_t workspace_bytes, void* stream) {
    cfx_comp_item it = {x, base, new_base, packet};
    return cfx_compress_batch(ctx, codec, N, C, param, flags, 1, &it, workspace, workspace_bytes, stream);
}

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

int cfx_decompress(cfx_ctx* ctx, int codec, const void* packet, const void* base, void* recon, int N, int C, int param, void* stream) {
    cfx_decomp_item it = {packet, base, recon};
    return cfx_decompress_batch(ctx, codec, N, C, param, 1, &it, stream);
}


// ---- entry points for cfx_plan.hip (plan replay, exchange lane) ------------------------------------------------------------
}  // extern "C"
bool cfx_i_shape_ok(int codec, int N, int C, int param) { return shape_ok(codec, N, C, param); }
// The 2-bit layer launch takes the external gate too (k_int2_compress_gated's group D).  With the exchange as a one-wave kernel on an exchange
// stream it measured 2.40-2.46 ms per FLUX step against 2.03 for three launches in stream order - a resident polling kernel on another queue
// alone costs that launch 4 us per layer (tools/xgate_probe.py, kind gated+poller) -; with the exchange INSIDE the launch (P2PInline) 2.11.
bool cfx_i_has_xlayer_form(int codec) { return codec >= CFX_CODEC_BINARY && codec <= CFX_CODEC_TOPK; }
unsigned* cfx_i_ticket_block(cfx_ctx* ctx, void* stream) {
    if (!ctx->tick && cfx_prepare(ctx) != CFX_OK) return nullptr;
    return ctx->tick + (size_t)ticket_slot(ctx, stream) * CFX_MAX_BATCH * TICK_WORDS;
}
int cfx_i_decompress_impl(cfx_ctx* ctx, int codec, int N, int C, int param, int batch, const cfx_decomp_item* items, void* stream,
                          unsigned* pre, unsigned pre_val) {
    return decompress_impl(ctx, codec, N, C, param, batch, items, stream, pre, pre_val);
}
size_t cfx_i_ws_words(int codec, int N, int C) { return ws_words(codec, N, C); }
int cfx_i_stream_cus(cfx_ctx* ctx, void* stream) { return stream_cu_count(ctx, stream); }
int cfx_i_compress_impl(cfx_ctx* ctx, int codec, int N, int C, int param, int flags, int batch, const cfx_comp_item* items,
                        int n_ride, const cfx_decomp_item* ride, int n_gated, const cfx_decomp_item* gated,
                        void* workspace, size_t workspace_bytes, void* stream, CfxXGate* xg) {
    return compress_impl(ctx, codec, N, C, param, flags, batch, items, n_ride, ride, n_gated, gated, workspace, workspace_bytes, stream, xg);
}
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
    if (N <= 0 || C <= 0 || (C % 8) || rank < 1 || rank > 8 || (((size_t)N * (C / 8)) % 2)) return 0;
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
    if (!cfx_binary_rank_packet_bytes(N, C, rank)) return fail(ctx, CFX_ERR_SHAPE, "binary rank-K compress: bad shape / rank (1 .. 8)");
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
    if (!cfx_binary_rank_packet_bytes(N, C, rank)) return fail(ctx, CFX_ERR_SHAPE, "binary rank-K decompress: bad shape / rank (1 .. 8)");
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

int cfx_residual2_delta(cfx_ctx* ctx, const void* x, const void* base, const void* delta_base, void* dd, size_t n, void* stream) {
    if (!ctx || !x || !base || !delta_base || !dd) return fail(ctx, CFX_ERR_NULL, "residual2_delta: null pointer");
    if (n == 0 || (n & 7)) return fail(ctx, CFX_ERR_SHAPE, "residual2_delta: element count must be a positive multiple of 8");
    if (!AL16(x) || !AL16(base) || !AL16(delta_base) || !AL16(dd)) return fail(ctx, CFX_ERR_ALIGN, "residual2_delta: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const size_t n8 = n / 8;
    LAUNCH(ctx, KID_RES2_DELTA, s, k_residual2_delta, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, (const h16*)x, (const h16*)base,
           (const h16*)delta_base, (h16*)dd, n8);
    return check_launch(ctx, "residual2_delta launch");
}

int cfx_residual2_update(cfx_ctx* ctx, const void* base, const void* delta_base, const void* recv, void* new_base, void* new_delta_base,
                         float decay, size_t n, void* stream) {
    if (!ctx || !base || !delta_base || !recv || !new_base || !new_delta_base) return fail(ctx, CFX_ERR_NULL, "residual2_update: null pointer");
    if (n == 0 || (n & 7)) return fail(ctx, CFX_ERR_SHAPE, "residual2_update: element count must be a positive multiple of 8");
    if (!AL16(base) || !AL16(delta_base) || !AL16(recv) || !AL16(new_base) || !AL16(new_delta_base))
        return fail(ctx, CFX_ERR_ALIGN, "residual2_update: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const size_t n8 = n / 8;
    LAUNCH(ctx, KID_RES2_UPDATE, s, k_residual2_update, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, (const h16*)base,
           (const h16*)delta_base, (const h16*)recv, (h16*)new_base, (h16*)new_delta_base, decay, n8);
    return check_launch(ctx, "residual2_update launch");
}

int cfx_attn_merge_wait(cfx_ctx* ctx, void* out, void* lse, const void* block_out, const void* block_lse, int B, int S, int H, int D,
                        int block_out_bshd, int first, const void* wait_flag, unsigned wait_value, void* stream) {
    if (!ctx || !out || !lse || !block_out || !block_lse) return fail(ctx, CFX_ERR_NULL, "attn_merge: null pointer");
    if (B <= 0 || S <= 0 || H <= 0 || D <= 0 || (D & 7) || D > 512) return fail(ctx, CFX_ERR_SHAPE, "attn_merge: head dim must be a positive multiple of 8, at most 512");
    if (!AL16(out) || !AL16(block_out)) return fail(ctx, CFX_ERR_ALIGN, "attn_merge: pointers must be 16-byte aligned");
    if (wait_flag && !ctx->gate_err && cfx_prepare(ctx) != CFX_OK) return CFX_ERR_LAUNCH;
    if (ctx->gate_err && *(volatile unsigned*)ctx->gate_err) return fail(ctx, CFX_ERR_GATE, "attn_merge: an earlier flag / gate wait on this context timed out (cfx_gate_errors)");
    hipStream_t s = (hipStream_t)stream;
    int lg = 0;
    while ((1 << lg) < D / 8) ++lg;          // a row's D/8 threads in 2^lg lanes of one wave
    const size_t total = ((size_t)B * S * H) << lg;
    LAUNCH(ctx, KID_ATTN_MERGE, s, k_attn_merge, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (float*)out, (float*)lse,
           (const h16*)block_out, (const float*)block_lse, B, S, H, D, first,
           (size_t)S * H * D, block_out_bshd ? (size_t)H * D : (size_t)D, block_out_bshd ? (size_t)D : (size_t)S * D, lg,
           (const unsigned*)wait_flag, wait_value, ctx->gate_err, ctx->gate_timeout);
    return check_launch(ctx, "attn_merge launch");
}

int cfx_attn_merge(cfx_ctx* ctx, void* out, void* lse, const void* block_out, const void* block_lse, int B, int S, int H, int D,
                   int block_out_bshd, int first, void* stream) {
    return cfx_attn_merge_wait(ctx, out, lse, block_out, block_lse, B, S, H, D, block_out_bshd, first, nullptr, 0u, stream);
}

int cfx_copy_probe(cfx_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream) {
    if (!ctx || !dst || !src) return fail(ctx, CFX_ERR_NULL, "copy_probe: null");
    if ((bytes & 15) || !AL16(dst) || !AL16(src)) return fail(ctx, CFX_ERR_ALIGN, "copy_probe: 16-byte granularity");
    { hipStream_t s = (hipStream_t)stream; LAUNCH(ctx, KID_COPY_PROBE, s, k_copy_probe, dim3(2048), dim3(256), 0, s, (uint4*)dst, (const uint4*)src, bytes / 16); }
    return check_launch(ctx, "copy_probe launch");
}

}  // extern "C"

// Device code shared by the codec families of libcfx.so (cfx_absmean.hip, cfx_minmax.hip, cfx_topk.hip, cfx_api.hip): element types, loads /
// stores, the tagged-word arenas, tile coordinates, the ticket / gate geometry and the in-launch waits, the peer-to-peer exchange a layer
// launch runs inside itself.  (Round 6: cfx_kernels.hip was one 4 130-line translation unit; it is now this header + one file per family
// + the C-ABI.)
#ifndef CFX_DEVICE_H
#define CFX_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cfx.h"
#include "cfx_internal.h"

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

#define MML_MAX_TILES 2048     // statistics tiles of one min/max layer launch = flag words per ring and kind (cfx_prepare sizes them)
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


// The plain copy the roofline's `achievable` is measured with (bench.py) and the PMC counters are calibrated on (tools/pmc_summary.py).  Every
// workgroup copies ONE contiguous chunk, 16 bytes a lane, four loads in flight per thread before the first store, non-temporal both ways
// (a one-touch stream): 6.1 TB/s at 96 MiB with 8192 workgroups (round 6, build/cb/copybench.hip: the grid-stride form it replaces gave
// 5.1-5.7, hipMemcpyAsync 5.1; the guide's float4 copy: 6.3).
static __global__ __launch_bounds__(256) void k_copy_probe(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_* s4 = reinterpret_cast<const u32x4_*>(src);
    u32x4_* d4 = reinterpret_cast<u32x4_*>(dst);
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
    for (size_t i = b0 + threadIdx.x; i < b1; i += 256 * 4) {
        u32x4_ a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + 256 * u < b1) a[u] = __builtin_nontemporal_load(s4 + i + 256 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + 256 * u < b1) __builtin_nontemporal_store(a[u], d4 + i + 256 * u);
    }
}

#endif

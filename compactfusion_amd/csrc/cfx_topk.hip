// libcfx.so - the 1:m top-1 sparsifier (compress_topk.py:11-163): compress / decompress kernels and the layer launch (k_topk_layer).
// Shared device code: cfx_device.h; the C-ABI and the dispatch: cfx_api.hip.
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
// host side: this family's launches (validated and dispatched by cfx_api.hip)
// ---------------------------------------------------------------------------------------------------
int cfx_i_topk_compress(CompressCall& cc) {
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

int cfx_i_topk_decompress(cfx_ctx* ctx, int N, int C, int param, int batch, const BatchD& b, void* stream, unsigned* pre, unsigned pre_val) {
    hipStream_t s = (hipStream_t)stream;
    const size_t E = (size_t)N * C;
    const dim3 g((unsigned)((E / 8 + 255) / 256), batch);
    switch (param) {
        case 1: LAUNCH(ctx, KID_TOPK_DECOMPRESS, s, k_topk_decompress<1>, g, dim3(256), 0, s, b, E, pre, pre_val); break;
        case 2: LAUNCH(ctx, KID_TOPK_DECOMPRESS, s, k_topk_decompress<2>, g, dim3(256), 0, s, b, E, pre, pre_val); break;
        case 4: LAUNCH(ctx, KID_TOPK_DECOMPRESS, s, k_topk_decompress<4>, g, dim3(256), 0, s, b, E, pre, pre_val); break;
        case 8: LAUNCH(ctx, KID_TOPK_DECOMPRESS, s, k_topk_decompress<8>, g, dim3(256), 0, s, b, E, pre, pre_val); break;
        default: LAUNCH(ctx, KID_TOPK_DECOMPRESS, s, k_topk_decompress<16>, g, dim3(256), 0, s, b, E, pre, pre_val); break;
    }
    return check_launch(ctx, "decompress launch");
}

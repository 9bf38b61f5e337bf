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
#include <algorithm>
#include "cfx.h"
#include "cfx_internal.h"
#include "cfx_device.h"
#define CFX_API_TU
#include "cfx_host.h"

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

int cfx_i_auto_rows(const cfx_ctx* ctx, int N, int C, int batch, bool stats) {
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
static int stream_cu_count_impl(cfx_ctx* ctx, void* stream) {
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
int cfx_i_decompress_checked(cfx_ctx* ctx, int codec, int N, int C, int param, int batch, const cfx_decomp_item* items, void* stream,
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
    const int R = auto_rows(ctx, N, C, batch, false);
    switch (codec) {
        case CFX_CODEC_BINARY:
        case CFX_CODEC_INT2: return cfx_i_absmean_decompress(ctx, codec, N, C, batch, b, R, stream, pre, pre_val);
        case CFX_CODEC_INT4:
        case CFX_CODEC_INT8: return cfx_i_minmax_decompress(ctx, codec, N, C, batch, b, R, stream, pre, pre_val);
        default: return cfx_i_topk_decompress(ctx, N, C, param, batch, b, stream, pre, pre_val);
    }
}

int cfx_decompress_batch(cfx_ctx* ctx, int codec, int N, int C, int param, int batch, const cfx_decomp_item* items, void* stream) {
    return cfx_i_decompress_checked(ctx, codec, N, C, param, batch, items, stream, nullptr, 0u);
}

int cfx_i_fused_rows(const cfx_ctx* ctx, int N, int C, int batch, int cus) {
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

// Ticket / gate blocks are handed out round-robin from a ring PER STREAM (launches of one stream are in order, so a ring slot is never
// shared by two launches in flight; one ring for every stream would let a stalled stream's launch meet a slot that another stream has
// cycled back to).  Needs cfx_prepare.
unsigned cfx_i_ticket_slot(cfx_ctx* ctx, void* stream) {
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
void cfx_i_fill_p2p(cfx_ctx* ctx, CfxXGate* xg, P2PInline& p) {
    memset(&p, 0, sizeof(p));
    if (!xg->p2p_own) return;
    p.own = xg->p2p_own;
    p.n_peers = xg->p2p_n;
    for (int i = 0; i < xg->p2p_n; ++i) p.peer[i] = xg->p2p_peer[i];
    p.timeout = ctx->gate_timeout;
    xg->inline_done = 1;
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
    CompressCall cc;
    cc.ctx = ctx; cc.codec = codec; cc.N = N; cc.C = C; cc.param = param; cc.flags = flags; cc.batch = batch; cc.items = items;
    cc.n_ride = n_ride; cc.n_gated = n_gated; cc.gated = gated; cc.stream = stream; cc.xg = xg; cc.b = b; cc.rd = rd; cc.gd = gd;
    cc.ws = ws; cc.wstride = wstride; cc.CB = CB; cc.upd = upd; cc.capturing = capturing;
    cc.fused = false; cc.tick = nullptr; cc.slot = 0; cc.stream_cus = 0; cc.R = cc.P = 0;
    if (codec == CFX_CODEC_TOPK) return cfx_i_topk_compress(cc);

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
    cc.fused = fused; cc.tick = tick; cc.slot = slot; cc.stream_cus = stream_cus; cc.R = R; cc.P = P;
    return (codec == CFX_CODEC_BINARY || codec == CFX_CODEC_INT2) ? cfx_i_absmean_compress(cc) : cfx_i_minmax_compress(cc);
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
    return cfx_i_decompress_checked(ctx, codec, N, C, param, batch, items, stream, pre, pre_val);
}
size_t cfx_i_ws_words(int codec, int N, int C) { return ws_words(codec, N, C); }
int cfx_i_stream_cus(cfx_ctx* ctx, void* stream) { return stream_cu_count_impl(ctx, stream); }
int cfx_i_compress_impl(cfx_ctx* ctx, int codec, int N, int C, int param, int flags, int batch, const cfx_comp_item* items,
                        int n_ride, const cfx_decomp_item* ride, int n_gated, const cfx_decomp_item* gated,
                        void* workspace, size_t workspace_bytes, void* stream, CfxXGate* xg) {
    return compress_impl(ctx, codec, N, C, param, flags, batch, items, n_ride, ride, n_gated, gated, workspace, workspace_bytes, stream, xg);
}

extern "C" {

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
    { hipStream_t s = (hipStream_t)stream; const size_t n16 = bytes / 16; const unsigned g = (unsigned)(n16 / 1024 < 8192 ? (n16 / 1024 ? n16 / 1024 : 1) : 8192); LAUNCH(ctx, KID_COPY_PROBE, s, k_copy_probe, dim3(g), dim3(256), 0, s, (uint4*)dst, (const uint4*)src, n16); }
    return check_launch(ctx, "copy_probe launch");
}

}  // extern "C"

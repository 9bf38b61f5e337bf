// Internals shared by the translation units of libcfx.so (not part of the ABI).
#ifndef CFX_INTERNAL_H
#define CFX_INTERNAL_H
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include "cfx.h"

enum {
    KID_ABSMEAN_STATS_BITS = 1, KID_ABSMEAN_STATS = 2, KID_ABSMEAN_FINALIZE = 3, KID_BINARY_DEQUANT = 4,
    KID_INT2_QUANT = 5, KID_INT2_DEQUANT = 6, KID_MINMAX_STATS = 7, KID_MINMAX_FINALIZE = 8,
    KID_INT8_QUANT = 9, KID_INT8_DEQUANT = 10, KID_INT4_QUANT = 11, KID_INT4_DEQUANT = 12,
    KID_TOPK_COMPRESS = 13, KID_TOPK_DECOMPRESS = 14, KID_COPY_PROBE = 15, KID_BINARY_EF = 16,
    KID_LR_PREP = 17, KID_LR_AQ = 18, KID_LR_ATY = 19, KID_LR_CHOL = 20, KID_LR_APPLY = 21, KID_LR_DECODE = 22,
    KID_BINARY_PIPE = 23, KID_BINARY_PIPE_EDGE = 24, KID_RES2_DELTA = 25, KID_RES2_UPDATE = 26,
    KID_ABSMEAN_COMPRESS_BITS = 27, KID_ABSMEAN_COMPRESS = 28, KID_MINMAX_COMPRESS = 29, KID_ATTN_MERGE = 30,
    KID_ABSMEAN_COMPRESS_GATED = 31, KID_MAX = 32
};
static const char* const kid_names[KID_MAX] = {
    "", "k_absmean_stats<bits>", "k_absmean_stats", "k_absmean_finalize", "k_binary_dequant", "k_int2_quant", "k_int2_dequant",
    "k_minmax_stats", "k_minmax_finalize", "k_int8_quant", "k_int8_dequant", "k_int4_quant", "k_int4_dequant",
    "k_topk_compress", "k_topk_decompress", "k_copy_probe", "k_binary_dequant(ef)",
    "k_lr_prep", "k_lr_aq", "k_lr_aty", "k_lr_chol", "k_lr_apply", "k_lr_decode",
    "k_binary_pipe", "k_binary_pipe(prologue/epilogue)", "k_residual2_delta", "k_residual2_update",
    "k_absmean_compress<bits>", "k_absmean_compress", "k_minmax_compress", "k_attn_merge",
    "gated layer launch (k_absmean_compress<bits,gated> / k_int2_compress_gated)"};

struct ProfRec { int kid; hipEvent_t a, b; };

struct cfx_ctx {
    int device;
    int rows_per_tile;
    // native per-launch timing (hipEvents recorded on the launch stream around selected kernels)
    ProfRec* prof;
    int prof_cap, prof_n;
    int prof_stride;                // record every prof_stride-th eligible launch of each kernel id
    int prof_seen[KID_MAX];
    unsigned prof_mask;
    // in-launch finalize: ticket blocks (device memory, zeroed once, self-resetting) handed out round-robin, one per launch
    unsigned* tick;
    unsigned tick_next;
    // gated reconstruction: one monotonic arrival counter per ticket-ring slot (64 B apart, after the ticket blocks), the value
    // at which the slot's next launch opens, and one error word (a gate that never opened)
    unsigned* gate;
    unsigned gate_expect[2 * 256];  // two gates per slot (the 2-bit layer launch has two)
    unsigned* gate_err;
    int fused;                      // 1 (default): compress = statistics + in-launch finalize; 0: separate finalize kernel
    void* dbg_stamps;               // developer hook (cfx_debug_stamps)
    int stats_rows;                 // CFX_STATS_ROWS override of the statistics tile height (experiments), 0 = automatic
    char err[256];
};

// Returns the record slot for this launch or -1.  A profiled launch goes through hipExtLaunchKernelGGL, which ties the
// two events to the dispatch packet itself: their elapsed time is the kernel's execution time (as rocprofv3 reports it),
// not the kernel plus the command processor's event handling that a hipEventRecord bracket would add (~6 us here).
static inline int prof_slot(cfx_ctx* ctx, int kid) {
    if (!ctx->prof_mask || !(ctx->prof_mask & (1u << kid)) || ctx->prof_n >= ctx->prof_cap) return -1;
    if ((ctx->prof_seen[kid]++ % ctx->prof_stride) != 0) return -1;
    const int i = ctx->prof_n++;
    ctx->prof[i].kid = kid;
    return i;
}
#define LAUNCH(ctx, kid, s, kern, grid, block, shm, strm, ...) do { \
        const int _pi = prof_slot(ctx, kid); \
        if (_pi >= 0) hipExtLaunchKernelGGL(kern, grid, block, shm, strm, (ctx)->prof[_pi].a, (ctx)->prof[_pi].b, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid, block, shm, strm, __VA_ARGS__); \
    } while (0)

// Same, with an optional event that must fire when THIS launch has finished (used to hand work to another stream).  When
// the launch is not being profiled the event rides on the dispatch packet itself (no marker packet on the stream: a
// marker costs ~6 us of queue time between two kernels on this stack); otherwise it is recorded right after the launch.
#define LAUNCH_DONE(ctx, kid, s, done_ev, kern, grid, block, shm, strm, ...) do { \
        const int _pi = prof_slot(ctx, kid); \
        if (_pi >= 0) { \
            hipExtLaunchKernelGGL(kern, grid, block, shm, strm, (ctx)->prof[_pi].a, (ctx)->prof[_pi].b, 0, __VA_ARGS__); \
            if (done_ev) (void)hipEventRecord(done_ev, strm); \
        } else if (done_ev) hipExtLaunchKernelGGL(kern, grid, block, shm, strm, nullptr, done_ev, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid, block, shm, strm, __VA_ARGS__); \
    } while (0)

static inline int fail(cfx_ctx* ctx, int code, const char* msg) {
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s", msg);
    return code;
}


static inline int check_launch(cfx_ctx* ctx, const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[200];
        snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
        return fail(ctx, CFX_ERR_LAUNCH, buf);
    }
    return CFX_OK;
}

#define AL16(p) ((((uintptr_t)(p)) & 15) == 0)
#endif

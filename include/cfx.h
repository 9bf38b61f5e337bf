/*
 * cfx.h - C-ABI of libcfx.so: MI355X (gfx950) residual-compressed activation exchange kernels.
 *
 * This is the drop-in boundary for CompactFusion's compressor hot path.  The reference has no
 * FFI (it is Python + Triton); these entry points are what a binding for that
 * path would call, one per reference kernel wrapper:
 *
 *   cfx_compress[_batch]   replaces  binary_quant_fastpath   xfuser/compact/fastpath.py:124-228  (+ eager scale prologue :150-166)
 *                                    int2_quant_fastpath     xfuser/compact/fastpath.py:584-669  (+ prologue :614-625)
 *                                    quantize_int8 on delta  xfuser/compact/compress_quantize.py:428-471 composed with main.py:227-233
 *                                    quantize_int4 on delta  xfuser/compact/compress_quantize.py:522-583 composed with main.py:227-233
 *                                    topk_compress           xfuser/compact/compress_topk.py:11-105 (slowpath.py:76-79)
 *                                    and the wire packing    xfuser/compact/main.py:149-152, slowpath.py:83
 *   cfx_decompress[_batch] replaces  binary_dequant_fastpath xfuser/compact/fastpath.py:371-438
 *                                    int2_dequant_fastpath   xfuser/compact/fastpath.py:745-811
 *                                    dequantize_int8/int4    xfuser/compact/compress_quantize.py:473-484, :585-640 (+ base add main.py:376)
 *                                    topk_decompress         xfuser/compact/compress_topk.py:108-163
 *                                    and the wire unpacking  xfuser/compact/main.py:285-304, slowpath.py:137-169
 *   cfx_packet_bytes       replaces  the size arithmetic of  xfuser/compact/main.py:285-293, slowpath.py:111-135
 *
 * Conventions
 *   - All tensor pointers are DEVICE pointers to contiguous row-major (N, C) fp16 ("half") data.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls are asynchronous.
 *   - No torch types, no exceptions: every call returns CFX_OK (0) or a negative error code and
 *     records a message retrievable with cfx_last_error_string().
 *   - The library keeps no state besides the opaque cfx_ctx (device id, tuning, last error).
 *     Workspace is caller-provided so calls can be captured in a hipGraph.
 *
 * Wire layouts (little endian; identical to the reference where the reference defines one)
 *   BINARY  [ bits  N*C/8 B : bit i of byte j of row n = (x-base)[n,8j+i] >= 0 | U  N fp16 | V  C fp16 ]   main.py:149-152
 *   INT2    [ codes N*C/4 B : 2-bit (sign<<1|mag) , element j at bits 2(j%4)   | tok N fp16 | chan C fp16 ]   main.py:149-152
 *   INT4    [ codes N*C/2 B : byte [n/2][c] = q[n][c] | q[n+1][c]<<4           | scale C fp16 | min C fp16 ]   compress_quantize.py:566-573
 *   INT8    [ q     N*C   B : int8                                             | scale C fp16 | zp  C int16 ]  compress_quantize.py:463-471
 *   TOPK    [ val N*C/m fp16 | idx N*C/(2m) B : (i1<<4)|i2 per 2m-block of the flat (-1,1024) view ]           slowpath.py:76-79
 */
#ifndef CFX_H
#define CFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFX_ABI_VERSION 1
#define CFX_MAX_BATCH 16

typedef struct cfx_ctx cfx_ctx;

enum cfx_status {
    CFX_OK = 0,
    CFX_ERR_NULL = -1,       /* a required pointer is NULL */
    CFX_ERR_SHAPE = -2,      /* N/C/param violate the codec's divisibility rules */
    CFX_ERR_ALIGN = -3,      /* a tensor pointer is not 16-byte aligned / packet not 2-byte aligned */
    CFX_ERR_CODEC = -4,      /* unknown codec id */
    CFX_ERR_BATCH = -5,      /* batch < 1 or > CFX_MAX_BATCH */
    CFX_ERR_LAUNCH = -6,     /* hipGetLastError() after a launch was not hipSuccess */
    CFX_ERR_WORKSPACE = -7,  /* workspace NULL or smaller than cfx_workspace_bytes() */
    CFX_ERR_GATE = -8,       /* an in-launch gate / lane flag wait of an EARLIER launch on this context timed out (cfx_gate_errors) */
    CFX_ERR_QUEUES = -9      /* flag-ordered streams asked for, but the process did not give HIP's streams hardware queues of their own
                              * (GPU_MAX_HW_QUEUES unset: cfx_hw_queues_ok) */
};

enum cfx_codec {
    CFX_CODEC_BINARY = 1,    /* COMPACT_COMPRESS_TYPE.BINARY fastpath, comp_rank = -1 */
    CFX_CODEC_INT2 = 2,      /* COMPACT_COMPRESS_TYPE.INT2 fastpath */
    CFX_CODEC_INT4 = 3,      /* per-channel min/max 16 levels, rows paired per byte */
    CFX_CODEC_INT8 = 4,      /* per-channel affine int8, zero point int16 */
    CFX_CODEC_TOPK = 5       /* COMPACT_COMPRESS_TYPE.SPARSE, param = m in {1,2,4,8,16} */
};

enum cfx_flags {
    CFX_FLAG_UPDATE_CACHE = 1, /* write new_base (compact_compress(update_cache=True)) */
    CFX_FLAG_NO_EF = 2         /* error feedback off: new_base = x (main.py:233 `else x`) */
};

/* One tensor of a compress batch.  base may be NULL (compress_residual == 0: the codec sees x itself);
 * new_base may alias base (in-place error-feedback update) and is ignored unless CFX_FLAG_UPDATE_CACHE. */
typedef struct cfx_comp_item {
    const void* x;        /* (N,C) fp16, 16-byte aligned */
    const void* base;     /* (N,C) fp16, 16-byte aligned, or NULL */
    void*       new_base; /* (N,C) fp16, 16-byte aligned, or NULL */
    void*       packet;   /* cfx_packet_bytes() bytes, 2-byte aligned (16-byte aligned is faster) */
} cfx_comp_item;

/* One tensor of a decompress batch.  recon = base + decode(packet) (base NULL: recon = decode(packet)).
 * recon may alias base (the receiver's cache update, main.py:317-319). */
typedef struct cfx_decomp_item {
    const void* packet;
    const void* base;
    void*       recon;
} cfx_decomp_item;

int         cfx_abi_version(void);
cfx_ctx*    cfx_create(int device);
void        cfx_destroy(cfx_ctx* ctx);
const char* cfx_last_error_string(cfx_ctx* ctx);

/* rows per workgroup tile for the streaming kernels (0 = automatic). */
int         cfx_set_rows_per_tile(cfx_ctx* ctx, int rows);

/* Size in bytes of one packet / of the scratch workspace a compress batch needs. 0 on invalid arguments. */
size_t      cfx_packet_bytes(int codec, int N, int C, int param);
size_t      cfx_workspace_bytes(int codec, int N, int C, int param, int batch);

int cfx_compress_batch(cfx_ctx* ctx, int codec, int N, int C, int param, int flags,
                       int batch, const cfx_comp_item* items,
                       void* workspace, size_t workspace_bytes, void* stream);

int cfx_decompress_batch(cfx_ctx* ctx, int codec, int N, int C, int param,
                         int batch, const cfx_decomp_item* items, void* stream);

/* cfx_compress_batch plus `n_ride` reconstruction items of the SAME codec and shape (1-bit codec only) that are
 * co-scheduled in the statistics launch: bandwidth work that does not depend on this call's scales - typically the previous
 * layer's deferred error-feedback update (own packet applied to own state; the reference does it inside
 * _binary_quant_fastpath, fastpath.py:88-120, but nothing reads that state before the next denoise step: the local
 * attention block uses the uncompressed K,V, ring.py:207-209) - streams while the scale reduction, which is pure
 * latency, completes.  The ride items must not alias this call's x / base / packet operands. */
int cfx_compress_batch_ex(cfx_ctx* ctx, int codec, int N, int C, int param, int flags,
                          int batch, const cfx_comp_item* items, int n_ride, const cfx_decomp_item* ride,
                          void* workspace, size_t workspace_bytes, void* stream);

/* cfx_compress_batch_ex plus `n_gated` reconstruction items (any streaming codec, same shape) whose PACKETS ARE PRODUCED BY THIS CALL:
 * typically the call's own packets applied to the rank's own state (the error-feedback update, fastpath.py:88-120) and -
 * when logical peers are looped back on one GPU - to the peers' states.  They run in the SAME launch as the compress: their
 * workgroups first pull the state tiles into registers (bandwidth work that overlaps the scale reduction, which is pure
 * latency), wait on an arrival counter until the packet is complete, then finish from registers.  One launch per layer instead
 * of two, 2.125 instead of 4.125 B/el behind the dependency.  With CFX_FLAG_UPDATE_CACHE the call's own error-feedback update
 * (new_base) is part of the launch too (1-bit: as two more gated tensors; 2-bit: the statistics workgroups quantise their own
 * tiles from registers once the scales exist, a second gate releases the gated items; int4 / int8: k_minmax_layer, the statistics tile kept
 * in registers, codes and the state update from them; top-k: k_topk_layer, nothing global to wait for - the gate counts the compress
 * workgroups).  Results are identical to compress
 * followed by cfx_decompress_batch; when the shape does not qualify (C % 128 != 0, statistics tile != 32 rows, in-launch
 * finalize off, too many items for one launch) exactly that sequence runs.
 * A gated item's base/recon must not alias this call's x / packet operands; recon may equal base, and a gated item may
 * update a compress item's `base` in place (the compress group has finished reading it when the gate opens).
 * Inside the launch (round 5): the 1-bit / 2-bit statistics tiles hand their partial sums over as TAGGED 8-byte words {24-bit launch
 * tag | 40-bit sum} in an arena the context keeps per stream (zeroed at allocation, tags never reused) - the data is its own "published"
 * mark: no drain, no ticket, and no 32-bit cliff (round 4's words gave out at an average |residual| of 0.5; beyond 40 bits - an average
 * of 128 - the exact sum travels in the caller's workspace); FIXED workgroups (the last-dispatched tiles) poll the words they reduce and
 * publish the scales into the packet and, tagged, into the arena.  2-bit: a statistics tile polls those tagged scales (no gate 1), and a
 * reconstruction tile waits for the 4 - 6 tiles whose codes it reads (per-tile flags) instead of the slowest of all (2.10 -> 1.89 ms per
 * FLUX step).  1-bit: the reconstruction tiles keep the arrival gate (XCD-relayed; tagged scales measured 10 % slower there).
 * Stream capture (round 6): the ONE-launch forms take the value their gates open at, their ticket-ring slot and their launch tags as launch
 * arguments the host advances with every launch (monotonic arrival counters, no reset, no memset node) - a replayed node would wait for
 * numbers that have gone by.  So on a CAPTURING stream this call - and the exchange-layer ops of a plan - enqueue the capturable sequence
 * instead: compress (tickets that reset themselves) ; reconstruct the gated items in stream order; identical results, two launches (int4 /
 * int8: three) instead of one, no host call per replay.  tests/test_gpu_api.py::test_layer_calls_are_graph_capturable replays every codec's
 * layer call and the peer-to-peer layer op four times between eager launches of the same context.  The ungated launches (cfx_compress_batch
 * / _ex) are capturable as they are.  (An int4 / int8 compress call outside a capture runs as ONE launch too - statistics, scales, codes
 * and the state update from the tile in registers, k_minmax_layer - handing its partials over as sequence-tagged words in an arena the
 * context keeps per stream; under stream capture the same call runs the capturable sequence statistics ; quantise, with identical results.) */
int cfx_compress_batch_gated(cfx_ctx* ctx, int codec, int N, int C, int param, int flags,
                             int batch, const cfx_comp_item* items, int n_ride, const cfx_decomp_item* ride,
                             int n_gated, const cfx_decomp_item* gated,
                             void* workspace, size_t workspace_bytes, void* stream);
/* Number of in-launch gate / lane flag waits that timed out since the last call (a bounded spin gives up instead of hanging the GPU;
 * always 0 unless the library is broken or a stream was starved), or < 0 on error.  Reads and clears a pinned host word: no device
 * synchronisation (a complete count needs the streams drained first).  While the count is non-zero every compress / plan / merge
 * call on the context returns CFX_ERR_GATE. */
int cfx_gate_errors(cfx_ctx* ctx);
/* Every in-launch wait (arrival gates, tagged-word and flag polls, the peer-to-peer exchange inside a layer launch, the low-rank
 * hand-overs) gives up on ONE time base: the 100 MHz wall clock against the context's gate timeout (default 5 s,
 * cfx_set_gate_timeout_ms), never on an iteration count.  A workgroup whose wait gave up does NOT store: reconstructions from packets
 * that have not arrived are never written, the states the workgroup owns stay as they were, and the error word counts the failure.
 * The launch it belonged to leaves the context's arrival counters short; cfx_gate_recover drains the device, resets them (and the
 * low-rank hand-over arenas) and clears the error word, after which the context is usable again.  Returns the number of failed waits
 * that were pending, or < 0.  What the caller does about the layer whose exchange failed (run it again over another transport, or drop
 * the generation) is its business: compact/xlayer.py re-validates. */
int cfx_gate_recover(cfx_ctx* ctx);

/* The statistics pass of a compress call reduces its partial sums INSIDE the launch (last-arriving workgroups, ticket
 * counters; replaces the eager scale prologue of fastpath.py:150-166 / compress_quantize.py:452-463 and the separate
 * finalize kernel).  The tickets live in device memory owned by the context: cfx_prepare allocates them (idempotent;
 * otherwise the first compress call does - call it before capturing compress calls into a hipGraph).  Compress calls
 * on one context may come from several streams (a ticket ring per stream, 8 rings, the least recently used one is reassigned;
 * launches of one stream are in order).
 * cfx_set_fused_finalize(ctx, 0) selects the two-kernel sequence (statistics, finalize); results are bit-identical. */
int cfx_prepare(cfx_ctx* ctx);
int cfx_set_fused_finalize(cfx_ctx* ctx, int on);
/* (Developer probes - per-workgroup phase stamps, the launch-tag test hook, early exits of the compress kernel - are NOT part of this
 * library: include/cfx_dev.h, libcfx_dev.so, built with -DCFX_DEV_PROBES.) */
/* Tuning / measurement switches of a context.  The library reads NO environment variable for its behaviour (the one variable it looks
 * at, GPU_MAX_HW_QUEUES, belongs to the HIP runtime: see cfx_prepare below); what earlier builds read from the environment is set here:
 *   cfx_set_stats_rows    statistics tile height of the one-launch compress (multiple of 16; 0 = automatic)
 *   cfx_set_gated_launch  0: the gated / exchange-layer ops always run as compress ; exchange ; reconstruct in stream order (1 = default:
 *                         the one-launch forms where they qualify)
 *   cfx_set_lr_chain      low-rank factor chain: 0 = automatic (slab-resident single launch where its workgroups fit the stream, else
 *                         the six-launch N-space chain, else the C-space chain), 1 = never the single launch, 2 = C-space chain only
 *   cfx_set_lr_decode     low-rank reconstruction kernel: 0 = automatic (MFMA form at rank 32), 1 = VALU form, 2 = MFMA form */
int cfx_set_stats_rows(cfx_ctx* ctx, int rows);
int cfx_set_gated_launch(cfx_ctx* ctx, int on);
int cfx_set_lr_chain(cfx_ctx* ctx, int chain);
int cfx_set_lr_decode(cfx_ctx* ctx, int mode);
/* Flag-ordered launches (the exchange-layer ops, the exchange lane) need the streams they order to sit on hardware queues of their own:
 * a polling kernel is never scheduled out for the kernel it waits for.  HIP multiplexes streams over a pool of hardware queues; with
 * GPU_MAX_HW_QUEUES unset a stream created after a collective library initialised can be time-sliced against the exchange stream's
 * queue (measured: the layer launch 25 -> 50-100 us, and gate time-outs under stream churn).  cfx_hw_queues_ok() returns 1 when the
 * process set the variable to >= 2 before HIP started (any explicit value restores one queue per stream), 0 otherwise.  With 0 the
 * plan ops that would order two streams by flags (cfx_plan_add_exchange_layer[_p2p] in their one-launch form, cfx_plan_run_lane) are
 * refused with CFX_ERR_QUEUES - the exchange-layer ops then run in stream order on one stream (same results) - unless
 * cfx_set_allow_shared_queues(ctx, 1) says the caller knows better. */
int cfx_hw_queues_ok(void);
int cfx_set_allow_shared_queues(cfx_ctx* ctx, int on);

/* Single-tensor conveniences (batch of one). */
int cfx_compress(cfx_ctx* ctx, int codec, const void* x, const void* base, void* new_base, void* packet,
                 int N, int C, int param, int flags, void* workspace, size_t workspace_bytes, void* stream);
int cfx_decompress(cfx_ctx* ctx, int codec, const void* packet, const void* base, void* recon,
                   int N, int C, int param, void* stream);

/* The 2-bit quantise kernel alone, SCALES GIVEN: each item's packet tail already holds tok (N fp16, after the N*C/4 code bytes) and chan
 * (C fp16); codes (+ error-feedback state with CFX_FLAG_UPDATE_CACHE) are written.  Replaces the Triton kernel _int2_quant_fastpath
 * (xfuser/compact/fastpath.py:486-580) as the reference launches it behind its eager scale prologue (:614-625) - what a parity test
 * needs to compare codes bit for bit given the reference's own scale vectors. */
int cfx_int2_quantize(cfx_ctx* ctx, int N, int C, int flags, int batch, const cfx_comp_item* items, void* stream);

/* Low-rank residual codecs (compactfusion_amd/csrc/cfx_lowrank.hip).
 *   cfx_lr_compress_batch   replaces subspace_iter (xfuser/compact/compress_lowrank.py:14-61) + the LOW_RANK / LOW_RANK_Q
 *                           encode of slowpath.py:54-75 + the residual / error-feedback flow of main.py:227-233
 *   cfx_lr_decompress_batch replaces slowpath.py:120-131, :151-164 (torch.matmul(u, v), int4 factor dequant) + main.py:376
 * quantized = 0: LOW_RANK   wire [ U (N,r) fp16 | V (r,C) fp16 ]
 * quantized = 1: LOW_RANK_Q wire [ int4(U) (N/2,r) | scale r | min r | int4(V^T) (C/2,r) | scale r | min r ]   (rank % 8 == 0)
 * rank even, <= 32.  init_q[i] -> device C x RP fp32 row-major start matrix, RP = 8/16/32 = rank rounded up, columns >= rank
 * zero; it need not be orthonormal (the iteration only sees its span); its entries enter the first product as fp16 (hi + lo): keep them
 * below 65504 in magnitude (randn is what the reference draws).  Workspace from cfx_lr_workspace_bytes.
 * rank <= 16 on a shard of at most 576 tokens: ONE persistent launch (csrc/cfx_lrslab.hip) whose workgroups wait for each other - taken
 * only where C / 32 workgroups per tensor fit the CUs of `stream` (otherwise a multi-launch chain runs; same results within the codec's
 * tolerance).  It hands its partial sums over through an arena the context owns (allocated on the first call of a stream, zeroed when
 * the shape changes): make one call before capturing the stream into a hipGraph; the launch itself is capturable. */
size_t cfx_lr_packet_bytes(int quantized, int N, int C, int rank);
size_t cfx_lr_workspace_bytes(int quantized, int N, int C, int rank, int batch);
int    cfx_lr_compress_batch(cfx_ctx* ctx, int quantized, int N, int C, int rank, int flags, int batch,
                             const cfx_comp_item* items, const void* const* init_q,
                             void* workspace, size_t workspace_bytes, void* stream);
int    cfx_lr_decompress_batch(cfx_ctx* ctx, int quantized, int N, int C, int rank, int batch,
                               const cfx_decomp_item* items, void* workspace, size_t workspace_bytes, void* stream);

/* 1-bit codec with rank-K scales (COMPACT_COMPRESS_TYPE.BINARY with comp_rank >= 1; deprecated in the reference, main.py:188-189):
 *   cfx_binary_rank_compress_batch   replaces binary_quant_fastpath(rank >= 1) (xfuser/compact/fastpath.py:124-228: subspace_iter(|x - base|)
 *                                    for the scales + the K-loop of _binary_quant_fastpath :88-120) and quantize_1bit(rank >= 1)
 *                                    (compress_quantize.py:37-49)
 *   cfx_binary_rank_decompress_batch replaces binary_dequant_fastpath (fastpath.py:371-438, K-loop :330-360) / dequantize_1bit
 * scale[n, c] = fp16(sum_k fp16(U[n,k] * V[c,k])), out = base + (2 b - 1) * scale.  Wire [ bits N*C/8 | U (N,K) fp16 | V (C,K) fp16 ]
 * (main.py:149-152).  rank 1 .. 32 (the reference's Triton kernels take the powers of two, fastpath.py:91 tl.arange(0, RANK); 32 is the
 * factor chain's limit); init_q[i] as for cfx_lr_compress_batch (device C x {8, 16, 32} fp32 - the rank rounded up -, columns >= rank zero). */
size_t cfx_binary_rank_packet_bytes(int N, int C, int rank);
size_t cfx_binary_rank_workspace_bytes(int N, int C, int rank, int batch);
int    cfx_binary_rank_compress_batch(cfx_ctx* ctx, int N, int C, int rank, int flags, int batch, const cfx_comp_item* items,
                                      const void* const* init_q, void* workspace, size_t workspace_bytes, void* stream);
int    cfx_binary_rank_decompress_batch(cfx_ctx* ctx, int N, int C, int rank, int batch, const cfx_decomp_item* items, void* stream);

/* Native per-launch timing.  When enabled, every `stride`-th launch of a kernel whose id bit is set in kernel_mask
 * is bracketed by hipEvents recorded on the launch stream (up to `capacity` records; capacity 0 disables).  An event
 * pair costs ~2-5 us of stream time, hence the stride.
 * cfx_profile_read synchronises on the recorded events, returns their count and resets the log.
 * Kernel ids: 1 absmean_stats<bits>, 2 absmean_stats, 3 absmean_finalize, 4 binary_dequant, 5 int2_quant,
 * 6 int2_dequant, 7 minmax_stats, 8 minmax_finalize, 9 int8_quant, 10 int8_dequant, 11 int4_quant,
 * 12 int4_dequant, 13 topk_compress, 14 topk_decompress, 15 copy_probe, 16 binary_dequant launched as the
 * sender's error-feedback update, 17-22 low-rank chain (prep, aq, aty, chol, apply, decode), 23 binary_pipe (steady-state
 * fused launch of cfx_plan_run_pipelined), 24 binary_pipe prologue / epilogue / ragged-unit launches, 25 residual2_delta,
 * 26 residual2_update, 27 absmean_compress<bits> (statistics + sign bits + in-launch finalize [+ ride-along reconstruction]),
 * 28 absmean_compress (2-bit statistics + in-launch finalize), 29 minmax_compress (int4 / int8 statistics + in-launch finalize), 30 attn_merge. */
int         cfx_profile_enable(cfx_ctx* ctx, int capacity, unsigned kernel_mask, int stride);
int         cfx_profile_read(cfx_ctx* ctx, int* kernel_ids, float* ms, int cap);
const char* cfx_kernel_name(int kernel_id);

/* Plan: a prebuilt schedule of batch ops (e.g. the 57 layers x {compress, reconstruct} of one denoise step) that is
 * replayed from native code - the MI355X-native replacement for the reference's per-call Python dispatch
 * (xfuser/compact/ring.py:188-206, main.py:169-270).  cfx_plan_add_* copy their arguments and return the op index
 * (>= 0) or an error (< 0); cfx_plan_run launches ops [first_op, first_op + n_ops) on `stream`. */
typedef struct cfx_plan cfx_plan;
cfx_plan* cfx_plan_create(cfx_ctx* ctx);
void      cfx_plan_destroy(cfx_plan* plan);
int       cfx_plan_add_compress(cfx_plan* plan, int codec, int N, int C, int param, int flags, int batch,
                                const cfx_comp_item* items, void* workspace, size_t workspace_bytes);
int       cfx_plan_add_compress_ex(cfx_plan* plan, int codec, int N, int C, int param, int flags, int batch,
                                   const cfx_comp_item* items, int n_ride, const cfx_decomp_item* ride,
                                   void* workspace, size_t workspace_bytes);   /* cfx_compress_batch_ex as a plan op */
int       cfx_plan_add_compress_gated(cfx_plan* plan, int codec, int N, int C, int param, int flags, int batch,
                                      const cfx_comp_item* items, int n_ride, const cfx_decomp_item* ride,
                                      int n_gated, const cfx_decomp_item* gated,
                                      void* workspace, size_t workspace_bytes);   /* cfx_compress_batch_gated as a plan op */
int       cfx_plan_add_decompress(cfx_plan* plan, int codec, int N, int C, int param, int batch,
                                  const cfx_decomp_item* items);
/* cfx_lr_compress_batch / cfx_lr_decompress_batch as plan ops (a LOW_RANK / LOW_RANK_Q layer replayed without per-op marshalling:
 * slowpath.py:54-75, :120-131, :151-164 once per layer of patchpara/fwd.py:108-131).  Pointers (items, init_q, workspace) must stay
 * valid while the plan is replayed. */
int       cfx_plan_add_lr_compress(cfx_plan* plan, int quantized, int N, int C, int rank, int flags, int batch,
                                   const cfx_comp_item* items, const void* const* init_q, void* workspace, size_t workspace_bytes);
int       cfx_plan_add_lr_decompress(cfx_plan* plan, int quantized, int N, int C, int rank, int batch,
                                     const cfx_decomp_item* items, void* workspace, size_t workspace_bytes);
/* Exchange ops.  An all-gather op is ordered after everything the plan enqueued on the main stream before it (the
 * packets are complete); with a side stream (modes 1, 2) a wait op makes the main stream wait for that gather, so
 * "compress(l+1), gather(l+1), wait(l), reconstruct(l)" overlaps the wire with the codec; in mode 0 waits are no-ops. */
typedef struct cfx_comm cfx_comm;
/* where all-gather ops run: 0 = in order on the main stream (default; measured best for layer-sized work, a
 * cross-stream event hop costs ~10 us here), 1 = side stream, 2 = prioritised side stream.  Set before adding ops.
 * cfx_plan_run_pipelined with mode 1 / 2 issues the collectives of unit u on that stream underneath the fused launch that
 * follows finalize(u) (one more unit of look-ahead; no wait ops needed). */
int       cfx_plan_set_exchange_stream(cfx_plan* plan, int mode);
/* Use the caller's stream as this plan's exchange stream (mode 1 semantics; the plan does not own it).  Many plans - one
 * per layer - should share ONE exchange stream: every stream is a hardware queue to the dispatcher. */
int       cfx_plan_use_exchange_stream(cfx_plan* plan, void* stream);
/* layers (1..7, default 7) per unit of cfx_plan_run_pipelined; set before the first replay */
int       cfx_plan_set_pipe_unit_layers(cfx_plan* plan, int layers);
int       cfx_plan_add_all_gather(cfx_plan* plan, cfx_comm* comm, const void* send, void* recv, size_t bytes_per_rank);
int       cfx_plan_add_wait(cfx_plan* plan, int gather_op);
/* Exchange layer: compress ; all-gather ; reconstruct as ONE op - the in-order layer of the gather schedules (reference
 * xfuser/compact/ring.py:188-206 + 265-269, patchpara/fwd.py:108-137: quantise, exchange, dequantise every peer's shard onto its
 * state).  The reconstruction workgroups are launched WITH the compress group on the stream cfx_plan_run is given: they pull their
 * state tiles into registers while the statistics chain and the collective run, and continue when the plan's exchange stream - which
 * waits (a flag kernel) until the launch's packets are complete, issues ncclAllGather(send, recv, bytes_per_rank) and then sets the
 * launch's external gate - says the packets have arrived.  `recon` items read their packets from wherever the collective leaves them
 * (or from this op's own packet operands: with CFX_FLAG_UPDATE_CACHE the rank's own error-feedback update joins them).  comm NULL:
 * nothing moves, the exchange stream runs one relay kernel (wait + set).  Any codec; the one-launch form exists for the 1-bit codec.
 * When the one-launch form is not available (codec or shape
 * without it, a run stream masked below 128 CUs, the legacy NULL stream beside a BLOCKING exchange stream - every CU-masked stream is
 * one -, cfx_hw_queues_ok() == 0, cfx_plan_run_async / _lane) the op runs as compress ; all-gather ; reconstruct in order - same results.  The exchange stream must own a hardware queue (see "Exchange lane" below): the plan creates a
 * CU-masked one unless cfx_plan_use_exchange_stream supplied it (one stream should serve all plans).  A gate that never opens times
 * out like any flag wait (CFX_ERR_GATE at the next call).
 * With more than one rank the collective is a KERNEL that has to be placed while the reconstruction workgroups hold their CUs: the
 * one-launch form is then taken only if that group leaves 32 workgroup slots of the run stream's CUs free (room for RCCL's workgroups:
 * 256 threads x ~280 VGPRs on gfx950), otherwise the op runs in order.  Returns the op index. */
int       cfx_plan_add_exchange_layer(cfx_plan* plan, int codec, int N, int C, int param, int flags, int batch,
                                      const cfx_comp_item* items, int n_recon, const cfx_decomp_item* recon,
                                      cfx_comm* comm, const void* send, void* recv, size_t bytes_per_rank,
                                      void* workspace, size_t workspace_bytes);
/* The exchange layer WITHOUT a collective, for the GPUs of one node: every rank's packets stay where the compress launch wrote them - in
 * a buffer from cfx_ipc_alloc that the peers have opened with cfx_ipc_open - and a peer's reconstruction workgroups read them from
 * there over xGMI (`recon` items point into the opened mappings).  What is exchanged is one 4-byte word per rank and layer, and the exchange
 * runs INSIDE the launch: workgroup 0, once its own tile is done, waits until the launch's packets are complete, advances *own_flag by one
 * (the op's execution count 1, 2, ... - taken from the word itself on the device, so a rebuilt plan keeps counting where the words stand;
 * every rank executes the same ops equally often), waits until every peer_flags[i] has reached that count and opens the launch's gate.  ONE
 * launch per layer on ONE stream: no exchange stream, no collective kernel that has to find CUs beside the waiting workgroups, no
 * hardware-queue requirement, any run stream (the legacy NULL stream included).  own_flag and the peers' flags live in cfx_ipc_alloc memory
 * (zero-initialised), one word per op and plan; a rank may rewrite a layer's packets only after its peers have moved past that layer: a plan
 * has at least two such ops per replay, or the caller double-buffers packets and flag words by execution parity (compact/xlayer.py).
 * Replaces the all-gather of ring.py:188-206 / patchpara/fwd.py:108-109 on a single node.  1-bit, 2-bit, int4, int8 (the min/max and 2-bit
 * codecs where all their statistics workgroups are co-resident); otherwise - shape / stream without the one-launch form - compress ; a
 * one-wave kernel that publishes and waits ; reconstruct in stream order: same results.  Time-outs as for cfx_plan_add_exchange_layer. */
#define CFX_P2P_MAX_PEERS 15
int       cfx_plan_add_exchange_layer_p2p(cfx_plan* plan, int codec, int N, int C, int param, int flags, int batch,
                                          const cfx_comp_item* items, int n_recon, const cfx_decomp_item* recon,
                                          void* own_flag, int n_peers, const void* const* peer_flags,
                                          void* workspace, size_t workspace_bytes);
/* The publish-and-wait step of the p2p exchange as an op of its own, for chains that keep separate launches (compress ; p2p_sync ;
 * reconstruct peer 1 ; reconstruct peer 2 ; ... - the lane plan of compact_fwd): runs in stream order where the op range runs; what
 * the ops before it wrote is complete when *own_flag is published, the ops behind it start after every peer has published.  Flags
 * as in cfx_plan_add_exchange_layer_p2p (one word per op and plan, cfx_ipc_alloc memory, execution count as the epoch). */
int       cfx_plan_add_p2p_sync(cfx_plan* plan, void* own_flag, int n_peers, const void* const* peer_flags);
/* Device memory shared between the processes of a node (hipIpcGetMemHandle / hipIpcOpenMemHandle; on hosts with dmabuf IPC only the
 * processes need HSA_ENABLE_IPC_MODE_LEGACY=0).  cfx_ipc_alloc: zeroed device memory + its 64-byte handle (send it to the peers by any
 * means); cfx_ipc_open: map a peer's allocation; close / free when done.
 * The memory is UNCACHED device memory (hipExtMallocWithFlags(hipDeviceMallocUncached); fine-grained, then ordinary memory as fall-backs,
 * cfx_ipc_memory_kind says which): peers poll words in it and read packets from it while the producing kernel is still running, and the
 * same addresses are rewritten every step - ordinary device memory is only promised coherent across devices at kernel boundaries. */
int       cfx_ipc_alloc(cfx_ctx* ctx, size_t bytes, void** ptr, void* handle64);
int       cfx_ipc_memory_kind(cfx_ctx* ctx);   /* what the last cfx_ipc_alloc of the context returned: 2 uncached, 1 fine-grained, 0 ordinary */
int       cfx_set_ipc_memory_kind(cfx_ctx* ctx, int kind);   /* what cfx_ipc_alloc asks for first (default 2; measurements: 1, 0) */
int       cfx_ipc_open(cfx_ctx* ctx, const void* handle64, void** ptr);
int       cfx_ipc_close(cfx_ctx* ctx, void* ptr);
int       cfx_ipc_free(cfx_ctx* ctx, void* ptr);
/* One hop of the ring relay (reference xfuser/compact/ring.py:193-195,265-269: RingComm.send_recv / commit / wait): send
 * `bytes` to rank+1 and receive `bytes` from rank-1 as one grouped ncclSend + ncclRecv, on the exchange stream like an
 * all-gather op (cfx_plan_add_wait applies).  W-1 hops relay every rank's packet around the ring. */
int       cfx_plan_add_ring_hop(cfx_plan* plan, cfx_comm* comm, const void* send, void* recv, size_t bytes);
/* Re-point the activation of item `item` of compress op `op` (the K / V tensor a layer hands over changes from call to
 * call; state, packet and workspace operands are bound once). */
int       cfx_plan_set_input(cfx_plan* plan, int op, int item, const void* x);
int       cfx_plan_size(const cfx_plan* plan);
int       cfx_plan_copy_op(cfx_plan* dst, const cfx_plan* src, int op);   /* append a copy of a (de)compress op of `src` */
int       cfx_plan_run(cfx_plan* plan, int first_op, int n_ops, void* stream);
/* cfx_plan_run after re-pointing the n_xs activations of the first compress op of the range (cfx_plan_set_input + run in
 * one host call: what a layer's K,V hand-over costs). */
int       cfx_plan_run_x(cfx_plan* plan, int first_op, int n_ops, const void* const* xs, int n_xs, void* stream);
/* The op range - compress, collective, reconstruction - on the plan's EXCHANGE stream, forked off `main_stream` (an event)
 * so that it runs beside what the caller enqueues on `main_stream` next: the local attention block, which needs none of it
 * (reference ring.py:207-209).  Exchange ops of the range run in order on that stream; wait ops are no-ops.
 * cfx_plan_join(plan, main_stream) makes `main_stream` wait for the range (before the first peer block).  Two host calls per
 * layer.  The activations must stay alive until the join. */
int       cfx_plan_run_async(cfx_plan* plan, int first_op, int n_ops, const void* const* xs, int n_xs, void* main_stream);
int       cfx_plan_join(cfx_plan* plan, void* main_stream);
/* Exchange lane: the layer's chain (compress, collective, per-peer reconstruction) on its own - normally CU-masked - stream, ordered
 * with the compute stream ONLY through flag words in device memory (no events: a cross-stream event hop costs ~14 us of idle queue
 * time on MI355X / ROCm 7.2, a flag written by one stream's kernel and polled by the other's ~1.7 us; tools/lane_probe.hip).
 * Replaces the reference's fork / join of the K,V exchange around the local attention block (xfuser/compact/ring.py:191-269:
 * RingComm.send_recv + commit before the block, wait after it, decompress on the compute stream).
 *   cfx_plan_flags(plan, n)       allocates the plan's n flag words (64 bytes apart, zeroed; once per plan) and returns the device
 *                                 address of flag 0, or NULL.  Flags hold monotonic EPOCHS: one per cfx_plan_run_lane call.
 *   cfx_plan_add_flag_wait / _set plan ops: wait until flag i has reached the current epoch / set flag i to it.  A set op makes
 *                                 everything the ops before it wrote visible to whoever waits (kernel boundary in an in-order stream).
 *   cfx_plan_run_lane             advances the epoch, launches "set flag `ready_flag`" on `compute_stream` (behind the kernels that
 *                                 produce the activations xs, see cfx_plan_run_x) and replays the op range on the plan's exchange
 *                                 stream (cfx_plan_use_exchange_stream); the range normally starts with a wait op on `ready_flag`.
 *                                 *epoch_out receives the epoch the range's set ops will publish.  One host call per layer.
 *   cfx_attn_merge_wait           cfx_attn_merge whose launch also waits (one lane, after its merge work) until *wait_flag has
 *                                 reached wait_value: the attention block that follows it in the compute stream then finds the
 *                                 peer's reconstructed K,V complete.  wait_flag NULL = cfx_attn_merge.
 *   cfx_flag_set / cfx_flag_wait  the same two tiny kernels for callers that order other work (flag = any 4-byte-aligned device word).
 * A wait gives up after the context's gate timeout (default 5 s, cfx_set_gate_timeout_ms) and counts the failure in a pinned host
 * word: the next plan / compress / merge call on the context returns CFX_ERR_GATE, cfx_gate_errors reads and clears the count
 * without synchronising the device.
 * The two streams a flag orders must NOT share a hardware queue (the polling kernel would block the kernel it waits for until the
 * timeout): HIP multiplexes ordinary streams over a small pool of queues, a CU-masked stream owns its queue - create at least the
 * exchange stream with cfx_stream_create_masked (a full mask, first_cu = 0, n_cus = all, is fine).
 *   cfx_stream_create_masked      a stream restricted to CU-mask bits [first_cu, first_cu + n_cus) (hipExtStreamCreateWithCUMask; on
 *                                 MI355X bit i = CU i/8 of XCD i%8, so a contiguous range is the same share of every XCD).  The lane
 *                                 uses two with DISJOINT ranges - e.g. 32 CUs for the exchange, 224 for the compute stream the
 *                                 attention kernels run on - so that neither slows the other's workgroups (tools/sdpa_mask_probe.py:
 *                                 SDPA 37 us alone, 38 us beside a saturating copy on the other 32 CUs, 93 us when it shares them). */
void*     cfx_plan_flags(cfx_plan* plan, int n);
int       cfx_plan_add_flag_wait(cfx_plan* plan, int flag);
int       cfx_plan_add_flag_set(cfx_plan* plan, int flag);
int       cfx_plan_set_pre_flag(cfx_plan* plan, int op, int flag);   /* reconstruction op `op` sets `flag` as the first thing its launch
                                                                       * does (= after the op in front of it, without a launch of its own) */
unsigned  cfx_plan_epoch(const cfx_plan* plan);
int       cfx_plan_run_lane(cfx_plan* plan, int first_op, int n_ops, const void* const* xs, int n_xs, int ready_flag,
                            void* compute_stream, unsigned* epoch_out);
/* cfx_plan_run_lane's first half on its own: advance the epoch and launch "set flag `ready_flag`" on `compute_stream`.  A later
 * cfx_plan_run_lane with compute_stream = NULL then only replays the op range (same epoch): the caller enqueues the local attention
 * block in between, so the chain's host issue overlaps GPU work. */
int       cfx_plan_lane_begin(cfx_plan* plan, int ready_flag, void* compute_stream, unsigned* epoch_out);
int       cfx_flag_set(cfx_ctx* ctx, void* flag, unsigned value, void* stream);
int       cfx_flag_wait(cfx_ctx* ctx, const void* flag, unsigned value, void* stream);
int       cfx_stream_create_masked(cfx_ctx* ctx, int first_cu, int n_cus, void** stream);
int       cfx_stream_destroy(cfx_ctx* ctx, void* stream);
int       cfx_set_gate_timeout_ms(cfx_ctx* ctx, int ms);
/* Software-pipelined replay (replaces the reference's strictly sequential quantise -> cat -> send -> dequantise per layer,
 * xfuser/compact/ring.py:188-260 with fastpath.py:124-228, 371-438).  If ops [first_op, first_op + n_ops) are a sequence of "groups"
 *     k x compress (BINARY, flags without UPDATE_CACHE)   { all-gather }*   k x decompress (BINARY)      of one shape,
 * (k >= 1 layers whose packets travel in one collective) consecutive groups are merged into units of up to 7 layers
 * (cfx_plan_set_pipe_unit_layers; at most 112 reconstruction and 16 compress items per unit) and replayed on `stream` as
 *     { all-gathers of unit t-2 } ; [dequant(unit t-2) | finalize(unit t-1) | stats(unit t)]          t = 0, 1, ...
 * with every bracket ONE fused launch, so the small statistics kernels of later layers run underneath the
 * reconstruction of earlier ones.  Results are bit-identical to cfx_plan_run; packet / state buffers must be distinct
 * per layer within the call; the statistics workspaces are plan-owned (the ops' own are not used).  Any other op
 * sequence is replayed by cfx_plan_run. */
int       cfx_plan_run_pipelined(cfx_plan* plan, int first_op, int n_ops, void* stream);
/* Call once after the last cfx_plan_add_*: recognises the unit schedule of the whole plan for cfx_plan_run_pipelined and
 * makes every allocation a replay needs (statistics workspaces, the context's ticket blocks), so that cfx_plan_run /
 * cfx_plan_run_pipelined do no host allocation, environment lookup or device allocation per step (optional: the first
 * replay of a range does the same work otherwise). */
int       cfx_plan_finalize(cfx_plan* plan);

/* Process-global state, the only one besides cfx_ctx: the table of RCCL entry points filled by cfx_rccl_load (RCCL is a
 * process-wide library; loading it twice would give two collective runtimes).  Communicators themselves are per cfx_comm.
 * cfx_comm_create leaves the caller's current device unchanged.
 * RCCL communicator owned by the library (replaces yunchang's RingComm + torch.distributed P2P of the reference,
 * xfuser/compact/ring.py:172,193-195,265-267).  RCCL is loaded at run time (cfx_rccl_load: pass the path of the
 * librccl the process already uses, or NULL to search); rank 0 makes the 128-byte unique id (cfx_comm_unique_id),
 * the host shares it by any means, every rank calls cfx_comm_create (collective). */
int       cfx_rccl_load(const char* path);   /* a different explicit path replaces the table for communicators created afterwards */
int       cfx_comm_unique_id(cfx_ctx* ctx, void* out128);
cfx_comm* cfx_comm_create(cfx_ctx* ctx, const void* id128, int nranks, int rank);
void      cfx_comm_destroy(cfx_comm* comm);
int       cfx_comm_all_gather(cfx_comm* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);
int       cfx_comm_ring_hop(cfx_comm* comm, const void* send, void* recv, size_t bytes, void* stream);

/* Second-order residual (CompactConfig(residual=2), xfuser/compact/main.py:244-266 compress, :378-384 decompress): the
 * predictor arithmetic around any codec, n fp16 elements (multiple of 8), one fp16 rounding per reference operation.
 *   cfx_residual2_delta :  dd = (x - base) - delta_base                        -> compress dd with base = NULL
 *   cfx_residual2_update:  new_base = (base + delta_base) + recv ; new_delta_base = fp16(fp32(fp16(delta_base + recv)) * decay)
 * (recv = decode(packet) with base = NULL; decay = CompactConfig.delta_decay_factor, main.py:272-273).  new_base may alias
 * base and new_delta_base may alias delta_base. */
int cfx_residual2_delta(cfx_ctx* ctx, const void* x, const void* base, const void* delta_base, void* dd, size_t n, void* stream);
int cfx_residual2_update(cfx_ctx* ctx, const void* base, const void* delta_base, const void* recv, void* new_base,
                         void* new_delta_base, float decay, size_t n, void* stream);

/* Ring-attention block merge - the consumer side of the exchange (reference xfuser/compact/ring.py:263 calls
 * yunchang.ring.utils.update_out_and_lse, un-vendored; published formula):
 *   out <- out - sigmoid(lse_b - lse) * (out - out_b) ;  lse <- lse - logsigmoid(lse - lse_b)         (fp32)
 * out fp32 [B][S][H][D], lse fp32 [B][S][H]; block_out fp16 [B][H][S][D] (block_out_bshd = 0) or [B][S][H][D] (= 1), block_lse fp32 [B][H][S]
 * (the fused SDPA kernel's own output layouts).  first != 0 initialises out / lse from the block.  D % 8 == 0.  One launch per block instead of ~10
 * eager elementwise kernels. */
int cfx_attn_merge(cfx_ctx* ctx, void* out, void* lse, const void* block_out, const void* block_lse, int B, int S, int H, int D,
                   int block_out_bshd, int first, void* stream);
int cfx_attn_merge_wait(cfx_ctx* ctx, void* out, void* lse, const void* block_out, const void* block_lse, int B, int S, int H, int D,
                        int block_out_bshd, int first, const void* wait_flag, unsigned wait_value, void* stream);

/* Bandwidth probe: dst[i] = src[i] over `bytes` (multiple of 16) - the achievable-HBM reference
 * against which bench.py reports roofline fractions. */
int cfx_copy_probe(cfx_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CFX_H */

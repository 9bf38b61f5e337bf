// Internals shared by the translation units of libcfx.so (not part of the ABI).
#ifndef CFX_INTERNAL_H
#define CFX_INTERNAL_H
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include "cfx.h"
#ifdef CFX_DEV_PROBES
#include "cfx_dev.h"
#endif

enum {
    KID_ABSMEAN_STATS_BITS = 1, KID_ABSMEAN_STATS = 2, KID_ABSMEAN_FINALIZE = 3, KID_BINARY_DEQUANT = 4,
    KID_INT2_QUANT = 5, KID_INT2_DEQUANT = 6, KID_MINMAX_STATS = 7, KID_MINMAX_FINALIZE = 8,
    KID_INT8_QUANT = 9, KID_INT8_DEQUANT = 10, KID_INT4_QUANT = 11, KID_INT4_DEQUANT = 12,
    KID_TOPK_COMPRESS = 13, KID_TOPK_DECOMPRESS = 14, KID_COPY_PROBE = 15, KID_BINARY_EF = 16,
    KID_LR_PREP = 17, KID_LR_AQ = 18, KID_LR_ATY = 19, KID_LR_CHOL = 20, KID_LR_APPLY = 21, KID_LR_DECODE = 22,
    KID_BINARY_PIPE = 23, KID_BINARY_PIPE_EDGE = 24, KID_RES2_DELTA = 25, KID_RES2_UPDATE = 26,
    KID_ABSMEAN_COMPRESS_BITS = 27, KID_ABSMEAN_COMPRESS = 28, KID_MINMAX_COMPRESS = 29, KID_ATTN_MERGE = 30,
    KID_ABSMEAN_COMPRESS_GATED = 31, KID_MAX = 32,
    KID_LR_CHAIN = KID_LR_PREP      // the single-launch chain has no separate prep launch: it takes that id (the kernel mask is 32 bits)
};
static const char* const kid_names[KID_MAX] = {
    "", "k_absmean_stats<bits>", "k_absmean_stats", "k_absmean_finalize", "k_binary_dequant", "k_int2_quant", "k_int2_dequant",
    "k_minmax_stats", "k_minmax_finalize", "k_int8_quant", "k_int8_dequant", "k_int4_quant", "k_int4_dequant",
    "k_topk_compress", "k_topk_decompress", "k_copy_probe", "k_binary_dequant(ef)",
    "k_lr_prep | k_lrs (slab-resident chain, one launch)", "k_lr_aq", "k_lr_aty", "k_lrg_gy (last arriver: factorisations)", "k_lr_apply", "k_lr_decode",
    "k_binary_pipe", "k_binary_pipe(prologue/epilogue)", "k_residual2_delta", "k_residual2_update",
    "k_absmean_compress<bits>", "k_absmean_compress", "k_minmax_compress", "k_attn_merge",
    "gated layer launch (k_absmean_compress<bits,gated> / k_int2_compress_gated / k_minmax_layer / k_topk_layer)"};

// Developer probes (include/cfx_dev.h; `python -m compactfusion_amd.build --dev-probes` builds libcfx_dev.so with -DCFX_DEV_PROBES): phase
// times of a workgroup on the 100 MHz wall clock, 16 words a workgroup.  In the PRODUCT build `Probe` is an empty type - no kernel
// argument, no register, no branch in any kernel - and the entry points that would set one up do not exist.
#ifdef CFX_DEV_PROBES
struct Probe {
    unsigned long long* p;
    __host__ __device__ Probe(unsigned long long* q = nullptr) : p(q) {}
    __device__ __forceinline__ bool on() const { return p != nullptr; }
    __device__ __forceinline__ void at(int k) const { if (p && threadIdx.x == 0) p[k] = (unsigned long long)wall_clock64(); }
    __device__ __forceinline__ void set(int k, unsigned long long v) const { if (p && threadIdx.x == 0) p[k] = v; }
    __device__ __forceinline__ void copy(int dst, int src) const { if (p && threadIdx.x == 0) p[dst] = p[src]; }
    __device__ __forceinline__ Probe of(size_t wg) const { return Probe(p ? p + wg * 16 : nullptr); }
};
#else
struct Probe {
    __host__ __device__ Probe() {}
    __device__ __forceinline__ constexpr bool on() const { return false; }
    __device__ __forceinline__ void at(int) const {}
    __device__ __forceinline__ void set(int, unsigned long long) const {}
    __device__ __forceinline__ void copy(int, int) const {}
    __device__ __forceinline__ Probe of(size_t) const { return Probe(); }
};
#endif

struct ProfRec { int kid; hipEvent_t a, b; };
#define CFX_RING_STREAMS 8       // ticket / gate rings of a context: one per stream that issues compress launches

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
    unsigned tick_next[CFX_RING_STREAMS];
    void* ring_stream[CFX_RING_STREAMS];   // the stream each ring serves
    int n_ring_streams;
    unsigned long long ring_used[CFX_RING_STREAMS], ring_clock;   // least recently used ring changes hands when all are taken
    void* cu_cache_stream[8];              // stream -> CUs its queue may use (CU-masked streams: fewer than the device has)
    int cu_cache_n[8], n_cu_cache;
    unsigned cu_cache_next;
    // gated reconstruction: one monotonic arrival counter per ticket-ring slot (64 B apart, after the ticket blocks), the value
    // at which the slot's next launch opens, and one error word (a gate that never opened)
    unsigned* gate;
    unsigned gate_expect[3 * 256 * CFX_RING_STREAMS];  // three gates per slot (the 2-bit exchange layer has three)
    unsigned* colgate;              // tile flags of the min/max layer launch: per ring 2048 words ("codes published")
    unsigned mml_seq;               // sequence number of the min/max layer launches of this context: the value of a launch's flags and the tag of its partials
    unsigned long long* mml_arena[CFX_RING_STREAMS];        // per ring: tagged partials + scales of the min/max layer launch (zeroed at allocation, grown on demand)
    size_t mml_arena_bytes[CFX_RING_STREAMS];
    bool mml_arena_owned[CFX_RING_STREAMS];
    void* mml_arena_owner[CFX_RING_STREAMS];   // the stream whose launches use the ring's arena (a change of owner waits for the previous one's launches)
    unsigned long long* abs_arena[CFX_RING_STREAMS];        // per ring: tagged partial sums of the 1-bit / 2-bit layer launches (zeroed at allocation, grown on demand)
    size_t abs_arena_bytes[CFX_RING_STREAMS];
    unsigned abs_seq;               // tags of those launches: 24 bits of this counter, never 0
    unsigned* gate_err;             // pinned HOST word (device-visible): waits that timed out since the last cfx_gate_errors
    long long gate_timeout;         // ticks of the 100 MHz wall clock a flag wait may last
    int fused;                      // 1 (default): compress = statistics + in-launch finalize; 0: separate finalize kernel
    void* dev_buf;                  // cfx_dev_stamps (developer build): where the probes write
    int stats_rows;                 // cfx_set_stats_rows: override of the statistics tile height (experiments), 0 = automatic
    int gated_on;                   // cfx_set_gated_launch: 1 (default) the one-launch gated / exchange-layer forms where they qualify
    int lr_chain, lr_decode;        // cfx_set_lr_chain / cfx_set_lr_decode (0 = automatic)
    int dev_probe;                  // cfx_set_dev_probe (developer builds)
    int allow_shared_queues;        // cfx_set_allow_shared_queues: flag-ordered streams even when cfx_hw_queues_ok() == 0
    int ipc_want;                   // cfx_set_ipc_memory_kind: what cfx_ipc_alloc asks for first (2 uncached - default -, 1 fine-grained, 0 ordinary)
    int ipc_kind;                   // what the last cfx_ipc_alloc returned: 2 uncached, 1 fine-grained, 0 ordinary device memory
    // hand-over arenas of the slab-resident low-rank chain: one per stream that launches it (zeroed when allocated and whenever the
    // shape it is laid out for changes: its words carry sequence tags that only make sense against what the chain itself wrote)
    void* lrs_stream[8];
    char* lrs_arena[8];
    size_t lrs_bytes[8];
    unsigned long long lrs_key[8];
    int lrs_n;
    unsigned lrs_next;
    void* lrs_last_stream;          // the stream of the last slab-resident launch (launches of two streams must not be in flight together)
    hipEvent_t lrs_ev;
    char err[256];
};

// what the launches of a context hand their kernels as `Probe` (developer build: the buffer cfx_dev_stamps set; product build: nothing)
static inline Probe cfx_i_probe(const cfx_ctx* ctx) {
#ifdef CFX_DEV_PROBES
    return Probe((unsigned long long*)ctx->dev_buf);
#else
    (void)ctx;
    return Probe();
#endif
}


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

// ---------------------------------------------------------------------------------------------------
// Plan / communicator internals, shared by cfx_absmean.hip (the fused pipeline launch) and cfx_plan.hip (everything else)
// ---------------------------------------------------------------------------------------------------
// ---- plan: a prebuilt schedule of batch ops replayed from native code (no per-op Python marshalling) -------------
// ---- RCCL, loaded at run time (the same library instance PyTorch-ROCm uses; no link-time dependency) ----------------
typedef struct { char internal[128]; } cfx_nccl_uid;
typedef void* cfx_nccl_comm;
struct RcclApi {
    void* handle;
    int (*GetUniqueId)(cfx_nccl_uid*);
    int (*CommInitRank)(cfx_nccl_comm*, int, cfx_nccl_uid, int);
    int (*AllGather)(const void*, void*, size_t, int, cfx_nccl_comm, hipStream_t);
    int (*CommDestroy)(cfx_nccl_comm);
    const char* (*GetErrorString)(int);
    int (*Send)(const void*, size_t, int, int, cfx_nccl_comm, hipStream_t);
    int (*Recv)(void*, size_t, int, int, cfx_nccl_comm, hipStream_t);
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    char path[512];
};

struct cfx_comm {
    cfx_ctx* ctx;
    cfx_nccl_comm comm;
    int nranks, rank;
    RcclApi api;
};

// what an exchange-layer launch tells its caller: the packets of the launch are complete once counter *p_gate has reached p_expect;
// the gated reconstruction group proceeds once *f_gate == f_expect
// needs_room (in): a collective kernel will run while the reconstruction group waits - take the one-launch form only if the group leaves it CUs
// p_count: the packets are complete once the p_count CONSECUTIVE words at p_gate have all reached p_expect (1: one counter word)
// p2p_own (in): the peer-to-peer exchange runs INSIDE the launch (workgroup 0 publishes p2p_own, awaits the p2p_n words p2p_peer[], opens the
// gate); inline_done (out): the launch does that - nothing to launch on an exchange stream
struct CfxXGate { int taken; unsigned* p_gate; unsigned p_expect; int p_count; unsigned* f_gate; unsigned f_expect; int needs_room;
                  int remote;      // remote (in): the reconstruction items' packets may sit in a peer GPU's memory
                  unsigned* p2p_own; const unsigned* const* p2p_peer; int p2p_n; int inline_done; };
struct PlanOp {
    int kind;   // 0 compress, 1 decompress, 2 all-gather on the side stream, 3 main stream waits for gather op `ref`, 4 ring hop,
                // 5 wait until flag `ref` has reached the plan's epoch, 6 set flag `ref` to the epoch,
                // 7 low-rank compress (codec = quantized, param = rank), 8 low-rank decompress,
                // 11 p2p sync: publish this rank's word, wait for the peers' (in stream order),
                // 10 = 9 without a collective (peers' packets read in place, one published word per rank and layer),
                // 9 exchange layer: compress (c) ; all-gather (comm, send, recv; comm NULL = none) ; reconstruct (g) - one launch on the
                //   main stream whose reconstruction group preloads its state and waits for the collective's arrival
    int codec, N, C, param, flags, batch;
    cfx_comp_item c[CFX_MAX_BATCH];
    cfx_decomp_item d[CFX_MAX_BATCH];     // kind 1: the items; kind 0: ride-along reconstruction items (n_ride of them)
    int n_ride;
    cfx_decomp_item g[CFX_MAX_BATCH];     // kind 0: gated reconstruction items (n_gated of them)
    int n_gated;
    void* ws;
    size_t ws_bytes;
    const void* q0[CFX_MAX_BATCH];        // kind 7: the start matrices
    // kind 2 / 3
    cfx_comm* comm;
    const void* send;
    void* recv;
    size_t bytes_per_rank;
    hipEvent_t ev_pre, ev_done;
    int ref;
    int pre_flag;         // kind 1: flag this reconstruction launch publishes FIRST (the epoch), or -1
    // kind 10 (exchange layer without a collective: peers' packets read in place through IPC mappings)
    unsigned* own_flag;                              // this rank's "packets of this layer complete" word (in memory the peers have mapped)
    const unsigned* peer_flag[CFX_P2P_MAX_PEERS];    // the peers' words
    int n_peers;
};
struct PipeSched;
struct cfx_plan {
    cfx_ctx* ctx;
    PlanOp* ops;
    int n, cap;
    hipStream_t side;     // exchange stream (created on first all-gather op, or the caller's: side_owned = false)
    bool side_owned;
    hipEvent_t ev_fork, ev_join;   // cfx_plan_run_async / cfx_plan_join
    int side_mode;        // 0: issue collectives on the main stream (no cross-stream events), 1: side stream, 2: prioritised side stream
    void* pipe_ws;        // cfx_plan_run_pipelined: two statistics workspaces of CFX_MAX_BATCH tensors each (stats of unit
    size_t pipe_ws_bytes; //   t runs beside the finalize of unit t-1)
    PipeSched* sched;          // unit schedule of the pipelined replay, built once (cfx_plan_finalize or the first replay of a range)
    unsigned* flags;      // exchange lane: n_flags flag words, a 64-byte line each (cfx_plan_flags)
    int n_flags;
    unsigned epoch;       // value the lane's flags take in the current replay (cfx_plan_run_lane advances it)
    unsigned* p2p_sink;   // a device word nobody reads: where the in-order form of a p2p exchange layer "opens its gate"
    int pipe_unit_layers; // cfx_plan_set_pipe_unit_layers: layers per unit of the pipelined replay (default 7)
};


#define CFX_PIPE_MAX_DQ 112    // reconstruction items per fused launch (a "unit" of up to 7 layers x 16 tensors; kernel arguments stay below 4 KB)
struct PipeUnit { int first_layer, n_layers, n_comp_items, n_dq_items; };
// The recognised schedule of an op range: built once per plan and range, replayed without host allocations, environment
// lookups or device allocations.
struct PipeSched {
    int first_op, n_ops, n_plan_ops;      // the range it was built for (and the plan size at that time)
    bool ok;                              // false: the range is not a sequence of 1-bit groups -> cfx_plan_run replays it
    int L, n_ag, N, C, U;
    int *comp_op, *deq_op, *ag_op, *ag_unit;
    PipeUnit* units;
};

// cfx_api.hip (the fused pipeline launch: cfx_absmean.hip), for cfx_plan.hip (hidden: not part of the ABI)
#define CFX_HIDDEN __attribute__((visibility("hidden")))
CFX_HIDDEN bool cfx_i_shape_ok(int codec, int N, int C, int param);
CFX_HIDDEN size_t cfx_i_ws_words(int codec, int N, int C);
// a zeroed block of CFX_MAX_BATCH * 64 u32 ticket words for ONE launch on `stream`; whoever draws a word's final value resets it to 0
CFX_HIDDEN unsigned* cfx_i_ticket_block(cfx_ctx* ctx, void* stream);
// CUs the queue of `stream` may use (a CU-masked stream: fewer than the device has)
CFX_HIDDEN int cfx_i_stream_cus(cfx_ctx* ctx, void* stream);
// 1: the codec has a one-launch exchange-layer form (reconstruction group gated on an external word)
CFX_HIDDEN bool cfx_i_has_xlayer_form(int codec);
CFX_HIDDEN int cfx_i_decompress_impl(cfx_ctx* ctx, int codec, int N, int C, int param, int batch, const cfx_decomp_item* items, void* stream,
                                     unsigned* pre, unsigned pre_val);
// `xg` != NULL: the gated items wait on an EXTERNAL gate (their packets are delivered by a collective behind this launch), see
// compress_impl in cfx_api.hip; xg->taken == 0 on return: only the compress part was launched
CFX_HIDDEN int cfx_i_compress_impl(cfx_ctx* ctx, int codec, int N, int C, int param, int flags, int batch, const cfx_comp_item* items,
                                   int n_ride, const cfx_decomp_item* ride, int n_gated, const cfx_decomp_item* gated,
                                   void* workspace, size_t workspace_bytes, void* stream, CfxXGate* xg = nullptr);
CFX_HIDDEN int cfx_i_launch_pipe(cfx_plan* p, hipStream_t s, int N, int C, const int* comp_op, const int* deq_op,
                                 const PipeUnit* dq, const PipeUnit* fin, const PipeUnit* st, int fin_parity, int st_parity,
                                 hipEvent_t done_ev);
#endif

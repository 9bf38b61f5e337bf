// libcfx.so - plan replay, library-owned RCCL communicator and the flag-synchronised exchange lane (host side of the C-ABI of
// include/cfx.h; the streaming kernels are in cfx_absmean / cfx_minmax / cfx_topk.hip, their C-ABI in cfx_api.hip, the low-rank chain in cfx_lowrank.hip).
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

typedef unsigned long long u64;
#define shape_ok cfx_i_shape_ok
#define ws_words cfx_i_ws_words
#define compress_impl cfx_i_compress_impl
#define launch_pipe cfx_i_launch_pipe
#define PIPE_MAX_DQ CFX_PIPE_MAX_DQ

// The ONLY process-global state besides what hangs off a cfx_ctx (documented in cfx.h): the entry points of the collective
// library most recently loaded by cfx_rccl_load.  A communicator keeps its own copy of the table it was created with.
static RcclApi g_rccl = {};

static void sched_free(PipeSched* sc) {
    if (!sc) return;
    delete[] sc->comp_op; delete[] sc->deq_op; delete[] sc->ag_op; delete[] sc->ag_unit; delete[] sc->units;
    delete sc;
}


// ---------------------------------------------------------------------------------------------------
// Exchange lane: cross-stream ordering by in-memory flags instead of events.
// Measured on MI355X / ROCm 7.2 (tools/lane_probe.hip): an event hop between two streams (record + stream-wait) costs ~14 us of
// idle queue time per hop, a hipStreamWriteValue32 / WaitValue32 pair ~6 us, a flag written by one stream's kernel and polled by
// the other stream's kernel ~1.7 us (plus the 1.5 us any dependent tiny kernel costs in its stream).  So the layer's exchange
// chain runs on its own (CU-masked) stream and talks to the compute stream only through flag words:
//     compute stream:   k_flag_set(ready, e) ; attention(own K,V) ; merge + wait(peer1 >= e) ; attention(peer 1) ; merge + wait(peer2 >= e) ...
//     exchange stream:  k_flag_wait(ready >= e) ; compress ; all-gather ; reconstruct peer 1 ; k_flag_set(peer1, e) ; reconstruct peer 2 ; ...
// Flags hold monotonic epochs (one per replay of the plan), one 64-byte line each, never reset.  A set kernel runs after the
// kernel in front of it has finished and released its stores (in-order stream), a waiting kernel's successor starts with an
// acquire: the data behind a flag needs no fences of its own.
// ---------------------------------------------------------------------------------------------------
#define FLAG_WORDS 16      // u32 words per flag: a 64-byte line each
__global__ void k_flag_set(unsigned* flag, unsigned v) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one wave: wait until the `count` consecutive words at `flag` have all reached v (count 1: lane 0 polls one word); true = gave up
__device__ __forceinline__ bool wave_wait_words(const unsigned* flag, unsigned v, int count, long long t0, long long timeout) {
    const int lane = threadIdx.x;
    if (count <= 1) {
        bool gave_up = false;
        if (lane == 0) {
            while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - v) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > timeout) { gave_up = true; break; }
            }
        }
        return __builtin_amdgcn_ballot_w64(gave_up) != 0;
    }
    for (;;) {
        bool behind = false;
        for (int i = lane; i < count; i += 64)
            behind |= (int)(__hip_atomic_load(flag + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - v) < 0;
        if (__builtin_amdgcn_ballot_w64(behind) == 0) return false;
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > timeout) return true;
    }
}
__global__ void k_flag_wait(const unsigned* flag, unsigned v, int count, unsigned* err, long long timeout) {
    const bool gave_up = wave_wait_words(flag, v, count, wall_clock64(), timeout);
    if (gave_up && threadIdx.x == 0 && err) (void)__hip_atomic_fetch_add(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// wait + set in one launch: what sits between "packets complete" and "packets arrived" when no collective kernel does
__global__ void k_flag_relay(const unsigned* wait, unsigned wv, int count, unsigned* set, unsigned sv, unsigned* err, long long timeout) {
    const bool gave_up = wave_wait_words(wait, wv, count, wall_clock64(), timeout);
    if (threadIdx.x != 0) return;
    if (gave_up && err) (void)__hip_atomic_fetch_add(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(set, sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Peer-to-peer exchange without a collective (cfx_plan_add_exchange_layer_p2p): every rank's packets stay in ITS memory, mapped into
// the peers (cfx_ipc_*); the reconstruction workgroups of a peer read them from there.  What travels is one word per rank and layer:
//   wait until this rank's launch has completed its packets  ->  publish (own_pub = epoch, visible to the peers)
//   wait until every peer has published the epoch            ->  open this launch's gate
// One wave: lane 0 does the local steps, lane p polls peer p's word (system-scope loads: the word lives in another GPU's memory).
struct PeerFlags { const unsigned* pub[CFX_P2P_MAX_PEERS]; };
// The epoch a rank publishes is (its own word) + 1, taken on the DEVICE: every rank executes an op equally often, so the n-th execution
// publishes n on every rank whatever the host rebuilt in between (a plan that is rebuilt - compact_reset, a new state arena - keeps
// counting where the words stand; a host-side counter restarted at 0 there and every wait of the next generation passed at once).  Only
// this rank's flag kernels of this op write the word, one at a time (stream order, or behind the launch whose packets they wait for).
// p_gate NULL: nothing to wait for locally (the in-order forms: the packets are complete by stream order).
__global__ void k_flag_exchange(const unsigned* p_gate, unsigned p_expect, int p_count, unsigned* own_pub, PeerFlags peers, int n_peers,
                                unsigned* f_gate, unsigned f_expect, unsigned* err, long long timeout) {
    const int lane = threadIdx.x;
    const long long t0 = wall_clock64();
    bool gave_up = false;
    unsigned epoch = 0;
    // (the own word is read BEFORE the wait: only this kernel writes it, and a system-scope load of uncached memory is a microsecond that
    // would otherwise sit between "packets complete" and "gate open")
    if (lane == 0) epoch = __hip_atomic_load(own_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;
    if (p_gate) gave_up = wave_wait_words(p_gate, p_expect, p_count, t0, timeout);
    if (lane == 0) {
        if (p_gate) {
            // one-launch form: the packets were stored WRITE-THROUGH and their stores had completed before the last arrival was counted:
            // publishing after having SEEN the count orders them before the word for anybody who reads the word first - no release fence
            // (a system-scope release writes back this XCD's whole L2 while the launch's state updates are streaming through it)
            __hip_atomic_store(own_pub, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            // in-order forms: the packets were written by the kernels in front of this one in the stream, with plain stores
            __hip_atomic_store(own_pub, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    epoch = (unsigned)__builtin_amdgcn_readfirstlane((int)epoch);
    if (lane < n_peers) {
        while ((int)(__hip_atomic_load(peers.pub[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > timeout) { gave_up = true; break; }
        }
    }
    if (__builtin_amdgcn_ballot_w64(gave_up) != 0 && lane == 0 && err) (void)__hip_atomic_fetch_add(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (lane == 0) __hip_atomic_store(f_gate, f_expect, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static int gate_check(cfx_ctx* ctx, const char* what) {
    if (ctx->gate_err && *(volatile unsigned*)ctx->gate_err) {
        char buf[200];
        snprintf(buf, sizeof(buf), "%s: an earlier gate / flag wait on this context timed out (cfx_gate_errors reads and clears the count)", what);
        return fail(ctx, CFX_ERR_GATE, buf);
    }
    return CFX_OK;
}

extern "C" {

cfx_plan* cfx_plan_create(cfx_ctx* ctx) {
    if (!ctx) return nullptr;
    cfx_plan* p = new cfx_plan();
    p->ctx = ctx;
    p->ops = nullptr;
    p->n = p->cap = 0;
    p->side = nullptr;
    p->side_owned = true;
    p->ev_fork = p->ev_join = nullptr;
    p->pipe_ws = nullptr;
    p->pipe_ws_bytes = 0;
    p->sched = nullptr;
    p->flags = nullptr;
    p->n_flags = 0;
    p->epoch = 0;
    p->p2p_sink = nullptr;
    // Default: collectives in order on the main stream.  Measured on MI355X / ROCm 7: one cross-stream event hop costs
    // ~10 us of idle queue time, two per layer (main->side, side->main) = +1.1 ms per 57-layer step, whereas the
    // in-order exchange adds 0.1 ms; a side stream only pays when >> 20 us of independent work can overlap (attention).
    p->side_mode = 0;
    p->pipe_unit_layers = 7;
    return p;
}

int cfx_plan_set_exchange_stream(cfx_plan* p, int mode) {
    if (!p) return CFX_ERR_NULL;
    if (mode < 0 || mode > 2 || p->side) return fail(p->ctx, CFX_ERR_BATCH, "plan: exchange stream mode must be 0..2 and set before the first all-gather op");
    p->side_mode = mode;
    return CFX_OK;
}

int cfx_plan_set_pipe_unit_layers(cfx_plan* p, int layers) {
    if (!p) return CFX_ERR_NULL;
    if (layers < 1 || layers > 7) return fail(p->ctx, CFX_ERR_BATCH, "plan: a unit of the pipelined replay holds 1..7 layers");
    p->pipe_unit_layers = layers;
    sched_free(p->sched);
    p->sched = nullptr;
    return CFX_OK;
}

int cfx_plan_use_exchange_stream(cfx_plan* p, void* stream) {
    if (!p || !stream) return CFX_ERR_NULL;
    if (p->side) return fail(p->ctx, CFX_ERR_BATCH, "plan: the exchange stream must be chosen before the first exchange op");
    p->side = (hipStream_t)stream;
    p->side_owned = false;
    p->side_mode = 1;
    return CFX_OK;
}

void cfx_plan_destroy(cfx_plan* p) {
    if (!p) return;
    for (int i = 0; i < p->n; ++i) {
        if (p->ops[i].ev_pre) (void)hipEventDestroy(p->ops[i].ev_pre);
        if (p->ops[i].ev_done) (void)hipEventDestroy(p->ops[i].ev_done);
    }
    if (p->side && p->side_owned) (void)hipStreamDestroy(p->side);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    if (p->pipe_ws) (void)hipFree(p->pipe_ws);
    if (p->flags) (void)hipFree(p->flags);
    if (p->p2p_sink) (void)hipFree(p->p2p_sink);
    sched_free(p->sched);
    delete[] p->ops;
    delete p;
}

// op kinds whose items' x operands are the activations a run re-points (cfx_plan_run_x): the 1-bit .. top-k compress, the low-rank compress, the exchange layers
static inline bool takes_activations(int kind) { return kind == 0 || kind == 7 || kind == 9 || kind == 10; }

static PlanOp* plan_push(cfx_plan* p) {
    if (p->n == p->cap) {
        const int ncap = p->cap ? p->cap * 2 : 64;
        PlanOp* no = new PlanOp[ncap];
        if (p->n) memcpy(no, p->ops, sizeof(PlanOp) * p->n);
        delete[] p->ops;
        p->ops = no;
        p->cap = ncap;
    }
    return &p->ops[p->n++];
}

int cfx_plan_add_compress_gated(cfx_plan* p, int codec, int N, int C, int param, int flags, int batch, const cfx_comp_item* items,
                                int n_ride, const cfx_decomp_item* ride, int n_gated, const cfx_decomp_item* gated,
                                void* workspace, size_t workspace_bytes) {
    if (!p) return CFX_ERR_NULL;
    if (n_gated < 0 || n_gated > CFX_MAX_BATCH || (n_gated && !gated)) return fail(p->ctx, CFX_ERR_BATCH, "plan: gated batch out of range");
    const int op = cfx_plan_add_compress_ex(p, codec, N, C, param, flags, batch, items, n_ride, ride, workspace, workspace_bytes);
    if (op < 0) return op;
    p->ops[op].n_gated = n_gated;
    if (n_gated) memcpy(p->ops[op].g, gated, sizeof(cfx_decomp_item) * n_gated);
    return op;
}

int cfx_plan_add_compress_ex(cfx_plan* p, int codec, int N, int C, int param, int flags, int batch, const cfx_comp_item* items,
                             int n_ride, const cfx_decomp_item* ride, void* workspace, size_t workspace_bytes) {
    if (!p || !items) return CFX_ERR_NULL;
    if (batch < 1 || batch > CFX_MAX_BATCH) return fail(p->ctx, CFX_ERR_BATCH, "plan: batch out of range");
    if (n_ride < 0 || n_ride > CFX_MAX_BATCH || (n_ride && !ride)) return fail(p->ctx, CFX_ERR_BATCH, "plan: ride-along batch out of range");
    if (n_ride && codec != CFX_CODEC_BINARY) return fail(p->ctx, CFX_ERR_CODEC, "plan: ride-along reconstruction items need the 1-bit codec");
    if (!shape_ok(codec, N, C, param)) return fail(p->ctx, CFX_ERR_SHAPE, "plan: bad codec/shape");
    PlanOp* o = plan_push(p);
    memset(o, 0, sizeof(*o));
    o->kind = 0; o->codec = codec; o->N = N; o->C = C; o->param = param; o->flags = flags; o->batch = batch;
    memcpy(o->c, items, sizeof(cfx_comp_item) * batch);
    o->n_ride = n_ride;
    if (n_ride) memcpy(o->d, ride, sizeof(cfx_decomp_item) * n_ride);
    o->ws = workspace; o->ws_bytes = workspace_bytes;
    return p->n - 1;
}

int cfx_plan_add_compress(cfx_plan* p, int codec, int N, int C, int param, int flags, int batch, const cfx_comp_item* items,
                          void* workspace, size_t workspace_bytes) {
    return cfx_plan_add_compress_ex(p, codec, N, C, param, flags, batch, items, 0, nullptr, workspace, workspace_bytes);
}

int cfx_plan_add_decompress(cfx_plan* p, int codec, int N, int C, int param, int batch, const cfx_decomp_item* items) {
    if (!p || !items) return CFX_ERR_NULL;
    if (batch < 1 || batch > CFX_MAX_BATCH) return fail(p->ctx, CFX_ERR_BATCH, "plan: batch out of range");
    if (!shape_ok(codec, N, C, param)) return fail(p->ctx, CFX_ERR_SHAPE, "plan: bad codec/shape");
    PlanOp* o = plan_push(p);
    memset(o, 0, sizeof(*o));
    o->kind = 1; o->codec = codec; o->N = N; o->C = C; o->param = param; o->batch = batch;
    o->pre_flag = -1;
    memcpy(o->d, items, sizeof(cfx_decomp_item) * batch);
    return p->n - 1;
}

int cfx_plan_add_lr_compress(cfx_plan* p, int quantized, int N, int C, int rank, int flags, int batch, const cfx_comp_item* items,
                             const void* const* init_q, void* workspace, size_t workspace_bytes) {
    if (!p || !items || !init_q) return CFX_ERR_NULL;
    if (batch < 1 || batch > CFX_MAX_BATCH) return fail(p->ctx, CFX_ERR_BATCH, "plan: batch out of range");
    if (!cfx_lr_packet_bytes(quantized, N, C, rank)) return fail(p->ctx, CFX_ERR_SHAPE, "plan: bad low-rank shape");
    PlanOp* o = plan_push(p);
    memset(o, 0, sizeof(*o));
    o->kind = 7; o->codec = quantized; o->N = N; o->C = C; o->param = rank; o->flags = flags; o->batch = batch;
    o->pre_flag = -1;
    memcpy(o->c, items, sizeof(cfx_comp_item) * batch);
    for (int i = 0; i < batch; ++i) o->q0[i] = init_q[i];
    o->ws = workspace; o->ws_bytes = workspace_bytes;
    return p->n - 1;
}

int cfx_plan_add_lr_decompress(cfx_plan* p, int quantized, int N, int C, int rank, int batch, const cfx_decomp_item* items,
                               void* workspace, size_t workspace_bytes) {
    if (!p || !items) return CFX_ERR_NULL;
    if (batch < 1 || batch > CFX_MAX_BATCH) return fail(p->ctx, CFX_ERR_BATCH, "plan: batch out of range");
    if (!cfx_lr_packet_bytes(quantized, N, C, rank)) return fail(p->ctx, CFX_ERR_SHAPE, "plan: bad low-rank shape");
    PlanOp* o = plan_push(p);
    memset(o, 0, sizeof(*o));
    o->kind = 8; o->codec = quantized; o->N = N; o->C = C; o->param = rank; o->batch = batch;
    o->pre_flag = -1;
    memcpy(o->d, items, sizeof(cfx_decomp_item) * batch);
    o->ws = workspace; o->ws_bytes = workspace_bytes;
    return p->n - 1;
}

// Exchange layer: compress ; all-gather ; reconstruct as ONE op (the in-order layer of the ring / patch gather schedules, reference
// ring.py:188-206 + 265-269, patchpara/fwd.py:108-137).  The reconstruction workgroups are launched WITH the compress group: they pull
// their state tiles into registers while the statistics chain and the collective run, and continue when the exchange stream - which
// waits for the launch's packets, issues the collective and then sets the launch's external gate - says the packets have arrived.
static int add_exchange_layer(cfx_plan* p, int codec, int N, int C, int param, int flags, int batch, const cfx_comp_item* items,
                              int n_recon, const cfx_decomp_item* recon, cfx_comm* comm, const void* send, void* recv, size_t bytes_per_rank,
                              void* workspace, size_t workspace_bytes, bool need_side);
int cfx_plan_add_exchange_layer(cfx_plan* p, int codec, int N, int C, int param, int flags, int batch, const cfx_comp_item* items,
                                int n_recon, const cfx_decomp_item* recon, cfx_comm* comm, const void* send, void* recv, size_t bytes_per_rank,
                                void* workspace, size_t workspace_bytes) {
    return add_exchange_layer(p, codec, N, C, param, flags, batch, items, n_recon, recon, comm, send, recv, bytes_per_rank, workspace, workspace_bytes, true);
}
// need_side: the op's exchange runs on the plan's exchange stream (a collective, or the gate relay) - the peer-to-peer form runs it inside
// the launch and needs none
static int add_exchange_layer(cfx_plan* p, int codec, int N, int C, int param, int flags, int batch, const cfx_comp_item* items,
                              int n_recon, const cfx_decomp_item* recon, cfx_comm* comm, const void* send, void* recv, size_t bytes_per_rank,
                              void* workspace, size_t workspace_bytes, bool need_side) {
    if (!p) return CFX_ERR_NULL;
    if (n_recon < 1 || n_recon > CFX_MAX_BATCH || !recon) return fail(p->ctx, CFX_ERR_BATCH, "plan: exchange layer needs 1..CFX_MAX_BATCH reconstruction items");
    if (codec < CFX_CODEC_BINARY || codec > CFX_CODEC_TOPK) return fail(p->ctx, CFX_ERR_CODEC, "plan: exchange layer: unknown codec");
    if (comm && (!send || !recv)) return fail(p->ctx, CFX_ERR_NULL, "plan: exchange layer: null send/recv");
    if (!p->side && need_side) {
        // the flag kernels poll: they need a hardware queue of their own (cfx.h, exchange lane) - a CU-masked stream has one
        void* xs = nullptr;
        int total = 0;
        (void)hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, p->ctx->device);
        const int rc = cfx_stream_create_masked(p->ctx, 0, total, &xs);
        if (rc != CFX_OK) return rc;
        p->side = (hipStream_t)xs;
        p->side_owned = true;
        if (p->side_mode == 0) p->side_mode = 1;
    }
    const int op = cfx_plan_add_compress(p, codec, N, C, param, flags, batch, items, workspace, workspace_bytes);
    if (op < 0) return op;
    PlanOp* o = &p->ops[op];
    o->kind = 9;
    o->n_gated = n_recon;
    memcpy(o->g, recon, sizeof(cfx_decomp_item) * n_recon);
    o->comm = comm; o->send = send; o->recv = recv; o->bytes_per_rank = bytes_per_rank;
    return op;
}

// The exchange layer without a collective: the peers' packets are READ IN PLACE from the peers' memory (recon items point into buffers
// opened with cfx_ipc_open), and one word per rank and layer says when they are complete (k_flag_exchange above).  Nothing but two tiny
// kernels ever runs on the exchange stream: there is no collective kernel that would have to find CUs beside the waiting workgroups.
int cfx_plan_add_exchange_layer_p2p(cfx_plan* p, int codec, int N, int C, int param, int flags, int batch, const cfx_comp_item* items,
                                    int n_recon, const cfx_decomp_item* recon, void* own_flag, int n_peers, const void* const* peer_flags,
                                    void* workspace, size_t workspace_bytes) {
    if (!p) return CFX_ERR_NULL;
    if (n_peers < 0 || n_peers > CFX_P2P_MAX_PEERS || (n_peers && !peer_flags) || !own_flag)
        return fail(p->ctx, CFX_ERR_BATCH, "plan: p2p exchange layer needs an own flag and 0..CFX_P2P_MAX_PEERS peer flags");
    for (int i = 0; i < n_peers; ++i)
        if (!peer_flags[i] || ((uintptr_t)peer_flags[i] & 3)) return fail(p->ctx, CFX_ERR_NULL, "plan: p2p exchange layer: null / misaligned peer flag");
    const int op = add_exchange_layer(p, codec, N, C, param, flags, batch, items, n_recon, recon, nullptr, nullptr, nullptr, 0, workspace, workspace_bytes, false);
    if (op < 0) return op;
    PlanOp* o = &p->ops[op];
    o->kind = 10;
    o->own_flag = (unsigned*)own_flag;
    o->n_peers = n_peers;
    for (int i = 0; i < n_peers; ++i) o->peer_flag[i] = (const unsigned*)peer_flags[i];
    if (!p->p2p_sink && hipMalloc((void**)&p->p2p_sink, 64) != hipSuccess) {
        (void)hipGetLastError();
        p->p2p_sink = nullptr;
        return fail(p->ctx, CFX_ERR_LAUNCH, "plan: p2p exchange layer: hipMalloc failed");
    }
    return op;
}

// Publish-and-wait as an op of its own (what the p2p exchange layer does between compress and reconstruction, for chains that keep
// their own launches: compress ; p2p_sync ; reconstruct peer 1 ; reconstruct peer 2 ; ... - the lane plan of compact_fwd): in stream order
// on the stream the range runs on.  Everything the ops BEFORE it wrote is complete when the word is published (kernel boundary), and
// the ops behind it start after every peer has published (their packets can be read in place through the IPC mappings).
int cfx_plan_add_p2p_sync(cfx_plan* p, void* own_flag, int n_peers, const void* const* peer_flags) {
    if (!p) return CFX_ERR_NULL;
    if (n_peers < 0 || n_peers > CFX_P2P_MAX_PEERS || (n_peers && !peer_flags) || !own_flag)
        return fail(p->ctx, CFX_ERR_BATCH, "plan: p2p sync needs an own flag and 0..CFX_P2P_MAX_PEERS peer flags");
    for (int i = 0; i < n_peers; ++i)
        if (!peer_flags[i] || ((uintptr_t)peer_flags[i] & 3)) return fail(p->ctx, CFX_ERR_NULL, "plan: p2p sync: null / misaligned peer flag");
    if (!p->p2p_sink && hipMalloc((void**)&p->p2p_sink, 64) != hipSuccess) {
        (void)hipGetLastError();
        p->p2p_sink = nullptr;
        return fail(p->ctx, CFX_ERR_LAUNCH, "plan: p2p sync: hipMalloc failed");
    }
    PlanOp* o = plan_push(p);
    memset(o, 0, sizeof(*o));
    o->kind = 11;
    o->pre_flag = -1;
    o->own_flag = (unsigned*)own_flag;
    o->n_peers = n_peers;
    for (int i = 0; i < n_peers; ++i) o->peer_flag[i] = (const unsigned*)peer_flags[i];
    return p->n - 1;
}

// ---- device memory shared between the processes of a node (dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0) ---------------------------------
int cfx_ipc_alloc(cfx_ctx* ctx, size_t bytes, void** ptr, void* handle64) {
    if (!ctx || !ptr || !handle64 || !bytes) return fail(ctx, CFX_ERR_NULL, "ipc_alloc: null");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C-ABI hands IPC handles around as 64 bytes");
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != ctx->device && hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, CFX_ERR_LAUNCH, "ipc_alloc: hipSetDevice failed");
    int rc = CFX_OK;
    void* d = nullptr;
    hipIpcMemHandle_t h;
    // Peers poll flag words in this memory and read packets from it WHILE the producing kernel is still running, and the same addresses
    // are rewritten every step.  Ordinary (coarse-grained) device memory is only promised coherent across devices at kernel boundaries:
    // a reader's L2 may keep last step's lines.  So the buffer is UNCACHED device memory (what RCCL allocates for the buffers its kernels
    // exchange through on gfx94x / gfx950), else fine-grained, else - with a note in the context - ordinary memory; the device code
    // uses write-through stores on the producer and system-scope loads on the readers whatever the kind.
    int kind = ctx->ipc_want;
    hipError_t e = hipErrorUnknown;
    if (kind == 2) { e = hipExtMallocWithFlags(&d, bytes, hipDeviceMallocUncached); if (e != hipSuccess) { (void)hipGetLastError(); d = nullptr; kind = 1; } }
    if (kind == 1) { e = hipExtMallocWithFlags(&d, bytes, hipDeviceMallocFinegrained); if (e != hipSuccess) { (void)hipGetLastError(); d = nullptr; kind = 0; } }
    if (kind == 0) e = hipMalloc(&d, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); rc = fail(ctx, CFX_ERR_LAUNCH, "ipc_alloc: device allocation failed"); }
    else if (hipMemset(d, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess || hipIpcGetMemHandle(&h, d) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(d);
        rc = fail(ctx, CFX_ERR_LAUNCH, "ipc_alloc: hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 is needed on hosts with dmabuf IPC only)");
    } else { *ptr = d; memcpy(handle64, &h, 64); ctx->ipc_kind = kind; }
    if (cur >= 0 && cur != ctx->device) (void)hipSetDevice(cur);
    return rc;
}
int cfx_ipc_memory_kind(cfx_ctx* ctx) { return ctx ? ctx->ipc_kind : CFX_ERR_NULL; }
int cfx_set_ipc_memory_kind(cfx_ctx* ctx, int kind) {
    if (!ctx) return CFX_ERR_NULL;
    if (kind < 0 || kind > 2) return fail(ctx, CFX_ERR_BATCH, "ipc memory kind must be 2 (uncached), 1 (fine-grained) or 0 (ordinary device memory)");
    ctx->ipc_want = kind;
    return CFX_OK;
}
int cfx_ipc_open(cfx_ctx* ctx, const void* handle64, void** ptr) {
    if (!ctx || !ptr || !handle64) return fail(ctx, CFX_ERR_NULL, "ipc_open: null");
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != ctx->device && hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, CFX_ERR_LAUNCH, "ipc_open: hipSetDevice failed");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    int rc = CFX_OK;
    if (hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); rc = fail(ctx, CFX_ERR_LAUNCH, "ipc_open: hipIpcOpenMemHandle failed"); }
    if (cur >= 0 && cur != ctx->device) (void)hipSetDevice(cur);
    return rc;
}
int cfx_ipc_close(cfx_ctx* ctx, void* ptr) {
    if (!ctx || !ptr) return fail(ctx, CFX_ERR_NULL, "ipc_close: null");
    return hipIpcCloseMemHandle(ptr) == hipSuccess ? CFX_OK : fail(ctx, CFX_ERR_LAUNCH, "hipIpcCloseMemHandle failed");
}
int cfx_ipc_free(cfx_ctx* ctx, void* ptr) {
    if (!ctx || !ptr) return fail(ctx, CFX_ERR_NULL, "ipc_free: null");
    return hipFree(ptr) == hipSuccess ? CFX_OK : fail(ctx, CFX_ERR_LAUNCH, "hipFree failed");
}

int cfx_plan_size(const cfx_plan* p) { return p ? p->n : CFX_ERR_NULL; }

// append a copy of a compress / decompress op of another plan (to build differently ordered schedules from one op set)
int cfx_plan_copy_op(cfx_plan* dst, const cfx_plan* src, int op) {
    if (!dst || !src) return CFX_ERR_NULL;
    if (op < 0 || op >= src->n || src->ops[op].kind > 1) return fail(dst->ctx, CFX_ERR_BATCH, "plan: op to copy must be a compress/decompress op");
    const PlanOp tmp = src->ops[op];
    PlanOp* o = plan_push(dst);
    *o = tmp;
    o->ev_pre = o->ev_done = nullptr;      // events belong to the op they were created for
    o->pre_flag = -1;                      // ... and flags to the plan
    return dst->n - 1;
}

static int plan_exchange_stream(cfx_plan* p);
int cfx_plan_add_all_gather(cfx_plan* p, cfx_comm* comm, const void* send, void* recv, size_t bytes_per_rank) {
    if (!p || !comm || !send || !recv) return CFX_ERR_NULL;
    const int rs = plan_exchange_stream(p);
    if (rs != CFX_OK) return rs;
    PlanOp* o = plan_push(p);
    memset(o, 0, sizeof(*o));
    o->kind = 2; o->comm = comm; o->send = send; o->recv = recv; o->bytes_per_rank = bytes_per_rank;
    if (hipEventCreateWithFlags(&o->ev_pre, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&o->ev_done, hipEventDisableTiming) != hipSuccess)
        return fail(p->ctx, CFX_ERR_LAUNCH, "plan: cannot create events");
    return p->n - 1;
}

static int plan_exchange_stream(cfx_plan* p) {
    if (!p->side && p->side_mode != 0) {
        hipError_t e;
        if (p->side_mode == 2) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            e = hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, hi);   // a prioritised stream gets its own HW queue
        } else {
            e = hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking);
        }
        if (e != hipSuccess) return fail(p->ctx, CFX_ERR_LAUNCH, "plan: cannot create the exchange stream");
    }
    return CFX_OK;
}

int cfx_plan_add_ring_hop(cfx_plan* p, cfx_comm* comm, const void* send, void* recv, size_t bytes) {
    if (!p || !comm || !send || !recv) return CFX_ERR_NULL;
    const int rs = plan_exchange_stream(p);
    if (rs != CFX_OK) return rs;
    PlanOp* o = plan_push(p);
    memset(o, 0, sizeof(*o));
    o->kind = 4; o->comm = comm; o->send = send; o->recv = recv; o->bytes_per_rank = bytes;
    if (hipEventCreateWithFlags(&o->ev_pre, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&o->ev_done, hipEventDisableTiming) != hipSuccess)
        return fail(p->ctx, CFX_ERR_LAUNCH, "plan: cannot create events");
    return p->n - 1;
}

int cfx_plan_set_input(cfx_plan* p, int op, int item, const void* x) {
    if (!p || !x) return CFX_ERR_NULL;
    const int kind = (op >= 0 && op < p->n) ? p->ops[op].kind : -1;
    if ((kind != 0 && kind != 7 && kind != 9 && kind != 10) || item < 0 || item >= p->ops[op].batch)
        return fail(p->ctx, CFX_ERR_BATCH, "plan: set_input needs a compress op and an item of its batch");
    if (!AL16(x)) return fail(p->ctx, CFX_ERR_ALIGN, "plan: pointers must be 16-byte aligned");
    p->ops[op].c[item].x = x;
    return CFX_OK;
}

// cfx_plan_run with the activations of the range's FIRST compress op re-pointed first (one host call per layer phase)
int cfx_plan_run_x(cfx_plan* p, int first_op, int n_ops, const void* const* xs, int n_xs, void* stream) {
    if (!p) return CFX_ERR_NULL;
    if (first_op < 0 || n_ops < 0 || first_op + n_ops > p->n) return fail(p->ctx, CFX_ERR_BATCH, "plan: op range out of bounds");
    if (n_xs > 0) {
        if (!xs) return CFX_ERR_NULL;
        int op = first_op;
        while (op < first_op + n_ops && !takes_activations(p->ops[op].kind)) ++op;
        if (op == first_op + n_ops || p->ops[op].batch != n_xs) return fail(p->ctx, CFX_ERR_BATCH, "plan: run_x needs a compress op with n_xs items in the range");
        for (int i = 0; i < n_xs; ++i) {
            if (!xs[i]) return fail(p->ctx, CFX_ERR_NULL, "plan: null activation");
            if (!AL16(xs[i])) return fail(p->ctx, CFX_ERR_ALIGN, "plan: pointers must be 16-byte aligned");
            p->ops[op].c[i].x = xs[i];
        }
    }
    return cfx_plan_run(p, first_op, n_ops, stream);
}

int cfx_plan_add_wait(cfx_plan* p, int gather_op) {
    if (!p) return CFX_ERR_NULL;
    if (gather_op < 0 || gather_op >= p->n || (p->ops[gather_op].kind != 2 && p->ops[gather_op].kind != 4))
        return fail(p->ctx, CFX_ERR_BATCH, "plan: wait target is not an exchange op");
    PlanOp* o = plan_push(p);
    memset(o, 0, sizeof(*o));
    o->kind = 3; o->ref = gather_op;
    return p->n - 1;
}

// One hop of the ring relay (reference xfuser/compact/ring.py:193-195: RingComm.send_recv + commit): send `bytes` to rank+1,
// receive `bytes` from rank-1, as one grouped pair on `s`.
static int ring_hop(cfx_comm* c, const void* send, void* recv, size_t bytes, hipStream_t s) {
    if (!c->api.Send || !c->api.Recv || !c->api.GroupStart || !c->api.GroupEnd) return -1;
    const int nxt = (c->rank + 1) % c->nranks, prv = (c->rank + c->nranks - 1) % c->nranks;
    int r = c->api.GroupStart();
    if (r == 0) r = c->api.Send(send, bytes, /*ncclUint8*/ 1, nxt, c->comm, s);
    if (r == 0) r = c->api.Recv(recv, bytes, 1, prv, c->comm, s);
    const int e = c->api.GroupEnd();
    return r ? r : e;
}

// publish this rank's word for op `o`, wait for the peers' (k_flag_exchange) on `s`; p_gate != NULL: first wait until *p_gate has reached
// p_expect (the launch's packets are complete); afterwards *f_gate = f_expect
static int launch_flag_exchange(cfx_plan* p, const PlanOp* o, hipStream_t s, const unsigned* p_gate, unsigned p_expect, int p_count, unsigned* f_gate,
                                unsigned f_expect, const char* what) {
    PeerFlags pf;
    memset(&pf, 0, sizeof(pf));
    for (int q = 0; q < o->n_peers; ++q) pf.pub[q] = o->peer_flag[q];
    hipLaunchKernelGGL(k_flag_exchange, dim3(1), dim3(64), 0, s, p_gate, p_expect, p_count, o->own_flag, pf, o->n_peers, f_gate, f_expect,
                       p->ctx->gate_err, p->ctx->gate_timeout);
    return check_launch(p->ctx, what);
}

// `inline_exchange`: exchange ops run in order on `stream` itself whatever the plan's exchange-stream mode (cfx_plan_run_async:
// the whole range already runs on the exchange stream).
static int plan_run_impl(cfx_plan* p, int first_op, int n_ops, void* stream, bool inline_exchange) {
    if (!p) return CFX_ERR_NULL;
    if (first_op < 0 || n_ops < 0 || first_op + n_ops > p->n) return fail(p->ctx, CFX_ERR_BATCH, "plan: op range out of bounds");
    hipStream_t main_s = (hipStream_t)stream;
    { const int ge = gate_check(p->ctx, "plan run"); if (ge != CFX_OK) return ge; }
    const int side_mode = inline_exchange ? 0 : p->side_mode;
    for (int i = first_op; i < first_op + n_ops; ++i) {
        PlanOp* o = &p->ops[i];
        int rc = CFX_OK;
        switch (o->kind) {
            case 0: rc = compress_impl(p->ctx, o->codec, o->N, o->C, o->param, o->flags, o->batch, o->c, o->n_ride, o->d, o->n_gated, o->g, o->ws, o->ws_bytes, stream); break;
            case 1:
                rc = cfx_i_decompress_impl(p->ctx, o->codec, o->N, o->C, o->param, o->batch, o->d, stream,
                                           o->pre_flag >= 0 ? p->flags + (size_t)o->pre_flag * FLAG_WORDS : nullptr, p->epoch);
                break;
            case 2:
            case 4: {
                // exchange stream picks up after everything enqueued so far on the main stream (the packets are complete)
                hipStream_t xs = side_mode ? p->side : main_s;
                if (side_mode && (hipEventRecord(o->ev_pre, main_s) != hipSuccess || hipStreamWaitEvent(p->side, o->ev_pre, 0) != hipSuccess))
                    return fail(p->ctx, CFX_ERR_LAUNCH, "plan: event ordering failed");
                int r;
                if (o->kind == 2) r = o->comm->api.AllGather(o->send, o->recv, o->bytes_per_rank, /*ncclUint8*/ 1, o->comm->comm, xs);
                else r = ring_hop(o->comm, o->send, o->recv, o->bytes_per_rank, xs);
                if (r != 0) {
                    char buf[200];
                    snprintf(buf, sizeof(buf), "%s: %s", o->kind == 2 ? "ncclAllGather" : "ring hop (ncclSend/ncclRecv)",
                             o->comm->api.GetErrorString ? o->comm->api.GetErrorString(r) : "error");
                    return fail(p->ctx, CFX_ERR_LAUNCH, buf);
                }
                if (side_mode && hipEventRecord(o->ev_done, p->side) != hipSuccess) return fail(p->ctx, CFX_ERR_LAUNCH, "plan: event record failed");
            } break;
            case 3:
                if (side_mode && hipStreamWaitEvent(main_s, p->ops[o->ref].ev_done, 0) != hipSuccess) return fail(p->ctx, CFX_ERR_LAUNCH, "plan: wait failed");
                break;
            case 5:
                hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(64), 0, main_s, (const unsigned*)(p->flags + (size_t)o->ref * FLAG_WORDS), p->epoch, 1,
                                   p->ctx->gate_err, p->ctx->gate_timeout);
                rc = check_launch(p->ctx, "flag wait launch");
                break;
            case 6:
                hipLaunchKernelGGL(k_flag_set, dim3(1), dim3(64), 0, main_s, p->flags + (size_t)o->ref * FLAG_WORDS, p->epoch);
                rc = check_launch(p->ctx, "flag set launch");
                break;
            case 7: rc = cfx_lr_compress_batch(p->ctx, o->codec, o->N, o->C, o->param, o->flags, o->batch, o->c, o->q0, o->ws, o->ws_bytes, stream); break;
            case 8: rc = cfx_lr_decompress_batch(p->ctx, o->codec, o->N, o->C, o->param, o->batch, o->d, o->ws, o->ws_bytes, stream); break;
            case 11:
                rc = launch_flag_exchange(p, o, main_s, nullptr, 0u, 1, p->p2p_sink, 1u, "p2p sync launch");
                break;
            case 9:
            case 10: {
                // Which form: ONE launch (the reconstruction group launched with the compress group, gated on a word the exchange stream
                // sets) needs a codec that has it, an exchange stream that is not the run stream, and hardware queues of their own for the
                // two (cfx_hw_queues_ok).  The legacy NULL stream serialises with every BLOCKING stream - a CU-masked exchange stream is one:
                // its flag kernel would wait for the very launch it is meant to release - so beside the NULL stream only a non-blocking
                // exchange stream will do.  Everything else runs the same work in order on the run stream: compress ; exchange ; reconstruct.
                // (the peer-to-peer form needs NO second stream: workgroup 0 of the launch publishes / awaits the flag words itself)
                const bool inline_p2p = o->kind == 10 && cfx_i_has_xlayer_form(o->codec);
                bool own_stream = inline_p2p || (p->side && (hipStream_t)stream != p->side && !inline_exchange && cfx_i_has_xlayer_form(o->codec) &&
                                                 (cfx_hw_queues_ok() || p->ctx->allow_shared_queues));
                if (own_stream && !inline_p2p && stream == nullptr) {
                    unsigned sf = 0;
                    if (hipStreamGetFlags(p->side, &sf) != hipSuccess) { (void)hipGetLastError(); sf = 0; }
                    own_stream = (sf & hipStreamNonBlocking) != 0;
                }
                CfxXGate xg;
                memset(&xg, 0, sizeof(xg));
                xg.needs_room = o->comm && o->comm->nranks > 1;
                xg.remote = o->kind == 10 && o->n_peers > 0;
                if (inline_p2p) { xg.p2p_own = o->own_flag; xg.p2p_peer = o->peer_flag; xg.p2p_n = o->n_peers; }
                if (own_stream) {
                    rc = compress_impl(p->ctx, o->codec, o->N, o->C, o->param, o->flags, o->batch, o->c, 0, nullptr, o->n_gated, o->g, o->ws, o->ws_bytes, stream, &xg);
                } else if (o->kind == 9 && !o->comm) {
                    // nothing moves and nobody else publishes: the reconstruction items' packets are this launch's own - the ordinary gated launch
                    rc = compress_impl(p->ctx, o->codec, o->N, o->C, o->param, o->flags, o->batch, o->c, 0, nullptr, o->n_gated, o->g, o->ws, o->ws_bytes, stream);
                    break;
                } else {
                    rc = compress_impl(p->ctx, o->codec, o->N, o->C, o->param, o->flags, o->batch, o->c, 0, nullptr, 0, nullptr, o->ws, o->ws_bytes, stream);
                }
                if (rc != CFX_OK) break;
                if (!xg.taken) {
                    // in order on the run stream (also: a shape / stream without the one-launch form - compress_impl launched the compress only)
                    if (o->kind == 10) rc = launch_flag_exchange(p, o, main_s, nullptr, 0u, 1, p->p2p_sink, 1u, "p2p exchange layer: flag exchange launch (in order)");
                    else if (o->comm) {
                        const int r = o->comm->api.AllGather(o->send, o->recv, o->bytes_per_rank, /*ncclUint8*/ 1, o->comm->comm, main_s);
                        if (r != 0) {
                            char buf[200];
                            snprintf(buf, sizeof(buf), "ncclAllGather (exchange layer, in order): %s", o->comm->api.GetErrorString ? o->comm->api.GetErrorString(r) : "error");
                            return fail(p->ctx, CFX_ERR_LAUNCH, buf);
                        }
                    }
                    if (rc == CFX_OK) rc = cfx_i_decompress_impl(p->ctx, o->codec, o->N, o->C, o->param, o->n_gated, o->g, stream, nullptr, 0u);
                    break;
                }
                // one launch: what sits on the exchange stream between "packets complete" and "packets arrived"
                if (xg.inline_done) {
                    // nothing: the launch runs the peer-to-peer exchange itself
                } else if (o->kind == 10) {
                    rc = launch_flag_exchange(p, o, p->side, xg.p_gate, xg.p_expect, xg.p_count, xg.f_gate, xg.f_expect, "p2p exchange layer: flag exchange launch");
                } else if (!o->comm) {
                    hipLaunchKernelGGL(k_flag_relay, dim3(1), dim3(64), 0, p->side, (const unsigned*)xg.p_gate, xg.p_expect, xg.p_count, xg.f_gate, xg.f_expect,
                                       p->ctx->gate_err, p->ctx->gate_timeout);
                    rc = check_launch(p->ctx, "exchange layer: flag relay launch");
                } else {
                    // with a communicator: wait kernel ; ncclAllGather ; set kernel - the same three enqueues at every world size (a one-rank
                    // in-place all-gather enqueues nothing)
                    hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(64), 0, p->side, (const unsigned*)xg.p_gate, xg.p_expect, xg.p_count, p->ctx->gate_err, p->ctx->gate_timeout);
                    rc = check_launch(p->ctx, "exchange layer: flag wait launch");
                    if (rc != CFX_OK) break;
                    const int r = o->comm->api.AllGather(o->send, o->recv, o->bytes_per_rank, /*ncclUint8*/ 1, o->comm->comm, p->side);
                    if (r != 0) {
                        char buf[200];
                        snprintf(buf, sizeof(buf), "ncclAllGather: %s", o->comm->api.GetErrorString ? o->comm->api.GetErrorString(r) : "error");
                        return fail(p->ctx, CFX_ERR_LAUNCH, buf);
                    }
                    hipLaunchKernelGGL(k_flag_set, dim3(1), dim3(64), 0, p->side, xg.f_gate, xg.f_expect);
                    rc = check_launch(p->ctx, "exchange layer: flag set launch");
                }
            } break;
        }
        if (rc != CFX_OK) return rc;
    }
    return CFX_OK;
}

int cfx_plan_run(cfx_plan* p, int first_op, int n_ops, void* stream) { return plan_run_impl(p, first_op, n_ops, stream, false); }

// The whole op range on the plan's EXCHANGE stream, forked off `main_stream` and joined back later: everything a layer's exchange
// does - compress, collective, reconstruction - runs beside what the caller enqueues on `main_stream` in between (the local
// attention block).  cfx_plan_join makes `main_stream` wait for the range.  The activations (xs, see cfx_plan_run_x) must stay
// alive until the join.
int cfx_plan_run_async(cfx_plan* p, int first_op, int n_ops, const void* const* xs, int n_xs, void* main_stream) {
    if (!p) return CFX_ERR_NULL;
    if (!p->side) return fail(p->ctx, CFX_ERR_BATCH, "plan: run_async needs an exchange stream (cfx_plan_use_exchange_stream, or mode 1 / 2 and an exchange op)");
    if (!p->ev_fork && (hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) != hipSuccess))
        return fail(p->ctx, CFX_ERR_LAUNCH, "plan: cannot create events");
    if (first_op < 0 || n_ops < 0 || first_op + n_ops > p->n) return fail(p->ctx, CFX_ERR_BATCH, "plan: op range out of bounds");
    if (n_xs > 0) {
        if (!xs) return CFX_ERR_NULL;
        int op = first_op;
        while (op < first_op + n_ops && !takes_activations(p->ops[op].kind)) ++op;
        if (op == first_op + n_ops || p->ops[op].batch != n_xs) return fail(p->ctx, CFX_ERR_BATCH, "plan: run_async needs a compress op with n_xs items in the range");
        for (int i = 0; i < n_xs; ++i) {
            if (!xs[i] || !AL16(xs[i])) return fail(p->ctx, CFX_ERR_ALIGN, "plan: activations must be non-null and 16-byte aligned");
            p->ops[op].c[i].x = xs[i];
        }
    }
    if (hipEventRecord(p->ev_fork, (hipStream_t)main_stream) != hipSuccess || hipStreamWaitEvent(p->side, p->ev_fork, 0) != hipSuccess)
        return fail(p->ctx, CFX_ERR_LAUNCH, "plan: fork failed");
    const int rc = plan_run_impl(p, first_op, n_ops, (void*)p->side, true);
    if (hipEventRecord(p->ev_join, p->side) != hipSuccess) return fail(p->ctx, CFX_ERR_LAUNCH, "plan: join event failed");
    return rc;
}

int cfx_plan_join(cfx_plan* p, void* main_stream) {
    if (!p) return CFX_ERR_NULL;
    if (!p->ev_join) return fail(p->ctx, CFX_ERR_BATCH, "plan: join without run_async");
    return hipStreamWaitEvent((hipStream_t)main_stream, p->ev_join, 0) == hipSuccess ? CFX_OK : fail(p->ctx, CFX_ERR_LAUNCH, "plan: join failed");
}

// ---- exchange lane (see the flag kernels above) -------------------------------------------------------------------------
void* cfx_plan_flags(cfx_plan* p, int n) {
    if (!p || n < 1 || n > 4096) return nullptr;
    if (p->flags) return p->n_flags >= n ? (void*)p->flags : nullptr;          // one block per plan, sized once
    if (cfx_prepare(p->ctx) != CFX_OK) return nullptr;                         // the error word the waits report to
    void* m = nullptr;
    if (hipMalloc(&m, (size_t)n * FLAG_WORDS * sizeof(unsigned)) != hipSuccess || hipMemset(m, 0, (size_t)n * FLAG_WORDS * sizeof(unsigned)) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        if (m) (void)hipFree(m);
        fail(p->ctx, CFX_ERR_LAUNCH, "plan: cannot allocate the flag block");
        return nullptr;
    }
    p->flags = (unsigned*)m;
    p->n_flags = n;
    return m;
}

static int plan_add_flag_op(cfx_plan* p, int kind, int flag) {
    if (!p) return CFX_ERR_NULL;
    if (!p->flags || flag < 0 || flag >= p->n_flags) return fail(p->ctx, CFX_ERR_BATCH, "plan: flag index out of range (cfx_plan_flags first)");
    PlanOp* o = plan_push(p);
    memset(o, 0, sizeof(*o));
    o->kind = kind; o->ref = flag;
    return p->n - 1;
}
// The reconstruction op `op` publishes flag `flag` as the FIRST thing its launch does: the launch in front of it in the stream has
// finished by then (in-order stream), so this is "set flag after the previous op" without a launch of its own.
int cfx_plan_set_pre_flag(cfx_plan* p, int op, int flag) {
    if (!p) return CFX_ERR_NULL;
    if (op < 0 || op >= p->n || p->ops[op].kind != 1) return fail(p->ctx, CFX_ERR_BATCH, "plan: a pre-flag belongs to a reconstruction op");
    if (!p->flags || flag < 0 || flag >= p->n_flags) return fail(p->ctx, CFX_ERR_BATCH, "plan: flag index out of range (cfx_plan_flags first)");
    p->ops[op].pre_flag = flag;
    return CFX_OK;
}
int cfx_plan_add_flag_wait(cfx_plan* p, int flag) { return plan_add_flag_op(p, 5, flag); }
int cfx_plan_add_flag_set(cfx_plan* p, int flag) { return plan_add_flag_op(p, 6, flag); }

unsigned cfx_plan_epoch(const cfx_plan* p) { return p ? p->epoch : 0u; }

// Advance the epoch and publish "the activations exist" (flag `ready_flag`) behind whatever the caller has enqueued on the compute
// stream so far - the kernels that produce K, V.  Split from cfx_plan_run_lane so that the caller can enqueue the local attention
// block BETWEEN the two: the chain's host issue (a dozen launches) then overlaps GPU work instead of delaying it.
static int queues_check(cfx_ctx* ctx, const char* what) {
    if (cfx_hw_queues_ok() || ctx->allow_shared_queues) return CFX_OK;
    char buf[240];
    snprintf(buf, sizeof(buf), "%s: flag-ordered streams need hardware queues of their own - set GPU_MAX_HW_QUEUES (e.g. 8) before HIP "
             "initialises, or cfx_set_allow_shared_queues", what);
    return fail(ctx, CFX_ERR_QUEUES, buf);
}

int cfx_plan_lane_begin(cfx_plan* p, int ready_flag, void* compute_stream, unsigned* epoch_out) {
    if (!p) return CFX_ERR_NULL;
    { const int qc = queues_check(p->ctx, "plan: exchange lane"); if (qc != CFX_OK) return qc; }
    if (!p->flags || ready_flag < 0 || ready_flag >= p->n_flags) return fail(p->ctx, CFX_ERR_BATCH, "plan: ready flag index out of range");
    ++p->epoch;
    if (epoch_out) *epoch_out = p->epoch;
    hipLaunchKernelGGL(k_flag_set, dim3(1), dim3(64), 0, (hipStream_t)compute_stream, p->flags + (size_t)ready_flag * FLAG_WORDS, p->epoch);
    return check_launch(p->ctx, "ready flag launch");
}

int cfx_plan_run_lane(cfx_plan* p, int first_op, int n_ops, const void* const* xs, int n_xs, int ready_flag, void* compute_stream,
                      unsigned* epoch_out) {
    if (!p) return CFX_ERR_NULL;
    if (!p->side) return fail(p->ctx, CFX_ERR_BATCH, "plan: run_lane needs an exchange stream (cfx_plan_use_exchange_stream)");
    { const int qc = queues_check(p->ctx, "plan: exchange lane"); if (qc != CFX_OK) return qc; }
    if (first_op < 0 || n_ops < 0 || first_op + n_ops > p->n) return fail(p->ctx, CFX_ERR_BATCH, "plan: op range out of bounds");
    if (!p->flags || ready_flag < 0 || ready_flag >= p->n_flags) return fail(p->ctx, CFX_ERR_BATCH, "plan: ready flag index out of range");
    if (n_xs > 0) {
        if (!xs) return CFX_ERR_NULL;
        int op = first_op;
        while (op < first_op + n_ops && !takes_activations(p->ops[op].kind)) ++op;
        if (op == first_op + n_ops || p->ops[op].batch != n_xs) return fail(p->ctx, CFX_ERR_BATCH, "plan: run_lane needs a compress op with n_xs items in the range");
        for (int i = 0; i < n_xs; ++i) {
            if (!xs[i] || !AL16(xs[i])) return fail(p->ctx, CFX_ERR_ALIGN, "plan: activations must be non-null and 16-byte aligned");
            p->ops[op].c[i].x = xs[i];
        }
    }
    if (compute_stream) {
        const int rb = cfx_plan_lane_begin(p, ready_flag, compute_stream, epoch_out);
        if (rb != CFX_OK) return rb;
    } else if (epoch_out) *epoch_out = p->epoch;       // cfx_plan_lane_begin was called separately
    return plan_run_impl(p, first_op, n_ops, (void*)p->side, true);
}

int cfx_flag_set(cfx_ctx* ctx, void* flag, unsigned value, void* stream) {
    if (!ctx || !flag) return fail(ctx, CFX_ERR_NULL, "flag_set: null");
    hipLaunchKernelGGL(k_flag_set, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned*)flag, value);
    return check_launch(ctx, "flag set launch");
}

int cfx_flag_wait(cfx_ctx* ctx, const void* flag, unsigned value, void* stream) {
    if (!ctx || !flag) return fail(ctx, CFX_ERR_NULL, "flag_wait: null");
    if (!ctx->gate_err && cfx_prepare(ctx) != CFX_OK) return CFX_ERR_LAUNCH;
    const int ge = gate_check(ctx, "flag wait");
    if (ge != CFX_OK) return ge;
    hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned*)flag, value, 1, ctx->gate_err, ctx->gate_timeout);
    return check_launch(ctx, "flag wait launch");
}

// A stream whose kernels may only use CU-mask bits [first_cu, first_cu + n_cus).  On MI355X (8 XCDs x 32 CUs) bit i selects CU
// i / 8 of XCD i % 8, consecutive CUs of an XCD alternating over its shader engines (tools/lane_probe.hip): a contiguous bit range
// is the same number of CUs on every XCD, i.e. an even share of every L2 and of the fabric.
int cfx_stream_create_masked(cfx_ctx* ctx, int first_cu, int n_cus, void** stream) {
    if (!ctx || !stream) return fail(ctx, CFX_ERR_NULL, "stream_create_masked: null");
    int total = 0;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != ctx->device && hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, CFX_ERR_LAUNCH, "stream_create_masked: hipSetDevice failed");
    (void)hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, ctx->device);
    int rc = CFX_OK;
    if (total <= 0 || total > 512 || first_cu < 0 || n_cus < 1 || first_cu + n_cus > total) rc = fail(ctx, CFX_ERR_BATCH, "stream_create_masked: CU range outside the device");
    else {
        uint32_t mask[16] = {0};
        for (int b = first_cu; b < first_cu + n_cus; ++b) mask[b >> 5] |= 1u << (b & 31);
        hipStream_t s = nullptr;
        if (hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask) != hipSuccess) { (void)hipGetLastError(); rc = fail(ctx, CFX_ERR_LAUNCH, "hipExtStreamCreateWithCUMask failed"); }
        else *stream = (void*)s;
    }
    if (cur >= 0 && cur != ctx->device) (void)hipSetDevice(cur);
    return rc;
}

int cfx_stream_destroy(cfx_ctx* ctx, void* stream) {
    if (!ctx || !stream) return fail(ctx, CFX_ERR_NULL, "stream_destroy: null");
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? CFX_OK : fail(ctx, CFX_ERR_LAUNCH, "hipStreamDestroy failed");
}

// Software-pipelined replay of a 1-bit exchange step.  The op range must be a sequence of "groups"
//     k x compress [BINARY, no cache update]   { all-gather }*   k x decompress [BINARY]            (k >= 1, one shape)
// i.e. layers whose packets travel in one collective (k = 1: what the ring gather schedule builds; bench.py groups
// several layers per all-gather: fewer, larger collectives).  Consecutive whole groups are merged into UNITS of up to
// 7 layers (as many as fit 112 reconstruction items and 16 compress items: < 4 KB of kernel arguments), and the range is
// replayed on ONE stream as
// launch slots t = 0 .. U+1:
//     { all-gathers of unit t-2 }  ;  K_t = [dequant(unit t-2) | finalize(unit t-1) | stats(unit t)]
// where K_t is ONE k_binary_pipe launch: the latency-bound stats / finalize work of later layers runs underneath the
// bandwidth-bound reconstruction of earlier ones, and a launch is long enough (7 layers = ~0.9 GB of traffic) to
// amortise its ramp and tail (measured on MI355X, FLUX step: 1.82 ms in order, 1.35 ms one layer per launch, see
// DESIGN.md section 3 for the unit size; a two-stream version of the same idea loses to the ~10 us cross-stream event hops
// and to the slowdown of the small kernels under contention).  Results are bit-identical to cfx_plan_run (same device
// code per group of workgroups).  The statistics workspaces are plan-owned (two, alternating by unit); the ops' own
// workspaces are not used here.  A group too large for one unit, or any other op sequence, is replayed by cfx_plan_run.
// Recognise the group pattern of ops [first_op, first_op + n_ops), merging whole groups into units, and size the plan-owned
// statistics workspaces.  Everything that allocates or reads the environment happens HERE, once.
static int plan_build_sched(cfx_plan* p, int first_op, int n_ops) {
    sched_free(p->sched);
    p->sched = nullptr;
    const int end = first_op + n_ops;
    const int cap = n_ops > 0 ? n_ops : 1;
    PipeSched* sc = new PipeSched();
    sc->first_op = first_op; sc->n_ops = n_ops; sc->n_plan_ops = p->n;
    sc->comp_op = new int[cap]; sc->deq_op = new int[cap]; sc->ag_op = new int[cap]; sc->ag_unit = new int[cap];
    sc->units = new PipeUnit[cap];
    int* comp_op = sc->comp_op; int* deq_op = sc->deq_op; int* ag_op = sc->ag_op; int* ag_unit = sc->ag_unit;
    PipeUnit* units = sc->units;
    const int unit_layers = p->pipe_unit_layers;
    int L = 0, n_ag = 0, N = 0, C = 0, U = 0;
    bool ok = n_ops > 0;
    for (int i = first_op; ok && i < end;) {
        int k = 0, ncomp = 0, ndq = 0;
        const int ag0 = n_ag;
        while (i < end && p->ops[i].kind == 0) {
            const PlanOp* c = &p->ops[i];
            if (c->codec != CFX_CODEC_BINARY || (c->flags & CFX_FLAG_UPDATE_CACHE) || c->n_ride || c->n_gated) { ok = false; break; }
            if (L + k == 0) { N = c->N; C = c->C; }
            if (c->N != N || c->C != C) { ok = false; break; }
            ncomp += c->batch;
            comp_op[L + k++] = i++;
        }
        if (!ok || k == 0) { ok = false; break; }
        while (i < end && (p->ops[i].kind == 2 || p->ops[i].kind == 3)) {
            if (p->ops[i].kind == 2) ag_op[n_ag++] = i;
            ++i;
        }
        if (i < end && p->ops[i].kind == 4) { ok = false; break; }          // relay hops: in-order replay only
        for (int m = 0; m < k; ++m, ++i) {
            if (i >= end || p->ops[i].kind != 1 || p->ops[i].codec != CFX_CODEC_BINARY || p->ops[i].N != N || p->ops[i].C != C) { ok = false; break; }
            ndq += p->ops[i].batch;
            deq_op[L + m] = i;
        }
        if (!ok) break;
        if (ncomp > CFX_MAX_BATCH || ndq > PIPE_MAX_DQ) { ok = false; break; }      // a group must fit one launch
        // extend the current unit with this group if it still fits, else start a new unit
        if (U > 0 && units[U - 1].n_layers + k <= unit_layers && units[U - 1].n_comp_items + ncomp <= CFX_MAX_BATCH &&
            units[U - 1].n_dq_items + ndq <= PIPE_MAX_DQ) {
            units[U - 1].n_layers += k; units[U - 1].n_comp_items += ncomp; units[U - 1].n_dq_items += ndq;
        } else {
            units[U].first_layer = L; units[U].n_layers = k; units[U].n_comp_items = ncomp; units[U].n_dq_items = ndq;
            ++U;
        }
        for (int a = ag0; a < n_ag; ++a) ag_unit[a] = U - 1;
        L += k;
    }
    sc->ok = ok; sc->L = L; sc->n_ag = n_ag; sc->N = N; sc->C = C; sc->U = U;
    p->sched = sc;
    if (!ok) return CFX_OK;
    const size_t need = 2 * (size_t)CFX_MAX_BATCH * ws_words(CFX_CODEC_BINARY, N, C) * sizeof(u64);
    if (need > p->pipe_ws_bytes) {
        if (p->pipe_ws) (void)hipFree(p->pipe_ws);
        p->pipe_ws = nullptr; p->pipe_ws_bytes = 0;
        if (hipMalloc(&p->pipe_ws, need) != hipSuccess) { (void)hipGetLastError(); return fail(p->ctx, CFX_ERR_LAUNCH, "plan: cannot allocate the statistics workspaces"); }
        p->pipe_ws_bytes = need;
    }
    return CFX_OK;
}

int cfx_plan_finalize(cfx_plan* p) {
    if (!p) return CFX_ERR_NULL;
    if (cfx_prepare(p->ctx) != CFX_OK) return CFX_ERR_LAUNCH;      // ticket blocks of the in-order compress launches
    return plan_build_sched(p, 0, p->n);
}

int cfx_plan_run_pipelined(cfx_plan* p, int first_op, int n_ops, void* stream) {
    if (!p) return CFX_ERR_NULL;
    if (first_op < 0 || n_ops < 0 || first_op + n_ops > p->n) return fail(p->ctx, CFX_ERR_BATCH, "plan: op range out of bounds");
    if (!p->sched || p->sched->first_op != first_op || p->sched->n_ops != n_ops || p->sched->n_plan_ops != p->n) {
        const int rcb = plan_build_sched(p, first_op, n_ops);     // first replay of this range (or the plan grew): build once
        if (rcb != CFX_OK) return rcb;
    }
    const PipeSched* sc = p->sched;
    if (!sc->ok) return cfx_plan_run(p, first_op, n_ops, stream);
    const int *comp_op = sc->comp_op, *deq_op = sc->deq_op, *ag_op = sc->ag_op, *ag_unit = sc->ag_unit;
    const PipeUnit* units = sc->units;
    const int n_ag = sc->n_ag, N = sc->N, C = sc->C, U = sc->U;
    hipStream_t s = (hipStream_t)stream;
    auto unit = [&](int u) -> const PipeUnit* { return (u >= 0 && u < U) ? &units[u] : nullptr; };
    // Exchange stream mode 0: collectives in order on `stream`, right before the launch that consumes them.
    // Modes 1 / 2: one more unit of look-ahead; the collectives of unit u are issued on the exchange stream as soon as
    // the launch holding finalize(u) is queued and run UNDERNEATH the next launch; `stream` waits for them (an event that
    // has normally fired long before) only in front of the launch that reconstructs unit u.
    const int d = (p->side_mode != 0 && p->side && n_ag > 0) ? 1 : 0;
    auto all_gather = [&](const PlanOp* o, hipStream_t on) -> int {
        if (o->comm->api.AllGather(o->send, o->recv, o->bytes_per_rank, /*ncclUint8*/ 1, o->comm->comm, on) != 0)
            return fail(p->ctx, CFX_ERR_LAUNCH, "ncclAllGather failed");
        return CFX_OK;
    };
    int rc = CFX_OK, next_ag = 0, next_wait = 0;
    for (int t = 0; rc == CFX_OK && t <= U + 1 + d; ++t) {
        if (d == 0) {
            while (rc == CFX_OK && next_ag < n_ag && ag_unit[next_ag] + 2 <= t) rc = all_gather(&p->ops[ag_op[next_ag++]], s);
        } else {
            while (rc == CFX_OK && next_wait < n_ag && ag_unit[next_wait] + 3 <= t) {
                if (hipStreamWaitEvent(s, p->ops[ag_op[next_wait++]].ev_done, 0) != hipSuccess) rc = fail(p->ctx, CFX_ERR_LAUNCH, "plan: wait failed");
            }
        }
        if (rc != CFX_OK) break;
        // overlapped mode: the collectives of unit t-1 may start when THIS launch (which holds finalize(t-1)) has finished;
        // the first of them lends its event to the launch
        hipEvent_t done_ev = nullptr;
        if (d == 1 && next_ag < n_ag && ag_unit[next_ag] + 1 <= t) done_ev = p->ops[ag_op[next_ag]].ev_pre;
        bool launched = false;
        if (unit(t - 2 - d) || unit(t - 1) || unit(t)) {
            rc = launch_pipe(p, s, N, C, comp_op, deq_op, unit(t - 2 - d), unit(t - 1), unit(t), (t - 1) & 1, t & 1, done_ev);
            launched = true;
        }
        if (d == 1 && rc == CFX_OK && done_ev) {
            if ((!launched && hipEventRecord(done_ev, s) != hipSuccess) || hipStreamWaitEvent(p->side, done_ev, 0) != hipSuccess)
                rc = fail(p->ctx, CFX_ERR_LAUNCH, "plan: event ordering failed");
            while (rc == CFX_OK && next_ag < n_ag && ag_unit[next_ag] + 1 <= t) {
                PlanOp* o = &p->ops[ag_op[next_ag++]];
                rc = all_gather(o, p->side);
                if (rc == CFX_OK && hipEventRecord(o->ev_done, p->side) != hipSuccess) rc = fail(p->ctx, CFX_ERR_LAUNCH, "plan: event record failed");
            }
        }
    }
    return rc;
}

// ---- communicator -----------------------------------------------------------------------------------------------------
int cfx_rccl_load(const char* path) {
    if (g_rccl.handle && (!path || !path[0] || !strcmp(path, g_rccl.path))) return CFX_OK;    // already loaded (same library)
    void* h = nullptr;
    if (path && path[0]) h = dlopen(path, RTLD_NOW | RTLD_NOLOAD);
    if (!h && path && path[0]) h = dlopen(path, RTLD_NOW);
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (int i = 0; i < 2 && !h && !(path && path[0]); ++i) h = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
    for (int i = 0; i < 2 && !h && !(path && path[0]); ++i) h = dlopen(names[i], RTLD_NOW);
    if (!h) return CFX_ERR_NULL;
    RcclApi a = {};
    a.GetUniqueId = (int (*)(cfx_nccl_uid*))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(cfx_nccl_comm*, int, cfx_nccl_uid, int))dlsym(h, "ncclCommInitRank");
    a.AllGather = (int (*)(const void*, void*, size_t, int, cfx_nccl_comm, hipStream_t))dlsym(h, "ncclAllGather");
    a.CommDestroy = (int (*)(cfx_nccl_comm))dlsym(h, "ncclCommDestroy");
    a.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    a.Send = (int (*)(const void*, size_t, int, int, cfx_nccl_comm, hipStream_t))dlsym(h, "ncclSend");
    a.Recv = (int (*)(void*, size_t, int, int, cfx_nccl_comm, hipStream_t))dlsym(h, "ncclRecv");
    a.GroupStart = (int (*)(void))dlsym(h, "ncclGroupStart");
    a.GroupEnd = (int (*)(void))dlsym(h, "ncclGroupEnd");
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy) return CFX_ERR_NULL;
    a.handle = h;
    snprintf(a.path, sizeof(a.path), "%s", (path && path[0]) ? path : "");
    g_rccl = a;          // communicators created from now on use this library; existing ones keep the table they were made with
    return CFX_OK;
}

int cfx_comm_unique_id(cfx_ctx* ctx, void* out128) {
    if (!ctx || !out128) return CFX_ERR_NULL;
    if (!g_rccl.handle) return fail(ctx, CFX_ERR_NULL, "RCCL not loaded: call cfx_rccl_load first");
    cfx_nccl_uid id;
    const int r = g_rccl.GetUniqueId(&id);
    if (r != 0) return fail(ctx, CFX_ERR_LAUNCH, "ncclGetUniqueId failed");
    memcpy(out128, &id, 128);
    return CFX_OK;
}

cfx_comm* cfx_comm_create(cfx_ctx* ctx, const void* id128, int nranks, int rank) {
    if (!ctx || !id128 || !g_rccl.handle) return nullptr;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != ctx->device) (void)hipSetDevice(ctx->device);       // ncclCommInitRank binds the communicator to the current device
    cfx_nccl_uid id;
    memcpy(&id, id128, 128);
    cfx_comm* c = new cfx_comm();
    c->ctx = ctx; c->nranks = nranks; c->rank = rank; c->comm = nullptr;
    c->api = g_rccl;
    const int r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (cur >= 0 && cur != ctx->device) (void)hipSetDevice(cur);   // the caller's current device is left as it was
    if (r != 0) {
        char buf[200];
        snprintf(buf, sizeof(buf), "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error");
        fail(ctx, CFX_ERR_LAUNCH, buf);
        delete c;
        return nullptr;
    }
    return c;
}

void cfx_comm_destroy(cfx_comm* c) {
    if (!c) return;
    if (c->comm && c->api.CommDestroy) (void)c->api.CommDestroy(c->comm);
    delete c;
}

int cfx_comm_all_gather(cfx_comm* c, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
    if (!c || !send || !recv) return CFX_ERR_NULL;
    const int r = c->api.AllGather(send, recv, bytes_per_rank, 1, c->comm, (hipStream_t)stream);
    return r == 0 ? CFX_OK : fail(c->ctx, CFX_ERR_LAUNCH, "ncclAllGather failed");
}

int cfx_comm_ring_hop(cfx_comm* c, const void* send, void* recv, size_t bytes, void* stream) {
    if (!c || !send || !recv) return CFX_ERR_NULL;
    const int r = ring_hop(c, send, recv, bytes, (hipStream_t)stream);
    return r == 0 ? CFX_OK : fail(c->ctx, CFX_ERR_LAUNCH, "ring hop (ncclSend / ncclRecv) failed");
}

}  // extern "C"

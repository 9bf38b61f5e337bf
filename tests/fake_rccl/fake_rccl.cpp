// libcfx_fake_rccl.so - TEST INFRASTRUCTURE, never shipped and never loaded by the product on its own.
//
// A stand-in for the handful of RCCL entry points libcfx.so resolves at run time (cfx_rccl_load(path)), so that the native
// exchange paths - cfx_plan_add_all_gather / cfx_plan_add_ring_hop, the exchange-stream modes, cfx_plan_run_pipelined with
// world > 1 - can be exercised on ONE GPU (this pool has one GPU per box).  Two modes (env CFX_FAKE_RCCL_MODE):
//   threads  (default) : a communicator group of `nranks` ranks living in ONE process, one host thread per rank.  A
//                        collective rendez-vous' on a host barrier, then every rank enqueues on ITS stream: wait for the
//                        events the other ranks recorded on their streams at the call (their send buffers are complete),
//                        copy.  Semantics of ncclAllGather / grouped ncclSend + ncclRecv, stream-ordered like RCCL.
//   loopback           : only ONE rank of the group exists; every peer is that rank (an all-gather replicates the send
//                        buffer into all `nranks` slots, a ring hop receives what it sends) - the "8 logical ranks looped back
//                        on one GPU" measurement protocol of tools/overlap_bench.py.
// Build: hipcc -O2 -fPIC -shared tests/fake_rccl/fake_rccl.cpp -o tests/fake_rccl/libcfx_fake_rccl.so  (tests/fake_rccl/build.py)
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <map>
#include <mutex>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

// loopback all-gather: one launch replicates the send buffer into all slots (a real ncclAllGather is one enqueue too)
__global__ void k_replicate(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16, int copies) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const uint4 v = src[i];
    for (int r = 0; r < copies; ++r) dst[(size_t)r * n16 + i] = v;
}

// the same copy with the register footprint of RCCL's kernel on gfx950 (rcclGenericKernel: 256 threads, 261-280 VGPRs - read from
// librccl's code object): CFX_FAKE_RCCL_FAT=1.  What matters to the tests is WHERE such a workgroup can be placed, not what it does.
__global__ __launch_bounds__(256) void k_replicate_fat(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16, int copies) {
    asm volatile("v_mov_b32 v247, 0\n\tv_accvgpr_write_b32 a31, 0" ::: "v247", "a31");       // 248 VGPRs + 32 AGPRs
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const uint4 v = src[i];
    for (int r = 0; r < copies; ++r) dst[(size_t)r * n16 + i] = v;
}

// CFX_FAKE_RCCL_FAT=2: a whole SIMD's register file per wave (512): such a workgroup only fits a CU that holds nothing else
__global__ __launch_bounds__(256) void k_replicate_huge(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16, int copies) {
    asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255");
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const uint4 v = src[i];
    for (int r = 0; r < copies; ++r) dst[(size_t)r * n16 + i] = v;
}

namespace {
struct Pending { const void* send; void* recv; size_t bytes; int peer_send, peer_recv; };
struct Group {
    int nranks = 0;
    bool loopback = false;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    std::vector<const void*> send;     // per rank, current collective
    std::vector<size_t> bytes;
    std::vector<hipEvent_t> ready;     // per rank: recorded on the caller's stream at the call
    std::vector<hipEvent_t> done;      // per rank: recorded after the rank's copies (it has finished reading its peers' buffers)
    std::vector<std::vector<Pending>> p2p;   // per rank: the sends / recvs of the open group
};
struct Comm { Group* g; int rank; bool in_group = false; hipStream_t group_stream = nullptr; };
std::mutex g_mu;
std::map<uint64_t, Group*> g_groups;
uint64_t g_next_id = 1;

// host barrier: returns when all ranks of the group have arrived (threads mode)
void rendezvous(Group* g) {
    std::unique_lock<std::mutex> lk(g->mu);
    const long gen = g->generation;
    if (++g->arrived == g->nranks) { g->arrived = 0; ++g->generation; g->cv.notify_all(); }
    else g->cv.wait(lk, [&] { return g->generation != gen; });
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } fakeUid;

int ncclGetUniqueId(fakeUid* id) {
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id, 0, sizeof(*id));
    const uint64_t v = g_next_id++;
    memcpy(id->internal, &v, sizeof(v));
    memcpy(id->internal + 8, "cfxfake", 8);
    return 0;
}

int ncclCommInitRank(void** comm, int nranks, fakeUid id, int rank) {
    uint64_t key;
    memcpy(&key, id.internal, sizeof(key));
    const char* mode = getenv("CFX_FAKE_RCCL_MODE");
    Group* g;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_groups.find(key);
        if (it == g_groups.end()) {
            g = new Group();
            g->nranks = nranks;
            g->loopback = mode && !strcmp(mode, "loopback");
            g->send.assign(nranks, nullptr);
            g->bytes.assign(nranks, 0);
            g->ready.assign(nranks, nullptr);
            g->done.assign(nranks, nullptr);
            g->p2p.assign(nranks, {});
            g_groups[key] = g;
        } else g = it->second;
        if (!g->ready[rank] && hipEventCreateWithFlags(&g->ready[rank], hipEventDisableTiming) != hipSuccess) return 1;
        if (!g->done[rank] && hipEventCreateWithFlags(&g->done[rank], hipEventDisableTiming) != hipSuccess) return 1;
    }
    if (rank < 0 || rank >= nranks || g->nranks != nranks) return 4;
    Comm* c = new Comm();
    c->g = g; c->rank = rank;
    *comm = c;
    return 0;
}

int ncclCommDestroy(void* comm) { delete (Comm*)comm; return 0; }
const char* ncclGetErrorString(int r) { return r == 0 ? "no error" : "fake rccl error"; }

int ncclAllGather(const void* send, void* recv, size_t count, int /*datatype: bytes*/, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    Group* g = c->g;
    if (g->loopback) {
        if ((count & 15) == 0 && (((uintptr_t)send | (uintptr_t)recv) & 15) == 0) {
            const size_t n16 = count / 16;
            const char* fat = getenv("CFX_FAKE_RCCL_FAT");
            if (fat && fat[0] == '2')
                hipLaunchKernelGGL(k_replicate_huge, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, stream, (uint4*)recv, (const uint4*)send, n16, g->nranks);
            else if (fat && fat[0] == '1')
                hipLaunchKernelGGL(k_replicate_fat, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, stream, (uint4*)recv, (const uint4*)send, n16, g->nranks);
            else
                hipLaunchKernelGGL(k_replicate, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, stream, (uint4*)recv, (const uint4*)send, n16, g->nranks);
            return hipGetLastError() == hipSuccess ? 0 : 1;
        }
        for (int r = 0; r < g->nranks; ++r)
            if (hipMemcpyAsync((char*)recv + (size_t)r * count, send, count, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
        return 0;
    }
    g->send[c->rank] = send;
    g->bytes[c->rank] = count;
    if (hipEventRecord(g->ready[c->rank], stream) != hipSuccess) return 1;    // my send buffer is complete at this point of my stream
    rendezvous(g);                                                          // every rank has published its buffer and event
    for (int r = 0; r < g->nranks; ++r) {
        if (g->bytes[r] != count) return 4;
        if (r != c->rank && hipStreamWaitEvent(stream, g->ready[r], 0) != hipSuccess) return 1;
        if (hipMemcpyAsync((char*)recv + (size_t)r * count, g->send[r], count, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
    }
    // like the real collective, the op completes on a rank's stream only when every peer has read that rank's send buffer
    if (hipEventRecord(g->done[c->rank], stream) != hipSuccess) return 1;
    rendezvous(g);
    for (int r = 0; r < g->nranks; ++r)
        if (r != c->rank && hipStreamWaitEvent(stream, g->done[r], 0) != hipSuccess) return 1;
    rendezvous(g);                                                          // nobody re-records an event another rank still has to wait on
    return 0;
}

int ncclGroupStart(void) { return 0; }

// Grouped point-to-point: the calls between ncclGroupStart / ncclGroupEnd of ONE communicator are collected and executed
// at ncclGroupEnd (this stand-in supports one communicator per group, which is what a ring hop uses).
static thread_local Comm* t_open = nullptr;
int ncclSend(const void* send, size_t count, int, int peer, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    c->g->p2p[c->rank].push_back(Pending{send, nullptr, count, peer, -1});
    c->group_stream = stream; t_open = c;
    return 0;
}
int ncclRecv(void* recv, size_t count, int, int peer, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    c->g->p2p[c->rank].push_back(Pending{nullptr, recv, count, -1, peer});
    c->group_stream = stream; t_open = c;
    return 0;
}
int ncclGroupEnd(void) {
    Comm* c = t_open;
    t_open = nullptr;
    if (!c) return 0;
    Group* g = c->g;
    hipStream_t stream = c->group_stream;
    auto& mine = g->p2p[c->rank];
    if (g->loopback) {     // every peer is me: a receive from `peer` gets what I send (to anyone), in posting order
        size_t si = 0;
        for (auto& r : mine) {
            if (r.peer_recv < 0) continue;
            while (si < mine.size() && mine[si].peer_send < 0) ++si;
            if (si == mine.size()) return 4;
            if (hipMemcpyAsync(r.recv, mine[si].send, r.bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
            ++si;
        }
        mine.clear();
        return 0;
    }
    if (hipEventRecord(g->ready[c->rank], stream) != hipSuccess) return 1;
    rendezvous(g);
    for (auto& r : mine) {
        if (r.peer_recv < 0) continue;
        const int src = r.peer_recv;
        // the matching send of `src` to me: k-th receive from src pairs with src's k-th send to me
        int k = 0;
        for (auto& q : mine) { if (&q == &r) break; if (q.peer_recv == src) ++k; }
        const Pending* s = nullptr;
        for (auto& q : g->p2p[src]) if (q.peer_send == c->rank && k-- == 0) { s = &q; break; }
        if (!s || s->bytes != r.bytes) return 4;
        if (src != c->rank && hipStreamWaitEvent(stream, g->ready[src], 0) != hipSuccess) return 1;
        if (hipMemcpyAsync(r.recv, s->send, r.bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
    }
    if (hipEventRecord(g->done[c->rank], stream) != hipSuccess) return 1;
    rendezvous(g);
    for (int r = 0; r < g->nranks; ++r)
        if (r != c->rank && hipStreamWaitEvent(stream, g->done[r], 0) != hipSuccess) return 1;
    rendezvous(g);
    mine.clear();
    return 0;
}
}  // extern "C"

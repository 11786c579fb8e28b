// Built-in transports of sph_slab_step (include/chsph_mi.h, sph_slab_transport).  Included by sph_solver.hip after sph_ctx.
//
//   RCCL        one process per GPU: grouped ncclSend / ncclRecv between slab neighbours over the direct xGMI links, ncclAllReduce for
//               the residual / wrench / time-step reductions, everything enqueued on the stream the library hands over.
//   in-process  the ranks are threads of ONE process (contexts on one GPU or on several with peer access): the receiver pulls the
//               sender's buffer with hipMemcpyAsync on ITS stream behind an event the sender recorded on its own -- stream-ordered like
//               RCCL (nothing waits for a GPU; host threads only meet to hand over pointers and events), which is what lets one GPU
//               exercise the ordering of the overlapped halo path (pack / copy / unpack on the second stream against the force waves on
//               the first) without a host synchronisation hiding a missing dependency.
//
// The reference has no counterpart (single device).
#pragma once
#include <atomic>
#include <chrono>
#include <mutex>
#include <set>
#include <thread>

namespace {
// every transport this library hands out, so that sph_slab_transport_free / sph_destroy can tell a live one from a stale pointer
std::mutex g_transport_mutex;
std::set<const void*> g_live_transports;

struct TransportBase {
    sph_slab_transport vt;
    sph_ctx* c;
    int kind;  // 1 RCCL, 2 in-process
};

// ---- RCCL, bound with dlopen / dlsym when the first transport (or unique id) is asked for: a process that already holds an RCCL
// (PyTorch ships its own) keeps using that one, and a process that never asks loads none --------------------------------------
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    const char* (*GetErrorString)(ncclResult_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    void* handle = nullptr;
    std::string rccl_path, hip_path;
    bool ok = false, mixed = false;
};
const RcclApi& rccl_api() {
    static RcclApi A = [] {
        RcclApi a{};
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);  // one that is already in the process (torch's), else the ROCm one
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return a;
        a.handle = h;
#define CHSPH_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name))
        CHSPH_SYM(GetUniqueId, "ncclGetUniqueId"); CHSPH_SYM(CommInitRank, "ncclCommInitRank"); CHSPH_SYM(CommDestroy, "ncclCommDestroy");
        CHSPH_SYM(GroupStart, "ncclGroupStart"); CHSPH_SYM(GroupEnd, "ncclGroupEnd"); CHSPH_SYM(Send, "ncclSend"); CHSPH_SYM(Recv, "ncclRecv");
        CHSPH_SYM(GetErrorString, "ncclGetErrorString"); CHSPH_SYM(AllReduce, "ncclAllReduce");
#undef CHSPH_SYM
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Send && a.Recv && a.GetErrorString && a.AllReduce;
        // the RCCL picked must drive the HIP runtime this library drives: it sits beside that runtime's libamdhip64 (torch/lib or /opt/rocm/lib)
        a.rccl_path = module_of((const void*)a.Send);
        a.hip_path = module_of((const void*)&hipGetDeviceCount);
        if (a.ok && (runtime_census(nullptr) > 1 || dir_of(a.rccl_path) != dir_of(a.hip_path))) a.ok = false, a.mixed = true;
        return a;
    }();
    return A;
}
struct RcclTransport : TransportBase {
    ncclComm_t comm;
    uint32_t* d_counts;  // [0..1] send, [2..3] receive
    uint32_t* h_counts;  // pinned
};
#define NCCLCK(call)                                                                                       \
    do {                                                                                                   \
        ncclResult_t r_ = (call);                                                                          \
        if (r_ != ncclSuccess) { set_error(T->c, "RCCL: %s (%s)", R.GetErrorString(r_), #call); return 1; } \
    } while (0)
// the ranks to the left / right of this one (-1: none); a ring when the slabs lie along a periodic x axis (three ranks or more: sph_slab_configure)
void rccl_peers(const RcclTransport* T, int peer[2]) {
    const int r = T->vt.rank, n = T->vt.nranks;
    if (slab_is_ring(T->c)) { peer[0] = (r + n - 1) % n; peer[1] = (r + 1) % n; return; }
    peer[0] = r > 0 ? r - 1 : -1;
    peer[1] = r < n - 1 ? r + 1 : -1;
}
int rccl_exchange(void* user, const void* const send[2], const size_t sb[2], void* const recv[2], const size_t rb[2], int on_device, void* stream) {
    RcclTransport* T = (RcclTransport*)user;
    const RcclApi& R = rccl_api();
    if (!on_device) { set_error(T->c, "RCCL transport: host buffers are not supported"); return 1; }
    int peer[2];
    rccl_peers(T, peer);
    NCCLCK(R.GroupStart());
    for (int sd = 0; sd < 2; sd++) {
        if (peer[sd] < 0) continue;
        if (rb[sd]) NCCLCK(R.Recv(recv[sd], rb[sd], ncclUint8, peer[sd], T->comm, (hipStream_t)stream));
        if (sb[sd]) NCCLCK(R.Send(send[sd], sb[sd], ncclUint8, peer[sd], T->comm, (hipStream_t)stream));
    }
    NCCLCK(R.GroupEnd());
    return 0;
}
int rccl_exchange_counts(void* user, const uint32_t send[2], uint32_t recv[2]) {
    RcclTransport* T = (RcclTransport*)user;
    const RcclApi& R = rccl_api();
    hipStream_t st = T->c->stream;
    int peer[2];
    rccl_peers(T, peer);
    T->h_counts[0] = send[0]; T->h_counts[1] = send[1]; T->h_counts[2] = 0; T->h_counts[3] = 0;
    if (hipMemcpyAsync(T->d_counts, T->h_counts, 16, hipMemcpyHostToDevice, st) != hipSuccess) { set_error(T->c, "RCCL transport: count upload failed"); return 1; }
    NCCLCK(R.GroupStart());
    for (int sd = 0; sd < 2; sd++) {
        if (peer[sd] < 0) continue;
        NCCLCK(R.Recv(T->d_counts + 2 + sd, 1, ncclUint32, peer[sd], T->comm, st));
        NCCLCK(R.Send(T->d_counts + sd, 1, ncclUint32, peer[sd], T->comm, st));
    }
    NCCLCK(R.GroupEnd());
    if (hipMemcpyAsync(T->h_counts + 2, T->d_counts + 2, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        set_error(T->c, "RCCL transport: count download failed");
        return 1;
    }
    recv[0] = T->h_counts[2]; recv[1] = T->h_counts[3];
    return 0;
}
int rccl_allreduce(void* user, void* buf, size_t nwords, int op, int on_device, void* stream) {
    RcclTransport* T = (RcclTransport*)user;
    const RcclApi& R = rccl_api();
    if (!on_device) { set_error(T->c, "RCCL transport: host buffers are not supported"); return 1; }
    if (!nwords) return 0;
    switch (op) {
        case SPH_SLAB_REDUCE_MAX_U32: NCCLCK(R.AllReduce(buf, buf, nwords, ncclUint32, ncclMax, T->comm, (hipStream_t)stream)); return 0;
        case SPH_SLAB_REDUCE_SUM_I32: NCCLCK(R.AllReduce(buf, buf, nwords, ncclInt32, ncclSum, T->comm, (hipStream_t)stream)); return 0;
        case SPH_SLAB_REDUCE_MIN_F32: NCCLCK(R.AllReduce(buf, buf, nwords, ncclFloat32, ncclMin, T->comm, (hipStream_t)stream)); return 0;
    }
    set_error(T->c, "RCCL transport: unknown reduction %d", op);
    return 1;
}
#undef NCCLCK
void rccl_transport_release(RcclTransport* T) {
    (void)hipSetDevice(T->c->device);
    if (T->d_counts) (void)hipFree(T->d_counts);
    if (T->h_counts) (void)hipHostFree(T->h_counts);
    (void)rccl_api().CommDestroy(T->comm);
    delete T;
}

// ---- in-process transport ------------------------------------------------------------------------------------------------------
constexpr int INPROC_MAX_RANKS = 16;
struct InprocChannel {  // one direction between two slab neighbours
    std::atomic<uint64_t> posted{0}, consumed{0};    // device messages: sequence numbers of the sender / the receiver
    const void* ptr = nullptr;
    size_t bytes = 0;
    hipEvent_t ready = nullptr;  // recorded by the sender on its stream once the buffer is final
    hipEvent_t done = nullptr;   // recorded by the receiver on its stream behind its copy
    std::atomic<uint64_t> cposted{0}, cconsumed{0};  // host counts
    uint32_t count = 0;
};
struct InprocSlot {  // all-reduce: one per rank
    std::atomic<uint64_t> posted{0}, read{0};
    void* ptr = nullptr;
    int device = 0;
    hipEvent_t ready = nullptr, read_done = nullptr;
};
}  // namespace
struct sph_slab_group {
    int nranks = 0;
    InprocChannel toRight[INPROC_MAX_RANKS], toLeft[INPROC_MAX_RANKS];  // toRight[r]: r -> r + 1, toLeft[r]: r -> r - 1
    InprocSlot slot[INPROC_MAX_RANKS];
    std::atomic<int> failed{0};
    std::atomic<int> members{0};
    double timeout_s = 120.0;
};
namespace {
struct InprocTransport : TransportBase {
    sph_slab_group* g;
    uint64_t seq = 0, cseq = 0, aseq = 0;
    uint32_t* tmp = nullptr;
    size_t tmp_words = 0;
    uint32_t peer_mask = 0;
};
struct InprocPtrs { const uint32_t* p[INPROC_MAX_RANKS]; };
__global__ void k_inproc_allreduce(InprocPtrs src, int nranks, size_t nwords, int op, uint32_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwords) return;
    uint32_t acc = src.p[0][i];
    for (int r = 1; r < nranks; r++) {
        const uint32_t v = src.p[r][i];
        if (op == SPH_SLAB_REDUCE_MAX_U32) acc = v > acc ? v : acc;
        else if (op == SPH_SLAB_REDUCE_SUM_I32) acc = acc + v;  // two's complement: the same bits as a signed sum
        else acc = __uint_as_float(v) < __uint_as_float(acc) ? v : acc;
    }
    out[i] = acc;
}
// host threads meet here to hand over pointers and events; nobody waits for a GPU
bool inproc_wait(sph_slab_group* g, const std::atomic<uint64_t>& v, uint64_t want) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (v.load(std::memory_order_acquire) < want) {
        if (g->failed.load(std::memory_order_relaxed)) return false;
        if (++spins > 2000) {
            std::this_thread::yield();
            if ((spins & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > g->timeout_s) { g->failed.store(1); return false; }
        }
    }
    return true;
}
#define INPROC_FAIL(...)                                      \
    do {                                                      \
        set_error(T->c, __VA_ARGS__);                         \
        T->g->failed.store(1, std::memory_order_relaxed);     \
        return 1;                                             \
    } while (0)
#define INPROC_HIP(call)                                                                                   \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) INPROC_FAIL("in-process transport: %s (%s)", hipGetErrorString(e_), #call);  \
    } while (0)
// (slabs along a periodic x axis form a ring: toLeft[0] reaches the last rank, toRight[nranks - 1] the first)
InprocChannel* inproc_out(InprocTransport* T, int sd) {  // my message to side sd
    const int r = T->vt.rank, n = T->vt.nranks;
    const bool ring = slab_is_ring(T->c);
    if (sd == SPH_SLAB_LEFT) return (ring || r > 0) ? &T->g->toLeft[r] : nullptr;
    return (ring || r < n - 1) ? &T->g->toRight[r] : nullptr;
}
InprocChannel* inproc_in(InprocTransport* T, int sd) {  // the message side sd sends me
    const int r = T->vt.rank, n = T->vt.nranks;
    const bool ring = slab_is_ring(T->c);
    if (sd == SPH_SLAB_LEFT) return (ring || r > 0) ? &T->g->toRight[(r + n - 1) % n] : nullptr;
    return (ring || r < n - 1) ? &T->g->toLeft[(r + 1) % n] : nullptr;
}
int inproc_exchange(void* user, const void* const send[2], const size_t sb[2], void* const recv[2], const size_t rb[2], int on_device, void* stream) {
    InprocTransport* T = (InprocTransport*)user;
    hipStream_t st = (hipStream_t)stream;
    if (!on_device) INPROC_FAIL("in-process transport: host buffers are not supported");
    (void)hipSetDevice(T->c->device);
    const uint64_t seq = ++T->seq;
    for (int sd = 0; sd < 2; sd++) {  // 1. publish my buffers; the event says when the stream has finished writing them
        InprocChannel* ch = inproc_out(T, sd);
        if (!ch) continue;
        ch->ptr = send[sd]; ch->bytes = sb[sd];
        INPROC_HIP(hipEventRecord(ch->ready, st));
        ch->posted.store(seq, std::memory_order_release);
    }
    for (int sd = 0; sd < 2; sd++) {  // 2. pull the neighbours' buffers on my stream, behind their events
        InprocChannel* ch = inproc_in(T, sd);
        if (!ch) continue;
        if (!inproc_wait(T->g, ch->posted, seq)) INPROC_FAIL("in-process transport: rank %d waited in vain for the message of side %d (exchange %llu)", T->vt.rank, sd, (unsigned long long)seq);
        if (ch->bytes != rb[sd]) INPROC_FAIL("in-process transport: rank %d expects %zu bytes from side %d, the neighbour sends %zu", T->vt.rank, rb[sd], sd, ch->bytes);
        if (rb[sd]) {
            INPROC_HIP(hipStreamWaitEvent(st, ch->ready, 0));
            INPROC_HIP(hipMemcpyAsync(recv[sd], ch->ptr, rb[sd], hipMemcpyDefault, st));
        }
        INPROC_HIP(hipEventRecord(ch->done, st));
        ch->consumed.store(seq, std::memory_order_release);
    }
    for (int sd = 0; sd < 2; sd++) {  // 3. whatever my stream does next (the next pack into the same buffer) comes after the neighbour's copy
        InprocChannel* ch = inproc_out(T, sd);
        if (!ch) continue;
        if (!inproc_wait(T->g, ch->consumed, seq)) INPROC_FAIL("in-process transport: rank %d: side %d never took exchange %llu", T->vt.rank, sd, (unsigned long long)seq);
        if (sb[sd]) INPROC_HIP(hipStreamWaitEvent(st, ch->done, 0));
    }
    return 0;
}
int inproc_exchange_counts(void* user, const uint32_t send[2], uint32_t recv[2]) {
    InprocTransport* T = (InprocTransport*)user;
    const uint64_t seq = ++T->cseq;
    for (int sd = 0; sd < 2; sd++)
        if (InprocChannel* ch = inproc_out(T, sd)) { ch->count = send[sd]; ch->cposted.store(seq, std::memory_order_release); }
    for (int sd = 0; sd < 2; sd++) {
        recv[sd] = 0;
        InprocChannel* ch = inproc_in(T, sd);
        if (!ch) continue;
        if (!inproc_wait(T->g, ch->cposted, seq)) INPROC_FAIL("in-process transport: rank %d waited in vain for the count of side %d", T->vt.rank, sd);
        recv[sd] = ch->count;
        ch->cconsumed.store(seq, std::memory_order_release);
    }
    for (int sd = 0; sd < 2; sd++)
        if (InprocChannel* ch = inproc_out(T, sd))
            if (!inproc_wait(T->g, ch->cconsumed, seq)) INPROC_FAIL("in-process transport: rank %d: side %d never read its count", T->vt.rank, sd);
    return 0;
}
int inproc_allreduce(void* user, void* buf, size_t nwords, int op, int on_device, void* stream) {
    InprocTransport* T = (InprocTransport*)user;
    sph_slab_group* g = T->g;
    hipStream_t st = (hipStream_t)stream;
    if (!on_device) INPROC_FAIL("in-process transport: host buffers are not supported");
    if (op < SPH_SLAB_REDUCE_MAX_U32 || op > SPH_SLAB_REDUCE_MIN_F32) INPROC_FAIL("in-process transport: unknown reduction %d", op);
    if (!nwords || T->vt.nranks == 1) return 0;
    (void)hipSetDevice(T->c->device);
    if (T->tmp_words < nwords) {
        if (T->tmp) INPROC_HIP(hipFree(T->tmp));
        T->tmp = nullptr; T->tmp_words = 0;
        INPROC_HIP(hipMalloc((void**)&T->tmp, nwords * 5 / 4 * sizeof(uint32_t) + 256));
        T->tmp_words = nwords * 5 / 4 + 64;
    }
    const uint64_t seq = ++T->aseq;
    const int r = T->vt.rank, nr = T->vt.nranks;
    g->slot[r].ptr = buf; g->slot[r].device = T->c->device;
    INPROC_HIP(hipEventRecord(g->slot[r].ready, st));
    g->slot[r].posted.store(seq, std::memory_order_release);
    InprocPtrs P;
    for (int q = 0; q < nr; q++) {
        if (!inproc_wait(g, g->slot[q].posted, seq)) INPROC_FAIL("in-process transport: rank %d waited in vain for rank %d (reduction %llu)", r, q, (unsigned long long)seq);
        if (q != r) INPROC_HIP(hipStreamWaitEvent(st, g->slot[q].ready, 0));
        if (g->slot[q].device != T->c->device && !(T->peer_mask & (1u << q))) {  // the reduction kernel reads the other ranks' buffers in place
            const hipError_t pe = hipDeviceEnablePeerAccess(g->slot[q].device, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) INPROC_FAIL("in-process transport: no peer access from device %d to device %d", T->c->device, g->slot[q].device);
            (void)hipGetLastError();
            T->peer_mask |= 1u << q;
        }
        P.p[q] = (const uint32_t*)g->slot[q].ptr;
    }
    hipLaunchKernelGGL(k_inproc_allreduce, dim3(nblk(nwords)), dim3(256), 0, st, P, nr, nwords, op, T->tmp);
    INPROC_HIP(hipEventRecord(g->slot[r].read_done, st));
    g->slot[r].read.store(seq, std::memory_order_release);
    for (int q = 0; q < nr; q++) {  // nobody overwrites a buffer the others still read
        if (q == r) continue;
        if (!inproc_wait(g, g->slot[q].read, seq)) INPROC_FAIL("in-process transport: rank %d: rank %d never read reduction %llu", r, q, (unsigned long long)seq);
        INPROC_HIP(hipStreamWaitEvent(st, g->slot[q].read_done, 0));
    }
    INPROC_HIP(hipMemcpyAsync(buf, T->tmp, nwords * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    return 0;
}
void inproc_transport_release(InprocTransport* T) {
    (void)hipSetDevice(T->c->device);
    if (T->tmp) (void)hipFree(T->tmp);
    // the events stay with the group: a neighbour may still have a wait on them queued
    T->g->members.fetch_sub(1);
    delete T;
}
#undef INPROC_FAIL
#undef INPROC_HIP

void transport_register(sph_ctx* c, TransportBase* T) {
    std::lock_guard<std::mutex> lk(g_transport_mutex);
    g_live_transports.insert(&T->vt);
    c->transports.push_back(&T->vt);
}
// releases `t` if it is a live transport of this library; detaches it from its context
void transport_free(sph_slab_transport* t) {
    TransportBase* T = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_transport_mutex);
        if (!t || !g_live_transports.count(t)) return;  // not ours, or already released (sph_destroy releases what its context still holds)
        g_live_transports.erase(t);
        T = (TransportBase*)t->user;
        auto& v = T->c->transports;
        v.erase(std::remove(v.begin(), v.end(), (void*)t), v.end());
    }
    if (T->kind == 1) rccl_transport_release((RcclTransport*)T);
    else inproc_transport_release((InprocTransport*)T);
}
}  // namespace

extern "C" size_t sph_slab_rccl_unique_id(void* out, size_t capacity) {
    const RcclApi& R = rccl_api();
    ncclUniqueId id;
    if (!R.ok || capacity < sizeof(id) || R.GetUniqueId(&id) != ncclSuccess) return 0;
    memcpy(out, &id, sizeof(id));
    return sizeof(id);
}
extern "C" sph_slab_transport* sph_slab_transport_rccl(sph_ctx* c, const void* unique_id, size_t id_bytes, int rank, int nranks) {
    const RcclApi& R = rccl_api();
    if (R.mixed) {
        set_error(c, "sph_slab_transport_rccl: the RCCL in this process (%s) does not belong to the HIP runtime the solver uses (%s), or two runtimes are mapped: see INTEGRATION.md, 'One ROCm runtime per process'",
                  R.rccl_path.c_str(), R.hip_path.c_str());
        return nullptr;
    }
    if (!R.ok) { set_error(c, "sph_slab_transport_rccl: librccl.so could not be loaded"); return nullptr; }
    ncclUniqueId id;
    if (id_bytes != sizeof(id)) { set_error(c, "sph_slab_transport_rccl: the unique id has %zu bytes, expected %zu", id_bytes, sizeof(id)); return nullptr; }
    memcpy(&id, unique_id, sizeof(id));
    (void)hipSetDevice(c->device);
    RcclTransport* T = new RcclTransport();
    T->c = c; T->kind = 1;
    T->vt.user = T; T->vt.rank = rank; T->vt.nranks = nranks; T->vt.stream_ordered = 1;
    T->vt.exchange = rccl_exchange; T->vt.exchange_counts = rccl_exchange_counts; T->vt.allreduce = rccl_allreduce;
    T->d_counts = nullptr; T->h_counts = nullptr;
    ncclResult_t r = R.CommInitRank(&T->comm, nranks, id, rank);
    if (r != ncclSuccess) { set_error(c, "ncclCommInitRank: %s", R.GetErrorString(r)); delete T; return nullptr; }
    if (hipMalloc((void**)&T->d_counts, 16) != hipSuccess || hipHostMalloc((void**)&T->h_counts, 16, hipHostMallocDefault) != hipSuccess) {
        set_error(c, "sph_slab_transport_rccl: out of memory");
        rccl_transport_release(T);
        return nullptr;
    }
    transport_register(c, T);
    return &T->vt;
}

extern "C" sph_slab_group* sph_slab_group_create(int nranks) {
    if (nranks < 1 || nranks > INPROC_MAX_RANKS) return nullptr;
    sph_slab_group* g = new sph_slab_group();
    g->nranks = nranks;
    if (const char* e = getenv("CHSPH_INPROC_TIMEOUT")) g->timeout_s = atof(e);
    return g;
}
extern "C" int sph_slab_group_destroy(sph_slab_group* g) {
    if (!g) return 0;
    if (g->members.load() != 0) return 1;  // transports still attached: release them (or their contexts) first
    for (int r = 0; r < INPROC_MAX_RANKS; r++) {
        for (InprocChannel* ch : {&g->toRight[r], &g->toLeft[r]}) {
            if (ch->ready) (void)hipEventDestroy(ch->ready);
            if (ch->done) (void)hipEventDestroy(ch->done);
        }
        if (g->slot[r].ready) (void)hipEventDestroy(g->slot[r].ready);
        if (g->slot[r].read_done) (void)hipEventDestroy(g->slot[r].read_done);
    }
    delete g;
    return 0;
}
extern "C" sph_slab_transport* sph_slab_transport_inproc(sph_ctx* c, sph_slab_group* g, int rank) {
    if (!g || rank < 0 || rank >= g->nranks) { set_error(c, "sph_slab_transport_inproc: rank %d outside the group", rank); return nullptr; }
    (void)hipSetDevice(c->device);
    InprocTransport* T = new InprocTransport();
    T->c = c; T->kind = 2; T->g = g;
    T->vt.user = T; T->vt.rank = rank; T->vt.nranks = g->nranks; T->vt.stream_ordered = 1;
    T->vt.exchange = inproc_exchange; T->vt.exchange_counts = inproc_exchange_counts; T->vt.allreduce = inproc_allreduce;
    // events: `ready` of my outgoing channels and of my reduction slot live on my device, and so do the `done` events of my incoming channels
    bool ok = true;
    auto ev = [&](hipEvent_t* e) { if (!*e && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) ok = false; };
    // (all four channels of a ring member: whether the end ranks use theirs is known only once the context is configured as a slab rank)
    const int n = g->nranks;
    ev(&g->toLeft[rank].ready); ev(&g->toRight[rank].ready);
    ev(&g->toRight[(rank + n - 1) % n].done); ev(&g->toLeft[(rank + 1) % n].done);
    ev(&g->slot[rank].ready); ev(&g->slot[rank].read_done);
    if (!ok) { set_error(c, "sph_slab_transport_inproc: event creation failed"); delete T; return nullptr; }
    g->members.fetch_add(1);
    transport_register(c, T);
    return &T->vt;
}
extern "C" void sph_slab_transport_free(sph_slab_transport* t) { transport_free(t); }

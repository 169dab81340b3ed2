// TEST INFRASTRUCTURE — an in-process stand-in for RCCL's point-to-point API, so that the multi-rank code path of
// csrc/fluid_stripes.cpp (ncclCommInitRank, grouped ncclSend / ncclRecv to both neighbours on the comm stream) can run
// with SEVERAL ranks on a single-GPU box: every "rank" is a host thread with its own stripe context on the same
// device.  libfluid_hip.so dlopen()s this file instead of librccl.so when FLUID_RCCL_LIB points at it.
//
// Semantics kept from NCCL: CommInitRank blocks until all ranks of the id have arrived; sends and receives between a
// pair of ranks match in issue order; a group issues all its operations together; a send is complete on the sender's
// stream only after the receiver has copied the data; everything is ordered on the streams the caller passes.
// Never shipped, never on a product path.
//
// A link that takes time (the interior-first overlap had only ever met instantaneous copies): FAKE_RCCL_DELAY_US=<us> and /
// or FAKE_RCCL_GBPS=<GB/s> make every receive wait that long (latency + bytes / bandwidth) ON THE RECEIVER'S STREAM, behind the sender's
// "data exists" event and in front of the copy — a one-thread kernel spinning on the 100 MHz s_memrealtime counter, so the host never sleeps
// and the other streams of the device keep running, as they do while a real xGMI transfer is in flight (tools/overlap_vs_link.py).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <algorithm>
#include <string>
#include <vector>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclChar = 0, ncclFloat = 7 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct FakeComm* ncclComm_t;

}  // extern "C"

namespace {

struct Parcel {          // one send waiting for its receive
    const void* src;
    size_t bytes;
    hipEvent_t ready;    // recorded on the sender's stream: the data exists
    hipEvent_t copied;   // recorded on the receiver's stream: the data has been copied out
    bool taken = false;
};

struct World {
    int nranks = 0, arrived = 0;
    std::map<std::pair<int, int>, std::deque<Parcel*>> box;  // (src, dst) -> sends in issue order
};

std::mutex g_mu;
std::condition_variable g_cv;
std::map<std::string, World> g_worlds;
int g_next_id = 1;

}  // namespace

struct FakeComm {
    World* world;
    int rank, nranks;
};

namespace {

struct Op {
    bool send;
    void* ptr;
    size_t bytes;
    int peer;
    FakeComm* comm;
    hipStream_t stream;
};
__global__ void k_link_delay(unsigned long long ticks)   // 10 ns per tick
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

double link_us(size_t bytes)
{
    static const double lat = [] { const char* e = getenv("FAKE_RCCL_DELAY_US"); return e ? atof(e) : 0.0; }();
    static const double gbps = [] { const char* e = getenv("FAKE_RCCL_GBPS"); return e ? atof(e) : 0.0; }();
    return lat + (gbps > 0 ? (double)bytes / (gbps * 1e3) : 0.0);
}

thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

// FAKE_RCCL_LOOPBACK=1: ONE rank of a larger world runs alone — every receive from a neighbour is served from this rank's own send to
// that neighbour (the k-th receive from peer p takes the k-th send to p of the same group), nothing waits for another rank, and
// ncclCommInitRank returns at once.  The ghost rows then hold mirrored data, which is physically harmless, and the rank sees exactly the
// timing of a middle rank: its own compute plus exchanges that take `link_us` — with no other rank's kernels on the device
// (tools/overlap_vs_link.py).
bool loopback()
{
    static const bool on = [] { const char* e = getenv("FAKE_RCCL_LOOPBACK"); return e && atoi(e) != 0; }();
    return on;
}

// the link's time, ONCE per group and stream: a group's transfers to different neighbours run side by side over different links
ncclResult_t link_wait(std::vector<Op>& ops)
{
    std::map<hipStream_t, double> worst;
    for (Op& o : ops)
        if (!o.send) worst[o.stream] = std::max(worst[o.stream], link_us(o.bytes));
    for (auto& kv : worst)
        if (kv.second > 0) k_link_delay<<<1, 1, 0, kv.first>>>((unsigned long long)(kv.second * 100.0));
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t run_loopback(std::vector<Op>& ops)
{
    if (link_wait(ops) != ncclSuccess) return ncclUnhandledCudaError;   // every op of a group is on the caller's comm stream: sends precede in stream order
    std::map<int, std::vector<Op*>> sends;
    for (Op& o : ops)
        if (o.send) sends[o.peer].push_back(&o);
    std::map<int, size_t> next;
    for (Op& o : ops)
        if (!o.send) {
            auto& q = sends[o.peer];
            const size_t k = next[o.peer]++;
            if (k >= q.size() || q[k]->bytes != o.bytes) return ncclInvalidArgument;
            if (hipMemcpyAsync(o.ptr, q[k]->ptr, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) return ncclUnhandledCudaError;
        }
    return ncclSuccess;
}

ncclResult_t run(std::vector<Op>& ops)
{
    if (loopback()) return run_loopback(ops);
    std::vector<Parcel*> mine;
    // 1. post every send (never blocks)
    for (Op& o : ops)
        if (o.send) {
            Parcel* p = new Parcel{ o.ptr, o.bytes, nullptr, nullptr };
            if (hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
            if (hipEventCreateWithFlags(&p->copied, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
            if (hipEventRecord(p->ready, o.stream) != hipSuccess) return ncclUnhandledCudaError;
            {
                std::lock_guard<std::mutex> lk(g_mu);
                o.comm->world->box[{ o.comm->rank, o.peer }].push_back(p);
            }
            g_cv.notify_all();
            mine.push_back(p);
        }
    // 2. every receive takes the oldest unmatched send of its peer
    std::set<hipStream_t> delayed;
    for (Op& o : ops)
        if (!o.send) {
            Parcel* p = nullptr;
            {
                std::unique_lock<std::mutex> lk(g_mu);
                auto& q = o.comm->world->box[{ o.peer, o.comm->rank }];
                g_cv.wait(lk, [&] { return !q.empty(); });
                p = q.front();
                q.pop_front();
            }
            if (p->bytes != o.bytes) return ncclInvalidArgument;  // count mismatch between the two sides
            if (hipStreamWaitEvent(o.stream, p->ready, 0) != hipSuccess) return ncclUnhandledCudaError;
            if (!delayed.count(o.stream)) {   // once per group and stream, for the group's largest message (links to different peers run side by side)
                delayed.insert(o.stream);
                double us = 0;
                for (Op& q : ops)
                    if (!q.send && q.stream == o.stream) us = std::max(us, link_us(q.bytes));
                if (us > 0) k_link_delay<<<1, 1, 0, o.stream>>>((unsigned long long)(us * 100.0));
            }
            if (hipMemcpyAsync(o.ptr, p->src, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) return ncclUnhandledCudaError;
            if (hipEventRecord(p->copied, o.stream) != hipSuccess) return ncclUnhandledCudaError;
            {
                std::lock_guard<std::mutex> lk(g_mu);
                p->taken = true;
            }
            g_cv.notify_all();
        }
    // 3. a send completes on the sender's stream once the receiver has copied
    size_t k = 0;
    for (Op& o : ops)
        if (o.send) {
            Parcel* p = mine[k++];
            {
                std::unique_lock<std::mutex> lk(g_mu);
                g_cv.wait(lk, [&] { return p->taken; });
            }
            if (hipStreamWaitEvent(o.stream, p->copied, 0) != hipSuccess) return ncclUnhandledCudaError;
            (void)hipEventDestroy(p->ready);   // destruction is deferred by the runtime until the recorded work is done
            (void)hipEventDestroy(p->copied);
            delete p;
        }
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    std::lock_guard<std::mutex> lk(g_mu);
    std::memset(id, 0, sizeof *id);
    std::snprintf(id->internal, sizeof id->internal, "fake-rccl-%d", g_next_id++);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::unique_lock<std::mutex> lk(g_mu);
    World& w = g_worlds[std::string(id.internal)];
    if (w.nranks == 0) w.nranks = nranks;
    if (w.nranks != nranks) return ncclInvalidArgument;
    w.arrived++;
    g_cv.notify_all();
    if (!loopback()) g_cv.wait(lk, [&] { return w.arrived >= w.nranks; });  // collective: returns when every rank is here
    *comm = new FakeComm{ &w, rank, nranks };
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    t_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) return ncclInvalidArgument;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run(ops);
}

static ncclResult_t enqueue(bool send, void* ptr, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || (type != ncclFloat && type != ncclChar) || peer < 0 || peer >= comm->nranks) return ncclInvalidArgument;
    t_ops.push_back(Op{ send, ptr, count * (type == ncclFloat ? sizeof(float) : 1), peer, comm, stream });
    if (t_depth == 0) {
        std::vector<Op> ops;
        ops.swap(t_ops);
        return run(ops);
    }
    return ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    return enqueue(true, const_cast<void*>(buf), count, type, peer, comm, stream);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    return enqueue(false, buf, count, type, peer, comm, stream);
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "success";
    case ncclInvalidArgument: return "invalid argument (fake rccl: count / peer / type mismatch)";
    default: return "fake rccl error";
    }
}

}  // extern "C"

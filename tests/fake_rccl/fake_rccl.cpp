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
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
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

struct Reduce {           // one ncclAllReduce of the world: every rank's staged contribution
    int arrived = 0, departed = 0;
    std::vector<float*> stage;
    std::vector<hipEvent_t> ready, done;
};

struct World {
    int nranks = 0, arrived = 0;
    std::map<std::pair<int, int>, std::deque<Parcel*>> box;  // (src, dst) -> sends in issue order
    std::map<long, Reduce> reduces;                          // all-reduces by sequence number (every rank issues them in the same order)
};

std::mutex g_mu;
std::condition_variable g_cv;
std::map<std::string, World> g_worlds;
int g_next_id = 1;

}  // namespace

struct FakeComm {
    World* world;
    int rank, nranks;
    long reduce_seq = 0;
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

constexpr int kMaxRanks = 64;
struct ReduceArgs {
    const float* src[kMaxRanks];
    int n;
};
__global__ void k_all_reduce(float* dst, ReduceArgs a, size_t count, int op)
{
    for (size_t i = threadIdx.x; i < count; i += blockDim.x) {
        float v = a.src[0][i];
        for (int r = 1; r < a.n; r++) {
            const float x = a.src[r][i];
            v = op == ncclSum ? v + x : (op == ncclProd ? v * x : (op == ncclMax ? fmaxf(v, x) : fminf(v, x)));
        }
        dst[i] = v;
    }
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

// every rank's contribution is staged (in-place calls, and a sender that reuses its buffer), the ranks meet, each reduces all the staged
// vectors into its own receive buffer on its own stream; the last rank to leave waits for every reduction and frees the staging
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || type != ncclFloat || !sendbuff || !recvbuff || comm->nranks > kMaxRanks || t_depth != 0) return ncclInvalidArgument;
    if (loopback() || comm->nranks == 1) {
        if (sendbuff != recvbuff && hipMemcpyAsync(recvbuff, sendbuff, count * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
        return ncclSuccess;
    }
    float* stage = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;
    if (hipMalloc((void**)&stage, count * sizeof(float)) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(stage, sendbuff, count * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventRecord(ready, stream) != hipSuccess) return ncclUnhandledCudaError;
    const long seq = comm->reduce_seq++;
    ReduceArgs a{};
    std::vector<hipEvent_t> readies;
    {
        std::unique_lock<std::mutex> lk(g_mu);
        Reduce& R = comm->world->reduces[seq];
        if (R.stage.empty()) {
            R.stage.assign(comm->nranks, nullptr);
            R.ready.assign(comm->nranks, nullptr);
            R.done.assign(comm->nranks, nullptr);
        }
        R.stage[comm->rank] = stage;
        R.ready[comm->rank] = ready;
        R.arrived++;
        g_cv.notify_all();
        g_cv.wait(lk, [&] { return R.arrived >= comm->nranks; });   // collective: every rank is here
        a.n = comm->nranks;
        for (int r = 0; r < comm->nranks; r++) a.src[r] = R.stage[r];
        readies = R.ready;
    }
    for (hipEvent_t e : readies)
        if (hipStreamWaitEvent(stream, e, 0) != hipSuccess) return ncclUnhandledCudaError;
    k_all_reduce<<<1, 64, 0, stream>>>((float*)recvbuff, a, count, (int)op);
    if (hipGetLastError() != hipSuccess || hipEventRecord(done, stream) != hipSuccess) return ncclUnhandledCudaError;
    Reduce last;
    bool cleanup = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        Reduce& R = comm->world->reduces[seq];
        R.done[comm->rank] = done;
        if (++R.departed == comm->nranks) {
            last = R;
            cleanup = true;
            comm->world->reduces.erase(seq);
        }
    }
    if (cleanup)
        for (int r = 0; r < (int)last.stage.size(); r++) {
            (void)hipEventSynchronize(last.done[r]);
            (void)hipFree(last.stage[r]);
            (void)hipEventDestroy(last.ready[r]);
            (void)hipEventDestroy(last.done[r]);
        }
    return ncclSuccess;
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

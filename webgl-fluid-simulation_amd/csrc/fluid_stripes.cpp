// fluid_stripes.cpp — the reference's step() (script.js:1231-1294) on a row-stripe decomposition: one stripe
// context per GPU, ghost rows refreshed from the neighbouring ranks with RCCL send/recv over xGMI.
//
// The path shards by rows: every pass is a radius-1 stencil except the advection gather.  There is no global
// collective on the data path: a rank only ever talks to rank - 1 and rank + 1 (ncclSend / ncclRecv inside one
// ncclGroupStart / ncclGroupEnd per exchange, in place on the ghost rows — rows are contiguous in the field
// arrays, so nothing is packed or staged).
//
// Communication-avoiding plan (MI355X-first: xGMI is point to point, a neighbour message is small and each
// exchange costs a fixed latency): instead of one single-row exchange before each of the 7 + ITERS reference
// passes, a rank recomputes a few ghost rows redundantly and exchanges many rows at a time:
//
//     exchange { velocity H rows, pressure D0 + e0 rows }
//     curl -> vorticity -> divergence      (ghost rows out to H - 3)
//     clear + D0 Jacobi iterations; while iterations remain: exchange pressure, next block (D <= H - 3 each)
//     gradient subtract
//     exchange { velocity H rows, dye Hd rows }
//     advect velocity + dye
//
// With H = 32 and 50 iterations: 3 exchanges per step; H >= 54: 2.  Every recomputed ghost row is the same
// arithmetic on the same inputs as its owner's, so the decomposed result is BITWISE equal to the single-domain run.
//
// RCCL is loaded at run time (dlopen): the single-GPU path does not depend on it, and a process that already
// holds an RCCL (PyTorch bundles one) shares that copy instead of pulling in a second HIP runtime.
#include "fluid_internal.h"

#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

using namespace fluid;
using namespace fluid_impl;

namespace {

// ---- the plan: pure host logic, also exported for the CPU tests (fluid_stripe_plan) ----------------------------
void push_exchange(std::vector<fluid_stripe_op>& ops, int f0, int r0, int f1, int r1)
{
    fluid_stripe_op op{};
    op.kind = FLUID_OP_EXCHANGE;
    if (r0 > 0) {
        op.field[op.n_items] = f0;
        op.rows[op.n_items++] = r0;
    }
    if (f1 >= 0 && r1 > 0) {
        op.field[op.n_items] = f1;
        op.rows[op.n_items++] = r1;
    }
    if (op.n_items) ops.push_back(op);
}

void push_pass(std::vector<fluid_stripe_op>& ops, int kind, int iters, int ext)
{
    fluid_stripe_op op{};
    op.kind = kind;
    op.iters = iters;
    op.ext = ext;
    ops.push_back(op);
}

int build_plan(int halo, int dye_halo, int iterations, std::vector<fluid_stripe_op>& ops)
{
    if (halo < 4 || dye_halo < 1 || iterations < 0) return FLUID_ERR_INVALID;
    const int H = halo;
    // pressure blocks: divergence is valid H-3 rows out and iteration k of a block of d needs it d-k+e rows out
    // -> d <= H-3; the last block also leaves e = 1 valid ghost row (gradient subtract reads pressure one row out)
    struct Block { int d, e; };
    std::vector<Block> blocks;
    for (int remaining = iterations; remaining > 0;) {
        const int d = remaining < H - 3 ? remaining : H - 3;
        remaining -= d;
        blocks.push_back({ d, remaining == 0 ? 1 : 0 });
    }
    const int first = blocks.empty() ? 1 : blocks[0].d + blocks[0].e;
    push_exchange(ops, FLUID_VELOCITY, H, FLUID_PRESSURE, first);
    push_pass(ops, FLUID_OP_CURL_VORT_DIV, 0, H - 3);  // curl to H-1, vorticity to H-2, divergence to H-3 rows out
    if (blocks.empty()) push_pass(ops, FLUID_OP_CLEAR, 0, 1);
    for (size_t k = 0; k < blocks.size(); k++) {
        if (k == 0) {  // the exchanged ghost rows hold the neighbour's PRE-clear pressure: the clear covers them too
            push_pass(ops, FLUID_OP_CLEAR_JACOBI, blocks[k].d, blocks[k].e);
        } else {
            push_exchange(ops, FLUID_PRESSURE, blocks[k].d + blocks[k].e, -1, 0);
            push_pass(ops, FLUID_OP_JACOBI, blocks[k].d, blocks[k].e);
        }
    }
    push_pass(ops, FLUID_OP_GRADSUB, 0, 0);
    push_exchange(ops, FLUID_VELOCITY, H, FLUID_DYE, dye_halo);
    push_pass(ops, FLUID_OP_ADVECT, 0, 0);
    return FLUID_OK;
}

int run_pass(fluid_ctx* c, const fluid_stripe_op& op, float dt, const fluid_params* P)
{
    switch (op.kind) {
    case FLUID_OP_CURL_VORT_DIV: return pass_curl_vort_div(c, P->curl, dt, op.ext, nullptr);
    case FLUID_OP_CLEAR: return pass_clear(c, P->pressure, op.ext);
    case FLUID_OP_CLEAR_JACOBI: return pass_clear_jacobi(c, P->pressure, op.iters, op.ext, nullptr);
    case FLUID_OP_JACOBI: return pass_jacobi(c, op.iters, op.ext, 1.0f, nullptr);
    case FLUID_OP_GRADSUB: return pass_gradsub(c, op.ext);
    case FLUID_OP_ADVECT: return pass_advect(c, dt, P->velocity_dissipation, P->density_dissipation, nullptr);
    default: return c->fail(FLUID_ERR_INVALID, "unknown stripe op");
    }
}

// ghost / owned row blocks of one exchange item, as float pointers + float counts
struct Rows {
    float *send_lo, *recv_lo, *send_hi, *recv_hi;
    size_t count;
};

int rows_of(fluid_ctx* c, int field, int n, Rows* r)
{
    FieldRef f;
    CK(field_ref(c, field, &f));
    if (n < 1 || n > f.halo || n > f.rows) return c->fail(FLUID_ERR_INVALID, "exchange rows exceed the ghost rows / the stripe");
    const size_t rowf = (size_t)f.win->W * f.nc;
    float* base = (float*)f.ptr;
    const int h = f.halo, rr = f.rows;
    r->send_lo = base + (size_t)h * rowf;             // my lowest owned rows   -> lower neighbour's top ghost rows
    r->recv_lo = base + (size_t)(h - n) * rowf;       // my bottom ghost rows   <- lower neighbour's highest owned rows
    r->send_hi = base + (size_t)(h + rr - n) * rowf;  // my highest owned rows  -> upper neighbour's bottom ghost rows
    r->recv_hi = base + (size_t)(h + rr) * rowf;      // my top ghost rows      <- upper neighbour's lowest owned rows
    r->count = (size_t)n * rowf;
    return FLUID_OK;
}

// ---- RCCL, resolved at run time -------------------------------------------------------------------------------
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;
std::string g_rccl_path;  // fluid_comm_set_library()

const Rccl* rccl(std::string* why)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return &g_rccl;
    std::vector<std::string> names;
    if (!g_rccl_path.empty()) names.push_back(g_rccl_path);
    if (const char* e = getenv("FLUID_RCCL_LIB")) names.push_back(e);
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    std::string tried;
    for (const std::string& n : names) {
        void* h = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (!h) {
            tried += n + ": " + (dlerror() ? dlerror() : "?") + "; ";
            continue;
        }
        Rccl r;
        r.handle = h;
        bool ok = true;
        auto sym = [&](const char* s) {
            void* p = dlsym(h, s);
            if (!p) ok = false;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!ok) {
            tried += n + ": missing nccl symbols; ";
            dlclose(h);
            continue;
        }
        g_rccl = r;
        return &g_rccl;
    }
    if (why) *why = "cannot load RCCL (" + tried + ")";
    return nullptr;
}

int nccl_fail(fluid_ctx* c, const Rccl* R, ncclResult_t e, const char* what)
{
    return c->fail(FLUID_ERR_COMM, std::string(what) + ": " + (R && R->GetErrorString ? R->GetErrorString(e) : "RCCL error"));
}

#define NCCLCK(c, R, expr)                                        \
    do {                                                          \
        ncclResult_t _e = (expr);                                 \
        if (_e != ncclSuccess) return nccl_fail(c, R, _e, #expr); \
    } while (0)

// one batched neighbour exchange on the context stream: kernels before it have produced the owned rows it sends,
// kernels after it read the ghost rows it fills (stream order)
int rccl_exchange(fluid_ctx* c, const fluid_stripe_op& op)
{
    const Rccl* R = rccl(nullptr);
    if (!R || !c->comm) return c->fail(FLUID_ERR_COMM, "stripe context has no communicator (fluid_comm_init)");
    ncclComm_t comm = (ncclComm_t)c->comm;
    const int rank = c->desc.part, world = c->desc.parts;
    Rows rows[2];
    for (int i = 0; i < op.n_items; i++) CK(rows_of(c, op.field[i], op.rows[i], &rows[i]));
    NCCLCK(c, R, R->GroupStart());
    // per peer, sends and receives are issued in item order on both sides, so they pair up
    if (rank > 0)
        for (int i = 0; i < op.n_items; i++) {
            NCCLCK(c, R, R->Send(rows[i].send_lo, rows[i].count, ncclFloat, rank - 1, comm, c->stream));
            NCCLCK(c, R, R->Recv(rows[i].recv_lo, rows[i].count, ncclFloat, rank - 1, comm, c->stream));
        }
    if (rank < world - 1)
        for (int i = 0; i < op.n_items; i++) {
            NCCLCK(c, R, R->Send(rows[i].send_hi, rows[i].count, ncclFloat, rank + 1, comm, c->stream));
            NCCLCK(c, R, R->Recv(rows[i].recv_hi, rows[i].count, ncclFloat, rank + 1, comm, c->stream));
        }
    NCCLCK(c, R, R->GroupEnd());
    c->exchanges++;
    return FLUID_OK;
}

// the same exchange between stripe contexts that live in ONE process (the whole stripe set on one or several
// devices, driven by one host thread): device-to-device copies ordered with events.  This is how the plan and the
// windowed kernels are validated bit for bit on a single-GPU box; bench.py --gpus N uses the RCCL path.
int group_exchange(fluid_ctx** cs, int n, const fluid_stripe_op& op)
{
    // 1. every stripe's producers are done before anyone copies from it
    for (int r = 0; r < n; r++) {
        HIPCK(cs[r], hipSetDevice(cs[r]->device));
        HIPCK(cs[r], hipEventRecord(cs[r]->ev_group[0], cs[r]->stream));
    }
    for (int r = 0; r < n; r++) {
        fluid_ctx* c = cs[r];
        HIPCK(c, hipSetDevice(c->device));
        if (r > 0) HIPCK(c, hipStreamWaitEvent(c->stream, cs[r - 1]->ev_group[0], 0));
        if (r < n - 1) HIPCK(c, hipStreamWaitEvent(c->stream, cs[r + 1]->ev_group[0], 0));
        for (int i = 0; i < op.n_items; i++) {
            Rows me, lo, hi;
            CK(rows_of(c, op.field[i], op.rows[i], &me));
            if (r > 0) {
                CK(rows_of(cs[r - 1], op.field[i], op.rows[i], &lo));
                HIPCK(c, hipMemcpyAsync(me.recv_lo, lo.send_hi, me.count * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            }
            if (r < n - 1) {
                CK(rows_of(cs[r + 1], op.field[i], op.rows[i], &hi));
                HIPCK(c, hipMemcpyAsync(me.recv_hi, hi.send_lo, me.count * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            }
        }
        HIPCK(c, hipEventRecord(c->ev_group[1], c->stream));
        c->exchanges++;
    }
    // 2. nobody overwrites rows a neighbour is still copying from
    for (int r = 0; r < n; r++) {
        fluid_ctx* c = cs[r];
        HIPCK(c, hipSetDevice(c->device));
        if (r > 0) HIPCK(c, hipStreamWaitEvent(c->stream, cs[r - 1]->ev_group[1], 0));
        if (r < n - 1) HIPCK(c, hipStreamWaitEvent(c->stream, cs[r + 1]->ev_group[1], 0));
    }
    return FLUID_OK;
}

int plan_for(fluid_ctx* c, const fluid_params* P, std::vector<fluid_stripe_op>& ops)
{
    if (build_plan(c->desc.halo, c->dye_halo, P->iterations, ops) != FLUID_OK) return c->fail(FLUID_ERR_INVALID, "stripe plan: bad halo / iterations");
    return FLUID_OK;
}

}  // namespace

namespace fluid_impl {

int stripe_step_n(fluid_ctx* c, int n, float dt, const fluid_params* P)
{
    if (!c->comm) return c->fail(FLUID_ERR_COMM, "a stripe context steps with its communicator: call fluid_comm_init first (or drive the "
                                                  "passes and exchanges yourself through fluid_pass_* / fluid_field_device_ptr)");
    std::vector<fluid_stripe_op> ops;
    CK(plan_for(c, P, ops));
    for (int k = 0; k < n; k++)
        for (const fluid_stripe_op& op : ops) {
            if (op.kind == FLUID_OP_EXCHANGE) CK(rccl_exchange(c, op));
            else CK(run_pass(c, op, dt, P));
        }
    return FLUID_OK;
}

void stripes_release(fluid_ctx* c)
{
    if (c->comm) {
        const Rccl* R = rccl(nullptr);
        if (R) (void)R->CommDestroy((ncclComm_t)c->comm);
        c->comm = nullptr;
    }
    for (auto& e : c->ev_group)
        if (e) {
            (void)hipEventDestroy(e);
            e = nullptr;
        }
}

}  // namespace fluid_impl

// ================================================================================================================
extern "C" {

int fluid_stripe_plan(int halo, int dye_halo, int iterations, fluid_stripe_op* ops, int max_ops, int* n_ops)
{
    if (!n_ops) return FLUID_ERR_INVALID;
    std::vector<fluid_stripe_op> v;
    const int rc = build_plan(halo, dye_halo, iterations, v);
    if (rc != FLUID_OK) return rc;
    *n_ops = (int)v.size();
    if (ops) {
        if (max_ops < (int)v.size()) return FLUID_ERR_INVALID;
        std::memcpy(ops, v.data(), v.size() * sizeof(fluid_stripe_op));
    }
    return FLUID_OK;
}

int fluid_comm_set_library(const char* path)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return FLUID_ERR_INVALID;  // already resolved
    g_rccl_path = path ? path : "";
    return FLUID_OK;
}

int fluid_comm_unique_id(fluid_comm_id* id)
{
    if (!id) return FLUID_ERR_INVALID;
    static_assert(sizeof(fluid_comm_id) == sizeof(ncclUniqueId), "fluid_comm_id must be an ncclUniqueId");
    std::string why;
    const Rccl* R = rccl(&why);
    if (!R) {
        std::fprintf(stderr, "libfluid_hip: %s\n", why.c_str());
        return FLUID_ERR_COMM;
    }
    ncclUniqueId u;
    if (R->GetUniqueId(&u) != ncclSuccess) return FLUID_ERR_COMM;
    std::memcpy(id, &u, sizeof u);
    return FLUID_OK;
}

int fluid_comm_init(fluid_ctx* c, const fluid_comm_id* id)
{
    if (!c || !id) return FLUID_ERR_INVALID;
    if (c->comm) return c->fail(FLUID_ERR_INVALID, "communicator already initialised");
    std::string why;
    const Rccl* R = rccl(&why);
    if (!R) return c->fail(FLUID_ERR_COMM, why);
    HIPCK(c, hipSetDevice(c->device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    NCCLCK(c, R, R->CommInitRank(&comm, c->desc.parts, u, c->desc.part));
    c->comm = comm;
    return FLUID_OK;
}

int fluid_comm_selftest(fluid_ctx* c, int nfloats)
{
    // loop a buffer through ncclSend / ncclRecv to THIS rank inside one group, on the context stream, between two
    // kernels: checks that the library resolves, the communicator works and the transfers are stream-ordered
    if (!c || nfloats < 1) return FLUID_ERR_INVALID;
    const Rccl* R = rccl(nullptr);
    if (!R || !c->comm) return c->fail(FLUID_ERR_COMM, "no communicator");
    HIPCK(c, hipSetDevice(c->device));
    float *a = nullptr, *b = nullptr;
    HIPCK(c, hipMalloc((void**)&a, nfloats * sizeof(float)));
    HIPCK(c, hipMalloc((void**)&b, nfloats * sizeof(float)));
    int rc = FLUID_OK;
    do {
        if ((rc = c->hip(launch_fill(c->stream, a, (size_t)nfloats, 1, 3.25f, 0, 0, 0), "fill"))) break;
        if ((rc = c->hip(launch_fill(c->stream, b, (size_t)nfloats, 1, -1.0f, 0, 0, 0), "fill"))) break;
        ncclComm_t comm = (ncclComm_t)c->comm;
        const int me = c->desc.part;
        ncclResult_t e;
        if ((e = R->GroupStart()) != ncclSuccess || (e = R->Send(a, nfloats, ncclFloat, me, comm, c->stream)) != ncclSuccess ||
            (e = R->Recv(b, nfloats, ncclFloat, me, comm, c->stream)) != ncclSuccess || (e = R->GroupEnd()) != ncclSuccess) {
            rc = nccl_fail(c, R, e, "self send/recv");
            break;
        }
        std::vector<float> host(nfloats);
        if ((rc = c->hip(hipMemcpyAsync(host.data(), b, nfloats * sizeof(float), hipMemcpyDeviceToHost, c->stream), "copy"))) break;
        if ((rc = c->hip(hipStreamSynchronize(c->stream), "sync"))) break;
        for (int i = 0; i < nfloats; i++)
            if (host[i] != 3.25f) {
                rc = c->fail(FLUID_ERR_COMM, "self send/recv returned wrong data");
                break;
            }
    } while (0);
    (void)hipFree(a);
    (void)hipFree(b);
    return rc;
}

int fluid_group_step_n(fluid_ctx** cs, int n_ctx, int steps, float dt, const fluid_params* P)
{
    if (!cs || n_ctx < 1 || !P || steps < 0) return FLUID_ERR_INVALID;
    for (int r = 0; r < n_ctx; r++) {
        if (!cs[r]) return FLUID_ERR_INVALID;
        if (cs[r]->desc.parts != n_ctx || cs[r]->desc.part != r) return cs[r]->fail(FLUID_ERR_INVALID, "group must hold stripes 0..parts-1 in order");
        if (cs[r]->desc.halo != cs[0]->desc.halo) return cs[r]->fail(FLUID_ERR_INVALID, "stripes of a group share one halo");
    }
    if (n_ctx == 1) return fluid_step_n(cs[0], steps, dt, P);
    for (int r = 0; r < n_ctx; r++)
        for (auto& e : cs[r]->ev_group)
            if (!e) {
                HIPCK(cs[r], hipSetDevice(cs[r]->device));
                HIPCK(cs[r], hipEventCreateWithFlags(&e, hipEventDisableTiming));
            }
    std::vector<fluid_stripe_op> ops;
    CK(plan_for(cs[0], P, ops));
    for (int k = 0; k < steps; k++)
        for (const fluid_stripe_op& op : ops) {
            if (op.kind == FLUID_OP_EXCHANGE) {
                const int rc = group_exchange(cs, n_ctx, op);
                if (rc != FLUID_OK) return rc;
            } else {
                for (int r = 0; r < n_ctx; r++) {
                    HIPCK(cs[r], hipSetDevice(cs[r]->device));
                    CK(run_pass(cs[r], op, dt, P));
                }
            }
        }
    return FLUID_OK;
}

long fluid_exchange_count(const fluid_ctx* c) { return c ? c->exchanges : 0; }

}  // extern "C"

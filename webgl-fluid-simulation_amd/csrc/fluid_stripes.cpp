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
//     exchange { velocity A rows, dye Ad rows }     (A = the rows an advection back-trace may span, fluid_set_reach)
//     advect velocity + dye
//
// With H = 32 and 50 iterations: 3 exchanges per step; H >= 54: 2.  Every recomputed ghost row is the same
// arithmetic on the same inputs as its owner's, so the decomposed result is BITWISE equal to the single-domain run.
//
// Overlap: the exchanges in front of the two single-kernel pass groups (curl/vorticity/divergence, advection) run on
// a second HIP stream while the context stream computes the INTERIOR rows of that pass — the rows whose inputs are
// all owned — and only the thin strips next to the ghost rows wait for the transfer (events both ways).  The
// advection kernels count every tap outside the rows that are fresh for that launch, so a back-trace longer than
// the reach is reported (FLUID_ERR_HALO) instead of reading a row that is still in flight.
// The pressure blocks are cover as well (round 4): the leading launch(es) of a block behind an exchange — a pressure-only exchange
// between two blocks, or the step's first exchange, behind the curl pass's interior — are cut the same way (pass_jacobi, JacobiSplit):
// their interiors, one apron per launch inside the owned rectangle, compute while the ghost texels travel; the frame of each follows in
// one launch.  How many launches are cut is sized per exchange with a link model (exchange_us): a cut costs a thin frame launch.
//
// RCCL is loaded at run time (dlopen): the single-GPU path does not depend on it, and a process that already
// holds an RCCL (PyTorch bundles one) shares that copy instead of pulling in a second HIP runtime.
#include "fluid_internal.h"

#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
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

int build_plan(int halo, int dye_halo, int iterations, int advect_rows, int advect_dye_rows, std::vector<fluid_stripe_op>& ops)
{
    if (halo < 4 || dye_halo < 1 || iterations < 0) return FLUID_ERR_INVALID;
    if (advect_rows < 1 || advect_rows > halo || advect_dye_rows < 1 || advect_dye_rows > dye_halo) return FLUID_ERR_INVALID;
    const int H = halo;
    // pressure blocks: divergence is valid H-3 rows out and iteration k of a block of d needs it d-k+e rows out
    // -> d <= H-3; the last block also leaves e = 1 valid ghost row (gradient subtract reads pressure one row out)
    // How the iterations are cut into blocks (round 6): as few blocks as the ghost rows allow (every block costs an exchange), and inside that
    // BALANCED IN WHOLE LAUNCHES — the temporally blocked kernel runs ten iterations per launch at one trip of the field through memory, so a
    // block of 53 is six launches where 50 are five.  200 iterations at H = 56 were 53 + 53 + 53 + 41 = 23 launches; 4 x 50 are 20: a
    // 16384 x 2048 rank's step -9 % (profiles/r06/stripe_plan_blocks_ab.txt).  Where whole launches do not fit under the cap (H - 3) the
    // blocks are filled greedily as before.  Any cut leaves the same bits; fluid_hip/stripes.py's hosted schedule mirrors this one.
    struct Block { int d, e; };
    std::vector<Block> blocks;
    {
        const int cap = H - 3, depth = 10;
        std::vector<int> sizes;
        if (iterations > 0) {
            const int nb = (iterations + cap - 1) / cap, L = (iterations + depth - 1) / depth;
            int sum = 0;
            for (int k = 0; k < nb; k++) {
                const int d = k < nb - 1 ? (L / nb + (k < L % nb ? 1 : 0)) * depth : iterations - sum;
                sizes.push_back(d);
                sum += d;
            }
            bool ok = true;
            for (int d : sizes) ok = ok && d > 0 && d <= cap;
            if (!ok) {
                sizes.clear();
                for (int remaining = iterations; remaining > 0;) {
                    const int d = remaining < cap ? remaining : cap;
                    remaining -= d;
                    sizes.push_back(d);
                }
            }
        }
        for (size_t k = 0; k < sizes.size(); k++) blocks.push_back({ sizes[k], k + 1 == sizes.size() ? 1 : 0 });
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
    push_exchange(ops, FLUID_VELOCITY, advect_rows, FLUID_DYE, advect_dye_rows);
    push_pass(ops, FLUID_OP_ADVECT, 0, 0);
    return FLUID_OK;
}

// ghost depth of the dye field in dye rows (what `halo` sim rows are on the dye grid), whether or not this context has
// ghost ROWS at all (a 1 x N tile set only has ghost columns)
int dye_ghost_depth(const fluid_ctx* c) { return (int)(((long)c->desc.halo * c->dye.H + c->sim.H - 1) / c->sim.H); }

// rows refreshed in front of the advection: the reach of a back-trace (one more when the dye grid differs from the
// sim grid: the dye pass samples the NEW velocity bilinearly, so that is advected one ghost row out), and the same
// distance in dye rows
void advect_rows(const fluid_ctx* c, int* vel_rows, int* dye_rows)
{
    const bool same = c->sim.W == c->dye.W && c->sim.H == c->dye.H;
    int va = c->reach + (same ? 0 : 1);
    if (va > c->desc.halo) va = c->desc.halo;
    long vd = same ? va : ((long)c->reach * c->dye.H + c->sim.H - 1) / c->sim.H + 1;
    if (vd > dye_ghost_depth(c)) vd = dye_ghost_depth(c);
    if (vd < 1) vd = 1;
    if (va < 1) va = 1;
    *vel_rows = va;
    *dye_rows = (int)vd;
}

int dye_format_agree_end(fluid_ctx* c);

int run_pass(fluid_ctx* c, const fluid_stripe_op& op, float dt, const fluid_params* P)
{
    if (op.kind == FLUID_OP_ADVECT) CK(dye_format_agree_end(c));   // (a plan whose advection follows no dye exchange: the format is the set's all the same)
    switch (op.kind) {
    case FLUID_OP_CURL_VORT_DIV: return pass_curl_vort_div(c, P->curl, dt, op.ext, nullptr);
    case FLUID_OP_CLEAR: return pass_clear(c, P->pressure, op.ext);
    case FLUID_OP_CLEAR_JACOBI: return pass_clear_jacobi(c, P->pressure, op.iters, op.ext, nullptr, nullptr);
    case FLUID_OP_JACOBI: return pass_jacobi(c, op.iters, op.ext, 1.0f, nullptr, nullptr, nullptr);
    case FLUID_OP_GRADSUB: return pass_gradsub(c, op.ext);
    case FLUID_OP_ADVECT: return pass_advect(c, dt, P->velocity_dissipation, P->density_dissipation, nullptr);
    default: return c->fail(FLUID_ERR_INVALID, "unknown stripe op");
    }
}

// The last pressure block followed by the gradient subtract (plan: ... JACOBI(d, e = 1), GRADSUB(0)): K6 rides on the block's last
// launch where that launch has the instantiation (pass_jacobi), otherwise the two ops run in turn.  Same bits either way.
bool folds_gradsub(const std::vector<fluid_stripe_op>& ops, size_t i)
{
    return (ops[i].kind == FLUID_OP_CLEAR_JACOBI || ops[i].kind == FLUID_OP_JACOBI) && ops[i].iters > 0 && i + 1 < ops.size() &&
           ops[i + 1].kind == FLUID_OP_GRADSUB && ops[i + 1].ext == 0 && ops[i].ext >= 1;
}

int run_block_and_gradsub(fluid_ctx* c, const fluid_stripe_op& op, const fluid_stripe_op& gs, const fluid_params* P)
{
    bool folded = true;
    if (op.kind == FLUID_OP_CLEAR_JACOBI) CK(pass_clear_jacobi(c, P->pressure, op.iters, op.ext, nullptr, &folded));
    else CK(pass_jacobi(c, op.iters, op.ext, 1.0f, nullptr, &folded, nullptr));
    return folded ? (int)FLUID_OK : pass_gradsub(c, gs.ext);
}

// ghost / owned row blocks of one exchange item: addresses and bytes (fp32 or fp16 texels: the exchange moves bytes)
struct Rows {
    char *send_lo, *recv_lo, *send_hi, *recv_hi;
    size_t bytes;
};

int rows_of(fluid_ctx* c, int field, int n, Rows* r)
{
    FieldRef f;
    CK(field_ref(c, field, &f, false, true));   // the dye in the format it is in (dye_prepare ran in front of the exchange): 12-byte texels while packed
    if (n < 1 || n > f.halo || n > f.rows) return c->fail(FLUID_ERR_INVALID, "exchange rows exceed the ghost rows / the stripe");
    const size_t rowf = (size_t)f.win->P * f.texel();  // bytes per array row (pitch)
    char* base = (char*)f.ptr;
    const int h = f.halo, rr = f.rows;
    r->send_lo = base + (size_t)h * rowf;             // my lowest owned rows   -> lower neighbour's top ghost rows
    r->recv_lo = base + (size_t)(h - n) * rowf;       // my bottom ghost rows   <- lower neighbour's highest owned rows
    r->send_hi = base + (size_t)(h + rr - n) * rowf;  // my highest owned rows  -> upper neighbour's bottom ghost rows
    r->recv_hi = base + (size_t)(h + rr) * rowf;      // my top ghost rows      <- upper neighbour's lowest owned rows
    r->bytes = (size_t)n * rowf;
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
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
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
            const char* e = dlerror();  // a second call would return NULL: read it once
            tried += n + ": " + (e ? e : "?") + "; ";
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
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
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

// ---- exchanges -------------------------------------------------------------------------------------------------
// An exchange is begun (transfers enqueued on the comm stream, after everything the context stream has produced so
// far) and ended (the context stream waits for the ghost rows).  Between the two the context stream may compute
// whatever does not read the ghost rows in flight.  begin + end back to back is the plain synchronous exchange.
int ensure_comm_stream(fluid_ctx* c)
{
    // A context that exchanges over RCCL (one process per GPU) gets its comm stream at the HIGHEST stream priority: what runs there — the
    // pack / unpack launches of a tile exchange, RCCL's own send / receive kernels — is small and sits on the step's critical path, while
    // the context stream floods the chip with the interior rows of the next pass.  At equal priority those few workgroups wait for the big
    // launch's workgroups to retire one by one (every register of a CU is taken), and a 60 us link cost the centre tile of 3 x 3 +29 %
    // per step instead of +24.5 % (profiles/r04/overlap_vs_link_latency_visit6_two_rounds.txt).
    // The contexts of an in-process group (fluid_group_step_n: a whole stripe set on ONE device) keep the default priority: a priority
    // stream takes hardware queues of its own, and four contexts with two queues each oversubscribe them — the same group stepped 21 %
    // slower with overlap and 74 % slower without (2.02 -> 2.44 / 2.00 -> 3.48 ms, profiles/r04/group_priority_streams.txt).
    const bool high = c->comm != nullptr;
    if (c->comm_stream && c->comm_stream_high == high) return FLUID_OK;
    HIPCK(c, hipSetDevice(c->device));
    if (c->comm_stream) {   // created before the communicator was (fluid_set_overlap): once more, at the other priority
        HIPCK(c, hipStreamSynchronize(c->comm_stream));
        HIPCK(c, hipStreamDestroy(c->comm_stream));
        c->comm_stream = nullptr;
    }
    int prio_least = 0, prio_greatest = 0;
    HIPCK(c, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    if (high) HIPCK(c, hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, prio_greatest));
    else HIPCK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    c->comm_stream_high = high;
    if (c->ev_ready) return FLUID_OK;   // the events outlive the stream
    // FLUID_EVENT_SCOPE=device (lab build): the three events without a system-scope fence (hipEventDisableSystemFence).  On one GPU that
    // removes the 6-7 us of idle each record / wait on the context stream costs (profiles/r04/stripe_rank_timeline.txt); whether it is
    // safe between GPUs depends on RCCL never letting the peer device touch the field memory directly, which no one-GPU test can see —
    // a probe for the first multi-GPU box, not a product setting.
    unsigned ev_flags = hipEventDisableTiming;
    if (const char* e = fluid::lab_env("FLUID_EVENT_SCOPE"))
        if (e[0] == 'd') ev_flags |= hipEventDisableSystemFence;
    HIPCK(c, hipEventCreateWithFlags(&c->ev_ready, ev_flags));
    HIPCK(c, hipEventCreateWithFlags(&c->ev_landed, ev_flags));
    HIPCK(c, hipEventCreateWithFlags(&c->ev_mid, ev_flags));
    // between the two streams of ONE device, about texels this device wrote: no system-scope fence (as the step marks: fluid_solver.cpp).
    // ev_joined hands the context stream what the comm stream did behind the exchange; the ghost texels themselves were acquired at system
    // scope by the comm stream's own wait for ev_landed (comm_has_landed), in front of the first kernel that read them.  With a system-scope
    // ev_joined the thin launches on the comm stream cost a slow link one more 6-7 us fence per exchange than round 4's schedule
    // (profiles/r05/stripe_rank_strips_on_comm.txt, first form: the centre tile at 60 us per exchange +1 ... 2.5 %).
    HIPCK(c, hipEventCreateWithFlags(&c->ev_joined, hipEventDisableTiming | hipEventDisableSystemFence));
    HIPCK(c, hipEventCreateWithFlags(&c->ev_inner, hipEventDisableTiming | hipEventDisableSystemFence));
    if (const char* e = fluid::lab_env("FLUID_STRIPE_OVERLAP")) c->overlap = atoi(e) != 0;
    return FLUID_OK;
}

int rccl_exchange_begin(fluid_ctx* c, const fluid_stripe_op& op)
{
    const Rccl* R = rccl(nullptr);
    if (!R || !c->comm) return c->fail(FLUID_ERR_COMM, "stripe context has no communicator (fluid_comm_init)");
    ncclComm_t comm = (ncclComm_t)c->comm;
    const int rank = c->desc.part, world = c->desc.parts;
    Rows rows[2];
    for (int i = 0; i < op.n_items; i++) CK(rows_of(c, op.field[i], op.rows[i], &rows[i]));
    HIPCK(c, hipEventRecord(c->ev_ready, c->stream));
    HIPCK(c, hipStreamWaitEvent(c->comm_stream, c->ev_ready, 0));
    NCCLCK(c, R, R->GroupStart());
    // per peer, sends and receives are issued in item order on both sides, so they pair up
    if (rank > 0)
        for (int i = 0; i < op.n_items; i++) {
            NCCLCK(c, R, R->Send(rows[i].send_lo, rows[i].bytes, ncclChar, rank - 1, comm, c->comm_stream));
            NCCLCK(c, R, R->Recv(rows[i].recv_lo, rows[i].bytes, ncclChar, rank - 1, comm, c->comm_stream));
        }
    if (rank < world - 1)
        for (int i = 0; i < op.n_items; i++) {
            NCCLCK(c, R, R->Send(rows[i].send_hi, rows[i].bytes, ncclChar, rank + 1, comm, c->comm_stream));
            NCCLCK(c, R, R->Recv(rows[i].recv_hi, rows[i].bytes, ncclChar, rank + 1, comm, c->comm_stream));
        }
    NCCLCK(c, R, R->GroupEnd());
    HIPCK(c, hipEventRecord(c->ev_landed, c->comm_stream));
    c->exchanges++;
    return FLUID_OK;
}

int rccl_exchange_end(fluid_ctx* c)
{
    HIPCK(c, hipStreamWaitEvent(c->stream, c->ev_landed, 0));
    return FLUID_OK;
}

// The same exchange between stripe contexts that live in ONE process (the whole stripe set, driven by one host
// thread): device-to-device copies on each context's comm stream, ordered with the same two events per context.
// This is how the plan, the interior / strip split and the windowed kernels are validated bit for bit on a
// single-GPU box; bench.py --gpus N uses the RCCL path.
int group_exchange_begin(fluid_ctx** cs, int n, const fluid_stripe_op& op)
{
    for (int r = 0; r < n; r++) {
        HIPCK(cs[r], hipSetDevice(cs[r]->device));
        HIPCK(cs[r], hipEventRecord(cs[r]->ev_ready, cs[r]->stream));
    }
    for (int r = 0; r < n; r++) {
        fluid_ctx* c = cs[r];
        HIPCK(c, hipSetDevice(c->device));
        HIPCK(c, hipStreamWaitEvent(c->comm_stream, c->ev_ready, 0));
        if (r > 0) HIPCK(c, hipStreamWaitEvent(c->comm_stream, cs[r - 1]->ev_ready, 0));
        if (r < n - 1) HIPCK(c, hipStreamWaitEvent(c->comm_stream, cs[r + 1]->ev_ready, 0));
        for (int i = 0; i < op.n_items; i++) {
            Rows me, lo, hi;
            CK(rows_of(c, op.field[i], op.rows[i], &me));
            if (r > 0) {
                CK(rows_of(cs[r - 1], op.field[i], op.rows[i], &lo));
                HIPCK(c, hipMemcpyAsync(me.recv_lo, lo.send_hi, me.bytes, hipMemcpyDeviceToDevice, c->comm_stream));
            }
            if (r < n - 1) {
                CK(rows_of(cs[r + 1], op.field[i], op.rows[i], &hi));
                HIPCK(c, hipMemcpyAsync(me.recv_hi, hi.send_lo, me.bytes, hipMemcpyDeviceToDevice, c->comm_stream));
            }
        }
        HIPCK(c, hipEventRecord(c->ev_landed, c->comm_stream));
        c->exchanges++;
    }
    return FLUID_OK;
}

int group_exchange_end(fluid_ctx** cs, int n)
{
    // own ghost rows have landed, and no neighbour is still copying out of rows this stripe may overwrite next
    for (int r = 0; r < n; r++) {
        fluid_ctx* c = cs[r];
        HIPCK(c, hipSetDevice(c->device));
        HIPCK(c, hipStreamWaitEvent(c->stream, c->ev_landed, 0));
        if (r > 0) HIPCK(c, hipStreamWaitEvent(c->stream, cs[r - 1]->ev_landed, 0));
        if (r < n - 1) HIPCK(c, hipStreamWaitEvent(c->stream, cs[r + 1]->ev_landed, 0));
    }
    return FLUID_OK;
}

// ---- 2-D tiles (parts_x > 1): ghost COLUMNS as well ---------------------------------------------------------------
// Rank = part * parts_x + part_x.  The blocks are strided in the tile's arrays (pitch = owned + ghost columns): they travel through
// contiguous staging buffers, packed and unpacked by ONE kernel launch each (launch_copy_rects: every field and direction).  One
// message per neighbour carries all the fields of the exchange.  The interior-first overlap works as for stripes, with four strips
// around the interior.
//   * over RCCL (rccl_exchange_2d_begin): ONE message round to up to EIGHT neighbours — ghost columns from LEFT / RIGHT, ghost rows
//     from DOWN / UP over the owned columns, and the four corner blocks straight from the diagonal neighbours;
//   * inside one process (group_exchange_2d_begin): two phases — A moves ghost columns between left / right neighbours (owned rows
//     only), B then moves ghost rows between lower / upper neighbours over the owned columns PLUS the ghost columns phase A just
//     filled, so the corner blocks arrive without diagonal copies.
// Same bytes into the same ghost texels either way (a corner block of the diagonal neighbour's OWNED texels; tests/test_stripes_gpu.py
// holds both drivers to the single domain bit for bit).
struct Rect {
    char* p;             // first texel
    size_t pitch, line;  // bytes between rows, bytes per row of the block
    int nrows;
    size_t bytes() const { return line * (size_t)nrows; }
};

enum Dir { LEFT = 0, RIGHT = 1, DOWN = 2, UP = 3, DOWN_LEFT = 4, DOWN_RIGHT = 5, UP_LEFT = 6, UP_RIGHT = 7 };

struct Blocks {
    Rect send[4], recv[4];   // the in-process driver's two phases (DOWN / UP span owned + ghost columns)
    // the RCCL driver's single round: `core` = DOWN / UP over the owned columns only, `diag` = the corner blocks (index dir - DOWN_LEFT):
    // nx x n texels of this tile's OWNED corner out, the matching corner of its ghost frame in
    Rect send_core[4], recv_core[4];
    Rect send_diag[4], recv_diag[4];
};

int col_depth(const fluid_ctx* c, const FieldRef& f, int n)  // ghost columns that go with n ghost rows of this field
{
    if (f.win != &c->dye) return n < f.halo_x ? n : f.halo_x;      // sim fields: same depth both ways
    const long num = (long)n * c->dye.W * c->sim.H, den = (long)c->dye.H * c->sim.W;  // dye: same physical distance
    long nx = (num + den - 1) / den + 1;
    if (nx > f.halo_x) nx = f.halo_x;
    return (int)nx;
}

int blocks_of(fluid_ctx* c, int field, int n, Blocks* b)
{
    FieldRef f;
    CK(field_ref(c, field, &f, false, true));   // (as rows_of)
    if (n < 1 || (c->desc.parts > 1 && (n > f.halo || n > f.rows))) return c->fail(FLUID_ERR_INVALID, "exchange rows exceed the ghost rows / the tile");
    const size_t texel = f.texel(), pitch = (size_t)f.win->P * texel;
    const int nx = col_depth(c, f, n);
    if (nx < 1 || nx > f.cols) return c->fail(FLUID_ERR_INVALID, "exchange columns exceed the ghost columns / the tile");
    const int h = f.halo, R = f.rows, c0 = f.col0, c1 = f.col0 + f.cols;
    auto rect = [&](int arow, int nrows, int col, int ncols) {
        return Rect{ (char*)f.ptr + (size_t)arow * pitch + (size_t)(col - f.win->c0) * texel, pitch, (size_t)ncols * texel, nrows };  // col: global
    };
    // phase A: owned rows, columns next to the left / right tile border
    b->send[LEFT] = rect(h, R, c0, nx);
    b->recv[LEFT] = rect(h, R, c0 - nx, nx);
    b->send[RIGHT] = rect(h, R, c1 - nx, nx);
    b->recv[RIGHT] = rect(h, R, c1, nx);
    // phase B: owned + freshly received ghost columns, rows next to the lower / upper tile border
    const int cs = c->desc.part_x > 0 ? c0 - nx : c0, ce = c->desc.part_x < c->desc.parts_x - 1 ? c1 + nx : c1;
    b->send[DOWN] = rect(h, n, cs, ce - cs);
    b->recv[DOWN] = rect(h - n, n, cs, ce - cs);
    b->send[UP] = rect(h + R - n, n, cs, ce - cs);
    b->recv[UP] = rect(h + R, n, cs, ce - cs);
    b->send_core[DOWN] = rect(h, n, c0, c1 - c0);
    b->recv_core[DOWN] = rect(h - n, n, c0, c1 - c0);
    b->send_core[UP] = rect(h + R - n, n, c0, c1 - c0);
    b->recv_core[UP] = rect(h + R, n, c0, c1 - c0);
    b->send_diag[DOWN_LEFT - DOWN_LEFT] = rect(h, n, c0, nx);
    b->recv_diag[DOWN_LEFT - DOWN_LEFT] = rect(h - n, n, c0 - nx, nx);
    b->send_diag[DOWN_RIGHT - DOWN_LEFT] = rect(h, n, c1 - nx, nx);
    b->recv_diag[DOWN_RIGHT - DOWN_LEFT] = rect(h - n, n, c1, nx);
    b->send_diag[UP_LEFT - DOWN_LEFT] = rect(h + R - n, n, c0, nx);
    b->recv_diag[UP_LEFT - DOWN_LEFT] = rect(h + R, n, c0 - nx, nx);
    b->send_diag[UP_RIGHT - DOWN_LEFT] = rect(h + R - n, n, c1 - nx, nx);
    b->recv_diag[UP_RIGHT - DOWN_LEFT] = rect(h + R, n, c1, nx);
    return FLUID_OK;
}

bool has_neighbour(const fluid_ctx* c, int dir)
{
    const fluid_desc& d = c->desc;
    const bool l = d.part_x > 0, r = d.part_x < d.parts_x - 1, dn = d.part > 0, up = d.part < d.parts - 1;
    switch (dir) {
    case LEFT: return l;
    case RIGHT: return r;
    case DOWN: return dn;
    case UP: return up;
    case DOWN_LEFT: return dn && l;
    case DOWN_RIGHT: return dn && r;
    case UP_LEFT: return up && l;
    default: return up && r;
    }
}

int neighbour_rank(const fluid_ctx* c, int dir)
{
    const int r = c->desc.part * c->desc.parts_x + c->desc.part_x, px = c->desc.parts_x;
    switch (dir) {
    case LEFT: return r - 1;
    case RIGHT: return r + 1;
    case DOWN: return r - px;
    case UP: return r + px;
    case DOWN_LEFT: return r - px - 1;
    case DOWN_RIGHT: return r - px + 1;
    case UP_LEFT: return r + px - 1;
    default: return r + px + 1;
    }
}

// `from`: the context that owns the source block.  On the same device the copy is one k_copy_rects launch on c's comm stream; a
// neighbour on ANOTHER device is not dereferenced from a kernel (nothing here enables peer access): the runtime's 2-D copy moves the
// block, with or without peer access between the two GPUs.
int copy_rect(fluid_ctx* c, const fluid_ctx* from, const Rect& dst, const Rect& src, hipStream_t s)
{
    if (dst.line != src.line || dst.nrows != src.nrows) return c->fail(FLUID_ERR_INVALID, "exchange blocks of neighbouring tiles differ in shape");
    if (from->device != c->device) {
        HIPCK(c, hipMemcpy2DAsync(dst.p, dst.pitch, src.p, src.pitch, src.line, (size_t)src.nrows, hipMemcpyDeviceToDevice, s));
        return FLUID_OK;
    }
    fluid::CopyRects one{};
    one.n = 1;
    one.r[0] = fluid::copy_rect_of(src.p, dst.p, src.pitch, dst.pitch, src.line, src.nrows);
    HIPCK(c, fluid::launch_copy_rects(s, one));
    return FLUID_OK;
}

int ensure_stage(fluid_ctx* c, int slot, size_t bytes)
{
    if (c->stage_bytes[slot] >= bytes) return FLUID_OK;
    if (c->stage[slot]) (void)hipFree(c->stage[slot]);
    c->stage[slot] = nullptr;
    c->stage_bytes[slot] = 0;
    HIPCK(c, hipMalloc(&c->stage[slot], bytes));
    c->stage_bytes[slot] = bytes;
    return FLUID_OK;
}

// The whole exchange over RCCL on the comm stream (begun here, ended by rccl_exchange_end): pack, ONE ncclGroup with a send and a
// receive per neighbour — up to eight of them, every link side by side — unpack.  (Round 4 started with the two-phase form on this
// path too: columns, THEN rows over owned + fresh ghost columns put two full link times on every exchange, and against a link of 60 us
// the centre tile of 3 x 3 paid +25 % per step; a second, tiny round that forwarded the corner blocks still paid the latency twice —
// profiles/r04/overlap_vs_link_latency.txt.  The corner blocks now come straight from the diagonal neighbours: on an xGMI node every
// pair of GPUs has its own link.)
int rccl_exchange_2d_begin(fluid_ctx* c, const fluid_stripe_op& op)
{
    const Rccl* R = rccl(nullptr);
    if (!R || !c->comm) return c->fail(FLUID_ERR_COMM, "tile context has no communicator (fluid_comm_init)");
    ncclComm_t comm = (ncclComm_t)c->comm;
    Blocks blk[2];
    for (int i = 0; i < op.n_items; i++) CK(blocks_of(c, op.field[i], op.rows[i], &blk[i]));
    HIPCK(c, hipEventRecord(c->ev_ready, c->stream));
    HIPCK(c, hipStreamWaitEvent(c->comm_stream, c->ev_ready, 0));
    size_t total[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    fluid::CopyRects pack{}, unpack{};
    auto rects_of = [&](int dir, int i, const Rect** sr, const Rect** rr) {
        *sr = dir >= DOWN_LEFT ? &blk[i].send_diag[dir - DOWN_LEFT] : dir >= DOWN ? &blk[i].send_core[dir] : &blk[i].send[dir];
        *rr = dir >= DOWN_LEFT ? &blk[i].recv_diag[dir - DOWN_LEFT] : dir >= DOWN ? &blk[i].recv_core[dir] : &blk[i].recv[dir];
    };
    auto aligned = [](size_t off) { return (off + 15) & ~(size_t)15; };   // every block starts on 16 bytes of the staging: wide copies
    for (int dir = 0; dir < 8; dir++) {
        if (!has_neighbour(c, dir)) continue;
        const Rect *sr, *rr;
        // the message of one direction: the fields of the exchange in item order — both ends cut the same shapes, hence the same layout
        for (int i = 0; i < op.n_items; i++) {
            rects_of(dir, i, &sr, &rr);
            total[dir] = aligned(total[dir]) + sr->bytes();
        }
        if (!total[dir]) continue;
        CK(ensure_stage(c, 2 * dir, total[dir]));      // send staging of this direction
        CK(ensure_stage(c, 2 * dir + 1, total[dir]));  // receive staging (the neighbour's blocks have the same shapes)
        size_t off = 0;
        for (int i = 0; i < op.n_items; i++) {
            rects_of(dir, i, &sr, &rr);
            off = aligned(off);
            if (pack.n >= 16) return c->fail(FLUID_ERR_INVALID, "tile exchange: more blocks than one copy launch takes");
            pack.r[pack.n++] = fluid::copy_rect_of(sr->p, (char*)c->stage[2 * dir] + off, sr->pitch, sr->line, sr->line, sr->nrows);
            unpack.r[unpack.n++] = fluid::copy_rect_of((char*)c->stage[2 * dir + 1] + off, rr->p, rr->line, rr->pitch, rr->line, rr->nrows);
            off += sr->bytes();
        }
    }
    if (pack.n) {
        HIPCK(c, fluid::launch_copy_rects(c->comm_stream, pack));      // every field and direction: one launch
        NCCLCK(c, R, R->GroupStart());
        for (int dir = 0; dir < 8; dir++)
            if (total[dir]) {
                NCCLCK(c, R, R->Send(c->stage[2 * dir], total[dir], ncclChar, neighbour_rank(c, dir), comm, c->comm_stream));
                NCCLCK(c, R, R->Recv(c->stage[2 * dir + 1], total[dir], ncclChar, neighbour_rank(c, dir), comm, c->comm_stream));
            }
        NCCLCK(c, R, R->GroupEnd());
        HIPCK(c, fluid::launch_copy_rects(c->comm_stream, unpack));
    }
    HIPCK(c, hipEventRecord(c->ev_landed, c->comm_stream));
    c->exchanges++;
    return FLUID_OK;  // rccl_exchange_end() makes the context stream wait for it
}

// the same exchange for a whole tile set inside one process: direct rectangle copies, phase B after the neighbours' phase A
int group_exchange_2d_begin(fluid_ctx** cs, int n, const fluid_stripe_op& op)
{
    const int px = cs[0]->desc.parts_x;
    std::vector<Blocks> blk((size_t)n * 2);
    for (int r = 0; r < n; r++) {
        HIPCK(cs[r], hipSetDevice(cs[r]->device));
        for (int i = 0; i < op.n_items; i++) CK(blocks_of(cs[r], op.field[i], op.rows[i], &blk[(size_t)r * 2 + i]));
        HIPCK(cs[r], hipEventRecord(cs[r]->ev_ready, cs[r]->stream));
    }
    auto wait_neighbours = [&](int r, hipStream_t s, bool phase_a_done) -> int {
        fluid_ctx* c = cs[r];
        for (int dir = 0; dir < 4; dir++)
            if (has_neighbour(c, dir)) {
                fluid_ctx* o = cs[neighbour_rank(c, dir)];
                HIPCK(c, hipStreamWaitEvent(s, phase_a_done ? o->ev_mid : o->ev_ready, 0));
            }
        return FLUID_OK;
    };
    for (int r = 0; r < n; r++) {  // phase A
        fluid_ctx* c = cs[r];
        HIPCK(c, hipSetDevice(c->device));
        HIPCK(c, hipStreamWaitEvent(c->comm_stream, c->ev_ready, 0));
        CK(wait_neighbours(r, c->comm_stream, false));
        for (int i = 0; i < op.n_items; i++) {
            if (has_neighbour(c, LEFT)) CK(copy_rect(c, cs[r - 1], blk[(size_t)r * 2 + i].recv[LEFT], blk[(size_t)(r - 1) * 2 + i].send[RIGHT], c->comm_stream));
            if (has_neighbour(c, RIGHT)) CK(copy_rect(c, cs[r + 1], blk[(size_t)r * 2 + i].recv[RIGHT], blk[(size_t)(r + 1) * 2 + i].send[LEFT], c->comm_stream));
        }
        HIPCK(c, hipEventRecord(c->ev_mid, c->comm_stream));
    }
    for (int r = 0; r < n; r++) {  // phase B
        fluid_ctx* c = cs[r];
        HIPCK(c, hipSetDevice(c->device));
        CK(wait_neighbours(r, c->comm_stream, true));
        for (int i = 0; i < op.n_items; i++) {
            if (has_neighbour(c, DOWN)) CK(copy_rect(c, cs[r - px], blk[(size_t)r * 2 + i].recv[DOWN], blk[(size_t)(r - px) * 2 + i].send[UP], c->comm_stream));
            if (has_neighbour(c, UP)) CK(copy_rect(c, cs[r + px], blk[(size_t)r * 2 + i].recv[UP], blk[(size_t)(r + px) * 2 + i].send[DOWN], c->comm_stream));
        }
        HIPCK(c, hipEventRecord(c->ev_landed, c->comm_stream));
        c->exchanges++;
    }
    return FLUID_OK;
}

int group_exchange_2d_end(fluid_ctx** cs, int n)
{
    for (int r = 0; r < n; r++) {  // own blocks have landed; no neighbour still copies out of this tile
        fluid_ctx* c = cs[r];
        HIPCK(c, hipSetDevice(c->device));
        HIPCK(c, hipStreamWaitEvent(c->stream, c->ev_landed, 0));
        for (int dir = 0; dir < 4; dir++)
            if (has_neighbour(c, dir)) HIPCK(c, hipStreamWaitEvent(c->stream, cs[neighbour_rank(c, dir)]->ev_landed, 0));
    }
    return FLUID_OK;
}

// ---- interior-first forms of the two single-kernel pass groups ----------------------------------------------------
// rows of the band [ga, gb) that do not depend on ghost rows when every output row reads `dep` rows on each side
struct Split {
    int ga, gb, ia, ib;  // rows: whole band, interior
    int xa, xb, ja, jb;  // columns: whole band, interior (the whole width unless 2-D tiles)
};

// the band of a pass (owned rows / columns +- ext) and the part of it whose inputs are all owned when every output texel
// reads `dep` texels to each side (`depx` columns: whole float4 groups for the register-tile kernels)
Split split_band(const fluid_ctx* c, int ext, int dep, int depx)
{
    Split s;
    sim_band(c, ext, s.ga, s.gb);
    const Win w = sim_cols(c, ext);
    s.xa = w.x0;
    s.xb = w.x1;
    const fluid_desc& d = c->desc;
    const int r0 = c->sim_row0, r1 = r0 + c->sim_rows, c0 = c->sim_col0, c1 = c0 + c->sim_ncols;
    s.ia = d.part > 0 ? r0 + dep : s.ga;  // the bottom stripe has no lower neighbour: its low rows are interior
    s.ib = d.part < d.parts - 1 ? r1 - dep : s.gb;
    s.ja = d.part_x > 0 ? c0 + depx : s.xa;
    s.jb = d.part_x < d.parts_x - 1 ? c1 - depx : s.xb;
    if (s.ia > s.ib || s.ja > s.jb) {  // tile thinner than 2 * dep: no interior
        s.ia = s.ib = s.ga;
        s.ja = s.jb = s.xa;
    }
    return s;
}

int exchanged_reach(const fluid_ctx* c)  // velocity rows the exchange in front of the advection refreshes
{
    int va, vd;
    advect_rows(c, &va, &vd);
    return va;
}

bool overlap_ok(const fluid_ctx* c, const fluid_stripe_op& pass)
{
    if (!c->overlap) return false;
    const int A = exchanged_reach(c);
    const bool tiles = c->desc.parts_x > 1;
    if (pass.kind == FLUID_OP_CURL_VORT_DIV) return fused_cvd_applies(c) && c->sim_rows > 6 && (!tiles || c->sim_ncols > 8);
    if (pass.kind == FLUID_OP_ADVECT) return fused_advect_applies(c) && c->sim_rows > 2 * A && (!tiles || c->sim_ncols > 2 * A);
    return false;
}

// the four strips of a band around its interior: bottom and top over the full column range, left and right beside the interior
template <class F>
int for_each_strip(const Split& s, F&& launch)
{
    if (s.ia > s.ga) CK(launch(s.ga, s.ia, s.xa, s.xb));
    if (s.gb > s.ib) CK(launch(s.ib, s.gb, s.xa, s.xb));
    if (s.ib > s.ia && s.ja > s.xa) CK(launch(s.ia, s.ib, s.xa, s.ja));
    if (s.ib > s.ia && s.xb > s.jb) CK(launch(s.ia, s.ib, s.jb, s.xb));
    return FLUID_OK;
}

int pass_interior(fluid_ctx* c, const fluid_stripe_op& op, float dt, const fluid_params* P)
{
    if (op.kind == FLUID_OP_CURL_VORT_DIV) {
        const Split s = split_band(c, op.ext, 3, 4);  // a divergence texel reads velocity 3 texels away (4: float4 groups)
        return s.ib > s.ia && s.jb > s.ja ? cvd_band(c, P->curl, dt, s.ia, s.ib, s.ja, s.jb) : FLUID_OK;
    }
    const int A = exchanged_reach(c);
    const Split s = split_band(c, 0, A, A);  // an advected texel gathers from at most `reach` texels away ...
    const int r0 = c->sim_row0, r1 = r0 + c->sim_rows, c0 = c->sim_col0, c1 = c0 + c->sim_ncols;
    // ... and is held to it: only the owned rows and columns count as fresh for this launch
    return s.ib > s.ia && s.jb > s.ja
               ? advect_both_band(c, dt, P->velocity_dissipation, P->density_dissipation, s.ia, s.ib, s.ja, s.jb, r0, r1, c0, c1)
               : FLUID_OK;
}

// FLUID_STRIP_RECTS=0: one launch per strip, as in round 2 (A/B knob; same bits)
bool strip_rects_enabled()
{
    static const bool on = [] {
        const char* e = fluid::lab_env("FLUID_STRIP_RECTS");
        return !(e && atoi(e) == 0);
    }();
    return on;
}

int pass_strips(fluid_ctx* c, const fluid_stripe_op& op, float dt, const fluid_params* P)
{
    // the (up to four) strips of the band go into ONE launch (launch_*_rects): each is a few rows or columns of the tile
    fluid::BandRects B{};
    auto collect = [&](int ga, int gb, int xa, int xb) {
        B.r[B.n++] = fluid::BandRect{ xa, xb, ga, gb };
        return (int)FLUID_OK;
    };
    if (op.kind == FLUID_OP_CURL_VORT_DIV) {
        const Split s = split_band(c, op.ext, 3, 4);
        if (strip_rects_enabled()) {
            CK(for_each_strip(s, collect));
            CK(cvd_rects(c, P->curl, dt, B));
        } else {
            CK(for_each_strip(s, [&](int ga, int gb, int xa, int xb) { return cvd_band(c, P->curl, dt, ga, gb, xa, xb); }));
        }
        cvd_swap(c);
        return FLUID_OK;
    }
    const int A = exchanged_reach(c);
    const Split s = split_band(c, 0, A, A);
    // fresh now: owned + the rows / columns just exchanged
    const int v0 = c->sim_row0 - A, v1 = c->sim_row0 + c->sim_rows + A;
    const int u0 = c->desc.parts_x > 1 ? c->sim_col0 - A : 0, u1 = c->desc.parts_x > 1 ? c->sim_col0 + c->sim_ncols + A : c->sim.W;
    if (strip_rects_enabled()) {
        CK(for_each_strip(s, collect));
        CK(advect_both_rects(c, dt, P->velocity_dissipation, P->density_dissipation, B, v0, v1, u0, u1));
    } else {
        CK(for_each_strip(s, [&](int ga, int gb, int xa, int xb) {
            return advect_both_band(c, dt, P->velocity_dissipation, P->density_dissipation, ga, gb, xa, xb, v0, v1, u0, u1);
        }));
    }
    advect_both_note(c, dt, P->density_dissipation);   // interior + strips = this step's one advection of the dye
    advect_both_swap(c);
    return FLUID_OK;
}

// whole pass after a synchronous exchange; the advection windows are narrowed to the rows that were refreshed
int pass_whole(fluid_ctx* c, const fluid_stripe_op& op, float dt, const fluid_params* P)
{
    if (op.kind != FLUID_OP_ADVECT) return run_pass(c, op, dt, P);
    const Win sim = c->sim, dye = c->dye;
    int va, vd;
    advect_rows(c, &va, &vd);
    c->sim.v0 = c->sim_row0 - va;
    c->sim.v1 = c->sim_row0 + c->sim_rows + va;
    c->dye.v0 = c->dye_row0 - vd;
    c->dye.v1 = c->dye_row0 + c->dye_rows + vd;
    if (c->desc.parts_x > 1) {  // 2-D tiles: the same for the columns
        FieldRef fv, fd;
        CK(field_ref(c, FLUID_VELOCITY, &fv, true));
        CK(field_ref(c, FLUID_DYE, &fd, true));   // geometry only: asking for the dye's MEMORY here unpacked a packed field every step (2 x 2 tiles without overlap: +30 %)
        const int ax = col_depth(c, fv, va), dx = col_depth(c, fd, vd);
        c->sim.u0 = c->sim_col0 - ax;
        c->sim.u1 = c->sim_col0 + c->sim_ncols + ax;
        c->dye.u0 = c->dye_col0 - dx;
        c->dye.u1 = c->dye_col0 + c->dye_ncols + dx;
    }
    const int rc = run_pass(c, op, dt, P);
    c->sim = sim;
    c->dye = dye;
    return rc;
}

// A pressure-only exchange in front of a further Jacobi block (more iterations than the ghost rows carry: BASELINE configs[4] runs 200):
// nothing but that block follows, so the block's FIRST launch is cut — the rows whose inputs are all owned compute while the ghost rows
// travel, the strips next to them and every further launch follow (pass_jacobi split 1 / 2).  Round 4: against a link of 60 us per
// exchange the three such exchanges of a 200-iteration step were fully exposed (profiles/r04/overlap_vs_link_latency.txt).
bool jacobi_overlap_ok(const fluid_ctx* c, const std::vector<fluid_stripe_op>& ops, size_t i)
{
    if (!c->overlap || i + 1 >= ops.size() || ops[i].kind != FLUID_OP_EXCHANGE || ops[i].n_items != 1 || ops[i].field[0] != FLUID_PRESSURE) return false;
    return ops[i + 1].kind == FLUID_OP_JACOBI && jacobi_split_ok(c, ops[i + 1].iters, folds_gradsub(ops, i + 1));
}

// ---- how much cover an exchange needs --------------------------------------------------------------------------------
// A cut launch costs a thin frame launch (~10 us at 4096^2 per rank: one tile's latency), so a block's launches are cut only as far as the
// exchange is expected to last: its largest message over ONE link (the neighbours' links run side by side) at an assumed
// 20 us + bytes / 50 GB/s — the order of an RCCL point-to-point message of a few MB over one xGMI link — plus the pack / unpack launches
// of a tile exchange, against what a launch takes at the rates this library measures on the MI355X (profiles/r04: the Jacobi launch moves
// 13.2 B/texel at 5.0 TB/s, the curl / vorticity / divergence pass 20 B/texel at 5.0 TB/s).  Nothing breaks when the guess is off: a
// longer exchange shows, a shorter one has paid for a frame launch it did not need (profiles/r04/overlap_vs_link_latency.txt has both).
// fluid_set_link_model() replaces the two constants for a context; FLUID_LINK_MODEL="us,GB/s" (lab build) for the process.
struct LinkModel {
    double lat_us, gbps;
};

LinkModel link_model(const fluid_ctx* c)
{
    static const LinkModel forced = [] {
        LinkModel v{ -1.0, -1.0 };
        if (const char* e = fluid::lab_env("FLUID_LINK_MODEL")) {
            double a = 0, b = 0;
            if (sscanf(e, "%lf,%lf", &a, &b) == 2 && a >= 0 && b > 0) v = LinkModel{ a, b };
        }
        return v;
    }();
    return forced.gbps > 0 ? forced : LinkModel{ (double)c->link_lat_us, (double)c->link_gbps };
}

double exchange_us(fluid_ctx* c, const fluid_stripe_op& op)
{
    const bool tiles = c->desc.parts_x > 1;
    double rows_msg = 0, cols_msg = 0;   // bytes to a DOWN / UP neighbour, to a LEFT / RIGHT one
    for (int i = 0; i < op.n_items; i++) {
        FieldRef f;
        if (field_ref(c, op.field[i], &f, true) != FLUID_OK) return 0;
        const double texel = (double)f.texel();
        rows_msg += (double)op.rows[i] * (tiles ? f.cols : f.win->P) * texel;
        if (tiles) cols_msg += (double)f.rows * col_depth(c, f, op.rows[i]) * texel;
    }
    const LinkModel m = link_model(c);
    return m.lat_us + std::max(rows_msg, cols_msg) / (m.gbps * 1e3) + (tiles ? 30.0 : 0.0);
}

double jacobi_launch_us(const fluid_ctx* c) { return (double)c->sim_ncols * c->sim_rows * 13.2 / 5.0e6; }
double cvd_interior_us(const fluid_ctx* c) { return (double)c->sim_ncols * c->sim_rows * 20.0 / 5.0e6; }

int launches_for(double us, const fluid_ctx* c)   // Jacobi launches that take at least `us`, 0 .. 2
{
    if (us <= 0) return 0;
    const int n = (int)std::ceil(us / jacobi_launch_us(c));
    return n > 2 ? 2 : n;
}

// the pressure rows / columns an exchange is sending (0 if it carries no pressure): what a second cut launch must stay clear of
void pressure_in_flight(fluid_ctx* c, const fluid_stripe_op& op, JacobiSplit* sp)
{
    for (int i = 0; i < op.n_items; i++)
        if (op.field[i] == FLUID_PRESSURE) {
            FieldRef f;
            sp->guard_rows = op.rows[i];
            if (c->desc.parts_x > 1 && field_ref(c, FLUID_PRESSURE, &f, true) == FLUID_OK) sp->guard_cols = col_depth(c, f, op.rows[i]);
        }
}

// how to cut the Jacobi block behind a pressure-only exchange (cover >= 1: jacobi_overlap_ok said the block can be cut)
// `for_group`: the answer of ONE context of an in-process group, not rounded up — the group cuts as many launches as its least able
// context can (fluid_group_step_n), and none when that is 0
JacobiSplit jacobi_overlap_split(fluid_ctx* c, const std::vector<fluid_stripe_op>& ops, size_t i, bool for_group = false)
{
    JacobiSplit sp;
    pressure_in_flight(c, ops[i], &sp);
    sp.cover = std::max(1, launches_for(exchange_us(c, ops[i]), c));
    sp.cover = jacobi_split_launches(c, ops[i + 1].iters, folds_gradsub(ops, i + 1), sp);
    if (!for_group) sp.cover = std::max(1, sp.cover);
    return sp;
}

int jacobi_block_interior(fluid_ctx* c, const fluid_stripe_op& blk, JacobiSplit sp)
{
    sp.mode = 1;
    return pass_jacobi(c, blk.iters, blk.ext, 1.0f, nullptr, nullptr, nullptr, &sp);
}

// The step's FIRST exchange (velocity + pressure) has the interior of the curl / vorticity / divergence pass as its cover — 65 us at
// 4096^2 per rank, against pack + link + unpack of a tile exchange (profiles/r04/overlap_vs_link_latency.txt).  The first launch(es) of
// the pressure block behind it are cut the same way and join the cover: their texels 3 (4 columns) + one apron inside the owned rectangle
// read only divergence the interior launch already wrote and pressure nobody is exchanging into.  Then: the exchange lands, the strips of
// the curl / vorticity / divergence pass, the frames of those Jacobi launches, every further launch.
// cover = the launches to cut (0: the block is not cut).  FLUID_COVER_JACOBI=0 (lab build): off (A/B knob; same bits either way)
JacobiSplit jacobi_cover_split(fluid_ctx* c, const std::vector<fluid_stripe_op>& ops, size_t i)
{
    static const bool on = [] {
        const char* e = fluid::lab_env("FLUID_COVER_JACOBI");
        return !(e && atoi(e) == 0);
    }();
    JacobiSplit sp;
    sp.margin = 3;
    sp.cover = 0;
    if (!on || i + 2 >= ops.size() || ops[i + 1].kind != FLUID_OP_CURL_VORT_DIV || ops[i + 2].kind != FLUID_OP_CLEAR_JACOBI) return sp;
    pressure_in_flight(c, ops[i], &sp);
    sp.cover = launches_for(exchange_us(c, ops[i]) - cvd_interior_us(c), c);
    sp.cover = jacobi_split_launches(c, ops[i + 2].iters, folds_gradsub(ops, i + 2), sp);
    return sp;
}

int clear_jacobi_interior(fluid_ctx* c, const fluid_stripe_op& blk, const fluid_params* P, JacobiSplit sp)
{
    sp.mode = 1;
    return pass_jacobi(c, blk.iters, blk.ext, P->pressure, nullptr, nullptr, nullptr, &sp);
}

// the rest of a cut block (pscale: config.PRESSURE for the step's first block, whose first launch carries the clear; 1 otherwise);
// `gs` = the gradient-subtract op behind it, or null
int jacobi_block_rest(fluid_ctx* c, const fluid_stripe_op& blk, const fluid_stripe_op* gs, float pscale, JacobiSplit sp)
{
    sp.mode = 2;
    if (!gs) return pass_jacobi(c, blk.iters, blk.ext, pscale, nullptr, nullptr, nullptr, &sp);
    bool folded = true;
    CK(pass_jacobi(c, blk.iters, blk.ext, pscale, nullptr, &folded, nullptr, &sp));
    return folded ? (int)FLUID_OK : pass_gradsub(c, gs->ext);
}

// ---- the thin launches of an overlapped exchange on the COMM stream (round 5) ----
// Behind an exchange the context stream used to run, one after the other: the strips of the pass whose interior covered the exchange (11.7 us
// for the curl / vorticity / divergence pass of a 4096^2 stripe, 6.5 us for the advection) and the frame of the cut Jacobi launch (10.9 us) —
// launches of one tile's latency each, +5 % of the step (profiles/r04/stripe_rank_timeline.txt).  They depend on the ghost texels, not on the
// interior that is still running: they now go on the comm stream, directly behind RCCL's receive, and run BESIDE the interior launches wherever
// the link is faster than those (it is: 75 + 47 us of cover in front of the pressure loop, 147 us of advection).  The context stream then waits
// once, for ev_joined, where it waited for ev_landed.  A frame reads pressure / divergence the interiors wrote: the comm stream waits for
// ev_inner (device scope: same GPU) in front of it.  The strips' and frames' texels are disjoint from the interiors' (that is what makes them
// strips and frames), so nothing but those two events orders the streams.  FLUID_STRIPS_ON_COMM=0 (lab build): as before (A/B knob; same bits).
bool strips_on_comm(const fluid_ctx* c)
{
    static const bool on = [] {
        const char* e = fluid::lab_env("FLUID_STRIPS_ON_COMM");
        return !(e && atoi(e) == 0);
    }();
    return on && c->comm != nullptr && c->comm_stream != nullptr;
}

struct OnStream {   // the launchers enqueue on fluid_ctx::stream: point it somewhere else for a scope
    fluid_ctx* c;
    hipStream_t saved;
    OnStream(fluid_ctx* ctx, hipStream_t s) : c(ctx), saved(ctx->stream) { c->stream = s; }
    ~OnStream() { c->stream = saved; }
};

// the comm stream has the ghost texels (its own receive is done: the self-wait carries the acquire the context stream's wait used to make)
int comm_has_landed(fluid_ctx* c)
{
    HIPCK(c, hipStreamWaitEvent(c->comm_stream, c->ev_landed, 0));
    return FLUID_OK;
}

int join_comm(fluid_ctx* c)   // the context stream continues behind what the comm stream has been given so far
{
    HIPCK(c, hipEventRecord(c->ev_joined, c->comm_stream));
    HIPCK(c, hipStreamWaitEvent(c->stream, c->ev_joined, 0));
    return FLUID_OK;
}

// ---- the dye's wire format, agreed per call ----
// The ghost texels of the dye travel in the format the field is in: 12-byte texels while it is packed, RGBA otherwise — and the two ends of a
// message must cut the same size (a count mismatch between ncclSend and ncclRecv is undefined: a hang or torn ghost rows).  Whether a rank packs
// depends on state ONE rank can change alone between two calls — fluid_field_device_ptr(FLUID_DYE), fluid_write_field, fluid_halo_unpack, a
// splat into one context only, or simply a stripe below the texel count from which packing pays.  So every fluid_step_n on a communicator
// begins with ONE ncclAllReduce(min) of four floats on the comm stream — {this rank would pack, its alpha is known, alpha, -alpha} — which the
// host reads when the call reaches its first dye exchange (the step's first ~0.3 ms of launches are enqueued by then: nothing waits on the
// device).  The set packs iff every rank would and all hold the same alpha; ranks that disagree about the alpha itself (one was splatted
// alone) forget theirs, as a ghost-row unpack does: the texels they receive carry another.  Inside a call the format follows from the call's
// arguments alone (dt, the decays), which are the same on every rank.
int dye_format_agree_begin(fluid_ctx* c, float dt, const fluid_params* P)
{
    const Rccl* R = rccl(nullptr);
    if (!R || !c->comm) return c->fail(FLUID_ERR_COMM, "stripe context has no communicator (fluid_comm_init)");
    if (!c->agree_host) {
        float* h = nullptr;
        float* d = nullptr;
        hipEvent_t ev = nullptr;
        int rc = c->hip(hipHostMalloc((void**)&h, 8 * sizeof(float), hipHostMallocDefault), "hipHostMalloc (dye format words)");
        if (!rc) rc = c->hip(hipMalloc((void**)&d, 8 * sizeof(float)), "hipMalloc (dye format words)");
        if (!rc) rc = c->hip(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate (dye format)");
        if (rc) {
            if (h) (void)hipHostFree(h);
            if (d) (void)hipFree(d);
            return rc;
        }
        c->agree_host = h;
        c->agree_dev = d;
        c->ev_agreed = ev;
    }
    c->dye_set_rgba = false;   // this rank's own answer, asked without the set's veto
    const bool packs = dye_wants_packed(c, dt, P->velocity_dissipation, P->density_dissipation);
    float* h = c->agree_host;
    h[0] = packs ? 1.0f : 0.0f;
    h[1] = c->alpha_known ? 1.0f : 0.0f;
    h[2] = c->alpha_known ? c->dye_alpha : 0.0f;
    h[3] = c->alpha_known ? -c->dye_alpha : 0.0f;
    HIPCK(c, hipMemcpyAsync(c->agree_dev, h, 4 * sizeof(float), hipMemcpyHostToDevice, c->comm_stream));
    NCCLCK(c, R, R->AllReduce(c->agree_dev, c->agree_dev + 4, 4, ncclFloat, ncclMin, (ncclComm_t)c->comm, c->comm_stream));
    HIPCK(c, hipMemcpyAsync(h + 4, c->agree_dev + 4, 4 * sizeof(float), hipMemcpyDeviceToHost, c->comm_stream));
    HIPCK(c, hipEventRecord(c->ev_agreed, c->comm_stream));
    c->agree_pending = true;
    return FLUID_OK;
}

int dye_format_agree_end(fluid_ctx* c)
{
    if (!c->agree_pending) return FLUID_OK;
    c->agree_pending = false;
    HIPCK(c, hipEventSynchronize(c->ev_agreed));
    const float* m = c->agree_host + 4;
    const bool all_pack = m[0] == 1.0f, all_known = m[1] == 1.0f, one_alpha = all_known && m[2] == -m[3];
    if (!one_alpha) c->alpha_known = false;   // a neighbour's texels carry another alpha (or nobody knows theirs): no ONE alpha here either until the next splat
    c->dye_set_rgba = !(all_pack && one_alpha);
    return FLUID_OK;
}

// in front of an exchange that carries dye ghost texels: the field goes into the format this step's advection takes (packed or RGBA — what
// the set agreed on at the start of the call), so that both ends of every message cut the same texel size
int prepare_exchange(fluid_ctx* c, const fluid_stripe_op& op, float dt, const fluid_params* P)
{
    for (int i = 0; i < op.n_items; i++)
        if (op.field[i] == FLUID_DYE) {
            CK(dye_format_agree_end(c));
            return dye_prepare(c, dt, P);
        }
    return FLUID_OK;
}

int plan_for(fluid_ctx* c, const fluid_params* P, std::vector<fluid_stripe_op>& ops)
{
    int va, vd;
    advect_rows(c, &va, &vd);
    if (build_plan(c->desc.halo, dye_ghost_depth(c), P->iterations, va, vd, ops) != FLUID_OK)
        return c->fail(FLUID_ERR_INVALID, "stripe plan: bad halo / iterations / reach");
    return FLUID_OK;
}

}  // namespace

namespace fluid_impl {

int stripe_step_n(fluid_ctx* c, int n, float dt, const fluid_params* P)
{
    if (!c->comm) return c->fail(FLUID_ERR_COMM, "a stripe context steps with its communicator: call fluid_comm_init first (or drive the "
                                                  "passes and exchanges yourself through fluid_pass_* / fluid_field_device_ptr)");
    CK(ensure_comm_stream(c));
    std::vector<fluid_stripe_op> ops;
    CK(plan_for(c, P, ops));
    struct CurlGuard {
        fluid_ctx* c;
        ~CurlGuard() { c->keep_curl = true; }
    } guard{ c };
    const bool skip = skip_hidden_curl();
    if (n > 0) CK(dye_format_agree_begin(c, dt, P));   // collective, first thing on the comm stream: every rank of the set is in this call
    mark_step(c, 0);
    for (int k = 0; k < n; mark_step(c, ++k))
        for (size_t i = 0; i < ops.size(); i++) {
            c->keep_curl = c->curl_output && (k == n - 1 || !skip);   // only the call's last step leaves a curl field a caller can read (fluid_step_n) — none with fluid_set_curl_output(ctx, 0)
            const fluid_stripe_op& op = ops[i];
            if (op.kind != FLUID_OP_EXCHANGE) {
                if (folds_gradsub(ops, i)) {
                    CK(run_block_and_gradsub(c, op, ops[i + 1], P));
                    i++;
                    continue;
                }
                CK(pass_whole(c, op, dt, P));
                continue;
            }
            CK(prepare_exchange(c, op, dt, P));
            if (c->desc.parts_x > 1) CK(rccl_exchange_2d_begin(c, op));
            else CK(rccl_exchange_begin(c, op));
            const bool side = strips_on_comm(c);   // strips and frames beside the interiors, on the comm stream
            if (i + 1 < ops.size() && overlap_ok(c, ops[i + 1])) {
                JacobiSplit deep = jacobi_cover_split(c, ops, i);
                CK(pass_interior(c, ops[i + 1], dt, P));  // computes while the ghost rows travel
                // ONE cut Jacobi launch: its frame reads the DIVERGENCE of the pass's interior (its pressure input is the step's old
                // pressure) and nothing the Jacobi interior writes — it runs on the comm stream beside that interior.  With TWO cut
                // launches the second frame reads the first interior's output: the frames would wait for both interiors anyway, and a
                // hop to the comm stream and back costs more than it hides (the 200-iteration regime: +2 %, profiles/r05) — they stay
                // on the context stream, behind the strips.
                const bool frames_on_comm = side && deep.cover == 1;
                if (frames_on_comm) HIPCK(c, hipEventRecord(c->ev_inner, c->stream));
                if (deep.cover) CK(clear_jacobi_interior(c, ops[i + 2], P, deep));
                if (side) {
                    CK(comm_has_landed(c));
                    {
                        OnStream on(c, c->comm_stream);
                        CK(pass_strips(c, ops[i + 1], dt, P));
                    }
                    if (frames_on_comm) {
                        HIPCK(c, hipStreamWaitEvent(c->comm_stream, c->ev_inner, 0));
                        deep.frame_stream = c->comm_stream;
                        deep.frame_done = c->ev_joined;   // pass_jacobi joins the streams behind the frame
                    } else {
                        CK(join_comm(c));
                    }
                } else {
                    CK(rccl_exchange_end(c));
                    CK(pass_strips(c, ops[i + 1], dt, P));
                }
                i++;
                if (deep.cover) {
                    const bool gs = folds_gradsub(ops, i + 1);
                    CK(jacobi_block_rest(c, ops[i + 1], gs ? &ops[i + 2] : nullptr, P->pressure, deep));
                    i += gs ? 2 : 1;
                }
            } else if (jacobi_overlap_ok(c, ops, i)) {
                const bool gs = folds_gradsub(ops, i + 1);
                JacobiSplit sp = jacobi_overlap_split(c, ops, i);
                CK(jacobi_block_interior(c, ops[i + 1], sp));
                if (side && sp.cover == 1) {   // one cut launch: its frame reads nothing the interior writes, and runs beside it (two: see above)
                    CK(comm_has_landed(c));
                    sp.frame_stream = c->comm_stream;
                    sp.frame_done = c->ev_joined;
                } else {
                    CK(rccl_exchange_end(c));
                }
                CK(jacobi_block_rest(c, ops[i + 1], gs ? &ops[i + 2] : nullptr, 1.0f, sp));
                i += gs ? 2 : 1;
            } else {
                CK(rccl_exchange_end(c));
            }
        }
    // A back-trace longer than the refreshed ghost rows was counted by the advection kernels (Win::v0/v1): report it from the
    // call that produced it — a caller that never asks fluid_halo_check must not get FLUID_OK with stale rows in its fields.
    return n > 0 ? fluid_halo_check(c) : FLUID_OK;
}

void stripes_release(fluid_ctx* c)
{
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);  // nothing of this context is still in flight on the communicator
    if (c->comm) {
        const Rccl* R = rccl(nullptr);
        if (R) (void)R->CommDestroy((ncclComm_t)c->comm);
        c->comm = nullptr;
    }
    if (c->comm_stream) {
        (void)hipStreamDestroy(c->comm_stream);
        c->comm_stream = nullptr;
    }
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->ev_landed) (void)hipEventDestroy(c->ev_landed);
    if (c->ev_mid) (void)hipEventDestroy(c->ev_mid);
    if (c->ev_joined) (void)hipEventDestroy(c->ev_joined);
    if (c->ev_inner) (void)hipEventDestroy(c->ev_inner);
    if (c->ev_agreed) (void)hipEventDestroy(c->ev_agreed);
    if (c->agree_host) (void)hipHostFree(c->agree_host);
    if (c->agree_dev) (void)hipFree(c->agree_dev);
    c->agree_host = c->agree_dev = nullptr;
    c->agree_pending = false;
    c->ev_ready = c->ev_landed = c->ev_mid = c->ev_joined = c->ev_inner = c->ev_agreed = nullptr;
    for (int k = 0; k < 16; k++) {
        if (c->stage[k]) (void)hipFree(c->stage[k]);
        c->stage[k] = nullptr;
        c->stage_bytes[k] = 0;
    }
}

}  // namespace fluid_impl

// ================================================================================================================
extern "C" {

int fluid_stripe_plan(int halo, int dye_halo, int iterations, int advect_rows, int advect_dye_rows, fluid_stripe_op* ops, int max_ops,
                      int* n_ops)
{
    if (!n_ops) return FLUID_ERR_INVALID;
    std::vector<fluid_stripe_op> v;
    const int rc = build_plan(halo, dye_halo, iterations, advect_rows, advect_dye_rows, v);
    if (rc != FLUID_OK) return rc;
    *n_ops = (int)v.size();
    if (ops) {
        if (max_ops < (int)v.size()) return FLUID_ERR_INVALID;
        std::memcpy(ops, v.data(), v.size() * sizeof(fluid_stripe_op));
    }
    return FLUID_OK;
}

int fluid_set_reach(fluid_ctx* c, int rows)
{
    if (!c) return FLUID_ERR_INVALID;
    if (rows < 1) return c->fail(FLUID_ERR_INVALID, "reach must be >= 1 row");
    c->reach = rows;
    return FLUID_OK;
}

int fluid_set_link_model(fluid_ctx* c, float latency_us, float gbytes_per_s)
{
    if (!c) return FLUID_ERR_INVALID;
    if (!(latency_us >= 0.0f) || !(gbytes_per_s > 0.0f)) return c->fail(FLUID_ERR_INVALID, "link model: latency >= 0 us, bandwidth > 0 GB/s");
    c->link_lat_us = latency_us;
    c->link_gbps = gbytes_per_s;
    return FLUID_OK;
}

int fluid_set_overlap(fluid_ctx* c, int enabled)
{
    if (!c) return FLUID_ERR_INVALID;
    CK(ensure_comm_stream(c));
    c->overlap = enabled != 0;
    return FLUID_OK;
}

int fluid_advect_exchange_rows(const fluid_ctx* c, int* velocity_rows, int* dye_rows)
{
    if (!c || !velocity_rows || !dye_rows) return FLUID_ERR_INVALID;
    advect_rows(c, velocity_rows, dye_rows);
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
    NCCLCK(c, R, R->CommInitRank(&comm, c->desc.parts * c->desc.parts_x, u, c->desc.part * c->desc.parts_x + c->desc.part_x));
    c->comm = comm;
    return FLUID_OK;
}

int fluid_comm_selftest(fluid_ctx* c, int nfloats)
{
    // loop a buffer through ncclSend / ncclRecv to THIS rank inside one group, on the comm stream, fenced by the
    // same two events as a real exchange, between two kernels of the context stream
    if (!c || nfloats < 1) return FLUID_ERR_INVALID;
    const Rccl* R = rccl(nullptr);
    if (!R || !c->comm) return c->fail(FLUID_ERR_COMM, "no communicator");
    CK(ensure_comm_stream(c));
    HIPCK(c, hipSetDevice(c->device));
    float *a = nullptr, *b = nullptr;
    HIPCK(c, hipMalloc((void**)&a, nfloats * sizeof(float)));
    HIPCK(c, hipMalloc((void**)&b, nfloats * sizeof(float)));
    int rc = FLUID_OK;
    do {
        if ((rc = c->hip(launch_fill(c->stream, a, (size_t)nfloats, 1, 3.25f, 0, 0, 0), "fill"))) break;
        if ((rc = c->hip(launch_fill(c->stream, b, (size_t)nfloats, 1, -1.0f, 0, 0, 0), "fill"))) break;
        if ((rc = c->hip(hipEventRecord(c->ev_ready, c->stream), "record"))) break;
        if ((rc = c->hip(hipStreamWaitEvent(c->comm_stream, c->ev_ready, 0), "wait"))) break;
        ncclComm_t comm = (ncclComm_t)c->comm;
        const int me = c->desc.part * c->desc.parts_x + c->desc.part_x;
        ncclResult_t e;
        if ((e = R->GroupStart()) != ncclSuccess || (e = R->Send(a, nfloats, ncclFloat, me, comm, c->comm_stream)) != ncclSuccess ||
            (e = R->Recv(b, nfloats, ncclFloat, me, comm, c->comm_stream)) != ncclSuccess || (e = R->GroupEnd()) != ncclSuccess) {
            rc = nccl_fail(c, R, e, "self send/recv");
            break;
        }
        if ((rc = c->hip(hipEventRecord(c->ev_landed, c->comm_stream), "record"))) break;
        if ((rc = c->hip(hipStreamWaitEvent(c->stream, c->ev_landed, 0), "wait"))) break;
        std::vector<float> host(nfloats);
        if ((rc = c->hip(hipMemcpyAsync(host.data(), b, nfloats * sizeof(float), hipMemcpyDeviceToHost, c->stream), "copy"))) break;
        if ((rc = ctx_sync(c))) break;
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

int fluid_comm_calibrate_link(fluid_ctx* c, int reps, float* latency_us, float* gbytes_per_s)
{
    if (!c) return FLUID_ERR_INVALID;
    const Rccl* R = rccl(nullptr);
    if (!R || !c->comm) return c->fail(FLUID_ERR_COMM, "no communicator (fluid_comm_init)");
    CK(ensure_comm_stream(c));
    HIPCK(c, hipSetDevice(c->device));
    if (reps <= 0) reps = 20;
    if (reps > 200) reps = 200;
    const int warm = 3;
    // the neighbours a step exchanges whole rows / columns with (the corner blocks of a tile exchange are a few KB: not what the time is)
    const int py = c->desc.part, px = c->desc.part_x, ny = c->desc.parts, nx = c->desc.parts_x;
    std::vector<int> peers;
    if (py > 0) peers.push_back((py - 1) * nx + px);
    if (py < ny - 1) peers.push_back((py + 1) * nx + px);
    if (px > 0) peers.push_back(py * nx + px - 1);
    if (px < nx - 1) peers.push_back(py * nx + px + 1);
    auto report = [&] {
        if (latency_us) *latency_us = c->link_lat_us;
        if (gbytes_per_s) *gbytes_per_s = c->link_gbps;
        return (int)FLUID_OK;
    };
    if (peers.empty()) return report();
    // the largest message of a step: the first exchange's velocity + pressure ghost rows to one neighbour (every rank of a set computes the
    // same figure: the message sizes of the two sides of a pair must agree)
    const size_t small = 4096;
    // (from the descriptor only — the global width, the decomposition, the ghost depth: a tile's own pitch differs between border and inner tiles)
    size_t large = (size_t)c->desc.halo * ((size_t)c->desc.sim_w / (size_t)c->desc.parts_x) * 12u * (c->storage == FLUID_STORE_F16 ? 1u : 2u) / 2u;
    large = std::min<size_t>(std::max<size_t>(large & ~(size_t)255, (size_t)1 << 20), (size_t)64 << 20);
    char* buf = nullptr;
    HIPCK(c, hipMalloc((void**)&buf, 2 * peers.size() * large));
    std::vector<hipEvent_t> ev((size_t)warm + reps + 1, nullptr);
    int rc = FLUID_OK;
    auto measure = [&](size_t bytes, double* median_us) {
        ncclComm_t comm = (ncclComm_t)c->comm;
        for (size_t k = 0; k < ev.size() && !rc; k++) {
            if (k > 0) {
                ncclResult_t e = R->GroupStart();
                for (size_t p = 0; p < peers.size() && e == ncclSuccess; p++) {
                    e = R->Send(buf + (2 * p) * large, bytes, ncclChar, peers[p], comm, c->comm_stream);
                    if (e == ncclSuccess) e = R->Recv(buf + (2 * p + 1) * large, bytes, ncclChar, peers[p], comm, c->comm_stream);
                }
                if (e == ncclSuccess) e = R->GroupEnd();
                if (e != ncclSuccess) {
                    rc = nccl_fail(c, R, e, "link calibration send/recv");
                    break;
                }
            }
            rc = c->hip(hipEventRecord(ev[k], c->comm_stream), "record");
        }
        if (!rc) rc = c->hip(hipStreamSynchronize(c->comm_stream), "sync");
        if (rc) return;
        std::vector<float> us;
        for (size_t k = (size_t)warm; k + 1 < ev.size(); k++) {
            float ms = 0;
            if ((rc = c->hip(hipEventElapsedTime(&ms, ev[k], ev[k + 1]), "elapsed"))) return;
            us.push_back(ms * 1e3f);
        }
        std::sort(us.begin(), us.end());
        *median_us = us[us.size() / 2];
    };
    double t_small = 0, t_large = 0;
    do {
        if ((rc = c->hip(hipMemsetAsync(buf, 0, 2 * peers.size() * large, c->comm_stream), "memset"))) break;
        for (auto& e : ev)
            if ((rc = c->hip(hipEventCreate(&e), "hipEventCreate"))) break;
        if (rc) break;
        measure(small, &t_small);
        if (rc) break;
        measure(large, &t_large);
    } while (0);
    for (auto& e : ev)
        if (e) (void)hipEventDestroy(e);
    (void)hipFree(buf);
    CK(rc);
    // latency + bytes / bandwidth through the two points; a link on which the large message costs no more than the small one (an
    // instantaneous stand-in) gets a bandwidth that makes the byte term vanish
    const double lat = std::max(t_small, 0.0);
    const double extra = t_large - t_small;
    const double gbps = extra > 0.02 * std::max(t_small, 1.0) ? std::min((double)(large - small) / (extra * 1e3), 1e4) : 1e4;
    c->link_lat_us = (float)lat;
    c->link_gbps = (float)gbps;
    return report();
}

// the cut of a pressure-only exchange's Jacobi block for an in-process group: every context is asked (its own geometry, storage and
// column alignment — jacobi_overlap_ok / jacobi_split_launches), the group cuts what ALL of them can, and nothing (cover 0) when one cannot
static JacobiSplit group_overlap_split(fluid_ctx** cs, int n_ctx, const std::vector<fluid_stripe_op>& ops, size_t i)
{
    JacobiSplit none;
    none.cover = 0;
    for (int r = 0; r < n_ctx; r++)
        if (!jacobi_overlap_ok(cs[r], ops, i)) return none;
    JacobiSplit sp = jacobi_overlap_split(cs[0], ops, i, true);
    for (int r = 1; r < n_ctx && sp.cover > 0; r++) {
        const JacobiSplit o = jacobi_overlap_split(cs[r], ops, i, true);
        sp.cover = std::min(sp.cover, o.cover);
        sp.guard_rows = std::max(sp.guard_rows, o.guard_rows);
        sp.guard_cols = std::max(sp.guard_cols, o.guard_cols);
    }
    return sp;
}

int fluid_group_step_n(fluid_ctx** cs, int n_ctx, int steps, float dt, const fluid_params* P)
{
    if (!cs || n_ctx < 1 || !P || steps < 0) return FLUID_ERR_INVALID;
    for (int r = 0; r < n_ctx; r++) {
        if (!cs[r]) return FLUID_ERR_INVALID;
        const fluid_desc& d = cs[r]->desc;
        if (d.parts * d.parts_x != n_ctx || d.part * d.parts_x + d.part_x != r)
            return cs[r]->fail(FLUID_ERR_INVALID, "group must hold every tile, ordered by stripe then tile column");
        if (cs[r]->desc.halo != cs[0]->desc.halo || cs[r]->reach != cs[0]->reach || cs[r]->desc.schedule != cs[0]->desc.schedule ||
            cs[r]->storage != cs[0]->storage)
            return cs[r]->fail(FLUID_ERR_INVALID, "stripes of a group share halo, reach, schedule and storage");
    }
    if (n_ctx == 1) return fluid_step_n(cs[0], steps, dt, P);
    for (int r = 0; r < n_ctx; r++) CK(ensure_comm_stream(cs[r]));
    std::vector<fluid_stripe_op> ops;
    CK(plan_for(cs[0], P, ops));
    auto each = [&](auto&& fn) {
        for (int r = 0; r < n_ctx; r++) {
            const int rc_dev = cs[r]->hip(hipSetDevice(cs[r]->device), "hipSetDevice");
            if (rc_dev != FLUID_OK) return rc_dev;
            const int rc = fn(cs[r]);
            if (rc != FLUID_OK) return rc;
        }
        return (int)FLUID_OK;
    };
    struct CurlGuard {
        fluid_ctx** cs;
        int n;
        ~CurlGuard() { for (int r = 0; r < n; r++) cs[r]->keep_curl = true; }
    } guard{ cs, n_ctx };
    const bool skip = skip_hidden_curl();
    auto mark_all = [&](int k) { for (int r = 0; r < n_ctx; r++) mark_step(cs[r], k); };
    mark_all(0);
    for (int k = 0; k < steps; mark_all(++k))
        for (size_t i = 0; i < ops.size(); i++) {
            for (int r = 0; r < n_ctx; r++) cs[r]->keep_curl = cs[r]->curl_output && (k == steps - 1 || !skip);   // as in stripe_step_n
            const fluid_stripe_op& op = ops[i];
            if (op.kind != FLUID_OP_EXCHANGE) {
                if (folds_gradsub(ops, i)) {
                    const fluid_stripe_op& gs = ops[i + 1];
                    CK(each([&](fluid_ctx* c) { return run_block_and_gradsub(c, op, gs, P); }));
                    i++;
                    continue;
                }
                CK(each([&](fluid_ctx* c) { return pass_whole(c, op, dt, P); }));
                continue;
            }
            const bool tiles = cs[0]->desc.parts_x > 1;
            CK(each([&](fluid_ctx* c) { return prepare_exchange(c, op, dt, P); }));
            for (int r = 1; r < n_ctx; r++)   // what "collective" buys on an RCCL set, checked where one process sees every rank
                if (cs[r]->dye_packed != cs[0]->dye_packed || (cs[0]->dye_packed && std::memcmp(&cs[r]->dye_alpha, &cs[0]->dye_alpha, sizeof(float)) != 0))
                    return cs[r]->fail(FLUID_ERR_INVALID, "the contexts of a set disagree on the dye's format or alpha: splats, dye writes and raw dye pointers "
                                                          "are collective on a stripe / tile set");
            if (tiles) CK(group_exchange_2d_begin(cs, n_ctx, op));
            else CK(group_exchange_begin(cs, n_ctx, op));
            if (i + 1 < ops.size() && overlap_ok(cs[0], ops[i + 1])) {
                const fluid_stripe_op& next = ops[i + 1];
                // the pressure block's first launch(es) join the cover as far as every tile of the group can cut them
                JacobiSplit deep = jacobi_cover_split(cs[0], ops, i);
                for (int r = 1; r < n_ctx; r++) deep.cover = std::min(deep.cover, jacobi_cover_split(cs[r], ops, i).cover);
                CK(each([&](fluid_ctx* c) { return pass_interior(c, next, dt, P); }));
                if (deep.cover) CK(each([&](fluid_ctx* c) { return clear_jacobi_interior(c, ops[i + 2], P, deep); }));
                CK(tiles ? group_exchange_2d_end(cs, n_ctx) : group_exchange_end(cs, n_ctx));
                CK(each([&](fluid_ctx* c) { return pass_strips(c, next, dt, P); }));
                i++;
                if (deep.cover) {
                    const bool gs = folds_gradsub(ops, i + 1);
                    const fluid_stripe_op& blk = ops[i + 1];
                    CK(each([&](fluid_ctx* c) { return jacobi_block_rest(c, blk, gs ? &ops[i + 2] : nullptr, P->pressure, deep); }));
                    i += gs ? 2 : 1;
                }
            } else if (JacobiSplit sp = group_overlap_split(cs, n_ctx, ops, i); sp.cover > 0) {
                const bool gs = folds_gradsub(ops, i + 1);
                const fluid_stripe_op& blk = ops[i + 1];
                CK(each([&](fluid_ctx* c) { return jacobi_block_interior(c, blk, sp); }));
                CK(tiles ? group_exchange_2d_end(cs, n_ctx) : group_exchange_end(cs, n_ctx));
                CK(each([&](fluid_ctx* c) { return jacobi_block_rest(c, blk, gs ? &ops[i + 2] : nullptr, 1.0f, sp); }));
                i += gs ? 2 : 1;
            } else {
                CK(tiles ? group_exchange_2d_end(cs, n_ctx) : group_exchange_end(cs, n_ctx));
            }
        }
    if (steps > 0) {  // as in stripe_step_n: a reach violation fails the call that caused it (every stripe is checked and reset)
        int rc = FLUID_OK;
        for (int r = 0; r < n_ctx; r++) {
            const int rc_r = fluid_halo_check(cs[r]);
            if (rc == FLUID_OK) rc = rc_r;
        }
        return rc;
    }
    return FLUID_OK;
}

long fluid_exchange_count(const fluid_ctx* c) { return c ? c->exchanges : 0; }

}  // extern "C"

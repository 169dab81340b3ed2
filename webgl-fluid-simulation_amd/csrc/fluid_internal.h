// fluid_internal.h — what fluid_solver.cpp (fields, passes, whole-domain step) and fluid_stripes.cpp (the
// multi-GPU stripe plan, RCCL ghost-row exchange) share.  Internal; the public boundary is include/fluid_hip.h.
#pragma once
#include "../../include/fluid_hip.h"
#include "fluid_kernels.h"

#include <string>
#include <vector>

struct fluid_display_state;  // fluid_display.cpp

enum PassId { P_CURL, P_VORT, P_DIV, P_CLEAR, P_JACOBI, P_GRADSUB, P_ADVV, P_ADVD, P_COUNT };

struct fluid_ctx {
    fluid_desc desc{};
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;

    // local windows (sim grid, dye grid); rows include the ghost rows of a stripe
    fluid::Win sim{}, dye{};
    int sim_row0 = 0, sim_rows = 0, dye_row0 = 0, dye_rows = 0, dye_halo = 0;
    int sim_col0 = 0, sim_ncols = 0, dye_col0 = 0, dye_ncols = 0, dye_halo_x = 0;  // 2-D tiles: owned columns

    // field arrays: fp32 texels (float2 / float / float4) or, with desc.storage == FLUID_STORE_F16, half texels
    int storage = FLUID_STORE_F32;
    size_t esz = sizeof(float);            // bytes per channel
    void* vel[2] = { nullptr, nullptr };   // velocity.read / velocity.write   (2 channels)
    void* prs[2] = { nullptr, nullptr };   // pressure.read / pressure.write   (1)
    void* dyeb[2] = { nullptr, nullptr };  // dye.read / dye.write             (4)
    void* div = nullptr;
    void* curl = nullptr;
    unsigned int* miss = nullptr;  // advection taps that fell outside the window
    // false while fluid_step_n runs a step that is not the call's last: that step's curl field is overwritten before the call returns, so
    // the fused curl / vorticity / divergence kernel does not store it (4 of its 24 B/texel).  The per-pass kernels always need the field.
    bool keep_curl = true;
    // fluid_set_curl_output (ABI 10): off = no step of fluid_step / fluid_step_n stores its curl field (the reference's curl texture is read by
    // nothing outside step(): script.js:1234-1243) — one step per call then moves 4 B/texel less.  curl_valid: the curl field holds the last
    // step's curl; false after a step that did not store it: a read of FLUID_CURL fails instead of returning an older step's field
    bool curl_output = true;
    bool curl_valid = true;
    // The NEXT step's curl / vorticity / divergence, computed ahead by the launch that ended the last call (k_advect_cvd MODE 2; whole-domain
    // fp32 contexts where fluid_step_n chains): velocity after vorticity confinement, divergence, curl.  A page calls step() once per frame
    // (script.js:1176-1186): with this a frame is the six launches of a chained step instead of seven.  Valid until anything else touches the
    // fields (splat, write, resize, a per-pass call, a raw device pointer handed out) or the next step comes with another dt / CURL.
    void* pend_vel = nullptr;
    void* pend_div = nullptr;
    void* pend_curl = nullptr;
    bool pend_valid = false;
    bool pend_failed = false;   // the pending buffers could not be allocated (out of device memory): no run-ahead on this context
    float pend_dt = 0.0f, pend_curl_strength = 0.0f;
    void touched() { pend_valid = false; }   // call from every entry point that changes a field or hands out its memory

    // The dye's alpha channel, when the context KNOWS it to be one value everywhere (fluid_kernels.h rgb3): 1 after creation and after
    // every splat, divided by the dye's decay at every advection (the same fp32 division every texel sees); unknown after anything that can
    // write other values (a field write with non-uniform alpha, a raw pointer, a ghost-row unpack) until the next splat.  While it is known,
    // a whole-domain fp32 context at >= kSmallGridTexels keeps the dye PACKED (dye_packed: the two dye buffers then hold 12-byte texels)
    // between the fused advections; every other consumer of the dye goes through ensure_rgba() first.
    bool alpha_known = false;
    float dye_alpha = 1.0f;
    bool dye_packed = false;
    // Packing costs a conversion each way (28 B/texel) and saves 8 B/texel per advection: it pays from ~8 steps between two consumers of
    // the RGBA texels.  A host that renders or reads the dye every frame would lose: when a packed field is unpacked after fewer than 16
    // advections, the next 256 advections stay RGBA (pack_holdoff counts them down) before packing is tried again.
    int packed_advects = 0, pack_holdoff = 0;

    bool timing = false;
    hipEvent_t ev[P_COUNT + 1] = {};
    double acc_ms[P_COUNT] = {};
    double acc_total = 0;
    int acc_steps = 0, acc_jacobi_launches = 0, acc_folded_launches = 0;

    // step marks (fluid_set_step_marks): events between the steps of a call, nobody waits for them until they are read
    std::vector<hipEvent_t> marks;
    int marks_used = 0;   // events recorded by the last call (marked steps + 1)

    // stripe driver (fluid_stripes.cpp): RCCL communicator of the stripe set, exchange bookkeeping
    void* comm = nullptr;                // ncclComm_t, rank == desc.part, nranks == desc.parts
    hipStream_t comm_stream = nullptr;   // ghost rows travel here while the interior rows of the next pass compute
    float link_lat_us = 20.0f, link_gbps = 50.0f;   // one neighbour message: latency + bytes / bandwidth (fluid_set_link_model)
    bool comm_stream_high = false;       // created at the highest stream priority (contexts with an RCCL communicator: ensure_comm_stream)
    // lab (FLUID_JACOBI_CHAINS): a second stream and its events for the pressure loop cut into two row chains (pass_jacobi)
    // the chained Jacobi launch (k_jacobi_tb_chain; fp32 contexts of 3072^2 ... 20 M texels): its (block, tile row) counters on the
    // device, and two words of MAPPED HOST memory the kernel writes when a workgroup gives up waiting for a tile — read for free at every
    // synchronising call (chain_check): a pressure loop that timed out is an error of that call, never a silently wrong field
    unsigned int* chain_flags = nullptr;
    unsigned int* chain_err_host = nullptr;
    unsigned int* chain_err_dev = nullptr;
    fluid::ChainEpoch chain_epoch;
    bool chain_broken = false;           // a chained launch gave up once: this context keeps to plain launches from then on
    hipStream_t chain_stream = nullptr;
    std::vector<hipEvent_t> chain_ev;
    hipEvent_t ev_ready = nullptr;       // context stream -> comm stream: the rows to send exist
    hipEvent_t ev_landed = nullptr;      // comm stream -> context stream: the ghost rows have arrived
    hipEvent_t ev_order = nullptr;       // fluid_stream_wait_context / fluid_context_wait_stream: context stream <-> a caller's stream
    hipEvent_t ev_inner = nullptr;       // context stream -> comm stream, device scope: the interiors a cut Jacobi launch's frame reads are written
    hipEvent_t ev_joined = nullptr;      // comm stream -> context stream: the strips (and frames) that ran on the comm stream are done
    hipEvent_t ev_mid = nullptr;         // 2-D tiles: this tile's ghost columns are in (phase A), ghost rows may follow
    // The dye's WIRE FORMAT is agreed per fluid_step_n call over the communicator (fluid_stripes.cpp dye_format_agree): four floats in mapped
    // pinned host memory in and out of one ncclAllReduce(min) on the comm stream, read when the call reaches its first dye exchange.
    // dye_set_rgba: the set decided this call's dye travels (and advects) as RGBA — one rank at least cannot pack
    float* agree_host = nullptr;         // [0..3] this rank's {packs, alpha known, alpha, -alpha}, [4..7] the set's minima
    float* agree_dev = nullptr;          // 8 floats of device memory: the all-reduce's send and receive buffers
    hipEvent_t ev_agreed = nullptr;
    bool agree_pending = false;
    bool dye_set_rgba = false;
    void* stage[16] = {};                // 2-D tiles: contiguous staging of the strided blocks, send and receive per direction (4 sides + 4 corners)
    size_t stage_bytes[16] = {};
    long exchanges = 0;
    int reach = 24;                      // rows an advection back-trace may span (dt*|v| + 2); see fluid_set_reach
    int overlap = 1;                     // interior-first overlap of exchanges (FLUID_STRIPE_OVERLAP=0 turns it off)

    fluid_display_state* display = nullptr;  // bloom pyramid, sunrays, dithering texture, frame (fluid_display.cpp)

    int fail(int code, const std::string& what)
    {
        err = what;
        return code;
    }
    int hip(hipError_t e, const char* what)
    {
        if (e == hipSuccess) return FLUID_OK;
        err = std::string(what) + ": " + hipGetErrorString(e);
        (void)hipGetLastError();  // the runtime keeps the error until it is read: do not let it fail the next launch check
        return e == hipErrorOutOfMemory ? FLUID_ERR_OOM : FLUID_ERR_HIP;
    }
};

#define CK(expr)                                   \
    do {                                           \
        int _rc = (expr);                          \
        if (_rc != FLUID_OK) return _rc;           \
    } while (0)
#define HIPCK(ctx, expr) CK((ctx)->hip((expr), #expr))

// A launcher call on the context's fields in whichever storage they have: `expr` is compiled once per storage type with
// S = fluid::StoreF32 / fluid::StoreF16, and VEL / PRS / DYE / DIVG / CURL give the typed pointers.
#define STORE_CALL(c, expr)                                                                                    \
    ((c)->storage == FLUID_STORE_F16 ? [&] { using S = fluid::StoreF16; return (expr); }()                     \
                                     : [&] { using S = fluid::StoreF32; return (expr); }())
#define VEL(c, k) ((S::T2*)(c)->vel[k])
#define PRS(c, k) ((S::T1*)(c)->prs[k])
#define DYE(c, k) ((S::T4*)(c)->dyeb[k])
#define DIVG(c) ((S::T1*)(c)->div)
#define CURL(c) ((S::T1*)(c)->curl)
#define CURL_FUSED(c) ((c)->keep_curl ? CURL(c) : (S::T1*)nullptr)   // output of the fused kernel: null = not stored

namespace fluid_impl {

struct FieldRef {
    void* ptr;
    const fluid::Win* win;
    int row0, rows, halo, nc;
    int col0, cols, halo_x;  // owned columns and ghost columns (2-D tiles; 0, W, 0 otherwise)
    size_t esz;              // bytes per channel of the device array (4, or 2 with fp16 storage)
    size_t texel() const { return (size_t)nc * esz; }
};
int field_ref(fluid_ctx* c, int field, FieldRef* f, bool geometry_only = false, bool keep_packed = false);
int chain_check(fluid_ctx* c);   // behind a stream synchronisation: did a chained Jacobi launch give up waiting (FLUID_ERR_HIP)?
// THE synchronisation of every call that hands data or a status back to its caller: waits for `s` (the context's stream by default), then asks
// chain_check — so that a pressure loop that gave up is an error of whichever call synchronises first (fluid_sync, the reads and writes in
// both storages, fluid_halo_check, the display readbacks, the stripe driver's checks), never a silently wrong field or frame
int ctx_sync(fluid_ctx* c, hipStream_t s = nullptr);
int ensure_rgba(fluid_ctx* c);   // the dye buffers hold RGBA texels from here on (unpacks a packed dye field: fluid_ctx::dye_packed)
// Stripe / tile contexts pack their dye too (round 5): the ghost texels then travel as 12-byte texels, in place, and the FORMAT of the field is
// part of the message layout two neighbours must agree on.  It is therefore a function of nothing but what every rank of a set does alike:
// the splats (alpha becomes 1: packing may start), the steps (dye_prepare in front of every dye exchange: one predicate on dt, the decays
// and the packing state), and the calls that write dye texels behind the library's back (fluid_write_field, fluid_halo_unpack, a raw pointer:
// alpha unknown until the next splat) — which the header declares collective on such sets.  A READ converts into the spare buffer and
// changes nothing.  fluid_group_step_n asserts format and alpha equal across an in-process set.
bool dye_wants_packed(const fluid_ctx* c, float dt, float vel_diss, float dye_diss);
int dye_prepare(fluid_ctx* c, float dt, const fluid_params* P);          // the field in the format this step's advection takes
void advect_both_note(fluid_ctx* c, float dt, float dye_diss);           // one advection of the dye happened (band / rects forms: once per step)

// per-pass device time (fluid_set_timing): events on the context stream around each pass group
struct Timer {
    fluid_ctx* c;
    int idx = 0;
    explicit Timer(fluid_ctx* ctx) : c(ctx)
    {
        if (c->timing) (void)hipEventRecord(c->ev[0], c->stream);
    }
    void mark(int pass)  // closes `pass`: time since the previous mark is charged to it
    {
        if (!c->timing) return;
        (void)hipEventRecord(c->ev[1], c->stream);
        (void)hipEventSynchronize(c->ev[1]);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
        c->acc_ms[pass] += ms;
        c->acc_total += ms;
        (void)hipEventRecord(c->ev[0], c->stream);
    }
};

int pass_curl(fluid_ctx* c, int ext);
int pass_vorticity(fluid_ctx* c, float curl, float dt, int ext);
int pass_divergence(fluid_ctx* c, int ext);
int pass_curl_vort_div(fluid_ctx* c, float curl, float dt, int ext, Timer* t);
int pass_clear(fluid_ctx* c, float value, int ext);
struct Timer;
// how the stripe / tile driver cuts the leading launches of a pressure block around an exchange in flight (pass_jacobi)
struct JacobiSplit {
    int mode = 0;         // 1 = the interiors only (while the exchange travels), 2 = the rest of the block (after it landed)
    int margin = 0;       // texels the interiors stay inside the owned rectangle beyond their apron (0, or 3 behind the curl / vorticity / divergence interior)
    int cover = 1;        // leading launches cut (1 or 2)
    int guard_rows = 0, guard_cols = 0;   // pressure rows / columns next to the tile border that the exchange in flight is SENDING: the second
                                          // cut launch writes into the buffer they are read from and stays clear of them
    // mode 2: the frames of the cut launches go on this stream (the comm stream, behind the exchange that just landed) instead of the context
    // stream; `frame_done` is recorded there behind the last of them and the context stream waits for it before the block's further launches
    hipStream_t frame_stream = nullptr;
    hipEvent_t frame_done = nullptr;
};
int pass_jacobi(fluid_ctx* c, int iters, int ext_out, float pscale, int* launches, bool* gradsub, Timer* t, const JacobiSplit* split = nullptr);
bool jacobi_split_ok(const fluid_ctx* c, int iters, bool wants_gradsub, int margin = 0);
int jacobi_split_launches(const fluid_ctx* c, int iters, bool wants_gradsub, const JacobiSplit& sp);   // leading launches that can be cut (<= sp.cover)
bool gradsub_fold_enabled(long owned_texels);
int pass_clear_jacobi(fluid_ctx* c, float value, int iters, int ext_out, int* launches, bool* gradsub);
int pass_gradsub(fluid_ctx* c, int ext);
int pass_advect_velocity(fluid_ctx* c, float dt, float dissipation, int ext);
int pass_advect_dye(fluid_ctx* c, float dt, float dissipation);
int pass_advect(fluid_ctx* c, float dt, float vel_diss, float dye_diss, Timer* t);

// band forms (no ping-pong swap) of the single-kernel passes, for interior-first overlap in the stripe driver
bool jacobi_tb_applies(const fluid_ctx* c);
bool fused_cvd_applies(const fluid_ctx* c);
bool fused_advect_applies(const fluid_ctx* c);
void sim_band(const fluid_ctx* c, int ext, int& ga, int& gb);  // owned rows +- ext, clipped to domain and window
fluid::Win sim_cols(const fluid_ctx* c, int ext);              // the window with this launch's column range (2-D tiles)
fluid::Win dye_cols(const fluid_ctx* c, int ext);
int cvd_band(fluid_ctx* c, float curl, float dt, int ga, int gb, int xa, int xb);
int cvd_rects(fluid_ctx* c, float curl, float dt, const fluid::BandRects& B);
void cvd_swap(fluid_ctx* c);
int advect_both_band(fluid_ctx* c, float dt, float vel_diss, float dye_diss, int ga, int gb, int xa, int xb, int v0, int v1, int u0, int u1);
int advect_both_rects(fluid_ctx* c, float dt, float vel_diss, float dye_diss, const fluid::BandRects& B, int v0, int v1, int u0, int u1);
void advect_both_swap(fluid_ctx* c);

// fluid_stripes.cpp
void mark_step(fluid_ctx* c, int k);                                     // step marks: k = 0 in front of a call's first step, k behind its k-th
bool skip_hidden_curl();                                                 // FLUID_SKIP_CURL=0: every step stores its curl field (A/B knob)
int stripe_step_n(fluid_ctx* c, int n, float dt, const fluid_params* P);  // this rank's stripe, exchanges over RCCL
void stripes_release(fluid_ctx* c);                                       // frees the communicator (fluid_destroy)
// fluid_display.cpp
void display_release(fluid_ctx* c);

}  // namespace fluid_impl

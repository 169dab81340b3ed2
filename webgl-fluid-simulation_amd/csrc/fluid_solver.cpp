// fluid_solver.cpp — solver core behind the C ABI of include/fluid_hip.h.
//
// Owns the five simulation fields of the reference (`dye, velocity, divergence, curl, pressure`,
// script.js:950-954) as fp32 arrays in HBM, their read/write ping-pong (createDoubleFBO,
// script.js:1079-1106), the pass sequencing of step() (script.js:1231-1294) and splat()
// (script.js:1441-1455), and the row-stripe window used by the multi-GPU driver.
// There is NO CPU path here: without a HIP device fluid_create() fails.
#include "../../include/fluid_hip.h"
#include "fluid_internal.h"
#include "fluid_cut.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace fluid;
using namespace fluid_impl;

namespace {
thread_local std::string g_create_error;
}  // namespace

namespace {

size_t cells(const Win& w) { return (size_t)w.rows * (size_t)w.P; }  // texels allocated: rows x pitch (padding columns included)

void free_fields(fluid_ctx* c)
{
    for (int k = 0; k < 2; k++) {
        if (c->vel[k]) (void)hipFree(c->vel[k]);
        if (c->prs[k]) (void)hipFree(c->prs[k]);
        if (c->dyeb[k]) (void)hipFree(c->dyeb[k]);
        c->vel[k] = nullptr;
        c->prs[k] = nullptr;
        c->dyeb[k] = nullptr;
    }
    if (c->div) (void)hipFree(c->div);
    if (c->curl) (void)hipFree(c->curl);
    c->div = c->curl = nullptr;
    for (void** p : { &c->pend_vel, &c->pend_div, &c->pend_curl }) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    c->pend_valid = false;
}

// window geometry of this context's stripe for a (sim_w, sim_h, dye_w, dye_h) grid
int set_geometry(fluid_ctx* c, int sw, int sh, int dw, int dh)
{
    const fluid_desc& d = c->desc;
    if (sw < 1 || sh < 1 || dw < 1 || dh < 1) return c->fail(FLUID_ERR_INVALID, "field sizes must be >= 1");
    if (d.parts < 1 || d.part < 0 || d.part >= d.parts) return c->fail(FLUID_ERR_INVALID, "bad stripe index");
    if (d.parts_x < 1 || d.part_x < 0 || d.part_x >= d.parts_x) return c->fail(FLUID_ERR_INVALID, "bad tile column index");
    if (d.halo < 0) return c->fail(FLUID_ERR_INVALID, "negative halo");
    if (d.parts_x > 1) {
        if (sw % d.parts_x || dw % d.parts_x) return c->fail(FLUID_ERR_INVALID, "sim_w and dye_w must divide by parts_x");
        if ((sw / d.parts_x) % 4 || (dw / d.parts_x) % 4) return c->fail(FLUID_ERR_INVALID, "tile widths must be multiples of 4");
        if (d.halo < 4) return c->fail(FLUID_ERR_INVALID, "a tile needs halo >= 4");
        if (d.halo > sw / d.parts_x) return c->fail(FLUID_ERR_INVALID, "halo wider than a tile");
    }
    if (d.parts > 1) {
        if (sh % d.parts || dh % d.parts) return c->fail(FLUID_ERR_INVALID, "sim_h and dye_h must divide by parts");
        if (d.halo < 4) return c->fail(FLUID_ERR_INVALID, "a stripe needs halo >= 4");
        if (d.halo > sh / d.parts) return c->fail(FLUID_ERR_INVALID, "halo deeper than a stripe");
    }
    c->sim_rows = sh / d.parts;
    c->sim_row0 = d.part * c->sim_rows;
    c->dye_rows = dh / d.parts;
    c->dye_row0 = d.part * c->dye_rows;
    c->dye_halo = d.parts > 1 ? (int)(((long)d.halo * dh + sh - 1) / sh) : 0;
    c->sim_ncols = sw / d.parts_x;
    c->sim_col0 = d.part_x * c->sim_ncols;
    c->dye_ncols = dw / d.parts_x;
    c->dye_col0 = d.part_x * c->dye_ncols;
    c->dye_halo_x = d.parts_x > 1 ? (int)(((long)d.halo * dw + sw - 1) / sw) : 0;
    const int sh_halo = d.parts > 1 ? d.halo : 0;
    if (d.parts_x > 1) {
        // a tile holds its owned columns + ghost columns only (clipped to the domain; the origin is float4-aligned because tile
        // widths are multiples of 4 and the origin is rounded down to one): memory and clears shrink with parts_x
        const int sa = std::max((c->sim_col0 - d.halo) & ~3, 0), sb = std::min(c->sim_col0 + c->sim_ncols + d.halo, sw);
        const int da = std::max((c->dye_col0 - c->dye_halo_x) & ~3, 0), db = std::min(c->dye_col0 + c->dye_ncols + c->dye_halo_x, dw);
        c->sim = make_win_cols(sw, sh, c->sim_row0 - sh_halo, c->sim_rows + 2 * sh_halo, sa, sb);
        c->dye = make_win_cols(dw, dh, c->dye_row0 - c->dye_halo, c->dye_rows + 2 * c->dye_halo, da, db);
    } else {
        c->sim = make_win(sw, sh, c->sim_row0 - sh_halo, c->sim_rows + 2 * sh_halo);
        c->dye = make_win(dw, dh, c->dye_row0 - c->dye_halo, c->dye_rows + 2 * c->dye_halo);
    }
    return FLUID_OK;
}

// gl.clear after createFBO (script.js:1059) with clearColor (0,0,0,1) (script.js:136): dye alpha starts at 1
int zero_scalar_fields(fluid_ctx* c)
{
    for (void* p : { c->prs[0], c->prs[1], c->div, c->curl }) HIPCK(c, hipMemsetAsync(p, 0, cells(c->sim) * c->esz, c->stream));
    return FLUID_OK;
}

int alloc_fields(fluid_ctx* c)
{
    const size_t ns = cells(c->sim), nd = cells(c->dye);
    for (int k = 0; k < 2; k++) {
        HIPCK(c, hipMalloc(&c->vel[k], ns * 2 * c->esz));
        HIPCK(c, hipMalloc(&c->prs[k], ns * c->esz));
        HIPCK(c, hipMalloc(&c->dyeb[k], nd * 4 * c->esz));
    }
    HIPCK(c, hipMalloc(&c->div, ns * c->esz));
    HIPCK(c, hipMalloc(&c->curl, ns * c->esz));
    for (int k = 0; k < 2; k++) {
        HIPCK(c, hipMemsetAsync(c->vel[k], 0, ns * 2 * c->esz, c->stream));
        HIPCK(c, STORE_CALL(c, launch_fill(c->stream, (S::T1*)c->dyeb[k], nd, 4, 0.f, 0.f, 0.f, 1.f)));
    }
    return zero_scalar_fields(c);
}

// clip [row0 - ext, row0 + rows + ext) to the domain and to the window
void row_range(const Win& w, int row0, int rows, int ext, int& ga, int& gb)
{
    ga = std::max(std::max(row0 - ext, 0), w.g0);
    gb = std::min(std::min(row0 + rows + ext, w.H), w.g0 + w.rows);
}

int check_ext(fluid_ctx* c, int ext, int need_in)
{
    // every neighbour row a pass reads must exist in the window (rows beyond the domain edge are clamped,
    // and row_range() clips the computed rows to the domain, so a whole-domain context accepts any ext)
    if (ext < 0) return c->fail(FLUID_ERR_INVALID, "negative ext");
    if ((c->desc.parts > 1 || c->desc.parts_x > 1) && ext + need_in > c->desc.halo)  // 1 x N tile sets have ghost columns only: same bound
        return c->fail(FLUID_ERR_INVALID, "ext exceeds the ghost rows available to this pass");
    return FLUID_OK;
}

}  // namespace

namespace fluid_impl {

// The window of a launch that computes the owned columns +- ext (2-D tiles; whole float4 groups, clipped to the
// domain).  The window already holds the array's own column origin and pitch (set_geometry); only the launch's range changes.
Win cols_of(Win w, int parts_x, int col0, int cols, int ext)
{
    if (parts_x > 1) {
        w.x0 = std::max((col0 - ext) & ~3, 0);
        w.x1 = std::min((col0 + cols + ext + 3) & ~3, w.W);
    }
    return w;
}
Win sim_cols(const fluid_ctx* c, int ext) { return cols_of(c->sim, c->desc.parts_x, c->sim_col0, c->sim_ncols, ext); }
Win dye_cols(const fluid_ctx* c, int ext) { return cols_of(c->dye, c->desc.parts_x, c->dye_col0, c->dye_ncols, ext); }

// ---- passes -------------------------------------------------------------------------------------
int pass_curl(fluid_ctx* c, int ext)
{
    CK(check_ext(c, ext, 1));
    int ga, gb;
    row_range(c->sim, c->sim_row0, c->sim_rows, ext, ga, gb);
    c->curl_valid = true;
    return c->hip(STORE_CALL(c, launch_curl(c->stream, sim_cols(c, ext), VEL(c, 0), CURL(c), ga, gb)), "curl");
}

int pass_vorticity(fluid_ctx* c, float curl, float dt, int ext)
{
    CK(check_ext(c, ext, 1));
    int ga, gb;
    row_range(c->sim, c->sim_row0, c->sim_rows, ext, ga, gb);
    CK(c->hip(STORE_CALL(c, launch_vorticity(c->stream, sim_cols(c, ext), VEL(c, 0), CURL(c), VEL(c, 1), curl, dt, ga, gb)), "vorticity"));
    std::swap(c->vel[0], c->vel[1]);
    return FLUID_OK;
}

int pass_divergence(fluid_ctx* c, int ext)
{
    CK(check_ext(c, ext, 1));
    int ga, gb;
    row_range(c->sim, c->sim_row0, c->sim_rows, ext, ga, gb);
    return c->hip(STORE_CALL(c, launch_divergence(c->stream, sim_cols(c, ext), VEL(c, 0), DIVG(c), ga, gb)), "divergence");
}

// K1 + K2 + K3; one kernel when the fused schedule applies, the three passes otherwise (same bits either way)
int pass_curl_vort_div(fluid_ctx* c, float curl, float dt, int ext, Timer* t)
{
    if (fused_cvd_applies(c)) {
        CK(check_ext(c, ext, 3));
        int ga, gb;
        row_range(c->sim, c->sim_row0, c->sim_rows, ext, ga, gb);
        CK(c->hip(STORE_CALL(c, launch_curl_vort_div(c->stream, sim_cols(c, ext), VEL(c, 0), CURL_FUSED(c), VEL(c, 1), DIVG(c), curl, dt, ga, gb)),
                  "curl_vort_div"));
        c->curl_valid = c->keep_curl;
        std::swap(c->vel[0], c->vel[1]);
        if (t) t->mark(P_VORT);
        return FLUID_OK;
    }
    CK(pass_curl(c, ext + 2));
    if (t) t->mark(P_CURL);
    CK(pass_vorticity(c, curl, dt, ext + 1));
    if (t) t->mark(P_VORT);
    CK(pass_divergence(c, ext));
    if (t) t->mark(P_DIV);
    return FLUID_OK;
}

int pass_clear(fluid_ctx* c, float value, int ext)
{
    CK(check_ext(c, ext, 0));
    int ga, gb;
    row_range(c->sim, c->sim_row0, c->sim_rows, ext, ga, gb);
    CK(c->hip(STORE_CALL(c, launch_clear(c->stream, sim_cols(c, ext), PRS(c, 0), PRS(c, 1), value, ga, gb)), "clear"));
    std::swap(c->prs[0], c->prs[1]);
    return FLUID_OK;
}

// `iters` Jacobi iterations; pscale != 1 folds the clear pass into the first load (FUSED only).
// `gradsub` (in: fold K6 into the last launch if that launch has the instantiation; out: whether it was): the last block then writes
// the pressure AND velocity - grad(pressure) for the owned rows / columns (ext 0), and the caller skips pass_gradsub.  The blocks in
// front of it leave the pressure valid one ring further out (ext_out >= 1), which is what the separate pass needs as well.
// `split` (stripe / tile driver, an exchange in flight — fluid_stripes.cpp): 1 = ONLY the texels of the block's first launch whose inputs
// are all owned and `margin` texels inside the owned rectangle (they compute while the ghost rows / columns travel; no ping-pong swap
// yet), 2 = the rest of the block: the first launch's frame around that interior (ONE launch over its up to four rectangles,
// launch_jacobi_tb_rects), then every further launch.  1 then 2 leave exactly what 0 leaves (the same iterations over the same texels,
// cut differently).  `margin` = 0 behind a pressure-only exchange (the divergence is complete); 3 when the divergence itself is only
// there on the interior of the curl / vorticity / divergence pass in front (a divergence texel reads velocity 3 texels away; columns go
// by whole float4 groups: 4).  jacobi_split_ok() says whether a block can be cut this way.
static fluid::BlockCut jacobi_cut(const fluid_ctx* c, const Win& w, int ga, int gb, int shape, const JacobiSplit& sp, int level)
{
    const fluid_desc& d = c->desc;
    int dep, depx;
    fluid::cut_depths(level, jacobi_tb_depth(shape), jacobi_tb_apron_cols(shape), sp.margin, sp.guard_rows, sp.guard_cols, dep, depx);
    return fluid::block_cut(ga, gb, w.x0, w.x1, c->sim_row0, c->sim_row0 + c->sim_rows, c->sim_col0, c->sim_col0 + c->sim_ncols, d.part > 0,
                            d.part < d.parts - 1, d.part_x > 0, d.part_x < d.parts_x - 1, dep, depx);
}

// how many leading launches of a block of `iters` iterations can be cut this way, at most sp.cover (0: none)
int jacobi_split_launches(const fluid_ctx* c, int iters, bool wants_gradsub, const JacobiSplit& sp)
{
    if (!jacobi_tb_applies(c) || iters < 1 || sp.cover < 1) return 0;
    const long owned = (long)c->sim_ncols * c->sim_rows;
    const int shape = jacobi_tb_pick(owned), depth = jacobi_tb_depth(shape), hx = jacobi_tb_apron_cols(shape);
    const bool fold = wants_gradsub && jacobi_tb_has_gradsub(shape) && gradsub_fold_enabled(owned);
    int m = (iters + depth - 1) / depth - (fold ? 1 : 0);   // the launch that carries the gradient subtract is not cut
    if (m > sp.cover) m = sp.cover;
    if (m > 2) m = 2;
    const bool tiles = c->desc.parts_x > 1;
    if (tiles) {   // the frame has left / right parts — the rectangle launch (fp32 fields, the 10-row apron) or nothing
        if (c->storage != FLUID_STORE_F32 || depth > 10) return 0;
        if ((c->sim_col0 & 3) != 0 || (c->sim_ncols & 3) != 0) return 0;   // interior columns as whole float4 groups
    }
    // an interior worth a launch of its own, at every level
    auto worth = [&](int level) {
        int dep, depx;
        fluid::cut_depths(level, depth, hx, sp.margin, sp.guard_rows, sp.guard_cols, dep, depx);
        return c->sim_rows > 4 * dep && (!tiles || c->sim_ncols > 4 * depx);
    };
    while (m > 0 && !worth(m)) m--;
    return m;
}

bool jacobi_split_ok(const fluid_ctx* c, int iters, bool wants_gradsub, int margin)
{
    JacobiSplit sp;
    sp.margin = margin;
    return jacobi_split_launches(c, iters, wants_gradsub, sp) > 0;
}

#ifdef FLUID_PROBES
// Lab (FLUID_JACOBI_CHAINS="rows fraction", e.g. 0.5): the pressure loop of a whole-domain context as TWO row chains on two streams.  Launch j
// is cut at row m_j = M - 10 j: the lower part T_j = rows [0, m_j) reads nothing outside T_(j-1)'s output (one apron further up each
// launch), so the T chain runs on its own; the upper part B_j = rows [m_j, H) follows B_(j-1) and T_(j-1).  What it is for: a launch is its
// bytes over the bandwidth PLUS a fill / drain latency nothing overlaps (profiles/r03/jacobi_tail_probe.txt: ~10 us of 43.5) — with two
// chains out of phase, one chain's fill / drain falls into the other's steady state.  Same iterations over the same texels, same bits.
static double jacobi_chains()
{
    static const double f = [] {
        const char* e = fluid::lab_env("FLUID_JACOBI_CHAINS");
        const double v = e ? atof(e) : 0.0;
        return v > 0.0 && v < 1.0 ? v : 0.0;
    }();
    return f;
}

static int pass_jacobi_chains(fluid_ctx* c, int iters, float pscale, int shape, int* launches)
{
    const int depth = jacobi_tb_depth(shape), L = (iters + depth - 1) / depth;
    if (!c->chain_stream) {
        HIPCK(c, hipStreamCreateWithFlags(&c->chain_stream, hipStreamNonBlocking));
    }
    while ((int)c->chain_ev.size() < L + 2) {
        hipEvent_t e;
        HIPCK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->chain_ev.push_back(e);
    }
    int ga, gb;
    row_range(c->sim, c->sim_row0, c->sim_rows, 0, ga, gb);
    const int M = ga + (int)((gb - ga) * jacobi_chains());
    const Win w = sim_cols(c, 0);
    HIPCK(c, hipEventRecord(c->chain_ev[L], c->stream));           // everything in front of the loop
    HIPCK(c, hipStreamWaitEvent(c->chain_stream, c->chain_ev[L], 0));
    int done = 0, left = L;
    void *src = c->prs[0], *dst = c->prs[1];
    for (int j = 0; j < L; j++) {
        const int k = (iters - done + left - 1) / left;
        left--;
        const int m = M - depth * j;
        const float ps = j == 0 ? pscale : 1.0f;
        CK(c->hip(fluid::launch_jacobi_tb(c->stream, w, (const float*)src, (const float*)c->div, (float*)dst, ps, k, ga, m, shape), "jacobi_tb (lower chain)"));
        HIPCK(c, hipEventRecord(c->chain_ev[j], c->stream));
        if (j > 0) HIPCK(c, hipStreamWaitEvent(c->chain_stream, c->chain_ev[j - 1], 0));
        CK(c->hip(fluid::launch_jacobi_tb(c->chain_stream, w, (const float*)src, (const float*)c->div, (float*)dst, ps, k, m, gb, shape), "jacobi_tb (upper chain)"));
        std::swap(src, dst);
        done += k;
        if (launches) (*launches)++;
    }
    HIPCK(c, hipEventRecord(c->chain_ev[L + 1], c->chain_stream));
    HIPCK(c, hipStreamWaitEvent(c->stream, c->chain_ev[L + 1], 0));
    if (L & 1) std::swap(c->prs[0], c->prs[1]);
    return FLUID_OK;
}
#endif

int pass_jacobi(fluid_ctx* c, int iters, int ext_out, float pscale, int* launches, bool* gradsub, Timer* t, const JacobiSplit* sp)
{
    const int split = sp ? sp->mode : 0;
    if (iters < 0) return c->fail(FLUID_ERR_INVALID, "negative iteration count");
    const bool tb = jacobi_tb_applies(c);
    const long owned = (long)c->sim_ncols * c->sim_rows;
    const int shape = tb ? jacobi_tb_pick(owned) : 0;  // one tile geometry for every launch of this pass
    bool fold = gradsub && *gradsub && tb && iters > 0 && jacobi_tb_has_gradsub(shape) && gradsub_fold_enabled(owned);
    // lab (FLUID_CHAIN_GS=1): K6 as one more block of the chained launch (k_jacobi_tb_chain GSB) — whole-domain fp32 contexts, outside timing mode
    static const bool chain_gs_knob = [] { const char* e = fluid::lab_env("FLUID_CHAIN_GS"); return e && atoi(e) != 0; }();
    const bool chain_gs = chain_gs_knob && gradsub && *gradsub && !fold && !split && ext_out == 0 && c->desc.parts == 1 && c->desc.parts_x == 1 && !c->timing;
    if (gradsub) *gradsub = false;
    if (fold && ext_out < 1) ext_out = 1;
    CK(check_ext(c, ext_out, iters));
    int done = 0;
    if (!tb && pscale != 1.0f) return c->fail(FLUID_ERR_INVALID, "pscale needs the fused schedule");
    // balanced blocks: ceil(iters / max) launches of nearly equal depth (50 with max 8 -> 8,7,7,7,7,7,7)
    const int tb_max = tb ? jacobi_tb_depth(shape) : 1;
    int launches_left = tb ? (iters + tb_max - 1) / tb_max : iters;
#ifdef FLUID_PROBES
    if (tb && !split && !fold && !c->timing && jacobi_chains() > 0.0 && c->storage == FLUID_STORE_F32 && c->desc.parts == 1 && c->desc.parts_x == 1 && ext_out == 0 &&
        launches_left >= 2 && (int)(c->sim_rows * jacobi_chains()) > tb_max * (launches_left + 4) &&
        (int)(c->sim_rows * (1.0 - jacobi_chains())) > 4 * tb_max)
        return pass_jacobi_chains(c, iters, pscale, shape, launches);
#endif
    int cut_left = split && tb ? sp->cover : 0, level = 0;   // leading launches still to cut (split 1 / 2)
    void *pa = c->prs[0], *pb = c->prs[1];               // split 1: the interiors ping-pong here; the context's pair swaps when the frames run
    // Every launch that is left once the cut ones are through (all of them on a whole domain) as ONE launch of chained blocks of ten iterations,
    // where that is the faster schedule: fp32 grids whose pressure set fits the Infinity Cache (fluid::jacobi_chain_applies; k_jacobi_tb_chain).  A stripe's launches recompute
    // fewer ghost rows each: every block gets its own row range.  Counted as its blocks — each moves the field once, as a launch does.
    const bool chain_kind = tb && !fold && split != 1 && shape == 0 && c->storage == FLUID_STORE_F32 && !c->chain_broken;
    while (done < iters) {
        int ga, gb;
        if (chain_kind && cut_left == 0 && launches_left >= 2 && launches_left <= std::min(32, fluid::jacobi_chain_max_blocks())) {
            int it[33], ra[33], rb[33], xa[33], xb[33], d = done, left = launches_left;
            const int n = launches_left;
            for (int l = 0; l < n; l++) {
                it[l] = (iters - d + left - 1) / left;
                left--;
                const int ext = ext_out + (iters - d - it[l]);
                row_range(c->sim, c->sim_row0, c->sim_rows, ext, ra[l], rb[l]);
                const Win wl = sim_cols(c, ext);   // 2-D tiles: the columns of this block (the window's own everywhere else)
                xa[l] = wl.x0;
                xb[l] = wl.x1;
                d += it[l];
            }
            const Win w = sim_cols(c, ext_out + (iters - done - it[0]));
            if (fluid::jacobi_chain_applies(w, ra[0], rb[0], iters - done)) {
                if (!c->chain_flags || !c->chain_err_host || !c->chain_err_dev) {
                    // all three or none (a launch with the counters but without the error words would fault in its poll loop: ADVICE r05)
                    unsigned int *flags = nullptr, *eh = nullptr, *ed = nullptr;
                    int rc = c->hip(hipMalloc((void**)&flags, fluid::jacobi_chain_flag_bytes()), "hipMalloc (chain counters)");
                    if (!rc) rc = c->hip(hipHostMalloc((void**)&eh, 64 * sizeof(unsigned int), hipHostMallocMapped), "hipHostMalloc (chain error words)");
                    if (!rc) rc = c->hip(hipHostGetDevicePointer((void**)&ed, eh, 0), "hipHostGetDevicePointer");
                    if (rc) {
                        if (flags) (void)hipFree(flags);
                        if (eh) (void)hipHostFree(eh);
                        return rc;
                    }
                    for (int k = 0; k < 64; k++) eh[k] = 0;   // [0]: a chained launch gave up; [1]: lab ticket word; [8..31]: lab statistics
                    if (c->chain_flags) (void)hipFree(c->chain_flags);
                    if (c->chain_err_host) (void)hipHostFree(c->chain_err_host);
                    c->chain_flags = flags;
                    c->chain_err_host = eh;
                    c->chain_err_dev = ed;
                    c->chain_epoch = fluid::ChainEpoch{};
                }
                const bool gs = chain_gs && done == 0 && n < fluid::jacobi_chain_max_blocks();
                if (gs) {   // what the gradient-subtract block stores: the owned rows / columns (ext 0)
                    it[n] = 0;
                    row_range(c->sim, c->sim_row0, c->sim_rows, 0, ra[n], rb[n]);
                    const Win wg = sim_cols(c, 0);
                    xa[n] = wg.x0;
                    xb[n] = wg.x1;
                }
                const hipError_t e = fluid::launch_jacobi_tb_chain_ranges(c->stream, w, (float*)c->prs[0], (float*)c->prs[1], (const float*)c->div,
                                                                          done == 0 ? pscale : 1.0f, n, it, ra, rb, xa, xb, c->chain_flags, c->chain_err_dev, &c->chain_epoch,
                                                                          gs ? (const float2*)c->vel[0] : nullptr, gs ? (float2*)c->vel[1] : nullptr);
                if (e != hipErrorNotReady) {
                    CK(c->hip(e, "jacobi_tb (chain)"));
                    if (n & 1) std::swap(c->prs[0], c->prs[1]);
                    if (launches) *launches += n;
                    if (gs) {
                        std::swap(c->vel[0], c->vel[1]);
                        *gradsub = true;
                    }
                    return FLUID_OK;
                }
            }
        }
        if (tb) {
            const int k = (iters - done + launches_left - 1) / launches_left;
            launches_left--;
            if (split == 1 && cut_left == 0) return FLUID_OK;
            if (done + k == iters && fold) {
                if (split == 1) return FLUID_OK;
                row_range(c->sim, c->sim_row0, c->sim_rows, 0, ga, gb);
                if (t) t->mark(P_JACOBI);  // the launches so far; this one is timed as P_GRADSUB by the caller
                CK(c->hip(STORE_CALL(c, launch_jacobi_tb_gradsub(c->stream, sim_cols(c, 0), (const S::T1*)c->prs[0], (const S::T1*)c->div,
                                                                 (S::T1*)c->prs[1], (const S::T2*)c->vel[0], (S::T2*)c->vel[1],
                                                                 done == 0 ? pscale : 1.0f, k, ga, gb, shape)),
                          "jacobi_tb_gradsub"));
                std::swap(c->prs[0], c->prs[1]);
                std::swap(c->vel[0], c->vel[1]);
                if (launches) (*launches)++;
                *gradsub = true;
                return FLUID_OK;
            }
            row_range(c->sim, c->sim_row0, c->sim_rows, ext_out + (iters - done - k), ga, gb);
            if (cut_left > 0) {
                cut_left--;
                level++;
                const Win w = sim_cols(c, ext_out + (iters - done - k));
                const fluid::BlockCut q = jacobi_cut(c, w, ga, gb, shape, *sp, level);
                const float ps = done == 0 ? pscale : 1.0f;
                auto band = [&](const void* src, void* dst, int a, int b, int xa, int xb) {
                    if (b <= a || xb <= xa) return (int)FLUID_OK;
                    Win wb = w;
                    wb.x0 = xa;
                    wb.x1 = xb;
                    return c->hip(STORE_CALL(c, launch_jacobi_tb(c->stream, wb, (const S::T1*)src, (const S::T1*)c->div, (S::T1*)dst, ps, k, a, b, shape)),
                                  "jacobi_tb");
                };
                if (split == 1) {   // interiors only; the caller comes back with split == 2 once the ghost texels are in
                    CK(band(pa, pb, q.ia, q.ib, q.ja, q.jb));
                    std::swap(pa, pb);
                    done += k;
                    continue;
                }
                // the frame around that interior.  (With two launches cut, this one reads pressure up to one apron inside the first
                // interior, from the buffer the SECOND interior has already written into — further in: from 2 aprons + margin on.)
                // sp->frame_stream: the frames run on the comm stream, behind the exchange that brought their ghost texels
                hipStream_t main_stream = c->stream;
                if (sp->frame_stream) c->stream = sp->frame_stream;
                int rc_frame = FLUID_OK;
                if (c->storage == FLUID_STORE_F32 && k <= 10) {        // one launch
                    fluid::CutRect fr[4];
                    fluid::cut_frame(ga, gb, w.x0, w.x1, q, fr);
                    fluid::BandRects B{};
                    for (const fluid::CutRect& r : fr) B.r[B.n++] = fluid::BandRect{ r.xa, r.xb, r.ga, r.gb };
                    rc_frame = c->hip(fluid::launch_jacobi_tb_rects(c->stream, w, (const float*)c->prs[0], (const float*)c->div, (float*)c->prs[1], ps, k, B),
                                      "jacobi_tb (frame)");
                } else {   // fp16 storage / a deeper lab shape: stripes only (jacobi_split_launches), one launch per band
                    rc_frame = band(c->prs[0], c->prs[1], ga, q.ia, w.x0, w.x1);
                    if (!rc_frame) rc_frame = band(c->prs[0], c->prs[1], q.ib, gb, w.x0, w.x1);
                }
                c->stream = main_stream;
                CK(rc_frame);
                if (sp->frame_stream && cut_left == 0) {   // the last frame: the block's further launches (context stream) follow it
                    HIPCK(c, hipEventRecord(sp->frame_done, sp->frame_stream));
                    HIPCK(c, hipStreamWaitEvent(c->stream, sp->frame_done, 0));
                }
                if (launches) (*launches)++;
                std::swap(c->prs[0], c->prs[1]);
                done += k;
                continue;
            }
            CK(c->hip(STORE_CALL(c, launch_jacobi_tb(c->stream, sim_cols(c, ext_out + (iters - done - k)), (const S::T1*)c->prs[0], (const S::T1*)c->div,
                                                     (S::T1*)c->prs[1], done == 0 ? pscale : 1.0f, k, ga, gb, shape)),
                      "jacobi_tb"));
            done += k;
        } else {
            row_range(c->sim, c->sim_row0, c->sim_rows, ext_out + (iters - done - 1), ga, gb);
            CK(c->hip(STORE_CALL(c, launch_jacobi(c->stream, sim_cols(c, ext_out + (iters - done - 1)), PRS(c, 0), DIVG(c), PRS(c, 1), ga, gb)),
                      "jacobi"));
            done += 1;
        }
        std::swap(c->prs[0], c->prs[1]);
        if (launches) (*launches)++;
    }
    return FLUID_OK;
}

// K4 + K5 x iters; the clear rides on the first temporally blocked launch when that kernel applies
int pass_clear_jacobi(fluid_ctx* c, float value, int iters, int ext_out, int* launches, bool* gradsub)
{
    if (iters < 0) return c->fail(FLUID_ERR_INVALID, "negative iteration count");
    const bool fold = jacobi_tb_applies(c) && iters > 0;
    if (fold) return pass_jacobi(c, iters, ext_out, value, launches, gradsub, nullptr);
    if (gradsub) *gradsub = false;
    CK(pass_clear(c, value, ext_out + iters));
    return pass_jacobi(c, iters, ext_out, 1.0f, launches, nullptr, nullptr);
}

int pass_gradsub(fluid_ctx* c, int ext)
{
    CK(check_ext(c, ext, 1));
    int ga, gb;
    row_range(c->sim, c->sim_row0, c->sim_rows, ext, ga, gb);
    if (c->desc.schedule == FLUID_SCHED_FUSED && fused_supported(c->sim))
        CK(c->hip(STORE_CALL(c, launch_gradsub4(c->stream, sim_cols(c, ext), PRS(c, 0), VEL(c, 0), VEL(c, 1), ga, gb)), "gradsub"));
    else
        CK(c->hip(STORE_CALL(c, launch_gradsub(c->stream, sim_cols(c, ext), PRS(c, 0), VEL(c, 0), VEL(c, 1), ga, gb)), "gradsub"));
    std::swap(c->vel[0], c->vel[1]);
    return FLUID_OK;
}

// ---- the dye field packed to three floats per texel while its alpha is one known value (fluid_internal.h) ----
// FLUID_DYE_PACK=0 (lab build): never pack (A/B knob; same bits either way)
bool dye_pack_enabled()
{
    static const bool on = [] {
        const char* e = fluid::lab_env("FLUID_DYE_PACK");
        return !(e && atoi(e) == 0);
    }();
    return on;
}

int ensure_rgba(fluid_ctx* c)
{
    if (!c->dye_packed) return FLUID_OK;
    CK(c->hip(fluid::launch_dye_unpack(c->stream, (const fluid::rgb3*)c->dyeb[0], (float4*)c->dyeb[1], cells(c->dye), c->dye_alpha), "unpack dye"));
    std::swap(c->dyeb[0], c->dyeb[1]);
    c->dye_packed = false;
    // somebody needs RGBA texels every few steps: packing would cost more than it saves.  Whole-domain contexts only: on a stripe / tile the
    // format of the dye is part of the exchange's message layout, so it may depend on nothing one rank does alone (fluid_internal.h)
    if (c->packed_advects < 16 && c->desc.parts == 1 && c->desc.parts_x == 1) c->pack_holdoff = 256;
    return FLUID_OK;
}

int ensure_packed(fluid_ctx* c)
{
    if (c->dye_packed) return FLUID_OK;
    CK(c->hip(fluid::launch_dye_pack(c->stream, (const float4*)c->dyeb[0], (fluid::rgb3*)c->dyeb[1], cells(c->dye)), "pack dye"));
    std::swap(c->dyeb[0], c->dyeb[1]);
    c->dye_packed = true;
    c->packed_advects = 0;
    return FLUID_OK;
}

// the advection divides every dye texel, alpha included, by 1 + dissipation dt (advectionShader script.js:780-782): the same fp32 division here
void note_dye_advected(fluid_ctx* c, float dt, float dissipation)
{
    if (c->alpha_known) c->dye_alpha = c->dye_alpha / (1.0f + dissipation * dt);
}

// an fp32 context whose DYE passes are bandwidth-bound (at least kSmallGridTexels owned dye texels: on the sim grid that is where no launches
// are chained either) and whose alpha is known; fused schedule (the per-pass kernels read RGBA).  A stripe / tile context packs where the one
// fused advection kernel applies (dye grid = sim grid): its ghost rows then travel as 12-byte texels too (fluid_stripes.cpp).
bool dye_pack_applies(const fluid_ctx* c)
{
    const bool whole = c->desc.parts == 1 && c->desc.parts_x == 1;
    if (!whole && c->dye_set_rgba) return false;   // the set agreed on RGBA for this call: some rank cannot pack (fluid_stripes.cpp dye_format_agree)
    return dye_pack_enabled() && c->alpha_known && c->storage == FLUID_STORE_F32 && (whole || fused_advect_applies(c)) &&
           c->desc.schedule == FLUID_SCHED_FUSED && (long)c->dye_ncols * c->dye_rows >= fluid::kSmallGridTexels;
}

int pass_advect_velocity(fluid_ctx* c, float dt, float dissipation, int ext)
{
    CK(check_ext(c, ext, 0));
    int ga, gb;
    row_range(c->sim, c->sim_row0, c->sim_rows, ext, ga, gb);
    CK(c->hip(STORE_CALL(c, launch_advect_velocity(c->stream, sim_cols(c, ext), VEL(c, 0), VEL(c, 1), dt, dissipation, ga, gb, c->miss)),
              "advect velocity"));
    std::swap(c->vel[0], c->vel[1]);
    return FLUID_OK;
}

int pass_advect_dye(fluid_ctx* c, float dt, float dissipation)
{
    int ga, gb;
    row_range(c->dye, c->dye_row0, c->dye_rows, 0, ga, gb);
    if (c->pack_holdoff > 0 && !c->dye_packed) c->pack_holdoff--;
    if (dye_pack_applies(c) && (c->dye_packed || c->pack_holdoff == 0) &&
        fluid::advect_rgb_supported(c->sim, dye_cols(c, 0), dt, dissipation, dissipation)) {   // the dye as three floats per texel: 24 instead of 32 B/texel
        CK(ensure_packed(c));
        const hipError_t e = fluid::launch_advect_dye_rgb(c->stream, c->sim, (const float2*)c->vel[0], dye_cols(c, 0), (const fluid::rgb3*)c->dyeb[0],
                                                          (fluid::rgb3*)c->dyeb[1], dt, dissipation, ga, gb, c->miss);
        if (e != hipErrorNotReady) {
            CK(c->hip(e, "advect dye"));
            c->packed_advects++;
            note_dye_advected(c, dt, dissipation);
            std::swap(c->dyeb[0], c->dyeb[1]);
            return FLUID_OK;
        }
    }
    CK(ensure_rgba(c));
    note_dye_advected(c, dt, dissipation);
    CK(c->hip(STORE_CALL(c, launch_advect_dye(c->stream, c->sim, VEL(c, 0), dye_cols(c, 0), DYE(c, 0), DYE(c, 1), dt, dissipation, ga, gb, c->miss)),
              "advect dye"));
    std::swap(c->dyeb[0], c->dyeb[1]);
    return FLUID_OK;
}

// K7a + K7b: one kernel under the fused schedule when the dye grid is the sim grid, two launches otherwise.
// (In a stripe the velocity is advected one ghost row out when the grids differ: the dye pass then samples it bilinearly.)
int pass_advect(fluid_ctx* c, float dt, float vel_diss, float dye_diss, Timer* t)
{
    const bool same = c->sim.W == c->dye.W && c->sim.H == c->dye.H;
    if (fused_advect_applies(c)) {
        int ga, gb;
        row_range(c->sim, c->sim_row0, c->sim_rows, 0, ga, gb);
        if (c->pack_holdoff > 0 && !c->dye_packed) c->pack_holdoff--;
        if (dye_wants_packed(c, dt, vel_diss, dye_diss)) {   // 40 instead of 48 B/texel: the dye as three floats, its uniform alpha as a scalar
            CK(ensure_packed(c));
            c->packed_advects++;
            const hipError_t e = fluid::launch_advect_both_rgb(c->stream, sim_cols(c, 0), (const float2*)c->vel[0], (float2*)c->vel[1],
                                                               (const fluid::rgb3*)c->dyeb[0], (fluid::rgb3*)c->dyeb[1], dt, vel_diss, dye_diss, ga, gb, c->miss);
            if (e != hipErrorNotReady) {
                CK(c->hip(e, "advect"));
                note_dye_advected(c, dt, dye_diss);
                std::swap(c->vel[0], c->vel[1]);
                std::swap(c->dyeb[0], c->dyeb[1]);
                if (t) t->mark(P_ADVD);
                return FLUID_OK;
            }
        }
        CK(ensure_rgba(c));   // (also when the fast kernel does not apply to these decays: the general kernel reads RGBA)
        note_dye_advected(c, dt, dye_diss);
        CK(c->hip(STORE_CALL(c, launch_advect_both(c->stream, sim_cols(c, 0), VEL(c, 0), VEL(c, 1), DYE(c, 0), DYE(c, 1), dt, vel_diss, dye_diss, ga, gb,
                                                   c->miss)),
                  "advect"));
        std::swap(c->vel[0], c->vel[1]);
        std::swap(c->dyeb[0], c->dyeb[1]);
        if (t) t->mark(P_ADVD);
        return FLUID_OK;
    }
    CK(pass_advect_velocity(c, dt, vel_diss, (c->desc.parts > 1 && !same) ? 1 : 0));
    if (t) t->mark(P_ADVV);
    CK(pass_advect_dye(c, dt, dye_diss));
    if (t) t->mark(P_ADVD);
    return FLUID_OK;
}

// ---- band forms for the stripe driver: one row band of a single-kernel pass, WITHOUT the ping-pong swap, so that a
//      pass can run as "interior rows while the ghost rows are in flight, then the strips next to them" ----
// the temporally blocked Jacobi kernel exists for both storage types
// K6 inside the last Jacobi launch?  FLUID_FOLD_GRADSUB=0 / 1 forces it off / on (A/B knob; same bits either way).  Default: on small
// grids only, where a step is a chain of latency-bound launches and one launch fewer is worth 4-8 %; at 4096^2 the folded launch saves
// 12 us of pass time and the step is not faster for it (profiles/r03/gradsub_fold_ab.txt)
bool gradsub_fold_enabled(long owned_texels)
{
    static const int mode = [] {
        const char* e = fluid::lab_env("FLUID_FOLD_GRADSUB");
        return e ? (atoi(e) != 0 ? 1 : 0) : -1;
    }();
    return mode >= 0 ? mode == 1 : owned_texels < fluid::kSmallGridTexels;
}

void mark_step(fluid_ctx* c, int k)
{
    if (c->marks.empty()) return;
    if (k == 0) c->marks_used = 0;
    if (k < (int)c->marks.size() && k == c->marks_used) {
        (void)hipEventRecord(c->marks[k], c->stream);
        c->marks_used = k + 1;
    }
}

// FLUID_SKIP_CURL=0: every step of fluid_step_n stores its curl field (A/B knob; what a caller can read is the same either way)
bool skip_hidden_curl()
{
    static const bool on = [] {
        const char* e = fluid::lab_env("FLUID_SKIP_CURL");
        return !(e && atoi(e) == 0);
    }();
    return on;
}

bool jacobi_tb_applies(const fluid_ctx* c) { return c->desc.schedule == FLUID_SCHED_FUSED && jacobi_tb_supported(c->sim); }

bool fused_cvd_applies(const fluid_ctx* c) { return c->desc.schedule == FLUID_SCHED_FUSED && fused_supported(c->sim); }

bool fused_advect_applies(const fluid_ctx* c) { return c->desc.schedule == FLUID_SCHED_FUSED && c->sim.W == c->dye.W && c->sim.H == c->dye.H; }

void sim_band(const fluid_ctx* c, int ext, int& ga, int& gb) { row_range(c->sim, c->sim_row0, c->sim_rows, ext, ga, gb); }

int cvd_band(fluid_ctx* c, float curl, float dt, int ga, int gb, int xa, int xb)
{
    Win w = c->sim;
    w.x0 = xa;
    w.x1 = xb;
    c->curl_valid = c->keep_curl;
    return c->hip(STORE_CALL(c, launch_curl_vort_div(c->stream, w, VEL(c, 0), CURL_FUSED(c), VEL(c, 1), DIVG(c), curl, dt, ga, gb)), "curl_vort_div");
}

int cvd_rects(fluid_ctx* c, float curl, float dt, const BandRects& B)
{
    c->curl_valid = c->keep_curl;
    return c->hip(STORE_CALL(c, launch_curl_vort_div_rects(c->stream, c->sim, VEL(c, 0), CURL_FUSED(c), VEL(c, 1), DIVG(c), curl, dt, B)), "curl_vort_div");
}

void cvd_swap(fluid_ctx* c) { std::swap(c->vel[0], c->vel[1]); }

// the format the dye takes through THIS step's advection, decided once per step in front of the exchange that refreshes its ghost rows
// (fluid_stripes.cpp) — and by pass_advect for a whole domain: packed where packing applies, the fast kernel takes these decays and the
// field is not in a hold-off
bool dye_wants_packed(const fluid_ctx* c, float dt, float vel_diss, float dye_diss)
{
    if (!dye_pack_applies(c) || !(c->dye_packed || c->pack_holdoff == 0)) return false;
    const bool same = c->sim.W == c->dye.W && c->sim.H == c->dye.H;
    return same ? fluid::advect_rgb_supported(sim_cols(c, 0), sim_cols(c, 0), dt, vel_diss, dye_diss)
                : fluid::advect_rgb_supported(c->sim, dye_cols(c, 0), dt, dye_diss, dye_diss);
}

int dye_prepare(fluid_ctx* c, float dt, const fluid_params* P)
{
    return dye_wants_packed(c, dt, P->velocity_dissipation, P->density_dissipation) ? ensure_packed(c) : ensure_rgba(c);
}

void advect_both_note(fluid_ctx* c, float dt, float dye_diss)
{
    if (c->dye_packed) c->packed_advects++;
    note_dye_advected(c, dt, dye_diss);
}

int advect_both_band(fluid_ctx* c, float dt, float vel_diss, float dye_diss, int ga, int gb, int xa, int xb, int v0, int v1, int u0, int u1)
{
    Win w = c->sim;  // dye grid == sim grid: one window serves both gathers
    w.x0 = xa;
    w.x1 = xb;
    w.v0 = v0;
    w.v1 = v1;
    w.u0 = u0;
    w.u1 = u1;
    if (c->dye_packed) {   // (dye_prepare left it packed because the fast kernel takes these decays: never hipErrorNotReady)
        return c->hip(fluid::launch_advect_both_rgb(c->stream, w, (const float2*)c->vel[0], (float2*)c->vel[1], (const fluid::rgb3*)c->dyeb[0],
                                                    (fluid::rgb3*)c->dyeb[1], dt, vel_diss, dye_diss, ga, gb, c->miss),
                      "advect");
    }
    return c->hip(STORE_CALL(c, launch_advect_both(c->stream, w, VEL(c, 0), VEL(c, 1), DYE(c, 0), DYE(c, 1), dt, vel_diss, dye_diss, ga, gb, c->miss)),
                  "advect");
}

int advect_both_rects(fluid_ctx* c, float dt, float vel_diss, float dye_diss, const BandRects& B, int v0, int v1, int u0, int u1)
{
    Win w = c->sim;  // dye grid == sim grid: one window serves both gathers
    w.v0 = v0;
    w.v1 = v1;
    w.u0 = u0;
    w.u1 = u1;
    if (c->dye_packed) {
        return c->hip(fluid::launch_advect_both_rects_rgb(c->stream, w, (const float2*)c->vel[0], (float2*)c->vel[1], (const fluid::rgb3*)c->dyeb[0],
                                                          (fluid::rgb3*)c->dyeb[1], dt, vel_diss, dye_diss, B, c->miss),
                      "advect");
    }
    return c->hip(STORE_CALL(c, launch_advect_both_rects(c->stream, w, VEL(c, 0), VEL(c, 1), DYE(c, 0), DYE(c, 1), dt, vel_diss, dye_diss, B, c->miss)),
                  "advect");
}

void advect_both_swap(fluid_ctx* c)
{
    std::swap(c->vel[0], c->vel[1]);
    std::swap(c->dyeb[0], c->dyeb[1]);
}

}  // namespace fluid_impl

namespace {

// step(dt), script.js:1231-1294 — whole-domain contexts (a stripe runs the plan of fluid_stripes.cpp, with
// ghost-row exchanges between the pass groups).
// `lead`: the step starts with its own curl / vorticity / divergence launch (false: the previous step's advection launch already ran them,
// k_advect_cvd).  `chain`: 0 = the step ends with the advection launch; 1 / 2 = it ends with the launch that advects AND runs the next
// step's curl / vorticity / divergence (2: and writes the curl field — the chain's last such launch, whose curl a caller can read);
// 3 = the same at the END of a call: the advected velocity is stored for the caller and the next step's results go to the pending buffers.
int step_once(fluid_ctx* c, float dt, const fluid_params* P, bool lead = true, int chain = 0)
{
    Timer t(c);
    if (lead) CK(pass_curl_vort_div(c, P->curl, dt, 0, &t));
    int launches = 0;
    const bool fold_clear = jacobi_tb_applies(c) && P->iterations > 0;
    if (!fold_clear) {
        CK(pass_clear(c, P->pressure, 0));
        t.mark(P_CLEAR);
    }
    bool gradsub_done = true;  // ask for K6 inside the last Jacobi launch (taken on small grids: pass_jacobi)
    CK(pass_jacobi(c, P->iterations, 0, fold_clear ? P->pressure : 1.0f, &launches, &gradsub_done, &t));
    if (!gradsub_done) {
        t.mark(P_JACOBI);
        CK(pass_gradsub(c, 0));
    }
    t.mark(P_GRADSUB);
    const bool with_dye = fused_advect_applies(c);   // the launch advects the dye too (dye grid = sim grid)
    if (chain && with_dye) {
        CK(ensure_rgba(c));   // (chained launches run below kSmallGridTexels, packing at and above it: never both)
        note_dye_advected(c, dt, P->density_dissipation);
    }
    if (chain == 3) {  // advect, and run the NEXT step's curl / vorticity / divergence into the pending buffers
        int ga, gb;
        sim_band(c, 0, ga, gb);
        CK(c->hip(fluid::launch_advect_cvd(c->stream, sim_cols(c, 0), (const float2*)c->vel[0], (float2*)c->pend_vel,
                                           with_dye ? (const float4*)c->dyeb[0] : nullptr, with_dye ? (float4*)c->dyeb[1] : nullptr,
                                           (float*)c->pend_curl, (float*)c->pend_div, (float2*)c->vel[1], dt, P->velocity_dissipation,
                                           P->density_dissipation, P->curl, ga, gb),
                  "advect + the next step's curl_vort_div"));
        std::swap(c->vel[0], c->vel[1]);   // the advected velocity: what the caller reads — and what the dye pass samples
        if (with_dye) std::swap(c->dyeb[0], c->dyeb[1]);
        c->pend_valid = true;
        c->pend_dt = dt;
        c->pend_curl_strength = P->curl;
        if (with_dye) {
            t.mark(P_ADVD);
        } else {   // dye grid != sim grid: the dye pass is its own launch on its own grid, behind the advected velocity
            t.mark(P_ADVV);
            CK(pass_advect_dye(c, dt, P->density_dissipation));
            t.mark(P_ADVD);
        }
    } else if (chain) {
        int ga, gb;
        sim_band(c, 0, ga, gb);
        CK(c->hip(fluid::launch_advect_cvd(c->stream, sim_cols(c, 0), (const float2*)c->vel[0], (float2*)c->vel[1], (const float4*)c->dyeb[0],
                                           (float4*)c->dyeb[1], chain == 2 ? (float*)c->curl : nullptr, (float*)c->div, nullptr, dt,
                                           P->velocity_dissipation, P->density_dissipation, P->curl, ga, gb),
                  "advect + curl_vort_div"));
        std::swap(c->vel[0], c->vel[1]);
        std::swap(c->dyeb[0], c->dyeb[1]);
        t.mark(P_ADVD);  // charged to the advection: the next step's vorticity column then reads 0
    } else {
        CK(pass_advect(c, dt, P->velocity_dissipation, P->density_dissipation, &t));
    }
    if (c->timing) {
        c->acc_steps++;
        c->acc_jacobi_launches += launches;
        c->acc_folded_launches += gradsub_done ? 1 : 0;
    }
    return FLUID_OK;
}

// n steps in one call: nobody sees the fields between them, so each step but the last hands its advected velocity to the next step's
// curl / vorticity / divergence inside one launch (k_advect_cvd) instead of through memory.  After the call every field holds what n
// separate calls leave (tests/test_hip_properties.py::test_step_n_equals_n_steps, bitwise).
// FLUID_CHAIN=0 / 1 forces it off / on (A/B knob; same bits either way).  Default: below 3072^2 texels, like the folded gradient subtract
// and for the same reason: there a step is a chain of latency-bound launches and one launch fewer is worth 4-7 % (1024^2: 13.3 k -> 13.9 k
// steps/s, 2048^2: 6.3 k -> 6.75 k).  At 4096^2 the combined launch has the bytes of 0.70 of the two it replaces and takes their time
// (200-222 us against 142 + 73): one texel per lane is what the gathers want and costs the stencil stages a 22 % apron, and at 62-70 M
// VALU wave-instructions the launch is bound by issue slots, not by bytes (profiles/r03/advect_cvd_chain.txt).
bool chain_enabled(long owned_texels)
{
    static const int mode = [] {
        const char* e = fluid::lab_env("FLUID_CHAIN");
        return e ? (atoi(e) != 0 ? 1 : 0) : -1;
    }();
    return mode >= 0 ? mode == 1 : owned_texels < fluid::kSmallGridTexels;
}

// FLUID_RUN_AHEAD=0 (lab build): a call never ends with the launch that computes the next call's curl / vorticity / divergence ahead
bool run_ahead_enabled(long owned_texels)
{
    static const int mode = [] {
        const char* e = fluid::lab_env("FLUID_RUN_AHEAD");   // 0 / 1 force it off / on (lab build)
        return e ? (atoi(e) != 0 ? 1 : 0) : -1;
    }();
    return mode >= 0 ? mode == 1 : owned_texels < fluid::kRunAheadTexels;
}

int pending_buffers(fluid_ctx* c)   // allocated on first use: whole-domain fp32 contexts below 3072^2 texels only (<= 150 MB)
{
    if (c->pend_vel && c->pend_div && c->pend_curl) return FLUID_OK;
    if (c->pend_failed) return FLUID_ERR_OOM;   // tried before and the device had no room: this context steps without working ahead
    const size_t n = cells(c->sim);
    int rc = FLUID_OK;
    // all three or none: a partial set would look usable to the next call (pend_vel alone used to be the test) and hand k_advect_cvd null outputs
    if (!rc && !c->pend_vel) rc = c->hip(hipMalloc(&c->pend_vel, n * 2 * sizeof(float)), "hipMalloc pending velocity");
    if (!rc && !c->pend_div) rc = c->hip(hipMalloc(&c->pend_div, n * sizeof(float)), "hipMalloc pending divergence");
    if (!rc && !c->pend_curl) rc = c->hip(hipMalloc(&c->pend_curl, n * sizeof(float)), "hipMalloc pending curl");
    // the padding columns are never meaningful but are copied around: give them defined content once
    if (!rc) rc = c->hip(hipMemsetAsync(c->pend_vel, 0, n * 2 * sizeof(float), c->stream), "memset pending velocity");
    if (!rc) rc = c->hip(hipMemsetAsync(c->pend_div, 0, n * sizeof(float), c->stream), "memset pending divergence");
    if (!rc) rc = c->hip(hipMemsetAsync(c->pend_curl, 0, n * sizeof(float), c->stream), "memset pending curl");
    if (rc) {
        for (void** p : { &c->pend_vel, &c->pend_div, &c->pend_curl }) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        c->pend_failed = true;   // (a resize clears it: other sizes, another try)
    }
    return rc;
}

bool chain_applies(const fluid_ctx* c, float dt, const fluid_params* P)
{
    return c->storage == FLUID_STORE_F32 && fused_cvd_applies(c) && fused_advect_applies(c) && chain_enabled((long)c->sim_ncols * c->sim_rows) &&
           fluid::advect_cvd_supported(sim_cols(c, 0), dt, P->velocity_dissipation, P->density_dissipation);
}

// dye grid != sim grid (the reference's default shape, script.js:60-66): K7a and the next step's K1-K3 still go into one launch — without
// the dye, which keeps its own pass behind it — with EVERY step handing its successor's curl / vorticity / divergence over through the
// pending buffers (step_once chain 3).  Five launches per step become four on the tiny sim grids of that shape.
bool split_chain_applies(const fluid_ctx* c, float dt, const fluid_params* P)
{
    const long owned = (long)c->sim_ncols * c->sim_rows;
    return c->storage == FLUID_STORE_F32 && c->desc.parts == 1 && c->desc.parts_x == 1 && fused_cvd_applies(c) && !fused_advect_applies(c) &&
           chain_enabled(owned) && run_ahead_enabled(owned) && fluid::advect_cvd_velocity_supported(sim_cols(c, 0), dt, P->velocity_dissipation);
}

}  // namespace

namespace fluid_impl {

int chain_check(fluid_ctx* c)
{
    if (!c->chain_err_host || c->chain_err_host[0] == 0) return FLUID_OK;
    const unsigned int why = c->chain_err_host[0];
    c->chain_err_host[0] = 0;
    c->chain_epoch = fluid::ChainEpoch{};   // the counters are in no known state: a chained launch would have to zero them
    c->chain_broken = true;                 // ... and this context does not try again: plain launches from here on (pass_jacobi)
    return c->fail(FLUID_ERR_HIP, why == 2u ? "the chained Jacobi launch ran out of room for the items it had put aside: the fields of the calls since the last "
                                              "synchronisation are not valid; this context keeps to one launch per block of iterations from now on"
                                            : "the chained Jacobi launch gave up waiting for an item of the previous block of iterations: the fields of the calls "
                                              "since the last synchronisation are not valid; this context keeps to one launch per block of iterations from now on");
}

int ctx_sync(fluid_ctx* c, hipStream_t s)
{
    HIPCK(c, hipStreamSynchronize(s ? s : c->stream));
    return chain_check(c);
}

int field_ref(fluid_ctx* c, int field, FieldRef* f, bool geometry_only, bool keep_packed)
{
    const int h = c->desc.parts > 1 ? c->desc.halo : 0, hx = c->desc.parts_x > 1 ? c->desc.halo : 0;
    switch (field) {
    case FLUID_VELOCITY: *f = { c->vel[0], &c->sim, c->sim_row0, c->sim_rows, h, 2, c->sim_col0, c->sim_ncols, hx, c->esz }; break;
    case FLUID_PRESSURE: *f = { c->prs[0], &c->sim, c->sim_row0, c->sim_rows, h, 1, c->sim_col0, c->sim_ncols, hx, c->esz }; break;
    case FLUID_DIVERGENCE: *f = { c->div, &c->sim, c->sim_row0, c->sim_rows, h, 1, c->sim_col0, c->sim_ncols, hx, c->esz }; break;
    case FLUID_CURL:
        if (!geometry_only && !c->curl_valid)
            return c->fail(FLUID_ERR_INVALID, "the last step did not store its curl field (fluid_set_curl_output(ctx, 0)): switch the output on and step");
        *f = { c->curl, &c->sim, c->sim_row0, c->sim_rows, h, 1, c->sim_col0, c->sim_ncols, hx, c->esz };
        break;
    case FLUID_DYE:
        // whoever asks for the dye field's MEMORY (read, write, ghost rows, a raw pointer) gets RGBA texels — except the stripe / tile driver's
        // own exchanges (keep_packed), which move the ghost texels in whatever format the field is in: 3 channels while it is packed
        if (!geometry_only && !keep_packed) CK(ensure_rgba(c));
        *f = { c->dyeb[0], &c->dye, c->dye_row0, c->dye_rows, c->dye_halo, (keep_packed && !geometry_only && c->dye_packed) ? 3 : 4,
               c->dye_col0, c->dye_ncols, c->dye_halo_x, c->esz };
        break;
    default: return c->fail(FLUID_ERR_INVALID, "unknown field id");
    }
    return FLUID_OK;
}

}  // namespace fluid_impl

// ================================================================================================
extern "C" {

int fluid_abi_version(void) { return FLUID_ABI_VERSION; }

const char* fluid_build_flavor(void)
{
#ifdef FLUID_PROBES
    return "probes";
#else
    return "product";
#endif
}

const char* fluid_error_string(int status)
{
    switch (status) {
    case FLUID_OK: return "ok";
    case FLUID_ERR_INVALID: return "invalid argument";
    case FLUID_ERR_HIP: return "HIP runtime error";
    case FLUID_ERR_NO_DEVICE: return "no HIP device";
    case FLUID_ERR_OOM: return "out of device memory";
    case FLUID_ERR_HALO: return "advection back-trace left the stripe's ghost rows";
    case FLUID_ERR_UNSUPPORTED: return "unsupported";
    case FLUID_ERR_COMM: return "RCCL unavailable / failed, or stripe without communicator";
    default: return "unknown status";
    }
}

const char* fluid_last_error(const fluid_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int fluid_device_count(int* count)
{
    if (!count) return FLUID_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return FLUID_OK;
}

int fluid_create(const fluid_desc* desc, fluid_ctx** out)
{
    if (!desc || !out) return FLUID_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        g_create_error = "no HIP device visible: libfluid_hip has no CPU path";
        return FLUID_ERR_NO_DEVICE;
    }
    if (desc->device < 0 || desc->device >= n) {
        g_create_error = "device ordinal out of range";
        return FLUID_ERR_INVALID;
    }
    fluid_ctx* c = new (std::nothrow) fluid_ctx();
    if (!c) return FLUID_ERR_OOM;
    c->desc = *desc;
    if (c->desc.parts < 1) c->desc.parts = 1;
    if (c->desc.parts_x < 1) c->desc.parts_x = 1;
    if (c->desc.parts == 1 && c->desc.parts_x == 1) c->desc.halo = 0;
    if (desc->storage != FLUID_STORE_F32 && desc->storage != FLUID_STORE_F16) {
        g_create_error = "unknown storage mode";
        delete c;
        return FLUID_ERR_INVALID;
    }
    c->storage = desc->storage;
    c->esz = desc->storage == FLUID_STORE_F16 ? 2 : sizeof(float);
    c->device = desc->device;
    int rc = FLUID_OK;
    do {
        if ((rc = c->hip(hipSetDevice(c->device), "hipSetDevice"))) break;
        if ((rc = c->hip(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking), "hipStreamCreate"))) break;
        c->stream = c->own_stream;
        if ((rc = set_geometry(c, desc->sim_w, desc->sim_h, desc->dye_w, desc->dye_h))) break;
        if ((rc = c->hip(hipMalloc((void**)&c->miss, sizeof(unsigned int)), "hipMalloc"))) break;
        if ((rc = c->hip(hipMemsetAsync(c->miss, 0, sizeof(unsigned int), c->stream), "memset"))) break;
        for (auto& e : c->ev)
            if ((rc = c->hip(hipEventCreate(&e), "hipEventCreate"))) break;
        if (rc) break;
        if ((rc = alloc_fields(c))) break;
        if ((rc = c->hip(hipStreamSynchronize(c->stream), "sync"))) break;
    } while (0);
    if (rc != FLUID_OK) {
        g_create_error = c->err;
        fluid_destroy(c);
        return rc;
    }
    c->alpha_known = c->storage == FLUID_STORE_F32;   // alloc_fields filled alpha = 1, ghost rows and columns included
    c->dye_alpha = 1.0f;
    *out = c;
    return FLUID_OK;
}

int fluid_destroy(fluid_ctx* c)
{
    if (!c) return FLUID_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    stripes_release(c);
    display_release(c);
    free_fields(c);
    if (c->miss) (void)hipFree(c->miss);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : c->marks) (void)hipEventDestroy(e);
    for (auto& e : c->chain_ev) (void)hipEventDestroy(e);
    if (c->ev_order) (void)hipEventDestroy(c->ev_order);
    if (c->chain_flags) (void)hipFree(c->chain_flags);
    if (c->chain_err_host) (void)hipHostFree(c->chain_err_host);
    if (c->chain_stream) (void)hipStreamDestroy(c->chain_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return FLUID_OK;
}

int fluid_resize(fluid_ctx* c, int sw, int sh, int dw, int dh)
{
    if (!c) return FLUID_ERR_INVALID;
    c->touched();
    if (c->desc.parts != 1 || c->desc.parts_x != 1) return c->fail(FLUID_ERR_UNSUPPORTED, "resize of a stripe / tile context");
    if (sw < 1 || sh < 1 || dw < 1 || dh < 1) return c->fail(FLUID_ERR_INVALID, "field sizes must be >= 1");
    HIPCK(c, hipSetDevice(c->device));
    CK(ensure_rgba(c));
    const Win osim = c->sim, odye = c->dye;
    const bool sim_changed = (sw != osim.W || sh != osim.H), dye_changed = (dw != odye.W || dh != odye.H);
    const Win ns = make_win(sw, sh, 0, sh), nd = make_win(dw, dh, 0, dh);
    // Everything new is allocated and filled first; the context only changes once nothing can fail any more, so a
    // failed resize (out of memory at a larger size) leaves the old fields in place and usable.
    void* ndye[2] = { nullptr, nullptr };
    void* nvel[2] = { nullptr, nullptr };
    void* nscal[4] = { nullptr, nullptr, nullptr, nullptr };  // pressure.read, pressure.write, divergence, curl
    int rc = FLUID_OK;
    do {
        // resizeDoubleFBO (script.js:1116-1126): read <- bilinear copy of the old read, write <- fresh zero texture
        if (dye_changed) {
            for (int k = 0; k < 2 && !rc; k++) rc = c->hip(hipMalloc(&ndye[k], cells(nd) * 4 * c->esz), "hipMalloc dye");
            if (rc) break;
            if ((rc = c->hip(STORE_CALL(c, launch_resample(c->stream, odye, (const S::T1*)c->dyeb[0], 4, nd, (S::T1*)ndye[0])), "resample dye"))) break;
            if ((rc = c->hip(STORE_CALL(c, launch_fill(c->stream, (S::T1*)ndye[1], cells(nd), 4, 0.f, 0.f, 0.f, 1.f)), "fill dye"))) break;
        }
        if (sim_changed) {
            for (int k = 0; k < 2 && !rc; k++) rc = c->hip(hipMalloc(&nvel[k], cells(ns) * 2 * c->esz), "hipMalloc velocity");
            for (int k = 0; k < 4 && !rc; k++) rc = c->hip(hipMalloc(&nscal[k], cells(ns) * c->esz), "hipMalloc scalar field");
            if (rc) break;
            if ((rc = c->hip(STORE_CALL(c, launch_resample(c->stream, osim, (const S::T1*)c->vel[0], 2, ns, (S::T1*)nvel[0])), "resample velocity"))) break;
            if ((rc = c->hip(hipMemsetAsync(nvel[1], 0, cells(ns) * 2 * c->esz, c->stream), "memset velocity"))) break;
        }
        rc = c->hip(hipStreamSynchronize(c->stream), "sync");
    } while (0);
    if (rc != FLUID_OK) {
        for (auto p : ndye) if (p) (void)hipFree(p);
        for (auto p : nvel) if (p) (void)hipFree(p);
        for (auto p : nscal) if (p) (void)hipFree(p);
        return rc;
    }
    if (dye_changed)
        for (int k = 0; k < 2; k++) {
            (void)hipFree(c->dyeb[k]);
            c->dyeb[k] = ndye[k];
        }
    if (sim_changed) {
        for (int k = 0; k < 2; k++) {
            (void)hipFree(c->vel[k]);
            (void)hipFree(c->prs[k]);
            c->vel[k] = nvel[k];
            c->prs[k] = nscal[k];
        }
        (void)hipFree(c->div);
        (void)hipFree(c->curl);
        c->div = nscal[2];
        c->curl = nscal[3];
        for (void** p : { &c->pend_vel, &c->pend_div, &c->pend_curl }) {   // of the old size: reallocated when a call next works ahead
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        c->pend_failed = false;
    }
    c->desc.sim_w = sw;
    c->desc.sim_h = sh;
    c->desc.dye_w = dw;
    c->desc.dye_h = dh;
    CK(set_geometry(c, sw, sh, dw, dh));
    // divergence, curl and pressure are recreated on every initFramebuffers() (script.js:1004-1006)
    CK(zero_scalar_fields(c));
    return FLUID_OK;
}

int fluid_set_schedule(fluid_ctx* c, int schedule)
{
    if (!c) return FLUID_ERR_INVALID;
    c->touched();
    if (schedule != FLUID_SCHED_PASSES && schedule != FLUID_SCHED_FUSED) return c->fail(FLUID_ERR_INVALID, "unknown schedule");
    c->desc.schedule = schedule;
    return FLUID_OK;
}

int fluid_set_stream(fluid_ctx* c, void* hip_stream, int external)
{
    if (!c) return FLUID_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->device));
    CK(ctx_sync(c));
    c->stream = external ? (hipStream_t)hip_stream : c->own_stream;
    return FLUID_OK;
}

int fluid_pass_splat(fluid_ctx* c, int field, float x, float y, float aspect, float radius, float c0, float c1, float c2)
{
    if (!c) return FLUID_ERR_INVALID;
    c->touched();
    HIPCK(c, hipSetDevice(c->device));
    int ga, gb;
    if (field == FLUID_VELOCITY) {
        row_range(c->sim, c->sim.g0, c->sim.rows, 0, ga, gb);
        CK(c->hip(STORE_CALL(c, launch_splat_velocity(c->stream, sim_cols(c, c->desc.halo), VEL(c, 0), VEL(c, 1), x, y, aspect, radius, c0, c1, ga, gb)),
                  "splat velocity"));
        std::swap(c->vel[0], c->vel[1]);
    } else if (field == FLUID_DYE) {
        row_range(c->dye, c->dye.g0, c->dye.rows, 0, ga, gb);
        if (c->dye_packed)
            CK(c->hip(fluid::launch_splat_dye_rgb(c->stream, dye_cols(c, c->dye_halo_x), (const fluid::rgb3*)c->dyeb[0], (fluid::rgb3*)c->dyeb[1], x, y,
                                                  aspect, radius, c0, c1, c2, ga, gb),
                      "splat dye"));
        else
            CK(c->hip(STORE_CALL(c, launch_splat_dye(c->stream, dye_cols(c, c->dye_halo_x), DYE(c, 0), DYE(c, 1), x, y, aspect, radius, c0, c1, c2, ga, gb)),
                      "splat dye"));
        std::swap(c->dyeb[0], c->dyeb[1]);
        // the splat writes alpha = 1 into every texel of the context's window, ghost rows and columns included (fp16 storage rounds the decayed
        // alpha and does not track it).  On a stripe / tile set the value is the same on every rank because every rank gets every splat — the
        // header's contract for splats, writes and raw pointers on such sets (collective), asserted by fluid_group_step_n for in-process sets
        if (c->storage == FLUID_STORE_F32) {
            c->alpha_known = true;
            c->dye_alpha = 1.0f;
        }
    } else {
        return c->fail(FLUID_ERR_INVALID, "splat target must be velocity or dye");
    }
    return FLUID_OK;
}

int fluid_splat(fluid_ctx* c, float x, float y, float dx, float dy, float r, float g, float b, float aspect, float radius)
{
    CK(fluid_pass_splat(c, FLUID_VELOCITY, x, y, aspect, radius, dx, dy, 0.0f));
    return fluid_pass_splat(c, FLUID_DYE, x, y, aspect, radius, r, g, b);
}

int fluid_step_n(fluid_ctx* c, int n, float dt, const fluid_params* P)
{
    if (!c || !P) return FLUID_ERR_INVALID;
    if (n < 0) return c->fail(FLUID_ERR_INVALID, "negative step count");
    if (P->iterations < 0) return c->fail(FLUID_ERR_INVALID, "negative PRESSURE_ITERATIONS");
    HIPCK(c, hipSetDevice(c->device));
    if (c->desc.parts != 1 || c->desc.parts_x != 1) return stripe_step_n(c, n, dt, P);  // ghost-row exchanges over RCCL (fluid_stripes.cpp)
    // the curl field is a by-product that only a caller reads: of a call for n steps, the LAST step's (the header's contract)
    struct CurlGuard {   // whatever way the loop ends, the next call stores its curl again
        fluid_ctx* c;
        ~CurlGuard() { c->keep_curl = true; }
    } guard{ c };
    const bool skip = fluid_impl::skip_hidden_curl();
    fluid_impl::mark_step(c, 0);
    if (n == 0) return FLUID_OK;
    const bool chains = chain_applies(c, dt, P);
    const bool split = !chains && split_chain_applies(c, dt, P) && pending_buffers(c) == FLUID_OK;
    // did the launch that ended the previous call (or step) already run the next curl / vorticity / divergence?  Then adopt its buffers.
    auto adopt = [&]() {
        bool taken = false;
        if (c->pend_valid && (chains || split) && dt == c->pend_dt && P->curl == c->pend_curl_strength) {
            std::swap(c->vel[0], c->pend_vel);   // the velocity after vorticity confinement (the advected one moves to the spare buffer)
            std::swap(c->div, c->pend_div);
            std::swap(c->curl, c->pend_curl);
            taken = true;
        }
        c->pend_valid = false;
        return taken;
    };
    if (split) {   // dye grid != sim grid: every step works ahead for the next (split_chain_applies)
        for (int k = 0; k < n; k++) {
            const bool lead_k = !adopt();
            c->keep_curl = true;   // a lead launch writes this step's own curl; otherwise it came with the pending buffers
            CK(step_once(c, dt, P, lead_k, 3));
            fluid_impl::mark_step(c, k + 1);
        }
        c->curl_valid = true;
        return FLUID_OK;
    }
    bool lead = !adopt();
    const bool ahead = chains && run_ahead_enabled((long)c->sim_ncols * c->sim_rows) && pending_buffers(c) == FLUID_OK;   // end the call with the launch that works ahead
    if (chains && (n > 1 || ahead || !lead)) {
        for (int k = 0; k < n; k++) {
            c->keep_curl = !skip || n == 1;   // a lead launch of a call for ONE step writes the curl a caller reads; else the chain's last launch does
            const int chain = k < n - 2 ? 1 : (k == n - 2 ? 2 : (ahead ? 3 : 0));
            CK(step_once(c, dt, P, k == 0 && lead, chain));
            fluid_impl::mark_step(c, k + 1);
        }
        c->curl_valid = true;   // (the chain's last launch, or the pending buffers adopted, hold the last step's curl)
        return FLUID_OK;
    }
    for (int k = 0; k < n; k++) {
        // (curl_output off: not even the call's last step — the plain path only; the launches that work ahead or carry the next step's stencil
        // stages on small grids write their curl as before)
        c->keep_curl = c->curl_output && (k == n - 1 || !skip);
        CK(step_once(c, dt, P));
        fluid_impl::mark_step(c, k + 1);
    }
    return FLUID_OK;
}

int fluid_step(fluid_ctx* c, float dt, const fluid_params* P) { return fluid_step_n(c, 1, dt, P); }

int fluid_sync(fluid_ctx* c)
{
    if (!c) return FLUID_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->device));
    return ctx_sync(c);
}

int fluid_field_info_get(const fluid_ctx* c, int field, fluid_field_info* out)
{
    if (!c || !out) return FLUID_ERR_INVALID;
    FieldRef f;
    CK(field_ref(const_cast<fluid_ctx*>(c), field, &f, true));   // sizes and layout only: nothing is converted for a query
    *out = fluid_field_info{ f.win->W, f.win->H, f.nc, f.row0, f.rows, f.halo, f.col0, f.cols, f.halo_x, (int)f.esz, f.win->P, f.win->c0 };
    return FLUID_OK;
}

// The host side of read / write speaks fp32 whatever the storage.  With fp16 storage the owned rows go through a
// temporary fp32 copy on the device (widening is exact; narrowing rounds to nearest even, like any other store).
namespace {

struct HostBlock {
    FieldRef f;
    size_t line, rows_n;  // fp32 bytes per owned row segment, scalars in the owned rows over the whole pitch
    char* first_row;      // device address of the first owned row (array column 0)
};

// `peek`: a READ of a stripe / tile context's packed dye converts into the field's spare buffer and leaves the field as it is — on such a
// set the dye's format is part of the exchange's message layout and must not change because ONE rank was read (fluid_internal.h)
int host_block(fluid_ctx* c, int field, size_t bytes, const char* who, HostBlock* b, bool peek = false)
{
    if (peek && field == FLUID_DYE && c->dye_packed && (c->desc.parts > 1 || c->desc.parts_x > 1)) {
        CK(field_ref(c, field, &b->f, true));   // RGBA geometry
        CK(c->hip(fluid::launch_dye_unpack(c->stream, (const fluid::rgb3*)c->dyeb[0], (float4*)c->dyeb[1], cells(c->dye), c->dye_alpha), "unpack dye (read)"));
        b->f.ptr = c->dyeb[1];
    } else {
        CK(field_ref(c, field, &b->f));
    }
    const FieldRef& f = b->f;
    b->line = (size_t)f.cols * f.nc * sizeof(float);
    if (bytes != (size_t)f.rows * b->line) return c->fail(FLUID_ERR_INVALID, std::string(who) + ": byte count does not match the owned rows x columns (fp32)");
    b->rows_n = (size_t)f.rows * f.win->P * f.nc;
    b->first_row = (char*)f.ptr + (size_t)(f.row0 - f.win->g0) * f.win->P * f.texel();
    return FLUID_OK;
}

}  // namespace

int fluid_read_field(fluid_ctx* c, int field, float* host, size_t bytes)
{
    if (!c || !host) return FLUID_ERR_INVALID;
    HostBlock b;
    CK(host_block(c, field, bytes, "read_field", &b, true));
    HIPCK(c, hipSetDevice(c->device));
    const size_t pitch32 = (size_t)b.f.win->P * b.f.nc * sizeof(float);
    const size_t col = (size_t)(b.f.col0 - b.f.win->c0);  // array column of the first owned column
    if (c->storage == FLUID_STORE_F32) {
        HIPCK(c, hipMemcpy2DAsync(host, b.line, b.first_row + col * b.f.texel(), pitch32, b.line, b.f.rows, hipMemcpyDeviceToHost, c->stream));
        return ctx_sync(c);
    }
    float* tmp = nullptr;
    HIPCK(c, hipMalloc((void**)&tmp, b.rows_n * sizeof(float)));
    int rc = c->hip(launch_widen(c->stream, (const __half*)b.first_row, tmp, b.rows_n), "widen");
    if (!rc) rc = c->hip(hipMemcpy2DAsync(host, b.line, (char*)tmp + col * b.f.nc * sizeof(float), pitch32, b.line, b.f.rows, hipMemcpyDeviceToHost, c->stream), "copy");
    if (!rc) rc = ctx_sync(c);
    (void)hipFree(tmp);
    return rc;
}

int fluid_write_field(fluid_ctx* c, int field, const float* host, size_t bytes)
{
    if (!c || !host) return FLUID_ERR_INVALID;
    c->touched();
    HostBlock b;
    CK(host_block(c, field, bytes, "write_field", &b));
    HIPCK(c, hipSetDevice(c->device));
    const size_t pitch32 = (size_t)b.f.win->P * b.f.nc * sizeof(float);
    const size_t col = (size_t)(b.f.col0 - b.f.win->c0);  // array column of the first owned column
    if (field == FLUID_DYE && (c->desc.parts > 1 || c->desc.parts_x > 1)) c->alpha_known = false;   // the ghost texels keep what they held: no ONE alpha until the next splat
    if (field == FLUID_DYE && c->alpha_known) {   // does the caller's dye keep ONE alpha?
        const size_t n = (size_t)b.f.rows * b.f.cols;
        const float a0 = n ? host[3] : c->dye_alpha;
        bool uniform = true;
        for (size_t k = 0; k < n && uniform; k++) uniform = std::memcmp(&host[4 * k + 3], &a0, sizeof(float)) == 0;
        c->alpha_known = uniform;
        c->dye_alpha = a0;
    }
    if (c->storage == FLUID_STORE_F32) {
        HIPCK(c, hipMemcpy2DAsync(b.first_row + col * b.f.texel(), pitch32, host, b.line, b.line, b.f.rows, hipMemcpyHostToDevice, c->stream));
        return ctx_sync(c);
    }
    // the columns of these rows that this context does not own keep their values: widen, overlay the owned block, narrow
    float* tmp = nullptr;
    HIPCK(c, hipMalloc((void**)&tmp, b.rows_n * sizeof(float)));
    int rc = FLUID_OK;
    if (b.f.cols != b.f.win->P) rc = c->hip(launch_widen(c->stream, (const __half*)b.first_row, tmp, b.rows_n), "widen");
    if (!rc) rc = c->hip(hipMemcpy2DAsync((char*)tmp + col * b.f.nc * sizeof(float), pitch32, host, b.line, b.line, b.f.rows, hipMemcpyHostToDevice, c->stream), "copy");
    if (!rc) rc = c->hip(launch_narrow(c->stream, tmp, (__half*)b.first_row, b.rows_n), "narrow");
    if (!rc) rc = ctx_sync(c);
    (void)hipFree(tmp);
    return rc;
}

#define PASS_PROLOGUE()                      \
    if (!c) return FLUID_ERR_INVALID;        \
    HIPCK(c, hipSetDevice(c->device))

int fluid_pass_curl(fluid_ctx* c, int ext)
{
    if (c) c->touched();
    PASS_PROLOGUE();
    return pass_curl(c, ext);
}
int fluid_pass_vorticity(fluid_ctx* c, float curl, float dt, int ext)
{
    if (c) c->touched();
    PASS_PROLOGUE();
    return pass_vorticity(c, curl, dt, ext);
}
int fluid_pass_divergence(fluid_ctx* c, int ext)
{
    if (c) c->touched();
    PASS_PROLOGUE();
    return pass_divergence(c, ext);
}
int fluid_pass_curl_vorticity_divergence(fluid_ctx* c, float curl, float dt, int ext)
{
    if (c) c->touched();
    PASS_PROLOGUE();
    return pass_curl_vort_div(c, curl, dt, ext, nullptr);
}
int fluid_pass_clear(fluid_ctx* c, float value, int ext)
{
    if (c) c->touched();
    PASS_PROLOGUE();
    return pass_clear(c, value, ext);
}
int fluid_pass_jacobi(fluid_ctx* c, int iters, int ext_out)
{
    if (c) c->touched();
    PASS_PROLOGUE();
    return pass_jacobi(c, iters, ext_out, 1.0f, nullptr, nullptr, nullptr);
}
int fluid_pass_clear_jacobi(fluid_ctx* c, float value, int iters, int ext_out)
{
    if (c) c->touched();
    PASS_PROLOGUE();
    return pass_clear_jacobi(c, value, iters, ext_out, nullptr, nullptr);
}
int fluid_pass_gradsub(fluid_ctx* c, int ext)
{
    if (c) c->touched();
    PASS_PROLOGUE();
    return pass_gradsub(c, ext);
}
int fluid_pass_advect_velocity(fluid_ctx* c, float dt, float dissipation, int ext)
{
    if (c) c->touched();
    PASS_PROLOGUE();
    return pass_advect_velocity(c, dt, dissipation, ext);
}
int fluid_pass_advect_dye(fluid_ctx* c, float dt, float dissipation)
{
    if (c) c->touched();
    PASS_PROLOGUE();
    return pass_advect_dye(c, dt, dissipation);
}

int fluid_pass_advect(fluid_ctx* c, float dt, float velocity_dissipation, float density_dissipation)
{
    if (c) c->touched();
    PASS_PROLOGUE();
    return pass_advect(c, dt, velocity_dissipation, density_dissipation, nullptr);
}

static int halo_copy(fluid_ctx* c, int field, int side, int nrows, void* buf, bool pack)
{
    if (!c || !buf) return FLUID_ERR_INVALID;
    FieldRef f;
    CK(field_ref(c, field, &f));
    if (nrows < 1 || nrows > f.halo || nrows > f.rows) return c->fail(FLUID_ERR_INVALID, "halo rows out of range");
    if (side != 0 && side != 1) return c->fail(FLUID_ERR_INVALID, "side must be 0 (bottom) or 1 (top)");
    HIPCK(c, hipSetDevice(c->device));
    const size_t row_bytes = (size_t)f.win->P * f.texel();
    int first;  // first array row of the block
    if (pack) first = side == 0 ? f.halo : f.halo + f.rows - nrows;
    else first = side == 0 ? f.halo - nrows : f.halo + f.rows;
    char* p = (char*)f.ptr + (size_t)first * row_bytes;
    if (pack) HIPCK(c, hipMemcpyAsync(buf, p, nrows * row_bytes, hipMemcpyDeviceToDevice, c->stream));
    else HIPCK(c, hipMemcpyAsync(p, buf, nrows * row_bytes, hipMemcpyDeviceToDevice, c->stream));
    return FLUID_OK;
}

int fluid_halo_pack(fluid_ctx* c, int field, int side, int nrows, void* dev_buf) { return halo_copy(c, field, side, nrows, dev_buf, true); }

int fluid_halo_unpack(fluid_ctx* c, int field, int side, int nrows, const void* dev_buf)
{
    if (c) c->touched();
    if (c && field == FLUID_DYE) c->alpha_known = false;   // ghost texels from a buffer of the caller's: whatever alpha they carry
    return halo_copy(c, field, side, nrows, const_cast<void*>(dev_buf), false);
}

int fluid_field_device_ptr(fluid_ctx* c, int field, void** dev_ptr)
{
    if (!c || !dev_ptr) return FLUID_ERR_INVALID;
    c->touched();
    HIPCK(c, hipSetDevice(c->device));
    const bool was_packed = c->dye_packed;
    FieldRef f;
    CK(field_ref(c, field, &f));
    // The header's ordering rule (1): work THIS call had to enqueue is waited for here, so that `fluid_sync(); fluid_field_device_ptr();`
    // hands out finished memory.  Round 4 returned while k_dye_unpack was still writing the buffer behind the pointer, on a non-blocking
    // stream no other stream is ordered against: bench.py's torch.equal read it half-written (BENCH_r04.json, profiles/r05/device_view_race.txt).
    if (was_packed && !c->dye_packed) CK(ctx_sync(c));
    else CK(chain_check(c));   // (a pressure loop that gave up in a call the caller already synchronised with: no pointer to its fields)
    if (field == FLUID_DYE) c->alpha_known = false;   // a raw pointer: whatever gets written through it, the context does not see (until the next splat)
    *dev_ptr = f.ptr;
    return FLUID_OK;
}

// the two ordering calls of the zero-copy contract (include/fluid_hip.h, fluid_field_device_ptr): one event, recorded on the producing
// side and waited for by the consuming stream, all on the device
static int order_streams(fluid_ctx* c, hipStream_t from, hipStream_t to)
{
    HIPCK(c, hipSetDevice(c->device));
    if (from == to) return FLUID_OK;   // one stream: already in order
    if (!c->ev_order) HIPCK(c, hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
    HIPCK(c, hipEventRecord(c->ev_order, from));
    HIPCK(c, hipStreamWaitEvent(to, c->ev_order, 0));
    return FLUID_OK;
}

int fluid_stream_wait_context(fluid_ctx* c, void* hip_stream)
{
    if (!c) return FLUID_ERR_INVALID;
    // What this call can know without waiting: a chained pressure loop that gave up in work the device has ALREADY run is an error here too —
    // the consumer must not be ordered behind fields that are known to be wrong.  (Work still in flight cannot have failed yet: the caller
    // that wants the verdict on it synchronises — fluid_sync — as the header says.)
    CK(chain_check(c));
    return order_streams(c, c->stream, (hipStream_t)hip_stream);
}

int fluid_context_wait_stream(fluid_ctx* c, void* hip_stream)
{
    if (!c) return FLUID_ERR_INVALID;
    return order_streams(c, (hipStream_t)hip_stream, c->stream);
}

int fluid_halo_check(fluid_ctx* c)
{
    if (!c) return FLUID_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->device));
    unsigned int m = 0;
    HIPCK(c, hipMemcpyAsync(&m, c->miss, sizeof(m), hipMemcpyDeviceToHost, c->stream));
    CK(ctx_sync(c));
    if (m) {
        HIPCK(c, hipMemsetAsync(c->miss, 0, sizeof(unsigned int), c->stream));
        char msg[128];
        std::snprintf(msg, sizeof msg, "%u advection taps fell outside the stripe's ghost rows (raise halo)", m);
        return c->fail(FLUID_ERR_HALO, msg);
    }
    return FLUID_OK;
}

int fluid_set_curl_output(fluid_ctx* c, int enabled)
{
    if (!c) return FLUID_ERR_INVALID;
    c->curl_output = enabled != 0;
    return FLUID_OK;
}

int fluid_set_timing(fluid_ctx* c, int enabled)
{
    if (!c) return FLUID_ERR_INVALID;
    c->timing = enabled != 0;
    std::fill(std::begin(c->acc_ms), std::end(c->acc_ms), 0.0);
    c->acc_total = 0;
    c->acc_steps = c->acc_jacobi_launches = c->acc_folded_launches = 0;
    return FLUID_OK;
}

int fluid_schedule_info_get(fluid_ctx* c, int n_steps, float dt, const fluid_params* P, fluid_schedule_info* out)
{
    if (!c || !P || !out || n_steps < 0) return FLUID_ERR_INVALID;
    *out = fluid_schedule_info{};
    const bool whole = c->desc.parts == 1 && c->desc.parts_x == 1;
    const long owned = (long)c->sim_ncols * c->sim_rows;
    out->fused = c->desc.schedule == FLUID_SCHED_FUSED;
    const bool tb = fluid_impl::jacobi_tb_applies(c) && P->iterations > 0;
    out->jacobi_shape = tb ? fluid::jacobi_tb_pick(owned) : -1;
    const int depth = tb ? fluid::jacobi_tb_depth(out->jacobi_shape) : 1;
    out->jacobi_launches = tb ? (P->iterations + depth - 1) / depth : P->iterations;
    {   // ... which are ONE launch of chained blocks where that schedule applies (pass_jacobi)
        int ga, gb;
        row_range(c->sim, c->sim_row0, c->sim_rows, 0, ga, gb);
        // A stripe / tile rank chains the launches it has left behind its cut ones the same way (pass_jacobi: its blocks' row / column ranges shrink
        // from launch to launch; the rule is asked about the owned texels here, about the first uncut launch's there — the same answer but on a rank
        // within a ghost zone of the rule's limits): reported for them too (it said false there while the rank's loop did chain).
        // Not after a chained launch of this context has given up: chain_broken
        out->jacobi_chained = tb && !c->chain_broken && c->storage == FLUID_STORE_F32 && out->jacobi_shape == 0 && !fluid_impl::gradsub_fold_enabled(owned) &&
                              out->jacobi_launches >= 2 && fluid::jacobi_chain_applies(sim_cols(c, 0), ga, gb, P->iterations);
    }
    out->gradsub_folded = tb && fluid::jacobi_tb_has_gradsub(out->jacobi_shape) && fluid_impl::gradsub_fold_enabled(owned);
    const bool split = whole && n_steps > 0 && !chain_applies(c, dt, P) && split_chain_applies(c, dt, P);   // dye grid != sim grid
    const bool chains = whole && n_steps > 0 && (split || chain_applies(c, dt, P));
    out->pending_adopted = chains && c->pend_valid && dt == c->pend_dt && P->curl == c->pend_curl_strength;
    out->runs_ahead = chains && (split || run_ahead_enabled((long)c->sim_ncols * c->sim_rows));
    const bool chain = chains && (n_steps > 1 || out->runs_ahead || out->pending_adopted);
    out->chained = chain ? n_steps - 1 + out->runs_ahead : 0;
    const bool fused_cvd = fluid_impl::fused_cvd_applies(c);
    out->dye_packed = dye_pack_applies(c) && (c->dye_packed || c->pack_holdoff == 0);
    // how many times the call stores a curl field.  Chained / plain steps: hidden curls are skipped — the call's last step's is stored, plus
    // the one the closing launch works ahead.  Split path (dye grid != sim grid; step_once chain 3): EVERY step's closing launch stores the
    // next step's curl into the pending buffer, and a lead launch (nothing adopted) stores the first step's own.
    if (split) out->curl_stores = n_steps + (out->pending_adopted ? 0 : 1);
    else out->curl_stores = (fused_cvd && fluid_impl::skip_hidden_curl() && n_steps > 0) ? 1 + (out->runs_ahead ? 1 : 0) : n_steps + (out->runs_ahead ? 1 : 0);
    if (!c->curl_output && fused_cvd && !split && !chains) out->curl_stores = 0;   // fluid_set_curl_output(ctx, 0): the plain path stores none
    if (whole) {
        const int cvd = fused_cvd ? 1 : 3, clear = tb ? 0 : 1, gs = out->gradsub_folded ? 0 : 1;
        const int adv = fluid_impl::fused_advect_applies(c) ? 1 : 2;
        const int per_step = cvd + clear + (out->jacobi_chained ? 1 : out->jacobi_launches) + gs + adv;
        // a chained step has no curl launch of its own: only the call's first step does, unless the previous call already ran it ahead
        out->launches = chain ? n_steps * (per_step - 1) + (out->pending_adopted ? 0 : 1) : n_steps * per_step;
    }
    return FLUID_OK;
}

int fluid_set_step_marks(fluid_ctx* c, int capacity)
{
    if (!c || capacity < 0 || capacity > 4096) return c ? c->fail(FLUID_ERR_INVALID, "step marks: capacity 0 .. 4096") : FLUID_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->device));
    for (auto& e : c->marks) (void)hipEventDestroy(e);
    c->marks.clear();
    c->marks_used = 0;
    if (capacity > 0) {
        c->marks.resize((size_t)capacity + 1, nullptr);
        // timing-only events: no system-scope fence when one is recorded (hipEventDisableSystemFence — "for events that are only being
        // used to measure timing").  With the default flags every mark cost the stream 6 us of cache writeback and idle: the driver's 20
        // marked steps read 1.2 % slower than the same steps unmarked (profiles/r04/step_marks_cost.txt).
        for (auto& e : c->marks) HIPCK(c, hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
    }
    return FLUID_OK;
}

int fluid_get_step_marks(fluid_ctx* c, float* ms, int capacity, int* n_steps)
{
    if (!c || !n_steps || capacity < 0 || (capacity > 0 && !ms)) return FLUID_ERR_INVALID;
    const int n = c->marks_used > 0 ? c->marks_used - 1 : 0;
    *n_steps = n;
    if (n == 0) return FLUID_OK;
    HIPCK(c, hipEventSynchronize(c->marks[n]));
    for (int k = 0; k < n && k < capacity; k++) HIPCK(c, hipEventElapsedTime(&ms[k], c->marks[k], c->marks[k + 1]));
    return FLUID_OK;
}

int fluid_get_timings(fluid_ctx* c, fluid_timings* out)
{
    if (!c || !out) return FLUID_ERR_INVALID;
    out->curl_ms = (float)c->acc_ms[P_CURL];
    out->vorticity_ms = (float)c->acc_ms[P_VORT];
    out->divergence_ms = (float)c->acc_ms[P_DIV];
    out->clear_ms = (float)c->acc_ms[P_CLEAR];
    out->jacobi_ms = (float)c->acc_ms[P_JACOBI];
    out->gradsub_ms = (float)c->acc_ms[P_GRADSUB];
    out->advect_velocity_ms = (float)c->acc_ms[P_ADVV];
    out->advect_dye_ms = (float)c->acc_ms[P_ADVD];
    out->total_ms = (float)c->acc_total;
    out->jacobi_launches = c->acc_jacobi_launches;
    out->steps = c->acc_steps;
    out->folded_launches = c->acc_folded_launches;
    return FLUID_OK;
}

}  // extern "C"

// fluid_kernels.h — launch interface between the solver core (fluid_solver.cpp) and the
// gfx950 kernels (fluid_kernels.hip).  Internal; the public boundary is include/fluid_hip.h.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdlib>

namespace fluid {

// The FLUID_* tuning knobs (tile shapes, folds, chains, rows per thread: every A/B of the rounds' profiles/) exist in the lab build only —
// `make PROBES=1` -> libfluid_hip_probes.so, -DFLUID_PROBES — together with the kernel shapes they select.  The product library reads none of
// them: what it launches is decided by the grid alone (fluid_schedule_info_get says what).
// ONE threshold for everything the library does differently on small grids (owned texels of a context below it): the 40-row Jacobi tile
// instead of the 80-row one, the gradient subtract folded into the last Jacobi launch, fluid_step_n / fluid_step chaining the advection with
// the next step's curl / vorticity / divergence.  Below it a step is a chain of latency-bound launches; above it, of bandwidth-bound ones
// (profiles/r03/jacobi_shapes_small_grids.txt, gradsub_fold_ab.txt, advect_cvd_chain.txt).
constexpr long kSmallGridTexels = 3072l * 3072l;

// Working ahead (the launch that ends a call also runs the next call's curl / vorticity / divergence into pending buffers) keeps a third
// velocity buffer and second divergence / curl buffers alive: 80 B per texel instead of 64.  Up to 1536^2 that stays inside the 256 MB
// Infinity Cache (512^2: 0.98 of fluid_step_n per frame instead of 0.82, 1024^2: 0.99 instead of 0.90); at 2048^2 it falls out of it and the
// per-frame path gets SLOWER (0.83 against 0.92: profiles/r04/single_step_path.txt).
constexpr long kRunAheadTexels = 1536l * 1536l + 1;

inline const char* lab_env(const char* name)
{
#ifdef FLUID_PROBES
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// A window onto one field: the device array holds `rows` rows x P columns of a field whose global size is
// W x H; array row r is global row g0 + r (g0 < 0 is allowed: ghost rows below the domain are
// allocated but never addressed), array column k is global column c0 + k.  Whole domain: g0 = 0, rows = H, c0 = 0.
struct Win {
    int W, H;
    int g0;
    int rows;
    // Columns: c0 (a multiple of 4) is the global column of array column 0 and P the pitch — columns per array row, a multiple
    // of 4 so that every row starts float4-aligned whatever W is.  The array covers global columns [c0, c0 + P); columns
    // beyond W - 1 (W % 4 != 0) or beyond what a tile needs are padding: loaded by the four-texel lanes, never meaningful.
    // Whole width: c0 = 0, P = W rounded up to 4.  A 2-D tile: its owned columns + ghost columns only.
    int c0, P;
    // Global rows [v0, v1) hold data a GATHER (bilinear fetch) may use: the advection / resample kernels count every
    // tap outside as a miss.  Normally the whole window; the stripe driver narrows it to the rows that are fresh at
    // that moment (owned rows while an exchange is in flight, owned + exchanged rows afterwards).
    int v0, v1;
    // A launch writes columns [x0, x1) only, and gathers may use columns [u0, u1) (2-D tile decomposition; the whole
    // width otherwise: 0, W).
    int x0, x1;
    int u0, u1;
};
inline int pitch_of(int cols) { return (cols + 3) & ~3; }
inline double udiv_recip(float d) { return 1.0 / (double)d; }
inline bool udiv_decay_ok(float d) { return d >= 1.0f && d < 2.0f; }
inline Win make_win(int W, int H, int g0, int rows) { return Win{ W, H, g0, rows, 0, pitch_of(W), g0, g0 + rows, 0, W, 0, W }; }
// a window that holds global columns [ca, cb) only (ca a multiple of 4): the arrays of a 2-D tile
inline Win make_win_cols(int W, int H, int g0, int rows, int ca, int cb)
{
    return Win{ W, H, g0, rows, ca, pitch_of(cb - ca), g0, g0 + rows, 0, W, 0, W };
}

// Storage of the fields (fluid_desc.storage): fp32 texels, or half texels (FLUID_STORE_F16: what the reference's
// half-float textures hold on a real GPU, script.js:138, 145-147).  Arithmetic is fp32 either way; a store to a half
// field rounds to nearest even.
struct alignas(8) half4 {
    __half2 lo, hi;
};
// The dye field PACKED: three floats per texel.  The dye's alpha channel is spatially uniform by construction — splats write 1
// (script.js:544: vec4(base + splat, 1.0)), the advection's bilinear mix of four equal values is that value, and the decay divides every
// texel by the same number — so a context that knows the value (fluid_ctx::alpha_known) keeps it as ONE scalar and the big fused
// advection moves 40 instead of 48 B/texel.  Everything else sees RGBA: fluid_solver.cpp unpacks on demand (ensure_rgba).
struct rgb3 {
    float r, g, b;
};

struct StoreF32 {
    using T1 = float;
    using T2 = float2;
    using T4 = float4;
};
struct StoreF16 {
    using T1 = __half;
    using T2 = __half2;
    using T4 = half4;
};

// All launchers enqueue on `s` and return hipGetLastError().  Row ranges [ga, gb) are GLOBAL rows.
hipError_t launch_curl(hipStream_t s, Win w, const float2* vel, float* curl, int ga, int gb);
hipError_t launch_vorticity(hipStream_t s, Win w, const float2* vel, const float* curl, float2* vel_out,
                            float curl_strength, float dt, int ga, int gb);
hipError_t launch_divergence(hipStream_t s, Win w, const float2* vel, float* div, int ga, int gb);
hipError_t launch_clear(hipStream_t s, Win w, const float* p, float* p_out, float value, int ga, int gb);
hipError_t launch_jacobi(hipStream_t s, Win w, const float* p, const float* div, float* p_out, int ga, int gb);
hipError_t launch_gradsub(hipStream_t s, Win w, const float* p, const float2* vel, float2* vel_out, int ga, int gb);
// the same pass, four texels per lane
hipError_t launch_gradsub4(hipStream_t s, Win w, const float* p, const float2* vel, float2* vel_out, int ga, int gb);
hipError_t launch_advect_velocity(hipStream_t s, Win w, const float2* vel, float2* out, float dt, float dissipation,
                                  int ga, int gb, unsigned int* miss);
hipError_t launch_advect_dye(hipStream_t s, Win vw, const float2* vel, Win dw, const float4* dye, float4* out,
                             float dt, float dissipation, int ga, int gb, unsigned int* miss);
// K7a + K7b in one kernel; only when the dye grid equals the sim grid (same window)
hipError_t launch_advect_both(hipStream_t s, Win w, const float2* vel, float2* vel_out, const float4* dye, float4* dye_out,
                              float dt, float vel_dissipation, float dye_dissipation, int ga, int gb, unsigned int* miss);
// packed dye (rgb3): the fused advection where k_advect_both_fast applies (hipErrorNotReady otherwise: the caller unpacks and takes the
// RGBA path), the dye splat, and the two conversions over a whole array of n texels
hipError_t launch_advect_both_rgb(hipStream_t s, Win w, const float2* vel, float2* vel_out, const rgb3* dye, rgb3* dye_out, float dt,
                                  float vel_dissipation, float dye_dissipation, int ga, int gb, unsigned int* miss);
// whether the packed-dye kernels apply to these windows and decays (checked BEFORE a field is packed for them)
bool advect_rgb_supported(Win vw, Win dw, float dt, float vel_dissipation, float dye_dissipation);
hipError_t launch_advect_dye_rgb(hipStream_t s, Win vw, const float2* vel, Win dw, const rgb3* dye, rgb3* out, float dt, float dissipation, int ga,
                                 int gb, unsigned int* miss);
hipError_t launch_splat_dye_rgb(hipStream_t s, Win w, const rgb3* base, rgb3* out, float x, float y, float aspect, float radius, float c0,
                                float c1, float c2, int ga, int gb);
hipError_t launch_dye_pack(hipStream_t s, const float4* rgba, rgb3* rgb, size_t n);
hipError_t launch_dye_unpack(hipStream_t s, const rgb3* rgb, float4* rgba, size_t n, float alpha);
hipError_t launch_splat_velocity(hipStream_t s, Win w, const float2* base, float2* out, float x, float y, float aspect,
                                 float radius, float c0, float c1, int ga, int gb);
hipError_t launch_splat_dye(hipStream_t s, Win w, const float4* base, float4* out, float x, float y, float aspect,
                            float radius, float c0, float c1, float c2, int ga, int gb);
hipError_t launch_resample(hipStream_t s, Win sw, const float* src, int nc, Win dw, float* dst);
hipError_t launch_fill(hipStream_t s, float* dst, size_t n_vec, int nc, float v0, float v1, float v2, float v3);

// Strided block copies for the ghost-column exchange of 2-D tiles (fluid_stripes.cpp): up to 16 rectangles — every field and
// direction of one exchange (four sides + four corners, two fields) — packed into (or unpacked from) contiguous staging in ONE launch
// on the comm stream.  `unit` = bytes a thread moves at a time: the widest of 16 / 8 / 4 / 2 that both addresses, both pitches and the
// line are multiples of (copy_rect_of picks it; a velocity block of 56 columns moves as 16-byte pieces, a pressure block of 51 as 4-byte ones).
struct CopyRect {
    const char* src;
    char* dst;
    size_t spitch, dpitch;  // bytes between rows
    unsigned line_units, nrows, unit;
};
inline CopyRect copy_rect_of(const void* src, void* dst, size_t spitch, size_t dpitch, size_t line_bytes, int nrows)
{
    const size_t all = (size_t)src | (size_t)dst | spitch | dpitch | line_bytes;
    const unsigned unit = (all & 15) == 0 ? 16 : (all & 7) == 0 ? 8 : (all & 3) == 0 ? 4 : 2;
    return CopyRect{ (const char*)src, (char*)dst, spitch, dpitch, (unsigned)(line_bytes / unit), (unsigned)nrows, unit };
}
struct CopyRects {
    CopyRect r[16];
    int n;
};
hipError_t launch_copy_rects(hipStream_t s, const CopyRects& R);

// Several bands of a single-kernel pass group in ONE launch: the strips around the interior of a 2-D tile (fluid_stripes.cpp).  Each
// rectangle = columns [xa, xb) x rows [ga, gb) of the window (xa, xb whole float4 groups for the register-tile kernel); empty ones are
// skipped.  Same kernels bodies, same bits as one launch per rectangle.
struct BandRect { int xa, xb, ga, gb; };
struct BandRects { BandRect r[4]; int n; };
hipError_t launch_advect_both_rects(hipStream_t s, Win w, const float2* vel, float2* vel_out, const float4* dye, float4* dye_out, float dt,
                                    float vel_dissipation, float dye_dissipation, const BandRects& B, unsigned int* miss);
hipError_t launch_advect_both_rects(hipStream_t s, Win w, const __half2* vel, __half2* vel_out, const half4* dye, half4* dye_out, float dt,
                                    float vel_dissipation, float dye_dissipation, const BandRects& B, unsigned int* miss);
hipError_t launch_advect_both_rects_rgb(hipStream_t s, Win w, const float2* vel, float2* vel_out, const rgb3* dye, rgb3* dye_out, float dt,
                                        float vel_dissipation, float dye_dissipation, const BandRects& B, unsigned int* miss);   // packed dye (hipErrorNotReady: see launch_advect_both_rgb)
hipError_t launch_curl_vort_div_rects(hipStream_t s, Win w, const float2* vel, float* curl, float2* vel_out, float* div, float curl_strength,
                                      float dt, const BandRects& B);
// launch_jacobi_tb over several rectangles in one launch (fp32 fields, iters <= 10): the frame of a block's first launch around the interior
// that ran while an exchange was in flight.  Bitwise equal to launch_jacobi_tb over the same texels, whatever shape that one takes.
hipError_t launch_jacobi_tb_rects(hipStream_t s, Win w, const float* p, const float* div, float* p_out, float pscale, int iters, const BandRects& B);
hipError_t launch_curl_vort_div_rects(hipStream_t s, Win w, const __half2* vel, __half* curl, __half2* vel_out, __half* div, float curl_strength,
                                      float dt, const BandRects& B);

// Fused curl -> vorticity -> divergence (K1+K2+K3): reads velocity rows [ga-3, gb+3) (clamped), writes curl,
// the confined velocity and its divergence for rows [ga, gb).  Any width (the pitch keeps rows float4-aligned).  Bitwise equal to the three
// single-pass kernels run in turn.
bool fused_supported(Win w);
// K7a + K7b of one step and K1 + K2 + K3 of the next in one launch (fluid_step_n, n > 1): `vel` is the projected velocity, `vel_out` receives
// the next step's velocity after vorticity confinement; `curl` may be null (the field is then left as it is: only a chain's last launch
// writes it).  fp32, dye grid = sim grid, whole domain.
bool advect_cvd_supported(Win w, float dt, float vel_dissipation, float dye_dissipation);
// ... with `dye` == null (dye grid != sim grid): the velocity alone is advected and ALWAYS stored to `vel_adv` (the dye pass that follows
// samples it), the next step's curl / vorticity / divergence go to vel_out / curl / div as before
bool advect_cvd_velocity_supported(Win w, float dt, float vel_dissipation);
// `vel_adv` non-null (then `curl` must be too): the launch that ENDS a call — it also stores the advected velocity there (what a caller reads
// as the velocity field), and vel_out / curl / div are the context's pending buffers, which the next call adopts or drops.
hipError_t launch_advect_cvd(hipStream_t s, Win w, const float2* vel, float2* vel_out, const float4* dye, float4* dye_out, float* curl,
                             float* div, float2* vel_adv, float dt, float vel_dissipation, float dye_dissipation, float curl_strength, int ga,
                             int gb);
hipError_t launch_curl_vort_div(hipStream_t s, Win w, const float2* vel, float* curl, float2* vel_out, float* div,
                                float curl_strength, float dt, int ga, int gb);

// Temporally blocked Jacobi: `iters` (<= jacobi_tb_depth(shape)) iterations in one launch, every input
// value scaled by `pscale` on load (pscale = config.PRESSURE folds the clear pass, 1.0f otherwise).
// Reads p rows [ga - iters, gb + iters) (clamped to the domain), writes p_out rows [ga, gb).
// Any width.  Bitwise equal to `iters` launches of launch_jacobi.  `shape`: the register tile's geometry, picked ONCE per pass from the
// number of owned texels (jacobi_tb_pick: small grids take smaller, deeper tiles; FLUID_TB_VARIANT forces one).
int jacobi_tb_pick(long owned_texels);
int jacobi_tb_depth(int shape);
int jacobi_tb_apron_cols(int shape);   // columns of apron a tile of that shape loads on each side
bool jacobi_tb_has_gradsub(int shape);
bool jacobi_tb_supported(Win w);
// The pressure loop as ONE launch of chained blocks of ten iterations (fluid_kernels.hip, k_jacobi_tb_chain; fp32 fields, the 80-row tile):
// where jacobi_chain_applies() says so — grids whose pressure set fits the Infinity Cache — it is what pass_jacobi runs.  `flags`: jacobi_chain_flag_bytes() of device memory
// (zeroed by the launcher); `err`: two words the device can write and the HOST can read (mapped host memory): err[0] != 0 = a workgroup gave up
// waiting for a tile (the results of that call are not valid).  pa holds the input; the result is in pb when the number of blocks is odd.
struct ChainEpoch {   // per context: the shape of the last chained call and how many calls of that shape have counted the counters up
    unsigned int signature = 0xffffffffu, calls = 0;
    // the persistent form (k_jacobi_pchain, fluid_pchain.h): the shape of its last call and which half of the state words the next call counts in
    unsigned int psig[2] = { 0xffffffffu, 0u };
    int bank = 0;
};
int jacobi_chain_max_blocks();   // blocks of <= 10 iterations one chained launch can hold
bool jacobi_chain_applies(const Win& w, int ga, int gb, int iters);
size_t jacobi_chain_flag_bytes();
hipError_t launch_jacobi_tb_chain(hipStream_t s, Win w, float* pa, float* pb, const float* div, float pscale, int iters, int ga, int gb,
                                  unsigned int* flags, unsigned int* err, int* blocks, bool* result_in_b, ChainEpoch* ep = nullptr);
// ... with a row range per block (a stripe rank's launches behind its cut ones: each recomputes fewer ghost rows); <= 8 blocks of <= 10 iterations
hipError_t launch_jacobi_tb_chain_ranges(hipStream_t s, Win w, float* pa, float* pb, const float* div, float pscale, int nblocks, const int* iters,
                                         const int* ga, const int* gb, const int* xa, const int* xb, unsigned int* flags, unsigned int* err,
                                         ChainEpoch* ep = nullptr, const float2* vel = nullptr, float2* vel_out = nullptr);   // (vel_out: lab — K6 as one more block; ranges entry [nblocks])
hipError_t launch_jacobi_tb(hipStream_t s, Win w, const float* p, const float* div, float* p_out, float pscale,
                            int iters, int ga, int gb, int shape);
// The same launch with K6 (gradient subtract) folded in — for the LAST block of a step's loop: runs `iters` iterations, writes p_out rows
// [ga, gb) AND vel_out = vel - grad(p_out) for the same texels (the tile carries one more apron ring, so the pressure neighbours of every
// stored texel are exact in registers).  Reads p rows [ga - iters - 1, gb + iters + 1).  Bitwise equal to launch_jacobi_tb + launch_gradsub.
// Only shapes with jacobi_tb_has_gradsub().
hipError_t launch_jacobi_tb_gradsub(hipStream_t s, Win w, const float* p, const float* div, float* p_out, const float2* vel, float2* vel_out,
                                    float pscale, int iters, int ga, int gb, int shape);
hipError_t launch_jacobi_tb_gradsub(hipStream_t s, Win w, const __half* p, const __half* div, __half* p_out, const __half2* vel,
                                    __half2* vel_out, float pscale, int iters, int ga, int gb, int shape);
// the fused kernels on fp16-storage fields: every intermediate the reference would have rendered to a half-float texture
// between two of the fused passes (curl, the confined velocity, the advected velocity) is rounded to fp16 in registers,
// so each is bitwise equal to its single-pass half kernels run in turn
hipError_t launch_curl_vort_div(hipStream_t s, Win w, const __half2* vel, __half* curl, __half2* vel_out, __half* div, float curl_strength,
                                float dt, int ga, int gb);
hipError_t launch_gradsub4(hipStream_t s, Win w, const __half* p, const __half2* vel, __half2* vel_out, int ga, int gb);
hipError_t launch_advect_both(hipStream_t s, Win w, const __half2* vel, __half2* vel_out, const half4* dye, half4* dye_out, float dt,
                              float vel_dissipation, float dye_dissipation, int ga, int gb, unsigned int* miss);
// the same on fp16-storage fields: the clear (pscale) and every iteration round their output to fp16, so the launch is
// bitwise equal to `iters` launches of the half launch_jacobi
hipError_t launch_jacobi_tb(hipStream_t s, Win w, const __half* p, const __half* div, __half* p_out, float pscale, int iters, int ga,
                            int gb, int shape);


// ---- the same passes on fp16-storage fields (fluid_kernels_f16.hip): one kernel per reference pass ----
hipError_t launch_curl(hipStream_t s, Win w, const __half2* vel, __half* curl, int ga, int gb);
hipError_t launch_vorticity(hipStream_t s, Win w, const __half2* vel, const __half* curl, __half2* vel_out, float curl_strength, float dt,
                            int ga, int gb);
hipError_t launch_divergence(hipStream_t s, Win w, const __half2* vel, __half* div, int ga, int gb);
hipError_t launch_clear(hipStream_t s, Win w, const __half* p, __half* p_out, float value, int ga, int gb);
hipError_t launch_jacobi(hipStream_t s, Win w, const __half* p, const __half* div, __half* p_out, int ga, int gb);
hipError_t launch_gradsub(hipStream_t s, Win w, const __half* p, const __half2* vel, __half2* vel_out, int ga, int gb);
hipError_t launch_advect_velocity(hipStream_t s, Win w, const __half2* vel, __half2* out, float dt, float dissipation, int ga, int gb,
                                  unsigned int* miss);
hipError_t launch_advect_dye(hipStream_t s, Win vw, const __half2* vel, Win dw, const half4* dye, half4* out, float dt, float dissipation,
                             int ga, int gb, unsigned int* miss);
hipError_t launch_splat_velocity(hipStream_t s, Win w, const __half2* base, __half2* out, float x, float y, float aspect, float radius,
                                 float c0, float c1, int ga, int gb);
hipError_t launch_splat_dye(hipStream_t s, Win w, const half4* base, half4* out, float x, float y, float aspect, float radius, float c0,
                            float c1, float c2, int ga, int gb);
hipError_t launch_resample(hipStream_t s, Win sw, const __half* src, int nc, Win dw, __half* dst);
hipError_t launch_fill(hipStream_t s, __half* dst, size_t n_vec, int nc, float v0, float v1, float v2, float v3);
// host boundary (fluid_read_field / fluid_write_field speak fp32): n scalars, exact widening / round-to-nearest-even narrowing
hipError_t launch_widen(hipStream_t s, const __half* src, float* dst, size_t n);
hipError_t launch_narrow(hipStream_t s, const float* src, __half* dst, size_t n);

}  // namespace fluid

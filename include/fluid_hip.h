/*
 * fluid_hip.h — C ABI of libfluid_hip.so: the MI355X (gfx950) stable-fluids hot path.
 *
 * This is the drop-in boundary for the simulation path of PavelDoGreat/WebGL-Fluid-Simulation.
 * The reference has no FFI of its own (its passes are WebGL draw calls issued from one
 * script), so each entry point below names the reference code it replaces; a binding for
 * the reference's host language (JavaScript, N-API) is in webgl-fluid-simulation_amd/addon/
 * and INTEGRATION.md shows how script.js would call it.
 *
 * Conventions
 *   - Every function returns 0 (FLUID_OK) or a negative fluid_status; fluid_last_error()
 *     gives text.  The reference has no error convention (it only console.trace()s GL
 *     failures, script.js:402-403, 425-426), so nothing is mirrored there.
 *   - A context owns all device memory.  Host pointers are caller-owned.  Calls are
 *     asynchronous on the context's HIP stream except fluid_read_field / fluid_sync.
 *     Not thread-safe per context (the reference is single-threaded, script.js:1176-1186).
 *   - Fields are row-major, row 0 = BOTTOM (uv.y smallest), like gl.readPixels
 *     (script.js:301-307).  velocity = 2 floats/texel (RG), dye = 4 (RGBA), pressure /
 *     divergence / curl = 1.  All fp32 (what the SwiftShader reference stores).
 *   - Scalars arrive as float: the caller rounds its doubles exactly like gl.uniform1f does.
 *
 * Domain decomposition (multi-GPU): a context may own `part` of `parts` equal row stripes of
 * the global grid plus `halo` ghost rows on each side — and, for a 2-D tile set, column tile
 * `part_x` of `parts_x` of that stripe plus `halo` ghost columns.  part = 0, parts = 1, halo = 0
 * is the whole domain.  The per-pass entry points take `ext`: how many ghost rows (and columns)
 * beyond the owned ones are (re)computed, so a rank can trade halo exchanges for redundant work.
 */
#ifndef FLUID_HIP_H
#define FLUID_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLUID_ABI_VERSION 10

typedef enum fluid_status {
    FLUID_OK = 0,
    FLUID_ERR_INVALID = -1,       /* bad argument / size / state                  */
    FLUID_ERR_HIP = -2,           /* a HIP runtime call failed (see last_error)   */
    FLUID_ERR_NO_DEVICE = -3,     /* no gfx950-capable device visible             */
    FLUID_ERR_OOM = -4,           /* device allocation failed                     */
    FLUID_ERR_HALO = -5,          /* an advection back-trace left the ghost rows  */
    FLUID_ERR_UNSUPPORTED = -6,
    FLUID_ERR_COMM = -7           /* RCCL missing / failed, or a stripe without communicator */
} fluid_status;

/* the reference's five simulation fields: `let dye; let velocity; ...` script.js:950-954 */
typedef enum fluid_field {
    FLUID_VELOCITY = 0,   /* velocity.read   RG   */
    FLUID_PRESSURE = 1,   /* pressure.read   R    */
    FLUID_DIVERGENCE = 2, /* divergence      R    */
    FLUID_CURL = 3,       /* curl            R    */
    FLUID_DYE = 4,        /* dye.read        RGBA */
    FLUID_FIELD_COUNT = 5
} fluid_field;

/* how fluid_step() maps the reference's passes onto kernels */
typedef enum fluid_schedule {
    FLUID_SCHED_PASSES = 0, /* one kernel per reference pass (7 + ITERS launches)        */
    FLUID_SCHED_FUSED = 1   /* fused / temporally blocked kernels; same results bit for bit */
} fluid_schedule;

/* Storage of the simulation fields.  The reference keeps them in half-float textures on a real GPU (halfFloatTexType,
 * script.js:138; formats 145-147); the headless SwiftShader build the goldens come from keeps fp32.  Arithmetic is fp32
 * in both modes; FLUID_STORE_F16 rounds every pass output to fp16 (nearest even) and moves half the bytes per step.
 * The host boundary (read / write field) speaks fp32 in both modes. */
typedef enum fluid_storage {
    FLUID_STORE_F32 = 0,
    FLUID_STORE_F16 = 1
} fluid_storage;

/* replaces the size / format part of initFramebuffers(), script.js:982-1010 */
typedef struct fluid_desc {
    int sim_w, sim_h;   /* GLOBAL velocity/pressure/divergence/curl size (getResolution(SIM_RESOLUTION)) */
    int dye_w, dye_h;   /* GLOBAL dye size (getResolution(DYE_RESOLUTION))                               */
    int device;         /* HIP device ordinal                                                            */
    int part, parts;    /* row-stripe decomposition; 0, 1 for a whole-domain context                    */
    int halo;           /* ghost rows (in sim rows) on each side of the stripe; 0 for whole domain       */
    int schedule;       /* fluid_schedule for fluid_step()                                               */
    int part_x, parts_x; /* 2-D tile decomposition: column tile part_x of parts_x (0, 1 or 0, 0: full width); the   */
                        /* rank owns rows of stripe `part` x columns of tile `part_x`, ghost depth `halo` all round */
    int storage;        /* fluid_storage: FLUID_STORE_F32 (0, default) or FLUID_STORE_F16                */
} fluid_desc;

/* the per-step uniforms step() reads from `config`, script.js:1243, 1255, 1262, 1283, 1291 */
typedef struct fluid_params {
    float curl;                 /* config.CURL                  */
    float pressure;             /* config.PRESSURE              */
    int iterations;             /* config.PRESSURE_ITERATIONS   */
    float velocity_dissipation; /* config.VELOCITY_DISSIPATION  */
    float density_dissipation;  /* config.DENSITY_DISSIPATION   */
} fluid_params;

typedef struct fluid_field_info {
    int width, height;  /* global size                                   */
    int channels;       /* floats per texel                              */
    int row0, rows;     /* owned global rows [row0, row0 + rows)         */
    int halo;           /* ghost rows each side, in this field's rows    */
    int col0, cols;     /* owned global columns [col0, col0 + cols)      */
    int halo_x;         /* ghost columns each side (0 unless parts_x > 1) */
    int bytes_per_channel; /* 4 (FLUID_STORE_F32) or 2 (FLUID_STORE_F16): element size of the DEVICE array       */
    /* layout of the device array (fluid_field_device_ptr): array row r = global row row0 - halo + r, array column k = global
     * column array_col0 + k, `pitch` texels per array row (a multiple of 4 >= the columns held: rows stay 16-byte aligned for
     * every width; a 2-D tile holds its owned + ghost columns only).  Columns beyond the field's width are padding. */
    int pitch, array_col0;
} fluid_field_info;

/* per-pass device time of the last fluid_step*(), filled when timing is enabled */
typedef struct fluid_timings {
    float curl_ms, vorticity_ms, divergence_ms, clear_ms, jacobi_ms, gradsub_ms, advect_velocity_ms, advect_dye_ms;
    float total_ms;
    int jacobi_launches;   /* kernel launches the Jacobi loop took (ITERS, or fewer when temporally blocked) */
    int steps;             /* steps the sums cover */
    int folded_launches;   /* how many of jacobi_launches also carried the gradient subtract (fused schedule: the last launch of a
                            * step's loop).  Their time is in gradsub_ms; jacobi_ms covers the other jacobi_launches - folded_launches */
} fluid_timings;

typedef struct fluid_ctx fluid_ctx;

int fluid_abi_version(void);
/* "product" (make: the kernels the grid-driven schedule can pick, no tuning knobs read) or "probes" (make PROBES=1: libfluid_hip_probes.so, every
 * lab shape of profiles/ and the FLUID_* A/B knobs that select them).  Since ABI 8. */
const char *fluid_build_flavor(void);
const char *fluid_error_string(int status);
const char *fluid_last_error(const fluid_ctx *ctx); /* ctx may be NULL: last create() failure */
int fluid_device_count(int *count);

/* createDoubleFBO / createFBO for the five fields, script.js:1045-1106 + 982-1010: zero fields, dye alpha = 1 */
int fluid_create(const fluid_desc *desc, fluid_ctx **out);
int fluid_destroy(fluid_ctx *ctx);

/* initFramebuffers() after a resolution change, script.js:982-1010 + resizeDoubleFBO 1116-1126:
 * velocity and dye are bilinearly resampled (copyShader, 496-506) unless their size is unchanged;
 * pressure, divergence and curl are recreated zero.  Whole-domain contexts only. */
int fluid_resize(fluid_ctx *ctx, int sim_w, int sim_h, int dye_w, int dye_h);

int fluid_set_schedule(fluid_ctx *ctx, int schedule);
/* external != 0: run on the caller-owned hipStream_t `hip_stream` (e.g. torch's current stream; NULL is the
 * HIP null stream); external == 0: back to the context's own non-blocking stream */
int fluid_set_stream(fluid_ctx *ctx, void *hip_stream, int external);

/* splat(x, y, dx, dy, color), script.js:1441-1455: velocity pass then dye pass.
 * aspect = canvas.width / canvas.height (1444); radius = correctRadius(SPLAT_RADIUS / 100) (1447, 1457-1462) */
int fluid_splat(fluid_ctx *ctx, float x, float y, float dx, float dy, float r, float g, float b,
                float aspect, float radius);

/* step(dt), script.js:1231-1294.  On a stripe context (parts > 1): this rank's share, see the multi-GPU block below */
int fluid_step(fluid_ctx *ctx, float dt, const fluid_params *params);
/* n consecutive step(dt) without returning to the host (replays, offline runs, the benchmark loop).  Afterwards every field holds exactly
 * what n calls of fluid_step leave — velocity, pressure, dye, and the divergence and curl of the LAST step; what lies between two of the n
 * steps is nobody's to read, and on one grid held by one domain (fp32, below 3072^2 texels by default) the library uses that: each step's
 * advection launch also runs the next step's curl / vorticity / divergence on the velocity it has just advected, which then never goes
 * through memory (k_advect_cvd). */
int fluid_step_n(fluid_ctx *ctx, int n, float dt, const fluid_params *params);

int fluid_sync(fluid_ctx *ctx);

/* framebufferToTexture(target), script.js:301-307, in the field's native channel count.
 * Whole-domain: the full field.  Stripe context: the owned rows only. `bytes` must match. */
int fluid_read_field(fluid_ctx *ctx, int field, float *host, size_t bytes);
/* state injection / checkpoint restore (no reference equivalent; the test port) */
int fluid_write_field(fluid_ctx *ctx, int field, const float *host, size_t bytes);
int fluid_field_info_get(const fluid_ctx *ctx, int field, fluid_field_info *out);

/* ---- single passes: one reference program each, for per-pass parity tests and for the
 *      stripe driver.  ext = ghost rows beyond the owned rows to compute as well. ---- */
int fluid_pass_curl(fluid_ctx *ctx, int ext);                        /* curlProgram        script.js:1234-1237 */
int fluid_pass_vorticity(fluid_ctx *ctx, float curl, float dt, int ext); /* vorticityProgram 1239-1246 (swaps velocity) */
int fluid_pass_divergence(fluid_ctx *ctx, int ext);                  /* divergenceProgram  1248-1251 */
/* curl + vorticity + divergence (script.js:1234-1251) as ONE kernel under the fused schedule (three under
 * the per-pass schedule): velocity must be valid ext + 3 ghost rows out; curl, the confined velocity and its
 * divergence come back valid ext rows out.  Bit-identical to the three passes run in turn. */
int fluid_pass_curl_vorticity_divergence(fluid_ctx *ctx, float curl, float dt, int ext);
int fluid_pass_clear(fluid_ctx *ctx, float value, int ext);          /* clearProgram       1253-1257 (swaps pressure) */
/* `iters` Jacobi iterations (pressureProgram, 1259-1266); input must be valid `ext_out + iters`
 * rows beyond the owned rows, output is valid `ext_out` rows beyond.  Uses the context's schedule. */
int fluid_pass_jacobi(fluid_ctx *ctx, int iters, int ext_out);
/* clear (x value) followed by `iters` Jacobi iterations, script.js:1253-1266; under the fused schedule the clear is
 * folded into the first iteration's loads (same rounding).  Pressure must be valid ext_out + iters rows out. */
int fluid_pass_clear_jacobi(fluid_ctx *ctx, float value, int iters, int ext_out);
int fluid_pass_gradsub(fluid_ctx *ctx, int ext);                     /* gradienSubtractProgram 1268-1273 (swaps velocity) */
int fluid_pass_advect_velocity(fluid_ctx *ctx, float dt, float dissipation, int ext); /* advectionProgram 1275-1285 */
int fluid_pass_advect_dye(fluid_ctx *ctx, float dt, float dissipation);               /* advectionProgram 1287-1293 */
/* both advection draws (script.js:1275-1293): one kernel under the fused schedule when the dye grid equals the
 * sim grid, else the two passes.  Velocity AND dye ghost rows must be valid before the call. */
int fluid_pass_advect(fluid_ctx *ctx, float dt, float velocity_dissipation, float density_dissipation);
/* one splatProgram draw (script.js:1442-1454) into FLUID_VELOCITY (c2 ignored) or FLUID_DYE, owned + ghost rows */
int fluid_pass_splat(fluid_ctx *ctx, int field, float x, float y, float aspect, float radius,
                     float c0, float c1, float c2);

/* ---- ghost-row staging for the stripe driver (rows are contiguous, so this is a D2D copy) ----
 * side: 0 = bottom (low rows), 1 = top.  pack: copy the `nrows` OWNED rows nearest that side into
 * dev_buf; unpack: copy dev_buf into the `nrows` GHOST rows nearest the owned rows on that side.
 * nrows counts rows of that field (dye rows for FLUID_DYE). dev_buf is device memory. */
int fluid_halo_pack(fluid_ctx *ctx, int field, int side, int nrows, void *dev_buf);
int fluid_halo_unpack(fluid_ctx *ctx, int field, int side, int nrows, const void *dev_buf);
/* device address of array row 0 (ghost rows included) of the field's CURRENT read buffer — the zero-copy form of "DoubleFBO.read after a
 * call is the result" (script.js:1079-1106), and what the stripe driver sends / receives ghost rows through.
 * Passes that swap read/write invalidate it: query after each pass.
 * The texels behind the pointer are always the layout fluid_field_info describes (dye: RGBA).  Internally the library may keep state that
 * the pointer cannot see — the next step's curl / vorticity / divergence computed ahead, the dye packed to three floats while its alpha is
 * one known value: asking for a pointer drops the former and converts the latter back, and — because whatever is written through the
 * pointer is invisible to the library — the dye is not packed again until the next splat.  A cost in speed only, never in results.
 *
 * ORDERING (the rule every zero-copy consumer lives by).  The memory holds the field once the work the context has enqueued ON ITS STREAM
 * up to and including this call has run: the library's streams are non-blocking, so a kernel / copy / torch op on ANY other stream — the
 * null stream included — is NOT ordered against it by itself.  A consumer does one of:
 *   (1) fluid_sync(ctx) in front of the call.  Work the call itself has to enqueue (the packed-dye conversion above) is then waited for
 *       INSIDE the call: `fluid_sync(); fluid_field_device_ptr();` always returns memory that is complete for the host and every stream;
 *   (2) fluid_stream_wait_context(ctx, consumer_stream) behind the call: no host wait, the consumer's stream waits on the device;
 *   (3) run on the context's stream (fluid_set_stream gave the library the consumer's stream: the stripe driver's case).
 * The reverse hazard is the consumer's as well: the context's NEXT call overwrites / swaps these buffers, so work that still reads (or
 * writes) through the pointer must have finished, or fluid_context_wait_stream(ctx, consumer_stream) must be called, before that call. */
int fluid_field_device_ptr(fluid_ctx *ctx, int field, void **dev_ptr);
/* `hip_stream` (a hipStream_t; NULL = the legacy null stream) waits, on the device, for everything this context has enqueued so far —
 * its steps, passes, exchanges and conversions; work enqueued on `hip_stream` after the call sees their results.  Does not block the host. */
int fluid_stream_wait_context(fluid_ctx *ctx, void *hip_stream);
/* the other direction: whatever this context enqueues after the call runs behind everything enqueued on `hip_stream` so far (a consumer
 * still reading a field through a raw pointer, a producer that wrote ghost rows through one).  Does not block the host. */
int fluid_context_wait_stream(fluid_ctx *ctx, void *hip_stream);
/* synchronises, then returns FLUID_ERR_HALO if any advection tap since the last check fell outside the stripe's rows.
 * fluid_step / fluid_step_n / fluid_group_step_n on stripe and tile contexts call it themselves at the end of every call
 * (the step that sampled a row that was not refreshed fails; the call synchronises), so this entry point only matters to a
 * driver that runs the passes and exchanges itself (fluid_pass_*). */
int fluid_halo_check(fluid_ctx *ctx);

/* ---- multi-GPU: one stripe context per GPU, one process per GPU, ghost rows over RCCL (xGMI) ----
 * The reference is single-GPU, so nothing in script.js corresponds to this block; it is how step() (1231-1294)
 * runs when `parts` > 1.  fluid_step / fluid_step_n on a stripe context execute the plan below: pass groups on the
 * stripe's window with neighbour ncclSend / ncclRecv (one ncclGroup per exchange, in place on the ghost rows)
 * between them, all on the context stream.  No global collective is on the data path.
 *
 * WHAT IS COLLECTIVE ON A SET (every rank makes the call, with the same arguments, in the same order): fluid_step / fluid_step_n (they
 * exchange), fluid_comm_init, fluid_comm_calibrate_link — and, since ABI 9, everything that puts dye texels into a context behind the
 * library's back or resets their alpha: fluid_splat / fluid_pass_splat on FLUID_DYE (the reference's splat() draws into the whole dye
 * texture: every rank draws it on its window), fluid_write_field(FLUID_DYE), fluid_halo_unpack(FLUID_DYE), fluid_field_device_ptr(FLUID_DYE).
 * Reason: a rank of at least 3072^2 owned texels whose dye grid is its sim grid keeps the dye packed to three floats per texel through the
 * fused advection, as a whole-domain context does, and its ghost texels then travel as 12-byte texels, in place — the field's FORMAT is
 * part of the message layout two neighbours cut.  The library derives it from nothing but those calls and the steps' arguments, so ranks
 * that make them alike agree by construction (fluid_group_step_n checks format and alpha across an in-process set and refuses a set that
 * disagrees).  READS are not collective: fluid_read_field(FLUID_DYE) on one rank converts into the field's spare buffer and changes
 * nothing. */
typedef enum fluid_stripe_op_kind {
    FLUID_OP_EXCHANGE = 0,       /* refresh n_items fields' ghost rows from both neighbours                  */
    FLUID_OP_CURL_VORT_DIV = 1,  /* script.js:1234-1251, ghost rows out to `ext`                               */
    FLUID_OP_CLEAR = 2,          /* 1253-1257                                                                  */
    FLUID_OP_CLEAR_JACOBI = 3,   /* 1253-1266: clear, then `iters` Jacobi iterations leaving `ext` ghost rows  */
    FLUID_OP_JACOBI = 4,         /* `iters` more iterations                                                    */
    FLUID_OP_GRADSUB = 5,        /* 1268-1273.  The NATIVE driver (fluid_step_n / fluid_group_step_n) runs this op INSIDE the launch of the */
                                 /* JACOBI / CLEAR_JACOBI block in front of it where the library folds the gradient subtract (grids below   */
                                 /* 3072^2 owned texels: fluid_schedule_info.gradsub_folded).  That block then stores the pressure for the */
                                 /* OWNED rows / columns only (ext 0) — after the step the pressure ghost ring is stale, where the plan's  */
                                 /* JACOBI(ext 1) and a driver that runs the ops one by one (fluid_pass_*) leave it valid.  Nothing reads  */
                                 /* it: the next step's first exchange refreshes the pressure ghosts before its CLEAR_JACOBI.              */
    FLUID_OP_ADVECT = 6          /* 1275-1293                                                                  */
} fluid_stripe_op_kind;

typedef struct fluid_stripe_op {
    int kind;        /* fluid_stripe_op_kind                        */
    int iters, ext;  /* pass operands                               */
    int n_items;     /* EXCHANGE: how many (field, rows) pairs      */
    int field[2];    /* fluid_field                                 */
    int rows[2];     /* rows of that field to refresh on each side  */
} fluid_stripe_op;

/* the per-step plan for a stripe with `halo` sim ghost rows, `dye_halo` dye ghost rows and `iterations` Jacobi
 * iterations; advect_rows / advect_dye_rows = velocity / dye rows refreshed in front of the advection (pure host
 * logic, no device needed).  ops may be NULL to query *n_ops. */
int fluid_stripe_plan(int halo, int dye_halo, int iterations, int advect_rows, int advect_dye_rows,
                      fluid_stripe_op *ops, int max_ops, int *n_ops);
/* rows an advection back-trace may span: dt * max|v| + 2.  Default 24: dt <= 1/60 (script.js:1191); the vorticity pass
 * clamps |v| to 1000 (script.js:864), but the projection that follows overshoots it — 1106 measured on the 4096 x 32768
 * grid of the 8-rank bench (tools/max_velocity.py) = 18.4 rows; 24 holds up to |v| ~ 1300.  Only that many velocity / dye
 * ghost rows are refreshed before the advection; a longer back-trace is counted and reported by fluid_halo_check
 * (FLUID_ERR_HALO), never silently served from a stale row. */
int fluid_set_reach(fluid_ctx *ctx, int rows);
int fluid_advect_exchange_rows(const fluid_ctx *ctx, int *velocity_rows, int *dye_rows);
/* interior-first overlap of the exchanges with the curl/vorticity/divergence and advection kernels (default on) */
int fluid_set_overlap(fluid_ctx *ctx, int enabled);
/* What one neighbour message costs on this machine's links: latency_us + bytes / gbytes_per_s (default 20 us, 50 GB/s: an RCCL
 * point-to-point message of a few MB over one xGMI link).  With the overlap on, the driver also cuts the leading launch(es) of a
 * pressure block behind an exchange — their interior computes while the ghost texels travel — and sizes that cut (0, 1 or 2
 * launches, each at the price of one thin launch) with this figure; nothing but speed depends on it.  Every rank of a set may
 * keep its own.  Since ABI 8. */
int fluid_set_link_model(fluid_ctx *ctx, float latency_us, float gbytes_per_s);
/* MEASURES the two figures instead of assuming them (the defaults are guesses nobody calibrated on xGMI, and a wrong model costs 2-3 % either
 * way).  Collective over the stripe / tile set — every rank calls it once, behind fluid_comm_init: `reps` (<= 0: 20) grouped ncclSend /
 * ncclRecv exchanges with the row and column neighbours, as a step's exchange issues them, of a 4 KB message and of the largest message a step
 * sends (the velocity + pressure ghost rows), back to back on the comm stream with an event between every two; latency = the median small
 * exchange, bandwidth = the extra bytes / the extra time of the large one.  Sets the context's link model (fluid_set_link_model) and reports it.
 * A rank without neighbours keeps its model and reports that.  Since ABI 9. */
int fluid_comm_calibrate_link(fluid_ctx *ctx, int reps, float *latency_us, float *gbytes_per_s);

typedef struct fluid_comm_id {
    char bytes[128]; /* an ncclUniqueId */
} fluid_comm_id;

/* which RCCL to dlopen (default: FLUID_RCCL_LIB, then librccl.so.1).  A process that already holds an RCCL — PyTorch
 * bundles one — passes that file so both share one library and one HIP runtime.  Before the first comm call. */
int fluid_comm_set_library(const char *path);
/* rank 0 creates the id (ncclGetUniqueId) and hands it to every rank out of band (any launcher transport) */
int fluid_comm_unique_id(fluid_comm_id *id);
/* collective over the stripe set: ncclCommInitRank(parts, id, part) on the context's device */
int fluid_comm_init(fluid_ctx *ctx, const fluid_comm_id *id);
/* loops `nfloats` floats through ncclSend/ncclRecv to this very rank, stream-ordered between two kernels */
int fluid_comm_selftest(fluid_ctx *ctx, int nfloats);
/* neighbour exchanges issued so far by this context */
long fluid_exchange_count(const fluid_ctx *ctx);
/* the SAME plan for a whole stripe set living in one process (contexts 0..parts-1 in order, any devices): ghost
 * rows move by device-to-device copies instead of RCCL.  Validation path on a single-GPU box. */
int fluid_group_step_n(fluid_ctx **ctxs, int n_ctx, int steps, float dt, const fluid_params *params);

/* ---- display compositor: render(target), script.js:1296-1419 — the first consumer of dye.read ----
 * bloom (applyBloom 1346-1389), sunrays (applySunrays 1391-1403 + blur 1405-1419), then drawColor(BACK_COLOR) and
 * drawDisplay (shading, bloom, sunrays, dithering, gamma; shader 549-612) with the target != null blend state
 * (ONE, ONE_MINUS_SRC_ALPHA; blending off when TRANSPARENT).  Whole-domain contexts only. */
typedef struct fluid_display_params {
    int shading, bloom, sunrays, transparent;       /* config.SHADING / BLOOM / SUNRAYS / TRANSPARENT              */
    float back_r, back_g, back_b;                   /* normalizeColor(config.BACK_COLOR): components / 255          */
    int bloom_w, bloom_h;                           /* getResolution(config.BLOOM_RESOLUTION)                       */
    int bloom_iterations;                           /* config.BLOOM_ITERATIONS                                      */
    double bloom_intensity, bloom_threshold, bloom_soft_knee; /* doubles: the knee curve is computed in JS doubles (1354-1358) */
    int sunrays_w, sunrays_h;                       /* getResolution(config.SUNRAYS_RESOLUTION)                     */
    double sunrays_weight;                          /* config.SUNRAYS_WEIGHT                                        */
} fluid_display_params;

enum { FLUID_DISPLAY_BLOOM = 0, FLUID_DISPLAY_SUNRAYS = 1 };

/* the dithering texture (script.js:958, createTextureAsync 1128-1158): R channel in [0, 1], sampled LINEAR + REPEAT.
 * Default: the reference's 1 x 1 white placeholder (1135) — the blue-noise PNG is an asset of the page, not shipped. */
int fluid_set_dither(fluid_ctx *ctx, const float *host_r, int width, int height);
/* render(target) into the context's width x height float RGBA frame (captureScreenshot's target, script.js:287-290) */
int fluid_render(fluid_ctx *ctx, int width, int height, const fluid_display_params *params);
/* framebufferToTexture(target), script.js:301-307: RGBA floats, row 0 = bottom */
int fluid_read_frame(fluid_ctx *ctx, float *host_rgba, size_t bytes);
/* normalizeTexture(texture, w, h), script.js:309-323: clamp01 * 255 truncated to bytes, rows flipped (top row first) */
int fluid_read_frame_rgba8(fluid_ctx *ctx, unsigned char *host_rgba8, size_t bytes);
/* the bloom (RGBA) or blurred sunrays (R) buffer of the last render; host may be NULL to query the size */
int fluid_read_display_buffer(fluid_ctx *ctx, int which, float *host, size_t bytes, int *width, int *height);

/* (ABI 10) The curl field as an OUTPUT of fluid_step / fluid_step_n.  The reference writes its curl texture in every step (curlProgram,
 * script.js:1234-1237) and reads it in the same step only (vorticityProgram, 1239-1243): nothing outside step() looks at it.  On (default): a
 * call leaves the curl of its LAST step in FLUID_CURL, as the header's contract says.  Off: no step stores it — the fused curl / vorticity /
 * divergence launch keeps it in registers, 4 B/texel less per call; one step per call, the page's update() pattern (1176-1186), is then
 * within 2 % of a batched call at 4096^2 instead of 4 % (bench.py `per_frame`).  Reading FLUID_CURL (fluid_read_field, fluid_field_device_ptr,
 * a ghost-row pack) after a step that did not store it fails with FLUID_ERR_INVALID rather than return an older step's field.  The per-pass
 * entry points and the one-kernel-per-pass schedule always write it; so do the small-grid launches that carry the next step's stencil stages. */
int fluid_set_curl_output(fluid_ctx *ctx, int enabled);

int fluid_set_timing(fluid_ctx *ctx, int enabled);
int fluid_get_timings(fluid_ctx *ctx, fluid_timings *out);

/* What fluid_step_n(ctx, n_steps, dt, params) would launch on this context right now (since ABI 8) — the library picks kernels by grid
 * size (and by its FLUID_* A/B knobs), and a measurement has to be able to say which ones it timed (bench.py `config`). */
typedef struct fluid_schedule_info {
    int fused;             /* the context runs FLUID_SCHED_FUSED kernels                                                         */
    int jacobi_shape;      /* tile-shape table index of the temporally blocked Jacobi launches; -1: one launch per iteration     */
    int jacobi_launches;   /* Jacobi launches per step                                                                           */
    int gradsub_folded;    /* the gradient subtract rides in a step's last Jacobi launch                                         */
    int chained;           /* steps of the call whose advection launch also runs the next step's curl / vorticity / divergence  */
    int curl_stores;       /* steps of the call that store their curl field (the last one always does)                           */
    int launches;          /* kernel launches of the whole call (whole-domain contexts; 0 for stripes: see fluid_stripe_plan)    */
    int runs_ahead;        /* the call's LAST advection launch also runs the next call's curl / vorticity / divergence into       */
                           /* pending buffers (what makes one fluid_step per frame cost six launches instead of seven)            */
    int pending_adopted;   /* this call's first curl / vorticity / divergence was already run ahead by the previous call         */
    int dye_packed;        /* the fused advection runs on the dye PACKED to three floats per texel (its alpha is one known value:  */
                           /* 40 instead of 48 B/texel; whole-domain fp32 contexts at >= 3072^2 texels); anything that reads or     */
                           /* writes dye texels sees RGBA — the library converts on demand                                          */
    int jacobi_chained;    /* (ABI 9) the step's `jacobi_launches` blocks of iterations run as ONE launch whose tiles wait for the tiles   */
                           /* of the previous block they read (fp32 grids of 3072^2 ... 20 M texels: no fill / drain between the blocks; on a stripe / */
                           /* tile rank: the launches it has left behind the ones cut around an exchange)                                 */
} fluid_schedule_info;
int fluid_schedule_info_get(fluid_ctx *ctx, int n_steps, float dt, const fluid_params *params, fluid_schedule_info *out);

/* Step marks (since ABI 8): one event on the context's stream in front of the first and behind every step of the NEXT fluid_step_n /
 * fluid_group_step_n calls (the first `capacity` steps of each call; 0 switches the marks off).  Timing-only events (no system-scope
 * fence when one is recorded: with default events every mark cost the stream 6 us).  Unlike fluid_set_timing nothing waits:
 * the stream runs exactly as it does unmarked, so the marks show how a step's time develops INSIDE a timed window (bench.py
 * `timed_window_regime`: the first steps after an idle run slower than the steady state).  fluid_get_step_marks waits for the last
 * mark of the last call and returns the device time of each of its marked steps in milliseconds. */
int fluid_set_step_marks(fluid_ctx *ctx, int capacity);
int fluid_get_step_marks(fluid_ctx *ctx, float *ms, int capacity, int *n_steps);

#ifdef __cplusplus
}
#endif
#endif /* FLUID_HIP_H */

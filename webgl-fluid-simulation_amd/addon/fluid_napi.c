/*
 * fluid_napi.c — Node N-API addon over the C ABI of libfluid_hip.so (include/fluid_hip.h).
 *
 * The reference's host language is JavaScript; this is the thin native layer that lets a
 * JavaScript host (addon/fluid.js, which mirrors the reference's config / splat / multipleSplats /
 * step / initFramebuffers surface) call the HIP hot path.  One exported function per C-ABI entry
 * point the shim needs; numbers arrive as JS doubles and are narrowed to float here exactly like
 * gl.uniform1f does in the reference.  Errors become JS exceptions carrying fluid_last_error().
 *
 * Build: gcc -shared -fPIC -I/usr/include/node -I../../include fluid_napi.c -L.. -lfluid_hip
 */
#include <node_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fluid_hip.h"

#define NAPI_OK(call)                                                         \
    do {                                                                      \
        if ((call) != napi_ok) {                                              \
            napi_throw_error(env, NULL, "fluid_napi: N-API call failed: " #call); \
            return NULL;                                                      \
        }                                                                     \
    } while (0)

static napi_value throw_status(napi_env env, fluid_ctx *ctx, int rc)
{
    char msg[512];
    const char *detail = fluid_last_error(ctx);
    snprintf(msg, sizeof msg, "libfluid_hip: %s: %s", fluid_error_string(rc), detail ? detail : "");
    char code[16];
    snprintf(code, sizeof code, "%d", rc);
    napi_throw_error(env, code, msg);
    return NULL;
}

/* an optional trailing integer argument (absent or undefined -> dflt) */
static int get_opt_i(napi_env env, napi_callback_info info, size_t index, int dflt, int *out)
{
    napi_value argv[16];
    size_t argc = 16;
    napi_valuetype t;
    *out = dflt;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok) return 0;
    if (index >= argc || index >= 16) return 1;
    if (napi_typeof(env, argv[index], &t) != napi_ok) return 0;
    if (t == napi_undefined) return 1;
    int32_t v;
    if (napi_get_value_int32(env, argv[index], &v) != napi_ok) {
        napi_throw_type_error(env, NULL, "fluid_napi: expected an integer");
        return 0;
    }
    *out = (int)v;
    return 1;
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv)
{
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        napi_throw_type_error(env, NULL, "fluid_napi: wrong number of arguments");
        return 0;
    }
    return 1;
}

static int get_ctx(napi_env env, napi_value v, fluid_ctx **out)
{
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "fluid_napi: expected a fluid context handle");
        return 0;
    }
    *out = *(fluid_ctx **)p; /* the external holds a heap cell so destroy() can null it */
    if (!*out) {
        napi_throw_error(env, NULL, "fluid_napi: context already destroyed");
        return 0;
    }
    return 1;
}

static int get_f(napi_env env, napi_value v, float *out)
{
    double d;
    if (napi_get_value_double(env, v, &d) != napi_ok) {
        napi_throw_type_error(env, NULL, "fluid_napi: expected a number");
        return 0;
    }
    *out = (float)d;
    return 1;
}

static int get_i(napi_env env, napi_value v, int *out)
{
    int32_t i;
    if (napi_get_value_int32(env, v, &i) != napi_ok) {
        napi_throw_type_error(env, NULL, "fluid_napi: expected an integer");
        return 0;
    }
    *out = (int)i;
    return 1;
}

static void finalize_ctx(napi_env env, void *data, void *hint)
{
    fluid_ctx **cell = (fluid_ctx **)data;
    (void)env; (void)hint;
    if (cell) {
        if (*cell) fluid_destroy(*cell);
        free(cell);
    }
}

/* the JS handle of a context: an external holding a heap cell, freed (and the context destroyed) by the finalizer */
static napi_value wrap_ctx(napi_env env, fluid_ctx *ctx)
{
    napi_value ext;
    fluid_ctx **cell = (fluid_ctx **)malloc(sizeof *cell);
    if (cell) *cell = ctx;
    if (!cell || napi_create_external(env, cell, finalize_ctx, NULL, &ext) != napi_ok) {
        fluid_destroy(ctx);
        free(cell);
        napi_throw_error(env, NULL, "fluid_napi: cannot create the context handle");
        return NULL;
    }
    return ext;
}

/* create(simW, simH, dyeW, dyeH, device, schedule[, storage]) -> handle; storage 0 = fp32 fields (default), 1 = fp16 */
static napi_value n_create(napi_env env, napi_callback_info info)
{
    napi_value a[6];
    if (!get_args(env, info, 6, a)) return NULL;
    fluid_desc d;
    memset(&d, 0, sizeof d);
    d.parts = 1;
    if (!get_i(env, a[0], &d.sim_w) || !get_i(env, a[1], &d.sim_h) || !get_i(env, a[2], &d.dye_w) ||
        !get_i(env, a[3], &d.dye_h) || !get_i(env, a[4], &d.device) || !get_i(env, a[5], &d.schedule) ||
        !get_opt_i(env, info, 6, FLUID_STORE_F32, &d.storage))
        return NULL;
    fluid_ctx *ctx = NULL;
    int rc = fluid_create(&d, &ctx);
    if (rc != FLUID_OK) return throw_status(env, NULL, rc);
    return wrap_ctx(env, ctx);
}

/* createTile(simW, simH, dyeW, dyeH, device, schedule, part, parts, partX, partsX, halo[, storage]) -> handle: one rank's share of a
 * multi-GPU run (row stripe `part` of `parts`, column tile `partX` of `partsX`, `halo` ghost rows / columns) */
static napi_value n_create_tile(napi_env env, napi_callback_info info)
{
    napi_value a[11];
    if (!get_args(env, info, 11, a)) return NULL;
    fluid_desc d;
    memset(&d, 0, sizeof d);
    if (!get_i(env, a[0], &d.sim_w) || !get_i(env, a[1], &d.sim_h) || !get_i(env, a[2], &d.dye_w) || !get_i(env, a[3], &d.dye_h) ||
        !get_i(env, a[4], &d.device) || !get_i(env, a[5], &d.schedule) || !get_i(env, a[6], &d.part) || !get_i(env, a[7], &d.parts) ||
        !get_i(env, a[8], &d.part_x) || !get_i(env, a[9], &d.parts_x) || !get_i(env, a[10], &d.halo) ||
        !get_opt_i(env, info, 11, FLUID_STORE_F32, &d.storage))
        return NULL;
    fluid_ctx *ctx = NULL;
    int rc = fluid_create(&d, &ctx);
    if (rc != FLUID_OK) return throw_status(env, NULL, rc);
    return wrap_ctx(env, ctx);
}

/* commUniqueId() -> Buffer(128): rank 0 creates it (ncclGetUniqueId) and hands it to the other ranks */
static napi_value n_comm_unique_id(napi_env env, napi_callback_info info)
{
    (void)info;
    fluid_comm_id id;
    int rc = fluid_comm_unique_id(&id);
    if (rc != FLUID_OK) return throw_status(env, NULL, rc);
    napi_value buf;
    void *data = NULL;
    NAPI_OK(napi_create_buffer_copy(env, sizeof id.bytes, id.bytes, &data, &buf));
    return buf;
}

/* commInit(h, Buffer id): collective over the tile set (ncclCommInitRank); afterwards step() exchanges ghost rows itself */
static napi_value n_comm_init(napi_env env, napi_callback_info info)
{
    napi_value a[2];
    fluid_ctx *c;
    void *data = NULL;
    size_t len = 0;
    if (!get_args(env, info, 2, a) || !get_ctx(env, a[0], &c)) return NULL;
    if (napi_get_buffer_info(env, a[1], &data, &len) != napi_ok || len != sizeof(fluid_comm_id)) {
        napi_throw_type_error(env, NULL, "fluid_napi: commInit expects the 128-byte Buffer of commUniqueId()");
        return NULL;
    }
    fluid_comm_id id;
    memcpy(id.bytes, data, sizeof id.bytes);
    int rc = fluid_comm_init(c, &id);
    return rc == FLUID_OK ? NULL : throw_status(env, c, rc);
}

static napi_value n_set_reach(napi_env env, napi_callback_info info)
{
    napi_value a[2];
    fluid_ctx *c;
    int rows;
    if (!get_args(env, info, 2, a) || !get_ctx(env, a[0], &c) || !get_i(env, a[1], &rows)) return NULL;
    int rc = fluid_set_reach(c, rows);
    return rc == FLUID_OK ? NULL : throw_status(env, c, rc);
}

/* setLinkModel(h, latencyUs, gbytesPerS): what one neighbour message costs on this machine's links (fluid_set_link_model) */
static napi_value n_set_link_model(napi_env env, napi_callback_info info)
{
    napi_value a[3];
    fluid_ctx *c;
    float lat, gbps;
    if (!get_args(env, info, 3, a) || !get_ctx(env, a[0], &c) || !get_f(env, a[1], &lat) || !get_f(env, a[2], &gbps)) return NULL;
    int rc = fluid_set_link_model(c, lat, gbps);
    return rc == FLUID_OK ? NULL : throw_status(env, c, rc);
}

/* calibrateLink(h, reps) -> [latencyUs, gbytesPerS]: collective over the tile set, behind commInit — measures what a neighbour message
 * costs on this node's links and makes it the context's link model (fluid_comm_calibrate_link, ABI 9) */
static napi_value n_calibrate_link(napi_env env, napi_callback_info info)
{
    napi_value a[2], arr, v;
    fluid_ctx *c;
    int reps;
    float lat = 0.0f, gbps = 0.0f;
    if (!get_args(env, info, 2, a) || !get_ctx(env, a[0], &c) || !get_i(env, a[1], &reps)) return NULL;
    int rc = fluid_comm_calibrate_link(c, reps, &lat, &gbps);
    if (rc != FLUID_OK) return throw_status(env, c, rc);
    NAPI_OK(napi_create_array_with_length(env, 2, &arr));
    NAPI_OK(napi_create_double(env, (double)lat, &v));
    NAPI_OK(napi_set_element(env, arr, 0, v));
    NAPI_OK(napi_create_double(env, (double)gbps, &v));
    NAPI_OK(napi_set_element(env, arr, 1, v));
    return arr;
}

/* haloCheck(h): throws (code -5) if an advection back-trace left the refreshed ghost rows / columns */
static napi_value n_halo_check(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    fluid_ctx *c;
    if (!get_args(env, info, 1, a) || !get_ctx(env, a[0], &c)) return NULL;
    int rc = fluid_halo_check(c);
    return rc == FLUID_OK ? NULL : throw_status(env, c, rc);
}

static napi_value n_exchange_count(napi_env env, napi_callback_info info)
{
    napi_value a[1], v;
    fluid_ctx *c;
    if (!get_args(env, info, 1, a) || !get_ctx(env, a[0], &c)) return NULL;
    NAPI_OK(napi_create_int64(env, (int64_t)fluid_exchange_count(c), &v));
    return v;
}

static napi_value n_destroy(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    void *p = NULL;
    if (napi_get_value_external(env, a[0], &p) == napi_ok && p) {
        fluid_ctx **cell = (fluid_ctx **)p;
        if (*cell) fluid_destroy(*cell);
        *cell = NULL;
    }
    return NULL;
}

static napi_value n_resize(napi_env env, napi_callback_info info)
{
    napi_value a[5];
    fluid_ctx *c;
    int v[4];
    if (!get_args(env, info, 5, a) || !get_ctx(env, a[0], &c)) return NULL;
    for (int k = 0; k < 4; k++)
        if (!get_i(env, a[k + 1], &v[k])) return NULL;
    int rc = fluid_resize(c, v[0], v[1], v[2], v[3]);
    return rc == FLUID_OK ? NULL : throw_status(env, c, rc);
}

/* splat(h, x, y, dx, dy, r, g, b, aspect, radius) */
static napi_value n_splat(napi_env env, napi_callback_info info)
{
    napi_value a[10];
    fluid_ctx *c;
    float f[9];
    if (!get_args(env, info, 10, a) || !get_ctx(env, a[0], &c)) return NULL;
    for (int k = 0; k < 9; k++)
        if (!get_f(env, a[k + 1], &f[k])) return NULL;
    int rc = fluid_splat(c, f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8]);
    return rc == FLUID_OK ? NULL : throw_status(env, c, rc);
}

/* step(h, n, dt, curl, pressure, iterations, velocityDissipation, densityDissipation) */
static napi_value n_step(napi_env env, napi_callback_info info)
{
    napi_value a[8];
    fluid_ctx *c;
    int n;
    float dt;
    fluid_params P;
    memset(&P, 0, sizeof P);
    if (!get_args(env, info, 8, a) || !get_ctx(env, a[0], &c)) return NULL;
    if (!get_i(env, a[1], &n) || !get_f(env, a[2], &dt) || !get_f(env, a[3], &P.curl) || !get_f(env, a[4], &P.pressure) ||
        !get_i(env, a[5], &P.iterations) || !get_f(env, a[6], &P.velocity_dissipation) || !get_f(env, a[7], &P.density_dissipation))
        return NULL;
    int rc = fluid_step_n(c, n, dt, &P);
    return rc == FLUID_OK ? NULL : throw_status(env, c, rc);
}

static napi_value n_sync(napi_env env, napi_callback_info info)
{
    napi_value a[1];
    fluid_ctx *c;
    if (!get_args(env, info, 1, a) || !get_ctx(env, a[0], &c)) return NULL;
    int rc = fluid_sync(c);
    return rc == FLUID_OK ? NULL : throw_status(env, c, rc);
}

static napi_value n_set_schedule(napi_env env, napi_callback_info info)
{
    napi_value a[2];
    fluid_ctx *c;
    int s;
    if (!get_args(env, info, 2, a) || !get_ctx(env, a[0], &c) || !get_i(env, a[1], &s)) return NULL;
    int rc = fluid_set_schedule(c, s);
    return rc == FLUID_OK ? NULL : throw_status(env, c, rc);
}

/* fieldInfo(h, field) -> {width, height, channels} */
static napi_value n_field_info(napi_env env, napi_callback_info info)
{
    napi_value a[2], obj, v;
    fluid_ctx *c;
    int field;
    fluid_field_info fi;
    if (!get_args(env, info, 2, a) || !get_ctx(env, a[0], &c) || !get_i(env, a[1], &field)) return NULL;
    int rc = fluid_field_info_get(c, field, &fi);
    if (rc != FLUID_OK) return throw_status(env, c, rc);
    NAPI_OK(napi_create_object(env, &obj));
    NAPI_OK(napi_create_int32(env, fi.width, &v));
    NAPI_OK(napi_set_named_property(env, obj, "width", v));
    NAPI_OK(napi_create_int32(env, fi.height, &v));
    NAPI_OK(napi_set_named_property(env, obj, "height", v));
    NAPI_OK(napi_create_int32(env, fi.channels, &v));
    NAPI_OK(napi_set_named_property(env, obj, "channels", v));
    const int owned[4] = { fi.row0, fi.rows, fi.col0, fi.cols };   /* the block readField returns (the whole field unless a tile) */
    const char *names[4] = { "row0", "rows", "col0", "cols" };
    for (int k = 0; k < 4; k++) {
        NAPI_OK(napi_create_int32(env, owned[k], &v));
        NAPI_OK(napi_set_named_property(env, obj, names[k], v));
    }
    return obj;
}

/* readField(h, field) -> Float32Array(width * height * channels), row 0 = bottom */
static napi_value n_read_field(napi_env env, napi_callback_info info)
{
    napi_value a[2], buf, arr;
    fluid_ctx *c;
    int field;
    fluid_field_info fi;
    if (!get_args(env, info, 2, a) || !get_ctx(env, a[0], &c) || !get_i(env, a[1], &field)) return NULL;
    int rc = fluid_field_info_get(c, field, &fi);
    if (rc != FLUID_OK) return throw_status(env, c, rc);
    size_t n = (size_t)fi.cols * fi.rows * fi.channels; /* the owned block: the whole field for a whole-domain context */
    void *data = NULL;
    NAPI_OK(napi_create_arraybuffer(env, n * sizeof(float), &data, &buf));
    rc = fluid_read_field(c, field, (float *)data, n * sizeof(float));
    if (rc != FLUID_OK) return throw_status(env, c, rc);
    NAPI_OK(napi_create_typedarray(env, napi_float32_array, n, buf, 0, &arr));
    return arr;
}

/* writeField(h, field, Float32Array) */
static napi_value n_write_field(napi_env env, napi_callback_info info)
{
    napi_value a[3], ab;
    fluid_ctx *c;
    int field;
    napi_typedarray_type t;
    size_t len, off;
    void *data;
    if (!get_args(env, info, 3, a) || !get_ctx(env, a[0], &c) || !get_i(env, a[1], &field)) return NULL;
    if (napi_get_typedarray_info(env, a[2], &t, &len, &data, &ab, &off) != napi_ok || t != napi_float32_array) {
        napi_throw_type_error(env, NULL, "fluid_napi: writeField expects a Float32Array");
        return NULL;
    }
    int rc = fluid_write_field(c, field, (const float *)data, len * sizeof(float));
    return rc == FLUID_OK ? NULL : throw_status(env, c, rc);
}

static int get_d(napi_env env, napi_value v, double *out)
{
    if (napi_get_value_double(env, v, out) != napi_ok) {
        napi_throw_type_error(env, NULL, "fluid_napi: expected a number");
        return 0;
    }
    return 1;
}

/* setDither(h, Float32Array r, width, height): the dithering texture's R channel (script.js:958) */
static napi_value n_set_dither(napi_env env, napi_callback_info info)
{
    napi_value a[4], ab;
    fluid_ctx *c;
    int w, h;
    napi_typedarray_type t;
    size_t len, off;
    void *data;
    if (!get_args(env, info, 4, a) || !get_ctx(env, a[0], &c) || !get_i(env, a[2], &w) || !get_i(env, a[3], &h)) return NULL;
    if (napi_get_typedarray_info(env, a[1], &t, &len, &data, &ab, &off) != napi_ok || t != napi_float32_array || len != (size_t)w * h) {
        napi_throw_type_error(env, NULL, "fluid_napi: setDither expects a Float32Array of width * height");
        return NULL;
    }
    int rc = fluid_set_dither(c, (const float *)data, w, h);
    return rc == FLUID_OK ? NULL : throw_status(env, c, rc);
}

/* render(h, width, height, shading, bloom, sunrays, transparent, backR, backG, backB, bloomW, bloomH, bloomIterations,
 *        bloomIntensity, bloomThreshold, bloomSoftKnee, sunraysW, sunraysH, sunraysWeight): render(target), script.js:1296 */
static napi_value n_render(napi_env env, napi_callback_info info)
{
    napi_value a[19];
    fluid_ctx *c;
    int w, h;
    fluid_display_params P;
    memset(&P, 0, sizeof P);
    if (!get_args(env, info, 19, a) || !get_ctx(env, a[0], &c)) return NULL;
    if (!get_i(env, a[1], &w) || !get_i(env, a[2], &h) || !get_i(env, a[3], &P.shading) || !get_i(env, a[4], &P.bloom) ||
        !get_i(env, a[5], &P.sunrays) || !get_i(env, a[6], &P.transparent) || !get_f(env, a[7], &P.back_r) || !get_f(env, a[8], &P.back_g) ||
        !get_f(env, a[9], &P.back_b) || !get_i(env, a[10], &P.bloom_w) || !get_i(env, a[11], &P.bloom_h) ||
        !get_i(env, a[12], &P.bloom_iterations) || !get_d(env, a[13], &P.bloom_intensity) || !get_d(env, a[14], &P.bloom_threshold) ||
        !get_d(env, a[15], &P.bloom_soft_knee) || !get_i(env, a[16], &P.sunrays_w) || !get_i(env, a[17], &P.sunrays_h) ||
        !get_d(env, a[18], &P.sunrays_weight))
        return NULL;
    int rc = fluid_render(c, w, h, &P);
    return rc == FLUID_OK ? NULL : throw_status(env, c, rc);
}

/* readFrame(h, width, height) -> Float32Array(width * height * 4): framebufferToTexture(target), row 0 = bottom */
static napi_value n_read_frame(napi_env env, napi_callback_info info)
{
    napi_value a[3], buf, arr;
    fluid_ctx *c;
    int w, h;
    if (!get_args(env, info, 3, a) || !get_ctx(env, a[0], &c) || !get_i(env, a[1], &w) || !get_i(env, a[2], &h)) return NULL;
    size_t n = (size_t)w * h * 4;
    void *data = NULL;
    NAPI_OK(napi_create_arraybuffer(env, n * sizeof(float), &data, &buf));
    int rc = fluid_read_frame(c, (float *)data, n * sizeof(float));
    if (rc != FLUID_OK) return throw_status(env, c, rc);
    NAPI_OK(napi_create_typedarray(env, napi_float32_array, n, buf, 0, &arr));
    return arr;
}

/* readFrameRgba8(h, width, height) -> Uint8Array(width * height * 4): normalizeTexture, top row first */
static napi_value n_read_frame_rgba8(napi_env env, napi_callback_info info)
{
    napi_value a[3], buf, arr;
    fluid_ctx *c;
    int w, h;
    if (!get_args(env, info, 3, a) || !get_ctx(env, a[0], &c) || !get_i(env, a[1], &w) || !get_i(env, a[2], &h)) return NULL;
    size_t n = (size_t)w * h * 4;
    void *data = NULL;
    NAPI_OK(napi_create_arraybuffer(env, n, &data, &buf));
    int rc = fluid_read_frame_rgba8(c, (unsigned char *)data, n);
    if (rc != FLUID_OK) return throw_status(env, c, rc);
    NAPI_OK(napi_create_typedarray(env, napi_uint8_array, n, buf, 0, &arr));
    return arr;
}

static napi_value n_device_count(napi_env env, napi_callback_info info)
{
    napi_value v;
    int n = 0;
    (void)info;
    fluid_device_count(&n);
    NAPI_OK(napi_create_int32(env, n, &v));
    return v;
}

static napi_value n_set_timing(napi_env env, napi_callback_info info)
{
    napi_value a[2];
    fluid_ctx *c;
    int on;
    if (!get_args(env, info, 2, a) || !get_ctx(env, a[0], &c) || !get_i(env, a[1], &on)) return NULL;
    int rc = fluid_set_timing(c, on);
    return rc == FLUID_OK ? NULL : throw_status(env, c, rc);
}

/* setCurlOutput(handle, on): fluid_set_curl_output (ABI 10) — off: no step stores the curl field nothing outside step() reads */
static napi_value n_set_curl_output(napi_env env, napi_callback_info info)
{
    napi_value a[2];
    fluid_ctx *c;
    int on;
    if (!get_args(env, info, 2, a) || !get_ctx(env, a[0], &c) || !get_i(env, a[1], &on)) return NULL;
    int rc = fluid_set_curl_output(c, on);
    return rc == FLUID_OK ? NULL : throw_status(env, c, rc);
}

static napi_value n_get_timings(napi_env env, napi_callback_info info)
{
    napi_value a[1], obj, v;
    fluid_ctx *c;
    fluid_timings t;
    if (!get_args(env, info, 1, a) || !get_ctx(env, a[0], &c)) return NULL;
    int rc = fluid_get_timings(c, &t);
    if (rc != FLUID_OK) return throw_status(env, c, rc);
    NAPI_OK(napi_create_object(env, &obj));
    const char *names[] = { "curl", "vorticity", "divergence", "clear", "jacobi", "gradsub", "advectVelocity", "advectDye", "total" };
    const float vals[] = { t.curl_ms, t.vorticity_ms, t.divergence_ms, t.clear_ms, t.jacobi_ms, t.gradsub_ms,
                           t.advect_velocity_ms, t.advect_dye_ms, t.total_ms };
    for (int k = 0; k < 9; k++) {
        NAPI_OK(napi_create_double(env, vals[k], &v));
        NAPI_OK(napi_set_named_property(env, obj, names[k], v));
    }
    NAPI_OK(napi_create_int32(env, t.jacobi_launches, &v));
    NAPI_OK(napi_set_named_property(env, obj, "jacobiLaunches", v));
    NAPI_OK(napi_create_int32(env, t.steps, &v));
    NAPI_OK(napi_set_named_property(env, obj, "steps", v));
    NAPI_OK(napi_create_int32(env, t.folded_launches, &v));
    NAPI_OK(napi_set_named_property(env, obj, "foldedLaunches", v));
    return obj;
}

/* scheduleInfo(h, n, dt, curl, pressure, iterations, velocityDissipation, densityDissipation) -> what the next step(h, n, ...) with these
 * arguments would launch (fluid_schedule_info_get, since ABI 8); nothing runs */
static napi_value n_schedule_info(napi_env env, napi_callback_info info)
{
    napi_value a[8], obj, v;
    fluid_ctx *c;
    int n;
    float dt;
    fluid_params P;
    fluid_schedule_info S;
    memset(&P, 0, sizeof P);
    if (!get_args(env, info, 8, a) || !get_ctx(env, a[0], &c)) return NULL;
    if (!get_i(env, a[1], &n) || !get_f(env, a[2], &dt) || !get_f(env, a[3], &P.curl) || !get_f(env, a[4], &P.pressure) ||
        !get_i(env, a[5], &P.iterations) || !get_f(env, a[6], &P.velocity_dissipation) || !get_f(env, a[7], &P.density_dissipation))
        return NULL;
    int rc = fluid_schedule_info_get(c, n, dt, &P, &S);
    if (rc != FLUID_OK) return throw_status(env, c, rc);
    NAPI_OK(napi_create_object(env, &obj));
    const char *names[] = { "fused", "jacobiShape", "jacobiLaunches", "gradsubFolded", "chained", "curlStores", "launches", "runsAhead",
                            "pendingAdopted", "dyePacked", "jacobiChained" };
    const int vals[] = { S.fused, S.jacobi_shape, S.jacobi_launches, S.gradsub_folded, S.chained, S.curl_stores, S.launches, S.runs_ahead,
                         S.pending_adopted, S.dye_packed, S.jacobi_chained };
    for (int k = 0; k < 11; k++) {
        NAPI_OK(napi_create_int32(env, vals[k], &v));
        NAPI_OK(napi_set_named_property(env, obj, names[k], v));
    }
    return obj;
}

/* setStepMarks(h, capacity): an event in front of the first and behind every step of the next step() calls (fluid_set_step_marks) */
static napi_value n_set_step_marks(napi_env env, napi_callback_info info)
{
    napi_value a[2];
    fluid_ctx *c;
    int cap;
    if (!get_args(env, info, 2, a) || !get_ctx(env, a[0], &c) || !get_i(env, a[1], &cap)) return NULL;
    int rc = fluid_set_step_marks(c, cap);
    return rc == FLUID_OK ? NULL : throw_status(env, c, rc);
}

/* getStepMarks(h) -> [ms of each marked step of the last step() call] (waits for the last mark) */
static napi_value n_get_step_marks(napi_env env, napi_callback_info info)
{
    napi_value a[1], arr, v;
    fluid_ctx *c;
    float ms[256];
    int n = 0;
    if (!get_args(env, info, 1, a) || !get_ctx(env, a[0], &c)) return NULL;
    int rc = fluid_get_step_marks(c, ms, 256, &n);
    if (rc != FLUID_OK) return throw_status(env, c, rc);
    if (n > 256) n = 256;
    NAPI_OK(napi_create_array_with_length(env, (size_t)n, &arr));
    for (int k = 0; k < n; k++) {
        NAPI_OK(napi_create_double(env, ms[k], &v));
        NAPI_OK(napi_set_element(env, arr, (uint32_t)k, v));
    }
    return arr;
}

static napi_value init(napi_env env, napi_value exports)
{
    const struct { const char *name; napi_callback fn; } fns[] = {
        { "create", n_create }, { "createTile", n_create_tile }, { "commUniqueId", n_comm_unique_id }, { "commInit", n_comm_init },
        { "setReach", n_set_reach }, { "haloCheck", n_halo_check }, { "exchangeCount", n_exchange_count }, { "destroy", n_destroy }, { "resize", n_resize }, { "splat", n_splat },
        { "step", n_step }, { "sync", n_sync }, { "setSchedule", n_set_schedule }, { "fieldInfo", n_field_info },
        { "readField", n_read_field }, { "writeField", n_write_field }, { "deviceCount", n_device_count },
        { "setTiming", n_set_timing }, { "getTimings", n_get_timings }, { "setCurlOutput", n_set_curl_output },
        { "setDither", n_set_dither }, { "render", n_render }, { "readFrame", n_read_frame }, { "readFrameRgba8", n_read_frame_rgba8 },
        { "scheduleInfo", n_schedule_info }, { "setStepMarks", n_set_step_marks }, { "getStepMarks", n_get_step_marks },
        { "setLinkModel", n_set_link_model },
        { "calibrateLink", n_calibrate_link },
    };
    for (size_t k = 0; k < sizeof fns / sizeof fns[0]; k++) {
        napi_value f;
        if (napi_create_function(env, fns[k].name, NAPI_AUTO_LENGTH, fns[k].fn, NULL, &f) != napi_ok) return NULL;
        if (napi_set_named_property(env, exports, fns[k].name, f) != napi_ok) return NULL;
    }
    napi_value v;
    if (napi_create_int32(env, fluid_abi_version(), &v) == napi_ok) napi_set_named_property(env, exports, "abiVersion", v);
    if (napi_create_string_utf8(env, fluid_build_flavor(), NAPI_AUTO_LENGTH, &v) == napi_ok) napi_set_named_property(env, exports, "buildFlavor", v);
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)

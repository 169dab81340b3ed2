// fluid_display.cpp — render(target) of the reference (script.js:1296-1419) behind the C ABI: owns the bloom pyramid,
// the sunrays buffers, the dithering texture and the frame, and sequences the display kernels exactly like
// applyBloom / applySunrays / blur / drawColor / drawDisplay do.  SURVEY.md §8f N3: the first consumer of dye.read.
#include "fluid_display.h"
#include "fluid_internal.h"

#include <new>
#include <vector>

using namespace fluid;

struct fluid_display_state {
    struct Buf {
        void* p = nullptr;
        int w = 0, h = 0;
        size_t bytes = 0;
    };
    Buf bloom, sunrays, sunrays_tmp, frame, frame8, dither;
    Buf dye32[2];  // fp16 storage: fp32 copy of dye.read / stand-in for dye.write (the compositor computes on fp32 texels)
    std::vector<Buf> levels;
    bool bloom_valid = false;
};

namespace {

int ensure(fluid_ctx* c, fluid_display_state::Buf& b, int w, int h, size_t texel_bytes)
{
    const size_t need = (size_t)w * h * texel_bytes;
    if (b.p && b.bytes >= need) {
        b.w = w;
        b.h = h;
        return FLUID_OK;
    }
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    HIPCK(c, hipMalloc(&b.p, need));
    b.w = w;
    b.h = h;
    b.bytes = need;
    return FLUID_OK;
}

fluid_display_state* state(fluid_ctx* c)
{
    if (!c->display) c->display = new (std::nothrow) fluid_display_state();
    return c->display;
}

int default_dither(fluid_ctx* c, fluid_display_state* d)
{
    if (d->dither.p) return FLUID_OK;
    // createTextureAsync's placeholder (script.js:1135): one white texel until the image has loaded
    const float one = 1.0f;
    CK(ensure(c, d->dither, 1, 1, sizeof(float)));
    HIPCK(c, hipMemcpyAsync(d->dither.p, &one, sizeof one, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return FLUID_OK;
}

// dye.read / dye.write as fp32 texels.  With fp16 storage the compositor works on a widened copy of dye.read (exact) and
// scribbles the sunrays mask on a scratch buffer instead of dye.write; its own intermediates (bloom pyramid, sunrays) stay
// fp32 in both modes.
int dye_texels(fluid_ctx* c, fluid_display_state* d, const float4** read, float4** write)
{
    CK(fluid_impl::ensure_rgba(c));            // the compositor reads RGBA texels (a packed dye field is unpacked here)
    const bool dense = c->dye.P == c->dye.W;   // the compositor's kernels take dense W x H images
    if (c->storage == FLUID_STORE_F32 && dense) {
        *read = (const float4*)c->dyeb[0];
        *write = (float4*)c->dyeb[1];
        return FLUID_OK;
    }
    for (auto& b : d->dye32) CK(ensure(c, b, c->dye.P, c->dye.H, sizeof(float4)));
    const float4* src = (const float4*)c->dyeb[0];
    if (c->storage != FLUID_STORE_F32) {   // widen (exact) into the second scratch image, pitch and all
        HIPCK(c, launch_widen(c->stream, (const __half*)c->dyeb[0], (float*)d->dye32[1].p, (size_t)c->dye.P * c->dye.H * 4));
        src = (const float4*)d->dye32[1].p;
        if (dense) {
            *read = src;
            *write = (float4*)d->dye32[0].p;
            return FLUID_OK;
        }
    }
    // a width that is not a multiple of 4 keeps padding columns in every row (pitch > width): compact it
    HIPCK(c, hipMemcpy2DAsync(d->dye32[0].p, (size_t)c->dye.W * sizeof(float4), src, (size_t)c->dye.P * sizeof(float4), (size_t)c->dye.W * sizeof(float4),
                              (size_t)c->dye.H, hipMemcpyDeviceToDevice, c->stream));
    *read = (const float4*)d->dye32[0].p;
    *write = (float4*)d->dye32[1].p;   // the sunrays mask goes to scratch (the widened copy, if any, is consumed by now)
    return FLUID_OK;
}

// applyBloom(dye.read, bloom), script.js:1346-1389
int apply_bloom(fluid_ctx* c, fluid_display_state* d, const fluid_display_params* P, const float4* dye)
{
    std::vector<std::pair<int, int>> sizes;
    for (int i = 0; i < P->bloom_iterations; i++) {  // initBloomFramebuffers, script.js:1012-1032
        const int w = P->bloom_w >> (i + 1), h = P->bloom_h >> (i + 1);
        if (w < 2 || h < 2) break;
        sizes.push_back({ w, h });
    }
    const bool fresh = !d->bloom.p || d->bloom.w != P->bloom_w || d->bloom.h != P->bloom_h;
    CK(ensure(c, d->bloom, P->bloom_w, P->bloom_h, sizeof(float4)));
    if (fresh) {  // a new FBO is cleared to (0, 0, 0, 1) (script.js:1059, 136)
        HIPCK(c, launch_fill(c->stream, (float*)d->bloom.p, (size_t)P->bloom_w * P->bloom_h, 4, 0.f, 0.f, 0.f, 1.f));
    }
    if (sizes.size() < 2) return FLUID_OK;  // script.js:1347-1348: bloom keeps whatever it held
    d->levels.resize(sizes.size());
    for (size_t i = 0; i < sizes.size(); i++) CK(ensure(c, d->levels[i], sizes[i].first, sizes[i].second, sizeof(float4)));

    // the uniforms are computed in JS doubles and narrowed by gl.uniform3f / uniform1f
    const double knee = P->bloom_threshold * P->bloom_soft_knee + 0.0001;
    const float c0 = (float)(P->bloom_threshold - knee), c1 = (float)(knee * 2), c2 = (float)(0.25 / knee);
    HIPCK(c, launch_bloom_prefilter(c->stream, dye, c->dye.W, c->dye.H, (float4*)d->bloom.p, d->bloom.w, d->bloom.h, c0, c1, c2,
                                    (float)P->bloom_threshold));
    const fluid_display_state::Buf* last = &d->bloom;
    for (auto& lv : d->levels) {
        HIPCK(c, launch_box4(c->stream, (const float4*)last->p, last->w, last->h, (float4*)lv.p, lv.w, lv.h, 0, 0, 1.0f));
        last = &lv;
    }
    for (int i = (int)d->levels.size() - 2; i >= 0; i--) {  // blendFunc(ONE, ONE)
        auto& base = d->levels[i];
        HIPCK(c, launch_box4(c->stream, (const float4*)last->p, last->w, last->h, (float4*)base.p, base.w, base.h, 1, 0, 1.0f));
        last = &base;
    }
    HIPCK(c, launch_box4(c->stream, (const float4*)last->p, last->w, last->h, (float4*)d->bloom.p, d->bloom.w, d->bloom.h, 0, 1,
                         (float)P->bloom_intensity));
    return FLUID_OK;
}

// applySunrays(dye.read, dye.write, sunrays); blur(sunrays, sunraysTemp, 1) — script.js:1391-1419
int apply_sunrays(fluid_ctx* c, fluid_display_state* d, const fluid_display_params* P, const float4* dye, float4* dye_write)
{
    CK(ensure(c, d->sunrays, P->sunrays_w, P->sunrays_h, sizeof(float)));
    CK(ensure(c, d->sunrays_tmp, P->sunrays_w, P->sunrays_h, sizeof(float)));
    const size_t n = (size_t)c->dye.W * c->dye.H;
    HIPCK(c, launch_sunrays_mask(c->stream, dye, dye_write, n));  // the reference scribbles the mask on dye.write too
    HIPCK(c, launch_sunrays(c->stream, dye_write, c->dye.W, c->dye.H, (float*)d->sunrays.p, d->sunrays.w, d->sunrays.h, (float)P->sunrays_weight));
    HIPCK(c, launch_blur3(c->stream, (const float*)d->sunrays.p, (float*)d->sunrays_tmp.p, d->sunrays.w, d->sunrays.h, 1));
    HIPCK(c, launch_blur3(c->stream, (const float*)d->sunrays_tmp.p, (float*)d->sunrays.p, d->sunrays.w, d->sunrays.h, 0));
    return FLUID_OK;
}

}  // namespace

namespace fluid_impl {

void display_release(fluid_ctx* c)
{
    fluid_display_state* d = c->display;
    if (!d) return;
    for (auto* b : { &d->bloom, &d->sunrays, &d->sunrays_tmp, &d->frame, &d->frame8, &d->dither, &d->dye32[0], &d->dye32[1] })
        if (b->p) (void)hipFree(b->p);
    for (auto& b : d->levels)
        if (b.p) (void)hipFree(b.p);
    delete d;
    c->display = nullptr;
}

}  // namespace fluid_impl

extern "C" {

int fluid_set_dither(fluid_ctx* c, const float* host_r, int w, int h)
{
    if (!c || !host_r || w < 1 || h < 1) return FLUID_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->device));
    fluid_display_state* d = state(c);
    if (!d) return c->fail(FLUID_ERR_OOM, "out of host memory");
    CK(ensure(c, d->dither, w, h, sizeof(float)));
    HIPCK(c, hipMemcpyAsync(d->dither.p, host_r, (size_t)w * h * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return FLUID_OK;
}

int fluid_render(fluid_ctx* c, int width, int height, const fluid_display_params* P)
{
    if (!c || !P) return FLUID_ERR_INVALID;
    if (c->desc.parts != 1 || c->desc.parts_x != 1) return c->fail(FLUID_ERR_UNSUPPORTED, "render of a stripe / tile context");
    if (width < 1 || height < 1) return c->fail(FLUID_ERR_INVALID, "frame size must be >= 1");
    if (P->bloom && (P->bloom_w < 1 || P->bloom_h < 1 || P->bloom_iterations < 0)) return c->fail(FLUID_ERR_INVALID, "bad bloom size");
    if (P->sunrays && (P->sunrays_w < 1 || P->sunrays_h < 1)) return c->fail(FLUID_ERR_INVALID, "bad sunrays size");
    HIPCK(c, hipSetDevice(c->device));
    fluid_display_state* d = state(c);
    if (!d) return c->fail(FLUID_ERR_OOM, "out of host memory");
    CK(default_dither(c, d));
    const float4* dye = nullptr;
    float4* dye_write = nullptr;
    CK(dye_texels(c, d, &dye, &dye_write));
    if (P->bloom) CK(apply_bloom(c, d, P, dye));
    if (P->sunrays) CK(apply_sunrays(c, d, P, dye, dye_write));
    CK(ensure(c, d->frame, width, height, sizeof(float4)));
    DisplayArgs a{};
    a.dye = dye;
    a.dye_w = c->dye.W;
    a.dye_h = c->dye.H;
    a.bloom = P->bloom ? (const float4*)d->bloom.p : nullptr;
    a.bloom_w = d->bloom.w;
    a.bloom_h = d->bloom.h;
    a.sunrays = P->sunrays ? (const float*)d->sunrays.p : nullptr;
    a.sun_w = d->sunrays.w;
    a.sun_h = d->sunrays.h;
    a.dither = (const float*)d->dither.p;
    a.dither_w = d->dither.w;
    a.dither_h = d->dither.h;
    a.frame = (float4*)d->frame.p;
    a.w = width;
    a.h = height;
    a.shading = P->shading;
    a.transparent = P->transparent;
    a.back_r = P->back_r;
    a.back_g = P->back_g;
    a.back_b = P->back_b;
    HIPCK(c, launch_display(c->stream, a));
    return FLUID_OK;
}

int fluid_read_frame(fluid_ctx* c, float* host_rgba, size_t bytes)
{
    if (!c || !host_rgba) return FLUID_ERR_INVALID;
    fluid_display_state* d = c->display;
    if (!d || !d->frame.p) return c->fail(FLUID_ERR_INVALID, "no frame rendered yet");
    if (bytes != (size_t)d->frame.w * d->frame.h * sizeof(float4)) return c->fail(FLUID_ERR_INVALID, "read_frame: byte count does not match the frame");
    HIPCK(c, hipSetDevice(c->device));
    HIPCK(c, hipMemcpyAsync(host_rgba, d->frame.p, bytes, hipMemcpyDeviceToHost, c->stream));
    return fluid_impl::ctx_sync(c);   // (a frame composited from the dye of a pressure loop that gave up is an error, not pixels)
}

int fluid_read_frame_rgba8(fluid_ctx* c, unsigned char* host, size_t bytes)
{
    if (!c || !host) return FLUID_ERR_INVALID;
    fluid_display_state* d = c->display;
    if (!d || !d->frame.p) return c->fail(FLUID_ERR_INVALID, "no frame rendered yet");
    if (bytes != (size_t)d->frame.w * d->frame.h * 4) return c->fail(FLUID_ERR_INVALID, "read_frame_rgba8: byte count does not match the frame");
    HIPCK(c, hipSetDevice(c->device));
    CK(ensure(c, d->frame8, d->frame.w, d->frame.h, 4));
    HIPCK(c, launch_normalize(c->stream, (const float4*)d->frame.p, (unsigned char*)d->frame8.p, d->frame.w, d->frame.h));
    HIPCK(c, hipMemcpyAsync(host, d->frame8.p, bytes, hipMemcpyDeviceToHost, c->stream));
    return fluid_impl::ctx_sync(c);   // (a frame composited from the dye of a pressure loop that gave up is an error, not pixels)
}

int fluid_read_display_buffer(fluid_ctx* c, int which, float* host, size_t bytes, int* w, int* h)
{
    if (!c) return FLUID_ERR_INVALID;
    fluid_display_state* d = c->display;
    const fluid_display_state::Buf* b = !d ? nullptr : which == FLUID_DISPLAY_BLOOM ? &d->bloom : which == FLUID_DISPLAY_SUNRAYS ? &d->sunrays : nullptr;
    if (!b || !b->p) return c->fail(FLUID_ERR_INVALID, "display buffer not rendered");
    if (w) *w = b->w;
    if (h) *h = b->h;
    if (!host) return FLUID_OK;
    const size_t need = (size_t)b->w * b->h * (which == FLUID_DISPLAY_BLOOM ? sizeof(float4) : sizeof(float));
    if (bytes != need) return c->fail(FLUID_ERR_INVALID, "read_display_buffer: byte count mismatch");
    HIPCK(c, hipSetDevice(c->device));
    HIPCK(c, hipMemcpyAsync(host, b->p, need, hipMemcpyDeviceToHost, c->stream));
    return fluid_impl::ctx_sync(c);   // (a frame composited from the dye of a pressure loop that gave up is an error, not pixels)
}

}  // extern "C"

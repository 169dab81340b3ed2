// fluid_display.hip — gfx950 kernels of the display compositor: the reference's render(target)
// (script.js:1296-1419) re-expressed as HIP kernels — bloom prefilter / blur pyramid / final, sunrays mask / march /
// separable blur, and the display pass (shading, bloom, sunrays, dithering, gamma, back-colour blend).  The first
// consumer of dye.read; not on the step() path.  One thread per target texel, 256-thread blocks along a row; every
// pass is a handful of bilinear fetches (the reference's LINEAR + CLAMP_TO_EDGE textures) — bandwidth-trivial next to
// the simulation, so no tiling.  Arithmetic follows the GLSL line by line (-ffp-contract=off).
#include "fluid_display.h"

namespace fluid {

namespace {

constexpr int BX = 256;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float mixf(float a, float b, float t) { return a + (b - a) * t; }  // form validated against SwiftShader
__device__ __forceinline__ int modi(int a, int n)
{
    const int r = a % n;
    return r < 0 ? r + n : r;
}

struct Tap {
    long a, b, c, d;
    float fx, fy;
};

// texture2D on a LINEAR texture of W x H texels: CLAMP_TO_EDGE (every FBO, script.js:1051-1052) or REPEAT (dithering, 1133-1134)
template <bool REPEAT>
__device__ __forceinline__ Tap taps(int W, int H, float u, float v)
{
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const float fi = floorf(x), fj = floorf(y);
    Tap t;
    t.fx = x - fi;
    t.fy = y - fj;
    const int i0 = (int)fi, j0 = (int)fj;
    int ia, ib, ja, jb;
    if (REPEAT) {
        ia = modi(i0, W); ib = modi(i0 + 1, W); ja = modi(j0, H); jb = modi(j0 + 1, H);
    } else {
        ia = clampi(i0, 0, W - 1); ib = clampi(i0 + 1, 0, W - 1); ja = clampi(j0, 0, H - 1); jb = clampi(j0 + 1, 0, H - 1);
    }
    t.a = (long)ja * W + ia;
    t.b = (long)ja * W + ib;
    t.c = (long)jb * W + ia;
    t.d = (long)jb * W + ib;
    return t;
}

__device__ __forceinline__ float4 tex4(const float4* __restrict__ F, int W, int H, float u, float v)
{
    const Tap t = taps<false>(W, H, u, v);
    const float4 a = F[t.a], b = F[t.b], c = F[t.c], d = F[t.d];
    return make_float4(mixf(mixf(a.x, b.x, t.fx), mixf(c.x, d.x, t.fx), t.fy), mixf(mixf(a.y, b.y, t.fx), mixf(c.y, d.y, t.fx), t.fy),
                       mixf(mixf(a.z, b.z, t.fx), mixf(c.z, d.z, t.fx), t.fy), mixf(mixf(a.w, b.w, t.fx), mixf(c.w, d.w, t.fx), t.fy));
}

template <bool REPEAT>
__device__ __forceinline__ float tex1(const float* __restrict__ F, int W, int H, float u, float v)
{
    const Tap t = taps<REPEAT>(W, H, u, v);
    return mixf(mixf(F[t.a], F[t.b], t.fx), mixf(F[t.c], F[t.d], t.fx), t.fy);
}

__device__ __forceinline__ float alpha_of(const float4* __restrict__ F, int W, int H, float u, float v)  // .a of an RGBA texture
{
    const Tap t = taps<false>(W, H, u, v);
    return mixf(mixf(F[t.a].w, F[t.b].w, t.fx), mixf(F[t.c].w, F[t.d].w, t.fx), t.fy);
}

#define TEXEL(w, h)                                    \
    const int i = blockIdx.x * BX + threadIdx.x;       \
    const int j = blockIdx.y;                          \
    if (i >= (w)) return;                              \
    const float u = ((float)i + 0.5f) / (float)(w);    \
    const float v = ((float)j + 0.5f) / (float)(h)

// bloomPrefilterShader, script.js:614-631
__global__ void __launch_bounds__(BX) k_bloom_prefilter(const float4* __restrict__ dye, int dw, int dh, float4* __restrict__ out, int w, int h,
                                                         float c0, float c1, float c2, float threshold)
{
    TEXEL(w, h);
    const float4 s = tex4(dye, dw, dh, u, v);
    const float br = fmaxf(s.x, fmaxf(s.y, s.z));
    float rq = fminf(fmaxf(br - c0, 0.0f), c1);
    rq = c2 * rq * rq;
    const float k = fmaxf(rq, br - threshold) / fmaxf(br, 0.0001f);
    out[(long)j * w + i] = make_float4(s.x * k, s.y * k, s.z * k, 0.0f);
}

// bloomBlurShader / bloomFinalShader, script.js:633-675: 0.25 * (L + R + T + B), taps one SOURCE texel from vUv;
// `add`: gl.blendFunc(ONE, ONE) of the up-sampling leg (dst = src + dst); `scale`: BLOOM_INTENSITY of the final pass
__global__ void __launch_bounds__(BX) k_box4(const float4* __restrict__ src, int sw, int sh, float4* __restrict__ dst, int w, int h, int add,
                                              int scaled, float scale)
{
    TEXEL(w, h);
    const float tx = 1.0f / (float)sw, ty = 1.0f / (float)sh;
    float4 s = tex4(src, sw, sh, u - tx, v);
    const float4 r = tex4(src, sw, sh, u + tx, v), t = tex4(src, sw, sh, u, v + ty), b = tex4(src, sw, sh, u, v - ty);
    s = make_float4(s.x + r.x, s.y + r.y, s.z + r.z, s.w + r.w);
    s = make_float4(s.x + t.x, s.y + t.y, s.z + t.z, s.w + t.w);
    s = make_float4(s.x + b.x, s.y + b.y, s.z + b.z, s.w + b.w);
    s = make_float4(s.x * 0.25f, s.y * 0.25f, s.z * 0.25f, s.w * 0.25f);
    if (scaled) s = make_float4(s.x * scale, s.y * scale, s.z * scale, s.w * scale);
    const long c = (long)j * w + i;
    if (add) {
        const float4 d = dst[c];
        s = make_float4(s.x + d.x, s.y + d.y, s.z + d.z, s.w + d.w);
    }
    dst[c] = s;
}

// sunraysMaskShader, script.js:677-690 (drawn into dye.write at the dye resolution: vUv is the texel centre)
__global__ void __launch_bounds__(BX) k_sunrays_mask(const float4* __restrict__ dye, float4* __restrict__ mask, size_t n)
{
    for (size_t c = (size_t)blockIdx.x * BX + threadIdx.x; c < n; c += (size_t)gridDim.x * BX) {
        float4 s = dye[c];
        const float br = fmaxf(s.x, fmaxf(s.y, s.z));
        s.w = 1.0f - fminf(fmaxf(br * 20.0f, 0.0f), 0.8f);
        mask[c] = s;
    }
}

// sunraysShader, script.js:692-724: 16 steps toward the centre
__global__ void __launch_bounds__(BX) k_sunrays(const float4* __restrict__ mask, int mw, int mh, float* __restrict__ out, int w, int h, float weight)
{
    TEXEL(w, h);
    const float k = 1.0f / 16.0f * 0.3f;
    const float du = (u - 0.5f) * k, dv = (v - 0.5f) * k;
    float cu = u, cv = v, decay = 1.0f;
    float color = alpha_of(mask, mw, mh, u, v);
    for (int s = 0; s < 16; s++) {
        cu -= du;
        cv -= dv;
        const float col = alpha_of(mask, mw, mh, cu, cv);
        color += col * decay * weight;
        decay *= 0.95f;
    }
    out[(long)j * w + i] = color * 0.7f;
}

// blurVertexShader + blurShader, script.js:460-494: one leg of blur(), script.js:1405-1419
__global__ void __launch_bounds__(BX) k_blur3(const float* __restrict__ src, float* __restrict__ dst, int w, int h, int horizontal)
{
    TEXEL(w, h);
    const float ox = horizontal ? (1.0f / (float)w) * 1.33333333f : 0.0f;
    const float oy = horizontal ? 0.0f : (1.0f / (float)h) * 1.33333333f;
    float s = tex1<false>(src, w, h, u, v) * 0.29411764f;
    s += tex1<false>(src, w, h, u - ox, v - oy) * 0.35294117f;
    s += tex1<false>(src, w, h, u + ox, v + oy) * 0.35294117f;
    dst[(long)j * w + i] = s;
}

__device__ __forceinline__ float length3(float4 c) { return sqrtf(c.x * c.x + c.y * c.y + c.z * c.z); }
__device__ __forceinline__ float gamma1(float c)  // linearToGamma, script.js:566-569
{
    c = fmaxf(c, 0.0f);
    return fmaxf(1.055f * powf(c, 0.416666667f) - 0.055f, 0.0f);
}

// displayShaderSource (script.js:549-612) + drawColor / the blend state of render() (script.js:1305-1316)
__global__ void __launch_bounds__(BX) k_display(DisplayArgs A)
{
    TEXEL(A.w, A.h);
    float4 c = tex4(A.dye, A.dye_w, A.dye_h, u, v);
    if (A.shading) {
        const float tx = 1.0f / (float)A.w, ty = 1.0f / (float)A.h;
        const float4 lc = tex4(A.dye, A.dye_w, A.dye_h, u - tx, v), rc = tex4(A.dye, A.dye_w, A.dye_h, u + tx, v);
        const float4 tc = tex4(A.dye, A.dye_w, A.dye_h, u, v + ty), bc = tex4(A.dye, A.dye_w, A.dye_h, u, v - ty);
        const float dx = length3(rc) - length3(lc);
        const float dy = length3(tc) - length3(bc);
        const float lz = sqrtf(tx * tx + ty * ty);
        const float nz = lz / sqrtf(dx * dx + dy * dy + lz * lz);  // normalize(vec3(dx, dy, length(texelSize))).z = dot(n, l)
        const float diffuse = fminf(fmaxf(nz + 0.7f, 0.7f), 1.0f);
        c.x *= diffuse; c.y *= diffuse; c.z *= diffuse;
    }
    float4 bl = make_float4(0, 0, 0, 0);
    if (A.bloom) bl = tex4(A.bloom, A.bloom_w, A.bloom_h, u, v);
    if (A.sunrays) {
        const float s = tex1<false>(A.sunrays, A.sun_w, A.sun_h, u, v);
        c.x *= s; c.y *= s; c.z *= s;
        if (A.bloom) { bl.x *= s; bl.y *= s; bl.z *= s; }
    }
    if (A.bloom) {
        const float sx = (float)((double)A.w / (double)A.dither_w), sy = (float)((double)A.h / (double)A.dither_h);  // getTextureScale
        float noise = tex1<true>(A.dither, A.dither_w, A.dither_h, u * sx, v * sy);
        noise = noise * 2.0f - 1.0f;
        const float nz = noise / 255.0f;
        c.x += gamma1(bl.x + nz);
        c.y += gamma1(bl.y + nz);
        c.z += gamma1(bl.z + nz);
    }
    const float a = fmaxf(c.x, fmaxf(c.y, c.z));
    float4 o;
    if (A.transparent) {
        o = make_float4(c.x, c.y, c.z, a);
    } else {  // drawColor(BACK_COLOR), then blendFunc(ONE, ONE_MINUS_SRC_ALPHA)
        const float k = 1.0f - a;
        o = make_float4(c.x + A.back_r * k, c.y + A.back_g * k, c.z + A.back_b * k, a + 1.0f * k);
    }
    A.frame[(long)j * A.w + i] = o;
}

// normalizeTexture, script.js:309-323: clamp01(x) * 255 stored into a Uint8Array (truncation), rows flipped
__global__ void __launch_bounds__(BX) k_normalize(const float4* __restrict__ frame, uchar4* __restrict__ out, int w, int h)
{
    const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y;
    if (i >= w) return;
    const float4 s = frame[(long)j * w + i];
    auto q = [](float x) { return (unsigned char)(fmin(fmax((double)x, 0.0), 1.0) * 255.0); };
    out[(long)(h - 1 - j) * w + i] = make_uchar4(q(s.x), q(s.y), q(s.z), q(s.w));
}

inline dim3 grid2(int w, int h) { return dim3((w + BX - 1) / BX, h, 1); }

}  // namespace

hipError_t launch_bloom_prefilter(hipStream_t s, const float4* dye, int dw, int dh, float4* out, int w, int h, float c0, float c1, float c2,
                                  float threshold)
{
    k_bloom_prefilter<<<grid2(w, h), BX, 0, s>>>(dye, dw, dh, out, w, h, c0, c1, c2, threshold);
    return hipGetLastError();
}

hipError_t launch_box4(hipStream_t s, const float4* src, int sw, int sh, float4* dst, int w, int h, int add, int scaled, float scale)
{
    k_box4<<<grid2(w, h), BX, 0, s>>>(src, sw, sh, dst, w, h, add, scaled, scale);
    return hipGetLastError();
}

hipError_t launch_sunrays_mask(hipStream_t s, const float4* dye, float4* mask, size_t n)
{
    const unsigned g = (unsigned)((n + BX - 1) / BX < 8192 ? (n + BX - 1) / BX : 8192);
    k_sunrays_mask<<<g ? g : 1, BX, 0, s>>>(dye, mask, n);
    return hipGetLastError();
}

hipError_t launch_sunrays(hipStream_t s, const float4* mask, int mw, int mh, float* out, int w, int h, float weight)
{
    k_sunrays<<<grid2(w, h), BX, 0, s>>>(mask, mw, mh, out, w, h, weight);
    return hipGetLastError();
}

hipError_t launch_blur3(hipStream_t s, const float* src, float* dst, int w, int h, int horizontal)
{
    k_blur3<<<grid2(w, h), BX, 0, s>>>(src, dst, w, h, horizontal);
    return hipGetLastError();
}

hipError_t launch_display(hipStream_t s, const DisplayArgs& a)
{
    k_display<<<grid2(a.w, a.h), BX, 0, s>>>(a);
    return hipGetLastError();
}

hipError_t launch_normalize(hipStream_t s, const float4* frame, unsigned char* out_rgba8, int w, int h)
{
    k_normalize<<<grid2(w, h), BX, 0, s>>>(frame, reinterpret_cast<uchar4*>(out_rgba8), w, h);
    return hipGetLastError();
}

}  // namespace fluid

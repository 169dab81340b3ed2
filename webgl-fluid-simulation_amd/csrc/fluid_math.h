// fluid_math.h — per-texel device arithmetic shared by the fp32-storage kernels (fluid_kernels.hip) and the
// fp16-storage kernels (fluid_kernels_f16.hip): storage access (ld / st), the GL bilinear fetch, and one `*_texel`
// body per reference pass.  Everything computes in fp32, whatever the storage type; a store to an fp16 field rounds
// to nearest even.  Internal; each function cites the shader lines of the reference (script.js) it follows.
#pragma once
#include "fluid_kernels.h"

namespace fluid {
namespace {

// ---- storage access: a field is an array of fp32 texels or (FLUID_STORE_F16) of half texels; `ld` widens to fp32,
//      `st` narrows with round-to-nearest-even — what a write to a half-float render target does (script.js:138, 145-147)
__device__ __forceinline__ float ld(const float* p, long i) { return p[i]; }
__device__ __forceinline__ float2 ld(const float2* p, long i) { return p[i]; }
__device__ __forceinline__ float4 ld(const float4* p, long i) { return p[i]; }
__device__ __forceinline__ float ld(const __half* p, long i) { return __half2float(p[i]); }
__device__ __forceinline__ float2 ld(const __half2* p, long i) { return __half22float2(p[i]); }
__device__ __forceinline__ float4 ld(const half4* p, long i)
{
    const half4 v = p[i];
    const float2 a = __half22float2(v.lo), b = __half22float2(v.hi);
    return make_float4(a.x, a.y, b.x, b.y);
}
// what a field of that storage keeps of an fp32 value: v itself, or v rounded to fp16
__device__ __forceinline__ float kept(const float*, float v) { return v; }
__device__ __forceinline__ float kept(const float2*, float v) { return v; }
__device__ __forceinline__ float kept(const __half*, float v) { return __half2float(__float2half_rn(v)); }
__device__ __forceinline__ float kept(const __half2*, float v) { return __half2float(__float2half_rn(v)); }
// packed dye texels (fluid_kernels.h rgb3): the alpha lane of what a kernel computes is dropped — the context carries it as a scalar
__device__ __forceinline__ float4 ld(const rgb3* p, long i)
{
    const rgb3 v = p[i];
    return make_float4(v.r, v.g, v.b, 0.0f);
}
__device__ __forceinline__ void st(rgb3* p, long i, float4 v) { p[i] = rgb3{ v.x, v.y, v.z }; }
__device__ __forceinline__ void st(float* p, long i, float v) { p[i] = v; }
__device__ __forceinline__ void st(float2* p, long i, float2 v) { p[i] = v; }
__device__ __forceinline__ void st(float4* p, long i, float4 v) { p[i] = v; }
__device__ __forceinline__ void st(__half* p, long i, float v) { p[i] = __float2half_rn(v); }
__device__ __forceinline__ void st(__half2* p, long i, float2 v) { p[i] = __float22half2_rn(v); }
__device__ __forceinline__ void st(half4* p, long i, float4 v)
{
    half4 h;
    h.lo = __float22half2_rn(make_float2(v.x, v.y));
    h.hi = __float22half2_rn(make_float2(v.z, v.w));
    p[i] = h;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// array index of the clamped global texel (gj, gi)
__device__ __forceinline__ long widx(const Win& w, int gj, int gi)
{
    // CLAMP_TO_EDGE in global coordinates, folded with the array's own column range (wave-uniform bounds; the array range only
    // ever cuts texels no valid output reads: it keeps a launch at the rim of a tile's ghost zone inside the allocation)
    const int lo = max(w.c0, 0), hi = min(w.c0 + w.P, w.W) - 1;
    return (long)(clampi(gj, 0, w.H - 1) - w.g0) * w.P + (clampi(gi, lo, hi) - w.c0);
}
// array index of the global texel (gj, i) of the window (no clamping: the caller's texel)
__device__ __forceinline__ long at(const Win& w, int gj, int i) { return (long)(gj - w.g0) * w.P + (i - w.c0); }

// K2 vorticity confinement, one texel — vorticityShader script.js:835-866
__device__ __forceinline__ float2 vorticity_cell(float L, float R, float T, float B, float C, float2 v, float curl_strength, float dt)
{
    float fx = 0.5f * (fabsf(T) - fabsf(B));
    float fy = 0.5f * (fabsf(R) - fabsf(L));
    const float len = sqrtf(fx * fx + fy * fy) + 0.0001f;
    fx = fx / len;
    fy = fy / len;
    const float s = curl_strength * C;
    fx = fx * s;
    fy = fy * s;
    fy = fy * -1.0f;
    float vx = v.x + fx * dt;
    float vy = v.y + fy * dt;
    vx = fminf(fmaxf(vx, -1000.0f), 1000.0f);
    vy = fminf(fmaxf(vy, -1000.0f), 1000.0f);
    return make_float2(vx, vy);
}

// ------------------------------------------------------------------------------------------------
// GL LINEAR fetch with CLAMP_TO_EDGE (what texture2D does on the LINEAR-filtered velocity / dye
// textures).  mix(a, b, t) = a + (b - a) * t — the form validated against SwiftShader.
struct Taps {
    long a, b, c, d;  // array indices of the four taps
    float fx, fy;
    int miss;  // taps whose row is outside the window's valid rows (stripe ghost rows exhausted)
};

__device__ __forceinline__ Taps bil_taps(const Win& w, float u, float v)
{
    const float x = u * (float)w.W - 0.5f;
    const float y = v * (float)w.H - 0.5f;
    const float fi = floorf(x), fj = floorf(y);
    Taps t;
    t.fx = x - fi;
    t.fy = y - fj;
    const int i0 = (int)fi, j0 = (int)fj;
    const int ia = clampi(i0, 0, w.W - 1), ib = clampi(i0 + 1, 0, w.W - 1);
    const int ja = clampi(j0, 0, w.H - 1), jb = clampi(j0 + 1, 0, w.H - 1);  // CLAMP_TO_EDGE first, in global rows
    t.miss = (ja < w.v0 || ja >= w.v1) + (jb < w.v0 || jb >= w.v1)          // then: is that row fresh in this window?
             + (ia < w.u0 || ia >= w.u1) + (ib < w.u0 || ib >= w.u1);       //       ... and that column (2-D tiles)
    const int la = clampi(ja - w.g0, 0, w.rows - 1), lb = clampi(jb - w.g0, 0, w.rows - 1);  // a miss still loads: keep it inside the array
    const int ka = clampi(ia - w.c0, 0, w.P - 1), kb = clampi(ib - w.c0, 0, w.P - 1);
    t.a = (long)la * w.P + ka;
    t.b = (long)la * w.P + kb;
    t.c = (long)lb * w.P + ka;
    t.d = (long)lb * w.P + kb;
    return t;
}

__device__ __forceinline__ float mixf(float a, float b, float t) { return a + (b - a) * t; }

// ---- csrc/fluid_math.h ----
// x / d for a WAVE-UNIFORM divisor d, with r = 1.0 / (double)d rounded to double (host: udiv_recip): the fp32 quotient, CORRECTLY
// ROUNDED — bit for bit `x / d` — in three full-rate instructions (v_cvt_f64_f32, v_mul_f64, v_cvt_f32_f64) instead of the 10-instruction
// IEEE divide sequence (v_div_scale x 2, v_rcp, 5 fma, v_div_fmas, v_div_fixup: 14 fma-issue-slots on this chip,
// tools/micro/valu_rate2.hip).  Why it is exact: the product carries a relative error <= 2^-52 (r and the multiply, 2^-53 each), while a
// quotient of two 24-bit significands that is not itself a rounding boundary stays >= 2^-49 (relative) away from every boundary
// (midpoint of two neighbouring floats: a 25-bit significand M; |x/d - M| = |x - d M| / d and x - d M is a non-zero multiple of the product
// of their last places), and it never IS a boundary: d M would need >= 26 significant bits unless d is a power of two, where r is exact.
// Results in the subnormal range (a decaying dye does get there, and the reference keeps subnormals): the boundaries are coarser, the
// margin is 2^-48, and an exact tie would need d * (odd) to be an even multiple of the input's last place — impossible for 1 <= d < 2,
// which is what the callers guarantee for the divisors whose quotients can underflow (udiv_decay_ok).  Zeros keep their sign, infinities
// and NaNs propagate (a multiply by a positive finite number).  Checked against `x / d` on 2.4e8 (d, x) pairs of every class and on
// every (i + .5) / W, W <= 2200 and the BASELINE widths (tests/test_div_uniform.py).
__device__ __forceinline__ float div_uniform(float x, double r) { return (float)((double)x * r); }



// exp() as the reference's GL implementation evaluates it (splatShader, script.js:738, calls the GLSL built-in; the algorithm lives in
// the rasteriser the reference runs on here: SwiftShader as bundled with Chromium 88, src/Pipeline/ShaderCore.cpp `exponential()` /
// `exponential2()`): exp(x) = exp2(1.44269504 x), exp2(x) = 2^i * poly5(f), i = round-to-nearest-even(x - 0.5), f = x - i, the integer part
// written into the exponent field, a degree-5 polynomial for 2^f, every step a separately rounded fp32 multiply or add (the build has
// -ffp-contract=off).  With it a splat — and therefore every whole step that starts from splats — is bit-identical to the reference at
// power-of-two grid sizes (tests/test_hip_vs_golden.py); ocml's expf is an ulp away on 40 % of the texels, which the discontinuous
// vorticity force then amplifies.
__device__ __forceinline__ float exp_reference(float x)
{
    float x0 = 1.44269504f * x;
    x0 = fminf(x0, __uint_as_float(0x43010000u));  // 129.0
    x0 = fmaxf(x0, __uint_as_float(0xC2FDFFFFu));  // -126.99999
    const int i = __float2int_rn(x0 - 0.5f);
    const float ii = __uint_as_float((unsigned)(i + 127) << 23);
    const float f = x0 - (float)i;
    float ff = __uint_as_float(0x3AF61905u);       // 1.8775767e-3
    ff = ff * f + __uint_as_float(0x3C134806u);    // 8.9893397e-3
    ff = ff * f + __uint_as_float(0x3D64AA23u);    // 5.5826318e-2
    ff = ff * f + __uint_as_float(0x3E75EAD4u);    // 2.4015361e-1
    ff = ff * f + __uint_as_float(0x3F31727Bu);    // 6.9315308e-1
    ff = ff * f + 1.0f;
    return ii * ff;
}

// K8 splat weight — splatShader script.js:726-744
__device__ __forceinline__ float splat_weight(const Win& w, int i, int gj, float x, float y, float aspect, float radius)
{
    const float u = ((float)i + 0.5f) / (float)w.W;
    const float v = ((float)gj + 0.5f) / (float)w.H;
    const float px = (u - x) * aspect;
    const float py = v - y;
    return exp_reference(-(px * px + py * py) / radius);
}

// ================================================================================================================
// One texel of each reference pass, templated on the storage types of the fields it touches (V2 = velocity texel,
// S1 = scalar texel, D4 = dye texel: float2 / float / float4 or __half2 / __half / half4).  The per-pass kernels of
// both storage modes are thin wrappers around these; all arithmetic is fp32 in the shader's operand order.

// K1 curl — curlShader script.js:814-833
template <class V2, class S1>
__device__ __forceinline__ void curl_texel(const Win& w, const V2* __restrict__ vel, S1* __restrict__ curl, int i, int gj)
{
    const float L = ld(vel, widx(w, gj, i - 1)).y;
    const float R = ld(vel, widx(w, gj, i + 1)).y;
    const float T = ld(vel, widx(w, gj + 1, i)).x;
    const float B = ld(vel, widx(w, gj - 1, i)).x;
    const float vort = R - L - T + B;
    st(curl, at(w, gj, i), 0.5f * vort);
}

// K2 vorticity confinement — vorticityShader script.js:835-866
template <class V2, class S1>
__device__ __forceinline__ void vorticity_texel(const Win& w, const V2* __restrict__ vel, const S1* __restrict__ curl, V2* __restrict__ vel_out,
                                                float curl_strength, float dt, int i, int gj)
{
    const long c = at(w, gj, i);
    const float L = ld(curl, widx(w, gj, i - 1));
    const float R = ld(curl, widx(w, gj, i + 1));
    const float T = ld(curl, widx(w, gj + 1, i));
    const float B = ld(curl, widx(w, gj - 1, i));
    st(vel_out, c, vorticity_cell(L, R, T, B, ld(curl, c), ld(vel, c), curl_strength, dt));
}

// K3 divergence with the reflecting-wall rule — divergenceShader script.js:786-812
template <class V2, class S1>
__device__ __forceinline__ void divergence_texel(const Win& w, const V2* __restrict__ vel, S1* __restrict__ div, int i, int gj)
{
    const long c = at(w, gj, i);
    float L = ld(vel, widx(w, gj, i - 1)).x;
    float R = ld(vel, widx(w, gj, i + 1)).x;
    float T = ld(vel, widx(w, gj + 1, i)).y;
    float B = ld(vel, widx(w, gj - 1, i)).y;
    const float2 C = ld(vel, c);
    if (i == 0) L = -C.x;
    if (i == w.W - 1) R = -C.x;
    if (gj == w.H - 1) T = -C.y;
    if (gj == 0) B = -C.y;
    st(div, c, 0.5f * (R - L + T - B));
}

// K4 clear — clearShader script.js:508-519
template <class S1>
__device__ __forceinline__ void clear_texel(const Win& w, const S1* __restrict__ p, S1* __restrict__ p_out, float value, int i, int gj)
{
    const long c = at(w, gj, i);
    st(p_out, c, value * ld(p, c));
}

// K5 one Jacobi iteration — pressureShader script.js:868-890 (operand order of line 887)
template <class S1>
__device__ __forceinline__ void jacobi_texel(const Win& w, const S1* __restrict__ p, const S1* __restrict__ div, S1* __restrict__ p_out, int i, int gj)
{
    const long c = at(w, gj, i);
    const float L = ld(p, widx(w, gj, i - 1));
    const float R = ld(p, widx(w, gj, i + 1));
    const float T = ld(p, widx(w, gj + 1, i));
    const float B = ld(p, widx(w, gj - 1, i));
    st(p_out, c, (L + R + B + T - ld(div, c)) * 0.25f);
}

// K6 gradient subtract — gradientSubtractShader script.js:892-913
template <class S1, class V2>
__device__ __forceinline__ void gradsub_texel(const Win& w, const S1* __restrict__ p, const V2* __restrict__ vel, V2* __restrict__ vel_out, int i, int gj)
{
    const long c = at(w, gj, i);
    const float L = ld(p, widx(w, gj, i - 1));
    const float R = ld(p, widx(w, gj, i + 1));
    const float T = ld(p, widx(w, gj + 1, i));
    const float B = ld(p, widx(w, gj - 1, i));
    const float2 v = ld(vel, c);
    st(vel_out, c, make_float2(v.x - (R - L), v.y - (T - B)));
}

// GL LINEAR fetch with CLAMP_TO_EDGE of a two- / four-channel field (the filter runs in fp32 on the widened taps)
template <class V2>
__device__ __forceinline__ float2 bil2(const Win& w, const V2* __restrict__ F, float u, float v, int& miss)
{
    const Taps t = bil_taps(w, u, v);
    miss += t.miss;
    const float2 a = ld(F, t.a), b = ld(F, t.b), c = ld(F, t.c), d = ld(F, t.d);
    return make_float2(mixf(mixf(a.x, b.x, t.fx), mixf(c.x, d.x, t.fx), t.fy),
                       mixf(mixf(a.y, b.y, t.fx), mixf(c.y, d.y, t.fx), t.fy));
}

template <class D4>
__device__ __forceinline__ float4 bil4(const Win& w, const D4* __restrict__ F, float u, float v, int& miss)
{
    const Taps t = bil_taps(w, u, v);
    miss += t.miss;
    const float4 a = ld(F, t.a), b = ld(F, t.b), c = ld(F, t.c), d = ld(F, t.d);
    return make_float4(mixf(mixf(a.x, b.x, t.fx), mixf(c.x, d.x, t.fx), t.fy),
                       mixf(mixf(a.y, b.y, t.fx), mixf(c.y, d.y, t.fx), t.fy),
                       mixf(mixf(a.z, b.z, t.fx), mixf(c.z, d.z, t.fx), t.fy),
                       mixf(mixf(a.w, b.w, t.fx), mixf(c.w, d.w, t.fx), t.fy));
}

// K7a velocity self-advection — advectionShader script.js:746-784, call 1275-1285; returns the taps that missed the window
template <class V2>
__device__ __forceinline__ int advect_velocity_texel(const Win& w, const V2* __restrict__ vel, V2* __restrict__ out, float dt, float dissipation,
                                                     float tsx, float tsy, int i, int gj)
{
    const float u = ((float)i + 0.5f) / (float)w.W;
    const float v = ((float)gj + 0.5f) / (float)w.H;
    const long c = at(w, gj, i);
    const float2 vv = ld(vel, c);
    const float cu = u - dt * vv.x * tsx;
    const float cv = v - dt * vv.y * tsy;
    int miss = 0;
    const float2 r = bil2(w, vel, cu, cv, miss);
    const float decay = 1.0f + dissipation * dt;
    st(out, c, make_float2(r.x / decay, r.y / decay));
    return miss;
}

// K7b dye advection — same program, call script.js:1287-1293: the back-trace uses the SIM texel size (1276)
template <bool SAME_RES, class V2, class D4>
__device__ __forceinline__ int advect_dye_texel(const Win& vw, const V2* __restrict__ vel, const Win& dw, const D4* __restrict__ dye,
                                                D4* __restrict__ out, float dt, float dissipation, float tsx, float tsy, int i, int gj)
{
    const float u = ((float)i + 0.5f) / (float)dw.W;
    const float v = ((float)gj + 0.5f) / (float)dw.H;
    int miss = 0;
    float2 vv;
    if (SAME_RES) vv = ld(vel, at(vw, gj, i));
    else vv = bil2(vw, vel, u, v, miss);
    const float cu = u - dt * vv.x * tsx;
    const float cv = v - dt * vv.y * tsy;
    const float4 r = bil4(dw, dye, cu, cv, miss);
    const float decay = 1.0f + dissipation * dt;
    st(out, at(dw, gj, i), make_float4(r.x / decay, r.y / decay, r.z / decay, r.w / decay));
    return miss;
}

// K8 splat — splatShader script.js:726-744 (the dye target's alpha is forced to 1.0, line 742)
template <class V2>
__device__ __forceinline__ void splat_velocity_texel(const Win& w, const V2* __restrict__ base, V2* __restrict__ out, float x, float y,
                                                     float aspect, float radius, float c0, float c1, int i, int gj)
{
    const long c = at(w, gj, i);
    const float g = splat_weight(w, i, gj, x, y, aspect, radius);
    const float2 b = ld(base, c);
    st(out, c, make_float2(b.x + g * c0, b.y + g * c1));
}

template <class D4>
__device__ __forceinline__ void splat_dye_texel(const Win& w, const D4* __restrict__ base, D4* __restrict__ out, float x, float y, float aspect,
                                                float radius, float c0, float c1, float c2, int i, int gj)
{
    const long c = at(w, gj, i);
    const float g = splat_weight(w, i, gj, x, y, aspect, radius);
    const float4 b = ld(base, c);
    st(out, c, make_float4(b.x + g * c0, b.y + g * c1, b.z + g * c2, 1.0f));
}

// copyShader resample for resizeFBO — script.js:496-506, 1108-1114 (T = float or __half, NC channels per texel)
template <int NC, class T>
__device__ __forceinline__ void resample_texel(const Win& sw, const T* __restrict__ src, const Win& dw, T* __restrict__ dst, int i, int gj)
{
    const float u = ((float)i + 0.5f) / (float)dw.W;
    const float v = ((float)gj + 0.5f) / (float)dw.H;
    const Taps t = bil_taps(sw, u, v);
    for (int k = 0; k < NC; k++) {
        const float a = ld(src, t.a * NC + k), b = ld(src, t.b * NC + k), c = ld(src, t.c * NC + k), d = ld(src, t.d * NC + k);
        st(dst, at(dw, gj, i) * NC + k, mixf(mixf(a, b, t.fx), mixf(c, d, t.fx), t.fy));
    }
}

}  // namespace
}  // namespace fluid

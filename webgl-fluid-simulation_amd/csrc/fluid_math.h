// fluid_math.h — per-texel device arithmetic shared by the fp32-storage kernels (fluid_kernels.hip) and the
// fp16-storage kernels (fluid_kernels_f16.hip): everything here computes in fp32, whatever the storage type.
// Internal; each function cites the shader lines of the reference (script.js) it follows.
#pragma once
#include "fluid_kernels.h"

namespace fluid {
namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// array index of the clamped global texel (gj, gi)
__device__ __forceinline__ long widx(const Win& w, int gj, int gi)
{
    return (long)(clampi(gj, 0, w.H - 1) - w.g0) * w.W + clampi(gi, 0, w.W - 1);
}

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
    const int la = clampi(ja - w.g0, 0, w.rows - 1), lb = clampi(jb - w.g0, 0, w.rows - 1);
    t.a = (long)la * w.W + ia;
    t.b = (long)la * w.W + ib;
    t.c = (long)lb * w.W + ia;
    t.d = (long)lb * w.W + ib;
    return t;
}

__device__ __forceinline__ float mixf(float a, float b, float t) { return a + (b - a) * t; }

// K8 splat weight — splatShader script.js:726-744
__device__ __forceinline__ float splat_weight(const Win& w, int i, int gj, float x, float y, float aspect, float radius)
{
    const float u = ((float)i + 0.5f) / (float)w.W;
    const float v = ((float)gj + 0.5f) / (float)w.H;
    const float px = (u - x) * aspect;
    const float py = v - y;
    return expf(-(px * px + py * py) / radius);
}

}  // namespace
}  // namespace fluid

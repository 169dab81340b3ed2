// fluid_kernels.hip — gfx950 (MI355X / CDNA4) kernels for the stable-fluids hot path.
//
// Every kernel re-expresses one GLSL fragment program of the reference
// (PavelDoGreat/WebGL-Fluid-Simulation, script.js) as a coalesced fp32 stencil over the field
// arrays; the arithmetic (operand order, no FMA contraction: built with -ffp-contract=off)
// follows the shader source line by line so results match the reference pipeline bit for bit
// wherever the reference itself is bit-reproducible.
//
// Two families:
//   * one kernel per reference pass ("PASSES" schedule) — also the per-pass test entry points;
//   * fused / temporally blocked kernels ("FUSED" schedule) that produce the same bits with
//     fewer trips through HBM.  The path is ~0.5 flop/byte: HBM-bound, so no MFMA anywhere.
//
// Conventions: F[j][i], row j = 0 is the bottom row; texel centre uv = ((i+.5)/W, (j+.5)/H);
// CLAMP_TO_EDGE everywhere (script.js:1051-1052).  `Win` = window of a stripe-decomposed field.
#include "fluid_kernels.h"
#include "fluid_math.h"
#include "fluid_tiles.h"
#include "fluid_pchain.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace fluid {

namespace {

constexpr int BX = 256;  // threads per block for the per-pass kernels: 4 waves along a row

// ------------------------------------------------------------------------------------------------
// One kernel per reference pass: thread (i, gj) is one texel, the arithmetic is the pass's `*_texel` body of
// fluid_math.h (shared with the fp16-storage kernels of fluid_kernels_f16.hip).
#define TEXEL_OR_RETURN(win)                                  \
    const int i = (win).x0 + blockIdx.x * BX + threadIdx.x; \
    const int gj = ga + blockIdx.y;                           \
    if (i >= (win).x1) return

__global__ void __launch_bounds__(BX) k_curl(Win w, const float2* __restrict__ vel, float* __restrict__ curl, int ga)
{
    TEXEL_OR_RETURN(w);
    curl_texel(w, vel, curl, i, gj);
}

__global__ void __launch_bounds__(BX) k_vorticity(Win w, const float2* __restrict__ vel, const float* __restrict__ curl,
                                                   float2* __restrict__ vel_out, float curl_strength, float dt, int ga)
{
    TEXEL_OR_RETURN(w);
    vorticity_texel(w, vel, curl, vel_out, curl_strength, dt, i, gj);
}

__global__ void __launch_bounds__(BX) k_divergence(Win w, const float2* __restrict__ vel, float* __restrict__ div, int ga)
{
    TEXEL_OR_RETURN(w);
    divergence_texel(w, vel, div, i, gj);
}

__global__ void __launch_bounds__(BX) k_clear(Win w, const float* __restrict__ p, float* __restrict__ p_out, float value, int ga)
{
    TEXEL_OR_RETURN(w);
    clear_texel(w, p, p_out, value, i, gj);
}

__global__ void __launch_bounds__(BX) k_jacobi(Win w, const float* __restrict__ p, const float* __restrict__ div,
                                                float* __restrict__ p_out, int ga)
{
    TEXEL_OR_RETURN(w);
    jacobi_texel(w, p, div, p_out, i, gj);
}

__global__ void __launch_bounds__(BX) k_gradsub(Win w, const float* __restrict__ p, const float2* __restrict__ vel,
                                                 float2* __restrict__ vel_out, int ga)
{
    TEXEL_OR_RETURN(w);
    gradsub_texel(w, p, vel, vel_out, i, gj);
}

__global__ void __launch_bounds__(BX) k_advect_velocity(Win w, const float2* __restrict__ vel, float2* __restrict__ out,
                                                         float dt, float dissipation, float tsx, float tsy, int ga,
                                                         unsigned int* __restrict__ miss_out)
{
    TEXEL_OR_RETURN(w);
    const int miss = advect_velocity_texel(w, vel, out, dt, dissipation, tsx, tsy, i, gj);
    if (miss) atomicAdd(miss_out, (unsigned)miss);
}

template <bool SAME_RES>
__global__ void __launch_bounds__(BX) k_advect_dye(Win vw, const float2* __restrict__ vel, Win dw, const float4* __restrict__ dye,
                                                    float4* __restrict__ out, float dt, float dissipation, float tsx, float tsy,
                                                    int ga, unsigned int* __restrict__ miss_out)
{
    TEXEL_OR_RETURN(dw);
    const int miss = advect_dye_texel<SAME_RES>(vw, vel, dw, dye, out, dt, dissipation, tsx, tsy, i, gj);
    if (miss) atomicAdd(miss_out, (unsigned)miss);
}

// K7a + K7b fused (dye grid == sim grid): the dye back-trace needs the NEW velocity only at its own texel
// (script.js:1289: uVelocity sampled at vUv), so one thread advects the velocity texel, stores it, and
// advects the dye texel with it — the new velocity is not re-read from HBM (48 B/texel instead of 56).
// Same per-texel arithmetic as the two kernels above, hence the same bits.
// ROWS texels per thread (consecutive rows, same column), processed stage by stage so that the ROWS independent
// gathers of a stage are all in flight together (velocity -> 4 velocity taps -> 4 dye taps; two texels per thread measure 6 % faster
// than one).  advect_both_body is the GENERAL form (any array size, any decay); k_advect_both_fast below is what normally runs.
struct Fetch2 {
    float2 a, b, c, d;
    float fx, fy;
};
struct Fetch4 {
    float4 a, b, c, d;
    float fx, fy;
};

template <int ROWS, class V2, class D4>
__device__ __forceinline__ void advect_both_body(const Win& w, const V2* __restrict__ vel, V2* __restrict__ vel_out, const D4* __restrict__ dye,
                                                 D4* __restrict__ dye_out, float dt, float vel_dissipation, float dye_dissipation, float tsx,
                                                 float tsy, int ga, int gb, unsigned int* __restrict__ miss_out)
{
    const int i = w.x0 + blockIdx.x * BX + threadIdx.x;
    const int gj0 = ga + blockIdx.y * ROWS;
    if (i >= w.x1) return;
    const float u = ((float)i + 0.5f) / (float)w.W;
    const float vdecay = 1.0f + vel_dissipation * dt, ddecay = 1.0f + dye_dissipation * dt;
    int miss = 0;
    bool on[ROWS];
    long c[ROWS];
    float v[ROWS];
    float2 vv[ROWS], nv[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const int gj = gj0 + k;
        on[k] = gj < gb;
        const int gjc = on[k] ? gj : gb - 1;  // a row past the band repeats the last one (loads stay in bounds, nothing stored)
        v[k] = ((float)gjc + 0.5f) / (float)w.H;
        c[k] = at(w, gjc, i);
        vv[k] = ld(vel, c[k]);
    }
    Fetch2 f2[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const Taps t = bil_taps(w, u - dt * vv[k].x * tsx, v[k] - dt * vv[k].y * tsy);
        if (on[k]) miss += t.miss;
        f2[k].a = ld(vel, t.a); f2[k].b = ld(vel, t.b); f2[k].c = ld(vel, t.c); f2[k].d = ld(vel, t.d);
        f2[k].fx = t.fx; f2[k].fy = t.fy;
    }
    Fetch4 f4[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const Fetch2& f = f2[k];
        const float rx = mixf(mixf(f.a.x, f.b.x, f.fx), mixf(f.c.x, f.d.x, f.fx), f.fy);
        const float ry = mixf(mixf(f.a.y, f.b.y, f.fx), mixf(f.c.y, f.d.y, f.fx), f.fy);
        // the dye pass reads the velocity TEXTURE the velocity pass wrote: with fp16 storage that is the rounded value
        nv[k] = make_float2(kept(vel_out, rx / vdecay), kept(vel_out, ry / vdecay));
        if (on[k]) st(vel_out, c[k], nv[k]);
        const Taps t = bil_taps(w, u - dt * nv[k].x * tsx, v[k] - dt * nv[k].y * tsy);
        if (on[k]) miss += t.miss;
        f4[k].a = ld(dye, t.a); f4[k].b = ld(dye, t.b); f4[k].c = ld(dye, t.c); f4[k].d = ld(dye, t.d);
        f4[k].fx = t.fx; f4[k].fy = t.fy;
    }
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const Fetch4& f = f4[k];
        const float4 d = make_float4(mixf(mixf(f.a.x, f.b.x, f.fx), mixf(f.c.x, f.d.x, f.fx), f.fy),
                                     mixf(mixf(f.a.y, f.b.y, f.fx), mixf(f.c.y, f.d.y, f.fx), f.fy),
                                     mixf(mixf(f.a.z, f.b.z, f.fx), mixf(f.c.z, f.d.z, f.fx), f.fy),
                                     mixf(mixf(f.a.w, f.b.w, f.fx), mixf(f.c.w, f.d.w, f.fx), f.fy));
        if (on[k]) st(dye_out, c[k], make_float4(d.x / ddecay, d.y / ddecay, d.z / ddecay, d.w / ddecay));
    }
    if (miss) atomicAdd(miss_out, (unsigned)miss);
}

// ---- csrc/fluid_kernels.hip ----
// ---- the same kernel with everything that is not the reference's arithmetic taken off the VALU, and fewer tap loads ----
// advect_both_body spends 540 VALU instructions per two texels, of which the reference's own arithmetic (back-trace, filter weights,
// mixes) is a fifth: 15 IEEE divides (150 instructions) although every divisor is wave-uniform (W, H, the two decays) -> div_uniform (3
// each, bit-exact); CLAMP_TO_EDGE, array-range and freshness clamps on every tap of every fetch (~25 per fetch) although all but the border
// waves are interior -> one unsigned range test per axis on the first tap and a branch (the clamped path stays, per lane, for the texels
// that need it); 64-bit tap addresses (~14 per fetch) -> 32-bit byte offsets from one base (saddr + voffset loads).  259 instructions.
// Beyond its HBM bytes the kernel pays for the volume of its gathers through the vector L1 (nine loads, 104 B requested per texel for
// 24 B of new data: timing probes in profiles/r02/advect_experiments.txt), so the two horizontally adjacent velocity taps of a row —
// 16 contiguous bytes wherever no clamp intervenes — come in ONE load (gather_taps).  Measured inside the step at 4096^2, interleaved
// A/B: 170.0 us (general, two texels per thread) -> 159.2 (VALU) -> 157.1 (+ paired velocity taps) -> 145.5 us with four texels per
// thread, which the lighter kernel can afford (98 VGPRs); 1880 -> 1950 steps/s.
// Launched when the dye array fits 32-bit byte offsets (<= 4 GiB: holds up to BASELINE's 16384^2) and both decay divisors are in [1, 2);
// otherwise the general kernel.  Same fp32 operations in the same order on the same values, hence the same bits (tests: fused == per-pass
// kernels — which divide and clamp the plain way — array_equal on every shape, stripes and tiles with misses included).
struct TapBox {
    int xlo, ylo;
    unsigned nx, ny;
};
// the first taps (i0, j0) of a fetch whose four taps all lie inside the domain (no CLAMP_TO_EDGE), inside the array and inside the part of
// the window that is fresh (no miss)
__device__ __forceinline__ TapBox tap_box(const Win& w)
{
    const int xlo = max(max(w.u0, w.c0), 0), xhi = min(min(w.u1, w.c0 + w.P), w.W) - 2;
    const int ylo = max(max(w.v0, w.g0), 0), yhi = min(min(w.v1, w.g0 + w.rows), w.H) - 2;
    return TapBox{ xlo, ylo, (unsigned)max(xhi - xlo + 1, 0), (unsigned)max(yhi - ylo + 1, 0) };
}
struct Tap4 {
    unsigned a, b, c, d;  // BYTE offsets of the four taps from the field's base pointer
    float fx, fy;
    int miss;
};
// SZ = bytes per texel of the field fetched from
template <unsigned SZ>
__device__ __forceinline__ Tap4 taps32(const Win& w, const TapBox& B, float u, float v)
{
    const float x = u * (float)w.W - 0.5f;
    const float y = v * (float)w.H - 0.5f;
    const float fi = floorf(x), fj = floorf(y);
    Tap4 t;
    t.fx = x - fi;
    t.fy = y - fj;
    const int i0 = (int)fi, j0 = (int)fj;
    if ((unsigned)(i0 - B.xlo) < B.nx && (unsigned)(j0 - B.ylo) < B.ny) {
        // ((j0 - g0) P + (i0 - c0)) SZ with the uniform part on the scalar unit; j0 >= 0 and P SZ < 2^24 here: a 24-bit multiply-add
        const unsigned row_bytes = (unsigned)w.P * SZ;
        const unsigned k = 0u - (unsigned)(w.g0 * w.P + w.c0) * SZ;
        t.a = __umul24((unsigned)j0, row_bytes) + k + (unsigned)i0 * SZ;
        t.b = t.a + SZ;
        t.c = t.a + row_bytes;
        t.d = t.c + SZ;
        t.miss = 0;
    } else {  // bil_taps, fluid_math.h
        const int ia = clampi(i0, 0, w.W - 1), ib = clampi(i0 + 1, 0, w.W - 1);
        const int ja = clampi(j0, 0, w.H - 1), jb = clampi(j0 + 1, 0, w.H - 1);
        t.miss = (ja < w.v0 || ja >= w.v1) + (jb < w.v0 || jb >= w.v1) + (ia < w.u0 || ia >= w.u1) + (ib < w.u0 || ib >= w.u1);
        const int la = clampi(ja - w.g0, 0, w.rows - 1), lb = clampi(jb - w.g0, 0, w.rows - 1);
        const int ka = clampi(ia - w.c0, 0, w.P - 1), kb = clampi(ib - w.c0, 0, w.P - 1);
        t.a = (unsigned)(la * w.P + ka) * SZ;
        t.b = (unsigned)(la * w.P + kb) * SZ;
        t.c = (unsigned)(lb * w.P + ka) * SZ;
        t.d = (unsigned)(lb * w.P + kb) * SZ;
    }
    return t;
}
// taps32 that also says where the first tap is and whether the fetch took the interior path (the tap box of k_advect_dye_fast_rgb_box)
template <unsigned SZ>
__device__ __forceinline__ Tap4 taps32_ij(const Win& w, const TapBox& B, float u, float v, int& i0, int& j0, bool& interior)
{
    const float x = u * (float)w.W - 0.5f;
    const float y = v * (float)w.H - 0.5f;
    i0 = (int)floorf(x);
    j0 = (int)floorf(y);
    interior = (unsigned)(i0 - B.xlo) < B.nx && (unsigned)(j0 - B.ylo) < B.ny;
    return taps32<SZ>(w, B, u, v);   // (the same expressions: the compiler merges them)
}

// the texel at a 32-bit byte offset from the (uniform) base pointer
template <class T>
__device__ __forceinline__ const T* at_byte(const T* p, unsigned o)
{
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(p) + (size_t)o);
}
template <class T>
__device__ __forceinline__ T* at_byte(T* p, unsigned o)
{
    return reinterpret_cast<T*>(reinterpret_cast<char*>(p) + (size_t)o);
}


// two horizontally adjacent fp32 velocity texels in one 16-byte load (8-byte aligned: global loads only need dword alignment)
typedef float pair_f2 __attribute__((ext_vector_type(4), aligned(8)));
__device__ __forceinline__ void load_pair(const float2* F, unsigned o, float2& a, float2& b)
{
    const pair_f2 v = *reinterpret_cast<const pair_f2*>(reinterpret_cast<const char*>(F) + (size_t)o);
    a = make_float2(v.x, v.y);
    b = make_float2(v.z, v.w);
}
// two horizontally adjacent PACKED dye texels — 24 contiguous bytes, dword-aligned — in a 16-byte and an 8-byte load instead of two
// 12-byte ones: the texture addresser spends its cycles per instruction and lane group, not per byte, and a dwordx3 costs what a dwordx4 does
typedef float quad_u4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float pair_u4 __attribute__((ext_vector_type(2), aligned(4)));
[[maybe_unused]] __device__ __forceinline__ void load_pair(const rgb3* F, unsigned o, float4& a, float4& b)
{
    const char* p = reinterpret_cast<const char*>(F) + (size_t)o;
    const quad_u4 lo = *reinterpret_cast<const quad_u4*>(p);
    const pair_u4 hi = *reinterpret_cast<const pair_u4*>(p + 16);
    a = make_float4(lo.x, lo.y, lo.z, 0.0f);
    b = make_float4(lo.w, hi.x, hi.y, 0.0f);
}
// f[k].a .. f[k].d = the texels at byte offsets t[k].a .. t[k].d of field F
// PAIR3: the packed dye's two taps of a row in one 16-byte + one 8-byte load (load_pair above) wherever no clamp separates them
template <int ROWS, bool PAIR3 = false, class T, class FT>
__device__ __forceinline__ void gather_taps(const T* __restrict__ F, const Tap4 (&t)[ROWS], FT (&f)[ROWS])
{
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        bool paired = false;
        if constexpr (PAIR3 && sizeof(T) == sizeof(rgb3) && sizeof(f[k].a) == sizeof(float4)) {   // packed fp32 dye
            paired = t[k].b == t[k].a + 12u && t[k].d == t[k].c + 12u;
            if (paired) {
                load_pair(reinterpret_cast<const rgb3*>(F), t[k].a, f[k].a, f[k].b);
                load_pair(reinterpret_cast<const rgb3*>(F), t[k].c, f[k].c, f[k].d);
            }
        }
        if constexpr (sizeof(T) == sizeof(float2) && sizeof(f[k].a) == sizeof(float2)) {   // fp32 velocity
            paired = t[k].b == t[k].a + 8u && t[k].d == t[k].c + 8u;                       // no clamp between the two taps of a row
            if (paired) {
                load_pair(reinterpret_cast<const float2*>(F), t[k].a, f[k].a, f[k].b);
                load_pair(reinterpret_cast<const float2*>(F), t[k].c, f[k].c, f[k].d);
            }
        }
        if (!paired) {
            f[k].a = ld(at_byte(F, t[k].a), 0); f[k].b = ld(at_byte(F, t[k].b), 0); f[k].c = ld(at_byte(F, t[k].c), 0); f[k].d = ld(at_byte(F, t[k].d), 0);
        }
        f[k].fx = t[k].fx;
        f[k].fy = t[k].fy;
    }
}

// WY: waves of a block stacked in y (1 = the four waves side by side: 256 columns x ROWS rows per block; 4 = one wave wide: 64 columns x
// 4 ROWS rows — the block's waves then share the rows between them in the CU's L1, at the price of more column seams between XCDs)
template <int ROWS, class V2, class D4, int WY = 1, bool PAIR3 = false>
__device__ __forceinline__ void advect_both_fast_body(const Win& w, const V2* __restrict__ vel, V2* __restrict__ vel_out,
                                                      const D4* __restrict__ dye, D4* __restrict__ dye_out, float dt, double rW, double rH,
                                                      double rvd, double rdd, float tsx, float tsy, int ga, int gb,
                                                      unsigned int* __restrict__ miss_out, int bx, int by)
{
    // (bx, by): the block's position in its band — blockIdx for a one-band launch, or its place in one of the launch's rectangles
    // a lane past the last column repeats it (loads stay in bounds, nothing stored, nothing counted), like a row past the band
    constexpr int CW = BX / WY;  // columns per block
    const int tx = WY == 1 ? (int)threadIdx.x : (int)threadIdx.x % CW, ty = WY == 1 ? 0 : (int)threadIdx.x / CW;
    const int lane_i = w.x0 + bx * CW + tx;
    const bool live = lane_i < w.x1;
    const int i = live ? lane_i : w.x1 - 1;
    const int gj0 = ga + (by * WY + ty) * ROWS;
    const TapBox B = tap_box(w);
    const float u = div_uniform((float)i + 0.5f, rW);
    int miss = 0;
    bool on[ROWS];
    unsigned c[ROWS];  // texel index: the velocity and the dye arrays have different texel sizes
    float v[ROWS];
    float2 vv[ROWS], nv[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const int gj = gj0 + k;
        on[k] = live && gj < gb;
        const int gjc = gj < gb ? gj : gb - 1;  // a row past the band repeats the last one
        v[k] = div_uniform((float)gjc + 0.5f, rH);
        c[k] = (unsigned)((gjc - w.g0) * w.P + (i - w.c0));
        vv[k] = ld(at_byte(vel, c[k] * (unsigned)sizeof(V2)), 0);
    }
    Tap4 t[ROWS];
    Fetch2 f2[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        t[k] = taps32<sizeof(V2)>(w, B, u - dt * vv[k].x * tsx, v[k] - dt * vv[k].y * tsy);
        if (on[k]) miss += t[k].miss;
    }
    gather_taps<ROWS>(vel, t, f2);
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const Fetch2& f = f2[k];
        const float rx = mixf(mixf(f.a.x, f.b.x, f.fx), mixf(f.c.x, f.d.x, f.fx), f.fy);
        const float ry = mixf(mixf(f.a.y, f.b.y, f.fx), mixf(f.c.y, f.d.y, f.fx), f.fy);
        // the dye pass reads the velocity TEXTURE the velocity pass wrote: with fp16 storage that is the rounded value
        nv[k] = make_float2(kept(vel_out, div_uniform(rx, rvd)), kept(vel_out, div_uniform(ry, rvd)));
        if (on[k]) st(at_byte(vel_out, c[k] * (unsigned)sizeof(V2)), 0, nv[k]);
        t[k] = taps32<sizeof(D4)>(w, B, u - dt * nv[k].x * tsx, v[k] - dt * nv[k].y * tsy);
        if (on[k]) miss += t[k].miss;
    }
    Fetch4 f4[ROWS];
    gather_taps<ROWS, PAIR3>(dye, t, f4);
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const Fetch4& f = f4[k];
        const float4 d = make_float4(mixf(mixf(f.a.x, f.b.x, f.fx), mixf(f.c.x, f.d.x, f.fx), f.fy),
                                     mixf(mixf(f.a.y, f.b.y, f.fx), mixf(f.c.y, f.d.y, f.fx), f.fy),
                                     mixf(mixf(f.a.z, f.b.z, f.fx), mixf(f.c.z, f.d.z, f.fx), f.fy),
                                     mixf(mixf(f.a.w, f.b.w, f.fx), mixf(f.c.w, f.d.w, f.fx), f.fy));
        if (on[k])
            st(at_byte(dye_out, c[k] * (unsigned)sizeof(D4)), 0,
               make_float4(div_uniform(d.x, rdd), div_uniform(d.y, rdd), div_uniform(d.z, rdd), div_uniform(d.w, rdd)));
    }
    if (miss) atomicAdd(miss_out, (unsigned)miss);
}

// (98 VGPRs at four rows per thread = 4 waves per SIMD; held to 96 for 5 waves it is 1-4 % SLOWER: profiles/r03/advect_occupancy_ab.txt)
template <int ROWS>
__global__ void __launch_bounds__(BX) k_advect_both_fast(Win w, const float2* __restrict__ vel, float2* __restrict__ vel_out,
                                                          const float4* __restrict__ dye, float4* __restrict__ dye_out, float dt, double rW,
                                                          double rH, double rvd, double rdd, float tsx, float tsy, int ga, int gb,
                                                          unsigned int* __restrict__ miss_out)
{
    advect_both_fast_body<ROWS>(w, vel, vel_out, dye, dye_out, dt, rW, rH, rvd, rdd, tsx, tsy, ga, gb, miss_out, (int)blockIdx.x, (int)blockIdx.y);
}

// the same kernel on the PACKED dye field (fluid_kernels.h rgb3: 12-byte texels, the uniform alpha kept as a scalar by the context):
// 40 B/texel instead of 48.  Same body, same arithmetic on the three colour channels, hence the same bits.
template <int ROWS>
__global__ void __launch_bounds__(BX) k_advect_both_fast_rgb(Win w, const float2* __restrict__ vel, float2* __restrict__ vel_out,
                                                              const rgb3* __restrict__ dye, rgb3* __restrict__ dye_out, float dt, double rW,
                                                              double rH, double rvd, double rdd, float tsx, float tsy, int ga, int gb,
                                                              unsigned int* __restrict__ miss_out)
{
    advect_both_fast_body<ROWS>(w, vel, vel_out, dye, dye_out, dt, rW, rH, rvd, rdd, tsx, tsy, ga, gb, miss_out, (int)blockIdx.x, (int)blockIdx.y);
}

__global__ void __launch_bounds__(BX) k_splat_dye_rgb(Win w, const rgb3* __restrict__ base, rgb3* __restrict__ out, float x, float y, float aspect,
                                                       float radius, float c0, float c1, float c2, int ga)
{
    TEXEL_OR_RETURN(w);
    splat_dye_texel(w, base, out, x, y, aspect, radius, c0, c1, c2, i, gj);
}

__global__ void __launch_bounds__(BX) k_dye_pack(const float4* __restrict__ rgba, rgb3* __restrict__ rgb, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * BX + threadIdx.x; i < n; i += (size_t)gridDim.x * BX) {
        const float4 v = rgba[i];
        rgb[i] = rgb3{ v.x, v.y, v.z };
    }
}

__global__ void __launch_bounds__(BX) k_dye_unpack(const rgb3* __restrict__ rgb, float4* __restrict__ rgba, size_t n, float alpha)
{
    for (size_t i = (size_t)blockIdx.x * BX + threadIdx.x; i < n; i += (size_t)gridDim.x * BX) {
        const rgb3 v = rgb[i];
        rgba[i] = make_float4(v.r, v.g, v.b, alpha);
    }
}

#ifdef FLUID_PROBES
// lab (FLUID_RGB_PAIR=1): the packed-dye advection with the two dye taps of a row in a 16-byte + an 8-byte load (gather_taps PAIR3)
template <int ROWS>
__global__ void __launch_bounds__(BX) k_advect_both_fast_rgb_pair(Win w, const float2* __restrict__ vel, float2* __restrict__ vel_out,
                                                                   const rgb3* __restrict__ dye, rgb3* __restrict__ dye_out, float dt, double rW,
                                                                   double rH, double rvd, double rdd, float tsx, float tsy, int ga, int gb,
                                                                   unsigned int* __restrict__ miss_out)
{
    advect_both_fast_body<ROWS, float2, rgb3, 1, true>(w, vel, vel_out, dye, dye_out, dt, rW, rH, rvd, rdd, tsx, tsy, ga, gb, miss_out, (int)blockIdx.x, (int)blockIdx.y);
}

// lab: the packed-dye advection with its block's waves stacked in y (WY) AND the blocks handed to the XCDs in contiguous column ranges
// (block b runs on XCD b % 8: with the plain (bx, by) grid horizontally adjacent blocks never share an L2; here XCD k takes the k-th
// eighth of every block row, so that a tap row displaced across a block seam is refetched at 8 seams per row instead of at every one)
template <int ROWS, int WY>
__global__ void __launch_bounds__(BX) k_advect_both_fast_rgb_wy(Win w, const float2* __restrict__ vel, float2* __restrict__ vel_out,
                                                                 const rgb3* __restrict__ dye, rgb3* __restrict__ dye_out, float dt, double rW,
                                                                 double rH, double rvd, double rdd, float tsx, float tsy, int ga, int gb,
                                                                 unsigned int* __restrict__ miss_out, int gx, int xcd_cols)
{
    // xcd_cols (FLUID_ADVECT_XCD) bit 0: the XCD column ranges; bit 1 (round 6): the block rows from the LAST to the first — the launch in front (the
    // gradient subtract) walked them first to last, so what it wrote last is what the Infinity Cache still holds; bit 2: nothing (this kernel, plain)
    const int b = (int)blockIdx.x, r = b % gx;
    const int by = (xcd_cols & 2) ? (int)gridDim.x / gx - 1 - b / gx : b / gx;
    int bx = r;
    if ((xcd_cols & 1) && (gx & 7) == 0) {
        const int per = gx >> 3;            // blocks of a row per XCD
        bx = (r & 7) * per + (r >> 3);      // consecutive block ids of a row alternate XCDs: give XCD (r & 7) its (r >> 3)-th block
    }
    advect_both_fast_body<ROWS, float2, rgb3, WY>(w, vel, vel_out, dye, dye_out, dt, rW, rH, rvd, rdd, tsx, tsy, ga, gb, miss_out, bx, by);
}

template <int ROWS, int WY>
__global__ void __launch_bounds__(BX) k_advect_both_fast_wy(Win w, const float2* __restrict__ vel, float2* __restrict__ vel_out,
                                                             const float4* __restrict__ dye, float4* __restrict__ dye_out, float dt, double rW,
                                                             double rH, double rvd, double rdd, float tsx, float tsy, int ga, int gb,
                                                             unsigned int* __restrict__ miss_out)
{
    advect_both_fast_body<ROWS, float2, float4, WY>(w, vel, vel_out, dye, dye_out, dt, rW, rH, rvd, rdd, tsx, tsy, ga, gb, miss_out, (int)blockIdx.x, (int)blockIdx.y);
}
#endif

template <int ROWS>
__global__ void __launch_bounds__(BX) k_advect_both_fast_h(Win w, const __half2* __restrict__ vel, __half2* __restrict__ vel_out,
                                                            const half4* __restrict__ dye, half4* __restrict__ dye_out, float dt, double rW,
                                                            double rH, double rvd, double rdd, float tsx, float tsy, int ga, int gb,
                                                            unsigned int* __restrict__ miss_out)
{
    advect_both_fast_body<ROWS>(w, vel, vel_out, dye, dye_out, dt, rW, rH, rvd, rdd, tsx, tsy, ga, gb, miss_out, (int)blockIdx.x, (int)blockIdx.y);
}


// ---- several bands in ONE launch (2-D tiles: the four strips around a tile's interior, fluid_stripes.cpp pass_strips).  A strip is a few
// rows or columns wide: as a launch of its own it is mostly launch latency, and a step had eight of them (decomposition overhead of the
// 2 x 2 tiling: +10 ... 14 % on one GPU, profiles/r02/decomposition_overhead_one_gpu.txt).  Block b finds its rectangle from the prefix sums
// (wave-uniform, scalar unit) and runs the same body with that rectangle's column / row range.
struct AdvRects {
    int n;
    int xa[4], xb[4], ga[4], gb[4], nbx[4], blk0[5];
};

template <int ROWS, class V2, class D4>
__global__ void __launch_bounds__(BX) k_advect_both_fast_rects(Win w, AdvRects R, const V2* __restrict__ vel, V2* __restrict__ vel_out,
                                                                const D4* __restrict__ dye, D4* __restrict__ dye_out, float dt, double rW,
                                                                double rH, double rvd, double rdd, float tsx, float tsy,
                                                                unsigned int* __restrict__ miss_out)
{
    const int b = (int)blockIdx.x;
    int k = 0;
    while (k + 1 < R.n && b >= R.blk0[k + 1]) k++;
    const int local = b - R.blk0[k], nbx = R.nbx[k];
    w.x0 = R.xa[k];
    w.x1 = R.xb[k];
    advect_both_fast_body<ROWS>(w, vel, vel_out, dye, dye_out, dt, rW, rH, rvd, rdd, tsx, tsy, R.ga[k], R.gb[k], miss_out, local % nbx, local / nbx);
}

// ---- dye grid != sim grid (the reference's shipping defaults: SIM_RESOLUTION 128, DYE_RESOLUTION 1024, script.js:60-66): K7a and K7b stay
// two launches (the dye pass samples the NEW velocity bilinearly, script.js:1287-1293), each with the same treatment as the fused kernel:
// uniform divides as double multiplies, the interior tap fast path, 32-bit offsets, paired velocity taps, ROWS texels per thread.  Same
// fp32 operations in the same order as advect_velocity_texel / advect_dye_texel<false>, hence the same bits.
template <int ROWS, class V2>
__device__ __forceinline__ void advect_velocity_fast_body(const Win& w, const V2* __restrict__ vel, V2* __restrict__ vel_out, float dt, double rW,
                                                          double rH, double rvd, float tsx, float tsy, int ga, int gb,
                                                          unsigned int* __restrict__ miss_out)
{
    const int lane_i = w.x0 + blockIdx.x * BX + threadIdx.x;
    const bool live = lane_i < w.x1;
    const int i = live ? lane_i : w.x1 - 1;
    const int gj0 = ga + blockIdx.y * ROWS;
    const TapBox B = tap_box(w);
    const float u = div_uniform((float)i + 0.5f, rW);
    int miss = 0;
    bool on[ROWS];
    unsigned c[ROWS];
    float v[ROWS];
    float2 vv[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const int gj = gj0 + k;
        on[k] = live && gj < gb;
        const int gjc = gj < gb ? gj : gb - 1;
        v[k] = div_uniform((float)gjc + 0.5f, rH);
        c[k] = (unsigned)((gjc - w.g0) * w.P + (i - w.c0));
        vv[k] = ld(at_byte(vel, c[k] * (unsigned)sizeof(V2)), 0);
    }
    Tap4 t[ROWS];
    Fetch2 f2[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        t[k] = taps32<sizeof(V2)>(w, B, u - dt * vv[k].x * tsx, v[k] - dt * vv[k].y * tsy);
        if (on[k]) miss += t[k].miss;
    }
    gather_taps<ROWS>(vel, t, f2);
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const Fetch2& f = f2[k];
        const float rx = mixf(mixf(f.a.x, f.b.x, f.fx), mixf(f.c.x, f.d.x, f.fx), f.fy);
        const float ry = mixf(mixf(f.a.y, f.b.y, f.fx), mixf(f.c.y, f.d.y, f.fx), f.fy);
        if (on[k]) st(at_byte(vel_out, c[k] * (unsigned)sizeof(V2)), 0, make_float2(div_uniform(rx, rvd), div_uniform(ry, rvd)));
    }
    if (miss) atomicAdd(miss_out, (unsigned)miss);
}

// VT (fp32 velocity, WY == 1): the velocity is sampled at the dye texel's own position — a regular lattice, so the taps of a wave's 64
// dye columns x ROWS dye rows are a run of at most 64 / ratio + 2 sim columns in 2 or 3 sim rows (ratio = dye width / sim width: 8 in the
// reference's default configuration, script.js:60-61).  The wave fetches that run ONCE (one 8-byte load per lane, 60 lanes) into its own
// 480 bytes of LDS and every lane reads its four taps from there — two ds_read2_b64 — instead of gathering them through the texture
// addresser with four 16-byte loads per lane and row pair: the dye taps, which follow the flow, are the only gathers left.  A lane whose
// taps are not all inside that run, or touch the domain edge / the window's stale part (taps32's slow path), gathers as before.
// Same texels into the same arithmetic, hence the same bits.
constexpr int VT_COLS = 20, VT_ROWS = 3;

// BOX (packed fp32 dye, lab: FLUID_DYE_BOX=1): the wave's dye taps through LDS.  The dye taps follow the flow, but on a dye grid FINER than
// the sim grid the flow is smooth at the dye's scale: the taps of a wave's 64 columns x ROWS rows fill a box of about (64 + 2) x (ROWS + 2)
// texels.  The wave finds the box (min / max of its lanes' first taps: four 6-step xor-shuffle reductions), fetches its rows with ONE aligned
// 16-byte load per lane and row — 6 loads instead of the 16 twelve-byte gathers the texture addresser works off at 16 lanes per clock
// (TA busy 0.65-0.75 in round 4's kernel: profiles/r04/advect_l1_l2_requests.txt) — and every lane reads its taps from the wave's own 6 KB
// of LDS.  A lane whose taps are not interior (domain edge, stale window: taps32's slow path) gathers as before, as does a whole wave whose
// box does not fit (BOX_ROWS rows x 64 16-byte chunks).  Same texels into the same arithmetic, hence the same bits.
constexpr int BOX_ROWS = 6;
// LDS pointers keep their address space through the call (round 5: as generic pointers the wave's reads of its velocity tile compiled to
// flat_load — 28 of them per thread, every one a texture-addresser instruction, in the kernel whose bound IS the texture addresser)
// (clang's own vector types: HIP's float2 / float4 are classes whose operator= takes a generic `this`)
typedef float lds_v2 __attribute__((ext_vector_type(2)));
typedef float lds_v4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) lds_v2 lds_float2;
typedef __attribute__((address_space(3))) lds_v4 lds_float4;
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ float2 lds_get(const lds_float2* p, int i)
{
    const lds_v2 q = p[i];
    return make_float2(q.x, q.y);
}
[[maybe_unused]] __device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
    return v;
}
[[maybe_unused]] __device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
    return v;
}

template <int ROWS, class V2, class D4, int WY = 1, bool VT = false, bool PAIR3 = false, bool BOX = false>
__device__ __forceinline__ void advect_dye_fast_body(const Win& vw, const V2* __restrict__ vel, const Win& dw, const D4* __restrict__ dye,
                                                     D4* __restrict__ dye_out, float dt, double rW, double rH, double rdd, float tsx, float tsy,
                                                     int ga, int gb, unsigned int* __restrict__ miss_out, lds_float2* vtile = nullptr,
                                                     lds_float4* box = nullptr)
{
    static_assert(!VT || (WY == 1 && sizeof(V2) == sizeof(float2)), "the velocity tile: fp32 fields, waves side by side");
    constexpr int CW = BX / WY;  // columns per block (advect_both_fast_body)
    const int tx = WY == 1 ? (int)threadIdx.x : (int)threadIdx.x % CW, ty = WY == 1 ? 0 : (int)threadIdx.x / CW;
    const int lane_i = dw.x0 + (int)blockIdx.x * CW + tx;
    const bool live = lane_i < dw.x1;
    const int i = live ? lane_i : dw.x1 - 1;
    const int gj0 = ga + ((int)blockIdx.y * WY + ty) * ROWS;
    const TapBox Bv = tap_box(vw), Bd = tap_box(dw);
    const float u = div_uniform((float)i + 0.5f, rW);
    int miss = 0;
    bool on[ROWS];
    unsigned c[ROWS];
    float v[ROWS];
    Tap4 t[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const int gj = gj0 + k;
        on[k] = live && gj < gb;
        const int gjc = gj < gb ? gj : gb - 1;
        v[k] = div_uniform((float)gjc + 0.5f, rH);
        c[k] = (unsigned)((gjc - dw.g0) * dw.P + (i - dw.c0));
        if constexpr (!VT) {
            t[k] = taps32<sizeof(V2)>(vw, Bv, u, v[k]);  // uVelocity sampled at vUv: a LINEAR fetch on the sim grid
            if (on[k]) miss += t[k].miss;
        }
    }
    Fetch2 f2[ROWS];
    if constexpr (VT) {
        // first taps (i0, j0[k]) and weights exactly as taps32 computes them
        const float x = u * (float)vw.W - 0.5f, fi = floorf(x);
        const int i0 = (int)fi;
        int j0[ROWS];
        float fyv[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
            const float y = v[k] * (float)vw.H - 0.5f, fj = floorf(y);
            j0[k] = (int)fj;
            fyv[k] = y - fj;
        }
        // the wave's run of sim texels: origin = the first lane's first tap (columns and rows only grow from there)
        const int ib = __builtin_amdgcn_readfirstlane(i0), jb = __builtin_amdgcn_readfirstlane(j0[0]);
        const int lane = (int)threadIdx.x & 63;
        if (lane < VT_ROWS * VT_COLS) {
            const int r = lane / VT_COLS, q = lane - r * VT_COLS;
            const int lr = clampi(jb + r - vw.g0, 0, vw.rows - 1), lc = clampi(ib + q - vw.c0, 0, vw.P - 1);   // inside the array; a texel outside the tap box is never used
            const float2 tv0 = ld(vel, (size_t)lr * (size_t)vw.P + (size_t)lc);
            vtile[lane] = lds_v2{ tv0.x, tv0.y };
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
            const bool hit = (unsigned)(i0 - Bv.xlo) < Bv.nx && (unsigned)(j0[k] - Bv.ylo) < Bv.ny && (unsigned)(i0 - ib) < (unsigned)(VT_COLS - 1) &&
                             (unsigned)(j0[k] - jb) < (unsigned)(VT_ROWS - 1);
            if (hit) {
                const int o = (j0[k] - jb) * VT_COLS + (i0 - ib);
                f2[k].a = lds_get(vtile, o);
                f2[k].b = lds_get(vtile, o + 1);
                f2[k].c = lds_get(vtile, o + VT_COLS);
                f2[k].d = lds_get(vtile, o + VT_COLS + 1);
            } else {
                const Tap4 tv = taps32<sizeof(V2)>(vw, Bv, u, v[k]);
                if (on[k]) miss += tv.miss;
                f2[k].a = ld(at_byte(vel, tv.a), 0); f2[k].b = ld(at_byte(vel, tv.b), 0); f2[k].c = ld(at_byte(vel, tv.c), 0); f2[k].d = ld(at_byte(vel, tv.d), 0);
            }
            f2[k].fx = x - fi;
            f2[k].fy = fyv[k];
        }
    } else {
        gather_taps<ROWS>(vel, t, f2);
    }
    int di0[ROWS], dj0[ROWS];
    bool dint[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const Fetch2& f = f2[k];
        const float vx = mixf(mixf(f.a.x, f.b.x, f.fx), mixf(f.c.x, f.d.x, f.fx), f.fy);
        const float vy = mixf(mixf(f.a.y, f.b.y, f.fx), mixf(f.c.y, f.d.y, f.fx), f.fy);
        if constexpr (BOX) t[k] = taps32_ij<sizeof(D4)>(dw, Bd, u - dt * vx * tsx, v[k] - dt * vy * tsy, di0[k], dj0[k], dint[k]);
        else t[k] = taps32<sizeof(D4)>(dw, Bd, u - dt * vx * tsx, v[k] - dt * vy * tsy);
        if (on[k]) miss += t[k].miss;
    }
    Fetch4 f4[ROWS];
    bool boxed = false;
    if constexpr (BOX && sizeof(D4) == sizeof(rgb3)) {
        int lo_i = 0x7fffffff, hi_i = -0x7fffffff, lo_j = 0x7fffffff, hi_j = -0x7fffffff;
#pragma unroll
        for (int k = 0; k < ROWS; k++)
            if (dint[k]) {
                lo_i = min(lo_i, di0[k]); hi_i = max(hi_i, di0[k] + 1);
                lo_j = min(lo_j, dj0[k]); hi_j = max(hi_j, dj0[k] + 1);
            }
        lo_i = wave_min(lo_i); hi_i = wave_max(hi_i); lo_j = wave_min(lo_j); hi_j = wave_max(hi_j);
        const unsigned row_bytes = (unsigned)dw.P * 12u;
        const unsigned first16 = ((unsigned)(lo_i - dw.c0) * 12u) & ~15u;               // row pitch and array base are multiples of 16
        const unsigned chunks = hi_i >= lo_i ? (((unsigned)(hi_i - dw.c0) * 12u + 12u - first16 + 15u) >> 4) : 65u;
        const int bh = hi_j - lo_j + 1;
        boxed = hi_i >= lo_i && bh <= BOX_ROWS && chunks <= 64u;                        // wave-uniform
        if (boxed) {
            const int lane = (int)threadIdx.x & 63;
            const char* row0 = reinterpret_cast<const char*>(dye) + (size_t)(unsigned)(lo_j - dw.g0) * row_bytes + first16 + 16u * (unsigned)lane;
#pragma unroll
            for (int r = 0; r < BOX_ROWS; r++)
                if (r < bh && (unsigned)lane < chunks) {
                    const float4 q = *reinterpret_cast<const float4*>(row0 + (size_t)r * row_bytes);
                    box[r * 64 + lane] = lds_v4{ q.x, q.y, q.z, q.w };
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < ROWS; k++) {
                if (dint[k]) {
                    const unsigned o = (unsigned)(dj0[k] - lo_j) * 1024u + ((unsigned)(di0[k] - dw.c0) * 12u - first16);
                    const lds_float* p = reinterpret_cast<const lds_float*>(reinterpret_cast<const lds_char*>(box) + o);
                    f4[k].a = make_float4(p[0], p[1], p[2], 0.0f);
                    f4[k].b = make_float4(p[3], p[4], p[5], 0.0f);
                    f4[k].c = make_float4(p[256], p[257], p[258], 0.0f);
                    f4[k].d = make_float4(p[259], p[260], p[261], 0.0f);
                } else {
                    f4[k].a = ld(at_byte(dye, t[k].a), 0); f4[k].b = ld(at_byte(dye, t[k].b), 0); f4[k].c = ld(at_byte(dye, t[k].c), 0); f4[k].d = ld(at_byte(dye, t[k].d), 0);
                }
                f4[k].fx = t[k].fx;
                f4[k].fy = t[k].fy;
            }
        }
    }
    if (!boxed) gather_taps<ROWS, PAIR3>(dye, t, f4);
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const Fetch4& f = f4[k];
        const float4 d = make_float4(mixf(mixf(f.a.x, f.b.x, f.fx), mixf(f.c.x, f.d.x, f.fx), f.fy),
                                     mixf(mixf(f.a.y, f.b.y, f.fx), mixf(f.c.y, f.d.y, f.fx), f.fy),
                                     mixf(mixf(f.a.z, f.b.z, f.fx), mixf(f.c.z, f.d.z, f.fx), f.fy),
                                     mixf(mixf(f.a.w, f.b.w, f.fx), mixf(f.c.w, f.d.w, f.fx), f.fy));
        if (on[k])
            st(at_byte(dye_out, c[k] * (unsigned)sizeof(D4)), 0,
               make_float4(div_uniform(d.x, rdd), div_uniform(d.y, rdd), div_uniform(d.z, rdd), div_uniform(d.w, rdd)));
    }
    if (miss) atomicAdd(miss_out, (unsigned)miss);
}

template <int ROWS, class V2>
__global__ void __launch_bounds__(BX) k_advect_velocity_fast(Win w, const V2* __restrict__ vel, V2* __restrict__ vel_out, float dt, double rW,
                                                              double rH, double rvd, float tsx, float tsy, int ga, int gb,
                                                              unsigned int* __restrict__ miss_out)
{
    advect_velocity_fast_body<ROWS>(w, vel, vel_out, dt, rW, rH, rvd, tsx, tsy, ga, gb, miss_out);
}

template <int ROWS, class V2, class D4>
__global__ void __launch_bounds__(BX) k_advect_dye_fast(Win vw, const V2* __restrict__ vel, Win dw, const D4* __restrict__ dye,
                                                         D4* __restrict__ dye_out, float dt, double rW, double rH, double rdd, float tsx,
                                                         float tsy, int ga, int gb, unsigned int* __restrict__ miss_out)
{
    advect_dye_fast_body<ROWS>(vw, vel, dw, dye, dye_out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss_out);
}

// ... with the velocity taps from the wave's LDS run (advect_dye_fast_body, VT)
template <int ROWS>
__global__ void __launch_bounds__(BX) k_advect_dye_fast_vt(Win vw, const float2* __restrict__ vel, Win dw, const float4* __restrict__ dye,
                                                            float4* __restrict__ dye_out, float dt, double rW, double rH, double rdd, float tsx,
                                                            float tsy, int ga, int gb, unsigned int* __restrict__ miss_out)
{
    __shared__ float2 vt[BX / 64][VT_ROWS * VT_COLS];
    advect_dye_fast_body<ROWS, float2, float4, 1, true>(vw, vel, dw, dye, dye_out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss_out, (lds_float2*)vt[threadIdx.x >> 6]);
}

#ifdef FLUID_PROBES
template <int ROWS, int WY>
__global__ void __launch_bounds__(BX) k_advect_dye_fast_wy(Win vw, const float2* __restrict__ vel, Win dw, const float4* __restrict__ dye,
                                                            float4* __restrict__ dye_out, float dt, double rW, double rH, double rdd, float tsx,
                                                            float tsy, int ga, int gb, unsigned int* __restrict__ miss_out)
{
    advect_dye_fast_body<ROWS, float2, float4, WY>(vw, vel, dw, dye, dye_out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss_out);
}
#endif

// ... on the PACKED dye field (rgb3; the dye grid differs from the sim grid): 24 instead of 32 B/texel
template <int ROWS>
__global__ void __launch_bounds__(BX) k_advect_dye_fast_rgb(Win vw, const float2* __restrict__ vel, Win dw, const rgb3* __restrict__ dye,
                                                             rgb3* __restrict__ dye_out, float dt, double rW, double rH, double rdd, float tsx,
                                                             float tsy, int ga, int gb, unsigned int* __restrict__ miss_out)
{
    advect_dye_fast_body<ROWS>(vw, vel, dw, dye, dye_out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss_out);
}

template <int ROWS>
__global__ void __launch_bounds__(BX) k_advect_dye_fast_rgb_vt(Win vw, const float2* __restrict__ vel, Win dw, const rgb3* __restrict__ dye,
                                                                rgb3* __restrict__ dye_out, float dt, double rW, double rH, double rdd, float tsx,
                                                                float tsy, int ga, int gb, unsigned int* __restrict__ miss_out)
{
    __shared__ float2 vt[BX / 64][VT_ROWS * VT_COLS];
    advect_dye_fast_body<ROWS, float2, rgb3, 1, true>(vw, vel, dw, dye, dye_out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss_out, (lds_float2*)vt[threadIdx.x >> 6]);
}

#ifdef FLUID_PROBES
template <int ROWS>   // lab (FLUID_DYE_BOX=1): the dye pass on the packed field with the wave's tap box staged through LDS (advect_dye_fast_body BOX)
__global__ void __launch_bounds__(BX) k_advect_dye_fast_rgb_box(Win vw, const float2* __restrict__ vel, Win dw, const rgb3* __restrict__ dye,
                                                                 rgb3* __restrict__ dye_out, float dt, double rW, double rH, double rdd, float tsx,
                                                                 float tsy, int ga, int gb, unsigned int* __restrict__ miss_out)
{
    __shared__ float2 vt[BX / 64][VT_ROWS * VT_COLS];
    __shared__ float4 box[BX / 64][BOX_ROWS][64];
    advect_dye_fast_body<ROWS, float2, rgb3, 1, true, false, true>(vw, vel, dw, dye, dye_out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss_out, (lds_float2*)vt[threadIdx.x >> 6],
                                                                   (lds_float4*)&box[threadIdx.x >> 6][0][0]);
}

template <int ROWS>   // lab (FLUID_RGB_PAIR=1): the dye pass on the packed field with paired tap loads
__global__ void __launch_bounds__(BX) k_advect_dye_fast_rgb_vt_pair(Win vw, const float2* __restrict__ vel, Win dw, const rgb3* __restrict__ dye,
                                                                     rgb3* __restrict__ dye_out, float dt, double rW, double rH, double rdd, float tsx,
                                                                     float tsy, int ga, int gb, unsigned int* __restrict__ miss_out)
{
    __shared__ float2 vt[BX / 64][VT_ROWS * VT_COLS];
    advect_dye_fast_body<ROWS, float2, rgb3, 1, true, true>(vw, vel, dw, dye, dye_out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss_out, (lds_float2*)vt[threadIdx.x >> 6]);
}
#endif

template <int ROWS>
__global__ void __launch_bounds__(BX) k_advect_both(Win w, const float2* __restrict__ vel, float2* __restrict__ vel_out,
                                                     const float4* __restrict__ dye, float4* __restrict__ dye_out, float dt,
                                                     float vel_dissipation, float dye_dissipation, float tsx, float tsy, int ga, int gb,
                                                     unsigned int* __restrict__ miss_out)
{
    advect_both_body<ROWS>(w, vel, vel_out, dye, dye_out, dt, vel_dissipation, dye_dissipation, tsx, tsy, ga, gb, miss_out);
}

template <int ROWS>
__global__ void __launch_bounds__(BX) k_advect_both_h(Win w, const __half2* __restrict__ vel, __half2* __restrict__ vel_out,
                                                       const half4* __restrict__ dye, half4* __restrict__ dye_out, float dt,
                                                       float vel_dissipation, float dye_dissipation, float tsx, float tsy, int ga, int gb,
                                                       unsigned int* __restrict__ miss_out)
{
    advect_both_body<ROWS>(w, vel, vel_out, dye, dye_out, dt, vel_dissipation, dye_dissipation, tsx, tsy, ga, gb, miss_out);
}

__global__ void __launch_bounds__(BX) k_splat_velocity(Win w, const float2* __restrict__ base, float2* __restrict__ out, float x, float y,
                                                        float aspect, float radius, float c0, float c1, int ga)
{
    TEXEL_OR_RETURN(w);
    splat_velocity_texel(w, base, out, x, y, aspect, radius, c0, c1, i, gj);
}

__global__ void __launch_bounds__(BX) k_splat_dye(Win w, const float4* __restrict__ base, float4* __restrict__ out, float x, float y,
                                                   float aspect, float radius, float c0, float c1, float c2, int ga)
{
    TEXEL_OR_RETURN(w);
    splat_dye_texel(w, base, out, x, y, aspect, radius, c0, c1, c2, i, gj);
}

template <int NC>
__global__ void __launch_bounds__(BX) k_resample(Win sw, const float* __restrict__ src, Win dw, float* __restrict__ dst)
{
    const int i = blockIdx.x * BX + threadIdx.x;
    const int gj = blockIdx.y;
    if (i >= dw.W) return;
    resample_texel<NC>(sw, src, dw, dst, i, gj);
}

template <int NC>
__global__ void __launch_bounds__(BX) k_fill(float* __restrict__ dst, size_t n, float v0, float v1, float v2, float v3)
{
    const float vals[4] = { v0, v1, v2, v3 };
    for (size_t i = (size_t)blockIdx.x * BX + threadIdx.x; i < n; i += (size_t)gridDim.x * BX)
        for (int k = 0; k < NC; k++) dst[i * NC + k] = vals[k];
}

// ghost-column blocks of 2-D tiles: strided rectangle -> contiguous staging and back, all rectangles of an exchange in one launch
template <class U>
__device__ __forceinline__ void copy_rect_body(const CopyRect& q)
{
    const size_t n = (size_t)q.line_units * q.nrows;
    for (size_t idx = (size_t)blockIdx.x * BX + threadIdx.x; idx < n; idx += (size_t)gridDim.x * BX) {
        const size_t row = idx / q.line_units, k = idx - row * q.line_units;
        reinterpret_cast<U*>(q.dst + row * q.dpitch)[k] = reinterpret_cast<const U*>(q.src + row * q.spitch)[k];
    }
}

__global__ void __launch_bounds__(BX) k_copy_rects(CopyRects R)
{
    const CopyRect q = R.r[blockIdx.y];
    switch (q.unit) {   // block-uniform
    case 16: copy_rect_body<uint4>(q); break;
    case 8: copy_rect_body<uint2>(q); break;
    case 4: copy_rect_body<unsigned int>(q); break;
    default: copy_rect_body<unsigned short>(q); break;
    }
}

// ------------------------------------------------------------------------------------------------
// Temporally blocked Jacobi (the hot loop: 82 % of the reference's bytes at 50 iterations).
//
// A workgroup of NW waves owns a tile of 256 columns x NW*RY rows, held ENTIRELY IN REGISTERS:
// lane l of wave wv keeps columns 4l..4l+3 (one float4 per row, loaded with one coalesced 1 KiB
// wave transaction) of rows wv*RY..wv*RY+RY-1, for both pressure and divergence.  One iteration:
//   * left/right neighbours come from the adjacent lanes through full-wave DPP shifts
//     (wave_shr:1 / wave_shl:1 — folded into the v_add, no LDS traffic);
//   * top/bottom neighbours are the thread's own registers, except at the wave's first/last row,
//     which the neighbouring waves publish through a double-buffered 2-row LDS mailbox
//     (one barrier per iteration);
//   * rows are updated in place with a one-row delay register, so no second copy of the tile.
// After k iterations the outer k columns/rows of the tile are stale; the tile carries a HALO-deep
// apron so that up to HALO iterations fit in one launch and only the inner (256-2*HALO) x
// (NW*RY-2*HALO) texels are stored.  HBM traffic per launch is ~12 B/texel (+apron re-reads that
// hit L2 / Infinity Cache) for `iters` reference passes instead of 12 B each.
//
// Domain edges: CLAMP_TO_EDGE means an off-domain neighbour equals the centre texel; handled by
// selects in the EDGE instantiation, which only workgroups touching the domain border run.
// (JacobiTB tile geometry, the XCD-aware tile order and the DPP lane shifts: fluid_tiles.h)

// Two texels as one 64-bit register pair: arithmetic on it is v_pk_add_f32 / v_pk_mul_f32 (full rate on gfx950:
// two IEEE fp32 results per lane per instruction — the same bits as two scalar operations).
typedef float v2f __attribute__((ext_vector_type(2)));
// A lane's four consecutive texels (c0 c1 c2 c3) of one row, held as the OUTER pair (c0, c3) and the INNER pair
// (c1, c2).  With this pairing the horizontal sums need no register shuffling:
//   inner: (c0 + c2, c3 + c1) = outer + swap(inner)            one v_pk_add_f32 with op_sel
//   outer: (left + c1, c2 + right)                             two v_add_f32_dpp (lane shift folded in)
// and everything vertical is pair-wise on (outer, inner) of the rows above / below.
struct Quad {
    v2f o, i;
};
__device__ __forceinline__ Quad quad_of(float4 v)
{
    Quad q;
    q.o = v2f{ v.x, v.w };
    q.i = v2f{ v.y, v.z };
    return q;
}
__device__ __forceinline__ float4 float4_of(Quad q) { return make_float4(q.o.x, q.i.x, q.i.y, q.o.y); }
// mailbox rows travel in register order (no permutation on the way through LDS)
__device__ __forceinline__ float4 raw_of(Quad q) { return make_float4(q.o.x, q.o.y, q.i.x, q.i.y); }
__device__ __forceinline__ Quad quad_of_raw(float4 v)
{
    Quad q;
    q.o = v2f{ v.x, v.y };
    q.i = v2f{ v.z, v.w };
    return q;
}

// a lane's four texels to / from a pressure or divergence array: fp32 texels (16-byte access) or, with fp16 storage,
// half texels (8-byte access; widening is exact, the store narrows values that already are halves)
__device__ __forceinline__ Quad load_quad(const float* p, size_t idx) { return quad_of(*reinterpret_cast<const float4*>(p + idx)); }
__device__ __forceinline__ void store_quad(float* p, size_t idx, Quad q) { *reinterpret_cast<float4*>(p + idx) = float4_of(q); }
__device__ __forceinline__ Quad load_quad(const __half* p, size_t idx)
{
    const half4 h = *reinterpret_cast<const half4*>(p + idx);
    const float2 a = __half22float2(h.lo), b = __half22float2(h.hi);
    Quad q;
    q.o = v2f{ a.x, b.y };
    q.i = v2f{ a.y, b.x };
    return q;
}
__device__ __forceinline__ void store_quad(__half* p, size_t idx, Quad q)
{
    half4 h;
    h.lo = __float22half2_rn(make_float2(q.o.x, q.i.x));
    h.hi = __float22half2_rn(make_float2(q.i.y, q.o.y));
    *reinterpret_cast<half4*>(p + idx) = h;
}
// what a half-float render target keeps of two fp32 values (round to nearest even), widened again
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f round_half(v2f v)
{
    return __builtin_convertvector(__builtin_convertvector(v, v2h), v2f);  // fptrunc (RTNE: one v_cvt_pk_f16_f32 on gfx950) + fpext
}

// ------------------------------------------------------------------------------------------------
// Four consecutive texels of a lane, to / from fp32 or fp16 fields (the register-tile kernels below are templated on the
// storage type; the fp32 instantiations are the kernels the headline runs, the half ones serve FLUID_STORE_F16).
// (`kept(field, v)`, fluid_math.h: what a field of that storage keeps of v — applied to every intermediate the reference
// would have written to a texture between two of the fused passes.)

__device__ __forceinline__ float4 load_s4(const float* p, size_t idx) { return *reinterpret_cast<const float4*>(p + idx); }
__device__ __forceinline__ float4 load_s4(const __half* p, size_t idx)
{
    const half4 h = *reinterpret_cast<const half4*>(p + idx);
    const float2 a = __half22float2(h.lo), b = __half22float2(h.hi);
    return make_float4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ void store_s4(float* p, size_t idx, float4 v) { *reinterpret_cast<float4*>(p + idx) = v; }
__device__ __forceinline__ void store_s4(__half* p, size_t idx, float4 v)
{
    half4 h;
    h.lo = __float22half2_rn(make_float2(v.x, v.y));
    h.hi = __float22half2_rn(make_float2(v.z, v.w));
    *reinterpret_cast<half4*>(p + idx) = h;
}
// four velocity texels as two (x0 y0 x1 y1) groups
__device__ __forceinline__ void load_v4(const float2* vel, size_t idx, float4& a, float4& b)
{
    const float4* src = reinterpret_cast<const float4*>(vel + idx);
    a = src[0];
    b = src[1];
}
__device__ __forceinline__ void load_v4(const __half2* vel, size_t idx, float4& a, float4& b)
{
    struct alignas(16) H8 {
        __half2 t[4];
    };
    const H8 h = *reinterpret_cast<const H8*>(vel + idx);
    const float2 t0 = __half22float2(h.t[0]), t1 = __half22float2(h.t[1]), t2 = __half22float2(h.t[2]), t3 = __half22float2(h.t[3]);
    a = make_float4(t0.x, t0.y, t1.x, t1.y);
    b = make_float4(t2.x, t2.y, t3.x, t3.y);
}
__device__ __forceinline__ void store_v4(float2* vel, size_t idx, float4 a, float4 b)
{
    float4* dst = reinterpret_cast<float4*>(vel + idx);
    dst[0] = a;
    dst[1] = b;
}
__device__ __forceinline__ void store_v4(__half2* vel, size_t idx, float4 a, float4 b)
{
    struct alignas(16) H8 {
        __half2 t[4];
    };
    H8 h;
    h.t[0] = __float22half2_rn(make_float2(a.x, a.y));
    h.t[1] = __float22half2_rn(make_float2(a.z, a.w));
    h.t[2] = __float22half2_rn(make_float2(b.x, b.y));
    h.t[3] = __float22half2_rn(make_float2(b.z, b.w));
    *reinterpret_cast<H8*>(vel + idx) = h;
}

// One texel row of a Jacobi iteration (pressureShader script.js:881-888, operand order of line 887:
// ((L + R) + B) + T - div) * 0.25): 11 VALU instructions for the lane's four texels.  HALF: the iteration's output goes
// through fp16, as it does when the reference renders it into a half-float texture.
// EDGE: 0 = the tile is interior (no select at all); 1 = it touches the left / right domain border only, at a width that is a
// multiple of 4 (two selects per row); 2 = everything (bottom / top rows, the partly padded last quad of any other width).  Border
// tiles are 14 % of the 4096^2 grid and their selects cost arithmetic the kernel is bound by, so the cheap case has its own path.
template <int EDGE, bool HALF = false>
__device__ __forceinline__ Quad jacobi_row(Quad C, Quad T, Quad B, const Quad D, int gj, int H, bool at_left, int nv)
{
    if (EDGE == 2 && nv < 4) {  // the quad that holds column W - 1 of a width that is not a multiple of 4 (nv = its texels inside the
        if (nv < 2) C.i.x = C.o.x;  // domain): CLAMP_TO_EDGE inside the quad — the texels beyond the edge repeat the last one, so
        if (nv < 3) C.i.y = C.i.x;  // the horizontal sums below see the clamped neighbour (the padding columns hold no data)
        C.o.y = C.i.y;
    }
    float L = from_left_lane(C.o.y);   // column 4*lane - 1 = the left lane's c3
    float R = from_right_lane(C.o.x);  // column 4*lane + 4 = the right lane's c0
    if (EDGE) {  // CLAMP_TO_EDGE: an off-domain neighbour is the centre texel
        if (at_left) L = C.o.x;
        if (nv <= 4) R = C.o.y;  // the lane that holds column W - 1 (lanes beyond it only feed texels outside the domain)
    }
    if (EDGE == 2) {
        if (gj == 0) B = C;
        if (gj == H - 1) T = C;
    }
    const v2f quarter = v2f{ 0.25f, 0.25f };
    float h0 = L + C.i.x;  // texel 0: left + c1
    float h3 = C.i.y + R;  // texel 3: c2 + right
    if (!EDGE) {
        // keep these two adds scalar: as a packed pair (what the SLP vectoriser makes of them) they need two
        // v_mov_b32_dpp in front; scalar, the lane shift folds into the add itself (v_add_f32_dpp)
        asm("" : "+v"(h0));
        asm("" : "+v"(h3));
    }
    const v2f h_o = v2f{ h0, h3 };
    const v2f h_i = C.o + __builtin_shufflevector(C.i, C.i, 1, 0);   // texels 1, 2: c0 + c2, c3 + c1
    Quad n;
    n.o = (h_o + B.o + T.o - D.o) * quarter;
    n.i = (h_i + B.i + T.i - D.i) * quarter;
    if (HALF) {
        n.o = round_half(n.o);
        n.i = round_half(n.i);
    }
    return n;
}

// One Jacobi iteration over the wave's RY rows, updated in place.  The wave's first and last row need a row of the
// neighbouring waves (LDS mailbox, `box` is this iteration's slot); every other row only needs the wave's own
// registers.  So: publish, sweep the inner rows (one delay register carries the old row below), THEN meet the other
// waves at the barrier and finish the two outer rows — the mailbox round trip hides behind RY - 2 rows of arithmetic.
// SKIP (round 6): the rows of the tile's first and last wave that the apron has already reached are not swept any more.  After k iterations the
// outer k rows of a tile are stale (nobody reads them again: the next iteration's valid rows only look one row out); sweeping them all the
// same is 110 of the 800 row sweeps of an 80-row tile.  rows [lo_skip, RY - hi_skip) of this wave are swept (wave-uniform; 0, 0 = all).
// NOSYNC (lab probe, results NOT valid): no mailbox, no barrier — what the exchange between the waves costs
template <int NW, int RY, int EDGE, bool HALF, bool SKIP = false, bool NOSYNC = false>
__device__ __forceinline__ void jacobi_sweep(Quad (&P)[RY], const Quad (&D)[RY], float4 (*box)[2][64], int wv, int lane, int gy,
                                             int H, bool at_left, int nv, int lo_skip = 0, int hi_skip = 0)
{
    static_assert(RY >= 3, "a wave needs an inner row");
    if constexpr (NOSYNC) {
        const Quad old0 = P[0], old1 = P[1];
        Quad below = old0;
#pragma unroll
        for (int r = 1; r < RY - 1; r++) {
            const Quad C = P[r];
            P[r] = jacobi_row<EDGE, HALF>(C, P[r + 1], below, D[r], gy + r, H, at_left, nv);
            below = C;
        }
        P[RY - 1] = jacobi_row<EDGE, HALF>(P[RY - 1], old1, below, D[RY - 1], gy + RY - 1, H, at_left, nv);
        P[0] = jacobi_row<EDGE, HALF>(old0, old1, below, D[0], gy, H, at_left, nv);
        return;
    }
    box[wv][0][lane] = raw_of(P[0]);
    box[wv][1][lane] = raw_of(P[RY - 1]);
    const Quad old0 = P[0], old1 = P[1];
    Quad below = old0;
    const int r_end = RY - hi_skip;
#pragma unroll
    for (int r = 1; r < RY - 1; r++) {
        const Quad C = P[r];
        if (!SKIP || (r >= lo_skip && r < r_end)) P[r] = jacobi_row<EDGE, HALF>(C, P[r + 1], below, D[r], gy + r, H, at_left, nv);
        below = C;
    }
    __syncthreads();
    // unconditional b128 reads: the first/last wave reads its own mailbox, which only feeds the stale apron
    const Quad lo = quad_of_raw(box[wv > 0 ? wv - 1 : 0][1][lane]);
    const Quad hi = quad_of_raw(box[wv < NW - 1 ? wv + 1 : NW - 1][0][lane]);
    if (!SKIP || (RY - 1 >= lo_skip && hi_skip == 0)) P[RY - 1] = jacobi_row<EDGE, HALF>(P[RY - 1], hi, below, D[RY - 1], gy + RY - 1, H, at_left, nv);
    if (!SKIP || (lo_skip == 0 && hi_skip < RY)) P[0] = jacobi_row<EDGE, HALF>(old0, old1, lo, D[0], gy, H, at_left, nv);
}

// T = float (fp32 fields) or __half (fp16 storage: the clear and every iteration round their output to fp16)
// CHAIN (lab, k_jacobi_tb_chain): the pressure is read with `sc1` loads (past this CU's L1) and stored with `sc1` write-through stores, so
// that a tile of the NEXT block of iterations, on any XCD, may read it inside the same launch once this tile has said it is done — the
// guide's R1 hand-off (payload write-through, every storing wave drains, one flag), without a fence on either side.
typedef unsigned int chain_u4 __attribute__((ext_vector_type(4)));
[[maybe_unused]] __device__ __forceinline__ Quad load_quad_sc1(__amdgpu_buffer_rsrc_t r, size_t idx)
{
    const chain_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(idx * sizeof(float)), 0, 16);   // aux 16 = sc1
    return quad_of(make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)));
}
[[maybe_unused]] __device__ __forceinline__ void store_quad_sc1(__amdgpu_buffer_rsrc_t r, size_t idx, Quad q)
{
    const float4 f = float4_of(q);
    __builtin_amdgcn_raw_buffer_store_b128(chain_u4{ __float_as_uint(f.x), __float_as_uint(f.y), __float_as_uint(f.z), __float_as_uint(f.w) }, r,
                                           (unsigned)(idx * sizeof(float)), 0, 16);
}

// SKIP (round 6, k_jacobi_tb_chain): the tile's first and last 10-row block skip the rows the apron has reached (jacobi_sweep) — and those two
// blocks are given to the waves with threadIdx.y 0 and 2.  A workgroup's waves go to the SIMDs in the cyclic order 0 -> 2 -> 1 -> 3 from a start
// that advances by one for every workgroup a CU takes (MI355X guide, LDS section; profiles/r06/placement_probe.txt: waves w and w + 4 share a
// SIMD, and the two workgroups resident on a CU start one place apart 97 % of the time): with the light blocks on waves 0 and 2 of BOTH
// workgroups, every SIMD of the CU holds exactly one light wave and three full ones — 345 row sweeps per ten iterations instead of 400 on
// each SIMD, not 290 on two of them and 400 on the others.  Placement changes the speed only, never the result.
struct NoWait {
    __device__ __forceinline__ void operator()() const {}
};
// wait_deps (k_jacobi_tb_chain, round 6): called between the tile's DIVERGENCE loads and its pressure loads.  The divergence is an input of
// the whole launch — nobody writes it — so its ten loads per wave go out BEFORE the workgroup has asked whether the previous block's tiles are
// done: their round trip is the poll's, and only the pressure's ten loads follow the answer (a tile is its memory round trips in a row:
// profiles/r06/chain_bounds_probes.txt).
template <int NW, int RY, int HX, int HY, int EDGE, class T, bool GS = false, class V2 = float2, bool CHAIN = false, bool SKIP = false, bool LIGHT = false, int PROBE = 0, class FW = NoWait>
__device__ __forceinline__ void jacobi_tb_body_impl(const Win& w, const T* __restrict__ p, const T* __restrict__ div,
                                               T* __restrict__ p_out, float pscale, int iters, int ga, int gb, int x0,
                                               int y0, float4 (*mail)[NW][2][64], const V2* __restrict__ vel = nullptr,
                                               V2* __restrict__ vel_out = nullptr, FW&& wait_deps = FW{})
{
    constexpr bool HALF = sizeof(T) == 2;
    using G = JacobiTB<NW, RY, HX, HY>;
    const int lane = threadIdx.x;
    // the block is (64, NW): threadIdx.y is the wave index — tell the compiler it is wave-uniform so that all
    // per-row address arithmetic lands on the scalar unit and the loads use SGPR-base + lane-offset addressing
    const int hw = __builtin_amdgcn_readfirstlane(threadIdx.y);
    // which 10-row block of the tile this wave holds: its own number — or, with SKIP, the tile's first block on wave 0 and its last on wave 2
    const int wv = !SKIP ? hw : (hw == 2 ? NW - 1 : (hw > 2 ? hw - 1 : hw));
    const int cx = x0 + 4 * lane;     // first of this lane's 4 columns
    const int gy = y0 + wv * RY;      // global row of this wave's row 0

    // Load the whole tile with UNCONDITIONAL loads from clamped addresses (no per-row branch, so the 2*RY
    // 1 KiB wave loads are all in flight together).  Apron texels outside the domain or outside the
    // stripe's window then hold some other texel's finite value; that is fine: they are never stored,
    // and no in-domain texel reads an off-domain neighbour (EDGE selects) or a texel deeper in
    // the apron than `iters`.
    Quad P[RY], D[RY];
    const unsigned cxs = (unsigned)(min(max(cx, w.c0), w.c0 + w.P - 4) - w.c0);  // array column of the lane's quad, kept inside the array
    const v2f ps = v2f{ pscale, pscale };
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rin, rout;
    if constexpr (CHAIN) {
        const int bytes = (int)((size_t)w.rows * (size_t)w.P * sizeof(float));
        rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p), 0, bytes, 0x00020000);
        rout = __builtin_amdgcn_make_buffer_rsrc(p_out, 0, bytes, 0x00020000);
    }
    if constexpr (CHAIN) {
#pragma unroll
        for (int r = 0; r < RY; r++) {
            const int lr = min(max(gy + r - w.g0, 0), w.rows - 1);
            D[r] = load_quad(div, (size_t)lr * (size_t)w.P + cxs);
        }
        wait_deps();
#pragma unroll
        for (int r = 0; r < RY; r++) {
            const int lr = min(max(gy + r - w.g0, 0), w.rows - 1);
            P[r] = load_quad_sc1(rin, (size_t)lr * (size_t)w.P + cxs);
        }
    } else {
#pragma unroll
        for (int r = 0; r < RY; r++) {
            const int lr = min(max(gy + r - w.g0, 0), w.rows - 1);
            const size_t row = (size_t)lr * (size_t)w.P;  // wave-uniform
            P[r] = load_quad(p, row + cxs);
            D[r] = load_quad(div, row + cxs);
        }
    }
#pragma unroll
    for (int r = 0; r < RY; r++) {  // clearShader folded in: value * p, same rounding as the separate pass
        P[r].o = ps * P[r].o;
        P[r].i = ps * P[r].i;
        if (HALF) {  // the clear pass renders into a half-float texture too (pscale == 1: already halves, unchanged)
            P[r].o = round_half(P[r].o);
            P[r].i = round_half(P[r].i);
        }
    }

    const bool at_left = (cx == 0);
    const int nv = w.W - cx;  // texels of the lane's quad inside the domain: 4 or more everywhere but in the lane(s) at the right edge

    // two iterations per trip: the mailbox slot is a compile-time constant and the register allocator can hand the
    // second sweep's results back to the registers the first one read (no copies on the loop back-edge)
    int it = 0;
    if constexpr (SKIP) {
        static_assert(!GS && NW >= 4, "the skipping form is the plain tile's");
        // rows this wave leaves out in iteration it + 1: the first it + 1 rows of the tile's first block / the last it + 1 of its last block —
        // unless the tile holds the domain's edge on that side (CLAMP_TO_EDGE keeps the edge exact: nothing is stale there)
        // Two copies of the WHOLE body, picked per wave (jacobi_tb_body below): the six full waves run the plain sweep untouched, the two light
        // ones the sweep with the row conditions.  (The conditions in everybody's code: the register allocator pays for the joins with four
        // moves a row and the tile is 6.5 % SLOWER; two copies of the loop only: 247 spilled registers at the loop's joins — visit 7.)
        // Every wave meets the others at a barrier of its own copy: s_barrier counts arrivals.
        if constexpr (PROBE == 2) {   // lab: no mailbox, no barrier (results not valid)
            for (; it < iters; it++) jacobi_sweep<NW, RY, EDGE, HALF, false, true>(P, D, mail[0], wv, lane, gy, w.H, at_left, nv);
        } else if constexpr (PROBE == 1) {   // lab: no arithmetic at all — every wave skips every row (results not valid)
            for (; it + 2 <= iters; it += 2) {
                jacobi_sweep<NW, RY, EDGE, HALF, true>(P, D, mail[0], wv, lane, gy, w.H, at_left, nv, RY, 0);
                jacobi_sweep<NW, RY, EDGE, HALF, true>(P, D, mail[1], wv, lane, gy, w.H, at_left, nv, RY, 0);
            }
            if (it < iters) jacobi_sweep<NW, RY, EDGE, HALF, true>(P, D, mail[0], wv, lane, gy, w.H, at_left, nv, RY, 0);
        } else if constexpr (LIGHT) {
            const int lo_on = (wv == 0 && y0 > 0) ? 1 : 0, hi_on = (wv == NW - 1 && y0 + NW * RY < w.H) ? 1 : 0;
            for (; it + 2 <= iters; it += 2) {
                jacobi_sweep<NW, RY, EDGE, HALF, true>(P, D, mail[0], wv, lane, gy, w.H, at_left, nv, lo_on * min(it + 1, RY), hi_on * min(it + 1, RY));
                jacobi_sweep<NW, RY, EDGE, HALF, true>(P, D, mail[1], wv, lane, gy, w.H, at_left, nv, lo_on * min(it + 2, RY), hi_on * min(it + 2, RY));
            }
            if (it < iters) jacobi_sweep<NW, RY, EDGE, HALF, true>(P, D, mail[0], wv, lane, gy, w.H, at_left, nv, lo_on * min(it + 1, RY), hi_on * min(it + 1, RY));
        } else {
            for (; it + 2 <= iters; it += 2) {
                jacobi_sweep<NW, RY, EDGE, HALF>(P, D, mail[0], wv, lane, gy, w.H, at_left, nv);
                jacobi_sweep<NW, RY, EDGE, HALF>(P, D, mail[1], wv, lane, gy, w.H, at_left, nv);
            }
            if (it < iters) jacobi_sweep<NW, RY, EDGE, HALF>(P, D, mail[0], wv, lane, gy, w.H, at_left, nv);
        }
    } else {
        for (; it + 2 <= iters; it += 2) {
            jacobi_sweep<NW, RY, EDGE, HALF>(P, D, mail[0], wv, lane, gy, w.H, at_left, nv);
            jacobi_sweep<NW, RY, EDGE, HALF>(P, D, mail[1], wv, lane, gy, w.H, at_left, nv);
        }
        if (it < iters) jacobi_sweep<NW, RY, EDGE, HALF>(P, D, mail[0], wv, lane, gy, w.H, at_left, nv);
    }

    // store the texels the apron kept exact
    int xa, xb, out_lo, out_hi;
    tile_exact(x0, G::TX, HX, w.W, w.x0, w.x1, xa, xb);
    tile_exact(y0, G::TY, HY, w.H, ga, gb, out_lo, out_hi);
    const bool col_store = (cx >= xa) && (cx < xb);
#pragma unroll
    for (int r = 0; r < RY; r++) {
        const int gj = gy + r;
        if (col_store && gj >= out_lo && gj < out_hi) {
            if constexpr (CHAIN) store_quad_sc1(rout, (size_t)at(w, gj, cx), P[r]);
            else store_quad(p_out, (size_t)at(w, gj, cx), P[r]);
        }
    }

    // K6 folded into the LAST launch of the loop (gradientSubtractShader script.js:895-913): the tile still holds the final pressure,
    // exact one ring beyond what it stores (the launcher gives this instantiation an apron of iters + 1 rows, HX >= iters + 1 columns),
    // so velocity -= (R - L, T - B) needs no second trip of the pressure through HBM and no launch of its own: the step moves
    // 16 B/texel here (velocity in and out) instead of 20 in k_gradsub4 + the kernel boundary.  Same subtraction per texel as
    // gradsub_texel / gradsub4_body, hence the same bits.
    if constexpr (GS) {
        float4 (*box)[2][64] = mail[iters & 1];  // the slot the NEXT sweep would use: every wave is past the barrier behind its last readers
        box[wv][0][lane] = raw_of(P[0]);
        box[wv][1][lane] = raw_of(P[RY - 1]);
        __syncthreads();
        const Quad lo = quad_of_raw(box[wv > 0 ? wv - 1 : 0][1][lane]);
        const Quad hi = quad_of_raw(box[wv < NW - 1 ? wv + 1 : NW - 1][0][lane]);
        constexpr int CH = RY >= 8 ? (RY + 1) / 2 : RY;  // velocity rows in flight together (8 registers each)
#pragma unroll
        for (int r0 = 0; r0 < RY; r0 += CH) {
            float4 va[CH], vb[CH];
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const int r = r0 + k, gj = gy + r;
                if (r < RY && gj >= out_lo && gj < out_hi) {  // wave-uniform
                    if (col_store) load_v4(vel, (size_t)at(w, gj, cx), va[k], vb[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const int r = r0 + k, gj = gy + r;
                if (r < RY && gj >= out_lo && gj < out_hi) {
                    Quad C = P[r];
                    Quad Tq = r < RY - 1 ? P[r + 1] : hi;
                    Quad Bq = r > 0 ? P[r - 1] : lo;
                    if (EDGE == 2 && nv < 4) {  // CLAMP_TO_EDGE inside the partly padded last quad, as in jacobi_row
                        if (nv < 2) C.i.x = C.o.x;
                        if (nv < 3) C.i.y = C.i.x;
                        C.o.y = C.i.y;
                    }
                    float L = from_left_lane(C.o.y), R = from_right_lane(C.o.x);  // all lanes active here: the shifts see every neighbour
                    if (EDGE) {
                        if (at_left) L = C.o.x;
                        if (nv <= 4) R = C.o.y;
                    }
                    if (EDGE == 2) {
                        if (gj == 0) Bq = C;
                        if (gj == w.H - 1) Tq = C;
                    }
                    if (col_store) {
                        float4 oa, ob;
                        oa.x = va[k].x - (C.i.x - L);
                        oa.y = va[k].y - (Tq.o.x - Bq.o.x);
                        oa.z = va[k].z - (C.i.y - C.o.x);
                        oa.w = va[k].w - (Tq.i.x - Bq.i.x);
                        ob.x = vb[k].x - (C.o.y - C.i.x);
                        ob.y = vb[k].y - (Tq.i.y - Bq.i.y);
                        ob.z = vb[k].z - (R - C.i.y);
                        ob.w = vb[k].w - (Tq.o.y - Bq.o.y);
                        store_v4(vel_out, (size_t)at(w, gj, cx), oa, ob);
                    }
                }
            }
        }
    }
}

template <int NW, int RY, int HX, int HY, int EDGE, class T, bool GS = false, class V2 = float2, bool CHAIN = false, bool SKIP = false, int PROBE = 0, class FW = NoWait>
__device__ __forceinline__ void jacobi_tb_body(const Win& w, const T* __restrict__ p, const T* __restrict__ div, T* __restrict__ p_out, float pscale, int iters,
                                               int ga, int gb, int x0, int y0, float4 (*mail)[NW][2][64], const V2* __restrict__ vel = nullptr,
                                               V2* __restrict__ vel_out = nullptr, FW&& wait_deps = FW{})
{
    if constexpr (PROBE != 0) {
        jacobi_tb_body_impl<NW, RY, HX, HY, EDGE, T, GS, V2, CHAIN, true, false, PROBE>(w, p, div, p_out, pscale, iters, ga, gb, x0, y0, mail, vel, vel_out, wait_deps);
    } else if constexpr (SKIP) {
        const int hw = __builtin_amdgcn_readfirstlane(threadIdx.y);
        const bool light = (hw == 0 && y0 > 0) || (hw == 2 && y0 + NW * RY < w.H);   // the tile's first block is wave 0's, its last wave 2's
        if (light) jacobi_tb_body_impl<NW, RY, HX, HY, EDGE, T, GS, V2, CHAIN, true, true>(w, p, div, p_out, pscale, iters, ga, gb, x0, y0, mail, vel, vel_out, wait_deps);
        else jacobi_tb_body_impl<NW, RY, HX, HY, EDGE, T, GS, V2, CHAIN, true, false>(w, p, div, p_out, pscale, iters, ga, gb, x0, y0, mail, vel, vel_out, wait_deps);
    } else {
        jacobi_tb_body_impl<NW, RY, HX, HY, EDGE, T, GS, V2, CHAIN, false, false>(w, p, div, p_out, pscale, iters, ga, gb, x0, y0, mail, vel, vel_out, wait_deps);
    }
}

// Tile `b` of an nx x ny tile set over rows [ga, gb): its place in the XCD-aware order, then the body with the CLAMP_TO_EDGE selects its
// position needs (interior / left-right border / everything).  HYT = the tile's row apron (HY, or HY + 1 with the gradient subtract folded in).
template <int NW, int RY, int HX, int HYT, bool GS = false, class T, class V2 = float2>
__device__ __forceinline__ void jacobi_tb_tile(const Win& w, const T* __restrict__ p, const T* __restrict__ div, T* __restrict__ p_out, float pscale,
                                               int iters, int ga, int gb, int xs, int ys, int nx, int ny, int b, int remap,
                                               float4 (*mail)[NW][2][64], const V2* __restrict__ vel = nullptr, V2* __restrict__ vel_out = nullptr)
{
    using G = JacobiTB<NW, RY, HX, HYT>;
    int bx, by;
    tile_of_block(b, nx, ny, remap, bx, by);
#ifdef FLUID_PROBES
    // lab (FLUID_XCD_REMAP bit 2): every other octet of workgroups runs at a raised wave priority — when the two workgroups of a CU differ,
    // the favoured one iterates at the rate of a workgroup alone and reaches its store / load phase while the other still computes
    if ((remap & 4) && ((int)blockIdx.x & 8)) __builtin_amdgcn_s_setprio(2);
    // (bit 3: the second-dispatched half of a workgroup's waves — the arbitration losers on their SIMDs, MI355X guide — at priority 1)
    if ((remap & 8) && (int)threadIdx.y >= NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
    const int x0 = xs + bx * G::VX, y0 = ys + by * G::VY;
    const bool xedge = (x0 <= 0) || (x0 + G::TX >= w.W), yedge = (y0 <= 0) || (y0 + G::TY >= w.H);
    const bool ragged = (w.W & 3) != 0 && x0 + G::TX >= w.W;  // the tile holds the partly padded last quad
    if (yedge || ragged) jacobi_tb_body<NW, RY, HX, HYT, 2, T, GS, V2>(w, p, div, p_out, pscale, iters, ga, gb, x0, y0, mail, vel, vel_out);
    else if (xedge) jacobi_tb_body<NW, RY, HX, HYT, 1, T, GS, V2>(w, p, div, p_out, pscale, iters, ga, gb, x0, y0, mail, vel, vel_out);
    else jacobi_tb_body<NW, RY, HX, HYT, 0, T, GS, V2>(w, p, div, p_out, pscale, iters, ga, gb, x0, y0, mail, vel, vel_out);
}

// ---- the whole pressure loop as ONE launch (round 5) — shipped where it measured faster: 4096-wide grids --------
// A launch of ten iterations is its bytes over the bandwidth plus ~8 us in which the chip fills and drains, five times per step.  Here the
// five blocks of iterations are one grid of 5 x T workgroups: workgroup B = l T + b runs tile b of block l, reads the pressure buffer l % 2
// and writes the other one.  A tile of block l may start when the three tile ROWS of block l - 1 around it are complete (the rows its
// apron reads — and, because block l writes the buffer block l - 1 read, the rows whose readers must be through): one counter per
// (block, tile row), bumped by every tile behind its drained write-through stores, polled by one lane.
// The order that makes this pay: within a block every XCD takes a contiguous run of the row-major tile sequence (as k_jacobi_tb does), and
// ODD blocks walk their runs BACKWARDS.  A front that finishes block l at the bottom of its run starts block l + 1 right there — the rows
// it needs are its own last ones and the neighbouring front's FIRST ones, done long ago — and reaches the top of its run when the front
// above has long finished block l.  No front ever waits for another one to end; the drain of block l is the fill of block l + 1.
// Dependencies point to lower workgroup ids only: with workgroups dispatched in id order (what the hardware does; HIP does not promise it)
// a waiting workgroup waits for one that is resident or done.  The poll is bounded all the same: a workgroup that gives up sets *err and
// computes on stale data rather than hang the device.
constexpr int CHAIN_MAX_BLOCKS = 24, CHAIN_MAX_ROWS = 512, CHAIN_MAX_PANELS = 8;   // (200 iterations = 20 blocks; 16384-wide = 71 tiles = 4 panels)
struct ChainPlan {
    int blocks, tiles;            // blocks of iterations; workgroups per block (mode 0: nx * ny tiles; band-cyclic: 8 G band nx, some without a tile)
    int gs;                       // 1: one more block of workgroups behind the last block of iterations — the gradient subtract, tile by tile (GSB)
    int iters[CHAIN_MAX_BLOCKS + 1];
    int xa[CHAIN_MAX_BLOCKS + 1], xb[CHAIN_MAX_BLOCKS + 1];   // ... and columns (2-D tiles: the ghost columns shrink too; everywhere else the window's own)
    int ga[CHAIN_MAX_BLOCKS + 1], gb[CHAIN_MAX_BLOCKS + 1];   // rows block l stores (a stripe's blocks recompute fewer ghost rows each: the ranges shrink; the
                                                      // tiling is block 0's — the widest — for all of them, and a tile with nothing to store only counts itself)
                                                      // (entry [blocks]: the rows / columns the gradient-subtract block stores, gs == 1)
    int band;                     // > 0: the band-cyclic order below, `band` tile rows per band
    int pw;                       // ... in panels of pw tile columns (fluid_tiles.h; >= nx: one panel).  One counter per (block, tile row, panel)
    int rot;                      // block l gives slot-XCD k the bands (k + l * rot) % 8 of every group (fluid_tiles.h; 0 = the same XCD in every block)
    int tickets;                  // 1: a workgroup's place in the order is a ticket it draws when it starts (independent of the dispatch order)
    unsigned int target;          // a panel's tile row of the previous block is complete when its counter has reached this TIMES THE PANEL'S WIDTH:
                                  // the counters are never reset between calls of the same shape (no memset in the stream) — call number e: e + 1
    unsigned int timeout;         // 100 MHz ticks a workgroup waits for a tile row before it gives up (2 s; lab: FLUID_CHAIN_TIMEOUT_MS)
    int withhold;                 // lab (FLUID_CHAIN_WITHHOLD=row): the first tile of that tile row of block 0 never counts itself — what waits
                                  // for the row gives up: the give-up path, forced (tests/test_chain_safety.py); -1 = off
};

// GSB (round 6, lab: FLUID_CHAIN_GS=1): K6 as ONE MORE BLOCK of the chained launch.  Workgroup (blocks, b) subtracts the gradient of the final
// pressure from the velocity over the texels tile b of the last block of iterations STORED (the stored ranges partition the domain), as soon as
// the three tile rows of that block around it are complete — the same counters a block of iterations waits for, because velocity - grad p reads
// the pressure one ring out.  What it buys: the launch boundary between the loop and k_gradsub4 — the loop's last tiles drain beside gradient
// subtract workgroups instead of beside an emptying chip.  Same subtraction per texel as gradsub4_body (script.js:895-913), hence the same bits;
// CLAMP_TO_EDGE in y comes with the row clamp of the loads (a whole-domain window: the array's first / last row is the domain's).
// The wave holds the rows the Jacobi tile's wave wv holds, y0 + wv RY ... + RY, and loads one more on either side; pressure through `sc1`
// loads like a block of iterations (other CUs wrote it inside this launch), the velocity — an input of the whole launch — through plain ones.
template <int NW, int RY, int HX, int HY>
__device__ __forceinline__ void gradsub_chain_tile(const Win& w, const float* __restrict__ p, const float2* __restrict__ vel, float2* __restrict__ vel_out,
                                                   int ga, int gb, int x0, int y0)
{
    using G = JacobiTB<NW, RY, HX, HY>;
    const int lane = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int cx = x0 + 4 * lane, gy = y0 + wv * RY;
    int xa, xb, out_lo, out_hi;
    tile_exact(x0, G::TX, HX, w.W, w.x0, w.x1, xa, xb);
    tile_exact(y0, G::TY, HY, w.H, ga, gb, out_lo, out_hi);
    const int r_lo = max(out_lo - gy, 0), r_hi = min(out_hi - gy, RY);   // this wave's rows [r_lo, r_hi) are stored (wave-uniform)
    if (r_hi <= r_lo) return;
    const bool col_store = (cx >= xa) && (cx < xb);
    const unsigned cxs = (unsigned)(min(max(cx, w.c0), w.c0 + w.P - 4) - w.c0);
    const int bytes = (int)((size_t)w.rows * (size_t)w.P * sizeof(float));
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
    float4 Pr[RY + 2];   // rows gy - 1 ... gy + RY
#pragma unroll
    for (int r = 0; r < RY + 2; r++) {
        const int lr = min(max(gy - 1 + r - w.g0, 0), w.rows - 1);
        if (r + 1 > r_lo && r < r_hi + 2) {   // (wave-uniform: the rows this wave's stored rows read)
            const chain_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, (unsigned)(((size_t)lr * (size_t)w.P + cxs) * sizeof(float)), 0, 16);   // sc1
            Pr[r] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    }
    const bool at_left = (cx == 0);
    const int nv = w.W - cx;
    constexpr int CH = (RY + 1) / 2;   // velocity rows in flight together (8 registers each)
#pragma unroll
    for (int r0 = 0; r0 < RY; r0 += CH) {
        float4 va[CH], vb[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const int r = r0 + k;
            if (r < RY && r >= r_lo && r < r_hi && col_store) load_v4(vel, (size_t)at(w, gy + r, cx), va[k], vb[k]);
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const int r = r0 + k;
            if (r < RY && r >= r_lo && r < r_hi) {
                float4 C = Pr[r + 1];
                const float4 T = Pr[r + 2], B = Pr[r];
                if (nv < 4) {   // CLAMP_TO_EDGE inside the partly padded last quad, as in gradsub4_body
                    if (nv < 2) C.y = C.x;
                    if (nv < 3) C.z = C.y;
                    C.w = C.z;
                }
                float L = from_left_lane(C.w), R = from_right_lane(C.x);   // all lanes active here: the shifts see every neighbour
                if (at_left) L = C.x;
                if (nv <= 4) R = C.w;
                if (col_store) {
                    float4 oa, ob;
                    oa.x = va[k].x - (C.y - L);
                    oa.y = va[k].y - (T.x - B.x);
                    oa.z = va[k].z - (C.z - C.x);
                    oa.w = va[k].w - (T.y - B.y);
                    ob.x = vb[k].x - (C.w - C.y);
                    ob.y = vb[k].y - (T.z - B.z);
                    ob.z = vb[k].z - (R - C.z);
                    ob.w = vb[k].w - (T.w - B.w);
                    store_v4(vel_out, (size_t)at(w, gy + r, cx), oa, ob);
                }
            }
        }
    }
}

// DIAG (FLUID_JACOBI_CHAIN=2 / 3 / 4: timing probes whose RESULTS ARE NOT VALID): 1 = a tile counts itself done without draining its stores
// (what the wait for the write-through acknowledgements costs), 2 = plain pressure loads and stores instead of sc1 (what the cache policy
// costs), 3 = nobody waits for anybody (what the dependency waits cost)
template <int NW, int RY, int HX, int HY, int BPC, int DIAG = 0, bool SKIP = false, bool DFIRST = false, bool GSB = false>
__global__ void __launch_bounds__(64 * NW, (BPC * NW + 3) / 4) k_jacobi_tb_chain(Win w, float* __restrict__ pa, float* __restrict__ pb,
                                                              const float* __restrict__ div, float pscale, ChainPlan C, int xs,
                                                              int ys, int nx, int ny, unsigned int* __restrict__ done,
                                                              unsigned int* __restrict__ err, const float2* __restrict__ vel = nullptr,
                                                              float2* __restrict__ vel_out = nullptr)
{
    __shared__ float4 mail[2][NW][2][64];
    using G = JacobiTB<NW, RY, HX, HY>;
    // The workgroup's place in the ORDER is the order in which workgroups really start — a ticket — not blockIdx: a workgroup then only ever
    // waits for workgroups that have started, whatever order the hardware dispatches in (the first form relied on id order)
    __shared__ int ticket;
    if (C.tickets) {   // (one word for 6000 workgroups per step: 46 us of the step at 4096^2 — visit 12; off = blockIdx order, what the hardware dispatches in)
        if (threadIdx.x == 0 && threadIdx.y == 0) ticket = (int)__hip_atomic_fetch_add(err + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    const int B = C.tickets ? ticket : (int)blockIdx.x, l = B / C.tiles, b = B - l * C.tiles;
    int bx, by;
    if (C.band > 0) {
        // BAND-CYCLIC order, the same direction in every block: consecutive workgroups alternate XCDs (b % 8); XCD k walks bands of `band`
        // tile rows — band k of every group of eight bands, group after group — so that the rows a tile of block l + 1 needs (its own
        // band's and the neighbouring bands' of the SAME group) were finished a whole block ago, while the 8 x band x nx tiles in flight
        // together still share their aprons inside one XCD's L2.  (The first form — each XCD one contiguous run of the whole sequence, odd
        // blocks backwards — turned every front around on the tiles that had just been resident TOGETHER: a block's first 64 tiles per
        // front waited for the previous block's last 64, i.e. for its drain; +10 % instead of -2 %: profiles/r05/jacobi_chain_ab.txt.)
        if (!chain_tile_of_block(b, nx, ny, C.band, C.pw, bx, by, l * C.rot)) return;   // (fluid_tiles.h) the last group's bands beyond the grid: no tile (block-uniform)
    } else {
        // XCD b % 8 takes the (b / 8)-th tile of its contiguous run of the row-major sequence, from the far end in odd blocks
        const int n = C.tiles, q = n >> 3, r8 = n & 7, xcd = b & 7, slot = b >> 3;
        const int len = q + (xcd < r8 ? 1 : 0), start = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
        const int t = start + ((l & 1) ? len - 1 - slot : slot);
        by = t / nx;
        bx = t - by * nx;
    }
    const int ga = C.ga[l], gb = C.gb[l];
    w.x0 = C.xa[l];
    w.x1 = C.xb[l];
    const int y0t = ys + by * G::VY, x0t = xs + bx * G::VX;
    int st_lo, st_hi, sx_lo, sx_hi;
    tile_exact(y0t, G::TY, HY, w.H, ga, gb, st_lo, st_hi);
    tile_exact(x0t, G::TX, HX, w.W, w.x0, w.x1, sx_lo, sx_hi);
    const bool nothing = st_hi <= st_lo || sx_hi <= sx_lo;   // this block's ranges do not reach this tile (block-uniform): no reads, no writes — count and go
    auto wait_prev = [&]() {
    if (l > 0 && DIAG != 3 && !nothing) {
        // one lane per counter — three rows of this tile's panel, and of the panel next door where the tile sits on the panel's first / last
        // column (3 ... 9 counters) — so that they all come back in ONE memory round trip (one lane after the other: three — visit 12)
        const int t = (int)threadIdx.x, r = by - 1 + t % 3, pn = C.band > 0 ? bx / C.pw : 0, pq = pn - 1 + t / 3;
        const int np = C.band > 0 ? chain_panels(nx, C.pw) : 1, pw = C.band > 0 ? C.pw : nx;
        const bool needed = pq == pn || (pq < pn ? bx == pn * pw : (bx == pn * pw + pw - 1));
        if (threadIdx.y == 0 && t < 9 && r >= 0 && r < ny && pq >= 0 && pq < np && needed) {
            const unsigned int* flag = done + ((l - 1) * CHAIN_MAX_ROWS + r) * CHAIN_MAX_PANELS + pq;
            const unsigned int target = C.target * (unsigned)min(pw, nx - pq * pw);
            unsigned spins = 0;
            unsigned long long t0 = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                // never hang the device — bounded in WALL-CLOCK time (round 6: a count of polls is some tens of milliseconds, which a foreign
                // kernel holding the CUs can outlast; the 100 MHz clock does not care how slowly this wave gets to poll), and nobody waits once
                // somebody has given up (err is mapped HOST memory: looked at every 1024 polls only)
                if ((++spins & 63u) == 0) {
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if (t0 == 0) t0 = now;
                    if (now - t0 > (unsigned long long)C.timeout || ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) {
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
        }
        __syncthreads();
    }
    };
    if constexpr (!DFIRST) wait_prev();
    auto wait_in_body = [&]() {
        if constexpr (DFIRST) wait_prev();
    };
    const float* p = (l & 1) ? pb : pa;
    float* p_out = (l & 1) ? pa : pb;
    const int x0 = xs + bx * G::VX, y0 = ys + by * G::VY;
    if constexpr (GSB) {
        if (l == C.blocks) {   // the gradient-subtract block (block-uniform): reads the last block's pressure, stores velocity, counts nothing
            if constexpr (DFIRST) wait_prev();
            if (!nothing) gradsub_chain_tile<NW, RY, HX, HY>(w, p, vel, vel_out, ga, gb, x0, y0);
            return;
        }
    }
    const bool xedge = (x0 <= 0) || (x0 + G::TX >= w.W), yedge = (y0 <= 0) || (y0 + G::TY >= w.H);
    const bool ragged = (w.W & 3) != 0 && x0 + G::TX >= w.W;
    const float ps = l == 0 ? pscale : 1.0f;
    constexpr bool SC1 = DIAG != 2;
    constexpr int PROBE = DIAG == 5 ? 1 : (DIAG == 6 ? 2 : 0);   // lab: 5 = no arithmetic, 6 = no mailbox / barrier (results not valid)
    if (nothing) {
    } else if (yedge || ragged) jacobi_tb_body<NW, RY, HX, HY, 2, float, false, float2, SC1, SKIP, PROBE>(w, p, div, p_out, ps, C.iters[l], ga, gb, x0, y0, mail, (const float2*)nullptr, (float2*)nullptr, wait_in_body);
    else if (xedge) jacobi_tb_body<NW, RY, HX, HY, 1, float, false, float2, SC1, SKIP, PROBE>(w, p, div, p_out, ps, C.iters[l], ga, gb, x0, y0, mail, (const float2*)nullptr, (float2*)nullptr, wait_in_body);
    else jacobi_tb_body<NW, RY, HX, HY, 0, float, false, float2, SC1, SKIP, PROBE>(w, p, div, p_out, ps, C.iters[l], ga, gb, x0, y0, mail, (const float2*)nullptr, (float2*)nullptr, wait_in_body);
    // done: every storing wave drains its write-through stores, then ONE lane counts the tile (the guide's R1)
    if (DIAG != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0 && !(C.withhold >= 0 && l == 0 && by == C.withhold && bx == 0))
        __hip_atomic_fetch_add(done + (l * CHAIN_MAX_ROWS + by) * CHAIN_MAX_PANELS + (C.band > 0 ? bx / C.pw : 0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the pressure loop as ONE launch of PERSISTENT workgroups that take STACKS of tiles from per-XCD ticket heads (round 6) -------------
// What is new against k_jacobi_tb_chain (the geometry, the order and the safety argument: fluid_pchain.h):
//   * a workgroup keeps going: it draws an item (block, stack row, tile column), waits for the <= 6 counters around it in the previous block,
//     runs the stack's tiles, drains its write-through stores, counts itself, draws the next.  The order is whatever the tickets say — no
//     assumption about which workgroup the hardware starts first, and none about where it puts it;
//   * a STACK: the second and further tiles of a stack take the row below their first row from the tile before, level by level, through one
//     1 KiB LDS line per iteration (the row that tile had already published to its neighbour wave's mailbox: one more ds_write by one wave) —
//     their bottom apron is gone: 1.33 -> 1.23 (M = 2) rows of arithmetic per stored row, and fewer apron rows re-read;
//   * the counters live in two halves: a call counts in one and zeroes the other for the next call (no memset in the stream, no epoch that
//     could wrap).
// One tile of a stack: jacobi_tb_body's load / iterate / store with the carry line in and out.  `sm`: the two mailbox slots, then the two
// carry sets of HY lines.  lo_line: the LDS line (float4 index) wave 0 reads as the row below the tile — the previous tile's carry line of
// this level, or (first tile) its own mailbox line, which only feeds the stale bottom apron.
template <int NW, int RY, int EDGE>
__device__ __forceinline__ void jacobi_sweep_st(Quad (&P)[RY], const Quad (&D)[RY], float4* sm, int box, int lo_line, int out_line, int carry_row_wave,
                                                int wv, int lane, int gy, int H, bool at_left, int nv)
{
    static_assert(RY >= 3, "a wave needs an inner row");
    const int mine = box + wv * 128;
    sm[mine + lane] = raw_of(P[0]);
    sm[mine + 64 + lane] = raw_of(P[RY - 1]);
    if (wv == carry_row_wave && out_line >= 0) sm[out_line + lane] = raw_of(P[RY - 1]);   // wave-uniform: this level's row for the NEXT tile of the stack
    const Quad old0 = P[0], old1 = P[1];
    Quad below = old0;
#pragma unroll
    for (int r = 1; r < RY - 1; r++) {
        const Quad C = P[r];
        P[r] = jacobi_row<EDGE, false>(C, P[r + 1], below, D[r], gy + r, H, at_left, nv);
        below = C;
    }
    __syncthreads();
    const Quad lo = quad_of_raw(sm[(wv > 0 ? mine - 128 + 64 : lo_line) + lane]);
    const Quad hi = quad_of_raw(sm[(wv < NW - 1 ? mine + 128 : mine) + lane]);
    P[RY - 1] = jacobi_row<EDGE, false>(P[RY - 1], hi, below, D[RY - 1], gy + RY - 1, H, at_left, nv);
    P[0] = jacobi_row<EDGE, false>(old0, old1, lo, D[0], gy, H, at_left, nv);
}

// One tile of a stack.  claim != null (the stack's FIRST tile): wave 0 bumps the item's claim word in front of the tile's loads — the
// atomic's round trip is the loads' — and parks what it saw in *claimed (0: the item is this workgroup's).  hooks (the stack's LAST tile):
// wave 0 gets two moments BETWEEN sweeps, pre_a two trips before the end and pre_b one trip before it.  after_loads(): called by every wave
// once the tile's loads are out.  may_store: checked in front of the stores (a claim that was lost: somebody else runs this item).
template <int NW, int RY, int HX, int HY, int EDGE, bool SC1, class FL, class FA, class FB, class FS>
__device__ __forceinline__ void jacobi_stack_tile(const Win& w, __amdgpu_buffer_rsrc_t rin, __amdgpu_buffer_rsrc_t rout, const float* __restrict__ p,
                                                  float* __restrict__ p_out, const float* __restrict__ div, float pscale, int iters, int x0, int yt,
                                                  int row_lo, int row_hi, int col_lo, int col_hi, float4* sm, int carry_in, int carry_out,
                                                  unsigned int* claim, int* claimed, bool hooks, FL&& after_loads, FA&& pre_a, FB&& pre_b, FS&& may_store)
{
    using S = JacobiStack<NW, RY, HX, HY>;
    static_assert(S::CARRY_ROW == RY - 1, "the carried row is the one its wave publishes to the mailbox anyway (HY == RY)");
    const int lane = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int cx = x0 + 4 * lane, gy = yt + wv * RY;
    unsigned int seen = 0;
    const bool claimer = claim != nullptr && wv == 0 && lane == 0;
    if (claimer) seen = __hip_atomic_fetch_add(claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Quad P[RY], D[RY];
    const unsigned cxs = (unsigned)(min(max(cx, w.c0), w.c0 + w.P - 4) - w.c0);
    const v2f ps = v2f{ pscale, pscale };
#pragma unroll
    for (int r = 0; r < RY; r++) {   // unconditional loads from clamped addresses, all in flight together (jacobi_tb_body)
        const int lr = min(max(gy + r - w.g0, 0), w.rows - 1);
        const size_t row = (size_t)lr * (size_t)w.P;
        if constexpr (SC1) P[r] = load_quad_sc1(rin, row + cxs);
        else P[r] = load_quad(p, row + cxs);
        D[r] = load_quad(div, row + cxs);
    }
    after_loads();
    if (claimer) *claimed = (int)seen;
#pragma unroll
    for (int r = 0; r < RY; r++) {   // clearShader folded in (block 0: pscale; 1 elsewhere)
        P[r].o = ps * P[r].o;
        P[r].i = ps * P[r].i;
    }
    const bool at_left = (cx == 0);
    const int nv = w.W - cx;
    constexpr int BOX = NW * 128;   // float4 lines of one mailbox slot
    const int lo0 = carry_in >= 0 ? carry_in : 64, lo1 = carry_in >= 0 ? carry_in + 64 : BOX + 64;   // (no carry: wave 0's own mailbox line)
    const int lstep = carry_in >= 0 ? 128 : 0, ostep = 128;
    int it = 0, lo_a = lo0, lo_b = lo1, out_a = carry_out, out_b = carry_out >= 0 ? carry_out + 64 : -1;
    const int ntrips = iters >> 1, trip_a = max(ntrips - 2, 0), trip_b = max(ntrips - 1, 0);
    int trip = 0;
    if (hooks && ntrips == 0) {   // a single sweep: both moments in front of it
        pre_a();
        pre_b();
    }
    for (; it + 2 <= iters; it += 2, trip++) {   // two sweeps per trip: the mailbox slot is a compile-time constant
        if (hooks) {
            if (trip == trip_a) pre_a();
            if (trip == trip_b) pre_b();
        }
        jacobi_sweep_st<NW, RY, EDGE>(P, D, sm, 0, lo_a, out_a, S::CARRY_WAVE, wv, lane, gy, w.H, at_left, nv);
        jacobi_sweep_st<NW, RY, EDGE>(P, D, sm, BOX, lo_b, out_b, S::CARRY_WAVE, wv, lane, gy, w.H, at_left, nv);
        lo_a += lstep;
        lo_b += lstep;
        if (carry_out >= 0) {
            out_a += ostep;
            out_b += ostep;
        }
    }
    if (it < iters) jacobi_sweep_st<NW, RY, EDGE>(P, D, sm, 0, lo_a, out_a, S::CARRY_WAVE, wv, lane, gy, w.H, at_left, nv);
    if (!may_store()) return;   // block-uniform
    const bool col_store = (cx >= col_lo) && (cx < col_hi);
#pragma unroll
    for (int r = 0; r < RY; r++) {
        const int gj = gy + r;
        if (col_store && gj >= row_lo && gj < row_hi) {
            if constexpr (SC1) store_quad_sc1(rout, (size_t)at(w, gj, cx), P[r]);
            else store_quad(p_out, (size_t)at(w, gj, cx), P[r]);
        }
    }
}

// DIAG (lab): 3 = nobody waits for anybody (FLUID_JACOBI_CHAIN=4: a timing probe whose RESULTS ARE NOT VALID)
// FOURTH FORM.  What the probes of the per-tile launch say (profiles/r06/chain_bounds_probes.txt): the loop is bound by neither its arithmetic
// nor its barriers — a tile costs 14.4 us from launch to exit even with NO arithmetic, three memory round trips in a row (poll, loads, drain).
// A persistent workgroup can take two of the three off the critical path, if what it does in between costs no time of its own:
//   * WHICH item comes next is known without asking anybody: the workgroup registers once (a number g < S on its XCD's counter) and owns the
//     positions g, g + S, g + 2 S ... of that XCD's sequence — the order a freed slot would take them in anyway;
//   * the item is CLAIMED (its claim word bumped from 0) by an atomic that rides in front of its own loads; a workgroup that waits for an item
//     nobody has claimed claims and runs it itself, so nobody ever spins on an item that nobody runs, whoever owns it and wherever it runs;
//   * while the loads of item i are in flight — wave 0 has nothing else to do — it decodes item i + 1 and sends for the nine counters that
//     item depends on (LDS-DMA: no destination register); two trips of sweeps before the end it looks at what came back (and asks again if
//     it was too early), one trip before the end it posts READY;
//   * behind the stores of item i the workgroup goes straight to the loads of item i + 1; once those are out, every wave waits for its
//     STORES only (vmcnt(20)), one barrier, item i is counted.
// Not ready, a claim lost, a hole, a stack with nothing to store: the workgroup drains, counts, and wave 0 takes the CONTROL path (spin on
// the counters; claim and run what nobody has claimed; skip; count).
enum { PG_KIND = 0, PG_L, PG_X0, PG_Y0S, PG_ST_LO, PG_ST_HI, PG_SX_LO, PG_SX_HI, PG_CELL, PG_CLAIM, PG_BY, PG_BX, PG_WORDS };
enum { CT_READY = 0, CT_CLAIMED, CT_POS, CT_NSTACK, CT_PRECLAIMED, CT_WORDS };
enum { PK_DONE = 0, PK_RUN = 1, PK_COUNT_ONLY = 2 };   // nothing left | a stack to run | nothing to store here: claim, count and go
struct PChainLds {
    int item[2][PG_WORDS];   // the item being run and the one after it
    int ctl[CT_WORDS];
    int word[16];                      // lanes 0..8: the counters the next item depends on (word indices into the state half) ...
    unsigned int seen[16], want[16];   // ... what they read when asked (landed by LDS-DMA), and what they must reach
    int stack[PCHAIN_MAX_BLOCKS + 2];  // items put aside while this workgroup runs one that they wait for and nobody had claimed (l << 24 | by << 12 | bx)
};
constexpr int PCHAIN_NO_PENDING = -2;   // (-1: an item that is deliberately not counted — the lab's withheld item)
template <int NW, int RY, int HX, int HY, int DIAG = 0>
__global__ void __launch_bounds__(64 * NW, (2 * NW + 3) / 4) k_jacobi_pchain(Win w, float* __restrict__ pa, float* __restrict__ pb,
                                                                            const float* __restrict__ div, float pscale, PChainPlan plan,
                                                                            unsigned int* __restrict__ state, unsigned int* __restrict__ err)
{
    using G = JacobiTB<NW, RY, HX, HY>;
    using S = JacobiStack<NW, RY, HX, HY>;
    constexpr int BOX = NW * 128, CARRY0 = 2 * BOX;
    __shared__ float4 sm[2 * BOX + 2 * HY * 64];   // two mailbox slots, two carry sets of HY lines (a tile writes one set while it reads the other)
    __shared__ PChainLds L;
    __shared__ PChainPlan sP;                       // the plan, in LDS: read where it is used (a kernel argument's loads are hoisted into SGPRs the tile body has not got)
    const int lane = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    {
        const int* src = reinterpret_cast<const int*>(&plan);
        int* dst = reinterpret_cast<int*>(&sP);
        for (int i = wv * 64 + lane; i < (int)(sizeof(PChainPlan) / sizeof(int)); i += 64 * NW) dst[i] = src[i];
    }
    {   // the other half of the state: zero for the next call (nobody polls it before this launch has retired)
        unsigned int* const other = state + (size_t)(plan.d.bank ^ 1) * plan.d.bank_words;
        for (int i = (int)blockIdx.x * 64 * NW + wv * 64 + lane; i < plan.d.bank_words; i += (int)gridDim.x * 64 * NW) other[i] = 0u;
    }
    unsigned int* const mine = state + (size_t)plan.d.bank * plan.d.bank_words;   // (two SGPRs across the body: affordable)
    const int slots_per_xcd = (int)gridDim.x >> 3;   // S: workgroups (= owners of positions) per XCD sequence
    if (wv == 0 && lane == 0) {
        // register: a number g < S on the counter of the XCD this workgroup runs on (HW_REG_XCC_ID: affinity only) — or, if that XCD already has
        // its S owners, on the next one's: 8 S workgroups, 8 S numbers, each taken once wherever the hardware puts the workgroups
        const int x0 = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);
        int pos = -1;
        for (int k = 0; k < 8 && pos < 0; k++) {
            const int x = (x0 + k) & 7;
            const unsigned g = __hip_atomic_fetch_add(mine + x * PCHAIN_HEAD_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g < (unsigned)slots_per_xcd) pos = (x << 28) | (int)g;   // head << 28 | position in its sequence
        }
        L.ctl[CT_POS] = pos;
        L.ctl[CT_READY] = 0;
        L.ctl[CT_NSTACK] = 0;
        L.ctl[CT_PRECLAIMED] = 0;
        L.item[0][PG_KIND] = PK_DONE;
        L.item[1][PG_KIND] = PK_DONE;
    }
    __syncthreads();
    auto dims = [&]() -> PChainDims {   // (field by field: a by-value struct filled through a pointer lands in scratch memory)
        PChainDims d;
#define PD(f) d.f = __builtin_amdgcn_readfirstlane(sP.d.f)
        PD(blocks); PD(nx); PD(ny); PD(xs); PD(ys); PD(stack); PD(np); PD(pw); PD(bh); PD(nrg); PD(nb); PD(slots); PD(bank); PD(bank_words); PD(withhold); PD(stagger);
#undef PD
        d.timeout = (unsigned)__builtin_amdgcn_readfirstlane((int)sP.d.timeout);
#define PF(f) d.f = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sP.d.f)))
        PF(r_slots); PF(r_pw); PF(r_nb); PF(r_np); PF(r_bh);
#undef PF
        return d;
    };
    auto cap_of = [&](int x) -> int { return __builtin_amdgcn_readfirstlane(sP.d.cap[x]); };
    // item (l, by, bx): its geometry into item[slot] (lane 0 of wave 0), its dependencies into word / want (lanes 0..8); returns its kind
    auto post_item = [&](const PChainDims& C, int slot, int l, int by, int bx) -> int {
        const int x0 = C.xs + bx * G::VX, y0s = C.ys + by * (S::span(C.stack) - 2 * HY);
        int st_lo, st_hi, sx_lo, sx_hi;
        tile_exact(y0s, S::span(C.stack), HY, w.H, __builtin_amdgcn_readfirstlane(sP.ga[l]), __builtin_amdgcn_readfirstlane(sP.gb[l]), st_lo, st_hi);
        tile_exact(x0, G::TX, HX, w.W, __builtin_amdgcn_readfirstlane(sP.xa[l]), __builtin_amdgcn_readfirstlane(sP.xb[l]), sx_lo, sx_hi);
        const int kind = (st_hi <= st_lo || sx_hi <= sx_lo) ? PK_COUNT_ONLY : PK_RUN;   // (COUNT_ONLY: this block's ranges do not reach this stack — no reads, no writes)
        if (lane == 0) {
            int* g = L.item[slot];
            g[PG_KIND] = kind;
            g[PG_L] = l;
            g[PG_X0] = x0;
            g[PG_Y0S] = y0s;
            g[PG_ST_LO] = st_lo;
            g[PG_ST_HI] = st_hi;
            g[PG_SX_LO] = sx_lo;
            g[PG_SX_HI] = sx_hi;
            g[PG_CELL] = (C.withhold >= 0 && C.withhold == (l * C.ny + by) * C.nx + bx) ? -1 : pchain_cell(C, l, by, pchain_panel_of(C, bx));
            g[PG_CLAIM] = pchain_claim_word(C, l, by, bx);
            g[PG_BY] = by;
            g[PG_BX] = bx;
        }
        if (lane < 16) {
            const int r = by - 1 + lane / 3, c = bx - 1 + lane % 3;
            int word = 0;
            unsigned int need = 0;   // (a lane with nothing to look at reads word 0 and wants nothing of it)
            if (l > 0 && DIAG != 3 && lane < 9 && r >= 0 && r < C.ny && c >= 0 && c < C.nx) {
                const int pn = pchain_panel_of(C, c);
                need = (unsigned)pchain_panel_width(C, pn);
                word = pchain_cell(C, l - 1, r, pn);
            }
            L.word[lane] = word;
            L.want[lane] = need;
        }
        return kind;
    };
    auto send_poll = [&]() {   // lanes 0..15 of wave 0: the counters of word[] into seen[], by LDS-DMA (aux 16 = sc1: past this CU's L1)
        if (lane < 16) {
            typedef __attribute__((address_space(3))) unsigned int lds_u32;   // (the LDS base travels in M0: handed over as a wave-uniform value explicitly)
            lds_u32* const dst = (lds_u32*)(size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_u32*)L.seen);
            __builtin_amdgcn_global_load_lds(mine + L.word[lane], dst, 4, 0, 16);
        }
    };
    auto poll_ok = [&]() -> bool {   // wave 0, the poll has landed: is every counter there?
        const bool ok = lane >= 16 || L.seen[lane] >= L.want[lane];
        return __ballot(!ok) == 0;
    };
    // the next position this workgroup owns, decoded into item[slot]; holes are passed over; PK_DONE when the sequence is through
    auto next_owned = [&](const PChainDims& C, int slot) -> int {
        int pos = __builtin_amdgcn_readfirstlane(L.ctl[CT_POS]);
        int kind = PK_DONE;
        while (pos >= 0) {
            const int hx = (int)((unsigned)pos >> 28), ht = pos & 0x0fffffff;
            if (ht >= cap_of(hx)) {
                pos = -1;
                break;
            }
            int l, by, bx, q;
            const bool real = pchain_item(C, hx, ht, l, by, bx, q);
            pos += slots_per_xcd;   // (the position after this one, whatever this one turns out to be)
            if (real) {
                kind = post_item(C, slot, l, by, bx);
                break;
            }
        }
        if (lane == 0) {
            L.ctl[CT_POS] = pos;
            if (kind == PK_DONE) L.item[slot][PG_KIND] = PK_DONE;
        }
        return kind;
    };

    int slot = 0, pending = PCHAIN_NO_PENDING;   // pending: the cell of an item whose stores are still draining (counted behind the next item's loads)
    bool shadow_now = false;                     // the tile being run is its stack's last one
    // ---- the hooks of the tile body ----
    auto after_loads = [&]() {
        if (pending != PCHAIN_NO_PENDING) {   // block-uniform
            asm volatile("s_waitcnt vmcnt(20)" ::: "memory");   // this wave's stores of the item before are through (the 20 loads just issued stay in flight)
            __builtin_amdgcn_s_barrier();                        // ... and every other wave's
            if (wv == 0 && lane == 0 && pending >= 0) __hip_atomic_fetch_add(mine + pending, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pending = PCHAIN_NO_PENDING;
        }
        // ... and, while the loads of the stack's LAST tile are in flight, wave 0 prepares the item after this one: decode, ask for its counters
        if (shadow_now && wv == 0 && __builtin_amdgcn_readfirstlane(L.ctl[CT_NSTACK]) == 0 && __builtin_amdgcn_readfirstlane(L.ctl[CT_PRECLAIMED]) == 0) {
            __builtin_amdgcn_s_setprio(3);
            const PChainDims C = dims();
            const int kind = next_owned(C, slot ^ 1);
            if (kind == PK_RUN) send_poll();
            __builtin_amdgcn_s_setprio(0);
        }
    };
    auto pre_a = [&]() {   // wave 0, between two sweeps: did the counters read "all there" when the tile started?  If not, ask again
        if (wv != 0) return;
        bool ready = false;
        if (__builtin_amdgcn_readfirstlane(L.ctl[CT_NSTACK]) == 0 && __builtin_amdgcn_readfirstlane(L.ctl[CT_PRECLAIMED]) == 0 &&
            __builtin_amdgcn_readfirstlane(L.item[slot ^ 1][PG_KIND]) == PK_RUN) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the LDS-DMA landed long ago; nothing else of this wave is in flight between sweeps)
            ready = poll_ok();
            if (!ready) send_poll();
        }
        if (lane == 0) L.ctl[CT_READY] = ready ? 1 : 0;
    };
    auto pre_b = [&]() {   // one trip later: the second answer
        if (wv != 0) return;
        if (__builtin_amdgcn_readfirstlane(L.ctl[CT_READY]) == 0 && __builtin_amdgcn_readfirstlane(L.ctl[CT_NSTACK]) == 0 &&
            __builtin_amdgcn_readfirstlane(L.ctl[CT_PRECLAIMED]) == 0 && __builtin_amdgcn_readfirstlane(L.item[slot ^ 1][PG_KIND]) == PK_RUN) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (poll_ok() && lane == 0) L.ctl[CT_READY] = 1;
        }
    };
    auto may_store = [&]() -> bool { return __builtin_amdgcn_readfirstlane(L.ctl[CT_CLAIMED]) == 0; };   // (parked behind the loads; a barrier of sweeps ago)

    // ---- the first item: through the control path (nothing is ready yet) ----
    if (wv == 0) {
        const PChainDims C = dims();
        (void)next_owned(C, slot);
    }
    __syncthreads();
    bool fresh = true;   // item[slot] was decoded but its dependencies have not been looked at (or were not there): the control path decides
    while (true) {
        if (fresh) {
            if (wv == 0) {
                // ---- the CONTROL path (wave 0 only: no barrier in here; lanes talk through ballots): make item[slot] runnable ----
                __builtin_amdgcn_s_setprio(3);
                const PChainDims C = dims();
                int nstack = __builtin_amdgcn_readfirstlane(L.ctl[CT_NSTACK]);
                int preclaimed = __builtin_amdgcn_readfirstlane(L.ctl[CT_PRECLAIMED]);
                unsigned long long t_wait = 0, t_err = 0;
                bool waiting = false;
                while (true) {
                    int kind = __builtin_amdgcn_readfirstlane(L.item[slot][PG_KIND]);
                    if (kind == PK_DONE) {
                        if (nstack == 0) break;
                        // back to the item that was put aside
                        const int e = __builtin_amdgcn_readfirstlane(L.stack[--nstack]);
                        kind = post_item(C, slot, e >> 24, (e >> 12) & 0xfff, e & 0xfff);
                        preclaimed = 1;   // (it was claimed before it was put aside)
                        waiting = false;
                        continue;
                    }
                    const int l = __builtin_amdgcn_readfirstlane(L.item[slot][PG_L]);
                    if (kind == PK_COUNT_ONLY) {   // nothing to store: claim it, count it, next
                        unsigned int old = 0;
                        if (!preclaimed && lane == 0) old = __hip_atomic_fetch_add(mine + L.item[slot][PG_CLAIM], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
                        if (old == 0 && lane == 0 && L.item[slot][PG_CELL] >= 0) __hip_atomic_fetch_add(mine + L.item[slot][PG_CELL], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        preclaimed = 0;
                        if (nstack > 0) {
                            if (lane == 0) L.item[slot][PG_KIND] = PK_DONE;   // (pops the stack above)
                        } else (void)next_owned(C, slot);
                        continue;
                    }
                    if (l == 0 || DIAG == 3) break;   // runnable
                    // the nine counters (lanes 0..8) and, beside them, the claim words of the nine items they belong to (lanes 16..24)
                    const int by = __builtin_amdgcn_readfirstlane(L.item[slot][PG_BY]), bx = __builtin_amdgcn_readfirstlane(L.item[slot][PG_BX]);
                    bool ok = true;
                    {
                        const int k = lane & 15, r = by - 1 + k / 3, c = bx - 1 + k % 3;
                        if (k < 9 && (lane >> 4) < 2 && r >= 0 && r < C.ny && c >= 0 && c < C.nx) {
                            if (lane < 16) {
                                const int pn = pchain_panel_of(C, c);
                                ok = __hip_atomic_load(mine + pchain_cell(C, l - 1, r, pn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)pchain_panel_width(C, pn);
                            } else {
                                ok = __hip_atomic_load(mine + pchain_claim_word(C, l - 1, r, c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                            }
                        }
                    }
                    const unsigned long long bad = __ballot(!ok);   // wave-uniform
                    if ((bad & 0xffffull) == 0) break;              // every counter is there: runnable
                    if (bad & 0xffff0000ull) {
                        // an item this one waits for has not been claimed by anybody: claim it, put this one aside, run that one first — a
                        // workgroup only ever SPINS on items that somebody runs
                        const int k = __builtin_ctzll(bad & 0xffff0000ull) - 16, r = by - 1 + k / 3, c = bx - 1 + k % 3;
                        unsigned int old = 1;
                        if (lane == 0) old = __hip_atomic_fetch_add(mine + pchain_claim_word(C, l - 1, r, c), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
                        if (old == 0) {   // ours
                            if (!preclaimed) {   // this item must be claimed before it is put aside (else its owner and this workgroup could both run it)
                                unsigned int mineold = 0;
                                if (lane == 0) mineold = __hip_atomic_fetch_add(mine + L.item[slot][PG_CLAIM], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                mineold = (unsigned)__builtin_amdgcn_readfirstlane((int)mineold);
                                if (mineold == 0) {
                                    if (lane == 0) L.stack[nstack] = (l << 24) | (by << 12) | bx;
                                    nstack++;
                                }   // (else somebody else runs it: forget it)
                            } else {
                                if (lane == 0) L.stack[nstack] = (l << 24) | (by << 12) | bx;
                                nstack++;
                            }
                            (void)post_item(C, slot, l - 1, r, c);
                            preclaimed = 1;
                        }
                        waiting = false;
                        continue;
                    }
                    // everything waited for is being run by somebody: spin — bounded in wall-clock time, and nobody waits once somebody has given up
                    if (!waiting) {
                        waiting = true;
                        t_wait = __builtin_amdgcn_s_memrealtime();
                    }
                    __builtin_amdgcn_s_sleep(1);
                    const unsigned long long waited = __builtin_amdgcn_s_memrealtime() - t_wait;
                    bool go = false;
                    if (waited > 10000ull && waited - t_err > 10000ull) {   // (err lives in HOST memory: a PCIe round trip — only after 100 us, every 100 us)
                        t_err = waited;
                        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) go = true;
                    }
                    if (waited > (unsigned long long)C.timeout) {
                        if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        go = true;   // run on stale data rather than hang the device; the host learns through err
                    }
                    if (go) break;
                }
                if (lane == 0) {
                    L.ctl[CT_NSTACK] = nstack;
                    L.ctl[CT_PRECLAIMED] = preclaimed;
                    L.ctl[CT_READY] = 0;
                }
                __builtin_amdgcn_s_setprio(0);
            }
            __syncthreads();
            fresh = false;
        }
        const int kind = __builtin_amdgcn_readfirstlane(L.item[slot][PG_KIND]);
        if (kind == PK_DONE) break;
        // ---- run item[slot] (PK_RUN) ----
        {
            const int l = __builtin_amdgcn_readfirstlane(L.item[slot][PG_L]);
            Win wl = w;
            wl.x0 = __builtin_amdgcn_readfirstlane(sP.xa[l]);
            wl.x1 = __builtin_amdgcn_readfirstlane(sP.xb[l]);
            float* const p = (l & 1) ? pb : pa;
            float* const p_out = (l & 1) ? pa : pb;
            const int bytes = (int)((size_t)w.rows * (size_t)w.P * sizeof(float));
            const float ps = l == 0 ? pscale : 1.0f;
            const int iters = __builtin_amdgcn_readfirstlane(sP.iters[l]), stack = __builtin_amdgcn_readfirstlane(sP.d.stack);
            const bool preclaimed = __builtin_amdgcn_readfirstlane(L.ctl[CT_PRECLAIMED]) != 0;
            if (preclaimed && wv == 0 && lane == 0) L.ctl[CT_CLAIMED] = 0;   // (visible behind the first sweep's barrier, long before the stores)
            for (int t = 0; t < stack; t++) {
                const int* g = L.item[slot];
                const int x0 = __builtin_amdgcn_readfirstlane(g[PG_X0]), y0s = __builtin_amdgcn_readfirstlane(g[PG_Y0S]);
                const int st_lo = __builtin_amdgcn_readfirstlane(g[PG_ST_LO]), st_hi = __builtin_amdgcn_readfirstlane(g[PG_ST_HI]);
                const int sx_lo = __builtin_amdgcn_readfirstlane(g[PG_SX_LO]), sx_hi = __builtin_amdgcn_readfirstlane(g[PG_SX_HI]);
                const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000);
                const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(p_out, 0, bytes, 0x00020000);
                const bool xedge = (x0 <= 0) || (x0 + G::TX >= w.W), ragged = (w.W & 3) != 0 && x0 + G::TX >= w.W;
                int yt, a, b;
                bool last;
                stack_tile_rows(y0s, t, G::TY, HY, w.H, st_lo, st_hi, yt, a, b, last);
                const int cin = t > 0 ? CARRY0 + ((t - 1) & 1) * HY * 64 : -1, cout = last ? -1 : CARRY0 + (t & 1) * HY * 64;
                const bool yedge = (yt <= 0) || (yt + G::TY >= w.H);
                unsigned int* const claim = (t == 0 && !preclaimed) ? mine + __builtin_amdgcn_readfirstlane(g[PG_CLAIM]) : nullptr;
                if (t > 0) __syncthreads();   // the last sweep's mailbox readers are through before the next tile's first sweep publishes
                shadow_now = last;
                if (yedge || ragged)
                    jacobi_stack_tile<NW, RY, HX, HY, 2, true>(wl, rin, rout, p, p_out, div, ps, iters, x0, yt, a, b, sx_lo, sx_hi, sm, cin, cout, claim, &L.ctl[CT_CLAIMED], last, after_loads, pre_a, pre_b, may_store);
                else if (xedge)
                    jacobi_stack_tile<NW, RY, HX, HY, 1, true>(wl, rin, rout, p, p_out, div, ps, iters, x0, yt, a, b, sx_lo, sx_hi, sm, cin, cout, claim, &L.ctl[CT_CLAIMED], last, after_loads, pre_a, pre_b, may_store);
                else
                    jacobi_stack_tile<NW, RY, HX, HY, 0, true>(wl, rin, rout, p, p_out, div, ps, iters, x0, yt, a, b, sx_lo, sx_hi, sm, cin, cout, claim, &L.ctl[CT_CLAIMED], last, after_loads, pre_a, pre_b, may_store);
                if (last) break;
            }
        }
        // The stores are out (or were left out: the claim was lost).  READY: on to the next item's loads, this one is counted behind them.
        const bool stored = may_store();
        const bool helping = __builtin_amdgcn_readfirstlane(L.ctl[CT_NSTACK]) != 0 || __builtin_amdgcn_readfirstlane(L.ctl[CT_PRECLAIMED]) != 0;
        if (!helping && __builtin_amdgcn_readfirstlane(L.ctl[CT_READY]) != 0) {
            pending = stored ? __builtin_amdgcn_readfirstlane(L.item[slot][PG_CELL]) : PCHAIN_NO_PENDING;
            if (!stored) __syncthreads();   // (no count barrier will follow in after_loads: the mailbox hazard needs one)
            slot ^= 1;
            continue;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores, then ONE lane counts the item (the guide's R1 hand-off)
        __syncthreads();
        if (wv == 0 && lane == 0) {
            const int cell = L.item[slot][PG_CELL];
            if (stored && cell >= 0) __hip_atomic_fetch_add(mine + cell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (helping) {
                L.ctl[CT_PRECLAIMED] = 0;
                L.item[slot][PG_KIND] = PK_DONE;   // (the control path pops the stack, or — nothing there — takes the next owned position)
                if (L.ctl[CT_NSTACK] == 0) L.item[slot][PG_KIND] = -1;
            }
        }
        __syncthreads();
        if (helping) {
            if (__builtin_amdgcn_readfirstlane(L.item[slot][PG_KIND]) == -1) {   // the helped item was the last thing put aside... nothing: go on with the owned positions
                if (wv == 0) {
                    const PChainDims C = dims();
                    (void)next_owned(C, slot);
                }
                __syncthreads();
            }
        } else {
            slot ^= 1;   // the item decoded behind this one's loads — not ready then: the control path looks again
        }
        fresh = true;
    }
}

// BPC = workgroups that must fit on a CU together (their load / compute / store phases overlap each other):
// the second __launch_bounds__ argument is waves per SIMD, i.e. the VGPR budget the compiler has to meet.
template <int NW, int RY, int HX, int HY, int BPC>
__global__ void __launch_bounds__(64 * NW, (BPC * NW + 3) / 4) k_jacobi_tb(Win w, const float* __restrict__ p, const float* __restrict__ div,
                                                        float* __restrict__ p_out, float pscale, int iters, int ga, int gb,
                                                        int xs, int ys, int nx, int ny, int remap)
{
    __shared__ float4 mail[2][NW][2][64];
    jacobi_tb_tile<NW, RY, HX, HY>(w, p, div, p_out, pscale, iters, ga, gb, xs, ys, nx, ny, (int)blockIdx.x, remap, mail);
}

// A launch whose FIRST and / or LAST tiles are smaller (round 3).  The time of a launch is its bytes over the bandwidth plus the latency of
// the last tiles' iterations, which nothing overlaps any more (4096-wide grids of every height: 18.5 us + 8.2 ns per row for ten iterations
// against 3.3 us + 7.1 ns per row for one, tools/jacobi_tail_probe.py): the bulk of the rows takes the big tile (least apron traffic), the
// last rows a tile with fewer rows per wave, whose ten iterations drain sooner.  Up to three row segments, each tiled with the big (RYA) or
// the small (RYB) shape — so many band launches in one, dispatched in that order.
struct MixSegs {
    int n;
    int g[4];      // segment k covers rows [g[k], g[k + 1])
    int small[3];  // tiled with RYB (1) or RYA (0) rows per wave
    int ys[3], ny[3], blk0[4];
};

template <int NW, int RYA, int RYB, int HX, int HY, int BPC>
__global__ void __launch_bounds__(64 * NW, (BPC * NW + 3) / 4) k_jacobi_tb_mix(Win w, const float* __restrict__ p, const float* __restrict__ div,
                                                            float* __restrict__ p_out, float pscale, int iters, MixSegs S, int xs, int nx,
                                                            int remap)
{
    __shared__ float4 mail[2][NW][2][64];
    const int b = (int)blockIdx.x;
    int k = 0;
    while (k + 1 < S.n && b >= S.blk0[k + 1]) k++;
    if (S.small[k]) jacobi_tb_tile<NW, RYB, HX, HY>(w, p, div, p_out, pscale, iters, S.g[k], S.g[k + 1], xs, S.ys[k], nx, S.ny[k], b - S.blk0[k], remap, mail);
    else jacobi_tb_tile<NW, RYA, HX, HY>(w, p, div, p_out, pscale, iters, S.g[k], S.g[k + 1], xs, S.ys[k], nx, S.ny[k], b - S.blk0[k], remap, mail);
}

// The LAST launch of a step's loop with the gradient subtract folded in (jacobi_tb_body, GS): the row apron is HY + 1 so that `iters`
// <= HY iterations leave the pressure exact one ring beyond the texels the tile stores; reads and writes the velocity for those texels.
template <int NW, int RY, int HX, int HY, int BPC>
__global__ void __launch_bounds__(64 * NW, (BPC * NW + 3) / 4) k_jacobi_tb_gs(Win w, const float* __restrict__ p, const float* __restrict__ div,
                                                           float* __restrict__ p_out, const float2* __restrict__ vel,
                                                           float2* __restrict__ vel_out, float pscale, int iters, int ga, int gb, int xs,
                                                           int ys, int nx, int ny, int remap)
{
    __shared__ float4 mail[2][NW][2][64];
    jacobi_tb_tile<NW, RY, HX, HY + 1, true>(w, p, div, p_out, pscale, iters, ga, gb, xs, ys, nx, ny, (int)blockIdx.x, remap, mail, vel, vel_out);
}

template <int NW, int RY, int HX, int HY, int BPC>
__global__ void __launch_bounds__(64 * NW, (BPC * NW + 3) / 4) k_jacobi_tb_gs_h(Win w, const __half* __restrict__ p, const __half* __restrict__ div,
                                                             __half* __restrict__ p_out, const __half2* __restrict__ vel,
                                                             __half2* __restrict__ vel_out, float pscale, int iters, int ga, int gb, int xs,
                                                             int ys, int nx, int ny, int remap)
{
    __shared__ float4 mail[2][NW][2][64];
    jacobi_tb_tile<NW, RY, HX, HY + 1, true>(w, p, div, p_out, pscale, iters, ga, gb, xs, ys, nx, ny, (int)blockIdx.x, remap, mail, vel, vel_out);
}

// the same tile on fp16-storage fields (FLUID_STORE_F16): half the bytes per launch; every iteration's output is rounded to
// fp16 in registers, exactly where the reference's per-iteration render into a half-float texture rounds it
template <int NW, int RY, int HX, int HY, int BPC>
__global__ void __launch_bounds__(64 * NW, (BPC * NW + 3) / 4) k_jacobi_tb_h(Win w, const __half* __restrict__ p, const __half* __restrict__ div,
                                                          __half* __restrict__ p_out, float pscale, int iters, int ga, int gb,
                                                          int xs, int ys, int nx, int ny, int remap)
{
    __shared__ float4 mail[2][NW][2][64];
    jacobi_tb_tile<NW, RY, HX, HY>(w, p, div, p_out, pscale, iters, ga, gb, xs, ys, nx, ny, (int)blockIdx.x, remap, mail);
}

// ------------------------------------------------------------------------------------------------
// The temporally blocked Jacobi tile with TWO texels per lane: 128 columns x NW * RY rows (round 3, small grids).  At 1024^2 the four-texel
// tile makes 260 workgroups — one per CU, two waves per SIMD — and each of them runs its ten iterations as a latency chain (0.59 us an
// iteration: profiles/r03/jacobi_iter_cost.txt); half the columns per workgroup are twice the workgroups with half the arithmetic per row
// (6 instructions for two texels: two v_add_f32_dpp + four packed), three of them per CU.  The price is the column apron (12 of 128
// columns a side instead of 12 of 256) — arithmetic a latency-bound launch has to spare, which is why only grids below 1280^2 texels take
// this shape (jacobi_tb_pick).  Same operand order per texel as jacobi_row, hence the same bits.  fp32 fields only (fp16 storage keeps the
// four-texel tile).
template <int NW, int RY, int HX, int HY>
struct JacobiTB2 {
    static constexpr int TX = 128, TY = NW * RY;
    static constexpr int VX = TX - 2 * HX, VY = TY - 2 * HY;
    static_assert(HX % 2 == 0 && HX >= HY && VX > 0 && VY > 0, "pair tile geometry");
};

// EDGE as in jacobi_row; nv = texels of the lane's pair inside the domain (1 in the lane that holds column W - 1 of an odd width)
template <int EDGE>
__device__ __forceinline__ v2f jacobi_row2(v2f C, v2f T, v2f B, const v2f D, int gj, int H, bool at_left, int nv)
{
    if (EDGE == 2 && nv < 2) C.y = C.x;  // CLAMP_TO_EDGE inside the partly padded last pair
    float L = from_left_lane(C.y);   // column 2*lane - 1
    float R = from_right_lane(C.x);  // column 2*lane + 2
    if (EDGE) {
        if (at_left) L = C.x;
        if (nv <= 2) R = C.y;  // the lane that holds column W - 1 (lanes beyond it only feed texels outside the domain)
    }
    if (EDGE == 2) {
        if (gj == 0) B = C;
        if (gj == H - 1) T = C;
    }
    float h0 = L + C.y;  // texel 0: left + right
    float h1 = C.x + R;  // texel 1
    if (!EDGE) {  // scalar, so that the lane shift folds into the add (v_add_f32_dpp), as in jacobi_row
        asm("" : "+v"(h0));
        asm("" : "+v"(h1));
    }
    const v2f quarter = v2f{ 0.25f, 0.25f };
    return (v2f{ h0, h1 } + B + T - D) * quarter;
}

template <int NW, int RY, int EDGE>
__device__ __forceinline__ void jacobi_sweep2(v2f (&P)[RY], const v2f (&D)[RY], v2f (*box)[2][64], int wv, int lane, int gy, int H, bool at_left,
                                              int nv)
{
    static_assert(RY >= 3, "a wave needs an inner row");
    box[wv][0][lane] = P[0];
    box[wv][1][lane] = P[RY - 1];
    const v2f old0 = P[0], old1 = P[1];
    v2f below = old0;
#pragma unroll
    for (int r = 1; r < RY - 1; r++) {
        const v2f C = P[r];
        P[r] = jacobi_row2<EDGE>(C, P[r + 1], below, D[r], gy + r, H, at_left, nv);
        below = C;
    }
    __syncthreads();
    const v2f lo = box[wv > 0 ? wv - 1 : 0][1][lane];
    const v2f hi = box[wv < NW - 1 ? wv + 1 : NW - 1][0][lane];
    P[RY - 1] = jacobi_row2<EDGE>(P[RY - 1], hi, below, D[RY - 1], gy + RY - 1, H, at_left, nv);
    P[0] = jacobi_row2<EDGE>(old0, old1, lo, D[0], gy, H, at_left, nv);
}

typedef float pair_f __attribute__((ext_vector_type(2), aligned(8)));
template <int NW, int RY, int HX, int HY, int EDGE, bool GS>
__device__ __forceinline__ void jacobi_tb2_body(const Win& w, const float* __restrict__ p, const float* __restrict__ div, float* __restrict__ p_out,
                                                float pscale, int iters, int ga, int gb, int x0, int y0, v2f (*mail)[NW][2][64],
                                                const float2* __restrict__ vel, float2* __restrict__ vel_out)
{
    using G = JacobiTB2<NW, RY, HX, HY>;
    const int lane = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int cx = x0 + 2 * lane, gy = y0 + wv * RY;
    v2f P[RY], D[RY];
    const unsigned cxs = (unsigned)(min(max(cx, w.c0), w.c0 + w.P - 2) - w.c0);  // array column of the lane's pair, kept inside the array
    const v2f ps = v2f{ pscale, pscale };
#pragma unroll
    for (int r = 0; r < RY; r++) {  // unconditional loads from clamped addresses, as in jacobi_tb_body
        const int lr = min(max(gy + r - w.g0, 0), w.rows - 1);
        const size_t row = (size_t)lr * (size_t)w.P;
        P[r] = *reinterpret_cast<const pair_f*>(p + row + cxs);
        D[r] = *reinterpret_cast<const pair_f*>(div + row + cxs);
    }
#pragma unroll
    for (int r = 0; r < RY; r++) P[r] = ps * P[r];  // clearShader folded in

    const bool at_left = (cx == 0);
    const int nv = w.W - cx;
    int it = 0;
    for (; it + 2 <= iters; it += 2) {
        jacobi_sweep2<NW, RY, EDGE>(P, D, mail[0], wv, lane, gy, w.H, at_left, nv);
        jacobi_sweep2<NW, RY, EDGE>(P, D, mail[1], wv, lane, gy, w.H, at_left, nv);
    }
    if (it < iters) jacobi_sweep2<NW, RY, EDGE>(P, D, mail[0], wv, lane, gy, w.H, at_left, nv);

    int xa, xb, out_lo, out_hi;
    tile_exact(x0, G::TX, HX, w.W, w.x0, w.x1, xa, xb);
    tile_exact(y0, G::TY, HY, w.H, ga, gb, out_lo, out_hi);
    const bool col_store = (cx >= xa) && (cx < xb);
#pragma unroll
    for (int r = 0; r < RY; r++) {
        const int gj = gy + r;
        if (col_store && gj >= out_lo && gj < out_hi) *reinterpret_cast<pair_f*>(p_out + (size_t)at(w, gj, cx)) = P[r];
    }

    if constexpr (GS) {  // K6 on the final pressure, as in jacobi_tb_body
        v2f (*box)[2][64] = mail[iters & 1];
        box[wv][0][lane] = P[0];
        box[wv][1][lane] = P[RY - 1];
        __syncthreads();
        const v2f lo = box[wv > 0 ? wv - 1 : 0][1][lane];
        const v2f hi = box[wv < NW - 1 ? wv + 1 : NW - 1][0][lane];
        float4 va[RY];
#pragma unroll
        for (int r = 0; r < RY; r++) {
            const int gj = gy + r;
            if (gj >= out_lo && gj < out_hi && col_store) va[r] = *reinterpret_cast<const float4*>(vel + (size_t)at(w, gj, cx));
        }
#pragma unroll
        for (int r = 0; r < RY; r++) {
            const int gj = gy + r;
            if (gj >= out_lo && gj < out_hi) {  // wave-uniform
                v2f C = P[r];
                v2f Tq = r < RY - 1 ? P[r < RY - 1 ? r + 1 : r] : hi;
                v2f Bq = r > 0 ? P[r > 0 ? r - 1 : 0] : lo;
                if (EDGE == 2 && nv < 2) C.y = C.x;
                float L = from_left_lane(C.y), R = from_right_lane(C.x);  // all lanes active here
                if (EDGE) {
                    if (at_left) L = C.x;
                    if (nv <= 2) R = C.y;
                }
                if (EDGE == 2) {
                    if (gj == 0) Bq = C;
                    if (gj == w.H - 1) Tq = C;
                }
                if (col_store) {
                    float4 o;
                    o.x = va[r].x - (C.y - L);
                    o.y = va[r].y - (Tq.x - Bq.x);
                    o.z = va[r].z - (R - C.x);
                    o.w = va[r].w - (Tq.y - Bq.y);
                    *reinterpret_cast<float4*>(vel_out + (size_t)at(w, gj, cx)) = o;
                }
            }
        }
    }
}

template <int NW, int RY, int HX, int HYT, bool GS>
__device__ __forceinline__ void jacobi_tb2_tile(const Win& w, const float* __restrict__ p, const float* __restrict__ div, float* __restrict__ p_out,
                                                float pscale, int iters, int ga, int gb, int xs, int ys, int nx, int ny, int b, int remap,
                                                v2f (*mail)[NW][2][64], const float2* __restrict__ vel = nullptr, float2* __restrict__ vel_out = nullptr)
{
    using G = JacobiTB2<NW, RY, HX, HYT>;
    int bx, by;
    tile_of_block(b, nx, ny, remap, bx, by);
    const int x0 = xs + bx * G::VX, y0 = ys + by * G::VY;
    const bool xedge = (x0 <= 0) || (x0 + G::TX >= w.W), yedge = (y0 <= 0) || (y0 + G::TY >= w.H);
    const bool ragged = (w.W & 1) != 0 && x0 + G::TX >= w.W;  // the tile holds the partly padded last pair
    if (yedge || ragged) jacobi_tb2_body<NW, RY, HX, HYT, 2, GS>(w, p, div, p_out, pscale, iters, ga, gb, x0, y0, mail, vel, vel_out);
    else if (xedge) jacobi_tb2_body<NW, RY, HX, HYT, 1, GS>(w, p, div, p_out, pscale, iters, ga, gb, x0, y0, mail, vel, vel_out);
    else jacobi_tb2_body<NW, RY, HX, HYT, 0, GS>(w, p, div, p_out, pscale, iters, ga, gb, x0, y0, mail, vel, vel_out);
}

template <int NW, int RY, int HX, int HY, int BPC>
__global__ void __launch_bounds__(64 * NW, (BPC * NW + 3) / 4) k_jacobi_tb2(Win w, const float* __restrict__ p, const float* __restrict__ div,
                                                         float* __restrict__ p_out, float pscale, int iters, int ga, int gb, int xs, int ys,
                                                         int nx, int ny, int remap)
{
    __shared__ v2f mail[2][NW][2][64];
    jacobi_tb2_tile<NW, RY, HX, HY, false>(w, p, div, p_out, pscale, iters, ga, gb, xs, ys, nx, ny, (int)blockIdx.x, remap, mail);
}

template <int NW, int RY, int HX, int HY, int BPC>
__global__ void __launch_bounds__(64 * NW, (BPC * NW + 3) / 4) k_jacobi_tb2_gs(Win w, const float* __restrict__ p, const float* __restrict__ div,
                                                            float* __restrict__ p_out, const float2* __restrict__ vel,
                                                            float2* __restrict__ vel_out, float pscale, int iters, int ga, int gb, int xs,
                                                            int ys, int nx, int ny, int remap)
{
    __shared__ v2f mail[2][NW][2][64];
    jacobi_tb2_tile<NW, RY, HX, HY + 1, true>(w, p, div, p_out, pscale, iters, ga, gb, xs, ys, nx, ny, (int)blockIdx.x, remap, mail, vel, vel_out);
}

// ------------------------------------------------------------------------------------------------
// K6 gradient subtract, four texels per lane (fused schedule): the per-texel kernel issues six memory instructions for
// 20 bytes; here a wave moves one 256-texel row segment with five 16-byte loads and two 16-byte stores per lane, the
// horizontal pressure neighbours come from the lane's own float4 and the adjacent lanes (DPP), and only the two
// lanes at the segment ends fetch their outside neighbour.  Same subtraction per texel, hence the same bits.
template <class S1, class V2>
__device__ __forceinline__ void gradsub4_body(const Win& w, const S1* __restrict__ p, const V2* __restrict__ vel, V2* __restrict__ vel_out, int ga,
                                              int gb)
{
    const int lane = threadIdx.x;
    const int gj = ga + (int)blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int cx = w.x0 + (int)blockIdx.x * 256 + 4 * lane;  // x0, x1 multiples of 4 (the launcher rounds them outward)
    if (gj >= gb || cx >= w.x1) return;
    const int ax = cx - w.c0;  // array column of the lane's quad
    const size_t rowC = (size_t)(gj - w.g0) * (size_t)w.P;
    const size_t rowT = (size_t)(min(gj + 1, w.H - 1) - w.g0) * (size_t)w.P;
    const size_t rowB = (size_t)(max(gj - 1, 0) - w.g0) * (size_t)w.P;
    float4 C = load_s4(p, rowC + ax);
    const float4 T = load_s4(p, rowT + ax);
    const float4 B = load_s4(p, rowB + ax);
    float4 va, vb;
    load_v4(vel, rowC + ax, va, vb);
    const int nv = w.W - cx;  // texels of this quad inside the domain (< 4 only in the last quad of a width that is not a multiple of 4)
    if (nv < 4) {             // CLAMP_TO_EDGE inside the quad: the texels beyond the edge repeat the last one
        if (nv < 2) C.y = C.x;
        if (nv < 3) C.z = C.y;
        C.w = C.z;
    }
    float L = from_left_lane(C.w), R = from_right_lane(C.x);
    // the two lanes at the segment ends fetch their outside neighbour: CLAMP_TO_EDGE at the domain border, and never outside the array
    // (a tile array holds owned + ghost columns only; a launch that starts at its first column has no valid output there anyway)
    if (lane == 0) L = (cx > 0 && ax > 0) ? ld(p, (long)(rowC + ax - 1)) : C.x;
    if (lane == 63 || cx + 4 >= w.x1) R = (cx + 4 < w.W && ax + 4 < w.P) ? ld(p, (long)(rowC + ax + 4)) : C.w;
    float4 oa, ob;
    oa.x = va.x - (C.y - L);
    oa.y = va.y - (T.x - B.x);
    oa.z = va.z - (C.z - C.x);
    oa.w = va.w - (T.y - B.y);
    ob.x = vb.x - (C.w - C.y);
    ob.y = vb.y - (T.z - B.z);
    ob.z = vb.z - (R - C.z);
    ob.w = vb.w - (T.w - B.w);
    store_v4(vel_out, rowC + ax, oa, ob);
}

__global__ void __launch_bounds__(256) k_gradsub4(Win w, const float* __restrict__ p, const float2* __restrict__ vel,
                                                   float2* __restrict__ vel_out, int ga, int gb)
{
    gradsub4_body(w, p, vel, vel_out, ga, gb);
}

__global__ void __launch_bounds__(256) k_gradsub4_h(Win w, const __half* __restrict__ p, const __half2* __restrict__ vel,
                                                     __half2* __restrict__ vel_out, int ga, int gb)
{
    gradsub4_body(w, p, vel, vel_out, ga, gb);
}

// ------------------------------------------------------------------------------------------------
// Fused K1 + K2 + K3: curl -> vorticity confinement -> divergence in one trip through HBM
// (reads velocity once: 8 B/texel; writes curl 4, velocity 8, divergence 4 = 24 B/texel instead of
// 12 + 20 + 12 = 44).  Same register-tile layout as the Jacobi kernel: lane l of wave wv holds 4
// consecutive texels (two float4 = four RG pairs) of RY rows; x neighbours by full-wave DPP shifts,
// y neighbours in registers, wave-boundary rows through LDS mailboxes (one barrier per stage).
// Each stage shrinks the exact region by one ring, so the tile carries a 3-row / 4-column apron
// (4 keeps float4 alignment) and stores only its interior.  The per-texel arithmetic is the same
// device code as the single-pass kernels, so the result is bit-identical to running them in turn.
// Tile shape of the fused curl/vorticity/divergence kernel: 8 waves x 5 rows at <= 128 VGPRs, so that TWO workgroups
// share a CU and one's loads overlap the other's arithmetic (69 VALU instructions per texel: two IEEE divides and a
// square root).  Measured 85 us at 4096^2 against 99 us for 8 x 8 rows at 229 VGPRs / one workgroup per CU, although
// the smaller tile re-reads more apron rows (profiles/r01/cvd_tile_shapes.txt).
#ifndef VD_WAVES_PER_EU
#define VD_WAVES_PER_EU 4
#endif
template <int NW, int RY>
struct VortDiv {
    static constexpr int TX = 256, TY = NW * RY, AX = 4, AY = 3;
    static constexpr int VX = TX - 2 * AX, VY = TY - 2 * AY;
};

struct Row4 {  // four texels of one row held by a lane
    float x[4], y[4];
};

// EDGE: 0 interior, 1 left / right border only (width a multiple of 4), 2 everything — as for the Jacobi kernel
template <int NW, int RY, int EDGE, class V2, class S1>
__device__ __forceinline__ void vort_div_body(const Win& w, const V2* __restrict__ vel, S1* __restrict__ curl_out,
                                              V2* __restrict__ vel_out, S1* __restrict__ div_out, float curl_strength,
                                              float dt, int ga, int gb, int x0, int y0, float4 (*mail)[2][2][64])
{
    using G = VortDiv<NW, RY>;
    constexpr bool EX = EDGE >= 1, EY = EDGE == 2;  // selects for the left / right border; for the bottom / top rows and the padded quad
    const int lane = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.y);  // wave-uniform: row addressing on the SALU
    const int cx = x0 + 4 * lane;
    const int gy = y0 + wv * RY;
    const int cxs = min(max(cx, w.c0), w.c0 + w.P - 4) - w.c0;  // array column of the lane's quad, kept inside the array
    const int nv = w.W - cx;                                   // texels of this quad inside the domain
    const bool at_left = (cx == 0), at_right = (nv >= 1 && nv <= 4);  // the lane that holds column W - 1
    const int wb = wv > 0 ? wv - 1 : 0, wa = wv < NW - 1 ? wv + 1 : NW - 1;

    // ---- load velocity (unconditional, clamped addresses; see the Jacobi kernel) ----
    Row4 V[RY];
#pragma unroll
    for (int r = 0; r < RY; r++) {
        const int lr = min(max(gy + r - w.g0, 0), w.rows - 1);
        float4 a, b;
        load_v4(vel, (size_t)((long)lr * w.P + cxs), a, b);
        V[r].x[0] = a.x; V[r].y[0] = a.y; V[r].x[1] = a.z; V[r].y[1] = a.w;
        V[r].x[2] = b.x; V[r].y[2] = b.y; V[r].x[3] = b.z; V[r].y[3] = b.w;
        if (EY && nv < 4) {  // width not a multiple of 4: the quad's texels beyond column W - 1 repeat it (CLAMP_TO_EDGE inside the quad)
#pragma unroll
            for (int k = 1; k < 4; k++)
                if (k >= nv) {
                    V[r].x[k] = V[r].x[k - 1];
                    V[r].y[k] = V[r].y[k - 1];
                }
        }
    }

    // ---- stage 1: curl (needs vx of the rows above/below, vy of the columns left/right) ----
    mail[wv][0][0][lane] = make_float4(V[0].x[0], V[0].x[1], V[0].x[2], V[0].x[3]);
    mail[wv][1][0][lane] = make_float4(V[RY - 1].x[0], V[RY - 1].x[1], V[RY - 1].x[2], V[RY - 1].x[3]);
    __syncthreads();
    const float4 vxb = mail[wb][1][0][lane], vxa = mail[wa][0][0][lane];
    float C[RY][4];
#pragma unroll
    for (int r = 0; r < RY; r++) {
        const int gj = gy + r;
        float Lq = from_left_lane(V[r].y[3]), Rq = from_right_lane(V[r].y[0]);
        if (EX) {
            if (at_left) Lq = V[r].y[0];
            if (at_right) Rq = V[r].y[3];
        }
        const float bx[4] = { r > 0 ? V[r - 1].x[0] : vxb.x, r > 0 ? V[r - 1].x[1] : vxb.y, r > 0 ? V[r - 1].x[2] : vxb.z,
                              r > 0 ? V[r - 1].x[3] : vxb.w };
        const float tx[4] = { r < RY - 1 ? V[r + 1].x[0] : vxa.x, r < RY - 1 ? V[r + 1].x[1] : vxa.y,
                              r < RY - 1 ? V[r + 1].x[2] : vxa.z, r < RY - 1 ? V[r + 1].x[3] : vxa.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float L = k > 0 ? V[r].y[k - 1] : Lq;
            const float R = k < 3 ? V[r].y[k + 1] : Rq;
            float T = tx[k], B = bx[k];
            if (EY) {
                if (gj == 0) B = V[r].x[k];
                if (gj == w.H - 1) T = V[r].x[k];
            }
            const float vort = R - L - T + B;
            C[r][k] = kept(curl_out, 0.5f * vort);  // the vorticity pass reads the curl TEXTURE: fp16 storage rounds it here
        }
        if (EY && nv < 4) {
#pragma unroll
            for (int k = 1; k < 4; k++)
                if (k >= nv) C[r][k] = C[r][k - 1];
        }
    }

    // ---- stage 2: vorticity confinement (needs curl of the four neighbours) ----
    mail[wv][0][1][lane] = make_float4(C[0][0], C[0][1], C[0][2], C[0][3]);
    mail[wv][1][1][lane] = make_float4(C[RY - 1][0], C[RY - 1][1], C[RY - 1][2], C[RY - 1][3]);
    __syncthreads();
    const float4 cb = mail[wb][1][1][lane], ca = mail[wa][0][1][lane];
    Row4 N[RY];
#pragma unroll
    for (int r = 0; r < RY; r++) {
        const int gj = gy + r;
        float Lq = from_left_lane(C[r][3]), Rq = from_right_lane(C[r][0]);
        if (EX) {
            if (at_left) Lq = C[r][0];
            if (at_right) Rq = C[r][3];
        }
        const float bc[4] = { r > 0 ? C[r - 1][0] : cb.x, r > 0 ? C[r - 1][1] : cb.y, r > 0 ? C[r - 1][2] : cb.z, r > 0 ? C[r - 1][3] : cb.w };
        const float tc[4] = { r < RY - 1 ? C[r + 1][0] : ca.x, r < RY - 1 ? C[r + 1][1] : ca.y, r < RY - 1 ? C[r + 1][2] : ca.z,
                              r < RY - 1 ? C[r + 1][3] : ca.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float L = k > 0 ? C[r][k - 1] : Lq;
            const float R = k < 3 ? C[r][k + 1] : Rq;
            float T = tc[k], B = bc[k];
            if (EY) {
                if (gj == 0) B = C[r][k];
                if (gj == w.H - 1) T = C[r][k];
            }
            float2 conf = vorticity_cell(L, R, T, B, C[r][k], make_float2(V[r].x[k], V[r].y[k]), curl_strength, dt);
            asm volatile("" : "+v"(conf.x), "+v"(conf.y));  // finished HERE: left alone, the tail of a divide is sunk into the store branch of stage 3 and its operand spilled on the way (96 VGPRs, no scratch, against 128 + 8 B: profiles/r03/cvd_pin_ab.txt)
            N[r].x[k] = kept(vel_out, conf.x);  // likewise the divergence pass reads the stored velocity
            N[r].y[k] = kept(vel_out, conf.y);
        }
    }

    // ---- stage 3: divergence of the new velocity (vy of the rows above/below, vx of the columns left/right) ----
    __syncthreads();  // everyone is done reading the stage-1 mailboxes (slot 0) before they are reused
    mail[wv][0][0][lane] = make_float4(N[0].y[0], N[0].y[1], N[0].y[2], N[0].y[3]);
    mail[wv][1][0][lane] = make_float4(N[RY - 1].y[0], N[RY - 1].y[1], N[RY - 1].y[2], N[RY - 1].y[3]);
    __syncthreads();
    const float4 nyb = mail[wb][1][0][lane], nya = mail[wa][0][0][lane];

    int xa, xb, out_lo, out_hi;
    tile_exact(x0, G::TX, G::AX, w.W, w.x0, w.x1, xa, xb);
    tile_exact(y0, G::TY, G::AY, w.H, ga, gb, out_lo, out_hi);
    const bool col_store = (cx >= xa) && (cx < xb);
#pragma unroll
    for (int r = 0; r < RY; r++) {
        const int gj = gy + r;
        float Lq = from_left_lane(N[r].x[3]), Rq = from_right_lane(N[r].x[0]);
        const float by[4] = { r > 0 ? N[r - 1].y[0] : nyb.x, r > 0 ? N[r - 1].y[1] : nyb.y, r > 0 ? N[r - 1].y[2] : nyb.z,
                              r > 0 ? N[r - 1].y[3] : nyb.w };
        const float ty[4] = { r < RY - 1 ? N[r + 1].y[0] : nya.x, r < RY - 1 ? N[r + 1].y[1] : nya.y,
                              r < RY - 1 ? N[r + 1].y[2] : nya.z, r < RY - 1 ? N[r + 1].y[3] : nya.w };
        float dv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float L = k > 0 ? N[r].x[k - 1] : Lq;
            float R = k < 3 ? N[r].x[k + 1] : Rq;
            float T = ty[k], B = by[k];
            if (EX) {  // reflecting walls: an off-domain neighbour is MINUS the centre component (script.js:804-807)
                if (at_left && k == 0) L = -N[r].x[0];
                if (at_right && k == (EY ? nv - 1 : 3)) R = -N[r].x[k];  // the texel in column W - 1
            }
            if (EY) {
                if (gj == w.H - 1) T = -N[r].y[k];
                if (gj == 0) B = -N[r].y[k];
            }
            dv[k] = 0.5f * (R - L + T - B);
        }
        if (col_store && gj >= out_lo && gj < out_hi) {
            const size_t c = (size_t)at(w, gj, cx);
            if (curl_out) store_s4(curl_out, c, make_float4(C[r][0], C[r][1], C[r][2], C[r][3]));  // null: a step of fluid_step_n whose curl field nobody can read (wave-uniform)
            store_s4(div_out, c, make_float4(dv[0], dv[1], dv[2], dv[3]));
            store_v4(vel_out, c, make_float4(N[r].x[0], N[r].y[0], N[r].x[1], N[r].y[1]), make_float4(N[r].x[2], N[r].y[2], N[r].x[3], N[r].y[3]));
        }
    }
}

// tile `b` of an nx x ny tile set over rows [ga, gb): its place in the XCD-aware order, then the body with the border selects it needs
template <int NW, int RY, class V2, class S1>
__device__ __forceinline__ void vort_div_tile(const Win& w, const V2* __restrict__ vel, S1* __restrict__ curl_out, V2* __restrict__ vel_out,
                                              S1* __restrict__ div_out, float curl_strength, float dt, int ga, int gb, int xs, int ys, int nx,
                                              int ny, int b, int remap, float4 (*mail)[2][2][64])
{
    using G = VortDiv<NW, RY>;
    int bx, by;
    tile_of_block(b, nx, ny, remap, bx, by);
    const int x0 = xs + bx * G::VX, y0 = ys + by * G::VY;
    const bool xedge = (x0 <= 0) || (x0 + G::TX >= w.W), yedge = (y0 <= 0) || (y0 + G::TY >= w.H);
    const bool ragged = (w.W & 3) != 0 && x0 + G::TX >= w.W;  // the tile holds the partly padded last quad
    if (yedge || ragged) vort_div_body<NW, RY, 2>(w, vel, curl_out, vel_out, div_out, curl_strength, dt, ga, gb, x0, y0, mail);
    else if (xedge) vort_div_body<NW, RY, 1>(w, vel, curl_out, vel_out, div_out, curl_strength, dt, ga, gb, x0, y0, mail);
    else vort_div_body<NW, RY, 0>(w, vel, curl_out, vel_out, div_out, curl_strength, dt, ga, gb, x0, y0, mail);
}

template <int NW, int RY>
__global__ void __launch_bounds__(64 * NW, VD_WAVES_PER_EU) k_curl_vort_div(Win w, const float2* __restrict__ vel, float* __restrict__ curl_out,
                                                            float2* __restrict__ vel_out, float* __restrict__ div_out,
                                                            float curl_strength, float dt, int ga, int gb, int xs, int ys, int nx, int ny,
                                                            int remap)
{
    __shared__ float4 mail[NW][2][2][64];
    vort_div_tile<NW, RY>(w, vel, curl_out, vel_out, div_out, curl_strength, dt, ga, gb, xs, ys, nx, ny, (int)blockIdx.x, remap, mail);
}

template <int NW, int RY>
__global__ void __launch_bounds__(64 * NW, VD_WAVES_PER_EU) k_curl_vort_div_h(Win w, const __half2* __restrict__ vel, __half* __restrict__ curl_out,
                                                              __half2* __restrict__ vel_out, __half* __restrict__ div_out,
                                                              float curl_strength, float dt, int ga, int gb, int xs, int ys, int nx, int ny,
                                                              int remap)
{
    __shared__ float4 mail[NW][2][2][64];
    vort_div_tile<NW, RY>(w, vel, curl_out, vel_out, div_out, curl_strength, dt, ga, gb, xs, ys, nx, ny, (int)blockIdx.x, remap, mail);
}

struct TileRects {
    int n;
    int x0[4], x1[4], ga[4], gb[4], xs[4], ys[4], nx[4], ny[4], blk0[5];
};

// the fused curl / vorticity / divergence tile kernel over several rectangles in one launch (see k_advect_both_fast_rects)
template <int NW, int RY, class V2, class S1>
__global__ void __launch_bounds__(64 * NW, VD_WAVES_PER_EU) k_curl_vort_div_rects(Win w, TileRects R, const V2* __restrict__ vel, S1* __restrict__ curl_out,
                                                                  V2* __restrict__ vel_out, S1* __restrict__ div_out, float curl_strength,
                                                                  float dt, int remap)
{
    __shared__ float4 mail[NW][2][2][64];
    const int b = (int)blockIdx.x;
    int k = 0;
    while (k + 1 < R.n && b >= R.blk0[k + 1]) k++;
    w.x0 = R.x0[k];
    w.x1 = R.x1[k];
    vort_div_tile<NW, RY>(w, vel, curl_out, vel_out, div_out, curl_strength, dt, R.ga[k], R.gb[k], R.xs[k], R.ys[k], R.nx[k], R.ny[k], b - R.blk0[k], remap, mail);
}

// The temporally blocked Jacobi tile over several rectangles in one launch: the FRAME of a block's first launch around the interior that
// computed while the ghost rows / columns of an exchange were in flight (fluid_solver.cpp pass_jacobi, split 2; fluid_stripes.cpp).  The
// frame of a 4096^2 tile is a few hundred tiles, so the launch is one tile's latency: it takes the 56-row tile (7 rows per wave).
template <int NW, int RY, int HX, int HY, int BPC>
__global__ void __launch_bounds__(64 * NW, (BPC * NW + 3) / 4) k_jacobi_tb_rects(Win w, TileRects R, const float* __restrict__ p, const float* __restrict__ div,
                                                              float* __restrict__ p_out, float pscale, int iters, int remap)
{
    __shared__ float4 mail[2][NW][2][64];
    const int b = (int)blockIdx.x;
    int k = 0;
    while (k + 1 < R.n && b >= R.blk0[k + 1]) k++;
    w.x0 = R.x0[k];
    w.x1 = R.x1[k];
    jacobi_tb_tile<NW, RY, HX, HY>(w, p, div, p_out, pscale, iters, R.ga[k], R.gb[k], R.xs[k], R.ys[k], R.nx[k], R.ny[k], b - R.blk0[k], remap, mail);
}

// the same kernel with smaller tiles for a launch's first and last rows (see k_jacobi_tb_mix: a launch is its bytes over the bandwidth
// plus the fill / drain latency of its first / last tiles, and a tile of this kernel is 1400 VALU instructions per wave)
template <int NW, int RYA, int RYB>
__global__ void __launch_bounds__(64 * NW, VD_WAVES_PER_EU) k_curl_vort_div_mix(Win w, MixSegs S, const float2* __restrict__ vel, float* __restrict__ curl_out,
                                                                float2* __restrict__ vel_out, float* __restrict__ div_out, float curl_strength,
                                                                float dt, int xs, int nx, int remap)
{
    __shared__ float4 mail[NW][2][2][64];
    const int b = (int)blockIdx.x;
    int k = 0;
    while (k + 1 < S.n && b >= S.blk0[k + 1]) k++;
    if (S.small[k]) vort_div_tile<NW, RYB>(w, vel, curl_out, vel_out, div_out, curl_strength, dt, S.g[k], S.g[k + 1], xs, S.ys[k], nx, S.ny[k], b - S.blk0[k], remap, mail);
    else vort_div_tile<NW, RYA>(w, vel, curl_out, vel_out, div_out, curl_strength, dt, S.g[k], S.g[k + 1], xs, S.ys[k], nx, S.ny[k], b - S.blk0[k], remap, mail);
}


// ------------------------------------------------------------------------------------------------
// K7a + K7b of one step and K1 + K2 + K3 of the NEXT in one trip through HBM (fluid_step_n with n > 1, fluid_solver.cpp step_once(lead, chain)).
// Between the advection of step k and the curl pass of step k + 1 nothing happens to the velocity (splats arrive between CALLS, and a
// call that asks for n steps cannot be observed in between), so the advected velocity never has to reach memory: the tile advects it into
// registers — for its texels and a 3-ring apron, by the arithmetic of k_advect_both_fast — advects the dye of its own texels with it, and
// runs the three stencil stages of k_curl_vort_div on the registers.  Per texel: velocity 8 B in (+ taps, which hit the caches), dye
// 16 + 16, velocity after vorticity 8, divergence 4 (+ curl 4 only in the launch whose curl field a caller can read: the last of the
// chain) = 52 B against 49 + 25 for the two launches it replaces.
// Layout: ONE texel per lane and row (what the gathers want: a wave's tap addresses span 64 texels, not 256), a wave = 64 columns x RY
// rows, NW waves stacked in y with the three LDS mailbox exchanges of the stencil stages; x neighbours by DPP lane shifts, the x apron
// (AX columns a side: the three stencil stages need 3; 4 would keep the stored runs 32-byte aligned and measures the same) by redundancy
// inside the wave.
// Same fp32 operations in the same order on the same values as the two kernels: the same bits (tests: step(dt, n) against n x step(dt, 1)
// and against the per-pass schedule).
#ifndef FLUID_CHAIN_CH
#define FLUID_CHAIN_CH 4    // rows of the dye advection in flight together
#endif
#ifndef FLUID_CHAIN_CHA
#define FLUID_CHAIN_CHA 8   // rows of the velocity advection in flight together
#endif
template <int NW, int RY, int AX_>
struct AdvectCvd {
    static constexpr int TX = 64, TY = NW * RY, AX = AX_, AY = 3;
    static constexpr int VX = TX - 2 * AX, VY = TY - 2 * AY;
};

// The kernel is bound by its arithmetic (70 M VALU wave-instructions per launch at 4096^2 in the first version, VALU-busy 0.53 of 214 us),
// so everything that acts on an (x, y) pair is ONE packed instruction (v_pk_mul_f32 / v_pk_add_f32: two IEEE results, the same bits as two
// scalar operations): the back-trace, the texel coordinates and weights of a fetch, the velocity filter.
__device__ __forceinline__ v2f mix2(v2f a, v2f b, float t) { return a + (b - a) * v2f{ t, t }; }

// taps32 on a packed coordinate: the same operations (x = u W - 1/2, floor, fraction, range test, 24-bit row multiply)
template <unsigned SZ>
__device__ __forceinline__ Tap4 taps32p(const Win& w, const TapBox& B, v2f uv)
{
    const v2f xy = uv * v2f{ (float)w.W, (float)w.H } - v2f{ 0.5f, 0.5f };
    const v2f fl = v2f{ floorf(xy.x), floorf(xy.y) };
    const v2f fr = xy - fl;
    Tap4 t;
    t.fx = fr.x;
    t.fy = fr.y;
    const int i0 = (int)fl.x, j0 = (int)fl.y;
    if ((unsigned)(i0 - B.xlo) < B.nx && (unsigned)(j0 - B.ylo) < B.ny) {
        const unsigned row_bytes = (unsigned)w.P * SZ;
        const unsigned k = 0u - (unsigned)(w.g0 * w.P + w.c0) * SZ;
        t.a = __umul24((unsigned)j0, row_bytes) + k + (unsigned)i0 * SZ;
        t.b = t.a + SZ;
        t.c = t.a + row_bytes;
        t.d = t.c + SZ;
        t.miss = 0;
    } else {  // bil_taps, fluid_math.h
        const int ia = clampi(i0, 0, w.W - 1), ib = clampi(i0 + 1, 0, w.W - 1);
        const int ja = clampi(j0, 0, w.H - 1), jb = clampi(j0 + 1, 0, w.H - 1);
        t.miss = 0;
        const int la = clampi(ja - w.g0, 0, w.rows - 1), lb = clampi(jb - w.g0, 0, w.rows - 1);
        const int ka = clampi(ia - w.c0, 0, w.P - 1), kb = clampi(ib - w.c0, 0, w.P - 1);
        t.a = (unsigned)(la * w.P + ka) * SZ;
        t.b = (unsigned)(la * w.P + kb) * SZ;
        t.c = (unsigned)(lb * w.P + ka) * SZ;
        t.d = (unsigned)(lb * w.P + kb) * SZ;
    }
    return t;
}

// BORDER: the tile touches the domain's edge (CLAMP_TO_EDGE selects, reflecting walls, lanes / rows past the last column / row);
// an interior tile (88 % of a 4096^2 grid) runs without any of it
// MODE: 0 = a step in the middle of a fluid_step_n chain (the curl field is nobody's to read: not stored); 1 = the chain's last such launch
// (curl stored); 2 = the launch that ends a CALL (fluid_step, or the last step of fluid_step_n): the advected velocity is what the caller can
// read, so it is stored too (vel_adv_out), and the next step's curl / vorticity / divergence go to the context's `pending` buffers — the
// next call takes them over if nothing touched the fields in between (fluid_solver.cpp: pend_*), else they are simply dropped.
// DYE = false: the dye grid differs from the sim grid (the reference's default shape) — the launch advects the velocity only and always
// stores it (the separate dye pass that follows samples the ADVECTED velocity bilinearly, script.js:1287-1293); MODE is then 2.
template <int NW, int RY, int AX_, int MODE, bool BORDER, bool DYE = true>
__device__ __forceinline__ void advect_cvd_body(const Win& w, const float2* __restrict__ vel, float2* __restrict__ vel_out,
                                                const float4* __restrict__ dye, float4* __restrict__ dye_out, float* __restrict__ curl_out,
                                                float* __restrict__ div_out, float2* __restrict__ vel_adv_out, float dt, double rW, double rH, double rvd, double rdd, float tsx,
                                                float tsy, float curl_strength, int ga, int gb, int x0, int y0, float (*mail)[2][3][64])
{
    using G = AdvectCvd<NW, RY, AX_>;
    constexpr int CH = FLUID_CHAIN_CH;  // rows whose gathers are in flight together (k_advect_both_fast: four texels per thread)
    static_assert(RY % CH == 0, "rows per wave in whole chunks");
    const int lane = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int cx = x0 + lane, gy = y0 + wv * RY;
    const int i = BORDER ? min(cx, w.W - 1) : cx;  // a lane past the last column repeats it (never stored; column W - 1 overrides what it would lend)
    const bool at_left = BORDER && cx == 0, at_right = BORDER && cx == w.W - 1;
    const int wb = wv > 0 ? wv - 1 : 0, wa = wv < NW - 1 ? wv + 1 : NW - 1;
    const TapBox B = tap_box(w);
    const float u = div_uniform((float)i + 0.5f, rW);
    const v2f dt2 = v2f{ dt, dt }, ts2 = v2f{ tsx, tsy };
    const unsigned col = (unsigned)(i - w.c0);
    auto row_of = [&](int r) { return BORDER ? min(gy + r, w.H - 1) : gy + r; };  // a row past the domain repeats the last one (wave-uniform)

    // ---- K7a for the wave's RY rows: V = advected velocity.  CHA rows' loads in flight together: the launch is bound by how many dependent
    // round trips through memory a tile makes (own velocity -> its taps -> the dye taps), not by bytes or arithmetic ----
    constexpr int CHA = FLUID_CHAIN_CHA < RY ? FLUID_CHAIN_CHA : RY;
    static_assert(RY % CHA == 0, "rows per wave in whole chunks");
    v2f V[RY];
    auto vc_of = [&](int r) { return div_uniform((float)row_of(r) + 0.5f, rH); };  // the row's v coordinate (recomputed where needed: three instructions against a register per row)
#pragma unroll
    for (int r0 = 0; r0 < RY; r0 += CHA) {
        v2f vv[CHA];
#pragma unroll
        for (int k = 0; k < CHA; k++) {
            const float2 q = ld(at_byte(vel, ((unsigned)((row_of(r0 + k) - w.g0) * w.P) + col) * 8u), 0);
            vv[k] = v2f{ q.x, q.y };
        }
        Tap4 t[CHA];
        Fetch2 f2[CHA];
#pragma unroll
        for (int k = 0; k < CHA; k++) t[k] = taps32p<8>(w, B, v2f{ u, vc_of(r0 + k) } - dt2 * vv[k] * ts2);
        gather_taps<CHA>(vel, t, f2);
#pragma unroll
        for (int k = 0; k < CHA; k++) {
            const Fetch2& f = f2[k];
            const v2f m = mix2(mix2(v2f{ f.a.x, f.a.y }, v2f{ f.b.x, f.b.y }, f.fx), mix2(v2f{ f.c.x, f.c.y }, v2f{ f.d.x, f.d.y }, f.fx), f.fy);
            V[r0 + k] = v2f{ div_uniform(m.x, rvd), div_uniform(m.y, rvd) };
        }
    }

    int xa, xb, out_lo, out_hi;
    tile_exact(x0, G::TX, G::AX, w.W, w.x0, w.x1, xa, xb);
    tile_exact(y0, G::TY, G::AY, w.H, ga, gb, out_lo, out_hi);
    const bool col_store = (cx >= xa) && (cx < xb);

    if constexpr (MODE == 2) {  // the advected velocity itself, for whoever reads the field between two calls
        if (col_store) {
#pragma unroll
            for (int r = 0; r < RY; r++) {
                const int gj = gy + r;
                if (gj >= out_lo && gj < out_hi) *at_byte(vel_adv_out, ((unsigned)((gj - w.g0) * w.P) + col) * 8u) = make_float2(V[r].x, V[r].y);
            }
        }
    }

    // ---- K7b for the texels this tile stores, CH rows at a time (the apron columns sit it out; an apron row costs its gathers: 6 of
    // NW * RY) ----
    if (DYE && col_store) {
#pragma unroll
        for (int r0 = 0; r0 < RY; r0 += CH) {
            if (gy + r0 + CH <= out_lo || gy + r0 >= out_hi) continue;  // wave-uniform
            Tap4 t[CH];
            Fetch4 f4[CH];
#pragma unroll
            for (int k = 0; k < CH; k++) t[k] = taps32p<16>(w, B, v2f{ u, vc_of(r0 + k) } - dt2 * V[r0 + k] * ts2);
            gather_taps<CH>(dye, t, f4);
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const Fetch4& f = f4[k];
                const v2f lo = mix2(mix2(v2f{ f.a.x, f.a.y }, v2f{ f.b.x, f.b.y }, f.fx), mix2(v2f{ f.c.x, f.c.y }, v2f{ f.d.x, f.d.y }, f.fx), f.fy);
                const v2f hi = mix2(mix2(v2f{ f.a.z, f.a.w }, v2f{ f.b.z, f.b.w }, f.fx), mix2(v2f{ f.c.z, f.c.w }, v2f{ f.d.z, f.d.w }, f.fx), f.fy);
                const int gj = gy + r0 + k;
                if (gj >= out_lo && gj < out_hi)
                    *at_byte(dye_out, ((unsigned)((gj - w.g0) * w.P) + col) * 16u) =
                        make_float4(div_uniform(lo.x, rdd), div_uniform(lo.y, rdd), div_uniform(hi.x, rdd), div_uniform(hi.y, rdd));
            }
        }
    }

    // ---- K1 of the next step: curl of the advected velocity (vx of the rows below / above, vy of the columns left / right) ----
    mail[wv][0][0][lane] = V[0].x;
    mail[wv][1][0][lane] = V[RY - 1].x;
    __syncthreads();
    const float vxb = mail[wb][1][0][lane], vxa = mail[wa][0][0][lane];
    float C[RY];
#pragma unroll
    for (int r = 0; r < RY; r++) {
        const int gj = gy + r;
        float L = from_left_lane(V[r].y), R = from_right_lane(V[r].y);
        float Bq = r > 0 ? V[r > 0 ? r - 1 : 0].x : vxb, T = r < RY - 1 ? V[r < RY - 1 ? r + 1 : r].x : vxa;
        if (BORDER) {  // CLAMP_TO_EDGE: an off-domain neighbour is the texel itself
            if (at_left) L = V[r].y;
            if (at_right) R = V[r].y;
            if (gj == 0) Bq = V[r].x;
            if (gj == w.H - 1) T = V[r].x;
        }
        const float vort = R - L - T + Bq;
        C[r] = 0.5f * vort;
    }

    // ---- K2: vorticity confinement (curl of the four neighbours) ----
    mail[wv][0][1][lane] = C[0];
    mail[wv][1][1][lane] = C[RY - 1];
    __syncthreads();
    const float cb = mail[wb][1][1][lane], ca = mail[wa][0][1][lane];
    float2 N[RY];
#pragma unroll
    for (int r = 0; r < RY; r++) {
        const int gj = gy + r;
        float L = from_left_lane(C[r]), R = from_right_lane(C[r]);
        float Bq = r > 0 ? C[r > 0 ? r - 1 : 0] : cb, T = r < RY - 1 ? C[r < RY - 1 ? r + 1 : r] : ca;
        if (BORDER) {
            if (at_left) L = C[r];
            if (at_right) R = C[r];
            if (gj == 0) Bq = C[r];
            if (gj == w.H - 1) T = C[r];
        }
        N[r] = vorticity_cell(L, R, T, Bq, C[r], make_float2(V[r].x, V[r].y), curl_strength, dt);
    }

    // ---- K3: divergence of the new velocity, reflecting walls (script.js:804-807) ----
    mail[wv][0][2][lane] = N[0].y;
    mail[wv][1][2][lane] = N[RY - 1].y;
    __syncthreads();
    const float nyb = mail[wb][1][2][lane], nya = mail[wa][0][2][lane];
#pragma unroll
    for (int r = 0; r < RY; r++) {
        const int gj = gy + r;
        float L = from_left_lane(N[r].x), R = from_right_lane(N[r].x);
        float Bq = r > 0 ? N[r > 0 ? r - 1 : 0].y : nyb, T = r < RY - 1 ? N[r < RY - 1 ? r + 1 : r].y : nya;
        if (BORDER) {
            if (at_left) L = -N[r].x;
            if (at_right) R = -N[r].x;
            if (gj == w.H - 1) T = -N[r].y;
            if (gj == 0) Bq = -N[r].y;
        }
        const float dv = 0.5f * (R - L + T - Bq);
        if (col_store && gj >= out_lo && gj < out_hi) {
            const unsigned c = (unsigned)((gj - w.g0) * w.P) + col;
            if (MODE >= 1) *at_byte(curl_out, c * 4u) = C[r];
            *at_byte(div_out, c * 4u) = dv;
            *at_byte(vel_out, c * 8u) = N[r];
        }
    }
}

template <int NW, int RY, int AX_, int MODE, bool DYE = true>
__global__ void __launch_bounds__(64 * NW, RY <= 4 ? 5 : 4) k_advect_cvd(Win w, const float2* __restrict__ vel, float2* __restrict__ vel_out,
                                                                         const float4* __restrict__ dye, float4* __restrict__ dye_out,
                                                                         float* __restrict__ curl_out, float* __restrict__ div_out,
                                                                         float2* __restrict__ vel_adv_out, float dt,
                                                                         double rW, double rH, double rvd, double rdd, float tsx, float tsy,
                                                                         float curl_strength, int ga, int gb, int xs, int ys, int nx, int ny,
                                                                         int remap)
{
    using G = AdvectCvd<NW, RY, AX_>;
    __shared__ float mail[NW][2][3][64];  // [wave][first / last row][stage][lane]
    int bx, by;
    tile_of_block((int)blockIdx.x, nx, ny, remap, bx, by);
    const int x0 = xs + bx * G::VX, y0 = ys + by * G::VY;
    if (x0 <= 0 || x0 + G::TX >= w.W || y0 <= 0 || y0 + G::TY >= w.H)
        advect_cvd_body<NW, RY, AX_, MODE, true, DYE>(w, vel, vel_out, dye, dye_out, curl_out, div_out, vel_adv_out, dt, rW, rH, rvd, rdd, tsx, tsy, curl_strength, ga, gb, x0, y0, mail);
    else
        advect_cvd_body<NW, RY, AX_, MODE, false, DYE>(w, vel, vel_out, dye, dye_out, curl_out, div_out, vel_adv_out, dt, rW, rH, rvd, rdd, tsx, tsy, curl_strength, ga, gb, x0, y0, mail);
}

#ifndef VD_NW_
#define VD_NW_ 8
#endif
#ifndef VD_RY_
#define VD_RY_ 5
#endif
constexpr int VD_NW = VD_NW_, VD_RY = VD_RY_;

// tile-shape variants (NW waves x RY rows per wave, apron HX columns x HY rows, BPC workgroups per CU); FLUID_TB_VARIANT picks one (tuning knob,
// read once).  Without the knob the shape follows the grid (tb_variant_for): the 8 x 10 tile measured best on MI355X at 4096^2 and above
// (profiles/), but a 1024^2 launch is only 90 such tiles on 512 workgroup slots and each of them runs its ten iterations latency-bound
// (15.8 us per launch, profiles/r02/pass_time_vs_grid_size.txt) — small grids take tiles with fewer rows per wave: more workgroups, a
// shorter per-iteration chain.  `gs`: the shape has a gradient-subtract instantiation (k_jacobi_tb_gs: needs HX >= HY + 1).
struct TBVariant { int nw, ry, hx, hy, bpc; bool gs; };
constexpr TBVariant kTB[] = {
    {8, 10, 12, 10, 2, true},   // 0: default at >= 3072^2 — 126 VGPRs, two workgroups per CU, 50 iterations in 5 launches
    {8, 8, 8, 8, 2, false},     // 1: shallow apron, 7 launches (round 1's first shape)
    {8, 11, 12, 10, 2, false},  // 2: (spills)
    {8, 12, 12, 10, 2, true},   // 3: 972 tiles at 4096^2 instead of 1242 (spills: 0.78 ms per step against 0.50)
    {8, 12, 16, 13, 2, false},  // 4: 4 launches, 128 VGPRs (spills)
    {8, 16, 20, 17, 1, false},  // 5: 3 launches, one workgroup per CU
    {16, 12, 20, 17, 1, false}, // 6: 3 launches, 16-wave workgroup
    {4, 24, 16, 13, 2, false},  // 7: 4 waves x 24 rows
    {8, 5, 12, 10, 2, true},    // 8: small grids: 40-row tile
    {8, 6, 12, 10, 2, true},    // 9: 48-row tile
    {8, 7, 12, 10, 2, true},    // 10: 56-row tile
    {8, 4, 12, 10, 3, true},    // 11: 32-row tile, three workgroups per CU
    {8, 7, 20, 17, 2, true},    // 12: small grids, deeper: 56-row tile with a 17-row apron — 50 iterations in 3 launches
    {8, 8, 28, 25, 2, true},    // 13: 64-row tile with a 25-row apron — 2 launches
    {8, 6, 20, 17, 2, true},    // 14: 48-row tile, 17-row apron
    {16, 6, 28, 25, 1, true},   // 15: 16 waves x 6 rows, 25-row apron — 2 launches
    {4, 10, 12, 10, 4, false},  // 16: the 40-row tile as FOUR waves x 10 rows (one wave per SIMD: a 4-wave barrier)
    {4, 16, 12, 10, 2, false},  // 17: 4 waves x 16 rows
    {2, 20, 12, 10, 4, false},  // 18: the 40-row tile as TWO waves x 20 rows
    {16, 3, 12, 10, 1, false},  // 19: 16 waves x 3 rows (48-row tile)      (16-19: profiles/r03/jacobi_iter_cost.txt)
    {8, 5, 12, 10, 3, true},    // 20: the 40-row tile with TWO texels per lane (128 columns, k_jacobi_tb2): grids below 1280^2
};
constexpr int kPairTB = 20;     // not in TB_VARIANTS: its own kernels, fp32 fields only (fp16 storage runs shape 8 in its place)
constexpr int kNumTB = sizeof(kTB) / sizeof(kTB[0]);
constexpr int kDefaultTB = 0;  // measured best of the table at 4096^2 (profiles/r01/jacobi_variants.txt, there listed as "8x10 h12/10")

// The PRODUCT library (make) holds the shapes the grid-driven choice can return (jacobi_tb_pick, tb_rules: 0 at >= 3072^2 texels, 8 and the
// two-texel tile below); everything else in the table is a lab shape of some round's A/B and exists in libfluid_hip_probes.so only
// (make PROBES=1: -DFLUID_PROBES, where the FLUID_* knobs that select them are read at all — fluid_kernels.h lab_env).
#ifdef FLUID_PROBES
#define TB_VARIANTS(X)      \
    X(0, 8, 10, 12, 10, 2)  \
    X(1, 8, 8, 8, 8, 2)     \
    X(2, 8, 11, 12, 10, 2)  \
    X(3, 8, 12, 12, 10, 2)  \
    X(4, 8, 12, 16, 13, 2)  \
    X(5, 8, 16, 20, 17, 1)  \
    X(6, 16, 12, 20, 17, 1) \
    X(7, 4, 24, 16, 13, 2)  \
    X(8, 8, 5, 12, 10, 2)   \
    X(9, 8, 6, 12, 10, 2)   \
    X(10, 8, 7, 12, 10, 2)  \
    X(11, 8, 4, 12, 10, 3)  \
    X(12, 8, 7, 20, 17, 2)  \
    X(13, 8, 8, 28, 25, 2)  \
    X(14, 8, 6, 20, 17, 2)  \
    X(15, 16, 6, 28, 25, 1) \
    X(16, 4, 10, 12, 10, 4) \
    X(17, 4, 16, 12, 10, 2) \
    X(18, 2, 20, 12, 10, 4) \
    X(19, 16, 3, 12, 10, 1)
#define TB_GS_VARIANTS(X)  \
    X(0, 8, 10, 12, 10, 2) \
    X(3, 8, 12, 12, 10, 2) \
    X(8, 8, 5, 12, 10, 2)  \
    X(9, 8, 6, 12, 10, 2)  \
    X(10, 8, 7, 12, 10, 2) \
    X(11, 8, 4, 12, 10, 3) \
    X(12, 8, 7, 20, 17, 2) \
    X(13, 8, 8, 28, 25, 2) \
    X(14, 8, 6, 20, 17, 2) \
    X(15, 16, 6, 28, 25, 1)
#else
#define TB_VARIANTS(X)     \
    X(0, 8, 10, 12, 10, 2) \
    X(8, 8, 5, 12, 10, 2)
#define TB_GS_VARIANTS(X) X(8, 8, 5, 12, 10, 2)
#endif
#define TB_CHECK(k, NW, RY, HX, HY, BPC) \
    static_assert(kTB[k].nw == NW && kTB[k].ry == RY && kTB[k].hx == HX && kTB[k].hy == HY && kTB[k].bpc == BPC, "variant table");
TB_VARIANTS(TB_CHECK)
#define TB_GS_CHECK(k, NW, RY, HX, HY, BPC) TB_CHECK(k, NW, RY, HX, HY, BPC) static_assert(kTB[k].gs && HX >= HY + 1, "gs variant");
TB_GS_VARIANTS(TB_GS_CHECK)
// which shapes of the table THIS library holds kernels for
#define TB_IS(k, NW, RY, HX, HY, BPC) case k:
constexpr bool tb_built(int k)
{
    switch (k) {
        TB_VARIANTS(TB_IS)
    case kPairTB: return true;
    default: return false;
    }
}
constexpr bool tb_gs_built(int k)
{
    switch (k) {
        TB_GS_VARIANTS(TB_IS)
    case kPairTB: return true;
    default: return false;
    }
}
#undef TB_IS

int cvd_remap()  // FLUID_CVD_REMAP: tile order of the fused curl/vorticity/divergence kernel (same encoding)
{
    static const int v = [] {
        const char* e = lab_env("FLUID_CVD_REMAP");
        return (e ? atoi(e) : 3) & 3;  // row-major XCD runs, as for the Jacobi kernel (82 us vs 85 us column-major at 4096^2)
    }();
    return v;
}

int tb_variant_env()  // FLUID_TB_VARIANT, or -1
{
    static const int v = [] {
        const char* e = lab_env("FLUID_TB_VARIANT");
        const int k = e ? atoi(e) : -1;
        return (k >= 0 && k < kNumTB && tb_built(k)) ? k : -1;
    }();
    return v;
}

// FLUID_TB_SMALL="texels:shape,texels:shape,...": grids below `texels` owned texels take `shape` (first match; A/B knob for the
// grid-driven choice).  Default: below 2000^2 the 40-row tile with two texels per lane (shape 20), below 3072^2 the 40-row tile
// (profiles/r03/jacobi_shapes_small_grids.txt, jacobi_pair_tile_ab.txt).
struct TBRule { long below; int shape; };
const std::vector<TBRule>& tb_rules()
{
    static const std::vector<TBRule> rules = [] {
        std::vector<TBRule> r;
        if (const char* e = lab_env("FLUID_TB_SMALL")) {
            const char* p = e;
            while (*p) {
                char* q = nullptr;
                const long t = strtol(p, &q, 10);
                if (q == p || *q != ':') break;
                p = q + 1;
                const long sh = strtol(p, &q, 10);
                if (q == p) break;
                if (sh >= 0 && sh < kNumTB && tb_built((int)sh)) r.push_back({ t, (int)sh });
                p = *q == ',' ? q + 1 : q;
            }
        } else {
            r.push_back({ 2000l * 2000l, kPairTB });  // two texels per lane: 512^2 +35 %, 1024^2 +14 %, 1536^2 +5 %, level at 2048^2 (profiles/r03/jacobi_pair_tile_ab.txt)
            r.push_back({ kSmallGridTexels, 8 });
        }
        return r;
    }();
    return rules;
}

// How many of a launch's first / last rows take the small tiles (k_jacobi_tb_mix).  What fills the 512 workgroup slots does not depend on the
// grid's width, rows do: the cut is expressed in small TILES — default 192 at the head, 384 at the tail, of 8 x 7 tiles for bands of at
// least 4000 rows and 8 x 5 tiles below — and converted to whole tile rows at the launch's width.  At 4096^2 the launch goes from 46.3-47.2
// to 43.0-44.5 us and the step down 2.3 ... 3.2 % (five boxes, interleaved; everything from 180 to 540 tiles of 5-, 6- or 7-row tiles is
// within 0.5 % there); at 3072^2 the 5-row tiles are 3.5 % ahead of the 7-row ones, at 8192^2 the 7-row tiles 1 %; at 16384^2 a cut in ROWS
// (366 / 666: thousands of small tiles) is slower than no cut and the same tile counts gain 1.3 % (profiles/r03/jacobi_small_tile_head_tail.txt).
// A/B knobs: FLUID_TB_TAIL_TILES="head_tiles,tail_tiles,ry", FLUID_TB_TAIL="head_rows,tail_rows,ry" (rows win; 0,0 = one shape per launch).
struct TBTail { int head, tail, ry, min_rows; };   // head / tail in ROWS
inline TBTail tb_tail(int rows, int nx)
{
    static const TBTail forced_rows = [] {
        TBTail r{ -1, -1, 7, 0 };
        if (const char* e = lab_env("FLUID_TB_TAIL")) {
            r = TBTail{ 0, 0, 7, 0 };   // a forced setting applies to every band height (the knob-hash test runs it on small grids)
            sscanf(e, "%d,%d,%d", &r.head, &r.tail, &r.ry);
            if (r.ry != 2 && r.ry != 5 && r.ry != 6 && r.ry != 7) r.head = r.tail = 0;   // 2 = the two-texel tile (8 x 5 rows, 128 columns)
        }
        return r;
    }();
    if (forced_rows.head >= 0) return forced_rows;
    static const TBTail forced_tiles = [] {
        TBTail r{ -1, -1, 5, 1024 };
        if (const char* e = lab_env("FLUID_TB_TAIL_TILES")) {
            r = TBTail{ 0, 0, 5, 1024 };
            sscanf(e, "%d,%d,%d", &r.head, &r.tail, &r.ry);
            if (r.ry != 2 && r.ry != 5 && r.ry != 6 && r.ry != 7) r.head = r.tail = 0;   // 2 = the two-texel tile (8 x 5 rows, 128 columns)
        }
        return r;
    }();
    const TBTail tiles = forced_tiles.head >= 0 ? forced_tiles : TBTail{ 192, 384, rows >= 4000 ? 7 : 5, 1024 };
    if (tiles.ry == 2) nx = (nx * 232 + 103) / 104;   // tiles per row of the two-texel shape: 104 stored columns each instead of 232
    const int vy = 8 * (tiles.ry == 2 ? 5 : tiles.ry) - 20;  // rows a small tile stores (apron 10 on both sides)
    return TBTail{ (tiles.head + nx - 1) / nx * vy, (tiles.tail + nx - 1) / nx * vy, tiles.ry, tiles.min_rows };
}

template <int NW, int RYA, int RYB, int HX, int HY, int BPC>
hipError_t launch_tb_mix(hipStream_t s, Win w, const float* p, const float* div, float* p_out, float pscale, int iters, int ga, int gb, int head, int tail)
{
    using GA = JacobiTB<NW, RYA, HX, HY>;
    using GB = JacobiTB<NW, RYB, HX, HY>;
    const Axis ax = make_axis(w.x0, w.x1, w.W, GA::TX, HX);
    MixSegs S{};
    int total = 0;
    auto seg = [&](int a, int b, int small) {
        if (b <= a) return;
        const Axis ay = small ? make_axis(a, b, w.H, GB::TY, HY) : make_axis(a, b, w.H, GA::TY, HY);
        const int k = S.n++;
        S.g[k] = a; S.g[k + 1] = b; S.small[k] = small; S.ys[k] = ay.S; S.ny[k] = ay.n; S.blk0[k] = total;
        total += ax.n * ay.n;
    };
    // big tile b of the middle segment stores rows up to S + (TY - HY) + b VY (S = its first tile's origin): the cut in front of the tail
    // snaps down to such a boundary, so that no big tile is launched for a few rows
    const int lo = ga + head;
    int gmid = gb - tail;
    if (tail > 0) {
        const int base = (lo - HY > 0 ? lo - HY : 0) + GA::TY - HY;
        if (gmid > base) gmid = base + (gmid - base) / GA::VY * GA::VY;
    }
    if (gmid < lo) gmid = lo;
    seg(ga, lo, 1);
    seg(lo, gmid, 0);
    seg(gmid, gb, 1);
    S.blk0[S.n] = total;
    k_jacobi_tb_mix<NW, RYA, RYB, HX, HY, BPC><<<dim3(total, 1, 1), dim3(64, NW, 1), 0, s>>>(w, p, div, p_out, pscale, iters, S, ax.S, ax.n, xcd_remap());
    return hipGetLastError();
}

template <int NW, int RYA, int RYP, int HX, int HY, int BPC>
hipError_t launch_tb_mix2(hipStream_t s, Win w, const float* p, const float* div, float* p_out, float pscale, int iters, int ga, int gb, int head,
                          int tail);   // defined behind the two-texel tile's launchers

template <int NW, int RY, int HX, int HY, int BPC>
hipError_t launch_tb(hipStream_t s, Win w, const float* p, const float* div, float* p_out, float pscale, int iters, int ga, int gb)
{
    if constexpr (NW == 8 && RY == 10 && HX == 12 && HY == 10) {  // the default shape: with small tiles for the launch's first / last rows
        const int rows = gb - ga;
        TBTail t = tb_tail(rows, make_axis(w.x0, w.x1, w.W, 256, HX).n);
        const int full = 3 * (t.head + t.tail);
        if (full > 0 && rows < full) {  // a shorter band: the same proportions (at most a third of the rows in small tiles)
            t.head = (int)((long)t.head * rows / full);
            t.tail = (int)((long)t.tail * rows / full);
        }
        if (t.head + t.tail > 0 && rows >= t.min_rows) {
#ifdef FLUID_PROBES
            if (t.ry == 2) return launch_tb_mix2<NW, RY, 5, HX, HY, BPC>(s, w, p, div, p_out, pscale, iters, ga, gb, t.head, t.tail);
            if (t.ry == 6) return launch_tb_mix<NW, RY, 6, HX, HY, BPC>(s, w, p, div, p_out, pscale, iters, ga, gb, t.head, t.tail);
#endif
            if (t.ry == 5) return launch_tb_mix<NW, RY, 5, HX, HY, BPC>(s, w, p, div, p_out, pscale, iters, ga, gb, t.head, t.tail);
            return launch_tb_mix<NW, RY, 7, HX, HY, BPC>(s, w, p, div, p_out, pscale, iters, ga, gb, t.head, t.tail);
        }
    }
    using G = JacobiTB<NW, RY, HX, HY>;
    const Axis ax = make_axis(w.x0, w.x1, w.W, G::TX, HX), ay = make_axis(ga, gb, w.H, G::TY, HY);
    k_jacobi_tb<NW, RY, HX, HY, BPC><<<dim3(ax.n * ay.n, 1, 1), dim3(64, NW, 1), 0, s>>>(w, p, div, p_out, pscale, iters, ga, gb, ax.S, ay.S, ax.n,
                                                                                 ay.n, xcd_remap());
    return hipGetLastError();
}

template <int NW, int RY, int HX, int HY, int BPC>
hipError_t launch_tb(hipStream_t s, Win w, const __half* p, const __half* div, __half* p_out, float pscale, int iters, int ga, int gb)
{
    using G = JacobiTB<NW, RY, HX, HY>;
    const Axis ax = make_axis(w.x0, w.x1, w.W, G::TX, HX), ay = make_axis(ga, gb, w.H, G::TY, HY);
    k_jacobi_tb_h<NW, RY, HX, HY, BPC><<<dim3(ax.n * ay.n, 1, 1), dim3(64, NW, 1), 0, s>>>(w, p, div, p_out, pscale, iters, ga, gb, ax.S, ay.S,
                                                                                   ax.n, ay.n, xcd_remap());
    return hipGetLastError();
}

// the last launch with the gradient subtract folded in: the same shape with a row apron of HY + 1
template <int NW, int RY, int HX, int HY, int BPC>
hipError_t launch_tb_gs(hipStream_t s, Win w, const float* p, const float* div, float* p_out, const float2* vel, float2* vel_out, float pscale,
                        int iters, int ga, int gb)
{
    using G = JacobiTB<NW, RY, HX, HY + 1>;
    const Axis ax = make_axis(w.x0, w.x1, w.W, G::TX, HX), ay = make_axis(ga, gb, w.H, G::TY, HY + 1);
    k_jacobi_tb_gs<NW, RY, HX, HY, BPC><<<dim3(ax.n * ay.n, 1, 1), dim3(64, NW, 1), 0, s>>>(w, p, div, p_out, vel, vel_out, pscale, iters, ga, gb,
                                                                                    ax.S, ay.S, ax.n, ay.n, xcd_remap());
    return hipGetLastError();
}

template <int NW, int RY, int HX, int HY, int BPC>
hipError_t launch_tb_gs(hipStream_t s, Win w, const __half* p, const __half* div, __half* p_out, const __half2* vel, __half2* vel_out,
                        float pscale, int iters, int ga, int gb)
{
    using G = JacobiTB<NW, RY, HX, HY + 1>;
    const Axis ax = make_axis(w.x0, w.x1, w.W, G::TX, HX), ay = make_axis(ga, gb, w.H, G::TY, HY + 1);
    k_jacobi_tb_gs_h<NW, RY, HX, HY, BPC><<<dim3(ax.n * ay.n, 1, 1), dim3(64, NW, 1), 0, s>>>(w, p, div, p_out, vel, vel_out, pscale, iters, ga,
                                                                                      gb, ax.S, ay.S, ax.n, ay.n, xcd_remap());
    return hipGetLastError();
}

template <int NW, int RY, int HX, int HY, int BPC>
hipError_t launch_tb2(hipStream_t s, Win w, const float* p, const float* div, float* p_out, float pscale, int iters, int ga, int gb)
{
    using G = JacobiTB2<NW, RY, HX, HY>;
    const Axis ax = make_axis(w.x0, w.x1, w.W, G::TX, HX), ay = make_axis(ga, gb, w.H, G::TY, HY);
    k_jacobi_tb2<NW, RY, HX, HY, BPC><<<dim3(ax.n * ay.n, 1, 1), dim3(64, NW, 1), 0, s>>>(w, p, div, p_out, pscale, iters, ga, gb, ax.S, ay.S, ax.n,
                                                                                  ay.n, xcd_remap());
    return hipGetLastError();
}
template <int NW, int RY, int HX, int HY, int BPC>
hipError_t launch_tb2_gs(hipStream_t s, Win w, const float* p, const float* div, float* p_out, const float2* vel, float2* vel_out, float pscale,
                         int iters, int ga, int gb)
{
    using G = JacobiTB2<NW, RY, HX, HY + 1>;
    const Axis ax = make_axis(w.x0, w.x1, w.W, G::TX, HX), ay = make_axis(ga, gb, w.H, G::TY, HY + 1);
    k_jacobi_tb2_gs<NW, RY, HX, HY, BPC><<<dim3(ax.n * ay.n, 1, 1), dim3(64, NW, 1), 0, s>>>(w, p, div, p_out, vel, vel_out, pscale, iters, ga, gb,
                                                                                     ax.S, ay.S, ax.n, ay.n, xcd_remap());
    return hipGetLastError();
}

// The mixed launch with the TWO-texel tile (k_jacobi_tb2's 128-column tile) as the shape of its first / last rows: the shortest latency
// chain of all the tiles (6 instructions a row) for the part of a launch whose iterations nothing overlaps.  Same bits by construction
// (jacobi_row2).  FLUID_TB_TAIL_TILES / FLUID_TB_TAIL with 2 as the third field (A/B: profiles/r04/jacobi_pair_head_tail_ab.txt).
template <int NW, int RYA, int RYP, int HX, int HY, int BPC>
__global__ void __launch_bounds__(64 * NW, (BPC * NW + 3) / 4) k_jacobi_tb_mix2(Win w, const float* __restrict__ p, const float* __restrict__ div,
                                                             float* __restrict__ p_out, float pscale, int iters, MixSegs S, int xs, int nx,
                                                             int xs2, int nx2, int remap)
{
    __shared__ float4 mail[2][NW][2][64];
    const int b = (int)blockIdx.x;
    int k = 0;
    while (k + 1 < S.n && b >= S.blk0[k + 1]) k++;
    if (S.small[k])  // a head / tail segment: 128-column tiles of two texels per lane (their mailboxes fit in half of the same LDS)
        jacobi_tb2_tile<NW, RYP, HX, HY, false>(w, p, div, p_out, pscale, iters, S.g[k], S.g[k + 1], xs2, S.ys[k], nx2, S.ny[k], b - S.blk0[k], remap,
                                                reinterpret_cast<v2f(*)[NW][2][64]>(mail));
    else
        jacobi_tb_tile<NW, RYA, HX, HY>(w, p, div, p_out, pscale, iters, S.g[k], S.g[k + 1], xs, S.ys[k], nx, S.ny[k], b - S.blk0[k], remap, mail);
}

template <int NW, int RYA, int RYP, int HX, int HY, int BPC>
hipError_t launch_tb_mix2(hipStream_t s, Win w, const float* p, const float* div, float* p_out, float pscale, int iters, int ga, int gb, int head,
                          int tail)
{
    using GA = JacobiTB<NW, RYA, HX, HY>;
    using GP = JacobiTB2<NW, RYP, HX, HY>;
    const Axis ax = make_axis(w.x0, w.x1, w.W, GA::TX, HX), ax2 = make_axis(w.x0, w.x1, w.W, GP::TX, HX);
    MixSegs S{};
    int total = 0;
    auto seg = [&](int a, int b, int small) {
        if (b <= a) return;
        const Axis ay = small ? make_axis(a, b, w.H, GP::TY, HY) : make_axis(a, b, w.H, GA::TY, HY);
        const int k = S.n++;
        S.g[k] = a; S.g[k + 1] = b; S.small[k] = small; S.ys[k] = ay.S; S.ny[k] = ay.n; S.blk0[k] = total;
        total += (small ? ax2.n : ax.n) * ay.n;
    };
    const int lo = ga + head;
    int gmid = gb - tail;
    if (tail > 0) {  // the cut in front of the tail snaps down to a boundary of the big tiles (launch_tb_mix)
        const int base = (lo - HY > 0 ? lo - HY : 0) + GA::TY - HY;
        if (gmid > base) gmid = base + (gmid - base) / GA::VY * GA::VY;
    }
    if (gmid < lo) gmid = lo;
    seg(ga, lo, 1);
    seg(lo, gmid, 0);
    seg(gmid, gb, 1);
    S.blk0[S.n] = total;
    k_jacobi_tb_mix2<NW, RYA, RYP, HX, HY, BPC><<<dim3(total, 1, 1), dim3(64, NW, 1), 0, s>>>(w, p, div, p_out, pscale, iters, S, ax.S, ax.n, ax2.S, ax2.n,
                                                                                      xcd_remap());
    return hipGetLastError();
}

inline dim3 row_grid(const Win& w, int ga, int gb) { return dim3((w.x1 - w.x0 + BX - 1) / BX, gb - ga, 1); }  // columns [x0, x1) x rows [ga, gb)

}  // namespace

// ------------------------------------------------------------------------------------------------
#define ROWS_OR_RETURN() \
    if (gb <= ga) return hipSuccess

hipError_t launch_curl(hipStream_t s, Win w, const float2* vel, float* curl, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_curl<<<row_grid(w, ga, gb), BX, 0, s>>>(w, vel, curl, ga);
    return hipGetLastError();
}

hipError_t launch_vorticity(hipStream_t s, Win w, const float2* vel, const float* curl, float2* vel_out, float curl_strength,
                            float dt, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_vorticity<<<row_grid(w, ga, gb), BX, 0, s>>>(w, vel, curl, vel_out, curl_strength, dt, ga);
    return hipGetLastError();
}

hipError_t launch_divergence(hipStream_t s, Win w, const float2* vel, float* div, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_divergence<<<row_grid(w, ga, gb), BX, 0, s>>>(w, vel, div, ga);
    return hipGetLastError();
}

hipError_t launch_clear(hipStream_t s, Win w, const float* p, float* p_out, float value, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_clear<<<row_grid(w, ga, gb), BX, 0, s>>>(w, p, p_out, value, ga);
    return hipGetLastError();
}

hipError_t launch_jacobi(hipStream_t s, Win w, const float* p, const float* div, float* p_out, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_jacobi<<<row_grid(w, ga, gb), BX, 0, s>>>(w, p, div, p_out, ga);
    return hipGetLastError();
}

hipError_t launch_gradsub(hipStream_t s, Win w, const float* p, const float2* vel, float2* vel_out, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_gradsub<<<row_grid(w, ga, gb), BX, 0, s>>>(w, p, vel, vel_out, ga);
    return hipGetLastError();
}

hipError_t launch_gradsub4(hipStream_t s, Win w, const float* p, const float2* vel, float2* vel_out, int ga, int gb)
{
    ROWS_OR_RETURN();
    if (!fused_supported(w)) return hipErrorInvalidValue;
    w.x0 &= ~3;  // whole float4 groups: a few columns beyond the requested range are (re)computed as well
    w.x1 = (w.x1 + 3) & ~3;
    if (w.x1 > w.W) w.x1 = w.W;
    k_gradsub4<<<dim3((w.x1 - w.x0 + 255) / 256, (gb - ga + 3) / 4, 1), dim3(64, 4, 1), 0, s>>>(w, p, vel, vel_out, ga, gb);
    return hipGetLastError();
}

hipError_t launch_gradsub4(hipStream_t s, Win w, const __half* p, const __half2* vel, __half2* vel_out, int ga, int gb)
{
    ROWS_OR_RETURN();
    if (!fused_supported(w)) return hipErrorInvalidValue;
    w.x0 &= ~3;
    w.x1 = (w.x1 + 3) & ~3;
    if (w.x1 > w.W) w.x1 = w.W;
    k_gradsub4_h<<<dim3((w.x1 - w.x0 + 255) / 256, (gb - ga + 3) / 4, 1), dim3(64, 4, 1), 0, s>>>(w, p, vel, vel_out, ga, gb);
    return hipGetLastError();
}

// whether the fast advection kernels apply to a field of `texel_bytes` per texel held in window w, decaying by `decay`
// (32-bit byte offsets, 24-bit row multiply, divisor in [1, 2) for div_uniform); FLUID_ADVECT_FAST=0 keeps the general kernels (A/B knob)
static bool advect_fast_ok(const Win& w, size_t texel_bytes, float decay_a, float decay_b)
{
    static const bool enabled = [] {
        const char* e = lab_env("FLUID_ADVECT_FAST");
        return !(e && atoi(e) == 0);
    }();
    return enabled && udiv_decay_ok(decay_a) && udiv_decay_ok(decay_b) && (size_t)w.rows * (size_t)w.P * texel_bytes <= (1ull << 32) &&
           (size_t)w.P * texel_bytes < (1u << 24) && w.H < (1 << 24);
}

#ifdef FLUID_PROBES
static int advect_wy()   // FLUID_ADVECT_WY (lab): 1 (default), 2 or 4 waves of an advection block stacked in y
{
    static const int v = [] {
        const char* e = lab_env("FLUID_ADVECT_WY");
        const int k = e ? atoi(e) : 1;
        return (k == 2 || k == 4) ? k : 1;
    }();
    return v;
}
#endif

// texels per thread of the separate fast kernels: four, or fewer on small grids so that the launch still spreads over the chip
// (FLUID_ADVECT_SPLIT_ROWS=1 / 2 / 4 forces one: A/B knob)
// FLUID_VTILE=0 / 1 (lab build): the dye != sim advection gathers its velocity taps / reads them from the wave's LDS run (same bits)
static bool velocity_tile()
{
    static const int mode = [] {
        const char* e = lab_env("FLUID_VTILE");
        return e ? (atoi(e) != 0 ? 1 : 0) : -1;
    }();
    return mode >= 0 ? mode == 1 : true;
}

static int split_advect_rows(long texels)
{
    static const int forced = [] {
        const char* e = lab_env("FLUID_ADVECT_SPLIT_ROWS");
        const int k = e ? atoi(e) : 0;
        return (k == 1 || k == 2 || k == 4) ? k : 0;
    }();
    if (forced) return forced;
    return texels >= (1l << 20) ? 2 : 1;  // four rows per thread measured 10 % slower at 4096^2 (profiles/r03/shipping_rows.txt)
}

// ... of the kernels that read their velocity taps from the wave's LDS run (one run per wave whatever the rows): four rows from 6 M dye
// texels (2816^2: 51.3 us against 54.7 with two; 4096^2 packed: 106.4 against 108.2), two from 1 M (2048^2: 26.0 against 27.6 with four),
// one below (profiles/r04/dye_ne_sim_velocity_run_ab.txt)
#ifdef FLUID_PROBES
// FLUID_DYE_BOX=1 (lab build): the packed dye pass stages each wave's tap box through LDS (k_advect_dye_fast_rgb_box; measured +53 %: profiles/r05)
static bool dye_box()
{
    static const bool on = [] { const char* e = lab_env("FLUID_DYE_BOX"); return e && atoi(e) != 0; }();
    return on;
}
#endif

static int split_advect_rows_vt(long texels)
{
    static const bool forced = lab_env("FLUID_ADVECT_SPLIT_ROWS") != nullptr;   // (lab build)
    if (forced) return split_advect_rows(texels);
    return texels >= (6l << 20) ? 4 : (texels >= (1l << 20) ? 2 : 1);
}

template <class V2>
hipError_t launch_advect_velocity_any(hipStream_t s, Win w, const V2* vel, V2* out, float dt, float dissipation, int ga, int gb, unsigned int* miss)
{
    ROWS_OR_RETURN();
    const float tsx = (float)(1.0 / w.W), tsy = (float)(1.0 / w.H), decay = 1.0f + dissipation * dt;
    if (advect_fast_ok(w, sizeof(V2), decay, decay)) {
        const double rW = udiv_recip((float)w.W), rH = udiv_recip((float)w.H), rvd = udiv_recip(decay);
        const unsigned gx = (w.x1 - w.x0 + BX - 1) / BX;
        switch (split_advect_rows((long)(w.x1 - w.x0) * (gb - ga))) {
#ifdef FLUID_PROBES
        case 4: k_advect_velocity_fast<4><<<dim3(gx, (gb - ga + 3) / 4, 1), BX, 0, s>>>(w, vel, out, dt, rW, rH, rvd, tsx, tsy, ga, gb, miss); break;
#endif
        case 2: k_advect_velocity_fast<2><<<dim3(gx, (gb - ga + 1) / 2, 1), BX, 0, s>>>(w, vel, out, dt, rW, rH, rvd, tsx, tsy, ga, gb, miss); break;
        default: k_advect_velocity_fast<1><<<dim3(gx, gb - ga, 1), BX, 0, s>>>(w, vel, out, dt, rW, rH, rvd, tsx, tsy, ga, gb, miss); break;
        }
        return hipGetLastError();
    }
    return hipErrorNotReady;  // the caller launches the general per-texel kernel
}

template <class V2, class D4>
hipError_t launch_advect_dye_any(hipStream_t s, Win vw, const V2* vel, Win dw, const D4* dye, D4* out, float dt, float dissipation, int ga, int gb,
                                 unsigned int* miss)
{
    ROWS_OR_RETURN();
    const float tsx = (float)(1.0 / vw.W), tsy = (float)(1.0 / vw.H), decay = 1.0f + dissipation * dt;
    if (!(vw.W == dw.W && vw.H == dw.H) && advect_fast_ok(dw, sizeof(D4), decay, decay) && advect_fast_ok(vw, sizeof(V2), decay, decay)) {
        const double rW = udiv_recip((float)dw.W), rH = udiv_recip((float)dw.H), rdd = udiv_recip(decay);
        const unsigned gx = (dw.x1 - dw.x0 + BX - 1) / BX;
#ifdef FLUID_PROBES
        if constexpr (sizeof(V2) == sizeof(float2)) {
            const int rows_ = split_advect_rows((long)(dw.x1 - dw.x0) * (gb - ga));
            if (const int wy = advect_wy(); wy > 1 && (rows_ == 2 || rows_ == 4)) {
                const unsigned cw = BX / wy, gxx = (dw.x1 - dw.x0 + cw - 1) / cw, gyy = (gb - ga + rows_ * wy - 1) / (rows_ * wy);
#define WY_CASE(R, Y) if (rows_ == R && wy == Y) k_advect_dye_fast_wy<R, Y><<<dim3(gxx, gyy, 1), BX, 0, s>>>(vw, vel, dw, dye, out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss);
                WY_CASE(2, 2) WY_CASE(2, 4) WY_CASE(4, 2) WY_CASE(4, 4)
#undef WY_CASE
                return hipGetLastError();
            }
        }
#endif
        if constexpr (sizeof(V2) == sizeof(float2)) {
            if (velocity_tile()) {
                switch (split_advect_rows_vt((long)(dw.x1 - dw.x0) * (gb - ga))) {
                case 4: k_advect_dye_fast_vt<4><<<dim3(gx, (gb - ga + 3) / 4, 1), BX, 0, s>>>(vw, vel, dw, dye, out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss); break;
                case 2: k_advect_dye_fast_vt<2><<<dim3(gx, (gb - ga + 1) / 2, 1), BX, 0, s>>>(vw, vel, dw, dye, out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss); break;
                default: k_advect_dye_fast_vt<1><<<dim3(gx, gb - ga, 1), BX, 0, s>>>(vw, vel, dw, dye, out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss); break;
                }
                return hipGetLastError();
            }
        }
#ifndef FLUID_PROBES
        if constexpr (sizeof(V2) == sizeof(float2)) return hipErrorNotReady;   // (unreachable: the product library has no FLUID_VTILE=0)
        else
#endif
        {
            switch (split_advect_rows((long)(dw.x1 - dw.x0) * (gb - ga))) {
#ifdef FLUID_PROBES
            case 4: k_advect_dye_fast<4><<<dim3(gx, (gb - ga + 3) / 4, 1), BX, 0, s>>>(vw, vel, dw, dye, out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss); break;
#endif
            case 2: k_advect_dye_fast<2><<<dim3(gx, (gb - ga + 1) / 2, 1), BX, 0, s>>>(vw, vel, dw, dye, out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss); break;
            default: k_advect_dye_fast<1><<<dim3(gx, gb - ga, 1), BX, 0, s>>>(vw, vel, dw, dye, out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss); break;
            }
            return hipGetLastError();
        }
    }
    return hipErrorNotReady;
}

hipError_t launch_advect_velocity(hipStream_t s, Win w, const float2* vel, float2* out, float dt, float dissipation, int ga,
                                  int gb, unsigned int* miss)
{
    ROWS_OR_RETURN();
    const hipError_t e = launch_advect_velocity_any(s, w, vel, out, dt, dissipation, ga, gb, miss);
    if (e != hipErrorNotReady) return e;
    k_advect_velocity<<<row_grid(w, ga, gb), BX, 0, s>>>(w, vel, out, dt, dissipation, (float)(1.0 / w.W), (float)(1.0 / w.H), ga, miss);
    return hipGetLastError();
}

hipError_t launch_advect_dye(hipStream_t s, Win vw, const float2* vel, Win dw, const float4* dye, float4* out, float dt,
                             float dissipation, int ga, int gb, unsigned int* miss)
{
    ROWS_OR_RETURN();
    const hipError_t e = launch_advect_dye_any(s, vw, vel, dw, dye, out, dt, dissipation, ga, gb, miss);
    if (e != hipErrorNotReady) return e;
    const float tsx = (float)(1.0 / vw.W), tsy = (float)(1.0 / vw.H);  // velocity.texelSizeX/Y, script.js:1061-1062, 1276
    if (vw.W == dw.W && vw.H == dw.H)
        k_advect_dye<true><<<row_grid(dw, ga, gb), BX, 0, s>>>(vw, vel, dw, dye, out, dt, dissipation, tsx, tsy, ga, miss);
    else
        k_advect_dye<false><<<row_grid(dw, ga, gb), BX, 0, s>>>(vw, vel, dw, dye, out, dt, dissipation, tsx, tsy, ga, miss);
    return hipGetLastError();
}

// k_advect_both_fast serves when advect_fast_ok (above) holds for the dye array and both decay divisors; otherwise the general kernel
// texels per thread of the fast kernel (FLUID_ADVECT_ROWS / FLUID_ADVECT_ROWS_F16: A/B knobs); the general kernel runs with two
static int advect_rows(const char* env, int dflt)
{
    const char* e = lab_env(env);
    const int k = e ? atoi(e) : dflt;
    return (k == 1 || k == 2 || k == 3 || k == 4 || k == 6 || k == 8) ? k : dflt;
}
#define ADVECT_FAST_CASE(K, R)                                                                                                        \
    case R:                                                                                                                           \
        K<R><<<dim3(gx, (gb - ga + R - 1) / R, 1), BX, 0, s>>>(w, vel, vel_out, dye, dye_out, dt, rW, rH, rvd, rdd, tsx, tsy, ga, gb, miss); \
        break;

hipError_t launch_advect_both(hipStream_t s, Win w, const float2* vel, float2* vel_out, const float4* dye, float4* dye_out,
                              float dt, float vel_dissipation, float dye_dissipation, int ga, int gb, unsigned int* miss)
{
    ROWS_OR_RETURN();
    static const int rows = advect_rows("FLUID_ADVECT_ROWS", 4);
    const float tsx = (float)(1.0 / w.W), tsy = (float)(1.0 / w.H);
    const unsigned gx = (w.x1 - w.x0 + BX - 1) / BX;
    const float vdecay = 1.0f + vel_dissipation * dt, ddecay = 1.0f + dye_dissipation * dt;
    if (advect_fast_ok(w, sizeof(float4), vdecay, ddecay)) {
        const double rW = udiv_recip((float)w.W), rH = udiv_recip((float)w.H), rvd = udiv_recip(vdecay), rdd = udiv_recip(ddecay);
#ifdef FLUID_PROBES
        // FLUID_ADVECT_WY=2 / 4 (lab): the block's four waves stacked 2 x 2 / 1 x 4 instead of side by side, with FLUID_ADVECT_ROWS 4 or 2
        if (const int wy = advect_wy(); wy > 1 && (rows == 4 || rows == 2)) {
            const unsigned cw = BX / wy, gxx = (w.x1 - w.x0 + cw - 1) / cw, gyy = (gb - ga + rows * wy - 1) / (rows * wy);
#define WY_CASE(R, Y) if (rows == R && wy == Y) k_advect_both_fast_wy<R, Y><<<dim3(gxx, gyy, 1), BX, 0, s>>>(w, vel, vel_out, dye, dye_out, dt, rW, rH, rvd, rdd, tsx, tsy, ga, gb, miss);
            WY_CASE(4, 2) WY_CASE(4, 4) WY_CASE(2, 2) WY_CASE(2, 4)
#undef WY_CASE
            return hipGetLastError();
        }
#endif
        switch (rows) {
#ifdef FLUID_PROBES
            ADVECT_FAST_CASE(k_advect_both_fast, 1)
            ADVECT_FAST_CASE(k_advect_both_fast, 2)
            ADVECT_FAST_CASE(k_advect_both_fast, 3)
            ADVECT_FAST_CASE(k_advect_both_fast, 6)
            ADVECT_FAST_CASE(k_advect_both_fast, 8)
#endif
            ADVECT_FAST_CASE(k_advect_both_fast, 4)
        }
        return hipGetLastError();
    }
    k_advect_both<2><<<dim3(gx, (gb - ga + 1) / 2, 1), BX, 0, s>>>(w, vel, vel_out, dye, dye_out, dt, vel_dissipation, dye_dissipation, tsx, tsy, ga, gb, miss);
    return hipGetLastError();
}

hipError_t launch_advect_both(hipStream_t s, Win w, const __half2* vel, __half2* vel_out, const half4* dye, half4* dye_out, float dt,
                              float vel_dissipation, float dye_dissipation, int ga, int gb, unsigned int* miss)
{
    ROWS_OR_RETURN();
    static const int rows = advect_rows("FLUID_ADVECT_ROWS_F16", 4);
    const float tsx = (float)(1.0 / w.W), tsy = (float)(1.0 / w.H);
    const unsigned gx = (w.x1 - w.x0 + BX - 1) / BX;
    const float vdecay = 1.0f + vel_dissipation * dt, ddecay = 1.0f + dye_dissipation * dt;
    if (advect_fast_ok(w, sizeof(half4), vdecay, ddecay)) {
        const double rW = udiv_recip((float)w.W), rH = udiv_recip((float)w.H), rvd = udiv_recip(vdecay), rdd = udiv_recip(ddecay);
        switch (rows) {
#ifdef FLUID_PROBES
            ADVECT_FAST_CASE(k_advect_both_fast_h, 1)
            ADVECT_FAST_CASE(k_advect_both_fast_h, 2)
            ADVECT_FAST_CASE(k_advect_both_fast_h, 3)
            ADVECT_FAST_CASE(k_advect_both_fast_h, 6)
            ADVECT_FAST_CASE(k_advect_both_fast_h, 8)
#endif
            ADVECT_FAST_CASE(k_advect_both_fast_h, 4)
        }
        return hipGetLastError();
    }
    k_advect_both_h<2><<<dim3(gx, (gb - ga + 1) / 2, 1), BX, 0, s>>>(w, vel, vel_out, dye, dye_out, dt, vel_dissipation, dye_dissipation, tsx, tsy, ga, gb, miss);
    return hipGetLastError();
}

hipError_t launch_advect_both_rgb(hipStream_t s, Win w, const float2* vel, float2* vel_out, const rgb3* dye, rgb3* dye_out, float dt,
                                  float vel_dissipation, float dye_dissipation, int ga, int gb, unsigned int* miss)
{
    ROWS_OR_RETURN();
    const float vdecay = 1.0f + vel_dissipation * dt, ddecay = 1.0f + dye_dissipation * dt;
    if (!advect_fast_ok(w, sizeof(float4), vdecay, ddecay)) return hipErrorNotReady;   // the RGBA array's bound: what the buffer was sized for
    const float tsx = (float)(1.0 / w.W), tsy = (float)(1.0 / w.H);
    const unsigned gx = (w.x1 - w.x0 + BX - 1) / BX;
    const double rW = udiv_recip((float)w.W), rH = udiv_recip((float)w.H), rvd = udiv_recip(vdecay), rdd = udiv_recip(ddecay);
#ifdef FLUID_PROBES
    static const int rows = advect_rows("FLUID_ADVECT_ROWS", 4);   // (lab: texels per thread, as for the RGBA kernel)
    static const int xcd_cols = [] { const char* e = lab_env("FLUID_ADVECT_XCD"); return e ? atoi(e) : 0; }();
    if (const int wy = advect_wy(); (wy > 1 || xcd_cols) && (rows == 4 || rows == 2)) {   // FLUID_ADVECT_WY / FLUID_ADVECT_XCD (lab)
        const int cw = BX / wy, gxx = (w.x1 - w.x0 + cw - 1) / cw, gyy = (gb - ga + rows * wy - 1) / (rows * wy);
#define WY_CASE(R, Y) if (rows == R && wy == Y) k_advect_both_fast_rgb_wy<R, Y><<<dim3(gxx * gyy, 1, 1), BX, 0, s>>>(w, vel, vel_out, dye, dye_out, dt, rW, rH, rvd, rdd, tsx, tsy, ga, gb, miss, gxx, xcd_cols);
        WY_CASE(4, 1) WY_CASE(4, 2) WY_CASE(4, 4) WY_CASE(2, 2) WY_CASE(2, 4)
#undef WY_CASE
        return hipGetLastError();
    }
    static const bool pair3 = [] { const char* e = lab_env("FLUID_RGB_PAIR"); return e && atoi(e) != 0; }();
    if (pair3 && rows == 4) {
        k_advect_both_fast_rgb_pair<4><<<dim3(gx, (gb - ga + 3) / 4, 1), BX, 0, s>>>(w, vel, vel_out, dye, dye_out, dt, rW, rH, rvd, rdd, tsx, tsy, ga, gb, miss);
        return hipGetLastError();
    }
    switch (rows) {
        ADVECT_FAST_CASE(k_advect_both_fast_rgb, 2)
        ADVECT_FAST_CASE(k_advect_both_fast_rgb, 3)
        ADVECT_FAST_CASE(k_advect_both_fast_rgb, 6)
        ADVECT_FAST_CASE(k_advect_both_fast_rgb, 8)
    default: break;
    }
    if (rows == 2 || rows == 3 || rows == 6 || rows == 8) return hipGetLastError();
#endif
    k_advect_both_fast_rgb<4><<<dim3(gx, (gb - ga + 3) / 4, 1), BX, 0, s>>>(w, vel, vel_out, dye, dye_out, dt, rW, rH, rvd, rdd, tsx, tsy, ga, gb, miss);
    return hipGetLastError();
}

#undef ADVECT_FAST_CASE

bool advect_rgb_supported(Win vw, Win dw, float dt, float vel_dissipation, float dye_dissipation)
{
    const float vdecay = 1.0f + vel_dissipation * dt, ddecay = 1.0f + dye_dissipation * dt;
    if (vw.W == dw.W && vw.H == dw.H) return advect_fast_ok(dw, sizeof(float4), vdecay, ddecay);
    return advect_fast_ok(dw, sizeof(float4), ddecay, ddecay) && advect_fast_ok(vw, sizeof(float2), ddecay, ddecay);
}

// the dye pass alone on the packed field (dye grid != sim grid); hipErrorNotReady where the fast kernel does not apply
hipError_t launch_advect_dye_rgb(hipStream_t s, Win vw, const float2* vel, Win dw, const rgb3* dye, rgb3* out, float dt, float dissipation, int ga,
                                 int gb, unsigned int* miss)
{
    ROWS_OR_RETURN();
    const float tsx = (float)(1.0 / vw.W), tsy = (float)(1.0 / vw.H), decay = 1.0f + dissipation * dt;
    if ((vw.W == dw.W && vw.H == dw.H) || !advect_fast_ok(dw, sizeof(float4), decay, decay) || !advect_fast_ok(vw, sizeof(float2), decay, decay))
        return hipErrorNotReady;
    const double rW = udiv_recip((float)dw.W), rH = udiv_recip((float)dw.H), rdd = udiv_recip(decay);
    const unsigned gx = (dw.x1 - dw.x0 + BX - 1) / BX;
#ifdef FLUID_PROBES
    if (dye_box() && velocity_tile() && split_advect_rows_vt((long)(dw.x1 - dw.x0) * (gb - ga)) == 4) {
        k_advect_dye_fast_rgb_box<4><<<dim3(gx, (gb - ga + 3) / 4, 1), BX, 0, s>>>(vw, vel, dw, dye, out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss);
        return hipGetLastError();
    }
    static const bool pair3 = [] { const char* e = lab_env("FLUID_RGB_PAIR"); return e && atoi(e) != 0; }();
    if (pair3 && velocity_tile() && split_advect_rows_vt((long)(dw.x1 - dw.x0) * (gb - ga)) == 4) {
        k_advect_dye_fast_rgb_vt_pair<4><<<dim3(gx, (gb - ga + 3) / 4, 1), BX, 0, s>>>(vw, vel, dw, dye, out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss);
        return hipGetLastError();
    }
#endif
    if (velocity_tile()) {
        switch (split_advect_rows_vt((long)(dw.x1 - dw.x0) * (gb - ga))) {
        case 4: k_advect_dye_fast_rgb_vt<4><<<dim3(gx, (gb - ga + 3) / 4, 1), BX, 0, s>>>(vw, vel, dw, dye, out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss); break;
        case 2: k_advect_dye_fast_rgb_vt<2><<<dim3(gx, (gb - ga + 1) / 2, 1), BX, 0, s>>>(vw, vel, dw, dye, out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss); break;
        default: k_advect_dye_fast_rgb_vt<1><<<dim3(gx, gb - ga, 1), BX, 0, s>>>(vw, vel, dw, dye, out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss); break;
        }
        return hipGetLastError();
    }
#ifdef FLUID_PROBES
    if (split_advect_rows((long)(dw.x1 - dw.x0) * (gb - ga)) == 1)
        k_advect_dye_fast_rgb<1><<<dim3(gx, gb - ga, 1), BX, 0, s>>>(vw, vel, dw, dye, out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss);
    else
        k_advect_dye_fast_rgb<2><<<dim3(gx, (gb - ga + 1) / 2, 1), BX, 0, s>>>(vw, vel, dw, dye, out, dt, rW, rH, rdd, tsx, tsy, ga, gb, miss);
    return hipGetLastError();
#else
    return hipErrorNotReady;   // (unreachable: the product library has no FLUID_VTILE=0)
#endif
}

hipError_t launch_splat_dye_rgb(hipStream_t s, Win w, const rgb3* base, rgb3* out, float x, float y, float aspect, float radius, float c0,
                                float c1, float c2, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_splat_dye_rgb<<<row_grid(w, ga, gb), BX, 0, s>>>(w, base, out, x, y, aspect, radius, c0, c1, c2, ga);
    return hipGetLastError();
}

hipError_t launch_dye_pack(hipStream_t s, const float4* rgba, rgb3* rgb, size_t n)
{
    if (n == 0) return hipSuccess;
    k_dye_pack<<<(unsigned)std::min<size_t>((n + BX - 1) / BX, 8192), BX, 0, s>>>(rgba, rgb, n);
    return hipGetLastError();
}

hipError_t launch_dye_unpack(hipStream_t s, const rgb3* rgb, float4* rgba, size_t n, float alpha)
{
    if (n == 0) return hipSuccess;
    k_dye_unpack<<<(unsigned)std::min<size_t>((n + BX - 1) / BX, 8192), BX, 0, s>>>(rgb, rgba, n, alpha);
    return hipGetLastError();
}

hipError_t launch_splat_velocity(hipStream_t s, Win w, const float2* base, float2* out, float x, float y, float aspect,
                                 float radius, float c0, float c1, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_splat_velocity<<<row_grid(w, ga, gb), BX, 0, s>>>(w, base, out, x, y, aspect, radius, c0, c1, ga);
    return hipGetLastError();
}

hipError_t launch_splat_dye(hipStream_t s, Win w, const float4* base, float4* out, float x, float y, float aspect, float radius,
                            float c0, float c1, float c2, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_splat_dye<<<row_grid(w, ga, gb), BX, 0, s>>>(w, base, out, x, y, aspect, radius, c0, c1, c2, ga);
    return hipGetLastError();
}

hipError_t launch_resample(hipStream_t s, Win sw, const float* src, int nc, Win dw, float* dst)
{
    const dim3 g((dw.W + BX - 1) / BX, dw.H, 1);
    if (nc == 2) k_resample<2><<<g, BX, 0, s>>>(sw, src, dw, dst);
    else if (nc == 4) k_resample<4><<<g, BX, 0, s>>>(sw, src, dw, dst);
    else k_resample<1><<<g, BX, 0, s>>>(sw, src, dw, dst);
    return hipGetLastError();
}

hipError_t launch_copy_rects(hipStream_t s, const CopyRects& R)
{
    if (R.n < 1) return hipSuccess;
    if (R.n > 16) return hipErrorInvalidValue;
    size_t most = 0;
    for (int k = 0; k < R.n; k++) {
        const unsigned u = R.r[k].unit;
        if (u != 16 && u != 8 && u != 4 && u != 2) return hipErrorInvalidValue;
        most = std::max(most, (size_t)R.r[k].line_units * R.r[k].nrows);
    }
    if (most == 0) return hipSuccess;
    const unsigned gx = (unsigned)std::min<size_t>((most + BX - 1) / BX, 2048);
    k_copy_rects<<<dim3(gx, R.n, 1), BX, 0, s>>>(R);
    return hipGetLastError();
}

hipError_t launch_fill(hipStream_t s, float* dst, size_t n, int nc, float v0, float v1, float v2, float v3)
{
    if (n == 0) return hipSuccess;
    const unsigned g = (unsigned)((n + BX - 1) / BX < 4096 ? (n + BX - 1) / BX : 4096);
    if (nc == 2) k_fill<2><<<g, BX, 0, s>>>(dst, n, v0, v1, v2, v3);
    else if (nc == 4) k_fill<4><<<g, BX, 0, s>>>(dst, n, v0, v1, v2, v3);
    else k_fill<1><<<g, BX, 0, s>>>(dst, n, v0, v1, v2, v3);
    return hipGetLastError();
}

bool fused_supported(Win w) { return w.W >= 1 && w.P % 4 == 0 && w.c0 % 4 == 0; }  // any width: the pitch keeps every row float4-aligned

// The first / last rows of a launch that take 8 x 3-row tiles (k_curl_vort_div_mix), in small TILES (converted to whole tile rows at the
// launch's width, like tb_tail): default 360 / 360 — at 4096^2 378 rows each: the pass goes from 83-84 to 75 us (it is the HEAD that pays:
// the first tiles no longer finish loading together), the step -1.1 % (profiles/r03/cvd_small_tile_head_tail.txt).
// A/B knob: FLUID_CVD_TAIL="head_rows,tail_rows" (0,0 = one shape per launch)
struct CvdTail { int head, tail; };   // rows
static CvdTail cvd_tail(int nx)
{
    static const CvdTail forced = [] {
        CvdTail r{ -1, -1 };
        if (const char* e = lab_env("FLUID_CVD_TAIL")) {
            r = CvdTail{ 0, 0 };
            sscanf(e, "%d,%d", &r.head, &r.tail);
        }
        return r;
    }();
    if (forced.head >= 0) return forced;
    const int rows = (360 + nx - 1) / nx * 18;
    return CvdTail{ rows, rows };
}

hipError_t launch_curl_vort_div(hipStream_t s, Win w, const float2* vel, float* curl, float2* vel_out, float* div,
                                float curl_strength, float dt, int ga, int gb)
{
    ROWS_OR_RETURN();
    if (!fused_supported(w)) return hipErrorInvalidValue;
    const CvdTail t = cvd_tail(make_axis(w.x0, w.x1, w.W, 256, 4).n);
    if (VD_NW == 8 && VD_RY == 5 && t.head + t.tail > 0 && gb - ga >= 3 * (t.head + t.tail)) {
        using GA = VortDiv<8, 5>;
        using GB = VortDiv<8, 3>;
        const Axis ax = make_axis(w.x0, w.x1, w.W, GA::TX, GA::AX);
        MixSegs S{};
        int total = 0;
        auto seg = [&](int a, int b, int small) {
            if (b <= a) return;
            const Axis ay = small ? make_axis(a, b, w.H, GB::TY, GB::AY) : make_axis(a, b, w.H, GA::TY, GA::AY);
            const int k = S.n++;
            S.g[k] = a; S.g[k + 1] = b; S.small[k] = small; S.ys[k] = ay.S; S.ny[k] = ay.n; S.blk0[k] = total;
            total += ax.n * ay.n;
        };
        const int lo = ga + t.head;
        int gmid = gb - t.tail;
        if (t.tail > 0) {
            const int base = (lo - GA::AY > 0 ? lo - GA::AY : 0) + GA::TY - GA::AY;
            if (gmid > base) gmid = base + (gmid - base) / GA::VY * GA::VY;
        }
        if (gmid < lo) gmid = lo;
        seg(ga, lo, 1);
        seg(lo, gmid, 0);
        seg(gmid, gb, 1);
        S.blk0[S.n] = total;
        k_curl_vort_div_mix<8, 5, 3><<<dim3(total, 1, 1), dim3(64, 8, 1), 0, s>>>(w, S, vel, curl, vel_out, div, curl_strength, dt, ax.S, ax.n, cvd_remap());
        return hipGetLastError();
    }
    using G = VortDiv<VD_NW, VD_RY>;
    const Axis ax = make_axis(w.x0, w.x1, w.W, G::TX, G::AX), ay = make_axis(ga, gb, w.H, G::TY, G::AY);
    k_curl_vort_div<VD_NW, VD_RY><<<dim3(ax.n * ay.n, 1, 1), dim3(64, VD_NW, 1), 0, s>>>(w, vel, curl, vel_out, div, curl_strength, dt,
                                                                                    ga, gb, ax.S, ay.S, ax.n, ay.n, cvd_remap());
    return hipGetLastError();
}

hipError_t launch_curl_vort_div(hipStream_t s, Win w, const __half2* vel, __half* curl, __half2* vel_out, __half* div, float curl_strength,
                                float dt, int ga, int gb)
{
    ROWS_OR_RETURN();
    if (!fused_supported(w)) return hipErrorInvalidValue;
    using G = VortDiv<VD_NW, VD_RY>;
    const Axis ax = make_axis(w.x0, w.x1, w.W, G::TX, G::AX), ay = make_axis(ga, gb, w.H, G::TY, G::AY);
    k_curl_vort_div_h<VD_NW, VD_RY><<<dim3(ax.n * ay.n, 1, 1), dim3(64, VD_NW, 1), 0, s>>>(w, vel, curl, vel_out, div, curl_strength, dt, ga,
                                                                                      gb, ax.S, ay.S, ax.n, ay.n, cvd_remap());
    return hipGetLastError();
}

// K7a + K7b + the next step's K1 + K2 + K3 (k_advect_cvd) apply to: fp32 fields on one grid held by one domain, wherever the fast advection
// and the fused curl / vorticity / divergence kernels both do.  (Whether fluid_step_n uses it: fluid_solver.cpp chain_enabled.)
bool advect_cvd_supported(Win w, float dt, float vel_dissipation, float dye_dissipation)
{
    const float vdecay = 1.0f + vel_dissipation * dt, ddecay = 1.0f + dye_dissipation * dt;
    return fused_supported(w) && advect_fast_ok(w, sizeof(float4), vdecay, ddecay) && w.g0 == 0 && w.c0 == 0 && w.rows == w.H && w.x0 == 0 &&
           w.x1 == w.W;
}

// the same launch WITHOUT the dye (dye grid != sim grid): only the velocity's decay matters
bool advect_cvd_velocity_supported(Win w, float dt, float vel_dissipation)
{
    const float vdecay = 1.0f + vel_dissipation * dt;
    return fused_supported(w) && advect_fast_ok(w, sizeof(float2), vdecay, vdecay) && w.g0 == 0 && w.c0 == 0 && w.rows == w.H && w.x0 == 0 && w.x1 == w.W;
}

// FLUID_CHAIN_TILE="waves,rows,apron columns" (A/B knob): 4,8,3 | 8,4,3 | 4,4,3 | 4,8,4 | 8,8,3 | 8,8,4 | 16,8,4 | 16,4,3.  Default: four waves of
// eight rows — 64 x 32 texels, of which 58 x 26 are stored: on the small grids that chain by default the smaller workgroup wins (1024^2:
// 14.6 k steps/s against 13.9 k with eight waves, 2048^2: 6.9 k against 6.7 k) and at 4096^2 the two are level; below 768^2 texels the same
// tile as eight waves of four rows (512^2: 18.3 k against 17.6 k; level at 1024^2, behind at 2048^2) (profiles/r03/advect_cvd_chain.txt)
static void advect_cvd_shape(long texels, int& nw, int& ry, int& ax)
{
    static const int forced = [] {
        int a = 0, b = 0, c = 0;
        if (const char* e = lab_env("FLUID_CHAIN_TILE")) sscanf(e, "%d,%d,%d", &a, &b, &c);
        return a * 10000 + b * 100 + c;
    }();
    const int pick = forced ? forced : (texels < 768l * 768l ? 80403 : 40803);
    nw = pick / 10000;
    ry = pick / 100 % 100;
    ax = pick % 100;
}

hipError_t launch_advect_cvd(hipStream_t s, Win w, const float2* vel, float2* vel_out, const float4* dye, float4* dye_out, float* curl,
                             float* div, float2* vel_adv, float dt, float vel_dissipation, float dye_dissipation, float curl_strength, int ga, int gb)
{
    if (vel_adv && !curl) return hipErrorInvalidValue;   // the launch that ends a call stores the curl field as well
    if (!dye && !vel_adv) return hipErrorInvalidValue;   // without the dye the launch stores the advected velocity for the dye pass that follows
    ROWS_OR_RETURN();
    if (!(dye ? advect_cvd_supported(w, dt, vel_dissipation, dye_dissipation) : advect_cvd_velocity_supported(w, dt, vel_dissipation))) return hipErrorInvalidValue;
    const float tsx = (float)(1.0 / w.W), tsy = (float)(1.0 / w.H);
    const float vdecay = 1.0f + vel_dissipation * dt, ddecay = 1.0f + dye_dissipation * dt;
    const double rW = udiv_recip((float)w.W), rH = udiv_recip((float)w.H), rvd = udiv_recip(vdecay), rdd = udiv_recip(ddecay);
    int nw, ry, apron;
    advect_cvd_shape((long)(w.x1 - w.x0) * (gb - ga), nw, ry, apron);
#define ADVECT_CVD_CASE(NW_, RY_, AX_)                                                                                                  \
    if (nw == NW_ && ry == RY_ && apron == AX_) {                                                                                       \
        using G = AdvectCvd<NW_, RY_, AX_>;                                                                                             \
        const Axis ax = make_axis(w.x0, w.x1, w.W, G::TX, G::AX), ay = make_axis(ga, gb, w.H, G::TY, G::AY);                            \
        if (!dye)                                                                                                                       \
            k_advect_cvd<NW_, RY_, AX_, 2, false><<<dim3(ax.n * ay.n, 1, 1), dim3(64, NW_, 1), 0, s>>>(                                 \
                w, vel, vel_out, dye, dye_out, curl, div, vel_adv, dt, rW, rH, rvd, rdd, tsx, tsy, curl_strength, ga, gb, ax.S, ay.S, ax.n, ay.n, cvd_remap()); \
        else if (vel_adv)                                                                                                               \
            k_advect_cvd<NW_, RY_, AX_, 2><<<dim3(ax.n * ay.n, 1, 1), dim3(64, NW_, 1), 0, s>>>(                                        \
                w, vel, vel_out, dye, dye_out, curl, div, vel_adv, dt, rW, rH, rvd, rdd, tsx, tsy, curl_strength, ga, gb, ax.S, ay.S, ax.n, ay.n, cvd_remap()); \
        else if (curl)                                                                                                                  \
            k_advect_cvd<NW_, RY_, AX_, 1><<<dim3(ax.n * ay.n, 1, 1), dim3(64, NW_, 1), 0, s>>>(                                        \
                w, vel, vel_out, dye, dye_out, curl, div, vel_adv, dt, rW, rH, rvd, rdd, tsx, tsy, curl_strength, ga, gb, ax.S, ay.S, ax.n, ay.n, cvd_remap()); \
        else                                                                                                                            \
            k_advect_cvd<NW_, RY_, AX_, 0><<<dim3(ax.n * ay.n, 1, 1), dim3(64, NW_, 1), 0, s>>>(                                        \
                w, vel, vel_out, dye, dye_out, curl, div, vel_adv, dt, rW, rH, rvd, rdd, tsx, tsy, curl_strength, ga, gb, ax.S, ay.S, ax.n, ay.n, cvd_remap()); \
        return hipGetLastError();                                                                                                       \
    }
    ADVECT_CVD_CASE(4, 8, 3)
    ADVECT_CVD_CASE(8, 4, 3)
#ifdef FLUID_PROBES
    ADVECT_CVD_CASE(8, 8, 4)
    ADVECT_CVD_CASE(8, 8, 3)
    ADVECT_CVD_CASE(4, 8, 4)
    ADVECT_CVD_CASE(4, 4, 3)
    ADVECT_CVD_CASE(16, 4, 3)
    ADVECT_CVD_CASE(16, 8, 4)
#endif
#undef ADVECT_CVD_CASE
    return hipErrorInvalidValue;
}

// the tile shape for a pass over `texels` owned texels: the forced one (FLUID_TB_VARIANT), or by grid size
int jacobi_tb_pick(long texels)
{
    const int e = tb_variant_env();
    if (e >= 0) return e;
    for (const TBRule& r : tb_rules())
        if (texels < r.below) return r.shape;
    return kDefaultTB;
}
int jacobi_tb_depth(int shape) { return kTB[shape].hy; }
int jacobi_tb_apron_cols(int shape) { return kTB[shape].hx; }
bool jacobi_tb_has_gradsub(int shape) { return tb_gs_built(shape); }

bool jacobi_tb_supported(Win w) { return fused_supported(w); }


// rows per wave of the two-texel tile by grid size: 8 waves x 4 rows below 768^2 texels (512^2: 26.5 k steps/s against 24.7 k with 5 rows),
// x 5 below 1280^2 (1024^2: 16.7 k against 15.8 k / 15.6 k with 4 / 6), x 6 above (1536^2: 11.1 k against 10.6 k)
// (profiles/r03/jacobi_pair_tile_ab.txt).  FLUID_TB2="waves,rows" forces one (A/B knob): 8,5 | 8,4 | 8,6 | 4,10 | 16,3
static int tb2_shape(long texels)
{
    static const int forced = [] {
        int a = 0, b = 0;
        if (const char* e = lab_env("FLUID_TB2")) sscanf(e, "%d,%d", &a, &b);
        return a * 100 + b;
    }();
    if (forced) return forced;
    return texels < 768l * 768l ? 804 : (texels < 1280l * 1280l ? 805 : 806);
}

template <class T>
hipError_t launch_jacobi_tb_any(hipStream_t s, Win w, const T* p, const T* div, T* p_out, float pscale, int iters, int ga, int gb, int v)
{
    ROWS_OR_RETURN();
    if (!jacobi_tb_supported(w) || v < 0 || v >= kNumTB || !tb_built(v)) return hipErrorInvalidValue;
    if (iters < 1 || iters > kTB[v].hy) return hipErrorInvalidValue;
    if (v == kPairTB) {
        if constexpr (sizeof(T) == 4) {
            switch (tb2_shape((long)(w.x1 - w.x0) * (gb - ga))) {
            case 804: return launch_tb2<8, 4, 12, 10, 4>(s, w, p, div, p_out, pscale, iters, ga, gb);
            case 806: return launch_tb2<8, 6, 12, 10, 3>(s, w, p, div, p_out, pscale, iters, ga, gb);
#ifdef FLUID_PROBES
            case 410: return launch_tb2<4, 10, 12, 10, 4>(s, w, p, div, p_out, pscale, iters, ga, gb);
            case 1603: return launch_tb2<16, 3, 12, 10, 2>(s, w, p, div, p_out, pscale, iters, ga, gb);
#endif
            default: return launch_tb2<8, 5, 12, 10, 3>(s, w, p, div, p_out, pscale, iters, ga, gb);
            }
        }
        v = 8;  // fp16 storage: the same 40-row tile with four texels per lane
    }
    switch (v) {
#define TB_CASE(k, NW, RY, HX, HY, BPC) \
    case k: return launch_tb<NW, RY, HX, HY, BPC>(s, w, p, div, p_out, pscale, iters, ga, gb);
        TB_VARIANTS(TB_CASE)
#undef TB_CASE
    default: return hipErrorInvalidValue;
    }
}

template <class T, class V2>
hipError_t launch_jacobi_tb_gradsub_any(hipStream_t s, Win w, const T* p, const T* div, T* p_out, const V2* vel, V2* vel_out, float pscale,
                                        int iters, int ga, int gb, int v)
{
    ROWS_OR_RETURN();
    if (!jacobi_tb_supported(w) || v < 0 || v >= kNumTB || !tb_gs_built(v)) return hipErrorInvalidValue;
    if (iters < 1 || iters > kTB[v].hy) return hipErrorInvalidValue;
    // This launch stores the pressure AND the velocity over [x0, x1) x [ga, gb): a column range that is not made of whole float4 groups
    // would have to be widened for the velocity (as launch_gradsub4 does) and would then write pressure into columns whose apron inputs
    // are not exact.  The stripe / tile driver only hands whole groups (cols_of); anything else is a caller's error, not something to widen.
    if ((w.x0 & 3) != 0 || ((w.x1 & 3) != 0 && w.x1 != w.W)) return hipErrorInvalidValue;
    if (v == kPairTB) {
        if constexpr (sizeof(T) == 4) {
            switch (tb2_shape((long)(w.x1 - w.x0) * (gb - ga))) {
            case 804: return launch_tb2_gs<8, 4, 12, 10, 4>(s, w, p, div, p_out, vel, vel_out, pscale, iters, ga, gb);
            case 806: return launch_tb2_gs<8, 6, 12, 10, 3>(s, w, p, div, p_out, vel, vel_out, pscale, iters, ga, gb);
#ifdef FLUID_PROBES
            case 410: return launch_tb2_gs<4, 10, 12, 10, 4>(s, w, p, div, p_out, vel, vel_out, pscale, iters, ga, gb);
            case 1603: return launch_tb2_gs<16, 3, 12, 10, 2>(s, w, p, div, p_out, vel, vel_out, pscale, iters, ga, gb);
#endif
            default: return launch_tb2_gs<8, 5, 12, 10, 3>(s, w, p, div, p_out, vel, vel_out, pscale, iters, ga, gb);
            }
        }
        v = 8;
    }
    switch (v) {
#define TB_CASE(k, NW, RY, HX, HY, BPC) \
    case k: return launch_tb_gs<NW, RY, HX, HY, BPC>(s, w, p, div, p_out, vel, vel_out, pscale, iters, ga, gb);
        TB_GS_VARIANTS(TB_CASE)
#undef TB_CASE
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_jacobi_tb(hipStream_t s, Win w, const float* p, const float* div, float* p_out, float pscale, int iters, int ga, int gb, int shape)
{
    return launch_jacobi_tb_any(s, w, p, div, p_out, pscale, iters, ga, gb, shape);
}

static int jacobi_chain_mode()   // FLUID_JACOBI_CHAIN (lab build): 0 = never, 1 = wherever it can run; 2, 3, 4 = its timing probes (k_jacobi_tb_chain DIAG 1, 2, 3: invalid results); unset: the rule below
{
    static const int m = [] { const char* e = lab_env("FLUID_JACOBI_CHAIN"); return e ? atoi(e) : -1; }();
    return m;
}
// Where the chained launch is the shipped path: it removes all but one of a step's fill / drain phases (~8 us each at 4096^2) and pays two
// memory round trips per tile that do no work (the poll of 3 ... 9 counters in front, the drain of the write-through stores behind).  Measured
// over widths x set sizes (profiles/r06/chain_loop_map.txt, chain_rotation_ab.txt):
//   * it pays where the loop's set — 12 B/texel: two pressure buffers and the divergence — fits the 256 MB Infinity Cache, which answers those
//     round trips in a third of the time HBM takes under load: 4096 x 4880 -5 ... -11 % of the loop, 4096 x 5461 (268 MB) level, 4096 x 8192 +4 %,
//     8192^2 +6.5 %, a 16384 x 2048 rank at 200 iterations +2.7 %;
//   * at ANY width once the bands rotate over the XCDs from block to block (ChainPlan::rot; round 6): with the same XCD in every block a partly
//     empty last group of bands loads the XCDs unevenly — 8192 x 2048 (35 tile rows = 12 bands of 3: two per block for XCDs 0 ... 3, one for the
//     others) measured +10 %, 16384 x 1024 +12 % — and the waits couple every front to the slowest one.  Rotated by five bands per block:
//     3072^2 -12 %, 4096 x 2560 -15 ... -20 %, 4096^2 -12 ... -15 % (it was -7 %), 5120 x 3276 -10 %, 8192 x 2048 -4 %, 12288 x 1366 -14 %,
//     16384 x 1024 -5 %, 2048 x 8192 -14 %.
// So: the large-grid tile (>= 3072^2 texels: pass_jacobi), >= 1024 rows, <= 20 M texels (240 MB), 11 ... 240 iterations (24 blocks).
bool jacobi_chain_applies(const Win& w, int ga, int gb, int iters)
{
    const int mode = jacobi_chain_mode();
    if (mode == 0 || iters <= 10 || iters > 10 * jacobi_chain_max_blocks() || !jacobi_tb_supported(w)) return false;
    if (mode >= 1) return true;
    return gb - ga >= 1024 && (long)(gb - ga) * (long)(w.x1 - w.x0) <= 20000000L;
}
size_t jacobi_pchain_state_bytes();
// the counters of either form (k_jacobi_tb_chain: (block, tile row); k_jacobi_pchain: two halves of heads + (block, stack row, panel) cells)
size_t jacobi_chain_flag_bytes() { return std::max((size_t)(CHAIN_MAX_BLOCKS * CHAIN_MAX_ROWS * CHAIN_MAX_PANELS) * sizeof(unsigned int), jacobi_pchain_state_bytes()); }
// FLUID_CHAIN_PERSIST=1 (lab build): the persistent form, k_jacobi_pchain — placement-independent and general, and measured 20 ... 50 % slower on
// the loop than one workgroup per tile (profiles/r06/pchain_*.txt): a lab kernel.  Default, and all the product has: k_jacobi_tb_chain.
static bool chain_persistent()
{
    static const bool on = [] { const char* e = lab_env("FLUID_CHAIN_PERSIST"); return e && atoi(e) != 0; }();
    return on;
}
int jacobi_chain_max_blocks() { return chain_persistent() ? PCHAIN_MAX_BLOCKS : CHAIN_MAX_BLOCKS; }
hipError_t launch_jacobi_pchain_ranges(hipStream_t s, Win w, float* pa, float* pb, const float* div, float pscale, int nblocks, const int* iters,
                                       const int* ga, const int* gb, const int* xa, const int* xb, unsigned int* state, unsigned int* err, ChainEpoch* ep);

// `iters` iterations as ONE launch of ceil(iters / 10) chained blocks (k_jacobi_tb_chain; the 80-row tile of shape 0).  pa holds the input;
// the result is in pb when the number of blocks is odd, in pa when it is even (*result_in_b).  hipErrorNotReady: does not apply here.
// The general form: `nblocks` blocks of iters[l] (<= 10) iterations storing rows [ga[l], gb[l]) — ga non-decreasing, gb non-increasing (a stripe's
// launches recompute fewer ghost rows each); block 0's range carries the tiling.  pa holds the input; the result is in pb when nblocks is odd.
hipError_t launch_jacobi_tb_chain_ranges(hipStream_t s, Win w, float* pa, float* pb, const float* div, float pscale, int nblocks, const int* iters,
                                         const int* ga, const int* gb, const int* xa, const int* xb, unsigned int* flags, unsigned int* err,
                                         ChainEpoch* ep, const float2* vel, float2* vel_out)
{
    using G = JacobiTB<8, 10, 12, 10>;
    if (chain_persistent()) return vel_out ? hipErrorNotReady : launch_jacobi_pchain_ranges(s, w, pa, pb, div, pscale, nblocks, iters, ga, gb, xa, xb, flags, err, ep);
    if (nblocks < 2 || nblocks > CHAIN_MAX_BLOCKS || gb[0] <= ga[0] || xb[0] <= xa[0]) return hipErrorNotReady;
    ChainPlan C{};
    C.blocks = nblocks;
    if (vel_out) {   // the gradient subtract as one more block (GSB): entry [nblocks] of the ranges is what IT stores
#ifndef FLUID_PROBES
        return hipErrorNotReady;   // (a lab form)
#endif
        if (!vel || ga[nblocks] < ga[0] || gb[nblocks] > gb[0] || xa[nblocks] < xa[0] || xb[nblocks] > xb[0] || jacobi_chain_mode() > 1) return hipErrorNotReady;
        C.gs = 1;
        C.ga[nblocks] = ga[nblocks];
        C.gb[nblocks] = gb[nblocks];
        C.xa[nblocks] = xa[nblocks];
        C.xb[nblocks] = xb[nblocks];
    }
    for (int l = 0; l < nblocks; l++) {
        if (iters[l] < 1 || iters[l] > 10 || ga[l] < ga[0] || gb[l] > gb[0] || xa[l] < xa[0] || xb[l] > xb[0]) return hipErrorNotReady;
        C.iters[l] = iters[l];
        C.ga[l] = ga[l];
        C.gb[l] = gb[l];
        C.xa[l] = xa[l];
        C.xb[l] = xb[l];
    }
    const Axis ax = make_axis(xa[0], xb[0], w.W, G::TX, 12), ay = make_axis(ga[0], gb[0], w.H, G::TY, 10);
    if (ay.n > CHAIN_MAX_ROWS) return hipErrorNotReady;
    // panels, and rows per band: what keeps ONE band of an XCD's panel inside the 64 workgroups resident there (32 CUs x 2).  4096-wide: 18 tiles per
    // row, one panel -> 3 rows (4 rows = 72 tiles spill and the order alone costs 12 %; 2 rows leave a third of the XCD to the next band: +2 %:
    // profiles/r05/jacobi_chain_ab.txt).  Wider rows: panels of at most 21 tiles (8192-wide: 36 = 2 x 18; 16384-wide: 71 = 4 x 18) — one
    // panel of 36 tiles and one-row bands had no tile's vertical neighbour on its own XCD (+3.6 % at 8192^2 in round 5)
    static const int forced = [] { const char* e = lab_env("FLUID_CHAIN_BAND"); return e ? atoi(e) : -1; }();   // 0 = the first form (contiguous runs, odd blocks backwards)
    static const int forced_pw = [] { const char* e = lab_env("FLUID_CHAIN_PANEL"); return e ? atoi(e) : -1; }();   // tiles per panel; 0 = one panel (round 5)
    int pw = forced_pw > 0 ? std::min(forced_pw, ax.n) : (forced_pw == 0 ? ax.n : chain_panel_width(ax.n, 21));
    if (chain_panels(ax.n, pw) > CHAIN_MAX_PANELS) pw = (ax.n + CHAIN_MAX_PANELS - 1) / CHAIN_MAX_PANELS;
    const int band = forced >= 0 ? forced : std::max(1, 64 / pw);
    static const int tickets = [] { const char* e = lab_env("FLUID_CHAIN_TICKET"); return e ? atoi(e) : 0; }();
    C.band = band > 0 ? band : 0;
    C.pw = C.band > 0 ? pw : ax.n;
    // five bands on per block: the XCDs' loads even out over the launch (the rule above; FLUID_CHAIN_ROT: 3 and 5 measure alike, 7 is best at
    // 4096^2 alone and worst elsewhere, 0 = round 5's fixed assignment: profiles/r06/chain_rotation_ab.txt)
    static const int rot = [] { const char* e = lab_env("FLUID_CHAIN_ROT"); return e ? atoi(e) & 7 : 5; }();
    C.rot = rot;
    C.tickets = tickets != 0;   // (lab; err[1] is the ticket word)
    C.tiles = C.band > 0 ? chain_slots(ax.n, ay.n, C.band) : ax.n * ay.n;
    // the counters: zeroed when the shape of the call changes (tiles per row, tile rows, blocks: what decides which counters a call bumps, and by
    // how much), counted up from call to call otherwise — a memset in front of every launch was 5 us of the step and a kernel boundary
    hipError_t e = hipSuccess;
    const unsigned sig = (unsigned)ax.n | ((unsigned)ay.n << 7) | ((unsigned)nblocks << 16) | ((unsigned)C.band << 21) | ((unsigned)C.pw << 25);   // (ay.n <= 512, nblocks <= 24, band <= 64, pw <= 127)
    if (!ep || ep->signature != sig || ep->calls >= (1u << 24)) {
        e = hipMemsetAsync(flags, 0, jacobi_chain_flag_bytes(), s);
        if (e != hipSuccess) return e;
        if (ep) {
            ep->signature = sig;
            ep->calls = 0;
        }
    }
    static const int timeout_ms = [] { const char* e = lab_env("FLUID_CHAIN_TIMEOUT_MS"); return e ? atoi(e) : 2000; }();
    static const int withhold = [] { const char* e = lab_env("FLUID_CHAIN_WITHHOLD"); return e ? atoi(e) : -1; }();
    C.timeout = (unsigned int)timeout_ms * 100000u;
    C.withhold = withhold;
    C.target = (ep ? ep->calls : 0u) + 1u;
    const dim3 grid((unsigned)((C.blocks + C.gs) * C.tiles), 1, 1), block(64, 8, 1);
#ifdef FLUID_PROBES
    if (C.tickets) {
        e = hipMemsetAsync(err + 1, 0, sizeof(unsigned int), s);
        if (e != hipSuccess) return e;
    }
#endif
    switch (jacobi_chain_mode()) {
#ifdef FLUID_PROBES
    case 2: k_jacobi_tb_chain<8, 10, 12, 10, 2, 1><<<grid, block, 0, s>>>(w, pa, pb, div, pscale, C, ax.S, ay.S, ax.n, ay.n, flags, err); break;
    case 3: k_jacobi_tb_chain<8, 10, 12, 10, 2, 2><<<grid, block, 0, s>>>(w, pa, pb, div, pscale, C, ax.S, ay.S, ax.n, ay.n, flags, err); break;
    case 4: k_jacobi_tb_chain<8, 10, 12, 10, 2, 3><<<grid, block, 0, s>>>(w, pa, pb, div, pscale, C, ax.S, ay.S, ax.n, ay.n, flags, err); break;
    case 5: k_jacobi_tb_chain<8, 10, 12, 10, 2, 5><<<grid, block, 0, s>>>(w, pa, pb, div, pscale, C, ax.S, ay.S, ax.n, ay.n, flags, err); break;
    case 6: k_jacobi_tb_chain<8, 10, 12, 10, 2, 6><<<grid, block, 0, s>>>(w, pa, pb, div, pscale, C, ax.S, ay.S, ax.n, ay.n, flags, err); break;
#endif
    default: {
#ifdef FLUID_PROBES
        // Two lab forms of the tile, both bitwise, both measured LEVEL (profiles/r06/chain_dfirst_ab.txt, chain_skip_ab.txt) — the loop hides its
        // arithmetic and the poll's round trip alike (chain_bounds_probes.txt): FLUID_CHAIN_DFIRST=1: the tile's divergence loads go out in front of
        // the poll; FLUID_CHAIN_SKIP=1: its first / last wave skip the rows the apron has reached (light blocks on waves 0 and 2: one per SIMD)
        static const int dfirst = [] { const char* e = lab_env("FLUID_CHAIN_DFIRST"); return e ? atoi(e) : 0; }();
        static const int skip = [] { const char* e = lab_env("FLUID_CHAIN_SKIP"); return e ? atoi(e) : 0; }();
        if (C.gs) k_jacobi_tb_chain<8, 10, 12, 10, 2, 0, false, false, true><<<grid, block, 0, s>>>(w, pa, pb, div, pscale, C, ax.S, ay.S, ax.n, ay.n, flags, err, vel, vel_out);
        else if (dfirst) k_jacobi_tb_chain<8, 10, 12, 10, 2, 0, false, true><<<grid, block, 0, s>>>(w, pa, pb, div, pscale, C, ax.S, ay.S, ax.n, ay.n, flags, err);
        else if (skip) k_jacobi_tb_chain<8, 10, 12, 10, 2, 0, true, false><<<grid, block, 0, s>>>(w, pa, pb, div, pscale, C, ax.S, ay.S, ax.n, ay.n, flags, err);
        else
#endif
            k_jacobi_tb_chain<8, 10, 12, 10, 2, 0><<<grid, block, 0, s>>>(w, pa, pb, div, pscale, C, ax.S, ay.S, ax.n, ay.n, flags, err);
        break;
    }
    }
    const hipError_t rc = hipGetLastError();
    if (ep) {
        // the counters count up from call to call: only a launch that went out has bumped them (a failed one would leave every later call
        // waiting for a count nobody makes — ADVICE r05); anything else: the next call zeroes them
        if (rc == hipSuccess) ep->calls++;
        else ep->signature = 0xffffffffu;
        ep->psig[0] = 0xffffffffu;   // (the persistent form's state shares the memory)
    }
    return rc;
}

// ---- the persistent form (k_jacobi_pchain; fluid_pchain.h) ----
static int pchain_knob(const char* name, int dflt)
{
    const char* e = lab_env(name);
    return e ? atoi(e) : dflt;
}
size_t jacobi_pchain_state_bytes() { return 2 * (size_t)(8 * PCHAIN_HEAD_STRIDE + PCHAIN_MAX_BLOCKS * PCHAIN_MAX_CELLS + PCHAIN_MAX_ITEMS) * sizeof(unsigned int); }

// Tiles per stack, column panels and stack rows per band for an nx x ny tiling of `blocks` blocks.  A band should fill one XCD's 64 resident
// workgroups (its tiles share their aprons in that XCD's L2) and the bands of the launch should deal out evenly over the eight heads.
// FLUID_CHAIN_STACK / FLUID_CHAIN_PANEL / FLUID_CHAIN_BAND (lab build) force M / pw / bh.
void pchain_layout(PChainPlan& C)
{
    static const int f_pw = pchain_knob("FLUID_CHAIN_PANEL", 0), f_bh = pchain_knob("FLUID_CHAIN_BAND", 0);
    const int np0 = (C.d.nx + 20) / 21;
    C.d.pw = f_pw > 0 ? std::min(f_pw, C.d.nx) : (C.d.nx + np0 - 1) / np0;
    C.d.np = (C.d.nx + C.d.pw - 1) / C.d.pw;
    int best = 1;
    double best_score = -1.0;
    for (int bh = 1; bh <= std::max(1, 64 / C.d.pw); bh++) {
        const int tb = C.d.blocks * C.d.np * ((C.d.ny + bh - 1) / bh);
        const double balance = (double)tb / (8.0 * ((tb + 7) / 8)), fill = std::min(1.0, (double)(C.d.pw * bh) / 56.0);
        const double score = balance * (0.75 + 0.25 * fill);   // (an even deal first; among even deals the fuller band)
        if (score > best_score + 1e-9) {
            best_score = score;
            best = bh;
        }
    }
    C.d.bh = f_bh > 0 ? f_bh : best;
    pchain_finish(C.d);
}

// stacks of M tiles: tile rows per launch / M stack rows.  M = 2 by default (0.8125 of the arithmetic stored instead of 0.75; M = 3: 0.833 and
// half as many, twice as long items again); 1 = plain tiles
int pchain_stack()
{
    static const int m = std::max(1, std::min(8, pchain_knob("FLUID_CHAIN_STACK", 2)));
    return m;
}

hipError_t launch_jacobi_pchain_ranges(hipStream_t s, Win w, float* pa, float* pb, const float* div, float pscale, int nblocks, const int* iters,
                                       const int* ga, const int* gb, const int* xa, const int* xb, unsigned int* state, unsigned int* err, ChainEpoch* ep)
{
    using G = JacobiTB<8, 10, 12, 10>;
    using S = JacobiStack<8, 10, 12, 10>;
    if (nblocks < 2 || nblocks > PCHAIN_MAX_BLOCKS || gb[0] <= ga[0] || xb[0] <= xa[0] || !ep) return hipErrorNotReady;
    PChainPlan C{};
    C.d.blocks = nblocks;
    for (int l = 0; l < nblocks; l++) {
        if (iters[l] < 1 || iters[l] > 10 || ga[l] < ga[0] || gb[l] > gb[0] || xa[l] < xa[0] || xb[l] > xb[0]) return hipErrorNotReady;
        C.iters[l] = iters[l];
        C.ga[l] = ga[l];
        C.gb[l] = gb[l];
        C.xa[l] = xa[l];
        C.xb[l] = xb[l];
    }
    C.d.stack = pchain_stack();
    const Axis ax = make_axis(xa[0], xb[0], w.W, G::TX, 12), ay = make_axis(ga[0], gb[0], w.H, S::span(C.d.stack), 10);
    C.d.nx = ax.n;
    C.d.ny = ay.n;
    C.d.xs = ax.S;
    C.d.ys = ay.S;
    pchain_layout(C);
    if (C.d.ny * C.d.np > PCHAIN_MAX_CELLS || C.d.nx >= 4096 || C.d.ny >= 4096 || (long)C.d.blocks * C.d.nx * C.d.ny > PCHAIN_MAX_ITEMS) return hipErrorNotReady;
    static const int timeout_ms = pchain_knob("FLUID_CHAIN_TIMEOUT_MS", 2000);
    C.d.timeout = (unsigned int)timeout_ms * 100000u;
    static const int withhold = pchain_knob("FLUID_CHAIN_WITHHOLD", -1);
    C.d.withhold = withhold;
    static const int stagger_us = pchain_knob("FLUID_CHAIN_STAGGER_US", 0);
    C.d.stagger = stagger_us * 100;
    // the state words: zeroed when the shape of the call changes, else the half the previous call of this shape zeroed for us
    const unsigned sig = (unsigned)C.d.nx | ((unsigned)C.d.ny << 10) | ((unsigned)nblocks << 20) | ((unsigned)C.d.stack << 25) | ((unsigned)C.d.bh << 28);
    const unsigned sig2 = (unsigned)C.d.pw;
    if (ep->psig[0] != sig || ep->psig[1] != sig2) {
        const hipError_t e = hipMemsetAsync(state, 0, jacobi_pchain_state_bytes(), s);
        if (e != hipSuccess) return e;
        ep->psig[0] = 0xffffffffu;   // (until the launch below has gone out)
        ep->bank = 0;
    }
    C.d.bank = ep->bank;
    static const int cus = [] {
        hipDeviceProp_t p{};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
        return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }();
    static const int f_grid = pchain_knob("FLUID_CHAIN_GRID", 0);
    // 8 S workgroups: S owners per XCD sequence (S = what an XCD holds: two workgroups per CU), every position has exactly one owner
    const unsigned grid = (unsigned)std::max(8, ((f_grid > 0 ? f_grid : 2 * cus) / 8) * 8);
#ifdef FLUID_PROBES
    if (jacobi_chain_mode() == 4) k_jacobi_pchain<8, 10, 12, 10, 3><<<dim3(grid), dim3(64, 8), 0, s>>>(w, pa, pb, div, pscale, C, state, err);
    else k_jacobi_pchain<8, 10, 12, 10, 0><<<dim3(grid), dim3(64, 8), 0, s>>>(w, pa, pb, div, pscale, C, state, err);
#else
    (void)grid;
    return hipErrorNotReady;   // (the product library does not carry the persistent form)
#endif
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) {   // only a launch that went out has used its half and zeroed the other
        ep->psig[0] = sig;
        ep->psig[1] = sig2;
        ep->bank ^= 1;
    } else {
        ep->psig[0] = 0xffffffffu;
    }
    ep->signature = 0xffffffffu;   // (the other form's counters share the memory: whoever runs next zeroes them)
    return e;
}

// `iters` iterations over the rows [ga, gb) in every block (a whole domain) as ceil(iters / 10) balanced blocks
hipError_t launch_jacobi_tb_chain(hipStream_t s, Win w, float* pa, float* pb, const float* div, float pscale, int iters, int ga, int gb,
                                  unsigned int* flags, unsigned int* err, int* blocks, bool* result_in_b, ChainEpoch* ep)
{
    ROWS_OR_RETURN();
    if (!jacobi_chain_applies(w, ga, gb, iters)) return hipErrorNotReady;
    const int n = (iters + 9) / 10;
    int it[PCHAIN_MAX_BLOCKS], a[PCHAIN_MAX_BLOCKS], b[PCHAIN_MAX_BLOCKS], xa[PCHAIN_MAX_BLOCKS], xb[PCHAIN_MAX_BLOCKS], done = 0, left = n;
    for (int l = 0; l < n; l++) {   // balanced, as pass_jacobi cuts them
        it[l] = (iters - done + left - 1) / left;
        done += it[l];
        left--;
        a[l] = ga;
        b[l] = gb;
        xa[l] = w.x0;
        xb[l] = w.x1;
    }
    *blocks = n;
    *result_in_b = (n & 1) != 0;
    return launch_jacobi_tb_chain_ranges(s, w, pa, pb, div, pscale, n, it, a, b, xa, xb, flags, err, ep);
}
hipError_t launch_jacobi_tb(hipStream_t s, Win w, const __half* p, const __half* div, __half* p_out, float pscale, int iters, int ga, int gb, int shape)
{
    return launch_jacobi_tb_any(s, w, p, div, p_out, pscale, iters, ga, gb, shape);
}
hipError_t launch_jacobi_tb_gradsub(hipStream_t s, Win w, const float* p, const float* div, float* p_out, const float2* vel, float2* vel_out,
                                    float pscale, int iters, int ga, int gb, int shape)
{
    return launch_jacobi_tb_gradsub_any(s, w, p, div, p_out, vel, vel_out, pscale, iters, ga, gb, shape);
}
hipError_t launch_jacobi_tb_gradsub(hipStream_t s, Win w, const __half* p, const __half* div, __half* p_out, const __half2* vel,
                                    __half2* vel_out, float pscale, int iters, int ga, int gb, int shape)
{
    return launch_jacobi_tb_gradsub_any(s, w, p, div, p_out, vel, vel_out, pscale, iters, ga, gb, shape);
}

// ---- several bands in one launch (the strips of a 2-D tile) ----
template <class V2, class D4>
hipError_t launch_advect_both_rects_any(hipStream_t s, Win w, const V2* vel, V2* vel_out, const D4* dye, D4* dye_out, float dt, float vel_dissipation,
                                        float dye_dissipation, const BandRects& B, unsigned int* miss)
{
    const float vdecay = 1.0f + vel_dissipation * dt, ddecay = 1.0f + dye_dissipation * dt;
    AdvRects R{};
    constexpr int ROWS = 4;
    int total = 0;
    for (int k = 0; k < B.n; k++) {
        const BandRect& q = B.r[k];
        if (q.gb <= q.ga || q.xb <= q.xa) continue;
        const int i = R.n++;
        R.xa[i] = q.xa; R.xb[i] = q.xb; R.ga[i] = q.ga; R.gb[i] = q.gb;
        R.nbx[i] = (q.xb - q.xa + BX - 1) / BX;
        R.blk0[i] = total;
        total += R.nbx[i] * ((q.gb - q.ga + ROWS - 1) / ROWS);
    }
    R.blk0[R.n] = total;
    if (R.n == 0) return hipSuccess;
    if (R.n > 1 && advect_fast_ok(w, sizeof(D4), vdecay, ddecay)) {
        const float tsx = (float)(1.0 / w.W), tsy = (float)(1.0 / w.H);
        const double rW = udiv_recip((float)w.W), rH = udiv_recip((float)w.H), rvd = udiv_recip(vdecay), rdd = udiv_recip(ddecay);
        k_advect_both_fast_rects<ROWS><<<dim3(total, 1, 1), BX, 0, s>>>(w, R, vel, vel_out, dye, dye_out, dt, rW, rH, rvd, rdd, tsx, tsy, miss);
        return hipGetLastError();
    }
    for (int i = 0; i < R.n; i++) {  // one band, or the general kernel: a launch per rectangle
        Win wb = w;
        wb.x0 = R.xa[i];
        wb.x1 = R.xb[i];
        const hipError_t e = launch_advect_both(s, wb, vel, vel_out, dye, dye_out, dt, vel_dissipation, dye_dissipation, R.ga[i], R.gb[i], miss);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_advect_both_rects(hipStream_t s, Win w, const float2* vel, float2* vel_out, const float4* dye, float4* dye_out, float dt,
                                    float vel_dissipation, float dye_dissipation, const BandRects& B, unsigned int* miss)
{
    return launch_advect_both_rects_any(s, w, vel, vel_out, dye, dye_out, dt, vel_dissipation, dye_dissipation, B, miss);
}
hipError_t launch_advect_both_rects(hipStream_t s, Win w, const __half2* vel, __half2* vel_out, const half4* dye, half4* dye_out, float dt,
                                    float vel_dissipation, float dye_dissipation, const BandRects& B, unsigned int* miss)
{
    return launch_advect_both_rects_any(s, w, vel, vel_out, dye, dye_out, dt, vel_dissipation, dye_dissipation, B, miss);
}
// the strips of a stripe / tile on the PACKED dye field (the caller checked advect_rgb_supported: the fast kernel applies)
hipError_t launch_advect_both_rects_rgb(hipStream_t s, Win w, const float2* vel, float2* vel_out, const rgb3* dye, rgb3* dye_out, float dt,
                                        float vel_dissipation, float dye_dissipation, const BandRects& B, unsigned int* miss)
{
    const float vdecay = 1.0f + vel_dissipation * dt, ddecay = 1.0f + dye_dissipation * dt;
    if (!advect_fast_ok(w, sizeof(float4), vdecay, ddecay)) return hipErrorNotReady;
    AdvRects R{};
    constexpr int ROWS = 4;
    int total = 0;
    for (int k = 0; k < B.n; k++) {
        const BandRect& q = B.r[k];
        if (q.gb <= q.ga || q.xb <= q.xa) continue;
        const int i = R.n++;
        R.xa[i] = q.xa; R.xb[i] = q.xb; R.ga[i] = q.ga; R.gb[i] = q.gb;
        R.nbx[i] = (q.xb - q.xa + BX - 1) / BX;
        R.blk0[i] = total;
        total += R.nbx[i] * ((q.gb - q.ga + ROWS - 1) / ROWS);
    }
    R.blk0[R.n] = total;
    if (R.n == 0) return hipSuccess;
    const float tsx = (float)(1.0 / w.W), tsy = (float)(1.0 / w.H);
    const double rW = udiv_recip((float)w.W), rH = udiv_recip((float)w.H), rvd = udiv_recip(vdecay), rdd = udiv_recip(ddecay);
    k_advect_both_fast_rects<ROWS><<<dim3(total, 1, 1), BX, 0, s>>>(w, R, vel, vel_out, dye, dye_out, dt, rW, rH, rvd, rdd, tsx, tsy, miss);
    return hipGetLastError();
}

template <class V2, class S1>
hipError_t launch_curl_vort_div_rects_any(hipStream_t s, Win w, const V2* vel, S1* curl, V2* vel_out, S1* div, float curl_strength, float dt,
                                          const BandRects& B)
{
    if (!fused_supported(w)) return hipErrorInvalidValue;
    using G = VortDiv<VD_NW, VD_RY>;
    TileRects R{};
    int total = 0;
    for (int k = 0; k < B.n; k++) {
        const BandRect& q = B.r[k];
        if (q.gb <= q.ga || q.xb <= q.xa) continue;
        const int i = R.n++;
        const Axis ax = make_axis(q.xa, q.xb, w.W, G::TX, G::AX), ay = make_axis(q.ga, q.gb, w.H, G::TY, G::AY);
        R.x0[i] = q.xa; R.x1[i] = q.xb; R.ga[i] = q.ga; R.gb[i] = q.gb;
        R.xs[i] = ax.S; R.ys[i] = ay.S; R.nx[i] = ax.n; R.ny[i] = ay.n;
        R.blk0[i] = total;
        total += ax.n * ay.n;
    }
    R.blk0[R.n] = total;
    if (R.n == 0) return hipSuccess;
    k_curl_vort_div_rects<VD_NW, VD_RY><<<dim3(total, 1, 1), dim3(64, VD_NW, 1), 0, s>>>(w, R, vel, curl, vel_out, div, curl_strength, dt, cvd_remap());
    return hipGetLastError();
}

// every product shape has a 12-column / 10-row apron (kTB); the frame launch runs the four-texel tile with 7 rows per wave whatever
// shape the pass picked — same arithmetic per texel, hence the same bits
hipError_t launch_jacobi_tb_rects(hipStream_t s, Win w, const float* p, const float* div, float* p_out, float pscale, int iters, const BandRects& B)
{
    constexpr int NW = 8, RY = 7, HX = 12, HY = 10;
    using G = JacobiTB<NW, RY, HX, HY>;
    if (!jacobi_tb_supported(w) || iters < 1 || iters > HY) return hipErrorInvalidValue;
    TileRects R{};
    int total = 0;
    for (int k = 0; k < B.n; k++) {
        const BandRect& q = B.r[k];
        if (q.gb <= q.ga || q.xb <= q.xa) continue;
        const int i = R.n++;
        const Axis ax = make_axis(q.xa, q.xb, w.W, G::TX, HX), ay = make_axis(q.ga, q.gb, w.H, G::TY, HY);
        R.x0[i] = q.xa; R.x1[i] = q.xb; R.ga[i] = q.ga; R.gb[i] = q.gb;
        R.xs[i] = ax.S; R.ys[i] = ay.S; R.nx[i] = ax.n; R.ny[i] = ay.n;
        R.blk0[i] = total;
        total += ax.n * ay.n;
    }
    R.blk0[R.n] = total;
    if (R.n == 0) return hipSuccess;
    k_jacobi_tb_rects<NW, RY, HX, HY, 2><<<dim3(total, 1, 1), dim3(64, NW, 1), 0, s>>>(w, R, p, div, p_out, pscale, iters, xcd_remap());
    return hipGetLastError();
}

hipError_t launch_curl_vort_div_rects(hipStream_t s, Win w, const float2* vel, float* curl, float2* vel_out, float* div, float curl_strength,
                                      float dt, const BandRects& B)
{
    return launch_curl_vort_div_rects_any(s, w, vel, curl, vel_out, div, curl_strength, dt, B);
}
hipError_t launch_curl_vort_div_rects(hipStream_t s, Win w, const __half2* vel, __half* curl, __half2* vel_out, __half* div, float curl_strength,
                                      float dt, const BandRects& B)
{
    return launch_curl_vort_div_rects_any(s, w, vel, curl, vel_out, div, curl_strength, dt, B);
}

}  // namespace fluid

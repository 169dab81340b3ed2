// fluid_kernels_f16.hip — the simulation passes on fp16-STORAGE fields (fluid_desc.storage = FLUID_STORE_F16).
//
// on a real GPU the reference keeps every simulation field in half-float textures
// (`halfFloatTexType` script.js:138, formats 145-147, framebuffers 995-1006): each pass reads halves, computes in fp32
// and its render-target write rounds the result back to fp16.  These kernels do exactly that — the per-texel arithmetic
// is the SAME fp32 code as the fp32-storage kernels (`*_texel` bodies of fluid_math.h), only the loads widen and the
// stores narrow (round to nearest even, v_cvt_f16_f32) — so a step moves half the bytes of the fp32 mode.
// Parity: the oracle restatement with an fp16 round trip after every pass output (oracle/oracle.py storage="f16"), and the live
// reference with half-float render targets emulated around the unmodified page (tests/test_f16_vs_golden.py).
#include "fluid_kernels.h"
#include "fluid_math.h"

namespace fluid {

namespace {

constexpr int BX = 256;  // threads per block: 4 waves along a row

#define TEXEL_OR_RETURN(win)                                  \
    const int i = (win).x0 + blockIdx.x * BX + threadIdx.x; \
    const int gj = ga + blockIdx.y;                           \
    if (i >= (win).x1) return

__global__ void __launch_bounds__(BX) k_h_curl(Win w, const __half2* __restrict__ vel, __half* __restrict__ curl, int ga)
{
    TEXEL_OR_RETURN(w);
    curl_texel(w, vel, curl, i, gj);
}

__global__ void __launch_bounds__(BX) k_h_vorticity(Win w, const __half2* __restrict__ vel, const __half* __restrict__ curl,
                                                     __half2* __restrict__ vel_out, float curl_strength, float dt, int ga)
{
    TEXEL_OR_RETURN(w);
    vorticity_texel(w, vel, curl, vel_out, curl_strength, dt, i, gj);
}

__global__ void __launch_bounds__(BX) k_h_divergence(Win w, const __half2* __restrict__ vel, __half* __restrict__ div, int ga)
{
    TEXEL_OR_RETURN(w);
    divergence_texel(w, vel, div, i, gj);
}

__global__ void __launch_bounds__(BX) k_h_clear(Win w, const __half* __restrict__ p, __half* __restrict__ p_out, float value, int ga)
{
    TEXEL_OR_RETURN(w);
    clear_texel(w, p, p_out, value, i, gj);
}

__global__ void __launch_bounds__(BX) k_h_jacobi(Win w, const __half* __restrict__ p, const __half* __restrict__ div,
                                                  __half* __restrict__ p_out, int ga)
{
    TEXEL_OR_RETURN(w);
    jacobi_texel(w, p, div, p_out, i, gj);
}

__global__ void __launch_bounds__(BX) k_h_gradsub(Win w, const __half* __restrict__ p, const __half2* __restrict__ vel,
                                                   __half2* __restrict__ vel_out, int ga)
{
    TEXEL_OR_RETURN(w);
    gradsub_texel(w, p, vel, vel_out, i, gj);
}

__global__ void __launch_bounds__(BX) k_h_advect_velocity(Win w, const __half2* __restrict__ vel, __half2* __restrict__ out, float dt,
                                                           float dissipation, float tsx, float tsy, int ga,
                                                           unsigned int* __restrict__ miss_out)
{
    TEXEL_OR_RETURN(w);
    const int miss = advect_velocity_texel(w, vel, out, dt, dissipation, tsx, tsy, i, gj);
    if (miss) atomicAdd(miss_out, (unsigned)miss);
}

template <bool SAME_RES>
__global__ void __launch_bounds__(BX) k_h_advect_dye(Win vw, const __half2* __restrict__ vel, Win dw, const half4* __restrict__ dye,
                                                      half4* __restrict__ out, float dt, float dissipation, float tsx, float tsy, int ga,
                                                      unsigned int* __restrict__ miss_out)
{
    TEXEL_OR_RETURN(dw);
    const int miss = advect_dye_texel<SAME_RES>(vw, vel, dw, dye, out, dt, dissipation, tsx, tsy, i, gj);
    if (miss) atomicAdd(miss_out, (unsigned)miss);
}

__global__ void __launch_bounds__(BX) k_h_splat_velocity(Win w, const __half2* __restrict__ base, __half2* __restrict__ out, float x, float y,
                                                          float aspect, float radius, float c0, float c1, int ga)
{
    TEXEL_OR_RETURN(w);
    splat_velocity_texel(w, base, out, x, y, aspect, radius, c0, c1, i, gj);
}

__global__ void __launch_bounds__(BX) k_h_splat_dye(Win w, const half4* __restrict__ base, half4* __restrict__ out, float x, float y,
                                                     float aspect, float radius, float c0, float c1, float c2, int ga)
{
    TEXEL_OR_RETURN(w);
    splat_dye_texel(w, base, out, x, y, aspect, radius, c0, c1, c2, i, gj);
}

template <int NC>
__global__ void __launch_bounds__(BX) k_h_resample(Win sw, const __half* __restrict__ src, Win dw, __half* __restrict__ dst)
{
    const int i = blockIdx.x * BX + threadIdx.x;
    const int gj = blockIdx.y;
    if (i >= dw.W) return;
    resample_texel<NC>(sw, src, dw, dst, i, gj);
}

template <int NC>
__global__ void __launch_bounds__(BX) k_h_fill(__half* __restrict__ dst, size_t n, float v0, float v1, float v2, float v3)
{
    const float vals[4] = { v0, v1, v2, v3 };
    for (size_t i = (size_t)blockIdx.x * BX + threadIdx.x; i < n; i += (size_t)gridDim.x * BX)
        for (int k = 0; k < NC; k++) dst[i * NC + k] = __float2half_rn(vals[k]);
}

__global__ void __launch_bounds__(BX) k_h_widen(const __half* __restrict__ src, float* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * BX + threadIdx.x; i < n; i += (size_t)gridDim.x * BX) dst[i] = __half2float(src[i]);
}

__global__ void __launch_bounds__(BX) k_h_narrow(const float* __restrict__ src, __half* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * BX + threadIdx.x; i < n; i += (size_t)gridDim.x * BX) dst[i] = __float2half_rn(src[i]);
}

inline dim3 row_grid(const Win& w, int ga, int gb) { return dim3((w.x1 - w.x0 + BX - 1) / BX, gb - ga, 1); }
inline unsigned flat_grid(size_t n) { return (unsigned)((n + BX - 1) / BX < 4096 ? (n + BX - 1) / BX : 4096); }

}  // namespace

#define ROWS_OR_RETURN() \
    if (gb <= ga) return hipSuccess

hipError_t launch_curl(hipStream_t s, Win w, const __half2* vel, __half* curl, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_h_curl<<<row_grid(w, ga, gb), BX, 0, s>>>(w, vel, curl, ga);
    return hipGetLastError();
}

hipError_t launch_vorticity(hipStream_t s, Win w, const __half2* vel, const __half* curl, __half2* vel_out, float curl_strength, float dt,
                            int ga, int gb)
{
    ROWS_OR_RETURN();
    k_h_vorticity<<<row_grid(w, ga, gb), BX, 0, s>>>(w, vel, curl, vel_out, curl_strength, dt, ga);
    return hipGetLastError();
}

hipError_t launch_divergence(hipStream_t s, Win w, const __half2* vel, __half* div, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_h_divergence<<<row_grid(w, ga, gb), BX, 0, s>>>(w, vel, div, ga);
    return hipGetLastError();
}

hipError_t launch_clear(hipStream_t s, Win w, const __half* p, __half* p_out, float value, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_h_clear<<<row_grid(w, ga, gb), BX, 0, s>>>(w, p, p_out, value, ga);
    return hipGetLastError();
}

hipError_t launch_jacobi(hipStream_t s, Win w, const __half* p, const __half* div, __half* p_out, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_h_jacobi<<<row_grid(w, ga, gb), BX, 0, s>>>(w, p, div, p_out, ga);
    return hipGetLastError();
}

hipError_t launch_gradsub(hipStream_t s, Win w, const __half* p, const __half2* vel, __half2* vel_out, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_h_gradsub<<<row_grid(w, ga, gb), BX, 0, s>>>(w, p, vel, vel_out, ga);
    return hipGetLastError();
}

hipError_t launch_advect_velocity(hipStream_t s, Win w, const __half2* vel, __half2* out, float dt, float dissipation, int ga, int gb,
                                  unsigned int* miss)
{
    ROWS_OR_RETURN();
    k_h_advect_velocity<<<row_grid(w, ga, gb), BX, 0, s>>>(w, vel, out, dt, dissipation, (float)(1.0 / w.W), (float)(1.0 / w.H), ga, miss);
    return hipGetLastError();
}

hipError_t launch_advect_dye(hipStream_t s, Win vw, const __half2* vel, Win dw, const half4* dye, half4* out, float dt, float dissipation,
                             int ga, int gb, unsigned int* miss)
{
    ROWS_OR_RETURN();
    const float tsx = (float)(1.0 / vw.W), tsy = (float)(1.0 / vw.H);  // velocity.texelSizeX/Y, script.js:1061-1062, 1276
    if (vw.W == dw.W && vw.H == dw.H)
        k_h_advect_dye<true><<<row_grid(dw, ga, gb), BX, 0, s>>>(vw, vel, dw, dye, out, dt, dissipation, tsx, tsy, ga, miss);
    else
        k_h_advect_dye<false><<<row_grid(dw, ga, gb), BX, 0, s>>>(vw, vel, dw, dye, out, dt, dissipation, tsx, tsy, ga, miss);
    return hipGetLastError();
}

hipError_t launch_splat_velocity(hipStream_t s, Win w, const __half2* base, __half2* out, float x, float y, float aspect, float radius,
                                 float c0, float c1, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_h_splat_velocity<<<row_grid(w, ga, gb), BX, 0, s>>>(w, base, out, x, y, aspect, radius, c0, c1, ga);
    return hipGetLastError();
}

hipError_t launch_splat_dye(hipStream_t s, Win w, const half4* base, half4* out, float x, float y, float aspect, float radius, float c0,
                            float c1, float c2, int ga, int gb)
{
    ROWS_OR_RETURN();
    k_h_splat_dye<<<row_grid(w, ga, gb), BX, 0, s>>>(w, base, out, x, y, aspect, radius, c0, c1, c2, ga);
    return hipGetLastError();
}

hipError_t launch_resample(hipStream_t s, Win sw, const __half* src, int nc, Win dw, __half* dst)
{
    const dim3 g((dw.W + BX - 1) / BX, dw.H, 1);
    if (nc == 2) k_h_resample<2><<<g, BX, 0, s>>>(sw, src, dw, dst);
    else if (nc == 4) k_h_resample<4><<<g, BX, 0, s>>>(sw, src, dw, dst);
    else k_h_resample<1><<<g, BX, 0, s>>>(sw, src, dw, dst);
    return hipGetLastError();
}

hipError_t launch_fill(hipStream_t s, __half* dst, size_t n, int nc, float v0, float v1, float v2, float v3)
{
    if (n == 0) return hipSuccess;
    if (nc == 2) k_h_fill<2><<<flat_grid(n), BX, 0, s>>>(dst, n, v0, v1, v2, v3);
    else if (nc == 4) k_h_fill<4><<<flat_grid(n), BX, 0, s>>>(dst, n, v0, v1, v2, v3);
    else k_h_fill<1><<<flat_grid(n), BX, 0, s>>>(dst, n, v0, v1, v2, v3);
    return hipGetLastError();
}

hipError_t launch_widen(hipStream_t s, const __half* src, float* dst, size_t n)
{
    if (n == 0) return hipSuccess;
    k_h_widen<<<flat_grid(n), BX, 0, s>>>(src, dst, n);
    return hipGetLastError();
}

hipError_t launch_narrow(hipStream_t s, const float* src, __half* dst, size_t n)
{
    if (n == 0) return hipSuccess;
    k_h_narrow<<<flat_grid(n), BX, 0, s>>>(src, dst, n);
    return hipGetLastError();
}

}  // namespace fluid

// fluid_display.h — launch interface of the display-compositor kernels (fluid_display.hip).  Internal.
#pragma once
#include <hip/hip_runtime.h>

namespace fluid {

struct DisplayArgs {
    const float4* dye;
    int dye_w, dye_h;
    const float4* bloom;   // nullptr: BLOOM off
    int bloom_w, bloom_h;
    const float* sunrays;  // nullptr: SUNRAYS off
    int sun_w, sun_h;
    const float* dither;   // R channel in [0, 1], REPEAT + LINEAR
    int dither_w, dither_h;
    float4* frame;
    int w, h;
    int shading, transparent;
    float back_r, back_g, back_b;
};

hipError_t launch_bloom_prefilter(hipStream_t s, const float4* dye, int dw, int dh, float4* out, int w, int h, float c0, float c1, float c2,
                                  float threshold);
hipError_t launch_box4(hipStream_t s, const float4* src, int sw, int sh, float4* dst, int w, int h, int add, int scaled, float scale);
hipError_t launch_sunrays_mask(hipStream_t s, const float4* dye, float4* mask, size_t n);
hipError_t launch_sunrays(hipStream_t s, const float4* mask, int mw, int mh, float* out, int w, int h, float weight);
hipError_t launch_blur3(hipStream_t s, const float* src, float* dst, int w, int h, int horizontal);
hipError_t launch_display(hipStream_t s, const DisplayArgs& a);
hipError_t launch_normalize(hipStream_t s, const float4* frame, unsigned char* out_rgba8, int w, int h);

}  // namespace fluid

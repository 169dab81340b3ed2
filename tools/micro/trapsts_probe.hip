// What does a wave's sticky exception status (HW_REG_TRAPSTS.EXCP) record on gfx950?  One wave per case; each lane computes the same thing.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/micro/trapsts_probe.hip -o /tmp/trapsts_probe && /tmp/trapsts_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned excp()
{
    unsigned e;
    asm volatile("s_nop 15\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(e));
    return e;
}
__device__ __forceinline__ unsigned mode()
{
    unsigned e;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE, 0, 32)" : "=s"(e));
    return e;
}

__global__ void probe(const float* in, float* out, unsigned* flags, int which)
{
    const float a = in[0], b = in[1], c = in[2];
    unsigned e0 = excp();
    float r = 0;
    v2f pr = { 0, 0 };
    switch (which) {
    case 0: r = __builtin_fmaf(a, b, c); break;                                   // v_fma_f32
    case 1: { v2f A = { a, a }, B = { b, b }, C = { c, c }; pr = __builtin_elementwise_fma(A, B, C); r = pr.x; } break;  // v_pk_fma_f32
    case 2: r = a * b; break;                                                      // v_mul_f32
    case 3: { v2f A = { a, a }, B = { b, b }; pr = A * B; r = pr.y; } break;       // v_pk_mul_f32
    case 4: r = a + c; break;                                                      // v_add_f32
    }
    asm volatile("" : : "v"(r), "v"(pr));
    unsigned e1 = excp();
    out[which * 64 + threadIdx.x] = r;
    if (threadIdx.x == 0) { flags[which * 3] = e0; flags[which * 3 + 1] = e1; flags[which * 3 + 2] = mode(); }
}

int main()
{
    float *in, *out; unsigned* fl;
    hipMalloc(&in, 16); hipMalloc(&out, 5 * 64 * 4); hipMalloc(&fl, 5 * 3 * 4);
    struct { const char* what; float a, b, c; } cases[] = {
        { "normal, inexact            ", 1.1f, 0.3f, 0.7f },
        { "normal, exact              ", 1.0f, 0.25f, 0.5f },
        { "subnormal result, inexact  ", 1.1e-38f, 0.25f, 1e-39f },
        { "subnormal result, exact    ", 0x1p-125f, 0.25f, 0x1p-130f },
        { "subnormal inputs, exact sum", 1e-40f, 1.0f, 1e-41f },
    };
    for (auto& cs : cases) {
        float h[3] = { cs.a, cs.b, cs.c };
        hipMemcpy(in, h, 12, hipMemcpyHostToDevice);
        printf("%s a=%a b=%a c=%a\n", cs.what, cs.a, cs.b, cs.c);
        for (int w = 0; w < 5; w++) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, in, out, fl, w);
        hipDeviceSynchronize();
        float ho[5 * 64]; unsigned hf[15];
        hipMemcpy(ho, out, sizeof ho, hipMemcpyDeviceToHost); hipMemcpy(hf, fl, sizeof hf, hipMemcpyDeviceToHost);
        const char* nm[] = { "v_fma_f32   ", "v_pk_fma_f32", "v_mul_f32   ", "v_pk_mul_f32", "v_add_f32   " };
        for (int w = 0; w < 5; w++)
            printf("   %s -> %-14a EXCP before %03x after %03x (inval %d, in-denorm %d, underflow %d, inexact %d)  MODE %08x\n", nm[w], ho[w * 64], hf[w * 3],
                   hf[w * 3 + 1], hf[w * 3 + 1] & 1, hf[w * 3 + 1] >> 1 & 1, hf[w * 3 + 1] >> 4 & 1, hf[w * 3 + 1] >> 5 & 1, hf[w * 3 + 2]);
    }
    return 0;
}

// Micro-benchmark: issue cost of v_fma_f32 / v_add_f32 / v_pk_add_f32 / v_fma_mix_f32 on this GPU (independent chains, 8 waves
// per SIMD, no memory traffic).  Build and run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/rate valu_rate.hip && /tmp/rate
// Result on MI355X (profiles/r01/valu_issue_rates.txt): packed fp32 costs 1.75 x a scalar op (1.15 x the flops per cycle),
// v_fma_mix_f32 1.55 x — which is why the Jacobi tiles are VALU-bound at the plain fp32 rate whatever the encoding.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int n)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
    unsigned h = 0x3c003c00u + threadIdx.x;
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, 1.0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (KIND == 1) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(h));
            if (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        }
        if (KIND == 2) {
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a[0]) : "v"(*(double*)&a[2]));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a[2]) : "v"(*(double*)&a[4]));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a[4]) : "v"(*(double*)&a[6]));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a[6]) : "v"(*(double*)&a[0]));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
double run(const char* name, int per_iter, int results_per_instr)
{
    float* out = nullptr;
    if (hipMalloc((void**)&out, 256 * 2048 * 4 * sizeof(float)) != hipSuccess) return -1;
    const int n = 4096, blocks = 256 * 8;  // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    k<KIND><<<blocks, 256>>>(out, 16);
    (void)hipEventRecord(a);
    k<KIND><<<blocks, 256>>>(out, n);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * 4 /*waves*/ * n * per_iter;          // wave-instructions
    const double per_simd_per_s = instr / (256.0 * 4) / (ms * 1e-3);           // per SIMD
    printf("%-16s %8.3f ms  %.2f G wave-instr/s per SIMD  -> %.2f cycles per wave-instruction at 2.4 GHz\n", name, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
    (void)hipFree(out);
    return ms;
}
int main()
{
    run<0>("v_fma_f32", 8, 1);
    run<3>("v_add_f32", 8, 1);
    run<2>("v_pk_add_f32", 4, 2);
    run<1>("v_fma_mix_f32", 8, 1);
    return 0;
}

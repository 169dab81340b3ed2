// Micro-benchmark: issue cost of the VALU instructions the advection and curl/vorticity kernels are made of — IEEE divide pieces
// (v_div_scale / v_rcp / v_div_fmas / v_div_fixup), f64 multiply and conversions (the exact divide-by-a-uniform trick of fluid_math.h
// div_uniform), 64-bit address arithmetic, integer clamps.  Independent chains, 8 waves per SIMD, no memory traffic.
// Build and run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/rate2 tools/micro/valu_rate2.hip && /tmp/rate2
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAINS 8
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int n)
{
    float a[CHAINS];
    double d[CHAINS];
    int q[CHAINS];
    long L[CHAINS];
    for (int i = 0; i < CHAINS; i++) {
        a[i] = threadIdx.x * 0.001f + i + 1.0f;
        d[i] = a[i];
        q[i] = threadIdx.x + i;
        L[i] = q[i];
    }
    for (int it = 0; it < n; it++) {
#define ONE(i)                                                                                                                  \
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, 1.0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));                               \
    if (KIND == 1) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));                                              \
    if (KIND == 2) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));                                              \
    if (KIND == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));                                    \
    if (KIND == 4) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));                                \
    if (KIND == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));                                                              \
    if (KIND == 6) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]) : "vcc");             \
    if (KIND == 7) asm volatile("v_div_fmas_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]) : "vcc");                   \
    if (KIND == 8) asm volatile("v_div_fixup_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));                          \
    if (KIND == 9) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));                                                            \
    if (KIND == 10) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(q[i]) : "v"(a[i]));                                             \
    if (KIND == 11) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));                           \
    if (KIND == 12) asm volatile("v_lshl_add_u64 %0, %0, 4, %1" : "+v"(L[i]) : "v"(L[(i + 1) & 7]));                           \
    if (KIND == 13) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(L[i]) : "v"(q[i]), "v"(q[(i + 1) & 7]) : "vcc");   \
    if (KIND == 14) asm volatile("v_min_i32 %0, %0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));                                   \
    if (KIND == 15) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(q[i]) : "v"(q[(i + 1) & 7]) : "vcc");                  \
    if (KIND == 16) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));                                                            \
    if (KIND == 17) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc");                       \
    if (KIND == 18) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));                                   \
    if (KIND == 19) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));                                \
    if (KIND == 20) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));                            \
    if (KIND == 21) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[(i + 1) & 7])); \
    if (KIND == 22) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        REP8(ONE)
#undef ONE
    }
    float s = 0;
    for (int i = 0; i < CHAINS; i++) s += a[i] + (float)d[i] + (float)q[i] + (float)L[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double base_ms = 0;
template <int KIND>
void run(const char* name)
{
    float* out = nullptr;
    if (hipMalloc((void**)&out, 256 * 2048 * sizeof(float)) != hipSuccess) return;
    const int n = 4096, blocks = 256 * 8;  // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    k<KIND><<<blocks, 256>>>(out, 16);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a);
        k<KIND><<<blocks, 256>>>(out, n);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    if (KIND == 0) base_ms = best;
    printf("%-18s %8.3f ms   %.2f x v_fma_f32\n", name, best, best / base_ms);
    (void)hipFree(out);
}
int main()
{
    run<0>("v_fma_f32");
    run<18>("v_mul_f32");
    run<19>("v_pk_mul_f32");
    run<20>("v_pk_fma_f32");
    run<1>("v_cvt_f64_f32");
    run<2>("v_cvt_f32_f64");
    run<3>("v_mul_f64");
    run<4>("v_fma_f64");
    run<5>("v_rcp_f32");
    run<16>("v_sqrt_f32");
    run<6>("v_div_scale_f32");
    run<7>("v_div_fmas_f32");
    run<8>("v_div_fixup_f32");
    run<9>("v_floor_f32");
    run<10>("v_cvt_i32_f32");
    run<11>("v_mad_i32_i24");
    run<12>("v_lshl_add_u64");
    run<13>("v_mad_i64_i32");
    run<14>("v_min_i32");
    run<15>("v_cndmask_b32");
    run<17>("v_cmp_lt_f32");
    run<21>("v_mov_b32_dpp");
    run<22>("v_add_f32_dpp");
    return 0;
}

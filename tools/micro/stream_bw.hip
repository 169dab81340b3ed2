// stream_bw.hip — what this box's memory system sustains for the access mixes of the step kernels (float4 per lane,
// coalesced): read-only, write-only, copy (1R:1W), 2R:1W, 3R:2W (gradient subtract), and the byte mix of the
// advection kernel (24 B read, 24 B written per texel).  Buffers of `MB` MiB each (default 256: a 4096^2 float4 field).
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_bw stream_bw.hip ; run: ./stream_bw [MiB] [reps] [stagger bytes]
// `stagger`: buffer k starts k * stagger bytes into its allocation (a multiple of 16) — do streams that sit at the same offset of
// equally sized, equally aligned arrays collide in the channel / bank mapping?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NR, int NW, bool NT>
__global__ void __launch_bounds__(256) k_mix(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c,
                                             f4* __restrict__ x, f4* __restrict__ y, size_t n, float* sink)
{
    const size_t stride = (size_t)gridDim.x * 256;
    f4 acc = f4{0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        f4 v = f4{1, 2, 3, 4};
        if (NR >= 1) { const f4 t = NT ? __builtin_nontemporal_load(a + i) : a[i]; v += t; }
        if (NR >= 2) { const f4 t = NT ? __builtin_nontemporal_load(b + i) : b[i]; v += t; }
        if (NR >= 3) { const f4 t = NT ? __builtin_nontemporal_load(c + i) : c[i]; v += t; }
        if (NW >= 1) { if (NT) __builtin_nontemporal_store(v, x + i); else x[i] = v; }
        if (NW >= 2) { if (NT) __builtin_nontemporal_store(v, y + i); else y[i] = v; }
        if (NW == 0) { acc += v; }
    }
    if (NW == 0 && acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

template <int NR, int NW, bool NT>
double run(const char* name, f4** buf, size_t n, int blocks, int reps, float* sink)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k = 0; k < 3; k++) k_mix<NR, NW, NT><<<blocks, 256>>>(buf[0], buf[1], buf[2], buf[3], buf[4], n, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int k = 0; k < reps; k++) k_mix<NR, NW, NT><<<blocks, 256>>>(buf[0], buf[1], buf[2], buf[3], buf[4], n, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)(NR + NW) * n * 16.0 * reps;
    const double tbps = bytes / (ms * 1e-3) / 1e12;
    printf("%-28s blocks %6d  %7.1f us/launch  %6.3f TB/s\n", name, blocks, ms * 1e3 / reps, tbps);
    return tbps;
}

int main(int argc, char** argv)
{
    const size_t mib = argc > 1 ? atol(argv[1]) : 256;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const size_t stagger = argc > 3 ? (size_t)atol(argv[3]) & ~(size_t)15 : 0;
    const size_t n = mib * 1024 * 1024 / 16;
    f4* buf[5];
    for (int k = 0; k < 5; k++) {
        char* base;
        CK(hipMalloc(&base, n * 16 + 5 * stagger));
        CK(hipMemset(base, 0, n * 16 + 5 * stagger));
        buf[k] = (f4*)(base + k * stagger);
    }
    float* sink; CK(hipMalloc(&sink, 4));
    printf("# %zu MiB per buffer, %d launches per row, buffer k staggered by k * %zu bytes\n", mib, reps, stagger);
    const int grids_all[] = { 2048, 8192, 65536 };
    const int grids_one[] = { 65536 };
    const bool brief = argc > 3;
    const int* grids = brief ? grids_one : grids_all;
    const int ngrids = brief ? 1 : 3;
    for (int gi = 0; gi < ngrids; gi++) {
        const int g = grids[gi];
        run<1, 0, false>("read", buf, n, g, reps, sink);
        run<0, 1, false>("write", buf, n, g, reps, sink);
        run<1, 1, false>("copy 1R:1W", buf, n, g, reps, sink);
        run<1, 1, true>("copy 1R:1W nontemporal", buf, n, g, reps, sink);
        run<2, 1, false>("2R:1W", buf, n, g, reps, sink);
        run<2, 1, true>("2R:1W nontemporal", buf, n, g, reps, sink);
        run<3, 2, false>("3R:2W", buf, n, g, reps, sink);
        run<2, 2, false>("2R:2W", buf, n, g, reps, sink);
        run<2, 2, true>("2R:2W nontemporal", buf, n, g, reps, sink);
    }
    return 0;
}

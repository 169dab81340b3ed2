// mall_probe.hip — does FETCH_SIZE (the L2's fabric-side read counter) count reads that the 256 MiB Infinity Cache serves?
// A read-only sweep over a working set of S MiB, warm (the set was just read), timed with HIP events; run once plain (rates) and once
// under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (bytes per dispatch; each size is its own kernel name so the rows can be told apart).
// If a warm 192 MiB sweep is FASTER than HBM can deliver (> ~6.3 TB/s) while FETCH_SIZE x 2 KiB still equals the bytes requested, the
// counter includes Infinity-Cache hits and "traffic" measured with it is fabric traffic, not HBM traffic.
// Second part: the Jacobi loop's own mix (two reads, one write-through `sc1` store per texel) over 3 x 64 MiB.
// Build: hipcc --offload-arch=gfx950 -O3 -o mall_probe mall_probe.hip ; run: ./mall_probe [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int TAG>
__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ a, size_t n, float* sink)
{
    const size_t stride = (size_t)gridDim.x * 256;
    f4 acc = f4{ 0, 0, 0, 0 };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

// p_out = p + d, stored write-through (sc1) like k_jacobi_tb_chain's pressure stores
template <int TAG>
__global__ void __launch_bounds__(256) k_2r1w_sc1(const f4* __restrict__ p, const f4* __restrict__ d, f4* __restrict__ q, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)(n * 16), 0x00020000);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const f4 v = p[i] + d[i];
        __builtin_amdgcn_raw_buffer_store_b128(u4{ __float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w) }, r,
                                               (unsigned)(i * 16), 0, 16);
    }
}

template <int TAG>
static void sweep(const f4* buf, size_t mib, int reps, float* sink)
{
    const size_t n = mib * 1024 * 1024 / 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int k = 0; k < 3; k++) k_read<TAG><<<8192, 256>>>(buf, n, sink);   // warm: the set was just read
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int k = 0; k < reps; k++) k_read<TAG><<<8192, 256>>>(buf, n, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_read<%d>  %5zu MiB  %8.1f us per sweep  %6.3f TB/s  (%.1f MB requested per dispatch)\n", TAG, mib, ms * 1e3 / reps,
           (double)mib * 1048576.0 * reps / (ms * 1e-3) / 1e12, mib * 1.048576);
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    f4* buf;
    const size_t big = 2048;
    CK(hipMalloc(&buf, big * 1024 * 1024));
    CK(hipMemset(buf, 0, big * 1024 * 1024));
    float* sink;
    CK(hipMalloc(&sink, 4));
    sweep<0>(buf, 32, reps, sink);
    sweep<1>(buf, 64, reps, sink);
    sweep<2>(buf, 128, reps, sink);
    sweep<3>(buf, 192, reps, sink);
    sweep<4>(buf, 256, reps, sink);
    sweep<5>(buf, 384, reps, sink);
    sweep<6>(buf, 768, reps, sink);
    sweep<7>(buf, 1536, reps, sink);
    // the Jacobi mix: 64 MiB pressure in, 64 MiB divergence in, 64 MiB pressure out (sc1), ping-ponged like the loop does
    {
        const size_t n = 64ull * 1024 * 1024 / 16;
        f4 *p = buf, *d = buf + n, *q = buf + 2 * n;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int k = 0; k < 4; k++) k_2r1w_sc1<0><<<8192, 256>>>(k & 1 ? q : p, d, k & 1 ? p : q, n);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int k = 0; k < reps; k++) k_2r1w_sc1<0><<<8192, 256>>>(k & 1 ? q : p, d, k & 1 ? p : q, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_2r1w_sc1<0>  3 x 64 MiB (192 MiB set)  %8.1f us per pass  %6.3f TB/s  (201.3 MB moved per dispatch: 134.2 read, 67.1 written)\n", ms * 1e3 / reps,
               3.0 * 64 * 1048576.0 * reps / (ms * 1e-3) / 1e12);
        // the same mix on a set that cannot sit in the Infinity Cache: 3 x 640 MiB
        const size_t nb = 640ull * 1024 * 1024 / 16;
        f4 *pb = buf, *db = buf + nb, *qb = buf + 2 * nb;
        for (int k = 0; k < 2; k++) k_2r1w_sc1<1><<<8192, 256>>>(k & 1 ? qb : pb, db, k & 1 ? pb : qb, nb);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int k = 0; k < reps; k++) k_2r1w_sc1<1><<<8192, 256>>>(k & 1 ? qb : pb, db, k & 1 ? pb : qb, nb);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_2r1w_sc1<1>  3 x 640 MiB (1920 MiB set)  %8.1f us per pass  %6.3f TB/s  (2013.3 MB moved per dispatch)\n", ms * 1e3 / reps,
               3.0 * 640 * 1048576.0 * reps / (ms * 1e-3) / 1e12);
    }
    return 0;
}

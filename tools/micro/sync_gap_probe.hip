// sync_gap_probe.hip — what does a cross-stream ordering primitive cost the stream it sits in?  (round 6, VERDICT r05 item 3: a stripe rank's
// step has four idle gaps of 5-12 us on the context stream, one at every event record / wait: profiles/r06/stripe_rank_timeline.txt.)
// Two chip-filling kernels A and B back to back on stream S, with ONE primitive between them; a second stream T plays the comm stream.  Both
// kernels stamp the 100 MHz wall clock (first start of B minus last end of A = the gap the primitive costs S).  Forms:
//   0  nothing between A and B (the stream's own in-order hand-over)
//   1  hipEventRecord(e, S), default event (system-scope release)             — ev_ready today
//   2  hipEventRecord(e, S), hipEventDisableSystemFence                        — ev_inner today
//   3  hipStreamWaitEvent(S, e) on an event T recorded long ago (default)      — ev_landed
//   4  the same with a hipEventDisableSystemFence event                        — ev_joined today
//   5  record (2) + wait (4) together                                          — the step boundary
//   6  hipStreamWriteValue32(S, flag, n)                                       — a memory flag instead of an event record
//   7  hipStreamWaitValue32(S, flag >= n) on a flag T wrote long ago           — a memory flag instead of an event wait
//   8  record (1) on S, T waits and runs a small kernel, S does NOT wait       — does the consumer on T change the cost for S?
//   9  nothing on S; A itself ends by bumping a device counter, a one-thread kernel on T spins for it (the kernel-level hand-off)
// Build: hipcc --offload-arch=gfx950 -O3 -o sync_gap_probe sync_gap_probe.hip ; run: ./sync_gap_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// streams 64 MB through the chip (about 25 us); stamps[0] = min start, stamps[1] = max end (ticks of 10 ns)
__global__ void __launch_bounds__(256) k_big(const float4* __restrict__ in, float4* __restrict__ out, size_t n, unsigned long long* stamps, unsigned int* counter)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = in[i];
        v.x += 1.0f;
        out[i] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(stamps, t0);
        atomicMax(stamps + 1, __builtin_amdgcn_s_memrealtime());
        if (counter) {
            __threadfence();
            atomicAdd(counter, 1u);
        }
    }
}
__global__ void k_small(float* p) { p[threadIdx.x] += 1.0f; }
__global__ void k_spin_for(const unsigned int* counter, unsigned int target, float* p)
{
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(4);
    p[0] += 1.0f;
}

int main()
{
    const size_t n = (size_t)64 << 20 >> 4;   // float4 elements of 64 MB
    float4 *a, *b, *c;
    CK(hipMalloc(&a, n * 16));
    CK(hipMalloc(&b, n * 16));
    CK(hipMalloc(&c, n * 16));
    CK(hipMemset(a, 0, n * 16));
    unsigned long long* stamps;   // DEVICE memory (2048 workgroups' atomics into mapped host memory took tens of microseconds themselves)
    CK(hipMalloc(&stamps, 6 * sizeof(unsigned long long)));
    unsigned int *flag, *counter;
    CK(hipMalloc(&flag, 64));
    CK(hipMemset(flag, 0, 64));
    CK(hipMalloc(&counter, 64));
    float* scratch;
    CK(hipMalloc(&scratch, 4096));
    CK(hipMemset(scratch, 0, 4096));
    hipStream_t S, T;
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&T, hipStreamNonBlocking, hi));
    hipEvent_t e_sys, e_dev, t_sys, t_dev;
    CK(hipEventCreateWithFlags(&e_sys, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e_dev, hipEventDisableTiming | hipEventDisableSystemFence));
    CK(hipEventCreateWithFlags(&t_sys, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&t_dev, hipEventDisableTiming | hipEventDisableSystemFence));
    const char* names[] = { "nothing", "record, system-scope event", "record, device-scope event", "wait, system-scope event (complete)", "wait, device-scope event (complete)",
                            "record + wait, device scope", "hipStreamWriteValue32", "hipStreamWaitValue32 (satisfied)", "record (system) + consumer on T, S does not wait",
                            "kernel-level hand-off: A bumps a counter, T spins" };
    const int grid = 256 * 8;
    for (int form = 0; form < 10; form++) {
        std::vector<double> gaps;
        bool supported = true;
        for (int rep = 0; rep < 24 && supported; rep++) {
            unsigned long long *sa = stamps, *sb = stamps + 2;
            const unsigned long long init[6] = { ~0ull, 0, ~0ull, 0, ~0ull, 0 };
            CK(hipMemcpy(stamps, init, sizeof init, hipMemcpyHostToDevice));
            CK(hipMemsetAsync(counter, 0, 4, S));
            // T did something long ago and recorded its events / wrote its flag
            k_small<<<1, 64, 0, T>>>(scratch);
            CK(hipEventRecord(t_sys, T));
            CK(hipEventRecord(t_dev, T));
            if (form == 7 && hipStreamWriteValue32(T, flag, (unsigned)(rep + 1), 0) != hipSuccess) supported = false;
            CK(hipStreamSynchronize(T));
            CK(hipStreamSynchronize(S));
            for (int k = 0; k < 8; k++) k_big<<<grid, 256, 0, S>>>(a, c, n, stamps + 4, nullptr);   // ~200 us of work in front: A, the primitive and B are all enqueued long before A runs
            k_big<<<grid, 256, 0, S>>>(a, b, n, sa, form == 9 ? counter : nullptr);
            switch (form) {
            case 1: CK(hipEventRecord(e_sys, S)); break;
            case 2: CK(hipEventRecord(e_dev, S)); break;
            case 3: CK(hipStreamWaitEvent(S, t_sys, 0)); break;
            case 4: CK(hipStreamWaitEvent(S, t_dev, 0)); break;
            case 5: CK(hipEventRecord(e_dev, S)); CK(hipStreamWaitEvent(S, t_dev, 0)); break;
            case 6: if (hipStreamWriteValue32(S, flag + 4, (unsigned)(rep + 1), 0) != hipSuccess) supported = false; break;
            case 7: if (hipStreamWaitValue32(S, flag, (unsigned)(rep + 1), hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) supported = false; break;
            case 8: CK(hipEventRecord(e_sys, S)); CK(hipStreamWaitEvent(T, e_sys, 0)); k_small<<<1, 64, 0, T>>>(scratch); break;
            case 9: k_spin_for<<<1, 1, 0, T>>>(counter, (unsigned)grid, scratch); break;
            default: break;
            }
            k_big<<<grid, 256, 0, S>>>(b, c, n, sb, nullptr);
            CK(hipStreamSynchronize(S));
            CK(hipStreamSynchronize(T));
            unsigned long long got[6];
            CK(hipMemcpy(got, stamps, sizeof got, hipMemcpyDeviceToHost));
            if (rep >= 4) gaps.push_back(((double)got[2] - (double)got[1]) * 0.01);
        }
        if (!supported) {
            (void)hipGetLastError();
            printf("form %d  %-56s not supported by this runtime\n", form, names[form]);
            continue;
        }
        std::sort(gaps.begin(), gaps.end());
        printf("form %d  %-56s gap between A's last workgroup and B's first: median %6.2f us  (min %6.2f, max %6.2f)\n", form, names[form], gaps[gaps.size() / 2], gaps.front(), gaps.back());
    }
    return 0;
}

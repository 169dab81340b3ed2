// clock_probe.hip — measurement tool, not product code (tools/first_steps.py): a one-wave kernel that times a fixed dependent VALU chain
// with both of the chip's counters: s_memtime (ticks at the SHADER clock, MI355X_MICROARCH.md "s_memtime tick = shader cycle") and
// s_memrealtime (a constant 100 MHz).  ticks / real time = the effective shader clock during those few microseconds, seen from inside the
// stream — something rocm-smi's once-a-second samples cannot resolve.  Launched between the steps of a run on the solver's own stream.
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/micro/clock_probe.hip -o tools/micro/libclock_probe.so
#include <hip/hip_runtime.h>

__global__ void __launch_bounds__(64) k_clock_probe(unsigned long long* out, int slot, int chain)
{
    if (threadIdx.x != 0) return;
    float x = 1.0f + (float)slot * 1e-9f, y = 0.999999f;
    const unsigned long long r0 = wall_clock64();
    const unsigned long long t0 = clock64();
    for (int i = 0; i < chain; i++) {
        asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    }
    const unsigned long long t1 = clock64();
    const unsigned long long r1 = wall_clock64();
    out[4 * slot + 0] = t1 - t0;   // shader-clock ticks
    out[4 * slot + 1] = r1 - r0;   // 100 MHz ticks (10 ns)
    out[4 * slot + 2] = r0;        // when (100 MHz time base): orders the samples against each other
    out[4 * slot + 3] = (unsigned long long)__float_as_uint(x);
}

extern "C" int clock_probe_launch(void* stream, unsigned long long* out, int slot, int chain)
{
    k_clock_probe<<<1, 64, 0, (hipStream_t)stream>>>(out, slot, chain);
    return (int)hipGetLastError();
}

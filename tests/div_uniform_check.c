/* TEST INFRASTRUCTURE — model check of csrc/fluid_math.h div_uniform: (float)((double)x * (1.0 / (double)d)) against x / d, bit for bit.
 * Host arithmetic is IEEE (SSE2): the same three operations the device executes (v_cvt_f64_f32, v_mul_f64, v_cvt_f32_f64, round to nearest
 * even, denormals kept).  Run by tests/test_div_uniform.py. */
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
static inline float udiv(float x, double r) { return (float)((double)x * r); }
static inline uint32_t bits(float f){ uint32_t u; memcpy(&u,&f,4); return u; }
static inline float fromb(uint32_t u){ float f; memcpy(&f,&u,4); return f; }
static uint64_t s = 88172645463325252ull;
static inline uint64_t rnd(void){ s ^= s<<13; s ^= s>>7; s ^= s<<17; return s; }
int main(void){
    long bad = 0, n = 0;
    /* coordinates: (i + .5) / W, all W <= 4096 and a few large */
    for (int W = 1; W <= 16384; W++) {
        if (W > 2200 && W != 4096 && W != 8192 && W != 16384 && W != 3000 && W != 5000 && W != 10000 && W != 12345 && W != 16383) continue;
        double r = 1.0 / (double)(float)W;
        for (int i = 0; i < W; i++) { float x = (float)i + 0.5f; float a = x / (float)W, b = udiv(x, r); n++; if (bits(a) != bits(b)) { if (bad < 5) printf("coord W=%d i=%d\n", W, i); bad++; } }
    }
    printf("coordinates: %ld checked, %ld differ\n", n, bad);
    /* decays d in [1,2): random d (many with few mantissa bits), random x of every class */
    long bad2 = 0, n2 = 0;
    for (int k = 0; k < 4000; k++) {
        uint32_t m = (uint32_t)(rnd() & 0x7FFFFF);
        if (k % 4 == 1) m &= 0x7FF000; if (k % 4 == 2) m &= 0x7C0000; if (k % 16 == 3) m = 0x7FFFFF; if (k % 16 == 7) m = 0; if (k%16==11) m = 1;
        float d = fromb(0x3F800000u | m);
        if (k < 64) d = 1.0f + (float)k * 0.25f * 0.016666f;
        double r = 1.0 / (double)d;
        for (int j = 0; j < 60000; j++) {
            uint32_t xb;
            int c = j % 8;
            uint64_t q = rnd();
            if (c == 0) xb = (uint32_t)q;                                     /* anything */
            else if (c == 1) xb = (uint32_t)(q & 0x807FFFFF);                 /* subnormal inputs */
            else if (c == 2) xb = (uint32_t)(q & 0x80FFFFFF) | 0x00800000u;   /* tiny normals -> subnormal results */
            else if (c == 3) xb = (uint32_t)(q & 0x81FFFFFF);                 /* around the subnormal boundary */
            else if (c == 4) xb = ((uint32_t)q & 0x807FFFFF) | ((uint32_t)(100 + (q >> 40) % 60) << 23);  /* ordinary magnitudes */
            else if (c == 5) xb = ((uint32_t)q & 0x80000FFF) | ((uint32_t)((q >> 40) % 255) << 23);      /* few mantissa bits */
            else if (c == 6) xb = ((uint32_t)q | 0x007FF000u);                /* mantissa nearly all ones */
            else xb = (j & 8) ? 0x80000000u : ((j & 16) ? 0x7F800000u : 0u);  /* -0, +inf, +0 */
            float x = fromb(xb);
            float a = x / d, b = udiv(x, r);
            n2++;
            if (bits(a) != bits(b) && !(a != a && b != b)) { if (bad2 < 5) printf("decay d=%a x=%a  %a vs %a\n", d, x, a, b); bad2++; }
        }
    }
    printf("decays: %ld checked, %ld differ\n", n2, bad2);
    return (bad || bad2) ? 1 : 0;
}

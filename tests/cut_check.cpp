// Host check of the geometry of a pressure launch cut around an exchange in flight (csrc/fluid_cut.h, used by fluid_solver.cpp pass_jacobi):
// for random tiles, bands, aprons, margins and guards
//   1. the interior and the four frame rectangles cover the launch's band once and only once;
//   2. an interior texel's inputs — the tile's whole apron around it — lie `margin` inside the owned rectangle on every side with a
//      neighbour (nothing of the ghost zone, nothing of the strips the curl / vorticity / divergence pass has not written yet);
//   3. the second cut launch reads only what the first one's INTERIOR wrote (one apron inside it), stays clear of the pressure rows /
//      columns the exchange is sending (it writes into the buffer they are read from), and leaves the first launch's FRAME its inputs:
//      the frame of launch 1 reads up to one apron inside interior 1, which interior 2 — written into the same buffer — does not touch.
// Built and run by tests/test_cut.py (g++, no GPU).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fluid_cut.h"

using namespace fluid;

static long fails = 0, cases = 0;
static unsigned long long seed = 88172645463325252ull;
static int rnd(int lo, int hi)   // inclusive
{
    seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17;
    return lo + (int)(seed % (unsigned long long)(hi - lo + 1));
}

#define CHECK(cond, ...) do { if (!(cond)) { if (fails < 20) { printf(__VA_ARGS__); printf("\n"); } fails++; } } while (0)

int main()
{
    for (int it = 0; it < 200000; it++) {
        cases++;
        const int ar = rnd(1, 25), ac = 4 * rnd(1, 7);                  // apron rows / columns of the tile shape
        const int margin = rnd(0, 1) ? 3 : 0;
        const int halo = rnd(4, 64);
        const bool below = rnd(0, 1), above = rnd(0, 1), left = rnd(0, 1), right = rnd(0, 1);
        const int r0 = below ? rnd(halo, 300) : 0, rows = rnd(8, 600), r1 = r0 + rows;
        const int c0 = left ? 4 * rnd(halo / 4 + 1, 80) : 0, cols = 4 * rnd(2, 150), c1 = c0 + cols;
        // the band: owned rows / columns + ext ghost texels on the sides that have a neighbour
        const int ext = rnd(0, halo);
        const int ga = below ? r0 - ext : 0, gb = above ? r1 + ext : r1;
        const int ex4 = (ext + 3) & ~3;
        const int x0 = left ? c0 - std::min(ex4, c0) : 0, x1 = right ? c1 + ex4 : c1;
        const int guard_r = rnd(0, halo), guard_c = rnd(0, halo);
        BlockCut q[3];
        int dep[3], depx[3];
        for (int level = 1; level <= 2; level++) {
            cut_depths(level, ar, ac, margin, guard_r, guard_c, dep[level], depx[level]);
            q[level] = block_cut(ga, gb, x0, x1, r0, r1, c0, c1, below, above, left, right, dep[level], depx[level]);
            const BlockCut& k = q[level];
            CHECK(ga <= k.ia && k.ia <= k.ib && k.ib <= gb && x0 <= k.ja && k.ja <= k.jb && k.jb <= x1, "interior outside the band (level %d)", level);
            CHECK((k.ja & 3) == 0 || k.ja == x0, "interior column %d not a float4 group", k.ja);
            // 1. cover once and only once
            CutRect fr[4];
            cut_frame(ga, gb, x0, x1, k, fr);
            long area = (long)std::max(k.ib - k.ia, 0) * std::max(k.jb - k.ja, 0);
            for (const CutRect& r : fr) area += (long)std::max(r.gb - r.ga, 0) * std::max(r.xb - r.xa, 0);
            CHECK(area == (long)(gb - ga) * (x1 - x0), "interior + frame area %ld != band %ld", area, (long)(gb - ga) * (x1 - x0));
            for (int a = 0; a < 4; a++) {
                const CutRect& r = fr[a];
                if (r.gb <= r.ga || r.xb <= r.xa) continue;
                const bool hits_interior = r.ga < k.ib && k.ia < r.gb && r.xa < k.jb && k.ja < r.xb;
                CHECK(!hits_interior, "frame rectangle %d overlaps the interior", a);
                for (int b = a + 1; b < 4; b++) {
                    const CutRect& s = fr[b];
                    if (s.gb <= s.ga || s.xb <= s.xa) continue;
                    CHECK(!(r.ga < s.gb && s.ga < r.gb && r.xa < s.xb && s.xa < r.xb), "frame rectangles %d and %d overlap", a, b);
                }
            }
            // 2. the inputs of a non-empty interior
            if (k.ib > k.ia && k.jb > k.ja) {
                if (below) CHECK(k.ia - level * ar >= r0 + margin, "level %d reads row %d below r0 + margin %d", level, k.ia - level * ar, r0 + margin);
                if (above) CHECK(k.ib + level * ar <= r1 - margin, "level %d reads past r1 - margin", level);
                if (left) CHECK(k.ja - level * ac >= c0 + margin, "level %d reads column %d left of c0 + margin %d", level, k.ja - level * ac, c0 + margin);
                if (right) CHECK(k.jb + level * ac <= c1 - margin, "level %d reads past c1 - margin", level);
            }
        }
        // 3. launch 2 against launch 1
        const BlockCut &a = q[1], &b = q[2];
        if (b.ib > b.ia && b.jb > b.ja) {
            if (below) CHECK(b.ia - ar >= a.ia && b.ia >= r0 + guard_r && b.ia >= a.ia + ar, "level 2 rows: reads below interior 1, into the rows in flight, or into frame 1's inputs");
            if (above) CHECK(b.ib + ar <= a.ib && b.ib <= r1 - guard_r, "level 2 rows (top)");
            if (left) CHECK(b.ja - ac >= a.ja && b.ja >= c0 + guard_c, "level 2 columns: %d against interior 1 from %d, guard %d", b.ja, a.ja, c0 + guard_c);
            if (right) CHECK(b.jb + ac <= a.jb && b.jb <= c1 - guard_c, "level 2 columns (right)");
        }
    }
    if (fails) {
        printf("FAILED: %ld of %ld cases\n", fails, cases);
        return 1;
    }
    printf("ok: %ld cases\n", cases);
    return 0;
}

// Host check of the tiling arithmetic the register-tile kernels share (csrc/fluid_tiles.h: make_axis, tile_exact, tile_of_block), on the real
// header: for every (domain, output range, tile span, apron) a kernel uses — and a few thousand random ones — the tiles' exact ranges must
// cover the output range once and only once, each exact range must keep its apron from the tile's rim (except where the rim is the domain's),
// and the XCD-aware block order must be a bijection.  Built and run by tests/test_tile_cover.py (hipcc, no GPU involved).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fluid_tiles.h"

using namespace fluid;

static long fails = 0, cases = 0;

static void check_axis(int dom, int lo, int hi, int T, int A)
{
    cases++;
    const Axis ax = make_axis(lo, hi, dom, T, A);
    std::vector<int> cover(hi - lo, 0);
    if (ax.V != T - 2 * A || ax.S < 0 || ax.n < 1) {
        printf("axis dom %d [%d, %d) T %d A %d: S %d V %d n %d\n", dom, lo, hi, T, A, ax.S, ax.V, ax.n);
        fails++;
        return;
    }
    for (int b = 0; b < ax.n; b++) {
        const int t0 = ax.S + b * ax.V;
        int a, e;
        tile_exact(t0, T, A, dom, lo, hi, a, e);
        for (int i = a; i < e; i++) {
            const bool apron_ok = (i - t0 >= A || t0 <= 0) && (t0 + T - 1 - i >= A || t0 + T >= dom) && i >= t0 && i < t0 + T;
            if (!apron_ok || i < lo || i >= hi) {
                printf("axis dom %d [%d, %d) T %d A %d: tile %d at %d claims %d\n", dom, lo, hi, T, A, b, t0, i);
                fails++;
                return;
            }
            cover[i - lo]++;
        }
    }
    for (int i = lo; i < hi; i++)
        if (cover[i - lo] != 1) {
            printf("axis dom %d [%d, %d) T %d A %d (S %d V %d n %d): position %d stored %d times\n", dom, lo, hi, T, A, ax.S, ax.V, ax.n, i, cover[i - lo]);
            fails++;
            return;
        }
}

static void check_order(int nx, int ny, int remap)
{
    cases++;
    std::vector<int> seen(nx * ny, 0);
    for (int b = 0; b < nx * ny; b++) {
        int bx = -1, by = -1;
        tile_of_block(b, nx, ny, remap, bx, by);
        if (bx < 0 || bx >= nx || by < 0 || by >= ny || seen[by * nx + bx]++) {
            printf("order %d x %d remap %d: block %d -> (%d, %d)\n", nx, ny, remap, b, bx, by);
            fails++;
            return;
        }
    }
}

// the band-cyclic order of the chained Jacobi launch: chain_slots() workgroup slots per block map ONTO the nx x ny tiles (every tile once, the
// other slots none), consecutive slots alternate XCDs (slot b belongs to XCD b % 8, whose tiles are the rows of its own bands only), and every
// tile of band group g comes before every tile of group g + 1 ON ITS XCD — so that what a tile of the next block needs (its own band's rows and
// the neighbouring bands' of the same group, or the first / last row of the next / previous group) was taken earlier in the previous block
// ... and, with panels of pw tile columns (round 6), every tile of a group's panel p in front of every tile of its panel p + 1 ON ITS XCD (the
// tiles resident together are a band x pw rectangle)
static void check_chain_order(int nx, int ny, int band, int pw, int rot = 0)
{
    cases++;
    const int slots = chain_slots(nx, ny, band);
    std::vector<int> seen((size_t)nx * ny, 0), last_group(8, -1), last_panel(8, -1);
    if (chain_panels(nx, pw) * pw < nx || (chain_panels(nx, pw) - 1) * pw >= nx) { printf("chain nx %d pw %d: %d panels\n", nx, pw, chain_panels(nx, pw)); fails++; return; }
    for (int b = 0; b < slots; b++) {
        int bx = -1, by = -1;
        if (!chain_tile_of_block(b, nx, ny, band, pw, bx, by, rot)) {
            if (by < ny) { printf("chain nx %d ny %d band %d pw %d: slot %d refused with row %d\n", nx, ny, band, pw, b, by); fails++; return; }
            if (bx < 0 || bx >= nx) { printf("chain nx %d ny %d band %d pw %d: slot %d (no tile) -> column %d\n", nx, ny, band, pw, b, bx); fails++; return; }
            continue;
        }
        if (bx < 0 || bx >= nx || by < 0 || by >= ny) { printf("chain nx %d ny %d band %d pw %d: slot %d -> (%d, %d)\n", nx, ny, band, pw, b, bx, by); fails++; return; }
        seen[(size_t)by * nx + bx]++;
        const int bandno = by / band, xcd = bandno & 7, group = bandno >> 3, panel = bx / pw;
        // rot: the launch hands slot-XCD k the bands (k + rot) % 8 (it passes block * 5)
        if (xcd != (((b & 7) + rot) & 7)) { printf("chain nx %d ny %d band %d pw %d: slot %d (XCD %d) got a tile of XCD %d's band\n", nx, ny, band, pw, b, b & 7, xcd); fails++; return; }
        if (group < last_group[xcd]) { printf("chain nx %d ny %d band %d pw %d: XCD %d goes back from group %d to %d\n", nx, ny, band, pw, xcd, last_group[xcd], group); fails++; return; }
        if (group > last_group[xcd]) last_panel[xcd] = -1;
        if (panel < last_panel[xcd]) { printf("chain nx %d ny %d band %d pw %d: XCD %d goes back from panel %d to %d\n", nx, ny, band, pw, xcd, last_panel[xcd], panel); fails++; return; }
        last_group[xcd] = group;
        last_panel[xcd] = panel;
    }
    for (int t = 0; t < nx * ny; t++)
        if (seen[t] != 1) { printf("chain nx %d ny %d band %d pw %d: tile %d taken %d times\n", nx, ny, band, pw, t, seen[t]); fails++; return; }
}

int main()
{
    // (tile span, apron) of every kernel: Jacobi columns 256 / 12 and 128 / 12 (two texels per lane), rows NW * RY with apron 10 / 11 (K6 folded) and
    // the deep shapes; curl-vorticity-divergence 256 / 4 columns, 40 / 24 rows apron 3; k_advect_cvd 64 / 3, 64 / 4 columns, 32 / 64 / 128 rows apron 3
    const int shapes[][2] = { { 256, 12 }, { 128, 12 }, { 80, 10 }, { 80, 11 }, { 56, 10 }, { 48, 10 }, { 48, 11 }, { 40, 10 }, { 40, 11 }, { 32, 10 }, { 32, 11 },
                              { 64, 8 }, { 96, 10 }, { 96, 13 }, { 56, 17 }, { 64, 25 }, { 256, 20 }, { 256, 28 }, { 256, 4 }, { 40, 3 }, { 24, 3 },
                              { 64, 3 }, { 64, 4 }, { 32, 3 }, { 128, 3 } };
    const int doms[] = { 1, 2, 3, 5, 31, 40, 57, 58, 63, 64, 65, 100, 127, 128, 129, 250, 256, 257, 300, 455, 512, 700, 1001, 1024, 2048, 4096, 8192, 16384 };
    for (const auto& sh : shapes)
        for (int dom : doms) {
            check_axis(dom, 0, dom, sh[0], sh[1]);                                  // the whole domain
            if (dom >= 8) {
                check_axis(dom, dom / 3, dom - dom / 5, sh[0], sh[1]);              // a stripe's band
                check_axis(dom, 0, dom / 2 + 1, sh[0], sh[1]);
                check_axis(dom, dom / 2 - 1, dom, sh[0], sh[1]);
                check_axis(dom, dom / 2, dom / 2 + 1, sh[0], sh[1]);                // one row / column
            }
        }
    srand(20260923);
    for (int k = 0; k < 20000; k++) {
        const auto& sh = shapes[rand() % (sizeof shapes / sizeof shapes[0])];
        const int dom = 1 + rand() % 5000, lo = rand() % dom, hi = lo + 1 + rand() % (dom - lo);
        check_axis(dom, lo, hi, sh[0], sh[1]);
    }
    for (int band = 1; band <= 5; band++)
        for (int nx = 1; nx <= 40; nx += (nx < 24 ? 1 : 5))
            for (int ny : { 1, 2, 3, 7, 8, 9, 23, 24, 25, 35, 69, 70, 71, 137, 274, 511, 512 }) {
                check_chain_order(nx, ny, band, nx);   // one panel: round 5's order
                for (int pw : { 1, 2, 3, 7, 13, 18, 21 })
                    if (pw < nx) check_chain_order(nx, ny, band, pw);
                check_chain_order(nx, ny, band, chain_panel_width(nx, 21));
            }
    for (int rot = 1; rot < 8; rot++)   // every rotation is a bijection of the same kind (the XCD of a band moves, nothing else)
        for (int nx : { 9, 18, 36, 71 })
            for (int ny : { 18, 35, 69, 137 }) check_chain_order(nx, ny, std::max(1, 64 / chain_panel_width(nx, 21)), chain_panel_width(nx, 21), rot);
    for (int nx : { 36, 54, 71, 72, 142 })   // 8192-, 12288-, 16384-wide and a 32768-wide row: what the launch picks
        for (int ny : { 35, 36, 137, 274 }) check_chain_order(nx, ny, std::max(1, 64 / chain_panel_width(nx, 21)), chain_panel_width(nx, 21));
    for (int remap = 0; remap < 4; remap++)
        for (int nx = 1; nx <= 40; nx++)
            for (int ny = 1; ny <= 40; ny += (ny < 12 ? 1 : 7)) check_order(nx, ny, remap);
    printf("%s: %ld cases, %ld failed\n", fails ? "FAILED" : "ok", cases, fails);
    return fails ? 1 : 0;
}

// Host check of the persistent chained Jacobi launch's bookkeeping (csrc/fluid_pchain.h, the real header; k_jacobi_pchain in fluid_kernels.hip):
//   1. tickets -> items: over the eight heads every (block, stack row, tile column) comes up exactly once, holes only beyond the grid, every
//      head's sequence in non-decreasing block order, and pchain_cap() is exactly where each sequence ends;
//   2. dependencies: the <= 3 x 3 items an item reads / overwrites lie in EARLIER bands, and the cells and bands the kernel's poll lanes look
//      at are exactly theirs;
//   3. stacks: the tiles of the stacks of a row range store every row exactly once, every stored row keeps its apron from the tile's rim except
//      where the rim is the domain's or the row below the tile came from the tile before, and a tile that hands a row on really holds it;
//   4. the protocol itself, simulated: W workgroups on arbitrary "XCDs" (all on one, a random spread, fewer workgroups than heads) stepped in
//      random order through draw / poll / shelve-and-help / spin / run / count — every item runs exactly once, AFTER everything it depends on,
//      nobody ever spins on an item that nobody holds, and the run ends.  That is the argument of fluid_pchain.h's header, executed.
// Built and run by tests/test_pchain_cpu.py (hipcc for the HIP headers; no GPU involved).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "fluid_pchain.h"

using namespace fluid;

static long fails = 0, cases = 0;
#define FAIL(...) do { printf(__VA_ARGS__); printf("\n"); fails++; return; } while (0)

static PChainDims dims(int blocks, int nx, int ny, int pw, int bh, int stack = 2)
{
    PChainDims C{};
    C.blocks = blocks;
    C.nx = nx;
    C.ny = ny;
    C.stack = stack;
    C.pw = pw;
    C.bh = bh;
    pchain_finish(C);
    C.withhold = -1;
    return C;
}

static void check_order(const PChainDims& C)
{
    cases++;
    std::vector<int> seen((size_t)C.blocks * C.nx * C.ny, 0);
    for (int x = 0; x < 8; x++) {
        const int cap = pchain_cap(C, x);
        int last_l = 0, last_q = -1;
        for (int t = 0; t < cap + 3 * C.slots; t++) {
            int l, by, bx, q;
            const bool real = pchain_item(C, x, t, l, by, bx, q);
            if (t >= cap) {
                if (q < pchain_total_bands(C)) FAIL("order: head %d ticket %d beyond cap %d is band %d < %d", x, t, cap, q, pchain_total_bands(C));
                continue;
            }
            if (q >= pchain_total_bands(C) || (q & 7) != x || q < last_q) FAIL("order: head %d ticket %d -> band %d (total %d)", x, t, q, pchain_total_bands(C));
            last_q = q;
            if (!real) continue;
            if (l < last_l || l >= C.blocks || by < 0 || bx < 0) FAIL("order: head %d ticket %d -> block %d after block %d", x, t, l, last_l);
            last_l = l;
            if (pchain_band_of(C, l, by, pchain_panel_of(C, bx)) != q) FAIL("order: band_of (%d, %d, %d) != %d", l, by, bx, q);
            if (seen[((size_t)l * C.ny + by) * C.nx + bx]++) FAIL("order: item (%d, %d, %d) twice", l, by, bx);
        }
    }
    for (size_t i = 0; i < seen.size(); i++)
        if (seen[i] != 1) FAIL("order: blocks %d nx %d ny %d pw %d bh %d: item %zu came up %d times", C.blocks, C.nx, C.ny, C.pw, C.bh, i, seen[i]);
    // cells: one per (block, stack row, panel), inside the half, disjoint from the heads
    std::vector<int> cell(C.bank_words, 0);
    for (int l = 0; l < C.blocks; l++)
        for (int by = 0; by < C.ny; by++)
            for (int pn = 0; pn < C.np; pn++) {
                const int c = pchain_cell(C, l, by, pn);
                if (c < 8 * PCHAIN_HEAD_STRIDE || c >= C.bank_words || cell[c]++) FAIL("cells: (%d, %d, %d) -> %d", l, by, pn, c);
                if (pchain_panel_width(C, pn) < 1 || pchain_panel_width(C, pn) > C.pw) FAIL("panel width %d", pchain_panel_width(C, pn));
            }
}

// the cells / bands the kernel's poll lanes look at for item (l, by, bx): mirrors k_jacobi_pchain's control wave (lanes 0..8 and 16..24)
struct Poll {
    int ncell, cell[9], want[9];
    int nband, band[9];
};
static Poll poll_of(const PChainDims& C, int l, int by, int bx)
{
    Poll P{};
    for (int k = 0; k < 9; k++) {
        const int r = by - 1 + k / 3, c = bx - 1 + k % 3;
        if (r < 0 || r >= C.ny || c < 0 || c >= C.nx) continue;
        const int pn = pchain_panel_of(C, c);
        P.cell[P.ncell] = pchain_cell(C, l - 1, r, pn);
        P.want[P.ncell++] = pchain_panel_width(C, pn);
        P.band[P.nband++] = pchain_band_of(C, l - 1, r, pn);
    }
    return P;
}

static void check_deps(const PChainDims& C)
{
    cases++;
    for (int l = 1; l < C.blocks; l++)
        for (int by = 0; by < C.ny; by++)
            for (int bx = 0; bx < C.nx; bx++) {
                const int q = pchain_band_of(C, l, by, pchain_panel_of(C, bx));
                const Poll P = poll_of(C, l, by, bx);
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        const int r = by + dy, c = bx + dx;
                        if (r < 0 || r >= C.ny || c < 0 || c >= C.nx) continue;
                        const int pn = pchain_panel_of(C, c), qd = pchain_band_of(C, l - 1, r, pn), cd = pchain_cell(C, l - 1, r, pn);
                        if (qd >= q) FAIL("deps: item (%d, %d, %d) band %d needs band %d", l, by, bx, q, qd);
                        bool cell_ok = false, band_ok = false;
                        for (int k = 0; k < P.ncell; k++) cell_ok |= P.cell[k] == cd;
                        for (int k = 0; k < P.nband; k++) band_ok |= P.band[k] == qd;
                        if (!cell_ok || !band_ok) FAIL("deps: item (%d, %d, %d): neighbour (%d, %d) cell %d band %d not polled (%d cells, %d bands)", l, by, bx, r, c, cd, qd, P.ncell, P.nband);
                    }
            }
}

template <int NW, int RY, int HX, int HY>
static void check_stack_rows(int H, int ga, int gb, int M)
{
    cases++;
    using G = JacobiTB<NW, RY, HX, HY>;
    using S = JacobiStack<NW, RY, HX, HY>;
    const Axis ay = make_axis(ga, gb, H, S::span(M), HY);
    std::vector<int> cover(gb - ga, 0);
    for (int by = 0; by < ay.n; by++) {
        const int y0s = ay.S + by * (S::span(M) - 2 * HY);
        int st_lo, st_hi;
        tile_exact(y0s, S::span(M), HY, H, ga, gb, st_lo, st_hi);
        if (st_hi <= st_lo) continue;
        bool ended = false;
        for (int t = 0; t < M; t++) {
            int yt, a, b;
            bool last;
            stack_tile_rows(y0s, t, G::TY, HY, H, st_lo, st_hi, yt, a, b, last);
            for (int i = a; i < b; i++) {
                const bool below_ok = (t > 0) || i - yt >= HY || yt <= 0;             // the row below the tile came from the tile before, or an apron, or the domain edge
                const bool above_ok = yt + G::TY - 1 - i >= HY || yt + G::TY >= H;
                if (!below_ok || !above_ok || i < yt || i >= yt + G::TY || i < ga || i >= gb) FAIL("stack H %d [%d, %d) M %d: stack %d tile %d at %d claims row %d", H, ga, gb, M, by, t, yt, i);
                cover[i - ga]++;
            }
            if (!last) {   // this tile hands its row CARRY_SLOT on: it must hold it (inside the domain, exact at every level: outside the top apron)
                const int cr = yt + S::CARRY_SLOT;
                if (cr >= H || cr + 1 != yt + S::STEP || G::TY - 1 - S::CARRY_SLOT < HY) FAIL("stack H %d M %d: tile %d at %d hands on row %d", H, M, t, yt, cr);
                if (t + 1 >= M) FAIL("stack H %d [%d, %d) M %d: stack %d does not reach its own range [%d, %d)", H, ga, gb, M, by, st_lo, st_hi);
            }
            ended = last;
            if (last) break;   // (as the kernel does)
        }
        if (!ended) FAIL("stack H %d [%d, %d) M %d: stack %d never ends", H, ga, gb, M, by);
    }
    for (int i = ga; i < gb; i++)
        if (cover[i - ga] != 1) FAIL("stack H %d [%d, %d) M %d (S %d n %d): row %d stored %d times", H, ga, gb, M, ay.S, ay.n, i, cover[i - ga]);
}

// ---- 4. the protocol, simulated ----
// A workgroup HOLDS tickets (its shelf: the ticket drawn ahead at the end of the last item, helper tickets, items put aside) and always
// works on the held item of the lowest band — the mirror of k_jacobi_pchain's control wave.
struct SimWG {
    int xcc;
    std::vector<unsigned> shelf;
    unsigned cur = PCHAIN_NONE;
    int phase = 0;   // 0: pick / poll, 2: running `cur` (draws ahead and counts at its next step), 3: finished
};

static void simulate(const PChainDims& C, int nwg, int placement, unsigned seed)
{
    cases++;
    std::mt19937 rng(seed);
    std::vector<unsigned> st(C.bank_words, 0);
    std::vector<int> done((size_t)C.blocks * C.nx * C.ny, 0), held((size_t)C.blocks * C.nx * C.ny, 0);
    auto idx = [&](int l, int by, int bx) { return ((size_t)l * C.ny + by) * C.nx + bx; };
    auto note_held = [&](unsigned e) {
        int l, by, bx, q;
        if (e != PCHAIN_NONE && pchain_item(C, (int)(e >> 28), (int)(e & 0x0fffffffu), l, by, bx, q)) held[idx(l, by, bx)] = 1;
    };
    auto draw_from = [&](int x0) -> unsigned {
        for (int k = 0; k < 8; k++) {
            const int x = (x0 + k) & 7, cap = pchain_cap(C, x);
            if (cap == 0) continue;
            const unsigned t = st[x * PCHAIN_HEAD_STRIDE]++;
            if (t < (unsigned)cap) {
                note_held(((unsigned)x << 28) | t);
                return ((unsigned)x << 28) | t;
            }
        }
        return PCHAIN_NONE;
    };
    auto band_of_ticket = [&](unsigned e) { return (int)(((e & 0x0fffffffu) / (unsigned)C.slots) * 8u + (e >> 28)); };
    std::vector<SimWG> wg(nwg);
    for (int i = 0; i < nwg; i++) {
        wg[i].xcc = placement == 0 ? i % 8 : (placement == 1 ? 3 : (int)(rng() % 8));
        const unsigned e = draw_from(wg[i].xcc);
        if (e != PCHAIN_NONE) wg[i].shelf.push_back(e);
    }
    long steps = 0, items_run = 0;
    const long total = (long)C.blocks * C.nx * C.ny, limit = 400 * total + 100000;
    int live = nwg;
    while (live > 0) {
        if (++steps > limit) FAIL("sim: blocks %d nx %d ny %d pw %d bh %d, %d workgroups, placement %d: no end after %ld steps (%ld of %ld items)", C.blocks, C.nx, C.ny, C.pw, C.bh, nwg, placement, steps, items_run, total);
        SimWG& g = wg[rng() % nwg];
        if (g.phase == 3) continue;
        if (g.phase == 0) {
            if (g.shelf.empty()) {
                const unsigned e = draw_from(g.xcc);
                if (e == PCHAIN_NONE) {
                    g.phase = 3;
                    live--;
                    continue;
                }
                g.shelf.push_back(e);
            }
            size_t pick = 0;
            for (size_t k = 1; k < g.shelf.size(); k++)
                if (band_of_ticket(g.shelf[k]) < band_of_ticket(g.shelf[pick])) pick = k;
            const unsigned cur = g.shelf[pick];
            int l, by, bx, q;
            const bool real = pchain_item(C, (int)(cur >> 28), (int)(cur & 0x0fffffffu), l, by, bx, q);
            bool go = !real || l == 0;
            if (!go) {
                const Poll P = poll_of(C, l, by, bx);
                bool cells = true;
                for (int k = 0; k < P.ncell; k++) cells &= st[P.cell[k]] >= (unsigned)P.want[k];
                if (cells) go = true;
                else {
                    int lag = -1;
                    for (int k = 0; k < P.nband && lag < 0; k++)
                        if (st[(P.band[k] & 7) * PCHAIN_HEAD_STRIDE] < (unsigned)((P.band[k] >> 3) + 1) * (unsigned)C.slots) lag = P.band[k] & 7;
                    if (lag >= 0) {   // hold a ticket of that head too
                        if ((int)g.shelf.size() >= PCHAIN_SHELF) FAIL("sim: shelf overflow");
                        const unsigned e = draw_from(lag);
                        if (e != PCHAIN_NONE) g.shelf.push_back(e);
                        continue;
                    }
                    // spinning: everything waited for must be HELD by somebody (or done)
                    for (int dy = -1; dy <= 1; dy++)
                        for (int dx = -1; dx <= 1; dx++) {
                            const int r = by + dy, c = bx + dx;
                            if (r < 0 || r >= C.ny || c < 0 || c >= C.nx) continue;
                            if (!done[idx(l - 1, r, c)] && !held[idx(l - 1, r, c)]) FAIL("sim: (%d, %d, %d) spins on (%d, %d, %d), which nobody holds", l, by, bx, l - 1, r, c);
                        }
                    continue;
                }
            }
            g.shelf[pick] = g.shelf.back();
            g.shelf.pop_back();
            g.cur = cur;
            if (!real) continue;   // a hole: nothing to run, nothing to count; pick again
            if (l > 0)
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        const int r = by + dy, c = bx + dx;
                        if (r < 0 || r >= C.ny || c < 0 || c >= C.nx) continue;
                        if (!done[idx(l - 1, r, c)]) FAIL("sim: (%d, %d, %d) runs before (%d, %d, %d)", l, by, bx, l - 1, r, c);
                    }
            g.phase = 2;
            continue;
        }
        // phase 2: the stack ran; draw the next ticket ahead (only when nothing else is held), then count
        int l, by, bx, q;
        pchain_item(C, (int)(g.cur >> 28), (int)(g.cur & 0x0fffffffu), l, by, bx, q);
        if (g.shelf.empty()) {
            const unsigned e = draw_from(g.xcc);
            if (e != PCHAIN_NONE) g.shelf.push_back(e);
        }
        if (done[idx(l, by, bx)]++) FAIL("sim: item (%d, %d, %d) ran twice", l, by, bx);
        st[pchain_cell(C, l, by, pchain_panel_of(C, bx))]++;
        items_run++;
        g.phase = 0;
    }
    if (items_run != total) FAIL("sim: %ld of %ld items ran (blocks %d nx %d ny %d pw %d bh %d, %d workgroups, placement %d)", items_run, total, C.blocks, C.nx, C.ny, C.pw, C.bh, nwg, placement);
}

int main()
{
    std::mt19937 rng(20260930);
    for (int k = 0; k < 2000000; k++) {   // the multiply-and-correct division of the ticket decode, against the real one
        const int a = (int)(rng() % (1u << 24)), b = 1 + (int)(rng() % (k & 1 ? 4096 : 70));
        if (pchain_div(a, b, 1.0f / (float)b) != a / b) {
            printf("pchain_div(%d, %d) = %d\n", a, b, pchain_div(a, b, 1.0f / (float)b));
            fails++;
            break;
        }
    }
    cases++;
    // the shapes the library picks and their neighbours: 4096^2 (18 x 32 stacks of two), 3072, 6144, 8192, 16384-wide ranks
    const int shapes[][5] = { { 5, 18, 32, 18, 2 }, { 5, 18, 32, 18, 3 }, { 5, 18, 69, 18, 3 }, { 5, 14, 24, 14, 4 }, { 5, 27, 48, 14, 2 }, { 5, 36, 63, 18, 3 },
                              { 20, 71, 17, 18, 3 }, { 2, 1, 1, 1, 1 }, { 3, 2, 9, 2, 5 }, { 8, 19, 20, 19, 1 }, { 24, 5, 7, 3, 2 }, { 5, 43, 12, 21, 3 } };
    for (auto& s : shapes) {
        const PChainDims C = dims(s[0], s[1], s[2], s[3], s[4]);
        check_order(C);
        check_deps(C);
    }
    for (int k = 0; k < 400; k++) {
        const int nx = 1 + rng() % 40, ny = 1 + rng() % 40, pw = 1 + rng() % std::min(nx, 24), bh = 1 + rng() % 5, blocks = 2 + rng() % 8;
        const PChainDims C = dims(blocks, nx, ny, pw, bh);
        check_order(C);
        check_deps(C);
    }
    for (int M = 1; M <= 4; M++) {
        const int Hs[] = { 1, 7, 59, 60, 61, 80, 81, 130, 131, 140, 141, 150, 151, 300, 1024, 2048, 3072, 4096, 4097 };
        for (int H : Hs) {
            check_stack_rows<8, 10, 12, 10>(H, 0, H, M);
            for (int k = 0; k < 40; k++) {
                const int a = rng() % H, b = a + 1 + rng() % (H - a);
                check_stack_rows<8, 10, 12, 10>(H, a, b, M);
            }
        }
    }
    // the protocol: every placement, from one workgroup to more than the chip holds, three seeds each
    for (auto& s : shapes) {
        if ((long)s[0] * s[1] * s[2] > 9000) continue;
        const PChainDims C = dims(s[0], s[1], s[2], s[3], s[4]);
        for (int placement = 0; placement < 3; placement++)
            for (int nwg : { 1, 3, 8, 64, 512 })
                for (unsigned seed = 1; seed <= 2; seed++) simulate(C, nwg, placement, seed * 7919u + nwg);
    }
    for (int k = 0; k < 60; k++) {
        const int nx = 1 + rng() % 12, ny = 1 + rng() % 12, pw = 1 + rng() % nx, bh = 1 + rng() % 4, blocks = 2 + rng() % 5;
        simulate(dims(blocks, nx, ny, pw, bh), 1 + rng() % 40, (int)(rng() % 3), (unsigned)rng());
    }
    if (fails) {
        printf("FAILED: %ld of %ld cases\n", fails, cases);
        return 1;
    }
    printf("ok: %ld cases\n", cases);
    return 0;
}

// fluid_pchain.h — the pressure loop (pressureShader script.js:868-890, loop 1259-1266) as ONE launch of PERSISTENT workgroups (round 6;
// k_jacobi_pchain, fluid_kernels.hip): what a workgroup works on, in which order it is handed out, and what it has to wait for.
// Everything here is integer arithmetic callable on the host: tests/pchain_check.cpp holds it to its invariants without a GPU.  Internal.
//
// ITEMS.  The loop is `blocks` blocks of <= 10 iterations.  In every block the stored range is tiled by nx columns of 256-column tiles
// (12-column apron, fluid_tiles.h) and ny rows of STACKS: a stack is `stack` (M) register tiles of TY = 80 rows taken one after the other
// by the same workgroup, bottom to top, each starting TY - HY rows above the one before.  The first tile of a stack loses a 10-row apron at
// both ends, like a tile of k_jacobi_tb; every further tile gets the row below its first row from the tile before — that tile's row
// TY - HY - 1, which its top apron has not reached at any level 0 ... 9 — through an LDS line per level, so its OWN first row is exact and
// it only loses the apron at the top: M tiles store 60 + 70 (M - 1) rows for 80 M rows of arithmetic (M = 2: 0.8125, M = 1: 0.75).
// An item = one stack of one block: (l, by, bx).
//
// ORDER.  Items are handed out by tickets, one ticket counter ("head") per XCD.  Within a block the stacks form BANDS of bh stack rows x pw
// tile columns (a band fills an XCD's 64 resident workgroups; wide grids have several column PANELS); the bands of all blocks form ONE
// sequence q = l * nb + (row group * np + panel), and band q belongs to the sequence of XCD q % 8 (ticket t of that head: band
// (t / slots) * 8 + x, slot t % slots inside it, row-major; a slot beyond the grid's last row / column is a hole).  A workgroup draws from
// the head of the XCD it runs on (HW_REG_XCC_ID) — neighbours in flight together then share their aprons in that XCD's L2 — and from the
// other heads once its own is exhausted.
//
// DEPENDENCIES.  An item of block l > 0 reads, and overwrites what was read by, the <= 3 x 3 items around it in block l - 1: it waits for
// the counters of their (stack row, panel) cells — one counter per (block, stack row, panel), bumped once per item, complete at the
// panel's width.  All of those lie in bands EARLIER in the sequence (pchain_check).  What makes the wait safe under ANY placement and
// dispatch order (HIP promises none, MI355X guide "Workgroup dispatch"): before it spins, a workgroup checks that every band it waits for
// has been DRAWN completely (its head is past the band); if one has not, the workgroup shelves its own item and draws from THAT head
// itself.  Heads are drawn in order, so an item that was drawn has all of its sequence's earlier items drawn too; a spinning workgroup
// therefore only ever waits for items that are resident or done, and the lowest unfinished drawn item has nothing left to wait for.
#pragma once
#include "fluid_tiles.h"

namespace fluid {

constexpr int PCHAIN_MAX_BLOCKS = 24;    // blocks of <= 10 iterations in one launch (configs[4]: 200 iterations = 20 blocks)
constexpr int PCHAIN_HEAD_STRIDE = 16;   // words between two heads: one 64-byte line each
constexpr int PCHAIN_MAX_CELLS = 1024;   // (stack row, panel) cells per block
constexpr int PCHAIN_SHELF = 32;         // items a workgroup can have shelved while it helps a head that lags
constexpr unsigned int PCHAIN_NONE = 0xffffffffu;   // "no item" (a ticket word is head << 28 | ticket)

struct PChainDims {
    int blocks;
    int nx, ny, xs, ys;   // tile columns x stack rows, and where the first tile column / stack row starts (global coordinates)
    int stack;            // tiles per stack (M)
    int np, pw;           // column panels; tile columns per panel (the last panel may be narrower)
    int bh, nrg, nb;      // stack rows per band; row groups = ceil(ny / bh); bands per block = nrg * np
    int slots;            // ticket slots per band = pw * bh
    int bank, bank_words; // which half of the state words this call counts in (it zeroes the other half for the next call); words per half
    unsigned int timeout; // 100 MHz ticks a workgroup waits for a dependency before it gives up (sets err, computes on stale data)
    int stagger;          // 100 MHz ticks over which the workgroups' first loads are spread (k_jacobi_pchain: persistent workgroups all start at
                          // once and, with equal items, STAY in phase: every tile on the chip loads, iterates and stores at the same time)
    int withhold;         // lab (FLUID_CHAIN_WITHHOLD): the item of band `withhold`, slot 0 never counts itself — forces the give-up path; -1 = off
    // derived (pchain_finish): reciprocals for the divisions of the decode — a workgroup decodes a ticket between two items, on a SIMD it shares
    // with the other workgroup's tile arithmetic, and a 32-bit integer division is ~35 instructions there, a multiply-and-correct 7 —
    // and how many tickets each head's sequence has
    float r_slots, r_pw, r_nb, r_np, r_bh;
    int cap[8];
};
struct PChainPlan {
    PChainDims d;
    int iters[PCHAIN_MAX_BLOCKS];
    int xa[PCHAIN_MAX_BLOCKS], xb[PCHAIN_MAX_BLOCKS];   // columns and rows block l stores (a stripe / tile rank's blocks recompute fewer ghost
    int ga[PCHAIN_MAX_BLOCKS], gb[PCHAIN_MAX_BLOCKS];   // texels each: the ranges shrink; the tiling is block 0's — the widest — for all of them)
};

// a / b for 0 <= a < 2^24, b >= 1, rb = 1.0f / b: the float product is within one of the quotient; one correction makes it exact
__host__ __device__ __forceinline__ int pchain_div(int a, int b, float rb)
{
    int q = (int)((float)a * rb);
    const int r = a - q * b;
    q += (r >= b) - (r < 0);
    return q;
}

__host__ __device__ __forceinline__ int pchain_total_bands(const PChainDims& C) { return C.blocks * C.nb; }
// tickets in the sequence of XCD x: its bands x, x + 8, ... below the total, `slots` each
__host__ __device__ __forceinline__ int pchain_cap(const PChainDims& C, int x) { return C.cap[x]; }
__host__ __device__ __forceinline__ int pchain_panel_of(const PChainDims& C, int bx) { return pchain_div(bx, C.pw, C.r_pw); }
__host__ __device__ __forceinline__ int pchain_panel_width(const PChainDims& C, int pn) { return min(C.pw, C.nx - pn * C.pw); }
// global band number of the band that holds stack row `by`, panel `pn` of block l
__host__ __device__ __forceinline__ int pchain_band_of(const PChainDims& C, int l, int by, int pn) { return l * C.nb + pchain_div(by, C.bh, C.r_bh) * C.np + pn; }
// ticket t of XCD x's sequence -> (l, by, bx) and its global band; false: a hole (a slot of a ragged band beyond the grid)
__host__ __device__ __forceinline__ bool pchain_item(const PChainDims& C, int x, int t, int& l, int& by, int& bx, int& q)
{
    const int i = pchain_div(t, C.slots, C.r_slots), j = t - i * C.slots;
    q = i * 8 + x;
    l = pchain_div(q, C.nb, C.r_nb);
    const int qb = q - l * C.nb, rg = pchain_div(qb, C.np, C.r_np), pn = qb - rg * C.np, jy = pchain_div(j, C.pw, C.r_pw), jx = j - jy * C.pw;
    by = rg * C.bh + jy;
    bx = pn * C.pw + jx;
    return by < C.ny && bx < C.nx;
}
// words of state per half: the eight heads, then one counter per (block, stack row, panel)
__host__ __device__ __forceinline__ int pchain_cell(const PChainDims& C, int l, int by, int pn) { return 8 * PCHAIN_HEAD_STRIDE + (l * C.ny + by) * C.np + pn; }
// ... then one CLAIM word per item (k_jacobi_pchain's fourth form: an item is run by whoever bumps its claim word from 0 — its owner, or a
// workgroup that waits for it and finds it unclaimed)
__host__ __device__ __forceinline__ int pchain_claim_word(const PChainDims& C, int l, int by, int bx) { return 8 * PCHAIN_HEAD_STRIDE + C.blocks * C.ny * C.np + (l * C.ny + by) * C.nx + bx; }
__host__ __device__ __forceinline__ int pchain_bank_words(const PChainDims& C) { return 8 * PCHAIN_HEAD_STRIDE + C.blocks * C.ny * C.np + C.blocks * C.ny * C.nx; }
constexpr int PCHAIN_MAX_ITEMS = 131072;   // items (claim words) per launch
// blocks, nx, ny, stack, pw, bh set: everything that follows from them
__host__ inline void pchain_finish(PChainDims& C)
{
    C.np = (C.nx + C.pw - 1) / C.pw;
    C.nrg = (C.ny + C.bh - 1) / C.bh;
    C.nb = C.nrg * C.np;
    C.slots = C.pw * C.bh;
    C.r_slots = 1.0f / (float)C.slots;
    C.r_pw = 1.0f / (float)C.pw;
    C.r_nb = 1.0f / (float)C.nb;
    C.r_np = 1.0f / (float)C.np;
    C.r_bh = 1.0f / (float)C.bh;
    const int tb = C.blocks * C.nb;
    for (int x = 0; x < 8; x++) C.cap[x] = x < tb ? ((tb - x + 7) / 8) * C.slots : 0;
    C.bank_words = pchain_bank_words(C);
}

// geometry of a stack of M tiles of JacobiTB<NW, RY, HX, HY>: the span it reads, what it stores, where its tiles start
template <int NW, int RY, int HX, int HY>
struct JacobiStack {
    using G = JacobiTB<NW, RY, HX, HY>;
    static constexpr int STEP = G::TY - HY;             // rows between the first rows of two tiles of a stack
    static constexpr int CARRY_SLOT = G::TY - HY - 1;   // the tile row handed to the next tile (level by level): the row below that tile's first row
    static constexpr int CARRY_WAVE = CARRY_SLOT / RY, CARRY_ROW = CARRY_SLOT % RY;
    __host__ __device__ static constexpr int span(int M) { return G::TY + (M - 1) * STEP; }
    static_assert(CARRY_SLOT >= HY, "the carried row must be outside the tile's bottom apron too (first tile of a stack)");
};

// rows tile t of a stack starting at y0s stores: [a, b) inside the stack's own exact range [st_lo, st_hi); `last`: no further tile of the stack stores anything
__host__ __device__ __forceinline__ void stack_tile_rows(int y0s, int t, int TY, int HY, int H, int st_lo, int st_hi, int& yt, int& a, int& b, bool& last)
{
    yt = y0s + t * (TY - HY);
    a = t == 0 ? (yt <= 0 ? 0 : yt + HY) : yt;
    const bool top = yt + TY >= H;   // the tile holds the domain's top edge: exact up to it
    b = top ? H : yt + TY - HY;
    last = top || b >= st_hi;
    a = max(a, st_lo);
    b = min(b, st_hi);
}

}  // namespace fluid

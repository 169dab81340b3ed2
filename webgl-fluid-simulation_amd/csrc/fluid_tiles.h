// fluid_tiles.h — what the register-tile kernels share (fluid_kernels.hip, fluid_kernels_f16.hip): the tiling of an output
// range by apron-carrying tiles, the XCD-aware order in which workgroups take tiles, and the full-wave DPP lane shifts.
// Internal.
#pragma once
#include "fluid_kernels.h"

#include <cstdlib>

namespace fluid {
namespace {

// ------------------------------------------------------------------------------------------------
// Tiling of an output range [lo, hi) of a domain [0, dom) by tiles of span T that lose an apron A on each
// side — except on a side where the tile contains the domain edge: CLAMP_TO_EDGE makes the edge exact, so
// no apron is needed there (at W = 4096, T = 256, A = 8 exactly 17 column tiles cover the row instead of 18).
// Tile b spans [S + b*V, S + b*V + T) with V = T - 2A and S = max(lo - A, 0).
struct Axis {
    int S, V, n;
};

__host__ __device__ inline int ceil_div_pos(int a, int b) { return a <= 0 ? 0 : (a + b - 1) / b; }

__host__ inline Axis make_axis(int lo, int hi, int dom, int T, int A)
{
    Axis ax;
    ax.V = T - 2 * A;
    ax.S = lo - A > 0 ? lo - A : 0;
    const int n1 = ceil_div_pos(hi - ax.S - T + A, ax.V) + 1;  // last tile's exact range reaches hi
    const int n2 = ceil_div_pos(dom - ax.S - T, ax.V) + 1;     // or the last tile contains the far domain edge
    ax.n = n1 < n2 ? n1 : n2;
    return ax;
}

// XCD-aware tile order (MI355X: 8 XCDs with private 4 MiB L2s; workgroup b is dispatched to XCD b % 8).
// Adjacent tiles share their aprons, so each XCD is handed a CONTIGUOUS run of the tile sequence (remap bit 0): the
// ~64 workgroups resident on an XCD at any time are then neighbours and the shared apron texels hit that XCD's L2
// instead of being fetched once per tile.  Bit 1 picks the sequence: row-major (the tiles in flight together span
// whole rows of the field, i.e. long contiguous address runs — measured 7 % faster for the Jacobi kernel at 4096^2,
// profiles/r01/jacobi_tile_order.txt) or column-major.  Bijective for any tile count; placement only affects
// speed, never results.
__host__ __device__ __forceinline__ void tile_of_block(int b, int nx, int ny, int remap, int& bx, int& by)  // (host: tests/tile_cover_check.cpp)
{
    const int n = nx * ny;
    int t = b;
    if (remap & 1) {
        const int q = n >> 3, r = n & 7, xcd = b & 7, slot = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    if (remap & 2) {  // row-major tile sequence: the tiles in flight together span whole rows of the field
        by = t / nx;
        bx = t - by * nx;
        // ... starting with the TOP tile row: the tiles on the bottom / top domain border run the instantiation with every
        // CLAMP_TO_EDGE select (2.6 x the arithmetic of an interior tile) — taken first, they do not end up as the launch's tail
        by = by == 0 ? ny - 1 : by - 1;
    } else {
        bx = t / ny;
        by = t - bx * ny;
    }
}

// The BAND-CYCLIC order of the chained Jacobi launch (k_jacobi_tb_chain, fluid_kernels.hip): within one block of iterations, consecutive
// workgroups alternate XCDs (b % 8, as the hardware places them); XCD k walks bands of `band` tile rows — band k of every group of eight
// bands, group after group.  chain_slots() workgroups per block; a slot whose band lies beyond the grid has no tile (false).
// PANELS (round 6): a tile row longer than an XCD holds (8192-wide: 36 tiles against 64 workgroup slots) is cut into panels of `pw` tile
// columns, and an XCD walks its band panel after panel — `band` rows x pw columns resident together, the 4096-wide picture (3 x 18) at any
// width; pw >= nx: one panel, the order of round 5.  Every panel but the last is pw wide.
// Host-callable: tests/tile_cover_check.cpp holds it to a bijection onto the nx x ny tiles with every group's tiles in front of the next group's.
// (Bands of NEARLY EQUAL height — every XCD the same number of tile rows where the last group of exact bands is partly empty: 8192 x 2048 has
// 35 tile rows = 12 bands of 3, two per block for XCDs 0 ... 3 and one for the others — were measured too: bitwise, and slower wherever a band
// then holds slots without a tile, 4096^2 included; an XCD that runs out of one band's tiles early takes the next band's beside it, which is
// what four-row bands did.  profiles/r06/chain_loop_map.txt.)
__host__ __device__ __forceinline__ int chain_slots(int nx, int ny, int band) { return 8 * ((ny + 8 * band - 1) / (8 * band)) * band * nx; }
__host__ __device__ __forceinline__ int chain_panels(int nx, int pw) { return (nx + pw - 1) / pw; }
// `rot` (round 6): the band of a group that slot-XCD k walks is band (k + rot) % 8 — the launch passes rot = l * step for block l, so that the
// bands of a partly empty last group (8192 x 2048: 35 tile rows = 12 bands: XCDs 0 ... 3 two bands per block, XCDs 4 ... 7 one) fall to other
// XCDs in the next block and the XCDs' loads even out over the launch; 0 = the same XCD in every block (round 5)
__host__ __device__ __forceinline__ bool chain_tile_of_block(int b, int nx, int ny, int band, int pw, int& bx, int& by, int rot = 0)
{
    const int xcd = ((b & 7) + rot) & 7, i = b >> 3, per_band = band * nx, g = i / per_band, q = i - g * per_band;
    const int per_panel = band * pw, p = q / per_panel, j = q - p * per_panel, wp = min(pw, nx - p * pw);   // (the last panel: what is left of the row)
    const int r = j / wp;
    by = (g * 8 + xcd) * band + r;
    bx = p * pw + (j - r * wp);
    return by < ny;
}
// the panel width the launch picks: the row in equal parts of at most `most` tiles (21: three rows of which fill an XCD's 64 slots)
__host__ inline int chain_panel_width(int nx, int most) { const int np = (nx + most - 1) / most; return (nx + np - 1) / np; }

// exact (storable) global range [a, b) of the tile starting at t0, intersected with [lo, hi)
__host__ __device__ __forceinline__ void tile_exact(int t0, int T, int A, int dom, int lo, int hi, int& a, int& b)
{
    a = t0 <= 0 ? 0 : t0 + A;
    b = t0 + T >= dom ? dom : t0 + T - A;
    a = max(a, lo);
    b = min(b, hi);
}

// geometry of a temporally blocked Jacobi tile (see fluid_kernels.hip)
template <int NW, int RY, int HX, int HY>
struct JacobiTB {
    static constexpr int TX = 256;          // columns per tile (64 lanes x float4)
    static constexpr int TY = NW * RY;      // rows per tile
    static constexpr int VX = TX - 2 * HX;  // HX-column / HY-row apron: up to min(HX, HY) iterations per launch
    static constexpr int VY = TY - 2 * HY;
    static_assert(HX % 4 == 0, "column apron must keep float4 alignment");
    static_assert(HX >= HY, "iterations per launch are bounded by the row apron");
    static_assert(VX > 0 && VY > 0, "tile smaller than its apron");
};

// Undefined `old` operand (mov_dpp, not update_dpp with a zero): the lane without a source gets an unspecified value
// — every caller either overrides it (EDGE) or only feeds the stale apron with it — and the DPP-combine pass is then
// free to fold the shift into the consuming v_add_f32 (v_add_f32_dpp: no extra instruction, no zero-initialisation).
__device__ __forceinline__ float from_left_lane(float v)  // value held by lane-1 (unspecified in lane 0)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right_lane(float v)  // value held by lane+1 (unspecified in lane 63)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}

inline int xcd_remap()  // FLUID_XCD_REMAP: tile order of the Jacobi kernel (A/B knob); bit 0 = XCD-contiguous runs, bit 1 = row-major
{
    static const int v = [] {
        const char* e = lab_env("FLUID_XCD_REMAP");
        return (e ? atoi(e) : 3) & 15;   // bits 2, 3 (lab): wave priorities inside the Jacobi tile kernel (jacobi_tb_tile)
    }();
    return v;
}

}  // namespace
}  // namespace fluid

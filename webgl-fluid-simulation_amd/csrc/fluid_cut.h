// fluid_cut.h — the geometry of a pressure launch cut around an exchange in flight (fluid_solver.cpp pass_jacobi, JacobiSplit): which texels
// of the launch's band compute while the ghost texels travel (the interior) and which follow (the frame).  Plain integer arithmetic, no
// device code: tests/cut_check.cpp holds it to its invariants on the host.  Internal.
#pragma once
#include <algorithm>

namespace fluid {

struct BlockCut {
    int ia, ib, ja, jb;   // the interior: rows [ia, ib) x columns [ja, jb)
};
struct CutRect {
    int xa, xb, ga, gb;   // columns [xa, xb) x rows [ga, gb); empty if xb <= xa or gb <= ga
};

// How far the interior of cut launch `level` (1 = the block's first launch, 2 = its second) stays inside the owned rectangle.  A tile
// loads its whole apron (apron_rows x apron_cols texels) whatever the iteration count, and launch `level` reads what launch level - 1's
// INTERIOR wrote: one apron per level.  margin: texels beyond that (3 behind the interior of the curl / vorticity / divergence pass, whose
// divergence reads velocity 3 texels away; columns go by whole float4 groups).  guard: pressure rows / columns next to the border that
// the exchange in flight is SENDING — the second cut launch writes into the buffer they are read from and stays clear of them.
inline void cut_depths(int level, int apron_rows, int apron_cols, int margin, int guard_rows, int guard_cols, int& dep, int& depx)
{
    dep = level * apron_rows + margin;
    depx = level * apron_cols + ((margin + 3) & ~3);
    if (level > 1) {
        dep = std::max(dep, guard_rows);
        depx = std::max(depx, (guard_cols + 3) & ~3);
    }
}

// The interior of a launch over the band rows [ga, gb) x columns [x0, x1) on a tile that owns rows [r0, r1) x columns [c0, c1) and has a
// neighbour below / above / left / right (a side without one has no ghost texels: the interior reaches the band's edge there).
inline BlockCut block_cut(int ga, int gb, int x0, int x1, int r0, int r1, int c0, int c1, bool below, bool above, bool left, bool right, int dep,
                          int depx)
{
    BlockCut q;
    q.ia = below ? std::min(std::max(r0 + dep, ga), gb) : ga;
    q.ib = above ? std::max(std::min(r1 - dep, gb), q.ia) : gb;
    q.ja = left ? std::min(std::max(c0 + depx, x0), x1) : x0;
    q.jb = right ? std::max(std::min(c1 - depx, x1), q.ja) : x1;
    return q;
}

// The frame around it: the band minus the interior as four rectangles (bottom and top over the full column range, left and right beside
// the interior).  Together with the interior they cover the band once and only once.
inline void cut_frame(int ga, int gb, int x0, int x1, const BlockCut& q, CutRect (&r)[4])
{
    r[0] = CutRect{ x0, x1, ga, q.ia };
    r[1] = CutRect{ x0, x1, q.ib, gb };
    r[2] = CutRect{ x0, q.ja, q.ia, q.ib };
    r[3] = CutRect{ q.jb, x1, q.ia, q.ib };
}

}  // namespace fluid

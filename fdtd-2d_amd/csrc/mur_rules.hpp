// The reference's boundary treatment of Ez (python-src/main.py:29-61) as pure functions.
//
// update_Ez runs four stages in order -- A interior curl update, B left/right 5-px Mur
// bands (rows 1..R-2), C top/bottom bands (columns 1..C-2), D the four 5x5 corner blocks --
// and each stage reads only values of the stage before it.  So the
// final value of every cell is a pure function of P (Ez before the call) and the freshly
// updated Hx, Hy in a small neighbourhood.  MurRules evaluates that function over any
// storage through an accessor, which lets the same code serve the global-memory frame
// kernel and the LDS-resident boundary-zone kernel, and removes every ordering hazard
// between boundary cells (nothing reads a value another thread is writing).
//
// Accessor interface (global row i, global column j):
//   T p(i,j)   Ez before the E half-step          T hx(i,j), hy(i,j)  updated H fields
//   T ce(i,j)  dt/(eps*dx)                        int R, C            global grid size
#pragma once
#include <hip/hip_runtime.h>

namespace fdtd {

template <class T, class Acc> struct MurRules {
    Acc m;
    T k;   // (c*dt - dx)/(c*dt + dx) from the [0,0] material cell, main.py:30-31

    // stage A (main.py:21-27); edge cells are not touched by it
    __device__ __forceinline__ T a(int i, int j) const
    {
        const T e = m.p(i, j);
        if (i < 1 || i > m.R - 2 || j < 1 || j > m.C - 2) return e;
        return e + ((m.hy(i, j) - m.hy(i, j - 1)) - (m.hx(i, j) - m.hx(i - 1, j))) * m.ce(i, j);
    }
    // after the left/right bands (main.py:34-41)
    __device__ __forceinline__ T b(int i, int j) const
    {
        if (i >= 1 && i <= m.R - 2) {
            if (j < 5) return m.p(i, j + 1) + k * (a(i, j + 1) - m.p(i, j));
            if (j >= m.C - 5) return m.p(i, j - 1) + k * (a(i, j - 1) - m.p(i, j));
        }
        return a(i, j);
    }
    // after the top/bottom bands (main.py:44-51)
    __device__ __forceinline__ T c(int i, int j) const
    {
        if (j >= 1 && j <= m.C - 2) {
            if (i < 5) return m.p(i + 1, j) + k * (b(i + 1, j) - m.p(i, j));
            if (i >= m.R - 5) return m.p(i - 1, j) + k * (b(i - 1, j) - m.p(i, j));
        }
        return b(i, j);
    }
    // after the corner rule (main.py:54-61): mean of the inward row- and column-neighbour
    __device__ __forceinline__ T d(int i, int j) const
    {
        const bool top = i < 5, bot = i >= m.R - 5, lef = j < 5, rig = j >= m.C - 5;
        if (top && lef) return (c(i, j + 1) + c(i + 1, j)) / T(2);
        if (top && rig) return (c(i, j - 1) + c(i + 1, j)) / T(2);
        if (bot && lef) return (c(i - 1, j) + c(i, j + 1)) / T(2);
        if (bot && rig) return (c(i - 1, j) + c(i, j - 1)) / T(2);
        return c(i, j);
    }
};

}  // namespace fdtd

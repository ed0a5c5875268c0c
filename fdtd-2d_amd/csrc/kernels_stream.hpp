// Temporally blocked leapfrog for gfx950: NT full time steps per pass over HBM.
//
// One pass reads the state at time n from one buffer set and writes the state at time
// n+NT to the other, so HBM traffic per cell-step drops from 24-32 B to (24-32)/NT B (plus
// halo overlap) and the kernel becomes VALU-bound instead of HBM-bound.  The arithmetic per
// cell and per step is exactly that of the single-step kernels (same operations, same
// order, one rounding each), only the schedule differs, so results stay value-identical.
//
//  k_stream  -- the bulk.  The rows [band_lo, band_hi) are cut into bands, the columns into
//    strips of 64 lanes x V columns (V = 16 B / sizeof(T): one dwordx4 per lane and row).
//    ONE WAVE owns one (band, strip) and streams down its rows with the NT time levels
//    skewed by one row each: at tick tau it loads row tau (level 0) and level t updates row
//    tau - t from level t-1's rows tau-t and tau-t+1, all held in registers.  Row
//    neighbours (i +- 1) are therefore register values of the same lane; column neighbours
//    (j +- 1) come from the adjacent lane by DPP wave shifts (v_mov_b32_dpp wave_shl/shr:1)
//    -- no LDS, no barriers, no inter-wave communication.  Strips overlap by HC columns and
//    bands by NT rows on each side (recomputed redundantly; validity shrinks one cell per
//    level from a non-physical edge).  Strips that touch the left/right grid edge take the
//    EDGE path, which applies the reference's column masks and the left/right Mur band
//    (row-local, python-src/main.py:34-41) in registers.
//
//  k_zone    -- the top and bottom ZO = 5 + NT rows, where the horizontal Mur bands and the
//    corner rule (main.py:44-61) couple rows in ways the row-skewed pipeline cannot follow.
//    A workgroup keeps a (ZO + NT + 1) x (64 + 2 NT + 2) tile of Ez (double-buffered), Hx,
//    Hy in LDS and performs the NT steps there, each cell through the pure boundary
//    function of mur_rules.hpp.  k_stream never comes closer than 5 rows to the grid's top
//    or bottom at any level, so the two kernels together cover the grid exactly once.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_step.hpp"
#include "mur_rules.hpp"

namespace fdtd {

constexpr int STREAM_HC = 8;        // halo columns per strip side (>= NT, multiple of V)
constexpr int STREAM_MAX_NT = 8;
constexpr int ZONE_WZ = 64;         // output columns per zone tile

template <class T> struct PassParams {
    const T *ez_in, *hx_in, *hy_in;
    T *ez_out, *hx_out, *hy_out;
    const T *ce, *ch;          // coefficient arrays (unused when uniform)
    T ce_u, ch_u, k;
    Geom g;
    int band_lo, band_hi;      // rows the streaming kernel produces
    int band_rows, nstrips;
    int zone_top, zone_bot;    // 1 if this launch owns the grid's top / bottom zone
    int src_row, src_col;      // -1: no source
    double amp[STREAM_MAX_NT]; // amplitude added after step s = 1..NT of this pass
};

// ---- lane shifts -----------------------------------------------------------------------
// from_next(x): lane l gets lane l+1's x; from_prev(x): lane l gets lane l-1's x.
// (lane 63 / lane 0 get 0: those are strip-edge lanes whose results are never used)
__device__ __forceinline__ int dpp_next(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x130, 0xF, 0xF, true); }
__device__ __forceinline__ int dpp_prev(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xF, 0xF, true); }
__device__ __forceinline__ float from_next(float x) { return __builtin_bit_cast(float, dpp_next(__builtin_bit_cast(int, x))); }
__device__ __forceinline__ float from_prev(float x) { return __builtin_bit_cast(float, dpp_prev(__builtin_bit_cast(int, x))); }
__device__ __forceinline__ double from_next(double x)
{
    const long long b = __builtin_bit_cast(long long, x);
    const unsigned lo = (unsigned)dpp_next((int)(unsigned)b), hi = (unsigned)dpp_next((int)(b >> 32));
    return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double from_prev(double x)
{
    const long long b = __builtin_bit_cast(long long, x);
    const unsigned lo = (unsigned)dpp_prev((int)(unsigned)b), hi = (unsigned)dpp_prev((int)(b >> 32));
    return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}

template <class T> struct Row3 {
    Vec<T> e, x, y;
};

// ---- the streaming kernel ------------------------------------------------------------------
template <class T, int NT, bool CE_ARR, bool CH_ARR, bool EDGE>
__device__ __forceinline__ void stream_body(const PassParams<T> &p, const int strip, const int ra,
                                            const int rb)
{
    constexpr int V = Vec<T>::N;
    constexpr int SW = 64 * V, OW = SW - 2 * STREAM_HC;
    constexpr int U = 4;   // ticks per unrolled loop body = prefetch distance in rows
    const Geom g = p.g;
    const int lane = threadIdx.x;
    const int j0 = strip * OW - STREAM_HC + V * lane;
    const bool ld_ok = j0 >= 0 && j0 < g.C;
    const bool st_ok = ld_ok && j0 >= strip * OW && j0 < (strip + 1) * OW;
    const size_t col = (size_t)(ld_ok ? j0 : 0);

    // level state: E[t], HX[t], HY[t] = level t at the row it processed in the previous tick
    Vec<T> E[NT], HX[NT + 1], HY[NT];
#pragma unroll
    for (int t = 0; t <= NT; ++t)
#pragma unroll
        for (int v = 0; v < V; ++v) {
            HX[t].v[v] = T(0);
            if (t < NT) E[t].v[v] = HY[t].v[v] = T(0);
        }

    const int tau0 = ra - NT, tau1 = rb + NT;   // level-0 rows [tau0, tau1)

    auto load_row = [&](int i) {
        Row3<T> r;
        if (ld_ok && i < tau1) {
            const size_t o = at(g, i, 0) + col;
            r.e = ldv(p.ez_in + o);
            r.x = ldv(p.hx_in + o);
            r.y = ldv(p.hy_in + o);
        } else {
#pragma unroll
            for (int v = 0; v < V; ++v) r.e.v[v] = r.x.v[v] = r.y.v[v] = T(0);
        }
        return r;
    };

    Row3<T> pre[U];
#pragma unroll
    for (int u = 0; u < U; ++u) pre[u] = load_row(tau0 + u);

    for (int tb = tau0; tb < tau1; tb += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int tau = tb + u;
            if (tau >= tau1) break;
            Vec<T> nE = pre[u].e, nX = pre[u].x, nY = pre[u].y;   // level 0, row tau
            pre[u] = load_row(tau + U);
#pragma unroll
            for (int t = 1; t <= NT; ++t) {
                const int i = tau - t;    // row this level updates now
                const Vec<T> &Po = E[t - 1];    // level t-1, row i   (P of this E half-step)
                Vec<T> cx, cy;
                if (CH_ARR || CE_ARR) {
                    const bool ok = ld_ok && i >= tau0;   // rows above tau0 are pipeline fill
                    const size_t o = ok ? at(g, i, 0) + col : 0;
                    if (CH_ARR) cx = ldv(p.ch + o);
                    if (CE_ARR) cy = ldv(p.ce + o);
                }
                // H half-step of row i (main.py:66-76)
                const T e_next_lane = from_next(Po.v[0]);
                Vec<T> hx, hy, en;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const T ch = CH_ARR ? cx.v[v] : p.ch_u;
                    const T right = (v + 1 < V) ? Po.v[v + 1] : e_next_lane;
                    hx.v[v] = HX[t - 1].v[v] - ch * (nE.v[v] - Po.v[v]);
                    hy.v[v] = HY[t - 1].v[v] + ch * (right - Po.v[v]);
                    if (EDGE) {
                        const int j = j0 + v;
                        const bool upd = j >= 0 && j <= g.C - 2;
                        hx.v[v] = upd ? hx.v[v] : HX[t - 1].v[v];
                        hy.v[v] = upd ? hy.v[v] : HY[t - 1].v[v];
                    }
                }
                // E half-step of row i, stage A (main.py:21-27); rows here are always interior
                const T hy_prev_lane = from_prev(hy.v[V - 1]);
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const T ce = CE_ARR ? cy.v[v] : p.ce_u;
                    const T left = (v > 0) ? hy.v[v - 1] : hy_prev_lane;
                    en.v[v] = Po.v[v] + ((hy.v[v] - left) - (hx.v[v] - HX[t].v[v])) * ce;
                    if (EDGE) {
                        const int j = j0 + v;
                        en.v[v] = (j >= 1 && j <= g.C - 2) ? en.v[v] : Po.v[v];
                    }
                }
                if (EDGE) {
                    // stage B, left/right 5-px Mur band of this row (main.py:34-41)
                    const T a_next = from_next(en.v[0]), a_prev = from_prev(en.v[V - 1]);
                    const T p_prev = from_prev(Po.v[V - 1]);
                    Vec<T> out;
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const int j = j0 + v;
                        const T pr = (v + 1 < V) ? Po.v[v + 1] : e_next_lane;
                        const T ar = (v + 1 < V) ? en.v[v + 1] : a_next;
                        const T pl = (v > 0) ? Po.v[v - 1] : p_prev;
                        const T al = (v > 0) ? en.v[v - 1] : a_prev;
                        const T bl = pr + p.k * (ar - Po.v[v]);
                        const T br = pl + p.k * (al - Po.v[v]);
                        out.v[v] = (j >= 0 && j < 5) ? bl : ((j >= g.C - 5 && j < g.C) ? br : en.v[v]);
                    }
                    en = out;
                }
                if (i == p.src_row) {   // point source after step t of this pass (fdtd.py:34)
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        if (j0 + v == p.src_col) en.v[v] = (T)((double)en.v[v] + p.amp[t - 1]);
                }
                // level t-1's row i+1 becomes its "previous" row; level t's row i flows on
                E[t - 1] = nE;
                HX[t - 1] = nX;
                HY[t - 1] = nY;
                nE = en;
                nX = hx;
                nY = hy;
            }
            HX[NT] = nX;
            const int io = tau - NT;
            if (io >= ra && st_ok) {
                const size_t o = at(g, io, 0) + col;
                stv(p.ez_out + o, nE);
                stv(p.hx_out + o, nX);
                stv(p.hy_out + o, nY);
            }
        }
    }
}

template <class T, int NT, bool CE_ARR, bool CH_ARR>
__global__ __launch_bounds__(64) void k_stream(const PassParams<T> p)
{
    constexpr int V = Vec<T>::N;
    constexpr int SW = 64 * V, OW = SW - 2 * STREAM_HC;
    const int strip = blockIdx.x % p.nstrips, band = blockIdx.x / p.nstrips;
    const int ra = p.band_lo + band * p.band_rows;
    const int rb = min(ra + p.band_rows, p.band_hi);
    if (ra >= rb) return;
    const int x0 = strip * OW - STREAM_HC;
    // all SW columns plain interior columns (5 <= j <= C-6)?  wave-uniform
    if (x0 >= 5 && x0 + SW <= p.g.C - 5)
        stream_body<T, NT, CE_ARR, CH_ARR, false>(p, strip, ra, rb);
    else
        stream_body<T, NT, CE_ARR, CH_ARR, true>(p, strip, ra, rb);
}

// ---- the top/bottom zone kernel ----------------------------------------------------------------
template <int NT> struct ZoneDims {
    static constexpr int ZO = 5 + NT;          // rows written per zone
    static constexpr int ZR = ZO + NT + 1;     // rows held in LDS
    static constexpr int M = NT + 1;           // margin columns per side
    static constexpr int WL = ZONE_WZ + 2 * M; // columns held in LDS
    static constexpr int WLP = WL + 1;         // padded LDS row
};

template <class T, bool CE_ARR> struct TileAcc {
    const T *P, *x, *y;   // LDS tiles, row stride WLP
    const T *cearr;
    T ce_u;
    Geom g;
    int R, C, z0, c0, wlp;
    __device__ __forceinline__ int idx(int i, int j) const { return (i - z0) * wlp + (j - c0); }
    __device__ __forceinline__ T p(int i, int j) const { return P[idx(i, j)]; }
    __device__ __forceinline__ T hx(int i, int j) const { return x[idx(i, j)]; }
    __device__ __forceinline__ T hy(int i, int j) const { return y[idx(i, j)]; }
    __device__ __forceinline__ T ce(int i, int j) const { return CE_ARR ? cearr[at(g, i, j)] : ce_u; }
};

template <class T, int NT, bool CE_ARR, bool CH_ARR>
__global__ __launch_bounds__(256) void k_zone(const PassParams<T> p)
{
    using D = ZoneDims<NT>;
    __shared__ T sE[2][D::ZR * D::WLP];
    __shared__ T sX[D::ZR * D::WLP];
    __shared__ T sY[D::ZR * D::WLP];
    const Geom g = p.g;
    const bool bottom = p.zone_top ? (blockIdx.y == 1) : true;
    // rows held [z0, z1), rows written [o0, o1)
    const int z0 = bottom ? g.R - D::ZR : 0, z1 = z0 + D::ZR;
    const int o0 = bottom ? g.R - D::ZO : 0, o1 = o0 + D::ZO;
    // columns held [c0, c1), columns written [w0, w1)
    // the last tile is shifted left to full width (it then recomputes a few columns of its
    // neighbour, writing identical values) so that the right Mur band never sits next to a
    // non-physical tile edge
    const int w0 = min((int)blockIdx.x * ZONE_WZ, max(0, g.C - ZONE_WZ)), w1 = min(w0 + ZONE_WZ, g.C);
    const int c0 = max(0, w0 - D::M), c1 = min(g.C, w0 + ZONE_WZ + D::M);
    const int wl = c1 - c0;
    const int ncell = D::ZR * wl;

    for (int n = threadIdx.x; n < ncell; n += 256) {
        const int li = n / wl, lj = n - li * wl;
        const size_t o = at(g, z0 + li, c0 + lj);
        const int s = li * D::WLP + lj;
        sE[0][s] = p.ez_in[o];
        sX[s] = p.hx_in[o];
        sY[s] = p.hy_in[o];
    }
    __syncthreads();

    int cur = 0;
#pragma unroll 1
    for (int step = 1; step <= NT; ++step) {
        const T *Eo = sE[cur];
        T *En = sE[cur ^ 1];
        // H half-step (main.py:66-76) on every cell whose i+1 / j+1 neighbours are in the tile
        for (int n = threadIdx.x; n < ncell; n += 256) {
            const int li = n / wl, lj = n - li * wl;
            const int i = z0 + li, j = c0 + lj;
            if (i <= g.R - 2 && j <= g.C - 2 && li + 1 < D::ZR && lj + 1 < wl) {
                const int s = li * D::WLP + lj;
                const T ch = CH_ARR ? p.ch[at(g, i, j)] : p.ch_u;
                const T e = Eo[s];
                sX[s] = sX[s] - ch * (Eo[s + D::WLP] - e);
                sY[s] = sY[s] + ch * (Eo[s + 1] - e);
            }
        }
        __syncthreads();
        // E half-step: stages A-D as one pure function of (Eo, new H) per cell
        MurRules<T, TileAcc<T, CE_ARR>> rules{{Eo, sX, sY, p.ce, p.ce_u, g, g.R, g.C, z0, c0, D::WLP}, p.k};
        for (int n = threadIdx.x; n < ncell; n += 256) {
            const int li = n / wl, lj = n - li * wl;
            const int i = z0 + li, j = c0 + lj;
            const int s = li * D::WLP + lj;
            // needs the row above and the column to the left inside the tile unless the
            // cell sits on the physical edge (where the rules never look outward)
            const bool ok = (li >= 1 || i == 0) && (lj >= 1 || j == 0);
            T val = ok ? rules.d(i, j) : Eo[s];
            if (i == p.src_row && j == p.src_col) val = (T)((double)val + p.amp[step - 1]);
            En[s] = val;
        }
        __syncthreads();
        cur ^= 1;
    }

    const T *Ef = sE[cur];
    const int ow = w1 - w0;
    for (int n = threadIdx.x; n < D::ZO * ow; n += 256) {
        const int r = n / ow, q = n - r * ow;
        const int i = o0 + r, j = w0 + q;
        const int s = (i - z0) * D::WLP + (j - c0);
        const size_t o = at(g, i, j);
        p.ez_out[o] = Ef[s];
        p.hx_out[o] = sX[s];
        p.hy_out[o] = sY[s];
    }
}

}  // namespace fdtd

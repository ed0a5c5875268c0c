// Half-step kernels for boundary = FDTD2D_BOUNDARY_PML: Berenger's split-field PML for the
// TE-mode system (build-defined: the reference has no time-domain PML; parity unpinned, see
// oracle/pml_numpy.py for the definition these kernels follow operation for operation).
// Ez = Ezx + Ezy; the engine stores Ez (total, ping-pong) and Ezx (updated in place).
// Loss factors are eight 1-D arrays (rows: ahr, bhr, aer, ber; columns: ahc, bhc, aec, bec),
// equal to 1 outside the L-cell layer, so HBM traffic per cell grows only by the Ezx field.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_step.hpp"
#include "kernels_stream.hpp"

namespace fdtd {

template <class T> struct PmlFactors {
    const T *ahr, *bhr, *aer, *ber;   // indexed by global row
    const T *ahc, *bhc, *aec, *bec;   // indexed by column
    int L, R, C;                      // layer depth in cells, grid size
    __host__ __device__ bool row_in(int i) const { return i < L || i > R - 1 - L; }
    __host__ __device__ bool col_in(int j) const { return j < L || j > C - 1 - L; }
};

template <class T, bool CH_ARR, int RPT>
__global__ __launch_bounds__(256) void k_update_h_pml(const T *__restrict__ ez, T *__restrict__ hx,
                                                      T *__restrict__ hy, const T *__restrict__ ch,
                                                      T ch_u, PmlFactors<T> f, Geom g, int lo, int hi)
{
    constexpr int V = Vec<T>::N;
    const int j0 = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int i0 = lo + (blockIdx.y * blockDim.y + threadIdx.y) * RPT;
    if (j0 > g.C - 2 || i0 >= hi) return;
    Vec<T> e = ldv(ez + at(g, i0, j0));
    T ac[V], bc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int j = j0 + v < g.C ? j0 + v : g.C - 1;
        ac[v] = f.ahc[j];
        bc[v] = f.bhc[j];
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int i = i0 + r;
        if (i >= hi) break;
        const size_t o = at(g, i, j0);
        const Vec<T> en = ldv(ez + o + g.pitch);
        const T er = (j0 + V < g.C) ? ez[o + V] : T(0);
        Vec<T> x = ldv(hx + o), y = ldv(hy + o), c;
        if (CH_ARR) c = ldv(ch + o);
        const T ar = f.ahr[i], br = f.bhr[i];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (j0 + v <= g.C - 2) {
                const T cc = CH_ARR ? c.v[v] : ch_u;
                const T right = (v + 1 < V) ? e.v[v + 1] : er;
                x.v[v] = ar * x.v[v] - (br * cc) * (en.v[v] - e.v[v]);
                y.v[v] = ac[v] * y.v[v] + (bc[v] * cc) * (right - e.v[v]);
            }
        }
        stv(hx + o, x);
        stv(hy + o, y);
        e = en;
    }
}

template <class T, bool CE_ARR, int RPT>
__global__ __launch_bounds__(256) void k_update_e_pml(const T *__restrict__ ez_old, T *__restrict__ ez_new,
                                                      T *__restrict__ ezx, const T *__restrict__ hx,
                                                      const T *__restrict__ hy, const T *__restrict__ ce,
                                                      T ce_u, PmlFactors<T> f, Geom g, int lo, int hi)
{
    constexpr int V = Vec<T>::N;
    const int j0 = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int i0 = lo + (blockIdx.y * blockDim.y + threadIdx.y) * RPT;
    if (j0 >= g.C || i0 >= hi) return;
    T ac[V], bc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int j = j0 + v < g.C ? j0 + v : g.C - 1;
        ac[v] = f.aec[j];
        bc[v] = f.bec[j];
    }
    Vec<T> xu;
    if (i0 >= 1) {
        xu = ldv(hx + at(g, i0 - 1, j0));
    } else {
#pragma unroll
        for (int v = 0; v < V; ++v) xu.v[v] = T(0);
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int i = i0 + r;
        if (i >= hi) break;
        const size_t o = at(g, i, j0);
        const Vec<T> x = ldv(hx + o), y = ldv(hy + o), e = ldv(ez_old + o);
        const bool rin = f.row_in(i);
        const bool touch = rin || f.col_in(j0) || f.col_in(j0 + V - 1);   // any cell of this vector in the layer
        Vec<T> ex;
        if (touch) ex = ldv(ezx + o);
        const T yl = (j0 > 0) ? hy[o - 1] : T(0);
        Vec<T> c, out;
        if (CE_ARR) c = ldv(ce + o);
        const bool row_in = (i >= 1) && (i <= g.R - 2);
        const T ar = f.aer[i], br = f.ber[i];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int j = j0 + v;
            const T cc = CE_ARR ? c.v[v] : ce_u;
            const T left = (v > 0) ? y.v[v - 1] : yl;
            const T dhy = y.v[v] - left, dhx = x.v[v] - xu.v[v];
            if (row_in && j >= 1 && j <= g.C - 2) {
                if (rin || f.col_in(j)) {          // split-field update inside the layer
                    T ey = e.v[v] - ex.v[v];
                    ex.v[v] = ac[v] * ex.v[v] + (bc[v] * cc) * dhy;
                    ey = ar * ey - (br * cc) * dhx;
                    out.v[v] = ex.v[v] + ey;
                } else {                            // the reference's update, main.py:21-27
                    out.v[v] = e.v[v] + (dhy - dhx) * cc;
                }
            } else {
                out.v[v] = e.v[v];
            }
        }
        stv(ez_new + o, out);
        if (touch) stv(ezx + o, ex);
        xu = x;
    }
}

// ---- temporally blocked pass with the PML (8 steps per launch) ----------------------------------
// Same streaming scheme as k_bulk (kernels_stream.hpp): one wave per (band, strip), row-slot
// ring in registers, DPP lane shifts.  There are no top/bottom zones in PML mode (no row
// coupling beyond the stencil), so the bands cover all rows and the body guards the grid's
// first/last row itself.  Waves whose dependency cone touches the layer run pml_body, which
// carries the split field Ezx in the slot and selects per cell between the split update
// (inside the layer) and the reference's update (outside); all other waves run the plain
// mask-free body of k_bulk and never touch Ezx.
template <class T> struct PmlPass {
    PmlFactors<T> f;
    const T *ezx_in;
    T *ezx_out;
    // Launch layout.  Waves that run pml_body are several times slower per row than plain
    // waves, so they get short bands (short_rows) and come first in launch order:
    //   class 1: strips 0 and last, rows [band_lo, band_hi) in short bands (n1 per strip)
    //   class 2: strips 1..nstrips-2, rows [band_lo, a_hi) and [c_lo, band_hi) in short bands
    //   class 3: strips 1..nstrips-2, rows [a_hi, c_lo) in normal bands (plain body)
    int short_rows, a_hi, c_lo, n1, nA, nC, nB;
};

template <class T, bool CE_ARR> struct PmlSlot {
    Vec<T> e, x, y, ex, ce;
};

template <class T, int NT, bool CE_ARR>
__device__ __forceinline__ void pml_body(const PassParams<T> &p, const PmlPass<T> &q, const int strip,
                                         const int ra, const int rb)
{
    constexpr int V = Vec<T>::N;
    constexpr int SW = 64 * V, HC = stream_hc(NT), OW = SW - 2 * HC;
    constexpr int PF = 1;
    constexpr int S = NT + PF + 2;
    const Geom g = p.g;
    const PmlFactors<T> &f = q.f;
    const int lane = threadIdx.x;
    const int x0 = strip_x0<T, NT>(p, strip);
    const int j0 = x0 + V * lane;
    const bool ld_ok = j0 >= 0 && j0 < g.C;
    const bool st_ok = ld_ok && j0 >= strip * OW && j0 < (strip + 1) * OW;
    const size_t col = (size_t)(ld_ok ? j0 : 0);
    const int tau0 = ra - NT, tau1 = rb + NT;

    // per-element column data, fixed for the strip
    Vec<T> ahc, bhc, aec, bec;
    bool mh[V], me[V], cin[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int j = j0 + v;
        const int jc = j < 0 ? 0 : (j >= g.C ? g.C - 1 : j);
        ahc.v[v] = f.ahc[jc];
        bhc.v[v] = f.bhc[jc];
        aec.v[v] = f.aec[jc];
        bec.v[v] = f.bec[jc];
        mh[v] = j >= 0 && j <= g.C - 2;       // Hx, Hy exist / are updated (main.py:70,74)
        me[v] = j >= 1 && j <= g.C - 2;       // Ez interior column
        cin[v] = f.col_in(jc);
    }

    PmlSlot<T, CE_ARR> slot[S];
#pragma unroll
    for (int k = 0; k < S; ++k)
#pragma unroll
        for (int v = 0; v < V; ++v)
            slot[k].e.v[v] = slot[k].x.v[v] = slot[k].y.v[v] = slot[k].ex.v[v] = slot[k].ce.v[v] = T(0);

    // unconditional memory operations, as in stream_body: rows clamped into what exists (rows
    // outside the grid or the band are never processed), lanes outside the grid zeroed
    const int row_lo = max(tau0, max(0, g.row_base)), row_hi = min(tau1, g.R) - 1;
    auto load_row = [&](PmlSlot<T, CE_ARR> &r, int i) {
        const size_t o = at(g, min(max(i, row_lo), row_hi), 0) + col;
        r.e = ldv(p.ez_in + o);
        r.x = ldv(p.hx_in + o);
        r.y = ldv(p.hy_in + o);
        r.ex = ldv(q.ezx_in + o);
        if (CE_ARR) r.ce = ldv(p.ce + o);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            r.e.v[v] = ld_ok ? r.e.v[v] : T(0);
            r.x.v[v] = ld_ok ? r.x.v[v] : T(0);
            r.y.v[v] = ld_ok ? r.y.v[v] : T(0);
            r.ex.v[v] = ld_ok ? r.ex.v[v] : T(0);
        }
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) load_row(slot[k], tau0 + k);

    for (int tb = tau0; tb < tau1; tb += S) {
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const int tau = tb + k;
            if (tau >= tau1) break;
            load_row(slot[(k + PF) % S], tau + PF);
#pragma unroll
            for (int t = 1; t <= NT; ++t) {
                const int i = tau - t;
                if (i < ra - (NT - t) - 1 || i >= rb + (NT - t)) continue;
                if (i < 0 || i > g.R - 1) continue;                     // outside the grid
                PmlSlot<T, CE_ARR> &c = slot[(k - t + 2 * S) % S];
                const PmlSlot<T, CE_ARR> &nx = slot[(k - t + 1 + 2 * S) % S];
                const PmlSlot<T, CE_ARR> &pv = slot[(k - t - 1 + 2 * S) % S];
                const T e_next_lane = from_next(c.e.v[0]);
                if (i <= g.R - 2) {                                     // H rows 0..R-2
                    const T ar = f.ahr[i], br = f.bhr[i];
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const T right = (v + 1 < V) ? c.e.v[v + 1] : e_next_lane;
                        const T hx = ar * c.x.v[v] - (br * p.ch_u) * (nx.e.v[v] - c.e.v[v]);
                        const T hy = ahc.v[v] * c.y.v[v] + (bhc.v[v] * p.ch_u) * (right - c.e.v[v]);
                        c.x.v[v] = mh[v] ? hx : c.x.v[v];
                        c.y.v[v] = mh[v] ? hy : c.y.v[v];
                    }
                }
                if (i >= 1 && i <= g.R - 2) {                           // Ez rows 1..R-2
                    const T hy_prev_lane = from_prev(c.y.v[V - 1]);
                    const bool rin = f.row_in(i);
                    const T ar = f.aer[i], br = f.ber[i];
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const T ce = CE_ARR ? c.ce.v[v] : p.ce_u;
                        const T left = (v > 0) ? c.y.v[v - 1] : hy_prev_lane;
                        const T dhy = c.y.v[v] - left, dhx = c.x.v[v] - pv.x.v[v];
                        const T plain = c.e.v[v] + (dhy - dhx) * ce;
                        T ey = c.e.v[v] - c.ex.v[v];
                        const T ex = aec.v[v] * c.ex.v[v] + (bec.v[v] * ce) * dhy;
                        ey = ar * ey - (br * ce) * dhx;
                        const bool lay = rin || cin[v];
                        c.ex.v[v] = (me[v] && lay) ? ex : c.ex.v[v];
                        c.e.v[v] = me[v] ? (lay ? ex + ey : plain) : c.e.v[v];
                    }
                }
                if (i >= p.src_row && i < p.src_row1) {
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        if (j0 + v >= p.src_col && j0 + v < p.src_col1)
                            c.e.v[v] = (T)((double)c.e.v[v] + p.amp[t - 1]);
                }
            }
            {
                const int io = tau - NT;
                const PmlSlot<T, CE_ARR> &o_ = slot[(k - NT + 2 * S) % S];
                const bool keep = st_ok && io >= ra;
                const size_t o = at(g, max(io, ra), 0) + col;
                const size_t d = (size_t)(blockIdx.x % TRASH_SLOTS) * (TRASH_SLOT_BYTES / sizeof(T)) + (size_t)lane * V;
                stv(keep ? p.ez_out + o : p.trash + d, o_.e);
                stv(keep ? p.hx_out + o : p.trash + d + 64 * V, o_.x);
                stv(keep ? p.hy_out + o : p.trash + d + 128 * V, o_.y);
                stv(keep ? q.ezx_out + o : p.trash + d + 192 * V, o_.ex);
            }
        }
    }
}

template <class T, bool CE_ARR>
__global__ __launch_bounds__(64, 2) void k_pass_pml(const PassParams<T> p, const PmlPass<T> q)
{
    constexpr int NT = 8;
    constexpr int V = Vec<T>::N;
    constexpr int SW = 64 * V;
    int b = blockIdx.x;
    int strip, ra, rb;
    const int inner = max(0, p.nstrips - 2);
    if (b < 2 * q.n1) {                               // class 1
        const int sidx = b / q.n1, band = b - sidx * q.n1;
        if (sidx == 1 && p.nstrips == 1) return;
        strip = sidx == 0 ? 0 : p.nstrips - 1;
        ra = p.band_lo + band * q.short_rows;
        rb = min(ra + q.short_rows, p.band_hi);
    } else if ((b -= 2 * q.n1) < inner * (q.nA + q.nC)) {   // class 2
        const int per = q.nA + q.nC;
        const int sidx = b / per, band = b - sidx * per;
        strip = sidx + 1;
        if (band < q.nA) {
            ra = p.band_lo + band * q.short_rows;
            rb = min(ra + q.short_rows, q.a_hi);
        } else {
            ra = q.c_lo + (band - q.nA) * q.short_rows;
            rb = min(ra + q.short_rows, p.band_hi);
        }
    } else {                                          // class 3
        b -= inner * (q.nA + q.nC);
        const int sidx = b / q.nB, band = b - sidx * q.nB;
        strip = sidx + 1;
        ra = q.a_hi + band * p.band_rows;
        rb = min(ra + p.band_rows, q.c_lo);
    }
    if (ra >= rb) return;
    const int x0 = strip_x0<T, NT>(p, strip);
    const int L = q.f.L;
    // does the wave's cone (rows [ra-2NT, rb+NT), columns [x0, x0+SW)) touch the layer or the edge?
    const bool layer = x0 < L + 1 || x0 + SW > p.g.C - 1 - L || ra - 2 * NT < L + 1 ||
                       rb + NT > p.g.R - 1 - L;
    const bool src = p.src_row1 > ra - 2 * NT && p.src_row < rb + NT && p.src_col1 > x0 &&
                     p.src_col < x0 + SW;
    if (layer)
        pml_body<T, NT, CE_ARR>(p, q, strip, ra, rb);
    else if (src)
        stream_body<T, NT, CE_ARR, false, true>(p, strip, ra, rb);
    else
        stream_body<T, NT, CE_ARR, false, false>(p, strip, ra, rb);
}

// 4-field halo message (Ez, Ezx, Hx, Hy) for slabs in PML mode
template <class T, bool PACK>
__global__ __launch_bounds__(256) void k_halo4(T *__restrict__ f0, T *__restrict__ f1, T *__restrict__ f2,
                                               T *__restrict__ f3, T *__restrict__ msg, Geom g,
                                               int row_first, int nrows)
{
    const size_t per = (size_t)nrows * g.C;
    const size_t n = 4 * per;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) {
        const int fi = (int)(t / per);
        const size_t r = t % per;
        const int i = row_first + (int)(r / g.C), j = (int)(r % g.C);
        T *fld = fi == 0 ? f0 : (fi == 1 ? f1 : (fi == 2 ? f2 : f3));
        if (PACK) msg[t] = fld[at(g, i, j)];
        else fld[at(g, i, j)] = msg[t];
    }
}

}  // namespace fdtd

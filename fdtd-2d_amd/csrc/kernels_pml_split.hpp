// Level-split form of the temporally blocked pass for boundary = FDTD2D_BOUNDARY_PML (build-defined
// split-field layer, oracle/pml_numpy.py; parity unpinned): 16 steps per launch like the Mur frame's
// k_bulk_split, instead of the 8 steps of the single-wave k_pass_pml.
//
// A pass is TWO kernels that read the same input buffers and write disjoint cells of the output
// buffers, so they run side by side (main stream / side stream):
//   * k_bulk_split (kernels_split.hpp, unchanged) on every (band, strip) whose 16-step dependency cone
//     stays clear of the layer and of the grid edge: there the split update never applies and Ezx is
//     identically zero, so the reference's own update on (Ez, Hx, Hy) is the whole story;
//   * k_bulk_split_pml (this file) on the rest -- the first / last strips over all rows and, on the
//     slabs that own the grid's top / bottom, the first / last bands of every strip.  Its rows carry the
//     fourth field Ezx through the slot ring and the LDS hand-off and select per cell between the
//     split update (inside the layer) and the reference's (outside), exactly as k_update_e_pml does.
// There are no zone tiles in PML mode (no row coupling beyond the stencil): bands cover every row and
// the body guards the grid's first / last row itself.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_pml.hpp"
#include "kernels_split.hpp"

namespace fdtd {

// which (band, strip) tasks the layer kernel owns: strips [0, n_left) and [nstrips - n_right, nstrips)
// over all bands of `rows_e` rows, the strips between them over the rows [band_lo, a_hi) and
// [c_lo, band_hi) in bands of `rows_tb` rows (one band each where the slab owns the grid's top / bottom: the
// rows a 16-step cone from the layer reaches, not rounded up to the edge strips' band height -- two 64-row
// bands per end cost 230 ticks for 73 needed rows, one 80-row band 131)
template <class T> struct PmlSplit {
    PmlFactors<T> f;
    const T *ezx_in;
    T *ezx_out;
    int n_left, n_right;       // layer strips at the left / right end
    int rows_e;                // band height of the layer tasks of the end strips
    int rows_tb;               // band height of the top / bottom tasks of the strips between them
    int a_hi, c_lo;            // rows [band_lo, a_hi) and [c_lo, band_hi) belong to the layer kernel in every strip
    int n_all, n_top, n_bot;   // bands per edge strip / top bands / bottom bands per inner strip
};

template <class T, bool CE_ARR, int V> struct PmlStripMath {
    using VT = VecN<T, V>;
    struct Row {
        VT e, x, y, ex;
        VT ce;          // travels with the row when eps is an array
    };
    const PassParams<T> &p;
    const PmlFactors<T> &f;
    int j0;
    bool ld_ok;
    bool inner;         // a strip whose columns all lie outside the column layers (uniform per workgroup)
    VT ahc, bhc, aec, bec;
    bool mh[V], me[V], cin[V];

    __device__ __forceinline__ PmlStripMath(const PassParams<T> &pp, const PmlFactors<T> &ff, int x0, int lane,
                                            bool inner_strip)
        : p(pp), f(ff), inner(inner_strip)
    {
        j0 = x0 + V * lane;
        ld_ok = j0 >= 0 && j0 < p.g.C;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int j = j0 + v;
            const int jc = j < 0 ? 0 : (j >= p.g.C ? p.g.C - 1 : j);
            ahc.v[v] = f.ahc[jc];
            bhc.v[v] = f.bhc[jc];
            aec.v[v] = f.aec[jc];
            bec.v[v] = f.bec[jc];
            mh[v] = j >= 0 && j <= p.g.C - 2;       // Hx, Hy exist / are updated (main.py:70,74)
            me[v] = j >= 1 && j <= p.g.C - 2;       // Ez interior column
            cin[v] = f.col_in(jc);
        }
    }

    // row i: level t-1 -> t, in place; same operations, in the same order, as k_update_h_pml /
    // k_update_e_pml (and oracle/pml_numpy.py step())
    __device__ __forceinline__ void level(Row &c, const VT &nxe, const VT &pvx, int t, int i) const
    {
        if (i < 0 || i > p.g.R - 1) return;                         // outside the grid
        // Rows outside the row layers of a strip outside the column layers: every factor is exactly 1 there
        // (fdtd2d_set_pml checks the arrays it is given: H factors on rows L .. R-2-L, E factors on L .. R-1-L) and
        // no cell is in the layer, so the split update IS the reference's (1*x - (1*ch)*d == x - ch*d bit for
        // bit) and Ezx keeps its value: the plain staged body, 44 instructions instead of ~110, and no factor
        // loads.  Most rows of the top / bottom tasks are such rows (the task's cone touches the layer, its rows
        // mostly do not).
        if (inner && i >= f.L && i <= f.R - 2 - f.L) {
            staged_level<T, V>(c.e, c.x, c.y, nxe, pvx, [&](int) { return p.ch_u; },
                               [&](int v) { return CE_ARR ? c.ce.v[v] : p.ce_u; });
            if (i >= p.src_row && i < p.src_row1) {
                const double amp = p.amp[t - 1];
#pragma unroll
                for (int v = 0; v < V; ++v)
                    if (j0 + v >= p.src_col && j0 + v < p.src_col1) c.e.v[v] = (T)((double)c.e.v[v] + amp);
            }
            return;
        }
        const T e_next_lane = from_next(c.e.v[0]);
        if (i <= p.g.R - 2) {                                       // H rows 0..R-2
            const T ar = f.ahr[i], br = f.bhr[i];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const T right = (v + 1 < V) ? c.e.v[v + 1] : e_next_lane;
                const T hx = ar * c.x.v[v] - (br * p.ch_u) * (nxe.v[v] - c.e.v[v]);
                const T hy = ahc.v[v] * c.y.v[v] + (bhc.v[v] * p.ch_u) * (right - c.e.v[v]);
                c.x.v[v] = mh[v] ? hx : c.x.v[v];
                c.y.v[v] = mh[v] ? hy : c.y.v[v];
            }
        }
        if (i >= 1 && i <= p.g.R - 2) {                             // Ez rows 1..R-2
            const T hy_prev_lane = from_prev(c.y.v[V - 1]);
            const bool rin = f.row_in(i);
            const T ar = f.aer[i], br = f.ber[i];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const T ce = CE_ARR ? c.ce.v[v] : p.ce_u;
                const T left = (v > 0) ? c.y.v[v - 1] : hy_prev_lane;
                const T dhy = c.y.v[v] - left, dhx = c.x.v[v] - pvx.v[v];
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
            const double amp = p.amp[t - 1];
#pragma unroll
            for (int v = 0; v < V; ++v)
                if (j0 + v >= p.src_col && j0 + v < p.src_col1) c.e.v[v] = (T)((double)c.e.v[v] + amp);
        }
    }
};

// the tick loop of split_body (kernels_split.hpp) with four-field rows; see there for the pipeline
template <class T, int NT, int SPLIT_NW, bool CE_ARR, int ROLE, int V>
__device__ __forceinline__ void split_body_pml(const PassParams<T> &p, const PmlSplit<T> &q, const int strip,
                                               const int ra, const int rb, const int w, VecN<T, V> *lds,
                                               const bool inner)
{
    using M = PmlStripMath<T, CE_ARR, V>;
    using Row = typename M::Row;
    constexpr int NF = 4 + (CE_ARR ? 1 : 0);                        // rows per hand-off
    constexpr int LV = NT / SPLIT_NW, LAG = LV + 1;
    constexpr int HC = stream_hc(NT);
    constexpr int SW = 64 * V, OW = SW - 2 * HC;
    constexpr int PF = ROLE == 0 ? STREAM_PF : 0;
    constexpr int S = LV + 2 + PF;
    const Geom g = p.g;
    const int lane = threadIdx.x & 63;
    const int x0 = strip_x0<T, NT, V>(p, strip);
    const M m(p, q.f, x0, lane, inner);
    const int j0 = m.j0;
    const bool st_ok = m.ld_ok && j0 >= strip * OW && j0 < (strip + 1) * OW;
    const size_t col = (size_t)(m.ld_ok ? j0 : 0);
    const int tau0 = ra - NT, tau1 = rb + NT;
    const int tend = rb + LV + (SPLIT_NW - 1) * LAG;
    const int shift = w * LAG, t0 = w * LV;
    const int r_end = rb + NT - t0;
    int first[LV + 1];
#pragma unroll
    for (int l = 1; l <= LV; ++l) first[l] = t0 + l <= p.nlev ? ra - NT + t0 + 2 * l - 1 : (1 << 30);
    auto buf = [&](int h, int d, int field) { return lds + ((h * HAND_DEPTH + d) * NF + field) * 64 + lane; };

    Row slot[S];
#pragma unroll
    for (int k = 0; k < S; ++k)
#pragma unroll
        for (int v = 0; v < V; ++v)
            slot[k].e.v[v] = slot[k].x.v[v] = slot[k].y.v[v] = slot[k].ex.v[v] = slot[k].ce.v[v] = T(0);

    // unconditional loads: rows clamped into what this handle stores (rows outside the grid or the
    // band are never processed), lanes outside the grid zeroed
    const int row_lo = max(tau0, max(0, g.row_base)), row_hi = min(tau1, g.R) - 1;
    // addresses as scalar row pointer + the lane's 32-bit byte offset, as in split_body (kernels_split.hpp)
    unsigned lane_off = (unsigned)col * (unsigned)sizeof(T);
    typedef const char __attribute__((address_space(1))) *gcptr;
    auto row_ptr = [&](const T *field, int i) {
        gcptr rp = (gcptr)(field + at(g, i, 0));
        asm("" : "+s"(rp));
        return (const char *)rp;
    };
    auto lane_off_now = [&]() { asm("" : "+v"(lane_off)); return lane_off; };
    auto load_global = [&](Row &r, int i) {
        const int ic = min(max(i, row_lo), row_hi);
        const unsigned lo = lane_off_now();
        r.e = ldn<V>(reinterpret_cast<const T *>(row_ptr(p.ez_in, ic) + lo));
        r.x = ldn<V>(reinterpret_cast<const T *>(row_ptr(p.hx_in, ic) + lo));
        r.y = ldn<V>(reinterpret_cast<const T *>(row_ptr(p.hy_in, ic) + lo));
        r.ex = ldn<V>(reinterpret_cast<const T *>(row_ptr(q.ezx_in, ic) + lo));
        if (CE_ARR) r.ce = ldn<V>(reinterpret_cast<const T *>(row_ptr(p.ce, ic) + lo));
        if (!inner) {           // (every lane of a strip between the column layers is inside the grid)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                r.e.v[v] = m.ld_ok ? r.e.v[v] : T(0);
                r.x.v[v] = m.ld_ok ? r.x.v[v] : T(0);
                r.y.v[v] = m.ld_ok ? r.y.v[v] : T(0);
                r.ex.v[v] = m.ld_ok ? r.ex.v[v] : T(0);
            }
        }
    };
    if (ROLE == 0) {
#pragma unroll
        for (int k = 0; k < PF; ++k) load_global(slot[k], tau0 + k);
    }

    for (int tb = tau0; tb < tend; tb += S) {
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const int tau = tb + k;
            if (tau >= tend) break;
            const int r = tau - shift;
            if (ROLE == 0) {
                load_global(slot[(k + PF) % S], r + PF);
            } else {
                const int dr = (tau + 1) & 1;
                Row &in = slot[k];
                in.e = *buf(w - 1, dr, 0);
                in.x = *buf(w - 1, dr, 1);
                in.y = *buf(w - 1, dr, 2);
                in.ex = *buf(w - 1, dr, 3);
                if (CE_ARR) in.ce = *buf(w - 1, dr, 4);
            }
            if (__builtin_expect(r < r_end, 1)) {
#pragma unroll
                for (int l = 1; l <= LV; ++l) {
                    if (__builtin_expect(r < first[l], 0)) continue;
                    Row &c = slot[(k - l + 2 * S) % S];
                    m.level(c, slot[(k - l + 1 + 2 * S) % S].e, slot[(k - l - 1 + 2 * S) % S].x, t0 + l, r - l);
                }
            }
            const Row &f = slot[(k - LV + 2 * S) % S];
            if (ROLE == 2) {
                const int io = r - LV;
                if (io >= ra && io < rb) {            // a row of the band (uniform) ...
                    if (st_ok) {                      // ... and a column this strip owns (lanes masked off otherwise)
                        const unsigned lo = lane_off_now();
                        stn<V>(reinterpret_cast<T *>(const_cast<char *>(row_ptr(p.ez_out, io)) + lo), f.e);
                        stn<V>(reinterpret_cast<T *>(const_cast<char *>(row_ptr(p.hx_out, io)) + lo), f.x);
                        stn<V>(reinterpret_cast<T *>(const_cast<char *>(row_ptr(p.hy_out, io)) + lo), f.y);
                        stn<V>(reinterpret_cast<T *>(const_cast<char *>(row_ptr(q.ezx_out, io)) + lo), f.ex);
                    }
                }
            } else {
                const int d = tau & 1;
                *buf(w, d, 0) = f.e;
                *buf(w, d, 1) = f.x;
                *buf(w, d, 2) = f.y;
                *buf(w, d, 3) = f.ex;
                if (CE_ARR) *buf(w, d, 4) = f.ce;
            }
            __syncthreads();
        }
    }
}

// Launch order: the layer strips at the two ends (all bands), then per inner strip its top and
// bottom bands.
template <class T, int NT, int SPLIT_NW, bool CE_ARR, int V = Vec<T>::N>
__global__ __launch_bounds__(64 * SPLIT_NW) void k_bulk_split_pml(const PassParams<T> p, const PmlSplit<T> q)
{
    static_assert(NT % SPLIT_NW == 0, "levels must divide evenly over the waves");
    constexpr int NF = 4 + (CE_ARR ? 1 : 0);
    constexpr int HAND = (SPLIT_NW - 1) * HAND_DEPTH * NF * 64;
    __shared__ VecN<T, V> lds[HAND];
    int b = blockIdx.x;
    int strip, ra, rb;
    const int n_edge = q.n_left + q.n_right;
    const bool inner = b >= n_edge * q.n_all;       // the top / bottom tasks of the strips between the column layers
    if (b < n_edge * q.n_all) {
        const int sidx = b / q.n_all, band = b - sidx * q.n_all;
        strip = sidx < q.n_left ? sidx : p.nstrips - n_edge + sidx;
        ra = p.band_lo + band * q.rows_e;
        rb = min(ra + q.rows_e, p.band_hi);
    } else {
        b -= n_edge * q.n_all;
        const int per = q.n_top + q.n_bot;
        const int sidx = b / per, band = b - sidx * per;
        strip = q.n_left + sidx;
        if (band < q.n_top) {
            ra = p.band_lo + band * q.rows_tb;
            rb = min(ra + q.rows_tb, q.a_hi);
        } else {
            ra = q.c_lo + (band - q.n_top) * q.rows_tb;
            rb = min(ra + q.rows_tb, p.band_hi);
        }
    }
    if (ra >= rb) return;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int n = threadIdx.x; n < HAND; n += 64 * SPLIT_NW)
#pragma unroll
        for (int v = 0; v < V; ++v) lds[n].v[v] = T(0);
    __syncthreads();
    if (w == 0) split_body_pml<T, NT, SPLIT_NW, CE_ARR, 0, V>(p, q, strip, ra, rb, w, lds, inner);
    else if (w == SPLIT_NW - 1) split_body_pml<T, NT, SPLIT_NW, CE_ARR, 2, V>(p, q, strip, ra, rb, w, lds, inner);
    else split_body_pml<T, NT, SPLIT_NW, CE_ARR, 1, V>(p, q, strip, ra, rb, w, lds, inner);
}

}  // namespace fdtd

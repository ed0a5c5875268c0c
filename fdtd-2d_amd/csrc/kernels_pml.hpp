// Half-step kernels for boundary = FDTD2D_BOUNDARY_PML: Berenger's split-field PML for the
// TE-mode system (build-defined: the reference has no time-domain PML; parity unpinned, see
// oracle/pml_numpy.py for the definition these kernels follow operation for operation).
// Ez = Ezx + Ezy; the engine stores Ez (total, ping-pong) and Ezx (updated in place).
// Loss factors are eight 1-D arrays (rows: ahr, bhr, aer, ber; columns: ahc, bhc, aec, bec),
// equal to 1 outside the L-cell layer, so HBM traffic per cell grows only by the Ezx field.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_step.hpp"

namespace fdtd {

template <class T> struct PmlFactors {
    const T *ahr, *bhr, *aer, *ber;   // indexed by global row
    const T *ahc, *bhc, *aec, *bec;   // indexed by column
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
        Vec<T> ex = ldv(ezx + o);
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
            if (row_in && j >= 1 && j <= g.C - 2) {
                T ey = e.v[v] - ex.v[v];
                ex.v[v] = ac[v] * ex.v[v] + (bc[v] * cc) * (y.v[v] - left);
                ey = ar * ey - (br * cc) * (x.v[v] - xu.v[v]);
                out.v[v] = ex.v[v] + ey;
            } else {
                out.v[v] = e.v[v];
            }
        }
        stv(ez_new + o, out);
        stv(ezx + o, ex);
        xu = x;
    }
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

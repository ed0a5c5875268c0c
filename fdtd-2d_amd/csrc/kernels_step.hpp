// Single half-step kernels for gfx950 (CDNA4): the drop-in path behind
// fdtd2d_update_h / fdtd2d_update_e / fdtd2d_add_point.
//
// Arithmetic follows the reference cell for cell (python-src/main.py:12-76) in the
// engine's type T, one rounding per operation (the library is built with
// -ffp-contract=off: NumPy never fuses a*b+c), so results are value-identical to the
// reference run on arrays of type T.
//
// Storage: every field is `rows_stored x pitch` elements, row-major, pitch a multiple
// of 64 elements; element (i, j) of the global grid lives at (i - row_base)*pitch + j.
// Hx's missing column C-1 and Hy's missing row R-1 exist as permanent zeros.
// Each lane moves 16 bytes per access (float4 / double2): a wave covers 1 KiB of a row.
#pragma once
#include <hip/hip_runtime.h>

#include "mur_rules.hpp"

namespace fdtd {

template <class T> struct alignas(16) Vec {
    static constexpr int N = 16 / sizeof(T);
    T v[N];
};

// N elements per lane, naturally aligned (N * sizeof(T) = 8 or 16 bytes -> dwordx2 / dwordx4)
template <class T, int NN> struct alignas(NN * sizeof(T)) VecN {
    static constexpr int N = NN;
    T v[NN];
};
template <int NN, class T>
__device__ __forceinline__ VecN<T, NN> ldn(const T *p) { return *reinterpret_cast<const VecN<T, NN> *>(p); }
template <int NN, class T>
__device__ __forceinline__ void stn(T *p, const VecN<T, NN> &x) { *reinterpret_cast<VecN<T, NN> *>(p) = x; }

template <class T>
__device__ __forceinline__ Vec<T> ldv(const T *p) { return *reinterpret_cast<const Vec<T> *>(p); }
template <class T>
__device__ __forceinline__ void stv(T *p, const Vec<T> &x) { *reinterpret_cast<Vec<T> *>(p) = x; }

struct Geom {
    int R, C;           // global grid
    int row_base;       // global row of stored row 0
    long long pitch;    // elements per stored row
};

__host__ __device__ __forceinline__ size_t at(const Geom &g, int i, int j)
{
    return (size_t)(i - g.row_base) * (size_t)g.pitch + (size_t)j;
}

// ---- H half-step: main.py:66-76 ------------------------------------------------------
// rows [lo, hi) with hi <= R-1, columns 0..C-2.  Each thread owns V columns and marches
// RPT rows down, carrying Ez[i+1] into the next iteration's Ez[i].
template <class T, bool CH_ARR, int RPT>
__global__ __launch_bounds__(256) void k_update_h(const T *__restrict__ ez, T *__restrict__ hx,
                                                  T *__restrict__ hy, const T *__restrict__ ch,
                                                  T ch_u, Geom g, int lo, int hi)
{
    constexpr int V = Vec<T>::N;
    const int j0 = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int i0 = lo + (blockIdx.y * blockDim.y + threadIdx.y) * RPT;
    if (j0 > g.C - 2 || i0 >= hi) return;
    Vec<T> e = ldv(ez + at(g, i0, j0));
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int i = i0 + r;
        if (i >= hi) break;
        const size_t o = at(g, i, j0);
        const Vec<T> en = ldv(ez + o + g.pitch);
        const T er = (j0 + V < g.C) ? ez[o + V] : T(0);
        Vec<T> x = ldv(hx + o), y = ldv(hy + o), c;
        if (CH_ARR) c = ldv(ch + o);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (j0 + v <= g.C - 2) {
                const T cc = CH_ARR ? c.v[v] : ch_u;
                const T right = (v + 1 < V) ? e.v[v + 1] : er;
                x.v[v] = x.v[v] - cc * (en.v[v] - e.v[v]);
                y.v[v] = y.v[v] + cc * (right - e.v[v]);
            }
        }
        stv(hx + o, x);
        stv(hy + o, y);
        e = en;
    }
}

// ---- E half-step, stage A: main.py:18-27 ---------------------------------------------
// rows [lo, hi); writes EVERY cell of those rows into ez_new: interior cells get the curl
// update, edge cells (row 0, row R-1, column 0, column C-1, padding) a copy of ez_old.
template <class T, bool CE_ARR, int RPT>
__global__ __launch_bounds__(256) void k_update_e(const T *__restrict__ ez_old,
                                                  T *__restrict__ ez_new,
                                                  const T *__restrict__ hx,
                                                  const T *__restrict__ hy,
                                                  const T *__restrict__ ce, T ce_u, Geom g, int lo,
                                                  int hi)
{
    constexpr int V = Vec<T>::N;
    const int j0 = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int i0 = lo + (blockIdx.y * blockDim.y + threadIdx.y) * RPT;
    if (j0 >= g.C || i0 >= hi) return;
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
        const T yl = (j0 > 0) ? hy[o - 1] : T(0);
        Vec<T> c, out;
        if (CE_ARR) c = ldv(ce + o);
        const bool row_in = (i >= 1) && (i <= g.R - 2);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int j = j0 + v;
            const T cc = CE_ARR ? c.v[v] : ce_u;
            const T left = (v > 0) ? y.v[v - 1] : yl;
            const T upd = e.v[v] + ((y.v[v] - left) - (x.v[v] - xu.v[v])) * cc;
            out.v[v] = (row_in && j >= 1 && j <= g.C - 2) ? upd : e.v[v];
        }
        stv(ez_new + o, out);
        xu = x;
    }
}

// ---- E half-step, stages B, C, D: main.py:29-61 -----------------------------------------
// Every frame cell is a pure function (mur_rules.hpp) of P (= ez_old) and the new H fields
// in a small neighbourhood, written to ez_new; nothing here reads ez_new, so there is no
// ordering hazard between frame cells and the kernel can follow k_update_e on the stream.
template <class T, bool CE_ARR> struct GlobalAcc {
    const T *P, *x, *y, *c;
    T ce_u;
    Geom g;
    int R, C;
    __device__ __forceinline__ T p(int i, int j) const { return P[at(g, i, j)]; }
    __device__ __forceinline__ T hx(int i, int j) const { return x[at(g, i, j)]; }
    __device__ __forceinline__ T hy(int i, int j) const { return y[at(g, i, j)]; }
    __device__ __forceinline__ T ce(int i, int j) const { return CE_ARR ? c[at(g, i, j)] : ce_u; }
};
template <class T, bool CE_ARR> using FrameCtx = MurRules<T, GlobalAcc<T, CE_ARR>>;

// Thread map: first n_lr threads cover the left/right bands of rows [lo, hi) (16 slots
// per row: 5 left, 5 right, 6 idle); then, if has_top / has_bot, 5 x C threads each for
// the horizontal bands.  Rows inside a horizontal band are covered by that band only.
template <class T, bool CE_ARR>
__global__ __launch_bounds__(256) void k_frame_mur(FrameCtx<T, CE_ARR> f, T *__restrict__ ez_new,
                                                   int lo, int hi, int has_top, int has_bot)
{
    const Geom &g = f.m.g;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int vlo = has_top ? (lo > 5 ? lo : 5) : lo;
    const int vhi = has_bot ? (hi < g.R - 5 ? hi : g.R - 5) : hi;
    const long long n_lr = (vhi > vlo) ? (long long)(vhi - vlo) * 16 : 0;
    int i, j;
    if (t < n_lr) {
        i = vlo + (int)(t >> 4);
        const int s = (int)(t & 15);
        if (s < 5) j = s;
        else if (s < 10) j = g.C - 10 + s;
        else return;
        if (i < 1 || i > g.R - 2) return;   // bands B only touch rows 1..R-2
    } else {
        long long u = t - n_lr;
        const long long band = 5LL * g.C;
        if (has_top && u < band) {
            i = (int)(u / g.C);
            j = (int)(u % g.C);
        } else {
            if (has_top) u -= band;
            if (!has_bot || u >= band) return;
            i = g.R - 5 + (int)(u / g.C);
            j = (int)(u % g.C);
        }
    }
    ez_new[at(g, i, j)] = f.d(i, j);
}

// ---- point source: fdtd.py:34 ---------------------------------------------------------------
// (one cell, or the same amplitude on every cell of an nr x nc rectangle: a line / patch source)
template <class T> __global__ void k_add_point(T *ez, Geom g, int row, int col, int nr, int nc, double amp)
{
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < nr * nc; n += gridDim.x * blockDim.x) {
        const size_t off = at(g, row + n / nc, col + n % nc);
        ez[off] = (T)((double)ez[off] + amp);
    }
}

// ---- coefficient arrays: x -> dt/(x*dx) in T (main.py:27,70,74) ------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_coef(T *__restrict__ a, size_t n, T dt, T dx)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) {
        const T x = a[t];
        a[t] = (x != T(0)) ? dt / (x * dx) : T(0);   // padding (0) stays 0
    }
}

// ---- snapshot: Ez -> colour-map index, decimated (main.py:155,167-168) -----------------------
// idx = trunc(256 * (clip(Ez, vmin, vmax) - vmin) / (vmax - vmin)), 256 -> 255: the LUT index
// matplotlib's Colormap.__call__ derives for the reference's `cmap((normed - vmin)/(vmax - vmin))`,
// computed in T like NumPy does for an array of type T.  One byte per sampled cell leaves the
// device instead of sizeof(T) bytes per cell.
template <class T>
__global__ __launch_bounds__(256) void k_snapshot(const T *__restrict__ ez, unsigned char *__restrict__ out,
                                                  Geom g, int row_first, int nrows_out, int ncols_out,
                                                  int stride, T vmin, T vmax, T range)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (q >= ncols_out || r >= nrows_out) return;
    T v = ez[at(g, row_first + r * stride, q * stride)];
    v = v < vmin ? vmin : v;
    v = v > vmax ? vmax : v;
    const T y = ((v - vmin) / range) * T(256);
    int idx = (int)y;
    idx = (y == T(256)) ? 255 : idx;
    idx = idx < 0 ? 0 : (idx > 255 ? 255 : idx);
    out[(size_t)r * ncols_out + q] = (unsigned char)idx;
}

// ---- reductions over the owned rows: sum of squares and max |.| of one field ------------------
// Per-block partials (double) are written to `part`; the host adds them (blocks <= 1024).
template <class T>
__global__ __launch_bounds__(256) void k_reduce(const T *__restrict__ f, double *__restrict__ part,
                                                Geom g, int row_first, int nrows, int ncols)
{
    __shared__ double ssum[256], smax[256];
    double s = 0, m = 0;
    const size_t n = (size_t)nrows * ncols;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
        const int i = row_first + (int)(t / ncols), j = (int)(t % ncols);
        const double v = (double)f[at(g, i, j)];
        s += v * v;
        const double a = v < 0 ? -v : v;
        m = a > m ? a : m;
    }
    ssum[threadIdx.x] = s;
    smax[threadIdx.x] = m;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            ssum[threadIdx.x] += ssum[threadIdx.x + w];
            smax[threadIdx.x] = smax[threadIdx.x] > smax[threadIdx.x + w] ? smax[threadIdx.x] : smax[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = ssum[0];
        part[2 * blockIdx.x + 1] = smax[0];
    }
}

// ---- host-layout <-> device-layout rows with a change of element type ----------------------------
// (fdtd2d_upload / download / set_materials when the host arrays are not of the engine's type:
// the conversion runs on the device, one rounding per element exactly as a NumPy astype)
template <class S, class D>
__global__ __launch_bounds__(256) void k_convert2d(const S *__restrict__ src, size_t src_pitch,
                                                   D *__restrict__ dst, size_t dst_pitch, int rows, int cols)
{
    const size_t n = (size_t)rows * cols;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
        const size_t i = t / cols, j = t - i * cols;
        dst[i * dst_pitch + j] = (D)src[i * src_pitch + j];
    }
}

// ---- min / max of a material array on the device (fdtd.py:25-26) --------------
// Per-block partials {min, max, number of elements that are not > 0 (NaN included)}; the host
// folds the <= 1024 partials.  min == max <=> the array is uniform.
template <class T>
__global__ __launch_bounds__(256) void k_minmax(const T *__restrict__ f, double *__restrict__ part,
                                                size_t pitch, int nrows, int ncols)
{
    __shared__ double smin[256], smax[256], sbad[256];
    double mn = 1e300, mx = -1e300, bad = 0;
    const size_t n = (size_t)nrows * ncols;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
        const size_t i = t / ncols, j = t - i * ncols;
        const double v = (double)f[i * pitch + j];
        if (!(v > 0)) bad += 1;
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
    smin[threadIdx.x] = mn;
    smax[threadIdx.x] = mx;
    sbad[threadIdx.x] = bad;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            smin[threadIdx.x] = smin[threadIdx.x] < smin[threadIdx.x + w] ? smin[threadIdx.x] : smin[threadIdx.x + w];
            smax[threadIdx.x] = smax[threadIdx.x] > smax[threadIdx.x + w] ? smax[threadIdx.x] : smax[threadIdx.x + w];
            sbad[threadIdx.x] += sbad[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[3 * blockIdx.x] = smin[0];
        part[3 * blockIdx.x + 1] = smax[0];
        part[3 * blockIdx.x + 2] = sbad[0];
    }
}

// ---- halo rows <-> contiguous message (3 fields x nrows x C) ---------------------------------
template <class T, bool PACK>
__global__ __launch_bounds__(256) void k_halo(T *__restrict__ f0, T *__restrict__ f1,
                                              T *__restrict__ f2, T *__restrict__ msg, Geom g,
                                              int row_first, int nrows)
{
    const size_t per = (size_t)nrows * g.C;
    const size_t n = 3 * per;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) {
        const int f = (int)(t / per);
        const size_t r = t % per;
        const int i = row_first + (int)(r / g.C), j = (int)(r % g.C);
        T *fld = f == 0 ? f0 : (f == 1 ? f1 : f2);
        if (PACK) msg[t] = fld[at(g, i, j)];
        else fld[at(g, i, j)] = msg[t];
    }
}

}  // namespace fdtd

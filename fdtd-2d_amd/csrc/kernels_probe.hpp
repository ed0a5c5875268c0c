// Point probe: the time series of Ez at one cell, one value per step, while the grid advances
// NT steps per launch.
//
// The intermediate time levels of a temporally blocked pass never reach HBM, and a store under
// a branch inside the pass kernels' tick loop would cost them their counted s_waitcnt.  So the
// probe does not touch those kernels: one extra workgroup recomputes the probe cell's own
// dependency cone from the pass's input buffers -- a (4 NT + 5)^2 tile in LDS, advanced NT
// steps with the same per-cell functions as the zone tiles (MurRules / the plain update), the
// centre value written out after every step.  A tile edge that is not a grid edge goes stale at
// most two cells per step (the rate inside the Mur bands), the margin is 2 NT + 2.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_stream.hpp"

namespace fdtd {

struct ProbeParams {
    int row, col;          // probe cell
    int r_lo, r_hi;        // rows that are current in the input buffers (tile rows are clipped to them)
    double *out;           // device buffer, one double per step
    long long base, cap;   // index of this pass's first step, capacity
};

template <class T, int NT, bool CE_ARR, bool CH_ARR>
__global__ __launch_bounds__(256) void k_probe(const PassParams<T> p, const ProbeParams q)
{
    constexpr int M = 2 * NT + 2, W = 2 * M + 1, WP = W + 1, ZS = W * WP;
    __shared__ T smem[4 * ZS];          // Ez (two buffers), Hx, Hy -- addressed by offset only
    T *const sX = smem + 2 * ZS;
    T *const sY = smem + 3 * ZS;
    const Geom g = p.g;
    const int z0 = max(q.r_lo, q.row - M), z1 = min(q.r_hi, q.row + M + 1);
    const int c0 = max(0, q.col - M), c1 = min(g.C, q.col + M + 1);
    const int nr = z1 - z0, nc = c1 - c0, cells = nr * nc;
    for (int n = threadIdx.x; n < cells; n += 256) {
        const int li = n / nc, lj = n - li * nc;
        const size_t o = at(g, z0 + li, c0 + lj);
        const int s = li * WP + lj;
        smem[s] = p.ez_in[o];
        sX[s] = p.hx_in[o];
        sY[s] = p.hy_in[o];
    }
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int step = 1; step <= p.nlev; ++step) {
        const T *Eo = smem + cur * ZS;
        T *En = smem + (cur ^ 1) * ZS;
        for (int n = threadIdx.x; n < cells; n += 256) {          // H half-step (main.py:66-76)
            const int li = n / nc, lj = n - li * nc, i = z0 + li, j = c0 + lj;
            if (i <= g.R - 2 && j <= g.C - 2 && li + 1 < nr && lj + 1 < nc) {
                const int s = li * WP + lj;
                const T ch = CH_ARR ? p.ch[at(g, i, j)] : p.ch_u;
                const T e = Eo[s];
                sX[s] = sX[s] - ch * (Eo[s + WP] - e);
                sY[s] = sY[s] + ch * (Eo[s + 1] - e);
            }
        }
        __syncthreads();
        MurRules<T, TileAcc<T, CE_ARR>> rules{{Eo, sX, sY, p.ce, p.ce_u, g, g.R, g.C, z0, c0, WP, nr, nc}, p.k};
        for (int n = threadIdx.x; n < cells; n += 256) {          // E half-step, stages A-D per cell
            const int li = n / nc, lj = n - li * nc, i = z0 + li, j = c0 + lj;
            const int s = li * WP + lj;
            T val = Eo[s];
            if ((li >= 1 || i == 0) && (lj >= 1 || j == 0)) {
                if (i < 5 || i >= g.R - 5 || j < 5 || j >= g.C - 5) {
                    val = rules.d(i, j);
                } else {
                    const T ce = CE_ARR ? p.ce[at(g, i, j)] : p.ce_u;
                    val = val + ((sY[s] - sY[s - 1]) - (sX[s] - sX[s - WP])) * ce;
                }
            }
            if (i >= p.src_row && i < p.src_row1 && j >= p.src_col && j < p.src_col1)
                val = (T)((double)val + p.amp[step - 1]);
            En[s] = val;
        }
        __syncthreads();
        cur ^= 1;
        if (threadIdx.x == 0 && q.base + step - 1 < q.cap)
            q.out[q.base + step - 1] = (double)(smem + cur * ZS)[(q.row - z0) * WP + (q.col - c0)];
    }
}

// after a single step of the half-step kernels: copy the cell
template <class T> __global__ void k_probe_copy(const T *ez, size_t off, double *out, long long idx)
{
    out[idx] = (double)ez[off];
}

// one sample of the running Fourier transform: acc_re[k][cell] += Ez * c[k], acc_im[k][cell] += Ez * s[k]
struct DftPhasors {
    double c[16], s[16];
};
template <class T>
__global__ __launch_bounds__(256) void k_dft(const T *__restrict__ ez, Geom g, int row_lo, int nrows, int col0, int ncols,
                                              int nfreq, DftPhasors ph, double *__restrict__ acc)
{
    const size_t cells = (size_t)nrows * ncols, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x; n < cells; n += stride) {
        const int i = row_lo + (int)(n / ncols), j = col0 + (int)(n % ncols);
        const double e = (double)ez[at(g, i, j)];
        for (int k = 0; k < nfreq; ++k) {
            acc[(size_t)(2 * k) * cells + n] += e * ph.c[k];
            acc[(size_t)(2 * k + 1) * cells + n] += e * ph.s[k];
        }
    }
}

}  // namespace fdtd

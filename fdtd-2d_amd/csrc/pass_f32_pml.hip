// float32 16-step passes with the split-field PML: host side + kernel instantiations.
#include "engine.hpp"

namespace fdtd_host {

template <bool CE_ARR> static int launch_pml_split(fdtd2d *h, fdtd::PassParams<float> &p)
{
    using T = float;
    constexpr int NT = 16, NW = 4, V = 4, SW = 64 * V, HC = fdtd::stream_hc(NT), OW = SW - 2 * HC;
    const int L = h->pml_L, R = h->rows, C = h->cols, ns = p.nstrips;
    const int region = std::max(0, p.band_hi - p.band_lo);
    if (region == 0) return 0;
    auto x0_of = [&](int s) {
        int x = s * OW - HC;
        if (s == ns - 1) x = std::min(x, (C - SW + 3) & ~3);
        return x;
    };
    // strips whose columns reach into the layer (or the grid edge) at either end
    int n_left = 0, n_right = 0;
    while (n_left < ns && x0_of(n_left) < L + 1) ++n_left;
    while (n_right < ns - n_left && x0_of(ns - 1 - n_right) + SW > C - 1 - L) ++n_right;
    const int inner = ns - n_left - n_right;
    // Rows the plain kernel must stay clear of: a band [ra, rb) computes level t >= 1 on the rows >= ra - NT (level 1
    // reaches furthest up; rows above that only enter as level-0 data), and every row it COMPUTES must be outside the row
    // layers, i.e. >= L -- so its first band may start at L + NT, and the layer kernel's top task holds L + NT rows (56
    // for L = 40; until round 3 this was the whole 16-step cone of the output rows, L + 1 + 2 NT -> 80 rows: 131 ticks
    // per task instead of 107).  Mirror image at the bottom.
    const int reach = L + NT;
    const int re = std::max(16, h->shape_now.edge_rows > 0 ? h->shape_now.edge_rows : h->pml_layer_rows);
    auto up8 = [&](int x) { return (x + 7) / 8 * 8; };
    int a_hi = p.band_lo, c_lo = p.band_hi;
    if (inner > 0) {
        if (h->top()) a_hi = std::min(p.band_hi, p.band_lo + up8(std::max(0, reach - p.band_lo)));
        if (h->bottom()) c_lo = std::max(a_hi, p.band_hi - up8(std::max(0, p.band_hi - (R - reach))));
    }
    // top / bottom tasks: one band per end (taller ends -- a slab whose halo starts far above the reach -- in
    // bands of at most 128 rows)
    const int rtb = std::max(8, std::min(128, std::max(a_hi - p.band_lo, p.band_hi - c_lo)));
    fdtd::PmlSplit<T> q{pml_factors<T>(h), (const T *)h->ezxb[h->hcur], (T *)h->ezxb[h->hcur ^ 1],
                        n_left, n_right, re, rtb, a_hi, c_lo, (region + re - 1) / re,
                        (a_hi - p.band_lo + rtb - 1) / rtb, (p.band_hi - c_lo + rtb - 1) / rtb};
    const long long layer_blocks = (long long)(n_left + n_right) * q.n_all + (long long)inner * (q.n_top + q.n_bot);
    // the plain kernel: inner strips x the rows between the layer bands, no edge strips, no zones
    fdtd::PassParams<T> pp = p;
    pp.band_lo = a_hi;
    pp.band_hi = c_lo;
    pp.strip_first = n_left;
    pp.nbands_e = 0;
    pp.band_rows_e = pp.band_rows;
    pp.nbands = (std::max(0, c_lo - a_hi) + pp.band_rows - 1) / pp.band_rows;
    pp.n_inner = inner;
    pp.band_rows2 = pp.nbands2 = 0;
    pp.split_row = c_lo;
    long long plain_blocks = (long long)pp.nbands * inner;
    // the plain kernel's tasks dealt out XCD by XCD (FDTD2D_OPT_XCD_MAP / the tuner's choice), as in launch_pass_impl
    if ((h->xcd_map >= 0 ? h->xcd_map != 0 : h->shape_now.xcd != 0) && plain_blocks > 0) {
        pp.xcd_map = 1;
        pp.main_tasks = (int)plain_blocks;
        pp.main_per = (pp.main_tasks + 7) / 8;
        pp.main_pad = 0;
        plain_blocks = 8LL * pp.main_per;
    }
    // side by side on two streams where the piece is large; the 16-row pieces next to a slab's cuts run their two
    // kernels one after the other on the handle's stream (a fork / join pair of events costs more than they take)
    const bool both = layer_blocks > 0 && plain_blocks > 0 && region >= 256;
    if (both) {
        HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
    }
    if (layer_blocks > 0) {          // the slower workgroups: started first, beside the plain kernel
        hipLaunchKernelGGL((fdtd::k_bulk_split_pml<T, NT, NW, CE_ARR, V>), dim3((unsigned)layer_blocks), dim3(64 * NW), 0,
                           both ? h->side_stream : h->stream, p, q);
        HIPCHK(h, hipGetLastError());
    }
    if (both) HIPCHK(h, hipEventRecord(h->ev_join, h->side_stream));
    if (plain_blocks > 0) {
        hipLaunchKernelGGL((fdtd::k_bulk_split<T, NT, NW, false, CE_ARR, false, V>), dim3((unsigned)plain_blocks), dim3(64 * NW), 0,
                           h->stream, pp);
        HIPCHK(h, hipGetLastError());
    }
    if (both) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
    h->pass_launches++;
    return 0;
}

int launch_pml_split_f32(fdtd2d *h, fdtd::PassParams<float> &p)
{
    return h->ce_uniform ? launch_pml_split<false>(h, p) : launch_pml_split<true>(h, p);
}

}  // namespace fdtd_host

// Shared between the translation units of libfdtd2d.so: the handle, error helpers and the
// entry point of the temporally blocked pass (defined per element type in pass_f32.hip /
// pass_f64.hip so that the heavy kernel instantiations compile in parallel).
#pragma once
#include "../../include/fdtd2d.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <map>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "kernels_step.hpp"
#include "kernels_stream.hpp"
#include "kernels_pml.hpp"
#include "kernels_split.hpp"
#include "kernels_pml_split.hpp"
#include "kernels_probe.hpp"

using fdtd::Geom;

struct fdtd2d_slab;
struct Range {
    int lo, hi;
};

struct fdtd2d {
    int rows = 0, cols = 0;      // global grid
    int row0 = 0, nrows = 0;     // owned rows
    int halo = 0;                // halo rows kept on each side (storage is symmetric)
    int dtype = FDTD2D_F32, boundary = FDTD2D_BOUNDARY_MUR5, device = 0;
    double dt = 0, dx = 0;
    long long pitch = 0;         // elements per stored row
    int stored = 0;              // stored rows = nrows + 2*halo
    size_t esz = 4;              // element size
    size_t field_bytes = 0;      // bytes of one stored field (without guard)

    void *ez[2] = {nullptr, nullptr};
    int cur = 0;                 // ez[cur] is the current Ez
    void *hxb[2] = {nullptr, nullptr}, *hyb[2] = {nullptr, nullptr};
    int hcur = 0;                // hxb[hcur], hyb[hcur] are the current Hx, Hy
    void *ce = nullptr, *ch = nullptr;    // coefficient arrays (nullptr when uniform)
    void *ezxb[2] = {nullptr, nullptr};   // PML only: the x-part of the split Ez (set follows hcur)
    void *pml = nullptr;                  // PML only: 4 row + 4 column factor arrays, back to back
    bool have_pml = false;
    bool pml_unit_outside = false;        // the factor arrays equal 1 outside the layer (checked by fdtd2d_set_pml)
    int pml_L = 0;
    int pml_short_rows = 16;     // band height of the layer waves (8-step k_pass_pml)
    int pml_layer_rows = 64;     // band height of the layer workgroups (16-step k_bulk_split_pml)
    int nfields() const { return boundary == FDTD2D_BOUNDARY_PML ? 4 : 3; }
    bool have_mat = false, ce_uniform = true, ch_uniform = true;
    double ce_u = 0, ch_u = 0;   // uniform coefficients, already rounded to T
    double k_mur = 0;            // Mur factor, already rounded to T
    double eps_min = 0, mu_min = 0;

    Range ev{0, 0}, hv{0, 0};    // global rows on which Ez / (Hx,Hy) are current
    long long step = 0;
    long long pass_launches = 0, step_launches = 0;
    // point probe (fdtd2d_set_probe): Ez[probe_row, probe_col] after every step since probe_step0
    int probe_row = 0, probe_col = 0;
    long long probe_cap = 0, probe_step0 = 0;
    double *probe_dev = nullptr;
    bool probe_pending = false;       // the next launch_pass records this pass's steps
    // running Fourier transform (fdtd2d_set_dft): window, frequencies, accumulators re[k][cell], im[k][cell]
    int dft_row0 = 0, dft_col0 = 0, dft_rows = 0, dft_cols = 0, dft_n = 0, dft_every = 0;
    double dft_omega[16] = {};
    double *dft_acc = nullptr;
    long long dft_step0 = 0;
    int dft_lo() const { return std::max(dft_row0, row0); }                               // owned window rows
    int dft_hi() const { return std::min(dft_row0 + dft_rows, row0 + nrows); }
    // steps until the next sampled step (a large number without a transform)
    int dft_gap() const { return dft_n ? dft_every - (int)((step - dft_step0) % dft_every) : (1 << 30); }
    int src_rows = 1, src_cols = 1;   // extent of the source: (row, col) of run/add_point is its first cell
    // a pass issued in pieces (fdtd2d_pass_rows) and not yet committed
    int pend_nt = 0;
    std::vector<Range> pend_done;

    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    hipStream_t side_stream = nullptr;   // zone tiles run here, concurrently with the bulk
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void *trash = nullptr;       // destination of masked-off stores (one 4 KiB slot per workgroup index mod 1024)
    void *scratch = nullptr;     // device scratch for snapshots / reduction partials
    size_t scratch_bytes = 0;
    std::string err;
    struct fdtd2d_slab *slab = nullptr;   // state of the row-slab run loop (slab_loop.hip), owned by the handle
#ifdef FDTD2D_TRACE
    void *trace_dev = nullptr;   // 4 x u64 per workgroup of the last level-split launch (profiling build)
    long long trace_blocks = 0;
#endif

    bool top() const { return row0 == 0; }
    bool bottom() const { return row0 + nrows == rows; }
    int row_base() const { return row0 - halo; }
    int store_lo() const { return std::max(0, row0 - halo); }
    int store_hi() const { return std::min(rows, row0 + nrows + halo); }
    Geom geom() const { return Geom{rows, cols, row_base(), pitch}; }
    void *ezx() const { return ezxb[hcur]; }
    void *hx() const { return hxb[hcur]; }
    void *hy() const { return hyb[hcur]; }
    int stream_band_rows = 0;    // 0 = heuristic (FDTD2D_OPT_BAND_ROWS)
    int level_split = -1;        // k_bulk_split for 8-step passes: -1 / 1 = yes (measured faster than
                                 // k_bulk at every size, float32 and float64:
                                 // profiles/r01_split8_vs_bulk.txt), 0 = k_bulk
    // 20-step passes (4 waves x 5 levels with one row of HBM prefetch: the same 8 register slots and
    // 4 waves per SIMD as the 16-step form; zones as k_zone beside the bulk): float32 + Mur frame only,
    // whole grids (a slab's 16-row halo does not feed them), no probe tile.  One such pass takes
    // 17..20 remaining steps in ONE sweep; it pays where the sweep dominates -- us per run(20) as one
    // 20-step pass vs a 16- and a 4-step pass: 16384^2 1827 vs 2914, 8192^2 870 vs 942, 4096^2 317 vs
    // 250 (profiles/r02_long_passes.txt); with the staged level body and a source in the run: 8192^2 617 vs
    // 626, 6144^2 407 vs 494, 4096^2 249 vs 268, 2048^2 193 vs 127 (profiles/r02_nt20_small.txt) -- hence the
    // size rule of 16 Mi cells (FDTD2D_OPT_MAX_PASS_STEPS = 20 lifts it, as 16 does for the 16-step rule).
    bool long_passes() const
    {
        if (dtype != FDTD2D_F32 || boundary != FDTD2D_BOUNDARY_MUR5 || max_nt < 20 || probe_cap) return false;
        return max_nt_forced || (size_t)nrows * cols >= (size_t)16 << 20;
    }
    // 16-step PML passes: the level-split pair k_bulk_split / k_bulk_split_pml (float32, uniform mu)
    bool pml_split(int nt) const
    {
        return boundary == FDTD2D_BOUNDARY_PML && nt == 16 && dtype == FDTD2D_F32 && ch_uniform && have_pml &&
               pml_unit_outside;
    }
    bool use_level_split(int nt, int band_lo, int band_hi) const
    {
        (void)band_lo, (void)band_hi;
        if ((nt != 8 && nt != 16 && nt != 20) || boundary != FDTD2D_BOUNDARY_MUR5) return false;
        // 16- and 20-step passes exist in this form only; 20 steps in float32 only; float64's 16-step kernel has no
        // probe tile (its LDS tile would not fit) -- such handles stay with 8-step passes
        if (nt > 16) return dtype == FDTD2D_F32;
        if (nt == 16) return dtype == FDTD2D_F32 || !probe_cap;
        // array materials: only the build with fused zone tiles exists
        if ((!ce_uniform || !ch_uniform) && zone_split == 1) return false;
        return level_split != 0;
    }
    // Launch shapes measured on this GPU for (pass length, first row, last row) of large passes:
    // band height (0 = the rule in launch_pass) and waves per strip (0 = the rule below).  Filled
    // by tune_pass() with uncommitted trial launches; results never depend on it.
    struct Shape {
        int band_rows, waves;
        int edge_rows = 0;       // band height of the first / last strip (0 = band_rows): their body
                                 // is ~2x slower per row, equal heights make them the tail of a launch
        int side = 1;            // waves side by side per level group (1, 2 or 4; kernels_stream.hpp, strip_x0)
        int xcd = 0;             // 1: tasks dealt out XCD by XCD (FDTD2D_OPT_XCD_MAP), chosen per shape by the tuner
        // One-round launches (every workgroup resident from the start): the zone tiles are done after a fraction of
        // the launch and their slots would idle.  The last n_short bands of every inner strip are `short_rows` tall
        // and come last in launch order: they start in the slots the zone tiles free and end with the tall bands.
        int short_rows = 0, n_short = 0;
        // float32 20-step passes, one wave per level group: zone tiles as workgroups of the bulk launch (1) instead of
        // k_zone on the side stream (0) -- the tuner measures both (16384^2 run(20) 1.74 -> 1.68 ms, 4096^2 0.222 -> 0.199,
        // 8192^2 0.508 -> 0.543 with the shapes tuned for the side stream: profiles/r03_zone20_fused.txt)
        int fuse = 0;
    };
    // strips of several waves side by side exist for the float32 16- and 20-step level-split kernels with 4 waves
    // per level group, on grids wide enough for a few of them
    bool side_ok(int nt, int sd) const
    {
        if (sd == 1) return true;
        // (4 waves side by side = 1024 threads, at most 128 VGPRs each: only without coefficient rows in the slots)
        // (float64: 2 columns per lane, strips of 128 / 248 / 488 columns: the overlap costs 25 / 13 / 7 % there)
        const int w = dtype == FDTD2D_F32 ? 256 : 128;
        return (sd == 2 || (sd == 4 && ce_uniform && ch_uniform)) && boundary == FDTD2D_BOUNDARY_MUR5 &&
               (nt == 16 || (nt == 20 && dtype == FDTD2D_F32)) && cols >= 8 * w * sd;
    }
    std::map<std::array<int, 3>, Shape> tuned;
    int autotune = 1;            // FDTD2D_OPT_AUTOTUNE
    // fdtd2d_set_shape: launch shapes given by the caller (measured elsewhere, e.g. by another process), by
    // pass length; key 0 = the full-length passes (cycle_steps())
    std::map<int, Shape> given_shape;
    const Shape *shape_given(int nt) const
    {
        auto it = given_shape.find(nt);
        if (it == given_shape.end() && nt == cycle_steps()) it = given_shape.find(0);
        return it != given_shape.end() && it->second.band_rows > 0 ? &it->second : nullptr;
    }
    Shape shape_now{0, 0};       // shape of the launch being issued (set by launch_pass)
    Shape shape_last{0, 0};      // band height / waves per strip actually used by the last pass
    int last_nt = 0;             // its kernel length
    int split_waves = 0;         // waves per strip in k_bulk_split: 0 = automatic, 4 or 8
    int split_waves_for(int nt, int lo, int hi) const
    {
        if (nt > 16) return 4;                   // 20 steps: 4 waves x 5 levels
        if (!split_waves && shape_now.waves) return shape_now.waves;
        // 4 waves x NT/4 levels on large slabs (what the tuner keeps picking from 4096^2 up); 8 waves
        // x NT/8 levels shorten the tick chain that bounds small ones: 8-step passes 27.8 vs 33.7 us
        // at 512^2, 31 vs 37 at 1024^2, equal at 2048^2, 63 vs 49 at 3072^2; 16-step passes 38 vs 56
        // at 2048^2, 57 vs 59 at 3072^2, 87 vs 78 at 4096^2 (profiles/r01_split_waves_sweep.txt)
        if (split_waves) return split_waves;
        const size_t cells = (size_t)std::max(0, hi - lo) * cols;
        return cells < (nt == 16 ? (size_t)12 << 20 : (size_t)3 << 20) ? 8 : 4;
    }
    bool max_nt_forced = false;  // set_option(MAX_PASS_STEPS): no size rule for 16-step passes
    int cycle_steps() const      // longest pass this configuration runs
    {
        // 16-step passes (k_bulk_split<16>) halve the HBM traffic per step but have twice the
        // fill/drain latency: measured faster from 4096^2 up, slower up to 3072^2
        // (profiles/r01_nt16_sweep.txt)
        const bool big = (size_t)nrows * cols >= (size_t)12 << 20;
        // (PML: only when the 16-step pair can really run -- factor arrays set and equal to 1 outside the
        // layer, uniform mu, no probe: callers use this figure as their exchange cycle and then ask
        // fdtd2d_pass_rows for passes of exactly that length)
        // (float64, round 3: the same level-split kernel with 2 columns per lane; its zone tiles need 79 KB of LDS and
        // run as k_zone beside the bulk; no probe tile)
        if (max_nt >= 16 && (big || max_nt_forced) && have_mat &&
            ((boundary == FDTD2D_BOUNDARY_MUR5 && (dtype == FDTD2D_F32 || !probe_cap)) ||
             (dtype == FDTD2D_F32 && pml_split(16) && !probe_cap)))
            return 16;
        return std::min(max_nt, 8);
    }
    int xcd_map = -1;            // FDTD2D_OPT_XCD_MAP: -1 = the tuner's choice per shape, 0 / 1 forced
    int side_waves = 0;          // FDTD2D_OPT_SIDE_WAVES: 0 = the tuner's choice, 1 / 2 / 4 forced
    unsigned long long *clk_dev = nullptr;   // clock probe stamps (fdtd2d_clock_probe_*)
    hipStream_t clk_stream = nullptr;
    int zone_split = -1;         // -1: by launch size; 0/1: force fused / side-stream zones (FDTD2D_OPT_ZONE_SPLIT)
    int max_nt = 20;             // longest pass; FDTD2D_OPT_MAX_PASS_STEPS (0: step kernels only)
};


namespace fdtd_host {

extern thread_local std::string g_create_error;

inline int fail(fdtd2d *h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIPCHK(h, expr)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fdtd_host::fail((h), -(1000 + (int)e_), "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <class T> fdtd::PmlFactors<T> pml_factors(const fdtd2d *h)
{
    const T *b = (const T *)h->pml;
    const size_t R = h->rows, C = h->cols;
    return fdtd::PmlFactors<T>{b, b + R, b + 2 * R, b + 3 * R, b + 4 * R, b + 4 * R + C, b + 4 * R + 2 * C,
                               b + 4 * R + 3 * C, h->pml_L, h->rows, h->cols};
}


// Can a pass of nt steps run from the current state?  Fills the row range of the bulk.
bool pass_geometry(const fdtd2d *h, int nt, int *band_lo, int *band_hi);

// One pass of nt steps (or the rows [band_lo, band_hi) of it); see pass_impl.hpp.
template <class T>
int launch_pass(fdtd2d *h, int nt, int band_lo, int band_hi, int src_row, int src_col,
                const double *amps, bool ztop, bool zbot, bool commit, int full_lo, int full_hi,
                int nlev = 0);
// the 16-step PML pass (pass_f32_pml.hip): k_bulk_split on the cells clear of the layer, k_bulk_split_pml on the rest
int launch_pml_split_f32(fdtd2d *h, fdtd::PassParams<float> &p);
extern template int launch_pass<float>(fdtd2d *, int, int, int, int, int, const double *, bool, bool, bool, int, int, int);
extern template int launch_pass<double>(fdtd2d *, int, int, int, int, int, const double *, bool, bool, bool, int, int, int);

}  // namespace fdtd_host

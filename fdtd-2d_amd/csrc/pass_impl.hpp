// Host side of the temporally blocked pass, instantiated once per element type
// (pass_f32.hip, pass_f64.hip).
#pragma once
#include "engine.hpp"

namespace fdtd_host {

// geometry of the strips of a pass: columns held, columns written
inline int pass_strip_ow(int nt, int v, int nw, int sd)
{
    return fdtd::strip_width(nt / nw, v, sd) - 2 * fdtd::stream_hc(nt);
}

template <class T, int NT, bool CE_ARR, bool CH_ARR, int V = fdtd::Vec<T>::N>
int launch_pass_impl(fdtd2d *h, fdtd::PassParams<T> &p)
{
    using D = fdtd::ZoneDims<NT>;
    const int sd = h->shape_now.side > 1 ? h->shape_now.side : 1;
    const int region = std::max(0, p.band_hi - p.band_lo);
    p.nbands = (region + p.band_rows - 1) / p.band_rows;
    p.nbands_e = (region + p.band_rows_e - 1) / p.band_rows_e;
    p.band_rows2 = p.nbands2 = 0;
    p.split_row = p.band_hi;
    // inner strips that hold source columns (at most two neighbours; wider sources get no special bands)
    p.src_strip = 0;
    p.n_src = 0;
    p.band_rows_s = std::max(16, std::min(p.band_rows_e, p.band_rows / 3));
    {   // short bands only on the rows whose workgroups can see the source: a band [ra, rb) runs the general body iff
        // ra - 2 NT < src_row1 and rb + NT > src_row, so the bands above src_row - NT and below src_row1 + 2 NT are plain
        auto clampi = [](int v, int a, int b) { return std::min(std::max(v, a), b); };
        p.src_lo = clampi(p.src_row - NT, p.band_lo, p.band_hi);
        p.src_hi = clampi(p.src_row1 + 2 * NT, p.src_lo, p.band_hi);
        p.nsrc_top = (p.src_lo - p.band_lo + p.band_rows - 1) / p.band_rows;
        p.nsrc_mid = (p.src_hi - p.src_lo + p.band_rows_s - 1) / p.band_rows_s;
        p.nbands_s = p.nsrc_top + p.nsrc_mid + (p.band_hi - p.src_hi + p.band_rows - 1) / p.band_rows;
    }
    if (p.src_col1 > p.src_col && p.band_rows_s < p.band_rows && p.nstrips > 2) {
        const int SWc = sd > 1 ? fdtd::strip_width(NT / 4, V, sd) : 64 * V, OWc = SWc - 2 * fdtd::stream_hc(NT);
        int s0 = -1, s1 = -1;
        for (int st = 1; st <= p.nstrips - 2; ++st) {
            const int x0 = st * OWc - fdtd::stream_hc(NT);
            if (p.src_col1 > x0 && p.src_col < x0 + SWc) {
                if (s0 < 0) s0 = st;
                s1 = st;
            }
        }
        if (s0 >= 0 && s1 - s0 + 1 <= 2) {
            p.src_strip = s0;
            p.n_src = s1 - s0 + 1;
        }
    }
    p.zone_tiles = (h->cols + D::WZ - 1) / D::WZ;
#ifdef FDTD2D_TUNE_LOG      // profiling build only: the launch WITHOUT its zone tiles (wrong results; what would cheaper tiles buy?)
    if (std::getenv("FDTD2D_DEBUG_NOZONES")) p.zone_top = p.zone_bot = 0;
#endif
    // launch order: zone tiles, 2 edge-strip slots (the second stays empty with one strip),
    // then strips 1 .. nstrips-2
    p.n_inner = std::max(0, p.nstrips - 2 - p.n_src);
    {   // filler bands (Shape::short_rows): the bottom rows of the inner strips in shorter bands that come last
        const fdtd2d::Shape &sh = h->shape_now;
        const bool xcd_now = h->xcd_map >= 0 ? h->xcd_map != 0 : sh.xcd != 0;
        if (sh.short_rows >= 8 && sh.n_short > 0 && !xcd_now && NT >= 8 && p.n_inner > 0 &&
            (long long)sh.short_rows * sh.n_short < region && h->use_level_split(NT, p.band_lo, p.band_hi)) {
            p.band_rows2 = sh.short_rows;
            p.nbands2 = sh.n_short;
            p.split_row = p.band_hi - sh.short_rows * sh.n_short;
            p.nbands = (p.split_row - p.band_lo + p.band_rows - 1) / p.band_rows;
        }
    }
    long long bulk = 2LL * p.nbands_e + (long long)p.n_src * p.nbands_s +
                     (long long)(p.nbands + p.nbands2) * p.n_inner;
    // 20-step passes (float32, one wave per level group): the register-resident zone tiles may ride in the bulk launch too
    // (Shape::fuse, the tuner's choice)
    const bool fuse_long = NT > 16 && sizeof(T) == 4 && sd == 1 && h->shape_now.fuse != 0;
    const int zone_tiles_all = (p.zone_top + p.zone_bot) * p.zone_tiles;
    // workgroups of the zone tiles: in the fused launch (64 x waves-per-strip threads) and as k_zone (256 threads)
    const int nw_now = NT >= 8 && h->use_level_split(NT, p.band_lo, p.band_hi) ? h->split_waves_for(NT, p.band_lo, p.band_hi) : 1;
    const int zones_fused = nw_now == 8 ? fdtd::zone_wgs_for<T, NT, 512>(zone_tiles_all)
                          : (nw_now == 4 ? fdtd::zone_wgs_for<T, NT, 256>(zone_tiles_all) : zone_tiles_all);
    const int zones_side = NT >= 8 && h->use_level_split(NT, p.band_lo, p.band_hi)
                               ? fdtd::zone_wgs_for<T, NT, fdtd::PASS_THREADS>(zone_tiles_all) : zone_tiles_all;
    const long long zones = zones_fused;
    p.zone_wgs = zones_fused;
    // Launches of several rounds: the zone workgroups -- short tasks since their tiles live in registers -- go LAST, into the
    // thinning tail of the launch, instead of holding slots of the first round: 16384^2 1270 -> 1232 us per 16-step pass,
    // eps + mu arrays 1798 -> 1764 (one process, alternating: profiles/r03_zone_last.txt).  One-round launches keep them
    // first (their slots are re-used by the filler bands).
    {
        const long long slots = 256LL * 16 / std::max(1, nw_now);
        p.zone_last = zones_fused > 0 && bulk * 100 > slots * 135 ? 1 : 0;
    }
    p.xcd_map = 0;
    p.main_pad = p.main_per = p.main_tasks = 0;
    const bool xcd = h->xcd_map >= 0 ? h->xcd_map != 0 : h->shape_now.xcd != 0;
    if (xcd && NT >= 8 && h->use_level_split(NT, p.band_lo, p.band_hi) && p.nstrips - 2 - p.n_src > 0) {
        // (the zone tiles ride in front of the bulk when fused: the pad makes the first inner-strip task a multiple
        // of 8 in the index the hardware sees)
        const bool side = zones > 0 && (h->zone_split == 1 || (NT > 16 && !fuse_long) || sd > 1);
        const long long front = ((side || p.zone_last) ? 0 : zones) + 2LL * p.nbands_e + (long long)p.n_src * p.nbands_s;
        p.xcd_map = 1;
        p.main_tasks = p.nbands * p.n_inner;
        p.main_per = (p.main_tasks + 7) / 8;
        p.main_pad = (int)((8 - front % 8) % 8);
        bulk = 2LL * p.nbands_e + (long long)p.n_src * p.nbands_s + p.main_pad + 8LL * p.main_per;
    }
    if (bulk + zones == 0) return 0;
    if constexpr (NT >= 8) {
        if (h->use_level_split(NT, p.band_lo, p.band_hi)) {     // 4 waves per (band, strip), 2 levels each
            // zone tiles: the first workgroups of the same launch (default: saves the side-stream
            // launch and two cross-stream event waits per pass -- 38 vs 84 us per 8 steps at
            // 2048^2, 96 vs 99 at 4096^2, equal at 16384^2: profiles/r01_zone_fuse_split.txt) or
            // k_zone on the side stream (zone_split = 1)
            // (20-step passes: 4 waves x 5 levels at 4 workgroups per CU; a fused 46-row LDS zone tile would
            // take 48 KB of LDS from every workgroup and leave 3, so their zones ran as k_zone only -- until the
            // register-resident tiles: float32 launches of one wave per level group may now fuse them, fuse_long)
            // (an LDS tile that does not fit a static allocation would have to run as k_zone with dynamic LDS, never fused:
            // float64 16-step passes until their tiles moved into the registers of four waves)
            constexpr bool big_tile = !fdtd::zone_in_registers<T, NT>() && (size_t)D::LDS_ELEMS * sizeof(T) > 65536;
            const bool side = zones > 0 && (h->zone_split == 1 || (NT > 16 && !fuse_long) || sd > 1 || big_tile);
            p.fused_zones = zones > 0 && !side;
            auto launch_side_zones = [&]() -> int {
                bool wide = false;
                if constexpr (NT > 16 && sizeof(T) == 4) {
                    // 128-column LDS tiles (dynamic LDS: 95 KB) on grids of 16384 columns and more (ZoneDims); below that
                    // the register-resident tiles of kernels_zone.hpp (round 3: run(20) at 8192^2 0.555 vs 0.585 ms,
                    // equal at 16384^2)
                    using DW = fdtd::ZoneDims<NT, true>;
                    if (h->cols >= 16384) {
                        wide = true;
                        constexpr size_t zone_dyn = (size_t)DW::LDS_ELEMS * sizeof(T);
                        // (every time: the attribute belongs to the device's code object, and a 20-step pass is a
                        // millisecond-scale launch)
                        HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&fdtd::k_zone<T, NT, CE_ARR, CH_ARR, true>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)zone_dyn));
                        p.zone_tiles = (h->cols + DW::WZ - 1) / DW::WZ;
                        const long long wzones = (long long)(p.zone_top + p.zone_bot) * p.zone_tiles;
                        hipLaunchKernelGGL((fdtd::k_zone<T, NT, CE_ARR, CH_ARR, true>), dim3((unsigned)wzones),
                                           dim3(fdtd::PASS_THREADS), zone_dyn, h->side_stream, p);
                    }
                }
                if (!wide) {
                    size_t zdyn = 0;
                    if constexpr (big_tile) {
                        zdyn = (size_t)D::LDS_ELEMS * sizeof(T);
                        HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&fdtd::k_zone<T, NT, CE_ARR, CH_ARR>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)zdyn));
                    }
                    hipLaunchKernelGGL((fdtd::k_zone<T, NT, CE_ARR, CH_ARR>), dim3((unsigned)zones_side),
                                       dim3(fdtd::PASS_THREADS), zdyn, h->side_stream, p);
                }
                HIPCHK(h, hipGetLastError());
                HIPCHK(h, hipEventRecord(h->ev_join, h->side_stream));
                return 0;
            };
            // Register-resident tiles are short tasks: their kernel is enqueued BEHIND the bulk kernel, so that they fill in as
            // slots free up instead of taking slots of the bulk's first round -- run(20) at 8192^2 0.556 -> 0.505 ms; the
            // 128-column LDS tiles of wide grids stay in front (16384^2: 1.636 either way; register tiles there 1.68)
            // (one process, alternating: profiles/r03_zone_last.txt).
            const bool wide_tiles = NT > 16 && sizeof(T) == 4 && h->cols >= 16384;
            const bool zones_after = side && fdtd::zone_in_registers<T, NT>() && !wide_tiles;
            if (side) {
                HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
                HIPCHK(h, hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
            }
            if (side && !zones_after) {
                if (int rcz = launch_side_zones()) return rcz;
            }
            const long long blocks = bulk + (p.fused_zones ? zones : 0);
#ifdef FDTD2D_TRACE
            h->trace_blocks = std::min<long long>(blocks, 1 << 16);
#endif
            if (blocks > 0) {
                // (array materials: one more row per slot and hand-off for each coefficient array)
                const int nw = h->split_waves_for(NT, p.band_lo, p.band_hi);
                const dim3 grid((unsigned)blocks), wg(64 * nw);
                if (sd > 1) {
                    // SD waves side by side per level group (4 waves x NT / 4 levels each): 64 * 4 * SD threads, the
                    // joint hand-off rows in dynamic LDS
                    if constexpr (NT == 16 || (NT == 20 && sizeof(T) == 4)) {
                        constexpr int NFc = 3 + (CE_ARR ? 1 : 0) + (CH_ARR ? 1 : 0);
                        auto go = [&](auto sdc) -> int {
                            constexpr int SDc = decltype(sdc)::value;
                            constexpr size_t dyn = (size_t)3 * fdtd::HAND_DEPTH * NFc * fdtd::side_units(NT / 4, V, SDc) * sizeof(fdtd::VecN<T, V>);
                            auto kern = &fdtd::k_bulk_split<T, NT, 4, false, CE_ARR, CH_ARR, V, SDc>;
                            HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
                            hipLaunchKernelGGL(kern, grid, dim3(64 * 4 * SDc), dyn, h->stream, p);
                            return 0;
                        };
                        int rc2;
                        if (sd == 2) rc2 = go(std::integral_constant<int, 2>{});
                        else if constexpr (!CE_ARR && !CH_ARR) rc2 = go(std::integral_constant<int, 4>{});
                        else return fail(h, FDTD2D_E_ARG, "4 waves side by side: uniform materials only");
                        if (rc2) return rc2;
                    } else {
                        return fail(h, FDTD2D_E_ARG, "strips of several waves side by side: 16-step passes and float32 20-step passes only");
                    }
                } else if constexpr (NT > 16) {
                    // 20 steps: 4 waves x 5 levels; zone tiles on the side stream or (float32, Shape::fuse) in this launch
                    bool launched = false;
                    if constexpr (sizeof(T) == 4) {
                        if (p.fused_zones) {
                            hipLaunchKernelGGL((fdtd::k_bulk_split<T, NT, 4, true, CE_ARR, CH_ARR, V>), grid, wg, 0, h->stream, p);
                            launched = true;
                        }
                    }
                    if (!launched) hipLaunchKernelGGL((fdtd::k_bulk_split<T, NT, 4, false, CE_ARR, CH_ARR, V>), grid, wg, 0, h->stream, p);
                } else {
                const bool w8 = nw == 8;
                // (a piece without zone tiles can run on either build; 8-step passes over array
                // materials only have the fused one)
                if (p.fused_zones || (!side && NT == 8 && (CE_ARR || CH_ARR))) {
                    if constexpr (big_tile) {
                        return fail(h, FDTD2D_E_STATE, "float64 16-step passes take their zone tiles from k_zone");
                    } else {
                    if (w8) hipLaunchKernelGGL((fdtd::k_bulk_split<T, NT, 8, true, CE_ARR, CH_ARR, V>), grid, wg, 0, h->stream, p);
                    else hipLaunchKernelGGL((fdtd::k_bulk_split<T, NT, 4, true, CE_ARR, CH_ARR, V>), grid, wg, 0, h->stream, p);
                    }
                } else if constexpr (NT == 16 || (!CE_ARR && !CH_ARR)) {
                    if (w8) hipLaunchKernelGGL((fdtd::k_bulk_split<T, NT, 8, false, CE_ARR, CH_ARR, V>), grid, wg, 0, h->stream, p);
                    else hipLaunchKernelGGL((fdtd::k_bulk_split<T, NT, 4, false, CE_ARR, CH_ARR, V>), grid, wg, 0, h->stream, p);
                } else {
                    return fail(h, FDTD2D_E_STATE, "8-step level-split passes over array materials are built with fused zones only");
                }
                }
                HIPCHK(h, hipGetLastError());
            }
            if (side && zones_after) {
                if (int rcz = launch_side_zones()) return rcz;
            }
            if (side) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
            h->pass_launches++;
            return 0;
        }
    }
    if constexpr (NT > 8) {
        return fail(h, FDTD2D_E_ARG, "16- and 20-step passes run on the level-split kernel only");
    } else {
    // Small launches: zone tiles as k_zone on the side stream (ordered behind what is already
    // on h->stream; everything later on h->stream waits for both).  Large launches: fused.
    const bool split = zones > 0 && (h->zone_split < 0 ? bulk < 1600 : h->zone_split != 0);
    p.fused_zones = zones > 0 && !split;
    if (split) {
        HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
        hipLaunchKernelGGL((fdtd::k_zone<T, NT, CE_ARR, CH_ARR>), dim3((unsigned)zones),
                           dim3(fdtd::PASS_THREADS), 0, h->side_stream, p);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipEventRecord(h->ev_join, h->side_stream));
    }
    const long long blocks = bulk + (p.fused_zones ? zones : 0);
    if (blocks > 0) {
        hipLaunchKernelGGL((fdtd::k_bulk<T, NT, CE_ARR, CH_ARR, V>), dim3((unsigned)blocks), dim3(64), 0,
                           h->stream, p);
        HIPCHK(h, hipGetLastError());
    }
    if (split) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
    h->pass_launches++;
    return 0;
    }
}

template <class T, int NT> int launch_pass_nt(fdtd2d *h, fdtd::PassParams<T> &p)
{
    if (h->ce_uniform && h->ch_uniform) return launch_pass_impl<T, NT, false, false>(h, p);
    if (!h->ce_uniform && h->ch_uniform) return launch_pass_impl<T, NT, true, false>(h, p);
    if (h->ce_uniform && !h->ch_uniform) return launch_pass_impl<T, NT, false, true>(h, p);
    return launch_pass_impl<T, NT, true, true>(h, p);
}

template <class T, int NT> int launch_probe_nt(fdtd2d *h, const fdtd::PassParams<T> &p, const fdtd::ProbeParams &q)
{
    const dim3 one(1), wg(256);
    if (h->ce_uniform && h->ch_uniform) hipLaunchKernelGGL((fdtd::k_probe<T, NT, false, false>), one, wg, 0, h->stream, p, q);
    else if (!h->ce_uniform && h->ch_uniform) hipLaunchKernelGGL((fdtd::k_probe<T, NT, true, false>), one, wg, 0, h->stream, p, q);
    else if (h->ce_uniform) hipLaunchKernelGGL((fdtd::k_probe<T, NT, false, true>), one, wg, 0, h->stream, p, q);
    else hipLaunchKernelGGL((fdtd::k_probe<T, NT, true, true>), one, wg, 0, h->stream, p, q);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// Records the nt steps of the pass described by p, if this handle's current rows hold the
// probe cell's output (on a slab: the owner and the ranks that keep the row as a halo).
template <class T> int launch_probe(fdtd2d *h, int nt, const fdtd::PassParams<T> &p)
{
    const int a = std::max(h->ev.lo, h->hv.lo), b = std::min(h->ev.hi, h->hv.hi);
    fdtd::ProbeParams q{h->probe_row, h->probe_col, h->top() ? 0 : a, h->bottom() ? h->rows : b,
                        h->probe_dev, h->step - h->probe_step0, h->probe_cap};
    const int out_lo = h->top() ? 0 : a + nt, out_hi = h->bottom() ? h->rows : b - nt;
    if (q.row < out_lo || q.row >= out_hi || q.base < 0 || q.base >= q.cap) return 0;
    switch (nt) {
    case 16:
        if constexpr (sizeof(T) == 4) return launch_probe_nt<T, 16>(h, p, q);
        return fail(h, FDTD2D_E_ARG, "16-step passes are float32");
    case 8: return launch_probe_nt<T, 8>(h, p, q);
    case 4: return launch_probe_nt<T, 4>(h, p, q);
    case 2: return launch_probe_nt<T, 2>(h, p, q);
    case 1: return launch_probe_nt<T, 1>(h, p, q);
    default: return fail(h, FDTD2D_E_ARG, "no probe kernel for %d-step passes", nt);
    }
}

#ifdef FDTD_PASS_LONG_EXTERN   // the 16- and 20-step float32 kernels are built in translation units of their own
extern template int launch_pass_nt<float, 16>(fdtd2d *, fdtd::PassParams<float> &);
extern template int launch_pass_nt<float, 20>(fdtd2d *, fdtd::PassParams<float> &);
extern template int launch_pass_nt<double, 16>(fdtd2d *, fdtd::PassParams<double> &);
#endif

// One pass of nt in {1,2,4,8,16} steps; amps = nt amplitudes or nullptr.
template <class T> int launch_pass(fdtd2d *h, int nt, int band_lo, int band_hi, int src_row,
                                   int src_col, const double *amps, bool ztop, bool zbot,
                                   bool commit, int full_lo, int full_hi, int nlev)
{
    // (k_bulk also instantiates with 2 columns per lane -- 86 VGPRs, 4-5 waves per SIMD -- but
    // that measured 20 % slower than 4 columns: profiles/r01_kpass_ablation.txt)
    constexpr int V = fdtd::Vec<T>::N;
    fdtd::PassParams<T> p;
    p.ez_in = (const T *)h->ez[h->cur];
    p.hx_in = (const T *)h->hxb[h->hcur];
    p.hy_in = (const T *)h->hyb[h->hcur];
    p.ez_out = (T *)h->ez[h->cur ^ 1];
    p.hx_out = (T *)h->hxb[h->hcur ^ 1];
    p.hy_out = (T *)h->hyb[h->hcur ^ 1];
    p.ce = (const T *)h->ce;
    p.ch = (const T *)h->ch;
    p.ce_u = (T)h->ce_u;
    p.ch_u = (T)h->ch_u;
    p.k = (T)h->k_mur;
    p.g = h->geom();
    p.band_lo = band_lo;
    p.band_hi = band_hi;
    int br = h->stream_band_rows;
    h->shape_now = fdtd2d::Shape{0, 0};
    if (const fdtd2d::Shape *gs = br <= 0 ? h->shape_given(nt) : nullptr) {
        h->shape_now = *gs;
        br = gs->band_rows;
    }
    if (br <= 0) {
        auto it = h->tuned.find({nt, band_lo, band_hi});
        if (it != h->tuned.end()) {
            h->shape_now = it->second;
            br = it->second.band_rows;
        }
    }
    // waves side by side per level group (strips 2 or 4 waves wide): the caller's option, else the shape's
    {
        int sd = h->side_waves > 0 ? h->side_waves : std::max(1, h->shape_now.side);
        if (!h->side_ok(nt, sd) || !h->use_level_split(nt, band_lo, band_hi) || h->pml_split(nt) ||
            (h->split_waves != 0 && h->split_waves != 4))
            sd = 1;
        h->shape_now.side = sd;
        if (sd > 1) h->shape_now.waves = 4;
        const int OW = pass_strip_ow(nt, V, 4, sd);
        p.nstrips = (h->cols + OW - 1) / OW;
    }
    if (br <= 0) {
        // Measured on MI355X (interleaved A/B, profiles/r01_band_sweep.txt): the pass is fastest
        // with about one wave per wave slot (1024 SIMDs x 3 waves) and bands of 16..128 rows:
        // 16 rows at 2048^2, 24 at 4096^2, 64-96 at 8192^2, 96-192 at 16384^2.  Shorter bands pay
        // too much pipeline fill, taller ones leave SIMDs without a second wave to switch to.
        const int region = std::max(0, band_hi - band_lo);
        const bool split4 = h->use_level_split(nt, band_lo, band_hi);
        const int slots = split4 ? 2304 : 3072;     // measured: the level-split kernel likes ~32-row bands too
        const int want = std::max(1, (slots + p.nstrips - 1) / p.nstrips);
        // 16-step passes: ~58 bands whatever the size is what tune_pass() keeps finding on square
        // grids (64 rows at 4096^2, 144 at 8192^2, 304 at 16384^2: profiles/r01_autotune.txt)
        br = nt >= 16 ? std::min(std::max((region + 57) / 58, 64), 448)
                     : std::min(std::max(region / want, split4 ? 32 : 16), 128);
        // 8-step level-split passes: 1280 workgroups are resident at once (256 CUs x 5 at 93
        // VGPRs).  A launch of 1.0-1.6 times that leaves a thin second round; one round of taller
        // bands measured 5-10 % faster (3072^2: profiles/r01_split_waves_sweep.txt).
        if (split4 && nt <= 8) {
            const int cap = 1280, fit = std::max(1, (cap * 9 / 10) / p.nstrips);
            const long long wgs = (long long)((region + br - 1) / br) * p.nstrips;
            if (wgs > cap * 92 / 100 && wgs <= cap * 16 / 10) br = std::max(br, (region + fit - 1) / fit);
        }
    }
    p.band_rows = std::max(br, 1);
    p.band_rows_e = h->shape_now.edge_rows > 0 ? h->shape_now.edge_rows : p.band_rows;
    h->last_nt = nt;
    h->shape_last = fdtd2d::Shape{p.band_rows,
                                  h->pml_split(nt) ? 4 : (h->use_level_split(nt, band_lo, band_hi) ? h->split_waves_for(nt, band_lo, band_hi) : 1),
                                  h->pml_split(nt) ? (h->shape_now.edge_rows > 0 ? h->shape_now.edge_rows : h->pml_layer_rows) : p.band_rows_e,
                                  h->shape_now.side, h->xcd_map >= 0 ? h->xcd_map : h->shape_now.xcd,
                                  h->shape_now.short_rows, h->shape_now.n_short,
                                  nt > 16 && h->dtype == FDTD2D_F32 && h->shape_now.side <= 1 ? h->shape_now.fuse : 0};
    p.zone_top = ztop;
    p.zone_bot = zbot;
    p.trash = (T *)h->trash;
    constexpr int NONE = -(1 << 30);
    p.src_row = amps ? src_row : NONE;
    p.src_col = amps ? src_col : NONE;
    p.src_row1 = amps ? src_row + h->src_rows : NONE;
    p.src_col1 = amps ? src_col + h->src_cols : NONE;
    // a short pass: the nt-step kernel and geometry, advancing only nlev < nt levels (one sweep
    // over the grid for a tail of 3, 5, 6, 7, 9..15 steps instead of one per power of two)
    p.strip_first = 1;
    p.xcd_map = p.main_pad = p.main_per = p.main_tasks = p.n_inner = 0;
    p.zone_wgs = p.zone_last = 0;
    p.band_rows2 = p.nbands2 = 0;
    p.split_row = band_hi;
    p.src_strip = p.n_src = 0;
    p.band_rows_s = p.nbands_s = 1;
    p.src_lo = p.src_hi = band_hi;
    p.nsrc_top = p.nsrc_mid = 0;
    p.nlev = nlev > 0 ? std::min(nlev, nt) : nt;
    if (p.nlev != nt && !h->use_level_split(nt, band_lo, band_hi) && !h->pml_split(nt))
        return fail(h, FDTD2D_E_ARG, "short passes run on the level-split kernels only");
    for (int s = 0; s < fdtd::STREAM_MAX_NT; ++s) p.amp[s] = (amps && s < p.nlev) ? amps[s] : 0.0;
#ifdef FDTD2D_TRACE
    p.trace = (unsigned long long *)h->trace_dev;
    h->trace_blocks = 0;
#endif
    int rc;
    if (h->probe_pending) {     // the probe cell's own cone, recomputed by one extra workgroup
        h->probe_pending = false;
        if ((rc = launch_probe<T>(h, nt, p))) return rc;
    }
    if (h->boundary == FDTD2D_BOUNDARY_PML && h->pml_split(nt)) {
        p.zone_top = p.zone_bot = 0;
        p.zone_tiles = 0;
        p.fused_zones = 0;
        if constexpr (sizeof(T) == 4) rc = launch_pml_split_f32(h, p);
        else return fail(h, FDTD2D_E_ARG, "16-step PML passes are built for float32 only");
    } else if (h->boundary == FDTD2D_BOUNDARY_PML) {
        p.zone_top = p.zone_bot = 0;
        p.zone_tiles = 0;
        const int region = std::max(0, p.band_hi - p.band_lo);
        p.nbands = (region + p.band_rows - 1) / p.band_rows;
        p.nbands_e = p.nbands;
        fdtd::PmlPass<T> q{pml_factors<T>(h), (const T *)h->ezxb[h->hcur], (T *)h->ezxb[h->hcur ^ 1],
                           0, 0, 0, 0, 0, 0, 0};
        // rows whose 8-step cone can touch the top / bottom layer: [0, L+1+16) and the mirror
        const int reach = h->pml_L + 1 + 2 * nt;
        q.short_rows = h->pml_short_rows;
        auto up = [&](int x) { return (x + q.short_rows - 1) / q.short_rows * q.short_rows; };
        q.a_hi = std::min(p.band_hi, std::max(p.band_lo, h->top() ? p.band_lo + up(reach - p.band_lo) : p.band_lo));
        q.c_lo = std::max(q.a_hi, std::min(p.band_hi, h->bottom() ? p.band_hi - up(p.band_hi - (h->rows - reach)) : p.band_hi));
        q.n1 = (region + q.short_rows - 1) / q.short_rows;
        q.nA = (q.a_hi - p.band_lo + q.short_rows - 1) / q.short_rows;
        q.nC = (p.band_hi - q.c_lo + q.short_rows - 1) / q.short_rows;
        q.nB = (std::max(0, q.c_lo - q.a_hi) + p.band_rows - 1) / p.band_rows;
        const int inner = std::max(0, p.nstrips - 2);
        const long long blocks = 2LL * q.n1 + (long long)inner * (q.nA + q.nC + q.nB);
        if (blocks > 0) {
            if (h->ce_uniform)
                hipLaunchKernelGGL((fdtd::k_pass_pml<T, false>), dim3((unsigned)blocks), dim3(64), 0, h->stream, p, q);
            else
                hipLaunchKernelGGL((fdtd::k_pass_pml<T, true>), dim3((unsigned)blocks), dim3(64), 0, h->stream, p, q);
            HIPCHK(h, hipGetLastError());
            h->pass_launches++;
        }
        rc = 0;
    } else
    switch (nt) {
    case 20:
        if constexpr (sizeof(T) == 4) { rc = launch_pass_nt<T, 20>(h, p); break; }
        return fail(h, FDTD2D_E_ARG, "20-step passes are built for float32 only");
    case 16: rc = launch_pass_nt<T, 16>(h, p); break;
    case 8: rc = launch_pass_nt<T, 8>(h, p); break;
    case 4: rc = launch_pass_nt<T, 4>(h, p); break;
    case 2: rc = launch_pass_nt<T, 2>(h, p); break;
    case 1: rc = launch_pass_nt<T, 1>(h, p); break;
    default: return fail(h, FDTD2D_E_ARG, "unsupported pass length %d", nt);
    }
    if (rc) return rc;
    if (commit) {
        h->cur ^= 1;
        h->hcur ^= 1;
        h->ev = h->hv = Range{h->top() ? 0 : full_lo, h->bottom() ? h->rows : full_hi};
        h->step += p.nlev;
    }
    return 0;
}


}  // namespace fdtd_host

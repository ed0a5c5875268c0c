// Register-resident zone tiles (round 3): the top / bottom ZO = 5 + NT rows of a 16- or 20-step float32 pass and of a
// 16-step float64 pass (four waves per tile there: a float64 row takes two registers per value).
//
// zone_body (kernels_stream.hpp) keeps a tile in LDS and sends every cell of every step through it: ~40 k
// wave-instructions per 64-column tile and 16 steps, with four waves that share a CU with three other workgroups -- at
// 4096^2 the 274 tiles hold their slots for half the launch, at 8192^2 with an eps array a launch without them is 14 %
// shorter (profiles/r03_zone_cost.txt).  Here a tile is TWO waves whose lanes are the tile's 64 columns and whose
// registers hold the rows: wave 0 the upper half of the ZR = ZO + NT + 1 rows, wave 1 the lower half, each
// (Ez, Hx, Hy [, ce, ch]) x RW rows.  Row neighbours are registers of the same lane, column neighbours come by DPP, and
// the two waves exchange one row of Ez and one of Hx per step through LDS (two barriers per step).  The boundary rules
// are the reference's stages A-D (main.py:18-61) evaluated in the reference's order, written for rows in registers:
//   A  interior update of the row                                   (rows 1..R-2, columns 1..C-2)
//   B  left / right band: B[j] = P[j+-1] + k (A[j+-1] - P[j])          (rows 1..R-2, columns 0..4 and C-5..C-1)
//   C  top / bottom band: C[i] = P[i+-1] + k (B[i+-1] - P[i])          (columns 1..C-2, rows 0..4 and R-5..R-1)
//   D  corner blocks: mean of the inward row- and column-neighbour's C
// (P = Ez before the step).  Same operations, same order, one rounding each as MurRules (mur_rules.hpp), which stays the
// definition for the LDS tiles (float64, 8-step passes, the 128-column tiles of the 20-step pass on wide grids) and for
// k_frame_mur; tests/test_gpu_parity.py compares every path with the oracle cell for cell.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_stream.hpp"

namespace fdtd {

// which passes take their zone tiles from registers
// (float64: 16-step passes; its rows take two registers per value, so a tile is spread over four waves)
template <class T, int NT, bool WIDE = false> constexpr bool zone_in_registers()
{
    return !WIDE && (sizeof(T) == 4 ? NT >= 16 : NT == 16);
}
template <class T> constexpr int zone_waves() { return sizeof(T) == 4 ? 2 : 4; }       // waves per register-resident tile

// xch: 2 x 64 elements of LDS per wave of this tile (the row of Ez a wave hands up, the row of Hx it hands down);
// `half` = the wave's place in the tile, 0 .. NQ - 1 from the top
template <class T, int NT, bool CE_ARR, bool CH_ARR>
__device__ __forceinline__ void zone_wave(const PassParams<T> &p, const int tile, const bool bottom, const int half,
                                          const bool active, T *xch)
{
    using D = ZoneDims<NT>;
    static_assert(D::WL == 64, "one lane per tile column");
    constexpr int NQ = zone_waves<T>();
    constexpr int RW = (D::ZR + NQ - 1) / NQ;               // rows per wave
    static_assert(D::ZR - (NQ - 1) * RW >= 6 && RW >= 6, "the band rows and the row beyond them must sit in ONE wave");
    const Geom g = p.g;
    const int R = g.R, C = g.C;
    const int lane = threadIdx.x & 63;
    const int z0 = bottom ? R - D::ZR : 0, o0 = bottom ? R - D::ZO : 0;
    const int w0 = min(tile * D::WZ, max(0, C - D::WZ)), w1 = min(w0 + D::WZ, C);
    const int c0 = max(0, w0 - D::M), c1 = min(C, w0 + D::WZ + D::M);
    const int j = c0 + lane;
    const bool col_ok = lane < c1 - c0;
    const int l0 = half * RW;                                  // first tile row of this wave
    const int nrow = min(RW, D::ZR - l0);                      // rows it holds
    const int i0 = z0 + l0;                                    // ... = global rows [i0, i0 + nrow)
    const bool mh = col_ok && j <= C - 2;                      // Hx, Hy exist (main.py:70,74)
    const bool me = col_ok && j >= 1 && j <= C - 2;            // interior column (stages A and C)
    const bool in_l = col_ok && j < 5, in_r = col_ok && j >= C - 5;
    const bool has_l = c0 < 5, has_r = c1 > C - 5;             // the tile holds band columns (wave-uniform)
    const T k = p.k;

    // coefficients with the update masks folded in (x - 0 * d == x exactly for finite d): per row and lane when they
    // are arrays, else one value per lane and a wave-uniform row test
    T e[RW], x[RW], y[RW], cea[CE_ARR ? RW : 1], cha[CH_ARR ? RW : 1];
    const T ce_lane = me ? p.ce_u : T(0), ch_lane = mh ? p.ch_u : T(0);
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int i = min(i0 + r, R - 1);
        const size_t o = at(g, i, col_ok ? j : 0);
        const bool ok = col_ok && r < nrow;
        e[r] = ok ? p.ez_in[o] : T(0);
        x[r] = ok ? p.hx_in[o] : T(0);
        y[r] = ok ? p.hy_in[o] : T(0);
        if (CE_ARR) cea[r] = (ok && me && i0 + r >= 1 && i0 + r <= R - 2) ? p.ce[o] : T(0);
        if (CH_ARR) cha[r] = (ok && mh && i0 + r <= R - 2) ? p.ch[o] : T(0);
    }
    auto cem = [&](int r) { return CE_ARR ? cea[CE_ARR ? r : 0] : ((i0 + r >= 1 && i0 + r <= R - 2) ? ce_lane : T(0)); };
    auto chm = [&](int r) { return CH_ARR ? cha[CH_ARR ? r : 0] : ((i0 + r <= R - 2) ? ch_lane : T(0)); };

    for (int step = 1; step <= p.nlev; ++step) {
        // rows still inside the cone of the ZO output rows (as zone_body): `keep` rows next to the grid edge
        const int keep = min(D::ZR, D::ZO + (p.nlev - step) + 2);
        const int k_lo = bottom ? D::ZR - keep : 0, k_hi = bottom ? D::ZR : keep;      // tile rows [k_lo, k_hi)
        // ---- every wave hands its first row of Ez up to the wave above
        xch[128 * half + lane] = e[0];
        __syncthreads();
        const T e_below = half + 1 < NQ ? xch[128 * (half + 1) + lane] : e[RW - 1];
        // ---- H half-step (main.py:66-76)
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            if (l0 + r < k_lo || l0 + r >= k_hi || r >= nrow) continue;
            const T en = (r + 1 < RW) ? ((r + 1 < nrow) ? e[r + 1] : e[r]) : e_below;
            const T right = from_next(e[r]);
            const T ch = chm(r);
            x[r] = x[r] - ch * (en - e[r]);
            y[r] = y[r] + ch * (right - e[r]);
        }
        // ---- every wave hands its last row of Hx down to the wave below (every wave but the last holds RW rows)
        xch[128 * half + 64 + lane] = x[RW - 1];
        __syncthreads();
        const T x_above = half > 0 ? xch[128 * (half - 1) + 64 + lane] : x[0];
        // ---- E half-step, rows in increasing order; a top-band row is finished one row later (it needs B of the
        // row below), a bottom-band row at once (it needs B of the row above)
        T p_prev = T(0), b_prev = T(0);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            if (l0 + r < k_lo || l0 + r >= k_hi || r >= nrow) continue;
            const int i = i0 + r;
            const T po = e[r];
            const T xa = r > 0 ? x[r - 1] : x_above;
            const T left = from_prev(y[r]);
            const T a = po + ((y[r] - left) - (x[r] - xa)) * cem(r);                    // stage A
            T b = a;
            if (i >= 1 && i <= R - 2) {                                                  // stage B
                if (has_r) {
                    const T pl = from_prev(po), al = from_prev(a);
                    const T br = pl + k * (al - po);
                    b = in_r ? br : b;
                }
                if (has_l) {
                    const T pr = from_next(po), ar = from_next(a);
                    const T bl = pr + k * (ar - po);
                    b = in_l ? bl : b;
                }
            }
            T c = b;
            if (i >= R - 5 && i >= 5) {                                                  // stage C, bottom band
                const T cb = p_prev + k * (b_prev - po);
                c = me ? cb : b;
            }
            if (r > 0 && i - 1 < 5) {                                                    // stage C, top band: row i - 1
                const T ct = po + k * (b - p_prev);
                e[r - 1] = me ? ct : b_prev;
            }
            e[r] = c;
            p_prev = po;
            b_prev = b;
        }
        // ---- stage D: the corner blocks, from the C values of the whole tile
        if ((has_l || has_r) && !bottom && half == 0) {
#pragma unroll
            for (int r = 0; r < 5; ++r) {                      // rows 0..4, increasing: e[r + 1] is still its C value
                const T cn = from_next(e[r]), cp = from_prev(e[r]);
                const T dl = (cn + e[r + 1]) / T(2), dr = (cp + e[r + 1]) / T(2);
                e[r] = in_l ? dl : (in_r ? dr : e[r]);
            }
        }
        if ((has_l || has_r) && bottom && half == NQ - 1) {
#pragma unroll
            for (int q = 0; q < 5; ++q) {                      // rows R-1..R-5, decreasing: e[r - 1] is still its C value
                const int r = (D::ZR - (NQ - 1) * RW) - 1 - q;  // the last wave holds ZR - (NQ - 1) RW rows
                const T cn = from_next(e[r]), cp = from_prev(e[r]);
                const T dl = (e[r - 1] + cn) / T(2), dr = (e[r - 1] + cp) / T(2);
                e[r] = in_l ? dl : (in_r ? dr : e[r]);
            }
        }
        // ---- source after the step (fdtd.py:34)
        if (p.src_row1 > i0 && p.src_row < i0 + nrow && j >= p.src_col && j < p.src_col1) {
            const double amp = p.amp[step - 1];
#pragma unroll
            for (int r = 0; r < RW; ++r)
                if (r < nrow && i0 + r >= p.src_row && i0 + r < p.src_row1) e[r] = (T)((double)e[r] + amp);
        }
    }

    if (active && col_ok && j >= w0 && j < w1) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int i = i0 + r;
            if (r < nrow && i >= o0 && i < o0 + D::ZO) {
                const size_t o = at(g, i, j);
                p.ez_out[o] = e[r];
                p.hx_out[o] = x[r];
                p.hy_out[o] = y[r];
            }
        }
    }
}

// The waves of a workgroup of THREADS threads take THREADS / 128 tiles; workgroup `wg` of the zone part of a launch.
// smem: 128 elements per wave.
template <class T, int NT, bool CE_ARR, bool CH_ARR, int THREADS>
__device__ __forceinline__ void zone_wave_group(const PassParams<T> &p, const int wg, T *smem)
{
    constexpr int NQ = zone_waves<T>(), TPW = THREADS / (64 * NQ);
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int q = wid / NQ, half = wid - q * NQ;
    const int total = (p.zone_top + p.zone_bot) * p.zone_tiles;
    const int gt = wg * TPW + q;
    const bool active = gt < total;                      // (a workgroup's last tiles may not exist: they run along for the
    const int gtc = min(gt, total - 1);                  // barriers and store nothing)
    const int z = gtc / p.zone_tiles;
    zone_wave<T, NT, CE_ARR, CH_ARR>(p, gtc - z * p.zone_tiles, p.zone_top ? z == 1 : true, half, active, smem + q * (128 * NQ));
}

template <class T, int THREADS> constexpr int zone_tiles_per_wg() { return THREADS / (64 * zone_waves<T>()); }
// LDS elements a workgroup's register-resident tiles need
template <class T, int THREADS> constexpr int zone_xch_elems() { return (THREADS / 64) * 128; }

// workgroups the zone tiles of a launch need: one per LDS tile, one per THREADS / 128 register-resident tiles
template <class T, int NT, int THREADS, bool WIDE = false> __host__ __device__ constexpr int zone_wgs_for(int tiles)
{
    return zone_in_registers<T, NT, WIDE>() ? (tiles + zone_tiles_per_wg<T, THREADS>() - 1) / zone_tiles_per_wg<T, THREADS>() : tiles;
}

// ---- the zone tiles as a launch of their own (side stream, beside the bulk) -----------------------------------------
template <class T, int NT, bool CE_ARR, bool CH_ARR, bool WIDE = false>
__global__ __launch_bounds__(PASS_THREADS) void k_zone(const PassParams<T> p)
{
    if constexpr (zone_in_registers<T, NT, WIDE>()) {
        __shared__ T xch[zone_xch_elems<T, PASS_THREADS>()];
        zone_wave_group<T, NT, CE_ARR, CH_ARR, PASS_THREADS>(p, blockIdx.x, xch);
        return;
    }
    const int z = blockIdx.x / p.zone_tiles;
    if constexpr (ZoneDims<NT, WIDE>::LDS_ELEMS * sizeof(T) > 65536) {      // beyond the static limit: dynamic LDS
        extern __shared__ __attribute__((aligned(16))) unsigned char zone_dyn[];
        zone_body<T, NT, CE_ARR, CH_ARR, PASS_THREADS, WIDE>(p, blockIdx.x - z * p.zone_tiles, p.zone_top ? z == 1 : true,
                                                             reinterpret_cast<T *>(zone_dyn));
    } else {
        __shared__ T smem[ZoneDims<NT, WIDE>::LDS_ELEMS];
        zone_body<T, NT, CE_ARR, CH_ARR, PASS_THREADS, WIDE>(p, blockIdx.x - z * p.zone_tiles, p.zone_top ? z == 1 : true, smem);
    }
}


}  // namespace fdtd

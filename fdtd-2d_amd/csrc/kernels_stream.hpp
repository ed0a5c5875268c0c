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

constexpr int PASS_THREADS = 256;    // threads per zone tile (4 waves share the tile through LDS)
// Masked-off stores (pipeline fill, overlap lanes) go to a scratch area instead of being
// branched around.  One 4 KiB slot per workgroup (index mod 1024): a single shared line would
// be written by every CU at once.
constexpr int TRASH_SLOTS = 1024, TRASH_SLOT_BYTES = 4096;
constexpr int STREAM_MAX_NT = 20;   // longest pass: 8 levels in one wave; 16 (4 or 8 waves) and 20 (4 waves x 5) with the level-split kernel
// halo columns per strip side: >= NT (validity shrinks one column per level from a strip
// edge) and a multiple of 4 so that every lane's 16-byte access stays aligned
constexpr int stream_hc(int nt) { return nt <= 4 ? 4 : (nt <= 8 ? 8 : (nt <= 16 ? 16 : 20)); }

template <class T> struct PassParams {
    const T *ez_in, *hx_in, *hy_in;
    T *ez_out, *hx_out, *hy_out;
    const T *ce, *ch;          // coefficient arrays (unused when uniform)
    T ce_u, ch_u, k;
    Geom g;
    int band_lo, band_hi;      // rows the streaming kernel produces
    int band_rows, nstrips, nbands;
    int band_rows_e, nbands_e;   // shorter bands for the first / last strip (their GENERAL body is
                               // ~2x slower per row: with equal heights they end a launch alone)
    int strip_first;           // first strip of the "inner" set (1; the PML pass gives its layer strips
                               // to another kernel and sets nbands_e = 0)
    int src_strip, n_src;      // inner strips [src_strip, src_strip + n_src) hold the source columns: their
    int band_rows_s, nbands_s; // workgroups near the source rows run the slower GENERAL body too and would end
                               // the launch alone, so the rows [src_lo, src_hi) around the source get short bands of
                               // their own in these strips (n_src = 0: none); the rows above and below keep band_rows:
    int src_lo, src_hi;        // nsrc_top bands of band_rows, then nsrc_mid of band_rows_s, then the rest (nbands_s in all)
    int nsrc_top, nsrc_mid;
    int xcd_map;               // 1: the inner strips' tasks are dealt out XCD by XCD (FDTD2D_OPT_XCD_MAP)
    int main_pad, main_per, main_tasks, n_inner;   // (with xcd_map) empty blocks in front of them, tasks per XCD, tasks, strips
    int band_rows2, nbands2, split_row;   // "filler" bands: the rows [split_row, band_hi) of the inner strips in nbands2
                               // shorter bands of band_rows2 rows, LAST in launch order (one-round launches: they start in
                               // the slots the zone tiles free; nbands2 = 0: none, split_row = band_hi)
    int zone_top, zone_bot;    // 1 if this launch owns the grid's top / bottom zone
    int zone_tiles;            // column tiles per zone
    int zone_wgs;              // workgroups of the zone part of a fused launch (register-resident tiles: several per workgroup)
    int zone_last;             // 1: they are the LAST workgroups of the launch (launches of several rounds: short tasks for the tail)
    int fused_zones;           // 1: the zone tiles are the first workgroups of the k_bulk launch
                               // (one wave each); 0: k_zone runs them on a side stream
    T *trash;                  // >= 3 x 1 KiB of device scratch: where masked-off stores land
    int src_row, src_col;      // first source cell; the source is the rectangle
    int src_row1, src_col1;    // [src_row, src_row1) x [src_col, src_col1) (all four very negative: none)
    int nlev;                  // time levels this launch really advances (<= NT; the level-split
                               // kernel, the zone tiles and the probe tile skip the rest)
    double amp[STREAM_MAX_NT]; // amplitude added after step s = 1..NT of this pass
#ifdef FDTD2D_TRACE              // profiling build only (tools/Makefile.trace): per-workgroup time stamps
    unsigned long long *trace;
#endif
};

#ifdef FDTD2D_TRACE
// {start, end} in shader cycles (s_memtime), kind (0 zone tile, 1 edge strip, 2 plain strip), the
// hardware id (XCC << 32 | HW_ID: wave slot, SIMD, CU, SE, workgroup slot of wave 0) of every workgroup of a launch, and the
// cycles each of its first four waves spent waiting at the tick barrier (8 words per workgroup).
struct TraceScope {
    unsigned long long *q;
    unsigned long long t0;
    int kind = 0;
    __device__ __forceinline__ TraceScope(unsigned long long *base)
        : q(base ? base + 8 * (size_t)blockIdx.x : nullptr), t0(__builtin_amdgcn_s_memtime()) {}
    __device__ __forceinline__ ~TraceScope()
    {
        if (q && threadIdx.x == 0) {
            q[0] = t0;
            q[1] = __builtin_amdgcn_s_memtime();
            q[2] = (unsigned long long)kind;
            q[3] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                   (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    }
};
#endif

// ---- lane shifts -----------------------------------------------------------------------
// from_next(x): lane l gets lane l+1's x; from_prev(x): lane l gets lane l-1's x.
// (lane 63 / lane 0 get 0: those are strip-edge lanes whose results are never used)
// A DPP-encoded instruction in the wrong place stalls the SIMD's issue once several waves share the SIMD
// (tools/ubench_dpp.hip, ubench_align.hip; profiles/r02_ubench_dpp.txt, r02_ubench_align.txt: one DPP form per
// 16-32 plain VALU instructions costs the SIMD 11-16 ns when it directly follows VALU work, 2-3 ns behind a few
// s_nop or an alignment pad, nothing in a run of adjacent DPP forms).  In the level update of the pass kernels
// (staged_level, kernels_split.hpp) the two lane shifts are therefore written out by hand at fixed places
// (diff_next / diff_prev below): 6-8 % of a launch.  The generic shifts here -- Mur band of the edge strips,
// k_bulk tails, PML layer body -- stay with the compiler (hand-written s_nop + aligned v_mov_b32_dpp forms
// measured 0-2 % slower there, profiles/r02_dpp_alignment.txt).
__device__ __forceinline__ int dpp_next_c(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x130, 0xF, 0xF, true); }
__device__ __forceinline__ int dpp_prev_c(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xF, 0xF, true); }
__device__ __forceinline__ float from_next(float x) { return __builtin_bit_cast(float, dpp_next_c(__builtin_bit_cast(int, x))); }
__device__ __forceinline__ float from_prev(float x) { return __builtin_bit_cast(float, dpp_prev_c(__builtin_bit_cast(int, x))); }
// diff_next(a, b) = a of lane l+1  -  b;  diff_prev(a, b) = a  -  b of lane l-1  (one instruction each for float).
// The CALLER keeps two instructions between a VALU write of the shifted operand and these (no s_nop inside).
#define DPP_PRE ".p2align 3\n\t"
__device__ __forceinline__ float diff_next(float a, float b)
{
    float r;
    asm volatile(DPP_PRE "v_sub_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float diff_prev(float a, float b)
{
    float r;
    asm volatile(DPP_PRE "v_subrev_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(b), "v"(a));
    return r;
}
// float64: no DPP form of the 64-bit arithmetic exists, a shift is two v_mov_b32_dpp; these stay with the
// compiler (its own placement and hazard handling) and the level keeps its chained form: the staged body with a
// hand-placed run of the two moves measured the same within 1 % from 512^2 to 8192^2 (profiles/r02_dpp_alignment.txt).
__device__ __forceinline__ double from_next(double x)
{
    const long long b = __builtin_bit_cast(long long, x);
    const unsigned lo = (unsigned)dpp_next_c((int)(unsigned)b), hi = (unsigned)dpp_next_c((int)(b >> 32));
    return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double from_prev(double x)
{
    const long long b = __builtin_bit_cast(long long, x);
    const unsigned lo = (unsigned)dpp_prev_c((int)(unsigned)b), hi = (unsigned)dpp_prev_c((int)(b >> 32));
    return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}

// One row of the strip on its way through the time levels: the registers of a slot are
// loaded with level 0 of row r and then updated IN PLACE to level 1, 2, ... NT on the
// following ticks, so no value is ever copied between registers.
// First column held by a strip.  Strips advance by OW = SW - 2 HC columns and overlap by HC
// on each side.  The LAST strip is shifted left until it ends at the grid's right edge:
// inside the right Mur band a cell depends on columns j-2..j of the previous level, so
// invalid data entering from a strip's left edge would advance two columns per level
// there; with the shift that edge is a full strip width away from the band.
//
// SD > 1 (level-split kernel only): SD waves side by side share a strip.  Within a tick a wave advances LV levels,
// so its values within LV columns of its window's INNER edges go stale -- LOV = ceil(LV / V) lanes per side.  The
// windows of neighbouring waves therefore overlap by 2 LOV lanes, each wave hands on only the lanes it still owns,
// and the next level group reads full 64-lane windows out of the joint row in LDS: the waves exchange their boundary
// columns through the hand-off they perform anyway.  Only the strip's two OUTER edges pay the HC = NT columns of
// overlap: 32 of 504 columns (SD = 2) or of 1000 (SD = 4) instead of 32 of 256.
constexpr int side_lov(int lv, int v, int sd) { return sd > 1 ? (lv + v - 1) / v : 0; }
constexpr int side_units(int lv, int v, int sd) { return sd * 64 - (sd - 1) * 2 * side_lov(lv, v, sd); }   // lanes of a joint row
constexpr int strip_width(int lv, int v, int sd) { return side_units(lv, v, sd) * v; }                      // columns held
template <class T, int NT, int V = Vec<T>::N, int SD = 1, int LV = 4>
__device__ __forceinline__ int strip_x0(const PassParams<T> &p, int strip)
{
    constexpr int WT = strip_width(LV, V, SD), HC = stream_hc(NT), OW = WT - 2 * HC;
    int x0 = strip * OW - HC;
    if (strip == p.nstrips - 1) x0 = min(x0, (p.g.C - WT + 3) & ~3);
    return x0;
}

// Launch order of the (band, strip) tasks: first the edge strips (0 and last: their GENERAL body
// is the slower one), then strips 1, 2, ...; all bands of one strip are consecutive.
template <class T>
__device__ __forceinline__ bool strip_of_block(const PassParams<T> &p, int b, int *strip, int *ra, int *rb)
{
    if (b < 2 * p.nbands_e) {
        const int sidx = b / p.nbands_e, band = b - sidx * p.nbands_e;
        if (sidx == 1 && p.nstrips == 1) return false;    // the second edge slot stays empty
        *strip = sidx == 0 ? 0 : p.nstrips - 1;
        *ra = p.band_lo + band * p.band_rows_e;
        *rb = min(*ra + p.band_rows_e, p.band_hi);
    } else if ((b -= 2 * p.nbands_e) < p.n_src * p.nbands_s) {
        const int sidx = b / p.nbands_s;
        int band = b - sidx * p.nbands_s;
        *strip = p.src_strip + sidx;
        if (band < p.nsrc_top) {
            *ra = p.band_lo + band * p.band_rows;
            *rb = min(*ra + p.band_rows, p.src_lo);
        } else if ((band -= p.nsrc_top) < p.nsrc_mid) {
            *ra = p.src_lo + band * p.band_rows_s;
            *rb = min(*ra + p.band_rows_s, p.src_hi);
        } else {
            *ra = p.src_hi + (band - p.nsrc_mid) * p.band_rows;
            *rb = min(*ra + p.band_rows, p.band_hi);
        }
    } else {
        b -= p.n_src * p.nbands_s;
        int sidx, band;
        if (p.xcd_map) {
            // Workgroups b, b + 8, b + 16 ... share an XCD (observed round-robin placement; speed only).  XCD x takes
            // the tasks [x * per, (x + 1) * per) of the list ordered band by band with the strips of a band next to
            // each other: neighbouring strips run at the same time on the same L2 and fetch the lines they share once.
            b -= p.main_pad;
            if (b < 0) return false;
            const int x = b & 7, q = b >> 3, t = x * p.main_per + q;
            if (q >= p.main_per || t >= p.main_tasks) return false;
            band = t / p.n_inner;
            sidx = t - band * p.n_inner;
        } else if (b < p.n_inner * p.nbands || p.nbands2 == 0) {
            sidx = b / p.nbands;
            band = b - sidx * p.nbands;
        } else {                        // the filler bands of the inner strips
            b -= p.n_inner * p.nbands;
            sidx = b / p.nbands2;
            band = b - sidx * p.nbands2;
            int st = sidx + p.strip_first;
            if (p.n_src > 0 && st >= p.src_strip) st += p.n_src;
            *strip = st;
            *ra = p.split_row + band * p.band_rows2;
            *rb = min(*ra + p.band_rows2, p.band_hi);
            return *ra < *rb;
        }
        int st = sidx + p.strip_first;
        if (p.n_src > 0 && st >= p.src_strip) st += p.n_src;      // the source strips were dealt with above
        *strip = st;
        *ra = p.band_lo + band * p.band_rows;
        *rb = min(*ra + p.band_rows, p.split_row);
    }
    return *ra < *rb;
}

template <class T, bool CE_ARR, bool CH_ARR, int V = Vec<T>::N> struct Slot {
    VecN<T, V> e, x, y;
    VecN<T, V> ce, ch;   // the row's coefficients ride along (only touched when arrays)
};

constexpr int STREAM_PF = 2;   // rows in flight ahead of level 0 (3 or 4 cost a wave per SIMD: profiles/r01_short_pass_cost.txt)

// ---- the streaming kernel ------------------------------------------------------------------
// GENERAL = false: every column of the strip is a plain interior column and the point
// source is outside the wave's dependency cone -- no masks at all.
// GENERAL = true adds what the reference does at the left/right grid edge and the source:
//   * cells the reference does not update (Hx, Hy beyond column C-2, Ez in columns 0 and
//     C-1, padding) keep their value: their coefficient is replaced by 0, and x -/+ 0*d == x
//     exactly for finite d, so no select is needed;
//   * the left/right 5-px Mur band of the row (main.py:34-41), under wave-uniform branches
//     that only the first / last strip takes;
//   * the point source (fdtd.py:34), under a wave-uniform row test.
template <class T, int NT, bool CE_ARR, bool CH_ARR, bool GENERAL, int V = Vec<T>::N>
__device__ __forceinline__ void stream_body(const PassParams<T> &p, const int strip, const int ra,
                                            const int rb)
{
    using VT = VecN<T, V>;
    using SlotT = Slot<T, CE_ARR, CH_ARR, V>;
    constexpr int HC = stream_hc(NT);
    constexpr int SW = 64 * V, OW = SW - 2 * HC;
    // rows in flight ahead of level 0 (the GENERAL body gives one up to stay within the
    // 168 VGPRs that allow 3 waves per SIMD for the whole kernel)
    constexpr int PF = GENERAL ? 1 : STREAM_PF;
    constexpr int S = NT + PF + 2;         // ring of row slots; the tick loop is unrolled S times
    const Geom g = p.g;
    const int lane = threadIdx.x;
    const int x0 = strip_x0<T, NT, V>(p, strip);
    const int j0 = x0 + V * lane;
    const bool ld_ok = j0 >= 0 && j0 < g.C;
    const bool st_ok = ld_ok && j0 >= strip * OW && j0 < (strip + 1) * OW;
    const size_t col = (size_t)(ld_ok ? j0 : 0);
    const int tau0 = ra - NT, tau1 = rb + NT;   // level-0 rows [tau0, tau1)

    // GENERAL: per-element masks (as 0/1 factors) and band membership, fixed for the strip
    VT mh, me;
    bool in_l[V], in_r[V];
    const bool has_l = GENERAL && x0 < 5, has_r = GENERAL && x0 + SW > g.C - 5;
    if (GENERAL) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int j = j0 + v;
            mh.v[v] = (j >= 0 && j <= g.C - 2) ? T(1) : T(0);
            me.v[v] = (j >= 1 && j <= g.C - 2) ? T(1) : T(0);
            in_l[v] = j >= 0 && j < 5;
            in_r[v] = j >= g.C - 5 && j < g.C;
        }
    }

    SlotT slot[S];
#pragma unroll
    for (int k = 0; k < S; ++k)
#pragma unroll
        for (int v = 0; v < V; ++v)
            slot[k].e.v[v] = slot[k].x.v[v] = slot[k].y.v[v] = slot[k].ce.v[v] = slot[k].ch.v[v] = T(0);

    // Memory operations in the tick loop are UNCONDITIONAL (rows clamped into the band's
    // range, masked-off lanes redirected): with loads or stores under a branch the compiler
    // can no longer count outstanding operations exactly and falls back to s_waitcnt
    // vmcnt(0) in front of every level -- which serialises the prefetch (measured: 58 % of
    // wave time in s_waitcnt, profiles/r01_kpass_ablation.txt).
    auto load_row = [&](SlotT &r, int i) {
        const int ic = min(i, tau1 - 1);             // rows past the band are never used
        const size_t o = at(g, ic, 0) + col;         // lanes outside the grid read column 0
        r.e = ldn<V>(p.ez_in + o);
        r.x = ldn<V>(p.hx_in + o);
        r.y = ldn<V>(p.hy_in + o);
        if (CE_ARR) r.ce = ldn<V>(p.ce + o);
        if (CH_ARR) r.ch = ldn<V>(p.ch + o);
        if (GENERAL) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                // lanes outside the grid hold zeros; update masks fold into the coefficients
                r.e.v[v] = ld_ok ? r.e.v[v] : T(0);
                r.x.v[v] = ld_ok ? r.x.v[v] : T(0);
                r.y.v[v] = ld_ok ? r.y.v[v] : T(0);
                if (CE_ARR) r.ce.v[v] = (ld_ok && me.v[v] != T(0)) ? r.ce.v[v] : T(0);
                if (CH_ARR) r.ch.v[v] = (ld_ok && mh.v[v] != T(0)) ? r.ch.v[v] : T(0);
            }
        }
    };
    VT ceu, chu;   // uniform coefficients, masked per element when GENERAL
#pragma unroll
    for (int v = 0; v < V; ++v) {
        ceu.v[v] = GENERAL ? (me.v[v] != T(0) ? p.ce_u : T(0)) : p.ce_u;
        chu.v[v] = GENERAL ? (mh.v[v] != T(0) ? p.ch_u : T(0)) : p.ch_u;
    }

#pragma unroll
    for (int k = 0; k < PF; ++k) load_row(slot[k], tau0 + k);

    for (int tb = tau0; tb < tau1; tb += S) {
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const int tau = tb + k;        // row tau lives in slot k (tb - tau0 is a multiple of S)
            if (tau >= tau1) break;
            load_row(slot[(k + PF) % S], tau + PF);
#pragma unroll
            for (int t = 1; t <= NT; ++t) {
                const int i = tau - t;                                   // row level t updates now
                // level t is only needed on rows [ra-(NT-t)-1, rb+(NT-t)): skip the rest of the
                // pipeline fill and drain (wave-uniform)
                if (i < ra - (NT - t) - 1 || i >= rb + (NT - t)) continue;
                SlotT &c = slot[(k - t + 2 * S) % S];      // row i, level t-1 -> t
                const SlotT &nx = slot[(k - t + 1 + 2 * S) % S];  // row i+1, level t-1
                const SlotT &pv = slot[(k - t - 1 + 2 * S) % S];  // row i-1, level t
                // H half-step of row i (main.py:66-76)
                const T e_next_lane = from_next(c.e.v[0]);
                VT po;      // Ez of row i before this step's E half-step (band rows only)
                if (GENERAL) po = c.e;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const T ch = CH_ARR ? c.ch.v[v] : (GENERAL ? chu.v[v] : p.ch_u);
                    const T right = (v + 1 < V) ? c.e.v[v + 1] : e_next_lane;
                    c.x.v[v] = c.x.v[v] - ch * (nx.e.v[v] - c.e.v[v]);
                    c.y.v[v] = c.y.v[v] + ch * (right - c.e.v[v]);
                }
                // E half-step of row i, stage A (main.py:21-27); rows here are always interior
                const T hy_prev_lane = from_prev(c.y.v[V - 1]);
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const T ce = CE_ARR ? c.ce.v[v] : (GENERAL ? ceu.v[v] : p.ce_u);
                    const T left = (v > 0) ? c.y.v[v - 1] : hy_prev_lane;
                    c.e.v[v] = c.e.v[v] + ((c.y.v[v] - left) - (c.x.v[v] - pv.x.v[v])) * ce;
                }
                if (GENERAL) {
                    // stage B: the 5-px Mur band of this row, B[j] = P[j+-1] + k (A[j+-1] - P[j])
                    if (has_l) {
                        const T a_next = from_next(c.e.v[0]);
                        VT out;
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            const T pr = (v + 1 < V) ? po.v[v + 1] : e_next_lane;
                            const T ar = (v + 1 < V) ? c.e.v[v + 1] : a_next;
                            const T bl = pr + p.k * (ar - po.v[v]);
                            out.v[v] = in_l[v] ? bl : c.e.v[v];
                        }
                        c.e = out;
                    }
                    if (has_r) {
                        const T a_prev = from_prev(c.e.v[V - 1]), p_prev = from_prev(po.v[V - 1]);
                        VT out;
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            const T pl = (v > 0) ? po.v[v - 1] : p_prev;
                            const T al = (v > 0) ? c.e.v[v - 1] : a_prev;
                            const T br = pl + p.k * (al - po.v[v]);
                            out.v[v] = in_r[v] ? br : c.e.v[v];
                        }
                        c.e = out;
                    }
                    if (i >= p.src_row && i < p.src_row1) {   // source after step t of this pass (fdtd.py:34)
#pragma unroll
                        for (int v = 0; v < V; ++v)
                            if (j0 + v >= p.src_col && j0 + v < p.src_col1)
                                c.e.v[v] = (T)((double)c.e.v[v] + p.amp[t - 1]);
                    }
                }
            }
            {
                const int io = tau - NT;
                const SlotT &f = slot[(k - NT + 2 * S) % S];
                const bool keep = st_ok && io >= ra;           // else: pipeline fill / overlap lanes
                const size_t o = at(g, max(io, ra), 0) + col;
                const size_t d = (size_t)(blockIdx.x % TRASH_SLOTS) * (TRASH_SLOT_BYTES / sizeof(T)) + (size_t)lane * V;
                stn<V>(keep ? p.ez_out + o : p.trash + d, f.e);
                stn<V>(keep ? p.hx_out + o : p.trash + d + 64 * V, f.x);
                stn<V>(keep ? p.hy_out + o : p.trash + d + 128 * V, f.y);
            }
        }
    }
}

// ---- the top/bottom zone tiles ---------------------------------------------------------------
// One wave per tile: 32 columns x (2 NT + 6) rows of Ez (double-buffered), Hx, Hy in LDS.
template <int NT, bool WIDE = false> struct ZoneDims {
    static constexpr int ZO = 5 + NT;          // rows written per zone
    static constexpr int ZR = ZO + NT + 1;     // rows held in LDS
    static constexpr int M = NT + 1;           // margin columns per side
    // columns held in LDS (a power of two >= 2 M + 8).  WIDE: the 20-step pass on wide grids -- its zones run as
    // k_zone beside the bulk kernel, with LDS of their own: 128 columns (86 written of 128 held instead of 22 of 64;
    // one 95 KB tile per CU leaves room for three bulk workgroups where three 48 KB tiles left room for none:
    // run(20) 1738-1773 vs 1830-1866 us at 16384^2, 595 vs 621-630 at 8192^2 -- but 364 vs 239 at 4096^2, where the
    // 96 long-lived tiles outlast the bulk: narrow tiles below 8192 columns, profiles/r02_nt20_small.txt)
    static constexpr int WL = NT <= 8 ? 32 : (WIDE ? 128 : 64);
    static constexpr int WZ = WL - 2 * M;      // columns written per tile
    static constexpr int WLP = WL + 1;         // padded LDS row
    static constexpr int LDS_ELEMS = 4 * ZR * WLP;   // Ez (two buffers), Hx, Hy
};

template <class T, bool CE_ARR> struct TileAcc {
    const T *P, *x, *y;   // LDS tiles, row stride wlp
    const T *cearr;
    T ce_u;
    Geom g;
    int R, C, z0, c0, wlp;
    int nr, nc;           // rows / columns the tile holds
    // Cells next to a non-physical tile edge ask the rules for neighbours the tile does not hold
    // (their results are outside the validity cone and never used).  The index is clamped into the
    // tile so that such a read stays inside the arrays whatever address space they live in.
    __device__ __forceinline__ int idx(int i, int j) const
    {
        return min(max(i - z0, 0), nr - 1) * wlp + min(max(j - c0, 0), nc - 1);
    }
    __device__ __forceinline__ T p(int i, int j) const { return P[idx(i, j)]; }
    __device__ __forceinline__ T hx(int i, int j) const { return x[idx(i, j)]; }
    __device__ __forceinline__ T hy(int i, int j) const { return y[idx(i, j)]; }
    __device__ __forceinline__ T ce(int i, int j) const
    {
        return CE_ARR ? cearr[at(g, min(max(i, 0), R - 1), min(max(j, 0), C - 1))] : ce_u;
    }
};

// smem: ZoneDims<NT>::LDS_ELEMS elements of LDS owned by the calling kernel (k_bulk_split shares
// the allocation with its hand-off buffers: a workgroup is either a zone tile or a strip)
template <class T, int NT, bool CE_ARR, bool CH_ARR, int THREADS, bool WIDE = false>
__device__ __forceinline__ void zone_body(const PassParams<T> &p, const int tile, const bool bottom,
                                          T *smem)
{
    using D = ZoneDims<NT, WIDE>;
    static_assert((D::WL & (D::WL - 1)) == 0 && THREADS % D::WL == 0 && D::WZ >= 8, "tile shape");
    // (offsets from the one LDS base, never a table of pointers: those become generic pointers and
    // flat loads; every tile index is clamped into the tile, see TileAcc::idx)
    constexpr int ZS = D::ZR * D::WLP;
    T *const sX = smem + 2 * ZS;
    T *const sY = smem + 3 * ZS;
    const Geom g = p.g;
    // thread -> (row group, column): THREADS/WL tile rows per sweep, no integer division
    constexpr int RG = THREADS / D::WL;
    const int lj = threadIdx.x & (D::WL - 1), lr = threadIdx.x / D::WL;
    // rows held [z0, z0+ZR), rows written [o0, o0+ZO)
    const int z0 = bottom ? g.R - D::ZR : 0;
    const int o0 = bottom ? g.R - D::ZO : 0;
    // columns written [w0, w1), columns held [c0, c1).  The last tile is shifted left to full
    // width (it recomputes a few columns of its neighbour, writing identical values) so that
    // the right Mur band never sits next to a non-physical tile edge.
    const int w0 = min(tile * D::WZ, max(0, g.C - D::WZ)), w1 = min(w0 + D::WZ, g.C);
    const int c0 = max(0, w0 - D::M), c1 = min(g.C, w0 + D::WZ + D::M);
    const int wl = c1 - c0;
    const int j = c0 + lj;
    const bool col_ok = lj < wl;
    // does this tile hold columns of the left/right Mur band (or the edge columns)?  wave-uniform
    const bool lr_band = c0 < 5 || c1 > g.C - 5;
    const bool col_band = j < 5 || j >= g.C - 5;

#pragma unroll
    for (int it = 0; it < (D::ZR + RG - 1) / RG; ++it) {     // unrolled: all loads in flight at once
        const int li = lr + it * RG;
        if (col_ok && li < D::ZR) {
            const size_t o = at(g, z0 + li, j);
            const int s = li * D::WLP + lj;
            smem[s] = p.ez_in[o];
            sX[s] = p.hx_in[o];
            sY[s] = p.hy_in[o];
        }
    }
    __syncthreads();

    int cur = 0;
#pragma unroll 1
    for (int step = 1; step <= p.nlev; ++step) {
        const T *Eo = smem + cur * ZS;
        T *En = smem + (cur ^ 1) * ZS;
        // Rows still inside the cone of the ZO output rows: with k steps to go after this one,
        // level `step` is needed on the ZO + k rows next to the grid edge (the Mur rows grow two
        // rows per step, 4 + 2 k + 2 <= ZO + k + 2); one row more for the H of the next step.
        const int keep = min(D::ZR, D::ZO + (p.nlev - step) + 2);
        const int r_lo = bottom ? D::ZR - keep : 0, r_hi = bottom ? D::ZR : keep;
        // H half-step (main.py:66-76) on every cell whose i+1 / j+1 neighbours are in the tile
        for (int li = r_lo + lr; li < r_hi; li += RG) {
            const int i = z0 + li;
            if (col_ok && i <= g.R - 2 && j <= g.C - 2 && li + 1 < D::ZR && lj + 1 < wl) {
                const int s = li * D::WLP + lj;
                const T ch = CH_ARR ? p.ch[at(g, i, j)] : p.ch_u;
                const T e = Eo[s];
                sX[s] = sX[s] - ch * (Eo[s + D::WLP] - e);
                sY[s] = sY[s] + ch * (Eo[s + 1] - e);
            }
        }
        __syncthreads();
        // E half-step: stages A-D as one pure function of (Eo, new H) per cell; plain interior
        // cells (the vast majority) take the direct formula
        MurRules<T, TileAcc<T, CE_ARR>> rules{{Eo, sX, sY, p.ce, p.ce_u, g, g.R, g.C, z0, c0, D::WLP, D::ZR, wl}, p.k};
        for (int li = r_lo + lr; li < r_hi; li += RG) {
            const int i = z0 + li;
            const int s = li * D::WLP + lj;
            if (col_ok) {
                // needs the row above and the column to the left inside the tile unless the
                // cell sits on the physical edge (where the rules never look outward)
                const bool ok = (li >= 1 || i == 0) && (lj >= 1 || j == 0);
                T val = Eo[s];
                if (ok) {
                    const bool row_band = i < 5 || i >= g.R - 5;
                    if (row_band || (lr_band && col_band)) {
                        val = rules.d(i, j);
                    } else {
                        const T ce = CE_ARR ? p.ce[at(g, i, j)] : p.ce_u;
                        val = val + ((sY[s] - sY[s - 1]) - (sX[s] - sX[s - D::WLP])) * ce;
                    }
                }
                if (i >= p.src_row && i < p.src_row1 && j >= p.src_col && j < p.src_col1)
                    val = (T)((double)val + p.amp[step - 1]);
                En[s] = val;
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    const T *Ef = smem + cur * ZS;
    for (int r = lr; r < D::ZO; r += RG) {
        const int i = o0 + r;
        if (col_ok && j >= w0 && j < w1) {
            const int s = (i - z0) * D::WLP + lj;
            const size_t o = at(g, i, j);
            p.ez_out[o] = Ef[s];
            p.hx_out[o] = sX[s];
            p.hy_out[o] = sY[s];
        }
    }
}

// ---- launches ---------------------------------------------------------------------------------
// k_bulk: one wave per workgroup, one (band, strip) task each; launch order: the edge strips
// (0 and last: they run the slower GENERAL body), then strips 1, 2, ...; no barriers.
// Zone tiles either ride in the same launch (first workgroups, one wave per tile) or run as
// k_zone (256 threads per tile) on a side stream, joined back by events.  A tile is a long
// dependent chain of LDS phases (~80 us with one wave, ~25 us with four): on small grids it
// would set the pass latency, so they take k_zone; on large grids the bulk dominates and the
// cross-stream join costs more than it saves, so they take the fused form
// (profiles/r01_kpass_ablation.txt).
// Minimum waves per SIMD the register allocator must leave room for: 3 (<= 168 VGPRs) for
// uniform materials -- the slot ring alone is 144 -- and 2 when coefficient rows ride along.
template <class T, int NT, bool CE_ARR, bool CH_ARR, int V = Vec<T>::N>
__global__ __launch_bounds__(64, 2)
void k_bulk(const PassParams<T> p)
{
    constexpr int SW = 64 * V;
    int b = blockIdx.x;
    if (p.fused_zones) {     // zone tiles as the first workgroups of this launch (one wave each)
        const int nzone = (p.zone_top + p.zone_bot) * p.zone_tiles;
        if (b < nzone) {
            __shared__ T smem[ZoneDims<NT>::LDS_ELEMS];
            const int z = b / p.zone_tiles;
            zone_body<T, NT, CE_ARR, CH_ARR, 64>(p, b - z * p.zone_tiles, p.zone_top ? z == 1 : true, smem);
            return;
        }
        b -= nzone;
    }
    int strip, ra, rb;
    if (!strip_of_block(p, b, &strip, &ra, &rb)) return;
    const int x0 = strip_x0<T, NT, V>(p, strip);
    // wave-uniform choice: all SW columns plain interior (5 <= j <= C-6) and the source cell
    // outside the rows/columns this wave ever touches -> mask-free body
    const bool edge = x0 < 5 || x0 + SW > p.g.C - 5;
    const bool src = p.src_row1 > ra - 2 * NT && p.src_row < rb + NT && p.src_col1 > x0 &&
                     p.src_col < x0 + SW;
    if (edge || src)
        stream_body<T, NT, CE_ARR, CH_ARR, true, V>(p, strip, ra, rb);
    else
        stream_body<T, NT, CE_ARR, CH_ARR, false, V>(p, strip, ra, rb);
}

}  // namespace fdtd

// Level-split variant of the temporally blocked pass (8 or 16 steps per launch).
//
// k_bulk (kernels_stream.hpp) lets ONE wave carry all 8 time levels of a (band, strip): 12 row
// slots = 144 VGPRs, 3 waves per SIMD, and to keep ~3000 waves busy on a 4096^2 grid the bands
// can only be ~24 rows tall, so every band re-reads 16 neighbour rows (HBM reads 2x ideal).
// Here a workgroup of NW = 4 waves shares one (band, strip): wave w advances levels 2w+1 and
// 2w+2 and hands each finished row to wave w+1 through LDS (double-buffered by tick parity,
// one s_barrier per tick).  Each wave then needs 4-6 slots (<= 100 VGPRs, 5 waves per SIMD),
// and for the same number of waves the bands are 4x taller: less re-reading, less redundant
// arithmetic.  The arithmetic per cell is the same as everywhere else (value-identical).
//
// Tick tau of the workgroup: wave w takes as input row r = tau - 3w at level 2w (wave 0 from
// HBM with a 2-row prefetch, the others from the LDS buffer wave w-1 filled in the previous
// tick), updates row r-1 to level 2w+1 and row r-2 to level 2w+2 in place (same slot ring as
// k_bulk), and hands row r-2 on (the last wave stores it).  Three lag rows per wave: the
// level-8 row leaves at tick r+11.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_stream.hpp"
#include "kernels_zone.hpp"

namespace fdtd {

// NW waves per workgroup (4 or 8), steps per pass NT = 8 or 16: NT / NW levels per wave, one
// more row of lag per hand-off.  More waves per strip = fewer levels per wave and tick, so the
// same number of resident waves covers taller bands (less fill per band) -- the better trade on
// grids that cannot fill the GPU otherwise.

// One level of one row, in place: the reference's operations (main.py:66-76, 12-27), one rounding each, written
// stage by stage -- eight independent differences, eight products, eight sums ... -- instead of three-instruction
// chains through one temporary, with the two cross-lane operands folded into v_sub_f32_dpp forms written out at
// fixed places of that order (diff_next / diff_prev, kernels_stream.hpp).  44 VALU instructions per 4 cells;
// sched_barrier keeps the stages in this order.  ch(v), ce(v): the coefficients of column v of the lane.
template <class T, int V, class CH, class CE>
__device__ __forceinline__ void staged_level(VecN<T, V> &e, VecN<T, V> &x, VecN<T, V> &y, const VecN<T, V> &nxe,
                                             const VecN<T, V> &pvx, CH ch, CE ce)
{
    if constexpr (sizeof(T) == 8) {         // float64: the chained form, lane shifts placed by the compiler (kernels_stream.hpp)
        const T e_next_lane = from_next(e.v[0]);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const T c = ch(v);
            const T right = (v + 1 < V) ? e.v[v + 1] : e_next_lane;
            x.v[v] = x.v[v] - c * (nxe.v[v] - e.v[v]);
            y.v[v] = y.v[v] + c * (right - e.v[v]);
        }
        const T hy_prev_lane = from_prev(y.v[V - 1]);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const T left = (v > 0) ? y.v[v - 1] : hy_prev_lane;
            e.v[v] = e.v[v] + ((y.v[v] - left) - (x.v[v] - pvx.v[v])) * ce(v);
        }
    } else {
    VecN<T, V> dx, dy;
    dy.v[V - 1] = diff_next(e.v[0], e.v[V - 1]);                   // Ez[i, j+1] - Ez[i, j] across the lane edge
#pragma unroll
    for (int v = 0; v < V; ++v) dx.v[v] = nxe.v[v] - e.v[v];
#pragma unroll
    for (int v = 0; v + 1 < V; ++v) dy.v[v] = e.v[v + 1] - e.v[v];
    __builtin_amdgcn_sched_barrier(0);
#ifdef FDTD2D_FUSED      // the tolerance build: x - c * dx and y + c * dy as one v_fma_f32 each (8 instead of 16 here)
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const T c = ch(v);
        x.v[v] = __builtin_fmaf(-c, dx.v[v], x.v[v]);
        y.v[v] = __builtin_fmaf(c, dy.v[v], y.v[v]);
    }
#else
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const T c = ch(v);
        dx.v[v] = c * dx.v[v];
        dy.v[v] = c * dy.v[v];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < V; ++v) {
        x.v[v] = x.v[v] - dx.v[v];
        y.v[v] = y.v[v] + dy.v[v];
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < V; ++v) dx.v[v] = x.v[v] - pvx.v[v];
#pragma unroll
    for (int v = 1; v < V; ++v) dy.v[v] = y.v[v] - y.v[v - 1];
    __builtin_amdgcn_sched_barrier(0);                // (the DPP read of Hy comes >= 2 instructions after its write)
    dy.v[0] = diff_prev(y.v[0], y.v[V - 1]);                       // Hy[i, j] - Hy[i, j-1] across the lane edge
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < V; ++v) dy.v[v] = dy.v[v] - dx.v[v];
    __builtin_amdgcn_sched_barrier(0);
#ifdef FDTD2D_FUSED
#pragma unroll
    for (int v = 0; v < V; ++v) e.v[v] = __builtin_fmaf(dy.v[v], ce(v), e.v[v]);
#else
#pragma unroll
    for (int v = 0; v < V; ++v) dy.v[v] = dy.v[v] * ce(v);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < V; ++v) e.v[v] = e.v[v] + dy.v[v];
#endif
    __builtin_amdgcn_sched_barrier(0);
    }
}

// per-lane constants of a strip + the level update (same operations as stream_body)
template <class T, bool GENERAL, bool CE_ARR, bool CH_ARR, int V> struct StripMath {
    using VT = VecN<T, V>;
    struct Row {
        VT e, x, y;
        VT ce, ch;      // coefficient rows travel with the field rows (array materials only)
    };
    const PassParams<T> &p;
    int j0;
    bool ld_ok, has_l, has_r;
    VT ceu, chu;
    bool in_l[V], in_r[V], m_e[V], m_h[V];

    __device__ __forceinline__ StripMath(const PassParams<T> &pp, int x0, int lane) : p(pp)
    {
        constexpr int SW = 64 * V;
        j0 = x0 + V * lane;
        ld_ok = j0 >= 0 && j0 < p.g.C;
        has_l = GENERAL && x0 < 5;
        has_r = GENERAL && x0 + SW > p.g.C - 5;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int j = j0 + v;
            const bool mh = j >= 0 && j <= p.g.C - 2, me = j >= 1 && j <= p.g.C - 2;
            m_e[v] = me;
            m_h[v] = mh;
            ceu.v[v] = GENERAL ? (me ? p.ce_u : T(0)) : p.ce_u;
            chu.v[v] = GENERAL ? (mh ? p.ch_u : T(0)) : p.ch_u;
            in_l[v] = j >= 0 && j < 5;
            in_r[v] = j >= p.g.C - 5 && j < p.g.C;
        }
    }

    // row i: level t-1 -> t, in place.  nx = row i+1 at level t-1, pvx = Hx of row i-1 at level t
    __device__ __forceinline__ void level(Row &c, const VT &nxe, const VT &pvx, int t, int i) const
    {
        VT po;
        if (GENERAL) po = c.e;
        staged_level<T, V>(
            c.e, c.x, c.y, nxe, pvx, [&](int v) { return CH_ARR ? c.ch.v[v] : (GENERAL ? chu.v[v] : p.ch_u); },
            [&](int v) { return CE_ARR ? c.ce.v[v] : (GENERAL ? ceu.v[v] : p.ce_u); });
        if (GENERAL) {
            if (has_l) {
                const T a_next = from_next(c.e.v[0]), e_next_lane = from_next(po.v[0]);
                VT out;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const T pr = (v + 1 < V) ? po.v[v + 1] : e_next_lane;
                    const T ar = (v + 1 < V) ? c.e.v[v + 1] : a_next;
                    out.v[v] = in_l[v] ? pr + p.k * (ar - po.v[v]) : c.e.v[v];
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
                    out.v[v] = in_r[v] ? pl + p.k * (al - po.v[v]) : c.e.v[v];
                }
                c.e = out;
            }
            if (i >= p.src_row && i < p.src_row1) {
                const double amp = p.amp[t - 1];
#pragma unroll
                for (int v = 0; v < V; ++v)
                    if (j0 + v >= p.src_col && j0 + v < p.src_col1) c.e.v[v] = (T)((double)c.e.v[v] + amp);
            }
        }
    }
};

// Hand-off between the waves of a strip: wave w writes its finished row into its buffer of tick
// parity tau & 1, waits for the writes, joins the barrier; wave w+1 reads it at the start of tick
// tau + 1.  (Measured and rejected, profiles/r02_handoff_pipelining.txt: a 3-deep ring with the
// reads prefetched one tick ahead and only the OLDER writes awaited before the barrier.  It takes
// the LDS round trip out of every wave's tick but adds 2 ticks of fill per hand-off, and was 4-7 %
// slower at 4096^2, 8192^2 and 16384^2: with 4 waves per SIMD the other waves already cover that
// latency, the ticks run at ~95 % of the VALU issue rate those 4 waves can reach.)
constexpr int HAND_DEPTH = 2;

// (Measured and rejected, profiles/r02_lds_dma_loader.txt: the loading wave keeping 3 or 5 rows in flight
// through LDS-DMA -- global_load_lds_dwordx4 into a ring in LDS, read back with ds_read when due -- instead
// of 2 rows through registers.  Value-identical, 99 VGPRs, but 6-10 % slower at 8192^2 / 16384^2 and
// insensitive to the depth: the pass does not wait for the depth of its HBM prefetch.)

// (Per-wave barrier waits, profiles/r02_trace_barrier.txt: the loading wave is the one the others wait for
// at the tick barrier -- it waits 8-9 % of a workgroup's life, waves 1-3 23-34 %.  Giving it one level fewer
// and a third row of prefetch, the storing wave one more -- 3 / 4 / 4 / 5 -- was measured and rejected:
// 158 vs 148.5 us at 4096^2, 1590 vs 1495 at 16384^2: what the loading wave waits for is the data, and the
// longer last wave then sets the tick.)
// (Batched lane shifts, profiles/r02_dpp_alignment.txt: the four Ez shifts of a tick issued together, then the four Hy
// updates, the four Hy shifts together, then Hx / Ez level by level -- runs of adjacent DPP forms are free in the
// microbenchmark.  Same values, 118 VGPRs, but 10 % SLOWER at 4096^2 (150.5 vs 136 us), 5 % at 8192^2, 1 % at 16384^2:
// four cone checks per level instead of one and 4-wide instead of 8-wide stages cost more than the shifts do.)
// ROLE 0: first wave (HBM -> LDS), 1: middle (LDS -> LDS), 2: last (LDS -> HBM)
template <class T, int NT, int SPLIT_NW, bool CE_ARR, bool CH_ARR, bool GENERAL, int ROLE, int V, int SD = 1>
__device__ __forceinline__ void split_body(const PassParams<T> &p, const int strip, const int ra,
                                           const int rb, const int w, const int side, VecN<T, V> *lds)
{
    using M = StripMath<T, GENERAL, CE_ARR, CH_ARR, V>;
    constexpr int NF = 3 + (CE_ARR ? 1 : 0) + (CH_ARR ? 1 : 0);     // rows per hand-off
    using Row = typename M::Row;
    constexpr int LV = NT / SPLIT_NW, LAG = LV + 1;
    constexpr int HC = stream_hc(NT);
    // SD waves side by side (kernels_stream.hpp, strip_x0): LOV lanes per inner side go stale within a tick
    constexpr int LOV = side_lov(LV, V, SD), UW = 64 - 2 * LOV, U = side_units(LV, V, SD);
    constexpr int OW = U * V - 2 * HC;
    // only the first wave hides HBM latency (the 5-level form keeps one row in flight instead of two:
    // 8 slots of 12 registers like the 4-level form, 4 waves per SIMD)
    constexpr int PF = ROLE == 0 ? (LV > 4 ? 1 : STREAM_PF) : 0;
    constexpr int S = LV + 2 + PF;              // ring of row slots, tick loop unrolled S times
    const Geom g = p.g;
    const int lane = threadIdx.x & 63;
    const int x0 = strip_x0<T, NT, V, SD, LV>(p, strip) + side * (UW * V);      // this wave's window of the strip
    const M m(p, x0, lane);
    const int j0 = m.j0;
    // lanes whose values survive a tick (all of them with one wave per strip)
    const bool own = SD == 1 || ((side == 0 || lane >= LOV) && (side == SD - 1 || lane < 64 - LOV));
    const bool st_ok = m.ld_ok && own && j0 >= strip * OW && j0 < (strip + 1) * OW;
    const size_t col = (size_t)(m.ld_ok ? j0 : 0);
    const int tau0 = ra - NT, tau1 = rb + NT;                    // level-0 rows [tau0, tau1)
    const int tend = rb + LV + (SPLIT_NW - 1) * LAG;              // ticks [tau0, tend) for every wave
    const int shift = w * LAG;                                    // this wave's input row = tau - shift
    const int t0 = w * LV;                                        // level of the input rows
    // Level t0 + l of this wave updates row r - l in the tick whose input row is r; it is inside the
    // band's cone iff  r >= first[l]  and  r < r_end  (level t is needed on rows [ra-(NT-t)-1,
    // rb+(NT-t)); r_end does not depend on l), and part of a short pass iff t0 + l <= nlev.
    const int r_end = rb + NT - t0;
    int first[LV + 1];
#pragma unroll
    for (int l = 1; l <= LV; ++l) first[l] = t0 + l <= p.nlev ? ra - NT + t0 + 2 * l - 1 : (1 << 30);
    // hand-off ring: [hand-off h][slot d][field][lane]
    auto buf = [&](int h, int d, int field) { return lds + ((h * HAND_DEPTH + d) * NF + field) * U + side * UW + lane; };

#ifdef FDTD2D_TRACE
    unsigned long long trace_bar = 0;       // cycles this wave waits at the tick barrier (profiling build)
#endif
    Row slot[S];
#pragma unroll
    for (int k = 0; k < S; ++k)
#pragma unroll
        for (int v = 0; v < V; ++v)
            slot[k].e.v[v] = slot[k].x.v[v] = slot[k].y.v[v] = slot[k].ce.v[v] = slot[k].ch.v[v] = T(0);

    // Addresses as (row pointer in scalar registers) + (the lane's byte offset, 32 bits, constant over the band):
    // global_load / global_store take that pair directly; the 64-bit per-lane pointer arithmetic it replaces was 7
    // VALU instructions per tick in the loading wave -- the wave the others wait for.
    unsigned lane_off = (unsigned)col * (unsigned)sizeof(T);
    // (the row pointer goes through an empty asm as a scalar: left alone, the compiler folds the lane offset into a
    // per-lane copy of the field pointer and adds the row offset with a 64-bit VALU add per access)
    // (it stays a GLOBAL pointer through the asm -- a generic one would turn the accesses into flat_load / flat_store)
    typedef const char __attribute__((address_space(1))) *gcptr;
    // The byte offset of the row in hand is carried from tick to tick (the rows of a band come one after the
    // other): the 64-bit row * pitch product is a chain of ten scalar instructions in front of every tick's
    // loads, which small grids -- one or two waves per SIMD, ticks bound by latency -- paid with 8 %.
    const long long pitch_b = (long long)g.pitch * (long long)sizeof(T);
    int cur_row = ROLE == 2 ? ra : min(tau0, tau1 - 1);
    long long cur_off = (long long)at(g, cur_row, 0) * (long long)sizeof(T);
    auto row_ptr = [&](const T *field, int i) {
        if (__builtin_expect(i == cur_row + 1, 1)) cur_off += pitch_b;            // the tick loop: the next row
        else if (i != cur_row) cur_off = (long long)at(g, i, 0) * (long long)sizeof(T);
        cur_row = i;
        gcptr rp = (gcptr)(reinterpret_cast<const char *>(field) + cur_off);
        asm("" : "+s"(rp));             // (not volatile: volatile statements keep their order among the DPP ones)
        return (const char *)rp;
    };
    // (and the lane offset through one as a vector register at each use: hoisted out of the loop as a 64-bit
    // value it is no longer the zero-extended 32-bit offset the scalar-base addressing form takes)
    // (in place, so that it keeps its own register: a copy lands in a register of the slot about to be loaded,
    // and the compiler then waits for that slot's previous load -- the whole prefetch -- before writing it)
    auto lane_off_now = [&]() { asm("" : "+v"(lane_off)); return lane_off; };
    auto load_global = [&](Row &r, int i) {
        const int ic = min(i, tau1 - 1);
        const unsigned lo = lane_off_now();
        r.e = ldn<V>(reinterpret_cast<const T *>(row_ptr(p.ez_in, ic) + lo));
        r.x = ldn<V>(reinterpret_cast<const T *>(row_ptr(p.hx_in, ic) + lo));
        r.y = ldn<V>(reinterpret_cast<const T *>(row_ptr(p.hy_in, ic) + lo));
        if (CE_ARR) r.ce = ldn<V>(reinterpret_cast<const T *>(row_ptr(p.ce, ic) + lo));
        if (CH_ARR) r.ch = ldn<V>(reinterpret_cast<const T *>(row_ptr(p.ch, ic) + lo));
        if (GENERAL) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                r.e.v[v] = m.ld_ok ? r.e.v[v] : T(0);
                r.x.v[v] = m.ld_ok ? r.x.v[v] : T(0);
                r.y.v[v] = m.ld_ok ? r.y.v[v] : T(0);
                if (CE_ARR) r.ce.v[v] = (m.ld_ok && m.m_e[v]) ? r.ce.v[v] : T(0);
                if (CH_ARR) r.ch.v[v] = (m.ld_ok && m.m_h[v]) ? r.ch.v[v] : T(0);
            }
        }
    };
    // (Measured and rejected, profiles/r03_rejected.txt: 2 waves x 8 levels -- half the hand-offs, 6 workgroups per CU at 3 waves
    // per SIMD: 1420 vs 1295 us at 16384^2, 440 vs 366 at 8192^2, 153 vs 126 at 4096^2 -- and s_setprio 3 for the loading wave, the one the others wait for at the
    // tick barrier -- 127.5 vs 126.5-132 us at 4096^2, 366 vs 364 at 8192^2, 1308 vs 1302-1320 at 16384^2: nothing.)
    if (ROLE == 0) {
#pragma unroll
        for (int k = 0; k < PF; ++k) load_global(slot[k], tau0 + k);
    }
    // (the other waves start from the zeroed buffers: their first input rows are far above the cone)

    for (int tb = tau0; tb < tend; tb += S) {
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const int tau = tb + k;
            if (tau >= tend) break;                 // same trip count in every wave of the workgroup
            const int r = tau - shift;              // input row of this wave, level t0; lives in slot k
            if (ROLE == 0) {
                load_global(slot[(k + PF) % S], r + PF);
            } else {                                // row r: written by wave w-1 in the previous tick
                const int dr = (tau + 1) & 1;
                Row &in = slot[k];
                in.e = *buf(w - 1, dr, 0);
                in.x = *buf(w - 1, dr, 1);
                in.y = *buf(w - 1, dr, 2);
                if (CE_ARR) in.ce = *buf(w - 1, dr, 3);
                if (CH_ARR) in.ch = *buf(w - 1, dr, NF - 1);
            }
            if (__builtin_expect(r < r_end, 1)) {
#pragma unroll
                for (int l = 1; l <= LV; ++l) {
                    if (__builtin_expect(r < first[l], 0)) continue;     // outside the cone / beyond a short pass
                    Row &c = slot[(k - l + 2 * S) % S];
                    m.level(c, slot[(k - l + 1 + 2 * S) % S].e, slot[(k - l - 1 + 2 * S) % S].x, t0 + l, r - l);
                }
            }
            const Row &f = slot[(k - LV + 2 * S) % S];       // row r - LV, now at level t0 + LV
            if (ROLE == 2) {
                const int io = r - LV;
                if (io >= ra && io < rb) {            // a row of the band (uniform) ...
                    char *pe = const_cast<char *>(row_ptr(p.ez_out, io)), *px = const_cast<char *>(row_ptr(p.hx_out, io)),
                         *py = const_cast<char *>(row_ptr(p.hy_out, io));
                    if (st_ok) {                      // ... and a column this strip owns (lanes masked off otherwise)
                        const unsigned lo = lane_off_now();
                        stn<V>(reinterpret_cast<T *>(pe + lo), f.e);
                        stn<V>(reinterpret_cast<T *>(px + lo), f.x);
                        stn<V>(reinterpret_cast<T *>(py + lo), f.y);
                    }
                }
            } else {
                const int d = tau & 1;
                if (own) {          // (neighbouring windows overlap: each lane of the joint row has ONE writer)
                    *buf(w, d, 0) = f.e;
                    *buf(w, d, 1) = f.x;
                    *buf(w, d, 2) = f.y;
                    if (CE_ARR) *buf(w, d, 3) = f.ce;
                    if (CH_ARR) *buf(w, d, NF - 1) = f.ch;
                }
            }
#ifdef FDTD2D_TRACE
            const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
            __syncthreads();
            trace_bar += __builtin_amdgcn_s_memtime() - tb0;
#else
            __syncthreads();
#endif
        }
    }
#ifdef FDTD2D_TRACE
    if (p.trace && lane == 0 && w < 4 && side == 0) p.trace[8 * (size_t)blockIdx.x + 4 + w] = trace_bar;
#endif
}

// FUSE: the zone tiles are the first workgroups of the launch and share its LDS allocation (a
// workgroup is either a tile or a strip).  That saves the side-stream k_zone launch and its two
// cross-stream event waits per pass; the build without the zone code serves zone_split = 1.
template <class T, int NT, int SPLIT_NW, bool FUSE, bool CE_ARR = false, bool CH_ARR = false, int V = Vec<T>::N, int SD = 1>
__global__ __launch_bounds__(64 * SPLIT_NW * SD) void k_bulk_split(const PassParams<T> p)
{
    static_assert(NT % SPLIT_NW == 0, "levels must divide evenly over the waves");
    static_assert(SD == 1 || !FUSE, "strips of several waves side by side take their zone tiles from k_zone");
    constexpr int SW = 64 * V, LV = NT / SPLIT_NW;
    constexpr int NF = 3 + (CE_ARR ? 1 : 0) + (CH_ARR ? 1 : 0);
    // one LDS allocation, used either as the hand-off buffers of a strip or as a zone tile
    constexpr int HAND = (SPLIT_NW - 1) * HAND_DEPTH * NF * side_units(LV, V, SD);      // VecN units
    constexpr int ZONE_ELEMS = zone_in_registers<T, NT>() ? zone_xch_elems<T, 64 * SPLIT_NW>() : ZoneDims<NT>::LDS_ELEMS;
    constexpr int ZONE = !FUSE ? 0 : (ZONE_ELEMS * (int)sizeof(T) + (int)sizeof(VecN<T, V>) - 1) / (int)sizeof(VecN<T, V>);
    // (several waves side by side: the joint rows can exceed the 64 KB a static allocation may have)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
    __shared__ VecN<T, V> lds_static[SD > 1 ? 1 : (HAND > ZONE ? HAND : ZONE)];
    VecN<T, V> *const lds = SD > 1 ? reinterpret_cast<VecN<T, V> *>(lds_dyn) : lds_static;
    int b = blockIdx.x;
#ifdef FDTD2D_TRACE
    TraceScope trace(p.trace);
#endif
    if constexpr (FUSE) {    // zone tiles are the first workgroups of the launch (all NW waves per tile)
        const int nzone = p.zone_wgs;
        const int bz = p.zone_last ? b - ((int)gridDim.x - nzone) : b;      // index among the zone workgroups (first or last in the launch)
        if (bz >= 0 && bz < nzone) {
            b = bz;
            if constexpr (zone_in_registers<T, NT>()) {      // two waves per tile, rows in registers (kernels_zone.hpp)
                zone_wave_group<T, NT, CE_ARR, CH_ARR, 64 * SPLIT_NW>(p, b, reinterpret_cast<T *>(lds));
            } else {
                const int z = b / p.zone_tiles;
                zone_body<T, NT, CE_ARR, CH_ARR, 64 * SPLIT_NW>(p, b - z * p.zone_tiles, p.zone_top ? z == 1 : true,
                                                                reinterpret_cast<T *>(lds));
            }
            return;
        }
        if (!p.zone_last) b -= nzone;
    }
    int strip, ra, rb;
    if (!strip_of_block(p, b, &strip, &ra, &rb)) return;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int w = SD == 1 ? wid : wid % SPLIT_NW, side = SD == 1 ? 0 : wid / SPLIT_NW;
    // the columns of THIS wave's window decide which body it runs: only the waves at the grid's edges and on the
    // source columns take the general one
    const int x0 = strip_x0<T, NT, V, SD, LV>(p, strip) + side * ((64 - 2 * side_lov(LV, V, SD)) * V);
    const bool edge = x0 < 5 || x0 + SW > p.g.C - 5;
    const bool src = p.src_row1 > ra - 2 * NT && p.src_row < rb + NT && p.src_col1 > x0 &&
                     p.src_col < x0 + SW;
#ifdef FDTD2D_TRACE
    trace.kind = (edge || src) ? 1 : 2;
#endif
    // zero the hand-off buffers: the first ticks read rows nobody has written yet
    for (int n = threadIdx.x; n < HAND; n += 64 * SPLIT_NW * SD)
#pragma unroll
        for (int v = 0; v < V; ++v) lds[n].v[v] = T(0);
    __syncthreads();
    if (edge || src) {
        if (w == 0) split_body<T, NT, SPLIT_NW, CE_ARR, CH_ARR, true, 0, V, SD>(p, strip, ra, rb, w, side, lds);
        else if (w == SPLIT_NW - 1) split_body<T, NT, SPLIT_NW, CE_ARR, CH_ARR, true, 2, V, SD>(p, strip, ra, rb, w, side, lds);
        else split_body<T, NT, SPLIT_NW, CE_ARR, CH_ARR, true, 1, V, SD>(p, strip, ra, rb, w, side, lds);
    } else {
        if (w == 0) split_body<T, NT, SPLIT_NW, CE_ARR, CH_ARR, false, 0, V, SD>(p, strip, ra, rb, w, side, lds);
        else if (w == SPLIT_NW - 1) split_body<T, NT, SPLIT_NW, CE_ARR, CH_ARR, false, 2, V, SD>(p, strip, ra, rb, w, side, lds);
        else split_body<T, NT, SPLIT_NW, CE_ARR, CH_ARR, false, 1, V, SD>(p, strip, ra, rb, w, side, lds);
    }
}

}  // namespace fdtd

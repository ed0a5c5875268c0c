// libfdtd2d.so -- host side of the C ABI declared in include/fdtd2d.h.
// MI355X (gfx950) only.  No CPU fallback: every compute entry point needs the device.
#include "engine.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace fdtd_host {
thread_local std::string g_create_error = "";
}

using namespace fdtd_host;

namespace {

int use_device(fdtd2d *h)
{
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess)
        return fail(h, FDTD2D_E_NODEVICE, "hipSetDevice(%d): %s", h->device, hipGetErrorString(e));
    return 0;
}

template <class T> T round_to(double x) { return (T)x; }

// dt/(x*dx) in T, as main.py:27,70,74 evaluate it for arrays of type T
template <class T> double coef_of(double x, double dt, double dx)
{
    const T xt = (T)x, dtt = (T)dt, dxt = (T)dx;
    volatile T prod = xt * dxt;
    volatile T q = dtt / prod;
    return (double)q;
}

// (c*dt - dx)/(c*dt + dx), c = 1/sqrt(mu00*eps00), every operation in T (main.py:30-31)
template <class T> double mur_of(double eps00, double mu00, double dt, double dx)
{
    const T e = (T)eps00, m = (T)mu00, dtt = (T)dt, dxt = (T)dx;
    volatile T prod = m * e;
    volatile T s = std::sqrt((T)prod);
    volatile T c = (T)1 / s;
    volatile T cdt = c * dtt;
    volatile T num = cdt - dxt, den = cdt + dxt;
    volatile T k = num / den;
    return (double)k;
}

double get_elem(const void *p, int dt, size_t i)
{
    return dt == FDTD2D_F64 ? ((const double *)p)[i] : (double)((const float *)p)[i];
}

// Byte offset of a field's first element inside its allocation.  64 would put the strips of the
// 16-step pass (which begin 16 columns left of a multiple of 224) on 128-B line boundaries; measured
// on one box against 0, us per 16-step pass: 4096^2 157.0 vs 156.5, 8192^2 445.7 vs 444.0, 16384^2
// 1510 vs 1483 (profiles/r02_field_shift.txt) -- the partly used lines are not what the 16384^2 pass
// waits for, so the fields stay on the allocation's own alignment.
constexpr size_t FIELD_SHIFT = 0, FIELD_GUARD = 256;

int alloc_field(fdtd2d *h, void **p)
{
    char *raw = nullptr;
    if (hipMalloc((void **)&raw, h->field_bytes + FIELD_GUARD + FIELD_SHIFT) != hipSuccess)
        return fail(h, FDTD2D_E_NOMEM, "hipMalloc of %zu bytes failed", h->field_bytes);
    if (hipMemsetAsync(raw, 0, h->field_bytes + FIELD_GUARD + FIELD_SHIFT, h->stream) != hipSuccess) {
        (void)hipFree(raw);
        return fail(h, FDTD2D_E_NOMEM, "hipMemset of a new field failed");
    }
    *p = raw + FIELD_SHIFT;
    return 0;
}

void free_field(void **p)
{
    if (*p) (void)hipFree((char *)*p - FIELD_SHIFT);
    *p = nullptr;
}

int need_scratch(fdtd2d *h, size_t bytes)
{
    if (h->scratch_bytes >= bytes) return 0;
    if (h->scratch) (void)hipFree(h->scratch);
    h->scratch = nullptr;
    h->scratch_bytes = 0;
    if (hipMalloc(&h->scratch, bytes) != hipSuccess) return fail(h, FDTD2D_E_NOMEM, "scratch allocation failed");
    h->scratch_bytes = bytes;
    return 0;
}

// rows per staging chunk when host and engine types differ (64 MiB of the wider type)
int convert_chunk_rows(int host_cols) { return std::max(1, (int)((64u << 20) / ((size_t)host_cols * 8))); }

// Copy host rows (host_cols elements each, of host_dtype) into device rows starting at
// stored row `srow`.  When the types differ the rows are staged on the device in the host's type
// and converted there (no host-side loop over the elements).
int copy_in(fdtd2d *h, void *dev, const void *host, int host_dtype, int srow, int nrows,
            int host_cols)
{
    if (nrows <= 0) return 0;
    char *d = (char *)dev + (size_t)srow * h->pitch * h->esz;
    if (host_dtype == h->dtype) {
        HIPCHK(h, hipMemcpy2D(d, h->pitch * h->esz, host, (size_t)host_cols * h->esz,
                              (size_t)host_cols * h->esz, nrows, hipMemcpyHostToDevice));
        return 0;
    }
    const size_t hsz = host_dtype == FDTD2D_F64 ? 8 : 4;
    const int chunk = std::min(nrows, convert_chunk_rows(host_cols));
    int rc = need_scratch(h, (size_t)chunk * host_cols * hsz);
    if (rc) return rc;
    for (int r = 0; r < nrows; r += chunk) {
        const int n = std::min(chunk, nrows - r);
        const size_t cnt = (size_t)n * host_cols;
        HIPCHK(h, hipMemcpyAsync(h->scratch, (const char *)host + (size_t)r * host_cols * hsz, cnt * hsz,
                                 hipMemcpyHostToDevice, h->stream));
        const unsigned blocks = (unsigned)std::min<size_t>((cnt + 255) / 256, 4096);
        char *dd = d + (size_t)r * h->pitch * h->esz;
        if (h->dtype == FDTD2D_F32)
            hipLaunchKernelGGL((fdtd::k_convert2d<double, float>), dim3(blocks), dim3(256), 0, h->stream,
                               (const double *)h->scratch, (size_t)host_cols, (float *)dd, (size_t)h->pitch, n, host_cols);
        else
            hipLaunchKernelGGL((fdtd::k_convert2d<float, double>), dim3(blocks), dim3(256), 0, h->stream,
                               (const float *)h->scratch, (size_t)host_cols, (double *)dd, (size_t)h->pitch, n, host_cols);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipStreamSynchronize(h->stream));      // the staging buffer is reused by the next chunk
    }
    return 0;
}

int copy_out(fdtd2d *h, const void *dev, void *host, int host_dtype, int srow, int nrows,
             int host_cols)
{
    if (nrows <= 0) return 0;
    const char *d = (const char *)dev + (size_t)srow * h->pitch * h->esz;
    if (host_dtype == h->dtype) {
        HIPCHK(h, hipMemcpy2D(host, (size_t)host_cols * h->esz, d, h->pitch * h->esz,
                              (size_t)host_cols * h->esz, nrows, hipMemcpyDeviceToHost));
        return 0;
    }
    const size_t hsz = host_dtype == FDTD2D_F64 ? 8 : 4;
    const int chunk = std::min(nrows, convert_chunk_rows(host_cols));
    int rc = need_scratch(h, (size_t)chunk * host_cols * hsz);
    if (rc) return rc;
    for (int r = 0; r < nrows; r += chunk) {
        const int n = std::min(chunk, nrows - r);
        const size_t cnt = (size_t)n * host_cols;
        const unsigned blocks = (unsigned)std::min<size_t>((cnt + 255) / 256, 4096);
        const char *dd = d + (size_t)r * h->pitch * h->esz;
        if (h->dtype == FDTD2D_F32)
            hipLaunchKernelGGL((fdtd::k_convert2d<float, double>), dim3(blocks), dim3(256), 0, h->stream,
                               (const float *)dd, (size_t)h->pitch, (double *)h->scratch, (size_t)host_cols, n, host_cols);
        else
            hipLaunchKernelGGL((fdtd::k_convert2d<double, float>), dim3(blocks), dim3(256), 0, h->stream,
                               (const double *)dd, (size_t)h->pitch, (float *)h->scratch, (size_t)host_cols, n, host_cols);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync((char *)host + (size_t)r * host_cols * hsz, h->scratch, cnt * hsz,
                                 hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return 0;
}

int zero_fields(fdtd2d *h)
{
    for (void *p : {h->ez[0], h->ez[1], h->hxb[0], h->hxb[1], h->hyb[0], h->hyb[1]})
        HIPCHK(h, hipMemsetAsync(p, 0, h->field_bytes + FIELD_GUARD, h->stream));
    for (void *p : {h->ezxb[0], h->ezxb[1]})
        if (p) HIPCHK(h, hipMemsetAsync(p, 0, h->field_bytes + FIELD_GUARD, h->stream));
    h->cur = 0;
    h->hcur = 0;
    h->ev = h->hv = Range{h->store_lo(), h->store_hi()};
    h->step = 0;
    return 0;
}

// ---- launches --------------------------------------------------------------------------

template <class T> int launch_h(fdtd2d *h, int lo, int hi)
{
    if (hi <= lo) return 0;
    constexpr int V = fdtd::Vec<T>::N, RPT = 4;
    const Geom g = h->geom();
    dim3 block(64, 4);
    dim3 grid((unsigned)((h->cols - 1 + 64 * V - 1) / (64 * V)),
              (unsigned)((hi - lo + 4 * RPT - 1) / (4 * RPT)));
    const T *ez = (const T *)h->ez[h->cur];
    if (h->boundary == FDTD2D_BOUNDARY_PML) {
        const fdtd::PmlFactors<T> f = pml_factors<T>(h);
        if (h->ch_uniform)
            hipLaunchKernelGGL((fdtd::k_update_h_pml<T, false, RPT>), grid, block, 0, h->stream, ez,
                               (T *)h->hx(), (T *)h->hy(), (const T *)nullptr, (T)h->ch_u, f, g, lo, hi);
        else
            hipLaunchKernelGGL((fdtd::k_update_h_pml<T, true, RPT>), grid, block, 0, h->stream, ez,
                               (T *)h->hx(), (T *)h->hy(), (const T *)h->ch, (T)0, f, g, lo, hi);
        HIPCHK(h, hipGetLastError());
        h->step_launches++;
        return 0;
    }
    if (h->ch_uniform)
        hipLaunchKernelGGL((fdtd::k_update_h<T, false, RPT>), grid, block, 0, h->stream, ez,
                           (T *)h->hx(), (T *)h->hy(), (const T *)nullptr, (T)h->ch_u, g, lo, hi);
    else
        hipLaunchKernelGGL((fdtd::k_update_h<T, true, RPT>), grid, block, 0, h->stream, ez,
                           (T *)h->hx(), (T *)h->hy(), (const T *)h->ch, (T)0, g, lo, hi);
    HIPCHK(h, hipGetLastError());
    h->step_launches++;
    return 0;
}

template <class T, bool CE_ARR> int launch_e_impl(fdtd2d *h, int lo, int hi)
{
    constexpr int V = fdtd::Vec<T>::N, RPT = 4;
    const Geom g = h->geom();
    const T *ez_old = (const T *)h->ez[h->cur];
    T *ez_new = (T *)h->ez[h->cur ^ 1];
    const T *ce = CE_ARR ? (const T *)h->ce : (const T *)nullptr;
    const T ce_u = CE_ARR ? (T)0 : (T)h->ce_u;
    dim3 block(64, 4);
    dim3 grid((unsigned)((h->cols + 64 * V - 1) / (64 * V)),
              (unsigned)((hi - lo + 4 * RPT - 1) / (4 * RPT)));
    if (h->boundary == FDTD2D_BOUNDARY_PML) {
        hipLaunchKernelGGL((fdtd::k_update_e_pml<T, CE_ARR, RPT>), grid, block, 0, h->stream, ez_old,
                           ez_new, (T *)h->ezx(), (const T *)h->hx(), (const T *)h->hy(), ce, ce_u,
                           pml_factors<T>(h), g, lo, hi);
        HIPCHK(h, hipGetLastError());
        h->cur ^= 1;
        h->step_launches++;
        return 0;
    }
    hipLaunchKernelGGL((fdtd::k_update_e<T, CE_ARR, RPT>), grid, block, 0, h->stream, ez_old,
                       ez_new, (const T *)h->hx(), (const T *)h->hy(), ce, ce_u, g, lo, hi);
    HIPCHK(h, hipGetLastError());
    if (h->boundary == FDTD2D_BOUNDARY_MUR5) {
        const int has_top = lo == 0, has_bot = hi == h->rows;
        const int vlo = has_top ? std::max(lo, 5) : lo;
        const int vhi = has_bot ? std::min(hi, h->rows - 5) : hi;
        long long n = (vhi > vlo ? (long long)(vhi - vlo) * 16 : 0) +
                      (long long)(has_top + has_bot) * 5 * h->cols;
        if (n > 0) {
            fdtd::FrameCtx<T, CE_ARR> f{{ez_old, (const T *)h->hx(), (const T *)h->hy(), ce, ce_u, g,
                                         g.R, g.C}, (T)h->k_mur};
            hipLaunchKernelGGL((fdtd::k_frame_mur<T, CE_ARR>), dim3((unsigned)((n + 255) / 256)),
                               dim3(256), 0, h->stream, f, ez_new, lo, hi, has_top, has_bot);
            HIPCHK(h, hipGetLastError());
        }
    }
    h->cur ^= 1;
    h->step_launches++;
    return 0;
}

template <class T> int launch_e(fdtd2d *h, int lo, int hi)
{
    if (hi <= lo) return 0;
    return h->ce_uniform ? launch_e_impl<T, false>(h, lo, hi) : launch_e_impl<T, true>(h, lo, hi);
}

template <class T> int launch_point(fdtd2d *h, int row, int col, double amp)
{
    const Geom g = h->geom();
    // rows of the source rectangle that are current on this slab
    const int r0 = std::max(row, h->ev.lo), r1 = std::min(row + h->src_rows, h->ev.hi);
    if (r0 >= r1) return 0;
    const int n = (r1 - r0) * h->src_cols;
    hipLaunchKernelGGL((fdtd::k_add_point<T>), dim3((n + 255) / 256), dim3(n == 1 ? 1 : 256), 0, h->stream,
                       (T *)h->ez[h->cur], g, r0, col, r1 - r0, h->src_cols, amp);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int need_ready(fdtd2d *h)
{
    if (!h) return FDTD2D_E_ARG;
    if (!h->have_mat)
        return fail(h, FDTD2D_E_STATE, "materials not set: call fdtd2d_set_materials first");
    if (h->boundary == FDTD2D_BOUNDARY_PML && !h->have_pml)
        return fail(h, FDTD2D_E_STATE, "PML factors not set: call fdtd2d_set_pml first");
    return use_device(h);
}

// the loop entry points refuse to run an unstable configuration (the assert of fdtd.py:28)
int need_stable(fdtd2d *h)
{
    int rc = need_ready(h);
    if (rc) return rc;
    const double c = fdtd2d_courant(h);
    if (c > 1.0)
        return fail(h, FDTD2D_E_COURANT, "Courant stability condition not met: %.17g > 1.0", c);
    return 0;
}

int do_update_h(fdtd2d *h)
{
    // Hx[i] needs Ez[i], Ez[i+1] (main.py:69-70): the H range shrinks to where both are current
    const int lo = std::max(h->hv.lo, h->ev.lo);
    const int hi = std::min(h->hv.hi, h->ev.hi == h->rows ? h->rows : h->ev.hi - 1);
    if (hi <= lo) return fail(h, FDTD2D_E_STATE, "no rows with current Ez left: refresh the halo");
    const int khi = std::min(hi, h->rows - 1);   // row R-1 of Hx/Hy is never updated
    int rc = h->dtype == FDTD2D_F32 ? launch_h<float>(h, lo, khi) : launch_h<double>(h, lo, khi);
    if (rc) return rc;
    h->hv = Range{lo, hi};
    return 0;
}

int do_update_e(fdtd2d *h)
{
    // Ez[i] needs Hx[i-1], Hx[i], Hy[i] (main.py:21-27)
    const int lo = std::max(h->ev.lo, h->hv.lo == 0 ? 0 : h->hv.lo + 1);
    const int hi = std::min(h->ev.hi, h->hv.hi);
    if (hi <= lo) return fail(h, FDTD2D_E_STATE, "no rows with current H left: refresh the halo");
    if (h->boundary == FDTD2D_BOUNDARY_MUR5) {
        // the horizontal bands need rows 0..5 / R-6..R-1 together
        if ((lo > 0 && lo < 6) || (hi < h->rows && hi > h->rows - 6))
            return fail(h, FDTD2D_E_STATE, "current rows [%d,%d) cut through the Mur band", lo, hi);
    }
    int rc = h->dtype == FDTD2D_F32 ? launch_e<float>(h, lo, hi) : launch_e<double>(h, lo, hi);
    if (rc) return rc;
    h->ev = Range{lo, hi};
    h->step++;
    return 0;
}

// the half-step path: Ez[probe] after the step that just finished
int probe_after_step(fdtd2d *h)
{
    const long long idx = h->step - 1 - h->probe_step0;
    if (!h->probe_cap || idx < 0 || idx >= h->probe_cap || h->probe_row < h->ev.lo || h->probe_row >= h->ev.hi)
        return 0;
    const size_t off = fdtd::at(h->geom(), h->probe_row, h->probe_col);
    if (h->dtype == FDTD2D_F32)
        hipLaunchKernelGGL((fdtd::k_probe_copy<float>), dim3(1), dim3(1), 0, h->stream, (const float *)h->ez[h->cur], off, h->probe_dev, idx);
    else
        hipLaunchKernelGGL((fdtd::k_probe_copy<double>), dim3(1), dim3(1), 0, h->stream, (const double *)h->ez[h->cur], off, h->probe_dev, idx);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// the running Fourier transform: one sample if the step that just finished is a sampled one
int dft_after_step(fdtd2d *h)
{
    if (!h->dft_n || (h->step - h->dft_step0) % h->dft_every != 0) return 0;
    const int lo = h->dft_lo(), hi = h->dft_hi();
    if (hi <= lo) return 0;
    if (lo < h->ev.lo || hi > h->ev.hi) return fail(h, FDTD2D_E_STATE, "the transform's window rows are not current");
    fdtd::DftPhasors ph;
    const double t = (double)h->step * h->dt;
    for (int k = 0; k < h->dft_n; ++k) {
        ph.c[k] = std::cos(h->dft_omega[k] * t);
        ph.s[k] = -std::sin(h->dft_omega[k] * t);
    }
    const size_t cells = (size_t)(hi - lo) * h->dft_cols;
    const unsigned blocks = (unsigned)std::min<size_t>((cells + 255) / 256, 4096);
    if (h->dtype == FDTD2D_F32)
        hipLaunchKernelGGL((fdtd::k_dft<float>), dim3(blocks), dim3(256), 0, h->stream, (const float *)h->ez[h->cur], h->geom(), lo,
                           hi - lo, h->dft_col0, h->dft_cols, h->dft_n, ph, h->dft_acc);
    else
        hipLaunchKernelGGL((fdtd::k_dft<double>), dim3(blocks), dim3(256), 0, h->stream, (const double *)h->ez[h->cur], h->geom(), lo,
                           hi - lo, h->dft_col0, h->dft_cols, h->dft_n, ph, h->dft_acc);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int do_add_point(fdtd2d *h, int row, int col, double amp)
{
    if (row < 0 || row + h->src_rows > h->rows || col < 0 || col + h->src_cols > h->cols)
        return fail(h, FDTD2D_E_ARG, "source (%d,%d)+%dx%d outside the %dx%d grid", row, col,
                    h->src_rows, h->src_cols, h->rows, h->cols);
    return h->dtype == FDTD2D_F32 ? launch_point<float>(h, row, col, amp)
                                  : launch_point<double>(h, row, col, amp);
}


}  // namespace

namespace fdtd_host {
// Can a pass of nt steps run from the current state?  Fills the row range of the bulk.
bool pass_geometry(const fdtd2d *h, int nt, int *band_lo, int *band_hi)
{
    if (h->probe_cap && h->boundary == FDTD2D_BOUNDARY_PML)
        return false;          // no probe tile for these: the run falls back to shorter passes / single steps
    if (h->boundary == FDTD2D_BOUNDARY_PML) {
        // k_pass_pml: 8-step passes; k_bulk_split + k_bulk_split_pml: 16 steps (float32); uniform mu,
        // bands over all rows (no zones)
        if ((nt != 8 && !h->pml_split(nt)) || nt > h->max_nt || !h->ch_uniform || !h->have_pml) return false;
        if (h->rows < 64 || h->cols < 64) return false;
        const int a = std::max(h->ev.lo, h->hv.lo), b = std::min(h->ev.hi, h->hv.hi);
        const int lo = h->top() ? 0 : a + nt, hi = h->bottom() ? h->rows : b - nt;
        if (h->top() && a > 0) return false;
        if (h->bottom() && b < h->rows) return false;
        if (hi - lo < 1 || lo > h->row0 || hi < h->row0 + h->nrows) return false;
        *band_lo = lo;
        *band_hi = hi;
        return true;
    }
    if (h->boundary != FDTD2D_BOUNDARY_MUR5 || nt < 1 || nt > h->max_nt) return false;
    const int zo = 5 + nt, zr = zo + nt + 1;
    if (h->rows < 2 * zr || h->cols < 16) return false;   // small grids use the single-step path
    // A pass consumes nt rows of validity on every interior side: from rows [a, b) current
    // at time n it can produce rows [a+nt, b-nt) at time n+nt (all of them, halo rows
    // included, so that several passes can follow one halo exchange).
    const int a = std::max(h->ev.lo, h->hv.lo), b = std::min(h->ev.hi, h->hv.hi);
    const int lo = h->top() ? zo : a + nt;
    const int hi = h->bottom() ? h->rows - zo : b - nt;
    if (hi - lo < 1) return false;
    if (h->top() && a > 0) return false;
    if (h->bottom() && b < h->rows) return false;
    if (lo > std::max(h->row0, h->top() ? zo : 0) || hi < std::min(h->row0 + h->nrows, h->bottom() ? h->rows - zo : h->rows))
        return false;                                  // owned rows would not be covered
    if (!h->top() && lo - nt < 5) return false;        // bulk rows never touch the top/bottom
    if (!h->bottom() && hi + nt > h->rows - 5) return false;   // 5-row Mur band at any level
    *band_lo = lo;
    *band_hi = hi;
    return true;
}

}  // namespace fdtd_host

namespace {

template <class T> int make_coef(fdtd2d *h, void *arr)
{
    const size_t n = (size_t)h->stored * h->pitch;
    hipLaunchKernelGGL((fdtd::k_coef<T>), dim3(2048), dim3(256), 0, h->stream, (T *)arr, n,
                       (T)h->dt, (T)h->dx);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// min, max and positivity of the stored rows [srow, srow+nrows) x cols of a device array of type T
template <class T>
int scan_device(fdtd2d *h, const void *arr, int srow, int nrows, double *mn, double *mx, bool *positive)
{
    const int blocks = 1024;
    int rc = need_scratch(h, (size_t)blocks * 3 * sizeof(double));
    if (rc) return rc;
    hipLaunchKernelGGL((fdtd::k_minmax<T>), dim3(blocks), dim3(256), 0, h->stream,
                       (const T *)arr + (size_t)srow * h->pitch, (double *)h->scratch, (size_t)h->pitch, nrows, h->cols);
    HIPCHK(h, hipGetLastError());
    std::vector<double> part((size_t)blocks * 3);
    HIPCHK(h, hipMemcpyAsync(part.data(), h->scratch, part.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double a = 1e300, b = -1e300, bad = 0;
    for (int k = 0; k < blocks; ++k) {
        a = std::min(a, part[3 * k]);
        b = std::max(b, part[3 * k + 1]);
        bad += part[3 * k + 2];
    }
    *mn = a;
    *mx = b;
    *positive = bad == 0;
    return 0;
}

int create_impl(fdtd2d_t **out, int rows, int cols, int row0, int nrows, int halo, double dt,
                double dx, int dtype, int boundary, int device)
{
    if (!out) return fail(nullptr, FDTD2D_E_ARG, "out is NULL");
    *out = nullptr;
    if (rows < 11 || cols < 11)
        return fail(nullptr, FDTD2D_E_ARG,
                    "grid %dx%d is below the 11x11 minimum of the 5-px Mur band", rows, cols);
    if (dtype != FDTD2D_F32 && dtype != FDTD2D_F64)
        return fail(nullptr, FDTD2D_E_ARG, "dtype must be FDTD2D_F32 or FDTD2D_F64");
    if (boundary != FDTD2D_BOUNDARY_NONE && boundary != FDTD2D_BOUNDARY_MUR5 &&
        boundary != FDTD2D_BOUNDARY_PML)
        return fail(nullptr, FDTD2D_E_ARG, "unknown boundary %d", boundary);
    if (!(dt > 0) || !(dx > 0)) return fail(nullptr, FDTD2D_E_ARG, "dt and dx must be positive");
    if (row0 < 0 || nrows <= 0 || row0 + nrows > rows || halo < 0)
        return fail(nullptr, FDTD2D_E_ARG, "slab [%d,%d) does not fit a %d-row grid", row0,
                    row0 + nrows, rows);
    const bool whole = (row0 == 0 && nrows == rows);
    if (!whole) {
        if (halo < 1) return fail(nullptr, FDTD2D_E_ARG, "a slab with neighbours needs halo >= 1");
        // the horizontal Mur bands (6 rows deep incl. their inputs) must sit inside one slab,
        // and a neighbour's halo must not reach into them
        if ((row0 > 0 && row0 - halo < 6) || (row0 + nrows < rows && row0 + nrows + halo > rows - 6))
            return fail(nullptr, FDTD2D_E_ARG,
                        "slab [%d,%d) with halo %d reaches into the 6-row boundary band of a "
                        "neighbouring slab", row0, row0 + nrows, halo);
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, FDTD2D_E_NODEVICE,
                    "no HIP device available (%s); libfdtd2d has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev)
        return fail(nullptr, FDTD2D_E_ARG, "device %d out of range (%d visible)", device, ndev);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess)
        return fail(nullptr, FDTD2D_E_NODEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, FDTD2D_E_NODEVICE, "device %d is %s; this library is built for gfx950",
                    device, prop.gcnArchName);

    fdtd2d *h = new fdtd2d();
    h->rows = rows; h->cols = cols; h->row0 = row0; h->nrows = nrows;
    h->halo = whole ? 0 : halo;
    h->dt = dt; h->dx = dx; h->dtype = dtype; h->boundary = boundary; h->device = device;
    h->esz = dtype == FDTD2D_F32 ? 4 : 8;
    h->pitch = ((long long)cols + 63) / 64 * 64;
    h->stored = nrows + 2 * h->halo;
    h->field_bytes = (size_t)h->stored * h->pitch * h->esz;
    int rc = use_device(h);
    auto bail = [&](int code) {
        g_create_error = h->err;
        fdtd2d_destroy(h);
        return code;
    };
    if (rc) return bail(rc);
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(h, FDTD2D_E_NODEVICE, "hipStreamCreate failed"));
    h->stream = h->own_stream;
    if (hipEventCreate(&h->t0) != hipSuccess || hipEventCreate(&h->t1) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(h, FDTD2D_E_NODEVICE, "hipEventCreate / hipStreamCreate failed"));
    for (void **p : {&h->ez[0], &h->ez[1], &h->hxb[0], &h->hxb[1], &h->hyb[0], &h->hyb[1]}) {
        // (+256 B guard: the last lane of a row may look one vector past the row end)
        if ((rc = alloc_field(h, p))) return bail(rc);
    }
    if (hipMalloc(&h->trash, fdtd::TRASH_SLOTS * fdtd::TRASH_SLOT_BYTES) != hipSuccess)
        return bail(fail(h, FDTD2D_E_NOMEM, "hipMalloc of the scratch line failed"));
    if (boundary == FDTD2D_BOUNDARY_PML) {
        const size_t fb = (size_t)(h->rows + h->cols) * 4 * h->esz;
        if (alloc_field(h, &h->ezxb[0]) || alloc_field(h, &h->ezxb[1]) || hipMalloc(&h->pml, fb) != hipSuccess)
            return bail(fail(h, FDTD2D_E_NOMEM, "hipMalloc of the PML arrays failed"));
    }
#ifdef FDTD2D_TRACE
    if (hipMalloc(&h->trace_dev, (size_t)(1 << 16) * 64) != hipSuccess || hipMemset(h->trace_dev, 0, (size_t)(1 << 16) * 64) != hipSuccess)
        return bail(fail(h, FDTD2D_E_NOMEM, "hipMalloc of the trace buffer failed"));
#endif
    rc = zero_fields(h);
    if (rc) return bail(rc);
    if (hipStreamSynchronize(h->stream) != hipSuccess)
        return bail(fail(h, FDTD2D_E_NODEVICE, "device sync failed after allocation"));
    *out = h;
    return 0;
}

// Upload eps / mu for the stored rows, find min / uniformity ON THE DEVICE (the Courant number of
// fdtd.py:25-26 needs min(eps), min(mu); a constant array is replaced by its scalar coefficient),
// then turn the arrays that stay into coefficient arrays in place.
template <class T>
int set_materials_impl(fdtd2d *h, const void *eps, const void *mu, int host_dtype,
                       const double *corner, int allow_uniform)
{
    const int slo = h->store_lo(), shi = h->store_hi();
    double eps00, mu00;
    if (corner) {
        eps00 = corner[0];
        mu00 = corner[1];
    } else if (slo == 0) {
        eps00 = get_elem(eps, host_dtype, 0);
        mu00 = get_elem(mu, host_dtype, 0);
    } else {
        return fail(h, FDTD2D_E_ARG, "corner {eps[0,0], mu[0,0]} is required for a slab that does "
                                     "not store global row 0");
    }
    h->have_mat = false;
    auto setup = [&](void **arr, const void *host, double *mn, bool *uniform, double *cu) -> int {
        int rc;
        if (!*arr && (rc = alloc_field(h, arr))) return rc;
        else HIPCHK(h, hipMemsetAsync(*arr, 0, h->field_bytes + FIELD_GUARD, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const int srow = slo - h->row_base();
        if ((rc = copy_in(h, *arr, host, host_dtype, srow, shi - slo, h->cols))) return rc;
        double mx;
        bool positive;
        if ((rc = scan_device<T>(h, *arr, srow, shi - slo, mn, &mx, &positive))) return rc;
        if (!positive) return fail(h, FDTD2D_E_ARG, "eps and mu must be positive");
        *uniform = (*mn == mx) && allow_uniform;
        if (*uniform) {
            free_field(arr);
            *cu = coef_of<T>(*mn, h->dt, h->dx);
            return 0;
        }
        return make_coef<T>(h, *arr);
    };
    double emin = 0, mmin = 0;
    int rc = setup(&h->ce, eps, &emin, &h->ce_uniform, &h->ce_u);
    if (rc) return rc;
    rc = setup(&h->ch, mu, &mmin, &h->ch_uniform, &h->ch_u);
    if (rc) return rc;
    h->eps_min = emin;
    h->mu_min = mmin;
    h->k_mur = mur_of<T>(eps00, mu00, h->dt, h->dx);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_mat = true;
    return 0;
}

int halo_rows(fdtd2d *h, int side, bool pack, int *first)
{
    if (side != 0 && side != 1) return fail(h, FDTD2D_E_ARG, "side must be 0 (top) or 1 (bottom)");
    if (h->halo == 0) return fail(h, FDTD2D_E_STATE, "this handle has no halo");
    if ((side == 0 && h->top()) || (side == 1 && h->bottom()))
        return fail(h, FDTD2D_E_STATE, "no neighbour on side %d", side);
    if (pack) *first = side == 0 ? h->row0 : h->row0 + h->nrows - h->halo;
    else *first = side == 0 ? h->row0 - h->halo : h->row0 + h->nrows;
    return 0;
}

template <class T, bool PACK> int launch_halo(fdtd2d *h, int first, void *buf)
{
    const size_t n = (size_t)h->nfields() * h->halo * h->cols;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    if (h->boundary == FDTD2D_BOUNDARY_PML) {
        hipLaunchKernelGGL((fdtd::k_halo4<T, PACK>), dim3(blocks), dim3(256), 0, h->stream,
                           (T *)h->ez[h->cur], (T *)h->ezx(), (T *)h->hx(), (T *)h->hy(), (T *)buf, h->geom(),
                           first, h->halo);
        HIPCHK(h, hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL((fdtd::k_halo<T, PACK>), dim3(blocks), dim3(256), 0, h->stream,
                       (T *)h->ez[h->cur], (T *)h->hx(), (T *)h->hy(), (T *)buf, h->geom(), first,
                       h->halo);
    HIPCHK(h, hipGetLastError());
    return 0;
}

}  // namespace

// ===================================== C ABI ============================================

extern "C" {

const char *fdtd2d_version(void)
{
#ifdef FDTD2D_FUSED
    return "fdtd2d-mi355x 0.3 (gfx950, fused multiply-add: tolerance build)";
#else
    return "fdtd2d-mi355x 0.3 (gfx950, one rounding per operation: value-identical build)";
#endif
}

int fdtd2d_create(fdtd2d_t **out, int rows, int cols, double dt, double dx, int dtype,
                  int boundary, int device)
{
    return create_impl(out, rows, cols, 0, rows, 0, dt, dx, dtype, boundary, device);
}

int fdtd2d_create_slab(fdtd2d_t **out, int rows, int cols, int row0, int nrows, int halo,
                       double dt, double dx, int dtype, int boundary, int device)
{
    return create_impl(out, rows, cols, row0, nrows, halo, dt, dx, dtype, boundary, device);
}

void fdtd2d_destroy(fdtd2d_t *h)
{
    if (!h) return;
    (void)fdtd2d_slab_detach(h);
    if (hipSetDevice(h->device) == hipSuccess) {
        if (h->stream && h->stream != h->own_stream) (void)hipStreamSynchronize(h->stream);
        if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
        for (void **p : {&h->ez[0], &h->ez[1], &h->hxb[0], &h->hxb[1], &h->hyb[0], &h->hyb[1], &h->ce, &h->ch, &h->ezxb[0], &h->ezxb[1]})
            free_field(p);
        if (h->pml) (void)hipFree(h->pml);
        if (h->scratch) (void)hipFree(h->scratch);
        if (h->trash) (void)hipFree(h->trash);
        if (h->probe_dev) (void)hipFree(h->probe_dev);
        if (h->dft_acc) (void)hipFree(h->dft_acc);
        if (h->clk_dev) (void)hipFree(h->clk_dev);
        if (h->clk_stream) { (void)hipStreamSynchronize(h->clk_stream); (void)hipStreamDestroy(h->clk_stream); }
        if (h->side_stream) { (void)hipStreamSynchronize(h->side_stream); (void)hipStreamDestroy(h->side_stream); }
        if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
        if (h->ev_join) (void)hipEventDestroy(h->ev_join);
        if (h->t0) (void)hipEventDestroy(h->t0);
        if (h->t1) (void)hipEventDestroy(h->t1);
        if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    }
    delete h;
}

const char *fdtd2d_last_error(const fdtd2d_t *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

long long fdtd2d_info(const fdtd2d_t *h, int what)
{
    if (!h) return FDTD2D_E_ARG;
    switch (what) {
    case FDTD2D_INFO_ROWS: return h->rows;
    case FDTD2D_INFO_COLS: return h->cols;
    case FDTD2D_INFO_ROW0: return h->row0;
    case FDTD2D_INFO_NROWS: return h->nrows;
    case FDTD2D_INFO_HALO: return h->halo;
    case FDTD2D_INFO_PITCH: return h->pitch;
    case FDTD2D_INFO_DTYPE: return h->dtype;
    case FDTD2D_INFO_BOUNDARY: return h->boundary;
    case FDTD2D_INFO_DEVICE: return h->device;
    case FDTD2D_INFO_EPS_UNIFORM: return h->have_mat && h->ce_uniform;
    case FDTD2D_INFO_MU_UNIFORM: return h->have_mat && h->ch_uniform;
    case FDTD2D_INFO_E_VALID_LO: return h->ev.lo;
    case FDTD2D_INFO_E_VALID_HI: return h->ev.hi;
    case FDTD2D_INFO_H_VALID_LO: return h->hv.lo;
    case FDTD2D_INFO_H_VALID_HI: return h->hv.hi;
    case FDTD2D_INFO_STEP: return h->step;
    case FDTD2D_INFO_PASS_LAUNCHES: return h->pass_launches;
    case FDTD2D_INFO_STEP_LAUNCHES: return h->step_launches;
    case FDTD2D_INFO_CYCLE_STEPS: return h->cycle_steps();
    case FDTD2D_INFO_LAST_BAND_ROWS: return h->shape_last.band_rows;
    case FDTD2D_INFO_LAST_WAVES: return h->shape_last.waves;
    case FDTD2D_INFO_LAST_EDGE_ROWS: return h->shape_last.edge_rows;
    case FDTD2D_INFO_LAST_PASS_STEPS: return h->last_nt;
    case FDTD2D_INFO_LAST_SIDE_WAVES: return h->shape_last.side;
    case FDTD2D_INFO_LAST_XCD_MAP: return h->shape_last.xcd;
    default: return FDTD2D_E_ARG;
    }
}

int fdtd2d_set_stream(fdtd2d_t *h, void *hip_stream)
{
    if (!h) return FDTD2D_E_ARG;
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return 0;
}

int fdtd2d_set_materials(fdtd2d_t *h, const void *eps, const void *mu, int host_dtype,
                         const double *corner, int allow_uniform)
{
    if (!h) return FDTD2D_E_ARG;
    if (!eps || !mu) return fail(h, FDTD2D_E_ARG, "eps and mu must not be NULL");
    if (host_dtype != FDTD2D_F32 && host_dtype != FDTD2D_F64)
        return fail(h, FDTD2D_E_ARG, "bad host_dtype");
    h->tuned.clear();     // the kernel variant may change with the material layout
    int rc = use_device(h);
    if (rc) return rc;
    return h->dtype == FDTD2D_F32
               ? set_materials_impl<float>(h, eps, mu, host_dtype, corner, allow_uniform)
               : set_materials_impl<double>(h, eps, mu, host_dtype, corner, allow_uniform);
}

int fdtd2d_set_materials_uniform(fdtd2d_t *h, double eps, double mu)
{
    if (!h) return FDTD2D_E_ARG;
    if (!(eps > 0) || !(mu > 0)) return fail(h, FDTD2D_E_ARG, "eps and mu must be positive");
    h->tuned.clear();
    int rc = use_device(h);
    if (rc) return rc;
    for (void **p : {&h->ce, &h->ch}) free_field(p);
    h->ce_uniform = h->ch_uniform = true;
    h->eps_min = eps;
    h->mu_min = mu;
    if (h->dtype == FDTD2D_F32) {
        h->ce_u = coef_of<float>(eps, h->dt, h->dx);
        h->ch_u = coef_of<float>(mu, h->dt, h->dx);
        h->k_mur = mur_of<float>(eps, mu, h->dt, h->dx);
    } else {
        h->ce_u = coef_of<double>(eps, h->dt, h->dx);
        h->ch_u = coef_of<double>(mu, h->dt, h->dx);
        h->k_mur = mur_of<double>(eps, mu, h->dt, h->dx);
    }
    h->have_mat = true;
    return 0;
}

int fdtd2d_set_pml(fdtd2d_t *h, const void *row_factors, const void *col_factors, int host_dtype,
                   int layer_cells)
{
    if (!h) return FDTD2D_E_ARG;
    if (h->boundary != FDTD2D_BOUNDARY_PML) return fail(h, FDTD2D_E_STATE, "handle was not created with FDTD2D_BOUNDARY_PML");
    if (!row_factors || !col_factors) return fail(h, FDTD2D_E_ARG, "factor arrays must not be NULL");
    if (host_dtype != h->dtype) return fail(h, FDTD2D_E_ARG, "PML factors must have the engine's dtype");
    if (layer_cells < 1 || 2 * layer_cells + 3 > std::min(h->rows, h->cols))
        return fail(h, FDTD2D_E_ARG, "a %d-cell layer does not fit a %dx%d grid", layer_cells, h->rows, h->cols);
    h->pml_L = layer_cells;
    // The 16-step pair runs the reference's own update wherever its 16-step cone stays clear of the layer,
    // and inside the layer kernel on rows / strips outside the layers: both are the split update only if
    // every factor is exactly 1 there -- H factors on L .. n-2-L (half-cell positions), E factors on
    // L .. n-1-L.  Arrays without that structure keep the 8-step kernel, which reads them everywhere.
    auto unit_outside = [&](const void *fac, int n) {
        auto at = [&](int which, int i) {
            return h->dtype == FDTD2D_F32 ? (double)((const float *)fac)[(size_t)which * n + i]
                                           : ((const double *)fac)[(size_t)which * n + i];
        };
        for (int i = layer_cells; i <= n - 1 - layer_cells; ++i) {
            if (i <= n - 2 - layer_cells && (at(0, i) != 1.0 || at(1, i) != 1.0)) return false;    // ah, bh
            if (at(2, i) != 1.0 || at(3, i) != 1.0) return false;                                   // ae, be
        }
        return true;
    };
    h->pml_unit_outside = unit_outside(row_factors, h->rows) && unit_outside(col_factors, h->cols);
    int rc = use_device(h);
    if (rc) return rc;
    char *d = (char *)h->pml;
    HIPCHK(h, hipMemcpy(d, row_factors, (size_t)4 * h->rows * h->esz, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d + (size_t)4 * h->rows * h->esz, col_factors, (size_t)4 * h->cols * h->esz,
                        hipMemcpyHostToDevice));
    h->have_pml = true;
    return 0;
}

int fdtd2d_transfer_ezx(fdtd2d_t *h, void *host, int host_dtype, int to_device)
{
    if (!h || !host) return FDTD2D_E_ARG;
    if (!h->ezxb[0]) return fail(h, FDTD2D_E_STATE, "no split field: handle was not created with FDTD2D_BOUNDARY_PML");
    if (host_dtype != FDTD2D_F32 && host_dtype != FDTD2D_F64) return fail(h, FDTD2D_E_ARG, "bad host_dtype");
    int rc = use_device(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return to_device ? copy_in(h, h->ezx(), host, host_dtype, h->halo, h->nrows, h->cols)
                     : copy_out(h, h->ezx(), host, host_dtype, h->halo, h->nrows, h->cols);
}

double fdtd2d_courant(const fdtd2d_t *h)
{
    if (!h || !h->have_mat) return -1.0;
    const double c = 1 / std::sqrt(h->eps_min * h->mu_min);
    return (c * h->dt) / h->dx;
}

int fdtd2d_upload(fdtd2d_t *h, const void *Ez, const void *Hx, const void *Hy, int host_dtype)
{
    if (!h) return FDTD2D_E_ARG;
    if (host_dtype != FDTD2D_F32 && host_dtype != FDTD2D_F64)
        return fail(h, FDTD2D_E_ARG, "bad host_dtype");
    int rc = use_device(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int s = h->halo;   // stored row of the first owned row
    if (Ez && (rc = copy_in(h, h->ez[h->cur], Ez, host_dtype, s, h->nrows, h->cols))) return rc;
    if (Hx && (rc = copy_in(h, h->hx(), Hx, host_dtype, s, h->nrows, h->cols - 1))) return rc;
    const int hy_rows = std::min(h->row0 + h->nrows, h->rows - 1) - h->row0;
    if (Hy && (rc = copy_in(h, h->hy(), Hy, host_dtype, s, hy_rows, h->cols))) return rc;
    // uploaded fields are current on the owned rows; halo rows are not until the next exchange
    const Range owned{h->row0, h->row0 + h->nrows};
    if (Ez) h->ev = owned;
    if (Hx || Hy) h->hv = owned;
    if (h->pend_nt) {          // an uncommitted partial pass refers to the old fields: drop it
        h->pend_nt = 0;
        h->pend_done.clear();
    }
    return 0;
}

int fdtd2d_download(fdtd2d_t *h, void *Ez, void *Hx, void *Hy, int host_dtype)
{
    if (!h) return FDTD2D_E_ARG;
    if (host_dtype != FDTD2D_F32 && host_dtype != FDTD2D_F64)
        return fail(h, FDTD2D_E_ARG, "bad host_dtype");
    int rc = use_device(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int r0 = h->row0, r1 = h->row0 + h->nrows;
    if ((Ez && (h->ev.lo > r0 || h->ev.hi < r1)) || ((Hx || Hy) && (h->hv.lo > r0 || h->hv.hi < r1)))
        return fail(h, FDTD2D_E_STATE, "owned rows [%d,%d) are not current (Ez [%d,%d), H [%d,%d))",
                    r0, r1, h->ev.lo, h->ev.hi, h->hv.lo, h->hv.hi);
    const int s = h->halo;
    if (Ez && (rc = copy_out(h, h->ez[h->cur], Ez, host_dtype, s, h->nrows, h->cols))) return rc;
    if (Hx && (rc = copy_out(h, h->hx(), Hx, host_dtype, s, h->nrows, h->cols - 1))) return rc;
    const int hy_rows = std::min(r1, h->rows - 1) - r0;
    if (Hy && (rc = copy_out(h, h->hy(), Hy, host_dtype, s, hy_rows, h->cols))) return rc;
    return 0;
}

int fdtd2d_reset(fdtd2d_t *h)
{
    if (!h) return FDTD2D_E_ARG;
    int rc = use_device(h);
    if (rc) return rc;
    return zero_fields(h);
}

int fdtd2d_update_h(fdtd2d_t *h)
{
    int rc = need_ready(h);
    return rc ? rc : do_update_h(h);
}

int fdtd2d_update_e(fdtd2d_t *h)
{
    int rc = need_ready(h);
    return rc ? rc : do_update_e(h);
}

int fdtd2d_set_probe(fdtd2d_t *h, int row, int col, long long capacity)
{
    if (!h) return FDTD2D_E_ARG;
    if (capacity < 0 || (capacity > 0 && (row < 0 || row >= h->rows || col < 0 || col >= h->cols)))
        return fail(h, FDTD2D_E_ARG, "probe cell (%d,%d) outside the %dx%d grid, or negative capacity", row, col, h->rows, h->cols);
    int rc = use_device(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->probe_dev) { (void)hipFree(h->probe_dev); h->probe_dev = nullptr; }
    h->probe_cap = 0;
    if (capacity > 0) {
        if (hipMalloc((void **)&h->probe_dev, (size_t)capacity * sizeof(double)) != hipSuccess)
            return fail(h, FDTD2D_E_NOMEM, "hipMalloc of %lld probe samples failed", capacity);
        HIPCHK(h, hipMemsetAsync(h->probe_dev, 0, (size_t)capacity * sizeof(double), h->stream));
        h->probe_row = row;
        h->probe_col = col;
        h->probe_cap = capacity;
        h->probe_step0 = h->step;
    }
    return 0;
}

int fdtd2d_read_probe(fdtd2d_t *h, double *out, long long first, long long count)
{
    if (!h) return FDTD2D_E_ARG;
    if (!h->probe_cap) return fail(h, FDTD2D_E_STATE, "no probe is set");
    if (!out || first < 0 || count < 0 || first + count > h->probe_cap)
        return fail(h, FDTD2D_E_ARG, "samples [%lld,%lld) outside the probe's capacity %lld", first, first + count, h->probe_cap);
    int rc = use_device(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (count) HIPCHK(h, hipMemcpy(out, h->probe_dev + first, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int fdtd2d_set_dft(fdtd2d_t *h, int row0, int col0, int nrows, int ncols, int nfreq, const double *omega, int every)
{
    if (!h) return FDTD2D_E_ARG;
    int rc = use_device(h);
    if (rc) return rc;
    if (h->pend_nt) return fail(h, FDTD2D_E_STATE, "a partial pass is pending: commit it first");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->dft_acc) { (void)hipFree(h->dft_acc); h->dft_acc = nullptr; }
    h->dft_n = 0;
    if (nfreq == 0) return 0;
    if (nfreq < 0 || nfreq > 16 || !omega || every < 1 || nrows < 1 || ncols < 1 || row0 < 0 || col0 < 0 ||
        row0 + nrows > h->rows || col0 + ncols > h->cols)
        return fail(h, FDTD2D_E_ARG, "need 1..16 frequencies, every >= 1 and a window inside the %dx%d grid", h->rows, h->cols);
    h->dft_row0 = row0; h->dft_col0 = col0; h->dft_rows = nrows; h->dft_cols = ncols; h->dft_every = every;
    for (int k = 0; k < nfreq; ++k) h->dft_omega[k] = omega[k];
    h->dft_step0 = h->step;
    const size_t own = (size_t)std::max(0, h->dft_hi() - h->dft_lo()) * ncols;
    if (own) {
        if (hipMalloc((void **)&h->dft_acc, own * 2 * nfreq * sizeof(double)) != hipSuccess)
            return fail(h, FDTD2D_E_NOMEM, "hipMalloc of the transform's accumulators failed");
        HIPCHK(h, hipMemsetAsync(h->dft_acc, 0, own * 2 * nfreq * sizeof(double), h->stream));
    }
    h->dft_n = nfreq;
    return 0;
}

int fdtd2d_read_dft(fdtd2d_t *h, double *re, double *im)
{
    if (!h || !re || !im) return FDTD2D_E_ARG;
    if (!h->dft_n) return fail(h, FDTD2D_E_STATE, "no transform is set");
    int rc = use_device(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t own = (size_t)std::max(0, h->dft_hi() - h->dft_lo()) * h->dft_cols;
    for (int k = 0; k < h->dft_n && own; ++k) {
        HIPCHK(h, hipMemcpy(re + (size_t)k * own, h->dft_acc + (size_t)(2 * k) * own, own * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(im + (size_t)k * own, h->dft_acc + (size_t)(2 * k + 1) * own, own * sizeof(double), hipMemcpyDeviceToHost));
    }
    return 0;
}

int fdtd2d_set_source_extent(fdtd2d_t *h, int nrows, int ncols)
{
    if (!h) return FDTD2D_E_ARG;
    if (nrows < 1 || ncols < 1 || nrows > h->rows || ncols > h->cols)
        return fail(h, FDTD2D_E_ARG, "source extent %dx%d does not fit the %dx%d grid", nrows, ncols, h->rows, h->cols);
    h->src_rows = nrows;
    h->src_cols = ncols;
    return 0;
}

int fdtd2d_add_point(fdtd2d_t *h, int row, int col, double amp)
{
    if (!h) return FDTD2D_E_ARG;
    int rc = use_device(h);
    return rc ? rc : do_add_point(h, row, col, amp);
}

// The next pass of a run with `rem` steps to go: kernel length *nt (the geometry: halo, strip
// overlap, zone depth) and the levels *nlev <= *nt it advances.  Every pass is one sweep over the
// grid whatever its length, so the plan minimises sweeps and never leaves a scrap: a remainder of
// at most one pass runs on the shortest kernel that holds it (level-split kernels stop after nlev
// levels; on very large grids a 20-step kernel takes 17..20 steps in one sweep), a remainder
// between one and two passes is cut in halves, longer runs take full passes.  Returns false when
// no temporally blocked pass is possible from the current state (small grids, an exhausted halo):
// the caller then takes one plain step.
static bool plan_pass(const fdtd2d *h, int rem, int *nt, int *nlev, int *lo, int *hi)
{
    const int C = h->cycle_steps();
    if (C <= 0 || rem <= 0) return false;
    const bool longp = C == 16 && h->long_passes();
    int take = rem;
    if (!(longp && rem <= 20)) {
        if (rem > 2 * C) take = C;
        else if (rem > C) take = (rem + 1) / 2;
    }
    const int lens[] = {1, 2, 4, 8, 16, 20};
    auto avail = [&](int c) { return c <= C || (longp && c == 20); };
    for (int c : lens)                      // shortest kernel that holds `take`
        if (c >= take && avail(c) && pass_geometry(h, c, lo, hi) &&
            (c == take || h->use_level_split(c, *lo, *hi) || h->pml_split(c))) {
            *nt = c;
            *nlev = take;
            return true;
        }
    for (int k = 5; k >= 0; --k)            // else the longest full pass below it
        if (lens[k] < take && avail(lens[k]) && pass_geometry(h, lens[k], lo, hi)) {
            *nt = *nlev = lens[k];
            return true;
        }
    return false;
}

// Measure the launch shape of a large pass once: trial launches write only into the buffers
// the next committed pass overwrites anyway (commit = false), so the state is untouched.
// Candidates: the rule of launch_pass, a ladder of band heights, and for 16-step passes both
// 4 and 8 waves per strip.  Costs about 40 launches the first time a (length, rows) pair is run.
static int tune_pass(fdtd2d *h, int nt, int lo, int hi, bool zt, bool zb, int src_row = 0, int src_col = 0,
                     bool has_src = false)
{
    // (a source changes the launch: its strips run the slower body in short bands of their own, which
    // can push a launch that just filled the GPU's workgroup slots into a second round -- 4096^2:
    // 147 -> 170 us per pass, profiles/r02_source_cost.txt -- so the trials carry the source, with
    // amplitude 0, when the run will)
    static const double zero_amps[fdtd::STREAM_MAX_NT] = {};
    const double *trial_amps = has_src ? zero_amps : nullptr;
    const std::array<int, 3> key{nt, lo, hi};
    if (!h->autotune || h->stream_band_rows > 0 || nt < 8 ||
        (h->boundary != FDTD2D_BOUNDARY_MUR5 && !h->pml_split(nt)) ||
        h->shape_given(nt) ||
        (size_t)std::max(0, hi - lo) * h->cols < ((size_t)4 << 20) || h->tuned.count(key))
        return 0;
    std::vector<fdtd2d::Shape> cand{{0, 0}};
    const bool split = h->use_level_split(nt, lo, hi);
    const std::vector<int> ladder = nt >= 16 ? std::vector<int>{64, 96, 144, 208, 304, 448}
                                             : std::vector<int>{16, 24, 32, 48, 64, 96, 128};
    const bool both_nw = split && nt == 16 && !h->split_waves;
    // waves side by side per level group (strips 256, 504 or 1000 columns wide): the caller's choice or every width
    // the configuration has (float32 16- / 20-step level-split passes, 4 waves per level group)
    std::vector<int> sides{1};
    if (split && !h->pml_split(nt) && (h->split_waves == 0 || h->split_waves == 4)) {
        if (h->side_waves > 1) {
            if (h->side_ok(nt, h->side_waves)) sides = {h->side_waves};
        } else if (h->side_waves == 0) {
            for (int sd : {2, 4})
                if (h->side_ok(nt, sd)) sides.push_back(sd);
        }
    }
    for (int sd : sides)
        for (int nw : {4, 8}) {
            if (nw == 8 && (!both_nw || sd > 1)) continue;
            for (int br : ladder)
                if (br * 4 <= hi - lo) cand.push_back({br, (both_nw || sd > 1) ? nw : 0, 0, sd});
        }
    if (h->pml_split(nt))                       // the PML pair: plain band height x band height of the layer's end strips
        for (int re : {128, 256})
            for (int br : ladder)
                if (br * 4 <= hi - lo && re * 4 <= hi - lo) cand.push_back({br, 0, re});
    // Shapes that fill the GPU's workgroup slots in k whole rounds, with the first / last strip
    // (~2x the work per row) cut into shorter bands: a launch lasts as long as its longest-lived
    // workgroup, and one that needs 1.1 rounds lasts as long as two.  The zone tiles come first in
    // launch order and hold a slot each.  (4096^2, us per 16-step pass: 64-row bands 158; 144-row
    // bands with 32-row edge bands = 1021 workgroups for 1024 slots 147.5; 119 / 32 = 1123
    // workgroups 184: profiles/r02_shape_sweep.txt)
    if (split) {
        const int region = hi - lo, V = h->dtype == FDTD2D_F32 ? 4 : 2;
        for (int sd : sides) {
        const int wt = sd > 1 ? fdtd::strip_width(nt / 4, V, sd) : 64 * V;
        const int ow = wt - 2 * fdtd::stream_hc(nt);
        const int ns = (h->cols + ow - 1) / ow;
        // ZoneDims<NT>::WZ; 20-step passes and strips of several waves take their zones from k_zone beside the bulk
        const int zw = sd > 1 ? 0 : (nt == 16 ? 30 : (nt == 8 ? 14 : 0));
        const int zone_tiles = zw ? ((zt ? 1 : 0) + (zb ? 1 : 0)) * ((h->cols + zw - 1) / zw) : 0;
        int n_src = 0;                           // inner strips that hold source columns (bands of their own)
        for (int st = 1; has_src && st <= ns - 2; ++st) {
            const int x0 = st * ow - fdtd::stream_hc(nt);
            if (src_col + h->src_cols > x0 && src_col < x0 + wt) ++n_src;
        }
        if (n_src > 2) n_src = 0;
        for (int nw : {4, 8}) {
            if (nw == 8 && (!both_nw || sd > 1)) continue;
            // resident workgroups (VGPR / LDS limits): 16 waves per CU
            const int slots = 256 * (nw == 8 ? 2 : 4) / sd;
            // workgroups of the fused zone tiles: 16-step passes keep a tile in the registers of two (float32) or four
            // (float64) waves (kernels_zone.hpp); the LDS tiles take a workgroup each
            const int tpw = (h->dtype == FDTD2D_F32 && nt >= 16) ? nw / 2 : ((h->dtype == FDTD2D_F64 && nt == 16) ? nw / 4 : 1);
            const int zones = (zone_tiles + tpw - 1) / tpw;
            const double fill = 2.0 * nt + nw - 1;
            for (int k : {1, 2, 3, 4}) {
                for (double w_e : {1.0, 1.5, 2.0, 3.0}) {            // edge bands as tall, 1/2, 1/3 as long-lived
                    const double tasks = (double)k * slots - (k == 1 ? zones : 0);
                    fdtd2d::Shape best{0, 0};
                    for (int nb = 1; nb <= region / 8; ++nb) {
                        const double life = (double)region / nb + fill;      // ticks of a plain task
                        const double er = life / w_e - fill;                   // rows of an equally long edge task
                        const int ne = w_e == 1.0 ? nb : (er >= 8 ? (int)std::ceil(region / er) : region / 8);
                        const int brs = std::max(16, std::min((region + ne - 1) / ne, (region + nb - 1) / nb / 3));
                        // workgroups of the source strips: short bands on the ~3 nt rows around the source only
                        const int n_s = n_src * (nb + 1 + (3 * nt + h->src_rows + brs - 1) / brs);
                        if ((double)std::max(0, ns - 2 - n_src) * nb + 2.0 * ne + n_s > tasks) break;
                        best = fdtd2d::Shape{(region + nb - 1) / nb, (both_nw || sd > 1) ? nw : 0,
                                             w_e == 1.0 ? 0 : std::max(8, (region + ne - 1) / ne), sd};
                    }
                    if (best.band_rows >= 8) cand.push_back(best);
                }
                // One round with fused zone tiles: the tiles are done after tz ticks and free their slots.  The last
                // `nshort` bands of every inner strip are tz rows shorter and come last in launch order: they start in
                // those slots and end with the tall bands (4096^2: 750 + 274 bulk workgroups instead of 750).
                const int n_i = std::max(1, ns - 2 - n_src), nshort = zones / n_i;
                for (int tz : {32, 48, 64, 80, 96, 112}) {
                    if (k != 1 || zones == 0 || sd != 1 || nshort < 1) break;
                    for (double w_e : {1.0, 1.5, 2.0, 3.0}) {
                        fdtd2d::Shape best{0, 0};
                        for (int nl = 1; nl <= region / 8; ++nl) {
                            const int rl = (region + nshort * tz + nl + nshort - 1) / (nl + nshort);     // rows of a tall band
                            const int rs = rl - tz;
                            if (rs < 16) break;
                            const double life = rl + fill, er = life / w_e - fill;
                            const int ne = w_e == 1.0 ? (region + rl - 1) / rl : (er >= 8 ? (int)std::ceil(region / er) : region / 8);
                            const int brs = std::max(16, std::min((region + ne - 1) / ne, rl / 3));
                            const int n_s = n_src * (nl + nshort + 1 + (3 * nt + h->src_rows + brs - 1) / brs);
                            if ((double)n_i * nl + 2.0 * ne + n_s > (double)slots - zones) break;
                            best = fdtd2d::Shape{rl, both_nw ? nw : 0, w_e == 1.0 ? 0 : std::max(8, (region + ne - 1) / ne), 1, 0, rs, nshort};
                        }
                        if (best.band_rows >= 8) cand.push_back(best);
                    }
                }
            }
        }
        }
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        return fail(h, FDTD2D_E_NOMEM, "hipEventCreate failed");
    }
    const long long launches = h->pass_launches;
    // The trials alternate direction like the committed passes of a run (set A -> B, then B -> A): launches
    // that always read the same set and write the other rank the shapes differently from a real run (8192^2
    // ring map: 314-row bands x 8 waves 0.53 ms in one-way trials like 157 x 4, 0.57 vs 0.53 ms in a run;
    // profiles/r02_tuner_view.txt).  The current set is kept in a scratch copy meanwhile and put back at the
    // end; without the memory for it the trials stay one-way.
    // (Not while pieces of a pass are pending -- fdtd2d_pass_rows has already written rows next to the cuts
    // into the other set, and a trial in the reverse direction would be followed by trials that overwrite
    // them with later time levels: one-way trials only rewrite this piece's rows with the values they get
    // anyway.)
    const int cur0 = h->cur, hcur0 = h->hcur;
    std::vector<std::pair<void *, void *>> saved;          // (scratch copy, original)
    if (h->pend_nt == 0) {
        std::vector<void *> orig{h->ez[cur0], h->hxb[hcur0], h->hyb[hcur0]};
        if (h->boundary == FDTD2D_BOUNDARY_PML && h->ezxb[hcur0]) orig.push_back(h->ezxb[hcur0]);
        for (void *o : orig) {
            void *c = nullptr;
            if (hipMalloc(&c, h->field_bytes) != hipSuccess ||
                hipMemcpyAsync(c, o, h->field_bytes, hipMemcpyDeviceToDevice, h->stream) != hipSuccess) {
                (void)hipGetLastError();
                if (c) (void)hipFree(c);
                for (auto &sv : saved) (void)hipFree(sv.first);
                saved.clear();
                break;
            }
            saved.push_back({c, o});
        }
    }
    const bool pingpong = !saved.empty();
    auto trial = [&](const fdtd2d::Shape &c, int reps) {
        h->tuned[key] = c;
        int rc = 0;
        for (int n = 0; n < reps && rc == 0; ++n) {
            rc = h->dtype == FDTD2D_F32
                     ? launch_pass<float>(h, nt, lo, hi, src_row, src_col, trial_amps, zt, zb, false, lo, hi)
                     : launch_pass<double>(h, nt, lo, hi, src_row, src_col, trial_amps, zt, zb, false, lo, hi);
            if (pingpong) {
                h->cur ^= 1;
                h->hcur ^= 1;
            }
        }
        return rc;
    };
    // clocks up, code objects loaded: an idle chip needs ~10-20 ms of work before it holds its clock (the first
    // repetitions of a bench run are 10-30 % slower), and candidates measured on the ramp would lose to later ones
    int rc = trial(cand[0], 3);
    {
        hipEvent_t w0 = nullptr, w1 = nullptr;
        if (hipEventCreate(&w0) == hipSuccess && hipEventCreate(&w1) == hipSuccess) {
            float warm_ms = 0;
            (void)hipEventRecord(w0, h->stream);
            for (int n = 0; n < 64 && rc == 0 && warm_ms < 30.0f; ++n) {
                rc = trial(cand[0], 2);
                (void)hipEventRecord(w1, h->stream);
                if (hipEventSynchronize(w1) != hipSuccess || hipEventElapsedTime(&warm_ms, w0, w1) != hipSuccess) break;
            }
        }
        if (w0) (void)hipEventDestroy(w0);
        if (w1) (void)hipEventDestroy(w1);
    }
    auto timed = [&](const fdtd2d::Shape &c, int reps, float *ms_per_launch) {
        (void)hipEventRecord(e0, h->stream);
        int r = trial(c, reps);
        if (r) return r;
        (void)hipEventRecord(e1, h->stream);
        float ms = 0;
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
            return fail(h, FDTD2D_E_STATE, "timing a trial launch failed");
        *ms_per_launch = ms / reps;
        return 0;
    };
    std::vector<float> best_of(cand.size(), 1e30f);
    for (int round = 0; round < 2 && rc == 0; ++round)
        for (size_t n = 0; n < cand.size() && rc == 0; ++n) {
            if (round == 0 && (rc = trial(cand[n], 1))) break;    // first use of this kernel variant
            float ms = 0;
            if ((rc = timed(cand[n], 2, &ms))) break;
            best_of[n] = std::min(best_of[n], ms);
        }
    // Finals: neighbouring shapes differ by a few per cent and a pair of launches scatters by as much (the
    // pick flipped between 4 and 8 waves per strip from run to run on the 8192^2 ring map, 0.50 vs 0.58 ms),
    // so the four fastest are measured again, six launches at a time, three times over.
    std::vector<size_t> order(cand.size());
    for (size_t n = 0; n < order.size(); ++n) order[n] = n;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return best_of[a] < best_of[b]; });
#ifdef FDTD2D_TUNE_LOG
    for (size_t k = 0; k < order.size(); ++k) {
        const fdtd2d::Shape &c = cand[order[k]];
        fprintf(stderr, "tune nt=%d first rounds #%zu: (%d, %d, %d, side %d, xcd %d, filler %d x %d) %.4f ms\n", nt, k, c.band_rows,
                c.waves, c.edge_rows, c.side, c.xcd, c.short_rows, c.n_short, best_of[order[k]]);
    }
#endif
    // the fastest four go to the finals -- together with their twins whose tasks are dealt out XCD by XCD
    // (FDTD2D_OPT_XCD_MAP: worth 3-5 % on launches of several rounds, -7 % on one-round launches) unless the caller
    // has fixed that choice
    {
        // (the fastest four, and the fastest two of every strip width: the widths respond differently to the
        // XCD-wise order -- 16384^2: 1 wave per level group 1549 -> 1467 us, 4 side by side 1507 -> 1507)
        std::vector<fdtd2d::Shape> fin;
        std::vector<float> fin_ms;
        int per_side[5] = {0, 0, 0, 0, 0};
        for (size_t k = 0; k < order.size(); ++k) {
            const fdtd2d::Shape &c = cand[order[k]];
            const int sd = std::min(std::max(c.side, 1), 4);
            if (k < 4 || per_side[sd] < 2) {
                fin.push_back(c);
                fin_ms.push_back(best_of[order[k]]);
            }
            per_side[sd]++;
        }
        if ((split || h->pml_split(nt)) && h->xcd_map < 0)
            for (size_t k = 0, n = fin.size(); k < n; ++k) {
                fdtd2d::Shape t = fin[k];
                t.xcd = 1;
                fin.push_back(t);
                fin_ms.push_back(fin_ms[k]);
            }
        // 20-step float32 passes: every finalist of one wave per level group again with its zone tiles in the bulk launch
        if (nt > 16 && h->dtype == FDTD2D_F32 && split)
            for (size_t k = 0, n = fin.size(); k < n; ++k) {
                if (fin[k].side > 1) continue;
                fdtd2d::Shape t = fin[k];
                t.fuse = 1;
                fin.push_back(t);
                fin_ms.push_back(fin_ms[k]);
            }
        cand = fin;
        best_of = fin_ms;
        order.resize(cand.size());
        for (size_t n = 0; n < order.size(); ++n) order[n] = n;
    }
    fdtd2d::Shape best = cand[order[0]];
    float best_ms = 1e30f;
    const size_t finalists = order.size();
    for (int round = 0; round < 3 && rc == 0; ++round)
        for (size_t k = 0; k < finalists && rc == 0; ++k) {
            float ms = 0;
            if ((rc = timed(cand[order[k]], 6, &ms))) break;
#ifdef FDTD2D_TUNE_LOG      // profiling builds only (tools/): what the tuner saw
            fprintf(stderr, "tune nt=%d final %d: (%d, %d, %d, side %d, xcd %d, filler %d x %d, fuse %d) first rounds %.4f ms, now %.4f ms\n", nt, round,
                    cand[order[k]].band_rows, cand[order[k]].waves, cand[order[k]].edge_rows, cand[order[k]].side,
                    cand[order[k]].xcd, cand[order[k]].short_rows, cand[order[k]].n_short, cand[order[k]].fuse, best_of[order[k]], ms);
#endif
            // (8 waves per strip run 5-7 % slower on a run's real fields than in these trials on the fields at
            // hand -- zero in a fresh engine: 8192^2 ring map 0.53 ms in trials, 0.57 ms in the run, while the
            // 4-wave shapes measure 0.52 both ways, profiles/r02_tuner_view.txt -- so they must win by that much)
            const float score = cand[order[k]].waves == 8 ? ms * 1.06f : ms;
            if (score < best_ms) {
                best_ms = score;
                best = cand[order[k]];
            }
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    h->cur = cur0;
    h->hcur = hcur0;
    for (auto &sv : saved) {
        if (hipMemcpyAsync(sv.second, sv.first, h->field_bytes, hipMemcpyDeviceToDevice, h->stream) != hipSuccess && !rc)
            rc = fail(h, FDTD2D_E_STATE, "restoring the fields after the trial launches failed");
    }
    if (!saved.empty()) (void)hipStreamSynchronize(h->stream);
    for (auto &sv : saved) (void)hipFree(sv.first);
    h->pass_launches = launches;
    h->tuned[key] = rc ? fdtd2d::Shape{0, 0} : best;
    return rc;
}

int fdtd2d_prepare(fdtd2d_t *h, int nsteps) { return fdtd2d_prepare_run(h, nsteps, 0, 0, 0); }

int fdtd2d_prepare_run(fdtd2d_t *h, int nsteps, int src_row, int src_col, int with_source)
{
    int rc = need_stable(h);
    if (rc) return rc;
    if (h->pend_nt) return fail(h, FDTD2D_E_STATE, "a partial pass is pending: commit it first");
    // the pass lengths fdtd2d_run(nsteps) will use from the current state (same decisions as its
    // loop): the long ones are tuned, tail kernels get one uncommitted launch (code objects loaded)
    const long long launches = h->pass_launches;
    auto warm = [&](int c, int lo, int hi, int nlev) {
        return h->dtype == FDTD2D_F32
                   ? launch_pass<float>(h, c, lo, hi, 0, 0, nullptr, h->top(), h->bottom(), false, lo, hi, nlev)
                   : launch_pass<double>(h, c, lo, hi, 0, 0, nullptr, h->top(), h->bottom(), false, lo, hi, nlev);
    };
    int left = nsteps;
    while (left > 0 && rc == 0) {
        int nt = 0, nlev = 0, lo = 0, hi = 0;
        if (!plan_pass(h, left, &nt, &nlev, &lo, &hi)) break;      // single-step kernels: nothing to prepare
        // (a short pass runs on the launch shape of its kernel's full passes: tuned here as well -- inside
        // fdtd2d_run a one-off tail does not pay for 20-120 ms of trial launches, this call is the set-up)
        rc = (nt >= 8 && (h->boundary == FDTD2D_BOUNDARY_MUR5 || h->pml_split(nt))) ? tune_pass(h, nt, lo, hi, h->top(), h->bottom(), src_row, src_col, with_source != 0)
                                                                            : 0;
        if (!rc && nlev != nt) rc = warm(nt, lo, hi, nlev);
        else if (!rc && nt < 8) rc = warm(nt, lo, hi, nlev);
        if (nlev == nt && left >= 2 * nt) left %= nt;               // the full passes of a long run are all alike
        else left -= nlev;
    }
    if (rc) return rc;
    h->pass_launches = launches;
    return hipStreamSynchronize(h->stream) == hipSuccess ? 0 : fail(h, FDTD2D_E_STATE, "stream sync failed");
}

int fdtd2d_run(fdtd2d_t *h, int nsteps, int src_row, int src_col, const double *amps)
{
    int rc = need_stable(h);
    if (rc) return rc;
    if (h->pend_nt) return fail(h, FDTD2D_E_STATE, "a partial pass is pending: commit it first");
    if (nsteps < 0) return fail(h, FDTD2D_E_ARG, "nsteps < 0");
    if (amps && (src_row < 0 || src_row + h->src_rows > h->rows || src_col < 0 ||
                 src_col + h->src_cols > h->cols))
        return fail(h, FDTD2D_E_ARG, "source (%d,%d)+%dx%d outside the %dx%d grid", src_row, src_col,
                    h->src_rows, h->src_cols, h->rows, h->cols);
    int n = 0;
    while (n < nsteps) {
        int nt = 0, nlev = 0, lo = 0, hi = 0;
        if (plan_pass(h, std::min(nsteps - n, h->dft_gap()), &nt, &nlev, &lo, &hi)) {
            const double *a = amps ? amps + n : nullptr;
            // (only full passes are tuned: a one-off tail does not pay for 20-120 ms of trial launches;
            // it uses the shape measured for full passes of its kernel if there is one)
            if (nlev == nt && (rc = tune_pass(h, nt, lo, hi, h->top(), h->bottom(), src_row, src_col, amps != nullptr))) return rc;
            h->probe_pending = h->probe_cap > 0;
            rc = h->dtype == FDTD2D_F32
                     ? launch_pass<float>(h, nt, lo, hi, src_row, src_col, a, h->top(), h->bottom(), true, lo, hi, nlev)
                     : launch_pass<double>(h, nt, lo, hi, src_row, src_col, a, h->top(), h->bottom(), true, lo, hi, nlev);
            if (rc) return rc;
            if ((rc = dft_after_step(h))) return rc;
            n += nlev;
            continue;
        }
        if ((rc = do_update_h(h))) return rc;
        if ((rc = do_update_e(h))) return rc;
        if (amps && (rc = do_add_point(h, src_row, src_col, amps[n]))) return rc;
        if ((rc = probe_after_step(h))) return rc;
        if ((rc = dft_after_step(h))) return rc;
        ++n;
    }
    return 0;
}

int fdtd2d_pass_rows(fdtd2d_t *h, int nt, int row_lo, int row_hi, int src_row, int src_col,
                     const double *amps)
{
    int rc = need_stable(h);
    if (rc) return rc;
    if (h->pend_nt && h->pend_nt != nt)
        return fail(h, FDTD2D_E_STATE, "a %d-step pass is pending; cannot add rows of a %d-step pass",
                    h->pend_nt, nt);
    if (nt != 1 && nt != 2 && nt != 4 && nt != 8 && nt != 16)
        return fail(h, FDTD2D_E_ARG, "pass length must be 1, 2, 4, 8 or 16");
    if (nt == 16 && h->cycle_steps() != 16)
        return fail(h, FDTD2D_E_STATE, "16-step passes need float32 and the Mur frame");
    int lo = 0, hi = 0;
    if (!pass_geometry(h, nt, &lo, &hi))
        return fail(h, FDTD2D_E_STATE, "a %d-step pass is not possible from the current state "
                    "(rows current: Ez [%d,%d), H [%d,%d))", nt, h->ev.lo, h->ev.hi, h->hv.lo, h->hv.hi);
    const int zo = h->boundary == FDTD2D_BOUNDARY_PML ? 0 : 5 + nt;   // no zones with the PML
    const int p_lo = h->top() ? 0 : lo, p_hi = h->bottom() ? h->rows : hi;   // rows a full pass writes
    if (row_lo < p_lo || row_hi > p_hi || row_lo >= row_hi)
        return fail(h, FDTD2D_E_ARG, "rows [%d,%d) are outside what this pass produces, [%d,%d)",
                    row_lo, row_hi, p_lo, p_hi);
    // the top / bottom zone is written as a whole or not at all
    if (h->top() && row_lo != 0 && row_lo < zo) return fail(h, FDTD2D_E_ARG, "rows cut through the top zone [0,%d)", zo);
    if (h->top() && row_lo == 0 && row_hi < zo) return fail(h, FDTD2D_E_ARG, "rows cut through the top zone [0,%d)", zo);
    if (h->bottom() && row_hi != h->rows && row_hi > h->rows - zo)
        return fail(h, FDTD2D_E_ARG, "rows cut through the bottom zone [%d,%d)", h->rows - zo, h->rows);
    if (h->bottom() && row_hi == h->rows && row_lo > h->rows - zo)
        return fail(h, FDTD2D_E_ARG, "rows cut through the bottom zone [%d,%d)", h->rows - zo, h->rows);
    const bool zt = h->top() && row_lo == 0, zb = h->bottom() && row_hi == h->rows;
    const int b_lo = std::max(row_lo, lo), b_hi = std::min(row_hi, hi);
    if ((rc = tune_pass(h, nt, b_lo, b_hi, zt, zb, src_row, src_col, amps != nullptr))) return rc;
    h->probe_pending = h->probe_cap > 0 && h->pend_nt == 0;      // once per pass: with its first piece
    rc = h->dtype == FDTD2D_F32
             ? launch_pass<float>(h, nt, b_lo, b_hi, src_row, src_col, amps, zt, zb, false, lo, hi)
             : launch_pass<double>(h, nt, b_lo, b_hi, src_row, src_col, amps, zt, zb, false, lo, hi);
    if (rc) return rc;
    h->pend_nt = nt;
    h->pend_done.push_back(Range{row_lo, row_hi});
    return 0;
}

int fdtd2d_pass_commit(fdtd2d_t *h)
{
    if (!h) return FDTD2D_E_ARG;
    if (!h->pend_nt) return fail(h, FDTD2D_E_STATE, "no partial pass is pending");
    const int nt = h->pend_nt;
    int lo = 0, hi = 0;
    if (!pass_geometry(h, nt, &lo, &hi)) return fail(h, FDTD2D_E_STATE, "state changed under a pending pass");
    const int p_lo = h->top() ? 0 : lo, p_hi = h->bottom() ? h->rows : hi;
    // the pieces must cover the owned rows without a gap; rows of [p_lo, p_hi) outside the
    // pieces (halo rows a full pass would also have advanced) simply stop being current
    std::sort(h->pend_done.begin(), h->pend_done.end(), [](const Range &a, const Range &b) { return a.lo < b.lo; });
    const int own_lo = h->row0, own_hi = h->row0 + h->nrows;
    int c_lo = h->pend_done.empty() ? p_hi : std::max(p_lo, h->pend_done.front().lo), at = c_lo;
    for (const Range &r : h->pend_done) {
        if (r.lo > at) break;
        at = std::max(at, r.hi);
    }
    at = std::min(at, p_hi);
    if (c_lo > own_lo || at < own_hi) {
        h->pend_nt = 0;
        h->pend_done.clear();
        return fail(h, FDTD2D_E_STATE, "pending pass covers rows [%d,%d) of the owned [%d,%d): dropped",
                    c_lo, at, own_lo, own_hi);
    }
    // (a running Fourier transform samples at fixed steps: a pass issued in pieces must end ON the next sampled step or
    // before it -- fdtd2d_run cuts its passes there by itself, callers of fdtd2d_pass_rows choose nt accordingly)
    const bool skipped = h->dft_n && h->dft_gap() < nt;
    h->cur ^= 1;
    h->hcur ^= 1;
    h->ev = h->hv = Range{c_lo, at};
    h->step += nt;
    h->pend_nt = 0;
    h->pend_done.clear();
    if (skipped)
        return fail(h, FDTD2D_E_STATE, "the %d-step pass ran over a step the running Fourier transform samples (every %d steps)",
                    nt, h->dft_every);
    return dft_after_step(h);
}

double fdtd2d_source_amplitude(int src_kind, double t, double fc)
{
    const double pi = 3.141592653589793;
    if (src_kind == FDTD2D_SRC_RICKER) {
        const double tau = pi * fc * (t - 1 / fc);
        return (1 - 2 * (tau * tau)) * std::exp(-(tau * tau));
    }
    if (src_kind == FDTD2D_SRC_SINUSOIDAL) {
        const double d = t - 3000 / fc, w = 2 / fc;
        const double envelope = 1 - std::exp(-(d * d) / (2 * (w * w)));
        return envelope * std::sin(2 * pi * fc * t);
    }
    return 0.0;
}

int fdtd2d_run_waveform(fdtd2d_t *h, int nsteps, int src_kind, int src_row, int src_col,
                        double fc, long long step0)
{
    if (!h) return FDTD2D_E_ARG;
    if (nsteps < 0) return fail(h, FDTD2D_E_ARG, "nsteps < 0");
    if (src_kind == FDTD2D_SRC_NONE) return fdtd2d_run(h, nsteps, 0, 0, nullptr);
    if (src_kind != FDTD2D_SRC_RICKER && src_kind != FDTD2D_SRC_SINUSOIDAL)
        return fail(h, FDTD2D_E_ARG, "unknown source kind %d", src_kind);
    std::vector<double> amps((size_t)nsteps);
    for (int n = 0; n < nsteps; ++n)
        amps[n] = fdtd2d_source_amplitude(src_kind, (double)(step0 + n) * h->dt, fc);
    return fdtd2d_run(h, nsteps, src_row, src_col, amps.data());
}

int fdtd2d_set_option(fdtd2d_t *h, int option, long long value)
{
    if (!h) return FDTD2D_E_ARG;
    switch (option) {
    case FDTD2D_OPT_MAX_PASS_STEPS:
        if (value < 0 || value > fdtd::STREAM_MAX_NT) return fail(h, FDTD2D_E_ARG, "pass length must be 0..20");
        h->max_nt = (int)value;
        h->max_nt_forced = true;
        return 0;
    case FDTD2D_OPT_BAND_ROWS:
        if (value < 0) return fail(h, FDTD2D_E_ARG, "band rows must be >= 0");
        h->stream_band_rows = (int)value;
        return 0;
    case FDTD2D_OPT_LEVEL_SPLIT:
        if (value < -1 || value > 1) return fail(h, FDTD2D_E_ARG, "level split must be -1, 0 or 1");
        h->level_split = (int)value;
        h->tuned.clear();
        return 0;
    case FDTD2D_OPT_AUTOTUNE:
        h->autotune = value != 0;
        h->tuned.clear();
        return 0;
    case FDTD2D_OPT_SPLIT_WAVES:
        if (value != 0 && value != 4 && value != 8) return fail(h, FDTD2D_E_ARG, "split waves must be 0, 4 or 8");
        h->split_waves = (int)value;
        h->tuned.clear();
        return 0;
    case FDTD2D_OPT_ZONE_SPLIT:
        if (value < -1 || value > 1) return fail(h, FDTD2D_E_ARG, "zone split must be -1, 0 or 1");
        h->zone_split = (int)value;
        h->tuned.clear();
        return 0;
    case FDTD2D_OPT_XCD_MAP:
        if (value < -1 || value > 1) return fail(h, FDTD2D_E_ARG, "xcd map must be -1, 0 or 1");
        h->xcd_map = (int)value;
        h->tuned.clear();
        return 0;
    case FDTD2D_OPT_SIDE_WAVES:
        if (value != 0 && value != 1 && value != 2 && value != 4) return fail(h, FDTD2D_E_ARG, "side waves must be 0, 1, 2 or 4");
        h->side_waves = (int)value;
        h->tuned.clear();
        return 0;
    default: return fail(h, FDTD2D_E_ARG, "unknown option %d", option);
    }
}

int fdtd2d_set_shape(fdtd2d_t *h, int pass_steps, const int *shape, int n)
{
    if (!h || !shape || n < 1) return FDTD2D_E_ARG;
    int v[FDTD2D_SHAPE_LEN] = {0, 0, 0, 1, 0, 0, 0, 0};
    for (int k = 0; k < n && k < FDTD2D_SHAPE_LEN; ++k) v[k] = shape[k];
    if (v[3] == 0) v[3] = 1;
    if (pass_steps < 0 || pass_steps > fdtd::STREAM_MAX_NT || v[0] < 0 || (v[1] != 0 && v[1] != 4 && v[1] != 8) || v[2] < 0 ||
        (v[3] != 1 && v[3] != 2 && v[3] != 4) || (v[4] != 0 && v[4] != 1) || v[5] < 0 || v[6] < 0 ||
        (v[7] != 0 && v[7] != 1))
        return fail(h, FDTD2D_E_ARG, "shape = {band rows, waves 0|4|8, edge band rows, side 1|2|4, xcd 0|1, filler rows, fillers per strip, "
                                     "zone tiles fused 0|1}");
    h->given_shape[pass_steps] = fdtd2d::Shape{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    return 0;
}

int fdtd2d_last_shape(const fdtd2d_t *h, int *shape, int n)
{
    if (!h || !shape || n < 1) return FDTD2D_E_ARG;
    const fdtd2d::Shape &s = h->shape_last;
    const int v[FDTD2D_SHAPE_LEN] = {s.band_rows, s.waves, s.edge_rows, s.side, s.xcd, s.short_rows, s.n_short, s.fuse};
    for (int k = 0; k < n; ++k) shape[k] = k < FDTD2D_SHAPE_LEN ? v[k] : 0;
    return 0;
}

int fdtd2d_sync(fdtd2d_t *h)
{
    if (!h) return FDTD2D_E_ARG;
    int rc = use_device(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
#ifdef FDTD2D_TRACE      // profiling build: dump the stamps of the last level-split launch
    if (const char *path = std::getenv("FDTD2D_TRACE_FILE")) {
        if (h->trace_blocks > 0) {
            std::vector<unsigned long long> t((size_t)h->trace_blocks * 8);
            HIPCHK(h, hipMemcpy(t.data(), h->trace_dev, t.size() * 8, hipMemcpyDeviceToHost));
            if (FILE *f = std::fopen(path, "w")) {
                for (long long b = 0; b < h->trace_blocks; ++b)
                    std::fprintf(f, "%lld %llu %llu %llu %llu %llu %llu %llu %llu\n", b, t[8 * b], t[8 * b + 1], t[8 * b + 2],
                                 t[8 * b + 3], t[8 * b + 4], t[8 * b + 5], t[8 * b + 6], t[8 * b + 7]);
                std::fclose(f);
            }
        }
    }
#endif
    return 0;
}

long long fdtd2d_halo_bytes(const fdtd2d_t *h)
{
    return h ? (long long)h->nfields() * h->halo * h->cols * (long long)h->esz : FDTD2D_E_ARG;
}

int fdtd2d_halo_pack(fdtd2d_t *h, int side, void *dev_buf)
{
    if (!h || !dev_buf) return FDTD2D_E_ARG;
    int rc = use_device(h), first = 0;
    if (rc || (rc = halo_rows(h, side, true, &first))) return rc;
    if (h->pend_nt) {
        // pack the rows of the pending (not yet committed) pass: they must have been issued
        bool ok = false;
        for (const Range &r : h->pend_done) ok = ok || (r.lo <= first && r.hi >= first + h->halo);
        if (!ok) return fail(h, FDTD2D_E_STATE, "rows [%d,%d) of the pending pass were not issued yet",
                             first, first + h->halo);
        h->cur ^= 1;
        h->hcur ^= 1;
        rc = h->dtype == FDTD2D_F32 ? launch_halo<float, true>(h, first, dev_buf)
                                    : launch_halo<double, true>(h, first, dev_buf);
        h->cur ^= 1;
        h->hcur ^= 1;
        return rc;
    }
    const int r0 = h->row0, r1 = h->row0 + h->nrows;
    if (h->ev.lo > r0 || h->ev.hi < r1 || h->hv.lo > r0 || h->hv.hi < r1)
        return fail(h, FDTD2D_E_STATE, "owned rows are not current; cannot pack a halo message");
    if (h->nrows < h->halo) return fail(h, FDTD2D_E_STATE, "slab thinner than its halo");
    return h->dtype == FDTD2D_F32 ? launch_halo<float, true>(h, first, dev_buf)
                                  : launch_halo<double, true>(h, first, dev_buf);
}

int fdtd2d_halo_unpack(fdtd2d_t *h, int side, const void *dev_buf)
{
    if (!h || !dev_buf) return FDTD2D_E_ARG;
    int rc = use_device(h), first = 0;
    if (rc || (rc = halo_rows(h, side, false, &first))) return rc;
    rc = h->dtype == FDTD2D_F32 ? launch_halo<float, false>(h, first, (void *)dev_buf)
                                : launch_halo<double, false>(h, first, (void *)dev_buf);
    if (rc) return rc;
    // the rows just written extend the current range, provided the owned rows adjoin it
    if (side == 0) {
        if (h->ev.lo <= h->row0) h->ev.lo = first;
        if (h->hv.lo <= h->row0) h->hv.lo = first;
    } else {
        if (h->ev.hi >= h->row0 + h->nrows) h->ev.hi = first + h->halo;
        if (h->hv.hi >= h->row0 + h->nrows) h->hv.hi = first + h->halo;
    }
    return 0;
}

int fdtd2d_snapshot_index(fdtd2d_t *h, double vmin, double vmax, int stride, unsigned char *out)
{
    if (!h || !out) return FDTD2D_E_ARG;
    if (stride < 1 || !(vmax > vmin)) return fail(h, FDTD2D_E_ARG, "need stride >= 1 and vmax > vmin");
    int rc = use_device(h);
    if (rc) return rc;
    const int r0 = h->row0, r1 = h->row0 + h->nrows;
    if (h->ev.lo > r0 || h->ev.hi < r1) return fail(h, FDTD2D_E_STATE, "owned rows of Ez are not current");
    const int first = ((r0 + stride - 1) / stride) * stride;          // global rows i with i % stride == 0
    const int nro = first < r1 ? (r1 - 1 - first) / stride + 1 : 0;
    const int nco = (h->cols - 1) / stride + 1;
    if (nro == 0) return 0;
    const size_t bytes = (size_t)nro * nco;
    if ((rc = need_scratch(h, bytes))) return rc;
    dim3 grid((unsigned)((nco + 255) / 256), (unsigned)nro);
    if (h->dtype == FDTD2D_F32)
        hipLaunchKernelGGL((fdtd::k_snapshot<float>), grid, dim3(256), 0, h->stream, (const float *)h->ez[h->cur],
                           (unsigned char *)h->scratch, h->geom(), first, nro, nco, stride, (float)vmin,
                           (float)vmax, (float)(vmax - vmin));
    else
        hipLaunchKernelGGL((fdtd::k_snapshot<double>), grid, dim3(256), 0, h->stream, (const double *)h->ez[h->cur],
                           (unsigned char *)h->scratch, h->geom(), first, nro, nco, stride, vmin, vmax,
                           vmax - vmin);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, h->scratch, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int fdtd2d_reduce(fdtd2d_t *h, int field, double *sum_sq, double *max_abs)
{
    if (!h) return FDTD2D_E_ARG;
    int rc = use_device(h);
    if (rc) return rc;
    const void *f = fdtd2d_device_ptr(h, field);
    if (!f) return fail(h, FDTD2D_E_ARG, "unknown field %d", field);
    const int r0 = h->row0, r1 = h->row0 + h->nrows;
    const Range &v = field == FDTD2D_FIELD_EZ ? h->ev : h->hv;
    if (v.lo > r0 || v.hi < r1) return fail(h, FDTD2D_E_STATE, "owned rows are not current");
    const int blocks = 1024;
    if ((rc = need_scratch(h, (size_t)blocks * 2 * sizeof(double)))) return rc;
    if (h->dtype == FDTD2D_F32)
        hipLaunchKernelGGL((fdtd::k_reduce<float>), dim3(blocks), dim3(256), 0, h->stream, (const float *)f,
                           (double *)h->scratch, h->geom(), r0, h->nrows, h->cols);
    else
        hipLaunchKernelGGL((fdtd::k_reduce<double>), dim3(blocks), dim3(256), 0, h->stream, (const double *)f,
                           (double *)h->scratch, h->geom(), r0, h->nrows, h->cols);
    HIPCHK(h, hipGetLastError());
    std::vector<double> part((size_t)blocks * 2);
    HIPCHK(h, hipMemcpyAsync(part.data(), h->scratch, part.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double s = 0, m = 0;
    for (int b = 0; b < blocks; ++b) {
        s += part[2 * b];
        m = std::max(m, part[2 * b + 1]);
    }
    if (sum_sq) *sum_sq = s;
    if (max_abs) *max_abs = m;
    return 0;
}

int fdtd2d_timer_start(fdtd2d_t *h)
{
    if (!h) return FDTD2D_E_ARG;
    int rc = use_device(h);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(h->t0, h->stream));
    return 0;
}

int fdtd2d_timer_stop(fdtd2d_t *h, float *ms)
{
    if (!h || !ms) return FDTD2D_E_ARG;
    int rc = use_device(h);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(h->t1, h->stream));
    HIPCHK(h, hipEventSynchronize(h->t1));
    HIPCHK(h, hipEventElapsedTime(ms, h->t0, h->t1));
    return 0;
}

int fdtd2d_time_launches(fdtd2d_t *h, int nlaunch, int steps_each, float *ms)
{
    int rc = need_ready(h);
    if (rc) return rc;
    if (!ms || nlaunch < 1 || nlaunch > 256 || steps_each < 1) return fail(h, FDTD2D_E_ARG, "bad arguments");
    std::vector<hipEvent_t> ev((size_t)2 * nlaunch, nullptr);
    auto cleanup = [&]() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); };
    for (hipEvent_t &e : ev)
        if (hipEventCreate(&e) != hipSuccess) { cleanup(); return fail(h, FDTD2D_E_NOMEM, "hipEventCreate failed"); }
    for (int n = 0; n < nlaunch && rc == 0; ++n) {
        if (hipEventRecord(ev[2 * n], h->stream) != hipSuccess) rc = fail(h, FDTD2D_E_STATE, "hipEventRecord failed");
        if (!rc) rc = fdtd2d_run(h, steps_each, 0, 0, nullptr);
        if (!rc && hipEventRecord(ev[2 * n + 1], h->stream) != hipSuccess) rc = fail(h, FDTD2D_E_STATE, "hipEventRecord failed");
    }
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, FDTD2D_E_STATE, "stream sync failed");
    for (int n = 0; n < nlaunch && rc == 0; ++n)
        if (hipEventElapsedTime(&ms[n], ev[2 * n], ev[2 * n + 1]) != hipSuccess) rc = fail(h, FDTD2D_E_STATE, "hipEventElapsedTime failed");
    cleanup();
    return rc;
}

namespace {
constexpr int CLK_PROBES = 16;
// one wave per workgroup: sleeps `ticks` of the 100 MHz counter, stamps the shader-cycle counter at both ends
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long *out, unsigned long long ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r = r0;
    while (r - r0 < ticks) {
        __builtin_amdgcn_s_sleep(64);
        r = __builtin_amdgcn_s_memrealtime();
    }
    unsigned long long *q = out + 3 * (size_t)blockIdx.x;
    q[0] = __builtin_amdgcn_s_memtime() - c0;
    q[1] = r - r0;
    q[2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 11) | 20) & 0xf;     // HW_REG_XCC_ID, bits 3:0
}
}  // namespace

namespace {
// 8 x 16 bytes per lane, all eight loads in flight before the first store (what a streaming copy needs to come near the
// rate the memory system sustains); one workgroup per 32 KiB
__global__ __launch_bounds__(256) void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    const size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x;
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = base + 256 * k < n ? src[base + 256 * k] : uint4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (base + 256 * k < n) dst[base + 256 * k] = v[k];
}
}  // namespace

int fdtd2d_measure_copy(fdtd2d_t *h, int reps, double *gbps)
{
    int rc = need_stable(h);
    if (rc) return rc;
    if (!gbps || reps < 1 || reps > 1000) return fail(h, FDTD2D_E_ARG, "bad arguments");
    if (h->pend_nt) return fail(h, FDTD2D_E_STATE, "a partial pass is pending: commit it first");
    const size_t n = h->field_bytes / 16;
    const void *src[3] = {h->ez[h->cur], h->hxb[h->hcur], h->hyb[h->hcur]};
    void *dst[3] = {h->ez[h->cur ^ 1], h->hxb[h->hcur ^ 1], h->hyb[h->hcur ^ 1]};
    auto once = [&]() {
        for (int f = 0; f < 3; ++f)
            hipLaunchKernelGGL(k_copy16, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, h->stream, (const uint4 *)src[f], (uint4 *)dst[f], n);
    };
    once();                                  // first touch
    HIPCHK(h, hipEventRecord(h->t0, h->stream));
    for (int r = 0; r < reps; ++r) once();
    HIPCHK(h, hipEventRecord(h->t1, h->stream));
    HIPCHK(h, hipEventSynchronize(h->t1));
    HIPCHK(h, hipGetLastError());
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->t0, h->t1));
    *gbps = ms > 0 ? 2.0 * 3.0 * (double)(n * 16) * reps / (ms * 1e-3) / 1e9 : 0.0;
    return 0;
}

int fdtd2d_clock_probe_start(fdtd2d_t *h, int micros)
{
    if (!h || micros < 1 || micros > 10000000) return FDTD2D_E_ARG;
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->clk_dev) {
        if (hipMalloc((void **)&h->clk_dev, CLK_PROBES * 3 * sizeof(unsigned long long)) != hipSuccess ||
            hipStreamCreateWithFlags(&h->clk_stream, hipStreamNonBlocking) != hipSuccess)
            return fail(h, FDTD2D_E_NOMEM, "clock probe buffers");
    }
    HIPCHK(h, hipMemsetAsync(h->clk_dev, 0, CLK_PROBES * 3 * sizeof(unsigned long long), h->clk_stream));
    hipLaunchKernelGGL(k_clock_probe, dim3(CLK_PROBES), dim3(64), 0, h->clk_stream, h->clk_dev,
                       (unsigned long long)micros * 100ull);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int fdtd2d_clock_probe_read(fdtd2d_t *h, double *mhz8)
{
    if (!h || !mhz8) return FDTD2D_E_ARG;
    if (!h->clk_dev) return fail(h, FDTD2D_E_STATE, "no clock probe was started");
    int rc = use_device(h);
    if (rc) return rc;
    unsigned long long v[CLK_PROBES * 3];
    HIPCHK(h, hipStreamSynchronize(h->clk_stream));
    HIPCHK(h, hipMemcpy(v, h->clk_dev, sizeof v, hipMemcpyDeviceToHost));
    double sum[8] = {}, cnt[8] = {};
    for (int k = 0; k < CLK_PROBES; ++k)
        if (v[3 * k + 1] > 0 && v[3 * k + 2] < 8) {
            sum[v[3 * k + 2]] += (double)v[3 * k] / (double)v[3 * k + 1] * 100.0;      // cycles per 10 ns tick -> MHz
            cnt[v[3 * k + 2]] += 1;
        }
    for (int x = 0; x < 8; ++x) mhz8[x] = cnt[x] > 0 ? sum[x] / cnt[x] : 0.0;
    return 0;
}

int fdtd2d_bytes_per_cell_step(const fdtd2d_t *h)
{
    if (!h) return FDTD2D_E_ARG;
    const int words = 6 + (h->ce_uniform ? 0 : 1) + (h->ch_uniform ? 0 : 1);
    return words * (int)h->esz;
}

void *fdtd2d_device_ptr(fdtd2d_t *h, int field)
{
    if (!h) return nullptr;
    switch (field) {
    case FDTD2D_FIELD_EZ: return h->ez[h->cur];
    case FDTD2D_FIELD_HX: return h->hx();
    case FDTD2D_FIELD_HY: return h->hy();
    default: return nullptr;
    }
}

}  // extern "C"

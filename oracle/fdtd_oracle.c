/*
 * CPU oracle (plain C) for the 2D TE-mode FDTD leapfrog -- TEST INFRASTRUCTURE ONLY.
 *
 * A scalar, per-cell restatement of the reference algorithm.  It is the checker
 * for the HIP path at sizes where the NumPy oracle is too slow, and the
 * multi-threaded "port" leg of bench.py's cpu_baseline.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product library (libfdtd2d.so) never links or calls it.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks every entry point
 * bit-for-bit against vectors produced by running the reference itself
 * (tests/golden/make_golden.py), for float64 and for float32 arrays.
 *
 * Reference map (relative to the reference checkout):
 *   orc_update_h_*   <- python-src/main.py:66-76   update_Hx_Hy
 *   orc_update_e_*   <- python-src/main.py:12-63   update_Ez (curl, Mur bands, corners)
 *   orc_mur_coef_*   <- python-src/main.py:30-31
 *   orc_add_point_*  <- python-src/fdtd.py:34 + main.py:185-186
 *   orc_ricker       <- python-src/main.py:183-184
 *   orc_run_*        <- python-src/fdtd.py:30-34   (H, E, source; t = i*dt)
 *
 * Layout = the reference's: row-major, Ez R x C, Hx R x (C-1), Hy (R-1) x C,
 * eps/mu R x C.  Arithmetic type = the array type: the float instantiation
 * rounds dt and dx to float first and then works in float throughout, which
 * is what NumPy does when the reference is handed float32 arrays (Python
 * scalars are weak).  Build with -ffp-contract=off: NumPy never fuses a*b+c.
 *
 * The boundary stages below run in the reference's own sequential order
 * (ascending band index, reading the not-yet-overwritten inward neighbour),
 * so unlike the staged NumPy oracle this file has no 11x11 minimum.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BAND 5

double orc_ricker(double t, double fc)
{
    const double pi = 3.141592653589793;
    double tau = pi * fc * (t - 1 / fc);
    return (1 - 2 * (tau * tau)) * exp(-(tau * tau));
}

#define DEFINE_ORACLE(T, SUF, SQRT)                                                         \
                                                                                            \
T orc_mur_coef_##SUF(T mu00, T eps00, double dt_, double dx_)                               \
{                                                                                           \
    T dt = (T)dt_, dx = (T)dx_;                                                             \
    T c = (T)1 / SQRT(mu00 * eps00);                                                        \
    T cdt = c * dt;                                                                         \
    return (cdt - dx) / (cdt + dx);                                                         \
}                                                                                           \
                                                                                            \
int orc_update_h_##SUF(const T *Ez, T *Hx, T *Hy, const T *mu, int R, int C,                \
                       double dt_, double dx_)                                              \
{                                                                                           \
    const T dt = (T)dt_, dx = (T)dx_;                                                       \
    _Pragma("omp parallel for schedule(static)")                                            \
    for (int i = 0; i < R - 1; ++i) {                                                       \
        const T *e0 = Ez + (size_t)i * C, *e1 = e0 + C;                                     \
        const T *m = mu + (size_t)i * C;                                                    \
        T *hx = Hx + (size_t)i * (C - 1);                                                   \
        T *hy = Hy + (size_t)i * C;                                                         \
        for (int j = 0; j < C - 1; ++j) {                                                   \
            T ch = dt / (m[j] * dx);                                                        \
            T dr = e1[j] - e0[j];                                                           \
            T dc = e0[j + 1] - e0[j];                                                       \
            hx[j] = hx[j] - ch * dr;                                                        \
            hy[j] = hy[j] + ch * dc;                                                        \
        }                                                                                   \
    }                                                                                       \
    return 0;                                                                               \
}                                                                                           \
                                                                                            \
/* P: caller scratch of R*C elements, or NULL (allocated here). */                          \
int orc_update_e_##SUF(T *Ez, const T *Hx, const T *Hy, const T *mu, const T *eps,          \
                       int R, int C, double dt_, double dx_, T *P)                          \
{                                                                                           \
    const T dt = (T)dt_, dx = (T)dx_;                                                       \
    int own = 0;                                                                            \
    if (R < BAND + 1 || C < BAND + 1) return -1;                                            \
    if (!P) { P = (T *)malloc((size_t)R * C * sizeof(T)); own = 1; if (!P) return -2; }     \
    memcpy(P, Ez, (size_t)R * C * sizeof(T));                                               \
    /* interior curl */                                                                     \
    _Pragma("omp parallel for schedule(static)")                                            \
    for (int i = 1; i < R - 1; ++i) {                                                       \
        T *e = Ez + (size_t)i * C;                                                          \
        const T *ep = eps + (size_t)i * C;                                                  \
        const T *hy = Hy + (size_t)i * C;                                                   \
        const T *hx = Hx + (size_t)i * (C - 1), *hxu = hx - (C - 1);                        \
        for (int j = 1; j < C - 1; ++j) {                                                   \
            T dhy = hy[j] - hy[j - 1];                                                      \
            T dhx = hx[j] - hxu[j];                                                         \
            T ce = dt / (ep[j] * dx);                                                       \
            e[j] = e[j] + (dhy - dhx) * ce;                                                 \
        }                                                                                   \
    }                                                                                       \
    const T k = orc_mur_coef_##SUF(mu[0], eps[0], dt_, dx_);                                \
    /* left, then right band: rows 1..R-2 */                                                \
    for (int b = 0; b < BAND; ++b)                                                          \
        for (int i = 1; i < R - 1; ++i) {                                                   \
            size_t o = (size_t)i * C;                                                       \
            Ez[o + b] = P[o + b + 1] + k * (Ez[o + b + 1] - P[o + b]);                      \
        }                                                                                   \
    for (int b = 0; b < BAND; ++b)                                                          \
        for (int i = 1; i < R - 1; ++i) {                                                   \
            size_t o = (size_t)i * C + (C - 1 - b);                                         \
            Ez[o] = P[o - 1] + k * (Ez[o - 1] - P[o]);                                      \
        }                                                                                   \
    /* top, then bottom band: columns 1..C-2 */                                             \
    for (int b = 0; b < BAND; ++b) {                                                        \
        size_t o = (size_t)b * C;                                                           \
        for (int j = 1; j < C - 1; ++j)                                                     \
            Ez[o + j] = P[o + C + j] + k * (Ez[o + C + j] - P[o + j]);                      \
    }                                                                                       \
    for (int b = 0; b < BAND; ++b) {                                                        \
        size_t o = (size_t)(R - 1 - b) * C;                                                 \
        for (int j = 1; j < C - 1; ++j)                                                     \
            Ez[o + j] = P[o - C + j] + k * (Ez[o - C + j] - P[o + j]);                      \
    }                                                                                       \
    /* corner blocks: mean of the inward row- and column-neighbour */                       \
    for (int a = 0; a < BAND; ++a)                                                          \
        for (int b = 0; b < BAND; ++b) {                                                    \
            size_t t = (size_t)a * C, u = (size_t)(R - 1 - a) * C;                          \
            int l = b, r = C - 1 - b;                                                       \
            Ez[t + l] = (Ez[t + l + 1] + Ez[t + C + l]) / 2;                                \
            Ez[t + r] = (Ez[t + r - 1] + Ez[t + C + r]) / 2;                                \
            Ez[u + l] = (Ez[u - C + l] + Ez[u + l + 1]) / 2;                                \
            Ez[u + r] = (Ez[u - C + r] + Ez[u + r - 1]) / 2;                                \
        }                                                                                   \
    if (own) free(P);                                                                       \
    return 0;                                                                               \
}                                                                                           \
                                                                                            \
int orc_add_point_##SUF(T *Ez, int R, int C, int row, int col, double amp)                  \
{                                                                                           \
    if (row < 0 || row >= R || col < 0 || col >= C) return -1;                              \
    size_t o = (size_t)row * C + col;                                                       \
    Ez[o] = (T)((double)Ez[o] + amp);                                                       \
    return 0;                                                                               \
}                                                                                           \
                                                                                            \
/* amps: nsteps doubles, or NULL for ricker(fc) at t = (step0 + n) * dt. */                 \
int orc_run_##SUF(T *Ez, T *Hx, T *Hy, const T *eps, const T *mu, int R, int C,             \
                  double dt, double dx, int nsteps, int src_row, int src_col,               \
                  const double *amps, double fc, long long step0)                           \
{                                                                                           \
    T *P = (T *)malloc((size_t)R * C * sizeof(T));                                          \
    if (!P) return -2;                                                                      \
    int rc = 0;                                                                             \
    for (int n = 0; n < nsteps && rc == 0; ++n) {                                           \
        rc = orc_update_h_##SUF(Ez, Hx, Hy, mu, R, C, dt, dx);                              \
        if (!rc) rc = orc_update_e_##SUF(Ez, Hx, Hy, mu, eps, R, C, dt, dx, P);             \
        double a = amps ? amps[n] : orc_ricker((double)(step0 + n) * dt, fc);               \
        if (!rc) rc = orc_add_point_##SUF(Ez, R, C, src_row, src_col, a);                   \
    }                                                                                       \
    free(P);                                                                                \
    return rc;                                                                              \
}

DEFINE_ORACLE(float, f32, sqrtf)
DEFINE_ORACLE(double, f64, sqrt)

void orc_set_threads(int n)
{
    extern void omp_set_num_threads(int);
    if (n > 0) omp_set_num_threads(n);
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/*
 * fdtd2d.h -- C ABI of libfdtd2d.so, the MI355X (gfx950) engine for the 2D TE-mode
 * Yee-grid Ez/Hx/Hy leapfrog.
 *
 * The reference (skunnavakkam/fdtd-2d) has no FFI for this path: its boundary is
 * the Python call surface of python-src/main.py as driven by python-src/fdtd.py.
 * Each entry point below names the reference interface it stands in for; the
 * ctypes binding a maintainer would add is shown in INTEGRATION.md and lives in
 * fdtd-2d_amd/_abi.py.
 *
 * Conventions
 *   - plain C types only; no torch/numpy types cross this boundary;
 *   - return 0 on success, a negative code on failure (FDTD2D_E_*; HIP errors are
 *     reported as -(1000 + hipError_t)); text via fdtd2d_last_error();
 *   - host buffers are borrowed for the duration of the call only;
 *   - the handle owns all device memory; a handle is not thread-safe, distinct
 *     handles are; no global state except the create-time error string;
 *   - all launches go to the handle's stream (fdtd2d_set_stream) and are
 *     asynchronous unless stated; fdtd2d_sync() waits;
 *   - there is no CPU fallback anywhere: without a usable gfx950 device every
 *     compute call fails with FDTD2D_E_NODEVICE.
 *
 * Index convention = the reference's: first axis row i, second axis column j;
 * host arrays are row-major: Ez R x C, Hx R x (C-1), Hy (R-1) x C, eps/mu R x C
 * (python-src/main.py:13-15,79-85).
 */
#ifndef FDTD2D_H
#define FDTD2D_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fdtd2d fdtd2d_t;

/* arithmetic / storage type of the device fields (also used for host buffers) */
#define FDTD2D_F32 0
#define FDTD2D_F64 1

/* treatment of the outer 5-cell frame of Ez */
#define FDTD2D_BOUNDARY_NONE 0 /* frame cells left at their stage-A value (edge cells never change) */
#define FDTD2D_BOUNDARY_MUR5 1 /* the reference: 5-px first-order Mur + corner rule, main.py:29-61 */
#define FDTD2D_BOUNDARY_PML  2 /* build-defined absorbing layer for BASELINE config 5 (no reference) */

/* point-source waveforms evaluated by the library (host side, float64) */
#define FDTD2D_SRC_NONE       0
#define FDTD2D_SRC_RICKER     1 /* main.py:182-187 */
#define FDTD2D_SRC_SINUSOIDAL 2 /* main.py:190-195 */

/* field selectors */
#define FDTD2D_FIELD_EZ 0
#define FDTD2D_FIELD_HX 1
#define FDTD2D_FIELD_HY 2

/* error codes */
#define FDTD2D_E_ARG      (-1)  /* bad argument / unsupported size (grids below 11x11) */
#define FDTD2D_E_NODEVICE (-2)  /* no gfx950 device / HIP runtime unusable */
#define FDTD2D_E_NOMEM    (-3)
#define FDTD2D_E_STATE    (-4)  /* call not valid in the current state (e.g. stale halo) */
#define FDTD2D_E_COURANT  (-5)  /* Courant number > 1 (fdtd.py:28): returned by fdtd2d_run*, fdtd2d_prepare
                                   and fdtd2d_pass_rows; nothing is launched */
/* HIP errors: -(1000 + hipError_t) */

/* info selectors for fdtd2d_info() */
#define FDTD2D_INFO_ROWS        0
#define FDTD2D_INFO_COLS        1
#define FDTD2D_INFO_ROW0        2
#define FDTD2D_INFO_NROWS       3
#define FDTD2D_INFO_HALO        4
#define FDTD2D_INFO_PITCH       5  /* elements per stored row */
#define FDTD2D_INFO_DTYPE       6
#define FDTD2D_INFO_BOUNDARY    7
#define FDTD2D_INFO_DEVICE      8
#define FDTD2D_INFO_EPS_UNIFORM 9
#define FDTD2D_INFO_MU_UNIFORM  10
#define FDTD2D_INFO_E_VALID_LO  11 /* global row range on which Ez is current */
#define FDTD2D_INFO_E_VALID_HI  12
#define FDTD2D_INFO_H_VALID_LO  13
#define FDTD2D_INFO_H_VALID_HI  14
#define FDTD2D_INFO_STEP        15 /* completed E half-steps since create/upload */
#define FDTD2D_INFO_PASS_LAUNCHES 16 /* temporally blocked pass kernels launched so far */
#define FDTD2D_INFO_STEP_LAUNCHES 17 /* single half-step kernels launched so far */
#define FDTD2D_INFO_LAST_BAND_ROWS 19 /* band height of the last temporally blocked pass */
#define FDTD2D_INFO_LAST_WAVES    20 /* its waves per (band, strip): 1 (k_bulk), 4 or 8 (k_bulk_split) */
#define FDTD2D_INFO_LAST_EDGE_ROWS 21 /* its band height on the first / last strip */
#define FDTD2D_INFO_LAST_PASS_STEPS 22 /* the kernel length (1, 2, 4, 8, 16, 20) of the last pass; 0: none yet */
#define FDTD2D_INFO_LAST_SIDE_WAVES 23 /* its waves side by side per level group (strip width 256, 504 or 1000 columns) */
#define FDTD2D_INFO_LAST_XCD_MAP    24 /* 1 if its tasks were dealt out XCD by XCD */
#define FDTD2D_INFO_CYCLE_STEPS   18 /* longest pass the current configuration runs: 16 (float32, Mur
                                         frame, >= 12 Mi cells per GPU), else 8, 0 if passes are off */

/* ---- lifetime ------------------------------------------------------------------ */

/* Whole-grid engine on one device.  Replaces grid_init (main.py:79-85): fields
 * start at zero.  rows, cols >= 11 (the staged Mur form). */
int fdtd2d_create(fdtd2d_t **out, int rows, int cols, double dt, double dx,
                  int dtype, int boundary, int device);

/* Row-slab engine for the multi-GPU decomposition: owns global rows
 * [row0, row0+nrows) of a rows x cols grid and keeps `halo` extra rows of every
 * field on each interior side.  One halo refresh allows up to `halo` full steps. */
int fdtd2d_create_slab(fdtd2d_t **out, int rows, int cols, int row0, int nrows, int halo,
                       double dt, double dx, int dtype, int boundary, int device);

void fdtd2d_destroy(fdtd2d_t *h);

/* Message of the last failure on this handle (or of the last failed create when h
 * is NULL).  Never NULL. */
const char *fdtd2d_last_error(const fdtd2d_t *h);

long long fdtd2d_info(const fdtd2d_t *h, int what);

/* Use an existing hipStream_t (e.g. torch's current stream) for all launches; NULL
 * restores the handle's own stream. */
int fdtd2d_set_stream(fdtd2d_t *h, void *hip_stream);

/* ---- materials ----------------------------------------------------------------- */

/* Replaces material_init (main.py:88-123) as the consumer of eps/mu.  eps and mu
 * point at the host rows this handle stores, i.e. global rows
 * [max(0,row0-halo), min(rows,row0+nrows+halo)), row-major with `cols` elements per
 * row, of type host_dtype.  The library forms ce = dt/(eps*dx) and ch = dt/(mu*dx)
 * once, in the engine's arithmetic type, exactly as main.py:27,70,74 do per step.
 * corner = {eps[0,0], mu[0,0]} of the GLOBAL grid (the Mur factor uses only that
 * cell, main.py:30-31); NULL is allowed when this handle stores global row 0.
 * allow_uniform != 0 lets the library detect constant arrays and switch to scalar
 * coefficients (same values, fewer bytes). */
int fdtd2d_set_materials(fdtd2d_t *h, const void *eps, const void *mu, int host_dtype,
                         const double *corner, int allow_uniform);

/* material_init(None, ...) path (main.py:103-106) and any other constant medium. */
int fdtd2d_set_materials_uniform(fdtd2d_t *h, double eps, double mu);

/* FDTD2D_BOUNDARY_PML only (BASELINE config 5; build-defined, the reference has no
 * time-domain PML): Berenger split-field layer.  row_factors = 4*rows values {ahr, bhr, aer,
 * ber}, col_factors = 4*cols values {ahc, bhc, aec, bec} (a = (1-s)/(1+s), b = 1/(1+s), 1
 * outside the layer; see oracle/pml_numpy.py / fdtd2d_amd.pml_profiles), in the engine's
 * dtype, for the GLOBAL grid (slabs index them by global row); layer_cells = L, the depth of
 * the layer: only cells within L of an edge use the split update, the rest the reference's
 * (main.py:21-27).  Ez is stored as total field plus its x-part Ezx (zero outside the layer); fdtd2d_transfer_ezx moves the owned rows of Ezx (rows x cols). */
int fdtd2d_set_pml(fdtd2d_t *h, const void *row_factors, const void *col_factors, int host_dtype,
                   int layer_cells);
int fdtd2d_transfer_ezx(fdtd2d_t *h, void *host, int host_dtype, int to_device);

/* Courant number c*dt/dx from the smallest eps and mu given so far (fdtd.py:25-26). */
double fdtd2d_courant(const fdtd2d_t *h);

/* ---- field transfer ------------------------------------------------------------ */

/* Host -> device for the OWNED rows.  Pointers address the first owned row of each
 * array in the reference's shapes (Ez: cols per row, Hx: cols-1, Hy: cols; Hy has
 * no row rows-1).  Any pointer may be NULL (field left as is).  Synchronous. */
int fdtd2d_upload(fdtd2d_t *h, const void *Ez, const void *Hx, const void *Hy, int host_dtype);

/* Device -> host for the OWNED rows, same layouts.  Synchronous (implies a sync). */
int fdtd2d_download(fdtd2d_t *h, void *Ez, void *Hx, void *Hy, int host_dtype);

/* Reset fields to zero and the step counter to 0 (grid_init again). */
int fdtd2d_reset(fdtd2d_t *h);

/* ---- the hot path -------------------------------------------------------------- */

/* update_Hx_Hy(Ez,Hx,Hy,mu,eps,dt,dx), main.py:66-76. */
int fdtd2d_update_h(fdtd2d_t *h);

/* update_Ez(Ez,Hx,Hy,mu,eps,dt,dx), main.py:12-63 (interior + Mur bands + corners). */
int fdtd2d_update_e(fdtd2d_t *h);

/* `Ez += source` with one non-zero cell, fdtd.py:34: Ez[row,col] =
 * round(float64(Ez[row,col]) + amp).  A cell outside this handle's rows is ignored. */
int fdtd2d_add_point(fdtd2d_t *h, int row, int col, double amp);

/* Line / patch sources (the reference's FDTD only has the one-cell source):
 * from now on the source of fdtd2d_add_point / fdtd2d_run* / fdtd2d_pass_rows is the rectangle of
 * nrows x ncols cells whose first cell is the (row, col) given there; every cell gets the same
 * amplitude, rounded per cell like the one-cell source.  Default 1 x 1. */
int fdtd2d_set_source_extent(fdtd2d_t *h, int nrows, int ncols);

/* Point probe: from now on fdtd2d_run* / fdtd2d_pass_rows record
 * Ez[row, col] after the source of every step -- also the steps inside a temporally blocked pass,
 * which never reach memory otherwise -- as float64 into a device buffer of `capacity` samples
 * (sample 0 = the first step after this call; capacity 0 removes the probe).  On a slab only
 * handles whose current rows hold the cell record it.  fdtd2d_read_probe waits for the stream
 * and copies samples [first, first + count) to the host.  With the PML boundary a probed run
 * uses the single-step kernels. */
int fdtd2d_set_probe(fdtd2d_t *h, int row, int col, long long capacity);
int fdtd2d_read_probe(fdtd2d_t *h, double *out, long long first, long long count);

/* The loop of fdtd.py:30-34 for nsteps steps: H, E, source.  amps = nsteps float64
 * amplitudes (host) or NULL for no source.  Asynchronous.  For a slab with
 * neighbours, nsteps must not exceed the halo validity left. */
int fdtd2d_run(fdtd2d_t *h, int nsteps, int src_row, int src_col, const double *amps);

/* Optional: measure the launch shapes fdtd2d_run(nsteps) will use (FDTD2D_OPT_AUTOTUNE) now
 * instead of inside its first passes.  Blocks for the 20-120 ms of trial launches; the fields
 * and the step counter are untouched. */
int fdtd2d_prepare(fdtd2d_t *h, int nsteps);
/* The same for a run with its source: the trial launches carry the source rectangle at (src_row,
 * src_col) with amplitude 0 when with_source != 0 (the strips that hold the source run a slower
 * body in bands of their own, which changes the best shape on grids that fill the GPU exactly once). */
int fdtd2d_prepare_run(fdtd2d_t *h, int nsteps, int src_row, int src_col, int with_source);

/* One temporally blocked pass of nt in {1,2,4,8,16} steps issued in pieces, so that a caller can
 * compute the rows its neighbours wait for first, send them, and overlap the transfer with
 * the rest: fdtd2d_pass_rows() launches the pass for output rows [row_lo,row_hi) only (reading
 * the current fields, writing the other buffer set; pieces may go to different streams via
 * fdtd2d_set_stream); fdtd2d_pass_commit() makes the new set current once the pieces tile
 * everything a full pass writes.  Between the two, fdtd2d_halo_pack() packs from the NEW
 * set.  amps = nt amplitudes (the same for every piece) or NULL. */
int fdtd2d_pass_rows(fdtd2d_t *h, int nt, int row_lo, int row_hi, int src_row, int src_col,
                     const double *amps);
int fdtd2d_pass_commit(fdtd2d_t *h);

/* Same with the waveform evaluated by the library at t = (step0+n)*dt. */
int fdtd2d_run_waveform(fdtd2d_t *h, int nsteps, int src_kind, int src_row, int src_col,
                        double fc, long long step0);

/* Waveform scalar (float64), main.py:183-184 / 193-194. */
double fdtd2d_source_amplitude(int src_kind, double t, double fc);

int fdtd2d_sync(fdtd2d_t *h);

/* Tuning knobs of fdtd2d_run (results do not depend on them, only speed):
 *   FDTD2D_OPT_MAX_PASS_STEPS  longest temporally blocked pass, 0..16 (0 = plain single steps
 *                              with the half-step kernels); default 16.  16-step passes exist
 *                              for float32 with the Mur frame (uniform or array materials);
 *                              everything else runs 8-step passes.  By default they are used
 *                              from 12 Mi cells per GPU up (faster from 4096^2, slower below:
 *                              profiles/r01_nt16_sweep.txt); setting this option to 16
 *                              explicitly uses them at every size.
 *   FDTD2D_OPT_BAND_ROWS       rows per streaming band (0 = heuristic) */
#define FDTD2D_OPT_MAX_PASS_STEPS 0
#define FDTD2D_OPT_BAND_ROWS      1
#define FDTD2D_OPT_LEVEL_SPLIT    3   /* 8-step passes with the level-split kernel (4 waves per band/strip,
                                         rows handed over through LDS): -1 / 1 yes (default), 0 the
                                         single-wave kernel.  16-step passes always use it. */
#define FDTD2D_OPT_ZONE_SPLIT     2   /* top/bottom zone tiles: 0 fused into the pass launch, 1 their own
                                         kernel on a side stream, -1 (default) fused for the level-split
                                         kernel and by launch size for the single-wave kernel */
#define FDTD2D_OPT_SPLIT_WAVES    4   /* waves per band/strip in the level-split kernel: 0 automatic
                                         (default), 4 or 8 */
#define FDTD2D_OPT_AUTOTUNE       5   /* 1 (default): the first pass of >= 8 steps over >= 4 Mi cells
                                         times a ladder of band heights (and 4 / 8 waves per strip)
                                         with uncommitted trial launches and keeps the fastest;
                                         0: fixed rules.  Results are identical either way. */
/* (6: reserved -- launch shapes are handed over with fdtd2d_set_shape) */
#define FDTD2D_OPT_XCD_MAP         7   /* task order of the level-split pass: 1 = every XCD (workgroups b, b + 8, ...
                                         under the round-robin placement the hardware is observed to use) gets a
                                         contiguous run of (band, strip) tasks with the strips of one band next to
                                         each other, so that the cache lines neighbouring strips share are fetched
                                         once per XCD L2; 0 = all bands of one strip consecutive; -1 (default) = the
                                         launch-shape tuner decides per shape.  Speed only. */
#define FDTD2D_OPT_SIDE_WAVES      8   /* float32 16- / 20-step passes: 2 or 4 waves side by side per level group share a
                                         strip of 504 / 1000 columns and exchange their boundary columns through the
                                         LDS hand-off -- the 32 overlap columns of a strip are paid once per 504 / 1000
                                         columns instead of per 256; 1 = never; 0 (default) = the tuner decides. */
int fdtd2d_set_option(fdtd2d_t *h, int option, long long value);

/* Launch shape of the temporally blocked passes of `pass_steps` steps (0 = the full-length passes, 16 or 8): re-use
 * a shape the tuner found elsewhere (another process, an earlier run).  shape[0..n): band rows, waves per level group
 * (0 = automatic, 4, 8), band rows of the first / last strip (0 = the same), waves side by side per level group (1, 2,
 * 4), xcd map (0 / 1), and for launches that fit the GPU in one round: rows and number per strip of the shorter "filler"
 * bands that take over the slots the zone tiles free (0, 0 = none), and for float32 20-step passes whether the zone tiles
 * ride in the bulk launch (1) or run as their own kernel on a side stream (0); missing trailing entries are 0 (side: 1);
 * shape[0] = 0 clears it.  fdtd2d_last_shape returns the shape the last pass ran with, in the same order.  Results never
 * depend on the shape. */
#define FDTD2D_SHAPE_LEN 8
int fdtd2d_set_shape(fdtd2d_t *h, int pass_steps, const int *shape, int n);
int fdtd2d_last_shape(const fdtd2d_t *h, int *shape, int n);

/* ---- row-slab halo exchange (transport is the caller's: RCCL via torch.distributed) -- */

/* Bytes of one halo message: 3 fields (4 with the PML's Ezx) x halo rows x cols elements. */
long long fdtd2d_halo_bytes(const fdtd2d_t *h);

/* Pack the `halo` owned rows nearest to side (0 = top/lower row index, 1 = bottom)
 * of Ez, Hx, Hy into dev_buf (device memory, fdtd2d_halo_bytes()).  Asynchronous. */
int fdtd2d_halo_pack(fdtd2d_t *h, int side, void *dev_buf);

/* Unpack a neighbour's message into this handle's halo rows on `side` and mark them
 * current.  Both sides that have a neighbour must be unpacked to restore validity. */
int fdtd2d_halo_unpack(fdtd2d_t *h, int side, const void *dev_buf);

/* ---- the row-slab run loop in C ("distributed communication backend") --------
 * fdtd2d_run_slab() enqueues a rank's WHOLE run -- per cycle of `cycle` steps: the rows next to the
 * cuts first (on an internal edge stream), pack, neighbour exchange, meanwhile the interior on the
 * handle's stream, commit, unpack -- so that no host-language code runs per exchange cycle.  It
 * replaces the per-cycle calls above for callers that attach a transport:
 *   fdtd2d_slab_attach_rccl   built-in: RCCL point-to-point (ncclSend / ncclRecv, one group per cycle on
 *                             the edge stream) with ranks rank-1 / rank+1 of a communicator created
 *                             from `unique_id128` (fdtd2d_rccl_unique_id() on rank 0, distributed by the
 *                             caller, e.g. torch.distributed.broadcast); the library loads librccl.so
 *                             with dlopen and owns the message buffers;
 *   fdtd2d_slab_attach        the caller's transport function and device message buffers
 *                             (fdtd2d_halo_bytes() each; NULL for a side without neighbour).
 * fn(ctx, send_top, recv_top, send_bottom, recv_bottom, bytes, stream) must ENQUEUE on `stream` (a
 * hipStream_t) -- or complete before returning -- the transfer of this rank's packed send buffers to
 * the neighbours' recv buffers (NULL pointers: no neighbour on that side); return 0 on success.
 * Every rank must pass the same nsteps, cycle (<= halo; what all ranks' fdtd2d_info(CYCLE_STEPS)
 * agree on) and overlap flag: ranks that disagree post their transfers in different orders.
 * overlap != 0 needs EVERY slab to be at least 2 * halo + 5 rows tall (two edge pieces of `halo` rows and an
 * interior that holds the whole top / bottom zone on the first / last rank); a rank whose slab is shorter
 * returns FDTD2D_E_ARG before it posts anything.  With the PML and a probe set no temporally blocked pass
 * exists (fdtd2d_info(CYCLE_STEPS) says 8, passes fail): run such handles with overlap = 0 on every rank.
 * The loop state belongs to the handle (no global table): distinct handles may run on distinct threads. */
typedef int (*fdtd2d_exchange_fn)(void *ctx, void *send_top, void *recv_top, void *send_bottom,
                                  void *recv_bottom, long long bytes, void *stream);
int fdtd2d_slab_attach(fdtd2d_t *h, void *send_top, void *recv_top, void *send_bottom, void *recv_bottom,
                       fdtd2d_exchange_fn fn, void *ctx);
int fdtd2d_rccl_unique_id(void *out128);
/* Self-test of the built-in transport's glue on ONE GPU: a one-rank communicator and one grouped
 * ncclSend / ncclRecv of `count` floats to itself through the same entry points.  0 = the message arrived. */
int fdtd2d_rccl_selftest(int device, long long count);
int fdtd2d_slab_attach_rccl(fdtd2d_t *h, const void *unique_id128, int rank, int world);
int fdtd2d_slab_detach(fdtd2d_t *h);
/* Ranks of the communicator behind the attached transport as RCCL itself counts them (ncclCommCount) and this
 * handle's rank in it (ncclCommUserRank): rank * 65536 + count; 0 for a caller-supplied transport; < 0 on error. */
long long fdtd2d_slab_ranks(fdtd2d_t *h);
int fdtd2d_run_slab(fdtd2d_t *h, int nsteps, int cycle, int overlap, int src_row, int src_col,
                    const double *amps);

/* ---- consumers of Ez next to the loop ---------------------- */

/* Running Fourier transform of Ez at given angular frequencies over a window of cells (the data
 * that connects the time-domain loop to the reference's frequency-domain fields, fdfd.py:111-118).  From this call on, after
 * every `every`-th completed step n (counted by fdtd2d_info(STEP)) the engine adds Ez[i,j] * exp(-i * omega_k * n * dt) to
 * accumulator k of every cell of the window [row0, row0+nrows) x [col0, col0+ncols) (float64 accumulation on the device; a
 * slab accumulates the window rows it owns).  While a transform is set the loop cuts its temporally blocked passes at the
 * sampled steps (every = 16 costs nothing, every = 1 runs single steps); fdtd2d_run_slab needs `every` to be a multiple of
 * its cycle.  nfreq <= 16.  nfreq = 0 removes it.  fdtd2d_read_dft copies the accumulators: re and im, each nfreq x (owned
 * window rows) x ncols float64, row-major; synchronous. */
int fdtd2d_set_dft(fdtd2d_t *h, int row0, int col0, int nrows, int ncols, int nfreq, const double *omega, int every);
int fdtd2d_read_dft(fdtd2d_t *h, double *re, double *im);

/* Device-side first half of capture_snapshot (main.py:153-179): clip Ez to [vmin,vmax], map
 * to the 0..255 colour-map index exactly as `cmap((normed - vmin)/(vmax - vmin))` does for an
 * array of the engine's type, keep every `stride`-th row and column (global indices that are
 * multiples of stride), and copy the bytes to `out` (row-major, ceil(cols/stride) per row,
 * owned rows only).  The host applies the colour table (fdtd2d_amd.capture_snapshot).
 * Synchronous. */
int fdtd2d_snapshot_index(fdtd2d_t *h, double vmin, double vmax, int stride, unsigned char *out);

/* Sum of squares and max |.| of one field over the owned rows (float64 accumulation).
 * Either output may be NULL.  Synchronous. */
int fdtd2d_reduce(fdtd2d_t *h, int field, double *sum_sq, double *max_abs);

/* ---- measurement --------------------------------------------------------------- */

/* HIP events on the handle's stream. stop returns elapsed milliseconds (syncs). */
int fdtd2d_timer_start(fdtd2d_t *h);
int fdtd2d_timer_stop(fdtd2d_t *h, float *ms);

/* Run `nlaunch` back-to-back passes of `steps_each` steps (no source) with one HIP event
 * before and one after EACH of them on the handle's stream, wait, and store the elapsed
 * milliseconds of every launch in ms[0..nlaunch-1].  For roofline measurements: this is the
 * per-kernel duration rocprofv3's kernel trace reports.  nlaunch <= 256. */
int fdtd2d_time_launches(fdtd2d_t *h, int nlaunch, int steps_each, float *ms);

/* Algorithmic HBM bytes per cell-step of the current configuration (SURVEY.md
 * section 8 M2): 24 + 4 per non-uniform coefficient array, times sizeof(T)/4. */
/* Shader clock under load: _start enqueues (on an internal side stream, NOT ordered with the handle's stream) 16
 * single-wave workgroups that sleep for `micros` microseconds and stamp the shader-cycle counter (s_memtime) against
 * the constant 100 MHz counter (s_memrealtime) at both ends; launch the work to be observed right after it.
 * _read waits for them and returns the clock in MHz per XCC id (0 where no probe landed).  Measurement aid. */
int fdtd2d_clock_probe_start(fdtd2d_t *h, int micros);
/* What a plain copy reaches on this device right now: copies the three current field arrays into the other buffer set
 * (which the next pass overwrites anyway) `reps` times with a 16-byte-per-lane grid-stride kernel and returns
 * (bytes read + bytes written) / time in GB/s.  The yardstick beside the 8 TB/s of the data sheet.  Synchronous. */
int fdtd2d_measure_copy(fdtd2d_t *h, int reps, double *gbps);
int fdtd2d_clock_probe_read(fdtd2d_t *h, double *mhz8);

int fdtd2d_bytes_per_cell_step(const fdtd2d_t *h);

/* Device pointer of a field's storage (row-major, pitch elements per row, first
 * stored row = global row row0-halo) for zero-copy interop.  NULL on error. */
void *fdtd2d_device_ptr(fdtd2d_t *h, int field);

/* Names the build: libfdtd2d.so = "... one rounding per operation: value-identical build" (-ffp-contract=off),
 * libfdtd2d_fused.so = "... fused multiply-add: tolerance build" (the same sources with a*b+c contracted into FMA:
 * 8 instead of 11 operations per cell-step, results within rounding of the exact build -- SURVEY.md M3's fp32
 * tolerance -- and still independent of the launch shape). */
const char *fdtd2d_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FDTD2D_H */

// Row-slab run loop in C: fdtd2d_slab_attach* / fdtd2d_run_slab (include/fdtd2d.h).
//
// One rank's whole run -- edge rows, pack, neighbour exchange, interior, commit, unpack, cycle after
// cycle -- is enqueued by ONE call; no host-language code runs per exchange cycle.  The transport is a
// function the caller attaches:
//   * fdtd2d_slab_attach_rccl(): the built-in one, RCCL point-to-point with the two neighbours
//     (ncclSend / ncclRecv in one group on the edge stream; librccl.so is loaded with dlopen so that the
//     library itself has no link-time dependency on it);
//   * fdtd2d_slab_attach(): any function of the caller's (tests stage the messages through the host
//     and gloo on a single GPU with it, so the sequencing below is exercised without several GPUs).
#include "engine.hpp"

#include <dlfcn.h>

#include <cstring>
#include <vector>

using namespace fdtd_host;

struct fdtd2d_slab {
    bool has[2] = {false, false};            // neighbour above (side 0) / below (side 1)
    void *send[2] = {nullptr, nullptr}, *recv[2] = {nullptr, nullptr};
    bool own_bufs = false;
    fdtd2d_exchange_fn fn = nullptr;
    void *ctx = nullptr;
    hipStream_t edge = nullptr;              // rows next to the cuts, pack, transfer
    hipEvent_t ev_main = nullptr, ev_edge = nullptr;
    // built-in RCCL transport
    void *lib = nullptr, *comm = nullptr;
    int rank = 0, world = 1;
    long long count = 0;                     // elements per message
    int nccl_dtype = 7;                      // ncclFloat32
    int (*p_send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*p_recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*p_gstart)() = nullptr, (*p_gend)() = nullptr;
    int (*p_destroy)(void *) = nullptr;
    const char *(*p_errstr)(int) = nullptr;
};

namespace {

// the loop state lives in the handle (no global: distinct handles may be driven by distinct threads)
fdtd2d_slab *slab_of(fdtd2d *h) { return h->slab; }

// are the halo rows current (an exchange has happened since the last pass / upload)?
bool halo_fresh(const fdtd2d *h)
{
    const int lo = h->store_lo(), hi = h->store_hi();
    return h->ev.lo <= lo && h->ev.hi >= hi && h->hv.lo <= lo && h->hv.hi >= hi;
}

int rccl_exchange(void *ctx, void *send_top, void *recv_top, void *send_bot, void *recv_bot, long long bytes,
                  void *stream)
{
    (void)bytes;
    fdtd2d_slab *s = (fdtd2d_slab *)ctx;
    hipStream_t st = (hipStream_t)stream;
    int rc = s->p_gstart();
    if (!rc && send_top) rc = s->p_send(send_top, (size_t)s->count, s->nccl_dtype, s->rank - 1, s->comm, st);
    if (!rc && recv_top) rc = s->p_recv(recv_top, (size_t)s->count, s->nccl_dtype, s->rank - 1, s->comm, st);
    if (!rc && send_bot) rc = s->p_send(send_bot, (size_t)s->count, s->nccl_dtype, s->rank + 1, s->comm, st);
    if (!rc && recv_bot) rc = s->p_recv(recv_bot, (size_t)s->count, s->nccl_dtype, s->rank + 1, s->comm, st);
    const int rc2 = s->p_gend();
    return rc ? rc : rc2;
}

// The process may already hold an RCCL (torch.distributed's "nccl" backend loads the copy that ships with
// torch): take THAT one -- two RCCL builds in one process would each export the same symbols -- and load the
// system's library only when none is there yet.
void *load_rccl(std::string *err)
{
    for (const char *name : {"librccl.so.1", "librccl.so"})
        if (void *l = dlopen(name, RTLD_NOW | RTLD_NOLOAD)) return l;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"})
        if (void *l = dlopen(name, RTLD_NOW | RTLD_LOCAL)) return l;
    const char *e = dlerror();
    *err = e ? e : "librccl.so not found";
    return nullptr;
}

int attach_common(fdtd2d *h, fdtd2d_slab *s)
{
    if (hipStreamCreateWithFlags(&s->edge, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_main, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_edge, hipEventDisableTiming) != hipSuccess)
        return fail(h, FDTD2D_E_NODEVICE, "stream / event creation for the slab loop failed");
    h->slab = s;
    return 0;
}

// owned edge rows -> neighbours' halos, everything on the handle's stream
int exchange_plain(fdtd2d *h, fdtd2d_slab *s)
{
    int rc;
    for (int side = 0; side < 2; ++side)
        if (s->has[side] && (rc = fdtd2d_halo_pack(h, side, s->send[side]))) return rc;
    rc = s->fn(s->ctx, s->has[0] ? s->send[0] : nullptr, s->has[0] ? s->recv[0] : nullptr,
               s->has[1] ? s->send[1] : nullptr, s->has[1] ? s->recv[1] : nullptr, fdtd2d_halo_bytes(h), h->stream);
    if (rc) return fail(h, FDTD2D_E_STATE, "halo transport failed with code %d", rc);
    for (int side = 0; side < 2; ++side)
        if (s->has[side] && (rc = fdtd2d_halo_unpack(h, side, s->recv[side]))) return rc;
    return 0;
}

// one pass of n steps with the exchange for the NEXT pass hidden behind the interior
int cycle_overlapped(fdtd2d *h, fdtd2d_slab *s, int n, int src_row, int src_col, const double *amps)
{
    const int r0 = h->row0, r1 = h->row0 + h->nrows, hl = h->halo;
    hipStream_t main_stream = h->stream;
    HIPCHK(h, hipEventRecord(s->ev_main, main_stream));
    HIPCHK(h, hipStreamWaitEvent(s->edge, s->ev_main, 0));
    h->stream = s->edge;
    int rc = 0;
    for (int side = 0; side < 2 && !rc; ++side) {
        if (!s->has[side]) continue;
        const int lo = side == 0 ? r0 : r1 - hl, hi = side == 0 ? r0 + hl : r1;
        rc = fdtd2d_pass_rows(h, n, lo, hi, src_row, src_col, amps);
        if (!rc) rc = fdtd2d_halo_pack(h, side, s->send[side]);
    }
    if (!rc) {
        const int t = s->fn(s->ctx, s->has[0] ? s->send[0] : nullptr, s->has[0] ? s->recv[0] : nullptr,
                            s->has[1] ? s->send[1] : nullptr, s->has[1] ? s->recv[1] : nullptr, fdtd2d_halo_bytes(h), s->edge);
        if (t) rc = fail(h, FDTD2D_E_STATE, "halo transport failed with code %d", t);
    }
    h->stream = main_stream;
    if (rc) return rc;
    const int lo = s->has[0] ? r0 + hl : 0, hi = s->has[1] ? r1 - hl : h->rows;
    if ((rc = fdtd2d_pass_rows(h, n, lo, hi, src_row, src_col, amps))) return rc;
    HIPCHK(h, hipEventRecord(s->ev_edge, s->edge));
    HIPCHK(h, hipStreamWaitEvent(main_stream, s->ev_edge, 0));
    if ((rc = fdtd2d_pass_commit(h))) return rc;
    for (int side = 0; side < 2; ++side)
        if (s->has[side] && (rc = fdtd2d_halo_unpack(h, side, s->recv[side]))) return rc;
    return 0;
}

}  // namespace

extern "C" {

int fdtd2d_slab_attach(fdtd2d_t *h, void *send_top, void *recv_top, void *send_bottom, void *recv_bottom,
                       fdtd2d_exchange_fn fn, void *ctx)
{
    if (!h || !fn) return FDTD2D_E_ARG;
    if (h->halo == 0) return fail(h, FDTD2D_E_STATE, "this handle is not a slab with neighbours");
    if (slab_of(h)) return fail(h, FDTD2D_E_STATE, "a slab loop is already attached");
    const bool top = !h->top(), bot = !h->bottom();
    if ((top && (!send_top || !recv_top)) || (bot && (!send_bottom || !recv_bottom)))
        return fail(h, FDTD2D_E_ARG, "message buffers are needed for every side with a neighbour");
    fdtd2d_slab *s = new fdtd2d_slab();
    s->has[0] = top;
    s->has[1] = bot;
    s->send[0] = send_top; s->recv[0] = recv_top; s->send[1] = send_bottom; s->recv[1] = recv_bottom;
    s->fn = fn;
    s->ctx = ctx;
    int rc = attach_common(h, s);
    if (rc) delete s;
    return rc;
}

int fdtd2d_rccl_unique_id(void *out128)
{
    if (!out128) return FDTD2D_E_ARG;
    std::string err;
    void *lib = load_rccl(&err);
    if (!lib) return fail(nullptr, FDTD2D_E_NODEVICE, "cannot load librccl.so: %s", err.c_str());
    auto get = (int (*)(void *))dlsym(lib, "ncclGetUniqueId");
    if (!get) return fail(nullptr, FDTD2D_E_NODEVICE, "ncclGetUniqueId not found in librccl.so");
    const int rc = get(out128);                 // ncclUniqueId is 128 bytes
    return rc ? fail(nullptr, FDTD2D_E_STATE, "ncclGetUniqueId failed with code %d", rc) : 0;
}

// One-rank communicator, one grouped ncclSend / ncclRecv to itself on a non-blocking stream: the same
// entry points, argument types and enum values as rccl_exchange() above, runnable on a single GPU.
int fdtd2d_rccl_selftest(int device, long long count)
{
    if (count < 1) return FDTD2D_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, FDTD2D_E_NODEVICE, "hipSetDevice(%d) failed", device);
    std::string err;
    void *lib = load_rccl(&err);
    if (!lib) return fail(nullptr, FDTD2D_E_NODEVICE, "cannot load librccl.so: %s", err.c_str());
    struct Id { char b[128]; } id;
    fdtd2d_slab s;
    auto get = (int (*)(void *))dlsym(lib, "ncclGetUniqueId");
    auto init = (int (*)(void **, int, Id, int))dlsym(lib, "ncclCommInitRank");
    s.p_send = (decltype(s.p_send))dlsym(lib, "ncclSend");
    s.p_recv = (decltype(s.p_recv))dlsym(lib, "ncclRecv");
    s.p_gstart = (decltype(s.p_gstart))dlsym(lib, "ncclGroupStart");
    s.p_gend = (decltype(s.p_gend))dlsym(lib, "ncclGroupEnd");
    s.p_destroy = (decltype(s.p_destroy))dlsym(lib, "ncclCommDestroy");
    if (!get || !init || !s.p_send || !s.p_recv || !s.p_gstart || !s.p_gend || !s.p_destroy)
        return fail(nullptr, FDTD2D_E_NODEVICE, "librccl.so lacks the point-to-point entry points");
    int nrc = get(&id);
    if (!nrc) nrc = init(&s.comm, 1, id, 0);
    if (nrc) return fail(nullptr, FDTD2D_E_STATE, "RCCL communicator set-up failed with code %d", nrc);
    int rc = 0;
    float *a = nullptr, *b = nullptr;
    hipStream_t st = nullptr;
    std::vector<float> host((size_t)count);
    for (long long n = 0; n < count; ++n) host[(size_t)n] = (float)(n % 8191) * 0.5f;
    if (hipMalloc(&a, (size_t)count * 4) != hipSuccess || hipMalloc(&b, (size_t)count * 4) != hipSuccess ||
        hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess ||
        hipMemcpy(a, host.data(), (size_t)count * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(b, 0, (size_t)count * 4) != hipSuccess)
        rc = fail(nullptr, FDTD2D_E_NOMEM, "self-test buffers");
    if (!rc) {
        nrc = s.p_gstart();
        if (!nrc) nrc = s.p_send(a, (size_t)count, 7, 0, s.comm, st);
        if (!nrc) nrc = s.p_recv(b, (size_t)count, 7, 0, s.comm, st);
        const int nrc2 = s.p_gend();
        if (nrc || nrc2) rc = fail(nullptr, FDTD2D_E_STATE, "grouped ncclSend / ncclRecv failed with code %d", nrc ? nrc : nrc2);
    }
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = fail(nullptr, FDTD2D_E_STATE, "stream sync after the transfer failed");
    if (!rc) {
        std::vector<float> back((size_t)count);
        if (hipMemcpy(back.data(), b, (size_t)count * 4, hipMemcpyDeviceToHost) != hipSuccess || back != host)
            rc = fail(nullptr, FDTD2D_E_STATE, "the received message differs from the one sent");
    }
    if (st) (void)hipStreamDestroy(st);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    s.p_destroy(s.comm);
    return rc;
}

int fdtd2d_slab_attach_rccl(fdtd2d_t *h, const void *unique_id128, int rank, int world)
{
    if (!h || !unique_id128) return FDTD2D_E_ARG;
    if (h->halo == 0) return fail(h, FDTD2D_E_STATE, "this handle is not a slab with neighbours");
    if (slab_of(h)) return fail(h, FDTD2D_E_STATE, "a slab loop is already attached");
    if (world < 2 || rank < 0 || rank >= world || (rank == 0) != h->top() || (rank == world - 1) != h->bottom())
        return fail(h, FDTD2D_E_ARG, "rank %d of %d does not match this slab's position", rank, world);
    int rc = 0;
    if (hipSetDevice(h->device) != hipSuccess) return fail(h, FDTD2D_E_NODEVICE, "hipSetDevice failed");
    fdtd2d_slab *s = new fdtd2d_slab();
    std::string err;
    s->lib = load_rccl(&err);
    if (!s->lib) { delete s; return fail(h, FDTD2D_E_NODEVICE, "cannot load librccl.so: %s", err.c_str()); }
    struct Id { char b[128]; } id;
    std::memcpy(&id, unique_id128, 128);
    auto init = (int (*)(void **, int, Id, int))dlsym(s->lib, "ncclCommInitRank");
    s->p_send = (decltype(s->p_send))dlsym(s->lib, "ncclSend");
    s->p_recv = (decltype(s->p_recv))dlsym(s->lib, "ncclRecv");
    s->p_gstart = (decltype(s->p_gstart))dlsym(s->lib, "ncclGroupStart");
    s->p_gend = (decltype(s->p_gend))dlsym(s->lib, "ncclGroupEnd");
    s->p_destroy = (decltype(s->p_destroy))dlsym(s->lib, "ncclCommDestroy");
    s->p_errstr = (decltype(s->p_errstr))dlsym(s->lib, "ncclGetErrorString");
    if (!init || !s->p_send || !s->p_recv || !s->p_gstart || !s->p_gend) {
        delete s;
        return fail(h, FDTD2D_E_NODEVICE, "librccl.so lacks the point-to-point entry points");
    }
    const int nrc = init(&s->comm, world, id, rank);
    if (nrc) {
        const char *m = s->p_errstr ? s->p_errstr(nrc) : "?";
        delete s;
        return fail(h, FDTD2D_E_STATE, "ncclCommInitRank failed: %s (%d)", m, nrc);
    }
    s->rank = rank;
    s->world = world;
    s->has[0] = rank > 0;
    s->has[1] = rank < world - 1;
    s->count = fdtd2d_halo_bytes(h) / (long long)h->esz;
    s->nccl_dtype = h->dtype == FDTD2D_F32 ? 7 : 8;          // ncclFloat32 / ncclFloat64
    s->own_bufs = true;
    for (int side = 0; side < 2 && !rc; ++side) {
        if (!s->has[side]) continue;
        if (hipMalloc(&s->send[side], (size_t)fdtd2d_halo_bytes(h)) != hipSuccess ||
            hipMalloc(&s->recv[side], (size_t)fdtd2d_halo_bytes(h)) != hipSuccess)
            rc = fail(h, FDTD2D_E_NOMEM, "hipMalloc of the halo messages failed");
    }
    s->fn = rccl_exchange;
    s->ctx = s;
    if (!rc) rc = attach_common(h, s);
    if (rc) {
        for (void *p : {s->send[0], s->recv[0], s->send[1], s->recv[1]})
            if (p) (void)hipFree(p);
        if (s->comm && s->p_destroy) s->p_destroy(s->comm);
        delete s;
    }
    return rc;
}

int fdtd2d_slab_detach(fdtd2d_t *h)
{
    if (!h) return FDTD2D_E_ARG;
    fdtd2d_slab *s = slab_of(h);
    if (!s) return 0;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (s->edge) { (void)hipStreamSynchronize(s->edge); (void)hipStreamDestroy(s->edge); }
    if (s->ev_main) (void)hipEventDestroy(s->ev_main);
    if (s->ev_edge) (void)hipEventDestroy(s->ev_edge);
    if (s->own_bufs)
        for (void *p : {s->send[0], s->recv[0], s->send[1], s->recv[1]})
            if (p) (void)hipFree(p);
    if (s->comm && s->p_destroy) s->p_destroy(s->comm);
    h->slab = nullptr;
    delete s;
    return 0;
}

long long fdtd2d_slab_ranks(fdtd2d_t *h)
{
    if (!h) return FDTD2D_E_ARG;
    fdtd2d_slab *s = slab_of(h);
    if (!s) return fail(h, FDTD2D_E_STATE, "no slab loop attached");
    if (!s->comm) return 0;
    auto count = (int (*)(void *, int *))dlsym(s->lib, "ncclCommCount");
    auto urank = (int (*)(void *, int *))dlsym(s->lib, "ncclCommUserRank");
    int n = 0, r = 0;
    if (!count || !urank || count(s->comm, &n) || urank(s->comm, &r))
        return fail(h, FDTD2D_E_STATE, "ncclCommCount / ncclCommUserRank failed");
    return (long long)r * 65536 + n;
}

int fdtd2d_run_slab(fdtd2d_t *h, int nsteps, int cycle, int overlap, int src_row, int src_col, const double *amps)
{
    if (!h) return FDTD2D_E_ARG;
    fdtd2d_slab *s = slab_of(h);
    if (!s) return fail(h, FDTD2D_E_STATE, "no slab loop attached: call fdtd2d_slab_attach[_rccl] first");
    if (nsteps < 0 || cycle < 1 || cycle > h->halo)
        return fail(h, FDTD2D_E_ARG, "need nsteps >= 0 and 1 <= cycle <= halo (%d)", h->halo);
    if (h->dft_n && h->dft_every % cycle != 0)
        return fail(h, FDTD2D_E_ARG, "a running Fourier transform samples every %d steps: not a multiple of the %d-step cycle", h->dft_every, cycle);
    if (hipSetDevice(h->device) != hipSuccess) return fail(h, FDTD2D_E_NODEVICE, "hipSetDevice failed");
    // Overlapped cycles need a temporally blocked pass of `cycle` steps.  Only quantities every rank
    // shares enter this decision (whether EVERY slab is tall enough for two edge pieces and an
    // interior is the caller's to establish before it passes overlap != 0): ranks that decided
    // differently would post their transfers in different orders.
    const bool can_overlap = overlap && h->rows >= 2 * (2 * cycle + 6) && (cycle == 8 || cycle == 16);
    if (can_overlap) {
        // two edge pieces of `halo` rows and an interior; on the first / last rank the interior piece must hold
        // the whole top / bottom zone (5 + cycle rows with the Mur frame), which is written as a whole or not at
        // all.  Checked before anything is posted: the caller passes overlap != 0 only when EVERY slab is at least
        // 2 * halo + 5 rows tall (include/fdtd2d.h).
        const int zo = h->boundary == FDTD2D_BOUNDARY_MUR5 ? 5 + cycle : 0;
        const int need = std::max(2 * h->halo + 1, (h->top() || h->bottom()) ? h->halo + zo : 0);
        if (h->nrows < need)
            return fail(h, FDTD2D_E_ARG, "overlapped cycles need a slab of at least %d rows on this rank (it has %d): "
                        "pass overlap = 0 on every rank", need, h->nrows);
    }
    int done = 0, rc = 0;
    while (done < nsteps && !rc) {
        const int n = std::min(cycle, nsteps - done);
        const double *a = amps ? amps + done : nullptr;
        if (!halo_fresh(h) && (rc = exchange_plain(h, s))) break;
        if (can_overlap && n == cycle) rc = cycle_overlapped(h, s, n, src_row, src_col, a);
        else rc = fdtd2d_run(h, n, src_row, src_col, a);
        done += n;
    }
    return rc;
}

}  // extern "C"

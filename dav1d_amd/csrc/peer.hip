// Multi-GPU behind the C ABI (gfx950 + RCCL over xGMI): the data-path collectives of SURVEY 8e, callable from the C host of
// INTEGRATION.md — one process per GPU, one Dav1dHipPeer per process.
//
//   * frames decoded elsewhere (config C4): dav1d_hip_peer_broadcast_picture — the owner's picture, one ncclBroadcast of the whole
//     allocation (planes are one contiguous allocation: dav1d_hip_picture_alloc), the frame-granular counterpart of the row
//     publication of src/thread_task.c:416-433;
//   * tile columns (config C3): dav1d_hip_peer_allgather_columns — every rank packs the strip it reconstructed (all planes) into
//     one contiguous message with a strided copy kernel, ONE ncclAllGather (strips padded to the widest column: direct over xGMI,
//     7 x 13 MB per rank at 8K), a second kernel scatters the other ranks' strips straight into the planes;
//   * in-loop filters across the tile edge: dav1d_hip_peer_exchange_halo — the `halo` luma columns on either side of a rank's column
//     from its neighbours, the same pack / gather / scatter on 16-column strips.
// Everything is enqueued on the context's stream (RCCL calls take the stream): no host synchronisation before a collective, the
// pack kernel follows the frame's launches in stream order.  RCCL is resolved at run time (dlopen("librccl.so")): a host without
// it gets -ENOSYS from dav1d_hip_peer_open and keeps everything else.  The SIMT-emulated build (CPU tests) takes a shared-memory
// stand-in for the same entry points (tests/emu/emu_rccl.h).
#include "common.h"
#include "capi.h"
#include <string.h>
#include <type_traits>
#ifdef DAV1D_HIP_EMU
#include "emu_rccl.h"
#else
#include <dlfcn.h>
#include <rccl/rccl.h>
#endif

struct Dav1dHipPeer {
    Dav1dHipContext *c;
    int rank, world;
    ncclComm_t comm;
    uint8_t *send, *recv;          // staging for the strided exchanges on the CONTEXT's stream (gather, halo), grown on demand
    size_t send_cap, recv_cap;
    // staging of the asynchronous gathers (side stream): buffers of their own, because nothing orders the side stream against the halo
    // exchange / synchronous gather of the next frame on the context's stream, which would otherwise pack and receive into the bytes an
    // in-flight gather is still reading.  Gathers on the side stream follow each other in order, so one pair is enough.
    uint8_t *asend, *arecv;
    size_t asend_cap, arecv_cap;
#ifndef DAV1D_HIP_EMU
    void *dl;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
#endif
    // the collectives can run on a stream of their own (dav1d_hip_peer_allgather_columns_async): the gather of frame n's columns
    // next to the reconstruction of frame n + 1, which the caller has already enqueued on the context's stream
    hipStream_t side;
    hipEvent_t ev_in, ev_out[8];      // ev_out[k & 7]: the k-th asynchronous gather is through
    unsigned n_async;
};

namespace {

// strips[k] = columns [x0, x1) of plane `pl` (pixels), rows [0, rows): strided <-> contiguous, 16 bytes per lane where the strip allows
struct StripPlan { int x0[3], w[3], rows[3], stride[3]; size_t off[3]; size_t bytes; };

// up to MAX_STRIPS strips per launch: blockIdx.z = strip * 3 + plane (the scatter of a gather is ONE launch over the strips of all
// the other ranks, each read from its own slot of the receive buffer)
enum { MAX_STRIPS = 8 };
struct StripSet { StripPlan p[MAX_STRIPS]; size_t slot[MAX_STRIPS]; int n; };

template <bool PACK>
__global__ __launch_bounds__(256) void strip_copy_kernel(uint8_t *plane0, uint8_t *plane1, uint8_t *plane2, uint8_t *buf0, const StripSet set, const int bps)
{
    const int si = blockIdx.z / 3, pl = blockIdx.z - 3 * si;
    if (si >= set.n) return;
    // (kernel arguments are not indexed by run-time values: a chain of selects over the few strips)
    StripPlan p = set.p[0];
    size_t slot = set.slot[0];
#pragma unroll
    for (int k = 1; k < MAX_STRIPS; k++) if (si == k) { p = set.p[k]; slot = set.slot[k]; }
    uint8_t *const buf = buf0 + slot;
    uint8_t *const plane = pl == 0 ? plane0 : pl == 1 ? plane1 : plane2;
    const int w = pl == 0 ? p.w[0] : pl == 1 ? p.w[1] : p.w[2], rows = pl == 0 ? p.rows[0] : pl == 1 ? p.rows[1] : p.rows[2];
    if (!plane || w <= 0) return;
    const int x0 = pl == 0 ? p.x0[0] : pl == 1 ? p.x0[1] : p.x0[2], stride = pl == 0 ? p.stride[0] : pl == 1 ? p.stride[1] : p.stride[2];
    const size_t off = pl == 0 ? p.off[0] : pl == 1 ? p.off[1] : p.off[2];
    const int row_bytes = w * bps;
    const int y = blockIdx.y;
    if (y >= rows) return;
    uint8_t *const a = plane + (size_t) y * stride * bps + (size_t) x0 * bps, *const b = buf + off + (size_t) y * row_bytes;
    // 16-byte pieces when both sides are aligned for them (tile columns start at multiples of 128 pixels: they are), bytes otherwise
    const bool wide = !(((uintptr_t) a | (uintptr_t) b | (uintptr_t) row_bytes) & 15);
    if (wide) {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < row_bytes / 16; i += gridDim.x * 256) {
            if (PACK) reinterpret_cast<uint4 *>(b)[i] = reinterpret_cast<const uint4 *>(a)[i];
            else reinterpret_cast<uint4 *>(a)[i] = reinterpret_cast<const uint4 *>(b)[i];
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < row_bytes; i += gridDim.x * 256) {
            if (PACK) b[i] = a[i]; else a[i] = b[i];
        }
    }
}

int plan_strip(const Dav1dHipPicture *pic, int x0, int x1, StripPlan *p, size_t fixed_w /* luma pixels of a padded strip, 0 = exact */) {
    const int bps = pic->bpc > 8 ? 2 : 1;
    const int ss_hor = pic->layout == DAV1D_HIP_LAYOUT_I420 || pic->layout == DAV1D_HIP_LAYOUT_I422;
    size_t off = 0;
    memset(p, 0, sizeof(*p));
    for (int pl = 0; pl < 3; pl++) {
        if (!pic->p[pl].data) continue;
        const int s = pl ? ss_hor : 0;
        // rows of the allocation that belong to the picture: the visible ones rounded up to the 8-row block grid
        p->rows[pl] = (pic->p[pl].h + 7) & ~7;
        p->x0[pl] = x0 >> s;
        p->w[pl] = (x1 - x0 + s) >> s;
        p->stride[pl] = (int) (pic->p[pl].stride / bps);
        p->off[pl] = off;
        const size_t wpad = fixed_w ? (fixed_w + s) >> s : (size_t) p->w[pl];
        off += (((size_t) p->rows[pl] * wpad * bps) + 255) & ~(size_t) 255;
    }
    p->bytes = off;
    return 0;
}

int grow(uint8_t **p, size_t *cap, size_t bytes) {
    if (*cap >= bytes) return 0;
    if (*p) (void) hipFree(*p);
    *p = nullptr; *cap = 0;
    size_t want = 1 << 20;
    while (want < bytes) want <<= 1;
    if (hipMalloc((void **) p, want) != hipSuccess) return -ENOMEM;
    *cap = want;
    return 0;
}

void launch_strips(bool pack, const Dav1dHipPicture *pic, uint8_t *buf, const StripSet &set, hipStream_t st) {
    const int bps = pic->bpc > 8 ? 2 : 1;
    int max_rows = 0;
    for (int k = 0; k < set.n; k++)
        for (int pl = 0; pl < 3; pl++) max_rows = set.p[k].rows[pl] > max_rows ? set.p[k].rows[pl] : max_rows;
    if (!max_rows || !set.n) return;
    const dim3 grid(2, (unsigned) max_rows, 3 * (unsigned) set.n), block(256);
    if (pack) hipLaunchKernelGGL((strip_copy_kernel<true>), grid, block, 0, st, (uint8_t *) pic->p[0].data, (uint8_t *) pic->p[1].data, (uint8_t *) pic->p[2].data, buf, set, bps);
    else hipLaunchKernelGGL((strip_copy_kernel<false>), grid, block, 0, st, (uint8_t *) pic->p[0].data, (uint8_t *) pic->p[1].data, (uint8_t *) pic->p[2].data, buf, set, bps);
}
void launch_strip(bool pack, const Dav1dHipPicture *pic, uint8_t *buf, const StripPlan &p, hipStream_t st) {
    StripSet set;
    memset(&set, 0, sizeof(set));
    set.p[0] = p; set.slot[0] = 0; set.n = 1;
    launch_strips(pack, pic, buf, set, st);
}

// columns of the ranks: even, inside the plane, one after the other (tile columns are), each at least `min_w` wide; the rows a strip
// copies must lie inside the allocation (a caller-wrapped picture: planes of at least round8(h) rows)
int check_columns(const Dav1dHipPicture *pic, const int *x0, const int *x1, const int world, const int min_w) {
    if (!pic->p[0].data || pic->p[0].w <= 0) return -EINVAL;
    const int wmax = (pic->p[0].w + 7) & ~7;
    for (int g = 0; g < world; g++) {
        if (x0[g] < 0 || x1[g] <= x0[g] || x1[g] > wmax || (x0[g] & 1) || (x1[g] & 1) || x1[g] - x0[g] < min_w) return -EINVAL;
        if (g && x0[g] < x1[g - 1]) return -EINVAL;
    }
    if (pic->alloc && pic->alloc_size) {
        const int bps = pic->bpc > 8 ? 2 : 1;
        for (int pl = 0; pl < 3; pl++) {
            if (!pic->p[pl].data) continue;
            const size_t end = (size_t) ((const uint8_t *) pic->p[pl].data - (const uint8_t *) pic->alloc) + (size_t) ((pic->p[pl].h + 7) & ~7) * (size_t) pic->p[pl].stride;
            (void) bps;
            if (end > pic->alloc_size + (size_t) pic->p[pl].stride) return -EINVAL;
        }
    }
    return 0;
}

} // namespace

#ifdef DAV1D_HIP_EMU
#define NCCL(p, fn) nccl##fn
#define STREAM_ARG(st) nullptr
#else
#define NCCL(p, fn) (p)->fn
#define STREAM_ARG(st) (st)
#endif

extern "C" {

int dav1d_hip_peer_unique_id(uint8_t id[128]) {
    if (!id) return -EINVAL;
    ncclUniqueId u;
#ifdef DAV1D_HIP_EMU
    if (ncclGetUniqueId(&u) != ncclSuccess) return -EIO;
#else
    void *dl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!dl) dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!dl) return -ENOSYS;
    ncclResult_t (*get)(ncclUniqueId *) = (ncclResult_t (*)(ncclUniqueId *)) dlsym(dl, "ncclGetUniqueId");
    const bool ok = get && get(&u) == ncclSuccess;
    dlclose(dl);                       // (dav1d_hip_peer_open takes its own reference)
    if (!ok) return -EIO;
#endif
    static_assert(sizeof(u) == 128, "ncclUniqueId");
    memcpy(id, &u, 128);
    return 0;
}

static void peer_free(Dav1dHipPeer *p) {
    if (p->ev_in) (void) hipEventDestroy(p->ev_in);
    for (int k = 0; k < 8; k++) if (p->ev_out[k]) (void) hipEventDestroy(p->ev_out[k]);
    if (p->side) (void) hipStreamDestroy(p->side);
#ifndef DAV1D_HIP_EMU
    if (p->dl) dlclose(p->dl);
#endif
    delete p;
}

int dav1d_hip_peer_open(Dav1dHipContext *c, Dav1dHipPeer **out, const uint8_t id[128], int rank, int world) {
    if (!c || !out || !id || world < 1 || rank < 0 || rank >= world) return -EINVAL;
    *out = nullptr;
    Dav1dHipPeer *p = new (std::nothrow) Dav1dHipPeer();
    if (!p) return -ENOMEM;
    p->c = c; p->rank = rank; p->world = world; p->comm = nullptr;
    p->send = p->recv = nullptr; p->send_cap = p->recv_cap = 0;
    p->asend = p->arecv = nullptr; p->asend_cap = p->arecv_cap = 0;
    p->side = nullptr; p->ev_in = nullptr; p->n_async = 0;
    for (int k = 0; k < 8; k++) p->ev_out[k] = nullptr;
    ncclUniqueId u;
    memcpy(&u, id, 128);
#ifndef DAV1D_HIP_EMU
    p->dl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!p->dl) p->dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!p->dl) { peer_free(p); return -ENOSYS; }
    *(void **) &p->CommInitRank = dlsym(p->dl, "ncclCommInitRank");
    *(void **) &p->CommDestroy = dlsym(p->dl, "ncclCommDestroy");
    *(void **) &p->AllGather = dlsym(p->dl, "ncclAllGather");
    *(void **) &p->Broadcast = dlsym(p->dl, "ncclBroadcast");
    *(void **) &p->Send = dlsym(p->dl, "ncclSend");
    *(void **) &p->Recv = dlsym(p->dl, "ncclRecv");
    *(void **) &p->GroupStart = dlsym(p->dl, "ncclGroupStart");
    *(void **) &p->GroupEnd = dlsym(p->dl, "ncclGroupEnd");
    if (!p->CommInitRank || !p->CommDestroy || !p->AllGather || !p->Broadcast || !p->Send || !p->Recv || !p->GroupStart || !p->GroupEnd) { peer_free(p); return -ENOSYS; }
    if (hipSetDevice(c->device) != hipSuccess) { peer_free(p); return -ENODEV; }
#endif
    bool ev_ok = hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&p->ev_in, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < 8 && ev_ok; k++) ev_ok = hipEventCreateWithFlags(&p->ev_out[k], hipEventDisableTiming) == hipSuccess;
    if (!ev_ok) { peer_free(p); return -ENOMEM; }
    if (NCCL(p, CommInitRank)(&p->comm, world, u, rank) != ncclSuccess) { peer_free(p); return -EIO; }
    *out = p;
    return 0;
}

void dav1d_hip_peer_close(Dav1dHipPeer *p) {
    if (!p) return;
    (void) hipStreamSynchronize(p->c->stream);
    if (p->side) (void) hipStreamSynchronize(p->side);
    if (p->comm) (void) NCCL(p, CommDestroy)(p->comm);
    if (p->send) (void) hipFree(p->send);
    if (p->recv) (void) hipFree(p->recv);
    if (p->asend) (void) hipFree(p->asend);
    if (p->arecv) (void) hipFree(p->arecv);
    peer_free(p);
}

int dav1d_hip_peer_rank(const Dav1dHipPeer *p) { return p ? p->rank : -EINVAL; }
int dav1d_hip_peer_world(const Dav1dHipPeer *p) { return p ? p->world : -EINVAL; }

// every rank ends up with `owner`'s planes of `pic` (same geometry on every rank).  Pictures of dav1d_hip_picture_alloc are one
// allocation: one broadcast; caller-wrapped ones go plane by plane.
int dav1d_hip_peer_broadcast_picture(Dav1dHipPeer *p, Dav1dHipPicture *pic, int owner) {
    if (!p || !pic || owner < 0 || owner >= p->world) return -EINVAL;
    pic->twin_ok = pic->twin_ok && p->rank == owner;            // the raster planes of the other ranks change under their twins
    if (p->world == 1) return 0;
    hipStream_t st = p->c->stream;
    if (pic->alloc && pic->alloc_size) {
        if (NCCL(p, Broadcast)(pic->alloc, pic->alloc, pic->alloc_size, ncclUint8, owner, p->comm, STREAM_ARG(st)) != ncclSuccess) return -EIO;
        return 0;
    }
    for (int pl = 0; pl < 3; pl++) {
        if (!pic->p[pl].data) continue;
        const size_t bytes = (size_t) pic->p[pl].stride * (size_t) ((pic->p[pl].h + 7) & ~7);
        if (NCCL(p, Broadcast)(pic->p[pl].data, pic->p[pl].data, bytes, ncclUint8, owner, p->comm, STREAM_ARG(st)) != ncclSuccess) return -EIO;
    }
    return 0;
}

// Rank g reconstructed luma columns [x0[g], x1[g]) of `pic` (tile column g; chroma follows the layout): afterwards every rank holds
// every column.  One message per rank (all planes of its strip, padded to the widest column), ONE all-gather, ONE scatter launch over
// the strips of all the other ranks.  `st`: the stream it all runs on.
static int allgather_on(Dav1dHipPeer *p, Dav1dHipPicture *pic, const int *x0, const int *x1, hipStream_t st, const bool async) {
    uint8_t **const sendp = async ? &p->asend : &p->send, **const recvp = async ? &p->arecv : &p->recv;
    size_t *const send_cap = async ? &p->asend_cap : &p->send_cap, *const recv_cap = async ? &p->arecv_cap : &p->recv_cap;
    int wmax = 0;
    for (int g = 0; g < p->world; g++) wmax = x1[g] - x0[g] > wmax ? x1[g] - x0[g] : wmax;
    StripPlan mine, slot;
    plan_strip(pic, x0[p->rank], x1[p->rank], &mine, (size_t) wmax);
    plan_strip(pic, 0, wmax, &slot, (size_t) wmax);
    const size_t per = slot.bytes;
    // (growing frees the old buffer with hipFree, which waits for the device: an earlier gather still reading it is through by then)
    int rc = grow(sendp, send_cap, per);
    if (!rc) rc = grow(recvp, recv_cap, per * (size_t) p->world);
    if (rc) return rc;
    uint8_t *const send = *sendp, *const recv = *recvp;
    // (the strips are raster rows: a picture that lives in its tiled twin only gets its raster planes back first)
    if (pic->twin_ok == DAV1D_HIP_TWIN_ONLY) if (const int ru = dav1d_hip_picture_untile(p->c, pic)) return ru;
    pic->twin_ok = 0;
    // the strip's rows are packed with the strip's own width as row pitch inside a slot laid out for the widest strip
    StripPlan pk = mine;
    for (int pl = 0; pl < 3; pl++) pk.off[pl] = slot.off[pl];
    launch_strip(true, pic, send, pk, st);
    if (NCCL(p, AllGather)(send, recv, per, ncclUint8, p->comm, STREAM_ARG(st)) != ncclSuccess) return -EIO;
    StripSet set;
    memset(&set, 0, sizeof(set));
    for (int g = 0; g < p->world; g++) {
        if (g == p->rank) continue;
        StripPlan up;
        plan_strip(pic, x0[g], x1[g], &up, (size_t) wmax);
        for (int pl = 0; pl < 3; pl++) up.off[pl] = slot.off[pl];
        set.p[set.n] = up; set.slot[set.n] = (size_t) g * per;
        if (++set.n == MAX_STRIPS) { launch_strips(false, pic, recv, set, st); set.n = 0; }       // (more than 9 ranks: a launch per 8 strips)
    }
    if (set.n) launch_strips(false, pic, recv, set, st);
    return hip_rc(hipGetLastError());
}

int dav1d_hip_peer_allgather_columns(Dav1dHipPeer *p, Dav1dHipPicture *pic, const int *x0, const int *x1) {
    if (!p || !pic || !x0 || !x1) return -EINVAL;
    const int rc = check_columns(pic, x0, x1, p->world, 2);
    if (rc) return rc;
    if (p->world == 1) return 0;
    return allgather_on(p, pic, x0, x1, p->c->stream, false);
}

// The same on the peer's side stream: it starts when what the context's stream holds so far is through (the frame that wrote the
// columns) and runs NEXT TO what the caller enqueues afterwards — the reconstruction of the next frame, which reads other pictures.
// dav1d_hip_peer_wait makes the context's stream wait for it (before the first launch that reads `pic` as a reference).
int dav1d_hip_peer_allgather_columns_async(Dav1dHipPeer *p, Dav1dHipPicture *pic, const int *x0, const int *x1) {
    if (!p || !pic || !x0 || !x1) return -EINVAL;
    int rc = check_columns(pic, x0, x1, p->world, 2);
    if (rc) return rc;
    if (p->world == 1) return 0;
    if (hipEventRecord(p->ev_in, p->c->stream) != hipSuccess || hipStreamWaitEvent(p->side, p->ev_in, 0) != hipSuccess) return -EIO;
    rc = allgather_on(p, pic, x0, x1, p->side, true);
    if (!rc && hipEventRecord(p->ev_out[p->n_async & 7], p->side) != hipSuccess) rc = -EIO;
    if (!rc) p->n_async++;
    return rc;
}
// lag = 0: every asynchronous gather issued so far; lag = k (< 8): all but the k most recent (a caller that recycles pictures a few
// frames later bounds how far the context's stream may run ahead of the gathers without giving the overlap up)
int dav1d_hip_peer_wait(Dav1dHipPeer *p, int lag) {
    if (!p || lag < 0 || lag > 7) return -EINVAL;
    if (p->world == 1 || p->n_async <= (unsigned) lag) return 0;
    return hip_rc(hipStreamWaitEvent(p->c->stream, p->ev_out[(p->n_async - 1 - (unsigned) lag) & 7], 0));
}

// The `halo` luma columns beyond either side of this rank's column [x0[rank], x1[rank]) from the neighbours that reconstructed them
// (the in-loop filters read across the tile edge: SURVEY 8e; 16 columns cover deblocking + CDEF + restoration).  Neighbour to
// neighbour: a send / receive pair per side inside one group — the data is needed next door only (an all-gather moved every edge to
// every rank).
int dav1d_hip_peer_exchange_halo(Dav1dHipPeer *p, Dav1dHipPicture *pic, const int *x0, const int *x1, int halo) {
    if (!p || !pic || !x0 || !x1 || halo <= 0 || (halo & 1)) return -EINVAL;
    int rc = check_columns(pic, x0, x1, p->world, halo);
    if (rc) return rc;
    if (p->world == 1) return 0;
    const int r = p->rank;
    StripPlan edge;
    plan_strip(pic, 0, halo, &edge, (size_t) halo);
    const size_t eb = edge.bytes;
    rc = grow(&p->send, &p->send_cap, 2 * eb);
    if (!rc) rc = grow(&p->recv, &p->recv_cap, 2 * eb);
    if (rc) return rc;
    hipStream_t st = p->c->stream;
    if (pic->twin_ok == DAV1D_HIP_TWIN_ONLY) if (const int ru = dav1d_hip_picture_untile(p->c, pic)) return ru;
    pic->twin_ok = 0;
    const bool has_l = r > 0, has_r = r + 1 < p->world;
    // my own left edge [x0, x0 + halo) -> send[0], right edge [x1 - halo, x1) -> send[1]: one pack launch
    StripSet set;
    memset(&set, 0, sizeof(set));
    if (has_l) { plan_strip(pic, x0[r], x0[r] + halo, &set.p[set.n], (size_t) halo); set.slot[set.n++] = 0; }
    if (has_r) { plan_strip(pic, x1[r] - halo, x1[r], &set.p[set.n], (size_t) halo); set.slot[set.n++] = eb; }
    launch_strips(true, pic, p->send, set, st);
#ifdef DAV1D_HIP_EMU
    (void) ncclGroupStart();
    if (has_l) { (void) ncclSend(p->send, eb, ncclUint8, r - 1, p->comm, nullptr); (void) ncclRecv(p->recv, eb, ncclUint8, r - 1, p->comm, nullptr); }
    if (has_r) { (void) ncclSend(p->send + eb, eb, ncclUint8, r + 1, p->comm, nullptr); (void) ncclRecv(p->recv + eb, eb, ncclUint8, r + 1, p->comm, nullptr); }
    if (emu_nccl_group_end(p->comm) != ncclSuccess) return -EIO;
#else
    bool ok = p->GroupStart() == ncclSuccess;
    if (has_l) ok = ok && p->Send(p->send, eb, ncclUint8, r - 1, p->comm, st) == ncclSuccess && p->Recv(p->recv, eb, ncclUint8, r - 1, p->comm, st) == ncclSuccess;
    if (has_r) ok = ok && p->Send(p->send + eb, eb, ncclUint8, r + 1, p->comm, st) == ncclSuccess && p->Recv(p->recv + eb, eb, ncclUint8, r + 1, p->comm, st) == ncclSuccess;
    ok = (p->GroupEnd() == ncclSuccess) && ok;
    if (!ok) return -EIO;
#endif
    // the left neighbour's right edge -> [x0 - halo, x0), the right neighbour's left edge -> [x1, x1 + halo): one scatter launch
    memset(&set, 0, sizeof(set));
    if (has_l) { plan_strip(pic, x0[r] - halo, x0[r], &set.p[set.n], (size_t) halo); set.slot[set.n++] = 0; }
    if (has_r) { plan_strip(pic, x1[r], x1[r] + halo, &set.p[set.n], (size_t) halo); set.slot[set.n++] = eb; }
    launch_strips(false, pic, p->recv, set, st);
    return hip_rc(hipGetLastError());
}

} // extern "C"

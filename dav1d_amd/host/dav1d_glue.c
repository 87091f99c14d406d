/* The reference-side binding of the HIP backend (see dav1d_glue.h).  Compiled into dav1d next to patches/dav1d-1.5.4-hip.patch.
 *
 * A frame's way through the glue:
 *   dav1d_hip_glue_frame_init      (worker thread, Dav1dHooks.after_init)      dav1d_hip_frame_begin + dav1d_hip_lister_create, the frame's
 *                                                                              filter stages (f->bd_fn.filter_sbrow_*) pointed at the filter lister
 *   dav1d_hip_glue_recon_tile_sbrow(worker threads, instead of pass 2)          dav1d_hip_lister_tile_sbrow: task lists from Av1Block / cbi / cf
 *   filter tasks                   (worker threads)                             dav1d_hip_lister_filter_sbrow, once per superblock row
 *   dav1d_hip_glue_frame_complete  (worker thread, scheduler lock held)         the frame queues up for the three stage threads:
 *     stage 1 (any order)   uploads: coefficients (unless packed), level cache, palette indices — a context and stream of its own
 *     stage 2 (oldest first among the frames whose references have ENDED: where a reference's final pixels are is known then)
 *                           dav1d_hip_frame_set_refs / _set_filters / dav1d_hip_frame_end; rows go to progress[1] as they become final
 *     stage 3               the picture to the host planes the application sees, then dav1d_hip_frame_done: dav1d_decode_frame_exit
 * Several devices (option n_devices = N): a picture is allocated on device (allocation number mod N) — dav1d allocates one per frame, in
 * frame order, so with n_fc a multiple of N frame context k always lands on device k mod N — and its frame ends THERE: every device has its
 * three contexts and its three stage threads.  A reference that lives on another device is made resident first (a mirror picture on the
 * reading device, dav1d_hip_picture_copy_peer: xGMI between peers), once per picture and reading device.  Frames that do not predict from
 * each other (key frames, the layers of a pyramid) end side by side on different devices; a chain of inter frames ends in turn, as on one.
 * The hooks carry no user pointer: one glue per process at a time (g_glue). */
#include "config.h"
#include <dlfcn.h>
#include <errno.h>
#include <limits.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "dav1d/dav1d.h"
#include "common/frame.h"
#include "src/internal.h"
#include "src/picture.h"
#include "src/decode.h"
#include "src/thread_task.h"
#include "dav1d_glue.h"

typedef struct Hip {
    void *dl;
    int (*open)(Dav1dHipContext **, int, void *);
    void (*close)(Dav1dHipContext *);
    int (*sync)(Dav1dHipContext *);
    int (*malloc_)(Dav1dHipContext *, void **, size_t);
    int (*free_)(Dav1dHipContext *, void *);
    int (*upload)(Dav1dHipContext *, void *, const void *, size_t);
    int (*memset_)(Dav1dHipContext *, void *, int, size_t);
    int (*host_picture_alloc)(Dav1dHipContext *, Dav1dHipHostPicture *, int, int, int, int);
    int (*host_picture_release)(Dav1dHipContext *, Dav1dHipHostPicture *);
    int (*host_picture_fetch)(Dav1dHipContext *, const Dav1dHipHostPicture *, const Dav1dHipPicture *, int, int);
    int (*host_picture_wait)(Dav1dHipContext *);
    int (*frame_begin)(Dav1dHipContext *, Dav1dHipFrame **, const Dav1dHipPicture *, const Dav1dHipPicture *, int);
    int (*frame_set_refs)(Dav1dHipFrame *, const Dav1dHipPicture *, int);
    int (*frame_set_filters)(Dav1dHipFrame *, const uint8_t *, ptrdiff_t, const uint8_t *, const uint8_t *, int, const Dav1dHipFilmGrainData *, int);
    int (*frame_end)(Dav1dHipFrame *, void *, int16_t *, uint8_t *, Dav1dHipPicture *, const Dav1dHipPicture *);
    void (*frame_destroy)(Dav1dHipFrame *);
    int (*lister_create)(Dav1dHipLister **, const Dav1dHipFrameDesc *, Dav1dHipFrame *);
    int (*lister_tile_sbrow)(Dav1dHipLister *, int, int, int);
    int (*lister_filter_sbrow)(Dav1dHipLister *, const Dav1dHipFilterDesc *, int);
    size_t (*lister_prep_elems)(const Dav1dHipLister *);
    size_t (*lister_mask_bytes)(const Dav1dHipLister *);
    const uint8_t *(*lister_const_masks)(size_t *);
    void (*lister_destroy)(Dav1dHipLister *);
    int (*frame_submit_intra_step)(Dav1dHipFrame *, size_t, const Dav1dHipIpredTask *, size_t, const Dav1dHipItxTask *, size_t, uint8_t *);
    int (*frame_set_super_res)(Dav1dHipFrame *, int);
    int (*fg_apply)(Dav1dHipContext *, const Dav1dHipPicture *, const Dav1dHipPicture *, const Dav1dHipFilmGrainData *, int);
    int (*picture_alloc)(Dav1dHipContext *, Dav1dHipPicture *, int, int, int, int);
    int (*picture_free)(Dav1dHipContext *, Dav1dHipPicture *);
    int (*plane_download)(Dav1dHipContext *, const Dav1dHipPicture *, int, void *, ptrdiff_t, int);
    int (*frame_set_progress_callback)(Dav1dHipFrame *, void (*)(void *, int, const Dav1dHipPicture *), void *);
    int (*live_objects)(long long *);
    int (*device_count)(void);
    int (*use)(Dav1dHipContext *);
    int (*enable_peer_access)(Dav1dHipContext *, Dav1dHipContext *);
    int (*current_device)(void);
    int (*set_device)(int);
    int (*picture_copy_peer)(Dav1dHipContext *, Dav1dHipPicture *, Dav1dHipContext *, const Dav1dHipPicture *);
    int (*picture_copy_peer_rows)(Dav1dHipContext *, Dav1dHipPicture *, Dav1dHipContext *, const Dav1dHipPicture *, int, int);
    int (*picture_retile)(Dav1dHipContext *, Dav1dHipPicture *);
} Hip;

/* per frame context */
typedef struct FcState {
    Dav1dHipFrameDesc desc;
    Dav1dHipFilterDesc fd;
    Dav1dHipFrame *frame;
    Dav1dHipLister *lister;
    atomic_int *filter_listed;   /* [sby]: the filter tasks of the row are listed */
    int sbh_cap;
    int dev;                     /* the device the frame ends on: its picture's */
    struct FcBufs {              /* device buffers of the frame context, per device it has had frames on */
        void *coef, *lvl, *prep, *mask, *pal_idx;
        size_t coef_cap, lvl_cap, prep_cap, mask_cap, pal_idx_cap;
    } b[DAV1D_HIP_GLUE_MAX_DEVICES];
    /* the frame's way through the three stage threads: 0 idle, 1 tasks through, 2 uploaded (or failed), 3 ended */
    Dav1dFrameContext *q_f;
    int q_state, q_rc;
    uint64_t q_arrival;
    Dav1dHipPicture q_filtered;
} FcState;

/* a device: the context frames begin, end and are fetched on, one for uploads and one for output work (streams of their own), three stage threads */
typedef struct Dev {
    Dav1dHipContext *ctx, *ctx_up, *ctx_out;
    Dav1dHipContext *ctx_peer;   /* (several devices) the stream the bands of other devices' pictures arrive on */
    pthread_t thread[3];
    int have_threads;
    void *targ[3][3];
    atomic_int n_frames, n_peer_copies, n_band_copies;
} Dev;

/* a frame ends badly because a frame it predicts from did: dav1d's error (DAV1D_ERR(EINVAL), as check_tile makes it), not the backend's */
#define GLUE_REF_FAILED (-1000)

struct Dav1dHipGlue {
    Dav1dHipGlueOptions o;
    Hip hip;
    Dev dev[DAV1D_HIP_GLUE_MAX_DEVICES];
    int n_dev;
    unsigned alloc_seq;
    Dav1dContext *c;
    unsigned n_fc;
    FcState *fcs;
    pthread_mutex_t q_mtx;
    pthread_cond_t q_cond;
    uint64_t q_arrivals;
    int q_stop;
    /* pictures between uses (dav1d's default allocator pools them too, src/picture.c:46-82 + src/mem.c) */
    Dav1dHipGluePicture *free_pics[32];
    int n_free_pics, closing;
    pthread_mutex_t pic_mtx;
    atomic_int n_row_publications, n_backend_failures;
};

static Dav1dHipGlue *g_glue;

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
static void stat_add(const Dav1dHipGlue *const g, const int what, const double t0) { if (g->o.stat) g->o.stat(g->o.cookie, what, t0); }
static FcState *state_of(const Dav1dFrameContext *const f) { return &g_glue->fcs[f - f->c->fc]; }

/* ------------------------------------------------------------------------------------------------ descriptors */
void dav1d_hip_glue_frame_desc(Dav1dHipFrameDesc *d, const Dav1dFrameContext *f) {
    memset(d, 0, sizeof(*d));
    d->w = f->cur.p.w; d->h = f->cur.p.h; d->layout = f->cur.p.layout; d->bpc = f->cur.p.bpc;
    d->sb128 = f->seq_hdr->sb128; d->intra_edge_filter = f->seq_hdr->intra_edge_filter;
    d->is_inter = IS_INTER_OR_SWITCH(f->frame_hdr);
    d->n_tile_cols = f->frame_hdr->tiling.cols; d->n_tile_rows = f->frame_hdr->tiling.rows;
    memcpy(d->col_start_sb, f->frame_hdr->tiling.col_start_sb, sizeof(d->col_start_sb));
    memcpy(d->row_start_sb, f->frame_hdr->tiling.row_start_sb, sizeof(d->row_start_sb));
    d->b4_stride = f->b4_stride;
    d->b = (const Dav1dHipAv1Block *) f->frame_thread.b;           /* same 32-byte layout, pinned by tests/test_lister.py */
    d->cbi = (const int16_t *) f->frame_thread.cbi;
    d->tile_start_off = f->frame_thread.tile_start_off;
    d->pal = f->frame_thread.pal;
    memcpy(d->svc, f->svc, sizeof(d->svc));
    for (int i = 0; i < 7; i++) { d->ref_w[i] = f->refp[i].p.p.w; d->ref_h[i] = f->refp[i].p.p.h; }
    memcpy(d->gmv, f->frame_hdr->gmv, sizeof(d->gmv));             /* Dav1dHipWarpParams == Dav1dWarpedMotionParams */
    memcpy(d->gmv_warp_allowed, f->gmv_warp_allowed, sizeof(d->gmv_warp_allowed));
    memcpy(d->jnt_weights, f->jnt_weights, sizeof(d->jnt_weights));
    d->cf_align64 = ARCH_X86_64;                                   /* the cf cursor realignment of src/decode.c:2209-2218 */
    memcpy(d->lossless, f->frame_hdr->segmentation.lossless, sizeof(d->lossless));   /* mask builder: src/decode.c:1889-1893 */
}

/* Dav1dSettings.inloop_filters (include/dav1d/dav1d.h:61-69): dav1d_filter_sbrow_deblock_cols / _rows return before they deblock without
 * DAV1D_INLOOPFILTER_DEBLOCK (src/recon_tmpl.c:1988, 2014 — the rows CDEF and restoration read across their borders are still kept,
 * un-deblocked: the lister takes them from the picture either way), dav1d_filter_sbrow_cdef without _CDEF (:2027), dav1d_filter_sbrow_lr
 * without _RESTORATION (:2089).  The filter lister deblocks when a luma level is set, runs CDEF when cdef_enabled and restores the planes
 * that have a type: a filter the application switched off is described as absent. */
void dav1d_hip_glue_filter_desc(Dav1dHipFilterDesc *fd, const Dav1dFrameContext *f) {
    const unsigned on = f->c->inloop_filters;
    memset(fd, 0, sizeof(*fd));
    if (on & DAV1D_INLOOPFILTER_DEBLOCK) {
        fd->lf_level_y[0] = f->frame_hdr->loopfilter.level_y[0]; fd->lf_level_y[1] = f->frame_hdr->loopfilter.level_y[1];
    }
    fd->lf_level_u = f->frame_hdr->loopfilter.level_u; fd->lf_level_v = f->frame_hdr->loopfilter.level_v;
    fd->lf_mask = (const Dav1dHipAv1Filter *) f->lf.mask;
    fd->tx_lpf_right_edge[0] = f->lf.tx_lpf_right_edge[0]; fd->tx_lpf_right_edge[1] = f->lf.tx_lpf_right_edge[1];
    fd->a_tx_lpf_y = f->a[0].tx_lpf_y; fd->a_tx_lpf_uv = f->a[0].tx_lpf_uv; fd->a_stride = sizeof(BlockContext);
    fd->cdef_enabled = f->seq_hdr->cdef && (on & DAV1D_INLOOPFILTER_CDEF);
    fd->cdef_damping = f->frame_hdr->cdef.damping;
    for (int i = 0; i < 8; i++) { fd->cdef_y_strength[i] = f->frame_hdr->cdef.y_strength[i]; fd->cdef_uv_strength[i] = f->frame_hdr->cdef.uv_strength[i]; }
    if (on & DAV1D_INLOOPFILTER_RESTORATION)
        for (int i = 0; i < 3; i++) fd->lr_type[i] = f->frame_hdr->restoration.type[i];
    fd->lr_unit_size[0] = f->frame_hdr->restoration.unit_size[0]; fd->lr_unit_size[1] = f->frame_hdr->restoration.unit_size[1];
    fd->lr_mask = (const Dav1dHipAv1Restoration *) f->lf.lr_mask;
    fd->sr_w = f->frame_hdr->width[0] != f->frame_hdr->width[1] ? f->sr_cur.p.p.w : 0;      /* restoration works on the upscaled frame */
}

/* HIP's current device belongs to the thread.  On a thread that is not the glue's (the allocator callbacks and the life-cycle calls run on the
 * application's) the glue selects its device and puts back what the thread had. */
static int borrow_thread(const Dav1dHipGlue *const g, Dav1dHipContext *const ctx) {
    const int prev = g->hip.current_device();
    (void) g->hip.use(ctx);
    return prev;
}
static void return_thread(const Dav1dHipGlue *const g, const int prev) { if (prev >= 0) (void) g->hip.set_device(prev); }

/* ------------------------------------------------------------------------------------------------ Dav1dPicAllocator */
static void free_mirrors(Dav1dHipGlue *g, Dav1dHipGluePicture *hp);
static int glue_alloc_picture(Dav1dPicture *const p, void *const cookie) {
    Dav1dHipGlue *const g = cookie;
    const double t0 = now_s();
    Dav1dHipGluePicture *hp = NULL;
    pthread_mutex_lock(&g->pic_mtx);
    const int dev = (int) (g->alloc_seq++ % (unsigned) g->n_dev);
    for (int i = 0; i < g->n_free_pics; i++) {
        const Dav1dHipPicture *const d = &g->free_pics[i]->hp.dev;
        if (g->free_pics[i]->dev == dev && d->p[0].w == p->p.w && d->p[0].h == p->p.h && d->layout == (int) p->p.layout && d->bpc == p->p.bpc) {
            hp = g->free_pics[i];
            g->free_pics[i] = g->free_pics[--g->n_free_pics];
            break;
        }
    }
    pthread_mutex_unlock(&g->pic_mtx);
    Dav1dHipContext *const ctx = g->dev[dev].ctx;
    const int thread_dev = borrow_thread(g, ctx);
    int rc = 0;
    const int pooled = hp != NULL;
    if (hp) rc = g->hip.memset_(ctx, hp->hp.dev.alloc, 0, hp->hp.dev.alloc_size);       /* as a fresh one: zero, padding included */
    if (!hp) {
        hp = calloc(1, sizeof(*hp));
        if (!hp) { return_thread(g, thread_dev); return DAV1D_ERR(ENOMEM); }
        hp->dev = dev;
        rc = g->hip.host_picture_alloc(ctx, &hp->hp, p->p.w, p->p.h, p->p.layout, p->p.bpc);   /* layouts share their values */
    }
    if (rc && pooled) {        /* a pooled picture owns pinned planes, device planes and mirrors: all of it goes, not just the record */
        free_mirrors(g, hp);
        (void) g->hip.host_picture_release(ctx, &hp->hp);
    }
    return_thread(g, thread_dev);
    stat_add(g, DAV1D_HIP_GLUE_STAT_PICTURE_ALLOC, t0);
    if (rc) { free(hp); return DAV1D_ERR(ENOMEM); }
    for (int i = 0; i < 3; i++) p->data[i] = hp->hp.data[i];
    p->stride[0] = hp->hp.stride[0]; p->stride[1] = hp->hp.stride[1];
    p->allocator_data = hp;
    hp->ref = hp->hp.dev;
    hp->ref_dev = hp->dev;
    memset(hp->mirror_ok, 0, sizeof(hp->mirror_ok));
    memset(hp->mirror_rows, 0, sizeof(hp->mirror_rows));
    atomic_store(&hp->failed, 0);
    atomic_store(&hp->final, 0);
    return 0;
}
static void free_mirrors(Dav1dHipGlue *const g, Dav1dHipGluePicture *const hp) {
    for (int d = 0; d < g->n_dev; d++)
        if (hp->mirror[d].alloc || hp->mirror[d].twin_alloc) { (void) g->hip.use(g->dev[d].ctx); g->hip.picture_free(g->dev[d].ctx, &hp->mirror[d]); }
}
static void glue_release_picture(Dav1dPicture *const p, void *const cookie) {
    Dav1dHipGlue *const g = cookie;
    Dav1dHipGluePicture *const hp = p->allocator_data;
    const double t0 = now_s();
    const int thread_dev = borrow_thread(g, g->dev[hp->ref_dev].ctx);       /* (a stage thread of the glue as often as an application thread) */
    if (hp->frame) g->hip.frame_destroy(hp->frame);
    hp->frame = NULL;
    hp->ref = hp->hp.dev;
    hp->ref_dev = hp->dev;
    hp->ref.twin_ok = hp->hp.dev.twin_ok = 0;
    memset(hp->mirror_ok, 0, sizeof(hp->mirror_ok));          /* (the mirrors' storage stays with the picture while it is pooled) */
    memset(hp->mirror_rows, 0, sizeof(hp->mirror_rows));
    pthread_mutex_lock(&g->pic_mtx);
    if (!g->closing && g->n_free_pics < 32) {
        g->free_pics[g->n_free_pics++] = hp;
        pthread_mutex_unlock(&g->pic_mtx);
        return_thread(g, thread_dev);
        stat_add(g, DAV1D_HIP_GLUE_STAT_PICTURE_RELEASE, t0);
        return;
    }
    pthread_mutex_unlock(&g->pic_mtx);
    free_mirrors(g, hp);
    (void) g->hip.use(g->dev[hp->dev].ctx);
    g->hip.host_picture_release(g->dev[hp->dev].ctx, &hp->hp);
    free(hp);
    return_thread(g, thread_dev);
    stat_add(g, DAV1D_HIP_GLUE_STAT_PICTURE_RELEASE, t0);
}

/* ------------------------------------------------------------------------------------------------ worker-thread side */
static void drop_frame_objects(Dav1dHipGlue *const g, FcState *const s) {
    if (s->lister || s->frame) (void) g->hip.use(g->dev[s->dev].ctx);
    if (s->lister) g->hip.lister_destroy(s->lister);
    if (s->frame) g->hip.frame_destroy(s->frame);
    s->lister = NULL; s->frame = NULL;
}

static void once_per_row(const Dav1dFrameContext *const f, const int sby) {
    Dav1dHipGlue *const g = g_glue;
    FcState *const s = state_of(f);
    if (atomic_exchange(&s->filter_listed[sby], 1)) return;
    const double t0 = now_s();
    (void) g->hip.use(g->dev[s->dev].ctx);
    const int rc = g->hip.lister_filter_sbrow(s->lister, &s->fd, sby);     /* instead of filter_sbrow* */
    if (rc) {
        /* the filter tasks return void (src/recon.h:47-53): the frame is marked the way a failed allocation marks it (src/thread_task.c:459-469) */
        atomic_store(&((Dav1dFrameContext *) f)->task_thread.error, -1);
        atomic_fetch_add(&g->n_backend_failures, 1);
        fprintf(stderr, "dav1d_hip_glue: dav1d_hip_lister_filter_sbrow(sby %d) = %d\n", sby, rc);
    }
    stat_add(g, DAV1D_HIP_GLUE_STAT_FILTER_LISTING, t0);
}
static void filter_f(Dav1dFrameContext *const f, const int sby) { once_per_row(f, sby); }
static void filter_t(Dav1dTaskContext *const tc, const int sby) { once_per_row(tc->f, sby); }

static int grow(Dav1dHipGlue *const g, Dav1dHipContext *const ctx, void **const p, size_t *const cap, const size_t bytes) {
    if (*cap >= bytes) return 0;
    if (*p) g->hip.free_(ctx, *p);
    *p = NULL; *cap = 0;
    size_t want = 1 << 16;
    while (want < bytes) want <<= 1;
    const int rc = g->hip.malloc_(ctx, p, want);
    if (!rc) *cap = want;
    return rc;
}

int dav1d_hip_glue_frame_init(Dav1dFrameContext *const f) {
    Dav1dHipGlue *const g = g_glue;
    FcState *const s = state_of(f);
    const Dav1dFrameHeader *const fh = f->frame_hdr;
    /* a frame that failed in pass 1 never reached dav1d_hip_glue_frame_complete: the task loop ended it itself, its objects are still here */
    drop_frame_objects(g, s);
    dav1d_hip_glue_frame_desc(&s->desc, f);
    Dav1dHipGluePicture *const cur = f->cur.allocator_data;        /* the picture of the CODED size (f->sr_cur's is the upscaled one under super-resolution) */
    s->dev = cur->dev;                                             /* the frame ends where its picture lives */
    Dav1dHipContext *const ctx = g->dev[s->dev].ctx;
    (void) g->hip.use(ctx);
    Dav1dHipPicture refs[7];
    const int n_refs = IS_INTER_OR_SWITCH(fh) ? 7 : 0;
    for (int i = 0; i < n_refs; i++) refs[i] = ((Dav1dHipGluePicture *) f->refp[i].p.allocator_data)->hp.dev;     /* geometry; the final ones at the end */
    int rc = g->hip.frame_begin(ctx, &s->frame, &cur->hp.dev, refs, n_refs);
    if (g->o.pack) s->desc.cf = f->frame_thread.cf;
    if (!rc) rc = g->hip.lister_create(&s->lister, &s->desc, s->frame);
    if (!rc && fh->width[0] != fh->width[1]) rc = g->hip.frame_set_super_res(s->frame, f->sr_cur.p.p.w);
    if (rc) { drop_frame_objects(g, s); return DAV1D_ERR(ENOMEM); }
    dav1d_hip_glue_filter_desc(&s->fd, f);
    if (f->sbh > s->sbh_cap) {
        free(s->filter_listed);
        s->filter_listed = calloc((size_t) f->sbh, sizeof(*s->filter_listed));
        if (!s->filter_listed) { s->sbh_cap = 0; drop_frame_objects(g, s); return DAV1D_ERR(ENOMEM); }
        s->sbh_cap = f->sbh;
    }
    for (int i = 0; i < f->sbh; i++) atomic_store(&s->filter_listed[i], 0);
    f->bd_fn.filter_sbrow_deblock_cols = filter_f;
    f->bd_fn.filter_sbrow_deblock_rows = filter_f;
    f->bd_fn.filter_sbrow_cdef = filter_t;
    f->bd_fn.filter_sbrow_resize = filter_f;
    f->bd_fn.filter_sbrow_lr = filter_f;
    return 0;
}

/* Listing ahead of the references (option free_listing): the rows of its references a tile-sbrow needs (decode_b's lowest_pixel
 * bookkeeping, src/decode.c:1957-1990) are forgotten — the pixels are read when the frame ends, and frames end in order — so check_tile
 * (src/thread_task.c:416-433) lets the pass-2 task through at once. */
void dav1d_hip_glue_after_entropy(Dav1dTaskContext *const t) {
    const Dav1dHipGlue *const g = g_glue;
    const Dav1dFrameContext *const f = t->f;
    if (!g->o.free_listing || !IS_INTER_OR_SWITCH(f->frame_hdr)) return;
    Dav1dTileState *const ts = t->ts;
    const int sby = (t->by - ts->tiling.row_start) >> f->sb_shift;
    for (int n = 0; n < 7; n++) ts->lowest_pixel[sby][n][0] = ts->lowest_pixel[sby][n][1] = INT_MIN;
}
static int glue_entropy(Dav1dTaskContext *const t) {
    const int rc = dav1d_decode_tile_sbrow(t);
    if (!rc) dav1d_hip_glue_after_entropy(t);
    return rc;
}

int dav1d_hip_glue_recon_tile_sbrow(Dav1dTaskContext *const t) {
    /* instead of dav1d_decode_tile_sbrow(tc) */
    Dav1dHipGlue *const g = g_glue;
    const Dav1dFrameContext *const f = t->f;
    const double t0 = now_s();
    (void) g->hip.use(g->dev[state_of(f)->dev].ctx);
    const int rc = g->hip.lister_tile_sbrow(state_of(f)->lister, t->ts->tiling.row, t->ts->tiling.col, t->by >> f->sb_shift);
    stat_add(g, DAV1D_HIP_GLUE_STAT_LISTING, t0);
    if (rc) {
        atomic_fetch_add(&g->n_backend_failures, 1);
        fprintf(stderr, "dav1d_hip_glue: dav1d_hip_lister_tile_sbrow(tile %d, %d, sby %d) = %d\n", t->ts->tiling.row, t->ts->tiling.col, t->by >> f->sb_shift, rc);
    }
    return rc ? 1 : 0;
}

/* the frame's tasks are through without an error (a worker thread, the scheduler's lock held): its turn on the stage threads comes */
void dav1d_hip_glue_frame_complete(Dav1dFrameContext *const f) {
    Dav1dHipGlue *const g = g_glue;
    FcState *const s = state_of(f);
    pthread_mutex_lock(&g->q_mtx);
    s->q_f = f;
    s->q_rc = 0;
    s->q_arrival = ++g->q_arrivals;
    s->q_state = 1;
    pthread_cond_broadcast(&g->q_cond);
    pthread_mutex_unlock(&g->q_mtx);
}

/* Dav1dHooks.before_init: a frame that failed in pass 1 never reached dav1d_hip_glue_frame_complete — the task loop ended it itself — and its objects are
 * still here, possibly with preparations of its last rows still packing coefficients out of f->frame_thread.cf on the library's threads (option
 * prep_async).  They are waited for (dav1d_hip_frame_destroy does) BEFORE dav1d_decode_frame_init may reallocate that array for the next frame. */
void dav1d_hip_glue_before_init(Dav1dFrameContext *const f) { drop_frame_objects(g_glue, state_of(f)); }

const Dav1dHooks dav1d_hip_glue_hooks = { dav1d_hip_glue_before_init, dav1d_hip_glue_frame_init, glue_entropy, dav1d_hip_glue_recon_tile_sbrow, dav1d_hip_glue_frame_complete };

/* ------------------------------------------------------------------------------------------------ the three stage threads */
/* stage 1 (any order): what the frame's launches read from the host side */
static int stage_upload(Dav1dHipGlue *const g, Dav1dFrameContext *const f) {
    FcState *const s = state_of(f);
    const Hip *const hip = &g->hip;
    Dev *const dv = &g->dev[s->dev];
    struct FcBufs *const b = &s->b[s->dev];
    const size_t cf_bytes = (size_t) f->frame_thread.cf_sz * 128 * 128 / 2;
    const size_t lvl_bytes = sizeof(*f->lf.level) * (size_t) f->sb128w * f->sb128h * 32 * 32;
    const size_t pal_idx_bytes = f->frame_hdr->allow_screen_content_tools ? (size_t) f->frame_thread.pal_idx_sz * 128 * 128 / 8 : 0;
    size_t n_const = 0;
    const uint8_t *const blob = hip->lister_const_masks(&n_const);
    const double t0 = now_s();
    int rc = g->o.pack ? 0 : grow(g, dv->ctx, &b->coef, &b->coef_cap, cf_bytes + 64);
    if (!rc) rc = grow(g, dv->ctx, &b->lvl, &b->lvl_cap, lvl_bytes + 64);
    if (!rc) rc = grow(g, dv->ctx, &b->prep, &b->prep_cap, hip->lister_prep_elems(s->lister) * 2 + 4096);
    const size_t mask_cap_before = b->mask_cap;
    if (!rc) rc = grow(g, dv->ctx, &b->mask, &b->mask_cap, hip->lister_mask_bytes(s->lister) + 4096);
    if (!rc && b->mask_cap != mask_cap_before) rc = hip->upload(dv->ctx_up, b->mask, blob, n_const);
    if (!rc && !g->o.pack) {
        rc = hip->upload(dv->ctx_up, b->coef, f->frame_thread.cf, cf_bytes);
        /* the arena has been consumed: the next frame's pass 1 finds it zero, as after the reference's inverse transforms
         * (src/itx_tmpl.c:60,108).  (The packing lister does that itself, block by block.) */
        if (!g->o.keep_cf) memset(f->frame_thread.cf, 0, cf_bytes);
    }
    if (!rc) rc = hip->upload(dv->ctx_up, b->lvl, f->lf.level, lvl_bytes);
    if (!rc && pal_idx_bytes && f->frame_thread.pal_idx) {
        /* palette indices (pal_pred's `idx`, src/recon_tmpl.c:1207-1224): the arena PAL tasks point into */
        rc = grow(g, dv->ctx, &b->pal_idx, &b->pal_idx_cap, pal_idx_bytes + 64);
        if (!rc) rc = hip->upload(dv->ctx_up, b->pal_idx, f->frame_thread.pal_idx, pal_idx_bytes);
        if (!rc) rc = hip->frame_submit_intra_step(s->frame, 0, NULL, 0, NULL, 0, b->pal_idx);
    }
    stat_add(g, DAV1D_HIP_GLUE_STAT_UPLOADS, t0);
    return rc;
}

/* have the frames this one predicts from ended?  (-1: one of them failed) */
static int refs_final(const Dav1dFrameContext *const f) {
    if (!IS_INTER_OR_SWITCH(f->frame_hdr)) return 1;
    int all = 1;
    for (int i = 0; i < 7; i++) {
        if (atomic_load(&f->refp[i].progress[1]) == FRAME_ERROR) return -1;
        Dav1dHipGluePicture *const rp = f->refp[i].p.allocator_data;
        const int fin = atomic_load(&rp->final);
        if (fin && atomic_load(&rp->failed)) return -1;       /* (ended badly, and dav1d_hip_frame_done has not said so yet) */
        all &= fin;
    }
    return all;
}

static int mirror_ready(Dav1dHipGlue *const g, Dav1dHipPicture *const m, const int d, const Dav1dHipPicture *const r) {
    if (m->alloc && (m->p[0].w != r->p[0].w || m->p[0].h != r->p[0].h || m->layout != r->layout || m->bpc != r->bpc)) g->hip.picture_free(g->dev[d].ctx, m);
    return m->alloc ? 0 : g->hip.picture_alloc(g->dev[d].ctx, m, r->p[0].w, r->p[0].h, r->layout, r->bpc);
}

/* Rows [.., rows) of `pic` — the picture frame f leaves, on device `from` — are final: the bands not yet across go to the devices on which a queued
 * frame predicts from the picture (dav1d starts frame n + 1 on the rows of frame n that progress[1] has published, src/thread_task.c:416-433;
 * across devices what can start early is the transfer).  Runs on `from`'s stage-2 thread inside dav1d_hip_frame_end, which goes on issuing work
 * for its own device afterwards: the thread's device is put back.  The consumer's side (resident_ref) only looks at mirror_rows once the
 * picture is final, i.e. after the last call of this. */
static void push_rows(Dav1dHipGlue *const g, Dav1dFrameContext *const f, const int rows, const Dav1dHipPicture *const pic) {
    Dav1dHipGluePicture *const out = f->sr_cur.p.allocator_data;
    const int from = state_of(f)->dev;
    if (!pic || pic->twin_ok == DAV1D_HIP_TWIN_ONLY) return;
    uint8_t want[DAV1D_HIP_GLUE_MAX_DEVICES] = { 0 };
    int any = 0;
    pthread_mutex_lock(&g->q_mtx);
    for (unsigned i = 0; i < g->n_fc; i++) {
        const FcState *const c = &g->fcs[i];
        if ((c->q_state != 1 && c->q_state != 2) || c->dev == from || !IS_INTER_OR_SWITCH(c->q_f->frame_hdr)) continue;
        for (int k = 0; k < 7; k++) if (c->q_f->refp[k].p.allocator_data == out) { want[c->dev] = 1; any = 1; }
    }
    pthread_mutex_unlock(&g->q_mtx);
    if (!any) return;
    const int h = pic->p[0].h, upto = rows >= h ? h : rows & ~7;
    for (int d = 0; d < g->n_dev; d++) {
        if (!want[d] || out->mirror_rows[d] < 0 || out->mirror_rows[d] >= upto) continue;
        (void) g->hip.use(g->dev[d].ctx);
        int rc = mirror_ready(g, &out->mirror[d], d, pic);
        if (!rc) rc = g->hip.picture_copy_peer_rows(g->dev[d].ctx_peer, &out->mirror[d], g->dev[from].ctx, pic, out->mirror_rows[d], upto);
        out->mirror_rows[d] = rc ? -1 : upto;
        if (!rc) atomic_fetch_add(&g->dev[d].n_band_copies, 1);
    }
    (void) g->hip.use(g->dev[from].ctx);
}

/* progress: rows of the frame's picture have become final on the device */
static void rows_final(void *const cookie, const int rows, const Dav1dHipPicture *const pic) {
    Dav1dFrameContext *const f = cookie;
    atomic_fetch_add(&g_glue->n_row_publications, 1);
    dav1d_hip_rows_done(f, (unsigned) rows);
    if (g_glue->n_dev > 1) push_rows(g_glue, f, rows, pic);
}

/* where frame-ending device `d` reads the final pixels of picture `rp`: the picture itself when it lives there, else its mirror on d, filled
 * on first use (stage 2 of a device is one thread: nobody else touches mirror[d] while the picture is referenced) */
static int resident_ref(Dav1dHipGlue *const g, Dav1dHipGluePicture *const rp, const int d, Dav1dHipPicture *const out) {
    if (rp->ref_dev == d) { *out = rp->ref; return 0; }
    const Hip *const hip = &g->hip;
    Dav1dHipPicture *const m = &rp->mirror[d];
    if (!rp->mirror_ok[d]) {
        const Dav1dHipPicture *const r = &rp->ref;
        int rc = mirror_ready(g, m, d, r);
        const int h = r->p[0].h;
        if (!rc && rp->mirror_rows[d] > 0 && r->twin_ok != DAV1D_HIP_TWIN_ONLY && m->p[0].data) {
            /* bands crossed while the picture's frame was ending (push_rows): what is missing follows on the same stream, the bands are
             * waited for, and the mirror's tiled twin is made HERE from its raster planes (a 15 us launch on this device instead of the twin's
             * 199 MB over the link) */
            if (rp->mirror_rows[d] < h) {
                rc = hip->picture_copy_peer_rows(g->dev[d].ctx_peer, m, g->dev[rp->ref_dev].ctx, r, rp->mirror_rows[d], h);
                if (!rc) atomic_fetch_add(&g->dev[d].n_band_copies, 1);
            }
            if (!rc) rc = hip->sync(g->dev[d].ctx_peer);
            (void) hip->use(g->dev[d].ctx);
            if (!rc && r->twin_ok && m->twin[0]) rc = hip->picture_retile(g->dev[d].ctx, m);
            if (!rc) rp->mirror_rows[d] = h;
        } else if (!rc && r->twin_ok == 1 && m->twin[0] && m->p[0].data) {
            /* raster planes and twin both valid: the raster planes cross (the picture is final: nothing of the source's stream to wait for), the
             * twin is made here — half the bytes over the link */
            rc = hip->picture_copy_peer_rows(g->dev[d].ctx, m, g->dev[rp->ref_dev].ctx, r, 0, h);
            if (!rc) rc = hip->picture_retile(g->dev[d].ctx, m);
        } else if (!rc) {
            rc = hip->picture_copy_peer(g->dev[d].ctx, m, g->dev[rp->ref_dev].ctx, r);
        }
        if (rc) return rc;
        rp->mirror_ok[d] = 1;
        atomic_fetch_add(&g->dev[d].n_peer_copies, 1);
    }
    *out = *m;
    return 0;
}

/* stage 2: "when the last task of the frame is in" */
static int stage_end(Dav1dHipGlue *const g, Dav1dFrameContext *const f, Dav1dHipPicture *const filtered) {
    FcState *const s = state_of(f);
    Dav1dHipGluePicture *const out = f->sr_cur.p.allocator_data;     /* the picture dav1d hands on: reference and output */
    const Hip *const hip = &g->hip;
    const struct FcBufs *const b = &s->b[s->dev];
    int rc = refs_final(f) < 0 ? GLUE_REF_FAILED : 0;
    const double t0 = now_s();
    if (g->o.frame_listed) g->o.frame_listed(g->o.cookie, f);
    if (!rc && IS_INTER_OR_SWITCH(f->frame_hdr)) {
        Dav1dHipPicture refs[7];
        for (int i = 0; i < 7 && !rc; i++) {            /* where those frames' final pixels are (a slot may repeat a picture: its mirror is made once) */
            rc = resident_ref(g, f->refp[i].p.allocator_data, s->dev, &refs[i]);
            if (rc) fprintf(stderr, "dav1d_hip_glue: reference %d could not be made resident on device %d: %d\n", i, s->dev, rc);
        }
        if (!rc) rc = hip->frame_set_refs(s->frame, refs, 7);
    }
    if (!rc) rc = hip->frame_set_filters(s->frame, b->lvl, f->b4_stride, f->lf.lim_lut.e, f->lf.lim_lut.i,
                                         f->frame_hdr->cdef.damping + f->cur.p.bpc - 8, NULL, 0);
    memset(filtered, 0, sizeof(*filtered));
    if (!rc && g->o.row_progress) rc = hip->frame_set_progress_callback(s->frame, rows_final, f);
    if (!rc) rc = hip->frame_end(s->frame, g->o.pack ? NULL : b->coef, b->prep, b->mask, filtered, NULL);
    atomic_fetch_add(&g->dev[s->dev].n_frames, 1);
    if (g->o.frame_end_seconds) g->o.frame_end_seconds(g->o.cookie, f, now_s() - t0);
    stat_add(g, DAV1D_HIP_GLUE_STAT_FRAME_END, t0);
    hip->lister_destroy(s->lister);
    s->lister = NULL;
    if (!rc) {
        out->ref = *filtered;                     /* later frames predict from this; the frame object lives as long as the picture */
        out->ref_dev = s->dev;                    /* (under super-resolution the upscaled picture dav1d allocated may be another device's: the pixels are the frame's) */
        out->frame = s->frame;
    } else {
        hip->frame_destroy(s->frame);
    }
    s->frame = NULL;
    return rc;
}

/* stage 3: the picture to the host planes the application sees, then the frame is done as far as dav1d is concerned */
static void stage_out(Dav1dHipGlue *const g, Dav1dFrameContext *const f, int rc, const Dav1dHipPicture *const filtered) {
    Dav1dHipGluePicture *const out = f->sr_cur.p.allocator_data;
    Dav1dHipContext *const ctx = g->dev[state_of(f)->dev].ctx;
    const double t0 = now_s();
    if (!rc) rc = g->hip.host_picture_fetch(ctx, &out->hp, filtered, 0, f->sr_cur.p.p.h);
    if (!rc) rc = g->hip.host_picture_wait(ctx);
    stat_add(g, DAV1D_HIP_GLUE_STAT_FETCH, t0);
    if (g->o.before_frame_done) g->o.before_frame_done(g->o.cookie, f, rc);
    if (rc && rc != GLUE_REF_FAILED) {
        atomic_fetch_add(&g->n_backend_failures, 1);
        fprintf(stderr, "dav1d_hip_glue: frame (order hint %d, type %d, %dx%d) failed in the backend: %d\n", f->frame_hdr->frame_offset, f->frame_hdr->frame_type, f->cur.p.w, f->cur.p.h, rc);
    }
    const int k = f->frame_hdr->frame_offset;
    dav1d_hip_frame_done(f, !rc ? 0 : rc == GLUE_REF_FAILED ? DAV1D_ERR(EINVAL) : DAV1D_ERR(EIO));
    if (g->o.after_frame_done) g->o.after_frame_done(g->o.cookie, k);
}

/* Stage 2 of device `dev` has nothing it may end yet: is there a frame queued for this device with a reference that HAS ended, on another
 * device, and is not resident here?  Copying it now hides the transfer behind the wait for the references that have not ended (q_mtx held). */
static Dav1dHipGluePicture *ref_to_fetch_ahead(const Dav1dHipGlue *const g, const int dev, const Dav1dHipGluePicture *const skip) {
    for (unsigned i = 0; i < g->n_fc; i++) {
        const FcState *const c = &g->fcs[i];
        if ((c->q_state != 1 && c->q_state != 2) || c->dev != dev || !IS_INTER_OR_SWITCH(c->q_f->frame_hdr)) continue;
        for (int k = 0; k < 7; k++) {
            Dav1dHipGluePicture *const rp = c->q_f->refp[k].p.allocator_data;
            if (rp == skip || !atomic_load(&rp->final) || atomic_load(&rp->failed) || atomic_load(&c->q_f->refp[k].progress[1]) == FRAME_ERROR) continue;
            if (rp->ref_dev != dev && !rp->mirror_ok[dev]) return rp;
        }
    }
    return NULL;
}

static void *stage_thread(void *const arg) {
    Dav1dHipGlue *const g = ((void **) arg)[0];
    const int stage = (int) (intptr_t) ((void **) arg)[1], dev = (int) (intptr_t) ((void **) arg)[2];
    (void) g->hip.use(g->dev[dev].ctx);          /* this thread's calls all go to one device */
    Dav1dHipGluePicture *no_luck = NULL;          /* a reference that could not be fetched ahead: left to stage_end, which reports */
    for (;;) {
        const double t_wait = now_s();
        FcState *s = NULL;
        pthread_mutex_lock(&g->q_mtx);
        for (;;) {
            if (g->q_stop) break;
            s = NULL;
            for (unsigned i = 0; i < g->n_fc; i++) {
                FcState *const c = &g->fcs[i];
                if (c->q_state != stage || c->dev != dev || (s && s->q_arrival < c->q_arrival)) continue;
                if (stage == 2 && !refs_final(c->q_f)) continue;        /* its references have not all ended yet */
                s = c;
            }
            if (s) break;
            Dav1dHipGluePicture *const ahead = stage == 2 && g->n_dev > 1 ? ref_to_fetch_ahead(g, dev, no_luck) : NULL;
            if (ahead) {
                Dav1dHipPicture unused;
                pthread_mutex_unlock(&g->q_mtx);
                if (resident_ref(g, ahead, dev, &unused)) no_luck = ahead;
                pthread_mutex_lock(&g->q_mtx);
                continue;
            }
            pthread_cond_wait(&g->q_cond, &g->q_mtx);
        }
        if (g->q_stop) { pthread_mutex_unlock(&g->q_mtx); break; }
        Dav1dFrameContext *const f = s->q_f;
        pthread_mutex_unlock(&g->q_mtx);
        if (stage == 2) stat_add(g, DAV1D_HIP_GLUE_STAT_GPU_IDLE, t_wait);
        if (stage == 1) s->q_rc = stage_upload(g, f);
        else if (stage == 2) {
            if (!s->q_rc) s->q_rc = stage_end(g, f, &s->q_filtered);
            else drop_frame_objects(g, s);
        }
        pthread_mutex_lock(&g->q_mtx);
        if (stage == 2) {
            Dav1dHipGluePicture *const op = f->sr_cur.p.allocator_data;
            if (s->q_rc) atomic_store(&op->failed, 1);      /* first: see Dav1dHipGluePicture.failed */
            atomic_store(&op->final, 1);                    /* well or badly: nobody waits for it any longer */
        }
        s->q_state = stage < 3 ? stage + 1 : 0;        /* (stage 3 first, its work after: dav1d may reuse the frame context from there on) */
        const int rc = s->q_rc;
        const Dav1dHipPicture filtered = s->q_filtered;
        pthread_cond_broadcast(&g->q_cond);
        pthread_mutex_unlock(&g->q_mtx);
        if (stage == 3) stage_out(g, f, rc, &filtered);
    }
    return NULL;
}

/* ------------------------------------------------------------------------------------------------ output */
int dav1d_hip_glue_output_with_grain(Dav1dHipGlue *const g, const Dav1dPicture *const pic, uint8_t *const dst[3]) {
    if (!g || !pic || !pic->allocator_data) return DAV1D_ERR(EINVAL);
    const Dav1dHipGluePicture *const hp = pic->allocator_data;
    const int bps = pic->p.bpc > 8 ? 2 : 1, n_pl = pic->p.layout == DAV1D_PIXEL_LAYOUT_I400 ? 1 : 3;
    const int ss_hor = pic->p.layout != DAV1D_PIXEL_LAYOUT_I444;
    Dav1dHipPicture grain;
    memset(&grain, 0, sizeof(grain));
    Dav1dHipContext *const ctx_out = g->dev[hp->ref_dev].ctx_out;
    const int thread_dev = borrow_thread(g, ctx_out);
    int rc = g->hip.picture_alloc(ctx_out, &grain, pic->p.w, pic->p.h, pic->p.layout, pic->p.bpc);
    if (rc) { return_thread(g, thread_dev); return DAV1D_ERR(ENOMEM); }
    rc = g->hip.fg_apply(ctx_out, &grain, &hp->ref, (const Dav1dHipFilmGrainData *) &pic->frame_hdr->film_grain.data,
                         pic->seq_hdr->mtrx == DAV1D_MC_IDENTITY);
    for (int pl = 0; pl < n_pl && !rc; pl++) {
        const int w = pl ? (pic->p.w + ss_hor) >> ss_hor : pic->p.w;
        rc = g->hip.plane_download(ctx_out, &grain, pl, dst[pl], (ptrdiff_t) w * bps, 0);
    }
    if (!rc) rc = g->hip.sync(ctx_out);
    g->hip.picture_free(ctx_out, &grain);
    return_thread(g, thread_dev);
    return rc ? DAV1D_ERR(EIO) : 0;
}

int dav1d_hip_glue_backend_failures(const Dav1dHipGlue *const g) { return g ? atomic_load(&g->n_backend_failures) : 0; }
int dav1d_hip_glue_row_publications(const Dav1dHipGlue *const g) { return g ? atomic_load(&g->n_row_publications) : 0; }
int dav1d_hip_glue_devices(const Dav1dHipGlue *const g) { return g ? g->n_dev : 0; }
int dav1d_hip_glue_device_stats(const Dav1dHipGlue *const g, const int d, int *const frames_ended, int *const peer_copies) {
    if (!g || d < 0 || d >= g->n_dev) return DAV1D_ERR(EINVAL);
    if (frames_ended) *frames_ended = atomic_load(&g->dev[d].n_frames);
    if (peer_copies) *peer_copies = atomic_load(&g->dev[d].n_peer_copies);
    return 0;
}
int dav1d_hip_glue_band_copies(const Dav1dHipGlue *const g, const int d) { return g && d >= 0 && d < g->n_dev ? atomic_load(&g->dev[d].n_band_copies) : 0; }
int dav1d_hip_glue_live_objects(const Dav1dHipGlue *const g, long long out[4]) { return g && g->hip.live_objects ? g->hip.live_objects(out) : DAV1D_ERR(EINVAL); }

/* ------------------------------------------------------------------------------------------------ life cycle */
#define SYM(field, name) do { *(void **) &g->hip.field = dlsym(g->hip.dl, name); if (!g->hip.field) goto fail; } while (0)

int dav1d_hip_glue_create(Dav1dHipGlue **const out, const Dav1dHipGlueOptions *const o) {
    if (!out || !o || !o->hip_lib) return DAV1D_ERR(EINVAL);
    *out = NULL;
    Dav1dHipGlue *const g = calloc(1, sizeof(*g));
    if (!g) return DAV1D_ERR(ENOMEM);
    g->o = *o;
    pthread_mutex_init(&g->q_mtx, NULL);
    pthread_mutex_init(&g->pic_mtx, NULL);
    pthread_cond_init(&g->q_cond, NULL);
    g->hip.dl = dlopen(o->hip_lib, RTLD_NOW | RTLD_LOCAL);
    if (!g->hip.dl) goto fail;
    SYM(open, "dav1d_hip_open"); SYM(close, "dav1d_hip_close"); SYM(sync, "dav1d_hip_sync"); SYM(malloc_, "dav1d_hip_malloc"); SYM(free_, "dav1d_hip_free");
    SYM(upload, "dav1d_hip_upload"); SYM(memset_, "dav1d_hip_memset"); SYM(host_picture_alloc, "dav1d_hip_host_picture_alloc"); SYM(host_picture_release, "dav1d_hip_host_picture_release");
    SYM(host_picture_fetch, "dav1d_hip_host_picture_fetch"); SYM(host_picture_wait, "dav1d_hip_host_picture_wait");
    SYM(frame_begin, "dav1d_hip_frame_begin"); SYM(frame_set_refs, "dav1d_hip_frame_set_refs"); SYM(frame_set_filters, "dav1d_hip_frame_set_filters");
    SYM(frame_end, "dav1d_hip_frame_end"); SYM(frame_destroy, "dav1d_hip_frame_destroy");
    SYM(lister_create, "dav1d_hip_lister_create"); SYM(lister_tile_sbrow, "dav1d_hip_lister_tile_sbrow"); SYM(lister_filter_sbrow, "dav1d_hip_lister_filter_sbrow");
    SYM(lister_prep_elems, "dav1d_hip_lister_prep_elems"); SYM(lister_mask_bytes, "dav1d_hip_lister_mask_bytes");
    SYM(lister_const_masks, "dav1d_hip_lister_const_masks"); SYM(lister_destroy, "dav1d_hip_lister_destroy");
    SYM(frame_submit_intra_step, "dav1d_hip_frame_submit_intra_step"); SYM(frame_set_super_res, "dav1d_hip_frame_set_super_res");
    SYM(fg_apply, "dav1d_hip_fg_apply"); SYM(picture_alloc, "dav1d_hip_picture_alloc"); SYM(picture_free, "dav1d_hip_picture_free");
    SYM(plane_download, "dav1d_hip_plane_download"); SYM(frame_set_progress_callback, "dav1d_hip_frame_set_progress_callback");
    SYM(live_objects, "dav1d_hip_live_objects"); SYM(device_count, "dav1d_hip_device_count"); SYM(use, "dav1d_hip_context_use");
    SYM(enable_peer_access, "dav1d_hip_enable_peer_access"); SYM(current_device, "dav1d_hip_current_device"); SYM(set_device, "dav1d_hip_set_device");
    SYM(picture_copy_peer, "dav1d_hip_picture_copy_peer"); SYM(picture_copy_peer_rows, "dav1d_hip_picture_copy_peer_rows"); SYM(picture_retile, "dav1d_hip_picture_retile");
    g->n_dev = o->n_devices > 1 ? o->n_devices : 1;
    if (g->n_dev > DAV1D_HIP_GLUE_MAX_DEVICES || o->device < 0 || o->device + g->n_dev > g->hip.device_count()) { g->n_dev = 0; goto fail; }
    const int thread_dev = g->hip.current_device();          /* (dav1d_hip_open selects the device it opens on) */
    for (int d = 0; d < g->n_dev; d++) {
        Dev *const dv = &g->dev[d];
        if (g->hip.open(&dv->ctx, o->device + d, NULL) || g->hip.open(&dv->ctx_up, o->device + d, NULL) || g->hip.open(&dv->ctx_out, o->device + d, NULL) ||
            (g->n_dev > 1 && g->hip.open(&dv->ctx_peer, o->device + d, NULL))) {
            return_thread(g, thread_dev);
            goto fail;
        }
    }
    for (int d = 0; d < g->n_dev; d++)               /* reference pictures cross between any two of them: direct copies where the devices are peers */
        for (int e = d + 1; e < g->n_dev; e++) (void) g->hip.enable_peer_access(g->dev[d].ctx, g->dev[e].ctx);
    return_thread(g, thread_dev);
    *out = g;
    return 0;
fail:
    dav1d_hip_glue_destroy(g);
    return DAV1D_ERR(ENOSYS);
}

void dav1d_hip_glue_settings(Dav1dHipGlue *const g, Dav1dSettings *const s) {
    s->allocator.cookie = g;
    s->allocator.alloc_picture_callback = glue_alloc_picture;
    s->allocator.release_picture_callback = glue_release_picture;
    s->apply_grain = 0;          /* dav1d_apply_grain reads host planes: with the backend the grain goes on on the device (dav1d_hip_glue_output_with_grain) */
}

int dav1d_hip_glue_attach(Dav1dHipGlue *const g, Dav1dContext *const c) {
    if (!g || !c || c->n_fc < 2) return DAV1D_ERR(EINVAL);         /* the two-pass hand-off only exists with frame threading (src/decode.c:2801, 3007) */
    if (!g->fcs) {
        g->c = c;
        g->n_fc = c->n_fc;
        g->fcs = calloc(g->n_fc, sizeof(*g->fcs));
        if (!g->fcs) return DAV1D_ERR(ENOMEM);
    }
    for (unsigned i = 0; i < g->n_fc; i++) g->fcs[i].q_state = 0;
    g->q_stop = 0;
    g_glue = g;
    for (int d = 0; d < g->n_dev; d++) {
        Dev *const dv = &g->dev[d];
        for (int k = 0; k < 3; k++) {
            dv->targ[k][0] = g; dv->targ[k][1] = (void *) (intptr_t) (k + 1); dv->targ[k][2] = (void *) (intptr_t) d;
            if (pthread_create(&dv->thread[k], NULL, stage_thread, dv->targ[k])) { dav1d_hip_glue_detach(g); return DAV1D_ERR(EAGAIN); }
            dv->have_threads = k + 1;
        }
    }
    dav1d_hooks = &dav1d_hip_glue_hooks;
    return 0;
}

void dav1d_hip_glue_detach(Dav1dHipGlue *const g) {
    if (!g) return;
    pthread_mutex_lock(&g->q_mtx);
    g->q_stop = 1;
    pthread_cond_broadcast(&g->q_cond);
    pthread_mutex_unlock(&g->q_mtx);
    for (int d = 0; d < g->n_dev; d++) {
        for (int k = 0; k < g->dev[d].have_threads; k++) pthread_join(g->dev[d].thread[k], NULL);
        g->dev[d].have_threads = 0;
    }
    if (dav1d_hooks == &dav1d_hip_glue_hooks) dav1d_hooks = NULL;
}

/* after dav1d_close (which releases the last pictures through the allocator) */
void dav1d_hip_glue_destroy(Dav1dHipGlue *const g) {
    if (!g) return;
    dav1d_hip_glue_detach(g);
    g->closing = 1;
    const int thread_dev = g->hip.current_device ? g->hip.current_device() : -1;
    for (int i = 0; i < g->n_free_pics; i++) {
        Dav1dHipGluePicture *const hp = g->free_pics[i];
        free_mirrors(g, hp);
        (void) g->hip.use(g->dev[hp->dev].ctx);
        g->hip.host_picture_release(g->dev[hp->dev].ctx, &hp->hp);
        free(hp);
    }
    g->n_free_pics = 0;
    if (g->fcs) {
        for (unsigned i = 0; i < g->n_fc; i++) {
            FcState *const s = &g->fcs[i];
            for (int d = 0; d < g->n_dev; d++) {
                struct FcBufs *const b = &s->b[d];
                Dav1dHipContext *const ctx = g->dev[d].ctx;
                if (!ctx) continue;
                (void) g->hip.use(ctx);
                if (b->coef) g->hip.free_(ctx, b->coef);
                if (b->lvl) g->hip.free_(ctx, b->lvl);
                if (b->prep) g->hip.free_(ctx, b->prep);
                if (b->mask) g->hip.free_(ctx, b->mask);
                if (b->pal_idx) g->hip.free_(ctx, b->pal_idx);
            }
            if (g->n_dev && g->dev[s->dev].ctx) drop_frame_objects(g, s);      /* frames that failed in pass 1 and were never followed by another frame on their context */
            free(s->filter_listed);
        }
        free(g->fcs);
    }
    for (int d = 0; d < DAV1D_HIP_GLUE_MAX_DEVICES; d++) {
        if (g->dev[d].ctx_peer) g->hip.close(g->dev[d].ctx_peer);
        if (g->dev[d].ctx_out) g->hip.close(g->dev[d].ctx_out);
        if (g->dev[d].ctx_up) g->hip.close(g->dev[d].ctx_up);
        if (g->dev[d].ctx) g->hip.close(g->dev[d].ctx);
    }
    if (thread_dev >= 0 && g->hip.set_device) (void) g->hip.set_device(thread_dev);
    if (g->hip.dl) dlclose(g->hip.dl);
    pthread_mutex_destroy(&g->q_mtx);
    pthread_mutex_destroy(&g->pic_mtx);
    pthread_cond_destroy(&g->q_cond);
    if (g_glue == g) g_glue = NULL;
    free(g);
}

/* The reference-side binding of the HIP backend: what a dav1d maintainer adds to dav1d (1.5.4) next to patches/dav1d-1.5.4-hip.patch.
 *
 * This file and dav1d_glue.c are compiled INTO dav1d (they include dav1d's internal headers, src/internal.h etc.) and talk to
 * libdav1d_hip.so through include/dav1d_hip.h, which they dlopen: no link-time dependency, a dav1d without the library behaves as
 * before.  The patch adds five hook points to src/thread_task.c (dav1d_hooks.h); the functions below are what goes behind them.
 * INTEGRATION.md 2 walks through it; oracle/ref_hooked.c (test infrastructure) is one user: it opens dav1d with the glue's allocator,
 * points the hooks here and compares every picture with dav1d's own.
 *
 *   Dav1dSettings s; dav1d_default_settings(&s);
 *   dav1d_hip_glue_create(&g, &opts);            // dlopen("libdav1d_hip.so"), three device contexts per device (opts.n_devices)
 *   dav1d_hip_glue_settings(g, &s);              // Dav1dPicAllocator on pinned host planes + device pictures
 *   dav1d_open(&c, &s);
 *   dav1d_hip_glue_attach(g, c);                 // per-frame-context state, the three stage threads, dav1d_hooks = the glue's
 *   ... dav1d_send_data / dav1d_get_picture as always ...
 *   dav1d_hip_glue_detach(g);                    // after the last picture has been drained: stage threads stop, hooks off
 *   dav1d_close(&c);                             // releases the last pictures through the allocator
 *   dav1d_hip_glue_destroy(g);
 *
 * Contract honoured here (reference file:line):
 *   - Dav1dSettings.inloop_filters (include/dav1d/dav1d.h:61-69, src/recon_tmpl.c:1988, 2014, 2027, 2089): a filter the application
 *     switched off is not listed (dav1d_hip_glue_filter_desc).
 *   - errors (src/thread_task.c:459-469, src/decode.c:3242-3251): a frame whose pass 1 fails never reaches the backend (the task loop
 *     ends it itself; its lister and frame object are dropped when the frame context is next used or at detach); a frame whose
 *     reference failed ends with DAV1D_ERR(EINVAL) like dav1d's own (check_tile, src/thread_task.c:393-439), and FRAME_ERROR goes
 *     into progress[1] for its dependants; a failure INSIDE the backend ends the frame with DAV1D_ERR(EIO) and is counted
 *     (dav1d_hip_glue_backend_failures).
 *   - progress (src/thread_task.c:888-896): rows are published as the backend reports them (option row_progress) and all at once
 *     when the frame ends. */
#ifndef DAV1D_HIP_GLUE_H
#define DAV1D_HIP_GLUE_H
#include "src/internal.h"
#include "dav1d_hooks.h"
#include "dav1d_hip.h"

typedef struct Dav1dHipGlue Dav1dHipGlue;
#define DAV1D_HIP_GLUE_MAX_DEVICES 8

/* what the glue hangs on a Dav1dPicture (Dav1dPicture.allocator_data) */
typedef struct Dav1dHipGluePicture {
    Dav1dHipHostPicture hp;      /* pinned host planes (what the application reads) + the device picture of the same geometry */
    Dav1dHipFrame *frame;        /* the frame that produced the picture: owns `ref` when that is not hp.dev */
    Dav1dHipPicture ref;         /* where the final pixels are on the device: what later frames predict from */
    atomic_int final;            /* the frame that produced the picture has ended (well or badly): `ref` is settled */
    atomic_int failed;           /* ... badly: set BEFORE final, so that whoever sees final also sees this — dav1d's own mark, progress[1] =
                                    FRAME_ERROR, is only stored a stage later (dav1d_hip_frame_done), and a frame that predicts from this picture
                                    must not end on its unwritten pixels in between */
    int dev, ref_dev;            /* index (from Dav1dHipGlueOptions.device) of the device hp was made on / the one `ref` lives on */
    Dav1dHipPicture mirror[DAV1D_HIP_GLUE_MAX_DEVICES];   /* `ref` once more on the other devices that end frames predicting from it */
    uint8_t mirror_ok[DAV1D_HIP_GLUE_MAX_DEVICES];
    int mirror_rows[DAV1D_HIP_GLUE_MAX_DEVICES];          /* luma rows of `ref` that have crossed to mirror[d] band by band while its frame was still ending
                                                             (rows_final; the rest follows in resident_ref); -1: a band failed, the whole picture goes */
} Dav1dHipGluePicture;

typedef struct Dav1dHipGlueOptions {
    const char *hip_lib;         /* path of libdav1d_hip.so */
    int device;                  /* the first device */
    int n_devices;               /* 0 / 1: that device only; N: devices device .. device + N - 1 — frame k's picture is allocated on device k mod N and
                                    the frame ends there; a reference that lives elsewhere is copied over first (dav1d_hip_picture_copy_peer) */
    int pack;                    /* 1: the lister packs the coefficients (Dav1dHipFrameDesc.cf = f->frame_thread.cf): what exists travels with the
                                    frame's lists, cf is left zero as the reference's inverse transforms leave it; 0: the dense arena is uploaded */
    int free_listing;            /* 1: a frame is listed without waiting for the rows of its references (the pixels are read when the frame ends,
                                    and frames end in order): decode_b's lowest_pixel notes are dropped; 0: dav1d's own rule */
    int row_progress;            /* 1: rows reach sr_cur.progress[1] band by band while the frame's last stage runs — and, with several devices, cross
                                    to the devices whose queued frames predict from the picture as they are published (dav1d_hip_picture_copy_peer_rows):
                                    when the frame has ended, its consumers on other devices find the picture (nearly) there */
    int keep_cf;                 /* (pack = 0) 1: f->frame_thread.cf is left as it is after the upload — a caller that lends the frame context
                                    arrays of its own; 0: zeroed, as the reference's inverse transforms leave it (src/itx_tmpl.c:60,108) */
    /* observers, all optional (a test harness keeps its books through them; a plain dav1d leaves them NULL) */
    void *cookie;
    void (*stat)(void *cookie, int what, double t0_seconds);                      /* `what`: DAV1D_HIP_GLUE_STAT_* below, t0 = when it started */
    void (*frame_listed)(void *cookie, Dav1dFrameContext *f);                      /* on the GPU thread, before the frame ends on the device */
    void (*frame_end_seconds)(void *cookie, Dav1dFrameContext *f, double seconds); /* what dav1d_hip_frame_end took */
    void (*before_frame_done)(void *cookie, Dav1dFrameContext *f, int rc);         /* the last moment the frame context is the glue's */
    void (*after_frame_done)(void *cookie, int frame_offset);
} Dav1dHipGlueOptions;
enum { DAV1D_HIP_GLUE_STAT_PICTURE_ALLOC = 0, DAV1D_HIP_GLUE_STAT_LISTING = 2, DAV1D_HIP_GLUE_STAT_FILTER_LISTING = 3, DAV1D_HIP_GLUE_STAT_GPU_IDLE = 4,
       DAV1D_HIP_GLUE_STAT_UPLOADS = 5, DAV1D_HIP_GLUE_STAT_FRAME_END = 6, DAV1D_HIP_GLUE_STAT_FETCH = 7, DAV1D_HIP_GLUE_STAT_PICTURE_RELEASE = 8 };

int dav1d_hip_glue_create(Dav1dHipGlue **out, const Dav1dHipGlueOptions *o);
void dav1d_hip_glue_settings(Dav1dHipGlue *g, Dav1dSettings *s);
int dav1d_hip_glue_attach(Dav1dHipGlue *g, Dav1dContext *c);
void dav1d_hip_glue_detach(Dav1dHipGlue *g);
void dav1d_hip_glue_destroy(Dav1dHipGlue *g);

/* ---- what goes behind the hook points (dav1d_hooks.h).  dav1d_hip_glue_hooks has all five; a caller with hooks of its own (the test
 * harness injects pass 1's output first) calls these from them. */
extern const Dav1dHooks dav1d_hip_glue_hooks;
void dav1d_hip_glue_before_init(Dav1dFrameContext *f);        /* Dav1dHooks.before_init: lets go of what a frame that failed in pass 1 left on the context */
int dav1d_hip_glue_frame_init(Dav1dFrameContext *f);           /* Dav1dHooks.after_init: frame + lister, the filter stages pointed at the filter lister */
void dav1d_hip_glue_after_entropy(Dav1dTaskContext *t);        /* behind a successful pass-1 tile-sbrow (option free_listing) */
int dav1d_hip_glue_recon_tile_sbrow(Dav1dTaskContext *t);      /* Dav1dHooks.recon_tile_sbrow: instead of dav1d_decode_tile_sbrow(pass 2) */
void dav1d_hip_glue_frame_complete(Dav1dFrameContext *f);      /* Dav1dHooks.frame_complete */

/* ---- the descriptors the backend is handed (also for callers that feed a generator with them) */
void dav1d_hip_glue_frame_desc(Dav1dHipFrameDesc *d, const Dav1dFrameContext *f);
void dav1d_hip_glue_filter_desc(Dav1dHipFilterDesc *fd, const Dav1dFrameContext *f);

/* ---- output: film grain on the device for a picture dav1d_get_picture handed out (Dav1dSettings.apply_grain = 0 with the backend:
 * dav1d_apply_grain, src/lib.c:311-329, reads host planes).  Tight rows into dst[pl] (row pitch = plane width in bytes). */
int dav1d_hip_glue_output_with_grain(Dav1dHipGlue *g, const Dav1dPicture *pic, uint8_t *const dst[3]);

int dav1d_hip_glue_backend_failures(const Dav1dHipGlue *g);    /* frames that failed INSIDE the backend (not: frames dav1d rejects) */
int dav1d_hip_glue_row_publications(const Dav1dHipGlue *g);
int dav1d_hip_glue_devices(const Dav1dHipGlue *g);
/* device d (0 .. devices - 1): frames that ended on it, reference pictures copied TO it from another device */
int dav1d_hip_glue_device_stats(const Dav1dHipGlue *g, int d, int *frames_ended, int *peer_copies);
/* bands of other devices' pictures that crossed to device d while their frames were still ending, or as the remainder behind such bands (option row_progress) */
int dav1d_hip_glue_band_copies(const Dav1dHipGlue *g, int d);
/* objects of libdav1d_hip alive (dav1d_hip_live_objects): contexts, frames, listers, host pictures */
int dav1d_hip_glue_live_objects(const Dav1dHipGlue *g, long long out[4]);
#endif

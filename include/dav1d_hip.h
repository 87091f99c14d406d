/*
 * dav1d_hip.h — C ABI of the MI355X (gfx950) block-reconstruction backend.
 *
 * This is the drop-in boundary for dav1d's post-entropy hot path: the seven
 * Dav1dDSPContext function tables (reference src/internal.h:62-70) and the
 * pass-2 reconstruction hand-off (reference src/internal.h:276-293).
 *
 * Three levels are exported, all plain C (pointers + sizes, no C++/torch types):
 *
 *   1. Batched entry points (dav1d_hip_*_batch, *_list_*): one call = one kernel family
 *      over a flat list of POD task descriptors, all buffers device-resident.
 *      This is what a pass-2 "lister" inside dav1d submits per tile-sbrow / frame.
 *
 *   2. One frame in flight (dav1d_hip_frame_*): the driver-level boundary -- tasks appended
 *      per tile-sbrow from any worker thread, one call runs the frame's stages in order.
 *
 *   3. A kernel-level drop-in table (Dav1dHipDSPContext, dav1d_hip_dsp_init_*):
 *      function pointers with the reference's exact DSP signatures that stage
 *      host memory through the same kernels one call at a time.  It exists so the
 *      unmodified reference call sites / parity tests can run against the backend;
 *      it is not the fast path.
 *
 * Conventions: every function returns 0 or a negative errno (like DAV1D_ERR(),
 * reference include/dav1d/common.h); strides are in BYTES like the reference's
 * ptrdiff_t strides; `bitdepth_max` is (1 << bpc) - 1 as in HIGHBD_DECL_SUFFIX
 * (reference include/common/bitdepth.h:69).  pixel = uint8_t @8bpc, uint16_t @10/12bpc;
 * coef = int16_t @8bpc, int32_t @10/12bpc (reference include/common/bitdepth.h:44-63).
 */
#ifndef DAV1D_HIP_H
#define DAV1D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DAV1D_HIP_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ context */

typedef struct Dav1dHipContext Dav1dHipContext;

/* Opens the backend on HIP device `device`.  `stream` is a hipStream_t owned by the
 * caller (e.g. torch.cuda.current_stream().cuda_stream) or NULL for a private
 * stream.  All batched calls are asynchronous on that stream.
 * Fails with -ENODEV if no gfx950 device is usable: there is no CPU fallback.
 * The context makes its side streams here, and the ORDER streams are made in decides which of them share a hardware queue (the runtime
 * deals them over GPU_MAX_HW_QUEUES = 4 queues, each in order): DAV1D_HIP_STREAM_PAD (default 1; "a,b,..." = per context in the order they
 * are opened) unused streams are made in front of the side streams, DAV1D_HIP_RECON_PAIR_FIRST (1 .. 3, default 2) is the first side stream
 * of the paired reconstruction launches — both read at open only; DESIGN.md 9 has the measurements. */
DAV1D_HIP_API int dav1d_hip_open(Dav1dHipContext **out, int device, void *stream);
DAV1D_HIP_API void dav1d_hip_close(Dav1dHipContext *c);
DAV1D_HIP_API int dav1d_hip_sync(Dav1dHipContext *c);
DAV1D_HIP_API void *dav1d_hip_stream(Dav1dHipContext *c);

/* Launch-bound sequences (the intra wavefront of a frame: ~100-200 small dependent launches; the reference runs the same
 * chain block by block inside recon_b_intra, src/recon_tmpl.c:1176-1560) can be recorded once and replayed as one HIP
 * graph.  Between begin and end, every *_list_run / *_run_batch call on this context is captured instead of executed; only
 * calls that neither allocate, copy to/from the host nor synchronise may be made (the list-run entry points qualify; the
 * *_batch conveniences and the first run of an mc list do not).  The graph replays the captured launches with the captured
 * pointers: the pictures, arenas and lists they name must stay alive, and the contents they read are whatever is there at
 * replay time.  -ENOSYS when the runtime cannot capture. */
typedef struct Dav1dHipGraph Dav1dHipGraph;
DAV1D_HIP_API int dav1d_hip_graph_begin(Dav1dHipContext *c);
DAV1D_HIP_API int dav1d_hip_graph_end(Dav1dHipContext *c, Dav1dHipGraph **out);
DAV1D_HIP_API int dav1d_hip_graph_launch(Dav1dHipContext *c, const Dav1dHipGraph *g);
DAV1D_HIP_API size_t dav1d_hip_graph_nodes(const Dav1dHipGraph *g);      /* launches (and other nodes) recorded */
DAV1D_HIP_API void dav1d_hip_graph_destroy(Dav1dHipContext *c, Dav1dHipGraph *g);
/* Tuning knobs are context state; the environment variables DESIGN.md lists only supply the defaults when the context is opened.
 * name = the variable without its DAV1D_HIP_ prefix, lower case ("recon_fuse", "recon_pipeline", "recon_lanes", "post_bands",
 * "serial", "cdef_unit", "flow_groups", "flow_mode", "flow_min_steps", "chunk_arena_min",
 * "intra_sb": the intra blocks of a frame superblock by superblock — 2, the default: every frame whose tiling is known; 1: only
 * wavefronts of at least flow_min_steps steps; 0: never; "intra_sb_lds": 1 = the form of that route that keeps the superblock's
 * pixels in LDS (4:2:0 / 4:0:0), 0 (default) = pixels handed over through the L2; "intra_sb_waves": 1 / 4 / 8 waves per superblock, 0 (default) = in the
 * one-launch form, for frames that copy no blocks, 1 where the superblocks hold "intra_sb_one_below" units or fewer on average (default 0: never — measured
 * no faster), 4 where a level holds 128 superblocks or more on average, else 8;
 * "intra_sb_flow": 1 (default) = all levels of a frame's superblocks as ONE launch, a superblock waiting for the neighbours it reads, 0 = a launch per level,
 * "prep_async": 1 (default) = the chunk preparation of a tile-sbrow the lister hands in runs on threads of the library for frames of at most 8 tiles (the
 * listing thread goes on with the tile's next row; errors surface at dav1d_hip_frame_end), 2 = for every frame, 0 = on the submitting thread;
 * "chunk_order": 1 = the prepared lists of a tile-sbrow are ordered for the device — by code path and reference, a few per cent on the
 * launches for a tenth more host time per frame; 0, the default, leaves decode order); -EINVAL for an unknown name. */
DAV1D_HIP_API int dav1d_hip_set_option(Dav1dHipContext *c, const char *name, long value);
/* Reads back what a context counts or was set to: "intra_sb_fallbacks" = frames of this context whose one-launch intra pass had workgroups give
 * up waiting for a neighbour and was finished by launches per level (0 in a sound run: the one-launch form rests on workgroups being dispatched
 * in index order and staying resident; the fall-back needs neither); "intra_sb_waves", "intra_sb_one_below", "recon_fuse", "recon_pair_streams",
 * "recon_pair_first", "ref_twin".  -EINVAL for an unknown name. */
DAV1D_HIP_API int dav1d_hip_get_option(Dav1dHipContext *c, const char *name, long *value);
DAV1D_HIP_API const char *dav1d_hip_version(void);
/* Measurement aid: device time (HIP events on the context's stream) of the kernel launches of the most recent
 * itx_add / cdef / lf / ipred / lr / fg *_batch call on this context -- excludes the task upload the batch calls do. */
DAV1D_HIP_API float dav1d_hip_last_kernel_ms(Dav1dHipContext *c);

/* Device memory helpers (callers may equally pass memory from their own allocator). */
DAV1D_HIP_API int dav1d_hip_malloc(Dav1dHipContext *c, void **dev, size_t bytes);
DAV1D_HIP_API int dav1d_hip_free(Dav1dHipContext *c, void *dev);
DAV1D_HIP_API int dav1d_hip_memset(Dav1dHipContext *c, void *dev, int v, size_t bytes);
/* Every entry point returns 0 or a negative errno.  When the cause was a HIP call (-EIO, -ENOMEM, -ENOSYS ...), this returns
 * the text of the last HIP error seen on the calling thread ("no error" if none) and its hipError_t value in *code (NULL: not
 * wanted).  Asynchronous faults of a kernel surface at the next synchronising call of the context. */
DAV1D_HIP_API const char *dav1d_hip_last_hip_error(int *code);
DAV1D_HIP_API int dav1d_hip_upload(Dav1dHipContext *c, void *dev, const void *host, size_t bytes);
DAV1D_HIP_API int dav1d_hip_download(Dav1dHipContext *c, void *host, const void *dev, size_t bytes);

/* ----------------------------------------------------------------- pictures */

enum Dav1dHipPixelLayout { /* == enum Dav1dPixelLayout, reference include/dav1d/headers.h */
    DAV1D_HIP_LAYOUT_I400 = 0,
    DAV1D_HIP_LAYOUT_I420 = 1,
    DAV1D_HIP_LAYOUT_I422 = 2,
    DAV1D_HIP_LAYOUT_I444 = 3,
};

typedef struct Dav1dHipPlane {
    void *data;        /* device pointer to pixel (0,0) */
    ptrdiff_t stride;  /* bytes */
    int w, h;          /* visible size in pixels */
} Dav1dHipPlane;

typedef struct Dav1dHipPicture {
    Dav1dHipPlane p[3];
    int bpc;           /* 8, 10 or 12 */
    int layout;        /* enum Dav1dHipPixelLayout */
    void *alloc;       /* base of the allocation (owned when made by picture_alloc) */
    size_t alloc_size;
    /* Optional "tiled twin": the same pixels once more, stored as 8x8 tiles (64 consecutive pixels per tile: one 128-byte memory
     * line at 10 / 12 bits; tile (tx, ty) of a plane at pixel offset ty * 8 * stride_in_pixels + tx * 64, row r of it 8 r further),
     * each plane with the size and stride of its raster plane.  Motion compensation reads a reference through its twin when
     * EVERY reference of the call has one that is valid (twin_ok): a prediction window then touches a third to a half of the
     * memory lines it touches in raster order.  dav1d_hip_picture_retile fills it; dav1d_hip_frame_end /
     * dav1d_hip_recon_list_run do so for the picture they produce when it has the storage.  Whoever changes the raster planes by
     * other means clears twin_ok (or retiles).  twin_alloc: the storage (owned when made by the library).
     * twin_ok: 0 = only the raster planes hold the picture; 1 = twin and raster planes agree; DAV1D_HIP_TWIN_ONLY = the picture
     * lives in its twin and the raster planes are stale (what dav1d_hip_recon_list_run_tiled leaves: the reconstruction wrote whole
     * tiles — an 8x8 block is one 128-byte line — and nothing else).  Such a picture may be handed to motion compensation as a
     * reference, to dav1d_hip_host_picture_fetch / dav1d_hip_plane_download (they un-tile the rows they copy: raster rows by the
     * address rules of the reference's src/picture.c:46-63 exist at the output only), to dav1d_hip_recon_list_run_tiled again and to
     * dav1d_hip_picture_untile, which gives the raster planes back (twin_ok = 1) for everything else. */
    void *twin[3];
    void *twin_alloc;
    int twin_ok;
} Dav1dHipPicture;
#define DAV1D_HIP_TWIN_ONLY 2

/* Allocation with the reference's geometry (src/picture.c:46-78): dimensions padded
 * to 128, stride = aligned_w << hbd, +64 B when a multiple of 1024. */
/* Objects of this library alive right now: out[0] contexts, [1] frames (dav1d_hip_frame_begin .. _destroy), [2] listers, [3] host
 * pictures.  For callers that must prove they do not leak on their error paths (a frame context of dav1d that fails in pass 1 drops its
 * frame and lister without ever ending the frame: reference src/decode.c:3242-3251). */
DAV1D_HIP_API int dav1d_hip_live_objects(long long out[4]);
/* ---- several devices in ONE process.  dav1d is one process with n_fc frame contexts (reference src/internal.h:354-388); its binding
 * (dav1d_amd/host/dav1d_glue.c, option n_devices) ends the frames of frame context k on device k mod N and makes a reference resident
 * where it is read.  A context belongs to the device it was opened on.  HIP's current device is a property of the calling thread: a
 * thread that serves contexts of several devices calls dav1d_hip_context_use before it calls in with one of them (dav1d_hip_open,
 * dav1d_hip_frame_begin, dav1d_hip_frame_end and dav1d_hip_picture_copy_peer do so themselves).
 * dav1d_hip_picture_copy_peer: `dst` (dst_c's device; same geometry and strides: dav1d_hip_picture_alloc under the same ref_twin
 * option) becomes a copy of `src` (src_c's device) — the raster planes unless src lives in its twin only, the twin when src has a valid
 * one and dst the storage — enqueued on dst_c's stream behind what src_c's stream holds at the call, hipMemcpyPeerAsync (xGMI between
 * peers).  dav1d_hip_frame_begin / _end answer -EXDEV for a picture that lives on another device than their context. */
DAV1D_HIP_API int dav1d_hip_device_count(void);
DAV1D_HIP_API int dav1d_hip_context_device(const Dav1dHipContext *c);
DAV1D_HIP_API int dav1d_hip_context_use(Dav1dHipContext *c);
/* for code that borrows somebody else's thread (dav1d's picture allocator runs on the application's): the thread's current device
 * before dav1d_hip_context_use, and back to it afterwards */
DAV1D_HIP_API int dav1d_hip_current_device(void);
DAV1D_HIP_API int dav1d_hip_set_device(int device);
DAV1D_HIP_API int dav1d_hip_picture_device(const Dav1dHipPicture *pic);
/* direct (xGMI) copies between the devices of two contexts, both directions: returns how many directions are enabled now (0 - 2; a pair
 * that cannot be peers still copies, through host memory), < 0 on bad arguments.  The caller's current device is left as it was. */
DAV1D_HIP_API int dav1d_hip_enable_peer_access(Dav1dHipContext *a, Dav1dHipContext *b);
DAV1D_HIP_API int dav1d_hip_picture_copy_peer(Dav1dHipContext *dst_c, Dav1dHipPicture *dst, Dav1dHipContext *src_c, const Dav1dHipPicture *src);
/* The same for luma rows [y0, y1) of the RASTER planes only (and the chroma rows under them; y0 a multiple of 8), on dst_c's stream, WITHOUT waiting
 * for src_c's stream: the caller says these rows are final on the source device — dav1d_hip_frame_set_progress_callback has reported them.  This is
 * how a picture crosses band by band while the frame that makes it is still ending, the way dav1d's frame threads start on the rows progress[1]
 * has published (reference src/thread_task.c:416-433, 888-896).  dst's tiled twin is stale afterwards: dav1d_hip_picture_retile on dst_c once the
 * last band is across.  -EINVAL for a source that lives in its twin only.  Leaves the thread on dst_c's device. */
DAV1D_HIP_API int dav1d_hip_picture_copy_peer_rows(Dav1dHipContext *dst_c, Dav1dHipPicture *dst, Dav1dHipContext *src_c, const Dav1dHipPicture *src, int y0, int y1);
DAV1D_HIP_API int dav1d_hip_picture_alloc(Dav1dHipContext *c, Dav1dHipPicture *pic,
                                          int w, int h, int layout, int bpc);
DAV1D_HIP_API int dav1d_hip_picture_free(Dav1dHipContext *c, Dav1dHipPicture *pic);
/* The tiled twin of a picture (see Dav1dHipPicture.twin).  Context option "ref_twin" ($DAV1D_HIP_REF_TWIN): 0 = twins are never
 * read; 1 (default) = motion compensation reads references through valid twins, which the caller makes (dav1d_hip_picture_retile
 * once a picture is final); 2 = dav1d_hip_picture_alloc also makes the storage along with the planes and dav1d_hip_frame_end
 * retiles the picture it produces; 3 = as 2, and a frame that is reconstruction and nothing else (no intra wavefront, in-loop filters,
 * super-resolution, warped or scaled predictions) leaves its picture in the twin ONLY (twin_ok = DAV1D_HIP_TWIN_ONLY on the picture
 * dav1d_hip_frame_end hands back: see Dav1dHipPicture).  _twin_alloc adds the storage to a picture that has none (also to a caller-wrapped one: strides
 * must be multiples of 8 pixels; the twin has the rows of dav1d's allocator, 128-row padding included).  _retile copies the raster planes into the twin on the
 * context's stream (asynchronous, like every launch) and sets twin_ok. */
DAV1D_HIP_API int dav1d_hip_picture_twin_alloc(Dav1dHipContext *c, Dav1dHipPicture *pic);
DAV1D_HIP_API int dav1d_hip_picture_retile(Dav1dHipContext *c, Dav1dHipPicture *pic);
/* twin -> raster planes for a picture that lives in its twin (twin_ok == DAV1D_HIP_TWIN_ONLY; twin_ok = 1 afterwards); a no-op for any
 * other picture.  On the context's stream. */
DAV1D_HIP_API int dav1d_hip_picture_untile(Dav1dHipContext *c, Dav1dHipPicture *pic);
/* ... on a side stream of the context: starts when what is enqueued so far is through, runs next to what is enqueued afterwards
 * (the copy is bound by bandwidth, a frame's launches by request latency and arithmetic).  Launches of THIS context that read
 * twins, and dav1d_hip_sync, wait for it; other contexts must not read the twin before this context has synchronised. */
DAV1D_HIP_API int dav1d_hip_picture_retile_overlapped(Dav1dHipContext *c, Dav1dHipPicture *pic);
/* The buffers behind a Dav1dPicAllocator (reference include/dav1d/picture.h:89-133, default implementation
 * src/picture.c:46-82): a decoded picture the application reads on the host — pinned memory, planes and strides laid out by the
 * rules of dav1d_default_picture_alloc (dimensions rounded up to 128, 64 more bytes of stride where it would be a multiple of
 * 1024, 64 bytes of slack at the end) — paired with the device picture the frame is reconstructed into, which has the same
 * geometry.  alloc_picture_callback stores data[] / stride[] in the Dav1dPicture and the struct in allocator_data;
 * release_picture_callback hands it back (INTEGRATION.md shows both).  dav1d_hip_host_picture_fetch copies luma rows
 * [row0, row1) and the chroma rows under them from `src` (NULL: hp->dev; or the picture a frame reported as filtered) to the host
 * planes on the context's copy stream and returns; dav1d_hip_host_picture_wait waits for the copies issued so far. */
typedef struct Dav1dHipHostPicture {
    void *data[3];             /* pinned host planes */
    ptrdiff_t stride[2];       /* bytes: luma, chroma */
    Dav1dHipPicture dev;       /* device picture of the same geometry */
    void *alloc;               /* base of the host allocation */
    size_t alloc_size;
} Dav1dHipHostPicture;
DAV1D_HIP_API int dav1d_hip_host_picture_alloc(Dav1dHipContext *c, Dav1dHipHostPicture *hp, int w, int h, int layout, int bpc);
DAV1D_HIP_API int dav1d_hip_host_picture_release(Dav1dHipContext *c, Dav1dHipHostPicture *hp);
DAV1D_HIP_API int dav1d_hip_host_picture_fetch(Dav1dHipContext *c, const Dav1dHipHostPicture *hp, const Dav1dHipPicture *src, int row0, int row1);
DAV1D_HIP_API int dav1d_hip_host_picture_wait(Dav1dHipContext *c);
/* host <-> device plane copies; host_stride in bytes; copies the PADDED plane
 * (aligned dimensions) when `padded` is non-zero, else the visible w x h. */
DAV1D_HIP_API int dav1d_hip_plane_upload(Dav1dHipContext *c, const Dav1dHipPicture *pic, int plane,
                                         const void *host, ptrdiff_t host_stride, int padded);
DAV1D_HIP_API int dav1d_hip_plane_download(Dav1dHipContext *c, const Dav1dHipPicture *pic, int plane,
                                           void *host, ptrdiff_t host_stride, int padded);

/* ---------------------------------------------------------------------- itx */

/* One inverse-transform-and-add, replaces one call of
 *   dsp->itx.itxfm_add[tx][txtp](dst, stride, coeff, eob HIGHBD)      (reference
 * src/recon_tmpl.c:811-816, 1924-1970; kernel src/itx_tmpl.c:43-124,184-203). */
typedef struct Dav1dHipItxTask {
    uint32_t dst_off;  /* pixel offset of the block's top-left inside its plane: y*(stride/sizeof(pixel)) + x */
    uint32_t cf_off;   /* offset (in coefs) of the block's slab in the coefficient arena;
                          slab = min(w,32)*min(h,32) coefs, column-major coeff[y + x*min(h,32)]
                          (reference src/itx_tmpl.c:98-105); must be a multiple of 16 bytes
                          (PACKED tasks: offset of the first of eob + 1 values, any alignment) */
    int16_t  eob;      /* as passed to itxfm_add (>= 0) */
    uint8_t  tx;       /* enum RectTxfmSize, reference src/levels.h:44-78 (0..18) */
    uint8_t  txtp;     /* enum TxfmType index of the itxfm_add table, reference src/levels.h:80-100 (16 = WHT_WHT) */
    uint8_t  plane;    /* 0..2 */
    uint8_t  flags;    /* DAV1D_HIP_ITX_* */
    uint8_t  rsv[2];   /* written by the library in its device copy; ignored on input */
} Dav1dHipItxTask;

/* flags.  PACKED: the block's coefficients are not a dense slab but the eob + 1 values the entropy decoder produced, in
 * the order it produced them (scan position 0 .. eob: dav1d_scans[tx] for the 2-D transform classes, slab order for the H
 * classes, column-interleaved for the V classes, reference src/recon_tmpl.c:458-520, 548-575), starting at cf_off of the
 * arena; the arena is read-only for such tasks (nothing to re-zero).  This is the sparse wire format of SURVEY 8(f)#1:
 * it shrinks the per-frame host -> device coefficient traffic from the full cf arena to the coefficients that exist. */
enum { DAV1D_HIP_ITX_PACKED = 1 };

/* Runs `n` tasks (any order, any mix of sizes; dst rectangles must be disjoint).
 * `tasks` is a HOST array (the CPU lister produces it), `coef` a DEVICE pointer to
 * the coefficient arena; every consumed slab is zeroed exactly as the reference does
 * (src/itx_tmpl.c:60,108: the whole slab, or coeff[0] on the dc-only path).  The
 * call bins the tasks by tx size, uploads them, launches and waits. */
DAV1D_HIP_API int dav1d_hip_itx_add_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst,
                                          const Dav1dHipItxTask *tasks, size_t n, void *coef);

/* A pre-binned, device-resident list (what the frame driver and bench use):
 * sort + upload once, launch many times. */
typedef struct Dav1dHipItxList Dav1dHipItxList;
DAV1D_HIP_API int dav1d_hip_itx_list_create(Dav1dHipContext *c, Dav1dHipItxList **out,
                                            const Dav1dHipItxTask *host_tasks, size_t n);
DAV1D_HIP_API void dav1d_hip_itx_list_destroy(Dav1dHipContext *c, Dav1dHipItxList *l);
DAV1D_HIP_API int dav1d_hip_itx_list_run(Dav1dHipContext *c, const Dav1dHipItxList *l,
                                         const Dav1dHipPicture *dst, void *coef);

/* Measurement aid: the same launches, each bracketed by HIP events on the context's
 * stream; ms[19] / counts[19] receive per-tx-size kernel durations and task counts. */
DAV1D_HIP_API int dav1d_hip_itx_list_run_timed(Dav1dHipContext *c, const Dav1dHipItxList *l,
                                               const Dav1dHipPicture *dst, void *coef,
                                               float *ms, size_t *counts);

/* ----------------------------------------------------------------------- mc */

enum Dav1dHipMcKind {
    DAV1D_HIP_MC_PUT  = 0,  /* dsp->mc.mc[filter_2d]   -> pixels into dst plane   (reference src/mc_tmpl.c:129-187, 434-489) */
    DAV1D_HIP_MC_PREP = 1,  /* dsp->mc.mct[filter_2d]  -> int16 into the prep arena (reference src/mc_tmpl.c:246-305, 516-586) */
    DAV1D_HIP_MC_PUT_TMP = 2, /* dsp->mc.mc[filter_2d] -> pixels into the scratch arena (row stride w): the `lap` predictions of
                                 obmc() (reference src/recon_tmpl.c:1052-1112); the arena is the `prep` pointer viewed as pixels */
};

/* One motion-compensated prediction, replaces one call of mc() in the reference
 * driver (src/recon_tmpl.c:938-1050): dsp->mc.emu_edge when the window leaves the
 * reference plane (:967-978) followed by dsp->mc.mc / mct (:983-989).  Edge
 * emulation is folded into the fetch: source coordinates are clamped to
 * [0, ref_w-1] x [0, ref_h-1] (== emu_edge_c, reference src/mc_tmpl.c:868-916). */
typedef struct Dav1dHipMcTask {
    uint32_t dst_off;   /* PUT: pixel offset in dst plane; PREP: int16 offset in the prep arena (w*h contiguous, row stride w) */
    int32_t  src_x;     /* integer position of the block's top-left in the reference plane (may be out of range) */
    int32_t  src_y;
    uint8_t  w, h;      /* 2..128 */
    uint8_t  mx, my;    /* 0..15 sixteenth-pel phases as passed to mc_fn */
    uint8_t  filter_2d; /* enum Filter2d, reference src/levels.h:184-196 (9 = bilinear) */
    uint8_t  kind;      /* enum Dav1dHipMcKind */
    uint8_t  plane;     /* plane index (same in dst and ref) */
    uint8_t  ref;       /* index into the refs[] array passed to the batch call */
    uint32_t pad;
} Dav1dHipMcTask;

/* `tasks` is a HOST array; `refs` is a host array of n_refs (<= 8) picture
 * descriptors with device planes; `prep` is the DEVICE int16 arena PREP tasks write to.
 * The tasks of one call / list write disjoint rectangles and run in no particular order (the library reorders them by
 * where they read and by code path). */
DAV1D_HIP_API int dav1d_hip_mc_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst,
                                     const Dav1dHipPicture *refs, int n_refs,
                                     const Dav1dHipMcTask *tasks, size_t n, int16_t *prep);

/* Pre-tiled, device-resident list: blocks are cut into <= 64x16 tiles and binned by
 * tile shape once; run many times. */
typedef struct Dav1dHipMcList Dav1dHipMcList;
DAV1D_HIP_API int dav1d_hip_mc_list_create(Dav1dHipContext *c, Dav1dHipMcList **out,
                                           const Dav1dHipMcTask *host_tasks, size_t n);
DAV1D_HIP_API void dav1d_hip_mc_list_destroy(Dav1dHipContext *c, Dav1dHipMcList *l);
DAV1D_HIP_API int dav1d_hip_mc_list_run(Dav1dHipContext *c, const Dav1dHipMcList *l,
                                        const Dav1dHipPicture *dst, const Dav1dHipPicture *refs,
                                        int n_refs, int16_t *prep);

/* Measurement aid, see dav1d_hip_itx_list_run_timed: ms[15] / counts[15] per tile-shape bin
 * (bin = 3*class(w) + class(h); w classes 4, 8, 16, 32, 64; h classes 4, 8, 16). */
DAV1D_HIP_API int dav1d_hip_mc_list_run_timed(Dav1dHipContext *c, const Dav1dHipMcList *l,
                                              const Dav1dHipPicture *dst, const Dav1dHipPicture *refs,
                                              int n_refs, int16_t *prep, float *ms, size_t *counts);

/* Compound combination of two prepared predictions, replaces dsp->mc.avg / w_avg /
 * mask / w_mask (reference src/recon_tmpl.c:1802-1826; src/mc_tmpl.c:628-681,724-794). */
enum Dav1dHipCompKind {
    DAV1D_HIP_COMP_AVG = 0,
    DAV1D_HIP_COMP_WAVG = 1,   /* arg = weight (jnt_weight) */
    DAV1D_HIP_COMP_MASK = 2,   /* mask_off -> w*h bytes in the mask arena */
    DAV1D_HIP_COMP_WMASK = 3,  /* arg = sign; ss = 0:444 1:422 2:420; mask_off -> output mask */
    /* dsp->mc.blend / blend_v / blend_h (reference src/mc_tmpl.c:682-722): dst = (dst*(64-m) + tmp*m + 32) >> 6 with
     * tmp = w*h pixels at tmp1_off of the scratch arena (see DAV1D_HIP_MC_PUT_TMP); used by OBMC and inter-intra */
    DAV1D_HIP_COMP_BLEND = 4,   /* mask_off -> w*h mask bytes */
    DAV1D_HIP_COMP_BLEND_V = 5, /* obmc mask along x, first 3w/4 columns; run after every other kind of the same batch / list
                                   (the order obmc() needs where a block's blend_h and blend_v areas overlap) */
    DAV1D_HIP_COMP_BLEND_H = 6, /* obmc mask along y, first 3h/4 rows */
};
typedef struct Dav1dHipCompTask {
    uint32_t dst_off;   /* pixel offset in dst plane */
    uint32_t tmp1_off;  /* int16 offsets in the prep arena */
    uint32_t tmp2_off;
    uint32_t mask_off;  /* byte offset in the mask arena (MASK: input, WMASK: output) */
    uint8_t  w, h;
    uint8_t  kind;      /* enum Dav1dHipCompKind */
    uint8_t  plane;
    int8_t   arg;
    uint8_t  ss;
    uint16_t pad;
} Dav1dHipCompTask;

/* `tasks` HOST array; `prep` / `mask` DEVICE arenas. */
DAV1D_HIP_API int dav1d_hip_comp_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst,
                                       const Dav1dHipCompTask *tasks, size_t n,
                                       const int16_t *prep, uint8_t *mask);
typedef struct Dav1dHipCompList Dav1dHipCompList;
DAV1D_HIP_API int dav1d_hip_comp_list_create(Dav1dHipContext *c, Dav1dHipCompList **out,
                                             const Dav1dHipCompTask *host_tasks, size_t n);
DAV1D_HIP_API void dav1d_hip_comp_list_destroy(Dav1dHipContext *c, Dav1dHipCompList *l);
DAV1D_HIP_API int dav1d_hip_comp_list_run(Dav1dHipContext *c, const Dav1dHipCompList *l,
                                          const Dav1dHipPicture *dst, const int16_t *prep, uint8_t *mask);

/* All inter prediction of a frame or tile-sbrow in one list: the PUT / PREP tasks and the
 * compound tasks that consume the PREP outputs, in the form the reference driver issues
 * them (src/recon_tmpl.c:1784-1826).  The builder fuses every AVG / W_AVG task with the two
 * PREP tasks that feed it (when nothing else reads them): such blocks are predicted twice
 * and combined in registers and never touch the prep arena, which is scratch (its contents
 * after a run are unspecified).  MASK / W_MASK compounds keep the two-step form. */
typedef struct Dav1dHipInterList Dav1dHipInterList;
DAV1D_HIP_API int dav1d_hip_inter_list_create(Dav1dHipContext *c, Dav1dHipInterList **out,
                                              const Dav1dHipMcTask *host_mc, size_t n_mc,
                                              const Dav1dHipCompTask *host_comp, size_t n_comp);
DAV1D_HIP_API void dav1d_hip_inter_list_destroy(Dav1dHipContext *c, Dav1dHipInterList *l);

/* Predictions and residuals of one frame (or tile-sbrow set) as ONE list: what dav1d_hip_inter_list_run followed by
 * dav1d_hip_itx_list_run does, except that the residual launch of a transform size waits only for the prediction launches
 * whose tiles lie under its blocks (worked out at creation on a 4x4-cell map of `geometry`, a picture of the size and strides
 * the list will run on), so the memory-bound small predictions overlap with the arithmetic-bound large transforms.  The
 * counterpart of the reference's per-block order mc -> itxfm_add inside recon_b_inter (src/recon_tmpl.c:1557-2000). */
typedef struct Dav1dHipReconList Dav1dHipReconList;
DAV1D_HIP_API int dav1d_hip_recon_list_create(Dav1dHipContext *c, Dav1dHipReconList **out, const Dav1dHipPicture *geometry,
                                              const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                                              const Dav1dHipItxTask *itx, size_t n_itx);
DAV1D_HIP_API int dav1d_hip_recon_list_run(Dav1dHipContext *c, const Dav1dHipReconList *l, const Dav1dHipPicture *dst,
                                           const Dav1dHipPicture *refs, int n_refs, int16_t *prep, uint8_t *mask, void *coef);
/* The same for a frame whose in-loop filters are off and that later frames predict from: afterwards dst's tiled twin
 * (Dav1dHipPicture.twin; the storage is made when missing) holds the frame's pixels and dst->twin_ok is set.  With tiled references,
 * blocks on AV1's grid and no mask / blend tasks the launches write the twin along with the raster planes (no extra pass over the
 * picture); otherwise the list runs as dav1d_hip_recon_list_run does and dav1d_hip_picture_retile follows. */
DAV1D_HIP_API int dav1d_hip_recon_list_run_twin(Dav1dHipContext *c, const Dav1dHipReconList *l, Dav1dHipPicture *dst,
                                                const Dav1dHipPicture *refs, int n_refs, int16_t *prep, uint8_t *mask, void *coef);
/* The same with the picture living in its twin ONLY (the native layout of a reconstructed picture on this backend): nothing is
 * written to the raster planes, dst->twin_ok = DAV1D_HIP_TWIN_ONLY afterwards (see Dav1dHipPicture).  The paired launches and the
 * residual launches leave whole tile rows (an 8x8 block = one 128-byte line, where the raster planes take eight 16-byte pieces in
 * eight lines), the prediction launches store their strips into tiles, the residual launches read the predicted pixels back from
 * there.  Needs what the direct form of _run_twin needs (tiled references, blocks on AV1's grid, no mask / blend tasks); a list that
 * cannot run that way runs on the raster planes and is retiled (twin_ok = 1).  If `dst` holds pixels the list does not overwrite
 * they are carried along: a picture with twin_ok == 0 is retiled first. */
DAV1D_HIP_API int dav1d_hip_recon_list_run_tiled(Dav1dHipContext *c, const Dav1dHipReconList *l, Dav1dHipPicture *dst,
                                                 const Dav1dHipPicture *refs, int n_refs, int16_t *prep, uint8_t *mask, void *coef);
DAV1D_HIP_API void dav1d_hip_recon_list_destroy(Dav1dHipContext *c, Dav1dHipReconList *l);
/* Measurement aid (bench.py): every launch of the list on its own, bracketed by events.  ms / counts hold 40 entries:
 * [0..4] paired launches (prediction + residual in one wave) by square size 4x4 .. 64x64, [5..19] prediction launches by tile
 * shape (3 * width class + height class), [20] the compound / blend launch, [21..39] residual launches by transform size. */
DAV1D_HIP_API int dav1d_hip_recon_list_run_timed(Dav1dHipContext *c, const Dav1dHipReconList *l, const Dav1dHipPicture *dst,
                                                 const Dav1dHipPicture *refs, int n_refs, int16_t *prep, uint8_t *mask, void *coef,
                                                 float *ms, size_t *counts);
/* ... of dav1d_hip_recon_list_run_tiled (-ENOTSUP when the list cannot run with its picture in the twin only) */
DAV1D_HIP_API int dav1d_hip_recon_list_run_tiled_timed(Dav1dHipContext *c, const Dav1dHipReconList *l, Dav1dHipPicture *dst,
                                                       const Dav1dHipPicture *refs, int n_refs, int16_t *prep, uint8_t *mask, void *coef,
                                                       float *ms, size_t *counts);
DAV1D_HIP_API int dav1d_hip_inter_list_run(Dav1dHipContext *c, const Dav1dHipInterList *l,
                                           const Dav1dHipPicture *dst, const Dav1dHipPicture *refs,
                                           int n_refs, int16_t *prep, uint8_t *mask);
/* ms[16] / counts[16]: the 15 tile-shape bins of the mc kernel, then the residual compound kernel */
DAV1D_HIP_API int dav1d_hip_inter_list_run_timed(Dav1dHipContext *c, const Dav1dHipInterList *l,
                                                 const Dav1dHipPicture *dst, const Dav1dHipPicture *refs,
                                                 int n_refs, int16_t *prep, uint8_t *mask,
                                                 float *ms, size_t *counts);
DAV1D_HIP_API size_t dav1d_hip_inter_list_fused(const Dav1dHipInterList *l);

/* -------------------------------------------------------------------- ipred */

enum Dav1dHipIpredKind {
    DAV1D_HIP_IPRED_PRED = 0,  /* dav1d_prepare_intra_edges + dsp->ipred.intra_pred[m]  (src/recon_tmpl.c:1256-1283) */
    DAV1D_HIP_IPRED_CFL = 1,   /* prepare (DC_PRED) + dsp->ipred.cfl_ac[layout-1] + cfl_pred[m]  (src/recon_tmpl.c:1367-1393) */
    DAV1D_HIP_IPRED_PAL = 2,   /* dsp->ipred.pal_pred  (src/recon_tmpl.c:1207-1224, 1395-1413) */
    /* Table-level kinds, one DSP entry each with the caller having run dav1d_prepare_intra_edges (used by the
     * reference-signature table; `aux` is then a scratch arena holding the prepared edge array / the ac block): */
    DAV1D_HIP_IPRED_DSP = 3,          /* intra_pred[mode]: mode = table index (0..13), pal[0] = the `angle` argument with its flag
                                         bits, aux_off = pixel offset of `topleft` in aux (elements -(h+min(w,h)) .. w+min(w,h) are read) */
    DAV1D_HIP_IPRED_DSP_CFL_AC = 4,   /* cfl_ac[layout - 1]: luma at aux_off of plane 0, max_w / max_h = w_pad / h_pad, output tw*4 x th*4
                                         int16 (row stride = width) at int16 offset pal[1] | pal[2] << 16 of aux */
    DAV1D_HIP_IPRED_DSP_CFL_PRED = 5, /* cfl_pred[mode] (mode 0, 3, 4, 5): edge as DSP, ac as CFL_AC's output, angle = alpha */
    /* The intra half of an inter-intra block (src/recon_tmpl.c:1606-1630, 1751-1784): edges prepared from the block's place in
     * the picture like PRED, prediction written to the scratch (prep) arena as pixels, row stride = block width, at pixel
     * offset aux_off; a DAV1D_HIP_COMP_BLEND task of the same wavefront step blends it in.  Frame API only. */
    DAV1D_HIP_IPRED_PRED_TMP = 6,
    /* Intra block copy (recon_b_inter on key / intra-only frames, src/recon_tmpl.c:1583-1597: mc() from the frame's own reconstruction
     * with the bilinear filter): the tw x th block is copied from position ((int16) pal[0], (int16) pal[1]) of its own plane, pixels of
     * the source window outside the coded area (whole 8x8 luma blocks) replicated as mc()'s emu_edge does, with the sixteenth-pel phases
     * pal[2] & 15 (x) and pal[2] >> 8 (y) — 0, or 8 in a subsampled chroma plane under an odd vector.  A prediction like any other of
     * its wavefront step: what it reads was written by earlier steps.  pal[7] = the highest step among the blocks it reads (pal[6] =
     * 0x8000): the superblock route waits for the superblocks under the source window to have passed it.  Blocks of up to 64x64. */
    DAV1D_HIP_IPRED_COPY = 7,
};

/* One intra prediction of one transform block.  Field names follow the arguments of
 * dav1d_prepare_intra_edges (src/ipred_prepare_tmpl.c:75-88) and of the intra_pred entries
 * (src/ipred.h:44-47).  Tasks of one batch must be independent: their edge pixels are final. */
typedef struct Dav1dHipIpredTask {
    uint32_t dst_off;    /* pixel offset of the block in its plane */
    uint32_t aux_off;    /* CFL: pixel offset of the co-located luma block in plane 0; PAL: byte offset of the packed indices */
    uint16_t x4, y4;     /* `x`, `y`: block position in 4-pixel units of the plane */
    uint16_t w4, h4;     /* `w`, `h`: end of the tile in the same units */
    uint8_t  tw, th;     /* transform block size in 4-pixel units */
    uint8_t  mode;       /* enum IntraPredMode of the bitstream (0 DC .. 12 PAETH, src/levels.h:108-122), 13 = FILTER_PRED */
    int8_t   angle;      /* angle delta (directional), filter index (FILTER_PRED) or alpha (CFL) */
    uint8_t  flags;      /* 1 have_left, 2 have_top, 4 top has right, 8 left has bottom (edge_flags already selected for the
                            layout), 16 seq_hdr->intra_edge_filter, 32 ANGLE_SMOOTH_EDGE_FLAG */
    uint8_t  plane;
    uint8_t  kind;       /* enum Dav1dHipIpredKind */
    uint8_t  pad;
    uint16_t max_w, max_h; /* PRED: max_width / max_height arguments in pixels; CFL: w_pad / h_pad in 4-pixel units */
    uint16_t pal[8];     /* PAL: the palette.  PRED / CFL / PRED_TMP tasks of a frame (optional, the pass-2 lister sets it): pal[6] = 0x8000 |
                            mask of the neighbouring superblocks (bit 0 left, 1 top-left, 2 top, 3 top-right) in which the prediction reads
                            pixels written by INTRA blocks of this frame, pal[7] = the highest wavefront step among those blocks.  With it on
                            every prediction of a superblock, the superblock route waits per block for exactly that
                            instead of for whole neighbouring superblocks; 0 = not known */
} Dav1dHipIpredTask;

/* `tasks` HOST array; `aux` DEVICE byte arena: packed palette indices for PAL tasks, scratch for the table-level
 * kinds (may be NULL when neither is used). */
DAV1D_HIP_API int dav1d_hip_ipred_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipIpredTask *tasks, size_t n,
                                        uint8_t *aux);

/* Device-resident wavefront of intra batches: uploaded once, batch k = the next batch_sizes[k] tasks of `tasks`.
 * run_batch() only enqueues the kernel on the context's stream (no host synchronisation), so the residual lists
 * (dav1d_hip_itx_list_run) of every wave can be interleaved on the same stream without a host round trip per wave. */
typedef struct Dav1dHipIpredList Dav1dHipIpredList;
DAV1D_HIP_API int dav1d_hip_ipred_list_create(Dav1dHipContext *c, Dav1dHipIpredList **out, const Dav1dHipIpredTask *tasks,
                                              const size_t *batch_sizes, size_t n_batches);
DAV1D_HIP_API int dav1d_hip_ipred_list_run_batch(Dav1dHipContext *c, const Dav1dHipIpredList *l, size_t batch,
                                                 const Dav1dHipPicture *dst, uint8_t *aux);
DAV1D_HIP_API void dav1d_hip_ipred_list_destroy(Dav1dHipContext *c, Dav1dHipIpredList *l);

/* The wavefront of an intra frame with both halves of every block: batch k = the predictions AND the residuals of the blocks
 * of step k (pred_sizes[k] / tx_sizes[k] consecutive tasks of `preds` / `txs`).  What dav1d_hip_ipred_list_run_batch(k)
 * followed by dav1d_hip_itx_list_run does, except that a 4x4 or 8x8 block whose residual covers exactly its prediction is
 * predicted and reconstructed by one wave (the reference's order inside recon_b_intra, src/recon_tmpl.c:1207-1360): one
 * launch instead of two for the steps that hold only small blocks.  run_batch only enqueues; it can be recorded into a graph. */
typedef struct Dav1dHipIntraList Dav1dHipIntraList;
DAV1D_HIP_API int dav1d_hip_intra_list_create(Dav1dHipContext *c, Dav1dHipIntraList **out, const Dav1dHipIpredTask *preds, const size_t *pred_sizes,
                                              const Dav1dHipItxTask *txs, const size_t *tx_sizes, size_t n_batches);
DAV1D_HIP_API int dav1d_hip_intra_list_run_batch(Dav1dHipContext *c, const Dav1dHipIntraList *l, size_t batch, const Dav1dHipPicture *dst,
                                                 void *coef, uint8_t *aux);
/* all batches of the list in order, enqueued back to back (no synchronisation in between) */
DAV1D_HIP_API int dav1d_hip_intra_list_run_all(Dav1dHipContext *c, const Dav1dHipIntraList *l, const Dav1dHipPicture *dst, void *coef,
                                               uint8_t *aux);
DAV1D_HIP_API void dav1d_hip_intra_list_destroy(Dav1dHipContext *c, Dav1dHipIntraList *l);

/* The same wavefront as ONE launch: the batches become a list of units (prediction + residual of one transform block)
 * sorted by step; the waves of the launch draw units in that order and a unit starts once every unit of the earlier steps
 * has finished (a counter in device memory), so a step boundary is a hand-off between running waves instead of a kernel
 * boundary.  Same results as the batch-by-batch routes.  create: -ENOTSUP when a task kind is not handled here (PRED_TMP of
 * inter-intra blocks, the DSP-level kinds) — use dav1d_hip_intra_list_* then.  run only enqueues. */
typedef struct Dav1dHipIntraFlow Dav1dHipIntraFlow;
DAV1D_HIP_API int dav1d_hip_intra_flow_create(Dav1dHipContext *c, Dav1dHipIntraFlow **out, const Dav1dHipIpredTask *preds, const size_t *pred_sizes,
                                              const Dav1dHipItxTask *txs, const size_t *tx_sizes, size_t n_batches);
DAV1D_HIP_API int dav1d_hip_intra_flow_run(Dav1dHipContext *c, const Dav1dHipIntraFlow *l, const Dav1dHipPicture *dst, void *coef, uint8_t *aux);
DAV1D_HIP_API void dav1d_hip_intra_flow_destroy(Dav1dHipContext *c, Dav1dHipIntraFlow *l);
DAV1D_HIP_API size_t dav1d_hip_intra_flow_units(const Dav1dHipIntraFlow *l);
/* after a run (synchronizes): out = { tickets drawn, units finished, waves that gave up waiting (0 unless something is broken) } */
DAV1D_HIP_API int dav1d_hip_intra_flow_status(Dav1dHipContext *c, const Dav1dHipIntraFlow *l, uint32_t out[3]);

/* The same wavefront SUPERBLOCK BY SUPERBLOCK: a workgroup owns a superblock and works its units off step by step behind
 * workgroup barriers; superblocks are ordered in levels — a superblock's level is one more than the highest among its left,
 * top-left, top and top-right neighbours of the same tile that hold intra units (all the reference's decode order lets a
 * block read: src/decode.c:2117-2375, src/ipred_prepare_tmpl.c:82-116).  The levels run as ONE launch in which a superblock
 * waits for the neighbours it reads (option intra_sb_flow, default), or as a launch per level.  The batches must be the
 * steps of a wavefront in which a block's step exceeds the step of every block it reads INSIDE its superblock (the listers'
 * and any decode-order-consistent numbering do).  geometry: a picture of the frame's size / layout / strides; sb128 and the
 * tile starts as in Dav1dHipFrameDesc.  -ENOTSUP for the task kinds dav1d_hip_intra_flow_create refuses.  run only enqueues. */
typedef struct Dav1dHipIntraSb Dav1dHipIntraSb;
DAV1D_HIP_API int dav1d_hip_intra_sb_create(Dav1dHipContext *c, Dav1dHipIntraSb **out, const Dav1dHipIpredTask *preds, const size_t *pred_sizes,
                                            const Dav1dHipItxTask *txs, const size_t *tx_sizes, size_t n_batches,
                                            const Dav1dHipPicture *geometry, int sb128, int n_tile_cols, const uint16_t *col_start_sb,
                                            int n_tile_rows, const uint16_t *row_start_sb);
DAV1D_HIP_API int dav1d_hip_intra_sb_run(Dav1dHipContext *c, const Dav1dHipIntraSb *l, const Dav1dHipPicture *dst, void *coef, uint8_t *aux);
/* The one-launch form (option intra_sb_flow) is a dataflow inside ONE grid: a workgroup spins on the flags of workgroups with LOWER
 * indices.  That makes forward progress only because (a) the hardware dispatches the workgroups of a grid in index order and (b) a
 * workgroup that has been dispatched keeps its compute unit until it ends (no preemption of a running wave by this process's own
 * launches) — true of gfx950 with the ROCm 7 firmware, NOT promised by the HIP programming model.  Should either stop holding, a
 * waiting workgroup gives up after about 0.8 s (SB_SPIN_LIMIT in csrc/intra_sb.hip), marks itself so that its dependants give up
 * too, and counts itself: dav1d_hip_frame_end then fails the frame with -EIO, and after dav1d_hip_intra_sb_run the caller asks
 * dav1d_hip_intra_sb_status (synchronizes): 0, or -EIO with *gave_up = superblocks left unreconstructed.  The launch-per-level form
 * (intra_sb_flow = 0) relies on neither assumption. */
DAV1D_HIP_API int dav1d_hip_intra_sb_status(Dav1dHipContext *c, const Dav1dHipIntraSb *l, uint32_t *gave_up);
DAV1D_HIP_API void dav1d_hip_intra_sb_destroy(Dav1dHipContext *c, Dav1dHipIntraSb *l);
DAV1D_HIP_API size_t dav1d_hip_intra_sb_levels(const Dav1dHipIntraSb *l);          /* levels of superblocks (launches per run with intra_sb_flow = 0) */
DAV1D_HIP_API size_t dav1d_hip_intra_sb_superblocks(const Dav1dHipIntraSb *l);     /* superblocks that hold units */

/* ------------------------------------------------- mc: warp, scaled, resize, emu_edge */

/* One 8x8 block of a warped prediction: dsp->mc.warp8x8 (kind PUT, pixels into dst) or warp8x8t (kind PREP, int16
 * into the prep arena with row stride tmp_stride) as issued by warp_affine() (reference src/recon_tmpl.c:1115-1174;
 * kernels src/mc_tmpl.c:799-866).  src_x / src_y = (dx, dy) of that driver, the integer position of the block's
 * top-left in the reference plane; the 15x15 window around it is fetched with clamped coordinates (== the driver's
 * emu_edge call at :1150-1160).  mx / my are the driver's values (already masked with ~0x3f). */
typedef struct Dav1dHipWarpTask {
    uint32_t dst_off;     /* PUT: pixel offset in dst plane; PREP: int16 offset in the prep arena */
    int32_t  src_x, src_y;
    int32_t  mx, my;
    int16_t  abcd[4];
    uint16_t tmp_stride;  /* PREP: row stride of the int16 output in elements */
    uint8_t  kind;        /* DAV1D_HIP_MC_PUT or DAV1D_HIP_MC_PREP */
    uint8_t  plane;
    uint8_t  ref;
    uint8_t  pad[3];
} Dav1dHipWarpTask;
DAV1D_HIP_API int dav1d_hip_warp_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *refs,
                                       int n_refs, const Dav1dHipWarpTask *tasks, size_t n, int16_t *prep);

/* One prediction from a reference of different size: dsp->mc.mc_scaled[filter_2d] (PUT) / mct_scaled[filter_2d]
 * (PREP) as issued by the scaled branch of mc() (reference src/recon_tmpl.c:990-1047; kernels src/mc_tmpl.c:189-244,
 * 307-357, 491-626).  src_x / src_y = (left, top) of that driver: the integer position in the reference plane that
 * the kernels' `src` pointer addresses; mx / my = (pos_x & 0x3ff, pos_y & 0x3ff); dx / dy = the 1/1024 pel steps
 * (svc[refidx][0].step, [1].step).  The window is fetched with clamped coordinates (== emu_edge, :1020-1032). */
typedef struct Dav1dHipMcScaledTask {
    uint32_t dst_off;     /* PUT: pixel offset in dst plane; PREP: int16 offset in the prep arena (row stride w) */
    int32_t  src_x, src_y;
    int16_t  mx, my;      /* 0..1023 */
    int16_t  dx, dy;      /* step in 1/1024 pel: 512..2048 in valid AV1 */
    uint8_t  w, h;        /* 2..128 */
    uint8_t  filter_2d;   /* enum Filter2d (9 = bilinear) */
    uint8_t  kind;        /* DAV1D_HIP_MC_PUT, DAV1D_HIP_MC_PREP or DAV1D_HIP_MC_PUT_TMP */
    uint8_t  plane;
    uint8_t  ref;
    uint8_t  pad[2];
} Dav1dHipMcScaledTask;
DAV1D_HIP_API int dav1d_hip_mc_scaled_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *refs,
                                            int n_refs, const Dav1dHipMcScaledTask *tasks, size_t n, int16_t *prep);

/* Horizontal super-resolution of one plane, replaces the per-row loop over dsp->mc.resize in dav1d_resize()
 * (reference src/recon_tmpl.c:2024-2049 -> src/mc_tmpl.c:918-944): dst rows y0..y0+h-1, dst_w pixels each, from the same
 * rows of src (src_w pixels each); dx = f->resize_step[ss_hor], mx0 = f->resize_start[ss_hor]. */
DAV1D_HIP_API int dav1d_hip_resize(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src,
                                   int plane, int dst_w, int y0, int h, int src_w, int dx, int mx0);

/* Stand-alone dsp->mc.emu_edge (reference src/mc_tmpl.c:868-916), same argument meaning: a bw x bh block whose
 * top-left sits at (x, y) of an iw x ih plane starting at `ref`, edge pixels replicated; strides in bytes, both
 * pointers DEVICE memory.  (The batched mc / warp / scaled paths never need it: they clamp coordinates.) */
DAV1D_HIP_API int dav1d_hip_emu_edge(Dav1dHipContext *c, int bpc, intptr_t bw, intptr_t bh, intptr_t iw, intptr_t ih,
                                     intptr_t x, intptr_t y, void *dst, ptrdiff_t dst_stride,
                                     const void *ref, ptrdiff_t ref_stride);

/* --------------------------------------------------------------------- cdef */

/* One 8x8 luma unit of CDEF as the reference driver handles it
 * (dav1d_cdef_brow, src/cdef_apply_tmpl.c:149-290): direction search on the luma block
 * (dsp->cdef.dir), variance-adjusted primary strength, dsp->cdef.fb[0] on luma and
 * dsp->cdef.fb[uv_idx] on both chroma blocks.  Out of place: `src` is the immutable
 * pre-CDEF (deblocked) picture, `dst` receives the filtered units; `dst` must already
 * hold a copy of `src` for the units that are not listed (skipped / zero strength). */
enum { DAV1D_HIP_CDEF_RAW = 1, DAV1D_HIP_CDEF_W4 = 2, DAV1D_HIP_CDEF_H4 = 4, DAV1D_HIP_CDEF_BOT_REP_Y = 8, DAV1D_HIP_CDEF_BOT_REP_UV = 16 };   /* Dav1dHipCdefTask.flags */
enum { DAV1D_HIP_CDEF_HAVE_LEFT = 1, DAV1D_HIP_CDEF_HAVE_RIGHT = 2, DAV1D_HIP_CDEF_HAVE_TOP = 4,
       DAV1D_HIP_CDEF_HAVE_BOTTOM = 8 };    /* == enum CdefEdgeFlags, src/cdef.h:36-41 */
typedef struct Dav1dHipCdefTask {
    uint16_t bx, by;     /* unit position in 8x8 luma units (RAW tasks: top-left in pixels of `plane`) */
    uint8_t  y_pri;      /* (y_strength >> 2) << (bpc - 8), before the variance adjustment */
    uint8_t  y_sec;      /* secondary strength incl. the 3 -> 4 fix-up, << (bpc - 8) */
    uint8_t  uv_pri, uv_sec;
    uint8_t  edges;      /* DAV1D_HIP_CDEF_HAVE_* */
    uint8_t  flags;      /* bit 0 RAW: one dsp->cdef.fb call (pri = y_pri, sec = y_sec, `dir`, no search / adjust) on
                            `plane`; bit 1: block is 4 wide, bit 2: block is 4 high (fb[1] = 4x8, fb[2] = 4x4);
                            bit 3 (DAV1D_HIP_CDEF_BOT_REP_Y) / bit 4 (_UV): of the two rows below the unit, the SECOND is a copy
                            of the first, in luma / in chroma.  With frame threading the last unit row of a superblock row's band
                            takes its two bottom rows from the deblocked lines backup_lpf() saved (src/cdef_apply_tmpl.c:222-232),
                            and where the picture's last row is the first of them, backup_lpf stores it twice
                            (`n_lines = 4 - (row + stripe_h + 1 == h)`, src/lf_apply_tmpl.c:77-97) */
    uint8_t  dir, plane; /* RAW only */
    uint8_t  pad[4];
} Dav1dHipCdefTask;

/* `tasks` HOST array.  `damping` = frame cdef.damping + (bpc - 8) (chroma uses damping - 1).
 * `dirvar`: optional DEVICE array of n words receiving dir | var << 3 of every non-RAW task. */
DAV1D_HIP_API int dav1d_hip_cdef_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src,
                                       const Dav1dHipCdefTask *tasks, size_t n, int damping, uint32_t *dirvar);

/* --------------------------------------------------------------- loop filter */

/* One call of dsp->lf.loop_filter_sb[plane != 0][dir] as the sbrow drivers issue it
 * (src/lf_apply_tmpl.c:176-311; kernels src/loopfilter_tmpl.c:37-257): a line of up to 32
 * edge units (4 pixels each) in one superblock column (dir 0: edges between columns, units
 * run down) or row (dir 1: edges between rows, units run across). */
typedef struct Dav1dHipLfTask {
    uint32_t dst_off;    /* pixel offset of the first unit's first pixel on the q side of the edge */
    uint32_t lvl_off;    /* index of the first unit's entry in the level array (uint8_t[4] per 4x4, src/internal.h:297) */
    uint32_t vmask[3];   /* unit u is filtered when bit u is set in any word; width 16 if set in [2] (luma), else 8 (chroma: 6) if set in [1], else 4 */
    uint8_t  plane;      /* 0..2 */
    uint8_t  dir;        /* 0: loop_filter_h_* (vertical edge), 1: loop_filter_v_* (horizontal edge) */
    uint8_t  lvl_comp;   /* which of the 4 level bytes: 0 Y cols, 1 Y rows, 2 U, 3 V */
    uint8_t  pad;
} Dav1dHipLfTask;

/* `tasks` HOST array (any order: all dir-0 tasks run, then all dir-1 tasks, which reproduces the
 * reference's column-then-row order); `lvl` DEVICE level array with row stride `b4_stride` entries;
 * lut_e / lut_i: the 64-entry E / I limit tables of Av1FilterLUT (src/lf_mask.h:36-40).  In place. */
DAV1D_HIP_API int dav1d_hip_lf_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipLfTask *tasks, size_t n,
                                     const uint8_t *lvl, ptrdiff_t b4_stride, const uint8_t lut_e[64], const uint8_t lut_i[64]);

/* --------------------------------------------------------- loop restoration */

enum { DAV1D_HIP_LR_HAVE_LEFT = 1, DAV1D_HIP_LR_HAVE_RIGHT = 2, DAV1D_HIP_LR_HAVE_TOP = 4, DAV1D_HIP_LR_HAVE_BOTTOM = 8 };
                                                                     /* == enum LrEdgeFlags, src/looprestoration.h:36-41 */
enum Dav1dHipLrType { DAV1D_HIP_LR_WIENER7 = 0, DAV1D_HIP_LR_WIENER5 = 1,      /* dsp->lr.wiener[0 / 1] */
                      DAV1D_HIP_LR_SGR_5X5 = 2, DAV1D_HIP_LR_SGR_3X3 = 3, DAV1D_HIP_LR_SGR_MIX = 4 };   /* dsp->lr.sgr[0 / 1 / 2] */

/* One call of a looprestorationfilter_fn on one restoration-unit stripe as lr_stripe() issues it
 * (src/lr_apply_tmpl.c:36-97): w <= 384, h <= 64.  Out of place: `src` = loop-restoration input
 * (CDEF output), `lpf` = deblocked (pre-CDEF) picture supplying the two rows above / below the
 * stripe (what the reference saves into lr_lpf_line, src/lf_apply_tmpl.c:41-102), `dst` = output. */
/* (A row "below the stripe" that lies beyond the plane's last row is that last row once more: backup_lpf() stores the picture's last
 * row twice where it is the first of the two, src/lf_apply_tmpl.c:77-97.) */
typedef struct Dav1dHipLrTask {
    uint16_t x, y;       /* stripe position in pixels of `plane` */
    uint16_t w, h;
    uint8_t  plane;
    uint8_t  edges;      /* DAV1D_HIP_LR_HAVE_* */
    uint8_t  type;       /* enum Dav1dHipLrType */
    uint8_t  pad;
    int16_t  filter[2][8]; /* Wiener: LooprestorationParams.filter as built by lr_stripe (:55-71): [0] horizontal, [1] vertical;
                              SGR: filter[0][0..3] = params.sgr.{s0, s1, w0, w1} (:72-82) */
} Dav1dHipLrTask;

/* `tasks` HOST array. */
DAV1D_HIP_API int dav1d_hip_lr_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src,
                                     const Dav1dHipPicture *lpf, const Dav1dHipLrTask *tasks, size_t n);

/* --------------------------------------------------------------- film grain */

/* Same members, order and types as Dav1dFilmGrainData (include/dav1d/headers.h:315-333), so a
 * pointer to frame_hdr->film_grain.data can be passed as is. */
typedef struct Dav1dHipFilmGrainData {
    unsigned seed;
    int num_y_points;
    uint8_t y_points[14][2];
    int chroma_scaling_from_luma;
    int num_uv_points[2];
    uint8_t uv_points[2][10][2];
    int scaling_shift;
    int ar_coeff_lag;
    int8_t ar_coeffs_y[24];
    int8_t ar_coeffs_uv[2][25 + 3];
    uint64_t ar_coeff_shift;
    int grain_scale_shift;
    int uv_mult[2];
    int uv_luma_mult[2];
    int uv_offset[2];
    int overlap_flag;
    int clip_to_restricted_range;
} Dav1dHipFilmGrainData;

/* dav1d_apply_grain (src/fg_apply_tmpl.c:97-241; called from src/lib.c:485-524) on the device:
 * grain templates (fg.generate_grain_y / _uv), scaling LUTs (generate_scaling), then
 * fg.fgy_32x32xn / fg.fguv_32x32xn over every 32x32 block of `src` into `dst`; planes without grain
 * are copied.  `is_id` = seq_hdr->mtrx == DAV1D_MC_IDENTITY. */
DAV1D_HIP_API int dav1d_hip_fg_apply(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src,
                                     const Dav1dHipFilmGrainData *data, int is_id);

/* The same in the two halves the reference has (dav1d_prep_grain / dav1d_apply_grain_row, src/fg_apply_tmpl.c:97-241): the grain
 * templates and scaling tables depend on the frame header only, so dav1d_hip_fg_prepare can be called as soon as the header is
 * parsed — it enqueues their generation on a side stream and returns — and dav1d_hip_fg_apply_prepared, at the end of the
 * frame, only waits for that event.  The ~0.23 ms the lone-wave template kernels take then hide behind the reconstruction. */
typedef struct Dav1dHipGrain Dav1dHipGrain;
DAV1D_HIP_API int dav1d_hip_fg_prepare(Dav1dHipContext *c, Dav1dHipGrain **out, const Dav1dHipFilmGrainData *data, int bpc, int layout);
DAV1D_HIP_API int dav1d_hip_fg_apply_prepared(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src,
                                              const Dav1dHipGrain *g, int is_id);
DAV1D_HIP_API void dav1d_hip_fg_grain_destroy(Dav1dHipContext *c, Dav1dHipGrain *g);
/* The grain templates alone (parity aid): host_lut receives grain_lut[3][73 + 1][82] as int16_t. */
DAV1D_HIP_API int dav1d_hip_fg_generate_grain(Dav1dHipContext *c, const Dav1dHipFilmGrainData *data, int bpc, int layout,
                                              int16_t *host_lut);

/* ------------------------------------------------------------ one frame in flight */

/* Driver-level boundary (reference: f->bd_fn.recon_b_intra / recon_b_inter / filter_sbrow_* called from the pass-2
 * workers, src/internal.h:247-262, src/thread_task.c:733-851).  A lister that walks f->frame_thread.b[] like decode_b()'s
 * pass-2 branch appends flat tasks per tile-sbrow / superblock row, from any worker thread; dav1d_hip_frame_end() runs
 * the frame in the reference's stage order and returns when every stage has completed (src/thread_task.c:888-896 then
 * publishes progress).  Intra blocks, warps and scaled references go through their own batch calls between
 * frame_begin and frame_end (they need the caller's wavefront order). */
typedef struct Dav1dHipFrame Dav1dHipFrame;
DAV1D_HIP_API int dav1d_hip_frame_begin(Dav1dHipContext *c, Dav1dHipFrame **out, const Dav1dHipPicture *cur,
                                        const Dav1dHipPicture *refs, int n_refs);
/* The references once more, any time before dav1d_hip_frame_end: same geometry as at dav1d_hip_frame_begin, other memory.  For
 * frame threading: frame n + 1 is listed while frame n is still in flight, and the picture that holds frame n's final pixels
 * (*filtered of its dav1d_hip_frame_end: `cur` or a picture that frame owns) is known only when frame n ends. */
DAV1D_HIP_API int dav1d_hip_frame_set_refs(Dav1dHipFrame *f, const Dav1dHipPicture *refs, int n_refs);
DAV1D_HIP_API int dav1d_hip_frame_submit_tile_sbrow(Dav1dHipFrame *f, const Dav1dHipMcTask *mc, size_t n_mc,
                                                    const Dav1dHipCompTask *comp, size_t n_comp,
                                                    const Dav1dHipItxTask *itx, size_t n_itx);
/* The values DAV1D_HIP_ITX_PACKED residual tasks of this frame refer to, a share at a time (a tile-sbrow's): n values (int32 at
 * more than 8 bits per component, int16 at 8) are copied into the frame's own coefficient arena; *base = the offset, in values, to
 * add to the cf_off of the tasks that index `vals` from 0.  Such a frame is ended with coef = NULL (a frame holds packed tasks or
 * dense ones, not both: -EINVAL at dav1d_hip_frame_end); the arena goes to the device with the prepared lists
 * (dav1d_hip_frame_flush, dav1d_hip_frame_end).  Thread-safe. */
DAV1D_HIP_API int dav1d_hip_frame_submit_coefs(Dav1dHipFrame *f, const void *vals, size_t n, uint32_t *base);
/* bytes of that arena in use so far (what will cross the host link for the frame's residuals) */
DAV1D_HIP_API size_t dav1d_hip_frame_coef_bytes(const Dav1dHipFrame *f);
/* Intra blocks of wavefront step `step` (their neighbours are final after the inter blocks of the frame and the intra blocks of
 * the steps before): predictions + residuals; `aux` = DEVICE arena of packed palette indices (NULL if no PAL task).  Thread-safe. */
DAV1D_HIP_API int dav1d_hip_frame_submit_intra_step(Dav1dHipFrame *f, size_t step, const Dav1dHipIpredTask *ipred, size_t n_ipred,
                                                    const Dav1dHipItxTask *itx, size_t n_itx, uint8_t *aux);
/* Every intra step of one submitter (a tile-sbrow) in ONE call: the tasks sorted by step, *_end[s] = end offset of step s in its
 * array, steps 0 .. n_steps - 1 (step 0 stays empty: it belongs to the inter blocks); blends = the step's inter-intra blends
 * (dav1d_hip_frame_submit_step_blend).  What the pass-2 lister calls: one lock per tile-sbrow instead of one per step. */
DAV1D_HIP_API int dav1d_hip_frame_submit_intra_sorted(Dav1dHipFrame *f, size_t n_steps, const Dav1dHipIpredTask *ipred, const size_t *ipred_end,
                                                      const Dav1dHipItxTask *itx, const size_t *itx_end, const Dav1dHipCompTask *blend,
                                                      const size_t *blend_end);
/* The frame's tiles in superblocks (seq_hdr->sb128, frame_hdr->tiling.cols / col_start_sb / rows / row_start_sb, as in
 * Dav1dHipFrameDesc).  With them the frame's intra blocks — the intra halves of inter-intra blocks and their blends included — run
 * superblock by superblock (dav1d_hip_intra_sb_*) instead of step by step, provided no submission carries intra block copies.  Before the first intra submission (-EINVAL
 * after it); dav1d_hip_lister_create calls it. */
DAV1D_HIP_API int dav1d_hip_frame_set_tiling(Dav1dHipFrame *f, int sb128, int n_tile_cols, const uint16_t *col_start_sb, int n_tile_rows,
                                             const uint16_t *row_start_sb);
/* Inter-intra blends of wavefront step `step` >= 1 (kind DAV1D_HIP_COMP_BLEND reading what the step's DAV1D_HIP_IPRED_PRED_TMP
 * tasks wrote): run between the step's predictions and its residuals.  Thread-safe. */
DAV1D_HIP_API int dav1d_hip_frame_submit_step_blend(Dav1dHipFrame *f, size_t step, const Dav1dHipCompTask *blend, size_t n);
/* Intra block copy (recon_b_inter on key / intra-only frames, reference src/recon_tmpl.c:1583-1597): predictions that read the
 * frame's OWN reconstruction.  tasks[i] is an ordinary prediction record (kind DAV1D_HIP_MC_PUT, filter_2d 9 = bilinear, `ref`
 * ignored: the source is the picture the frame is reconstructed into, bounded by its size rounded up to whole 8x8 blocks) that
 * runs in wavefront step steps[i] >= 1 — after every block of the earlier steps is final, before the step's residuals.  Frames that
 * carry such tasks run their wavefront as launches per step (not as the dataflow launch).  Thread-safe. */
DAV1D_HIP_API int dav1d_hip_frame_submit_step_copy(Dav1dHipFrame *f, const Dav1dHipMcTask *tasks, const uint16_t *steps, size_t n);
/* Warped predictions / predictions from references of another size of any tile-sbrow; run before the compound combinations. */
DAV1D_HIP_API int dav1d_hip_frame_submit_warp(Dav1dHipFrame *f, const Dav1dHipWarpTask *t, size_t n);
DAV1D_HIP_API int dav1d_hip_frame_submit_scaled(Dav1dHipFrame *f, const Dav1dHipMcScaledTask *t, size_t n);
/* Optional: once every dav1d_hip_frame_submit_tile_sbrow issued so far has returned (all tiles listed), start moving the prepared
 * lists to the device now rather than at dav1d_hip_frame_end; the transfer then runs under whatever the caller does next.  Must
 * not race with a submission in progress.  dav1d_hip_lister_run does this by itself when its threads have joined. */
DAV1D_HIP_API int dav1d_hip_frame_flush(Dav1dHipFrame *f);
/* In-loop filter tasks of a superblock row (or any other share of the frame; any order, any thread).  The tasks are checked
 * here (-EINVAL), and what can be prepared per share is — deblocking tasks ordered vertical edges first, CDEF units of a row that
 * sit side by side grouped for the strip kernel — so keep the units of one row of 8x8 blocks in one call, in raster order. */
DAV1D_HIP_API int dav1d_hip_frame_submit_filter_sbrow(Dav1dHipFrame *f, const Dav1dHipLfTask *lf, size_t n_lf,
                                                      const Dav1dHipCdefTask *cdef, size_t n_cdef,
                                                      const Dav1dHipLrTask *lr, size_t n_lr);
/* Super-resolution (reference src/recon_tmpl.c:2053-2086 dav1d_filter_sbrow_resize, src/lf_apply_tmpl.c:40-100 backup_lpf): the frame
 * is reconstructed, deblocked and CDEF-filtered at its coded width; between CDEF and restoration every row is upscaled to sr_w
 * pixels by the 8-tap resampler (mc.resize; step and first position from the two widths, AV1 spec 7.16), and so are the deblocked
 * rows restoration reads at its stripe borders.  The restoration tasks of such a frame are in upscaled coordinates
 * (Dav1dHipFilterDesc.sr_w); *filtered and grain_out of dav1d_hip_frame_end are sr_w wide.  Call before dav1d_hip_frame_end. */
DAV1D_HIP_API int dav1d_hip_frame_set_super_res(Dav1dHipFrame *f, int sr_w);
/* lvl: DEVICE level array; lut_e / lut_i: Av1FilterLUT tables; cdef_damping = frame damping + bpc - 8; grain may be NULL */
DAV1D_HIP_API int dav1d_hip_frame_set_filters(Dav1dHipFrame *f, const uint8_t *lvl, ptrdiff_t b4_stride, const uint8_t lut_e[64],
                                              const uint8_t lut_i[64], int cdef_damping, const Dav1dHipFilmGrainData *grain, int is_id);
/* coef / prep / mask: DEVICE arenas.  `cur` ends up reconstructed and deblocked; *filtered describes the picture after
 * CDEF + restoration (owned by the frame unless it is `cur`); grain_out (optional) receives the film grain output. */
DAV1D_HIP_API int dav1d_hip_frame_end(Dav1dHipFrame *f, void *coef, int16_t *prep, uint8_t *mask, Dav1dHipPicture *filtered,
                                      const Dav1dHipPicture *grain_out);
/* How the post filters of the last dav1d_hip_frame_end ran: the number of superblock-row bands deblocking, CDEF and
 * restoration were pipelined over (three streams, CDEF one band behind deblocking, restoration one behind CDEF — the
 * reference's filter_sbrow_* ordering, src/thread_task.c:783-851), or 0 when they ran stage by stage (small frames, units
 * not covering the frame, or DAV1D_HIP_POST_BANDS unset: the banded mode is an opt-in experiment, see frame.hip). */
DAV1D_HIP_API int dav1d_hip_frame_post_bands(const Dav1dHipFrame *f);
/* The same without blocking the caller: the frame runs on a thread of the library; `done` (optional) is called on that thread
 * when every row of the frame is final — where a dav1d build stores f->sr_cur.progress[1] and signals the task threads
 * (reference src/thread_task.c:888-896; :393-439 is the waiting side).  dav1d_hip_frame_progress: rows that are final (see
 * dav1d_hip_frame_set_progress_callback for their granularity); dav1d_hip_frame_wait joins and returns the result. */
DAV1D_HIP_API int dav1d_hip_frame_end_async(Dav1dHipFrame *f, void *coef, int16_t *prep, uint8_t *mask, const Dav1dHipPicture *grain_out,
                                            void (*done)(void *cookie, int rc, const Dav1dHipPicture *filtered), void *cookie);
DAV1D_HIP_API int dav1d_hip_frame_progress(const Dav1dHipFrame *f);
DAV1D_HIP_API int dav1d_hip_frame_wait(Dav1dHipFrame *f, Dav1dHipPicture *filtered);
/* Row-granular progress, the reference's per-superblock-row publication (src/thread_task.c:888-896: after a row's last filter,
 * `atomic_store(&f->sr_cur.progress[1], y)` and a broadcast): `progress` is called on the thread that ends the frame whenever
 * more rows of the filtered picture have become final — `rows` luma rows from the top, living in `pic` (the picture
 * dav1d_hip_frame_end reports as *filtered) — and once more with the picture height when the frame is through.  Rows arrive in
 * steps of a band when the in-loop filters run banded (option post_bands >= 3: bands of whole 256-row superblock-row pairs, each
 * followed through deblocking, CDEF and restoration; the band's event is waited for before the call).  On the default schedule (one
 * stage after the other over the whole frame) a listener makes the LAST stage — loop restoration — report from inside its launches:
 * its tasks are ordered by band of 256 luma rows, the workgroups count their band's completions and the last one of a band writes to
 * pinned host memory the ending thread polls; rows arrive in steps of 256 (17 publications for a 4320-row frame) while the later bands
 * are still being restored, at no measurable cost to the frame.  A frame without restoration tasks publishes once.
 * Set before dav1d_hip_frame_end / _end_async.  The callback runs with the frame's and the context's locks held: it may copy the
 * rows out (dav1d_hip_host_picture_fetch, dav1d_hip_download) and signal other threads, but it must not call any
 * dav1d_hip_frame_* entry point of this frame (submit, flush, wait, destroy) nor start a list run / frame end on the same
 * context — those would wait for the locks the caller of the callback holds. */
DAV1D_HIP_API int dav1d_hip_frame_set_progress_callback(Dav1dHipFrame *f, void (*progress)(void *cookie, int rows, const Dav1dHipPicture *pic),
                                                        void *cookie);
DAV1D_HIP_API void dav1d_hip_frame_destroy(Dav1dHipFrame *f);

/* ------------------------------------------------------------- pass-2 lister */

/* The hand-off of dav1d's frame threading (reference src/internal.h:276-293): pass 1 (entropy decoding) leaves, per frame,
 * an Av1Block per coded block, the eob / transform type of every transform block (cbi), the dequantised coefficients (cf)
 * and the palettes; pass 2 walks them in decode order and reconstructs (decode_b()'s pass-2 branch, src/decode.c:706-806,
 * calling f->bd_fn.recon_b_intra / recon_b_inter, src/recon_tmpl.c:1176-1985).  The lister below IS that walk: instead of
 * calling the DSP table it appends the flat task records above to a Dav1dHipFrame.  Host C (dav1d_amd/host/lister.c). */

/* == Av1Block, reference src/levels.h:262-287 (32 bytes; tests/test_lister.py pins size and member offsets against the
 * reference build).  Stored at the block's top-left 4x4: f->frame_thread.b[by * b4_stride + bx]. */
typedef struct Dav1dHipAv1Block {
    uint8_t bl, bs, bp;
    uint8_t intra, seg_id, skip_mode, skip, uvtx;
    union {
        struct {
            uint8_t y_mode, uv_mode, tx, pal_sz[2];
            int8_t y_angle, uv_angle, cfl_alpha[2];
        } i;                                           /* intra */
        struct {
            union {
                struct { int16_t mv[2][2]; /* [ref][y, x] */ uint8_t wedge_idx, mask_sign, interintra_mode; } m;
                struct { int16_t mv2d[2]; int16_t matrix[4]; } w;      /* MM_WARP blocks */
            } u;
            uint8_t comp_type, inter_mode, motion_mode, drl_idx;
            int8_t ref[2];
            uint8_t max_ytx, filter2d, interintra_type, tx_split0;
            uint16_t tx_split1;
        } p;                                           /* inter */
    } u;
} Dav1dHipAv1Block;

/* == Dav1dWarpedMotionParams, reference include/dav1d/headers.h:97-106 */
typedef struct Dav1dHipWarpParams {
    int type;                    /* enum Dav1dWarpedMotionType: 0 identity, 1 translation, 2 rot-zoom, 3 affine */
    int32_t matrix[6];
    union { struct { int16_t alpha, beta, gamma, delta; } p; int16_t abcd[4]; } u;
} Dav1dHipWarpParams;

/* Everything of a Dav1dFrameContext the walk reads, as plain values and HOST pointers (nothing is copied: the arrays must
 * stay alive until the last lister call of the frame).  Member comments name the reference field. */
typedef struct Dav1dHipFrameDesc {
    int w, h;                    /* f->cur.p.w / h: the coded size (super-resolution upscales after CDEF: dav1d_hip_frame_set_super_res) */
    int layout, bpc;             /* f->cur.p.layout / bpc */
    int sb128;                   /* seq_hdr->sb128 */
    int intra_edge_filter;       /* seq_hdr->intra_edge_filter */
    int is_inter;                /* IS_INTER_OR_SWITCH(frame_hdr): 0 on key / intra-only frames (their inter-coded blocks are intra block copies) */
    int n_tile_cols, n_tile_rows;           /* frame_hdr->tiling.cols / rows */
    uint16_t col_start_sb[65], row_start_sb[65];    /* frame_hdr->tiling.col_start_sb / row_start_sb */
    ptrdiff_t b4_stride;         /* f->b4_stride */
    const Dav1dHipAv1Block *b;   /* f->frame_thread.b */
    const int16_t *cbi;          /* f->frame_thread.cbi: eob << 5 | txtp per transform block */
    const unsigned *tile_start_off;         /* f->frame_thread.tile_start_off[tile]: start of the tile's share of cbi / cf / pal_idx */
    const void *pal;             /* f->frame_thread.pal: pixel[3][8] per 8x8 (NULL without screen content tools) */
    int32_t svc[7][2][2];        /* f->svc[ref][x / y]{ scale, step }: scale 0 = reference has the frame's size.  Not looked at (like ref_w / ref_h /
                                    gmv_warp_allowed) when !is_inter: a frame context keeps these from its last inter frame */
    int ref_w[7], ref_h[7];      /* f->refp[i].p.p.w / h */
    Dav1dHipWarpParams gmv[7];   /* frame_hdr->gmv */
    uint8_t gmv_warp_allowed[7]; /* f->gmv_warp_allowed */
    uint8_t jnt_weights[7][7];   /* f->jnt_weights */
    int cf_align64;              /* 1 when the producer is an x86-64 build of dav1d: decode_sb() realigns the coefficient cursor to
                                    64 bytes after every 8x8 split into 4x4s (#if ARCH_X86_64, src/decode.c:2209-2218) */
    uint8_t lossless[8];         /* frame_hdr->segmentation.lossless[seg_id]: the deblocking masks of an inter block of such a
                                    segment are built with 4x4 transforms whatever b->max_ytx / uvtx say (src/decode.c:1890-1893;
                                    a skipped block keeps the block's largest sizes in its record, :456-470) */
    void *cf;                    /* f->frame_thread.cf (HOST, writable) or NULL.  NULL: the residual tasks point into a dense copy of
                                    the arena that the caller sends to the device itself (dav1d_hip_frame_end's `coef`).  Otherwise
                                    the lister PACKS: per transform block it gathers the eob + 1 values the entropy decoder
                                    produced (scan order, see DAV1D_HIP_ITX_PACKED) into the frame's own coefficient arena
                                    (dav1d_hip_frame_submit_coefs) and zeroes them where they were — what the reference's
                                    itxfm_add does to the slab it consumed (src/itx_tmpl.c:60,108): only the values that exist
                                    cross the host link.  The values of a tile-sbrow move when the row is handed in — for frames
                                    of few tiles on threads of the library (option prep_async), so cf has to stay as it is, and
                                    is only ready for the next frame's pass 1, when dav1d_hip_frame_end (or _flush / _destroy)
                                    of this frame has returned.  dav1d_hip_frame_end is then called with coef = NULL. */
} Dav1dHipFrameDesc;

/* Byte offset of a tile's first coefficient in the cf arena, of its first cbi entry and of its first palette index byte, as
 * setup_tile() derives them from tile_start_off (reference src/decode.c:2438-2452). */
typedef struct Dav1dHipLister Dav1dHipLister;

/* Scratch layout of a listed frame: the prep arena holds, in int16 units, the PREP blocks of compound predictions and (as
 * pixels) the OBMC / inter-intra intermediates; the mask arena starts with the constant wedge / inter-intra masks
 * (dav1d_hip_lister_const_masks: upload them once per arena) followed by the frame's segmentation masks.  Sizes are known
 * once every tile-sbrow is listed. */
DAV1D_HIP_API int dav1d_hip_lister_create(Dav1dHipLister **out, const Dav1dHipFrameDesc *desc, Dav1dHipFrame *frame);
/* One tile-sbrow (what a DAV1D_TASK_TYPE_TILE_RECONSTRUCTION task runs, src/thread_task.c:733-752 -> dav1d_decode_tile_sbrow
 * with pass 2).  Thread-safe across tiles; the superblock rows of one tile must be listed top to bottom. */
DAV1D_HIP_API int dav1d_hip_lister_tile_sbrow(Dav1dHipLister *l, int tile_row, int tile_col, int sby);
/* All tiles of the frame on n_threads threads of the library (tiles in raster order, each walked top to bottom) — for callers
 * without a thread pool of their own; dav1d's task threads call dav1d_hip_lister_tile_sbrow instead. */
DAV1D_HIP_API int dav1d_hip_lister_run(Dav1dHipLister *l, int n_threads);
DAV1D_HIP_API size_t dav1d_hip_lister_prep_elems(const Dav1dHipLister *l);   /* int16 elements of the prep arena used so far */
DAV1D_HIP_API size_t dav1d_hip_lister_mask_bytes(const Dav1dHipLister *l);   /* bytes of the mask arena used so far (constant part included) */
DAV1D_HIP_API size_t dav1d_hip_lister_steps(const Dav1dHipLister *l);        /* wavefront steps the frame needs so far */
DAV1D_HIP_API const uint8_t *dav1d_hip_lister_const_masks(size_t *bytes);    /* HOST blob to copy to the start of the mask arena */
DAV1D_HIP_API void dav1d_hip_lister_destroy(Dav1dHipLister *l);
/* ---- the in-loop filters of a listed frame ----
 * What pass 1 leaves for the filters (reference src/lf_mask.h:42-63): per 128x128 the deblocking edge masks, the CDEF index of
 * its four 64x64s and the "has coefficients" mask of its 8x8s; per 128x128 and plane four restoration units; the level cache
 * f->lf.level goes to the device as is (dav1d_hip_frame_set_filters).  dav1d_hip_lister_filter_sbrow restates the drivers that
 * turn those into DSP calls — dav1d_loopfilter_sbrow_cols / _rows incl. the mask fix-ups at tile edges (src/lf_apply_tmpl.c:
 * 313-466), dav1d_cdef_brow (src/cdef_apply_tmpl.c:97-308), dav1d_lr_sbrow (src/lr_apply_tmpl.c:36-202) — as task records. */
typedef struct Dav1dHipRestorationUnit {       /* == Av1RestorationUnit */
    uint8_t type;                /* 0 none, 2 Wiener, 3 + sgr_idx self-guided */
    int8_t filter_h[3], filter_v[3], sgr_weights[2];
} Dav1dHipRestorationUnit;
typedef struct Dav1dHipAv1Filter {             /* == Av1Filter */
    uint16_t filter_y[2][32][3][2];
    uint16_t filter_uv[2][32][2][2];
    int8_t cdef_idx[4];
    uint16_t noskip_mask[16][2];
} Dav1dHipAv1Filter;
typedef struct Dav1dHipAv1Restoration { Dav1dHipRestorationUnit lr[3][4]; } Dav1dHipAv1Restoration;    /* == Av1Restoration */

typedef struct Dav1dHipFilterDesc {
    int lf_level_y[2], lf_level_u, lf_level_v;     /* frame_hdr->loopfilter.level_y / level_u / level_v */
    const Dav1dHipAv1Filter *lf_mask;              /* f->lf.mask, [sb128 row * sb128w + sb128 column] */
    const uint8_t *tx_lpf_right_edge[2];           /* f->lf.tx_lpf_right_edge */
    const uint8_t *a_tx_lpf_y, *a_tx_lpf_uv;       /* f->a[0].tx_lpf_y / .tx_lpf_uv of the pass-1 above contexts ... */
    size_t a_stride;                               /* ... sizeof(BlockContext) apart, [tile row * sb128w + sb128 column] */
    int cdef_enabled;                              /* seq_hdr->cdef */
    int cdef_damping;                              /* frame_hdr->cdef.damping */
    uint8_t cdef_y_strength[8], cdef_uv_strength[8];   /* frame_hdr->cdef.y_strength / uv_strength */
    int lr_type[3];                                /* frame_hdr->restoration.type: 0 = plane not restored */
    int lr_unit_size[2];                           /* frame_hdr->restoration.unit_size (log2), luma / chroma */
    const Dav1dHipAv1Restoration *lr_mask;         /* f->lf.lr_mask */
    int sr_w;                                      /* super-resolution: frame_hdr->width[1], the width restoration works at and
                                                      lr_mask is laid out for (f->sr_sb128w); 0 or the coded width: none */
} Dav1dHipFilterDesc;
/* One superblock row of filter tasks (what dav1d_filter_sbrow would execute, src/recon_tmpl.c:2100-2109), submitted to the
 * lister's frame.  Thread-safe; any order. */
DAV1D_HIP_API int dav1d_hip_lister_filter_sbrow(Dav1dHipLister *l, const Dav1dHipFilterDesc *fd, int sby);
/* every superblock row of the frame on n_threads threads of the library (see dav1d_hip_lister_run) */
DAV1D_HIP_API int dav1d_hip_lister_filter_run(Dav1dHipLister *l, const Dav1dHipFilterDesc *fd, int n_threads);
/* dav1d_hip_lister_run and dav1d_hip_lister_filter_run as one job of n_threads threads: the tiles first, the filter lists of the
 * superblock rows behind them in the same queue (they fill the time threads would otherwise wait for the slowest tile) */
DAV1D_HIP_API int dav1d_hip_lister_run_frame(Dav1dHipLister *l, const Dav1dHipFilterDesc *fd, int n_threads);

/* ---- deblocking masks and levels from the hand-off arrays (what pass 1 builds with dav1d_create_lf_mask_intra / _inter,
 * reference src/lf_mask.c:259-383, src/decode.c:1216-1226, 1882-1900, 1945-1956, 2730-2740), built on the device.
 * dav1d_hip_lf_rects (host walk, host/lf_rects.c): one rectangle per transform block — a whole skipped inter block counts as
 * one — in 4x4 units of its plane.  dav1d_hip_lf_masks_build (csrc/lfmask.hip): the rectangles painted into a cell map, the
 * masks, noskip_mask, level cache and tile-edge contexts read off it; outputs have the reference's layouts, so they go
 * straight into Dav1dHipFilterDesc / dav1d_hip_frame_set_filters. */
enum { DAV1D_HIP_LF_RECT_LUMA = 0, DAV1D_HIP_LF_RECT_CHROMA = 1, DAV1D_HIP_LF_RECT_NOSKIP = 2 };
typedef struct Dav1dHipLfRect {
    uint16_t x4, y4;     /* position, 4x4 units of the rectangle's plane (NOSKIP: luma units) */
    uint8_t  w4, h4;     /* size in the same units, clipped to the frame (NOSKIP: the unclipped block) */
    uint8_t  cls;        /* bits 0-1: min(2, log2(width / 4)) (chroma: min(1, .)), bits 2-3: the same for the height,
                            bit 4: the left side is an edge the filter visits, bit 5: the top side is */
    uint8_t  kind;       /* DAV1D_HIP_LF_RECT_* */
    uint8_t  lvl[2];     /* level cache entries [0], [1] (luma) / [2], [3] (chroma) of the covered cells */
    uint8_t  pad[2];
} Dav1dHipLfRect;
/* lflvl = ts->lflvl (== f->lf.lvl without delta_lf): [segment][0 y-vert, 1 y-hor, 2 u, 3 v][reference + 1][mode is not GLOBALMV].
 * *out is malloc'ed; release it with dav1d_hip_lf_rects_free. */
DAV1D_HIP_API int dav1d_hip_lf_rects(const Dav1dHipFrameDesc *d, const uint8_t lflvl[8][4][8][2], Dav1dHipLfRect **out, size_t *n);
/* The same with delta_lf (frame_hdr->delta.lf.present): pass 1 recomputes the level table whenever a superblock brings new deltas
 * (dav1d_calc_lf_values into ts->lflvlmem, reference src/decode.c:1180-1206), so a block's levels come from the table its
 * superblock was parsed with.  sb_lflvl[sb row * sbw + sb column] = that table (sbw = superblock columns of the frame; the glue
 * keeps a copy of ts->lflvlmem per superblock); NULL = dav1d_hip_lf_rects. */
DAV1D_HIP_API int dav1d_hip_lf_rects_sb(const Dav1dHipFrameDesc *d, const uint8_t lflvl[8][4][8][2], const uint8_t (*sb_lflvl)[8][4][8][2],
                                        Dav1dHipLfRect **out, size_t *n);
DAV1D_HIP_API void dav1d_hip_lf_rects_free(Dav1dHipLfRect *p);
/* masks_out: HOST, one Av1Filter per 128x128 (filter_y, filter_uv, noskip_mask written; cdef_idx zeroed: it comes from the
 * bitstream).  level_dev: DEVICE level cache (uint8_t[4] per 4x4, pitch d->b4_stride).  right_edge[0 luma, 1 chroma]: HOST,
 * == f->lf.tx_lpf_right_edge.  a_y / a_uv: HOST, 32 bytes per (tile row, sb128 column) == the tx_lpf_y / tx_lpf_uv members of
 * the pass-1 above contexts (Dav1dHipFilterDesc.a_stride = 32). */
DAV1D_HIP_API int dav1d_hip_lf_masks_build(Dav1dHipContext *c, const Dav1dHipFrameDesc *d, const Dav1dHipLfRect *rects, size_t n_rects,
                                           Dav1dHipAv1Filter *masks_out, uint8_t *level_dev, uint8_t *right_edge[2], uint8_t *a_y, uint8_t *a_uv);

/* ---- the data-parallel members of Dav1dRefmvsDSPContext (reference src/refmvs.c:763-803 save_tmvs, :914-923 splat_mv) on a
 * frame-level map of refmvs_block records in DEVICE memory: r[y4 * stride4 + x4], 12 bytes each ({ mv[2] (y, x int16 each),
 * int8 ref[2], bs, mf }), instead of the reference's ring of rows. */
typedef struct Dav1dHipSplatTask {
    uint16_t bx4, by4;       /* block position, 4x4 units */
    uint8_t  bw4, bh4;       /* size, 4x4 units (clipped to the frame by the caller, as decode_b does) */
    uint8_t  pad[2];
    uint32_t rmv[3];         /* the refmvs_block to splat, as its 12 bytes */
} Dav1dHipSplatTask;
/* Both calls enqueue on the context's stream and return (no host wait, no allocation once the context's pools are warm): results are
 * there for whatever runs on that stream next, or after dav1d_hip_sync. */
DAV1D_HIP_API int dav1d_hip_refmvs_splat_batch(Dav1dHipContext *c, void *r_dev, ptrdiff_t stride4, const Dav1dHipSplatTask *tasks, size_t n);
/* rp_dev: refmvs_temporal_block records (5 bytes: mv, ref), rp_stride per 8x8 row; ref_sign and the rectangle as the reference's
 * save_tmvs arguments (8x8 units). */
DAV1D_HIP_API int dav1d_hip_refmvs_save_tmvs(Dav1dHipContext *c, void *rp_dev, ptrdiff_t rp_stride, const void *r_dev, ptrdiff_t stride4,
                                             const uint8_t ref_sign[7], int col_start8, int col_end8, int row_start8, int row_end8);

/* Test aids (tests/test_host_tables.py pins the lister's derived AV1 geometry against the tables of the reference build). */
DAV1D_HIP_API long dav1d_hip_lister_mask_offset(int which, int c, int bs, int sign, int idx);
DAV1D_HIP_API void dav1d_hip_lister_tables(uint8_t *out);
DAV1D_HIP_API int dav1d_hip_lister_block_warp(Dav1dHipWarpParams *wm, const int16_t *matrix, const int16_t *mv2d, int bw4, int bh4, int bx4, int by4);

/* ------------------------------------------------- multi-GPU (RCCL over xGMI), one process per GPU
 *
 * The data-path collectives of the two shardings (one Dav1dFrameContext per frame thread on the reference side, src/internal.h:
 * 247-262, src/lib.c:140-301 n_fc): frames in flight one per GPU, where a frame that predicts from a frame decoded on another GPU
 * needs that picture (the publication rule of src/thread_task.c:416-433 at picture granularity) -> _broadcast_picture; tile columns one
 * per GPU (tiles are independent for reconstruction, src/recon_tmpl.c:1264-1268, not for inter prediction and the in-loop
 * filters, src/lf_apply_tmpl.c:330-398) -> _allgather_columns once per frame, _exchange_halo before the filters.  Rendezvous: rank
 * 0 makes a 128-byte id (dav1d_hip_peer_unique_id), the host hands it to every process by its own means (a file, a socket, MPI,
 * torch.distributed ...), every process opens its peer with the same id.  Everything is enqueued on the context's stream. */
typedef struct Dav1dHipPeer Dav1dHipPeer;
DAV1D_HIP_API int dav1d_hip_peer_unique_id(uint8_t id[128]);
DAV1D_HIP_API int dav1d_hip_peer_open(Dav1dHipContext *c, Dav1dHipPeer **out, const uint8_t id[128], int rank, int world);
DAV1D_HIP_API void dav1d_hip_peer_close(Dav1dHipPeer *p);
DAV1D_HIP_API int dav1d_hip_peer_rank(const Dav1dHipPeer *p);
DAV1D_HIP_API int dav1d_hip_peer_world(const Dav1dHipPeer *p);
/* every rank ends up with `owner`'s pixels of `pic` (same geometry everywhere; one ncclBroadcast for a picture of
 * dav1d_hip_picture_alloc) */
DAV1D_HIP_API int dav1d_hip_peer_broadcast_picture(Dav1dHipPeer *p, Dav1dHipPicture *pic, int owner);
/* rank g reconstructed luma columns [x0[g], x1[g]) (even; chroma follows the layout): afterwards every rank holds all of them.  One
 * strided pack kernel, ONE ncclAllGather of the strips (padded to the widest), scatter kernels straight into the planes. */
DAV1D_HIP_API int dav1d_hip_peer_allgather_columns(Dav1dHipPeer *p, Dav1dHipPicture *pic, const int *x0, const int *x1);
/* the `halo` luma columns on either side of this rank's column from the neighbours that reconstructed them (in-loop filters across
 * the tile edge; 16 covers deblocking + CDEF + restoration): neighbour to neighbour, an ncclSend / ncclRecv pair per side in one group */
DAV1D_HIP_API int dav1d_hip_peer_exchange_halo(Dav1dHipPeer *p, Dav1dHipPicture *pic, const int *x0, const int *x1, int halo);
/* dav1d_hip_peer_allgather_columns on the peer's side stream: it starts when what the context's stream holds so far is through (the frame
 * that wrote the columns) and runs next to what the caller enqueues afterwards — the next frame's reconstruction, which predicts from
 * other pictures.  dav1d_hip_peer_wait(p, lag): the context's stream waits for the asynchronous gathers issued so far but the `lag` (0 .. 7) most
 * recent ones — 0 before the first launch that reads `pic`; k for a caller that recycles a picture k + 1 frames later.
 * The asynchronous gathers stage through buffers of their own: a halo exchange or synchronous gather enqueued on the context's stream
 * while one is in flight does not touch its data.  The PICTURE is the caller's to keep apart: nothing on the context's stream may
 * write `pic` before dav1d_hip_peer_wait has covered the gather.
 * Column checks of all three calls (-EINVAL): even, one after the other, inside the plane, at least `halo` wide. */
DAV1D_HIP_API int dav1d_hip_peer_allgather_columns_async(Dav1dHipPeer *p, Dav1dHipPicture *pic, const int *x0, const int *x1);
DAV1D_HIP_API int dav1d_hip_peer_wait(Dav1dHipPeer *p, int lag);

/* ------------------------------------------------- reference-signature table */

/* The kernel-level drop-in: function pointer types with the reference's exact signatures and a table whose
 * layout is the reference's Dav1dDSPContext (src/internal.h:62-70: fg, ipred, mc, itx, lf, cdef, lr; members in the
 * order of src/filmgrain.h:74-80, src/ipred.h:81-90, src/mc.h:146-162, src/itx.h:70-72, src/loopfilter.h:45-53,
 * src/cdef.h:64-67, src/looprestoration.h:72-75), so `&c->dsp[bits]` of the reference can be handed to
 * dav1d_hip_dsp_init_* as is.  The 16 bpc flavour carries the trailing bitdepth_max (HIGHBD_DECL_SUFFIX), the 8 bpc
 * one does not; pixel = uint8_t / uint16_t, coef = int16_t / int32_t, grain entry = int8_t / int16_t
 * (include/common/bitdepth.h:44-63, src/filmgrain.h:36-44).  All pointers are HOST pointers: each call stages its
 * rectangles on the device, runs the same kernels the batched API runs, and copies the result back. */
#define DAV1D_HIP_ALIGN16 __attribute__((aligned(16)))
typedef struct Dav1dHipFilterLUT {      /* == Av1FilterLUT, reference src/lf_mask.h:36-40 */
    DAV1D_HIP_ALIGN16 uint8_t e[64];
    DAV1D_HIP_ALIGN16 uint8_t i[64];
    DAV1D_HIP_ALIGN16 uint64_t sharp[2];
} Dav1dHipFilterLUT;
typedef union Dav1dHipLrParams {        /* == LooprestorationParams, reference src/looprestoration.h:49-55 */
    DAV1D_HIP_ALIGN16 int16_t filter[2][8];
    struct { uint32_t s0, s1; int16_t w0, w1; } sgr;
} Dav1dHipLrParams;

#define DAV1D_HIP_N_RECT_TX_SIZES 19
#define DAV1D_HIP_N_TX_TYPES_PLUS_LL 17
#define DAV1D_HIP_N_2D_FILTERS 10
#define DAV1D_HIP_N_IMPL_INTRA_PRED_MODES 14
#define DAV1D_HIP_GRAIN_WIDTH 82

#define DAV1D_HIP_BD_NONE
#define DAV1D_HIP_BD_MAX , int bitdepth_max
#define DAV1D_HIP_DSP_TABLE(B, pixel, coef, entry, BD) \
typedef void (*dav1d_hip_itxfm_fn##B)(pixel *dst, ptrdiff_t stride, coef *coeff, int eob BD); \
typedef void (*dav1d_hip_mc_fn##B)(pixel *dst, ptrdiff_t dst_stride, const pixel *src, ptrdiff_t src_stride, int w, int h, int mx, int my BD); \
typedef void (*dav1d_hip_mc_scaled_fn##B)(pixel *dst, ptrdiff_t dst_stride, const pixel *src, ptrdiff_t src_stride, int w, int h, \
                                          int mx, int my, int dx, int dy BD); \
typedef void (*dav1d_hip_mct_fn##B)(int16_t *tmp, const pixel *src, ptrdiff_t src_stride, int w, int h, int mx, int my BD); \
typedef void (*dav1d_hip_mct_scaled_fn##B)(int16_t *tmp, const pixel *src, ptrdiff_t src_stride, int w, int h, int mx, int my, \
                                           int dx, int dy BD); \
typedef void (*dav1d_hip_avg_fn##B)(pixel *dst, ptrdiff_t dst_stride, const int16_t *tmp1, const int16_t *tmp2, int w, int h BD); \
typedef void (*dav1d_hip_w_avg_fn##B)(pixel *dst, ptrdiff_t dst_stride, const int16_t *tmp1, const int16_t *tmp2, int w, int h, int weight BD); \
typedef void (*dav1d_hip_mask_fn##B)(pixel *dst, ptrdiff_t dst_stride, const int16_t *tmp1, const int16_t *tmp2, int w, int h, \
                                     const uint8_t *mask BD); \
typedef void (*dav1d_hip_w_mask_fn##B)(pixel *dst, ptrdiff_t dst_stride, const int16_t *tmp1, const int16_t *tmp2, int w, int h, \
                                       uint8_t *mask, int sign BD); \
typedef void (*dav1d_hip_blend_fn##B)(pixel *dst, ptrdiff_t dst_stride, const pixel *tmp, int w, int h, const uint8_t *mask); \
typedef void (*dav1d_hip_blend_dir_fn##B)(pixel *dst, ptrdiff_t dst_stride, const pixel *tmp, int w, int h); \
typedef void (*dav1d_hip_warp8x8_fn##B)(pixel *dst, ptrdiff_t dst_stride, const pixel *src, ptrdiff_t src_stride, const int16_t *abcd, \
                                        int mx, int my BD); \
typedef void (*dav1d_hip_warp8x8t_fn##B)(int16_t *tmp, ptrdiff_t tmp_stride, const pixel *src, ptrdiff_t src_stride, const int16_t *abcd, \
                                         int mx, int my BD); \
typedef void (*dav1d_hip_emu_edge_fn##B)(intptr_t bw, intptr_t bh, intptr_t iw, intptr_t ih, intptr_t x, intptr_t y, \
                                         pixel *dst, ptrdiff_t dst_stride, const pixel *src, ptrdiff_t src_stride); \
typedef void (*dav1d_hip_resize_fn##B)(pixel *dst, ptrdiff_t dst_stride, const pixel *src, ptrdiff_t src_stride, int dst_w, int h, \
                                       int src_w, int dx, int mx BD); \
typedef void (*dav1d_hip_angular_ipred_fn##B)(pixel *dst, ptrdiff_t stride, const pixel *topleft, int width, int height, int angle, \
                                              int max_width, int max_height BD); \
typedef void (*dav1d_hip_cfl_ac_fn##B)(int16_t *ac, const pixel *y, ptrdiff_t stride, int w_pad, int h_pad, int cw, int ch); \
typedef void (*dav1d_hip_cfl_pred_fn##B)(pixel *dst, ptrdiff_t stride, const pixel *topleft, int width, int height, const int16_t *ac, \
                                         int alpha BD); \
typedef void (*dav1d_hip_pal_pred_fn##B)(pixel *dst, ptrdiff_t stride, const pixel *pal, const uint8_t *idx, int w, int h); \
typedef void (*dav1d_hip_loopfilter_sb_fn##B)(pixel *dst, ptrdiff_t stride, const uint32_t *mask, const uint8_t (*lvl)[4], \
                                              ptrdiff_t lvl_stride, const Dav1dHipFilterLUT *lut, int w BD); \
typedef int (*dav1d_hip_cdef_dir_fn##B)(const pixel *dst, ptrdiff_t dst_stride, unsigned *var BD); \
typedef void (*dav1d_hip_cdef_fn##B)(pixel *dst, ptrdiff_t stride, const pixel (*left)[2], const pixel *top, const pixel *bottom, \
                                     int pri_strength, int sec_strength, int dir, int damping, int edges BD); \
typedef void (*dav1d_hip_lr_fn##B)(pixel *dst, ptrdiff_t dst_stride, const pixel (*left)[4], const pixel *lpf, int w, int h, \
                                   const Dav1dHipLrParams *params, int edges BD); \
typedef void (*dav1d_hip_generate_grain_y_fn##B)(entry buf[][DAV1D_HIP_GRAIN_WIDTH], const Dav1dHipFilmGrainData *data BD); \
typedef void (*dav1d_hip_generate_grain_uv_fn##B)(entry buf[][DAV1D_HIP_GRAIN_WIDTH], const entry buf_y[][DAV1D_HIP_GRAIN_WIDTH], \
                                                  const Dav1dHipFilmGrainData *data, intptr_t uv BD); \
typedef void (*dav1d_hip_fgy_32x32xn_fn##B)(pixel *dst_row, const pixel *src_row, ptrdiff_t stride, const Dav1dHipFilmGrainData *data, \
                                            size_t pw, const uint8_t *scaling, const entry grain_lut[][DAV1D_HIP_GRAIN_WIDTH], \
                                            int bh, int row_num BD); \
typedef void (*dav1d_hip_fguv_32x32xn_fn##B)(pixel *dst_row, const pixel *src_row, ptrdiff_t stride, const Dav1dHipFilmGrainData *data, \
                                             size_t pw, const uint8_t *scaling, const entry grain_lut[][DAV1D_HIP_GRAIN_WIDTH], \
                                             int bh, int row_num, const pixel *luma_row, ptrdiff_t luma_stride, int uv_pl, int is_id BD); \
typedef struct Dav1dHipFilmGrainDSPContext##B { \
    dav1d_hip_generate_grain_y_fn##B generate_grain_y; \
    dav1d_hip_generate_grain_uv_fn##B generate_grain_uv[3]; \
    dav1d_hip_fgy_32x32xn_fn##B fgy_32x32xn; \
    dav1d_hip_fguv_32x32xn_fn##B fguv_32x32xn[3]; \
} Dav1dHipFilmGrainDSPContext##B; \
typedef struct Dav1dHipIntraPredDSPContext##B { \
    dav1d_hip_angular_ipred_fn##B intra_pred[DAV1D_HIP_N_IMPL_INTRA_PRED_MODES]; \
    dav1d_hip_cfl_ac_fn##B cfl_ac[3]; \
    dav1d_hip_cfl_pred_fn##B cfl_pred[6];    /* DC_PRED, -, -, LEFT_DC_PRED, TOP_DC_PRED, DC_128_PRED */ \
    dav1d_hip_pal_pred_fn##B pal_pred; \
} Dav1dHipIntraPredDSPContext##B; \
typedef struct Dav1dHipMCDSPContext##B { \
    dav1d_hip_mc_fn##B mc[DAV1D_HIP_N_2D_FILTERS]; \
    dav1d_hip_mc_scaled_fn##B mc_scaled[DAV1D_HIP_N_2D_FILTERS]; \
    dav1d_hip_mct_fn##B mct[DAV1D_HIP_N_2D_FILTERS]; \
    dav1d_hip_mct_scaled_fn##B mct_scaled[DAV1D_HIP_N_2D_FILTERS]; \
    dav1d_hip_avg_fn##B avg; \
    dav1d_hip_w_avg_fn##B w_avg; \
    dav1d_hip_mask_fn##B mask; \
    dav1d_hip_w_mask_fn##B w_mask[3]; \
    dav1d_hip_blend_fn##B blend; \
    dav1d_hip_blend_dir_fn##B blend_v; \
    dav1d_hip_blend_dir_fn##B blend_h; \
    dav1d_hip_warp8x8_fn##B warp8x8; \
    dav1d_hip_warp8x8t_fn##B warp8x8t; \
    dav1d_hip_emu_edge_fn##B emu_edge; \
    dav1d_hip_resize_fn##B resize; \
} Dav1dHipMCDSPContext##B; \
typedef struct Dav1dHipInvTxfmDSPContext##B { \
    dav1d_hip_itxfm_fn##B itxfm_add[DAV1D_HIP_N_RECT_TX_SIZES][DAV1D_HIP_N_TX_TYPES_PLUS_LL]; \
} Dav1dHipInvTxfmDSPContext##B; \
typedef struct Dav1dHipLoopFilterDSPContext##B { dav1d_hip_loopfilter_sb_fn##B loop_filter_sb[2][2]; } Dav1dHipLoopFilterDSPContext##B; \
typedef struct Dav1dHipCdefDSPContext##B { dav1d_hip_cdef_dir_fn##B dir; dav1d_hip_cdef_fn##B fb[3]; } Dav1dHipCdefDSPContext##B; \
typedef struct Dav1dHipLoopRestorationDSPContext##B { dav1d_hip_lr_fn##B wiener[2]; dav1d_hip_lr_fn##B sgr[3]; } \
    Dav1dHipLoopRestorationDSPContext##B; \
typedef struct Dav1dHipDSPContext##B { \
    Dav1dHipFilmGrainDSPContext##B fg; \
    Dav1dHipIntraPredDSPContext##B ipred; \
    Dav1dHipMCDSPContext##B mc; \
    Dav1dHipInvTxfmDSPContext##B itx; \
    Dav1dHipLoopFilterDSPContext##B lf; \
    Dav1dHipCdefDSPContext##B cdef; \
    Dav1dHipLoopRestorationDSPContext##B lr; \
} Dav1dHipDSPContext##B;

DAV1D_HIP_DSP_TABLE(8, uint8_t, int16_t, int8_t, DAV1D_HIP_BD_NONE)
DAV1D_HIP_DSP_TABLE(16, uint16_t, int32_t, int16_t, DAV1D_HIP_BD_MAX)

/* Counterparts of dav1d_{film_grain,intra_pred,mc,itx,loop_filter,cdef,loop_restoration}_dsp_init_{8,16}bpc (reference
 * src/lib.c:83-117 / src/decode.c:3387-3415): every entry of the table is overwritten with the HIP-backed function
 * (illegal itxfm_add combinations stay NULL, as in src/itx_tmpl.c:220-311).  They bind the table to the process-wide
 * default context (device 0 unless DAV1D_HIP_DEVICE is set), opened on first use; they return -ENODEV when that
 * fails and leave the table zeroed -- there is no CPU fallback. */
DAV1D_HIP_API int dav1d_hip_dsp_init_8bpc(Dav1dHipDSPContext8 *c);
DAV1D_HIP_API int dav1d_hip_dsp_init_16bpc(Dav1dHipDSPContext16 *c, int bpc);

#ifdef __cplusplus
}
#endif
#endif /* DAV1D_HIP_H */

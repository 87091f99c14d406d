// Internal declarations shared by the C-ABI translation units (host side).
#pragma once
#include "common.h"
#include <errno.h>
#include <mutex>
#include <vector>

struct Dav1dHipContext {
    int device;
    hipStream_t stream;
    bool own_stream;
    // scratch used by the reference-signature single-call wrappers (dsp_table.hip)
    void *scratch;
    size_t scratch_size;
    // The per-shape kernels of one family touch disjoint pixels, so they run concurrently on
    // side streams (fork from / join into `stream` with events): the long-latency, small-grid
    // shapes (64x64 transforms ...) overlap with the wide ones instead of serialising.
    enum { N_SIDE = 6 };
    hipStream_t side[N_SIDE];
    hipEvent_t ev_fork, ev_join[N_SIDE];
    hipEvent_t ev_pair[3];      // ends of the side streams the paired launches of a recon list went to (events of their own: StreamFan re-records ev_join[])
    hipEvent_t ev_bin[16];      // "this tile shape's predictions are in the picture" (recon list pipeline)
    bool concurrent;
    int flow_min_steps;         // wavefronts of at least this many steps run as one dataflow launch ($DAV1D_HIP_FLOW_MIN_STEPS, 0 = never)
    int flow_mode;              // $DAV1D_HIP_FLOW_MODE at open: hand-off variant of the intra dataflow launch (intra_flow.hip)
    int flow_groups;            // workgroups of the intra dataflow launch ($DAV1D_HIP_FLOW_GROUPS at open, default 512; every wave has to be resident: units are dealt out round-robin)
    int intra_sb;               // the intra wavefront superblock by superblock (intra_sb.hip; $DAV1D_HIP_INTRA_SB / option intra_sb): 0 never,
                                // 1 for wavefronts of at least flow_min_steps steps, 2 (default) for every frame whose tiling is known
    int intra_sb_flow;          // 1 (default): every level of a frame's superblocks in ONE launch, a superblock waiting for the flags of the neighbours it reads
                                // ($DAV1D_HIP_INTRA_SB_FLOW / option intra_sb_flow); 0: a launch per level
    int intra_sb_lds;           // 1: the superblock's pixels stay in LDS where that form exists (4:2:0); 0 (default): handed over through the L2 — measured
                                // equal or a little faster ($DAV1D_HIP_INTRA_SB_LDS / option intra_sb_lds)
    int intra_sb_waves;         // waves per workgroup of that route: 1, 4, 8 or 0 = the kernel form's own choice ($DAV1D_HIP_INTRA_SB_WAVES / option intra_sb_waves;
                                // the one-launch form chooses 1 where the superblocks hold intra_sb_one_below units or fewer on average and nothing is copied,
                                // 4 where the levels are wide (>= 128 superblocks on average) and nothing is copied, else 8; the per-level launches 8)
    long intra_sb_fallbacks;    // frames whose one-launch intra pass was finished by launches per level (workgroups gave up waiting: never in a sound run); dav1d_hip_get_stat
    int intra_sb_one_below;     // see intra_sb_waves ($DAV1D_HIP_INTRA_SB_ONE_BELOW / option intra_sb_one_below; 0: never one wave)
    int recon_fuse;             // bit mask of the square block sizes that run paired (DAV1D_HIP_RECON_FUSE)
    long recon_pipeline;        // smallest residual list a recon list pipelines on two streams (DAV1D_HIP_RECON_PIPELINE)
    int recon_lanes;            // side streams of the residual launches (DAV1D_HIP_RECON_LANES)
    int chunk_upload;           // 0: a frame's chunks go up as one transfer at frame end; 1: each chunk as it is submitted (DAV1D_HIP_CHUNK_UPLOAD)
    int recon_coop_below;       // paired kernels: launches of fewer groups than this take the cooperative form (recon.hip; DAV1D_HIP_RECON_COOP_BELOW)
    int post_bands;             // bands of the pipelined post filters, 0 = stage by stage (DAV1D_HIP_POST_BANDS)
    int recon_pair_streams;     // side streams the paired launches of a recon list are dealt over (DAV1D_HIP_RECON_PAIR_STREAMS, 1 .. 3; default 2)
    int recon_pair_first;       // ... starting at side stream 1 .. 3 (DAV1D_HIP_RECON_PAIR_FIRST; default 2): which HARDWARE queue a stream shares with which other is a matter of the order the
                                // streams were made in (the runtime deals them over GPU_MAX_HW_QUEUES = 4 queues), profiles/r06/hw_queues.txt
    hipStream_t pad_streams[8]; int n_pad_streams;      // (DAV1D_HIP_STREAM_PAD: unused streams made in front of the side streams — shifts that dealing; an experiment's knob)
    int ref_twin;               // tiled twins of reference pictures ($DAV1D_HIP_REF_TWIN): 0 never read, 1 (default) read when a picture has a valid
                                // one (dav1d_hip_picture_retile), 2 also made for every picture of dav1d_hip_picture_alloc and by dav1d_hip_frame_end
    bool cdef_rows;             // option cdef_rows (default 1) / $DAV1D_HIP_CDEF_ROWS: the filter lister hands over one record per unit row of a 64-pixel column and the device makes the unit records (cdef.hip cdef_expand_kernel)
    bool cdef_full_copy;        // option filter_full_copy / $DAV1D_HIP_FILTER_FULL_COPY=1: CDEF and restoration start from a copy of the whole picture (A/B aid)
    bool cdef_unit_kernel;      // $DAV1D_HIP_CDEF_UNIT=1 at open: one wave per 8x8 unit (the round-1 kernel) instead of strips; A/B aid
    // measurement aid: device time of the kernel launches of the most recent *_batch call (dav1d_hip_last_kernel_ms)
    hipEvent_t ev_t0, ev_t1;
    float last_ms;
    // chunked frames (chunk.hip): pinned slabs recycled between frames, the device arenas of the frame in flight, a copy stream
    struct Slab { uint8_t *host; size_t cap; };
    std::mutex pool_mtx;
    std::vector<Slab> free_slabs[48];          // by size class: slab capacities are powers of two, class = log2(cap)
    struct Arena { uint8_t *dev; size_t cap; };
    std::vector<Arena> free_arenas;            // chunk arenas of finished frames (a frame in flight owns one)
    std::vector<Arena> free_task_bufs;         // task-list buffers of the *_batch calls (TaskBuf below)
    std::vector<Dav1dHipPicture> free_pictures; // the frames' own pictures (CDEF / restoration outputs) between frames: hipMalloc and
                                               // above all hipFree (it waits for the device) stay out of the per-frame path
    size_t arena_hint;                         // what the largest frame so far needed
    uint32_t *band_cnt = nullptr, *band_flags = nullptr;   // frame_lr_banded (frame.hip): 64 counters + 64 targets on the device, 64 words of pinned host memory
    uint32_t band_seq = 0;
    int chunk_hints;                           // option chunk_hints (default 1): the lister's records are prepared from what the walk wrote into them (chunk.hip chunk_build_hinted); 0 = through the general preparation
    int prep_async;                            // option prep_async: the chunk preparation of a tile-sbrow on the library's own threads instead of the submitting one — 1 (default)
                                               // for frames of at most 8 tiles (few listing threads: the walk of a tile's next row runs next to the preparation of the last), 2 always, 0 never
    int chunk_order;                           // option chunk_order: the prepared lists of a tile-sbrow ordered for the device (1) or left in decode order (0)
    size_t arena_min;                          // size of a frame's chunk arena before anything is known (option chunk_arena_min; tests make it tiny)
    size_t uarena_hint = 0;                    // bytes of intra units the largest frame so far carried (sizes a frame's pinned unit arena)
    size_t carena_hint;                        // bytes of packed coefficients the largest frame so far carried (sizes the pinned twin)
    uint8_t *gather_dev, *segtab_dev;
    size_t gather_cap, segtab_cap;
    uint8_t *pending_slab;
    size_t pending_slab_cap;
    hipStream_t copy_stream;
    hipEvent_t ev_copy;
    hipEvent_t ev_untile;       // "the rows of this band are raster rows again" (dav1d_hip_host_picture_fetch of a picture that lives in its twin)
    hipEvent_t ev_retile;       // the end of the most recent overlapped retile (dav1d_hip_picture_retile_overlapped) ...
    bool retile_pending;        // ... which the next launches that read twins have to wait for
    std::mutex run_mtx;         // one multi-stream section (recon list run, banded post filters) at a time per context
};

// Device memory for the task list of one *_batch call, taken from a pool of the context.  hipMalloc + hipFree per call was the
// obvious form and is fine for one context; hipFree waits for the whole device, so with several contexts at work (frames in
// flight) every call of one frame stalled the others.  The pool only grows; dav1d_hip_close frees it.
struct TaskBuf {
    Dav1dHipContext *c;
    uint8_t *p = nullptr;
    size_t cap = 0;
    TaskBuf(Dav1dHipContext *ctx, size_t bytes) : c(ctx) {
        {
            std::lock_guard<std::mutex> lk(c->pool_mtx);
            int best = -1;
            for (int i = 0; i < (int) c->free_task_bufs.size(); i++)
                if (c->free_task_bufs[i].cap >= bytes && (best < 0 || c->free_task_bufs[i].cap < c->free_task_bufs[best].cap)) best = i;
            if (best >= 0) {
                p = c->free_task_bufs[best].dev; cap = c->free_task_bufs[best].cap;
                c->free_task_bufs.erase(c->free_task_bufs.begin() + best);
                return;
            }
        }
        size_t want = 1 << 16;
        while (want < bytes) want <<= 1;
        void *q = nullptr;
        if (hipMalloc(&q, want) == hipSuccess) { p = (uint8_t *) q; cap = want; }
    }
    ~TaskBuf() {
        if (!p) return;
        std::lock_guard<std::mutex> lk(c->pool_mtx);
        c->free_task_bufs.push_back({ p, cap });
    }
    TaskBuf(const TaskBuf &) = delete;
    TaskBuf &operator=(const TaskBuf &) = delete;
};


// brackets the launches of a batch call with events on the context's stream
struct KernelTimer {
    Dav1dHipContext *c;
    explicit KernelTimer(Dav1dHipContext *ctx) : c(ctx) { (void) hipEventRecord(c->ev_t0, c->stream); }
    void stop() {        // call after the launches, before the batch call's own synchronize
        (void) hipEventRecord(c->ev_t1, c->stream);
        (void) hipEventSynchronize(c->ev_t1);
        c->last_ms = 0.f;
        (void) hipEventElapsedTime(&c->last_ms, c->ev_t0, c->ev_t1);
    }
};

// fork/join helper: launches issued through next() land round-robin on the side streams
struct StreamFan {
    Dav1dHipContext *c;
    int used;
    bool on;      // small lists stay on the context's stream: forking costs more than their kernels run
    explicit StreamFan(Dav1dHipContext *ctx, bool worth_it = true) : c(ctx), used(0), on(ctx->concurrent && worth_it) {
        if (on) (void) hipEventRecord(c->ev_fork, c->stream);
    }
    hipStream_t next() {
        if (!on) return c->stream;
        const int i = used % Dav1dHipContext::N_SIDE;
        if (used < Dav1dHipContext::N_SIDE) (void) hipStreamWaitEvent(c->side[i], c->ev_fork, 0);
        used++;
        return c->side[i];
    }
    void join() {
        if (!on) return;
        const int n = used < Dav1dHipContext::N_SIDE ? used : Dav1dHipContext::N_SIDE;
        for (int i = 0; i < n; i++) {
            (void) hipEventRecord(c->ev_join[i], c->side[i]);
            (void) hipStreamWaitEvent(c->stream, c->ev_join[i], 0);
        }
        used = 0;
    }
};

// the HIP error behind the last non-zero return on this thread (dav1d_hip_last_hip_error): the C ABI speaks errno, and -EIO alone
// does not say whether a launch was refused, a kernel faulted or a copy failed
extern "C" { extern thread_local int dav1d_hip_tls_last_error; }
static inline int hip_rc(hipError_t e) {
    if (e == hipSuccess) return 0;
    dav1d_hip_tls_last_error = (int) e;
    if (e == hipErrorOutOfMemory) return -ENOMEM;
    if (e == hipErrorInvalidValue) return -EINVAL;
    if (e == hipErrorNotSupported) return -ENOSYS;
    return -EIO;
}
#define HIP_TRY(x) do { const int rc_ = hip_rc(x); if (rc_) return rc_; } while (0)

static inline DevPlanes dev_planes(const Dav1dHipPicture *p) {
    DevPlanes d;
    const int bps = p->bpc > 8 ? 2 : 1;
    for (int i = 0; i < 3; i++) {
        d.data[i] = p->p[i].data;
        d.stride[i] = (int) (p->p[i].stride / bps);
        d.w[i] = p->p[i].w;
        d.h[i] = p->p[i].h;
    }
    d.tiled = 0;
    return d;
}
static inline bool picture_twin_usable(const Dav1dHipPicture *p) {
    if (!p->twin_ok) return false;          // (1: twin and raster planes agree; DAV1D_HIP_TWIN_ONLY: the twin is the picture)
    const int bps = p->bpc > 8 ? 2 : 1;
    for (int i = 0; i < 3; i++)
        if (p->p[i].data && (!p->twin[i] || (p->p[i].stride / bps) % 8)) return false;
    return true;
}
// Before a launch that reads RASTER planes of pictures: the ones that live in their twin only (DAV1D_HIP_TWIN_ONLY, what a
// reconstruction in the tiled layout leaves) get their raster planes back first, on the context's stream — a raster reader of such a
// picture would take stale pixels for the picture without any sign of it.  (The pictures are const for the caller's sake: their pixels do
// not change; a caller that handed over a copy of the record un-tiles again next time.)
extern "C" int dav1d_hip_picture_untile(Dav1dHipContext *c, Dav1dHipPicture *pic);
static inline int raster_planes_valid(Dav1dHipContext *c, const Dav1dHipPicture *pics, int n) {
    for (int i = 0; i < n; i++)
        if (pics[i].twin_ok == DAV1D_HIP_TWIN_ONLY)
            if (const int rc = dav1d_hip_picture_untile(c, const_cast<Dav1dHipPicture *>(&pics[i]))) return rc;
    return 0;
}
// The planes motion compensation reads its references through: the tiled twins when the context uses them and EVERY reference
// of the call has a valid one (a launch is one kernel variant: all tiled or all raster), the raster planes otherwise (made valid first).
static inline int ref_planes(Dav1dHipContext *c, const Dav1dHipPicture *refs, int n_refs, DevPlanes *rp) {
    bool tiled = c->ref_twin != 0 && n_refs > 0;
    for (int i = 0; i < n_refs && tiled; i++) tiled = picture_twin_usable(&refs[i]);
    if (!tiled) if (const int rc = raster_planes_valid(c, refs, n_refs)) return rc;
    if (tiled && c->retile_pending) {          // a twin of this context may still be on its way on the side stream
        (void) hipStreamWaitEvent(c->stream, c->ev_retile, 0);
        c->retile_pending = false;
    }
    for (int i = 0; i < n_refs; i++) {
        rp[i] = dev_planes(&refs[i]);
        if (tiled) {
            for (int pl = 0; pl < 3; pl++) rp[i].data[pl] = refs[i].p[pl].data ? refs[i].twin[pl] : nullptr;
            rp[i].tiled = 1;
        }
    }
    return 0;
}

// kernel launchers (one per family translation unit)
extern "C" int dav1d_hip_launch_itx_all(const DevPlanes *dst, int bpc, const Dav1dHipItxTask *tasks, const size_t *off,
                                        void *coef, void *stream);
extern "C" int dav1d_hip_launch_itx_bin(const DevPlanes *dst, int bpc, int tx, const Dav1dHipItxTask *tasks,
                                        int n, void *coef, void *stream);
// Device-side unit of motion compensation: a tile of at most 16x16 cut out of a
// Dav1dHipMcTask by the host (mc list creation), filter rows already resolved
// (reference GET_H_FILTER / GET_V_FILTER, src/mc_tmpl.c:115-123).
// A tile carries one prediction (PUT / PREP) or, when the list builder could pair a
// compound task with the two PREP tasks that feed it, both predictions plus the combine
// (AVG / WAVG): the int16 intermediates then never leave registers.
enum { MCT_PUT = 0, MCT_PREP = 1, MCT_AVG = 2, MCT_WAVG = 3, MCT_PUT_TMP = 4 };
struct McRef {
    int32_t  src_x, src_y;// of the tile's top-left in the reference plane
    uint8_t  mx, my;
    uint8_t  fh, fv;      // row of av1_mc_subpel_filters (0..5) or 6 = bilinear
    uint8_t  ref;         // index into the reference picture set
    uint8_t  vspan;       // av1_mc_tap_span of (fv, my): the window rows the vertical taps reach (filled at list creation)
    uint8_t  hspan;       // the same of (fh, mx): the window columns the horizontal taps reach, origin src_x - 3 (tiled references)
    uint8_t  pad;
};
struct McTile {
    uint32_t dst_off;     // of the TASK: pixel offset in the dst plane; PREP: int16 offset in the prep arena
    uint8_t  w, h;        // tile size, <= 64 x 16
    uint8_t  kind, plane;
    uint8_t  bw;          // task width = row stride of a PREP block
    uint8_t  ox, oy;      // tile origin inside the task's block
    int8_t   weight;      // WAVG
    McRef    r[2];
};
// a wave-sized run of tiles of one shape inside the source-ordered tile list (all-shapes launch)
struct McGroup {
    uint32_t start;       // first tile
    uint16_t n;           // tiles in the group, <= 64 / lanes-per-tile of the shape
    uint16_t cls;         // tile-shape bin
};
extern "C" int dav1d_hip_launch_recon_fused(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc, int cls, const McTile *tiles,
                                            const Dav1dHipItxTask *tasks, int n, int16_t *prep, void *coef, int coop_below, void *stream);
extern "C" int dav1d_hip_launch_recon_fused_out(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc, int cls, const McTile *tiles,
                                                const Dav1dHipItxTask *tasks, int n, int16_t *prep, void *coef, int coop_below, int wide,
                                                const DevPlanes *dst_twin, void *stream);
extern "C" int dav1d_hip_launch_mc_bin_twin(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc, int cls,
                                            const McTile *tiles, int n, int16_t *prep, const DevPlanes *dst_twin, void *stream);
extern "C" int dav1d_hip_launch_itx_bin_out(const DevPlanes *dst, int bpc, int tx, const Dav1dHipItxTask *tasks, int n, void *coef, int wide,
                                            const DevPlanes *dst_twin, void *stream);
extern "C" int dav1d_hip_launch_mc_all(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc, const McTile *tiles,
                                       const McGroup *groups, int n_groups, int with_small, int16_t *prep, void *stream);
extern "C" int dav1d_hip_launch_mc_bin(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc, int cls,
                                       const McTile *tiles, int n, int16_t *prep, void *stream);
extern "C" int dav1d_hip_launch_comp(const DevPlanes *dst, int bpc, const Dav1dHipCompTask *tasks, int n,
                                     const int16_t *prep, uint8_t *mask, void *stream);

// One unit of the intra dataflow launch (intra_flow.hip): prediction and / or residual of one transform block, records included
// so that a wave has everything about a unit after ONE scalar fetch.
struct IntraUnit {
    uint32_t need;              // host side: units in earlier groups (a group = the units of one step that may run together);
                                // equal values mark a group.  The device copy also carries what the waves use:
    uint32_t has;               // bit 0: p holds a prediction, bit 1: t holds a residual, bit 2 (superblock route): the prediction is the intra half of an
                                // inter-intra block, blended into the picture's pixels with the mask at p.aux_off
    uint32_t grp;               // dense index of the unit's group
    uint32_t prev_n;            // units in the group before it (0 for the first): that many completions open this group
    Dav1dHipIpredTask p;
    Dav1dHipItxTask t;
    uint32_t pad2;
};
static_assert(sizeof(IntraUnit) == 80, "unit record layout");
enum { FLOW_SUB = 8, FLOW_SUB_STRIDE = 32 };       // completion counters per group, words between them (a 128-byte line each)
extern "C" int dav1d_hip_launch_intra_flow(const DevPlanes *dst, int bpc, int layout, const IntraUnit *units, int n_units, uint8_t *aux,
                                           void *coef, uint32_t *ctr, int n_waves, int mode, void *stream);
struct Dav1dHipIntraFlow;
extern "C" int dav1d_hip_intra_units_build(const Dav1dHipIpredTask *preds, const uint32_t *pred_end, const Dav1dHipItxTask *txs, const uint32_t *tx_end,
                                size_t n_steps, std::vector<IntraUnit> &units, std::vector<uint32_t> &ua_end, std::vector<uint32_t> &ub_end,
                                const Dav1dHipCompTask *blends, const uint32_t *blend_end);
extern "C" int dav1d_hip_intra_flow_from_units(Dav1dHipContext *c, Dav1dHipIntraFlow **out, IntraUnit *units, size_t n);
// batches (wavefront steps) of predictions + residuals -> device-resident unit list; -ENOTSUP when the list holds a task
// kind the dataflow launch does not run (PRED_TMP for inter-intra, DSP-level kinds): the caller keeps the stepped route
extern "C" int dav1d_hip_intra_flow_create(Dav1dHipContext *c, Dav1dHipIntraFlow **out, const Dav1dHipIpredTask *preds, const size_t *pred_sizes,
                                           const Dav1dHipItxTask *txs, const size_t *tx_sizes, size_t n_batches);
extern "C" int dav1d_hip_intra_flow_run(Dav1dHipContext *c, const Dav1dHipIntraFlow *l, const Dav1dHipPicture *dst, void *coef, uint8_t *aux);
extern "C" void dav1d_hip_intra_flow_destroy(Dav1dHipContext *c, Dav1dHipIntraFlow *l);
extern "C" size_t dav1d_hip_intra_flow_units(const Dav1dHipIntraFlow *l);
extern "C" int dav1d_hip_intra_flow_status(Dav1dHipContext *c, const Dav1dHipIntraFlow *l, uint32_t out[3]);

// ---- the intra wavefront superblock by superblock (intra_sb.hip): one workgroup per superblock, a launch per level
struct SbRegion { uint32_t first, n; uint16_t x0, y0; uint32_t pad; uint32_t dep[4]; };   // device: the superblock's records (header, then units) [first, first + n), its luma origin
struct SbPart { uint32_t sb, first, n; };          // host: superblock number (raster, frame-wide) and its run in a sorted unit array
struct SbTiling {                                   // the frame's tiles in superblocks (frame_hdr->tiling.col_start_sb / row_start_sb)
    int sb_log2, sbw, sbh, n_cols, n_rows;
    uint16_t col_start[65], row_start[65];
};
void dav1d_hip_sbw_set_fine(int on);
void dav1d_hip_sbw_set_fail_at(int at);
int dav1d_hip_sb_tiling_make(SbTiling *tl, int w, int h, int sb128, int n_cols, const uint16_t *col_start_sb, int n_rows, const uint16_t *row_start_sb);
enum : uint32_t { SB_NONE = 0xffffffffu };
struct SbSort {
    std::vector<uint32_t> pos, key;                 // per unit: its record in the output, its (step, kind) key
    std::vector<SbPart> parts;                      // per superblock: number, its HEADER record in the output, its units (they follow the header)
    size_t n_records;                               // units + superblocks
    std::vector<uint64_t> copy_deps;                // intra block copies: (superblock << 32 | superblock its source window lies in), other superblocks only
};
int dav1d_hip_sbw_prepare(std::vector<IntraUnit> &units, const std::vector<uint32_t> &ua_end, const std::vector<uint32_t> &ub_end,
                          const SbTiling &tl, const int strides[3], int ss_hor, int ss_ver, SbSort &st);
void dav1d_hip_sbw_emit(const std::vector<IntraUnit> &units, const SbSort &st, IntraUnit *out);
// extra (or nullptr): further (superblock << 32 | superblock it waits for) pairs, sorted — the sources of intra block copies
int dav1d_hip_sbw_levels(const SbTiling &tl, const uint32_t *sbs, size_t n, const uint8_t *dep, std::vector<int> &level_of_sb, std::vector<uint32_t> &level,
                         const std::vector<uint64_t> *extra = nullptr);
// where (with flags; DEVICE, one word per superblock of the frame, raster): the superblock's place in `regions` or SB_NONE; sbw: superblocks per row;
// done (DEVICE, a byte per record of `units`, or nullptr): with flags, zeroed by the caller and set behind every unit the launch reconstructs; without flags,
// the units a launch leaves alone — how launches per level finish a frame whose one-launch pass gave up (frame.hip)
extern "C" int dav1d_hip_launch_intra_sb(const DevPlanes *dst, int bpc, int layout, const IntraUnit *units, const SbRegion *regions, int n_regions,
                                         uint8_t *aux, const uint8_t *mask, void *coef, int waves, int sb_log2, int lds, uint32_t *flags, void *stream,
                                         const uint32_t *where = nullptr, int sbw = 0, uint8_t *done = nullptr);
// regions sorted by level + where each level starts, from the parts of any number of unit arrays laid end to end (base[k] = where
// array k starts): host-side plan of a frame's launches
struct SbPlan { std::vector<SbRegion> regions; std::vector<uint32_t> level_start; /* n_levels + 1 */ std::vector<uint32_t> where; /* superblock -> its region */ };
int dav1d_hip_sbw_plan(const SbTiling &tl, const std::vector<const std::vector<SbPart> *> &parts, const std::vector<size_t> &base, const uint8_t *dep,
                       SbPlan &plan, const std::vector<uint64_t> *extra = nullptr);

// raw_only: tasks without the RAW flag are left alone (they are run by groups, below)
extern "C" int dav1d_hip_launch_cdef(const DevPlanes *dst, const DevPlanes *src, int bpc, int layout,
                                     const Dav1dHipCdefTask *tasks, int n, int damping, uint32_t *dirvar, int raw_only, void *stream);

// Up to 16 listed units of ONE unit row, tasks[first .. first + n) in rising bx, all within [bx0, bx0 + span): what one wave of
// the strip kernel filters from a shared window.  edges: LEFT of the first unit, RIGHT of the last, TOP / BOTTOM of all.
struct CdefGroup {
    uint32_t first;
    uint16_t bx0, by;
    uint8_t n, span, edges, pad;      // pad: the units' DAV1D_HIP_CDEF_BOT_REP_* flags
};
// appends the groups of tasks[0 .. n) (indices offset by `base`) in list order; returns the number of RAW tasks met
size_t dav1d_hip_cdef_make_groups(const Dav1dHipCdefTask *tasks, size_t n, size_t base, std::vector<CdefGroup> &out);
bool dav1d_hip_cdef_strip_ok(const DevPlanes *dst, const DevPlanes *src, int bpc);
extern "C" int dav1d_hip_launch_cdef_expand(const void *rows, int w64, int h8, int bw4, int bh4, int w8, Dav1dHipCdefTask *tasks, void *groups,
                                            uint32_t *bitmap, void *stream);
extern "C" int dav1d_hip_launch_cdef_fill_unlisted(const DevPlanes *dst, const DevPlanes *src, int bpc, int layout, const Dav1dHipCdefTask *tasks,
                                                   int n, uint32_t *bitmap, int w8, int h8, void *stream);
extern "C" int dav1d_hip_launch_cdef_groups(const DevPlanes *dst, const DevPlanes *src, int bpc, int layout, const Dav1dHipCdefTask *tasks,
                                            const CdefGroup *groups, int n_groups, int damping, uint32_t *dirvar, void *stream);
// tasks + ready-made groups (host arrays) -> upload, strip kernel (+ the unit kernel for RAW tasks), synchronize
int dav1d_hip_cdef_run_groups(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src, const Dav1dHipCdefTask *tasks,
                              size_t n, const CdefGroup *groups, size_t n_groups, size_t n_raw, int damping, uint32_t *dirvar);

extern "C" int dav1d_hip_launch_lf(const DevPlanes *dst, int bpc, int dir, const Dav1dHipLfTask *tasks, int n, const uint8_t *lvl,
                                   int b4_stride, const uint8_t *lut_e, const uint8_t *lut_i, void *stream);

extern "C" int dav1d_hip_launch_intra_step(const DevPlanes *dst, int bpc, int layout, const Dav1dHipIpredTask *tasks, int n, int n_big,
                                           const Dav1dHipIpredTask *preds, const Dav1dHipItxTask *txs, int n_pairs, uint8_t *aux, void *tmp,
                                           void *coef, void *stream);
extern "C" int dav1d_hip_launch_intra_pairs(const DevPlanes *dst, int bpc, int layout, const Dav1dHipIpredTask *preds,
                                            const Dav1dHipItxTask *txs, int n, uint8_t *aux, void *coef, void *stream);
extern "C" int dav1d_hip_launch_ipred(const DevPlanes *dst, int bpc, int layout, const Dav1dHipIpredTask *tasks, int n, int n_big,
                                      uint8_t *pal_idx, void *tmp, void *stream);

// Completion of row bands signalled from INSIDE a launch (frame_lr_banded, frame.hip): every workgroup that has stored its pixels bumps its
// band's counter; the one that completes a band writes `seq` to the band's word in pinned host memory.  cnt == nullptr: off.
struct BandSignal { uint32_t *cnt; const uint32_t *target; uint32_t *host_flags; uint32_t seq; };
extern "C" int dav1d_hip_launch_wiener_sig(const DevPlanes *dst, const DevPlanes *src, const DevPlanes *lpf, int bpc,
                                           const Dav1dHipLrTask *tasks, int n, int max_w, const BandSignal *sig, void *stream);
extern "C" int dav1d_hip_launch_sgr_sig(const DevPlanes *dst, const DevPlanes *src, const DevPlanes *lpf, int bpc,
                                        const Dav1dHipLrTask *tasks, const void *waves, int n_waves, const BandSignal *sig, void *stream);
extern "C" int dav1d_hip_launch_wiener(const DevPlanes *dst, const DevPlanes *src, const DevPlanes *lpf, int bpc,
                                       const Dav1dHipLrTask *tasks, int n, int max_w, void *stream);

extern "C" int dav1d_hip_launch_fg_gen(int16_t *luts, const Dav1dHipFilmGrainData *data, int bpc, int layout, void *stream);
extern "C" int dav1d_hip_launch_fg_apply(const DevPlanes *dst, const DevPlanes *src, const int16_t *luts, const uint8_t *scaling,
                                         int scaling_size, const Dav1dHipFilmGrainData *data, int bpc, int layout, int is_id, uint8_t *offs,
                                         void *stream);

extern "C" int dav1d_hip_launch_fg_gen_part(int16_t *luts, const Dav1dHipFilmGrainData *data, int bpc, int layout, int part, void *stream);
extern "C" int dav1d_hip_launch_fg_apply_rows(const DevPlanes *dst, const DevPlanes *src, const int16_t *luts, const uint8_t *scaling,
                                              int scaling_size, const Dav1dHipFilmGrainData *data, int bpc, int layout, int is_id,
                                              int row_num, int pl, uint8_t *offs, void *stream);
extern "C" int dav1d_hip_launch_sgr(const DevPlanes *dst, const DevPlanes *src, const DevPlanes *lpf, int bpc,
                                    const Dav1dHipLrTask *tasks, const void *waves, int n_waves, void *stream);
void dav1d_hip_sgr_make_rows(Dav1dHipLrTask *tasks, size_t n, std::vector<uint32_t> &waves);

extern "C" int dav1d_hip_launch_warp(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc, const Dav1dHipWarpTask *tasks, int n,
                                     int16_t *prep, void *stream);
extern "C" int dav1d_hip_launch_mc_scaled(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc,
                                          const Dav1dHipMcScaledTask *tasks, int n, int16_t *prep, void *stream);
extern "C" int dav1d_hip_launch_resize(const DevPlanes *dst, const DevPlanes *src, int bpc, int plane, int dst_w, int y0, int h, int src_w,
                                       int dx, int mx0, void *stream);
extern "C" int dav1d_hip_launch_emu_edge(void *dst, ptrdiff_t dst_stride, const void *ref, ptrdiff_t ref_stride, int bw, int bh,
                                         int iw, int ih, int x, int y, int bpc, void *stream);

extern "C" { extern long long dav1d_hip_live[8]; }      // objects alive by kind (dav1d_hip_live_objects)
Dav1dHipContext *dav1d_hip_default_context(void);
// a picture for a frame's own use from the context's pool (zeroed like a fresh one) / back to it
extern "C" int pictures_on_device(const Dav1dHipContext *c, const Dav1dHipPicture *pics, int n);       // capi.hip: 0, or -EXDEV with several devices
extern "C" int dav1d_hip_picture_take(Dav1dHipContext *c, Dav1dHipPicture *pic, int w, int h, int layout, int bpc);
extern "C" void dav1d_hip_picture_give(Dav1dHipContext *c, Dav1dHipPicture *pic);
// the intra wavefront list with the blends of inter-intra blocks (frame driver)
extern "C" int dav1d_hip_intra_list_create_blend(Dav1dHipContext *c, Dav1dHipIntraList **out, const Dav1dHipIpredTask *preds, const size_t *pred_sizes,
                                                 const Dav1dHipItxTask *txs, const size_t *tx_sizes, const Dav1dHipCompTask *blends,
                                                 const size_t *blend_sizes, size_t n_batches);
extern "C" int dav1d_hip_intra_list_run_batch_blend(Dav1dHipContext *c, const Dav1dHipIntraList *l, size_t batch, const Dav1dHipPicture *dst,
                                                    void *coef, uint8_t *aux, int16_t *prep, uint8_t *mask);
int dav1d_hip_scratch(Dav1dHipContext *c, size_t bytes, void **out);

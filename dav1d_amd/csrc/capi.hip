// Host side of the C ABI declared in include/dav1d_hip.h: context, device memory,
// pictures, task-list binning and the batched entry points.
#include "capi.h"

#ifndef RECON_FUSE_DEFAULT
#define RECON_FUSE_DEFAULT 15       // which square block sizes run paired by default: see recon_fuse_mask() below
#endif
#include "lists.h"
#include "av1_scan_prefix.h"
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <atomic>
#include <new>
#include <algorithm>
#include <vector>
extern "C" void dav1d_hip_note_context_device(int device);

extern "C" {

// ------------------------------------------------------------------ context

// Objects alive right now, by kind (0 contexts, 1 frames, 2 listers, 3 host pictures): what a caller that must not leak — dav1d's frame
// contexts under error recovery, src/decode.c:3242-3251 — checks after it has closed everything (tests/test_stream_errors.py).
long long dav1d_hip_live[8];
int dav1d_hip_live_objects(long long out[4]) {
    if (!out) return -EINVAL;
    for (int i = 0; i < 4; i++) out[i] = __atomic_load_n(&dav1d_hip_live[i], __ATOMIC_RELAXED);
    return 0;
}

int dav1d_hip_open(Dav1dHipContext **out, int device, void *stream) {
    if (!out) return -EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return -ENODEV;
    if (hipSetDevice(device) != hipSuccess) return -ENODEV;
    Dav1dHipContext *c = new (std::nothrow) Dav1dHipContext();
    if (!c) return -ENOMEM;
    c->device = device;
    c->own_stream = stream == nullptr;
    dav1d_hip_note_context_device(device);
    c->scratch = nullptr;
    c->scratch_size = 0;
    if (stream) c->stream = (hipStream_t) stream;
    else if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return -ENODEV; }
    // the library holds gfx950 code only: any other device cannot run it
#ifndef DAV1D_HIP_EMU
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6)) { delete c; return -ENODEV; }
    }
#endif
    // tuning knobs: context state, the environment only supplies the defaults at open (dav1d_hip_set_option changes them later)
    auto env_int = [](const char *name, long dflt) { const char *e = getenv(name); return e ? atol(e) : dflt; };
    c->recon_fuse = (int) env_int("DAV1D_HIP_RECON_FUSE", RECON_FUSE_DEFAULT);
    c->recon_pipeline = env_int("DAV1D_HIP_RECON_PIPELINE", 16384);
    c->recon_lanes = (int) env_int("DAV1D_HIP_RECON_LANES", 1);
    c->chunk_upload = (int) env_int("DAV1D_HIP_CHUNK_UPLOAD", 0);
    c->recon_coop_below = (int) env_int("DAV1D_HIP_RECON_COOP_BELOW", 4096);
    c->post_bands = (int) env_int("DAV1D_HIP_POST_BANDS", 0);
    c->ref_twin = (int) env_int("DAV1D_HIP_REF_TWIN", 1);
    c->recon_pair_streams = (int) env_int("DAV1D_HIP_RECON_PAIR_STREAMS", 2);
    // (per-context lists "a,b,...": the i-th context opened in the process takes element i mod length)
    static std::atomic<int> n_opened{0};
    const int ctx_index = n_opened.fetch_add(1);
    auto env_list = [&](const char *name, long dflt) {
        const char *e = getenv(name);
        if (!e || !*e) return dflt;
        int n = 1;
        for (const char *q = e; *q; q++) n += *q == ',';
        int want = ctx_index % n;
        const char *q = e;
        while (want-- > 0) q = strchr(q, ',') + 1;
        return atol(q);
    };
    c->recon_pair_first = (int) std::max(1L, std::min(3L, env_list("DAV1D_HIP_RECON_PAIR_FIRST", 2)));
    // Which streams share a HARDWARE queue.  The runtime deals a process's streams over GPU_MAX_HW_QUEUES (4) hardware queues in the order they are
    // made, and a hardware queue runs its packets in order: a stream that waits for an event holds up every stream behind it in the same queue.  A
    // step's launches run on four streams of the context — main (4x4 pairs, 64-wide predictions, the join), side 0 (64x64 residuals), side 2 and 3 (the
    // paired launches).  With the side streams made straight behind the main stream, side 2 and 3 of the first context share ONE queue, and a second
    // context's side 2 joins them there while its side 3 sits behind its own main stream: three of the four streams that carry the long launches in one
    // queue (rocprofv3's Queue_Id per dispatch, profiles/r06/queue_map.txt).  One unused stream in front of the side streams shifts the dealing so that
    // each of those four has a queue to itself or shares it with a short stream: measured on the 8K step 0.274 -> 0.260 ms with one frame in flight,
    // 0.239 -> 0.229 with two, 4K 0.099 -> 0.088 with one (0.0735 -> 0.0765 with two; three 8K contexts 0.243 -> 0.259); more hardware queues (5 .. 16)
    // or fewer are all slower (profiles/r06/hw_queues.txt).
    c->n_pad_streams = (int) std::max(0L, std::min(8L, env_list("DAV1D_HIP_STREAM_PAD", 1)));
    for (int i = 0; i < c->n_pad_streams; i++)
        if (hipStreamCreateWithFlags(&c->pad_streams[i], hipStreamNonBlocking) != hipSuccess) { c->n_pad_streams = i; break; }
    const char *ser = getenv("DAV1D_HIP_SERIAL");
    c->concurrent = !(ser && atoi(ser));
    const char *cu = getenv("DAV1D_HIP_CDEF_UNIT");
    c->cdef_unit_kernel = cu && atoi(cu);
    c->cdef_full_copy = env_int("DAV1D_HIP_FILTER_FULL_COPY", 0) != 0;
    c->cdef_rows = env_int("DAV1D_HIP_CDEF_ROWS", 1) != 0;
    const char *fg = getenv("DAV1D_HIP_FLOW_GROUPS");
    c->flow_groups = fg && atoi(fg) > 0 ? atoi(fg) : 512;
    const char *fm = getenv("DAV1D_HIP_FLOW_MODE");
    c->flow_mode = fm ? atoi(fm) : 0;
    const char *fs = getenv("DAV1D_HIP_FLOW_MIN_STEPS");
    c->flow_min_steps = fs ? atoi(fs) : 200;
    const char *isb = getenv("DAV1D_HIP_INTRA_SB");
    c->intra_sb = isb ? atoi(isb) : 2;
    c->intra_sb_waves = (int) env_int("DAV1D_HIP_INTRA_SB_WAVES", 0);
    c->intra_sb_one_below = (int) env_int("DAV1D_HIP_INTRA_SB_ONE_BELOW", 0);
    c->intra_sb_fallbacks = 0;
    c->prep_async = (int) env_int("DAV1D_HIP_PREP_ASYNC", 1);
    c->intra_sb_lds = (int) env_int("DAV1D_HIP_INTRA_SB_LDS", 0);
    c->intra_sb_flow = (int) env_int("DAV1D_HIP_INTRA_SB_FLOW", 1);
    // (DAV1D_HIP_PAIR_PRIORITY=1, an experiment's knob: the streams of the paired launches are made with the device's highest stream priority — the
    // runtime keeps a pool of hardware queues per priority, so they are dealt over queues no other stream of the process is in.  Measured: 0.27 -
    // 0.49 ms per 8K step against 0.22 - 0.24, highest or lowest priority alike, profiles/r06/queue_search.txt — off)
    const long pair_prio = env_list("DAV1D_HIP_PAIR_PRIORITY", 0);
    int prio_least = 0, prio_greatest = 0;
#ifndef DAV1D_HIP_EMU
    if (pair_prio) (void) hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
#endif
    auto make_side = [&](const int i) {
#ifndef DAV1D_HIP_EMU
        if (pair_prio && i >= c->recon_pair_first && i < c->recon_pair_first + 3)
            return hipStreamCreateWithPriority(&c->side[i], hipStreamNonBlocking, pair_prio > 0 ? prio_greatest : prio_least);
#endif
        return hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking);
    };
    (void) prio_least; (void) prio_greatest;
    for (int i = 0; i < Dav1dHipContext::N_SIDE; i++) {
        if (make_side(i) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming) != hipSuccess) { delete c; return -ENODEV; }
        if (i < 3 && hipEventCreateWithFlags(&c->ev_pair[i], hipEventDisableTiming) != hipSuccess) { delete c; return -ENODEV; }
    }
    if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) { delete c; return -ENODEV; }
    for (int i = 0; i < 16; i++)
        if (hipEventCreateWithFlags(&c->ev_bin[i], hipEventDisableTiming) != hipSuccess) { delete c; return -ENODEV; }
    if (hipEventCreate(&c->ev_t0) != hipSuccess || hipEventCreate(&c->ev_t1) != hipSuccess) { delete c; return -ENODEV; }
    if (hipEventCreateWithFlags(&c->ev_retile, hipEventDisableTiming) != hipSuccess) { delete c; return -ENODEV; }
    c->retile_pending = false;
    c->last_ms = 0.f;
    c->gather_dev = c->segtab_dev = c->pending_slab = nullptr;
    c->gather_cap = c->segtab_cap = c->pending_slab_cap = 0;
    c->arena_hint = 0;
    c->arena_min = (size_t) 1 << 24;
    c->chunk_order = (int) env_int("DAV1D_HIP_CHUNK_ORDER", 0);
    c->chunk_hints = (int) env_int("DAV1D_HIP_CHUNK_HINTS", 1);
    c->carena_hint = 0;
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_untile, hipEventDisableTiming) != hipSuccess) { delete c; return -ENODEV; }
    *out = c;
    __atomic_fetch_add(&dav1d_hip_live[0], 1, __ATOMIC_RELAXED);
    return 0;
}

void dav1d_hip_close(Dav1dHipContext *c) {
    if (!c) return;
    __atomic_fetch_sub(&dav1d_hip_live[0], 1, __ATOMIC_RELAXED);
    (void) hipSetDevice(c->device);
    // everything this context (or a frame, a grain handle, a peer of it) still has in flight on ANY of its streams is through before the
    // streams, events and pools go: the per-stream waits below do not cover streams that belong to objects the caller destroyed without
    // waiting (a GPU test run died once in here, at the session's end, after 249 green tests)
    (void) hipDeviceSynchronize();
    hipStreamSynchronize(c->stream);
    if (c->scratch) hipFree(c->scratch);
    for (int i = 0; i < c->n_pad_streams; i++) hipStreamDestroy(c->pad_streams[i]);
    for (int i = 0; i < Dav1dHipContext::N_SIDE; i++) { hipStreamSynchronize(c->side[i]); hipStreamDestroy(c->side[i]); hipEventDestroy(c->ev_join[i]); if (i < 3) hipEventDestroy(c->ev_pair[i]); }
    hipEventDestroy(c->ev_fork);
    for (int i = 0; i < 16; i++) hipEventDestroy(c->ev_bin[i]);
    hipEventDestroy(c->ev_t0); hipEventDestroy(c->ev_t1); hipEventDestroy(c->ev_retile);
    hipStreamSynchronize(c->copy_stream); hipStreamDestroy(c->copy_stream); hipEventDestroy(c->ev_copy); hipEventDestroy(c->ev_untile);
    if (c->band_cnt) (void) hipFree(c->band_cnt);
    if (c->band_flags) (void) hipHostFree(c->band_flags);
    for (const Dav1dHipContext::Arena &ar : c->free_arenas) hipFree(ar.dev);
    for (const Dav1dHipContext::Arena &ar : c->free_task_bufs) hipFree(ar.dev);
    for (Dav1dHipPicture &q : c->free_pictures) { if (q.alloc) hipFree(q.alloc); if (q.twin_alloc) hipFree(q.twin_alloc); }
    if (c->gather_dev) hipFree(c->gather_dev);
    if (c->segtab_dev) hipFree(c->segtab_dev);
    if (c->pending_slab) hipHostFree(c->pending_slab);
    for (const std::vector<Dav1dHipContext::Slab> &cls : c->free_slabs) for (const Dav1dHipContext::Slab &sl : cls) hipHostFree(sl.host);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
}

int dav1d_hip_sync(Dav1dHipContext *c) {
    if (c->retile_pending) { (void) hipEventSynchronize(c->ev_retile); c->retile_pending = false; }
    return hip_rc(hipStreamSynchronize(c->stream));
}
void *dav1d_hip_stream(Dav1dHipContext *c) { return (void *) c->stream; }

// ---- recorded launch sequences (hipGraph)
struct Dav1dHipGraph {
    hipGraph_t graph;
    hipGraphExec_t exec;
    size_t nodes;
};

int dav1d_hip_graph_begin(Dav1dHipContext *c) {
    if (!c) return -EINVAL;
    return hip_rc(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
}

int dav1d_hip_graph_end(Dav1dHipContext *c, Dav1dHipGraph **out) {
    if (!c || !out) return -EINVAL;
    *out = nullptr;
    hipGraph_t graph = nullptr;
    HIP_TRY(hipStreamEndCapture(c->stream, &graph));
    if (!graph) return -EIO;
    Dav1dHipGraph *g = new (std::nothrow) Dav1dHipGraph();
    if (!g) { hipGraphDestroy(graph); return -ENOMEM; }
    g->graph = graph;
    g->nodes = 0;
    (void) hipGraphGetNodes(graph, nullptr, &g->nodes);
    const hipError_t e = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { hipGraphDestroy(graph); delete g; return hip_rc(e); }
    *out = g;
    return 0;
}

int dav1d_hip_graph_launch(Dav1dHipContext *c, const Dav1dHipGraph *g) {
    if (!c || !g) return -EINVAL;
    return hip_rc(hipGraphLaunch(g->exec, c->stream));
}

size_t dav1d_hip_graph_nodes(const Dav1dHipGraph *g) { return g ? g->nodes : 0; }

void dav1d_hip_graph_destroy(Dav1dHipContext *c, Dav1dHipGraph *g) {
    if (!g) return;
    if (c) hipStreamSynchronize(c->stream);
    hipGraphExecDestroy(g->exec);
    hipGraphDestroy(g->graph);
    delete g;
}
int dav1d_hip_get_option(Dav1dHipContext *c, const char *name, long *value) {
    if (!c || !name || !value) return -EINVAL;
    if (!strcmp(name, "intra_sb_fallbacks")) *value = c->intra_sb_fallbacks;
    else if (!strcmp(name, "intra_sb_waves")) *value = c->intra_sb_waves;
    else if (!strcmp(name, "intra_sb_one_below")) *value = c->intra_sb_one_below;
    else if (!strcmp(name, "recon_fuse")) *value = c->recon_fuse;
    else if (!strcmp(name, "recon_pair_streams")) *value = c->recon_pair_streams;
    else if (!strcmp(name, "recon_pair_first")) *value = c->recon_pair_first;
    else if (!strcmp(name, "ref_twin")) *value = c->ref_twin;
    else return -EINVAL;
    return 0;
}
// knobs by name (the environment variables of DESIGN.md without the DAV1D_HIP_ prefix, lower case); -EINVAL for an unknown name
int dav1d_hip_set_option(Dav1dHipContext *c, const char *name, long value) {
    if (!c || !name) return -EINVAL;
    if (!strcmp(name, "recon_fuse")) c->recon_fuse = (int) value;
    else if (!strcmp(name, "recon_pipeline")) c->recon_pipeline = value;
    else if (!strcmp(name, "recon_lanes")) c->recon_lanes = (int) value;
    else if (!strcmp(name, "chunk_upload")) c->chunk_upload = (int) value;
    else if (!strcmp(name, "recon_coop_below")) c->recon_coop_below = (int) value;
    else if (!strcmp(name, "post_bands")) c->post_bands = (int) value;
    else if (!strcmp(name, "ref_twin")) c->ref_twin = (int) value;
    else if (!strcmp(name, "recon_pair_streams")) c->recon_pair_streams = (int) value;
    else if (!strcmp(name, "recon_pair_first")) c->recon_pair_first = (int) std::max(1L, std::min(3L, (long) value));
    else if (!strcmp(name, "serial")) c->concurrent = !value;
    else if (!strcmp(name, "cdef_unit")) c->cdef_unit_kernel = value != 0;
    else if (!strcmp(name, "filter_full_copy")) c->cdef_full_copy = value != 0;
    else if (!strcmp(name, "cdef_rows")) c->cdef_rows = value != 0;
    else if (!strcmp(name, "flow_groups")) c->flow_groups = value > 0 ? (int) value : c->flow_groups;
    else if (!strcmp(name, "flow_mode")) c->flow_mode = (int) value;
    else if (!strcmp(name, "flow_min_steps")) c->flow_min_steps = (int) value;
    else if (!strcmp(name, "intra_sb")) c->intra_sb = (int) value;
    else if (!strcmp(name, "intra_sb_waves")) c->intra_sb_waves = value >= 8 ? 8 : value >= 4 ? 4 : value == 1 ? 1 : 0;
    else if (!strcmp(name, "intra_sb_one_below")) c->intra_sb_one_below = value > 0 ? (int) value : 0;
    else if (!strcmp(name, "intra_sb_lds")) c->intra_sb_lds = value != 0;
    else if (!strcmp(name, "intra_sb_flow")) c->intra_sb_flow = value != 0;
    else if (!strcmp(name, "intra_sb_fine")) dav1d_hip_sbw_set_fine(value != 0);
    else if (!strcmp(name, "intra_sb_fail_at")) dav1d_hip_sbw_set_fail_at((int) value);
    else if (!strcmp(name, "prep_async")) c->prep_async = value < 0 ? 0 : value > 2 ? 2 : (int) value;
    else if (!strcmp(name, "chunk_order")) c->chunk_order = value != 0;
    else if (!strcmp(name, "chunk_hints")) c->chunk_hints = value != 0;
    else if (!strcmp(name, "chunk_arena_min")) { if (value < 4096) return -EINVAL; c->arena_min = (size_t) 1 << 12; while (c->arena_min < (size_t) value) c->arena_min <<= 1; c->arena_hint = 0; }
    else return -EINVAL;
    return 0;
}

const char *dav1d_hip_version(void) { return "dav1d_hip 0.1 (gfx950)"; }
float dav1d_hip_last_kernel_ms(Dav1dHipContext *c) { return c ? c->last_ms : 0.f; }

int dav1d_hip_malloc(Dav1dHipContext *c, void **dev, size_t bytes) {
    (void) c;
    return hip_rc(hipMalloc(dev, bytes ? bytes : 1));
}
int dav1d_hip_free(Dav1dHipContext *c, void *dev) {
    hipStreamSynchronize(c->stream);
    return hip_rc(hipFree(dev));
}
int dav1d_hip_memset(Dav1dHipContext *c, void *dev, int v, size_t bytes) {
    return hip_rc(hipMemsetAsync(dev, v, bytes, c->stream));
}
thread_local int dav1d_hip_tls_last_error = 0;
const char *dav1d_hip_last_hip_error(int *code) {
    const int e = dav1d_hip_tls_last_error;
    if (code) *code = e;
    return e ? hipGetErrorString((hipError_t) e) : "no error";
}

int dav1d_hip_upload(Dav1dHipContext *c, void *dev, const void *host, size_t bytes) {
    HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    return hip_rc(hipStreamSynchronize(c->stream));
}
int dav1d_hip_download(Dav1dHipContext *c, void *host, const void *dev, size_t bytes) {
    HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    return hip_rc(hipStreamSynchronize(c->stream));
}

// ----------------------------------------------------------------- pictures

int dav1d_hip_picture_alloc(Dav1dHipContext *c, Dav1dHipPicture *pic, int w, int h, int layout, int bpc) {
    if (!pic || w <= 0 || h <= 0 || (bpc != 8 && bpc != 10 && bpc != 12) || layout < 0 || layout > 3)
        return -EINVAL;
    // geometry of the reference's default allocator, src/picture.c:46-78
    const int hbd = bpc > 8;
    const int aligned_w = (w + 127) & ~127, aligned_h = (h + 127) & ~127;
    const int has_chroma = layout != DAV1D_HIP_LAYOUT_I400;
    const int ss_ver = layout == DAV1D_HIP_LAYOUT_I420;
    const int ss_hor = layout != DAV1D_HIP_LAYOUT_I444;
    ptrdiff_t y_stride = (ptrdiff_t) aligned_w << hbd;
    ptrdiff_t uv_stride = has_chroma ? y_stride >> ss_hor : 0;
    if (!(y_stride & 1023)) y_stride += 64;
    if (!(uv_stride & 1023) && has_chroma) uv_stride += 64;
    const size_t y_sz = (size_t) y_stride * aligned_h;
    const size_t uv_sz = (size_t) uv_stride * (aligned_h >> ss_ver);
    const size_t total = y_sz + 2 * uv_sz + 64;
    void *buf = nullptr;
    HIP_TRY(hipMalloc(&buf, total));
    HIP_TRY(hipMemsetAsync(buf, 0, total, c->stream));
    memset(pic, 0, sizeof(*pic));
    pic->alloc = buf;
    pic->alloc_size = total;
    pic->bpc = bpc;
    pic->layout = layout;
    pic->p[0].data = buf;
    pic->p[0].stride = y_stride;
    pic->p[0].w = w;
    pic->p[0].h = h;
    for (int i = 1; i < 3; i++) {
        pic->p[i].data = has_chroma ? (uint8_t *) buf + y_sz + (i - 1) * uv_sz : nullptr;
        pic->p[i].stride = uv_stride;
        pic->p[i].w = has_chroma ? (w + ss_hor) >> ss_hor : 0;
        pic->p[i].h = has_chroma ? (h + ss_ver) >> ss_ver : 0;
    }
    if (c->ref_twin >= 2) {
        const int rc = dav1d_hip_picture_twin_alloc(c, pic);
        if (rc) { (void) hipFree(buf); memset(pic, 0, sizeof(*pic)); return rc; }
    }
    return 0;
}

int dav1d_hip_picture_take(Dav1dHipContext *c, Dav1dHipPicture *pic, int w, int h, int layout, int bpc) {
    {
        std::lock_guard<std::mutex> lk(c->pool_mtx);
        for (size_t i = 0; i < c->free_pictures.size(); i++) {
            const Dav1dHipPicture &q = c->free_pictures[i];
            if (q.p[0].w == w && q.p[0].h == h && q.layout == layout && q.bpc == bpc && !q.twin_alloc == !(c->ref_twin >= 2)) {
                *pic = q;
                c->free_pictures[i] = c->free_pictures.back();
                c->free_pictures.pop_back();
                pic->twin_ok = 0;
                // as a fresh allocation would be: zero, padding included
                return hip_rc(hipMemsetAsync(pic->alloc, 0, pic->alloc_size, c->stream));
            }
        }
    }
    return dav1d_hip_picture_alloc(c, pic, w, h, layout, bpc);
}

void dav1d_hip_picture_give(Dav1dHipContext *c, Dav1dHipPicture *pic) {
    if (!pic->alloc) return;
    {
        std::lock_guard<std::mutex> lk(c->pool_mtx);
        if (c->free_pictures.size() < 16) { c->free_pictures.push_back(*pic); memset(pic, 0, sizeof(*pic)); return; }
    }
    (void) dav1d_hip_picture_free(c, pic);
}

extern "C" int dav1d_hip_launch_retile(const DevPlanes *src, void *const twin[3], int bpc, void *stream);

// Rows of plane pl that exist in memory: a plane of dav1d_hip_picture_alloc (and of dav1d's own allocator, src/picture.c:46-63) is padded to
// a multiple of 128 luma rows — blocks on the picture's bottom edge reconstruct into that padding — a caller-wrapped plane only promises its
// visible rows.
static inline int picture_plane_rows(const Dav1dHipPicture *pic, int pl, bool padded) {
    if (!padded) return pic->p[pl].h;
    const int ss_ver = pic->layout == DAV1D_HIP_LAYOUT_I420;
    const int ah = (pic->p[0].h + 127) & ~127;
    return pl ? ah >> ss_ver : ah;
}
// the picture's planes for the retile / untile passes: every row the allocation holds when the library made it
static inline DevPlanes twin_pass_planes(const Dav1dHipPicture *pic) {
    DevPlanes d = dev_planes(pic);
    for (int pl = 0; pl < 3; pl++) if (pic->p[pl].data) d.h[pl] = picture_plane_rows(pic, pl, pic->alloc != nullptr);
    return d;
}

// Storage for the tiled twin: per plane stride x (rows padded as dav1d's allocator pads them: blocks on the bottom edge write below the
// visible rows, in the twin as in the raster plane) bytes, the planes one after the other.
int dav1d_hip_picture_twin_alloc(Dav1dHipContext *c, Dav1dHipPicture *pic) {
    if (!c || !pic || !pic->p[0].data) return -EINVAL;
    if (pic->twin_alloc) return 0;
    const int bps = pic->bpc > 8 ? 2 : 1;
    size_t off[3] = { 0, 0, 0 }, total = 0;
    for (int i = 0; i < 3; i++) {
        if (!pic->p[i].data) continue;
        if (pic->p[i].stride <= 0 || (pic->p[i].stride / bps) % 8 || pic->p[i].stride % 16) return -EINVAL;
        off[i] = total;
        total += (size_t) pic->p[i].stride * (size_t) picture_plane_rows(pic, i, true);
        total = (total + 255) & ~(size_t) 255;
    }
    void *buf = nullptr;
    HIP_TRY(hipMalloc(&buf, total + 256));
    HIP_TRY(hipMemsetAsync(buf, 0, total + 256, c->stream));
    pic->twin_alloc = buf;
    for (int i = 0; i < 3; i++) pic->twin[i] = pic->p[i].data ? (uint8_t *) buf + off[i] : nullptr;
    pic->twin_ok = 0;
    return 0;
}

extern "C" int dav1d_hip_launch_untile(const DevPlanes *dst, void *const twin[3], int bpc, const int row0[3], const int row1[3], int plane_mask, void *stream);
extern "C" int dav1d_hip_launch_retile(const DevPlanes *src, void *const twin[3], int bpc, void *stream);

// The same on a side stream of the context: the copy starts when the work enqueued so far is through and runs NEXT TO whatever the
// caller enqueues afterwards (the next frame's launches: they are bound by request latency and arithmetic, the copy by bandwidth).
// Launches of this context that read twins wait for it (ref_planes); dav1d_hip_sync does too.
int dav1d_hip_picture_retile_overlapped(Dav1dHipContext *c, Dav1dHipPicture *pic) {
    if (!c || !pic) return -EINVAL;
    if (!c->concurrent) return dav1d_hip_picture_retile(c, pic);
    if (!pic->twin_alloc && !pic->twin[0]) {
        const int rc = dav1d_hip_picture_twin_alloc(c, pic);
        if (rc) return rc;
    }
    hipStream_t side = c->side[Dav1dHipContext::N_SIDE - 1];
    HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
    HIP_TRY(hipStreamWaitEvent(side, c->ev_fork, 0));
    if (pic->twin_ok == DAV1D_HIP_TWIN_ONLY) return 0;          // the twin IS the picture
    const DevPlanes sp = twin_pass_planes(pic);
    const int rc = dav1d_hip_launch_retile(&sp, pic->twin, pic->bpc, side);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->ev_retile, side));
    c->retile_pending = true;
    pic->twin_ok = 1;
    return 0;
}

int dav1d_hip_picture_retile(Dav1dHipContext *c, Dav1dHipPicture *pic) {
    if (!c || !pic) return -EINVAL;
    if (!pic->twin_alloc && !pic->twin[0]) {
        const int rc = dav1d_hip_picture_twin_alloc(c, pic);
        if (rc) return rc;
    }
    if (pic->twin_ok == DAV1D_HIP_TWIN_ONLY) return 0;          // the twin IS the picture
    const DevPlanes sp = twin_pass_planes(pic);
    const int rc = dav1d_hip_launch_retile(&sp, pic->twin, pic->bpc, c->stream);
    if (!rc) pic->twin_ok = 1;
    return rc;
}

// The other way: a picture that lives in its twin only (twin_ok == DAV1D_HIP_TWIN_ONLY: what dav1d_hip_recon_list_run_tiled leaves) gets
// its raster planes back, on the context's stream; twin_ok becomes 1 (both valid).  No-op for any other picture.
int dav1d_hip_picture_untile(Dav1dHipContext *c, Dav1dHipPicture *pic) {
    if (!c || !pic) return -EINVAL;
    if (pic->twin_ok != DAV1D_HIP_TWIN_ONLY) return 0;
    if (!pic->twin[0]) return -EINVAL;
    const DevPlanes dp = twin_pass_planes(pic);
    const int rc = dav1d_hip_launch_untile(&dp, pic->twin, pic->bpc, nullptr, nullptr, 7, c->stream);
    if (!rc) pic->twin_ok = 1;
    return rc;
}

// Host side of a Dav1dPicAllocator: pinned planes with the geometry of the device picture (= the reference's default allocator,
// src/picture.c:46-82)
int dav1d_hip_host_picture_alloc(Dav1dHipContext *c, Dav1dHipHostPicture *hp, int w, int h, int layout, int bpc) {
    if (!c || !hp) return -EINVAL;
    memset(hp, 0, sizeof(*hp));
    const int rc = dav1d_hip_picture_alloc(c, &hp->dev, w, h, layout, bpc);
    if (rc) return rc;
    void *buf = nullptr;
    if (hipHostMalloc(&buf, hp->dev.alloc_size, hipHostMallocDefault) != hipSuccess) {
        (void) dav1d_hip_picture_free(c, &hp->dev);
        memset(hp, 0, sizeof(*hp));
        return -ENOMEM;
    }
    hp->alloc = buf;
    hp->alloc_size = hp->dev.alloc_size;
    for (int i = 0; i < 3; i++)
        hp->data[i] = hp->dev.p[i].data ? (uint8_t *) buf + ((const uint8_t *) hp->dev.p[i].data - (const uint8_t *) hp->dev.alloc) : nullptr;
    hp->stride[0] = hp->dev.p[0].stride;
    hp->stride[1] = hp->dev.p[1].stride;
    __atomic_fetch_add(&dav1d_hip_live[3], 1, __ATOMIC_RELAXED);
    return 0;
}

int dav1d_hip_host_picture_release(Dav1dHipContext *c, Dav1dHipHostPicture *hp) {
    if (!c || !hp) return -EINVAL;
    (void) hipStreamSynchronize(c->copy_stream);
    int rc = 0;
    if (hp->alloc) { rc = hip_rc(hipHostFree(hp->alloc)); __atomic_fetch_sub(&dav1d_hip_live[3], 1, __ATOMIC_RELAXED); }
    const int rc2 = dav1d_hip_picture_free(c, &hp->dev);
    memset(hp, 0, sizeof(*hp));
    return rc ? rc : rc2;
}

int dav1d_hip_host_picture_fetch(Dav1dHipContext *c, const Dav1dHipHostPicture *hp, const Dav1dHipPicture *src, int row0, int row1) {
    if (!c || !hp || !hp->alloc) return -EINVAL;
    if (!src) src = &hp->dev;
    if (src->bpc != hp->dev.bpc || src->layout != hp->dev.layout || src->p[0].w != hp->dev.p[0].w || src->p[0].h != hp->dev.p[0].h) return -EINVAL;
    if (row0 < 0) row0 = 0;
    if (row1 > src->p[0].h) row1 = src->p[0].h;
    if (row1 <= row0) return 0;
    const int ss_ver = src->layout == DAV1D_HIP_LAYOUT_I420, bps = src->bpc > 8 ? 2 : 1;
    if (src->twin_ok == DAV1D_HIP_TWIN_ONLY) {
        // the picture lives in its twin: the rows of this band become raster rows here, on their way out (the raster planes are the
        // staging; src is const, so the picture stays DAV1D_HIP_TWIN_ONLY and a later band / fetch does its own rows again)
        if (!src->twin[0]) return -EINVAL;
        int r0[3], r1[3];
        for (int pl = 0; pl < 3; pl++) {
            const int sv = pl ? ss_ver : 0;
            r0[pl] = row0 >> sv; r1[pl] = row1 >= src->p[0].h ? src->p[pl].h : row1 >> sv;
        }
        const DevPlanes dp = dev_planes(src);
        int rc = dav1d_hip_launch_untile(&dp, src->twin, src->bpc, r0, r1, 7, c->stream);
        if (!rc) rc = hip_rc(hipEventRecord(c->ev_untile, c->stream));
        if (!rc) rc = hip_rc(hipStreamWaitEvent(c->copy_stream, c->ev_untile, 0));
        if (rc) return rc;
    }
    for (int pl = 0; pl < 3; pl++) {
        if (!src->p[pl].data || !hp->data[pl]) continue;
        const int sv = pl ? ss_ver : 0;
        // chroma rows under luma rows [row0, row1): a band boundary is even, the last band ends with the picture
        const int r0 = row0 >> sv, r1 = row1 >= src->p[0].h ? src->p[pl].h : row1 >> sv;
        if (r1 <= r0) continue;
        const ptrdiff_t hs = hp->stride[pl ? 1 : 0];
        const hipError_t e = hipMemcpy2DAsync((uint8_t *) hp->data[pl] + (size_t) r0 * hs, hs,
                                              (const uint8_t *) src->p[pl].data + (size_t) r0 * src->p[pl].stride, src->p[pl].stride,
                                              (size_t) src->p[pl].w * bps, r1 - r0, hipMemcpyDeviceToHost, c->copy_stream);
        if (e != hipSuccess) return hip_rc(e);
    }
    return 0;
}

int dav1d_hip_host_picture_wait(Dav1dHipContext *c) {
    if (!c) return -EINVAL;
    return hip_rc(hipStreamSynchronize(c->copy_stream));
}

int dav1d_hip_picture_free(Dav1dHipContext *c, Dav1dHipPicture *pic) {
    if (!pic || (!pic->alloc && !pic->twin_alloc)) return 0;
    if (c) hipStreamSynchronize(c->stream); else (void) hipDeviceSynchronize();        // (a picture that outlived its context)
    int rc = pic->alloc ? hip_rc(hipFree(pic->alloc)) : 0;
    if (pic->twin_alloc) { const int rc2 = hip_rc(hipFree(pic->twin_alloc)); if (!rc) rc = rc2; }
    memset(pic, 0, sizeof(*pic));
    return rc;
}

// ---- more than one device in a process (dav1d is ONE process with n_fc frame contexts: the binding ends frame context k's frames on device
// k mod N, dav1d_amd/host/dav1d_glue.c).  The current device is a property of the calling THREAD in HIP: a thread that serves contexts of
// several devices says which one it means before it calls in.
int dav1d_hip_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : -ENODEV;
}
int dav1d_hip_context_device(const Dav1dHipContext *c) { return c ? c->device : -EINVAL; }
int dav1d_hip_context_use(Dav1dHipContext *c) {
    if (!c) return -EINVAL;
    return hipSetDevice(c->device) == hipSuccess ? 0 : -ENODEV;
}
// direct copies between the devices of two contexts (xGMI): without it hipMemcpyPeerAsync goes through host memory.  Both directions; a pair
// that is enabled already or cannot be peers is not an error (the copy still works, staged).  Leaves the caller's device current.
int dav1d_hip_enable_peer_access(Dav1dHipContext *a, Dav1dHipContext *b) {
    if (!a || !b) return -EINVAL;
    if (a->device == b->device) return 0;
    int prev = 0, enabled = 0;
    (void) hipGetDevice(&prev);
    const int dev[2] = { a->device, b->device };
    for (int k = 0; k < 2; k++) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dev[k], dev[k ^ 1]) != hipSuccess || !can) continue;
        if (hipSetDevice(dev[k]) != hipSuccess) continue;
        const hipError_t e = hipDeviceEnablePeerAccess(dev[k ^ 1], 0);
        if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) enabled++;
    }
    (void) hipGetLastError();
    (void) hipSetDevice(prev);
    return enabled;         // 0, 1 or 2 directions
}
// for callers that borrow a thread (an allocator callback on the application's thread): what the thread had, and back to it
int dav1d_hip_current_device(void) {
    int d = 0;
    return hipGetDevice(&d) == hipSuccess ? d : -ENODEV;
}
int dav1d_hip_set_device(int device) { return hipSetDevice(device) == hipSuccess ? 0 : -ENODEV; }
// the device a picture's planes live on (-EINVAL: not device memory the runtime knows)
int dav1d_hip_picture_device(const Dav1dHipPicture *pic) {
    if (!pic) return -EINVAL;
    const void *p = pic->twin_ok == DAV1D_HIP_TWIN_ONLY && pic->twin[0] ? pic->twin[0] : pic->p[0].data;
    hipPointerAttribute_t a;
    if (!p || hipPointerGetAttributes(&a, p) != hipSuccess || a.type != hipMemoryTypeDevice) { (void) hipGetLastError(); return -EINVAL; }
    return a.device;
}
// with more than one device: are these pictures where context c can launch on them?  (-EXDEV names the first that is not)
// (asked only when this process has opened contexts on more than one device: the question is a trip to the driver per picture, on the
// frame's critical path, and a process that uses one of a node's eight GPUs cannot have got a picture from another)
static std::atomic<uint64_t> g_ctx_devices{0};
void dav1d_hip_note_context_device(int device) { if (device >= 0 && device < 64) g_ctx_devices.fetch_or(1ull << device); }
int pictures_on_device(const Dav1dHipContext *c, const Dav1dHipPicture *pics, int n) {
    const uint64_t m = g_ctx_devices.load(std::memory_order_relaxed);
    if (!(m & (m - 1))) return 0;
    for (int i = 0; i < n; i++) {
        if (!pics[i].p[0].data) continue;
        const int d = dav1d_hip_picture_device(&pics[i]);
        if (d >= 0 && d != c->device) return -EXDEV;
    }
    return 0;
}
// `dst` (a picture of dst_c's device with src's geometry: dav1d_hip_picture_alloc under the same ref_twin option) becomes a copy of `src`
// (src_c's device): the raster planes unless src lives in its twin only, the twin when src has a valid one and dst the storage.  The copy is
// enqueued on dst_c's stream behind everything src_c's stream holds now (an event across the devices), over xGMI when the devices are peers
// (hipMemcpyPeerAsync stages through the host when they are not): a launch of dst_c that follows reads the copy.
int dav1d_hip_picture_copy_peer(Dav1dHipContext *dst_c, Dav1dHipPicture *dst, Dav1dHipContext *src_c, const Dav1dHipPicture *src) {
    if (!dst_c || !dst || !src_c || !src || !dst->p[0].data || !src->p[0].data) return -EINVAL;
    if (dst->bpc != src->bpc || dst->layout != src->layout) return -EINVAL;
    for (int pl = 0; pl < 3; pl++)
        if (dst->p[pl].w != src->p[pl].w || dst->p[pl].h != src->p[pl].h || dst->p[pl].stride != src->p[pl].stride || !dst->p[pl].data != !src->p[pl].data) return -EINVAL;
    const bool twin = src->twin_ok && src->twin[0] && dst->twin[0];
    if (src->twin_ok == DAV1D_HIP_TWIN_ONLY && !twin) return -EINVAL;
    // behind the source's work
    // (an event of this call's own: several devices may be copying from one source at a time, and src_c's thread goes on enqueuing.  A twin
    // made by dav1d_hip_picture_retile_overlapped is on a side stream: its maker waits for it — dav1d_hip_sync — before handing it out.)
    hipEvent_t ev;
    if (hipSetDevice(src_c->device) != hipSuccess) return -ENODEV;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, src_c->stream);
    if (hipSetDevice(dst_c->device) != hipSuccess) { (void) hipEventDestroy(ev); return -ENODEV; }
    if (e == hipSuccess) e = hipStreamWaitEvent(dst_c->stream, ev, 0);
    (void) hipEventDestroy(ev);             // (released once the wait has passed it)
    HIP_TRY(e);
    const bool padded = src->alloc != nullptr && dst->alloc != nullptr;
    for (int pl = 0; pl < 3; pl++) {
        if (!src->p[pl].data) continue;
        const size_t bytes = (size_t) src->p[pl].stride * (size_t) picture_plane_rows(src, pl, padded);
        if (src->twin_ok != DAV1D_HIP_TWIN_ONLY)
            HIP_TRY(hipMemcpyPeerAsync(dst->p[pl].data, dst_c->device, src->p[pl].data, src_c->device, bytes, dst_c->stream));
        if (twin)
            HIP_TRY(hipMemcpyPeerAsync(dst->twin[pl], dst_c->device, src->twin[pl], src_c->device, bytes, dst_c->stream));
    }
    dst->twin_ok = twin ? src->twin_ok : 0;
    return 0;
}

// Luma rows [y0, y1) of src's RASTER planes (the chroma rows under them) to dst on another device, on dst_c's stream.  The caller says the rows are
// final on the source device (dav1d_hip_frame_set_progress_callback reported them): nothing of src_c's stream is waited for, so the bands of a
// picture can cross while the frame that makes it is still ending (dav1d_glue.c).  y0 a multiple of 8; dst's twin is stale afterwards.
int dav1d_hip_picture_copy_peer_rows(Dav1dHipContext *dst_c, Dav1dHipPicture *dst, Dav1dHipContext *src_c, const Dav1dHipPicture *src, int y0, int y1) {
    if (!dst_c || !dst || !src_c || !src || !dst->p[0].data || !src->p[0].data) return -EINVAL;
    if (dst->bpc != src->bpc || dst->layout != src->layout || src->twin_ok == DAV1D_HIP_TWIN_ONLY) return -EINVAL;
    for (int pl = 0; pl < 3; pl++)
        if (dst->p[pl].w != src->p[pl].w || dst->p[pl].h != src->p[pl].h || dst->p[pl].stride != src->p[pl].stride || !dst->p[pl].data != !src->p[pl].data) return -EINVAL;
    if (y0 < 0 || (y0 & 7) || y1 < y0 || y1 > src->p[0].h) return -EINVAL;
    if (y1 == y0) return 0;
    if (hipSetDevice(dst_c->device) != hipSuccess) return -ENODEV;
    const bool padded = src->alloc != nullptr && dst->alloc != nullptr;
    const bool last = y1 == src->p[0].h;           // (the padding rows below the picture travel with its last band, as dav1d_hip_picture_copy_peer sends them)
    const int ss_ver = src->layout == DAV1D_HIP_LAYOUT_I420;
    for (int pl = 0; pl < 3; pl++) {
        if (!src->p[pl].data) continue;
        const int sv = pl ? ss_ver : 0;
        const int r0 = y0 >> sv, r1 = last ? picture_plane_rows(src, pl, padded) : (y1 + sv) >> sv;
        if (r1 <= r0) continue;
        const size_t off = (size_t) src->p[pl].stride * (size_t) r0, bytes = (size_t) src->p[pl].stride * (size_t) (r1 - r0);
        HIP_TRY(hipMemcpyPeerAsync((char *) dst->p[pl].data + off, dst_c->device, (const char *) src->p[pl].data + off, src_c->device, bytes, dst_c->stream));
    }
    dst->twin_ok = 0;
    return 0;
}

static void plane_extent(const Dav1dHipPicture *pic, int plane, int padded, size_t *row_bytes, int *rows) {
    const int bps = pic->bpc > 8 ? 2 : 1;
    if (padded) {
        const int ss_ver = plane && pic->layout == DAV1D_HIP_LAYOUT_I420;
        const int ss_hor = plane && pic->layout != DAV1D_HIP_LAYOUT_I444;
        const int aw = ((pic->p[0].w + 127) & ~127) >> ss_hor, ah = ((pic->p[0].h + 127) & ~127) >> ss_ver;
        *row_bytes = (size_t) aw * bps;
        *rows = ah;
    } else {
        *row_bytes = (size_t) pic->p[plane].w * bps;
        *rows = pic->p[plane].h;
    }
}

int dav1d_hip_plane_upload(Dav1dHipContext *c, const Dav1dHipPicture *pic, int plane,
                           const void *host, ptrdiff_t host_stride, int padded) {
    if (!pic || plane < 0 || plane > 2 || !pic->p[plane].data) return -EINVAL;
    size_t rb; int rows;
    plane_extent(pic, plane, padded, &rb, &rows);
    HIP_TRY(hipMemcpy2DAsync(pic->p[plane].data, pic->p[plane].stride, host, host_stride, rb, rows,
                             hipMemcpyHostToDevice, c->stream));
    return hip_rc(hipStreamSynchronize(c->stream));
}

int dav1d_hip_plane_download(Dav1dHipContext *c, const Dav1dHipPicture *pic, int plane,
                             void *host, ptrdiff_t host_stride, int padded) {
    if (!pic || plane < 0 || plane > 2 || !pic->p[plane].data) return -EINVAL;
    size_t rb; int rows;
    plane_extent(pic, plane, padded, &rb, &rows);
    if (pic->twin_ok == DAV1D_HIP_TWIN_ONLY) {
        // the picture lives in its twin: this plane's raster rows are made here (pic is const: the flag stays, the next call does it again)
        if (!pic->twin[plane]) return -EINVAL;
        const DevPlanes dp = twin_pass_planes(pic);
        const int rc = dav1d_hip_launch_untile(&dp, pic->twin, pic->bpc, nullptr, nullptr, 1 << plane, c->stream);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpy2DAsync(host, host_stride, pic->p[plane].data, pic->p[plane].stride, rb, rows,
                             hipMemcpyDeviceToHost, c->stream));
    return hip_rc(hipStreamSynchronize(c->stream));
}

} // extern "C"

int dav1d_hip_scratch(Dav1dHipContext *c, size_t bytes, void **out) {
    if (c->scratch_size < bytes) {
        hipStreamSynchronize(c->stream);
        if (c->scratch) hipFree(c->scratch);
        c->scratch = nullptr;
        c->scratch_size = 0;
        size_t sz = bytes < (1u << 20) ? (1u << 20) : bytes;
        HIP_TRY(hipMalloc(&c->scratch, sz));
        c->scratch_size = sz;
    }
    *out = c->scratch;
    return 0;
}

Dav1dHipContext *dav1d_hip_default_context(void) {
    static std::mutex mtx;
    static Dav1dHipContext *g = nullptr;
    std::lock_guard<std::mutex> lk(mtx);
    if (!g) {
        const char *e = getenv("DAV1D_HIP_DEVICE");
        if (dav1d_hip_open(&g, e ? atoi(e) : 0, nullptr)) g = nullptr;
    }
    return g;
}

// ---------------------------------------------------------------------- itx

static const uint8_t k_tx_w[19] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
static const uint8_t k_tx_h[19] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };


// legal (size, type) pairs, reference src/itx_tmpl.c:160-178 / 264-291
static bool itx_legal(int tx, int txtp) {
    if (tx < 0 || tx >= 19 || txtp < 0 || txtp > 16) return false;
    if (txtp == 16) return tx == 0;
    const int w = k_tx_w[tx], h = k_tx_h[tx];
    const int mx = w > h ? w : h;
    if (mx == 64) return txtp == 0;
    if (mx == 32) return txtp == 0 || txtp == 9;
    if (w == 16 && h == 16) return txtp <= 11;
    return true;
}

extern "C" {

} // extern "C"

bool itx_task_ok(const Dav1dHipItxTask &t) {
    if (!itx_legal(t.tx, t.txtp) || t.plane > 2 || t.eob < 0 || t.flags > DAV1D_HIP_ITX_PACKED) return false;
    return t.eob < av1_scan_prefix_off[t.tx + 1] - av1_scan_prefix_off[t.tx];
}

// How much of the slab can be non-zero: coefficients past the eob in scan order are zero by contract (the entropy
// decoder only writes scan positions <= eob, src/recon_tmpl.c:458-520, and itx leaves slabs zeroed), so the kernel
// reads and re-zeroes only the prefix [0, end).  2-D classes: the zig-zag's reach; H classes: the scan is the
// slab order itself; V classes: every column can be touched.  The device copy carries `end` in the pad bytes.
void itx_fill_prefix(Dav1dHipItxTask &t) {
    const int ncoef = av1_scan_prefix_off[t.tx + 1] - av1_scan_prefix_off[t.tx];
    int end = ncoef;
    if (t.txtp <= 9 || t.txtp == 16) end = av1_scan_prefix_end[av1_scan_prefix_off[t.tx] + t.eob];
    else if (t.txtp == 11 || t.txtp == 13 || t.txtp == 15) end = t.eob + 1;
    t.rsv[0] = (uint8_t) (end & 255);
    t.rsv[1] = (uint8_t) (end >> 8);
}

// code path of a transform block: dc-only shortcut, else its two 1-D kinds (txtp_kinds() in itx_body.h); 16 = WHT
int itx_path_key(const Dav1dHipItxTask &t) {
    static const uint8_t kinds[17] = {
        0 | 0 << 2, 0 | 1 << 2, 1 | 0 << 2, 1 | 1 << 2, 0 | 3 << 2, 3 | 0 << 2, 3 | 3 << 2, 3 | 1 << 2,
        1 | 3 << 2, 2 | 2 << 2, 2 | 0 << 2, 0 | 2 << 2, 2 | 1 << 2, 1 | 2 << 2, 2 | 3 << 2, 3 | 2 << 2, 16 };
    return t.txtp == 0 && t.eob < 1 ? 0 : 1 + kinds[t.txtp];
}

extern "C" {

int dav1d_hip_itx_list_create(Dav1dHipContext *c, Dav1dHipItxList **out, const Dav1dHipItxTask *tasks, size_t n) {
    if (!out || (!tasks && n)) return -EINVAL;
    *out = nullptr;
    Dav1dHipItxList *l = new (std::nothrow) Dav1dHipItxList();
    if (!l) return -ENOMEM;
    memset(l, 0, sizeof(*l));
    l->n = n;
    size_t cnt[19] = { 0 };
    for (size_t i = 0; i < n; i++) {
        const Dav1dHipItxTask &t = tasks[i];
        if (!itx_task_ok(t)) { delete l; return -EINVAL; }
        cnt[t.tx]++;
    }
    for (int b = 0; b < 19; b++) l->off[b + 1] = l->off[b] + cnt[b];
    if (n) {
        std::vector<Dav1dHipItxTask> sorted(n);
        size_t pos[19];
        for (int b = 0; b < 19; b++) pos[b] = l->off[b];
        for (size_t i = 0; i < n; i++) {
            Dav1dHipItxTask &t = sorted[pos[tasks[i].tx]++] = tasks[i];          // stable: keeps decode order inside a bin
            itx_fill_prefix(t);
        }
        // Blocks that share a wave should share their code path: a wave runs every 1-D kernel (and the dc-only shortcut) that
        // any of its blocks needs, one after the other.  Inside windows of consecutive blocks (still close together in the
        // picture, so the destination lines stay in L2) the blocks are grouped by (dc-only, first kind, second kind).  The
        // blocks of one list write disjoint pixels, so their order is free.  Speed only.
        static const int win_waves = getenv("DAV1D_HIP_ITX_SORT_WINDOW") ? atoi(getenv("DAV1D_HIP_ITX_SORT_WINDOW")) : 128;
        if (win_waves > 0) {
            auto key = [](const Dav1dHipItxTask &t) -> int { return itx_path_key(t); };
            for (int b = 0; b < 19; b++) {
                const int w = k_tx_w[b], h = k_tx_h[b];
                const int lanes = std::max(std::min(h, 32), w);
                const size_t win = (size_t) win_waves * (size_t) std::max(1, 64 / lanes);
                for (size_t lo = l->off[b]; lo < l->off[b + 1]; lo += win) {
                    const size_t hi = std::min(lo + win, l->off[b + 1]);
                    std::stable_sort(sorted.begin() + lo, sorted.begin() + hi,
                                     [&](const Dav1dHipItxTask &p, const Dav1dHipItxTask &q) { return key(p) < key(q); });
                }
            }
        }
        if (hipMalloc((void **) &l->dev, n * sizeof(Dav1dHipItxTask)) != hipSuccess) { delete l; return -ENOMEM; }
        const int rc = dav1d_hip_upload(c, l->dev, sorted.data(), n * sizeof(Dav1dHipItxTask));
        if (rc) { hipFree(l->dev); delete l; return rc; }
    }
    *out = l;
    return 0;
}

void dav1d_hip_itx_list_destroy(Dav1dHipContext *c, Dav1dHipItxList *l) {
    if (!l) return;
    hipStreamSynchronize(c->stream);
    if (l->dev) hipFree(l->dev);
    delete l;
}

int dav1d_hip_itx_list_run(Dav1dHipContext *c, const Dav1dHipItxList *l, const Dav1dHipPicture *dst, void *coef) {
    if (!l || !dst) return -EINVAL;
    const DevPlanes dp = dev_planes(dst);
    // longest-running shapes first (64-point, then 32-point ...), each on its own side stream
    static const uint8_t order[19] = { 4, 11, 12, 17, 18, 3, 9, 10, 15, 16, 2, 7, 8, 13, 14, 1, 5, 6, 0 };
    // short lists (the residuals of one intra wavefront step): every size in one launch
    static const size_t one_launch_below = getenv("DAV1D_HIP_ITX_ONE_LAUNCH") ? (size_t) atol(getenv("DAV1D_HIP_ITX_ONE_LAUNCH")) : 4096;
    if (l->n && l->n < one_launch_below) return dav1d_hip_launch_itx_all(&dp, dst->bpc, l->dev, l->off, coef, c->stream);
    StreamFan fan(c, l->n >= 16384);
    int rc = 0;
    for (int k = 0; k < 19 && !rc; k++) {
        const int b = order[k];
        const size_t cnt = l->off[b + 1] - l->off[b];
        if (!cnt) continue;
        rc = dav1d_hip_launch_itx_bin(&dp, dst->bpc, b, l->dev + l->off[b], (int) cnt, coef, fan.next());
    }
    fan.join();
    return rc;
}

// Same launches, each bracketed by HIP events on the context's stream; ms[b] receives the
// duration of bin b's kernel (0 for empty bins).  Measurement aid for bench.py.
int dav1d_hip_itx_list_run_timed(Dav1dHipContext *c, const Dav1dHipItxList *l, const Dav1dHipPicture *dst, void *coef,
                                 float *ms, size_t *counts) {
    if (!l || !dst || !ms) return -EINVAL;
    const DevPlanes dp = dev_planes(dst);
    hipEvent_t ev[20];
    for (int b = 0; b < 20; b++) HIP_TRY(hipEventCreate(&ev[b]));
    HIP_TRY(hipEventRecord(ev[0], c->stream));
    int rc = 0;
    for (int b = 0; b < 19 && !rc; b++) {
        const size_t cnt = l->off[b + 1] - l->off[b];
        if (counts) counts[b] = cnt;
        if (cnt) rc = dav1d_hip_launch_itx_bin(&dp, dst->bpc, b, l->dev + l->off[b], (int) cnt, coef, c->stream);
        hipEventRecord(ev[b + 1], c->stream);
    }
    hipStreamSynchronize(c->stream);
    for (int b = 0; b < 19; b++) { ms[b] = 0.f; hipEventElapsedTime(&ms[b], ev[b], ev[b + 1]); }
    for (int b = 0; b < 20; b++) hipEventDestroy(ev[b]);
    return rc;
}

int dav1d_hip_itx_add_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipItxTask *tasks,
                            size_t n, void *coef) {
    Dav1dHipItxList *l = nullptr;
    int rc = dav1d_hip_itx_list_create(c, &l, tasks, n);
    if (rc) return rc;
    KernelTimer kt(c);
    rc = dav1d_hip_itx_list_run(c, l, dst, coef);
    kt.stop();
    dav1d_hip_itx_list_destroy(c, l);     // synchronises the stream
    return rc;
}

} // extern "C"

// ----------------------------------------------------------------------- mc


// DAV1D_HIP_MC_FUSED: which tile shapes share one launch over a source-ordered list instead of one launch per shape.
//   0  none;  2  the shapes that are at least 16 wide (bins 6 .. 14);  1  all of them.
// Measured on MI355X (8K 10-bit synthetic frame): mode 1 cuts the fetch traffic by a third (lines are shared across
// shapes while they sit in L2) but runs 10 % slower than per-shape launches, because every wave then pays the LDS / VGPR
// footprint of the hungriest (small-tile) shape.
static int mc_fused_min_bin() {
    static const int mode = getenv("DAV1D_HIP_MC_FUSED") ? atoi(getenv("DAV1D_HIP_MC_FUSED")) : 0;
    return mode == 1 ? 0 : mode == 2 ? 6 : 15;
}

int tile_dim_class(int v) { return v <= 4 ? 0 : v <= 8 ? 1 : v <= 16 ? 2 : v <= 32 ? 3 : 4; }

extern "C" {

} // extern "C" (helpers below are C++)

int mc_task_valid(const Dav1dHipMcTask &t) {
    // widths are powers of two; heights too, except the 3/4-height `lap` predictions of obmc() (6, 12, 24 rows)
    return !(t.w < 2 || t.w > 128 || t.h < 2 || t.h > 128 || (t.w & (t.w - 1)) || (t.h & 1) ||
             t.mx > 15 || t.my > 15 || t.filter_2d > 9 || t.kind > 2 || t.plane > 2 || t.ref > 7);
}

McRef mc_ref_of(const Dav1dHipMcTask &t) {
    McRef r;
    memset(&r, 0, sizeof(r));
    r.src_x = t.src_x; r.src_y = t.src_y;
    r.mx = t.mx; r.my = t.my; r.ref = t.ref;
    if (t.filter_2d == 9) {
        r.fh = r.fv = 6;
    } else {
        // enum Filter2d -> (h type, v type) with REGULAR 0, SMOOTH 1, SHARP 2 (reference src/levels.h:184-196,
        // src/mc_tmpl.c:395-403); 4-tap rows for w <= 4 / h <= 4 (src/mc_tmpl.c:115-123)
        static const uint8_t ht[9] = { 0, 0, 0, 2, 2, 2, 1, 1, 1 };
        static const uint8_t vt[9] = { 0, 1, 2, 0, 1, 2, 0, 1, 2 };
        const int h_type = ht[t.filter_2d], v_type = vt[t.filter_2d];
        r.fh = t.w > 4 ? h_type : 3 + (h_type & 1);
        r.fv = t.h > 4 ? v_type : 3 + (v_type & 1);
    }
    r.vspan = av1_mc_tap_span_host[r.fv * 16 + r.my];
    r.hspan = av1_mc_tap_span_host[r.fh * 16 + r.mx];
    return r;
}

// cut one prediction block (or a fused pair) into <= 64x16 tiles and bin them by tile shape
void push_tiles(std::vector<McTile> *bins, const Dav1dHipMcTask &t, int kind, uint32_t dst_off,
                       const Dav1dHipMcTask *second, int weight, std::vector<McTile> *single) {
    McTile m;
    memset(&m, 0, sizeof(m));
    m.dst_off = dst_off;
    m.kind = kind; m.plane = t.plane; m.bw = t.w; m.weight = (int8_t) weight;
    const McRef r0 = mc_ref_of(t), r1 = second ? mc_ref_of(*second) : r0;
    const int tw = t.w < 64 ? t.w : 64, th = t.h < 16 ? t.h : 16;   // strips of one block, up to 64x16
    const int cls = tile_dim_class(tw) * 3 + tile_dim_class(th);
    for (int oy = 0; oy < t.h; oy += th)
        for (int ox = 0; ox < t.w; ox += tw) {
            m.w = tw; m.h = std::min(th, t.h - oy); m.ox = ox; m.oy = oy;
            m.r[0] = r0; m.r[0].src_x += ox; m.r[0].src_y += oy;
            m.r[1] = r1; m.r[1].src_x += ox; m.r[1].src_y += oy;
            if (single) single->push_back(m); else bins[cls].push_back(m);
        }
}

static int mc_list_from_bins(Dav1dHipContext *c, Dav1dHipMcList **out, std::vector<McTile> *bins) {
    Dav1dHipMcList *l = new (std::nothrow) Dav1dHipMcList();
    if (!l) return -ENOMEM;
    memset(l, 0, sizeof(*l));
    // Order the tiles of a bin by where they READ: (reference, plane, 64-row band, x).  The fetch of a tile is
    // row-granular (a 128-byte line per window row), so tiles that land on the same lines should run back to back on
    // one XCD while those lines sit in its L2; dst writes stay local because MVs are short.  Speed only.
    static const int sort_mode = getenv("DAV1D_HIP_MC_SORT") ? atoi(getenv("DAV1D_HIP_MC_SORT")) : 1;
    if (sort_mode) {
        auto key = [](const McTile &t) -> uint64_t {
            const McRef &r = t.r[0];
            const uint64_t y = (uint64_t) (r.src_y + 4096) & 0xffff, x = (uint64_t) (r.src_x + 4096) & 0xffff;
            if (sort_mode == 2) return ((uint64_t) r.ref << 56) | ((uint64_t) t.plane << 52) | ((x >> 9) << 40) | (y << 16) | x;
            if (sort_mode == 3) return ((uint64_t) t.plane << 52) | ((y >> 6) << 32) | (x << 8) | r.ref;
            return ((uint64_t) r.ref << 56) | ((uint64_t) t.plane << 52) | ((y >> 6) << 32) | x;
        };
        for (int b = 0; b < MC_BINS; b++)
            std::stable_sort(bins[b].begin(), bins[b].end(), [&](const McTile &p, const McTile &q) { return key(p) < key(q); });
    }
    std::vector<McTile> all;
    for (int b = 0; b < MC_BINS; b++) {
        l->off[b] = all.size();
        all.insert(all.end(), bins[b].begin(), bins[b].end());
    }
    l->off[MC_BINS] = all.size();
    l->n = all.size();
    for (const McTile &t : all) {
        const bool two = t.kind == MCT_AVG || t.kind == MCT_WAVG;
        l->max_ref = std::max(l->max_ref, std::max((int) t.r[0].ref, two ? (int) t.r[1].ref : 0));
    }
    if (l->n) {
        if (hipMalloc((void **) &l->dev, l->n * sizeof(McTile)) != hipSuccess) { delete l; return -ENOMEM; }
        int rc = dav1d_hip_upload(c, l->dev, all.data(), l->n * sizeof(McTile));
        if (!rc && !(l->host = (McTile *) malloc(l->n * sizeof(McTile)))) rc = -ENOMEM;
        if (rc) { hipFree(l->dev); delete l; return rc; }
        memcpy(l->host, all.data(), l->n * sizeof(McTile));
        // All shapes in one list: cells of (reference, plane, 64-row band, 512-pixel strip) of the SOURCE position, shapes
        // kept together inside a cell so that a wave gets a full group of one shape; a group never leaves its cell.
        struct Ent { uint64_t key; uint32_t idx; };
        const int fb = mc_fused_min_bin();
        l->n_fused = l->n - l->off[fb];
        std::vector<Ent> ord(l->n_fused);
        for (int b = fb; b < MC_BINS; b++)
            for (size_t i = l->off[b]; i < l->off[b + 1]; i++) {
                const McRef &r = all[i].r[0];
                const uint64_t y = (uint64_t) (r.src_y + 4096) & 0xffff, x = (uint64_t) (r.src_x + 4096) & 0xffff;
                Ent &e = ord[i - l->off[fb]];
                e.key = ((uint64_t) r.ref << 60) | ((uint64_t) all[i].plane << 58) | ((y >> 6) << 48) | ((x >> 9) << 42) |
                             ((uint64_t) b << 38) | (x << 16) | y;
                e.idx = (uint32_t) i;
            }
        std::sort(ord.begin(), ord.end(), [](const Ent &p, const Ent &q) { return p.key < q.key; });
        std::vector<McTile> fused(l->n_fused);
        std::vector<McGroup> groups;
        uint64_t cur = ~0ull;
        for (size_t i = 0; i < l->n_fused; i++) {
            fused[i] = all[ord[i].idx];
            const uint64_t cell_cls = ord[i].key >> 38;
            const int cls = (int) (cell_cls & 15);
            const int tw = 4 << (cls / 3), th = 4 << (cls % 3);
            const int per_wave = 64 / (tw * th / 4 < 64 ? tw * th / 4 : 64);
            if (cell_cls != cur || groups.back().n >= per_wave) {
                McGroup g = { (uint32_t) i, 0, (uint16_t) cls };
                groups.push_back(g);
                cur = cell_cls;
            }
            groups.back().n++;
        }
        l->n_groups = groups.size();
        if (l->n_fused) {
            if (hipMalloc((void **) &l->dev_all, l->n_fused * sizeof(McTile)) != hipSuccess ||
                hipMalloc((void **) &l->groups, groups.size() * sizeof(McGroup)) != hipSuccess) rc = -ENOMEM;
            if (!rc) rc = dav1d_hip_upload(c, l->dev_all, fused.data(), l->n_fused * sizeof(McTile));
            if (!rc) rc = dav1d_hip_upload(c, l->groups, groups.data(), groups.size() * sizeof(McGroup));
        }
        if (rc) { hipFree(l->dev); free(l->host); if (l->dev_all) hipFree(l->dev_all); if (l->groups) hipFree(l->groups); delete l; return rc; }
    }
    *out = l;
    return 0;
}

// Tiles that share a wave should share their code path: a wave runs the edge-emulating gather if ANY of its tiles leaves
// the reference plane, and the second prediction if ANY of them is a fused compound.  Inside windows of consecutive tiles
// (the source order, so the lines they read stay together) the tiles are grouped by (leaves the plane, kind).  Which tiles
// leave the plane depends on the reference geometry, known only at run time: done on the first run and again whenever the
// geometry changes.  The tiles of one list write disjoint rectangles (BLEND_V aside, which lives in the comp list), so
// their order is free.  Speed only.
uint64_t dav1d_hip_mc_geo_sig(const DevPlanes *rp, int n_refs) {
    uint64_t sig = 0xcbf29ce484222325ull;
    for (int r = 0; r < n_refs; r++)
        for (int p = 0; p < 3; p++) { sig = (sig ^ (uint32_t) rp[r].w[p]) * 0x100000001b3ull; sig = (sig ^ (uint32_t) rp[r].h[p]) * 0x100000001b3ull; }
    return sig | 1;
}

static int mc_regroup(Dav1dHipContext *c, Dav1dHipMcList *l, const DevPlanes *rp, int n_refs) {
    static const int win_waves = getenv("DAV1D_HIP_MC_GROUP_WINDOW") ? atoi(getenv("DAV1D_HIP_MC_GROUP_WINDOW")) : 128;
    if (win_waves <= 0 || !l->n) return 0;
    const uint64_t sig = dav1d_hip_mc_geo_sig(rp, n_refs);
    if (l->geo_sig == sig) return 0;
    std::vector<McTile> g(l->host, l->host + l->n);
    std::vector<uint8_t> key(l->n);
    for (int b = 0; b < MC_BINS; b++) {
        const int tw = 4 << (b / 3), th = 4 << (b % 3);
        const int ws = tw == 4 ? 12 : (tw + 8 + 7) & ~7, ext_x = (ws + 7) / 8 * 8, ext_y = th + 7;   // mc.hip: NCH * 8, WR - 1
        const int lanes = tw * th / 4 < 64 ? tw * th / 4 : 64;
        const size_t win = (size_t) win_waves * (size_t) (64 / lanes);
        if (64 / lanes < 2) continue;                    // one tile per wave: nothing to share
        for (size_t i = l->off[b]; i < l->off[b + 1]; i++) {
            const McTile &t = g[i];
            const bool two = t.kind == MCT_AVG || t.kind == MCT_WAVG;
            bool edge = false;
            for (int k = 0; k < (two ? 2 : 1); k++) {
                const McRef &r = t.r[k];
                const int x0 = r.src_x - 4, y0 = r.src_y - 3;
                edge |= x0 < 0 || y0 < 0 || x0 + ext_x > rp[r.ref].w[t.plane] || y0 + ext_y > rp[r.ref].h[t.plane];
            }
            key[i] = (uint8_t) ((edge ? 16 : 0) | t.kind << 1 | (t.r[0].src_x & 1));       // the column parity: see the paired blocks of recon lists
        }
        std::vector<uint32_t> idx;
        for (size_t lo = l->off[b]; lo < l->off[b + 1]; lo += win) {
            const size_t hi = std::min(lo + win, l->off[b + 1]);
            idx.resize(hi - lo);
            for (size_t i = lo; i < hi; i++) idx[i - lo] = (uint32_t) i;
            std::stable_sort(idx.begin(), idx.end(), [&](uint32_t p, uint32_t q) { return key[p] < key[q]; });
            for (size_t i = lo; i < hi; i++) g[i] = l->host[idx[i - lo]];
        }
    }
    hipStreamSynchronize(c->stream);                     // an earlier run may still be reading the old order
    const int rc = dav1d_hip_upload(c, l->dev, g.data(), l->n * sizeof(McTile));
    if (!rc) l->geo_sig = sig;
    return rc;
}

extern "C" {

int dav1d_hip_mc_list_create(Dav1dHipContext *c, Dav1dHipMcList **out, const Dav1dHipMcTask *tasks, size_t n) {
    if (!out || (!tasks && n)) return -EINVAL;
    *out = nullptr;
    std::vector<McTile> bins[MC_BINS];
    for (size_t i = 0; i < n; i++) {
        if (!mc_task_valid(tasks[i])) return -EINVAL;
        push_tiles(bins, tasks[i], tasks[i].kind == DAV1D_HIP_MC_PUT ? MCT_PUT : tasks[i].kind == DAV1D_HIP_MC_PREP ? MCT_PREP : MCT_PUT_TMP,
                   tasks[i].dst_off, nullptr, 0);
    }
    return mc_list_from_bins(c, out, bins);
}

void dav1d_hip_mc_list_destroy(Dav1dHipContext *c, Dav1dHipMcList *l) {
    if (!l) return;
    hipStreamSynchronize(c->stream);
    if (l->dev) hipFree(l->dev);
    free(l->host);
    if (l->dev_all) hipFree(l->dev_all);
    if (l->groups) hipFree(l->groups);
    delete l;
}

int dav1d_hip_mc_list_run(Dav1dHipContext *c, const Dav1dHipMcList *l, const Dav1dHipPicture *dst,
                          const Dav1dHipPicture *refs, int n_refs, int16_t *prep) {
    if (!l || !dst || !refs || n_refs < 1 || n_refs > 8 || (l->n && l->max_ref >= n_refs)) return -EINVAL;
    const DevPlanes dp = dev_planes(dst);
    DevPlanes rp[8];
    for (int i = 0; i < n_refs; i++) if (refs[i].bpc != dst->bpc) return -EINVAL;
    if (l->n_fused) {       // the all-shapes launch reads raster planes
        if (const int rv = raster_planes_valid(c, refs, n_refs)) return rv;
        for (int i = 0; i < n_refs; i++) rp[i] = dev_planes(&refs[i]);
    } else if (const int rv = ref_planes(c, refs, n_refs, rp)) return rv;
    const int fb = mc_fused_min_bin();
    int rc = mc_regroup(c, const_cast<Dav1dHipMcList *>(l), rp, n_refs);
    if (rc) return rc;
    StreamFan fan(c);
    if (l->n_fused) rc = dav1d_hip_launch_mc_all(&dp, rp, n_refs, dst->bpc, l->dev_all, l->groups, (int) l->n_groups, fb == 0, prep, fan.next());
    for (int b = fb - 1; b >= 0 && !rc; b--) {
        const size_t cnt = l->off[b + 1] - l->off[b];
        if (!cnt) continue;
        rc = dav1d_hip_launch_mc_bin(&dp, rp, n_refs, dst->bpc, b, l->dev + l->off[b], (int) cnt, prep, fan.next());
    }
    fan.join();
    return rc;
}

int dav1d_hip_mc_list_run_timed(Dav1dHipContext *c, const Dav1dHipMcList *l, const Dav1dHipPicture *dst,
                                const Dav1dHipPicture *refs, int n_refs, int16_t *prep, float *ms, size_t *counts) {
    if (!l || !dst || !refs || n_refs < 1 || n_refs > 8 || !ms || (l->n && l->max_ref >= n_refs)) return -EINVAL;
    const DevPlanes dp = dev_planes(dst);
    DevPlanes rp[8];
    if (const int rv = ref_planes(c, refs, n_refs, rp)) return rv;
    if (int rg = mc_regroup(c, const_cast<Dav1dHipMcList *>(l), rp, n_refs)) return rg;
    hipEvent_t ev[MC_BINS + 1];
    for (int b = 0; b <= MC_BINS; b++) HIP_TRY(hipEventCreate(&ev[b]));
    HIP_TRY(hipEventRecord(ev[0], c->stream));
    int rc = 0;
    for (int b = 0; b < MC_BINS && !rc; b++) {
        const size_t cnt = l->off[b + 1] - l->off[b];
        if (counts) counts[b] = cnt;
        if (cnt) rc = dav1d_hip_launch_mc_bin(&dp, rp, n_refs, dst->bpc, b, l->dev + l->off[b], (int) cnt, prep, c->stream);
        hipEventRecord(ev[b + 1], c->stream);
    }
    hipStreamSynchronize(c->stream);
    for (int b = 0; b < MC_BINS; b++) { ms[b] = 0.f; hipEventElapsedTime(&ms[b], ev[b], ev[b + 1]); }
    for (int b = 0; b <= MC_BINS; b++) hipEventDestroy(ev[b]);
    return rc;
}

int dav1d_hip_mc_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *refs, int n_refs,
                       const Dav1dHipMcTask *tasks, size_t n, int16_t *prep) {
    Dav1dHipMcList *l = nullptr;
    int rc = dav1d_hip_mc_list_create(c, &l, tasks, n);
    if (rc) return rc;
    rc = dav1d_hip_mc_list_run(c, l, dst, refs, n_refs, prep);
    dav1d_hip_mc_list_destroy(c, l);
    return rc;
}

} // extern "C"

// --------------------------------------------------------------------- comp
#include <unordered_set>


extern "C" {

int dav1d_hip_comp_list_create(Dav1dHipContext *c, Dav1dHipCompList **out, const Dav1dHipCompTask *tasks, size_t n) {
    if (!out || (!tasks && n)) return -EINVAL;
    *out = nullptr;
    for (size_t i = 0; i < n; i++) {
        const Dav1dHipCompTask &t = tasks[i];
        if (t.kind > 6 || t.plane > 2 || t.ss > 2 || t.w > 128 || t.h > 128 || t.w < 2 || t.h < 2 ||
            (t.kind <= 3 && (t.w < 4 || t.h < 4 || (t.w & 1) || (t.h & 1))))
            return -EINVAL;
    }
    Dav1dHipCompList *l = new (std::nothrow) Dav1dHipCompList();
    if (!l) return -ENOMEM;
    l->dev = nullptr;
    l->n = n;
    // obmc() blends the top neighbours' predictions (blend_h) before the left ones (blend_v) and the two overlap in the
    // block's top-left corner (reference src/recon_tmpl.c:1066-1111): keep that order with a second launch
    // The chroma planes of a COMP_INTER_SEG block are combined with the mask its luma W_MASK task wrote
    // (src/recon_tmpl.c:1812-1818, 1882-1889): those MASK tasks wait for the second launch as well.
    std::unordered_set<uint32_t> wmask_out;
    for (size_t i = 0; i < n; i++) if (tasks[i].kind == DAV1D_HIP_COMP_WMASK) wmask_out.insert(tasks[i].mask_off);
    auto second = [&](const Dav1dHipCompTask &t) {
        return t.kind == DAV1D_HIP_COMP_BLEND_V || (t.kind == DAV1D_HIP_COMP_MASK && wmask_out.count(t.mask_off));
    };
    std::vector<Dav1dHipCompTask> sorted;
    sorted.reserve(n);
    for (size_t i = 0; i < n; i++) if (!second(tasks[i])) sorted.push_back(tasks[i]);
    l->n_first = sorted.size();
    for (size_t i = 0; i < n; i++) if (second(tasks[i])) sorted.push_back(tasks[i]);
    if (n) {
        if (hipMalloc((void **) &l->dev, n * sizeof(Dav1dHipCompTask)) != hipSuccess) { delete l; return -ENOMEM; }
        const int rc = dav1d_hip_upload(c, l->dev, sorted.data(), n * sizeof(Dav1dHipCompTask));
        if (rc) { hipFree(l->dev); delete l; return rc; }
    }
    *out = l;
    return 0;
}

void dav1d_hip_comp_list_destroy(Dav1dHipContext *c, Dav1dHipCompList *l) {
    if (!l) return;
    hipStreamSynchronize(c->stream);
    if (l->dev) hipFree(l->dev);
    delete l;
}

int dav1d_hip_comp_list_run(Dav1dHipContext *c, const Dav1dHipCompList *l, const Dav1dHipPicture *dst,
                            const int16_t *prep, uint8_t *mask) {
    if (!l || !dst) return -EINVAL;
    const DevPlanes dp = dev_planes(dst);
    int rc = dav1d_hip_launch_comp(&dp, dst->bpc, l->dev, (int) l->n_first, prep, mask, c->stream);
    if (!rc) rc = dav1d_hip_launch_comp(&dp, dst->bpc, l->dev + l->n_first, (int) (l->n - l->n_first), prep, mask, c->stream);
    return rc;
}

int dav1d_hip_comp_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipCompTask *tasks, size_t n,
                         const int16_t *prep, uint8_t *mask) {
    Dav1dHipCompList *l = nullptr;
    int rc = dav1d_hip_comp_list_create(c, &l, tasks, n);
    if (rc) return rc;
    rc = dav1d_hip_comp_list_run(c, l, dst, prep, mask);
    dav1d_hip_comp_list_destroy(c, l);
    return rc;
}

} // extern "C"

// ------------------------------------------------------- inter list (mc + comp, fused)

// All inter prediction of a frame / tile-sbrow: the PUT / PREP tasks plus the compound
// tasks that consume the PREP outputs, exactly as the reference driver issues them
// (src/recon_tmpl.c:1784-1826).  Where an AVG / W_AVG task reads two PREP blocks that no
// other task reads, the three are fused into one tile kind (both predictions + combine in
// registers, nothing written to the prep arena).  MASK / W_MASK compounds keep the
// two-step form.
#include <unordered_map>
#include <algorithm>


// Recon lists: a transform block that covers exactly one prediction block (same plane, position and size, square 4x4 ..
// 64x64) is paired with it; the pair runs in one wave (recon.hip) and the prediction never reaches the picture on its own.
struct ReconPairing {
    std::unordered_map<uint64_t, uint32_t> by_pos;      // plane << 32 | dst_off -> index of the (square) transform task there
    const Dav1dHipItxTask *itx;
    std::vector<char> taken;                            // per transform task: paired
    std::vector<McTile> tiles[5];                       // per size class: tiles of the paired blocks, block by block
    std::vector<uint32_t> itx_idx[5];                   // per size class: the transform task of each block
    int mask;                                           // size classes that pair (bit k: 4 << k pixels square)
    int stride_px[3];                                   // picture strides (pixels) of the geometry the list is made for
    std::vector<uint8_t> blend_cells[3];                // per plane: 4x4 cells a blend task writes
    int cell_stride[3];
    void block_blend(const Dav1dHipCompTask &k) {
        const int sp = stride_px[k.plane];
        if (sp <= 0) return;
        const int x = (int) (k.dst_off % (uint32_t) sp), y = (int) (k.dst_off / (uint32_t) sp);
        for (int cy = y >> 2; cy <= (y + k.h - 1) >> 2; cy++)
            for (int cx = x >> 2; cx <= (x + k.w - 1) >> 2; cx++) {
                const size_t i = (size_t) cy * cell_stride[k.plane] + cx;
                if (cx < cell_stride[k.plane] && i < blend_cells[k.plane].size()) blend_cells[k.plane][i] = 1;
            }
    }
    bool blended(int plane, uint32_t dst_off, int w, int h) const {
        const int sp = stride_px[plane];
        if (sp <= 0 || blend_cells[plane].empty()) return false;
        const int x = (int) (dst_off % (uint32_t) sp), y = (int) (dst_off / (uint32_t) sp);
        for (int cy = y >> 2; cy <= (y + h - 1) >> 2; cy++)
            for (int cx = x >> 2; cx <= (x + w - 1) >> 2; cx++) {
                const size_t i = (size_t) cy * cell_stride[plane] + cx;
                if (cx < cell_stride[plane] && i < blend_cells[plane].size() && blend_cells[plane][i]) return true;
            }
        return false;
    }
    // the transform task a prediction of this rectangle pairs with, or -1
    long find(int plane, uint32_t dst_off, int w, int h) {
        if (w != h || blended(plane, dst_off, w, h)) return -1;
        auto it = by_pos.find((uint64_t) plane << 32 | dst_off);
        if (it == by_pos.end() || taken[it->second]) return -1;
        const Dav1dHipItxTask &t = itx[it->second];
        return (t.tx <= 4 && (mask >> t.tx & 1) && (4 << t.tx) == w) ? (long) it->second : -1;
    }
};

// marks the 4x4 cells of a w x h rectangle at pixel offset `off` of a plane
static void mark_cells(Dav1dHipInterList *l, int plane, uint32_t off, int w, int h, uint16_t bit) {
    const int sp = l->stride_px[plane], cs = l->cell_stride[plane];
    if (sp <= 0) return;
    const int x = (int) (off % (uint32_t) sp), y = (int) (off / (uint32_t) sp);
    for (int cy = y >> 2; cy <= (y + h - 1) >> 2; cy++)
        for (int cx = x >> 2; cx <= (x + w - 1) >> 2; cx++) {
            const size_t i = (size_t) cy * cs + cx;
            if (cx < cs && i < l->writers[plane].size()) l->writers[plane][i] |= bit;
        }
}

extern "C" {

} // extern "C"

static int inter_list_create_geo(Dav1dHipContext *c, Dav1dHipInterList **out, const Dav1dHipMcTask *mc, size_t n_mc,
                                 const Dav1dHipCompTask *comp, size_t n_comp, const Dav1dHipPicture *geom, ReconPairing *pair = nullptr);

extern "C" {

int dav1d_hip_inter_list_create(Dav1dHipContext *c, Dav1dHipInterList **out, const Dav1dHipMcTask *mc, size_t n_mc,
                                const Dav1dHipCompTask *comp, size_t n_comp) {
    return inter_list_create_geo(c, out, mc, n_mc, comp, n_comp, nullptr);
}

} // extern "C"

static int inter_list_create_geo(Dav1dHipContext *c, Dav1dHipInterList **out, const Dav1dHipMcTask *mc, size_t n_mc,
                                 const Dav1dHipCompTask *comp, size_t n_comp, const Dav1dHipPicture *geom, ReconPairing *pair) {
    if (!out || (!mc && n_mc) || (!comp && n_comp)) return -EINVAL;
    *out = nullptr;
    // prep offset -> producing PREP task, and how many compound inputs read that offset
    std::unordered_map<uint32_t, size_t> producer;
    std::unordered_map<uint32_t, int> readers;
    for (size_t i = 0; i < n_mc; i++) {
        if (!mc_task_valid(mc[i])) return -EINVAL;
        if (mc[i].kind == DAV1D_HIP_MC_PREP) producer[mc[i].dst_off] = i;
    }
    for (size_t i = 0; i < n_comp; i++) { readers[comp[i].tmp1_off]++; readers[comp[i].tmp2_off]++; }
    // A block some BLEND / BLEND_H / BLEND_V task writes on top of (OBMC, src/recon_tmpl.c:1052-1112) must not be paired:
    // the reference's order is prediction, blends, residual, and a paired wave would add the residual before the blends.
    if (pair)
        for (size_t i = 0; i < n_comp; i++)
            if (comp[i].kind >= DAV1D_HIP_COMP_BLEND) pair->block_blend(comp[i]);
    std::vector<char> fused_prep(n_mc, 0);
    std::vector<Dav1dHipCompTask> rest;
    std::vector<McTile> bins[MC_BINS];
    size_t n_fused = 0;
    for (size_t i = 0; i < n_comp; i++) {
        const Dav1dHipCompTask &k = comp[i];
        bool fuse = k.kind == DAV1D_HIP_COMP_AVG || k.kind == DAV1D_HIP_COMP_WAVG;
        size_t a = 0, b = 0;
        if (fuse) {
            auto pa = producer.find(k.tmp1_off), pb = producer.find(k.tmp2_off);
            fuse = pa != producer.end() && pb != producer.end() && k.tmp1_off != k.tmp2_off &&
                   readers[k.tmp1_off] == 1 && readers[k.tmp2_off] == 1;
            if (fuse) {
                a = pa->second; b = pb->second;
                fuse = mc[a].w == k.w && mc[a].h == k.h && mc[b].w == k.w && mc[b].h == k.h &&
                       mc[a].plane == k.plane && mc[b].plane == k.plane;
            }
        }
        if (fuse) {
            const long j = pair ? pair->find(k.plane, k.dst_off, k.w, k.h) : -1;
            if (j >= 0) { pair->taken[j] = 1; pair->itx_idx[pair->itx[j].tx].push_back((uint32_t) j); }
            push_tiles(bins, mc[a], k.kind == DAV1D_HIP_COMP_AVG ? MCT_AVG : MCT_WAVG, k.dst_off, &mc[b], k.arg,
                       j >= 0 ? &pair->tiles[pair->itx[j].tx] : nullptr);
            fused_prep[a] = fused_prep[b] = 1;
            n_fused++;
        } else {
            rest.push_back(k);
        }
    }
    for (size_t i = 0; i < n_mc; i++)
        if (!fused_prep[i]) {
            const long j = (pair && mc[i].kind == DAV1D_HIP_MC_PUT) ? pair->find(mc[i].plane, mc[i].dst_off, mc[i].w, mc[i].h) : -1;
            if (j >= 0) { pair->taken[j] = 1; pair->itx_idx[pair->itx[j].tx].push_back((uint32_t) j); }
            push_tiles(bins, mc[i], mc[i].kind == DAV1D_HIP_MC_PUT ? MCT_PUT : mc[i].kind == DAV1D_HIP_MC_PREP ? MCT_PREP : MCT_PUT_TMP,
                       mc[i].dst_off, nullptr, 0, j >= 0 ? &pair->tiles[pair->itx[j].tx] : nullptr);
        }
    Dav1dHipInterList *l = new (std::nothrow) Dav1dHipInterList();
    if (!l) return -ENOMEM;
    l->mc = nullptr; l->comp = nullptr; l->n_fused = n_fused;
    for (int p = 0; p < 3; p++) l->cell_stride[p] = l->stride_px[p] = 0;
    if (geom) {
        const int bps = geom->bpc > 8 ? 2 : 1;
        for (int p = 0; p < 3; p++) {
            if (!geom->p[p].data) continue;
            l->stride_px[p] = (int) (geom->p[p].stride / bps);
            l->cell_stride[p] = (l->stride_px[p] + 3) >> 2;
            l->writers[p].assign((size_t) l->cell_stride[p] * (size_t) ((geom->p[p].h + 127 + 3) >> 2), 0);
        }
        for (int b = 0; b < MC_BINS; b++)
            for (const McTile &t : bins[b])
                if (t.kind == MCT_PUT || t.kind == MCT_AVG || t.kind == MCT_WAVG)
                    mark_cells(l, t.plane, t.dst_off + (uint32_t) t.oy * (uint32_t) l->stride_px[t.plane] + t.ox, t.w, t.h, (uint16_t) (1u << b));
        for (const Dav1dHipCompTask &k : rest) mark_cells(l, k.plane, k.dst_off, k.w, k.h, 1u << 15);
    }
    int rc = mc_list_from_bins(c, &l->mc, bins);
    if (!rc) rc = dav1d_hip_comp_list_create(c, &l->comp, rest.data(), rest.size());
    if (rc) { dav1d_hip_mc_list_destroy(c, l->mc); delete l; return rc; }
    *out = l;
    return 0;
}

extern "C" {

void dav1d_hip_inter_list_destroy(Dav1dHipContext *c, Dav1dHipInterList *l) {
    if (!l) return;
    dav1d_hip_mc_list_destroy(c, l->mc);
    dav1d_hip_comp_list_destroy(c, l->comp);
    delete l;
}

int dav1d_hip_inter_list_run(Dav1dHipContext *c, const Dav1dHipInterList *l, const Dav1dHipPicture *dst,
                             const Dav1dHipPicture *refs, int n_refs, int16_t *prep, uint8_t *mask) {
    if (!l) return -EINVAL;
    int rc = dav1d_hip_mc_list_run(c, l->mc, dst, refs, n_refs, prep);
    if (!rc && l->comp->n) rc = dav1d_hip_comp_list_run(c, l->comp, dst, prep, mask);
    return rc;
}

int dav1d_hip_inter_list_run_timed(Dav1dHipContext *c, const Dav1dHipInterList *l, const Dav1dHipPicture *dst,
                                   const Dav1dHipPicture *refs, int n_refs, int16_t *prep, uint8_t *mask,
                                   float *ms, size_t *counts) {
    if (!l || !ms) return -EINVAL;
    int rc = dav1d_hip_mc_list_run_timed(c, l->mc, dst, refs, n_refs, prep, ms, counts);
    ms[MC_BINS] = 0.f;
    if (counts) counts[MC_BINS] = l->comp->n;
    if (!rc && l->comp->n) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, c->stream);
        rc = dav1d_hip_comp_list_run(c, l->comp, dst, prep, mask);
        hipEventRecord(e1, c->stream);
        hipStreamSynchronize(c->stream);
        hipEventElapsedTime(&ms[MC_BINS], e0, e1);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    return rc;
}

size_t dav1d_hip_inter_list_fused(const Dav1dHipInterList *l) { return l ? l->n_fused : 0; }

} // extern "C"

// --------------------------------------------------------------------- cdef

int dav1d_hip_cdef_run_groups(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src, const Dav1dHipCdefTask *tasks,
                              size_t n, const CdefGroup *groups, size_t n_groups, size_t n_raw, int damping, uint32_t *dirvar) {
    const size_t tb = (n * sizeof(Dav1dHipCdefTask) + 255) & ~(size_t) 255;
    TaskBuf dev_buf(c, tb + n_groups * sizeof(CdefGroup) + 256);
    uint8_t *const dev = reinterpret_cast<uint8_t *>(dev_buf.p);
    if (!dev) return -ENOMEM;
    int rc = dav1d_hip_upload(c, dev, tasks, n * sizeof(Dav1dHipCdefTask));
    if (!rc && n_groups) rc = dav1d_hip_upload(c, dev + tb, groups, n_groups * sizeof(CdefGroup));
    if (const int rv_ = raster_planes_valid(c, src, 1)) return rv_;      // (a source that lives in its tiled twin only: raster planes first)
    const DevPlanes dp = dev_planes(dst), sp = dev_planes(src);
    const Dav1dHipCdefTask *d_tasks = reinterpret_cast<const Dav1dHipCdefTask *>(dev);
    KernelTimer kt(c);
    if (!rc) rc = dav1d_hip_launch_cdef_groups(&dp, &sp, dst->bpc, dst->layout, d_tasks, reinterpret_cast<const CdefGroup *>(dev + tb),
                                               (int) n_groups, damping, dirvar, c->stream);
    if (!rc && n_raw) rc = dav1d_hip_launch_cdef(&dp, &sp, dst->bpc, dst->layout, d_tasks, (int) n, damping, dirvar, 1, c->stream);
    kt.stop();
    hipStreamSynchronize(c->stream);
    return rc;
}

extern "C" int dav1d_hip_cdef_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src,
                                    const Dav1dHipCdefTask *tasks, size_t n, int damping, uint32_t *dirvar) {
    if (!dst || !src || (!tasks && n) || dst->bpc != src->bpc || dst->layout != src->layout) return -EINVAL;
    if (!n) return 0;
    // one pass over the list (half a million units per 8K frame): the field checks as one OR-reduction
    unsigned bad = 0;
    for (size_t i = 0; i < n; i++) bad |= (unsigned) (tasks[i].edges > 15) | (unsigned) (tasks[i].plane > 2) | (unsigned) (tasks[i].dir > 7);
    if (bad) return -EINVAL;
    if (const int rv_ = raster_planes_valid(c, src, 1)) return rv_;      // (a source that lives in its tiled twin only: raster planes first)
    const DevPlanes dp = dev_planes(dst), sp = dev_planes(src);
    if (dav1d_hip_cdef_strip_ok(&dp, &sp, dst->bpc) && !c->cdef_unit_kernel) {
        // units that sit side by side share a wave (strip kernel); DSP-level RAW tasks keep the one-unit kernel
        std::vector<CdefGroup> groups;
        groups.reserve(n / 8 + 16);
        const size_t n_raw = dav1d_hip_cdef_make_groups(tasks, n, 0, groups);
        return dav1d_hip_cdef_run_groups(c, dst, src, tasks, n, groups.data(), groups.size(), n_raw, damping, dirvar);
    }
    TaskBuf dev_buf(c, n * sizeof(Dav1dHipCdefTask));
    Dav1dHipCdefTask *const dev = reinterpret_cast<Dav1dHipCdefTask *>(dev_buf.p);
    if (!dev) return -ENOMEM;
    int rc = dav1d_hip_upload(c, dev, tasks, n * sizeof(*dev));
    KernelTimer kt(c);
    if (!rc) rc = dav1d_hip_launch_cdef(&dp, &sp, dst->bpc, dst->layout, dev, (int) n, damping, dirvar, 0, c->stream);
    kt.stop();
    hipStreamSynchronize(c->stream);
    return rc;
}

// -------------------------------------------------------------- loop filter

extern "C" int dav1d_hip_lf_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipLfTask *tasks, size_t n,
                                  const uint8_t *lvl, ptrdiff_t b4_stride, const uint8_t lut_e[64], const uint8_t lut_i[64]) {
    if (!dst || (!tasks && n) || !lvl || !lut_e || !lut_i) return -EINVAL;
    if (!n) return 0;
    std::vector<Dav1dHipLfTask> sorted;
    sorted.reserve(n);
    size_t n0 = 0;
    for (int d = 0; d < 2; d++) {
        for (size_t i = 0; i < n; i++) {
            if (tasks[i].plane > 2 || tasks[i].dir > 1 || tasks[i].lvl_comp > 3) return -EINVAL;
            if (tasks[i].dir == d) sorted.push_back(tasks[i]);
        }
        if (d == 0) n0 = sorted.size();
    }
    TaskBuf dev_buf(c, n * sizeof(Dav1dHipLfTask));
    Dav1dHipLfTask *const dev = reinterpret_cast<Dav1dHipLfTask *>(dev_buf.p);
    if (!dev) return -ENOMEM;
    int rc = dav1d_hip_upload(c, dev, sorted.data(), n * sizeof(*dev));
    const DevPlanes dp = dev_planes(dst);
    // pass 1: every vertical edge; pass 2 (same stream, so after pass 1): every horizontal edge
    KernelTimer kt(c);
    if (!rc) rc = dav1d_hip_launch_lf(&dp, dst->bpc, 0, dev, (int) n0, lvl, (int) b4_stride, lut_e, lut_i, c->stream);
    if (!rc) rc = dav1d_hip_launch_lf(&dp, dst->bpc, 1, dev + n0, (int) (n - n0), lvl, (int) b4_stride, lut_e, lut_i, c->stream);
    kt.stop();
    hipStreamSynchronize(c->stream);
    return rc;
}

// -------------------------------------------------------------------- ipred

static int ipred_tasks_valid(const Dav1dHipIpredTask *tasks, size_t n, const uint8_t *aux) {
    for (size_t i = 0; i < n; i++) {
        const Dav1dHipIpredTask &t = tasks[i];
        if (t.plane > 2 || t.kind > DAV1D_HIP_IPRED_COPY || t.mode > 13 || !t.tw || !t.th || t.tw > 16 || t.th > 16) return -EINVAL;
        if (t.kind == DAV1D_HIP_IPRED_COPY) { if ((t.pal[2] & 0xf0f0) != 0) return -EINVAL; continue; }
        if (t.kind >= DAV1D_HIP_IPRED_PAL && t.kind != DAV1D_HIP_IPRED_PRED_TMP && !aux) return -EINVAL;
        if (t.kind == DAV1D_HIP_IPRED_PRED_TMP && (t.tw > 8 || t.th > 8 || t.mode > 12)) return -EINVAL;
        const bool cfl = t.kind == DAV1D_HIP_IPRED_CFL || t.kind >= DAV1D_HIP_IPRED_DSP_CFL_AC;
        if ((cfl || (t.kind != DAV1D_HIP_IPRED_PAL && t.mode == 13)) && (t.tw > 8 || t.th > 8)) return -EINVAL;   // both are limited to 32x32
        if (t.kind == DAV1D_HIP_IPRED_DSP_CFL_PRED && t.mode != 0 && (t.mode < 3 || t.mode > 5)) return -EINVAL;
    }
    return 0;
}

// Blocks of 1024 pixels or more whose predictor has no serial dependency are predicted by four workgroups each
// (ipred.hip: IPRED_PARTS); they go first in a batch so that the grid holds exactly 4 * n_big + n_small workgroups.
// The tasks of one batch are independent of each other, so their order is free.
static bool ipred_task_big(const Dav1dHipIpredTask &t) {
    if ((int) t.tw * t.th * 16 < 1024) return false;
    if (t.kind == DAV1D_HIP_IPRED_PAL) return true;
    return (t.kind == DAV1D_HIP_IPRED_PRED || t.kind == DAV1D_HIP_IPRED_DSP) && t.mode != 13;      // 13 = filter intra: serial
}
static size_t ipred_big_first(Dav1dHipIpredTask *t, size_t n) {
    return (size_t) (std::stable_partition(t, t + n, ipred_task_big) - t);
}

extern "C" int dav1d_hip_ipred_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipIpredTask *tasks, size_t n,
                                     uint8_t *pal_idx) {
    if (!dst || (!tasks && n)) return -EINVAL;
    if (!n) return 0;
    if (ipred_tasks_valid(tasks, n, pal_idx)) return -EINVAL;
    std::vector<Dav1dHipIpredTask> ordered(tasks, tasks + n);
    const size_t n_big = ipred_big_first(ordered.data(), n);
    TaskBuf dev_buf(c, n * sizeof(Dav1dHipIpredTask));
    Dav1dHipIpredTask *const dev = reinterpret_cast<Dav1dHipIpredTask *>(dev_buf.p);
    if (!dev) return -ENOMEM;
    int rc = dav1d_hip_upload(c, dev, ordered.data(), n * sizeof(*dev));
    const DevPlanes dp = dev_planes(dst);
    KernelTimer kt(c);
    if (!rc) rc = dav1d_hip_launch_ipred(&dp, dst->bpc, dst->layout, dev, (int) n, (int) n_big, pal_idx, nullptr, c->stream);
    kt.stop();
    hipStreamSynchronize(c->stream);
    return rc;
}

// Device-resident wavefront: the batches of an intra frame (or of the intra blocks of an inter frame) uploaded once; batch k
// = tasks [start[k], start[k + 1]).  run_batch() only enqueues the launch, so a caller can interleave the residual lists of
// every wave on the same stream without a host round trip per wave.
struct Dav1dHipIpredList {
    Dav1dHipIpredTask *dev;
    std::vector<size_t> start, n_big;     // batch k = tasks [start[k], start[k + 1]), its first n_big[k] are split four ways
    bool needs_aux, needs_tmp;
};

extern "C" int dav1d_hip_ipred_list_create(Dav1dHipContext *c, Dav1dHipIpredList **out, const Dav1dHipIpredTask *tasks,
                                           const size_t *batch_sizes, size_t n_batches) {
    if (!out || !batch_sizes) return -EINVAL;
    *out = nullptr;
    size_t n = 0;
    for (size_t k = 0; k < n_batches; k++) n += batch_sizes[k];
    if (n && !tasks) return -EINVAL;
    uint8_t dummy = 0;
    if (ipred_tasks_valid(tasks, n, &dummy)) return -EINVAL;
    Dav1dHipIpredList *l = new (std::nothrow) Dav1dHipIpredList();
    if (!l) return -ENOMEM;
    l->dev = nullptr;
    l->needs_aux = l->needs_tmp = false;
    for (size_t i = 0; i < n; i++) {
        if (tasks[i].kind >= DAV1D_HIP_IPRED_PAL && tasks[i].kind < DAV1D_HIP_IPRED_PRED_TMP) l->needs_aux = true;
        if (tasks[i].kind == DAV1D_HIP_IPRED_PRED_TMP) l->needs_tmp = true;
    }
    l->start.push_back(0);
    for (size_t k = 0; k < n_batches; k++) l->start.push_back(l->start.back() + batch_sizes[k]);
    if (n) {
        std::vector<Dav1dHipIpredTask> ordered(tasks, tasks + n);
        for (size_t k = 0; k < n_batches; k++) l->n_big.push_back(ipred_big_first(ordered.data() + l->start[k], batch_sizes[k]));
        if (hipMalloc((void **) &l->dev, n * sizeof(Dav1dHipIpredTask)) != hipSuccess) { delete l; return -ENOMEM; }
        const int rc = dav1d_hip_upload(c, l->dev, ordered.data(), n * sizeof(Dav1dHipIpredTask));
        if (rc) { hipFree(l->dev); delete l; return rc; }
    }
    *out = l;
    return 0;
}

// tmp: the scratch (prep) arena PRED_TMP tasks write to; NULL when the list holds none
static int ipred_list_run_batch_tmp(Dav1dHipContext *c, const Dav1dHipIpredList *l, size_t batch, const Dav1dHipPicture *dst, uint8_t *aux,
                                    void *tmp) {
    if (!l || !dst || batch + 1 >= l->start.size() || (l->needs_aux && !aux) || (l->needs_tmp && !tmp)) return -EINVAL;
    const size_t n = l->start[batch + 1] - l->start[batch];
    if (!n) return 0;
    const DevPlanes dp = dev_planes(dst);
    return dav1d_hip_launch_ipred(&dp, dst->bpc, dst->layout, l->dev + l->start[batch], (int) n, (int) l->n_big[batch], aux, tmp, c->stream);
}

extern "C" int dav1d_hip_ipred_list_run_batch(Dav1dHipContext *c, const Dav1dHipIpredList *l, size_t batch, const Dav1dHipPicture *dst,
                                              uint8_t *aux) {
    return ipred_list_run_batch_tmp(c, l, batch, dst, aux, nullptr);
}

extern "C" void dav1d_hip_ipred_list_destroy(Dav1dHipContext *c, Dav1dHipIpredList *l) {
    if (!l) return;
    hipStreamSynchronize(c->stream);
    if (l->dev) hipFree(l->dev);
    delete l;
}

// ------------------------------------------------- mc: warp, scaled, resize, emu_edge

template <typename T, typename Launch>
static int run_task_batch(Dav1dHipContext *c, const T *tasks, size_t n, Launch launch) {
    TaskBuf dev_buf(c, n * sizeof(T));
    T *const dev = reinterpret_cast<T *>(dev_buf.p);
    if (!dev) return -ENOMEM;
    int rc = dav1d_hip_upload(c, dev, tasks, n * sizeof(T));
    if (!rc) rc = launch(dev);
    hipStreamSynchronize(c->stream);
    return rc;
}

extern "C" int dav1d_hip_warp_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *refs, int n_refs,
                                    const Dav1dHipWarpTask *tasks, size_t n, int16_t *prep) {
    if (!dst || !refs || n_refs < 1 || n_refs > 8 || (!tasks && n)) return -EINVAL;
    if (!n) return 0;
    for (size_t i = 0; i < n; i++) {
        const Dav1dHipWarpTask &t = tasks[i];
        if (t.kind > DAV1D_HIP_MC_PREP || t.plane > 2 || t.ref >= n_refs) return -EINVAL;
        if (t.kind == DAV1D_HIP_MC_PREP && !prep) return -EINVAL;
    }
    DevPlanes rp[8];
    if (const int rv = raster_planes_valid(c, refs, n_refs)) return rv;          // (the warp kernels read raster planes)
    for (int i = 0; i < n_refs; i++) { if (refs[i].bpc != dst->bpc) return -EINVAL; rp[i] = dev_planes(&refs[i]); }
    const DevPlanes dp = dev_planes(dst);
    return run_task_batch(c, tasks, n, [&](const Dav1dHipWarpTask *dev) {
        return dav1d_hip_launch_warp(&dp, rp, n_refs, dst->bpc, dev, (int) n, prep, c->stream); });
}

extern "C" int dav1d_hip_mc_scaled_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *refs, int n_refs,
                                         const Dav1dHipMcScaledTask *tasks, size_t n, int16_t *prep) {
    if (!dst || !refs || n_refs < 1 || n_refs > 8 || (!tasks && n)) return -EINVAL;
    if (!n) return 0;
    for (size_t i = 0; i < n; i++) {
        const Dav1dHipMcScaledTask &t = tasks[i];
        if (t.kind > DAV1D_HIP_MC_PUT_TMP || t.plane > 2 || t.ref >= n_refs || t.filter_2d > 9) return -EINVAL;
        if (t.w < 2 || t.w > 128 || t.h < 2 || t.h > 128 || t.mx < 0 || t.mx > 1023 || t.my < 0 || t.my > 1023 || t.dx < 0 || t.dy < 0)
            return -EINVAL;
        if (t.kind != DAV1D_HIP_MC_PUT && !prep) return -EINVAL;
    }
    DevPlanes rp[8];
    if (const int rv = raster_planes_valid(c, refs, n_refs)) return rv;          // (so do the scaled ones)
    for (int i = 0; i < n_refs; i++) { if (refs[i].bpc != dst->bpc) return -EINVAL; rp[i] = dev_planes(&refs[i]); }
    const DevPlanes dp = dev_planes(dst);
    return run_task_batch(c, tasks, n, [&](const Dav1dHipMcScaledTask *dev) {
        return dav1d_hip_launch_mc_scaled(&dp, rp, n_refs, dst->bpc, dev, (int) n, prep, c->stream); });
}

extern "C" int dav1d_hip_resize(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src, int plane, int dst_w, int y0,
                                int h, int src_w, int dx, int mx0) {
    if (!dst || !src || dst->bpc != src->bpc || plane < 0 || plane > 2 || dst_w < 1 || src_w < 1 || h < 0 || y0 < 0) return -EINVAL;
    if (mx0 < 0 || mx0 > 0x3fff || dx < 0) return -EINVAL;
    if (!h) return 0;
    if (const int rv_ = raster_planes_valid(c, src, 1)) return rv_;      // (a source that lives in its tiled twin only: raster planes first)
    const DevPlanes dp = dev_planes(dst), sp = dev_planes(src);
    if (y0 + h > dp.h[plane] || y0 + h > sp.h[plane] || dst_w > dp.w[plane]) return -EINVAL;
    const int rc = dav1d_hip_launch_resize(&dp, &sp, dst->bpc, plane, dst_w, y0, h, src_w, dx, mx0, c->stream);
    hipStreamSynchronize(c->stream);
    return rc;
}

extern "C" int dav1d_hip_emu_edge(Dav1dHipContext *c, int bpc, intptr_t bw, intptr_t bh, intptr_t iw, intptr_t ih, intptr_t x, intptr_t y,
                                  void *dst, ptrdiff_t dst_stride, const void *ref, ptrdiff_t ref_stride) {
    if (!dst || !ref || bw < 1 || bh < 1 || iw < 1 || ih < 1 || (bpc != 8 && bpc != 10 && bpc != 12)) return -EINVAL;
    const int rc = dav1d_hip_launch_emu_edge(dst, dst_stride, ref, ref_stride, (int) bw, (int) bh, (int) iw, (int) ih, (int) x, (int) y, bpc,
                                             c->stream);
    hipStreamSynchronize(c->stream);
    return rc;
}

// --------------------------------------------------------- loop restoration

extern "C" int dav1d_hip_lr_batch(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src,
                                  const Dav1dHipPicture *lpf, const Dav1dHipLrTask *tasks, size_t n) {
    if (!dst || !src || !lpf || (!tasks && n) || dst->bpc != src->bpc || lpf->bpc != src->bpc) return -EINVAL;
    if (!n) return 0;
    for (size_t i = 0; i < n; i++) {
        const Dav1dHipLrTask &t = tasks[i];
        if (t.plane > 2 || t.edges > 15 || !t.w || t.w > 384 || !t.h || t.h > 64) return -EINVAL;
        if (t.type > DAV1D_HIP_LR_SGR_MIX) return -EINVAL;
    }
    // Wiener tasks first, self-guided tasks second: one launch each (tasks write disjoint stripes)
    std::vector<Dav1dHipLrTask> sorted;
    sorted.reserve(n);
    for (size_t i = 0; i < n; i++) if (tasks[i].type <= DAV1D_HIP_LR_WIENER5) sorted.push_back(tasks[i]);
    const size_t nw = sorted.size();
    for (size_t i = 0; i < n; i++) if (tasks[i].type > DAV1D_HIP_LR_WIENER5) sorted.push_back(tasks[i]);
    // self-guided: the units of a row share waves (lr.hip)
    std::vector<uint32_t> waves;
    dav1d_hip_sgr_make_rows(sorted.data() + nw, n - nw, waves);
    const size_t o_waves = (n * sizeof(Dav1dHipLrTask) + 15) & ~(size_t) 15;
    TaskBuf devb_buf(c, o_waves + waves.size() * 4 + 16);
    uint8_t *const devb = reinterpret_cast<uint8_t *>(devb_buf.p);
    if (!devb) return -ENOMEM;
    Dav1dHipLrTask *const dev = reinterpret_cast<Dav1dHipLrTask *>(devb);
    int rc = dav1d_hip_upload(c, dev, sorted.data(), n * sizeof(*dev));
    if (!rc && !waves.empty()) rc = dav1d_hip_upload(c, devb + o_waves, waves.data(), waves.size() * 4);
    if (const int rv_ = raster_planes_valid(c, src, 1)) return rv_;
    const DevPlanes dp = dev_planes(dst), sp = dev_planes(src), lp = dev_planes(lpf);
    KernelTimer kt(c);
    int max_w = 0;
    for (size_t i = 0; i < nw; i++) max_w = std::max(max_w, (int) sorted[i].w);
    if (!rc) rc = dav1d_hip_launch_wiener(&dp, &sp, &lp, dst->bpc, dev, (int) nw, max_w, c->stream);
    if (!rc) rc = dav1d_hip_launch_sgr(&dp, &sp, &lp, dst->bpc, dev + nw, devb + o_waves, (int) (waves.size() / 4), c->stream);
    kt.stop();
    hipStreamSynchronize(c->stream);
    return rc;
}

// --------------------------------------------------------------- film grain

// generate_scaling, reference src/fg_apply_tmpl.c:41-95 (piecewise-linear LUT over the scaling points;
// high bit depth interpolates between the 8-bit grid points)
static void fg_generate_scaling(const int bitdepth, const uint8_t points[][2], const int num, uint8_t *scaling) {
    const int shift_x = bitdepth - 8, scaling_size = 1 << bitdepth;
    if (num == 0) { memset(scaling, 0, scaling_size); return; }
    memset(scaling, points[0][1], (size_t) points[0][0] << shift_x);
    for (int i = 0; i < num - 1; i++) {
        const int bx = points[i][0], by = points[i][1], ex = points[i + 1][0], ey = points[i + 1][1];
        const int dx = ex - bx, dy = ey - by;
        const int delta = dy * ((0x10000 + (dx >> 1)) / dx);
        for (int x = 0, d = 0x8000; x < dx; x++) { scaling[(bx + x) << shift_x] = (uint8_t) (by + (d >> 16)); d += delta; }
    }
    const int n = points[num - 1][0] << shift_x;
    memset(&scaling[n], points[num - 1][1], scaling_size - n);
    if (shift_x) {
        const int pad = 1 << shift_x, rnd = pad >> 1;
        for (int i = 0; i < num - 1; i++) {
            const int bx = points[i][0] << shift_x, ex = points[i + 1][0] << shift_x, dx = ex - bx;
            for (int x = 0; x < dx; x += pad) {
                const int range = scaling[bx + x + pad] - scaling[bx + x];
                for (int k = 1, r = rnd; k < pad; k++) { r += range; scaling[bx + x + k] = (uint8_t) (scaling[bx + x] + (r >> shift_x)); }
            }
        }
    }
}

extern "C" int dav1d_hip_fg_generate_grain(Dav1dHipContext *c, const Dav1dHipFilmGrainData *data, int bpc, int layout, int16_t *host_lut) {
    if (!data || !host_lut || (bpc != 8 && bpc != 10 && bpc != 12)) return -EINVAL;
    const size_t bytes = 3 * 74 * 82 * sizeof(int16_t);
    TaskBuf dev_buf(c, bytes);
    int16_t *const dev = reinterpret_cast<int16_t *>(dev_buf.p);
    if (!dev) return -ENOMEM;
    hipMemsetAsync(dev, 0, bytes, c->stream);
    int rc = dav1d_hip_launch_fg_gen(dev, data, bpc, layout, c->stream);
    if (!rc) rc = dav1d_hip_download(c, host_lut, dev, bytes);
    return rc;
}

// Grain templates + scaling tables of one frame (dav1d_prep_grain, src/fg_apply_tmpl.c:97-163 up to the row loop): they depend
// on the frame header only, so they are generated on a side stream as soon as the parameters are known — a lone wave per
// template, ~0.23 ms of latency that then hides behind the reconstruction of the frame — and dav1d_hip_fg_apply_prepared
// (the dav1d_apply_grain_row part) only waits for their event.
struct Dav1dHipGrain {
    uint8_t *dev;
    size_t lut_bytes, scaling_size;
    int bpc, layout;
    Dav1dHipFilmGrainData data;
    std::vector<uint8_t> sc;       // host copy of the scaling tables: kept alive for the asynchronous upload
    hipEvent_t ready;
    hipStream_t side;
};

static int fg_prepare_on(Dav1dHipContext *c, Dav1dHipGrain **out, const Dav1dHipFilmGrainData *data, int bpc, int layout, hipStream_t stream) {
    if (!c || !out || !data || (bpc != 8 && bpc != 10 && bpc != 12) || layout < 0 || layout > 3) return -EINVAL;
    *out = nullptr;
    Dav1dHipGrain *g = new (std::nothrow) Dav1dHipGrain();
    if (!g) return -ENOMEM;
    g->dev = nullptr; g->bpc = bpc; g->layout = layout; g->data = *data;
    g->scaling_size = (size_t) 1 << bpc;
    g->lut_bytes = (3 * 74 * 82 * sizeof(int16_t) + 255) & ~(size_t) 255;      // keeps the scaling tables 16-byte aligned
    g->side = stream;
    if (hipEventCreateWithFlags(&g->ready, hipEventDisableTiming) != hipSuccess) { delete g; return -ENOMEM; }
    if (hipMalloc((void **) &g->dev, g->lut_bytes + 3 * g->scaling_size) != hipSuccess) { hipEventDestroy(g->ready); delete g; return -ENOMEM; }
    g->sc.assign(3 * g->scaling_size, 0);
    if (data->num_y_points || data->chroma_scaling_from_luma) fg_generate_scaling(bpc, data->y_points, data->num_y_points, &g->sc[0]);
    for (int i = 0; i < 2; i++)
        if (data->num_uv_points[i]) fg_generate_scaling(bpc, data->uv_points[i], data->num_uv_points[i], &g->sc[(size_t) (1 + i) * g->scaling_size]);
    int rc = hip_rc(hipMemsetAsync(g->dev, 0, g->lut_bytes, g->side));
    if (!rc) rc = hip_rc(hipMemcpyAsync(g->dev + g->lut_bytes, g->sc.data(), g->sc.size(), hipMemcpyHostToDevice, g->side));
    if (!rc) rc = dav1d_hip_launch_fg_gen((int16_t *) g->dev, data, bpc, layout, g->side);
    if (!rc) rc = hip_rc(hipEventRecord(g->ready, g->side));
    if (rc) { hipStreamSynchronize(g->side); hipFree(g->dev); hipEventDestroy(g->ready); delete g; return rc; }
    *out = g;
    return 0;
}

extern "C" int dav1d_hip_fg_prepare(Dav1dHipContext *c, Dav1dHipGrain **out, const Dav1dHipFilmGrainData *data, int bpc, int layout) {
    if (!c) return -EINVAL;
    return fg_prepare_on(c, out, data, bpc, layout, c->concurrent ? c->side[Dav1dHipContext::N_SIDE - 1] : c->stream);
}

extern "C" void dav1d_hip_fg_grain_destroy(Dav1dHipContext *c, Dav1dHipGrain *g) {
    if (!g) return;
    hipStreamSynchronize(g->side);
    if (c) hipStreamSynchronize(c->stream);
    hipFree(g->dev);
    hipEventDestroy(g->ready);
    delete g;
}

// the application proper on the context's stream (no timing, no synchronisation); offs: scratch for the per-block offsets
static int fg_apply_core(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src, const Dav1dHipGrain *g, int is_id,
                         uint8_t *offs) {
    const Dav1dHipFilmGrainData *data = &g->data;
    const int bpc = src->bpc;
    int rc = 0;
    // planes that get no grain are copied (dav1d_prep_grain, src/fg_apply_tmpl.c:127-163)
    const int ss_ver = src->layout == DAV1D_HIP_LAYOUT_I420;
    for (int pl = 0; pl < 3 && !rc; pl++) {
        if (pl && src->layout == DAV1D_HIP_LAYOUT_I400) break;
        const bool grain = pl ? (data->num_uv_points[pl - 1] || data->chroma_scaling_from_luma) : data->num_y_points != 0;
        if (grain) continue;
        const int rows = pl ? (src->p[0].h + ss_ver) >> ss_ver : src->p[0].h;
        const size_t rb = (size_t) src->p[pl].w * (bpc > 8 ? 2 : 1);
        rc = hip_rc(hipMemcpy2DAsync(dst->p[pl].data, dst->p[pl].stride, src->p[pl].data, src->p[pl].stride, rb, rows,
                                     hipMemcpyDeviceToDevice, c->stream));
    }
    if (const int rv_ = raster_planes_valid(c, src, 1)) return rv_;      // (a source that lives in its tiled twin only: raster planes first)
    const DevPlanes dp = dev_planes(dst), sp = dev_planes(src);
    if (!rc) rc = dav1d_hip_launch_fg_apply(&dp, &sp, (const int16_t *) g->dev, g->dev + g->lut_bytes, (int) g->scaling_size, data, bpc, src->layout,
                                            is_id, offs, c->stream);
    return rc;
}

static int fg_args_ok(const Dav1dHipPicture *dst, const Dav1dHipPicture *src) {
    return dst && src && dst->bpc == src->bpc && dst->layout == src->layout;
}

extern "C" int dav1d_hip_fg_apply_prepared(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src,
                                           const Dav1dHipGrain *g, int is_id) {
    if (!c || !g || !fg_args_ok(dst, src) || src->bpc != g->bpc || src->layout != g->layout) return -EINVAL;
    const size_t offs_bytes = (size_t) ((src->p[0].w + 31) / 32) * ((src->p[0].h + 31) / 32);
    TaskBuf offs_buf(c, offs_bytes + 16);
    uint8_t *const offs = reinterpret_cast<uint8_t *>(offs_buf.p);
    if (!offs) return -ENOMEM;
    int rc = hip_rc(hipStreamWaitEvent(c->stream, g->ready, 0));
    KernelTimer kt(c);
    if (!rc) rc = fg_apply_core(c, dst, src, g, is_id, offs);
    kt.stop();
    hipStreamSynchronize(c->stream);
    return rc;
}

// dav1d_apply_grain in one call: templates and application back to back on the context's stream
extern "C" int dav1d_hip_fg_apply(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src,
                                  const Dav1dHipFilmGrainData *data, int is_id) {
    if (!c || !data || !fg_args_ok(dst, src)) return -EINVAL;
    const size_t offs_bytes = (size_t) ((src->p[0].w + 31) / 32) * ((src->p[0].h + 31) / 32);
    TaskBuf offs_buf(c, offs_bytes + 16);
    uint8_t *const offs = reinterpret_cast<uint8_t *>(offs_buf.p);
    if (!offs) return -ENOMEM;
    Dav1dHipGrain *g = nullptr;
    KernelTimer kt(c);
    int rc = fg_prepare_on(c, &g, data, src->bpc, src->layout, c->stream);
    if (!rc) rc = fg_apply_core(c, dst, src, g, is_id, offs);
    kt.stop();
    hipStreamSynchronize(c->stream);
    dav1d_hip_fg_grain_destroy(c, g);
    return rc;
}

// ------------------------------------------------------------------ recon list: predictions and residuals pipelined
//
// dav1d_hip_inter_list_run followed by dav1d_hip_itx_list_run makes every residual wait for every prediction.  The residual
// launch of one transform size only needs the prediction launches whose tiles lie under its blocks; which ones those are is
// worked out here once, on a 4x4-cell map of the picture.  At run time the prediction launches go down the context's stream
// in the order largest tile shape first, each followed by an event; the residual launches go down a side stream, largest
// transform first, each waiting for the events of its own predecessors only.  The memory-bound predictions of the small
// shapes then overlap with the arithmetic-bound 64- and 32-point transforms instead of queueing in front of them.

// DAV1D_HIP_RECON_FUSE: which square block sizes get paired (transform block + the prediction block of the same rectangle in
// one wave, recon.hip): bit 0 4x4, bit 1 8x8, bit 2 16x16, bit 3 32x32, bit 4 64x64; 0 none.  Measured on MI355X (8K 10-bit
// frame, ms per frame, round 2 after the paired kernel's LDS regions were overlaid): 8x8 + 16x16 (6) 0.325-0.339,
// 8x8 + 16x16 + 32x32 (14, the default) 0.317-0.327.  Round 1 (three separate LDS arrays): none 0.362, 6 0.311 (older clock),
// 4x4 + 8x8 0.318, all 0.411.  What pays is that the paired launches move a quarter less HBM traffic AND run next to the
// pipelined launches of the other sizes on streams of their own; 4x4 and 64x64 pairs lose to their separate kernels.
// Round 6: the 4x4 pairs too (15, the default now): their launch fits five LDS pieces since the records pass through the window
// buffer (48 us against 37 + 23 for the two launches it replaces) and it runs on the MAIN stream, out of the way of the side streams'
// chains; with two frame contexts and two paired streams 0.2295 against 0.2424 ms per frame (profiles/r06/streams.txt).
int recon_fuse_mask(const Dav1dHipContext *c) { return c->recon_fuse & 31; }

extern "C" {

int dav1d_hip_recon_list_create(Dav1dHipContext *c, Dav1dHipReconList **out, const Dav1dHipPicture *geometry,
                                const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                                const Dav1dHipItxTask *itx, size_t n_itx) {
    if (!c || !out || !geometry) return -EINVAL;
    *out = nullptr;
    if ((!itx && n_itx) || n_itx > 0xffffffffu) return -EINVAL;
    for (size_t i = 0; i < n_itx; i++) if (!itx_task_ok(itx[i])) return -EINVAL;
    Dav1dHipReconList *l = new (std::nothrow) Dav1dHipReconList();
    if (!l) return -ENOMEM;
    const Dav1dHipItxTask *const itx_all = itx;
    const size_t n_itx_all = n_itx;
    l->inter = nullptr; l->itx = nullptr; l->wide_ok = false;
    for (int k = 0; k < 5; k++) { l->f_tiles[k] = nullptr; l->f_tasks[k] = nullptr; l->f_n[k] = 0; }
    l->f_max_ref = 0;
    ReconPairing pair;
    const bool fuse = recon_fuse_mask(c) != 0;
    if (fuse) {
        pair.mask = recon_fuse_mask(c);
        pair.itx = itx;
        bool any_blend = false;
        for (size_t i = 0; i < n_comp && !any_blend; i++) any_blend = comp[i].kind >= DAV1D_HIP_COMP_BLEND;
        for (int p = 0; p < 3; p++) {
            const int bps = geometry->bpc > 8 ? 2 : 1;
            pair.stride_px[p] = geometry->p[p].data ? (int) (geometry->p[p].stride / bps) : 0;
            pair.cell_stride[p] = (pair.stride_px[p] + 3) >> 2;
            if (any_blend && pair.stride_px[p]) pair.blend_cells[p].assign((size_t) pair.cell_stride[p] * (size_t) ((geometry->p[p].h + 127 + 3) >> 2), 0);
        }
        pair.taken.assign(n_itx, 0);
        for (size_t i = 0; i < n_itx; i++)
            if (itx[i].tx <= 4 && (pair.mask >> itx[i].tx & 1)) pair.by_pos[(uint64_t) itx[i].plane << 32 | itx[i].dst_off] = (uint32_t) i;
    }
    int rc = inter_list_create_geo(c, &l->inter, mc, n_mc, comp, n_comp, geometry, fuse ? &pair : nullptr);
    std::vector<Dav1dHipItxTask> rest;
    if (!rc && fuse) {
        rest.reserve(n_itx);
        for (size_t i = 0; i < n_itx; i++) if (!pair.taken[i]) rest.push_back(itx[i]);
        itx = rest.data();
        n_itx = rest.size();
    }
    if (!rc) rc = dav1d_hip_itx_list_create(c, &l->itx, itx, n_itx);
    // ---- the paired blocks of each size: ordered by where their first tile reads (as the tiles of mc lists are), then
    // grouped by the transform's code path inside windows of 128 waves (as the blocks of itx lists are); uploaded
    for (int k = 0; k < 5 && !rc && fuse; k++) {
        const size_t nblk = pair.itx_idx[k].size();
        if (!nblk) continue;
        const int tpb = k < 3 ? 1 : k == 3 ? 2 : 4, bpw = k == 0 ? 16 : k == 1 ? 8 : k == 2 ? 4 : k == 3 ? 2 : 1;
        if (pair.tiles[k].size() != nblk * tpb) { rc = -EINVAL; break; }
        std::vector<uint32_t> ord(nblk);
        std::vector<uint64_t> skey(nblk);
        for (size_t i = 0; i < nblk; i++) {
            ord[i] = (uint32_t) i;
            const McTile &t = pair.tiles[k][i * tpb];
            const uint64_t y = (uint64_t) (t.r[0].src_y + 4096) & 0xffff, x = (uint64_t) (t.r[0].src_x + 4096) & 0xffff;
            skey[i] = ((uint64_t) t.r[0].ref << 56) | ((uint64_t) t.plane << 52) | ((y >> 6) << 32) | x;
        }
        std::stable_sort(ord.begin(), ord.end(), [&](uint32_t p, uint32_t q) { return skey[p] < skey[q]; });
        const size_t win = (size_t) 128 * bpw;
        for (size_t lo = 0; lo < nblk; lo += win)
            std::stable_sort(ord.begin() + lo, ord.begin() + std::min(lo + win, nblk), [&](uint32_t p, uint32_t q) {
                const McTile &tp = pair.tiles[k][p * tpb], &tq = pair.tiles[k][q * tpb];
                // ... and the parity of the first reference column: with tiled references the horizontal pass picks its tap pairs by it
                // (mc_body.h, TILED), and a wave whose tiles agree runs one of the two forms instead of both
                const int kp = (itx_path_key(pair.itx[pair.itx_idx[k][p]]) * 8 + tp.kind) * 2 + (tp.r[0].src_x & 1);
                const int kq = (itx_path_key(pair.itx[pair.itx_idx[k][q]]) * 8 + tq.kind) * 2 + (tq.r[0].src_x & 1);
                return kp < kq;
            });
        std::vector<McTile> tiles(nblk * tpb);
        std::vector<Dav1dHipItxTask> tasks(nblk);
        for (size_t i = 0; i < nblk; i++) {
            for (int j = 0; j < tpb; j++) {
                const McTile &t = tiles[i * tpb + j] = pair.tiles[k][(size_t) ord[i] * tpb + j];
                const bool two = t.kind == MCT_AVG || t.kind == MCT_WAVG;
                l->f_max_ref = std::max(l->f_max_ref, std::max((int) t.r[0].ref, two ? (int) t.r[1].ref : 0));
            }
            tasks[i] = pair.itx[pair.itx_idx[k][ord[i]]];
            itx_fill_prefix(tasks[i]);
        }
        if (hipMalloc((void **) &l->f_tiles[k], tiles.size() * sizeof(McTile)) != hipSuccess ||
            hipMalloc((void **) &l->f_tasks[k], tasks.size() * sizeof(Dav1dHipItxTask)) != hipSuccess) { rc = -ENOMEM; break; }
        rc = dav1d_hip_upload(c, l->f_tiles[k], tiles.data(), tiles.size() * sizeof(McTile));
        if (!rc) rc = dav1d_hip_upload(c, l->f_tasks[k], tasks.data(), tasks.size() * sizeof(Dav1dHipItxTask));
        l->f_n[k] = nblk;
    }
    if (rc) { dav1d_hip_recon_list_destroy(c, l); return rc; }
    for (int b = 0; b < 19; b++) l->dep[b] = 0;
    for (int p = 0; p < 3; p++) l->stride_px[p] = l->inter->stride_px[p];
    l->wide_ok = true;
    for (size_t i = 0; i < n_itx_all && l->wide_ok; i++) {
        const Dav1dHipItxTask &t = itx_all[i];
        const int sp = l->stride_px[t.plane];
        l->wide_ok = sp > 0 && (int) (t.dst_off % (uint32_t) sp) % std::min((int) k_tx_w[t.tx], 8) == 0 && sp % 8 == 0;
    }
    for (size_t i = 0; i < n_itx; i++) {
        const Dav1dHipItxTask &t = itx[i];
        const int sp = l->stride_px[t.plane], cs = l->inter->cell_stride[t.plane];
        if (sp <= 0) { l->dep[t.tx] = 0xffff; continue; }
        const int x = (int) (t.dst_off % (uint32_t) sp), y = (int) (t.dst_off / (uint32_t) sp);
        const std::vector<uint16_t> &wr = l->inter->writers[t.plane];
        uint16_t m = 0;
        for (int cy = y >> 2; cy <= (y + k_tx_h[t.tx] - 1) >> 2; cy++)
            for (int cx = x >> 2; cx <= (x + k_tx_w[t.tx] - 1) >> 2; cx++) {
                const size_t j = (size_t) cy * cs + cx;
                m |= (cx < cs && j < wr.size()) ? wr[j] : (uint16_t) 0xffff;      // off the map: wait for everything
            }
        l->dep[t.tx] |= m;
    }
    // the maps are only needed for the dependency masks
    for (int p = 0; p < 3; p++) std::vector<uint16_t>().swap(l->inter->writers[p]);
    *out = l;
    return 0;
}

void dav1d_hip_recon_list_destroy(Dav1dHipContext *c, Dav1dHipReconList *l) {
    if (!l) return;
    if (l->inter) dav1d_hip_inter_list_destroy(c, l->inter);
    if (l->itx) dav1d_hip_itx_list_destroy(c, l->itx);
    hipStreamSynchronize(c->stream);
    for (int k = 0; k < 5; k++) { if (l->f_tiles[k]) hipFree(l->f_tiles[k]); if (l->f_tasks[k]) hipFree(l->f_tasks[k]); }
    delete l;
}

static int recon_list_run_impl(Dav1dHipContext *c, const Dav1dHipReconList *l, const Dav1dHipPicture *dst, const Dav1dHipPicture *refs, int n_refs,
                               int16_t *prep, uint8_t *mask, void *coef, bool wide, const DevPlanes *dst_twin);

int dav1d_hip_recon_list_run(Dav1dHipContext *c, const Dav1dHipReconList *l, const Dav1dHipPicture *dst,
                             const Dav1dHipPicture *refs, int n_refs, int16_t *prep, uint8_t *mask, void *coef) {
    return recon_list_run_impl(c, l, dst, refs, n_refs, prep, mask, coef, false, nullptr);
}

// The same, and the picture's tiled twin (Dav1dHipPicture.twin) holds the frame's pixels afterwards: the step of a frame whose in-loop
// filters are off, as later frames will predict from it.  When every launch of the list can write the twin along with the raster
// planes — tiled references, blocks on the 8-pixel grid, no mask / blend tasks (those go through a kernel that only knows raster
// planes) — it is written by the launches themselves (the paired kernels and the residual kernels through tile_write_out, the
// prediction kernels strip by strip); otherwise the list runs as always and dav1d_hip_picture_retile follows.  Sets dst->twin_ok.
// can every launch of the list write dst's twin itself?  (tiled references, blocks on the 8-pixel grid, no mask / blend tasks, aligned planes)
static bool recon_list_twin_direct(Dav1dHipContext *c, const Dav1dHipReconList *l, const Dav1dHipPicture *dst, const Dav1dHipPicture *refs, int n_refs) {
    bool direct = c->ref_twin != 0 && l->wide_ok && !l->inter->comp->n && mc_fused_min_bin() >= MC_BINS;
    const int bps = dst->bpc > 8 ? 2 : 1;
    for (int p = 0; p < 3 && direct; p++)
        if (dst->p[p].data) direct = dst->twin[p] && !((uintptr_t) dst->p[p].data & 15) && !((uintptr_t) dst->twin[p] & 15) && dst->p[p].stride % 16 == 0 &&
                                     (dst->p[p].stride / bps) % 8 == 0;
    for (int i = 0; i < n_refs && direct; i++) direct = picture_twin_usable(&refs[i]);
    static const bool debug_tiled = getenv("DAV1D_DEBUG_TILED") != nullptr;      // (asked once: this runs per frame)
    if (debug_tiled) fprintf(stderr, "twin_direct: ref_twin %d wide_ok %d comp %zu min_bin %d -> %d (refs ok: %d %d %d)\n", c->ref_twin, (int) l->wide_ok, (size_t) l->inter->comp->n, mc_fused_min_bin(), (int) direct, n_refs > 0 ? refs[0].twin_ok : -1, n_refs > 1 ? refs[1].twin_ok : -1, n_refs > 2 ? refs[2].twin_ok : -1);
    return direct;
}

int dav1d_hip_recon_list_run_twin(Dav1dHipContext *c, const Dav1dHipReconList *l, Dav1dHipPicture *dst,
                                  const Dav1dHipPicture *refs, int n_refs, int16_t *prep, uint8_t *mask, void *coef) {
    if (!c || !l || !dst || !refs) return -EINVAL;
    if (!dst->twin[0] && !dst->twin_alloc) { const int rc = dav1d_hip_picture_twin_alloc(c, dst); if (rc) return rc; }
    const bool direct = recon_list_twin_direct(c, l, dst, refs, n_refs);
    dst->twin_ok = 0;
    if (direct) {
        DevPlanes tw = dev_planes(dst);
        for (int p = 0; p < 3; p++) tw.data[p] = dst->p[p].data ? dst->twin[p] : nullptr;
        const int rc = recon_list_run_impl(c, l, dst, refs, n_refs, prep, mask, coef, true, &tw);
        if (!rc) dst->twin_ok = 1;
        return rc;
    }
    const int rc = recon_list_run_impl(c, l, dst, refs, n_refs, prep, mask, coef, false, nullptr);
    return rc ? rc : dav1d_hip_picture_retile(c, dst);
}

// The same with the picture living in its twin ONLY: nothing is written to the raster planes (dst->twin_ok = DAV1D_HIP_TWIN_ONLY
// afterwards) — an 8x8 block leaves as one 128-byte line instead of eight 16-byte row pieces, a 4x4 block as half a line instead of four
// 8-byte pieces, and the residual launches read the predicted pixels back the same way.  What reads such a picture: motion
// compensation of later frames (through the twin, as always), dav1d_hip_host_picture_fetch / dav1d_hip_plane_download (they un-tile
// on the way out: raster rows by the address rules of src/picture.c:46-63 exist at the output only) and dav1d_hip_picture_untile.
// `dst` on entry: any state; if it holds pixels the list does not overwrite (a partial list), they must be in the twin — a picture
// whose raster planes alone are valid is retiled first.  Lists that cannot run that way (recon_list_twin_direct) run on the raster
// planes and retile: twin_ok = 1 then.
int dav1d_hip_recon_list_run_tiled(Dav1dHipContext *c, const Dav1dHipReconList *l, Dav1dHipPicture *dst,
                                   const Dav1dHipPicture *refs, int n_refs, int16_t *prep, uint8_t *mask, void *coef) {
    if (!c || !l || !dst || !refs) return -EINVAL;
    if (!dst->twin[0] && !dst->twin_alloc) { const int rc = dav1d_hip_picture_twin_alloc(c, dst); if (rc) return rc; }
    if (!recon_list_twin_direct(c, l, dst, refs, n_refs)) {
        int rc = dav1d_hip_picture_untile(c, dst);
        if (!rc) rc = recon_list_run_impl(c, l, dst, refs, n_refs, prep, mask, coef, false, nullptr);
        dst->twin_ok = 0;
        return rc ? rc : dav1d_hip_picture_retile(c, dst);
    }
    // (a picture whose raster planes alone are valid is retiled first — once per picture: it lives in its twin from then on.  The contract
    // keeps every pixel the list does not write, the allocator's padding included, so "the list covers the visible picture" is no licence
    // to skip it)
    if (!dst->twin_ok) { const int rc = dav1d_hip_picture_retile(c, dst); if (rc) return rc; }
    DevPlanes tw = dev_planes(dst);
    for (int p = 0; p < 3; p++) tw.data[p] = dst->p[p].data ? dst->twin[p] : nullptr;
    tw.tiled = 2;
    const int rc = recon_list_run_impl(c, l, dst, refs, n_refs, prep, mask, coef, true, &tw);
    dst->twin_ok = rc ? 0 : DAV1D_HIP_TWIN_ONLY;
    return rc;
}

static int recon_list_run_impl(Dav1dHipContext *c, const Dav1dHipReconList *l, const Dav1dHipPicture *dst, const Dav1dHipPicture *refs, int n_refs,
                               int16_t *prep, uint8_t *mask, void *coef, const bool wide, const DevPlanes *dst_twin) {
    if (!c || !l || !dst || !refs) return -EINVAL;
    const int bps = dst->bpc > 8 ? 2 : 1;
    for (int p = 0; p < 3; p++)
        if (l->stride_px[p] && dst->p[p].stride / bps != l->stride_px[p]) return -EINVAL;    // not the geometry the list was made for
    size_t n_paired = 0;
    bool paired_on_side = false;
    int n_ps = 0;                                        // side streams the paired launches went to
    for (int k = 0; k < 5; k++) n_paired += l->f_n[k];
    if (n_paired) {
        // the paired blocks: one launch per size, largest first; independent of each other and of everything below
        if (n_refs < 1 || n_refs > 8 || l->f_max_ref >= n_refs) return -EINVAL;
        const DevPlanes dp = dev_planes(dst);
        DevPlanes rp[8];
        for (int i = 0; i < n_refs; i++) if (refs[i].bpc != dst->bpc) return -EINVAL;
        if (const int rv = ref_planes(c, refs, n_refs, rp)) return rv;
        // one after the other on a side stream of their own, next to the pipeline of the unpaired rest below (main stream:
        // predictions, side stream 0: residuals).  Measured: the paired launches on one stream 0.317 ms per frame, on two
        // streams that run side by side 0.334-0.343 — three launches at a time share the memory system better than four.
        const bool side = c->concurrent && n_paired >= 16384;
        int rc = 0, lane = 0;
        // c->recon_pair_streams: 1 = the paired launches one after the other on ONE side stream, 2 / 3 = dealt over that many (side
        // streams 2 .. 4; their ends are ev_pair[]).  Measured, round 6 (profiles/r06/streams.txt): with ONE frame in flight the
        // extra streams change nothing — round 2 measured the same — but with two frame contexts they are worth 11 %: the launches of
        // two frames on five streams keep every SIMD supplied with waves through the heads and tails of the single launches.
        n_ps = side ? std::max(1, std::min(3, c->recon_pair_streams)) : 0;
        if (side) {
            (void) hipEventRecord(c->ev_fork, c->stream);
            for (int j = 0; j < n_ps; j++) (void) hipStreamWaitEvent(c->side[c->recon_pair_first + j], c->ev_fork, 0);
        }
        for (int k = 4; k >= 0 && !rc; k--)
            if (l->f_n[k]) {
                // the 4x4 pairs go to the main stream, in front of the unpaired predictions: the side streams' chains of long launches are
                // what the step waits for, and the short launches of the main stream end long before them
                const bool on_main = side && k == 0;
                rc = dav1d_hip_launch_recon_fused_out(&dp, rp, n_refs, dst->bpc, k, l->f_tiles[k], l->f_tasks[k], (int) l->f_n[k], prep, coef, c->recon_coop_below,
                                                      wide, dst_twin, side && !on_main ? c->side[c->recon_pair_first + lane] : c->stream);
                if (!on_main && n_ps) lane = (lane + 1) % n_ps;
            }
        for (int j = 0; j < n_ps; j++) (void) hipEventRecord(c->ev_pair[j], c->side[c->recon_pair_first + j]);
        if (rc) return rc;
        paired_on_side = side;
        if (!l->inter->mc->n && !l->inter->comp->n && !l->itx->n) {
            for (int j = 0; j < n_ps; j++) (void) hipStreamWaitEvent(c->stream, c->ev_pair[j], 0);
            return 0;
        }
    }
    const Dav1dHipMcList *ml = l->inter->mc;
    // c->recon_pipeline = smallest residual list worth two streams (0: always pipeline, -1: never)
    const long min_tasks = c->recon_pipeline;
    auto join_paired = [&]() {
        if (paired_on_side) for (int j = 0; j < n_ps; j++) (void) hipStreamWaitEvent(c->stream, c->ev_pair[j], 0);
    };
    if (!dst_twin && (min_tasks < 0 || !c->concurrent || mc_fused_min_bin() < MC_BINS || (long) l->itx->n < min_tasks)) {
        int rc = dav1d_hip_inter_list_run(c, l->inter, dst, refs, n_refs, prep, mask);
        if (!rc) rc = dav1d_hip_itx_list_run(c, l->itx, dst, coef);
        join_paired();
        return rc;
    }
    if (n_refs < 1 || n_refs > 8 || (ml->n && ml->max_ref >= n_refs)) { join_paired(); return -EINVAL; }
    const DevPlanes dp = dev_planes(dst);
    DevPlanes rp[8];
    for (int i = 0; i < n_refs; i++) if (refs[i].bpc != dst->bpc) { join_paired(); return -EINVAL; }
    if (const int rv = ref_planes(c, refs, n_refs, rp)) { join_paired(); return rv; }
    int rc = mc_regroup(c, const_cast<Dav1dHipMcList *>(ml), rp, n_refs);
    if (rc) { join_paired(); return rc; }
    // DAV1D_HIP_RECON_LANES: side streams the residual launches are dealt over.  Measured (8K 10-bit): 1 lane 0.379 ms,
    // 2 lanes 0.394, 3 lanes 0.407, 5 lanes 0.420 per frame — residual launches running next to each other take bandwidth from
    // the predictions they are waiting for; one in-order residual stream keeps the pipeline a pipeline.
    const int n_lanes = paired_on_side ? 1      // side streams 1 and 2 carry the paired launches
                      : std::max(1, std::min((int) Dav1dHipContext::N_SIDE, c->recon_lanes));
    hipStream_t sm = c->stream;
    (void) hipEventRecord(c->ev_fork, sm);
    for (int i = 0; i < n_lanes; i++) (void) hipStreamWaitEvent(c->side[i], c->ev_fork, 0);
    // The prediction launches run in order on one stream, so a residual launch only has to wait for the LAST launch it depends
    // on — and only those launches get an event (a cross-stream event is a cache release / acquire: not free).
    static const uint8_t order[19] = { 4, 11, 12, 17, 18, 3, 9, 10, 15, 16, 2, 7, 8, 13, 14, 1, 5, 6, 0 };
    int seq[16], n_seq = 0;                              // launch order of the prediction side: bins descending, then the compound launch
    // largest tile shape first (measured: 16x16 first is as good, smallest first 8 % slower: its residuals are the shortest
    // and leave the long 64- and 32-point transforms for a tail that nothing overlaps)
    for (int b = MC_BINS - 1; b >= 0; b--) if (ml->off[b + 1] > ml->off[b]) seq[n_seq++] = b;
    if (l->inter->comp->n) seq[n_seq++] = 15;
    int last_dep[19];                                    // per transform size: position in seq[] of its last dependency, -1 none
    bool wanted[16] = { false };
    for (int b = 0; b < 19; b++) {
        last_dep[b] = -1;
        if (l->itx->off[b + 1] == l->itx->off[b]) continue;
        for (int k = 0; k < n_seq; k++) if (l->dep[b] >> seq[k] & 1) last_dep[b] = k;
        if (last_dep[b] >= 0) wanted[last_dep[b]] = true;
    }
    for (int k = 0; k < n_seq && !rc; k++) {
        const int b = seq[k];
        if (b == 15) rc = dav1d_hip_comp_list_run(c, l->inter->comp, dst, prep, mask);
        else if (dst_twin) rc = dav1d_hip_launch_mc_bin_twin(&dp, rp, n_refs, dst->bpc, b, ml->dev + ml->off[b], (int) (ml->off[b + 1] - ml->off[b]), prep, dst_twin, sm);
        else rc = dav1d_hip_launch_mc_bin(&dp, rp, n_refs, dst->bpc, b, ml->dev + ml->off[b], (int) (ml->off[b + 1] - ml->off[b]), prep, sm);
        if (wanted[k]) (void) hipEventRecord(c->ev_bin[k], sm);
    }
    int waited[Dav1dHipContext::N_SIDE];
    for (int i = 0; i < Dav1dHipContext::N_SIDE; i++) waited[i] = -1;
    int lane = 0;
    // residual launches in the order their predictions become ready (ties: largest transform first)
    int iorder[19];
    for (int k = 0; k < 19; k++) iorder[k] = order[k];
    std::stable_sort(iorder, iorder + 19, [&](int p, int q) { return last_dep[p] < last_dep[q]; });
    for (int k = 0; k < 19 && !rc; k++) {
        const int b = iorder[k];
        const size_t cnt = l->itx->off[b + 1] - l->itx->off[b];
        if (!cnt) continue;
        hipStream_t si = c->side[lane];
        if (last_dep[b] > waited[lane]) {                // a lane is in order too: an earlier wait covers everything before it
            (void) hipStreamWaitEvent(si, c->ev_bin[last_dep[b]], 0);
            waited[lane] = last_dep[b];
        }
        rc = dav1d_hip_launch_itx_bin_out(&dp, dst->bpc, b, l->itx->dev + l->itx->off[b], (int) cnt, coef, wide, dst_twin, si);
        lane = (lane + 1) % n_lanes;
    }
    for (int i = 0; i < n_lanes; i++) {
        (void) hipEventRecord(c->ev_join[i], c->side[i]);
        (void) hipStreamWaitEvent(sm, c->ev_join[i], 0);
    }
    join_paired();
    return rc;
}

// Measurement aid: the launches of a recon list one after the other on the context's stream, each bracketed by events.
// ms / counts: [0..4] the paired launches by size class (blocks), [5..19] the prediction launches by tile shape (tiles), [20] the
// compound / blend launch (tasks), [21..39] the residual launches by transform size (blocks).
static int recon_list_run_timed_impl(Dav1dHipContext *c, const Dav1dHipReconList *l, const Dav1dHipPicture *dst,
                                     const Dav1dHipPicture *refs, int n_refs, int16_t *prep, uint8_t *mask, void *coef,
                                     float *ms, size_t *counts, const DevPlanes *dst_twin);
int dav1d_hip_recon_list_run_timed(Dav1dHipContext *c, const Dav1dHipReconList *l, const Dav1dHipPicture *dst,
                                   const Dav1dHipPicture *refs, int n_refs, int16_t *prep, uint8_t *mask, void *coef,
                                   float *ms, size_t *counts) {
    return recon_list_run_timed_impl(c, l, dst, refs, n_refs, prep, mask, coef, ms, counts, nullptr);
}
// the launches of dav1d_hip_recon_list_run_tiled the same way (-ENOTSUP when the list cannot run with its picture in the twin only)
int dav1d_hip_recon_list_run_tiled_timed(Dav1dHipContext *c, const Dav1dHipReconList *l, Dav1dHipPicture *dst,
                                         const Dav1dHipPicture *refs, int n_refs, int16_t *prep, uint8_t *mask, void *coef,
                                         float *ms, size_t *counts) {
    if (!c || !l || !dst || !refs) return -EINVAL;
    if (!dst->twin[0] && !dst->twin_alloc) { const int rc = dav1d_hip_picture_twin_alloc(c, dst); if (rc) return rc; }
    if (!recon_list_twin_direct(c, l, dst, refs, n_refs)) return -ENOTSUP;
    if (!dst->twin_ok) { const int rc = dav1d_hip_picture_retile(c, dst); if (rc) return rc; }
    DevPlanes tw = dev_planes(dst);
    for (int p = 0; p < 3; p++) tw.data[p] = dst->p[p].data ? dst->twin[p] : nullptr;
    tw.tiled = 2;
    const int rc = recon_list_run_timed_impl(c, l, dst, refs, n_refs, prep, mask, coef, ms, counts, &tw);
    dst->twin_ok = rc ? 0 : DAV1D_HIP_TWIN_ONLY;
    return rc;
}
static int recon_list_run_timed_impl(Dav1dHipContext *c, const Dav1dHipReconList *l, const Dav1dHipPicture *dst,
                                     const Dav1dHipPicture *refs, int n_refs, int16_t *prep, uint8_t *mask, void *coef,
                                     float *ms, size_t *counts, const DevPlanes *dst_twin) {
    if (!c || !l || !dst || !refs || !ms || !counts || n_refs < 1 || n_refs > 8) return -EINVAL;
    const Dav1dHipMcList *ml = l->inter->mc;
    if ((ml->n && ml->max_ref >= n_refs) || l->f_max_ref >= n_refs) return -EINVAL;
    const DevPlanes dp = dev_planes(dst);
    DevPlanes rp[8];
    if (const int rv = ref_planes(c, refs, n_refs, rp)) return rv;
    int rc = mc_regroup(c, const_cast<Dav1dHipMcList *>(ml), rp, n_refs);
    if (rc) return rc;
    enum { N = 40 };
    hipEvent_t ev[N + 1];
    for (int k = 0; k <= N; k++) HIP_TRY(hipEventCreate(&ev[k]));
    HIP_TRY(hipEventRecord(ev[0], c->stream));
    for (int k = 0; k < N && !rc; k++) {
        size_t cnt = 0;
        if (k < 5) {
            cnt = l->f_n[k];
            if (cnt) rc = dav1d_hip_launch_recon_fused_out(&dp, rp, n_refs, dst->bpc, k, l->f_tiles[k], l->f_tasks[k], (int) cnt, prep, coef, c->recon_coop_below,
                                                           dst_twin != nullptr, dst_twin, c->stream);
        } else if (k < 20) {
            const int b = k - 5;
            cnt = ml->off[b + 1] - ml->off[b];
            if (cnt && dst_twin) rc = dav1d_hip_launch_mc_bin_twin(&dp, rp, n_refs, dst->bpc, b, ml->dev + ml->off[b], (int) cnt, prep, dst_twin, c->stream);
            else if (cnt) rc = dav1d_hip_launch_mc_bin(&dp, rp, n_refs, dst->bpc, b, ml->dev + ml->off[b], (int) cnt, prep, c->stream);
        } else if (k == 20) {
            cnt = l->inter->comp->n;
            if (cnt) rc = dav1d_hip_comp_list_run(c, l->inter->comp, dst, prep, mask);
        } else {
            const int b = k - 21;
            cnt = l->itx->off[b + 1] - l->itx->off[b];
            if (cnt) rc = dav1d_hip_launch_itx_bin_out(&dp, dst->bpc, b, l->itx->dev + l->itx->off[b], (int) cnt, coef, dst_twin != nullptr, dst_twin, c->stream);
        }
        counts[k] = cnt;
        (void) hipEventRecord(ev[k + 1], c->stream);
    }
    (void) hipStreamSynchronize(c->stream);
    for (int k = 0; k < N; k++) { ms[k] = 0.f; (void) hipEventElapsedTime(&ms[k], ev[k], ev[k + 1]); }
    for (int k = 0; k <= N; k++) (void) hipEventDestroy(ev[k]);
    return rc;
}

// ------------------------------------------------------------------ intra wavefront list
//
// The batches (wavefront steps) of an intra frame with both halves of every block: predictions and residuals.  A 4x4 or 8x8
// block whose residual covers exactly its prediction runs as a pair in one wave (intra_pair.hip); the other blocks of the
// step keep the prediction launch + residual launch route.  run_batch() only enqueues: at most three launches per step, one
// for the steps that hold nothing but small blocks (the second half of every superblock's wavefront).
struct Dav1dHipIntraList {
    Dav1dHipIpredList *preds;                 // unpaired predictions, batch by batch
    std::vector<Dav1dHipItxList *> itx;       // unpaired residuals, one list per batch
    Dav1dHipIpredTask *p_dev;                 // paired blocks of all batches: predictions ...
    Dav1dHipItxTask *t_dev;                   // ... and their residuals, same order
    std::vector<size_t> pair_start;           // batch k = pairs [pair_start[k], pair_start[k + 1])
    Dav1dHipCompTask *b_dev;                  // inter-intra blends of all batches (run between a batch's predictions and residuals)
    std::vector<size_t> blend_start;          // batch k = blends [blend_start[k], blend_start[k + 1])
    bool needs_aux;
};

void dav1d_hip_intra_list_destroy(Dav1dHipContext *c, Dav1dHipIntraList *l) {
    if (!l) return;
    if (l->preds) dav1d_hip_ipred_list_destroy(c, l->preds);
    for (Dav1dHipItxList *t : l->itx) if (t) dav1d_hip_itx_list_destroy(c, t);
    hipStreamSynchronize(c->stream);
    if (l->p_dev) hipFree(l->p_dev);
    if (l->t_dev) hipFree(l->t_dev);
    if (l->b_dev) hipFree(l->b_dev);
    delete l;
}

int dav1d_hip_intra_list_create(Dav1dHipContext *c, Dav1dHipIntraList **out, const Dav1dHipIpredTask *preds, const size_t *pred_sizes,
                                const Dav1dHipItxTask *txs, const size_t *tx_sizes, size_t n_batches) {
    return dav1d_hip_intra_list_create_blend(c, out, preds, pred_sizes, txs, tx_sizes, nullptr, nullptr, n_batches);
}

int dav1d_hip_intra_list_create_blend(Dav1dHipContext *c, Dav1dHipIntraList **out, const Dav1dHipIpredTask *preds, const size_t *pred_sizes,
                                      const Dav1dHipItxTask *txs, const size_t *tx_sizes, const Dav1dHipCompTask *blends,
                                      const size_t *blend_sizes, size_t n_batches) {
    if (!c || !out || !pred_sizes || !tx_sizes) return -EINVAL;
    *out = nullptr;
    size_t np = 0, nt = 0;
    for (size_t k = 0; k < n_batches; k++) { np += pred_sizes[k]; nt += tx_sizes[k]; }
    if ((np && !preds) || (nt && !txs)) return -EINVAL;
    uint8_t dummy = 0;
    if (ipred_tasks_valid(preds, np, &dummy)) return -EINVAL;
    for (size_t i = 0; i < nt; i++) if (!itx_task_ok(txs[i])) return -EINVAL;
    Dav1dHipIntraList *l = new (std::nothrow) Dav1dHipIntraList();
    if (!l) return -ENOMEM;
    l->preds = nullptr; l->p_dev = nullptr; l->t_dev = nullptr; l->b_dev = nullptr; l->needs_aux = false;
    for (size_t i = 0; i < np; i++) if (preds[i].kind >= DAV1D_HIP_IPRED_PAL && preds[i].kind < DAV1D_HIP_IPRED_PRED_TMP) l->needs_aux = true;
    l->blend_start.push_back(0);
    for (size_t k = 0; k < n_batches; k++) l->blend_start.push_back(l->blend_start.back() + (blend_sizes ? blend_sizes[k] : 0));
    if (l->blend_start.back()) {
        const size_t nb = l->blend_start.back();
        for (size_t i = 0; i < nb; i++)
            if (!blends || blends[i].kind != DAV1D_HIP_COMP_BLEND || blends[i].plane > 2 || blends[i].w < 4 || blends[i].h < 4) { delete l; return -EINVAL; }
        if (hipMalloc((void **) &l->b_dev, nb * sizeof(Dav1dHipCompTask)) != hipSuccess) { delete l; return -ENOMEM; }
        const int brc = dav1d_hip_upload(c, l->b_dev, blends, nb * sizeof(Dav1dHipCompTask));
        if (brc) { hipFree(l->b_dev); delete l; return brc; }
    }
    static const bool pairing = !(getenv("DAV1D_HIP_INTRA_PAIR") && !atoi(getenv("DAV1D_HIP_INTRA_PAIR")));
    std::vector<Dav1dHipIpredTask> rest_p, pair_p;
    std::vector<Dav1dHipItxTask> pair_t;
    std::vector<size_t> rest_p_sizes;
    int rc = 0;
    size_t p0 = 0, t0 = 0;
    l->pair_start.push_back(0);
    for (size_t k = 0; k < n_batches && !rc; k++) {
        std::unordered_map<uint64_t, size_t> tx_at;
        for (size_t i = 0; i < tx_sizes[k]; i++) {
            const Dav1dHipItxTask &t = txs[t0 + i];
            if (pairing && t.tx <= 1) tx_at[(uint64_t) t.plane << 32 | t.dst_off] = i;
        }
        std::vector<char> taken(tx_sizes[k], 0);
        size_t n_rest = 0;
        for (size_t i = 0; i < pred_sizes[k]; i++) {
            const Dav1dHipIpredTask &p = preds[p0 + i];
            long j = -1;
            if (p.tw == p.th && p.tw <= 2 && p.kind <= DAV1D_HIP_IPRED_PAL) {
                auto it = tx_at.find((uint64_t) p.plane << 32 | p.dst_off);
                if (it != tx_at.end() && !taken[it->second] && txs[t0 + it->second].tx == p.tw - 1) j = (long) it->second;
            }
            if (j >= 0) {
                taken[j] = 1;
                pair_p.push_back(p);
                pair_t.push_back(txs[t0 + j]);
                itx_fill_prefix(pair_t.back());
            } else {
                rest_p.push_back(p);
                n_rest++;
            }
        }
        rest_p_sizes.push_back(n_rest);
        l->pair_start.push_back(pair_p.size());
        std::vector<Dav1dHipItxTask> rest_t;
        for (size_t i = 0; i < tx_sizes[k]; i++) if (!taken[i]) rest_t.push_back(txs[t0 + i]);
        Dav1dHipItxList *tl = nullptr;
        rc = dav1d_hip_itx_list_create(c, &tl, rest_t.data(), rest_t.size());
        l->itx.push_back(tl);
        p0 += pred_sizes[k]; t0 += tx_sizes[k];
    }
    if (!rc) rc = dav1d_hip_ipred_list_create(c, &l->preds, rest_p.data(), rest_p_sizes.data(), n_batches);
    if (!rc && !pair_p.empty()) {
        if (hipMalloc((void **) &l->p_dev, pair_p.size() * sizeof(Dav1dHipIpredTask)) != hipSuccess ||
            hipMalloc((void **) &l->t_dev, pair_t.size() * sizeof(Dav1dHipItxTask)) != hipSuccess) rc = -ENOMEM;
        if (!rc) rc = dav1d_hip_upload(c, l->p_dev, pair_p.data(), pair_p.size() * sizeof(Dav1dHipIpredTask));
        if (!rc) rc = dav1d_hip_upload(c, l->t_dev, pair_t.data(), pair_t.size() * sizeof(Dav1dHipItxTask));
    }
    if (rc) { dav1d_hip_intra_list_destroy(c, l); return rc; }
    *out = l;
    return 0;
}

// ------------------------------------------------------------------ intra dataflow launch (intra_flow.hip)
struct Dav1dHipIntraFlow {
    IntraUnit *units;
    uint32_t *ctr;              // [0 .. 31]: error word; then FLOW_SUB counters of FLOW_SUB_STRIDE words per group
    size_t ctr_bytes;
    size_t n_units, n_steps, n_groups;
    bool needs_aux;
};

void dav1d_hip_intra_flow_destroy(Dav1dHipContext *c, Dav1dHipIntraFlow *l) {
    if (!l) return;
    hipStreamSynchronize(c->stream);
    if (l->units) hipFree(l->units);
    if (l->ctr) hipFree(l->ctr);
    delete l;
}
size_t dav1d_hip_intra_flow_units(const Dav1dHipIntraFlow *l) { return l ? l->n_units : 0; }
// after a run: tickets drawn, units finished, waves that gave up waiting (0 unless something is broken); synchronizes
int dav1d_hip_intra_flow_status(Dav1dHipContext *c, const Dav1dHipIntraFlow *l, uint32_t out[3]) {
    if (!c || !l || !out) return -EINVAL;
    // out[0]: unused (tickets are static), out[1]: units finished (sum of every group's counters), out[2]: waves that gave up
    std::vector<uint32_t> w(l->ctr_bytes / 4);
    const int rc = dav1d_hip_download(c, w.data(), l->ctr, l->ctr_bytes);
    uint64_t done = 0;
    for (size_t g = 0; g < l->n_groups; g++)
        for (int k = 0; k < FLOW_SUB; k++) done += w[32 + (g * FLOW_SUB + k) * FLOW_SUB_STRIDE];
    out[0] = 0; out[1] = (uint32_t) done; out[2] = w[0];
    return rc;
}

// Units of one set of tasks sorted by step (*_end[s] = end of step s): per step first the predictions, each with the residual
// of the same rectangle when there is one (that is how the reference walks an intra block: predict a transform block, add
// its residual, next one), then the residuals without a prediction of their own — those wait for every prediction of their
// step (a palette block: one prediction, many residuals).  need is left 0.  -ENOTSUP: a task kind the dataflow launch
// does not run (PRED_TMP of inter-intra blocks, the DSP-level kinds).
int dav1d_hip_intra_units_build(const Dav1dHipIpredTask *preds, const uint32_t *pred_end, const Dav1dHipItxTask *txs, const uint32_t *tx_end,
                                size_t n_steps, std::vector<IntraUnit> &units, std::vector<uint32_t> &ua_end, std::vector<uint32_t> &ub_end,
                                const Dav1dHipCompTask *blends, const uint32_t *blend_end) {
    const size_t np = n_steps ? pred_end[n_steps - 1] : 0, nt = n_steps ? tx_end[n_steps - 1] : 0;
    uint8_t dummy = 0;
    if (ipred_tasks_valid(preds, np, &dummy)) return -EINVAL;
    for (size_t i = 0; i < nt; i++) if (!itx_task_ok(txs[i])) return -EINVAL;
    // inter-intra blocks (kind PRED_TMP + a BLEND of the same rectangle in the same step) only where the caller brings the blends: the
    // unit then carries the blend — its mask offset in the place of the scratch offset nobody needs when the prediction stays in LDS
    for (size_t i = 0; i < np; i++) {
        const int k = preds[i].kind;
        if (k == DAV1D_HIP_IPRED_PRED_TMP && blends) continue;
        if (k != DAV1D_HIP_IPRED_PRED && k != DAV1D_HIP_IPRED_CFL && k != DAV1D_HIP_IPRED_PAL && k != DAV1D_HIP_IPRED_COPY) return -ENOTSUP;
    }
    if (blends) {
        const size_t nb = n_steps ? blend_end[n_steps - 1] : 0;
        for (size_t i = 0; i < nb; i++) if (blends[i].kind != DAV1D_HIP_COMP_BLEND || blends[i].plane > 2) return -ENOTSUP;
    }
    units.clear();
    units.reserve(np + nt / 4);
    ua_end.assign(n_steps, 0); ub_end.assign(n_steps, 0);
    auto unit = [&](const Dav1dHipIpredTask *p, const Dav1dHipItxTask *t) {
        IntraUnit u;
        memset(&u, 0, sizeof(u));
        if (p) { u.p = *p; u.has |= 1; }
        if (t) { u.t = *t; itx_fill_prefix(u.t); u.has |= 2; }
        units.push_back(u);
    };
    std::vector<uint32_t> slot;        // open-addressed map (plane, dst_off) -> transform task of the step
    std::vector<char> taken;
    for (size_t k = 0; k < n_steps; k++) {
        const size_t p0 = k ? pred_end[k - 1] : 0, t0 = k ? tx_end[k - 1] : 0;
        const size_t npk = pred_end[k] - p0, ntk = tx_end[k] - t0;
        if (npk || ntk) {
            size_t cap = 16;
            while (cap < 2 * ntk + 2) cap <<= 1;
            slot.assign(cap, 0xffffffffu);
            taken.assign(ntk, 0);
            auto hash = [&](uint32_t plane, uint32_t off) { return (size_t) ((off * 2654435761u) ^ (plane * 0x9e3779b9u)) & (cap - 1); };
            for (size_t i = 0; i < ntk; i++) {
                const Dav1dHipItxTask &t = txs[t0 + i];
                size_t h = hash(t.plane, t.dst_off);
                while (slot[h] != 0xffffffffu) h = (h + 1) & (cap - 1);
                slot[h] = (uint32_t) i;
            }
            for (size_t i = 0; i < npk; i++) {
                const Dav1dHipIpredTask &p = preds[p0 + i];
                uint32_t j = 0xffffffffu;
                for (size_t h = hash(p.plane, p.dst_off); slot[h] != 0xffffffffu; h = (h + 1) & (cap - 1)) {
                    const Dav1dHipItxTask &t = txs[t0 + slot[h]];
                    if (t.plane == p.plane && t.dst_off == p.dst_off && !taken[slot[h]] && k_tx_w[t.tx] == p.tw * 4 && k_tx_h[t.tx] == p.th * 4) {
                        j = slot[h];
                        break;
                    }
                }
                if (j != 0xffffffffu) taken[j] = 1;
                unit(&p, j == 0xffffffffu ? nullptr : &txs[t0 + j]);
                if (p.kind == DAV1D_HIP_IPRED_PRED_TMP) {
                    // its blend: the one of the step with the same rectangle
                    const size_t b0 = k ? blend_end[k - 1] : 0, b1 = blend_end[k];
                    const Dav1dHipCompTask *bl = nullptr;
                    for (size_t q = b0; q < b1 && !bl; q++)
                        if (blends[q].plane == p.plane && blends[q].dst_off == p.dst_off && blends[q].w == p.tw * 4 && blends[q].h == p.th * 4) bl = &blends[q];
                    if (!bl) return -EINVAL;
                    units.back().has |= 4;
                    units.back().p.aux_off = bl->mask_off;
                }
            }
            // units of a group are independent: put those that run the same code (transform size, then predictor) next to each
            // other, so that the waves of a CU — which are dealt consecutive units — share instruction cache lines.  The launch's
            // code is several hundred KB; with mixed sizes every wave misses on its own path.  Speed only.
            std::stable_sort(units.begin() + (k ? ub_end[k - 1] : 0), units.end(), [](const IntraUnit &a, const IntraUnit &b) {
                const int ka = ((a.has & 2) ? a.t.tx : 31) << 8 | a.p.mode, kb = ((b.has & 2) ? b.t.tx : 31) << 8 | b.p.mode;
                return ka < kb;
            });
            ua_end[k] = (uint32_t) units.size();
            for (size_t i = 0; i < ntk; i++) if (!taken[i]) unit(nullptr, &txs[t0 + i]);
        } else {
            ua_end[k] = (uint32_t) units.size();
        }
        ub_end[k] = (uint32_t) units.size();
    }
    return 0;
}

// units (host, need set, sorted) -> device-resident list; grp / prev_n are filled in here (the array is the caller's scratch)
int dav1d_hip_intra_flow_from_units(Dav1dHipContext *c, Dav1dHipIntraFlow **out, IntraUnit *units, size_t n) {
    if (!c || !out || (!units && n)) return -EINVAL;
    *out = nullptr;
    Dav1dHipIntraFlow *l = new (std::nothrow) Dav1dHipIntraFlow();
    if (!l) return -ENOMEM;
    memset(l, 0, sizeof(*l));
    l->n_units = n;
    for (size_t i = 0; i < n && !l->needs_aux; i++) if ((units[i].has & 1) && units[i].p.kind == DAV1D_HIP_IPRED_PAL) l->needs_aux = true;
    // groups: runs of equal `need`; the device copy gets the group index and the size of the group before
    size_t groups = 0, prev_n = 0, cur_start = 0;
    for (size_t i = 0; i < n; i++) {
        if (i && units[i].need != units[i - 1].need) { prev_n = i - cur_start; cur_start = i; groups++; }
        units[i].grp = (uint32_t) groups;
        units[i].prev_n = (uint32_t) prev_n;
    }
    l->n_groups = n ? groups + 1 : 0;
    l->ctr_bytes = (32 + l->n_groups * FLOW_SUB * FLOW_SUB_STRIDE) * sizeof(uint32_t);
    int rc = 0;
    if (hipMalloc((void **) &l->ctr, l->ctr_bytes) != hipSuccess) rc = -ENOMEM;
    if (!rc && n) {
        // one record past the end: the waves fetch a unit ahead
        if (hipMalloc((void **) &l->units, (n + 1) * sizeof(IntraUnit)) != hipSuccess) rc = -ENOMEM;
        if (!rc) rc = dav1d_hip_upload(c, l->units, units, n * sizeof(IntraUnit));
    }
    if (rc) { dav1d_hip_intra_flow_destroy(c, l); return rc; }
    *out = l;
    return 0;
}

int dav1d_hip_intra_flow_create(Dav1dHipContext *c, Dav1dHipIntraFlow **out, const Dav1dHipIpredTask *preds, const size_t *pred_sizes,
                                const Dav1dHipItxTask *txs, const size_t *tx_sizes, size_t n_batches) {
    if (!c || !out || !pred_sizes || !tx_sizes) return -EINVAL;
    *out = nullptr;
    std::vector<uint32_t> pe(n_batches), te(n_batches), ua, ub;
    size_t np = 0, nt = 0;
    for (size_t k = 0; k < n_batches; k++) {
        np += pred_sizes[k]; nt += tx_sizes[k];
        if (np >= 0xffffffffu || nt >= 0xffffffffu) return -ENOTSUP;
        pe[k] = (uint32_t) np; te[k] = (uint32_t) nt;
    }
    if ((np && !preds) || (nt && !txs)) return -EINVAL;
    std::vector<IntraUnit> units;
    const int rc = dav1d_hip_intra_units_build(preds, pe.data(), txs, te.data(), n_batches, units, ua, ub, nullptr, nullptr);
    if (rc) return rc;
    for (size_t k = 0, i = 0; k < n_batches; k++) {
        const uint32_t need_a = (uint32_t) i, need_b = ua[k];
        for (; i < ua[k]; i++) units[i].need = need_a;
        for (; i < ub[k]; i++) units[i].need = need_b;
    }
    return dav1d_hip_intra_flow_from_units(c, out, units.data(), units.size());
}

// enqueues: counters to zero, then the launch
int dav1d_hip_intra_flow_run(Dav1dHipContext *c, const Dav1dHipIntraFlow *l, const Dav1dHipPicture *dst, void *coef, uint8_t *aux) {
    if (!c || !l || !dst || (l->needs_aux && !aux)) return -EINVAL;
    if (!l->n_units) return 0;
    if (hipMemsetAsync(l->ctr, 0, l->ctr_bytes, c->stream) != hipSuccess) return -EIO;
    const DevPlanes dp = dev_planes(dst);
    // 8 one-wave workgroups per CU: more waves only poll
    return dav1d_hip_launch_intra_flow(&dp, dst->bpc, dst->layout, l->units, (int) l->n_units, aux, coef, l->ctr, c->flow_groups, c->flow_mode,
                                       c->stream);
}

// ------------------------------------------------------------------ intra wavefront superblock by superblock (intra_sb.hip)
struct Dav1dHipIntraSb {
    IntraUnit *units;
    SbRegion *regions;
    uint32_t *flags;            // n_regions + 1 words for the one-launch form
    std::vector<uint32_t> level_start;
    size_t n_units, n_regions;
    int sb_log2, sbw;
    bool needs_aux;
    bool has_copies;            // intra block copies among the units: the L2 hand-off kernel only, and (one launch) the `where` table
    uint32_t *where;            // superblock (raster) -> its region, for the copies' waits in the one-launch form
};

void dav1d_hip_intra_sb_destroy(Dav1dHipContext *c, Dav1dHipIntraSb *l) {
    if (!l) return;
    hipStreamSynchronize(c->stream);
    if (l->units) hipFree(l->units);
    if (l->regions) hipFree(l->regions);
    if (l->flags) hipFree(l->flags);
    if (l->where) hipFree(l->where);
    delete l;
}
size_t dav1d_hip_intra_sb_levels(const Dav1dHipIntraSb *l) { return l && !l->level_start.empty() ? l->level_start.size() - 1 : 0; }
size_t dav1d_hip_intra_sb_superblocks(const Dav1dHipIntraSb *l) { return l ? l->n_regions : 0; }

int dav1d_hip_intra_sb_create(Dav1dHipContext *c, Dav1dHipIntraSb **out, const Dav1dHipIpredTask *preds, const size_t *pred_sizes,
                              const Dav1dHipItxTask *txs, const size_t *tx_sizes, size_t n_batches, const Dav1dHipPicture *geometry,
                              int sb128, int n_tile_cols, const uint16_t *col_start_sb, int n_tile_rows, const uint16_t *row_start_sb) {
    if (!c || !out || !pred_sizes || !tx_sizes || !geometry) return -EINVAL;
    *out = nullptr;
    SbTiling tl;
    int rc = dav1d_hip_sb_tiling_make(&tl, geometry->p[0].w, geometry->p[0].h, sb128, n_tile_cols, col_start_sb, n_tile_rows, row_start_sb);
    if (rc) return rc;
    std::vector<uint32_t> pe(n_batches), te(n_batches), ua, ub;
    size_t np = 0, nt = 0;
    for (size_t k = 0; k < n_batches; k++) {
        np += pred_sizes[k]; nt += tx_sizes[k];
        if (np >= 0xffffffffu || nt >= 0xffffffffu) return -ENOTSUP;
        pe[k] = (uint32_t) np; te[k] = (uint32_t) nt;
    }
    if ((np && !preds) || (nt && !txs)) return -EINVAL;
    std::vector<IntraUnit> units;
    rc = dav1d_hip_intra_units_build(preds, pe.data(), txs, te.data(), n_batches, units, ua, ub, nullptr, nullptr);
    if (rc) return rc;
    const DevPlanes dp = dev_planes(geometry);
    SbSort st;
    rc = dav1d_hip_sbw_prepare(units, ua, ub, tl, dp.stride, geometry->layout != DAV1D_HIP_LAYOUT_I444, geometry->layout == DAV1D_HIP_LAYOUT_I420, st);
    if (rc) return rc;
    std::vector<IntraUnit> sorted(st.n_records);
    dav1d_hip_sbw_emit(units, st, sorted.data());
    bool has_pal = false;
    for (const IntraUnit &u : units) if ((u.has & 1) && u.p.kind == DAV1D_HIP_IPRED_PAL) { has_pal = true; break; }
    units.swap(sorted);
    const std::vector<SbPart> &parts = st.parts;
    SbPlan plan;
    std::sort(st.copy_deps.begin(), st.copy_deps.end());
    st.copy_deps.erase(std::unique(st.copy_deps.begin(), st.copy_deps.end()), st.copy_deps.end());
    rc = dav1d_hip_sbw_plan(tl, { &parts }, { 0 }, nullptr, plan, st.copy_deps.empty() ? nullptr : &st.copy_deps);
    if (rc) return rc;
    Dav1dHipIntraSb *l = new (std::nothrow) Dav1dHipIntraSb();
    if (!l) return -ENOMEM;
    l->units = nullptr; l->regions = nullptr; l->flags = nullptr; l->where = nullptr;
    l->has_copies = !st.copy_deps.empty(); l->sbw = tl.sbw;
    l->n_units = units.size(); l->n_regions = plan.regions.size();
    l->level_start = plan.level_start;
    l->sb_log2 = tl.sb_log2;
    l->needs_aux = has_pal;
    if (l->n_units) {
        if (hipMalloc((void **) &l->units, (l->n_units + 1) * sizeof(IntraUnit)) != hipSuccess) rc = -ENOMEM;
        if (!rc && hipMalloc((void **) &l->regions, l->n_regions * sizeof(SbRegion)) != hipSuccess) rc = -ENOMEM;
        if (!rc && hipMalloc((void **) &l->flags, (l->n_regions + 1) * sizeof(uint32_t)) != hipSuccess) rc = -ENOMEM;
        if (!rc) rc = dav1d_hip_upload(c, l->units, units.data(), l->n_units * sizeof(IntraUnit));
        if (!rc) rc = dav1d_hip_upload(c, l->regions, plan.regions.data(), l->n_regions * sizeof(SbRegion));
        if (!rc && l->has_copies) {
            if (hipMalloc((void **) &l->where, plan.where.size() * sizeof(uint32_t)) != hipSuccess) rc = -ENOMEM;
            if (!rc) rc = dav1d_hip_upload(c, l->where, plan.where.data(), plan.where.size() * sizeof(uint32_t));
        }
    }
    if (rc) { dav1d_hip_intra_sb_destroy(c, l); return rc; }
    *out = l;
    return 0;
}

// enqueues the launches on the context's stream: one per level, or (option intra_sb_flow, L2 hand-off form, more than one level) ONE
// for all of them with the superblocks waiting for their neighbours' flags
int dav1d_hip_intra_sb_run(Dav1dHipContext *c, const Dav1dHipIntraSb *l, const Dav1dHipPicture *dst, void *coef, uint8_t *aux) {
    if (!c || !l || !dst || (l->needs_aux && !aux)) return -EINVAL;
    const DevPlanes dp = dev_planes(dst);
    int rc = 0;
    const int lds = c->intra_sb_lds && !l->has_copies;          // (the LDS-resident form does not copy)
    if (c->intra_sb_flow && !lds && l->level_start.size() > 2) {
        if (hipMemsetAsync(l->flags, 0, (l->n_regions + 1) * sizeof(uint32_t), c->stream) != hipSuccess) return -EIO;
        // (four waves per workgroup where the levels are wide and nothing is copied, unless asked otherwise: see frame.hip)
        const bool wide_levels = l->n_regions >= 128 * (l->level_start.size() - 1);
        return dav1d_hip_launch_intra_sb(&dp, dst->bpc, dst->layout, l->units, l->regions, (int) l->n_regions, aux, nullptr, coef,
                                         c->intra_sb_waves ? c->intra_sb_waves : !l->has_copies && wide_levels ? 4 : 8, l->sb_log2, 0, l->flags, c->stream, l->where, l->sbw);
    }
    for (size_t k = 0; k + 1 < l->level_start.size() && !rc; k++)
        rc = dav1d_hip_launch_intra_sb(&dp, dst->bpc, dst->layout, l->units, l->regions + l->level_start[k],
                                       (int) (l->level_start[k + 1] - l->level_start[k]), aux, nullptr, coef, c->intra_sb_waves, l->sb_log2, lds, nullptr, c->stream);
    return rc;
}

// after dav1d_hip_intra_sb_run: waits for the stream; workgroups of the one-launch form that gave up waiting for a neighbour (none in
// a sound run, see include/dav1d_hip.h) left their superblocks unreconstructed -> -EIO
int dav1d_hip_intra_sb_status(Dav1dHipContext *c, const Dav1dHipIntraSb *l, uint32_t *gave_up) {
    if (!c || !l) return -EINVAL;
    uint32_t n = 0;
    if (gave_up) *gave_up = 0;
    if (!l->flags || !l->n_units) return dav1d_hip_sync(c);
    const int rc = dav1d_hip_download(c, &n, l->flags + l->n_regions, sizeof(n));
    if (rc) return rc;
    if (!(c->intra_sb_flow && !(c->intra_sb_lds && !l->has_copies) && l->level_start.size() > 2)) n = 0;      // the flags are only written by the one-launch form
    if (gave_up) *gave_up = n;
    return n ? -EIO : 0;
}

int dav1d_hip_intra_list_run_batch(Dav1dHipContext *c, const Dav1dHipIntraList *l, size_t batch, const Dav1dHipPicture *dst, void *coef,
                                   uint8_t *aux) {
    return dav1d_hip_intra_list_run_batch_blend(c, l, batch, dst, coef, aux, nullptr, nullptr);
}

// every batch of the list, in order, back to back on the context's stream (what a frame does; one call instead of one per step)
int dav1d_hip_intra_list_run_all(Dav1dHipContext *c, const Dav1dHipIntraList *l, const Dav1dHipPicture *dst, void *coef, uint8_t *aux) {
    if (!c || !l) return -EINVAL;
    int rc = 0;
    for (size_t k = 0; k + 1 < l->pair_start.size() && !rc; k++) rc = dav1d_hip_intra_list_run_batch_blend(c, l, k, dst, coef, aux, nullptr, nullptr);
    return rc;
}

// prep / mask: the scratch arena the PRED_TMP predictions of the batch go to and the blends read, and the mask arena
int dav1d_hip_intra_list_run_batch_blend(Dav1dHipContext *c, const Dav1dHipIntraList *l, size_t batch, const Dav1dHipPicture *dst, void *coef,
                                         uint8_t *aux, int16_t *prep, uint8_t *mask) {
    if (!c || !l || !dst || batch + 1 >= l->pair_start.size() || (l->needs_aux && !aux)) return -EINVAL;
    const size_t n_blend = l->blend_start[batch + 1] - l->blend_start[batch];
    if (n_blend && (!prep || !mask)) return -EINVAL;
    const DevPlanes dp = dev_planes(dst);
    int rc = 0;
    const size_t n_pairs = l->pair_start[batch + 1] - l->pair_start[batch];
    const Dav1dHipIpredList *pl = l->preds;
    const size_t n_rest = pl ? pl->start[batch + 1] - pl->start[batch] : 0;
    if (n_pairs && n_rest) {
        // the pairs and the other predictions of the step are independent: one launch, side by side (intra_pair.hip)
        if ((pl->needs_aux && !aux) || (pl->needs_tmp && !prep)) return -EINVAL;
        rc = dav1d_hip_launch_intra_step(&dp, dst->bpc, dst->layout, pl->dev + pl->start[batch], (int) n_rest, (int) pl->n_big[batch],
                                         l->p_dev + l->pair_start[batch], l->t_dev + l->pair_start[batch], (int) n_pairs, aux, prep, coef, c->stream);
    } else {
        if (n_pairs)
            rc = dav1d_hip_launch_intra_pairs(&dp, dst->bpc, dst->layout, l->p_dev + l->pair_start[batch], l->t_dev + l->pair_start[batch],
                                              (int) n_pairs, aux, coef, c->stream);
        if (!rc) rc = ipred_list_run_batch_tmp(c, l->preds, batch, dst, aux, prep);
    }
    if (!rc && n_blend) rc = dav1d_hip_launch_comp(&dp, dst->bpc, l->b_dev + l->blend_start[batch], (int) n_blend, prep, mask, c->stream);
    if (!rc && l->itx[batch]->n) rc = dav1d_hip_itx_list_run(c, l->itx[batch], dst, coef);
    return rc;
}

} // extern "C"

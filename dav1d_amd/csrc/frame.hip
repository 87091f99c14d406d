// Driver-level boundary: one frame in flight.  The reference's pass-2 workers call f->bd_fn.recon_b_* per block and
// filter_sbrow_* per superblock row (src/thread_task.c:733-851, src/recon_tmpl.c:1557-2135); the lister that replaces
// them appends the equivalent flat tasks here, tile-sbrow by tile-sbrow and from any worker thread, and
// dav1d_hip_frame_end() runs the whole frame in the reference's stage order:
//   inter prediction (+ fused compounds) -> residuals -> deblock (cols, rows) -> CDEF -> loop restoration -> film grain.
// Out-of-place stages land in pictures the frame owns; host code only: every kernel is reached through the batch API.
#include "capi.h"
#include <chrono>
#include <memory>
#include "cdef_rows.h"
#include "chunk.h"
extern "C" {
#include "../host/lister_priv.h"       // what the listers (host/*.c) and this file call of each other
}
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <new>
#include <vector>

struct Dav1dHipFrame {
    Dav1dHipContext *c;
    Dav1dHipPicture cur;
    Dav1dHipPicture refs[8];
    int n_refs;
    std::mutex mtx;
    std::vector<Dav1dHipChunk *> chunks;    // the tile-sbrows' inter predictions + residuals, preprocessed by their submitters (chunk.hip)
    uint8_t *arena;                         // device home of the chunks' blobs
    size_t arena_cap;
    // chunk preparations handed to the library's threads (option prep_async) and not finished yet; the first error among them
    std::atomic<int> prep_pending { 0 };
    std::mutex prep_mtx;
    std::condition_variable prep_cv;
    int prep_rc = 0;
    std::atomic<size_t> arena_used;
    // Pinned twin of the arena: a submitter copies its chunk's blob to the offset it drew and is done; the frame goes up as ONE
    // transfer at frame end (30 MB for an 8K frame: half a millisecond).  One hipMemcpyAsync per chunk — 1,800 calls per 8K frame,
    // serialised inside the runtime at about 4 us each — was what the listing threads queued on (c->chunk_upload: 1 = that way).
    uint8_t *harena;
    size_t harena_cap;
    size_t harena_flushed;                  // bytes of the twin already on their way to the device (dav1d_hip_frame_flush)
    // The frame's own coefficient arena (dav1d_hip_frame_submit_coefs): the eob + 1 values per transform block a packing lister
    // gathered, drawn segment by segment by the submitting threads.  The device side is sized for the frame's dense bound (no
    // packed frame can need more: a block's values fit its slab), the pinned twin for what frames have needed so far; a segment
    // that does not fit the twin is kept (late) and sent on its own at frame end.
    uint8_t *carena;
    size_t carena_cap;
    uint8_t *hcarena;
    size_t hcarena_cap, hcarena_flushed;
    std::atomic<size_t> carena_used;        // bytes
    std::once_flag carena_once;
    struct LateCoefs { size_t off, bytes; void *copy; };
    std::vector<LateCoefs> late_coefs;
    std::atomic<int> saw_dense, saw_packed; // kinds of residual tasks submitted so far
    // intra blocks: step k of the wavefront = the blocks whose neighbours are final after steps 0 .. k - 1 (and after the
    // inter blocks of the frame); predictions and residuals per step
    // One entry per submission (a tile-sbrow's blocks), tasks sorted by step with the end offset of every step; built — and
    // for the dataflow launch paired into units — on the submitting thread, merged step by step at frame end.
    struct StepChunk {
        std::vector<Dav1dHipIpredTask> ip;
        std::vector<Dav1dHipItxTask> ix;
        std::vector<Dav1dHipCompTask> bl;           // inter-intra blends (between a step's predictions and its residuals)
        std::vector<uint32_t> ip_end, ix_end, bl_end;       // [step] -> end offset of the step in ip / ix / bl
        std::vector<IntraUnit> units;               // per step: the units with a prediction, then the residuals on their own
        std::vector<uint32_t> ua_end, ub_end;       // [step] -> end of the step's first / second kind in units
        bool flow_ok;                               // every task is of a kind the dataflow launch runs
        std::vector<SbPart> parts;                  // sb_sorted: units are sorted by (superblock, step, kind), one part per superblock
        std::vector<uint64_t> copy_deps;            // intra block copies: (superblock << 32 | superblock under its source window)
        bool sb_sorted;
        IntraUnit *sorted;                          // ... and sit here: in the frame's pinned unit arena (in_arena) or in `units`
        size_t n_sorted;
        bool in_arena, has_pal, has_blend;
    };
    // The sorted units of the superblock route go straight into pinned memory, drawn chunk by chunk by the submitting threads (sized
    // by what frames have needed so far; a chunk that does not fit keeps its units and is copied at frame end): the upload is one
    // transfer from there.
    uint8_t *huarena = nullptr;
    size_t huarena_cap = 0;
    std::atomic<size_t> uarena_used { 0 };
    SbTiling tiling;                                // dav1d_hip_frame_set_tiling: what the superblock wavefront (intra_sb.hip) needs
    bool have_tiling;
    std::vector<uint8_t> sb_dep;                    // [superblock] -> neighbours its blocks read (dav1d_hip_frame_set_sb_deps; 15 until told)
    std::vector<StepChunk *> step_chunks;
    size_t n_steps;                                 // 1 + highest step submitted
    std::vector<Dav1dHipMcTask> step_copy;                      // intra block copies: predictions from the frame's own pixels ...
    std::vector<uint16_t> step_copy_step;                       // ... and the wavefront step each of them runs in
    std::vector<Dav1dHipWarpTask> warp;                         // warped predictions (step 0: they read reference pictures only)
    std::vector<Dav1dHipMcScaledTask> scaled;                   // predictions from references of another size
    uint8_t *aux;                    // DEVICE arena of the palette indices the intra tasks point into (may be NULL)
    std::vector<Dav1dHipLfTask> lf;
    std::vector<Dav1dHipCdefTask> cdef;
    std::vector<Dav1dHipLrTask> lr;
    // filter tasks as their submitters handed them over (malloc'ed arrays, owned): strung together at frame end.  Appending to the
    // three vectors above under the frame's lock — 11 MB per 8K frame, reallocations included — made the filter listing threads
    // queue (5 ms on 64 threads).
    // What can be done per piece is done by the thread that submits it: the checks of the batch calls, the deblocking tasks
    // partitioned into vertical edges first, the CDEF units grouped for the strip kernel (group.first relative to the piece).
    struct FilterPiece {
        Dav1dHipLfTask *lf; size_t n_lf, n_lf0;              // [0, n_lf0): dir 0 (edges between columns), the rest dir 1
        Dav1dHipCdefTask *cdef; size_t n_cdef, n_raw;
        std::vector<CdefGroup> *groups;
        Dav1dHipLrTask *lr; size_t n_lr;
    };
    std::vector<FilterPiece> filter_pieces;
    // Pinned slabs and device task buffers of launches that are enqueued but not waited for: the post filters of a frame are one
    // stream of uploads and launches with ONE wait at the end (frame_run), and what they read has to stay until then
    struct Deferred {
        std::vector<std::pair<uint8_t *, size_t>> slabs;
        std::vector<std::unique_ptr<TaskBuf>> bufs;
        void release(Dav1dHipContext *c) { for (auto &sl : slabs) dav1d_hip_slab_put(c, sl.first, sl.second); slabs.clear(); bufs.clear(); }
    } deferred;
    // the host half of the restoration stage (frame_lr_plan): made while the earlier stages' launches run
    struct LrPlan {
        bool valid = false, banded = false;
        int nb = 0, max_w = 0;
        size_t nw = 0, n_waves = 0;
        std::vector<Dav1dHipLrTask> sorted;
        std::vector<uint32_t> tail, target;
        std::vector<int> first_y, band;
        std::vector<size_t> off, pos;
        uint8_t *slab = nullptr;       // pinned copy of what the launches read
        size_t slab_cap = 0, bytes = 0, o_waves = 0;
    } lrplan;
    // CDEF as one record per unit row of a 64-pixel column (cdef_rows.h), written by the filter lister's threads straight into this
    // pinned table and cut into unit records on the device (cdef.hip cdef_expand_kernel); state: 0 not asked yet, 1 in use, -1 refused
    Dav1dHipCdefRow *cdef_rows = nullptr;
    size_t cdef_rows_cap = 0;
    int cdef_rows_state = 0, cdef_w64 = 0, cdef_h8 = 0;
    bool cdef_rows_dirty = false;
    std::atomic<size_t> cdef_row_units{0};
    const uint8_t *lvl;
    ptrdiff_t b4_stride;
    uint8_t lut_e[64], lut_i[64];
    bool have_lut;
    int cdef_damping;
    bool have_grain;
    Dav1dHipFilmGrainData grain;
    Dav1dHipGrain *prepared;         // templates + scaling tables, generated on a side stream from set_filters on
    int is_id;
    Dav1dHipPicture tmp[2];          // CDEF output, restoration output (allocated on first use)
    bool have_tmp[2];
    int sr_w;                        // super-resolution: width after the upscale between CDEF and restoration (0: none)
    Dav1dHipPicture sr[3];           // upscaled: CDEF output, deblocked rows (restoration's stripe borders), restoration output
    bool have_sr[3];
    int post_bands;                  // bands the post filters of the last dav1d_hip_frame_end ran in (0: stage by stage)
    std::thread worker;              // dav1d_hip_frame_end_async
    std::atomic<int> progress_rows;
    void (*progress_cb)(void *cookie, int rows, const Dav1dHipPicture *pic) = nullptr;
    void *progress_cookie = nullptr;
    // rows of the filtered picture that are final: stored for dav1d_hip_frame_progress, handed to the callback
    void publish(int rows, const Dav1dHipPicture *pic) {
        if (rows <= progress_rows.load()) return;
        progress_rows.store(rows);
        if (progress_cb) progress_cb(progress_cookie, rows, pic);
    }
    int async_rc;
    Dav1dHipPicture async_filtered;
};

static int frame_tmp(Dav1dHipFrame *f, int i) {
    if (f->have_tmp[i]) return 0;
    const int rc = dav1d_hip_picture_take(f->c, &f->tmp[i], f->cur.p[0].w, f->cur.p[0].h, f->cur.layout, f->cur.bpc);
    if (!rc) f->have_tmp[i] = true;
    return rc;
}

// ---------------------------------------------------------------- post filters, pipelined over bands of superblock rows
//
// deblock -> CDEF -> restoration of a whole frame one after the other leaves the arithmetic-bound CDEF kernel and the
// memory-bound deblocking / restoration kernels waiting for each other.  The reference overlaps them by superblock row
// (filter_sbrow_* tasks, src/thread_task.c:783-851: CDEF of row n runs once deblocking of row n + 1 is done, restoration one
// row behind that); the same here with bands of a few superblock rows: deblocking of band b on the context's stream, CDEF of
// band b on a side stream once deblocking of band b + 1 is through (its horizontal edges still change the last rows of band
// b), restoration of band b on a second side stream once CDEF of band b + 1 is through (it reads 3 rows below the band).
// Out of place as before: CDEF writes tmp[0] from cur, restoration writes tmp[1].  Returns 1 when the frame does not qualify
// (too small, a deblocking column task crossing a band, units not covering the frame): the caller then runs stage by stage.
static int post_filters_pipelined(Dav1dHipFrame *f, const Dav1dHipPicture **last_out) {
    Dav1dHipContext *c = f->c;
    const bool has_lf = !f->lf.empty(), has_cdef = !f->cdef.empty(), has_lr = !f->lr.empty();
    if (!c->concurrent || (int) has_lf + has_cdef + has_lr < 2) return 1;
    if (f->sr_w) return 1;                 // super-resolution: the upscale sits between CDEF and restoration, stage by stage
    // DAV1D_HIP_POST_BANDS = bands per frame.  Off unless asked for: measured on MI355X (8K 10-bit frame: deblock 0.15 + CDEF
    // 0.58 + restoration 0.26 = 0.99 ms stage by stage) the banded pipeline takes 1.14 ms with 3 bands, 1.21 with 6, 1.78 with
    // 17 — every cross-stream event costs a release / acquire of the caches, about 45 us per band, more than the overlap of
    // the arithmetic-bound CDEF with its memory-bound neighbours returns.  (The recon list, 5 events per frame, does gain.)
    const int want = c->post_bands;
    if (want < 2) return 1;
    const int H = f->cur.p[0].h, W = f->cur.p[0].w, layout = f->cur.layout, bps = f->cur.bpc > 8 ? 2 : 1;
    const int ss_ver = layout == DAV1D_HIP_LAYOUT_I420, ss_hor = layout != DAV1D_HIP_LAYOUT_I444;
    // band height: whole pairs of 128-row superblock rows (a chroma column task of 32 units covers 256 luma rows in 4:2:0)
    const int sbpairs = (H + 255) / 256, band_h = 256 * ((sbpairs + want - 1) / want), nb = (H + band_h - 1) / band_h;
    if (nb < 3) return 1;
    auto band_of = [&](int luma_y) { const int b = luma_y / band_h; return b < 0 ? 0 : b >= nb ? nb - 1 : b; };
    // ---- band of every task; anything irregular sends the frame down the stage-by-stage path
    std::vector<int> lf_b(f->lf.size()), cdef_b(f->cdef.size()), lr_b(f->lr.size());
    for (size_t i = 0; i < f->lf.size(); i++) {
        const Dav1dHipLfTask &t = f->lf[i];
        if (t.plane > 2 || t.dir > 1 || t.lvl_comp > 3) return -EINVAL;
        const int sv = t.plane ? ss_ver : 0, stride = (int) (f->cur.p[t.plane].stride / bps);
        if (stride <= 0) return -EINVAL;
        const int y = (int) (t.dst_off / (uint32_t) stride) << sv;
        lf_b[i] = band_of(y);
        if (t.dir == 0) {        // a column of units running down: must end inside its band
            const uint32_t m = t.vmask[0] | t.vmask[1] | t.vmask[2];
            const int units = m ? 32 - __builtin_clz(m) : 0;
            if (units && band_of(y + ((4 * units) << sv) - 1) != lf_b[i]) return 1;
        }
    }
    if (has_cdef && f->cdef.size() != (size_t) ((W + 7) / 8) * (size_t) ((H + 7) / 8)) return 1;   // unlisted units would need the copy
    for (size_t i = 0; i < f->cdef.size(); i++) {
        const Dav1dHipCdefTask &t = f->cdef[i];
        if (t.edges > 15 || t.plane > 2 || t.dir > 7) return -EINVAL;
        if (t.flags & 1) return 1;
        cdef_b[i] = band_of(t.by * 8);
    }
    long long area[3] = { 0, 0, 0 };
    for (size_t i = 0; i < f->lr.size(); i++) {
        const Dav1dHipLrTask &t = f->lr[i];
        if (t.plane > 2 || t.edges > 15 || !t.w || t.w > 384 || !t.h || t.h > 64 || t.type > DAV1D_HIP_LR_SGR_MIX) return -EINVAL;
        lr_b[i] = band_of((int) t.y << (t.plane ? ss_ver : 0));
        area[t.plane] += (long long) t.w * t.h;
    }
    if (has_lr)
        for (int p = 0; p < (layout == DAV1D_HIP_LAYOUT_I400 ? 1 : 3); p++)
            if (area[p] != (long long) f->cur.p[p].w * f->cur.p[p].h) return 1;
    (void) ss_hor;
    // ---- ordered copies: lf (band, columns before rows), cdef (band), lr (band, Wiener before self-guided)
    std::vector<Dav1dHipLfTask> lf_s(f->lf.size());
    std::vector<Dav1dHipCdefTask> cdef_s(f->cdef.size());
    std::vector<Dav1dHipLrTask> lr_s(f->lr.size());
    std::vector<size_t> lf_off(2 * nb + 1, 0), cdef_off(nb + 1, 0), lr_off(2 * nb + 1, 0);
    for (size_t i = 0; i < f->lf.size(); i++) lf_off[2 * lf_b[i] + f->lf[i].dir + 1]++;
    for (size_t i = 0; i < f->cdef.size(); i++) cdef_off[cdef_b[i] + 1]++;
    for (size_t i = 0; i < f->lr.size(); i++) lr_off[2 * lr_b[i] + (f->lr[i].type > DAV1D_HIP_LR_WIENER5) + 1]++;
    for (int k = 0; k < 2 * nb; k++) { lf_off[k + 1] += lf_off[k]; lr_off[k + 1] += lr_off[k]; }
    for (int k = 0; k < nb; k++) cdef_off[k + 1] += cdef_off[k];
    {
        std::vector<size_t> pos(lf_off.begin(), lf_off.end() - 1);
        for (size_t i = 0; i < f->lf.size(); i++) lf_s[pos[2 * lf_b[i] + f->lf[i].dir]++] = f->lf[i];
        pos.assign(cdef_off.begin(), cdef_off.end() - 1);
        for (size_t i = 0; i < f->cdef.size(); i++) cdef_s[pos[cdef_b[i]]++] = f->cdef[i];
        pos.assign(lr_off.begin(), lr_off.end() - 1);
        for (size_t i = 0; i < f->lr.size(); i++) lr_s[pos[2 * lr_b[i] + (f->lr[i].type > DAV1D_HIP_LR_WIENER5)]++] = f->lr[i];
    }
    if (has_lf && (!f->lvl || !f->have_lut)) return -EINVAL;
    int rc = 0;
    if (has_cdef) rc = frame_tmp(f, 0);
    if (!rc && has_lr) rc = frame_tmp(f, 1);
    if (rc) return rc;
    // CDEF: units that sit side by side share a wave (cdef.hip, strip kernel); the groups of every band, band after band
    std::vector<CdefGroup> cgroups;
    std::vector<size_t> cg_off(nb + 1, 0), raw_in_band(nb, 0);
    const DevPlanes cur_p = dev_planes(&f->cur);
    const bool strips = has_cdef && !c->cdef_unit_kernel && dav1d_hip_cdef_strip_ok(&cur_p, &cur_p, f->cur.bpc);
    if (strips)
        for (int k = 0; k < nb; k++) {
            // (tasks with the RAW flag — DSP-level callers — are not grouped: the band runs them through the unit kernel as well)
            raw_in_band[k] = dav1d_hip_cdef_make_groups(cdef_s.data() + cdef_off[k], cdef_off[k + 1] - cdef_off[k], cdef_off[k], cgroups);
            cg_off[k + 1] = cgroups.size();
        }
    // self-guided restoration: the units of a row share waves (lr.hip); rows and waves per band
    std::vector<uint32_t> sgr_waves;
    std::vector<size_t> sw_off(nb + 1, 0);
    for (int k = 0; k < nb; k++) {
        dav1d_hip_sgr_make_rows(lr_s.data() + lr_off[2 * k + 1], lr_off[2 * k + 2] - lr_off[2 * k + 1], sgr_waves);
        sw_off[k + 1] = sgr_waves.size() / 4;
    }
    const size_t bytes_lf = lf_s.size() * sizeof(Dav1dHipLfTask), bytes_cdef = cdef_s.size() * sizeof(Dav1dHipCdefTask),
                 bytes_lr = lr_s.size() * sizeof(Dav1dHipLrTask);
    const size_t o_cdef = (bytes_lf + 255) & ~(size_t) 255, o_lr = (o_cdef + bytes_cdef + 255) & ~(size_t) 255;
    const size_t o_cg = (o_lr + bytes_lr + 255) & ~(size_t) 255, bytes_cg = cgroups.size() * sizeof(CdefGroup);
    const size_t o_sw = (o_cg + bytes_cg + 255) & ~(size_t) 255;
    TaskBuf dev_buf(c, o_sw + sgr_waves.size() * 4 + 256);
    uint8_t *const dev = dev_buf.p;
    if (!dev) return -ENOMEM;
    if (bytes_lf) rc = dav1d_hip_upload(c, dev, lf_s.data(), bytes_lf);
    if (!rc && bytes_cdef) rc = dav1d_hip_upload(c, dev + o_cdef, cdef_s.data(), bytes_cdef);
    if (!rc && bytes_lr) rc = dav1d_hip_upload(c, dev + o_lr, lr_s.data(), bytes_lr);
    if (!rc && bytes_cg) rc = dav1d_hip_upload(c, dev + o_cg, cgroups.data(), bytes_cg);
    if (!rc && !sgr_waves.empty()) rc = dav1d_hip_upload(c, dev + o_sw, sgr_waves.data(), sgr_waves.size() * 4);
    const CdefGroup *d_cg = reinterpret_cast<const CdefGroup *>(dev + o_cg);
    const Dav1dHipLfTask *d_lf = reinterpret_cast<const Dav1dHipLfTask *>(dev);
    const Dav1dHipCdefTask *d_cdef = reinterpret_cast<const Dav1dHipCdefTask *>(dev + o_cdef);
    const Dav1dHipLrTask *d_lr = reinterpret_cast<const Dav1dHipLrTask *>(dev + o_lr);
    std::vector<hipEvent_t> ev(3 * nb, nullptr);       // [0, nb): deblocked, [nb, 2nb): CDEF done, [2nb, 3nb): the band's last stage done
    for (int k = 0; k < 3 * nb && !rc; k++) rc = hip_rc(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    // luma rows that are final once the last stage of band b is through: restoration stripes start 8 rows above the 64-row grid, so
    // a band's last stripe reaches into the next band's rows and the first rows of a band belong to the band before it
    std::vector<int> final_rows(nb, H);
    for (int b = nb - 2; b >= 0; b--) final_rows[b] = std::min(final_rows[b + 1], (b + 1) * band_h);
    if (has_lr) {
        std::vector<int> first_y(nb + 1, H);
        for (size_t i = 0; i < f->lr.size(); i++)
            first_y[lr_b[i]] = std::min(first_y[lr_b[i]], (int) f->lr[i].y << (f->lr[i].plane ? ss_ver : 0));
        // (when the NEXT band lists no unit, first_y comes from a later band: the rows are capped by what CDEF has finished when this
        // band's restoration ends — it waited for CDEF of the next band, whose last 8 rows the band after that may still change)
        for (int b = nb - 1; b >= 0; b--) {
            first_y[b] = std::min(first_y[b], first_y[b + 1]);
            final_rows[b] = b + 1 == nb ? H : std::min(first_y[b + 1], std::min(H, (b + 2) * band_h) - 8);
        }
    }
    const Dav1dHipPicture *cdef_out = has_cdef ? &f->tmp[0] : &f->cur;
    if (!rc) {
        const DevPlanes cur = dev_planes(&f->cur), t0 = dev_planes(&f->tmp[0]), t1 = dev_planes(&f->tmp[1]), co = dev_planes(cdef_out);
        hipStream_t sa = c->stream, sb = c->side[0], sc = c->side[1];
        (void) hipEventRecord(c->ev_t0, sa);          // dav1d_hip_last_kernel_ms(): device time of the pipelined section
        (void) hipEventRecord(c->ev_fork, sa);
        (void) hipStreamWaitEvent(sb, c->ev_fork, 0);
        (void) hipStreamWaitEvent(sc, c->ev_fork, 0);
        // the three stages are issued band by band, interleaved, so that no stream ever waits for an event that is recorded
        // later in issue order than its own next launch needs
        for (int b = 0; b < nb + 2 && !rc; b++) {
            if (has_lf && b < nb) {
                const size_t v0 = lf_off[2 * b], v1 = lf_off[2 * b + 1], v2 = lf_off[2 * b + 2];
                if (v1 > v0) rc = dav1d_hip_launch_lf(&cur, f->cur.bpc, 0, d_lf + v0, (int) (v1 - v0), f->lvl, (int) f->b4_stride, f->lut_e, f->lut_i, sa);
                if (!rc && v2 > v1) rc = dav1d_hip_launch_lf(&cur, f->cur.bpc, 1, d_lf + v1, (int) (v2 - v1), f->lvl, (int) f->b4_stride, f->lut_e, f->lut_i, sa);
                (void) hipEventRecord(ev[b], sa);
            }
            const int bc = b - 1;                    // CDEF runs one band behind deblocking
            if (has_cdef && bc >= 0 && bc < nb && !rc) {
                if (has_lf) (void) hipStreamWaitEvent(sb, ev[bc + 1 < nb ? bc + 1 : nb - 1], 0);
                const size_t n = cdef_off[bc + 1] - cdef_off[bc];
                // a listed unit whose strengths come out as zero (adjusted primary 0 and no secondary) is not written by the kernel:
                // the band's rows of the deblocked picture go to the output first
                for (int pl = 0; pl < 3 && !rc; pl++) {
                    if (!f->cur.p[pl].data) continue;
                    const int sv = pl ? ss_ver : 0;
                    const int r0 = (bc * band_h) >> sv, r1 = std::min(((bc + 1) * band_h) >> sv, f->cur.p[pl].h);
                    if (r1 > r0)
                        rc = hip_rc(hipMemcpy2DAsync((uint8_t *) f->tmp[0].p[pl].data + (size_t) r0 * f->tmp[0].p[pl].stride, f->tmp[0].p[pl].stride,
                                                     (const uint8_t *) f->cur.p[pl].data + (size_t) r0 * f->cur.p[pl].stride, f->cur.p[pl].stride,
                                                     (size_t) f->cur.p[pl].w * bps, r1 - r0, hipMemcpyDeviceToDevice, sb));
                }
                if (rc) break;
                if (n && strips) {
                    rc = dav1d_hip_launch_cdef_groups(&t0, &cur, f->cur.bpc, layout, d_cdef, d_cg + cg_off[bc], (int) (cg_off[bc + 1] - cg_off[bc]),
                                                      f->cdef_damping, nullptr, sb);
                    if (!rc && raw_in_band[bc])
                        rc = dav1d_hip_launch_cdef(&t0, &cur, f->cur.bpc, layout, d_cdef + cdef_off[bc], (int) n, f->cdef_damping, nullptr, 1, sb);
                } else if (n) rc = dav1d_hip_launch_cdef(&t0, &cur, f->cur.bpc, layout, d_cdef + cdef_off[bc], (int) n, f->cdef_damping, nullptr, 0, sb);
                (void) hipEventRecord(ev[nb + bc], sb);
                if (!has_lr) (void) hipEventRecord(ev[2 * nb + bc], sb);
            }
            const int br = b - 2;                    // restoration one band behind CDEF
            if (has_lr && br >= 0 && br < nb && !rc) {
                const int dep = br + 1 < nb ? br + 1 : nb - 1;
                if (has_cdef) (void) hipStreamWaitEvent(sc, ev[nb + dep], 0);
                else if (has_lf) (void) hipStreamWaitEvent(sc, ev[dep], 0);
                const size_t w0 = lr_off[2 * br], w1 = lr_off[2 * br + 1], w2 = lr_off[2 * br + 2];
                if (w1 > w0) rc = dav1d_hip_launch_wiener(&t1, &co, &cur, f->cur.bpc, d_lr + w0, (int) (w1 - w0), 384, sc);
                if (!rc && w2 > w1) rc = dav1d_hip_launch_sgr(&t1, &co, &cur, f->cur.bpc, d_lr + w1, dev + o_sw + 16 * sw_off[br], (int) (sw_off[br + 1] - sw_off[br]), sc);
                (void) hipEventRecord(ev[2 * nb + br], sc);
            }
        }
        (void) hipEventRecord(c->ev_join[0], sb);
        (void) hipEventRecord(c->ev_join[1], sc);
        (void) hipStreamWaitEvent(sa, c->ev_join[0], 0);
        (void) hipStreamWaitEvent(sa, c->ev_join[1], 0);
        (void) hipEventRecord(c->ev_t1, sa);
    }
    // the bands come through in order: every band's rows are published as soon as its last stage is (reference
    // src/thread_task.c:888-896, once per superblock row there)
    const Dav1dHipPicture *const out_pic = has_lr ? &f->tmp[1] : cdef_out;
    for (int b = 0; b < nb && !rc; b++) {
        if (hipEventSynchronize(ev[2 * nb + b]) != hipSuccess) break;
        if (b + 1 < nb) f->publish(final_rows[b], out_pic);         // the last band is published with the frame (frame_run)
    }
    (void) hipStreamSynchronize(c->stream);
    if (!rc) { c->last_ms = 0.f; (void) hipEventElapsedTime(&c->last_ms, c->ev_t0, c->ev_t1); }
    for (hipEvent_t e : ev) if (e) (void) hipEventDestroy(e);
    *last_out = has_lr ? &f->tmp[1] : cdef_out;
    if (!rc) f->post_bands = nb;
    return rc;
}

extern "C" {

int dav1d_hip_frame_begin(Dav1dHipContext *c, Dav1dHipFrame **out, const Dav1dHipPicture *cur, const Dav1dHipPicture *refs, int n_refs) {
    if (!c || !out || !cur || n_refs < 0 || n_refs > 8 || (n_refs && !refs)) return -EINVAL;
    *out = nullptr;
    for (int i = 0; i < n_refs; i++) if (refs[i].bpc != cur->bpc || refs[i].layout != cur->layout) return -EINVAL;
    // several devices in the process: the frame's own picture is on the context's device (its references are looked at when the frame ends:
    // a caller may only know their geometry yet)
    if (pictures_on_device(c, cur, 1)) return -EXDEV;
    (void) hipSetDevice(c->device);
    Dav1dHipFrame *f = new (std::nothrow) Dav1dHipFrame();
    if (!f) return -ENOMEM;
    __atomic_fetch_add(&dav1d_hip_live[1], 1, __ATOMIC_RELAXED);       // (every way out from here goes through dav1d_hip_frame_destroy or hands the frame over)
    f->c = c;
    f->cur = *cur;
    for (int i = 0; i < n_refs; i++) f->refs[i] = refs[i];
    f->n_refs = n_refs;
    f->lvl = nullptr;
    memset(f->lut_e, 0, 64); memset(f->lut_i, 0, 64);
    f->have_lut = false;
    f->b4_stride = 0;
    f->cdef_damping = 0;
    f->aux = nullptr;
    f->n_steps = 0;
    f->have_tiling = false;
    f->have_grain = false;
    f->prepared = nullptr;
    f->is_id = 0;
    f->have_tmp[0] = f->have_tmp[1] = false;
    f->sr_w = 0;
    f->have_sr[0] = f->have_sr[1] = f->have_sr[2] = false;
    f->post_bands = 0;
    f->progress_rows.store(0);
    f->async_rc = 0;
    memset(&f->async_filtered, 0, sizeof(f->async_filtered));
    // a chunk arena sized for the largest frame this context has seen (grown at frame end when it turns out too small)
    f->arena = nullptr; f->arena_cap = 0; f->arena_used = 0;
    {
        std::lock_guard<std::mutex> lk(c->pool_mtx);
        int best = -1;              // best fit: the coefficient arenas of packed frames live in the same pool and are much larger
        for (size_t i = 0; i < c->free_arenas.size(); i++)
            if (c->free_arenas[i].cap >= c->arena_hint && (best < 0 || c->free_arenas[i].cap < c->free_arenas[best].cap)) best = (int) i;
        if (best >= 0) {
            f->arena = c->free_arenas[best].dev; f->arena_cap = c->free_arenas[best].cap;
            c->free_arenas[best] = c->free_arenas.back();
            c->free_arenas.pop_back();
        }
    }
    if (!f->arena) {
        size_t want = c->arena_min;
        while (want < c->arena_hint + (c->arena_hint >> 2)) want <<= 1;
        if (hipMalloc((void **) &f->arena, want) == hipSuccess) f->arena_cap = want; else f->arena = nullptr;
    }
    f->carena = f->hcarena = nullptr; f->carena_cap = f->hcarena_cap = f->hcarena_flushed = 0; f->carena_used = 0;
    f->saw_dense = 0; f->saw_packed = 0;
    f->harena = nullptr; f->harena_cap = 0; f->harena_flushed = 0;
    if (f->arena && !c->chunk_upload) f->harena = dav1d_hip_slab_get(c, f->arena_cap, &f->harena_cap);
    *out = f;
    return 0;
}

// The references again, before the frame is ended: dav1d's frame threading lists frame n + 1 while frame n is still being filtered
// (src/thread_task.c:393-439 lets a tile task start once the rows it reads are final), and which picture holds frame n's final
// pixels — `cur` itself or a picture the frame owns (*filtered of dav1d_hip_frame_end) — is only known when frame n ends.
// Geometry, bit depth and layout must be those given to dav1d_hip_frame_begin: the lists were prepared for them.
int dav1d_hip_frame_set_refs(Dav1dHipFrame *f, const Dav1dHipPicture *refs, int n_refs) {
    if (!f || !refs || n_refs != f->n_refs) return -EINVAL;
    if (f->worker.joinable()) return -EBUSY;
    std::lock_guard<std::mutex> lk(f->mtx);
    for (int i = 0; i < n_refs; i++) {
        if (refs[i].bpc != f->refs[i].bpc || refs[i].layout != f->refs[i].layout) return -EINVAL;
        for (int p = 0; p < 3; p++)
            if (refs[i].p[p].w != f->refs[i].p[p].w || refs[i].p[p].h != f->refs[i].p[p].h || refs[i].p[p].stride != f->refs[i].p[p].stride ||
                !refs[i].p[p].data != !f->refs[i].p[p].data) return -EINVAL;
    }
    for (int i = 0; i < n_refs; i++) f->refs[i] = refs[i];
    return 0;
}

static void note_kinds(Dav1dHipFrame *f, const Dav1dHipItxTask *itx, size_t n) {
    unsigned packed = 0, dense = 0;
    for (size_t i = 0; i < n; i++) { const unsigned p = itx[i].flags & DAV1D_HIP_ITX_PACKED; packed |= p; dense |= p ^ 1u; }
    if (packed) f->saw_packed.store(1, std::memory_order_relaxed);
    if (dense) f->saw_dense.store(1, std::memory_order_relaxed);
}

// bytes of a dense coefficient arena of this picture: no packed frame holds more values than that
static size_t dense_coef_bytes(const Dav1dHipPicture *p) {
    size_t n = 0;
    for (int pl = 0; pl < 3; pl++)
        if (p->p[pl].data) n += (size_t) ((p->p[pl].w + 63) & ~63) * (size_t) ((p->p[pl].h + 63) & ~63);
    return n * (p->bpc > 8 ? 4 : 2);
}

// The values of the caller's PACKED residual tasks (a tile-sbrow's worth): copied into the frame's coefficient arena; *base = what
// to add to the tasks' cf_off.  The arena travels with the frame (dav1d_hip_frame_flush / dav1d_hip_frame_end with coef = NULL).
static int frame_coef_room(Dav1dHipFrame *f, size_t n, uint32_t *base, void **dst);
int dav1d_hip_frame_submit_coefs(Dav1dHipFrame *f, const void *vals, size_t n, uint32_t *base) {
    if (!f || !base || (!vals && n)) return -EINVAL;
    void *dst = nullptr;
    const int rc = frame_coef_room(f, n, base, &dst);
    if (!rc && n) memcpy(dst, vals, n * (f->cur.bpc > 8 ? 4 : 2));
    return rc;
}
// internal (host/lister.c): the room only; the lister's rows are packed straight into it (dav1d_hip_frame_submit_tile_sbrow_packing)
int dav1d_hip_frame_reserve_coefs(Dav1dHipFrame *f, size_t n, uint32_t *base, void **dst) {
    if (!f || !base || !dst) return -EINVAL;
    return frame_coef_room(f, n, base, dst);
}
// n values of room in the frame's packed-coefficient arena: in its pinned twin, or (the twin is sized by what frames have needed) in a buffer of
// its own that goes up at frame end
static int frame_coef_room(Dav1dHipFrame *f, size_t n, uint32_t *base, void **dst) {
    Dav1dHipContext *c = f->c;
    const size_t csz = f->cur.bpc > 8 ? 4 : 2;
    std::call_once(f->carena_once, [&]() {
        const size_t want = dense_coef_bytes(&f->cur) + 4096;
        {
            std::lock_guard<std::mutex> lk(c->pool_mtx);
            int best = -1;
            for (size_t i = 0; i < c->free_arenas.size(); i++)
                if (c->free_arenas[i].cap >= want && (best < 0 || c->free_arenas[i].cap < c->free_arenas[best].cap)) best = (int) i;
            if (best >= 0) {
                f->carena = c->free_arenas[best].dev; f->carena_cap = c->free_arenas[best].cap;
                c->free_arenas[best] = c->free_arenas.back();
                c->free_arenas.pop_back();
            }
        }
        if (!f->carena && hipMalloc((void **) &f->carena, want) == hipSuccess) f->carena_cap = want;
        if (f->carena) {
            const size_t twin = std::min(f->carena_cap, std::max(c->carena_hint + (c->carena_hint >> 2), want / 4));
            f->hcarena = dav1d_hip_slab_get(c, twin, &f->hcarena_cap);
        }
    });
    if (!f->carena) return -ENOMEM;
    const size_t bytes = (n * csz + 63) & ~(size_t) 63, off = f->carena_used.fetch_add(bytes);
    if (off + bytes > f->carena_cap || (off + bytes) / csz > 0xffffffffu) return -EINVAL;      // more values than the frame has coefficients
    *base = (uint32_t) (off / csz);
    *dst = nullptr;
    if (!n) return 0;
    if (f->hcarena && off + bytes <= f->hcarena_cap) { *dst = f->hcarena + off; return 0; }
    void *copy = malloc(n * csz);
    if (!copy) return -ENOMEM;
    *dst = copy;
    std::lock_guard<std::mutex> lk(f->mtx);
    f->late_coefs.push_back({ off, n * csz, copy });
    return 0;
}

size_t dav1d_hip_frame_coef_bytes(const Dav1dHipFrame *f) { return f ? f->carena_used.load() : 0; }

// Reconstruction tasks of one tile-sbrow (what decode_b()'s pass-2 branch would have executed, src/decode.c:706-806).
// Thread-safe; the order between tile-sbrows is free: inter tasks of a frame write disjoint pixels, and every residual
// is added after every prediction.
static int submit_tile_sbrow(Dav1dHipFrame *f, const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                            const Dav1dHipItxTask *itx, size_t n_itx, bool trusted, const uint16_t *itx_dep = nullptr,
                            const Dav1dHipPackRec *recs = nullptr, size_t n_recs = 0, void *cf = nullptr, void *dst = nullptr);
int dav1d_hip_frame_submit_tile_sbrow(Dav1dHipFrame *f, const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                                      const Dav1dHipItxTask *itx, size_t n_itx) {
    return submit_tile_sbrow(f, mc, n_mc, comp, n_comp, itx, n_itx, false);
}
// internal (host/lister.c): the same for records the library made itself — they are not validated a second time
int dav1d_hip_frame_submit_tile_sbrow_own(Dav1dHipFrame *f, const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                                          const Dav1dHipItxTask *itx, size_t n_itx, const uint16_t *itx_dep) {
    return submit_tile_sbrow(f, mc, n_mc, comp, n_comp, itx, n_itx, true, itx_dep);
}
int dav1d_hip_frame_submit_tile_sbrow_packing(Dav1dHipFrame *f, const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                                              const Dav1dHipItxTask *itx, size_t n_itx, const uint16_t *itx_dep,
                                              const Dav1dHipPackRec *recs, size_t n_recs, void *cf, void *dst) {
    return submit_tile_sbrow(f, mc, n_mc, comp, n_comp, itx, n_itx, true, itx_dep, recs, n_recs, cf, dst);
}
// ---- the library's preparation threads (option prep_async).  A frame of few tiles has few listing threads — dav1d lists the rows of a tile one
// after the other (the cursors into cbi / cf are only known behind the row before: src/decode.c:2594-2635) — and each of them used to walk a row,
// then prepare its chunk, then walk the next: with the preparation (45 % of the two) on a thread of its own, the walk of row k + 1 runs next to
// the preparation of row k.  Jobs are self-contained (they own copies of the row's records); a frame waits for its jobs before it flushes, ends or dies.
namespace {
struct PrepPool {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    int threads = 0;
    void start_locked() {
        static const int want = getenv("DAV1D_HIP_PREP_THREADS") ? atoi(getenv("DAV1D_HIP_PREP_THREADS")) : 4;
        while (threads < (want < 1 ? 1 : want > 32 ? 32 : want)) {
            std::thread([this]() {
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv.wait(lk, [this]() { return !q.empty(); });
                        job = std::move(q.front());
                        q.pop_front();
                    }
                    job();
                }
            }).detach();
            threads++;
        }
    }
    // as many jobs waiting as there are threads to take them: whoever asks does its row itself (frames of few tiles but many of them in flight —
    // dav1d's n_fc — bring more listing threads than the pool has)
    bool backed_up() {
        std::lock_guard<std::mutex> lk(m);
        return threads > 0 && q.size() >= (size_t) threads;
    }
    void push(std::function<void()> job) {
        std::lock_guard<std::mutex> lk(m);
        start_locked();
        q.push_back(std::move(job));
        cv.notify_one();
    }
};
PrepPool &prep_pool() { static PrepPool *p = new PrepPool; return *p; }       // (never destroyed: its threads outlive static destruction)
}
// every preparation the frame handed out is through; the first error among them (once)
static int frame_wait_prep(Dav1dHipFrame *f) {
    std::unique_lock<std::mutex> lk(f->prep_mtx);
    f->prep_cv.wait(lk, [f]() { return f->prep_pending.load() == 0; });
    const int rc = f->prep_rc;
    f->prep_rc = 0;
    return rc;
}

static int submit_tile_sbrow_now(Dav1dHipFrame *f, const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                                const Dav1dHipItxTask *itx, size_t n_itx, bool trusted, const uint16_t *itx_dep);
// recs / cf / dst (the lister's packing rows, or n_recs = 0): the row's coefficient values move from the hand-off's array into the frame's arena
// before the chunk is prepared — on the same thread as the preparation, whichever that is
static int submit_tile_sbrow(Dav1dHipFrame *f, const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                            const Dav1dHipItxTask *itx, size_t n_itx, bool trusted, const uint16_t *itx_dep,
                            const Dav1dHipPackRec *recs, size_t n_recs, void *cf, void *dst) {
    if (!f || (!mc && n_mc) || (!comp && n_comp) || (!itx && n_itx) || (n_recs && (!recs || !cf || !dst))) return -EINVAL;
    if (!n_mc && !n_comp && !n_itx && !n_recs) return 0;
    if ((n_mc || n_comp) && !f->n_refs) return -EINVAL;
    note_kinds(f, itx, n_itx);
    const int mode = f->c->prep_async, csz = f->cur.bpc > 8 ? 4 : 2;
    if (!trusted || !mode || (mode == 1 && !(f->have_tiling && f->tiling.n_cols * f->tiling.n_rows <= 8)) || prep_pool().backed_up()) {
        if (n_recs) dav1d_hip_pack_run(recs, n_recs, cf, dst, csz);
        return n_mc || n_comp || n_itx ? submit_tile_sbrow_now(f, mc, n_mc, comp, n_comp, itx, n_itx, trusted, itx_dep) : 0;
    }
    struct Job {
        std::vector<Dav1dHipMcTask> mc; std::vector<Dav1dHipCompTask> comp; std::vector<Dav1dHipItxTask> itx; std::vector<uint16_t> dep;
        std::vector<Dav1dHipPackRec> recs;
    };
    std::shared_ptr<Job> job;
    try {
        job = std::make_shared<Job>();
        job->mc.assign(mc, mc + n_mc); job->comp.assign(comp, comp + n_comp); job->itx.assign(itx, itx + n_itx);
        if (itx_dep) job->dep.assign(itx_dep, itx_dep + n_itx);
        if (n_recs) job->recs.assign(recs, recs + n_recs);
    } catch (const std::bad_alloc &) { return -ENOMEM; }
    f->prep_pending.fetch_add(1);
    prep_pool().push([f, job, cf, dst, csz]() {
        (void) hipSetDevice(f->c->device);
        if (!job->recs.empty()) dav1d_hip_pack_run(job->recs.data(), job->recs.size(), cf, dst, csz);
        const int rc = job->mc.empty() && job->comp.empty() && job->itx.empty() ? 0 :
                       submit_tile_sbrow_now(f, job->mc.data(), job->mc.size(), job->comp.data(), job->comp.size(), job->itx.data(), job->itx.size(), true,
                                             job->dep.empty() ? nullptr : job->dep.data());
        std::lock_guard<std::mutex> lk(f->prep_mtx);
        if (rc && !f->prep_rc) f->prep_rc = rc;
        f->prep_pending.fetch_sub(1);
        f->prep_cv.notify_all();
    });
    return 0;
}
static int submit_tile_sbrow_now(Dav1dHipFrame *f, const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                                const Dav1dHipItxTask *itx, size_t n_itx, bool trusted, const uint16_t *itx_dep) {
    // all the list preparation of this tile-sbrow happens here, on the submitting thread, without the frame's lock
    Dav1dHipChunk *ck = nullptr;
    // the blob's place in the arena is drawn when its size is known; it is written straight into the arena's pinned twin
    auto place = [](void *cookie, size_t bytes, size_t *dev_off) -> uint8_t * {
        Dav1dHipFrame *fr = (Dav1dHipFrame *) cookie;
        const size_t sz = (bytes + 255) & ~(size_t) 255, off = fr->arena_used.fetch_add(sz);
        *dev_off = off;
        return fr->harena && off + sz <= fr->harena_cap && off + sz <= fr->arena_cap ? fr->harena + off : nullptr;
    };
    const int rc = dav1d_hip_chunk_build(f->c, &ck, &f->cur, f->refs, f->n_refs, mc, n_mc, comp, n_comp, itx, n_itx, place, f, trusted, itx_dep);
    if (rc) return rc;
    // c->chunk_upload (no twin): up it goes on its own, while the other tile-sbrows are still being listed
    if (ck->used && ck->host && !f->harena && f->arena && ck->dev_off + ck->used <= f->arena_cap)
        ck->uploaded = hipMemcpyAsync(f->arena + ck->dev_off, ck->host, ck->used, hipMemcpyHostToDevice, f->c->copy_stream) == hipSuccess;
    std::lock_guard<std::mutex> lk(f->mtx);
    f->chunks.push_back(ck);
    return 0;
}

// Intra blocks (recon_b_intra, src/recon_tmpl.c:1176-1555) of wavefront step `step`: the transform blocks whose left / top /
// top-right neighbours are complete once every inter block of the frame and every intra block of the steps before it is — the
// lister derives the step from the block position inside its superblock and the superblock's own anti-diagonal, as the
// reference's tile threads do with their sbrow progress.  ipred[i] predicts a block; the residuals of the step's blocks go in
// `itx` (any order).  `aux`: DEVICE arena of packed palette indices the PAL tasks point into (NULL if none), the same for all
// calls of a frame.  Thread-safe; steps may arrive in any order.
int dav1d_hip_frame_submit_intra_step(Dav1dHipFrame *f, size_t step, const Dav1dHipIpredTask *ipred, size_t n_ipred,
                                      const Dav1dHipItxTask *itx, size_t n_itx, uint8_t *aux) {
    if (!f || (!ipred && n_ipred) || (!itx && n_itx) || step > 65535) return -EINVAL;
    std::vector<size_t> pe(step + 1, 0), xe(step + 1, 0), be(step + 1, 0);
    pe[step] = n_ipred; xe[step] = n_itx;
    const int rc = dav1d_hip_frame_submit_intra_sorted(f, step + 1, ipred, pe.data(), itx, xe.data(), nullptr, be.data());
    if (!rc && aux) { std::lock_guard<std::mutex> lk(f->mtx); f->aux = aux; }
    return rc;
}

int dav1d_hip_frame_submit_step_copy(Dav1dHipFrame *f, const Dav1dHipMcTask *tasks, const uint16_t *steps, size_t n) {
    if (!f || ((!tasks || !steps) && n)) return -EINVAL;
    for (size_t i = 0; i < n; i++)
        if (!steps[i] || tasks[i].kind != DAV1D_HIP_MC_PUT || tasks[i].plane > 2 || !f->cur.p[tasks[i].plane].data) return -EINVAL;
    std::lock_guard<std::mutex> lk(f->mtx);
    for (size_t i = 0; i < n; i++) {
        Dav1dHipMcTask t = tasks[i];
        t.ref = 0;                                   // the one "reference" these are run with is the frame's own picture
        f->step_copy.push_back(t);
        f->step_copy_step.push_back(steps[i]);
        f->n_steps = std::max(f->n_steps, (size_t) steps[i] + 1);
    }
    return 0;
}

// The blends of inter-intra blocks of wavefront step `step` (kind DAV1D_HIP_COMP_BLEND, tmp1_off = where the step's PRED_TMP task
// wrote the intra prediction): run after the step's predictions and before its residuals — the reference's order inside
// recon_b_inter (src/recon_tmpl.c:1606-1630: intra_pred into tmp, blend, later the residual).  Thread-safe.
int dav1d_hip_frame_submit_step_blend(Dav1dHipFrame *f, size_t step, const Dav1dHipCompTask *blend, size_t n) {
    if (!f || (!blend && n) || step > 65535 || !step) return -EINVAL;
    std::vector<size_t> pe(step + 1, 0), xe(step + 1, 0), be(step + 1, 0);
    be[step] = n;
    return dav1d_hip_frame_submit_intra_sorted(f, step + 1, nullptr, pe.data(), nullptr, xe.data(), blend, be.data());
}

// All intra steps of one submitter in one call: tasks sorted by step, *_end[s] = end offset of step s in its array (steps
// 0 .. n_steps - 1; step 0 stays empty: it is the inter blocks').  One lock per call instead of one per step, and the pairing
// of predictions with their residuals (dataflow launch, intra_flow.hip) happens here, on the submitting thread.
int dav1d_hip_frame_submit_intra_sorted(Dav1dHipFrame *f, size_t n_steps, const Dav1dHipIpredTask *ip, const size_t *ip_end,
                                        const Dav1dHipItxTask *ix, const size_t *ix_end, const Dav1dHipCompTask *bl, const size_t *bl_end) {
    if (!f || !n_steps || n_steps > 65536 || !ip_end || !ix_end || !bl_end) return -EINVAL;
    const size_t np = ip_end[n_steps - 1], nx = ix_end[n_steps - 1], nb = bl_end[n_steps - 1];
    if ((np && !ip) || (nx && !ix) || (nb && !bl) || np >= 0xffffffffu || nx >= 0xffffffffu) return -EINVAL;
    if (!np && !nx && !nb) return 0;
    note_kinds(f, ix, nx);
    Dav1dHipFrame::StepChunk *ck = new (std::nothrow) Dav1dHipFrame::StepChunk();
    if (!ck) return -ENOMEM;
    ck->ip.assign(ip, ip + np); ck->ix.assign(ix, ix + nx); ck->bl.assign(bl, bl + nb);
    ck->ip_end.resize(n_steps); ck->ix_end.resize(n_steps); ck->bl_end.resize(n_steps);
    for (size_t s = 0; s < n_steps; s++) {
        if (ip_end[s] > np || ix_end[s] > nx || bl_end[s] > nb || (s && (ip_end[s] < ip_end[s - 1] || ix_end[s] < ix_end[s - 1] || bl_end[s] < bl_end[s - 1]))) {
            delete ck;
            return -EINVAL;
        }
        ck->ip_end[s] = (uint32_t) ip_end[s]; ck->ix_end[s] = (uint32_t) ix_end[s]; ck->bl_end[s] = (uint32_t) bl_end[s];
    }
    // units (prediction + residual of a transform block): for the dataflow launch (no inter-intra blends) and for the superblock route
    // (which blends inside the unit)
    ck->has_blend = nb != 0;
    const bool want_sb = f->have_tiling && f->c->intra_sb > 0;
    ck->flow_ok = (nb == 0 && f->c->flow_min_steps > 0) || want_sb;
    if (ck->flow_ok) {
        int rc = dav1d_hip_intra_units_build(ck->ip.data(), ck->ip_end.data(), ck->ix.data(), ck->ix_end.data(), n_steps, ck->units, ck->ua_end, ck->ub_end,
                                             nb && want_sb ? ck->bl.data() : nullptr, ck->bl_end.data());
        if (rc == -ENOTSUP) { ck->flow_ok = false; ck->units.clear(); rc = 0; }
        if (rc) { delete ck; return rc; }
    }
    ck->sb_sorted = false;
    if (ck->flow_ok && f->have_tiling && f->c->intra_sb > 0) {
        // the superblock wavefront takes the units superblock by superblock: sorted here, on the submitting thread
        const DevPlanes dp = dev_planes(&f->cur);
        SbSort st;
        const int rc = dav1d_hip_sbw_prepare(ck->units, ck->ua_end, ck->ub_end, f->tiling, dp.stride, f->cur.layout != DAV1D_HIP_LAYOUT_I444,
                                             f->cur.layout == DAV1D_HIP_LAYOUT_I420, st);
        if (rc) { delete ck; return rc; }
        const size_t n = st.n_records, bytes = n * sizeof(IntraUnit);
        {
            std::lock_guard<std::mutex> lk(f->mtx);
            if (!f->huarena) {
                const size_t want = std::max<size_t>(f->c->uarena_hint + (f->c->uarena_hint >> 2), std::min<size_t>(f->c->arena_min, (size_t) 1 << 20));
                f->huarena = dav1d_hip_slab_get(f->c, want, &f->huarena_cap);
                if (!f->huarena) f->huarena_cap = 0;
            }
        }
        const size_t off = f->uarena_used.fetch_add(bytes);
        ck->in_arena = f->huarena && off + bytes <= f->huarena_cap;
        std::vector<IntraUnit> own;
        if (!ck->in_arena) own.resize(n);
        ck->sorted = ck->in_arena ? reinterpret_cast<IntraUnit *>(f->huarena + off) : own.data();
        ck->n_sorted = n;
        dav1d_hip_sbw_emit(ck->units, st, ck->sorted);
        ck->parts.swap(st.parts);
        ck->copy_deps.swap(st.copy_deps);
        ck->has_pal = false;
        for (size_t i = 0; i < ck->units.size() && !ck->has_pal; i++) ck->has_pal = (ck->units[i].has & 1) && ck->units[i].p.kind == DAV1D_HIP_IPRED_PAL;
        if (ck->in_arena) { std::vector<IntraUnit>().swap(ck->units); }
        else { ck->units.swap(own); ck->sorted = ck->units.data(); }
        ck->sb_sorted = true;
    }
    std::lock_guard<std::mutex> lk(f->mtx);
    f->step_chunks.push_back(ck);
    if (n_steps > f->n_steps) f->n_steps = n_steps;
    return 0;
}

// The frame's tiles in superblocks (frame_hdr->tiling, seq_hdr->sb128): with them the intra blocks of the frame run superblock by
// superblock (intra_sb.hip) instead of step by step.  Before the first intra submission; the listers call it.
int dav1d_hip_frame_set_tiling(Dav1dHipFrame *f, int sb128, int n_tile_cols, const uint16_t *col_start_sb, int n_tile_rows, const uint16_t *row_start_sb) {
    if (!f) return -EINVAL;
    std::lock_guard<std::mutex> lk(f->mtx);
    if (!f->step_chunks.empty()) return -EINVAL;
    const int rc = dav1d_hip_sb_tiling_make(&f->tiling, f->cur.p[0].w, f->cur.p[0].h, sb128, n_tile_cols, col_start_sb, n_tile_rows, row_start_sb);
    f->have_tiling = !rc;
    if (!rc) f->sb_dep.assign((size_t) f->tiling.sbw * f->tiling.sbh, 15);
    return rc;
}

// Which neighbouring superblocks the intra blocks of superblocks [first, first + n) (raster numbers, frame-wide) read pixels of that
// intra blocks of THIS frame wrote: bit 0 left, 1 top-left, 2 top, 3 top-right.  Without it every neighbour that holds intra units
// counts (correct, but an inter frame's scattered intra blocks then queue behind each other for nothing).  Callers write disjoint
// ranges (a tile-sbrow each); no lock.
int dav1d_hip_frame_set_sb_deps(Dav1dHipFrame *f, uint32_t first, size_t n, const uint8_t *mask) {
    if (!f || !mask || !f->have_tiling || (size_t) first + n > f->sb_dep.size()) return -EINVAL;
    memcpy(f->sb_dep.data() + first, mask, n);
    return 0;
}

// Warped predictions (warp_affine(), src/recon_tmpl.c:1115-1174) and predictions from references of another size (the scaled
// branch of mc(), :990-1047) of any tile-sbrow: they read reference pictures only and run before the compound combinations.
int dav1d_hip_frame_submit_warp(Dav1dHipFrame *f, const Dav1dHipWarpTask *t, size_t n) {
    if (!f || (!t && n)) return -EINVAL;
    std::lock_guard<std::mutex> lk(f->mtx);
    f->warp.insert(f->warp.end(), t, t + n);
    return 0;
}
int dav1d_hip_frame_submit_scaled(Dav1dHipFrame *f, const Dav1dHipMcScaledTask *t, size_t n) {
    if (!f || (!t && n)) return -EINVAL;
    std::lock_guard<std::mutex> lk(f->mtx);
    f->scaled.insert(f->scaled.end(), t, t + n);
    return 0;
}

// internal: the picture the frame reconstructs into (the lister needs its strides)
int dav1d_hip_frame_picture(const Dav1dHipFrame *f, Dav1dHipPicture *out) {
    if (!f || !out) return -EINVAL;
    *out = f->cur;
    return 0;
}

// In-loop filter tasks of one superblock row (dav1d_filter_sbrow_deblock_cols / _rows / _cdef / _lr, src/recon_tmpl.c:1985-2135).
// Loop restoration tasks must arrive in raster order inside a superblock row.  Thread-safe.
// the arrays become the frame's (malloc'ed by the caller, freed by the frame): what the filter lister calls
int dav1d_hip_frame_submit_filter_owned(Dav1dHipFrame *f, Dav1dHipLfTask *lf, size_t n_lf, Dav1dHipCdefTask *cdef, size_t n_cdef,
                                        Dav1dHipLrTask *lr, size_t n_lr) {
    if (!f || (!lf && n_lf) || (!cdef && n_cdef) || (!lr && n_lr)) return -EINVAL;
    // the checks of dav1d_hip_lf_batch / _cdef_batch / _lr_batch, here and in parallel instead of at frame end
    unsigned bad = 0;
    size_t n0 = 0;
    for (size_t i = 0; i < n_lf; i++) { bad |= (unsigned) (lf[i].plane > 2) | (unsigned) (lf[i].dir > 1) | (unsigned) (lf[i].lvl_comp > 3); n0 += lf[i].dir == 0; }
    for (size_t i = 0; i < n_cdef; i++) bad |= (unsigned) (cdef[i].edges > 15) | (unsigned) (cdef[i].plane > 2) | (unsigned) (cdef[i].dir > 7);
    for (size_t i = 0; i < n_lr; i++)
        bad |= (unsigned) (lr[i].plane > 2) | (unsigned) (lr[i].edges > 15) | (unsigned) (!lr[i].w) | (unsigned) (lr[i].w > 384) | (unsigned) (!lr[i].h) |
               (unsigned) (lr[i].h > 64) | (unsigned) (lr[i].type > DAV1D_HIP_LR_SGR_MIX);
    if (bad) return -EINVAL;
    Dav1dHipFrame::FilterPiece p = { lf, n_lf, n0, cdef, n_cdef, 0, nullptr, lr, n_lr };
    if (n_lf && n0 && n0 < n_lf) {
        // vertical edges first (stable): the two deblocking launches read the halves
        Dav1dHipLfTask *q = (Dav1dHipLfTask *) malloc(n_lf * sizeof(*q));
        if (!q) return -ENOMEM;
        size_t a = 0, b = n0;
        for (size_t i = 0; i < n_lf; i++) q[lf[i].dir ? b++ : a++] = lf[i];
        memcpy(lf, q, n_lf * sizeof(*q));                 // the caller's array stays the piece's array (it is handed over either way)
        free(q);
    }
    if (n_cdef) {
        p.groups = new (std::nothrow) std::vector<CdefGroup>();
        if (!p.groups) return -ENOMEM;
        p.groups->reserve(n_cdef / 8 + 16);
        p.n_raw = dav1d_hip_cdef_make_groups(cdef, n_cdef, 0, *p.groups);
    }
    std::lock_guard<std::mutex> lk(f->mtx);
    f->filter_pieces.push_back(p);
    return 0;
}

int dav1d_hip_frame_submit_filter_sbrow(Dav1dHipFrame *f, const Dav1dHipLfTask *lf, size_t n_lf, const Dav1dHipCdefTask *cdef, size_t n_cdef,
                                        const Dav1dHipLrTask *lr, size_t n_lr) {
    if (!f || (!lf && n_lf) || (!cdef && n_cdef) || (!lr && n_lr)) return -EINVAL;
    // copies made outside the lock, handed over as a piece
    Dav1dHipLfTask *a = n_lf ? (Dav1dHipLfTask *) malloc(n_lf * sizeof(*a)) : nullptr;
    Dav1dHipCdefTask *b = n_cdef ? (Dav1dHipCdefTask *) malloc(n_cdef * sizeof(*b)) : nullptr;
    Dav1dHipLrTask *d = n_lr ? (Dav1dHipLrTask *) malloc(n_lr * sizeof(*d)) : nullptr;
    if ((n_lf && !a) || (n_cdef && !b) || (n_lr && !d)) { free(a); free(b); free(d); return -ENOMEM; }
    if (n_lf) memcpy(a, lf, n_lf * sizeof(*a));
    if (n_cdef) memcpy(b, cdef, n_cdef * sizeof(*b));
    if (n_lr) memcpy(d, lr, n_lr * sizeof(*d));
    const int rc = dav1d_hip_frame_submit_filter_owned(f, a, n_lf, b, n_cdef, d, n_lr);
    if (rc) { free(a); free(b); free(d); }
    return rc;
}

// internal (host/filter_lister.c): the frame's table of CDEF unit rows, or nullptr when this frame takes unit records
extern "C" Dav1dHipCdefRow *dav1d_hip_frame_cdef_rows(Dav1dHipFrame *f, int *stride) {
    if (!f) return nullptr;
    std::lock_guard<std::mutex> lk(f->mtx);
    if (!f->cdef_rows_state) {
        Dav1dHipContext *c = f->c;
        const DevPlanes cur = dev_planes(&f->cur);
        f->cdef_rows_state = -1;
        if (c->cdef_rows && c->post_bands < 2 && !c->cdef_unit_kernel && dav1d_hip_cdef_strip_ok(&cur, &cur, f->cur.bpc)) {
            f->cdef_w64 = (f->cur.p[0].w + 63) >> 6; f->cdef_h8 = (f->cur.p[0].h + 7) >> 3;
            const size_t bytes = (size_t) f->cdef_w64 * f->cdef_h8 * sizeof(Dav1dHipCdefRow);
            f->cdef_rows = reinterpret_cast<Dav1dHipCdefRow *>(dav1d_hip_slab_get(c, bytes, &f->cdef_rows_cap));
            if (f->cdef_rows) { memset(f->cdef_rows, 0, bytes); f->cdef_rows_state = 1; }
        }
    }
    if (f->cdef_rows_state != 1) return nullptr;
    if (f->cdef_rows_dirty) { memset(f->cdef_rows, 0, (size_t) f->cdef_w64 * f->cdef_h8 * sizeof(Dav1dHipCdefRow)); f->cdef_rows_dirty = false; }
    *stride = f->cdef_w64;
    return f->cdef_rows;
}
extern "C" void dav1d_hip_frame_cdef_rows_add(Dav1dHipFrame *f, size_t n_units) { if (f) f->cdef_row_units.fetch_add(n_units); }

// the table as unit records, on the host (the routes that work on merged task lists; mixed submissions): a piece like any other
static int frame_cdef_rows_to_piece(Dav1dHipFrame *f) {
    const size_t n = f->cdef_row_units.load();
    if (f->cdef_rows_state != 1 || !n) return 0;
    Dav1dHipCdefTask *t = (Dav1dHipCdefTask *) malloc(n * sizeof(*t));
    if (!t) return -ENOMEM;
    const int bw4 = 2 * ((f->cur.p[0].w + 7) >> 3), bh4 = 2 * ((f->cur.p[0].h + 7) >> 3);      // f->bw, f->bh of the reference: whole 8x8 blocks
    size_t k = 0;
    for (int by = 0; by < f->cdef_h8; by++)
        for (int sbx = 0; sbx < f->cdef_w64; sbx++) {
            const Dav1dHipCdefRow &q = f->cdef_rows[(size_t) by * f->cdef_w64 + sbx];
            for (unsigned m = q.mask; m && k < n; m &= m - 1) {
                const int bx = 8 * sbx + __builtin_ctz(m);
                Dav1dHipCdefTask &o = t[k++];
                memset(&o, 0, sizeof(o));
                o.bx = (uint16_t) bx; o.by = (uint16_t) by;
                o.y_pri = q.y_pri; o.y_sec = q.y_sec; o.uv_pri = q.uv_pri; o.uv_sec = q.uv_sec;
                o.flags = q.flags;
                o.edges = (uint8_t) ((bx > 0 ? DAV1D_HIP_CDEF_HAVE_LEFT : 0) | (2 * bx + 2 < bw4 ? DAV1D_HIP_CDEF_HAVE_RIGHT : 0) |
                                     (by > 0 ? DAV1D_HIP_CDEF_HAVE_TOP : 0) | (2 * by + 2 < bh4 ? DAV1D_HIP_CDEF_HAVE_BOTTOM : 0));
            }
        }
    Dav1dHipFrame::FilterPiece p = { nullptr, 0, 0, t, k, 0, nullptr, nullptr, 0 };
    p.groups = new (std::nothrow) std::vector<CdefGroup>();
    if (!p.groups) { free(t); return -ENOMEM; }
    p.n_raw = dav1d_hip_cdef_make_groups(t, k, 0, *p.groups);
    f->filter_pieces.push_back(p);
    f->cdef_row_units.store(0);
    memset(f->cdef_rows, 0, (size_t) f->cdef_w64 * f->cdef_h8 * sizeof(Dav1dHipCdefRow));
    return 0;
}

// the pieces -> f->lf / f->cdef / f->lr (once, at frame end; submission order)
static int frame_merge_filter_pieces(Dav1dHipFrame *f) {
    const int rc_rows = frame_cdef_rows_to_piece(f);       // (-ENOMEM: the frame must not go on without its CDEF units — ADVICE r4)
    if (rc_rows) return rc_rows;
    if (f->filter_pieces.empty()) return 0;
    size_t a = f->lf.size(), b = f->cdef.size(), d = f->lr.size();
    for (const Dav1dHipFrame::FilterPiece &p : f->filter_pieces) { a += p.n_lf; b += p.n_cdef; d += p.n_lr; }
    f->lf.reserve(a); f->cdef.reserve(b); f->lr.reserve(d);
    for (Dav1dHipFrame::FilterPiece &p : f->filter_pieces) {
        f->lf.insert(f->lf.end(), p.lf, p.lf + p.n_lf);
        f->cdef.insert(f->cdef.end(), p.cdef, p.cdef + p.n_cdef);
        f->lr.insert(f->lr.end(), p.lr, p.lr + p.n_lr);
        f->lrplan.valid = false;            // (the plan was made for the list as it was)
        free(p.lf); free(p.cdef); free(p.lr);
        delete p.groups;
    }
    f->filter_pieces.clear();
    return 0;
}

static int copy_picture(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src);
static int frame_lr_plan(Dav1dHipFrame *f);
static int copy_unrestored_planes(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src, const std::vector<Dav1dHipLrTask> &lr);

struct FrameTrace {
    bool on;
    std::chrono::steady_clock::time_point t;
    double ms[8];
    FrameTrace() : on(getenv("DAV1D_HIP_TRACE_FRAME") != nullptr), t(std::chrono::steady_clock::now()) { for (double &v : ms) v = 0; }
    void mark(int k) { if (!on) return; const auto n = std::chrono::steady_clock::now(); ms[k] += std::chrono::duration<double, std::milli>(n - t).count(); t = n; }
};

// Deblocking and CDEF of a frame straight from the pieces: the tasks copied piece after piece into pinned memory (the larger
// arrays on a few threads), one upload each, the launches of dav1d_hip_lf_batch / dav1d_hip_cdef_run_groups.  *did_cdef: the
// CDEF output is in f->tmp[0].
static int frame_filters_from_pieces(Dav1dHipFrame *f, bool *did_cdef) {
    Dav1dHipContext *c = f->c;
    *did_cdef = false;
    FrameTrace tr;
    // CDEF came as unit rows: expanded on the device below — unless unit records were submitted as well (one list then, made here)
    bool rows = f->cdef_rows_state == 1 && f->cdef_row_units.load() > 0;
    if (rows) {
        bool mixed = false;
        for (const Dav1dHipFrame::FilterPiece &p : f->filter_pieces) mixed |= p.n_cdef != 0;
        if (mixed) { const int rc = frame_cdef_rows_to_piece(f); if (rc) return rc; rows = false; }
    }
    // restoration units: few; they go through the merged vector, and their host-side preparation (frame_lr_plan) runs below while the
    // device works on the stages before
    for (Dav1dHipFrame::FilterPiece &p : f->filter_pieces) { f->lr.insert(f->lr.end(), p.lr, p.lr + p.n_lr); p.n_lr = 0; f->lrplan.valid = false; }
    size_t n_lf = 0, n_lf0 = 0, n_cdef = 0, n_groups = 0, n_raw = 0;
    for (const Dav1dHipFrame::FilterPiece &p : f->filter_pieces) {
        n_lf += p.n_lf; n_lf0 += p.n_lf0; n_cdef += p.n_cdef; n_raw += p.n_raw;
        if (p.groups) n_groups += p.groups->size();
    }
    int rc = 0;
    const DevPlanes cur = dev_planes(&f->cur);
    if (n_lf) {
        if (!f->lvl || !f->have_lut) return -EINVAL;
        size_t cap = 0;
        Dav1dHipLfTask *host = reinterpret_cast<Dav1dHipLfTask *>(dav1d_hip_slab_get(c, n_lf * sizeof(Dav1dHipLfTask), &cap));
        if (!host) return -ENOMEM;
        size_t a = 0, b = n_lf0;
        for (const Dav1dHipFrame::FilterPiece &p : f->filter_pieces) {
            if (p.n_lf0) memcpy(host + a, p.lf, p.n_lf0 * sizeof(*host));
            if (p.n_lf > p.n_lf0) memcpy(host + b, p.lf + p.n_lf0, (p.n_lf - p.n_lf0) * sizeof(*host));
            a += p.n_lf0; b += p.n_lf - p.n_lf0;
        }
        tr.mark(0);
        f->deferred.slabs.push_back({ reinterpret_cast<uint8_t *>(host), cap });
        f->deferred.bufs.emplace_back(new TaskBuf(c, n_lf * sizeof(Dav1dHipLfTask)));
        Dav1dHipLfTask *const dev = reinterpret_cast<Dav1dHipLfTask *>(f->deferred.bufs.back()->p);
        if (!dev) rc = -ENOMEM;
        if (!rc) rc = hip_rc(hipMemcpyAsync(dev, host, n_lf * sizeof(*dev), hipMemcpyHostToDevice, c->stream));
        if (!rc) rc = dav1d_hip_launch_lf(&cur, f->cur.bpc, 0, dev, (int) n_lf0, f->lvl, (int) f->b4_stride, f->lut_e, f->lut_i, c->stream);
        if (!rc) rc = dav1d_hip_launch_lf(&cur, f->cur.bpc, 1, dev + n_lf0, (int) (n_lf - n_lf0), f->lvl, (int) f->b4_stride, f->lut_e, f->lut_i, c->stream);
        if (rc) return rc;
    }
    tr.mark(1);
    if (rows) {
        rc = frame_tmp(f, 0);
        if (rc) return rc;
        const DevPlanes t0 = dev_planes(&f->tmp[0]);
        const int w64 = f->cdef_w64, h8 = f->cdef_h8, w8 = (f->cur.p[0].w + 7) >> 3;
        const size_t n_units = f->cdef_row_units.load(), n_slots = (size_t) ((w64 + 1) >> 1) * h8;
        const bool covered = !c->cdef_full_copy && n_units == (size_t) w8 * h8;
        if (c->cdef_full_copy) rc = copy_picture(c, &f->tmp[0], &f->cur);
        if (rc) return rc;
        const size_t rb = ((size_t) w64 * h8 * sizeof(Dav1dHipCdefRow) + 255) & ~(size_t) 255, tb = n_slots * 16 * sizeof(Dav1dHipCdefTask);
        const size_t gb = (n_slots * sizeof(CdefGroup) + 255) & ~(size_t) 255;
        const size_t bm_bytes = covered || c->cdef_full_copy ? 0 : (((size_t) w8 * h8 + 31) / 32 * 4 + 255) & ~(size_t) 255;
        f->deferred.bufs.emplace_back(new TaskBuf(c, rb + tb + gb + bm_bytes + 256));
        uint8_t *const dev = f->deferred.bufs.back()->p;
        if (!dev) return -ENOMEM;
        rc = hip_rc(hipMemcpyAsync(dev, f->cdef_rows, (size_t) w64 * h8 * sizeof(Dav1dHipCdefRow), hipMemcpyHostToDevice, c->stream));
        Dav1dHipCdefTask *const d_tasks = reinterpret_cast<Dav1dHipCdefTask *>(dev + rb);
        uint32_t *const d_bm = bm_bytes ? reinterpret_cast<uint32_t *>(dev + rb + tb + gb) : nullptr;
        if (!rc) rc = dav1d_hip_launch_cdef_expand(dev, w64, h8, 2 * w8, 2 * h8, w8, d_tasks, dev + rb + tb, d_bm, c->stream);
        if (!rc && d_bm) rc = dav1d_hip_launch_cdef_fill_unlisted(&t0, &cur, f->cur.bpc, f->cur.layout, nullptr, 0, d_bm, w8, h8, c->stream);
        if (!rc) rc = dav1d_hip_launch_cdef_groups(&t0, &cur, f->cur.bpc, f->cur.layout, d_tasks, reinterpret_cast<const CdefGroup *>(dev + rb + tb), (int) n_slots,
                                                   f->cdef_damping, nullptr, c->stream);
        f->cdef_row_units.store(0);              // consumed, like the pieces; the table is wiped before anybody writes to it again (after frame_run's wait)
        f->cdef_rows_dirty = true;
        if (rc) return rc;
        *did_cdef = true;
    } else if (n_cdef) {
        rc = frame_tmp(f, 0);
        if (rc) return rc;
        const DevPlanes t0 = dev_planes(&f->tmp[0]);
        const bool strips = !c->cdef_unit_kernel && dav1d_hip_cdef_strip_ok(&t0, &cur, f->cur.bpc);
        // Units that are not listed keep their pixels.  The strip kernel writes every unit it is given (also the planes it does not
        // filter), so only the UNLISTED units have to be brought over: none when the list covers the frame, the ones a bitmap of the
        // listed units leaves out otherwise (cdef.hip cdef_fill_unlisted_kernel) — not a copy of the whole picture under the filter
        // (100 MB of traffic for an 8K frame, 40 us).  The unit-per-wave kernel and DSP-level (RAW) tasks keep the copy.
        const int w8 = (f->cur.p[0].w + 7) >> 3, h8 = (f->cur.p[0].h + 7) >> 3;
        const bool fill = strips && !n_raw && !c->cdef_full_copy;
        const bool covered = fill && n_cdef == (size_t) w8 * h8;
        const size_t bm_bytes = fill && !covered ? (((size_t) w8 * h8 + 31) / 32 * 4 + 255) & ~(size_t) 255 : 0;
        if (!fill) rc = copy_picture(c, &f->tmp[0], &f->cur);
        if (rc) return rc;
        const size_t tb = (n_cdef * sizeof(Dav1dHipCdefTask) + 255) & ~(size_t) 255;
        const size_t gb = (n_groups * sizeof(CdefGroup) + 255) & ~(size_t) 255;
        size_t cap = 0;
        uint8_t *host = dav1d_hip_slab_get(c, tb + n_groups * sizeof(CdefGroup) + 256, &cap);
        if (!host) return -ENOMEM;
        Dav1dHipCdefTask *ht = reinterpret_cast<Dav1dHipCdefTask *>(host);
        CdefGroup *hg = reinterpret_cast<CdefGroup *>(host + tb);
        // where every piece goes, then the copies on a few threads (8 MB of unit records for an 8K frame)
        const size_t np = f->filter_pieces.size();
        std::vector<size_t> t_off(np + 1, 0), g_off(np + 1, 0);
        for (size_t k = 0; k < np; k++) {
            t_off[k + 1] = t_off[k] + f->filter_pieces[k].n_cdef;
            g_off[k + 1] = g_off[k] + (f->filter_pieces[k].groups ? f->filter_pieces[k].groups->size() : 0);
        }
        auto copy_pieces = [&](size_t k0, size_t k1) {
            for (size_t k = k0; k < k1; k++) {
                const Dav1dHipFrame::FilterPiece &p = f->filter_pieces[k];
                if (p.n_cdef) memcpy(ht + t_off[k], p.cdef, p.n_cdef * sizeof(*ht));
                if (p.groups)
                    for (size_t i = 0; i < p.groups->size(); i++) { CdefGroup g = (*p.groups)[i]; g.first += (uint32_t) t_off[k]; hg[g_off[k] + i] = g; }
            }
        };
        {
            const unsigned hw = std::thread::hardware_concurrency();
            const size_t nt = std::max<size_t>(1, std::min<size_t>({ (size_t) 4, hw ? hw : 1, np / 4 + 1 }));
            std::vector<std::thread> th;
            for (size_t t = 1; t < nt; t++) th.emplace_back(copy_pieces, np * t / nt, np * (t + 1) / nt);
            copy_pieces(0, np / nt);
            for (std::thread &x : th) x.join();
        }
        f->deferred.slabs.push_back({ host, cap });
        f->deferred.bufs.emplace_back(new TaskBuf(c, tb + gb + bm_bytes + 256));
        uint8_t *const dev = f->deferred.bufs.back()->p;
        if (!dev) rc = -ENOMEM;
        if (!rc) rc = hip_rc(hipMemcpyAsync(dev, host, tb + n_groups * sizeof(CdefGroup), hipMemcpyHostToDevice, c->stream));
        const Dav1dHipCdefTask *d_tasks = reinterpret_cast<const Dav1dHipCdefTask *>(dev);
        if (!rc && bm_bytes)
            rc = dav1d_hip_launch_cdef_fill_unlisted(&t0, &cur, f->cur.bpc, f->cur.layout, d_tasks, (int) n_cdef,
                                                     reinterpret_cast<uint32_t *>(dev + tb + gb), w8, h8, c->stream);
        if (!rc && strips) {
            rc = dav1d_hip_launch_cdef_groups(&t0, &cur, f->cur.bpc, f->cur.layout, d_tasks, reinterpret_cast<const CdefGroup *>(dev + tb), (int) n_groups,
                                              f->cdef_damping, nullptr, c->stream);
            if (!rc && n_raw) rc = dav1d_hip_launch_cdef(&t0, &cur, f->cur.bpc, f->cur.layout, d_tasks, (int) n_cdef, f->cdef_damping, nullptr, 1, c->stream);
        } else if (!rc) {
            rc = dav1d_hip_launch_cdef(&t0, &cur, f->cur.bpc, f->cur.layout, d_tasks, (int) n_cdef, f->cdef_damping, nullptr, 0, c->stream);
        }
        if (rc) return rc;
        *did_cdef = true;
    }
    // the pieces' arrays were read by the copies above (host to pinned memory): they can go; the device works on, and the host half of
    // restoration is made meanwhile
    tr.mark(2);
    rc = frame_lr_plan(f);
    if (rc) return rc;
    tr.mark(3);
    for (Dav1dHipFrame::FilterPiece &p : f->filter_pieces) {
        free(p.lf); free(p.cdef); free(p.lr);
        delete p.groups;
    }
    f->filter_pieces.clear();
    tr.mark(4);
    if (tr.on) fprintf(stderr, "filters from pieces: deblocking tasks to pinned memory %.3f  enqueue %.3f  CDEF enqueue %.3f  restoration plan %.3f  pieces freed %.3f ms\n",
                       tr.ms[0], tr.ms[1], tr.ms[2], tr.ms[3], tr.ms[4]);
    return 0;
}

// Frame-level filter parameters: the level array (DEVICE, f->lf.level layout), the E / I tables of Av1FilterLUT, the frame's
// CDEF damping (frame_hdr->cdef.damping + bpc - 8), and optionally the film grain parameters.
int dav1d_hip_frame_set_filters(Dav1dHipFrame *f, const uint8_t *lvl, ptrdiff_t b4_stride, const uint8_t lut_e[64], const uint8_t lut_i[64],
                                int cdef_damping, const Dav1dHipFilmGrainData *grain, int is_id) {
    if (!f) return -EINVAL;
    std::lock_guard<std::mutex> lk(f->mtx);
    f->lvl = lvl;
    f->b4_stride = b4_stride;
    if (lut_e) memcpy(f->lut_e, lut_e, 64);
    if (lut_i) memcpy(f->lut_i, lut_i, 64);
    f->have_lut = lut_e && lut_i;
    f->cdef_damping = cdef_damping;
    f->have_grain = grain != nullptr;
    if (f->prepared) { dav1d_hip_fg_grain_destroy(f->c, f->prepared); f->prepared = nullptr; }
    if (grain) {
        f->grain = *grain;
        // the grain of the frame only depends on these parameters: start generating it now, next to the reconstruction
        const int rc = dav1d_hip_fg_prepare(f->c, &f->prepared, grain, f->cur.bpc, f->cur.layout);
        if (rc) return rc;
    }
    f->is_id = is_id;
    return 0;
}

static int copy_unrestored_planes(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src, const std::vector<Dav1dHipLrTask> &lr) {
    if (c->cdef_full_copy) return copy_picture(c, dst, src);
    uint64_t area[3] = { 0, 0, 0 };
    for (const Dav1dHipLrTask &t : lr) if (t.plane < 3) area[t.plane] += (uint64_t) t.w * t.h;
    const int bps = src->bpc > 8 ? 2 : 1;
    for (int pl = 0; pl < 3; pl++) {
        if (!src->p[pl].data || area[pl] == (uint64_t) src->p[pl].w * src->p[pl].h) continue;
        const int rc = hip_rc(hipMemcpy2DAsync(dst->p[pl].data, dst->p[pl].stride, src->p[pl].data, src->p[pl].stride,
                                               (size_t) src->p[pl].w * bps, src->p[pl].h, hipMemcpyDeviceToDevice, c->stream));
        if (rc) return rc;
    }
    return 0;
}

static int copy_picture(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src) {
    if (dst->alloc_size && dst->alloc_size == src->alloc_size && dst->alloc && src->alloc)
        return hip_rc(hipMemcpyAsync(dst->alloc, src->alloc, src->alloc_size, hipMemcpyDeviceToDevice, c->stream));
    const int bps = src->bpc > 8 ? 2 : 1;
    for (int pl = 0; pl < 3; pl++) {
        if (!src->p[pl].data) continue;
        // whole 8-pixel column groups (inside both strides): the super-resolution upscale resamples from the 8x8 block grid, and
        // columns [w, round8(w)) of an 8x8 unit CDEF leaves alone must be the deblocked ones here as in the whole-allocation copy
        size_t row_bytes = (size_t) ((src->p[pl].w + 7) & ~7) * bps;
        row_bytes = std::min(row_bytes, (size_t) std::min(src->p[pl].stride, dst->p[pl].stride));
        const int rc = hip_rc(hipMemcpy2DAsync(dst->p[pl].data, dst->p[pl].stride, src->p[pl].data, src->p[pl].stride,
                                               row_bytes, src->p[pl].h, hipMemcpyDeviceToDevice, c->stream));
        if (rc) return rc;
    }
    return 0;
}


// Row-granular progress without banding the whole frame (the reference publishes f->sr_cur.progress[1] after the last filter of every
// superblock row, src/thread_task.c:888-896; post_filters_pipelined() above follows every band through all three stages on three
// streams and pays 70 % for it).  Here only the LAST stage of a frame tells where it is, and from INSIDE its two launches (frame_lr_run): the
// restoration tasks are ordered by band of 256 luma rows, every workgroup that has written its pixels bumps its band's counter, and
// the one that completes a band writes to a word of pinned host memory the ending thread polls (lr.hip band_done).  Cutting the stage
// into a launch pair per band with an event behind each — the obvious way — was measured at 8K: 34 launches of 43 us each (a band
// does not fill the device, a launch's time is one workgroup's latency) against 126 us for the two, +58 % on the frame.
// Rows are final up to where the NEXT band's first stripe begins (restoration stripes start 8 rows above the 64-row grid,
// src/lr_apply_tmpl.c:176-199).
// The host half of the restoration stage — tasks validated, Wiener first, self-guided units sorted into rows and waves, for a listener
// band by band with the bands' wave counts — made as soon as the frame's restoration tasks are known, i.e. while the launches of the
// stages before are still running (frame_filters_from_pieces calls it ahead of its waits): sorting 10,000 stripes takes 0.14 ms, a
// tenth of an 8K frame's frame_end when the device sits idle through it.
static int frame_lr_plan(Dav1dHipFrame *f) {
    Dav1dHipFrame::LrPlan &pl = f->lrplan;
    if (pl.valid && pl.sorted.size() == f->lr.size() && pl.banded == (f->progress_cb && !f->sr_w && pl.nb >= 2 && pl.nb <= 64 && !f->lr.empty())) return 0;
    const int H = f->cur.p[0].h, ss_ver = f->cur.layout == DAV1D_HIP_LAYOUT_I420;
    const int band_h = 256;
    const size_t n = f->lr.size();
    pl.nb = (H + band_h - 1) / band_h;
    pl.banded = f->progress_cb && !f->sr_w && pl.nb >= 2 && pl.nb <= 64 && n;
    const int nb = pl.banded ? pl.nb : 1;
    std::vector<int> &band = pl.band;
    std::vector<size_t> &off = pl.off, &pos = pl.pos;
    band.resize(n);
    off.assign(2 * nb + 1, 0);                    // [kind * nb + band]
    pl.first_y.assign(nb + 1, H);
    for (size_t i = 0; i < n; i++) {
        const Dav1dHipLrTask &t = f->lr[i];
        if (t.plane > 2 || t.edges > 15 || !t.w || t.w > 384 || !t.h || t.h > 64 || t.type > DAV1D_HIP_LR_SGR_MIX) return -EINVAL;
        const int y = (int) t.y << (t.plane ? ss_ver : 0);
        band[i] = pl.banded ? std::min(y / band_h, nb - 1) : 0;
        pl.first_y[band[i]] = std::min(pl.first_y[band[i]], y);
        off[(t.type > DAV1D_HIP_LR_WIENER5) * nb + band[i] + 1]++;
    }
    for (int k = 0; k < 2 * nb; k++) off[k + 1] += off[k];
    // tasks: [Wiener, band by band][self-guided, band by band, each band's units sorted into rows]
    pl.sorted.resize(n);
    pos.assign(off.begin(), off.end() - 1);
    for (size_t i = 0; i < n; i++) {
        Dav1dHipLrTask &d = pl.sorted[pos[(f->lr[i].type > DAV1D_HIP_LR_WIENER5) * nb + band[i]]++] = f->lr[i];
        d.pad = (uint8_t) band[i];
    }
    for (int b = nb - 1; b >= 0; b--) pl.first_y[b] = std::min(pl.first_y[b], pl.first_y[b + 1]);
    pl.nw = off[nb];                               // Wiener tasks
    pl.tail.clear();                               // the wave descriptors, then (banded) the 64 band targets: one upload
    pl.target.assign(64, 0);
    pl.max_w = 0;
    for (size_t i = 0; i < pl.nw; i++) { pl.max_w = std::max(pl.max_w, (int) pl.sorted[i].w); pl.target[pl.sorted[i].pad] += (uint32_t) ((pl.sorted[i].w + 63) / 64); }
    for (int b = 0; b < nb; b++) {
        const size_t s0 = off[nb + b], s1 = off[nb + b + 1], w0 = pl.tail.size();
        dav1d_hip_sgr_make_rows(pl.sorted.data() + s0, s1 - s0, pl.tail);
        for (size_t k = w0; k < pl.tail.size(); k += 4) {       // the band's descriptors count from its first task: from the first self-guided task instead
            pl.tail[k] += (uint32_t) (s0 - pl.nw); pl.tail[k + 1] += (uint32_t) (s0 - pl.nw); pl.tail[k + 3] = (uint32_t) b;
        }
        pl.target[b] += (uint32_t) ((pl.tail.size() - w0) / 4);
    }
    pl.n_waves = pl.tail.size() / 4;
    if (pl.banded) pl.tail.insert(pl.tail.end(), pl.target.begin(), pl.target.end());
    // what goes to the device, in pinned memory: [tasks][wave descriptors (+ band targets)] — one asynchronous copy at run time
    pl.o_waves = (n * sizeof(Dav1dHipLrTask) + 15) & ~(size_t) 15;
    pl.bytes = pl.o_waves + pl.tail.size() * 4;
    if (pl.slab && pl.slab_cap < pl.bytes + 16) { dav1d_hip_slab_put(f->c, pl.slab, pl.slab_cap); pl.slab = nullptr; }
    if (!pl.slab && n) pl.slab = dav1d_hip_slab_get(f->c, pl.bytes + 16, &pl.slab_cap);
    if (n && !pl.slab) return -ENOMEM;
    if (n) {
        memcpy(pl.slab, pl.sorted.data(), n * sizeof(Dav1dHipLrTask));
        if (!pl.tail.empty()) memcpy(pl.slab + pl.o_waves, pl.tail.data(), pl.tail.size() * 4);
    }
    pl.valid = true;
    return 0;
}

static int frame_lr_run(Dav1dHipFrame *f, const Dav1dHipPicture *out, const Dav1dHipPicture *in, const Dav1dHipPicture *lpf) {
    Dav1dHipContext *c = f->c;
    FrameTrace tr;
    int rc = frame_lr_plan(f);
    if (rc) return rc;
    Dav1dHipFrame::LrPlan &pl = f->lrplan;
    const size_t n = pl.sorted.size();
    if (!n) return 0;
    if (out->bpc != in->bpc || lpf->bpc != in->bpc) return -EINVAL;
    const bool banded = pl.banded && f->progress_cb;
    const int nb = pl.nb;
    if (banded && !c->band_cnt) {
        if (hipMalloc((void **) &c->band_cnt, 128 * sizeof(uint32_t)) != hipSuccess) { c->band_cnt = nullptr; return -ENOMEM; }
        if (hipHostMalloc((void **) &c->band_flags, 64 * sizeof(uint32_t), 0) != hipSuccess) { c->band_flags = nullptr; return -ENOMEM; }
        memset(c->band_flags, 0, 64 * sizeof(uint32_t));
    }
    const size_t o_waves = pl.o_waves;
    f->deferred.bufs.emplace_back(new TaskBuf(c, pl.bytes + 16));
    uint8_t *const devb = f->deferred.bufs.back()->p;
    if (!devb) return -ENOMEM;
    Dav1dHipLrTask *const dev = reinterpret_cast<Dav1dHipLrTask *>(devb);
    tr.mark(0);
    rc = hip_rc(hipMemcpyAsync(devb, pl.slab, pl.bytes, hipMemcpyHostToDevice, c->stream));
    const DevPlanes dp = dev_planes(out), sp = dev_planes(in), lp = dev_planes(lpf);
    if (!banded) {
        // enqueued behind the stages before, not waited for: frame_run waits once, at its end
        if (!rc) rc = dav1d_hip_launch_wiener(&dp, &sp, &lp, out->bpc, dev, (int) pl.nw, pl.max_w, c->stream);
        if (!rc) rc = dav1d_hip_launch_sgr(&dp, &sp, &lp, out->bpc, dev + pl.nw, devb + o_waves, (int) pl.n_waves, c->stream);
        return rc;
    }
    if (!rc) rc = hip_rc(hipMemsetAsync(c->band_cnt, 0, 64 * sizeof(uint32_t), c->stream));
    const uint32_t seq = ++c->band_seq ? c->band_seq : ++c->band_seq;        // never 0: the flags start at 0
    const BandSignal sig = { c->band_cnt, reinterpret_cast<const uint32_t *>(devb + o_waves + pl.n_waves * 16), c->band_flags, seq };
    // "everything before restoration is through" (reconstruction, deblocking, CDEF, the copy of the units that are not listed): what a
    // band WITHOUT restoration tasks has to wait for before its rows count as final
    if (!rc) rc = hip_rc(hipEventRecord(c->ev_fork, c->stream));
    if (!rc) rc = dav1d_hip_launch_wiener_sig(&dp, &sp, &lp, out->bpc, dev, (int) pl.nw, pl.max_w, &sig, c->stream);
    if (!rc) rc = dav1d_hip_launch_sgr_sig(&dp, &sp, &lp, out->bpc, dev + pl.nw, devb + o_waves, (int) pl.n_waves, &sig, c->stream);
    tr.mark(1);
    // the bands come through (roughly) in order; the last one is published with the frame (frame_run).  A band without tasks has
    // nothing to wait for beyond the bands before it.
    if (!rc) {
        volatile uint32_t *const flags = c->band_flags;
        const std::vector<uint32_t> &target = pl.target;
        bool all_done = false, waited = false;
        for (int b = 0; b + 1 < nb; b++) {
            for (unsigned spin = 0; target[b] && flags[b] != seq && !all_done; spin++) {
                if ((spin & 4095) == 4095) all_done = hipStreamQuery(c->stream) == hipSuccess;      // (also the way out should a launch have failed)
                else __builtin_ia32_pause();
            }
            if (all_done && flags[b] != seq && target[b]) break;
            // a band's flag says "the restoration workgroups of this band have written their pixels", which orders it behind the
            // earlier stages only for bands that HAVE such workgroups: the leading bands without any wait for the stages themselves
            if (target[b]) waited = true;
            else if (!waited) { if (hipEventSynchronize(c->ev_fork) != hipSuccess) break; waited = true; }
            std::atomic_thread_fence(std::memory_order_acquire);
            f->publish(pl.first_y[b + 1], out);
        }
    }
    tr.mark(2);
    (void) hipStreamSynchronize(c->stream);
    tr.mark(3);
    if (tr.on) fprintf(stderr, "frame_lr_run (banded): upload buffers %.3f  uploads + launches %.3f  bands published %.3f  tail sync %.3f ms (%zu tasks, %d bands)\n",
                       tr.ms[0], tr.ms[1], tr.ms[2], tr.ms[3], n, nb);
    return rc;
}

// Runs the frame.  coef / prep / mask: the DEVICE arenas the task offsets refer to.  On return `cur` holds the
// reconstructed AND deblocked picture (deblocking is in place, as in the reference); *filtered receives a descriptor of
// the picture after CDEF and loop restoration (it is `cur` itself when neither stage has tasks; otherwise a picture owned by
// the frame, valid until dav1d_hip_frame_destroy); when film grain parameters were set and `grain_out` is given, the grain
// is applied from *filtered into grain_out (dav1d_apply_grain, src/lib.c:311-329).  Synchronous: every stage has
// completed on return, so the caller can publish progress the way src/thread_task.c:888-896 does.
static int frame_run(Dav1dHipFrame *f, void *coef, int16_t *prep, uint8_t *mask, Dav1dHipPicture *filtered, const Dav1dHipPicture *grain_out);
static int frame_flush_locked(Dav1dHipFrame *f);

int dav1d_hip_frame_end(Dav1dHipFrame *f, void *coef, int16_t *prep, uint8_t *mask, Dav1dHipPicture *filtered, const Dav1dHipPicture *grain_out) {
    if (!f) return -EINVAL;
    const int rc_prep = frame_wait_prep(f);          // (before the frame's lock: a preparation takes it to hand its chunk in)
    std::lock_guard<std::mutex> lk(f->mtx);
    Dav1dHipContext *c = f->c;
    // the multi-stream sections below use the context's side streams and events: one frame (or list run) at a time per context
    std::lock_guard<std::mutex> run_lk(c->run_mtx);
    (void) hipSetDevice(c->device);
    // several devices in the process: a reference that lives on another one has to be made resident here first (dav1d_hip_picture_copy_peer)
    const int rc_run = rc_prep ? rc_prep : pictures_on_device(c, f->refs, f->n_refs) ? -EXDEV : frame_run(f, coef, prep, mask, filtered, grain_out);
    // the frame has synchronised (or failed): the pinned chunk blobs go back to the context's pool
    (void) hipStreamSynchronize(c->copy_stream);
    for (Dav1dHipChunk *ck : f->chunks) { ck->release(c); delete ck; }
    f->chunks.clear();
    for (Dav1dHipFrame::StepChunk *sc : f->step_chunks) delete sc;
    f->step_chunks.clear();
    if (f->huarena) { dav1d_hip_slab_put(c, f->huarena, f->huarena_cap); f->huarena = nullptr; f->huarena_cap = 0; }
    f->uarena_used = 0;
    f->n_steps = 0;
    f->arena_used = 0;
    f->harena_flushed = 0;
    c->carena_hint = std::max(c->carena_hint, f->carena_used.load());
    f->carena_used = 0;
    f->hcarena_flushed = 0;
    f->saw_dense = 0; f->saw_packed = 0;
    for (Dav1dHipFrame::LateCoefs &lc : f->late_coefs) free(lc.copy);
    f->late_coefs.clear();
    if (c->pending_slab) {
        dav1d_hip_slab_put(c, c->pending_slab, c->pending_slab_cap);
        c->pending_slab = nullptr;
    }
    return rc_run;
}

} // extern "C"

// DAV1D_HIP_TRACE_FRAME=1: where a frame's time on the ending thread goes (host wall clock incl. the waits it makes), one line per frame

static int frame_run(Dav1dHipFrame *f, void *coef, int16_t *prep, uint8_t *mask, Dav1dHipPicture *filtered, const Dav1dHipPicture *grain_out) {
    Dav1dHipContext *c = f->c;
    int rc = 0;
    FrameTrace tr;
    // whatever the post filters enqueued without waiting (Dav1dHipFrame::deferred) is through before its buffers go back, on every way out
    struct DeferredGuard {
        Dav1dHipFrame *f;
        ~DeferredGuard() {
            if (f->deferred.slabs.empty() && f->deferred.bufs.empty()) return;
            (void) hipStreamSynchronize(f->c->stream);
            f->deferred.release(f->c);
        }
    } deferred_guard{ f };
    // the raster planes of every picture this frame writes change below: whatever tiled twin a recycled picture still carries is
    // stale from here on (*filtered is a copy of one of these descriptors, so it reports twin_ok = 0 unless the frame retiles it)
    f->cur.twin_ok = 0;
    for (int k = 0; k < 2; k++) f->tmp[k].twin_ok = 0;
    for (int k = 0; k < 3; k++) f->sr[k].twin_ok = 0;
    if (f->saw_packed.load()) {
        // a packed frame: every residual task points into the frame's own coefficient arena (a dense arena next to it is not
        // supported: the kernels take one base).  What the flushes have not sent goes now, the late segments behind it.
        if (coef || f->saw_dense.load() || !f->carena) return -EINVAL;
        rc = frame_flush_locked(f);
        for (const Dav1dHipFrame::LateCoefs &lc : f->late_coefs)
            if (!rc) rc = hip_rc(hipMemcpyAsync(f->carena + lc.off, lc.copy, lc.bytes, hipMemcpyHostToDevice, c->copy_stream));
        if (!rc) rc = hip_rc(hipEventRecord(c->ev_copy, c->copy_stream));
        if (!rc) rc = hip_rc(hipStreamWaitEvent(c->stream, c->ev_copy, 0));
        if (rc) return rc;
        coef = f->carena;
    }
    tr.mark(0);
    if (c->post_bands >= 2) { rc = frame_merge_filter_pieces(f); if (rc) return rc; }          // the banded route works on the merged lists
    // warped / scaled predictions first: the compound combinations of the list below read their PREP outputs
    if (!f->warp.empty()) rc = dav1d_hip_warp_batch(c, &f->cur, f->refs, f->n_refs, f->warp.data(), f->warp.size(), prep);
    if (!rc && !f->scaled.empty()) rc = dav1d_hip_mc_scaled_batch(c, &f->cur, f->refs, f->n_refs, f->scaled.data(), f->scaled.size(), prep);
    if (rc) return rc;
    tr.mark(1);
    // Option ref_twin = 3: a frame that is reconstruction and nothing else (no intra wavefront, no in-loop filters, no super-resolution,
    // no warped / scaled predictions — they write raster planes) leaves its picture in the tiled twin ONLY (dav1d_hip_recon_list_run_tiled);
    // film grain and the fetch to the host un-tile what they read.  Every other frame runs on the raster planes and is retiled at the end.
    const bool has_post = !f->filter_pieces.empty() || !f->lf.empty() || !f->cdef.empty() || !f->lr.empty() || (f->cdef_rows_state == 1 && f->cdef_row_units.load() > 0);
    const bool tiled_native = c->ref_twin >= 3 && f->warp.empty() && f->scaled.empty() && f->step_chunks.empty() && f->step_copy.empty() && !has_post && !f->sr_w;
    bool wrote_twin = false;
    if (!f->chunks.empty()) {
        // predictions and residuals as one pipelined list (the residual launch of a transform size waits only for the
        // prediction launches under its blocks), assembled from the chunks the submitting threads prepared: one upload per
        // chunk, one gather launch, then the frame's launches
        Dav1dHipReconList rl;
        Dav1dHipInterList il;
        Dav1dHipMcList ml;
        Dav1dHipCompList cl;
        Dav1dHipItxList xl;
        // a chunk that found no room in the twin (the first frames of a size: the arena is sized by what frames have needed) sits in
        // a slab of its own: the arena grows to what was drawn, the twin goes up again if it did, the late chunks follow
        bool regrown = false;
        FrameTrace t2;
        rc = dav1d_hip_chunks_grow_arena(c, &f->arena, &f->arena_cap, f->arena_used.load(), &regrown);
        if (regrown) f->harena_flushed = 0;
        if (!rc && f->harena) rc = frame_flush_locked(f);       // what dav1d_hip_frame_flush has not sent yet (the gather launch waits for the copy stream)
        t2.mark(0);
        if (!rc) rc = dav1d_hip_chunks_send_late(c, f->chunks, f->arena, f->arena_cap, regrown);      // after the twin: see chunk.h
        if (!rc) rc = dav1d_hip_chunks_to_recon_list(c, f->chunks, &f->arena, &f->arena_cap, f->refs, f->n_refs, &rl, &il, &ml, &cl, &xl);
        t2.mark(1);
        struct T2 { FrameTrace &t; ~T2() { t.mark(2); if (t.on) fprintf(stderr, "recon enqueue: arena + flush %.3f  chunks -> lists (gather) %.3f  launches %.3f ms\n", t.ms[0], t.ms[1], t.ms[2]); } } t2_end{ t2 };
        if (!rc) {
            if (ml.n || cl.n || rl.f_n[0] || rl.f_n[1] || rl.f_n[2] || rl.f_n[3] || rl.f_n[4]) {
                if (!f->n_refs) rc = -EINVAL;
                else if (tiled_native) {
                    // the frame's blocks cover the picture: whatever the recycled twin holds is overwritten, nothing has to be carried along
                    f->cur.twin_ok = DAV1D_HIP_TWIN_ONLY;
                    rc = dav1d_hip_recon_list_run_tiled(c, &rl, &f->cur, f->refs, f->n_refs, prep, mask, coef);
                    wrote_twin = !rc && f->cur.twin_ok != 0;
                }
                else rc = dav1d_hip_recon_list_run(c, &rl, &f->cur, f->refs, f->n_refs, prep, mask, coef);
            } else if (xl.n) {
                rc = dav1d_hip_itx_list_run(c, &xl, &f->cur, coef);
            }
        }
    }
    tr.mark(2);
    // intra blocks, wavefront step by step (each step: a paired launch for the small blocks, a prediction and a residual
    // launch for the others), enqueued back to back.  A frame may hold step copies (intra block copies) without any intra-step
    // submission (direct callers of the C API): they run all the same.
    if (!rc && (!f->step_chunks.empty() || !f->step_copy.empty())) {
        static const bool trace = getenv("DAV1D_HIP_TRACE_INTRA") != nullptr;
        const auto t_a = std::chrono::steady_clock::now();
        const size_t ns = f->n_steps;
        bool flow = c->flow_min_steps > 0 && ns >= (size_t) c->flow_min_steps;
        bool any_blend = false;
        for (const Dav1dHipFrame::StepChunk *ck : f->step_chunks) { flow = flow && ck->flow_ok && !ck->has_blend; any_blend = any_blend || ck->has_blend; }
        flow = flow && f->step_copy.empty();          // intra block copies are launches of their own between the steps
        // superblock by superblock (intra_sb.hip) when every submission was sorted for it (the frame's tiling is known, no intra block
        // copies; inter-intra blends are part of their units): option intra_sb = 2 for every frame, 1 for the long wavefronts only
        bool sbw = f->have_tiling && f->step_copy.empty() && !f->step_chunks.empty() && (c->intra_sb >= 2 || (c->intra_sb == 1 && flow));
        bool any_sorted = false;
        for (const Dav1dHipFrame::StepChunk *ck : f->step_chunks) { sbw = sbw && ck->flow_ok && ck->sb_sorted; any_sorted = any_sorted || ck->sb_sorted; }
        if (any_sorted) flow = false;                  // the dataflow launch wants the units in step order
        if (sbw) {
            // the chunks' units: those in the pinned arena form its prefix [0, in_arena) in the order they were drawn; the others (the
            // arena is sized by what earlier frames needed) follow, copied into a slab that then holds everything
            const size_t nck = f->step_chunks.size();
            std::vector<const std::vector<SbPart> *> parts(nck);
            std::vector<size_t> base(nck);
            size_t in_arena = 0, late = 0;
            bool needs_aux = false;
            for (size_t k = 0; k < nck; k++) {
                const Dav1dHipFrame::StepChunk *ck = f->step_chunks[k];
                parts[k] = &ck->parts;
                needs_aux = needs_aux || ck->has_pal;
                if (ck->in_arena) { base[k] = (size_t) (ck->sorted - reinterpret_cast<IntraUnit *>(f->huarena)); in_arena = std::max(in_arena, base[k] + ck->n_sorted); }
                else late += ck->n_sorted;
            }
            const size_t total = in_arena + late;
            c->uarena_hint = std::max(c->uarena_hint, total * sizeof(IntraUnit));
            size_t slab_cap = 0;
            uint8_t *slab = nullptr;
            const uint8_t *host = f->huarena;
            if (late) {
                slab = dav1d_hip_slab_get(c, total * sizeof(IntraUnit), &slab_cap);
                if (!slab) rc = -ENOMEM;
                else {
                    if (in_arena) memcpy(slab, f->huarena, in_arena * sizeof(IntraUnit));
                    size_t at = in_arena;
                    for (size_t k = 0; k < nck; k++) {
                        const Dav1dHipFrame::StepChunk *ck = f->step_chunks[k];
                        if (ck->in_arena) continue;
                        memcpy(slab + at * sizeof(IntraUnit), ck->sorted, ck->n_sorted * sizeof(IntraUnit));
                        base[k] = at; at += ck->n_sorted;
                    }
                    host = slab;
                }
            }
            SbPlan plan;
            std::vector<uint64_t> copy_deps;          // intra block copies: superblock -> the superblocks under its source windows
            for (const Dav1dHipFrame::StepChunk *ck : f->step_chunks) copy_deps.insert(copy_deps.end(), ck->copy_deps.begin(), ck->copy_deps.end());
            std::sort(copy_deps.begin(), copy_deps.end());
            copy_deps.erase(std::unique(copy_deps.begin(), copy_deps.end()), copy_deps.end());
            if (!rc) rc = dav1d_hip_sbw_plan(f->tiling, parts, base, f->sb_dep.empty() ? nullptr : f->sb_dep.data(), plan, copy_deps.empty() ? nullptr : &copy_deps);
            if (!rc && needs_aux && !f->aux) rc = -EINVAL;
            if (!rc && any_blend && !mask) rc = -EINVAL;
            const size_t ub = total * sizeof(IntraUnit), rb = plan.regions.size() * sizeof(SbRegion);
            if (!rc && total) {
                const int lds = c->intra_sb_lds && !any_blend && copy_deps.empty();          // (the LDS-resident form neither blends nor copies)
                const bool one_launch = c->intra_sb_flow && !lds && plan.level_start.size() > 2;
                const size_t fb = (plan.regions.size() + 1) * sizeof(uint32_t), o_flags = (ub + rb + 255) & ~(size_t) 255;
                const size_t o_where = (o_flags + fb + 255) & ~(size_t) 255, wb = one_launch ? plan.where.size() * sizeof(uint32_t) : 0;
                const size_t o_done = (o_where + wb + 255) & ~(size_t) 255, db = one_launch ? total : 0;       // a byte per record: reconstructed by the one launch
                TaskBuf dev_buf(c, o_done + db + 256);
                uint8_t *const dev = dev_buf.p;
                if (!dev) rc = -ENOMEM;
                if (!rc) rc = hip_rc(hipMemcpyAsync(dev, host, ub, hipMemcpyHostToDevice, c->stream));
                if (!rc && one_launch) rc = hip_rc(hipMemsetAsync(dev + o_flags, 0, fb, c->stream));
                if (!rc && one_launch) rc = hip_rc(hipMemsetAsync(dev + o_done, 0, db, c->stream));
                if (!rc) rc = dav1d_hip_upload(c, dev + ub, plan.regions.data(), rb);
                if (!rc && wb) rc = dav1d_hip_upload(c, dev + o_where, plan.where.data(), wb);
                const auto t_b = std::chrono::steady_clock::now();
                const DevPlanes dp = dev_planes(&f->cur);
                if (one_launch) {
                    // every level in one launch: superblocks wait for the flags of the neighbours they read (intra_sb.hip).  Workgroups of four
                    // waves where the frame has superblocks to fill the CUs with (>= 128 per level on average: many tiles, or isolated intra
                    // blocks in an inter frame): two superblocks per CU are in flight instead of one (256 registers per lane) — 8K key frame in
                    // 15 x 7 tiles 9.7 -> 7.5 ms.  Eight waves where the wavefront is narrow (4 tile columns: 25 superblocks per level; the
                    // CUs are not all busy and a superblock's own time counts: 30.8 fps against 28.2 on the 8K stream of bench.py) and for
                    // intra block copies (wide steps of whole-block copies: 12.1 ms against 13.1).  profiles/r05/intra_sb_waves_ab.jsonl
                    const bool wide_levels = plan.regions.size() >= 128 * (plan.level_start.size() - 1);
                    // ONE wave per superblock (its units one after the other, no barrier, eight superblocks per CU) is there for frames whose
                    // superblocks hold intra_sb_one_below units or fewer on average; off by default: measured no faster (the inter frame with 10 % intra
                    // blocks of bench.py's full table 0.435 -> 0.507 ms with it forced: a superblock is as slow as its units in a row, 8 - 10 us each,
                    // and the launch as slow as its fullest superblock; profiles/r06/intra_one_wave_ab.txt)
                    const bool few_units = c->intra_sb_one_below > 0 && total - plan.regions.size() <= (size_t) c->intra_sb_one_below * plan.regions.size();
                    const int sb_waves = c->intra_sb_waves ? c->intra_sb_waves : copy_deps.empty() && few_units ? 1 : copy_deps.empty() && wide_levels ? 4 : 8;
                    if (!rc) rc = dav1d_hip_launch_intra_sb(&dp, f->cur.bpc, f->cur.layout, reinterpret_cast<const IntraUnit *>(dev),
                                                            reinterpret_cast<const SbRegion *>(dev + ub), (int) plan.regions.size(), f->aux, mask, coef,
                                                            sb_waves, f->tiling.sb_log2, 0, reinterpret_cast<uint32_t *>(dev + o_flags), c->stream,
                                                            reinterpret_cast<const uint32_t *>(dev + o_where), f->tiling.sbw, dev + o_done);
                    uint32_t gave_up = 0;
                    if (!rc) rc = dav1d_hip_download(c, &gave_up, dev + o_flags + plan.regions.size() * sizeof(uint32_t), sizeof(gave_up));
                    if (!rc && gave_up) {
                        // Workgroups gave up waiting for a neighbour (never seen: the form rests on workgroups being dispatched in the order of
                        // their index and staying resident; a driver that preempts or reorders them would break it).  Nothing is lost: every unit
                        // the launch reconstructed is marked (`done`, a byte per record), and the launches per level below — no flags, no
                        // residency assumption — run what is left, level by level.
                        c->intra_sb_fallbacks++;
                        for (size_t l = 0; l + 1 < plan.level_start.size() && !rc; l++)
                            rc = dav1d_hip_launch_intra_sb(&dp, f->cur.bpc, f->cur.layout, reinterpret_cast<const IntraUnit *>(dev),
                                                           reinterpret_cast<const SbRegion *>(dev + ub) + plan.level_start[l],
                                                           (int) (plan.level_start[l + 1] - plan.level_start[l]), f->aux, mask, coef, sb_waves, f->tiling.sb_log2,
                                                           0, nullptr, c->stream, nullptr, 0, dev + o_done);
                    }
                } else
                for (size_t l = 0; l + 1 < plan.level_start.size() && !rc; l++)
                    rc = dav1d_hip_launch_intra_sb(&dp, f->cur.bpc, f->cur.layout, reinterpret_cast<const IntraUnit *>(dev),
                                                   reinterpret_cast<const SbRegion *>(dev + ub) + plan.level_start[l],
                                                   (int) (plan.level_start[l + 1] - plan.level_start[l]), f->aux, mask, coef, c->intra_sb_waves, f->tiling.sb_log2,
                                                   lds, nullptr, c->stream);
                // the device copy of the units goes back to the pool when this scope ends: the launches have to be through by then
                const int rs = hip_rc(hipStreamSynchronize(c->stream));
                if (!rc) rc = rs;
                if (trace) {
                    const auto t_d = std::chrono::steady_clock::now();
                    fprintf(stderr, "intra sb: %zu steps, %zu units (%zu late) in %zu superblocks, %zu levels; plan + upload %.2f ms, launches to finish %.2f ms\n", ns,
                            total, late, plan.regions.size(), plan.level_start.size() - 1, std::chrono::duration<double, std::milli>(t_b - t_a).count(),
                            std::chrono::duration<double, std::milli>(t_d - t_b).count());
                }
            }
            if (slab) dav1d_hip_slab_put(c, slab, slab_cap);
        } else if (flow) {
            // A long wavefront (a key frame: hundreds to thousands of steps, most of them narrow) goes down as ONE launch whose
            // waves hand the steps to each other (intra_flow.hip).  The chunks' units, merged step by step: first the units with
            // a prediction of every chunk, then the residuals on their own; `need` = where the group starts.
            std::vector<size_t> na(ns + 1, 0), nb(ns + 1, 0);
            for (const Dav1dHipFrame::StepChunk *ck : f->step_chunks)
                for (size_t s = 0; s < ck->ua_end.size(); s++) {
                    const uint32_t b0 = s ? ck->ub_end[s - 1] : 0;
                    na[s] += ck->ua_end[s] - b0; nb[s] += ck->ub_end[s] - ck->ua_end[s];
                }
            std::vector<size_t> oa(ns + 1, 0), ob(ns + 1, 0);
            size_t total = 0;
            for (size_t s = 0; s < ns; s++) { oa[s] = total; ob[s] = total + na[s]; total += na[s] + nb[s]; }
            // where every chunk's share of every step goes (a running position per step, chunk after chunk), then the copies —
            // 50 MB for an 8K key frame — on a few threads into pinned memory: the upload is one DMA from there
            const size_t nck = f->step_chunks.size();
            std::vector<uint32_t> da(nck * ns, 0), db(nck * ns, 0);
            {
                std::vector<size_t> pa(oa), pb(ob);
                for (size_t k = 0; k < nck; k++) {
                    const Dav1dHipFrame::StepChunk *ck = f->step_chunks[k];
                    for (size_t s = 0; s < ck->ua_end.size(); s++) {
                        const uint32_t b0 = s ? ck->ub_end[s - 1] : 0, a1 = ck->ua_end[s], b1 = ck->ub_end[s];
                        da[k * ns + s] = (uint32_t) pa[s]; pa[s] += a1 - b0;
                        db[k * ns + s] = (uint32_t) pb[s]; pb[s] += b1 - a1;
                    }
                }
            }
            size_t slab_cap = 0;
            IntraUnit *const all = total ? reinterpret_cast<IntraUnit *>(dav1d_hip_slab_get(c, total * sizeof(IntraUnit), &slab_cap)) : nullptr;
            if (total && !all) return -ENOMEM;
            auto copy_chunks = [&](size_t k0, size_t k1) {
                for (size_t k = k0; k < k1; k++) {
                    const Dav1dHipFrame::StepChunk *ck = f->step_chunks[k];
                    for (size_t s = 0; s < ck->ua_end.size(); s++) {
                        const uint32_t b0 = s ? ck->ub_end[s - 1] : 0, a1 = ck->ua_end[s], b1 = ck->ub_end[s];
                        IntraUnit *pa = all + da[k * ns + s], *pb = all + db[k * ns + s];
                        if (a1 > b0) { memcpy(pa, &ck->units[b0], (a1 - b0) * sizeof(IntraUnit)); for (uint32_t i = 0; i < a1 - b0; i++) pa[i].need = (uint32_t) oa[s]; }
                        if (b1 > a1) { memcpy(pb, &ck->units[a1], (b1 - a1) * sizeof(IntraUnit)); for (uint32_t i = 0; i < b1 - a1; i++) pb[i].need = (uint32_t) ob[s]; }
                    }
                }
            };
            {
                const unsigned hw = std::thread::hardware_concurrency();
                const size_t nt = std::max<size_t>(1, std::min<size_t>({ (size_t) 8, hw ? hw : 1, nck / 16 + 1 }));
                std::vector<std::thread> th;
                for (size_t t = 1; t < nt; t++) th.emplace_back(copy_chunks, nck * t / nt, nck * (t + 1) / nt);
                copy_chunks(0, nck / nt);
                for (std::thread &x : th) x.join();
            }
            Dav1dHipIntraFlow *fl = nullptr;
            rc = dav1d_hip_intra_flow_from_units(c, &fl, all, total);
            if (all) dav1d_hip_slab_put(c, reinterpret_cast<uint8_t *>(all), slab_cap);         // the upload has synchronised
            const auto t_b = std::chrono::steady_clock::now();
            if (!rc) rc = dav1d_hip_intra_flow_run(c, fl, &f->cur, coef, f->aux);
            uint32_t st[3] = { 0, 0, 0 };
            if (!rc) rc = dav1d_hip_intra_flow_status(c, fl, st);
            if (!rc && (st[2] || st[1] != total)) rc = -EIO;      // a wave gave up waiting: never in a sound run
            if (trace) {
                const auto t_d = std::chrono::steady_clock::now();
                fprintf(stderr, "intra flow: %zu steps, %zu units; merge + upload %.2f ms, launch to finish %.2f ms\n", ns, total,
                        std::chrono::duration<double, std::milli>(t_b - t_a).count(), std::chrono::duration<double, std::milli>(t_d - t_b).count());
            }
            if (fl) dav1d_hip_intra_flow_destroy(c, fl);
        } else {
            // A short wavefront (the intra blocks of an inter frame: tens of steps, the first ones wide) runs as up to three
            // launches per step (a paired launch for the small blocks, a prediction and a residual launch for the others),
            // which keep more waves in flight per step; so does any frame with inter-intra blends.  c->flow_min_steps = the
            // border ($DAV1D_HIP_FLOW_MIN_STEPS at open, default 200; 0 = always the launches).
            std::vector<Dav1dHipIpredTask> allp;
            std::vector<Dav1dHipItxTask> allt;
            std::vector<Dav1dHipCompTask> allb;
            std::vector<size_t> ps(ns, 0), ts(ns, 0), bs(ns, 0);
            for (size_t s = 0; s < ns; s++)
                for (const Dav1dHipFrame::StepChunk *ck : f->step_chunks) {
                    if (s >= ck->ip_end.size()) continue;
                    const uint32_t p0 = s ? ck->ip_end[s - 1] : 0, x0 = s ? ck->ix_end[s - 1] : 0, b0 = s ? ck->bl_end[s - 1] : 0;
                    allp.insert(allp.end(), ck->ip.begin() + p0, ck->ip.begin() + ck->ip_end[s]);
                    allt.insert(allt.end(), ck->ix.begin() + x0, ck->ix.begin() + ck->ix_end[s]);
                    allb.insert(allb.end(), ck->bl.begin() + b0, ck->bl.begin() + ck->bl_end[s]);
                    ps[s] += ck->ip_end[s] - p0; ts[s] += ck->ix_end[s] - x0; bs[s] += ck->bl_end[s] - b0;
                }
            // intra block copies by step: the frame's own picture as the one reference, its size rounded up to whole 8x8 blocks
            // (what mc() bounds an intrabc source by: f->bw * 4 >> ss_hor, src/recon_tmpl.c:960-966)
            std::vector<Dav1dHipMcTask> cp(f->step_copy.size());
            std::vector<size_t> cp_end(ns + 1, 0);
            if (!cp.empty()) {
                for (uint16_t s : f->step_copy_step) cp_end[(size_t) s + 1 <= ns ? s + 1 : ns]++;
                for (size_t s = 0; s < ns; s++) cp_end[s + 1] += cp_end[s];
                std::vector<size_t> pos(cp_end.begin(), cp_end.end() - 1);
                for (size_t i = 0; i < cp.size(); i++) cp[pos[f->step_copy_step[i]]++] = f->step_copy[i];
            }
            Dav1dHipPicture self = f->cur;
            {
                const int ssh = f->cur.layout != DAV1D_HIP_LAYOUT_I444, ssv = f->cur.layout == DAV1D_HIP_LAYOUT_I420;
                const int w8 = (f->cur.p[0].w + 7) & ~7, h8 = (f->cur.p[0].h + 7) & ~7;
                self.p[0].w = w8; self.p[0].h = h8;
                for (int p = 1; p < 3; p++) if (self.p[p].data) { self.p[p].w = w8 >> ssh; self.p[p].h = h8 >> ssv; }
            }
            Dav1dHipIntraList *xl = nullptr;
            rc = dav1d_hip_intra_list_create_blend(c, &xl, allp.data(), ps.data(), allt.data(), ts.data(), allb.data(), bs.data(), ps.size());
            for (size_t k = 0; k < ps.size() && !rc; k++) {
                if (!cp.empty() && cp_end[k + 1] > cp_end[k])
                    rc = dav1d_hip_mc_batch(c, &f->cur, &self, 1, cp.data() + cp_end[k], cp_end[k + 1] - cp_end[k], prep);
                if (!rc) rc = dav1d_hip_intra_list_run_batch_blend(c, xl, k, &f->cur, coef, f->aux, prep, mask);
            }
            if (xl) dav1d_hip_intra_list_destroy(c, xl);
        }
    }
    tr.mark(3);
    const Dav1dHipPicture *last = &f->cur;
    int piped = 1;
    f->post_bands = 0;
    if (!rc) piped = post_filters_pipelined(f, &last);
    if (piped < 0) return piped;
    if (piped == 1) {        // stage by stage
        last = &f->cur;
        if (!rc && (!f->filter_pieces.empty() || (f->cdef_rows_state == 1 && f->cdef_row_units.load() > 0))) {
            bool did_cdef = false;
            rc = frame_filters_from_pieces(f, &did_cdef);
            if (did_cdef) last = &f->tmp[0];
        }
        if (!rc && !f->lf.empty()) {
            if (!f->lvl || !f->have_lut) return -EINVAL;
            rc = dav1d_hip_lf_batch(c, &f->cur, f->lf.data(), f->lf.size(), f->lvl, f->b4_stride, f->lut_e, f->lut_i);
        }
        if (!rc && !f->cdef.empty()) {
            rc = frame_tmp(f, 0);
            if (!rc) rc = copy_picture(c, &f->tmp[0], &f->cur);       // units that are not listed keep their pixels
            if (!rc) rc = dav1d_hip_cdef_batch(c, &f->tmp[0], &f->cur, f->cdef.data(), f->cdef.size(), f->cdef_damping, nullptr);
            last = &f->tmp[0];
        }
        const Dav1dHipPicture *lpf = &f->cur;        // the deblocked picture: what restoration reads across its stripe borders
        if (!rc && f->sr_w) {
            // Super-resolution (dav1d_filter_sbrow_resize, src/recon_tmpl.c:2053-2086; backup_lpf's resize, src/lf_apply_tmpl.c:
            // 40-100): every row of the CDEF output — and of the deblocked picture, for the rows restoration reads at its stripe
            // borders — upscaled horizontally.  Step and first position: AV1 spec 7.16 (dav1d_submit_frame, src/decode.c:3531-3539);
            // the source width is the coded width rounded up to whole 8x8 blocks (4 * f->bw), whose columns the frame did write.
            const int ssh = f->cur.layout != DAV1D_HIP_LAYOUT_I444;
            const int in_w[2] = { f->cur.p[0].w, (f->cur.p[0].w + ssh) >> ssh }, out_w[2] = { f->sr_w, (f->sr_w + ssh) >> ssh };
            const int src_w[2] = { (f->cur.p[0].w + 7) & ~7, (((f->cur.p[0].w + 7) & ~7) + ssh) >> ssh };
            int step[2], start[2];
            for (int i = 0; i < 2; i++) {
                step[i] = ((in_w[i] << 14) + (out_w[i] >> 1)) / out_w[i];
                const int err = out_w[i] * step[i] - (in_w[i] << 14);
                start[i] = ((-((out_w[i] - in_w[i]) << 13) + (out_w[i] >> 1)) / out_w[i] + 128 - err / 2) & 0x3fff;
            }
            const bool want_lpf = !f->lr.empty();
            for (int k = 0; k < 2 && !rc; k++) {
                if (k == 1 && !want_lpf) break;
                if (!f->have_sr[k]) {
                    rc = dav1d_hip_picture_alloc(c, &f->sr[k], f->sr_w, f->cur.p[0].h, f->cur.layout, f->cur.bpc);
                    if (!rc) f->have_sr[k] = true;
                }
                const Dav1dHipPicture *from = k ? &f->cur : last;
                for (int pl = 0; pl < 3 && !rc; pl++)
                    if (f->cur.p[pl].data)
                        rc = dav1d_hip_resize(c, &f->sr[k], from, pl, out_w[!!pl], 0, f->cur.p[pl].h, src_w[!!pl], step[!!pl], start[!!pl]);
            }
            last = &f->sr[0];
            lpf = &f->sr[1];
        }
        if (!rc && !f->lr.empty()) {
            Dav1dHipPicture *out = &f->tmp[1];
            if (f->sr_w) {
                if (!f->have_sr[2]) {
                    rc = dav1d_hip_picture_alloc(c, &f->sr[2], f->sr_w, f->cur.p[0].h, f->cur.layout, f->cur.bpc);
                    if (!rc) f->have_sr[2] = true;
                }
                out = &f->sr[2];
            } else {
                rc = frame_tmp(f, 1);
            }
            // restoration units of type NONE are not listed and keep their pixels: a plane whose listed stripes cover it needs nothing
            // underneath, the others are copied plane by plane
            if (!rc) rc = copy_unrestored_planes(c, out, last, f->lr);
            // somebody listens for rows (dav1d_hip_frame_set_progress_callback): the LAST stage runs in bands of rows, each followed by
            // an event, and the rows are published band by band while the later bands still run
            if (!rc) rc = frame_lr_run(f, out, last, lpf);
            last = out;
        }
    }
    // the picture later frames predict from gets its tiled twin (Dav1dHipPicture.twin: what their motion compensation reads)
    if (!rc && c->ref_twin >= 2 && last->twin[0] && !(wrote_twin && last == &f->cur)) rc = dav1d_hip_picture_retile(c, const_cast<Dav1dHipPicture *>(last));
    // film grain reads raster rows: a picture that lives in its twin gets them back first (both valid afterwards)
    if (!rc && f->have_grain && grain_out && last->twin_ok == DAV1D_HIP_TWIN_ONLY) rc = dav1d_hip_picture_untile(c, const_cast<Dav1dHipPicture *>(last));
    if (!rc && filtered) *filtered = *last;
    if (!rc && f->have_grain && grain_out)
        rc = f->prepared ? dav1d_hip_fg_apply_prepared(c, grain_out, last, f->prepared, f->is_id)
                         : dav1d_hip_fg_apply(c, grain_out, last, &f->grain, f->is_id);
    tr.mark(4);
    if (!rc) rc = dav1d_hip_sync(c);
    else (void) dav1d_hip_sync(c);
    tr.mark(5);
    if (!rc) f->publish(f->cur.p[0].h, last);
    if (tr.on) fprintf(stderr, "frame_run %dx%d: coefs/flush %.2f  warp+scaled %.2f  chunks->lists+recon %.2f  intra %.2f  post filters %.2f  final sync %.2f ms  (chunks %zu, step chunks %zu, warp %zu, scaled %zu)\n",
                       f->cur.p[0].w, f->cur.p[0].h, tr.ms[0], tr.ms[1], tr.ms[2], tr.ms[3], tr.ms[4], tr.ms[5], f->chunks.size(), f->step_chunks.size(), f->warp.size(), f->scaled.size());
    return rc;
}

extern "C" {

// The chunk blobs in the pinned twin that are not on their way yet: one transfer on the copy stream.  Only complete blobs may go:
// the caller guarantees that no submission is in progress (dav1d_hip_frame_flush is called between the listing and the frame end).
static int frame_flush_locked(Dav1dHipFrame *f) {
    int rc = 0;
    if (f->hcarena) {
        // segments are drawn whole: one that crosses the end of the twin went the late way, everything below `used` that is in the twin is complete
        const size_t used = std::min(f->carena_used.load(), f->hcarena_cap);
        if (used > f->hcarena_flushed) {
            rc = hip_rc(hipMemcpyAsync(f->carena + f->hcarena_flushed, f->hcarena + f->hcarena_flushed, used - f->hcarena_flushed, hipMemcpyHostToDevice,
                                       f->c->copy_stream));
            if (!rc) f->hcarena_flushed = used;
        }
    }
    if (!f->harena || rc) return rc;
    const size_t used = std::min(std::min(f->arena_used.load(), f->arena_cap), f->harena_cap);
    if (used <= f->harena_flushed) return 0;
    rc = hip_rc(hipMemcpyAsync(f->arena + f->harena_flushed, f->harena + f->harena_flushed, used - f->harena_flushed, hipMemcpyHostToDevice,
                               f->c->copy_stream));
    if (!rc) f->harena_flushed = used;
    return rc;
}

// Optional, once every tile-sbrow submitted so far has returned: start the transfer of the chunks now instead of at frame end (it
// then runs under whatever the caller does next — the filter listing, the tail of the coefficient upload).  Not thread-safe
// against submissions in progress.
int dav1d_hip_frame_flush(Dav1dHipFrame *f) {
    if (!f) return -EINVAL;
    {   // (the preparations handed out so far: their blobs are what goes; an error stays for dav1d_hip_frame_end to report)
        std::unique_lock<std::mutex> lk(f->prep_mtx);
        f->prep_cv.wait(lk, [f]() { return f->prep_pending.load() == 0; });
    }
    std::lock_guard<std::mutex> lk(f->mtx);
    return frame_flush_locked(f);
}

int dav1d_hip_frame_set_super_res(Dav1dHipFrame *f, int sr_w) {
    if (!f || sr_w < 0) return -EINVAL;
    if (f->worker.joinable()) return -EBUSY;
    // AV1: the upscaled width is at most twice the coded one (denominators 9 .. 16 over 8), never smaller
    if (sr_w && (sr_w < f->cur.p[0].w || sr_w > 2 * f->cur.p[0].w)) return -EINVAL;
    f->sr_w = sr_w == f->cur.p[0].w ? 0 : sr_w;
    return 0;
}

int dav1d_hip_frame_set_progress_callback(Dav1dHipFrame *f, void (*progress)(void *cookie, int rows, const Dav1dHipPicture *pic), void *cookie) {
    if (!f) return -EINVAL;
    if (f->worker.joinable()) return -EBUSY;
    f->progress_cb = progress;
    f->progress_cookie = cookie;
    return 0;
}

int dav1d_hip_frame_post_bands(const Dav1dHipFrame *f) { return f ? f->post_bands : 0; }

// ---- asynchronous frame end + progress (reference src/thread_task.c:888-896 publishes f->sr_cur.progress[1] when a frame's
// rows are final; :393-439 check_tile waits on it).  The stages of a frame run over the WHOLE picture one after the other here, so
// rows become final together: progress is 0 until the frame is through and the picture height afterwards.  What the calling
// thread gets is that it does not have to wait: dav1d_hip_frame_end_async hands the blocking dav1d_hip_frame_end to a thread of
// the library and returns; the callback (optional) runs on that thread when the frame is final — the place to store
// progress[1] and wake the dependent frames' tasks.
int dav1d_hip_frame_end_async(Dav1dHipFrame *f, void *coef, int16_t *prep, uint8_t *mask, const Dav1dHipPicture *grain_out,
                              void (*done)(void *cookie, int rc, const Dav1dHipPicture *filtered), void *cookie) {
    if (!f) return -EINVAL;
    if (f->worker.joinable()) return -EBUSY;
    f->async_rc = 1;                                  // running
    f->progress_rows.store(0);
    const Dav1dHipPicture g = grain_out ? *grain_out : Dav1dHipPicture();
    const bool have_g = grain_out != nullptr;
    f->worker = std::thread([f, coef, prep, mask, g, have_g, done, cookie]() {
        (void) hipSetDevice(f->c->device);
        const int rc = dav1d_hip_frame_end(f, coef, prep, mask, &f->async_filtered, have_g ? &g : nullptr);
        f->async_rc = rc;
        if (done) done(cookie, rc, &f->async_filtered);
    });
    return 0;
}
// rows of the picture that are final (all in-loop filters applied)
int dav1d_hip_frame_progress(const Dav1dHipFrame *f) { return f ? f->progress_rows.load() : -EINVAL; }
// waits for the frame handed to dav1d_hip_frame_end_async; returns its result, *filtered as dav1d_hip_frame_end
int dav1d_hip_frame_wait(Dav1dHipFrame *f, Dav1dHipPicture *filtered) {
    if (!f) return -EINVAL;
    if (f->worker.joinable()) f->worker.join();
    if (filtered) *filtered = f->async_filtered;
    return f->async_rc == 1 ? -EINVAL : f->async_rc;
}

void dav1d_hip_frame_destroy(Dav1dHipFrame *f) {
    if (!f) return;
    __atomic_fetch_sub(&dav1d_hip_live[1], 1, __ATOMIC_RELAXED);
    if (f->worker.joinable()) f->worker.join();
    (void) frame_wait_prep(f);
    (void) hipStreamSynchronize(f->c->stream);
    if (f->prepared) dav1d_hip_fg_grain_destroy(f->c, f->prepared);
    (void) hipStreamSynchronize(f->c->copy_stream);
    for (Dav1dHipChunk *ck : f->chunks) { ck->release(f->c); delete ck; }
    for (Dav1dHipFrame::StepChunk *sc : f->step_chunks) delete sc;
    for (Dav1dHipFrame::FilterPiece &p : f->filter_pieces) { free(p.lf); free(p.cdef); free(p.lr); delete p.groups; }
    if (f->cdef_rows) dav1d_hip_slab_put(f->c, reinterpret_cast<uint8_t *>(f->cdef_rows), f->cdef_rows_cap);
    if (f->lrplan.slab) dav1d_hip_slab_put(f->c, f->lrplan.slab, f->lrplan.slab_cap);
    f->deferred.release(f->c);
    if (f->harena) dav1d_hip_slab_put(f->c, f->harena, f->harena_cap);
    if (f->huarena) dav1d_hip_slab_put(f->c, f->huarena, f->huarena_cap);
    if (f->hcarena) dav1d_hip_slab_put(f->c, f->hcarena, f->hcarena_cap);
    for (Dav1dHipFrame::LateCoefs &lc : f->late_coefs) free(lc.copy);
    if (f->arena || f->carena) {
        std::lock_guard<std::mutex> lk(f->c->pool_mtx);
        if (f->arena) f->c->free_arenas.push_back({ f->arena, f->arena_cap });
        if (f->carena) f->c->free_arenas.push_back({ f->carena, f->carena_cap });
    }
    for (int i = 0; i < 2; i++) if (f->have_tmp[i]) dav1d_hip_picture_give(f->c, &f->tmp[i]);
    for (int i = 0; i < 3; i++) if (f->have_sr[i]) dav1d_hip_picture_free(f->c, &f->sr[i]);
    delete f;
}

} // extern "C"

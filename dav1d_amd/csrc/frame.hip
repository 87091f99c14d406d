// Driver-level boundary: one frame in flight.  The reference's pass-2 workers call f->bd_fn.recon_b_* per block and
// filter_sbrow_* per superblock row (src/thread_task.c:733-851, src/recon_tmpl.c:1557-2135); the lister that replaces
// them appends the equivalent flat tasks here, tile-sbrow by tile-sbrow and from any worker thread, and
// dav1d_hip_frame_end() runs the whole frame in the reference's stage order:
//   inter prediction (+ fused compounds) -> residuals -> deblock (cols, rows) -> CDEF -> loop restoration -> film grain.
// Out-of-place stages land in pictures the frame owns; host code only: every kernel is reached through the batch API.
#include "capi.h"
#include <string.h>
#include <mutex>
#include <new>
#include <vector>

struct Dav1dHipFrame {
    Dav1dHipContext *c;
    Dav1dHipPicture cur;
    Dav1dHipPicture refs[8];
    int n_refs;
    std::mutex mtx;
    std::vector<Dav1dHipMcTask> mc;
    std::vector<Dav1dHipCompTask> comp;
    std::vector<Dav1dHipItxTask> itx;
    std::vector<Dav1dHipLfTask> lf;
    std::vector<Dav1dHipCdefTask> cdef;
    std::vector<Dav1dHipLrTask> lr;
    const uint8_t *lvl;
    ptrdiff_t b4_stride;
    uint8_t lut_e[64], lut_i[64];
    int cdef_damping;
    bool have_grain;
    Dav1dHipFilmGrainData grain;
    int is_id;
    Dav1dHipPicture tmp[2];          // CDEF output, restoration output (allocated on first use)
    bool have_tmp[2];
};

extern "C" {

int dav1d_hip_frame_begin(Dav1dHipContext *c, Dav1dHipFrame **out, const Dav1dHipPicture *cur, const Dav1dHipPicture *refs, int n_refs) {
    if (!c || !out || !cur || n_refs < 0 || n_refs > 8 || (n_refs && !refs)) return -EINVAL;
    *out = nullptr;
    for (int i = 0; i < n_refs; i++) if (refs[i].bpc != cur->bpc || refs[i].layout != cur->layout) return -EINVAL;
    Dav1dHipFrame *f = new (std::nothrow) Dav1dHipFrame();
    if (!f) return -ENOMEM;
    f->c = c;
    f->cur = *cur;
    for (int i = 0; i < n_refs; i++) f->refs[i] = refs[i];
    f->n_refs = n_refs;
    f->lvl = nullptr;
    f->b4_stride = 0;
    f->cdef_damping = 0;
    f->have_grain = false;
    f->is_id = 0;
    f->have_tmp[0] = f->have_tmp[1] = false;
    *out = f;
    return 0;
}

// Reconstruction tasks of one tile-sbrow (what decode_b()'s pass-2 branch would have executed, src/decode.c:706-806).
// Thread-safe; the order between tile-sbrows is free: inter tasks of a frame write disjoint pixels, and every residual
// is added after every prediction.
int dav1d_hip_frame_submit_tile_sbrow(Dav1dHipFrame *f, const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                                      const Dav1dHipItxTask *itx, size_t n_itx) {
    if (!f || (!mc && n_mc) || (!comp && n_comp) || (!itx && n_itx)) return -EINVAL;
    std::lock_guard<std::mutex> lk(f->mtx);
    f->mc.insert(f->mc.end(), mc, mc + n_mc);
    f->comp.insert(f->comp.end(), comp, comp + n_comp);
    f->itx.insert(f->itx.end(), itx, itx + n_itx);
    return 0;
}

// In-loop filter tasks of one superblock row (dav1d_filter_sbrow_deblock_cols / _rows / _cdef / _lr, src/recon_tmpl.c:1985-2135).
// Loop restoration tasks must arrive in raster order inside a superblock row.  Thread-safe.
int dav1d_hip_frame_submit_filter_sbrow(Dav1dHipFrame *f, const Dav1dHipLfTask *lf, size_t n_lf, const Dav1dHipCdefTask *cdef, size_t n_cdef,
                                        const Dav1dHipLrTask *lr, size_t n_lr) {
    if (!f || (!lf && n_lf) || (!cdef && n_cdef) || (!lr && n_lr)) return -EINVAL;
    std::lock_guard<std::mutex> lk(f->mtx);
    f->lf.insert(f->lf.end(), lf, lf + n_lf);
    f->cdef.insert(f->cdef.end(), cdef, cdef + n_cdef);
    f->lr.insert(f->lr.end(), lr, lr + n_lr);
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
    f->cdef_damping = cdef_damping;
    f->have_grain = grain != nullptr;
    if (grain) f->grain = *grain;
    f->is_id = is_id;
    return 0;
}

static int frame_tmp(Dav1dHipFrame *f, int i) {
    if (f->have_tmp[i]) return 0;
    const int rc = dav1d_hip_picture_alloc(f->c, &f->tmp[i], f->cur.p[0].w, f->cur.p[0].h, f->cur.layout, f->cur.bpc);
    if (!rc) f->have_tmp[i] = true;
    return rc;
}

static int copy_picture(Dav1dHipContext *c, const Dav1dHipPicture *dst, const Dav1dHipPicture *src) {
    if (dst->alloc_size && dst->alloc_size == src->alloc_size && dst->alloc && src->alloc)
        return hip_rc(hipMemcpyAsync(dst->alloc, src->alloc, src->alloc_size, hipMemcpyDeviceToDevice, c->stream));
    const int bps = src->bpc > 8 ? 2 : 1;
    for (int pl = 0; pl < 3; pl++) {
        if (!src->p[pl].data) continue;
        const int rc = hip_rc(hipMemcpy2DAsync(dst->p[pl].data, dst->p[pl].stride, src->p[pl].data, src->p[pl].stride,
                                               (size_t) src->p[pl].w * bps, src->p[pl].h, hipMemcpyDeviceToDevice, c->stream));
        if (rc) return rc;
    }
    return 0;
}

// Runs the frame.  coef / prep / mask: the DEVICE arenas the task offsets refer to.  On return `cur` holds the
// reconstructed AND deblocked picture (deblocking is in place, as in the reference); *filtered receives a descriptor of
// the picture after CDEF and loop restoration (it is `cur` itself when neither stage has tasks; otherwise a picture owned by
// the frame, valid until dav1d_hip_frame_destroy); when film grain parameters were set and `grain_out` is given, the grain
// is applied from *filtered into grain_out (dav1d_apply_grain, src/lib.c:311-329).  Synchronous: every stage has
// completed on return, so the caller can publish progress the way src/thread_task.c:888-896 does.
int dav1d_hip_frame_end(Dav1dHipFrame *f, void *coef, int16_t *prep, uint8_t *mask, Dav1dHipPicture *filtered, const Dav1dHipPicture *grain_out) {
    if (!f) return -EINVAL;
    std::lock_guard<std::mutex> lk(f->mtx);
    Dav1dHipContext *c = f->c;
    int rc = 0;
    if (!f->mc.empty() || !f->comp.empty()) {
        // predictions and residuals as one pipelined list (the residual launch of a transform size waits only for the
        // prediction launches under its blocks)
        if (!f->n_refs) return -EINVAL;
        Dav1dHipReconList *rl = nullptr;
        rc = dav1d_hip_recon_list_create(c, &rl, &f->cur, f->mc.data(), f->mc.size(), f->comp.data(), f->comp.size(),
                                         f->itx.data(), f->itx.size());
        if (!rc) rc = dav1d_hip_recon_list_run(c, rl, &f->cur, f->refs, f->n_refs, prep, mask, coef);
        if (rl) dav1d_hip_recon_list_destroy(c, rl);
    } else if (!f->itx.empty()) {
        rc = dav1d_hip_itx_add_batch(c, &f->cur, f->itx.data(), f->itx.size(), coef);
    }
    if (!rc && !f->lf.empty()) {
        if (!f->lvl) return -EINVAL;
        rc = dav1d_hip_lf_batch(c, &f->cur, f->lf.data(), f->lf.size(), f->lvl, f->b4_stride, f->lut_e, f->lut_i);
    }
    const Dav1dHipPicture *last = &f->cur;
    if (!rc && !f->cdef.empty()) {
        rc = frame_tmp(f, 0);
        if (!rc) rc = copy_picture(c, &f->tmp[0], &f->cur);       // units that are not listed keep their pixels
        if (!rc) rc = dav1d_hip_cdef_batch(c, &f->tmp[0], &f->cur, f->cdef.data(), f->cdef.size(), f->cdef_damping, nullptr);
        last = &f->tmp[0];
    }
    if (!rc && !f->lr.empty()) {
        rc = frame_tmp(f, 1);
        if (!rc) rc = copy_picture(c, &f->tmp[1], last);
        if (!rc) rc = dav1d_hip_lr_batch(c, &f->tmp[1], last, &f->cur, f->lr.data(), f->lr.size());
        last = &f->tmp[1];
    }
    if (!rc && filtered) *filtered = *last;
    if (!rc && f->have_grain && grain_out) rc = dav1d_hip_fg_apply(c, grain_out, last, &f->grain, f->is_id);
    if (!rc) rc = dav1d_hip_sync(c);
    return rc;
}

void dav1d_hip_frame_destroy(Dav1dHipFrame *f) {
    if (!f) return;
    (void) hipStreamSynchronize(f->c->stream);
    for (int i = 0; i < 2; i++) if (f->have_tmp[i]) dav1d_hip_picture_free(f->c, &f->tmp[i]);
    delete f;
}

} // extern "C"

/* Frame-level parity harness, linked INTO the oracle build of the reference (oracle/_ref/libdav1d_ref.so).
 * TEST INFRASTRUCTURE ONLY: nothing in the product library links or loads this.
 *
 * It builds a real Dav1dContext / Dav1dFrameContext / Dav1dTaskContext the way dav1d_submit_frame() does
 * (reference src/decode.c:3290-3560), lets the reference's own dav1d_decode_frame_init() (:2750-3140) allocate every
 * per-frame array, exposes those arrays so that a test can fill them with a synthetic pass-1 output, and then runs the
 * reference's OWN pass 2 — dav1d_decode_tile_sbrow() with frame_thread.pass = 2 (:2594-2635), i.e. decode_sb / decode_b /
 * dav1d_recon_b_intra / dav1d_recon_b_inter through the reference DSP table — and its OWN in-loop filters
 * (dav1d_filter_sbrow_{8,16}bpc, src/recon_tmpl.c:2100-2109) on the CPU.  No codec logic is restated here: only the set-up
 * of the context structs, the tile cursors of setup_tile() (:2438-2452, static in the reference) and, for the filter inputs,
 * the per-block calls of dav1d_create_lf_mask_intra / _inter that pass 1 makes (:1216-1226, 1882-1900). */
#include "config.h"
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include <limits.h>
#include "src/internal.h"
#include "src/tables.h"
#include "src/wedge.h"
#include "src/intra_edge.h"
#include "src/qm.h"
#include "src/warpmv.h"
#include "src/decode.h"
#include "src/recon.h"
#include "src/lf_mask.h"
#include "src/env.h"

typedef struct RefFrameParams {
    int w, h, layout, bpc, sb128;
    int is_inter;
    int n_tile_cols, n_tile_rows;
    uint16_t col_start_sb[65], row_start_sb[65];
    int intra_edge_filter;
    int allow_screen_content_tools;
    int switchable_comp_refs;
    int ref_w[7], ref_h[7];
    int ref_poc[7], cur_poc, order_hint_n_bits;
    int gmv_type[7];
    int32_t gmv_matrix[7][6];
    /* in-loop filters */
    int lf_level_y[2], lf_level_u, lf_level_v, lf_sharpness;
    int lf_mode_ref_delta_enabled;
    int lf_ref_delta[8], lf_mode_delta[2];
    int cdef_enabled, cdef_damping, cdef_n_bits;
    int cdef_y_strength[8], cdef_uv_strength[8];
    int lr_type[3], lr_unit_size[2];
    int sr_w;                     /* super-resolution: width of the upscaled frame (0 or w: none) */
    int delta_lf;                 /* 1: delta_lf present — every superblock gets level deltas of its own (2: one delta for all four levels) */
    /* segmentation: per-segment deblocking level deltas (seg_data.d[s].delta_lf_y_v, _y_h, _u, _v) and lossless segments */
    int seg_enabled;
    int seg_delta_lf[8][4];
    int seg_lossless[8];
} RefFrameParams;

typedef struct RefFrame {
    Dav1dContext c;
    Dav1dFrameContext f;
    Dav1dSequenceHeader seq;
    Dav1dFrameHeader fh, ref_fh[7];
    Dav1dTaskContext *tc;
    atomic_int flush_mem;
    void *pic_mem[9];             /* 0 current, 1 + i reference i, 8 the upscaled current picture (super-resolution) */
    size_t plane_bytes[9][3];
    refmvs_temporal_block *mvs;
    RefFrameParams p;
    /* pass-1 stand-in state for the loop filter masks: the above contexts are the pass-1 half of f->a itself */
    BlockContext lf_l;
    int cur_tile_row;
    uint32_t rng;
    /* delta_lf: the level table pass 1 would have had in ts->lflvlmem while it parsed superblock [sb row * sbw + sb column]
     * (dav1d_calc_lf_values with the superblock's deltas, src/decode.c:1180-1206) */
    uint8_t (*sb_lflvl)[8][4][8][2];
    int sbw;
} RefFrame;

static void once_init(void) {
    static int done;
    if (done) return;
    dav1d_init_cpu();
    dav1d_init_ii_wedge_masks();
    dav1d_init_intra_edge_tree();
    dav1d_init_qm_tables();
    done = 1;
}

/* geometry of dav1d_default_picture_alloc(), src/picture.c:46-78 */
static int alloc_picture(RefFrame *r, const int slot, Dav1dPicture *p, const int w, const int h, const int layout, const int bpc) {
    const int hbd = bpc > 8;
    const int aligned_w = (w + 127) & ~127, aligned_h = (h + 127) & ~127;
    const int has_chroma = layout != DAV1D_PIXEL_LAYOUT_I400;
    const int ss_ver = layout == DAV1D_PIXEL_LAYOUT_I420, ss_hor = layout != DAV1D_PIXEL_LAYOUT_I444;
    ptrdiff_t y_stride = aligned_w << hbd;
    ptrdiff_t uv_stride = has_chroma ? y_stride >> ss_hor : 0;
    if (!(y_stride & 1023)) y_stride += DAV1D_PICTURE_ALIGNMENT;
    if (!(uv_stride & 1023) && has_chroma) uv_stride += DAV1D_PICTURE_ALIGNMENT;
    const size_t y_sz = y_stride * aligned_h, uv_sz = uv_stride * (aligned_h >> ss_ver);
    void *buf = NULL;
    if (posix_memalign(&buf, 64, y_sz + 2 * uv_sz + DAV1D_PICTURE_ALIGNMENT)) return -1;
    memset(buf, 0, y_sz + 2 * uv_sz + DAV1D_PICTURE_ALIGNMENT);
    memset(p, 0, sizeof(*p));
    p->p.w = w; p->p.h = h; p->p.layout = layout; p->p.bpc = bpc;
    p->stride[0] = y_stride; p->stride[1] = uv_stride;
    p->data[0] = buf;
    p->data[1] = has_chroma ? (uint8_t *) buf + y_sz : NULL;
    p->data[2] = has_chroma ? (uint8_t *) buf + y_sz + uv_sz : NULL;
    r->pic_mem[slot] = buf;
    r->plane_bytes[slot][0] = y_sz; r->plane_bytes[slot][1] = r->plane_bytes[slot][2] = uv_sz;
    return 0;
}

void dav1d_ref_frame_destroy(void *h);

void *dav1d_ref_frame_create(const RefFrameParams *const p) {
    once_init();
    RefFrame *r = NULL;
    if (posix_memalign((void **) &r, 64, sizeof(*r))) return NULL;
    memset(r, 0, sizeof(*r));
    r->p = *p;
    Dav1dContext *const c = &r->c;
    Dav1dFrameContext *const f = &r->f;
    const int bits = p->bpc == 8 ? 0 : p->bpc == 10 ? 1 : 2;

    /* the frame-threaded configuration that produces the two-pass hand-off (src/lib.c:140-301: n_fc > 1) */
    c->n_fc = 2; c->n_tc = 2;
    c->fc = f;
    c->flush = &r->flush_mem;
    c->inloop_filters = DAV1D_INLOOPFILTER_ALL;
    if (posix_memalign((void **) &r->tc, 64, sizeof(*r->tc))) { free(r); return NULL; }
    memset(r->tc, 0, sizeof(*r->tc));
    c->tc = r->tc;
    if (p->bpc == 8) {
        dav1d_cdef_dsp_init_8bpc(&c->dsp[0].cdef); dav1d_intra_pred_dsp_init_8bpc(&c->dsp[0].ipred); dav1d_itx_dsp_init_8bpc(&c->dsp[0].itx, 8);
        dav1d_loop_filter_dsp_init_8bpc(&c->dsp[0].lf); dav1d_loop_restoration_dsp_init_8bpc(&c->dsp[0].lr, 8); dav1d_mc_dsp_init_8bpc(&c->dsp[0].mc);
        dav1d_film_grain_dsp_init_8bpc(&c->dsp[0].fg);
    } else {
        dav1d_cdef_dsp_init_16bpc(&c->dsp[bits].cdef); dav1d_intra_pred_dsp_init_16bpc(&c->dsp[bits].ipred); dav1d_itx_dsp_init_16bpc(&c->dsp[bits].itx, p->bpc);
        dav1d_loop_filter_dsp_init_16bpc(&c->dsp[bits].lf); dav1d_loop_restoration_dsp_init_16bpc(&c->dsp[bits].lr, p->bpc); dav1d_mc_dsp_init_16bpc(&c->dsp[bits].mc);
        dav1d_film_grain_dsp_init_16bpc(&c->dsp[bits].fg);
    }
    dav1d_refmvs_dsp_init(&c->refmvs_dsp);
    dav1d_pal_dsp_init(&c->pal_dsp);

    /* headers: the fields pass 2 and the in-loop filters read */
    Dav1dSequenceHeader *const seq = &r->seq;
    Dav1dFrameHeader *const fh = &r->fh;
    seq->sb128 = p->sb128;
    seq->hbd = bits;
    seq->layout = p->layout;
    seq->intra_edge_filter = p->intra_edge_filter;
    seq->order_hint_n_bits = p->order_hint_n_bits;
    seq->order_hint = p->order_hint_n_bits > 0;
    seq->cdef = p->cdef_enabled;
    seq->restoration = p->lr_type[0] || p->lr_type[1] || p->lr_type[2];
    fh->frame_type = p->is_inter ? DAV1D_FRAME_TYPE_INTER : DAV1D_FRAME_TYPE_KEY;
    fh->width[0] = fh->width[1] = p->w;
    if (p->sr_w && p->sr_w != p->w) {
        /* super-resolution: the frame is coded p->w wide and upscaled to p->sr_w between CDEF and restoration */
        fh->width[1] = p->sr_w;
        fh->super_res.enabled = 1;
        fh->super_res.width_scale_denominator = (8 * p->sr_w + p->w / 2) / p->w;       /* parse-side only; the filters use the widths */
    }
    fh->height = p->h;
    fh->frame_offset = p->cur_poc;
    fh->allow_screen_content_tools = p->allow_screen_content_tools;
    fh->switchable_comp_refs = p->switchable_comp_refs;
    fh->tiling.cols = p->n_tile_cols; fh->tiling.rows = p->n_tile_rows;
    for (int i = 0; i <= p->n_tile_cols; i++) fh->tiling.col_start_sb[i] = p->col_start_sb[i];
    for (int i = 0; i <= p->n_tile_rows; i++) fh->tiling.row_start_sb[i] = p->row_start_sb[i];
    fh->loopfilter.level_y[0] = p->lf_level_y[0]; fh->loopfilter.level_y[1] = p->lf_level_y[1];
    fh->loopfilter.level_u = p->lf_level_u; fh->loopfilter.level_v = p->lf_level_v;
    fh->loopfilter.sharpness = p->lf_sharpness;
    fh->loopfilter.mode_ref_delta_enabled = p->lf_mode_ref_delta_enabled;
    for (int i = 0; i < 8; i++) fh->loopfilter.mode_ref_deltas.ref_delta[i] = p->lf_ref_delta[i];
    for (int i = 0; i < 2; i++) fh->loopfilter.mode_ref_deltas.mode_delta[i] = p->lf_mode_delta[i];
    fh->cdef.damping = p->cdef_damping; fh->cdef.n_bits = p->cdef_n_bits;
    for (int i = 0; i < 8; i++) { fh->cdef.y_strength[i] = p->cdef_y_strength[i]; fh->cdef.uv_strength[i] = p->cdef_uv_strength[i]; }
    for (int i = 0; i < 3; i++) fh->restoration.type[i] = p->lr_type[i];
    fh->restoration.unit_size[0] = p->lr_unit_size[0]; fh->restoration.unit_size[1] = p->lr_unit_size[1];
    fh->txfm_mode = DAV1D_TX_SWITCHABLE;
    if (p->seg_enabled) {
        fh->segmentation.enabled = 1;
        for (int s = 0; s < 8; s++) {
            Dav1dSegmentationData *const sd = &fh->segmentation.seg_data.d[s];
            sd->delta_lf_y_v = p->seg_delta_lf[s][0]; sd->delta_lf_y_h = p->seg_delta_lf[s][1];
            sd->delta_lf_u = p->seg_delta_lf[s][2]; sd->delta_lf_v = p->seg_delta_lf[s][3];
            sd->ref = -1;
            fh->segmentation.lossless[s] = p->seg_lossless[s];
            fh->segmentation.qidx[s] = p->seg_lossless[s] ? 0 : 100;
        }
    }
    for (int i = 0; i < 7; i++) {
        fh->gmv[i] = dav1d_default_wm_params;
        fh->gmv[i].type = p->gmv_type[i];
        if (p->gmv_type[i]) memcpy(fh->gmv[i].matrix, p->gmv_matrix[i], sizeof(fh->gmv[i].matrix));
    }

    f->c = c;
    f->seq_hdr = seq;
    f->frame_hdr = fh;
    f->dsp = &c->dsp[bits];
    f->lf.last_sharpness = -1;
    /* bd_fn, src/decode.c:3419-3442 */
    if (p->bpc == 8) {
        f->bd_fn.recon_b_inter = dav1d_recon_b_inter_8bpc; f->bd_fn.recon_b_intra = dav1d_recon_b_intra_8bpc;
        f->bd_fn.filter_sbrow = dav1d_filter_sbrow_8bpc; f->bd_fn.backup_ipred_edge = dav1d_backup_ipred_edge_8bpc;
        f->bd_fn.read_coef_blocks = dav1d_read_coef_blocks_8bpc;
    } else {
        f->bd_fn.recon_b_inter = dav1d_recon_b_inter_16bpc; f->bd_fn.recon_b_intra = dav1d_recon_b_intra_16bpc;
        f->bd_fn.filter_sbrow = dav1d_filter_sbrow_16bpc; f->bd_fn.backup_ipred_edge = dav1d_backup_ipred_edge_16bpc;
        f->bd_fn.read_coef_blocks = dav1d_read_coef_blocks_16bpc;
    }

    /* pictures: current (no super-resolution: sr_cur is cur) and the seven references */
    if (alloc_picture(r, 0, &f->cur, p->w, p->h, p->layout, p->bpc)) goto fail;
    f->cur.seq_hdr = seq; f->cur.frame_hdr = fh;
    f->sr_cur.p = f->cur;
    if (fh->super_res.enabled) {
        /* dav1d_submit_frame(), src/decode.c:3524-3540: a picture of its own for the upscaled frame, the step and the first
         * position of the horizontal resampler (AV1 spec 7.16) per plane class */
        if (alloc_picture(r, 8, &f->sr_cur.p, p->sr_w, p->h, p->layout, p->bpc)) goto fail;
        f->sr_cur.p.seq_hdr = seq; f->sr_cur.p.frame_hdr = fh;
        const int ss_hor = p->layout != DAV1D_PIXEL_LAYOUT_I444;
        const int in_w[2] = { p->w, (p->w + ss_hor) >> ss_hor }, out_w[2] = { p->sr_w, (p->sr_w + ss_hor) >> ss_hor };
        for (int i = 0; i < 2; i++) {
            const int step = ((in_w[i] << 14) + (out_w[i] >> 1)) / out_w[i];
            const int err = out_w[i] * step - (in_w[i] << 14);
            f->resize_step[i] = step;
            f->resize_start[i] = ((-((out_w[i] - in_w[i]) << 13) + (out_w[i] >> 1)) / out_w[i] + 128 - err / 2) & 0x3fff;
        }
    }
    if (p->is_inter)
        for (int i = 0; i < 7; i++) {
            if (alloc_picture(r, 1 + i, &f->refp[i].p, p->ref_w[i], p->ref_h[i], p->layout, p->bpc)) goto fail;
            r->ref_fh[i].frame_offset = p->ref_poc[i];
            r->ref_fh[i].width[0] = r->ref_fh[i].width[1] = p->ref_w[i];
            r->ref_fh[i].height = p->ref_h[i];
            f->refp[i].p.frame_hdr = &r->ref_fh[i];
            f->refp[i].p.seq_hdr = seq;
            f->refpoc[i] = p->ref_poc[i];
            /* src/decode.c:3469-3487 */
            if (p->w != p->ref_w[i] || p->h != p->ref_h[i]) {
#define scale_fac(ref_sz, this_sz) ((((ref_sz) << 14) + ((this_sz) >> 1)) / (this_sz))
                f->svc[i][0].scale = scale_fac(p->ref_w[i], p->w);
                f->svc[i][1].scale = scale_fac(p->ref_h[i], p->h);
                f->svc[i][0].step = (f->svc[i][0].scale + 8) >> 4;
                f->svc[i][1].step = (f->svc[i][1].scale + 8) >> 4;
#undef scale_fac
            }
            f->gmv_warp_allowed[i] = fh->gmv[i].type > DAV1D_WM_TYPE_TRANSLATION && !fh->force_integer_mv &&
                                     !dav1d_get_shear_params(&fh->gmv[i]) && !f->svc[i][0].scale;
        }

    /* geometry, src/decode.c:3552-3562 */
    f->w4 = (p->w + 3) >> 2; f->h4 = (p->h + 3) >> 2;
    f->bw = ((p->w + 7) >> 3) << 1; f->bh = ((p->h + 7) >> 3) << 1;
    f->sb128w = (f->bw + 31) >> 5; f->sb128h = (f->bh + 31) >> 5;
    f->sb_shift = 4 + seq->sb128; f->sb_step = 16 << seq->sb128;
    f->sbh = (f->bh + f->sb_step - 1) >> f->sb_shift;
    f->b4_stride = (f->bw + 31) & ~31;
    f->bitdepth_max = (1 << p->bpc) - 1;
    if (p->is_inter) {
        r->mvs = calloc((size_t) f->sb128h * 16 * (f->b4_stride >> 1), sizeof(*r->mvs));
        f->mvs = r->mvs;
    }
    if (dav1d_decode_frame_init(f)) goto fail;
    if (p->is_inter) memset(f->rf.r, 0, sizeof(*f->rf.r) * 35 * 2 * f->rf.n_blocks * 2);
    memset(f->frame_thread.b, 0, sizeof(*f->frame_thread.b) * f->sb128w * f->sb128h * 32 * 32);
    memset(f->lf.level, 0, sizeof(*f->lf.level) * f->sb128w * f->sb128h * 32 * 32 + 3);
    memset(f->lf.lr_mask, 0, sizeof(*f->lf.lr_mask) * f->lf.lr_mask_sz);
    memset(f->lf.tx_lpf_right_edge[0], 0, (size_t) f->lf.re_sz * 32 * 2);

    /* setup_tile(), src/decode.c:2425-2509, minus the entropy decoder */
    {
        static const uint8_t ss_size_mul[4][2] = { { 4, 4 }, { 6, 5 }, { 8, 6 }, { 12, 8 } };
        const uint8_t *const size_mul = ss_size_mul[p->layout];
        for (int tr = 0, j = 0; tr < p->n_tile_rows; tr++)
            for (int tcol = 0; tcol < p->n_tile_cols; tcol++, j++) {
                Dav1dTileState *const ts = &f->ts[j];
                const unsigned off = f->frame_thread.tile_start_off[j];
                for (int q = 0; q < 2; q++) {
                    ts->frame_thread[q].pal_idx = f->frame_thread.pal_idx ? &f->frame_thread.pal_idx[(size_t) off * size_mul[1] / 8] : NULL;
                    ts->frame_thread[q].cbi = &f->frame_thread.cbi[(size_t) off * size_mul[0] / 64];
                    ts->frame_thread[q].cf = (uint8_t *) f->frame_thread.cf + (((size_t) off * size_mul[0]) >> !seq->hbd);
                }
                ts->tiling.row = tr; ts->tiling.col = tcol;
                ts->tiling.col_start = fh->tiling.col_start_sb[tcol] << f->sb_shift;
                ts->tiling.col_end = imin(fh->tiling.col_start_sb[tcol + 1] << f->sb_shift, f->bw);
                ts->tiling.row_start = fh->tiling.row_start_sb[tr] << f->sb_shift;
                ts->tiling.row_end = imin(fh->tiling.row_start_sb[tr + 1] << f->sb_shift, f->bh);
                ts->lflvl = f->lf.lvl;
            }
    }
    return r;
fail:
    dav1d_ref_frame_destroy(r);
    return NULL;
}

void dav1d_ref_frame_destroy(void *const h) {
    RefFrame *const r = h;
    if (!r) return;
    for (int i = 0; i < 9; i++) free(r->pic_mem[i]);
    free(r->tc);
    free(r->mvs);
    free(r->sb_lflvl);
    /* the per-frame arrays dav1d_decode_frame_init() allocated stay with the process: test infrastructure */
    free(r);
}

/* The kernel-level drop-in of INTEGRATION.md 1: the frame's DSP table (421 function pointers, src/internal.h:62-70) is overwritten
 * with a table of the same layout — the one dav1d_hip_dsp_init_{8,16}bpc fills — so that the reference's own pass 2 and in-loop
 * filters make every DSP call through it. */
int dav1d_ref_frame_use_dsp(void *const h, const void *const table, const size_t bytes) {
    RefFrame *const r = h;
    const int bits = r->p.bpc == 8 ? 0 : r->p.bpc == 10 ? 1 : 2;
    if (bytes != sizeof(r->c.dsp[bits])) return -1;
    memcpy(&r->c.dsp[bits], table, bytes);
    return 0;
}

/* named access to the arrays a test fills / reads */
void *dav1d_ref_frame_ptr(void *const h, const char *const name, size_t *const bytes) {
    RefFrame *const r = h;
    Dav1dFrameContext *const f = &r->f;
    const int num_sb128 = f->sb128w * f->sb128h;
    size_t n = 0;
    void *ptr = NULL;
#define IS(s) (!strcmp(name, s))
    if (IS("b")) { ptr = f->frame_thread.b; n = sizeof(Av1Block) * num_sb128 * 32 * 32; }
    else if (IS("cbi")) { ptr = f->frame_thread.cbi; n = sizeof(int16_t) * (size_t) f->frame_thread.cbi_sz * 32 * 32 / 4; }
    else if (IS("cf")) { ptr = f->frame_thread.cf; n = (size_t) f->frame_thread.cf_sz * 128 * 128 / 2; }
    else if (IS("pal")) { ptr = f->frame_thread.pal; n = (size_t) f->frame_thread.pal_sz * 16 * 16 * 24; }
    else if (IS("pal_idx")) { ptr = f->frame_thread.pal_idx; n = (size_t) f->frame_thread.pal_idx_sz * 128 * 128 / 8; }
    else if (IS("tile_start_off")) { ptr = f->frame_thread.tile_start_off; n = sizeof(unsigned) * f->n_ts; }
    else if (IS("svc")) { ptr = f->svc; n = sizeof(f->svc); }
    else if (IS("gmv_warp_allowed")) { ptr = f->gmv_warp_allowed; n = sizeof(f->gmv_warp_allowed); }
    else if (IS("jnt_weights")) { ptr = f->jnt_weights; n = sizeof(f->jnt_weights); }
    else if (IS("gmv")) { ptr = r->fh.gmv; n = sizeof(r->fh.gmv); }
    else if (IS("lf_mask")) { ptr = f->lf.mask; n = sizeof(*f->lf.mask) * num_sb128; }
    else if (IS("lflvl")) { ptr = f->lf.lvl; n = sizeof(f->lf.lvl); }
    else if (IS("sb_lflvl")) { ptr = r->sb_lflvl; n = r->sb_lflvl ? sizeof(*r->sb_lflvl) * (size_t) r->sbw * f->sbh : 0; }
    else if (IS("lf_level")) { ptr = f->lf.level; n = sizeof(*f->lf.level) * num_sb128 * 32 * 32; }
    else if (IS("lr_mask")) { ptr = f->lf.lr_mask; n = sizeof(*f->lf.lr_mask) * f->lf.lr_mask_sz; }
    else if (IS("lim_lut")) { ptr = &f->lf.lim_lut; n = sizeof(f->lf.lim_lut); }
    else if (IS("a")) { ptr = f->a; n = sizeof(*f->a) * f->a_sz; }
    else if (IS("tx_lpf_right_edge0")) { ptr = f->lf.tx_lpf_right_edge[0]; n = (size_t) f->lf.re_sz * 32; }
    else if (IS("tx_lpf_right_edge1")) { ptr = f->lf.tx_lpf_right_edge[1]; n = (size_t) f->lf.re_sz * 32; }
    else if (!strncmp(name, "pic", 3) && name[3] >= '0' && name[3] <= '8' && name[4] == '_' && name[5] >= '0' && name[5] <= '2') {
        /* pic<slot>_<plane>: slot 0 = the current picture, 1 + i = reference i, 8 = the upscaled current picture */
        const int slot = name[3] - '0', pl = name[5] - '0';
        const Dav1dPicture *pic = slot == 8 ? &f->sr_cur.p : slot ? &f->refp[slot - 1].p : &f->cur;
        ptr = pic->data[pl]; n = slot == 8 && !r->pic_mem[8] ? 0 : r->plane_bytes[slot][pl];
    }
    else if (IS("resize")) { ptr = f->resize_step; n = sizeof(f->resize_step) + sizeof(f->resize_start); }
#undef IS
    if (bytes) *bytes = n;
    return ptr;
}

/* geometry a test needs: [0] b4_stride, [1] bw, [2] bh, [3] sb128w, [4] sbh, [5..6] cur strides, [7 + 2 i ..] ref strides */
void dav1d_ref_frame_geometry(void *const h, int64_t *const out) {
    RefFrame *const r = h;
    const Dav1dFrameContext *const f = &r->f;
    out[0] = f->b4_stride; out[1] = f->bw; out[2] = f->bh; out[3] = f->sb128w; out[4] = f->sbh;
    out[5] = f->cur.stride[0]; out[6] = f->cur.stride[1];
    for (int i = 0; i < 7; i++) { out[7 + 2 * i] = f->refp[i].p.stride[0]; out[8 + 2 * i] = f->refp[i].p.stride[1]; }
    out[21] = f->sr_cur.p.stride[0]; out[22] = f->sr_cur.p.stride[1]; out[23] = f->sr_sb128w;
}

/* sizeof / offsetof of the reference's hand-off structs, for pinning the product's mirrors */
void dav1d_ref_layouts(int *const out) {
    int n = 0;
    out[n++] = (int) sizeof(Av1Block);
    out[n++] = (int) offsetof(Av1Block, bl); out[n++] = (int) offsetof(Av1Block, bs); out[n++] = (int) offsetof(Av1Block, bp);
    out[n++] = (int) offsetof(Av1Block, intra); out[n++] = (int) offsetof(Av1Block, seg_id); out[n++] = (int) offsetof(Av1Block, skip_mode);
    out[n++] = (int) offsetof(Av1Block, skip); out[n++] = (int) offsetof(Av1Block, uvtx);
    out[n++] = (int) offsetof(Av1Block, y_mode); out[n++] = (int) offsetof(Av1Block, uv_mode); out[n++] = (int) offsetof(Av1Block, tx);
    out[n++] = (int) offsetof(Av1Block, pal_sz); out[n++] = (int) offsetof(Av1Block, y_angle); out[n++] = (int) offsetof(Av1Block, uv_angle);
    out[n++] = (int) offsetof(Av1Block, cfl_alpha);
    out[n++] = (int) offsetof(Av1Block, mv); out[n++] = (int) offsetof(Av1Block, wedge_idx); out[n++] = (int) offsetof(Av1Block, mask_sign);
    out[n++] = (int) offsetof(Av1Block, interintra_mode); out[n++] = (int) offsetof(Av1Block, mv2d); out[n++] = (int) offsetof(Av1Block, matrix);
    out[n++] = (int) offsetof(Av1Block, comp_type); out[n++] = (int) offsetof(Av1Block, inter_mode); out[n++] = (int) offsetof(Av1Block, motion_mode);
    out[n++] = (int) offsetof(Av1Block, drl_idx); out[n++] = (int) offsetof(Av1Block, ref); out[n++] = (int) offsetof(Av1Block, max_ytx);
    out[n++] = (int) offsetof(Av1Block, filter2d); out[n++] = (int) offsetof(Av1Block, interintra_type); out[n++] = (int) offsetof(Av1Block, tx_split0);
    out[n++] = (int) offsetof(Av1Block, tx_split1);
    out[n++] = (int) sizeof(Dav1dWarpedMotionParams);
    out[n++] = (int) offsetof(Dav1dWarpedMotionParams, type); out[n++] = (int) offsetof(Dav1dWarpedMotionParams, matrix);
    out[n++] = (int) offsetof(Dav1dWarpedMotionParams, u);
    out[n++] = (int) sizeof(Av1Filter); out[n++] = (int) sizeof(Av1Restoration); out[n++] = (int) sizeof(Av1RestorationUnit);
    out[n++] = (int) offsetof(Av1Filter, filter_y); out[n++] = (int) offsetof(Av1Filter, filter_uv); out[n++] = (int) offsetof(Av1Filter, cdef_idx);
    out[n++] = (int) offsetof(Av1Filter, noskip_mask);
    out[n++] = (int) offsetof(Av1RestorationUnit, type); out[n++] = (int) offsetof(Av1RestorationUnit, filter_h);
    out[n++] = (int) offsetof(Av1RestorationUnit, filter_v); out[n++] = (int) offsetof(Av1RestorationUnit, sgr_weights);
    out[n++] = (int) sizeof(BlockContext); out[n++] = (int) offsetof(BlockContext, tx_lpf_y); out[n++] = (int) offsetof(BlockContext, tx_lpf_uv);
    out[n++] = (int) sizeof(Av1FilterLUT); out[n++] = (int) offsetof(Av1FilterLUT, e); out[n++] = (int) offsetof(Av1FilterLUT, i);
    out[n++] = -1;
}

/* pass 2 of the whole frame: every tile-sbrow in the order a single worker would take them
 * (dav1d_decode_frame_main, src/decode.c:3196-3240) */
int dav1d_ref_frame_recon(void *const h) {
    RefFrame *const r = h;
    Dav1dFrameContext *const f = &r->f;
    Dav1dTaskContext *const t = r->tc;
    const Dav1dFrameHeader *const fh = &r->fh;
    const int keyframe = !r->p.is_inter;
    t->c = &r->c; t->f = f;
    t->frame_thread.pass = 2;
    /* reset_context(), src/decode.c:2385-2413, of the pass-2 half of f->a (dav1d_decode_frame_init_cdf, :3182-3188) */
    for (int n = f->sb128w * fh->tiling.rows; n < f->a_sz; n++) {      /* the pass-1 half keeps what pass 1 left (tx_lpf_*) */
        memset(&f->a[n], 0, sizeof(f->a[n]));
        memset(f->a[n].intra, keyframe, sizeof(f->a[n].intra));
        memset(f->a[n].uvmode, DC_PRED, sizeof(f->a[n].uvmode));
        if (keyframe) memset(f->a[n].mode, DC_PRED, sizeof(f->a[n].mode));
    }
    /* the tile cursors start over (a second run after the arrays changed) */
    {
        static const uint8_t ss_size_mul[4][2] = { { 4, 4 }, { 6, 5 }, { 8, 6 }, { 12, 8 } };
        const uint8_t *const size_mul = ss_size_mul[r->p.layout];
        for (int j = 0; j < f->n_ts; j++) {
            const unsigned off = f->frame_thread.tile_start_off[j];
            Dav1dTileState *const ts = &f->ts[j];
            ts->frame_thread[0].pal_idx = f->frame_thread.pal_idx ? &f->frame_thread.pal_idx[(size_t) off * size_mul[1] / 8] : NULL;
            ts->frame_thread[0].cbi = &f->frame_thread.cbi[(size_t) off * size_mul[0] / 64];
            ts->frame_thread[0].cf = (uint8_t *) f->frame_thread.cf + (((size_t) off * size_mul[0]) >> !r->seq.hbd);
        }
    }
    for (int tile_row = 0; tile_row < fh->tiling.rows; tile_row++)
        for (int sby = fh->tiling.row_start_sb[tile_row]; sby < fh->tiling.row_start_sb[tile_row + 1]; sby++) {
            t->by = sby << f->sb_shift;
            for (int tile_col = 0; tile_col < fh->tiling.cols; tile_col++) {
                t->ts = &f->ts[tile_row * fh->tiling.cols + tile_col];
                if (dav1d_decode_tile_sbrow(t)) return -1;
            }
        }
    return 0;
}

/* The same pass 2 on n_threads workers, the split dav1d's frame threading makes (tile-sbrow tasks, src/thread_task.c:693-760):
 * a worker takes the next tile whose turn it is and runs its superblock rows top to bottom; tiles do not depend on each
 * other in pass 2 (prediction never crosses a tile edge; the reference pictures are complete).  Every worker has a
 * Dav1dTaskContext of its own, as dav1d's worker threads do. */
#include <pthread.h>
typedef struct MtJob {
    RefFrame *r;
    int next;                /* next tile, under mu */
    int failed;
    pthread_mutex_t mu;
} MtJob;

static void *mt_worker(void *const arg) {
    MtJob *const job = arg;
    RefFrame *const r = job->r;
    Dav1dFrameContext *const f = &r->f;
    const Dav1dFrameHeader *const fh = &r->fh;
    Dav1dTaskContext *t = NULL;
    if (posix_memalign((void **) &t, 64, sizeof(*t))) { job->failed = 1; return NULL; }
    memset(t, 0, sizeof(*t));
    t->c = &r->c; t->f = f;
    t->frame_thread.pass = 2;
    for (;;) {
        pthread_mutex_lock(&job->mu);
        const int tile = job->next < f->n_ts ? job->next++ : -1;
        pthread_mutex_unlock(&job->mu);
        if (tile < 0) break;
        const int tile_row = tile / fh->tiling.cols;
        t->ts = &f->ts[tile];
        for (int sby = fh->tiling.row_start_sb[tile_row]; sby < fh->tiling.row_start_sb[tile_row + 1]; sby++) {
            t->by = sby << f->sb_shift;
            if (dav1d_decode_tile_sbrow(t)) { job->failed = 1; break; }
        }
    }
    free(t);
    return NULL;
}

int dav1d_ref_frame_recon_mt(void *const h, int n_threads) {
    RefFrame *const r = h;
    Dav1dFrameContext *const f = &r->f;
    const Dav1dFrameHeader *const fh = &r->fh;
    const int keyframe = !r->p.is_inter;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > f->n_ts) n_threads = f->n_ts;
    for (int n = f->sb128w * fh->tiling.rows; n < f->a_sz; n++) {
        memset(&f->a[n], 0, sizeof(f->a[n]));
        memset(f->a[n].intra, keyframe, sizeof(f->a[n].intra));
        memset(f->a[n].uvmode, DC_PRED, sizeof(f->a[n].uvmode));
        if (keyframe) memset(f->a[n].mode, DC_PRED, sizeof(f->a[n].mode));
    }
    {
        static const uint8_t ss_size_mul[4][2] = { { 4, 4 }, { 6, 5 }, { 8, 6 }, { 12, 8 } };
        const uint8_t *const size_mul = ss_size_mul[r->p.layout];
        for (int j = 0; j < f->n_ts; j++) {
            const unsigned off = f->frame_thread.tile_start_off[j];
            Dav1dTileState *const ts = &f->ts[j];
            ts->frame_thread[0].pal_idx = f->frame_thread.pal_idx ? &f->frame_thread.pal_idx[(size_t) off * size_mul[1] / 8] : NULL;
            ts->frame_thread[0].cbi = &f->frame_thread.cbi[(size_t) off * size_mul[0] / 64];
            ts->frame_thread[0].cf = (uint8_t *) f->frame_thread.cf + (((size_t) off * size_mul[0]) >> !r->seq.hbd);
        }
    }
    MtJob job = { .r = r, .next = 0, .failed = 0 };
    pthread_mutex_init(&job.mu, NULL);
    pthread_t *const th = malloc(sizeof(*th) * (size_t) n_threads);
    if (!th) return -1;
    int started = 0;
    for (; started < n_threads; started++)
        if (pthread_create(&th[started], NULL, mt_worker, &job)) break;
    if (!started) mt_worker(&job);
    for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    free(th);
    pthread_mutex_destroy(&job.mu);
    return job.failed ? -1 : 0;
}

/* t->warpmv of a MM_WARP block exactly as decode_b() rebuilds it in pass 2 (src/decode.c:743-757); returns what
 * dav1d_get_shear_params() returned.  out = { matrix[6], alpha, beta, gamma, delta } */
int dav1d_ref_block_warp(const int16_t *const matrix, const int16_t *const mv2d, const int bw4, const int bh4, const int bx, const int by,
                         int32_t *const out)
{
    Dav1dWarpedMotionParams wm;
    memset(&wm, 0, sizeof(wm));
    wm.type = DAV1D_WM_TYPE_AFFINE;
    wm.matrix[2] = matrix[0] + 0x10000;
    wm.matrix[3] = matrix[1];
    wm.matrix[4] = matrix[2];
    wm.matrix[5] = matrix[3] + 0x10000;
    const mv m = { .y = mv2d[0], .x = mv2d[1] };
    dav1d_set_affine_mv2d(bw4, bh4, m, &wm, bx, by);
    const int rc = dav1d_get_shear_params(&wm);
    for (int i = 0; i < 6; i++) out[i] = wm.matrix[i];
    out[6] = wm.u.p.alpha; out[7] = wm.u.p.beta; out[8] = wm.u.p.gamma; out[9] = wm.u.p.delta;
    return rc;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Filter inputs.  Pass 1 of the reference builds the deblocking masks, the level cache, noskip_mask and cdef_idx block by
 * block while it parses (src/decode.c:938-956, 1216-1226, 1882-1900, 1945-1956, 2730-2740).  Here the reference's own
 * block walk (decode_sb with pass == 2) visits every block with the two reconstruction hooks pointed at the functions
 * below, which make the same calls of the reference's dav1d_create_lf_mask_intra / _inter with the same arguments. */
static RefFrame *g_walk;       /* the hooks have no user pointer */

static uint32_t walk_rnd(RefFrame *const r) { r->rng = r->rng * 1664525u + 1013904223u; return r->rng >> 8; }

static void walk_common(Dav1dTaskContext *const t, const enum BlockSize bs, const Av1Block *const b) {
    RefFrame *const r = g_walk;
    const Dav1dFrameContext *const f = t->f;
    Av1Filter *const lf_mask = f->lf.mask + (t->by >> 5) * f->sb128w + (t->bx >> 5);
    const uint8_t *const b_dim = dav1d_block_dimensions[bs];
    const int bx4 = t->bx & 31, by4 = t->by & 31, bw4 = b_dim[0], bh4 = b_dim[1];
    /* cdef index of the 64x64(s) the block touches: read with the first block that has coefficients (:938-956) */
    if (!b->skip) {
        const int idx = ((t->bx & 16) >> 4) + ((t->by & 16) >> 3);
        if (lf_mask->cdef_idx[idx] == -1) {
            const int v = walk_rnd(r) & ((1 << f->frame_hdr->cdef.n_bits) - 1);
            lf_mask->cdef_idx[idx] = v;
            if (bw4 > 16) lf_mask->cdef_idx[idx + 1] = v;
            if (bh4 > 16) lf_mask->cdef_idx[idx + 2] = v;
            if (bw4 == 32 && bh4 == 32) lf_mask->cdef_idx[idx + 3] = v;
        }
        /* :1945-1956 */
        uint16_t (*noskip_mask)[2] = &lf_mask->noskip_mask[by4 >> 1];
        const unsigned mask = (~0U >> (32 - bw4)) << (bx4 & 15);
        const int bx_idx = (bx4 & 16) >> 4;
        for (int y = 0; y < bh4; y += 2, noskip_mask++) {
            (*noskip_mask)[bx_idx] |= mask;
            if (bw4 == 32) (*noskip_mask)[1] |= mask;
        }
    }
}

/* ts->lflvl as pass 1 had it at this block: the frame's table, or the superblock's own with delta_lf */
static const uint8_t (*walk_lflvl(const RefFrame *const r, const Dav1dTaskContext *const t))[4][8][2] {
    if (!r->sb_lflvl) return t->ts->lflvl;
    const Dav1dFrameContext *const f = t->f;
    return (const uint8_t (*)[4][8][2]) r->sb_lflvl[(t->by >> f->sb_shift) * r->sbw + (t->bx >> f->sb_shift)];
}

static void walk_intra(Dav1dTaskContext *const t, const enum BlockSize bs, const enum EdgeFlags flags, const Av1Block *const b) {
    (void) flags;
    RefFrame *const r = g_walk;
    const Dav1dFrameContext *const f = t->f;
    const Dav1dTileState *const ts = t->ts;
    const int ss_ver = f->cur.p.layout == DAV1D_PIXEL_LAYOUT_I420, ss_hor = f->cur.p.layout != DAV1D_PIXEL_LAYOUT_I444;
    const uint8_t *const b_dim = dav1d_block_dimensions[bs];
    const int bx4 = t->bx & 31, by4 = t->by & 31, cbx4 = bx4 >> ss_hor, cby4 = by4 >> ss_ver;
    const int has_chroma = f->cur.p.layout != DAV1D_PIXEL_LAYOUT_I400 && (b_dim[0] > ss_hor || t->bx & 1) && (b_dim[1] > ss_ver || t->by & 1);
    BlockContext *const a = &f->a[r->cur_tile_row * f->sb128w + (t->bx >> 5)];
    if (f->frame_hdr->loopfilter.level_y[0] || f->frame_hdr->loopfilter.level_y[1])
        dav1d_create_lf_mask_intra(f->lf.mask + (t->by >> 5) * f->sb128w + (t->bx >> 5), f->lf.level, f->b4_stride,
                                   (const uint8_t (*)[8][2]) &walk_lflvl(r, t)[b->seg_id][0][0][0], t->bx, t->by, f->w4, f->h4, bs,
                                   b->tx, b->uvtx, f->cur.p.layout, &a->tx_lpf_y[bx4], &r->lf_l.tx_lpf_y[by4],
                                   has_chroma ? &a->tx_lpf_uv[cbx4] : NULL, has_chroma ? &r->lf_l.tx_lpf_uv[cby4] : NULL);
    walk_common(t, bs, b);
}

static int walk_inter(Dav1dTaskContext *const t, const enum BlockSize bs, const Av1Block *const b) {
    RefFrame *const r = g_walk;
    const Dav1dFrameContext *const f = t->f;
    const Dav1dTileState *const ts = t->ts;
    const int ss_ver = f->cur.p.layout == DAV1D_PIXEL_LAYOUT_I420, ss_hor = f->cur.p.layout != DAV1D_PIXEL_LAYOUT_I444;
    const uint8_t *const b_dim = dav1d_block_dimensions[bs];
    const int bx4 = t->bx & 31, by4 = t->by & 31, cbx4 = bx4 >> ss_hor, cby4 = by4 >> ss_ver;
    const int has_chroma = f->cur.p.layout != DAV1D_PIXEL_LAYOUT_I400 && (b_dim[0] > ss_hor || t->bx & 1) && (b_dim[1] > ss_ver || t->by & 1);
    BlockContext *const a = &f->a[r->cur_tile_row * f->sb128w + (t->bx >> 5)];
    if (f->frame_hdr->loopfilter.level_y[0] || f->frame_hdr->loopfilter.level_y[1]) {
        const int is_comp = b->comp_type != COMP_INTER_NONE;
        const int is_globalmv = b->inter_mode == (is_comp ? GLOBALMV_GLOBALMV : GLOBALMV);
        const uint8_t (*const lf_lvls)[8][2] = (const uint8_t (*)[8][2]) &walk_lflvl(r, t)[b->seg_id][0][b->ref[0] + 1][!is_globalmv];
        const uint16_t tx_split[2] = { b->tx_split0, b->tx_split1 };
        /* the caller's own adjustment for lossless segments, src/decode.c:1889-1893 */
        enum RectTxfmSize ytx = b->max_ytx, uvtx = b->uvtx;
        if (f->frame_hdr->segmentation.lossless[b->seg_id]) { ytx = (enum RectTxfmSize) TX_4X4; uvtx = (enum RectTxfmSize) TX_4X4; }
        dav1d_create_lf_mask_inter(f->lf.mask + (t->by >> 5) * f->sb128w + (t->bx >> 5), f->lf.level, f->b4_stride, lf_lvls,
                                   t->bx, t->by, f->w4, f->h4, b->skip, bs, ytx, tx_split, uvtx, f->cur.p.layout,
                                   &a->tx_lpf_y[bx4], &r->lf_l.tx_lpf_y[by4],
                                   has_chroma ? &a->tx_lpf_uv[cbx4] : NULL, has_chroma ? &r->lf_l.tx_lpf_uv[cby4] : NULL);
    }
    walk_common(t, bs, b);
    return 0;
}

int dav1d_ref_frame_build_filter_inputs(void *const h, const unsigned seed) {
    RefFrame *const r = h;
    Dav1dFrameContext *const f = &r->f;
    Dav1dTaskContext *const t = r->tc;
    const Dav1dFrameHeader *const fh = &r->fh;
    const int num_sb128 = f->sb128w * f->sb128h;
    r->rng = seed * 2654435761u + 12345u;
    if (r->p.delta_lf) {
        /* one set of deltas per superblock, as read_delta_lf leaves them in ts->last_delta_lf (multiples of 1 << res_log2 inside
         * +-63), turned into level tables by the reference's own dav1d_calc_lf_values */
        r->fh.delta.lf.present = 1;
        r->fh.delta.lf.multi = r->p.delta_lf == 1;
        r->sbw = (f->bw + f->sb_step - 1) >> f->sb_shift;
        free(r->sb_lflvl);
        r->sb_lflvl = calloc((size_t) r->sbw * f->sbh, sizeof(*r->sb_lflvl));
        if (!r->sb_lflvl) return -1;
        for (int i = 0; i < r->sbw * f->sbh; i++) {
            int8_t delta[4];
            for (int k = 0; k < 4; k++) delta[k] = (int8_t) ((int) (walk_rnd(r) % 97) - 48);
            dav1d_calc_lf_values(r->sb_lflvl[i], &r->fh, delta);
        }
    }
    memset(f->lf.mask, 0, sizeof(*f->lf.mask) * num_sb128);
    for (int i = 0; i < num_sb128; i++) memset(f->lf.mask[i].cdef_idx, -1, 4);
    memset(f->lf.level, 0, sizeof(*f->lf.level) * num_sb128 * 32 * 32);
    /* reset_context() of the pass-1 half of f->a: tx_lpf_y = 2, tx_lpf_uv = 1 (:2401-2402) */
    for (int n = 0; n < f->sb128w * fh->tiling.rows; n++) {
        memset(f->a[n].tx_lpf_y, 2, sizeof(f->a[n].tx_lpf_y));
        memset(f->a[n].tx_lpf_uv, 1, sizeof(f->a[n].tx_lpf_uv));
    }
    recon_b_intra_fn keep_intra = f->bd_fn.recon_b_intra;
    recon_b_inter_fn keep_inter = f->bd_fn.recon_b_inter;
    backup_ipred_edge_fn keep_edge = f->bd_fn.backup_ipred_edge;
    f->bd_fn.recon_b_intra = walk_intra;
    f->bd_fn.recon_b_inter = walk_inter;
    g_walk = r;
    t->c = &r->c; t->f = f;
    t->frame_thread.pass = 2;
    int rc = 0;
    const int ss_ver = f->cur.p.layout == DAV1D_PIXEL_LAYOUT_I420;
    for (int tile_row = 0; tile_row < fh->tiling.rows && !rc; tile_row++)
        for (int sby = fh->tiling.row_start_sb[tile_row]; sby < fh->tiling.row_start_sb[tile_row + 1] && !rc; sby++) {
            t->by = sby << f->sb_shift;
            for (int tile_col = 0; tile_col < fh->tiling.cols && !rc; tile_col++) {
                t->ts = &f->ts[tile_row * fh->tiling.cols + tile_col];
                r->cur_tile_row = tile_row;
                memset(r->lf_l.tx_lpf_y, 2, sizeof(r->lf_l.tx_lpf_y));
                memset(r->lf_l.tx_lpf_uv, 1, sizeof(r->lf_l.tx_lpf_uv));
                rc = dav1d_decode_tile_sbrow(t);
                /* :2730-2740: the left context at the tile's right edge, for the mask fix-ups across tile columns */
                int align_h = (f->bh + 31) & ~31;
                memcpy(&f->lf.tx_lpf_right_edge[0][align_h * tile_col + t->by], &r->lf_l.tx_lpf_y[t->by & 16], f->sb_step);
                align_h >>= ss_ver;
                memcpy(&f->lf.tx_lpf_right_edge[1][align_h * tile_col + (t->by >> ss_ver)], &r->lf_l.tx_lpf_uv[(t->by & 16) >> ss_ver],
                       f->sb_step >> ss_ver);
            }
        }
    f->bd_fn.recon_b_intra = keep_intra;
    f->bd_fn.recon_b_inter = keep_inter;
    f->bd_fn.backup_ipred_edge = keep_edge;
    return rc;
}

/* the reference's in-loop filters, superblock row by superblock row (dav1d_decode_frame_main, src/decode.c:3228-3232) */
int dav1d_ref_frame_filter(void *const h) {
    RefFrame *const r = h;
    Dav1dFrameContext *const f = &r->f;
    r->tc->c = &r->c; r->tc->f = f;
    r->tc->top_pre_cdef_toggle = 0;
    for (int sby = 0; sby < f->sbh; sby++) f->bd_fn.filter_sbrow(f, sby);
    return 0;
}

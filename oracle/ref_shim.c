/* Shim linked INTO the oracle build of the reference (oracle/_ref/libdav1d_ref.so).
 * It initialises the reference's own Dav1dDSPContext (src/internal.h:62-70) through
 * the reference's dav1d_*_dsp_init_{8,16}bpc entry points (src/decode.c:3387-3415)
 * and hands out raw function pointers by (family, i, j) so that test drivers
 * (ctypes / C harnesses) can call the reference C functions without knowing the
 * struct layout.  TEST INFRASTRUCTURE ONLY. */
#include "config.h"
#include <string.h>
#include <stdint.h>
#include "src/internal.h"
#include "src/tables.h"
#include "src/wedge.h"
#include "src/scan.h"
#include "src/itx_1d.h"

static Dav1dDSPContext g_dsp[3];
static int g_init[3];

#define INIT(bd, idx, depth) \
    dav1d_cdef_dsp_init_##bd##bpc(&g_dsp[idx].cdef); \
    dav1d_intra_pred_dsp_init_##bd##bpc(&g_dsp[idx].ipred); \
    dav1d_itx_dsp_init_##bd##bpc(&g_dsp[idx].itx, depth); \
    dav1d_loop_filter_dsp_init_##bd##bpc(&g_dsp[idx].lf); \
    dav1d_loop_restoration_dsp_init_##bd##bpc(&g_dsp[idx].lr, depth); \
    dav1d_mc_dsp_init_##bd##bpc(&g_dsp[idx].mc); \
    dav1d_film_grain_dsp_init_##bd##bpc(&g_dsp[idx].fg)

static Dav1dDSPContext *ctx(const int bpc) {
    const int idx = bpc == 8 ? 0 : bpc == 10 ? 1 : 2;
    if (!g_init[idx]) {
        static int once;
        if (!once) { dav1d_init_ii_wedge_masks(); once = 1; }
        if (bpc == 8) { INIT(8, 0, 8); }
        else if (bpc == 10) { INIT(16, 1, 10); }
        else { INIT(16, 2, 12); }
        g_init[idx] = 1;
    }
    return &g_dsp[idx];
}

void *dav1d_ref_dsp_context(const int bpc) { return ctx(bpc); }

/* family names follow the struct member names of the reference tables */
void *dav1d_ref_dsp_entry(const int bpc, const char *const family, const int i, const int j) {
    Dav1dDSPContext *const c = ctx(bpc);
#define F(n) (!strcmp(family, n))
    if (F("itxfm_add"))    return (void *) c->itx.itxfm_add[i][j];
    if (F("mc"))           return (void *) c->mc.mc[i];
    if (F("mct"))          return (void *) c->mc.mct[i];
    if (F("mc_scaled"))    return (void *) c->mc.mc_scaled[i];
    if (F("mct_scaled"))   return (void *) c->mc.mct_scaled[i];
    if (F("avg"))          return (void *) c->mc.avg;
    if (F("w_avg"))        return (void *) c->mc.w_avg;
    if (F("mask"))         return (void *) c->mc.mask;
    if (F("w_mask"))       return (void *) c->mc.w_mask[i];
    if (F("blend"))        return (void *) c->mc.blend;
    if (F("blend_v"))      return (void *) c->mc.blend_v;
    if (F("blend_h"))      return (void *) c->mc.blend_h;
    if (F("warp8x8"))      return (void *) c->mc.warp8x8;
    if (F("warp8x8t"))     return (void *) c->mc.warp8x8t;
    if (F("emu_edge"))     return (void *) c->mc.emu_edge;
    if (F("resize"))       return (void *) c->mc.resize;
    if (F("intra_pred"))   return (void *) c->ipred.intra_pred[i];
    if (F("cfl_ac"))       return (void *) c->ipred.cfl_ac[i];
    if (F("cfl_pred"))     return (void *) c->ipred.cfl_pred[i];
    if (F("pal_pred"))     return (void *) c->ipred.pal_pred;
    if (F("loop_filter_sb")) return (void *) c->lf.loop_filter_sb[i][j];
    if (F("cdef_dir"))     return (void *) c->cdef.dir;
    if (F("cdef_fb"))      return (void *) c->cdef.fb[i];
    if (F("wiener"))       return (void *) c->lr.wiener[i];
    if (F("sgr"))          return (void *) c->lr.sgr[i];
    if (F("generate_grain_y"))  return (void *) c->fg.generate_grain_y;
    if (F("generate_grain_uv")) return (void *) c->fg.generate_grain_uv[i];
    if (F("fgy_32x32xn"))  return (void *) c->fg.fgy_32x32xn;
    if (F("fguv_32x32xn")) return (void *) c->fg.fguv_32x32xn[i];
#undef F
    return NULL;
}

/* constant tables the parity tests cross-check the product's device tables against */
const void *dav1d_ref_table(const char *const name, size_t *const sz) {
#define T(n, sym) if (!strcmp(name, n)) { if (sz) *sz = sizeof(sym); return sym; }
    T("mc_subpel_filters", dav1d_mc_subpel_filters)
    T("mc_warp_filter", dav1d_mc_warp_filter)
    T("resize_filter", dav1d_resize_filter)
    T("sm_weights", dav1d_sm_weights)
    T("dr_intra_derivative", dav1d_dr_intra_derivative)
    T("filter_intra_taps", dav1d_filter_intra_taps)
    T("obmc_masks", dav1d_obmc_masks)
    T("gaussian_sequence", dav1d_gaussian_sequence)
    T("cdef_directions", dav1d_cdef_directions)
    T("sgr_params", dav1d_sgr_params)
    T("sgr_x_by_x", dav1d_sgr_x_by_x)
    T("txfm_dimensions", dav1d_txfm_dimensions)
    T("block_dimensions", dav1d_block_dimensions)
    T("max_txfm_size_for_bs", dav1d_max_txfm_size_for_bs)
    T("block_sizes", dav1d_block_sizes)
#undef T
    return NULL;
}

/* wedge / inter-intra masks (src/wedge.h:33-80), built by dav1d_init_ii_wedge_masks */
const void *dav1d_ref_masks(size_t *const sz) {
    ctx(8);
    if (sz) *sz = sizeof(dav1d_masks);
    return &dav1d_masks;
}

/* hidden-visibility tables of the reference re-exported for the parity tests */
void *dav1d_ref_tx1d_fn(const int sz, const int kind) { return (void *) dav1d_tx1d_fns[sz][kind]; }
void *dav1d_ref_wht4_1d(void) { return (void *) dav1d_inv_wht4_1d_c; }
const uint16_t *dav1d_ref_scan(const int tx) { return dav1d_scans[tx]; }
const uint8_t *dav1d_ref_last_nonzero_col_from_eob(const int tx) {
    dav1d_init_last_nonzero_col_from_eob_tables();
    return dav1d_last_nonzero_col_from_eob[tx];
}

/* ---- film grain: the reference's whole dav1d_apply_grain on caller-provided planes ---- */
#include "src/fg_apply.h"
int dav1d_ref_apply_grain(const int bpc, const Dav1dFilmGrainData *const data, const int w, const int h,
                          const int layout, const int is_id, void *const out_data[3], void *const in_data[3],
                          const ptrdiff_t y_stride, const ptrdiff_t uv_stride)
{
    Dav1dPicture in, out;
    Dav1dSequenceHeader seq;
    Dav1dFrameHeader fh;
    memset(&in, 0, sizeof(in)); memset(&out, 0, sizeof(out)); memset(&seq, 0, sizeof(seq)); memset(&fh, 0, sizeof(fh));
    seq.mtrx = is_id ? DAV1D_MC_IDENTITY : DAV1D_MC_BT709;
    fh.film_grain.data = *data;
    in.p.w = out.p.w = w; in.p.h = out.p.h = h;
    in.p.layout = out.p.layout = layout; in.p.bpc = out.p.bpc = bpc;
    in.seq_hdr = out.seq_hdr = &seq; in.frame_hdr = out.frame_hdr = &fh;
    for (int i = 0; i < 3; i++) { in.data[i] = in_data[i]; out.data[i] = out_data[i]; }
    in.stride[0] = out.stride[0] = y_stride; in.stride[1] = out.stride[1] = uv_stride;
    if (bpc == 8) dav1d_apply_grain_8bpc(&ctx(8)->fg, &out, &in);
    else dav1d_apply_grain_16bpc(&ctx(bpc)->fg, &out, &in);
    return 0;
}

/* grain templates as int16 [3][73 + 1][82] */
int dav1d_ref_generate_grain(const int bpc, const Dav1dFilmGrainData *const data, const int layout, int16_t *const out)
{
    Dav1dDSPContext *const c = ctx(bpc);
    memset(out, 0, sizeof(int16_t) * 3 * 74 * 82);
    if (bpc == 8) {
        static int8_t lut[3][74][82];
        memset(lut, 0, sizeof(lut));
        ((void (*)(int8_t (*)[82], const Dav1dFilmGrainData *)) c->fg.generate_grain_y)(lut[0], data);
        if (layout)
            for (int pl = 0; pl < 2; pl++)
                if (data->num_uv_points[pl] || data->chroma_scaling_from_luma)
                    ((void (*)(int8_t (*)[82], const int8_t (*)[82], const Dav1dFilmGrainData *, intptr_t))
                         c->fg.generate_grain_uv[layout - 1])(lut[1 + pl], (const int8_t (*)[82]) lut[0], data, pl);
        for (int i = 0; i < 3 * 74 * 82; i++) out[i] = ((int8_t *) lut)[i];
    } else {
        const int bdmax = (1 << bpc) - 1;
        int16_t (*lut)[74][82] = (int16_t (*)[74][82]) out;
        ((void (*)(int16_t (*)[82], const Dav1dFilmGrainData *, int)) c->fg.generate_grain_y)(lut[0], data, bdmax);
        if (layout)
            for (int pl = 0; pl < 2; pl++)
                if (data->num_uv_points[pl] || data->chroma_scaling_from_luma)
                    ((void (*)(int16_t (*)[82], const int16_t (*)[82], const Dav1dFilmGrainData *, intptr_t, int))
                         c->fg.generate_grain_uv[layout - 1])(lut[1 + pl], (const int16_t (*)[82]) lut[0], data, pl, bdmax);
    }
    return 0;
}

/* Dav1dRefmvsDSPContext of the reference build (C path): save_tmvs and splat_mv through thin calls */
#include "src/refmvs.h"
void dav1d_ref_refmvs_splat(refmvs_block **rr, const refmvs_block *rmv, int bx4, int bw4, int bh4) {
    Dav1dRefmvsDSPContext c;
    dav1d_refmvs_dsp_init(&c);
    c.splat_mv(rr, rmv, bx4, bw4, bh4);
}
void dav1d_ref_refmvs_save_tmvs(refmvs_temporal_block *rp, ptrdiff_t stride, refmvs_block *const *rr, const uint8_t *ref_sign,
                                int col_end8, int row_end8, int col_start8, int row_start8) {
    Dav1dRefmvsDSPContext c;
    dav1d_refmvs_dsp_init(&c);
    c.save_tmvs(rp, stride, rr, ref_sign, col_end8, row_end8, col_start8, row_start8);
}

// Reference-signature wrappers for the film grain, intra prediction, loop filter, CDEF and loop restoration
// tables (reference src/filmgrain.h:46-80, src/ipred.h:44-90, src/loopfilter.h:38-53, src/cdef.h:44-67,
// src/looprestoration.h:57-75).  One call = stage the host rectangles the C entry would touch, run the batched
// kernel on one task, copy back.  Parity aid for unmodified call sites -- the batched API is the fast path.
#include "dsp_stage.h"
#include <vector>

namespace {

using namespace dsp_stage;
#define LOCK std::lock_guard<std::mutex> lk(dsp_stage::mutex())

// ------------------------------------------------------------------------------------------- ipred

// intra_pred[mode]: the prepared edge array goes to the aux arena, the kernel skips its own edge preparation
template <typename pixel>
void ipred_call(int mode, pixel *dst, ptrdiff_t stride, const pixel *topleft, int w, int h, int angle, int max_w, int max_h, int bpc) {
    LOCK;
    Stage s(1 << 20);
    Dav1dHipPicture out = scratch_pic(s, w, h, bpc);
    const int m = w < h ? w : h, lo = h + m, hi = w + m;
    pixel *edge = (pixel *) s.take(sizeof(pixel) * (lo + hi + 1));
    hipMemcpyAsync(edge, topleft - lo, sizeof(pixel) * (lo + hi + 1), hipMemcpyHostToDevice, s.c->stream);
    Dav1dHipIpredTask t;
    memset(&t, 0, sizeof(t));
    t.kind = DAV1D_HIP_IPRED_DSP; t.mode = mode; t.tw = w >> 2; t.th = h >> 2;
    t.aux_off = lo; t.pal[0] = (uint16_t) angle; t.max_w = max_w; t.max_h = max_h;
    if (dav1d_hip_ipred_batch(s.c, &out, &t, 1, (uint8_t *) edge)) abort();
    down2d(s.c, dst, stride, out.p[0].data, out.p[0].stride, w * sizeof(pixel), h);
    s.sync();
}
template <int M> void ipred8(uint8_t *d, ptrdiff_t st, const uint8_t *tl, int w, int h, int a, int mw, int mh) { ipred_call<uint8_t>(M, d, st, tl, w, h, a, mw, mh, 8); }
template <int M> void ipred16(uint16_t *d, ptrdiff_t st, const uint16_t *tl, int w, int h, int a, int mw, int mh, int bm) {
    ipred_call<uint16_t>(M, d, st, tl, w, h, a, mw, mh, bpc_of(bm));
}

// cfl_ac[layout - 1]
template <typename pixel>
void cfl_ac_call(int layout, int16_t *ac, const pixel *ypx, ptrdiff_t stride, int w_pad, int h_pad, int cw, int ch, int bpc) {
    LOCK;
    Stage s(1 << 20);
    const int ss_hor = layout != DAV1D_HIP_LAYOUT_I444, ss_ver = layout == DAV1D_HIP_LAYOUT_I420;
    const int lw = (cw - 4 * w_pad) << ss_hor, lh = (ch - 4 * h_pad) << ss_ver;      // the luma pixels the entry reads
    Dav1dHipPicture pic = empty_pic(bpc, layout);
    add_plane(s, pic, 0, lw, lh);
    add_plane(s, pic, 1, cw, ch);
    up2d(s.c, pic.p[0].data, pic.p[0].stride, ypx, stride, lw * sizeof(pixel), lh);
    int16_t *dac = (int16_t *) s.take(sizeof(int16_t) * cw * ch);
    Dav1dHipIpredTask t;
    memset(&t, 0, sizeof(t));
    t.kind = DAV1D_HIP_IPRED_DSP_CFL_AC; t.plane = 1; t.tw = cw >> 2; t.th = ch >> 2; t.max_w = w_pad; t.max_h = h_pad;
    if (dav1d_hip_ipred_batch(s.c, &pic, &t, 1, (uint8_t *) dac)) abort();
    hipMemcpyAsync(ac, dac, sizeof(int16_t) * cw * ch, hipMemcpyDeviceToHost, s.c->stream);
    s.sync();
}
template <int L> void cfl_ac8(int16_t *ac, const uint8_t *y, ptrdiff_t st, int wp, int hp, int cw, int ch) { cfl_ac_call<uint8_t>(L, ac, y, st, wp, hp, cw, ch, 8); }
template <int L> void cfl_ac16(int16_t *ac, const uint16_t *y, ptrdiff_t st, int wp, int hp, int cw, int ch) { cfl_ac_call<uint16_t>(L, ac, y, st, wp, hp, cw, ch, 10); }

// cfl_pred[mode]
template <typename pixel>
void cfl_pred_call(int mode, pixel *dst, ptrdiff_t stride, const pixel *topleft, int w, int h, const int16_t *ac, int alpha, int bpc) {
    LOCK;
    Stage s(1 << 20);
    Dav1dHipPicture out = scratch_pic(s, w, h, bpc);
    const int m = w < h ? w : h, lo = h + m, hi = w + m;
    // arena: [ac: w*h int16][edge]
    int16_t *dac = (int16_t *) s.take(sizeof(int16_t) * w * h + sizeof(pixel) * (lo + hi + 1) + 16);
    pixel *edge = (pixel *) (dac + w * h);
    hipMemcpyAsync(dac, ac, sizeof(int16_t) * w * h, hipMemcpyHostToDevice, s.c->stream);
    // only the left column / top row a DC flavour averages is guaranteed to exist
    const bool need_left = mode == 0 || mode == 3, need_top = mode == 0 || mode == 4;
    if (need_left) hipMemcpyAsync(edge + lo - h, topleft - h, sizeof(pixel) * h, hipMemcpyHostToDevice, s.c->stream);
    if (need_top) hipMemcpyAsync(edge + lo + 1, topleft + 1, sizeof(pixel) * w, hipMemcpyHostToDevice, s.c->stream);
    Dav1dHipIpredTask t;
    memset(&t, 0, sizeof(t));
    t.kind = DAV1D_HIP_IPRED_DSP_CFL_PRED; t.mode = mode; t.tw = w >> 2; t.th = h >> 2; t.angle = (int8_t) alpha;
    t.aux_off = (uint32_t) (((uint8_t *) edge - (uint8_t *) dac) / sizeof(pixel)) + lo;
    if (dav1d_hip_ipred_batch(s.c, &out, &t, 1, (uint8_t *) dac)) abort();
    down2d(s.c, dst, stride, out.p[0].data, out.p[0].stride, w * sizeof(pixel), h);
    s.sync();
}
template <int M> void cfl_pred8(uint8_t *d, ptrdiff_t st, const uint8_t *tl, int w, int h, const int16_t *ac, int alpha) { cfl_pred_call<uint8_t>(M, d, st, tl, w, h, ac, alpha, 8); }
template <int M> void cfl_pred16(uint16_t *d, ptrdiff_t st, const uint16_t *tl, int w, int h, const int16_t *ac, int alpha, int bm) {
    cfl_pred_call<uint16_t>(M, d, st, tl, w, h, ac, alpha, bpc_of(bm));
}

template <typename pixel>
void pal_pred_call(pixel *dst, ptrdiff_t stride, const pixel *pal, const uint8_t *idx, int w, int h, int bpc) {
    LOCK;
    Stage s(1 << 20);
    Dav1dHipPicture out = scratch_pic(s, w, h, bpc);
    uint8_t *di = (uint8_t *) s.take((size_t) w * h / 2);
    hipMemcpyAsync(di, idx, (size_t) w * h / 2, hipMemcpyHostToDevice, s.c->stream);
    Dav1dHipIpredTask t;
    memset(&t, 0, sizeof(t));
    t.kind = DAV1D_HIP_IPRED_PAL; t.tw = w >> 2; t.th = h >> 2;
    for (int i = 0; i < 8; i++) t.pal[i] = pal[i];
    if (dav1d_hip_ipred_batch(s.c, &out, &t, 1, di)) abort();
    down2d(s.c, dst, stride, out.p[0].data, out.p[0].stride, w * sizeof(pixel), h);
    s.sync();
}
void pal_pred8(uint8_t *d, ptrdiff_t st, const uint8_t *pal, const uint8_t *idx, int w, int h) { pal_pred_call<uint8_t>(d, st, pal, idx, w, h, 8); }
void pal_pred16(uint16_t *d, ptrdiff_t st, const uint16_t *pal, const uint8_t *idx, int w, int h) { pal_pred_call<uint16_t>(d, st, pal, idx, w, h, 10); }

// ------------------------------------------------------------------------------------- loop filter

// loop_filter_sb[plane != 0][dir]: units = set bits of the masks; each unit is 4 lines along the edge
template <typename pixel>
void lf_call(int chroma, int dir, pixel *dst, ptrdiff_t stride, const uint32_t *mask, const uint8_t (*lvl)[4], ptrdiff_t lvl_stride,
             const Dav1dHipFilterLUT *lut, int bpc) {
    LOCK;
    const uint32_t vm = mask[0] | mask[1] | (chroma ? 0u : mask[2]);
    if (!vm) return;
    const int units = 32 - __builtin_clz(vm);
    // how far the widest filter in play reaches on each side of the edge (src/loopfilter_tmpl.c:37-161)
    // (edges between columns: the kernel fetches the taps of a line as one 8-pixel or two 8-pixel pieces around the edge)
    const int reach = dir ? (chroma ? (mask[1] ? 3 : 2) : (mask[2] ? 7 : mask[1] ? 4 : 2)) : ((!chroma && mask[2]) ? 8 : 4);
    const int along = 4 * units, across = 2 * reach;
    Stage s(1 << 20);
    Dav1dHipPicture pic = dir ? scratch_pic(s, along, across, bpc) : scratch_pic(s, across, along, bpc);
    const ptrdiff_t sp = stride / (ptrdiff_t) sizeof(pixel);
    pixel *const org = dir ? dst - reach * sp : dst - reach;
    up2d(s.c, pic.p[0].data, pic.p[0].stride, org, stride, (size_t) pic.p[0].w * sizeof(pixel), pic.p[0].h);
    // levels: the unit's own entry and its neighbour across the edge, component 0 of what the caller points at
    std::vector<uint8_t> hl((size_t) 2 * 32 * 4, 0);
    for (int u = 0; u < units; u++) {
        if (!(vm & (1u << u))) continue;
        // the neighbour's level is only consulted (and only guaranteed to exist) when the unit's own level is 0
        const uint8_t (*own)[4] = dir ? lvl + u : lvl + u * lvl_stride;
        const int L = own[0][0] ? own[0][0] : (dir ? own[-lvl_stride][0] : own[-1][0]);
        hl[(dir ? 32 + u : 2 * u + 1) * 4] = (uint8_t) L;
    }
    uint8_t *dl = (uint8_t *) s.take(hl.size());
    hipMemcpyAsync(dl, hl.data(), hl.size(), hipMemcpyHostToDevice, s.c->stream);
    Dav1dHipLfTask t;
    memset(&t, 0, sizeof(t));
    t.dst_off = dir ? (uint32_t) (reach * (pic.p[0].stride / (ptrdiff_t) sizeof(pixel))) : (uint32_t) reach;
    t.lvl_off = dir ? 32 : 1;
    t.vmask[0] = mask[0]; t.vmask[1] = mask[1]; t.vmask[2] = chroma ? 0 : mask[2];
    t.plane = 0; t.dir = dir; t.lvl_comp = 0;
    // the kernel tells luma from chroma by the plane index: chroma calls run on plane 1 of a two-plane picture
    Dav1dHipPicture run = pic;
    if (chroma) { run.layout = DAV1D_HIP_LAYOUT_I444; run.p[1] = pic.p[0]; run.p[2] = pic.p[0]; t.plane = 1; }
    if (dav1d_hip_lf_batch(s.c, &run, &t, 1, dl, dir ? 32 : 2, lut->e, lut->i)) abort();
    down2d(s.c, org, stride, pic.p[0].data, pic.p[0].stride, (size_t) pic.p[0].w * sizeof(pixel), pic.p[0].h);
    s.sync();
}
template <int C, int D> void lf8(uint8_t *d, ptrdiff_t st, const uint32_t *m, const uint8_t (*l)[4], ptrdiff_t ls, const Dav1dHipFilterLUT *lut, int) {
    lf_call<uint8_t>(C, D, d, st, m, l, ls, lut, 8);
}
template <int C, int D> void lf16(uint16_t *d, ptrdiff_t st, const uint32_t *m, const uint8_t (*l)[4], ptrdiff_t ls, const Dav1dHipFilterLUT *lut, int, int bm) {
    lf_call<uint16_t>(C, D, d, st, m, l, ls, lut, bpc_of(bm));
}

// -------------------------------------------------------------------------------------------- cdef

template <typename pixel>
int cdef_dir_call(const pixel *dst, ptrdiff_t stride, unsigned *var, int bpc) {
    LOCK;
    Stage s(1 << 20);
    Dav1dHipPicture pic = scratch_pic(s, 8, 8, bpc);
    up2d(s.c, pic.p[0].data, pic.p[0].stride, dst, stride, 8 * sizeof(pixel), 8);
    uint32_t *dv = (uint32_t *) s.take(sizeof(uint32_t));
    Dav1dHipCdefTask t;
    memset(&t, 0, sizeof(t));      // no strengths: the kernel only runs the direction search
    if (dav1d_hip_cdef_batch(s.c, &pic, &pic, &t, 1, 3 + (bpc - 8), dv)) abort();
    uint32_t r = 0;
    hipMemcpyAsync(&r, dv, sizeof(r), hipMemcpyDeviceToHost, s.c->stream);
    s.sync();
    *var = r >> 3;
    return (int) (r & 7);
}
int cdef_dir8(const uint8_t *d, ptrdiff_t st, unsigned *var) { return cdef_dir_call<uint8_t>(d, st, var, 8); }
int cdef_dir16(const uint16_t *d, ptrdiff_t st, unsigned *var, int bm) { return cdef_dir_call<uint16_t>(d, st, var, bpc_of(bm)); }

// fb[0] 8x8, fb[1] 4x8, fb[2] 4x4: the block plus its 2-pixel frame assembled from dst / left / top / bottom
template <typename pixel>
void cdef_fb_call(int w, int h, pixel *dst, ptrdiff_t stride, const pixel (*left)[2], const pixel *top, const pixel *bottom,
                  int pri, int sec, int dir, int damping, int edges, int bpc) {
    LOCK;
    Stage s(1 << 20);
    Dav1dHipPicture in = scratch_pic(s, w + 4, h + 4, bpc), out = scratch_pic(s, w + 4, h + 4, bpc);
    const bool hl = edges & DAV1D_HIP_CDEF_HAVE_LEFT, hr = edges & DAV1D_HIP_CDEF_HAVE_RIGHT;
    const int x0 = hl ? -2 : 0, x1 = w + (hr ? 2 : 0);
    up2d(s.c, px<pixel>(in, 0, 2, 2), in.p[0].stride, dst, stride, (size_t) (hr ? w + 2 : w) * sizeof(pixel), h);
    if (hl) up2d(s.c, px<pixel>(in, 0, 0, 2), in.p[0].stride, left, 2 * sizeof(pixel), 2 * sizeof(pixel), h);
    if (edges & DAV1D_HIP_CDEF_HAVE_TOP) up2d(s.c, px<pixel>(in, 0, 2 + x0, 0), in.p[0].stride, top + x0, stride, (size_t) (x1 - x0) * sizeof(pixel), 2);
    if (edges & DAV1D_HIP_CDEF_HAVE_BOTTOM) up2d(s.c, px<pixel>(in, 0, 2 + x0, h + 2), in.p[0].stride, bottom + x0, stride, (size_t) (x1 - x0) * sizeof(pixel), 2);
    Dav1dHipCdefTask t;
    memset(&t, 0, sizeof(t));
    t.bx = 2; t.by = 2; t.y_pri = (uint8_t) pri; t.y_sec = (uint8_t) sec; t.edges = (uint8_t) edges; t.dir = (uint8_t) dir;
    t.flags = 1 | (w == 4 ? 2 : 0) | (h == 4 ? 4 : 0);
    up2d(s.c, px<pixel>(out, 0, 2, 2), out.p[0].stride, dst, stride, w * sizeof(pixel), h);      // zero strengths leave the block as is
    if (dav1d_hip_cdef_batch(s.c, &out, &in, &t, 1, damping, nullptr)) abort();
    down2d(s.c, dst, stride, px<pixel>(out, 0, 2, 2), out.p[0].stride, w * sizeof(pixel), h);
    s.sync();
}
template <int W, int H> void cdef_fb8(uint8_t *d, ptrdiff_t st, const uint8_t (*l)[2], const uint8_t *t, const uint8_t *b, int pri, int sec, int dir, int damp, int e) {
    cdef_fb_call<uint8_t>(W, H, d, st, l, t, b, pri, sec, dir, damp, e, 8);
}
template <int W, int H> void cdef_fb16(uint16_t *d, ptrdiff_t st, const uint16_t (*l)[2], const uint16_t *t, const uint16_t *b, int pri, int sec, int dir, int damp, int e, int bm) {
    cdef_fb_call<uint16_t>(W, H, d, st, l, t, b, pri, sec, dir, damp, e, bpc_of(bm));
}

// ------------------------------------------------------------------------------- loop restoration

// The stripe sits at (4, 2) of three staged pictures: `in` = the unit with its left / right neighbours,
// `lpf` = the two rows above (rows 0, 1) and below (rows h + 2, h + 3), `out` = result
template <typename pixel>
void lr_call(int type, pixel *dst, ptrdiff_t stride, const pixel (*left)[4], const pixel *lpf, int w, int h,
             const Dav1dHipLrParams *params, int edges, int bpc) {
    LOCK;
    Stage s(1 << 20);
    const int W = w + 8, H = h + 4;
    Dav1dHipPicture in = scratch_pic(s, W, H, bpc), lp = scratch_pic(s, W, H, bpc), out = scratch_pic(s, W, H, bpc);
    const bool hl = edges & DAV1D_HIP_LR_HAVE_LEFT, hr = edges & DAV1D_HIP_LR_HAVE_RIGHT;
    const int x0 = hl ? -3 : 0, x1 = w + (hr ? 3 : 0);
    const ptrdiff_t sp = stride / (ptrdiff_t) sizeof(pixel);
    up2d(s.c, px<pixel>(in, 0, 4, 2), in.p[0].stride, dst, stride, (size_t) x1 * sizeof(pixel), h);
    if (hl) up2d(s.c, px<pixel>(in, 0, 0, 2), in.p[0].stride, left, 4 * sizeof(pixel), 4 * sizeof(pixel), h);
    if (edges & DAV1D_HIP_LR_HAVE_TOP) up2d(s.c, px<pixel>(lp, 0, 4 + x0, 0), lp.p[0].stride, lpf + x0, stride, (size_t) (x1 - x0) * sizeof(pixel), 2);
    if (edges & DAV1D_HIP_LR_HAVE_BOTTOM)
        up2d(s.c, px<pixel>(lp, 0, 4 + x0, h + 2), lp.p[0].stride, lpf + 6 * sp + x0, stride, (size_t) (x1 - x0) * sizeof(pixel), 2);
    Dav1dHipLrTask t;
    memset(&t, 0, sizeof(t));
    t.x = 4; t.y = 2; t.w = w; t.h = h; t.edges = (uint8_t) edges; t.type = (uint8_t) type;
    if (type <= DAV1D_HIP_LR_WIENER5) memcpy(t.filter, params->filter, sizeof(t.filter));
    else {
        // SGR parameters travel in filter[0][0..3] = s0, s1, w0, w1
        t.filter[0][0] = (int16_t) params->sgr.s0; t.filter[0][1] = (int16_t) params->sgr.s1;
        t.filter[0][2] = params->sgr.w0; t.filter[0][3] = params->sgr.w1;
    }
    if (dav1d_hip_lr_batch(s.c, &out, &in, &lp, &t, 1)) abort();
    down2d(s.c, dst, stride, px<pixel>(out, 0, 4, 2), out.p[0].stride, w * sizeof(pixel), h);
    s.sync();
}
template <int T> void lr8(uint8_t *d, ptrdiff_t st, const uint8_t (*l)[4], const uint8_t *lpf, int w, int h, const Dav1dHipLrParams *p, int e) {
    lr_call<uint8_t>(T, d, st, l, lpf, w, h, p, e, 8);
}
template <int T> void lr16(uint16_t *d, ptrdiff_t st, const uint16_t (*l)[4], const uint16_t *lpf, int w, int h, const Dav1dHipLrParams *p, int e, int bm) {
    lr_call<uint16_t>(T, d, st, l, lpf, w, h, p, e, bpc_of(bm));
}

// -------------------------------------------------------------------------------------- film grain

enum { GW = DAV1D_HIP_GRAIN_WIDTH, GH = 73, LUT_ELEMS = (GH + 1) * GW };

// widen / narrow between the caller's grain entries (int8_t at 8 bpc) and the device templates (int16_t)
template <typename entry> void lut_up(Stage &s, int16_t *dev, const entry (*host)[GW], int rows) {
    std::vector<int16_t> tmp((size_t) rows * GW);
    for (int y = 0; y < rows; y++) for (int x = 0; x < GW; x++) tmp[(size_t) y * GW + x] = host[y][x];
    hipMemcpyAsync(dev, tmp.data(), tmp.size() * sizeof(int16_t), hipMemcpyHostToDevice, s.c->stream);
    s.sync();
}
template <typename entry> void lut_down(Stage &s, entry (*host)[GW], const int16_t *dev, int rows, int cols) {
    std::vector<int16_t> tmp((size_t) rows * GW);
    hipMemcpyAsync(tmp.data(), dev, tmp.size() * sizeof(int16_t), hipMemcpyDeviceToHost, s.c->stream);
    s.sync();
    for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) host[y][x] = (entry) tmp[(size_t) y * GW + x];
}

// generate_grain_y (uv < 0) / generate_grain_uv[layout - 1]
template <typename entry>
void gen_grain_call(int layout, entry (*buf)[GW], const entry (*buf_y)[GW], const Dav1dHipFilmGrainData *data, int uv, int bpc) {
    LOCK;
    Stage s(1 << 20);
    int16_t *luts = (int16_t *) s.take_zero(3 * LUT_ELEMS * sizeof(int16_t));
    if (uv >= 0) lut_up<entry>(s, luts, buf_y, GH);
    if (dav1d_hip_launch_fg_gen_part(luts, data, bpc, layout, uv < 0 ? -1 : 1 + uv, s.c->stream)) abort();
    const int subx = uv >= 0 && layout != DAV1D_HIP_LAYOUT_I444, suby = uv >= 0 && layout == DAV1D_HIP_LAYOUT_I420;
    lut_down<entry>(s, buf, luts + (uv < 0 ? 0 : 1 + uv) * LUT_ELEMS, suby ? 38 : GH, subx ? 44 : GW);
}
void gen_y8(int8_t buf[][GW], const Dav1dHipFilmGrainData *d) { gen_grain_call<int8_t>(DAV1D_HIP_LAYOUT_I444, buf, nullptr, d, -1, 8); }
void gen_y16(int16_t buf[][GW], const Dav1dHipFilmGrainData *d, int bm) { gen_grain_call<int16_t>(DAV1D_HIP_LAYOUT_I444, buf, nullptr, d, -1, bpc_of(bm)); }
template <int L> void gen_uv8(int8_t buf[][GW], const int8_t buf_y[][GW], const Dav1dHipFilmGrainData *d, intptr_t uv) { gen_grain_call<int8_t>(L, buf, buf_y, d, (int) uv, 8); }
template <int L> void gen_uv16(int16_t buf[][GW], const int16_t buf_y[][GW], const Dav1dHipFilmGrainData *d, intptr_t uv, int bm) {
    gen_grain_call<int16_t>(L, buf, buf_y, d, (int) uv, bpc_of(bm));
}

// fgy_32x32xn (pl == 0) / fguv_32x32xn[layout - 1] (pl == 1 + uv_pl): one row of 32x32 blocks
template <typename pixel, typename entry>
void fg_row_call(int layout, int pl, pixel *dst_row, const pixel *src_row, ptrdiff_t stride, const Dav1dHipFilmGrainData *data, size_t pw,
                 const uint8_t *scaling, const entry (*grain_lut)[GW], int bh, int row_num, const pixel *luma_row, ptrdiff_t luma_stride,
                 int is_id, int bpc) {
    LOCK;
    const int sx = pl && layout != DAV1D_HIP_LAYOUT_I444, sy = pl && layout == DAV1D_HIP_LAYOUT_I420;
    const int lw = (int) pw << sx, lh = bh << sy, scaling_size = 1 << bpc;
    Stage s((size_t) 4 * (lw + 64) * (lh + 2) * sizeof(pixel) + 3 * LUT_ELEMS * sizeof(int16_t) + 3 * (size_t) scaling_size + (1 << 16));
    Dav1dHipPicture in = empty_pic(bpc, pl ? layout : DAV1D_HIP_LAYOUT_I400), out = in;
    add_plane(s, in, 0, lw, lh);
    out.p[0] = in.p[0];
    if (pl) {
        add_plane(s, in, pl, (int) pw, bh);
        add_plane(s, out, pl, (int) pw, bh);
        in.p[3 - pl] = in.p[pl]; out.p[3 - pl] = out.p[pl];
        up2d(s.c, in.p[0].data, in.p[0].stride, luma_row, luma_stride, (size_t) lw * sizeof(pixel), lh);
        up2d(s.c, in.p[pl].data, in.p[pl].stride, src_row, stride, pw * sizeof(pixel), bh);
    } else {
        add_plane(s, out, 0, lw, lh);
        up2d(s.c, in.p[0].data, in.p[0].stride, src_row, stride, pw * sizeof(pixel), bh);
    }
    int16_t *luts = (int16_t *) s.take_zero(3 * LUT_ELEMS * sizeof(int16_t));
    uint8_t *sc = (uint8_t *) s.take(3 * (size_t) scaling_size);
    lut_up<entry>(s, luts + pl * LUT_ELEMS, grain_lut, GH);
    // the kernel reads the luma table for chroma_scaling_from_luma, else the plane's own
    hipMemcpyAsync(sc + (size_t) ((pl && !data->chroma_scaling_from_luma) ? pl : 0) * scaling_size, scaling, scaling_size, hipMemcpyHostToDevice, s.c->stream);
    // the entry runs whenever it is called: the points that gate it in dav1d_apply_grain are forced on
    Dav1dHipFilmGrainData d = *data;
    if (!pl) { if (!d.num_y_points) d.num_y_points = 1; }
    else if (!d.num_uv_points[pl - 1] && !d.chroma_scaling_from_luma) d.num_uv_points[pl - 1] = 1;
    const DevPlanes dp = dev_planes(&out), sp = dev_planes(&in);
    uint8_t *offs = (uint8_t *) s.take(2 * (size_t) ((lw + 31) / 32) + 16);
    if (dav1d_hip_launch_fg_apply_rows(&dp, &sp, luts, sc, scaling_size, &d, bpc, in.layout, is_id, row_num, pl, offs, s.c->stream)) abort();
    down2d(s.c, dst_row, stride, out.p[pl].data, out.p[pl].stride, pw * sizeof(pixel), bh);
    s.sync();
}
void fgy8(uint8_t *d, const uint8_t *s, ptrdiff_t st, const Dav1dHipFilmGrainData *data, size_t pw, const uint8_t *sc, const int8_t lut[][GW], int bh, int row) {
    fg_row_call<uint8_t, int8_t>(DAV1D_HIP_LAYOUT_I400, 0, d, s, st, data, pw, sc, lut, bh, row, nullptr, 0, 0, 8);
}
void fgy16(uint16_t *d, const uint16_t *s, ptrdiff_t st, const Dav1dHipFilmGrainData *data, size_t pw, const uint8_t *sc, const int16_t lut[][GW], int bh, int row, int bm) {
    fg_row_call<uint16_t, int16_t>(DAV1D_HIP_LAYOUT_I400, 0, d, s, st, data, pw, sc, lut, bh, row, nullptr, 0, 0, bpc_of(bm));
}
template <int L> void fguv8(uint8_t *d, const uint8_t *s, ptrdiff_t st, const Dav1dHipFilmGrainData *data, size_t pw, const uint8_t *sc, const int8_t lut[][GW],
                            int bh, int row, const uint8_t *luma, ptrdiff_t ls, int uv, int is_id) {
    fg_row_call<uint8_t, int8_t>(L, 1 + uv, d, s, st, data, pw, sc, lut, bh, row, luma, ls, is_id, 8);
}
template <int L> void fguv16(uint16_t *d, const uint16_t *s, ptrdiff_t st, const Dav1dHipFilmGrainData *data, size_t pw, const uint8_t *sc, const int16_t lut[][GW],
                             int bh, int row, const uint16_t *luma, ptrdiff_t ls, int uv, int is_id, int bm) {
    fg_row_call<uint16_t, int16_t>(L, 1 + uv, d, s, st, data, pw, sc, lut, bh, row, luma, ls, is_id, bpc_of(bm));
}


} // namespace

void dav1d_hip_dsp_fill_post_8(Dav1dHipDSPContext8 *c) {
    c->fg.generate_grain_y = gen_y8;
    c->fg.generate_grain_uv[0] = gen_uv8<1>; c->fg.generate_grain_uv[1] = gen_uv8<2>; c->fg.generate_grain_uv[2] = gen_uv8<3>;
    c->fg.fgy_32x32xn = fgy8;
    c->fg.fguv_32x32xn[0] = fguv8<1>; c->fg.fguv_32x32xn[1] = fguv8<2>; c->fg.fguv_32x32xn[2] = fguv8<3>;
    dav1d_hip_angular_ipred_fn8 ip[14] = { ipred8<0>, ipred8<1>, ipred8<2>, ipred8<3>, ipred8<4>, ipred8<5>, ipred8<6>, ipred8<7>, ipred8<8>,
                                           ipred8<9>, ipred8<10>, ipred8<11>, ipred8<12>, ipred8<13> };
    for (int i = 0; i < 14; i++) c->ipred.intra_pred[i] = ip[i];
    c->ipred.cfl_ac[0] = cfl_ac8<1>; c->ipred.cfl_ac[1] = cfl_ac8<2>; c->ipred.cfl_ac[2] = cfl_ac8<3>;
    c->ipred.cfl_pred[0] = cfl_pred8<0>; c->ipred.cfl_pred[3] = cfl_pred8<3>; c->ipred.cfl_pred[4] = cfl_pred8<4>; c->ipred.cfl_pred[5] = cfl_pred8<5>;
    c->ipred.pal_pred = pal_pred8;
    c->lf.loop_filter_sb[0][0] = lf8<0, 0>; c->lf.loop_filter_sb[0][1] = lf8<0, 1>;
    c->lf.loop_filter_sb[1][0] = lf8<1, 0>; c->lf.loop_filter_sb[1][1] = lf8<1, 1>;
    c->cdef.dir = cdef_dir8;
    c->cdef.fb[0] = cdef_fb8<8, 8>; c->cdef.fb[1] = cdef_fb8<4, 8>; c->cdef.fb[2] = cdef_fb8<4, 4>;
    c->lr.wiener[0] = lr8<DAV1D_HIP_LR_WIENER7>; c->lr.wiener[1] = lr8<DAV1D_HIP_LR_WIENER5>;
    c->lr.sgr[0] = lr8<DAV1D_HIP_LR_SGR_5X5>; c->lr.sgr[1] = lr8<DAV1D_HIP_LR_SGR_3X3>; c->lr.sgr[2] = lr8<DAV1D_HIP_LR_SGR_MIX>;
}

void dav1d_hip_dsp_fill_post_16(Dav1dHipDSPContext16 *c) {
    c->fg.generate_grain_y = gen_y16;
    c->fg.generate_grain_uv[0] = gen_uv16<1>; c->fg.generate_grain_uv[1] = gen_uv16<2>; c->fg.generate_grain_uv[2] = gen_uv16<3>;
    c->fg.fgy_32x32xn = fgy16;
    c->fg.fguv_32x32xn[0] = fguv16<1>; c->fg.fguv_32x32xn[1] = fguv16<2>; c->fg.fguv_32x32xn[2] = fguv16<3>;
    dav1d_hip_angular_ipred_fn16 ip[14] = { ipred16<0>, ipred16<1>, ipred16<2>, ipred16<3>, ipred16<4>, ipred16<5>, ipred16<6>, ipred16<7>, ipred16<8>,
                                            ipred16<9>, ipred16<10>, ipred16<11>, ipred16<12>, ipred16<13> };
    for (int i = 0; i < 14; i++) c->ipred.intra_pred[i] = ip[i];
    c->ipred.cfl_ac[0] = cfl_ac16<1>; c->ipred.cfl_ac[1] = cfl_ac16<2>; c->ipred.cfl_ac[2] = cfl_ac16<3>;
    c->ipred.cfl_pred[0] = cfl_pred16<0>; c->ipred.cfl_pred[3] = cfl_pred16<3>; c->ipred.cfl_pred[4] = cfl_pred16<4>; c->ipred.cfl_pred[5] = cfl_pred16<5>;
    c->ipred.pal_pred = pal_pred16;
    c->lf.loop_filter_sb[0][0] = lf16<0, 0>; c->lf.loop_filter_sb[0][1] = lf16<0, 1>;
    c->lf.loop_filter_sb[1][0] = lf16<1, 0>; c->lf.loop_filter_sb[1][1] = lf16<1, 1>;
    c->cdef.dir = cdef_dir16;
    c->cdef.fb[0] = cdef_fb16<8, 8>; c->cdef.fb[1] = cdef_fb16<4, 8>; c->cdef.fb[2] = cdef_fb16<4, 4>;
    c->lr.wiener[0] = lr16<DAV1D_HIP_LR_WIENER7>; c->lr.wiener[1] = lr16<DAV1D_HIP_LR_WIENER5>;
    c->lr.sgr[0] = lr16<DAV1D_HIP_LR_SGR_5X5>; c->lr.sgr[1] = lr16<DAV1D_HIP_LR_SGR_3X3>; c->lr.sgr[2] = lr16<DAV1D_HIP_LR_SGR_MIX>;
}

/* oracle/port — CPU restatement of intra prediction.  TEST INFRASTRUCTURE ONLY.
 *
 * ipred.intra_pred[0..13] (reference src/ipred_tmpl.c:93-655), cfl_ac[3] (:657-715), cfl_pred[*] (:71-84, 103-218),
 * pal_pred (:717-730).  `topleft` is the edge array dav1d_prepare_intra_edges produces (src/ipred_prepare.h:57-73):
 * E[0] top-left, E[1..] the row above (and its extension), E[-1..] the left column downwards (and its extension). */
#include "port.h"

enum { M_DC, M_VERT, M_HOR, M_LEFT_DC, M_TOP_DC, M_DC_128, M_Z1, M_Z2, M_Z3, M_SMOOTH, M_SMOOTH_V, M_SMOOTH_H, M_PAETH, M_FILTER };

static int ctz(int v) { int n = 0; while (!(v & 1)) { v >>= 1; n++; } return n; }

/* dc of the top row / left column / both, dc_gen_* :93-166: the division by w + h of a rectangular block is a
 * multiply by 1/3 or 1/5 in fixed point */
static int dc_value(const int *E, int w, int h, int mode, int bitdepth_max)
{
    const int hbd = bitdepth_max > 255;
    if (mode == M_DC_128) return (bitdepth_max + 1) >> 1;
    int s = 0;
    if (mode != M_LEFT_DC) for (int i = 0; i < w; i++) s += E[1 + i];
    if (mode != M_TOP_DC) for (int i = 0; i < h; i++) s += E[-1 - i];
    if (mode == M_TOP_DC) return (s + (w >> 1)) >> ctz(w);
    if (mode == M_LEFT_DC) return (s + (h >> 1)) >> ctz(h);
    unsigned u = (unsigned) (s + ((w + h) >> 1)) >> ctz(w + h);
    if (w != h) {
        const int far = w > h * 2 || h > w * 2;
        u *= far ? (hbd ? 0x6667u : 0x3334u) : (hbd ? 0xAAABu : 0x5556u);
        u >>= hbd ? 17 : 16;
    }
    return (int) u;
}

/* get_filter_strength, :327-359 */
static int filter_strength(int wh, int angle, int is_sm)
{
    if (is_sm) {
        if (wh <= 8) return angle >= 64 ? 2 : angle >= 40 ? 1 : 0;
        if (wh <= 16) return angle >= 48 ? 2 : angle >= 20 ? 1 : 0;
        if (wh <= 24) return angle >= 4 ? 3 : 0;
        return 3;
    }
    if (wh <= 8) return angle >= 56 ? 1 : 0;
    if (wh <= 16) return angle >= 40 ? 1 : 0;
    if (wh <= 24) return angle >= 32 ? 3 : angle >= 16 ? 2 : angle >= 8 ? 1 : 0;
    if (wh <= 32) return angle >= 32 ? 3 : angle >= 4 ? 2 : 1;
    return 3;
}
static int use_upsample(int wh, int angle, int is_sm) { return angle < 40 && wh <= (16 >> is_sm); }

/* filter_edge, :361-384: out[0..sz) from in[] (valid on [from, to)), filtering only indices [lim_from, lim_to) */
static void filter_edge(int *out, int sz, int lim_from, int lim_to, const int *in, int from, int to, int strength)
{
    static const int k[3][5] = { { 0, 4, 8, 4, 0 }, { 0, 5, 6, 5, 0 }, { 2, 4, 4, 4, 2 } };
    for (int i = 0; i < sz; i++) {
        if (i < lim_from || i >= lim_to) { out[i] = in[port_iclip(i, from, to - 1)]; continue; }
        int s = 0;
        for (int j = 0; j < 5; j++) s += in[port_iclip(i - 2 + j, from, to - 1)] * k[strength - 1][j];
        out[i] = (s + 8) >> 4;
    }
}
/* upsample_edge, :390-405: 2*hsz - 1 outputs, odd ones interpolated with (-1, 9, 9, -1) / 16 */
static void upsample_edge(int *out, int hsz, const int *in, int from, int to, int bitdepth_max)
{
    for (int i = 0; i < hsz - 1; i++) {
        out[2 * i] = in[port_iclip(i, from, to - 1)];
        const int s = -in[port_iclip(i - 1, from, to - 1)] + 9 * in[port_iclip(i, from, to - 1)] +
                      9 * in[port_iclip(i + 1, from, to - 1)] - in[port_iclip(i + 2, from, to - 1)];
        out[2 * i + 1] = port_iclip((s + 8) >> 4, 0, bitdepth_max);
    }
    out[2 * hsz - 2] = in[port_iclip(hsz - 1, from, to - 1)];
}

void port_intra_pred(const int mode, void *const dst, const ptrdiff_t stride, const void *const topleft, const int w, const int h,
                     int angle, const int max_w, const int max_h, const int bitdepth_max)
{
    const int hbd = bitdepth_max > 255;
    const ptrdiff_t sp = hbd ? stride / 2 : stride;
    int edge[2 * 160 + 1], *const E = edge + 160;
    {
        const int m = w < h ? w : h;
        for (int k = -(h + m); k <= w + m; k++) E[k] = hbd ? ((const uint16_t *) topleft)[k] : ((const uint8_t *) topleft)[k];
    }
    int *out = malloc(sizeof(int) * (size_t) w * h);
    const int is_sm = (angle >> 9) & 1, edge_filter = angle >> 10;
    angle &= 511;

    if (mode <= M_DC_128 && mode != M_VERT && mode != M_HOR) {
        const int dc = dc_value(E, w, h, mode, bitdepth_max);
        for (int i = 0; i < w * h; i++) out[i] = dc;
    } else if (mode == M_VERT || mode == M_HOR) {
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) out[y * w + x] = mode == M_VERT ? E[1 + x] : E[-1 - y];
    } else if (mode == M_PAETH) {                 /* :244-265 */
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
            const int left = E[-1 - y], top = E[1 + x], tl = E[0], base = left + top - tl;
            const int dl = abs(base - left), dt = abs(base - top), dtl = abs(base - tl);
            out[y * w + x] = (dl <= dt && dl <= dtl) ? left : dt <= dtl ? top : tl;
        }
    } else if (mode >= M_SMOOTH && mode <= M_SMOOTH_H) {   /* :267-325 */
        const int right = E[w], bottom = E[-h];
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
            const int wv = av1_sm_weights[h + y], wh = av1_sm_weights[w + x];
            const int pv = wv * E[1 + x] + (256 - wv) * bottom, ph = wh * E[-1 - y] + (256 - wh) * right;
            out[y * w + x] = mode == M_SMOOTH ? (pv + ph + 256) >> 9 : mode == M_SMOOTH_V ? (pv + 128) >> 8 : (ph + 128) >> 8;
        }
    } else if (mode == M_Z1) {                    /* :407-456 */
        int top[64 + 64 + 16];
        int dx = av1_dr_intra_derivative[angle >> 1], max_base, ups = edge_filter && use_upsample(w + h, 90 - angle, is_sm);
        if (ups) {
            upsample_edge(top, w + h, E + 1, -1, w + port_imin(w, h), bitdepth_max);
            max_base = 2 * (w + h) - 2; dx <<= 1;
        } else {
            const int fs = edge_filter ? filter_strength(w + h, 90 - angle, is_sm) : 0;
            if (fs) { filter_edge(top, w + h, 0, w + h, E + 1, -1, w + port_imin(w, h), fs); max_base = w + h - 1; }
            else { for (int i = 0; i < w + port_imin(w, h); i++) top[i] = E[1 + i]; max_base = w + port_imin(w, h) - 1; }
        }
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
            const int xpos = dx * (y + 1), frac = xpos & 0x3E, base = (xpos >> 6) + x * (1 + ups);
            out[y * w + x] = base < max_base ? (top[base] * (64 - frac) + top[base + 1] * frac + 32) >> 6 : top[max_base];
        }
    } else if (mode == M_Z2) {                    /* :458-541 */
        int buf[64 * 2 + 64 * 2 + 1 + 16], *const tl = buf + 64 * 2 + 8;     /* tl[0] = corner, tl[1..] top, tl[-1..] left */
        int dy = av1_dr_intra_derivative[(angle - 90) >> 1], dx = av1_dr_intra_derivative[(180 - angle) >> 1];
        const int ups_l = edge_filter && use_upsample(w + h, 180 - angle, is_sm);
        const int ups_a = edge_filter && use_upsample(w + h, angle - 90, is_sm);
        if (ups_a) { upsample_edge(tl, w + 1, E, 0, w + 1, bitdepth_max); dx <<= 1; }
        else {
            const int fs = edge_filter ? filter_strength(w + h, angle - 90, is_sm) : 0;
            if (fs) filter_edge(tl + 1, w, 0, max_w, E + 1, -1, w, fs);
            else for (int i = 0; i < w; i++) tl[1 + i] = E[1 + i];
        }
        if (ups_l) { upsample_edge(tl - 2 * h, h + 1, E - h, 0, h + 1, bitdepth_max); dy <<= 1; }
        else {
            const int fs = edge_filter ? filter_strength(w + h, 180 - angle, is_sm) : 0;
            if (fs) filter_edge(tl - h, h, h - max_h, h, E - h, 0, h + 1, fs);
            else for (int i = 0; i < h; i++) tl[-h + i] = E[-h + i];
        }
        tl[0] = E[0];
        const int base_inc_x = 1 + ups_a;
        const int *const left = tl - (1 + ups_l);
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
            const int xpos = ((1 + ups_a) << 6) - dx * (y + 1);
            const int base_x = (xpos >> 6) + x * base_inc_x, frac_x = xpos & 0x3E;
            if (base_x >= 0) out[y * w + x] = (tl[base_x] * (64 - frac_x) + tl[base_x + 1] * frac_x + 32) >> 6;
            else {
                const int ypos = (y << (6 + ups_l)) - dy * (x + 1), base_y = ypos >> 6, frac_y = ypos & 0x3E;
                out[y * w + x] = (left[-base_y] * (64 - frac_y) + left[-(base_y + 1)] * frac_y + 32) >> 6;
            }
        }
    } else if (mode == M_Z3) {                    /* :543-598 */
        int lbuf[64 + 64 + 16], *left;            /* left[-k] = k-th sample going down the (processed) left edge */
        int dy = av1_dr_intra_derivative[(270 - angle) >> 1], max_base, ups = edge_filter && use_upsample(w + h, angle - 180, is_sm);
        if (ups) {
            upsample_edge(lbuf, w + h, E - (w + h), port_imax(w - h, 0), w + h + 1, bitdepth_max);
            left = lbuf + 2 * (w + h) - 2; max_base = 2 * (w + h) - 2; dy <<= 1;
        } else {
            const int fs = edge_filter ? filter_strength(w + h, angle - 180, is_sm) : 0;
            if (fs) {
                filter_edge(lbuf, w + h, 0, w + h, E - (w + h), port_imax(w - h, 0), w + h + 1, fs);
                left = lbuf + w + h - 1; max_base = w + h - 1;
            } else { left = E - 1; max_base = h + port_imin(w, h) - 1; }
        }
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
            const int ypos = dy * (x + 1), frac = ypos & 0x3E, base = (ypos >> 6) + y * (1 + ups);
            out[y * w + x] = base < max_base ? (left[-base] * (64 - frac) + left[-(base + 1)] * frac + 32) >> 6 : left[-max_base];
        }
    } else {                                      /* M_FILTER, :616-655: 4x2 sub-blocks in raster order */
        const int8_t *const flt = &av1_filter_intra_taps[(angle & 511) * 64];
        for (int y = 0; y < h; y += 2)
            for (int x = 0; x < w; x += 4) {
                int p[7];
                for (int q = 0; q < 5; q++) p[q] = y ? out[(y - 1) * w + x - 1 + q] : E[x + q];
                if (y && !x) p[0] = E[-y];
                p[5] = x ? out[y * w + x - 1] : E[-1 - y];
                p[6] = x ? out[(y + 1) * w + x - 1] : E[-2 - y];
                for (int yy = 0; yy < 2; yy++) for (int xx = 0; xx < 4; xx++) {
                    const int8_t *f = flt + (yy * 4 + xx) * 2;      /* tap layout of the x86 build of the tables (src/tables.c:751-763) */
                    const int acc = f[0] * p[0] + f[1] * p[1] + f[16] * p[2] + f[17] * p[3] + f[32] * p[4] + f[33] * p[5] + f[48] * p[6];
                    out[(y + yy) * w + x + xx] = port_iclip((acc + 8) >> 4, 0, bitdepth_max);
                }
            }
    }
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
        if (hbd) ((uint16_t *) dst)[y * sp + x] = (uint16_t) out[y * w + x]; else ((uint8_t *) dst)[y * sp + x] = (uint8_t) out[y * w + x];
    }
    free(out);
}

/* layout 0: 4:2:0, 1: 4:2:2, 2: 4:4:4 (the index of dsp->ipred.cfl_ac) */
void port_cfl_ac(const int layout, int16_t *const ac, const void *const ypx, const ptrdiff_t stride, const int w_pad, const int h_pad,
                 const int cw, const int ch, const int hbd)
{
    const int ss_hor = layout < 2, ss_ver = layout == 0;
    const ptrdiff_t sp = hbd ? stride / 2 : stride;
    const int wv = cw - 4 * w_pad, hv = ch - 4 * h_pad;
    int sum = 0;
    for (int y = 0; y < ch; y++)
        for (int x = 0; x < cw; x++) {
            const int xs = port_imin(x, wv - 1), ys = port_imin(y, hv - 1);       /* padding repeats the last visible column / row */
            int s = 0;
            for (int dy = 0; dy <= ss_ver; dy++) for (int dx = 0; dx <= ss_hor; dx++) {
                const ptrdiff_t i = ((ys << ss_ver) + dy) * sp + (xs << ss_hor) + dx;
                s += hbd ? ((const uint16_t *) ypx)[i] : ((const uint8_t *) ypx)[i];
            }
            ac[y * cw + x] = (int16_t) (s << (1 + !ss_ver + !ss_hor));
            sum += ac[y * cw + x];
        }
    const int log2sz = ctz(cw) + ctz(ch);
    const int mean = (sum + ((1 << log2sz) >> 1)) >> log2sz;
    for (int i = 0; i < cw * ch; i++) ac[i] = (int16_t) (ac[i] - mean);
}

void port_cfl_pred(const int mode, void *const dst, const ptrdiff_t stride, const void *const topleft, const int w, const int h,
                   const int16_t *const ac, const int alpha, const int bitdepth_max)
{
    const int hbd = bitdepth_max > 255;
    const ptrdiff_t sp = hbd ? stride / 2 : stride;
    int edge[2 * 64 + 1], *const E = edge + 64;
    for (int k = -h; k <= w; k++) {
        /* only the side a DC flavour averages is guaranteed to be readable */
        const int need = (k < 0 && (mode == M_DC || mode == M_LEFT_DC)) || (k > 0 && (mode == M_DC || mode == M_TOP_DC));
        E[k] = need ? (hbd ? ((const uint16_t *) topleft)[k] : ((const uint8_t *) topleft)[k]) : 0;
    }
    const int dc = dc_value(E, w, h, mode, bitdepth_max);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
        const int diff = alpha * ac[y * w + x];
        const int m = (abs(diff) + 32) >> 6;
        const int v = port_iclip(dc + (diff < 0 ? -m : m), 0, bitdepth_max);
        if (hbd) ((uint16_t *) dst)[y * sp + x] = (uint16_t) v; else ((uint8_t *) dst)[y * sp + x] = (uint8_t) v;
    }
}

void port_pal_pred(void *const dst, const ptrdiff_t stride, const void *const pal, const uint8_t *idx, const int w, const int h, const int hbd)
{
    const ptrdiff_t sp = hbd ? stride / 2 : stride;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x += 2) {
            const int i = *idx++;
            for (int k = 0; k < 2; k++) {
                const int c = k ? i >> 4 : i & 7;
                if (hbd) ((uint16_t *) dst)[y * sp + x + k] = ((const uint16_t *) pal)[c];
                else ((uint8_t *) dst)[y * sp + x + k] = ((const uint8_t *) pal)[c];
            }
        }
}

/* ---- dav1d_prepare_intra_edges (reference src/ipred_prepare_tmpl.c:75-204): maps the bitstream mode to a predictor and
 * assembles the edge array that predictor reads -- left column, bottom-left extension, top row, top-right extension,
 * corner -- from the picture around the block, with the reference's substitution rules where a neighbour is missing.
 * Same arguments as the reference function (x, y, w, h in 4-pixel units; w / h = end of the tile). */
static int needs(int mode, int what)      /* what: 0 left, 1 top, 2 corner, 3 top-right, 4 bottom-left (:50-73) */
{
    switch (what) {
    case 0: return mode == M_DC || mode == M_HOR || mode == M_LEFT_DC || mode == M_Z2 || mode == M_Z3 || (mode >= M_SMOOTH && mode <= M_FILTER);
    case 1: return mode == M_DC || mode == M_VERT || mode == M_TOP_DC || mode == M_Z1 || mode == M_Z2 || (mode >= M_SMOOTH && mode <= M_FILTER);
    case 2: return mode == M_Z1 || mode == M_Z2 || mode == M_Z3 || mode == M_PAETH || mode == M_FILTER;
    case 3: return mode == M_Z1;
    default: return mode == M_Z3;
    }
}

int port_prepare_intra_edges(const int x, const int have_left, const int y, const int have_top, const int w, const int h,
                             const int edge_flags, const void *const dst, const ptrdiff_t stride, const void *const sb_edge,
                             int mode, int *const angle, const int tw, const int th, const int filter_edge, void *const topleft_out,
                             const int bitdepth_max)
{
    const int hbd = bitdepth_max > 255;
    const ptrdiff_t sp = hbd ? stride / 2 : stride;
    int bd = 0;
    while (bitdepth_max >> bd) bd++;
#define PX(i) (hbd ? (int) ((const uint16_t *) dst)[i] : (int) ((const uint8_t *) dst)[i])
#define OUT(k, v) do { if (hbd) ((uint16_t *) topleft_out)[k] = (uint16_t) (v); else ((uint8_t *) topleft_out)[k] = (uint8_t) (v); } while (0)
#define GET(k) (hbd ? (int) ((uint16_t *) topleft_out)[k] : (int) ((uint8_t *) topleft_out)[k])
    /* bitstream mode -> predictor (:89-116) */
    if (mode >= 1 && mode <= 8) {
        static const uint8_t base[8] = { 90, 180, 45, 135, 113, 157, 203, 67 };
        *angle = base[mode - 1] + 3 * *angle;
        if (*angle <= 90) mode = (*angle < 90 && have_top) ? M_Z1 : M_VERT;
        else if (*angle < 180) mode = M_Z2;
        else mode = (*angle > 180 && have_left) ? M_Z3 : M_HOR;
    } else if (mode == 0) {
        mode = have_left ? (have_top ? M_DC : M_LEFT_DC) : (have_top ? M_TOP_DC : M_DC_128);
    } else if (mode == 12) {
        mode = have_left ? (have_top ? M_PAETH : M_HOR) : (have_top ? M_VERT : M_DC_128);
    }
    /* the row above: the picture, or the saved pre-filter row at the top of a superblock row */
    ptrdiff_t top0 = -sp;                      /* index of the pixel above the block's first column */
    const void *toprow = dst;
    if (sb_edge) { toprow = sb_edge; top0 = x * 4; }
#define TOP(i) (hbd ? (int) ((const uint16_t *) toprow)[top0 + (i)] : (int) ((const uint8_t *) toprow)[top0 + (i)])
    if (needs(mode, 0)) {
        const int sz = th * 4;
        if (have_left) {
            const int have = port_imin(sz, (h - y) * 4);
            for (int i = 0; i < sz; i++) OUT(-1 - i, PX(port_imin(i, have - 1) * sp - 1));
        } else {
            const int v = have_top ? TOP(0) : ((1 << bd) >> 1) + 1;
            for (int i = 0; i < sz; i++) OUT(-1 - i, v);
        }
        if (needs(mode, 4)) {
            const int have_bl = (!have_left || y + th >= h) ? 0 : (edge_flags & 8);
            if (have_bl) {
                const int have = port_imin(sz, (h - y - th) * 4);
                for (int i = 0; i < sz; i++) OUT(-sz - 1 - i, PX((sz + port_imin(i, have - 1)) * sp - 1));
            } else {
                const int v = GET(-sz);
                for (int i = 0; i < sz; i++) OUT(-sz - 1 - i, v);
            }
        }
    }
    if (needs(mode, 1)) {
        const int sz = tw * 4;
        if (have_top) {
            const int have = port_imin(sz, (w - x) * 4);
            for (int i = 0; i < sz; i++) OUT(1 + i, TOP(port_imin(i, have - 1)));
        } else {
            const int v = have_left ? PX(-1) : ((1 << bd) >> 1) - 1;
            for (int i = 0; i < sz; i++) OUT(1 + i, v);
        }
        if (needs(mode, 3)) {
            const int have_tr = (!have_top || x + tw >= w) ? 0 : (edge_flags & 1);
            if (have_tr) {
                const int have = port_imin(sz, (w - x - tw) * 4);
                for (int i = 0; i < sz; i++) OUT(1 + sz + i, TOP(sz + port_imin(i, have - 1)));
            } else {
                const int v = GET(sz);
                for (int i = 0; i < sz; i++) OUT(1 + sz + i, v);
            }
        }
    }
    if (needs(mode, 2)) {
        int v;
        if (have_left) v = have_top ? TOP(-1) : PX(-1);
        else v = have_top ? TOP(0) : (1 << bd) >> 1;
        OUT(0, v);
        if (mode == M_Z2 && tw + th >= 6 && filter_edge) OUT(0, ((GET(-1) + GET(1)) * 5 + v * 6 + 8) >> 4);
    }
#undef PX
#undef OUT
#undef GET
#undef TOP
    return mode;
}

int dav1d_prepare_intra_edges_8bpc(int x, int hl, int y, int ht, int w, int h, int ef, const void *dst, ptrdiff_t stride, const void *sb,
                                   int mode, int *angle, int tw, int th, int fe, void *out)
{ return port_prepare_intra_edges(x, hl, y, ht, w, h, ef, dst, stride, sb, mode, angle, tw, th, fe, out, 255); }
int dav1d_prepare_intra_edges_16bpc(int x, int hl, int y, int ht, int w, int h, int ef, const void *dst, ptrdiff_t stride, const void *sb,
                                    int mode, int *angle, int tw, int th, int fe, void *out, int bdmax)
{ return port_prepare_intra_edges(x, hl, y, ht, w, h, ef, dst, stride, sb, mode, angle, tw, th, fe, out, bdmax); }

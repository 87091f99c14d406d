/* oracle/port — CPU restatement of motion compensation.  TEST INFRASTRUCTURE ONLY.
 *
 * put/prep 8-tap and bilinear: reference src/mc_tmpl.c:129-187, 246-305, 434-489, 516-586;
 * avg / w_avg / mask / w_mask: :628-681, 724-794; blend*: :682-722; emu_edge: :868-916.
 * One generic routine per family, bit depth passed at run time, pixels accessed through
 * small helpers (clarity over speed). */
#include "port.h"

static inline int px_get(const void *p, ptrdiff_t stride_bytes, int x, int y, int hbd) {
    const uint8_t *row = (const uint8_t *) p + y * stride_bytes;
    return hbd ? ((const uint16_t *) row)[x] : row[x];
}
static inline void px_set(void *p, ptrdiff_t stride_bytes, int x, int y, int v, int hbd) {
    uint8_t *row = (uint8_t *) p + y * stride_bytes;
    if (hbd) ((uint16_t *) row)[x] = (uint16_t) v; else row[x] = (uint8_t) v;
}
static int bitdepth_of(int bitdepth_max) { int b = 0; while (bitdepth_max >> b) b++; return b; }

/* taps of direction d for phase m (0 = none); bilinear = {16-m, m} on 4 bits, 8-tap on 6 bits */
static void taps(int f[8], int set, int m) {
    for (int i = 0; i < 8; i++) f[i] = 0;
    if (!m) f[3] = 1;
    else if (set == 6) { f[3] = 16 - m; f[4] = m; }
    else for (int i = 0; i < 8; i++) f[i] = av1_mc_subpel_filters[(set * 15 + m - 1) * 8 + i];
}

/* dst != NULL: put (pixels); tmp != NULL: prep (int16, row stride w) */
void port_mc(void *dst, ptrdiff_t dst_stride, int16_t *tmp, const void *src, ptrdiff_t src_stride,
             int w, int h, int mx, int my, int filter_2d, int bitdepth_max)
{
    static const uint8_t ht[9] = { 0, 0, 0, 2, 2, 2, 1, 1, 1 }, vt[9] = { 0, 1, 2, 0, 1, 2, 0, 1, 2 };   /* src/levels.h:184-196 */
    const int hbd = bitdepth_max > 255;
    const int ib = hbd ? 14 - bitdepth_of(bitdepth_max) : 4;     /* intermediate_bits, src/mc_tmpl.c:39-49 */
    const int bias = hbd ? 8192 : 0;
    const int bilin = filter_2d == 9;
    const int fb = bilin ? 4 : 6;
    int fh[8], fv[8];
    taps(fh, bilin ? 6 : (w > 4 ? ht[filter_2d] : 3 + (ht[filter_2d] & 1)), mx);     /* GET_H_FILTER, :115-118 */
    taps(fv, bilin ? 6 : (h > 4 ? vt[filter_2d] : 3 + (vt[filter_2d] & 1)), my);     /* GET_V_FILTER, :120-123 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int v;
            if (mx && my) {
                int acc = 0;
                for (int j = 0; j < 8; j++) {
                    if (!fv[j]) continue;
                    int hs = 0;
                    for (int i = 0; i < 8; i++) if (fh[i]) hs += fh[i] * px_get(src, src_stride, x + i - 3, y + j - 3, hbd);
                    const int16_t mid = (int16_t) ((hs + ((1 << (fb - ib)) >> 1)) >> (fb - ib));
                    acc += fv[j] * mid;
                }
                v = dst ? (acc + ((1 << (fb + ib)) >> 1)) >> (fb + ib) : ((acc + ((1 << fb) >> 1)) >> fb) - bias;
            } else if (mx) {
                int hs = 0;
                for (int i = 0; i < 8; i++) if (fh[i]) hs += fh[i] * px_get(src, src_stride, x + i - 3, y, hbd);
                const int r1 = (1 << (fb - ib)) >> 1;
                if (!dst) v = ((hs + r1) >> (fb - ib)) - bias;
                else if (bilin) v = (((hs + r1) >> (fb - ib)) + ((1 << ib) >> 1)) >> ib;
                else v = (hs + 32 + r1) >> 6;
            } else if (my) {
                int vs = 0;
                for (int j = 0; j < 8; j++) if (fv[j]) vs += fv[j] * px_get(src, src_stride, x, y + j - 3, hbd);
                v = dst ? (vs + ((1 << fb) >> 1)) >> fb : ((vs + ((1 << (fb - ib)) >> 1)) >> (fb - ib)) - bias;
            } else {
                const int p = px_get(src, src_stride, x, y, hbd);
                v = dst ? p : (p << ib) - bias;
            }
            if (dst) px_set(dst, dst_stride, x, y, port_iclip(v, 0, bitdepth_max), hbd);
            else tmp[y * w + x] = (int16_t) v;
        }
}

/* kind 0 avg, 1 w_avg (arg = weight), 2 mask, 3 w_mask (arg = sign, ss 0:444 1:422 2:420) */
void port_comp(int kind, int ss, void *dst, ptrdiff_t dst_stride, const int16_t *tmp1, const int16_t *tmp2,
               int w, int h, int arg, const uint8_t *mask_in, uint8_t *mask_out, int bitdepth_max)
{
    const int hbd = bitdepth_max > 255;
    const int bd = bitdepth_of(bitdepth_max);
    const int ib = hbd ? 14 - bd : 4, bias = hbd ? 8192 : 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int a = tmp1[y * w + x], b = tmp2[y * w + x];
            int v;
            if (kind == 0) v = (a + b + (1 << ib) + bias * 2) >> (ib + 1);
            else if (kind == 1) v = (a * arg + b * (16 - arg) + (8 << ib) + bias * 16) >> (ib + 4);
            else if (kind == 2) { const int m = mask_in[y * w + x]; v = (a * m + b * (64 - m) + (32 << ib) + bias * 64) >> (ib + 6); }
            else {
                const int mask_sh = bd + ib - 4;
                const int m = port_imin(38 + ((abs(a - b) + (1 << (mask_sh - 5))) >> mask_sh), 64);
                v = ((a - b) * m + b * 64 + (32 << ib) + bias * 64) >> (ib + 6);
                /* the reference accumulates the sub-sampled mask in place (src/mc_tmpl.c:748-757) */
                if (ss == 0) mask_out[y * w + x] = (uint8_t) m;
                else if (ss == 1) {
                    uint8_t *mo = &mask_out[y * (w >> 1) + (x >> 1)];
                    if (x & 1) *mo = (uint8_t) ((*mo + m + 1 - arg) >> 1); else *mo = (uint8_t) m;
                } else {
                    uint8_t *mo = &mask_out[(y >> 1) * (w >> 1) + (x >> 1)];
                    if (!(y & 1) && !(x & 1)) *mo = (uint8_t) m;
                    else if ((y & 1) && (x & 1)) *mo = (uint8_t) ((*mo + m + 2 - arg) >> 2);
                    else *mo = (uint8_t) (*mo + m);
                }
            }
            px_set(dst, dst_stride, x, y, port_iclip(v, 0, bitdepth_max), hbd);
        }
}

void port_emu_edge(intptr_t bw, intptr_t bh, intptr_t iw, intptr_t ih, intptr_t x, intptr_t y,
                   void *dst, ptrdiff_t dst_stride, const void *ref, ptrdiff_t ref_stride, int hbd)
{
    for (int j = 0; j < bh; j++)
        for (int i = 0; i < bw; i++)
            px_set(dst, dst_stride, i, j,
                   px_get(ref, ref_stride, port_iclip((int) (x + i), 0, (int) iw - 1), port_iclip((int) (y + j), 0, (int) ih - 1), hbd), hbd);
}

/* dir 0: blend (mask array), 1: blend_v, 2: blend_h; (a*(64-m) + b*m + 32) >> 6 */
void port_blend(int dir, void *dst, ptrdiff_t dst_stride, const void *tmp, int w, int h, const uint8_t *mask, int hbd)
{
    const int bps = hbd ? 2 : 1;
    const int ww = dir == 1 ? (w * 3) >> 2 : w, hh = dir == 2 ? (h * 3) >> 2 : h;
    for (int y = 0; y < hh; y++)
        for (int x = 0; x < ww; x++) {
            const int m = dir == 0 ? mask[y * w + x] : dir == 1 ? av1_obmc_masks[w + x] : av1_obmc_masks[h + y];
            const int a = px_get(dst, dst_stride, x, y, hbd), b = px_get(tmp, (ptrdiff_t) w * bps, x, y, hbd);
            px_set(dst, dst_stride, x, y, (a * (64 - m) + b * m + 32) >> 6, hbd);
        }
}

/* ---- warp8x8 / warp8x8t (reference src/mc_tmpl.c:799-866): 15 rows filtered horizontally with a per-pixel filter chosen by
 * the affine position, then 8 rows vertically.  dst != NULL: pixels; tmp != NULL: int16 with PREP_BIAS. */
void port_warp8x8(void *dst, ptrdiff_t dst_stride, int16_t *tmp, ptrdiff_t tmp_stride, const void *src, ptrdiff_t src_stride,
                  const int16_t *abcd, int mx, int my, int bitdepth_max)
{
    const int hbd = bitdepth_max > 255;
    const int ib = hbd ? 14 - bitdepth_of(bitdepth_max) : 4;
    int16_t mid[15][8];
    for (int y = 0; y < 15; y++)
        for (int x = 0; x < 8; x++) {
            const int tmx = mx + y * abcd[1] + x * abcd[0];
            const int8_t *f = &av1_mc_warp_filter[(64 + ((tmx + 512) >> 10)) * 8];
            int s = 0;
            for (int k = 0; k < 8; k++) s += f[k] * px_get(src, src_stride, x + k - 3, y - 3, hbd);
            mid[y][x] = (int16_t) ((s + ((1 << (7 - ib)) >> 1)) >> (7 - ib));
        }
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            const int tmy = my + y * abcd[3] + x * abcd[2];
            const int8_t *f = &av1_mc_warp_filter[(64 + ((tmy + 512) >> 10)) * 8];
            int s = 0;
            for (int k = 0; k < 8; k++) s += f[k] * mid[y + k][x];
            if (dst) px_set(dst, dst_stride, x, y, port_iclip((s + ((1 << (7 + ib)) >> 1)) >> (7 + ib), 0, bitdepth_max), hbd);
            else tmp[y * tmp_stride + x] = (int16_t) (((s + 64) >> 7) - (hbd ? 8192 : 0));
        }
}

/* ---- put / prep with a scaled reference (reference src/mc_tmpl.c:189-244, 307-357, 491-626): output (x, y) samples the source at
 * (mx + x*dx, my + y*dy) in 1/1024 pel; the 1/16-pel phase of each coordinate picks the filter row. */
void port_mc_scaled(void *dst, ptrdiff_t dst_stride, int16_t *tmp, const void *src, ptrdiff_t src_stride, int w, int h,
                    int mx, int my, int dx, int dy, int filter_2d, int bitdepth_max)
{
    static const uint8_t ht[9] = { 0, 0, 0, 2, 2, 2, 1, 1, 1 }, vt[9] = { 0, 1, 2, 0, 1, 2, 0, 1, 2 };
    const int hbd = bitdepth_max > 255;
    const int ib = hbd ? 14 - bitdepth_of(bitdepth_max) : 4;
    const int bias = hbd ? 8192 : 0;
    const int bilin = filter_2d == 9;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int px = mx + x * dx, py = my + y * dy;
            const int sx = px >> 10, fx = (px & 0x3ff) >> 6, sy = py >> 10, fy = (py & 0x3ff) >> 6;
            int v;
            if (bilin) {
                int m[2];
                for (int r = 0; r < 2; r++) {
                    const int a = px_get(src, src_stride, sx, sy + r, hbd), b = px_get(src, src_stride, sx + 1, sy + r, hbd);
                    m[r] = (int16_t) ((16 * a + fx * (b - a) + ((1 << (4 - ib)) >> 1)) >> (4 - ib));
                }
                const int s = 16 * m[0] + fy * (m[1] - m[0]);
                v = dst ? (s + ((1 << (4 + ib)) >> 1)) >> (4 + ib) : ((s + 8) >> 4) - bias;
            } else {
                int fh[8], fv[8], mid[8];
                taps(fh, w > 4 ? ht[filter_2d] : 3 + (ht[filter_2d] & 1), fx);
                taps(fv, h > 4 ? vt[filter_2d] : 3 + (vt[filter_2d] & 1), fy);
                for (int r = 0; r < 8; r++) {
                    int s;
                    if (fx) {
                        s = 0;
                        for (int k = 0; k < 8; k++) s += fh[k] * px_get(src, src_stride, sx + k - 3, sy + r - 3, hbd);
                        s = (s + ((1 << (6 - ib)) >> 1)) >> (6 - ib);
                    } else s = px_get(src, src_stride, sx, sy + r - 3, hbd) << ib;
                    mid[r] = (int16_t) s;
                }
                if (fy) {
                    int s = 0;
                    for (int k = 0; k < 8; k++) s += fv[k] * mid[k];
                    v = dst ? (s + ((1 << (6 + ib)) >> 1)) >> (6 + ib) : ((s + 32) >> 6) - bias;
                } else v = dst ? (mid[3] + ((1 << ib) >> 1)) >> ib : mid[3] - bias;
            }
            if (dst) px_set(dst, dst_stride, x, y, port_iclip(v, 0, bitdepth_max), hbd);
            else tmp[y * w + x] = (int16_t) v;
        }
}

/* ---- resize (reference src/mc_tmpl.c:918-944): horizontal 8-tap upscale, position in 1/16384 pel steps of dx starting at mx0 */
void port_resize(void *dst, ptrdiff_t dst_stride, const void *src, ptrdiff_t src_stride, int dst_w, int h, int src_w, int dx, int mx0,
                 int bitdepth_max)
{
    const int hbd = bitdepth_max > 255;
    for (int y = 0; y < h; y++) {
        int mx = mx0, sx = -1;
        for (int x = 0; x < dst_w; x++) {
            const int8_t *F = &av1_resize_filter[(mx >> 8) * 8];
            int s = 0;
            for (int k = 0; k < 8; k++) s += F[k] * px_get(src, src_stride, port_iclip(sx + k - 3, 0, src_w - 1), y, hbd);
            px_set(dst, dst_stride, x, y, port_iclip((-s + 64) >> 7, 0, bitdepth_max), hbd);
            mx += dx;
            sx += mx >> 14;
            mx &= 0x3fff;
        }
    }
}

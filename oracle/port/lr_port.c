/* oracle/port — CPU restatement of loop restoration.  TEST INFRASTRUCTURE ONLY.
 *
 * lr.wiener[*] = wiener_c (reference src/looprestoration_tmpl.c:44-387), lr.sgr[*] = sgr_5x5_c / sgr_3x3_c / sgr_mix_c
 * (:389-1363).  The reference streams rows through rotating buffers; here both filters are written as formulas over a
 * virtual image V(r, c), r = -3 .. h+2, c = -3 .. w+2, that applies the reference's padding rules:
 *   rows of the unit: left of column 0 = `left[r][1..3]` (LR_HAVE_LEFT) or column 0 replicated; right of column w-1 =
 *     the picture (LR_HAVE_RIGHT) or column w-1 replicated;
 *   rows above: lpf row 1 for r = -1, lpf row 0 for r <= -2 (LR_HAVE_TOP), else unit row 0;
 *   rows below: lpf rows 6, 7 (LR_HAVE_BOTTOM, and only if the stripe is long enough for the reference's row loop to
 *     reach them: :274-355 for Wiener, the sgr_*_c drivers for SGR), else unit row h-1. */
#include "port.h"

typedef struct {
    const void *p, *left, *lpf;
    ptrdiff_t sp;           /* stride in pixels */
    int w, h, edges, hbd, use_bottom;
} VImg;

static int rdpx(const void *p, ptrdiff_t i, int hbd) { return hbd ? ((const uint16_t *) p)[i] : ((const uint8_t *) p)[i]; }

static int V(const VImg *v, int r, int c)
{
    enum { HL = 1, HR = 2, HT = 4 };
    if (r < 0 && !(v->edges & HT)) r = 0;
    if (r >= v->h && !v->use_bottom) r = v->h - 1;
    if (c < 0 && !(v->edges & HL)) c = 0;
    if (c >= v->w && !(v->edges & HR)) c = v->w - 1;
    if (r < 0) return rdpx(v->lpf, (r == -1 ? 1 : 0) * v->sp + c, v->hbd);
    if (r >= v->h) return rdpx(v->lpf, (6 + (r == v->h ? 0 : 1)) * v->sp + c, v->hbd);
    if (c < 0) return rdpx(v->left, r * 4 + 4 + c, v->hbd);
    return rdpx(v->p, r * v->sp + c, v->hbd);
}

static void wr(void *p, ptrdiff_t i, int val, int hbd) { if (hbd) ((uint16_t *) p)[i] = (uint16_t) val; else ((uint8_t *) p)[i] = (uint8_t) val; }

static VImg make_v(const void *p, ptrdiff_t stride, const void *left, const void *lpf, int w, int h, int edges, int bitdepth_max)
{
    VImg v;
    v.p = p; v.left = left; v.lpf = lpf; v.w = w; v.h = h; v.edges = edges; v.hbd = bitdepth_max > 255;
    v.sp = v.hbd ? stride / 2 : stride;
    v.use_bottom = 0;
    return v;
}

/* filter[0] = horizontal, filter[1] = vertical taps (7 used), as lr_stripe() builds them (src/lr_apply_tmpl.c:55-71) */
void port_wiener(void *const p, const ptrdiff_t stride, const void *const left, const void *const lpf, const int w, const int h,
                 const int16_t filter[2][8], const int edges, const int bitdepth_max)
{
    VImg v = make_v(p, stride, left, lpf, w, h, edges, bitdepth_max);
    v.use_bottom = (edges & 8) && h >= ((edges & 4) ? 4 : 6);
    int bd = 0;
    while (bitdepth_max >> bd) bd++;
    const int rb_h = bd == 12 ? 5 : 3, rb_v = bd == 12 ? 9 : 11;          /* :50-56, :173-176 */
    const int clip_limit = 1 << (bd + 1 + 7 - rb_h);
    const int round_offset = 1 << (bd + rb_v - 1);
    int *hor = malloc(sizeof(int) * (size_t) (h + 6) * w);
    for (int r = -3; r < h + 3; r++)
        for (int x = 0; x < w; x++) {
            int sum = 1 << (bd + 6);
            if (!v.hbd) sum += V(&v, r, x) * 128;                           /* :59-61 */
            for (int i = 0; i < 7; i++) sum += V(&v, r, x + i - 3) * filter[0][i];
            hor[(r + 3) * w + x] = port_iclip((sum + (1 << (rb_h - 1))) >> rb_h, 0, clip_limit - 1);
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int sum = -round_offset;
            for (int k = 0; k < 7; k++) sum += hor[(y + k) * w + x] * filter[1][k];
            wr(p, y * v.sp + x, port_iclip((sum + (1 << (rb_v - 1))) >> rb_v, 0, bitdepth_max), v.hbd);
        }
    free(hor);
}

/* sgr_calc_row_ab, :505-523: (A, B) of one box from its sum and sum of squares */
static void calc_ab(int *A, int *B, int sumsq, int sum, int s, int bd8, int n, int one_by_x)
{
    const int a = (sumsq + ((1 << (2 * bd8)) >> 1)) >> (2 * bd8);
    const int b = (sum + ((1 << bd8) >> 1)) >> bd8;
    const unsigned p = (unsigned) port_imax(a * n - b * b, 0);
    const unsigned z = (p * (unsigned) s + (1u << 19)) >> 20;
    const unsigned x = av1_sgr_x_by_x[z < 255 ? z : 255];
    *A = (int) ((x * (unsigned) sum * (unsigned) one_by_x + (1 << 11)) >> 12);
    *B = (int) x;
}

/* type 0: 5x5, 1: 3x3, 2: mix (dsp->lr.sgr[type]) */
void port_sgr(const int type, void *const p, const ptrdiff_t stride, const void *const left, const void *const lpf, const int w, const int h,
              const unsigned s0, const unsigned s1, const int w0, const int w1, const int edges, const int bitdepth_max)
{
    VImg v = make_v(p, stride, left, lpf, w, h, edges, bitdepth_max);
    const int do5 = type != 1, do3 = type != 0;
    if (do5) v.use_bottom = (edges & 8) && !(h & 1) && h >= ((edges & 4) ? 4 : 6);
    else v.use_bottom = (edges & 8) && h >= 3;
    int bd = 0;
    while (bitdepth_max >> bd) bd++;
    const int bd8 = bd - 8, W = w + 2, H = h + 2;
    /* surfaces on rows -1 .. h, columns -1 .. w (5x5: odd rows only) */
    int *A3 = calloc((size_t) W * H, sizeof(int)), *B3 = calloc((size_t) W * H, sizeof(int));
    int *A5 = calloc((size_t) W * H, sizeof(int)), *B5 = calloc((size_t) W * H, sizeof(int));
#define AT(S, j, c) S[((j) + 1) * W + (c) + 1]
    for (int j = -1; j <= h; j++)
        for (int c = -1; c <= w; c++) {
            if (do3) {
                int sum = 0, sq = 0;
                for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) { const int q = V(&v, j + dy, c + dx); sum += q; sq += q * q; }
                calc_ab(&AT(A3, j, c), &AT(B3, j, c), sq, sum, (int) s1, bd8, 9, 455);
            }
            if (do5 && (j & 1)) {
                int sum = 0, sq = 0;
                for (int dy = -2; dy <= 2; dy++) for (int dx = -2; dx <= 2; dx++) { const int q = V(&v, j + dy, c + dx); sum += q; sq += q * q; }
                calc_ab(&AT(A5, j, c), &AT(B5, j, c), sq, sum, (int) s0, bd8, 25, 164);
            }
        }
    int *out = malloc(sizeof(int) * (size_t) w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int px = V(&v, y, x);
            int acc = 0;
            if (do3) {          /* sgr_finish_filter_row1, :555-571: centre cross x4, corners x3 */
                int a = 0, b = 0;
                for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
                    const int wt = (dy && dx) ? 3 : 4;
                    a += wt * AT(B3, y + dy, x + dx); b += wt * AT(A3, y + dy, x + dx);
                }
                acc += w1 * ((b - a * px + (1 << 8)) >> 9);
            }
            if (do5) {          /* sgr_finish_filter2, :574-600: even rows blend the surfaces above and below, odd rows use their own */
                int a = 0, b = 0, t5;
                if (!(y & 1)) {
                    for (int dy = -1; dy <= 1; dy += 2) for (int dx = -1; dx <= 1; dx++) {
                        const int wt = dx ? 5 : 6;
                        a += wt * AT(B5, y + dy, x + dx); b += wt * AT(A5, y + dy, x + dx);
                    }
                    t5 = (b - a * px + (1 << 8)) >> 9;
                } else {
                    for (int dx = -1; dx <= 1; dx++) {
                        const int wt = dx ? 5 : 6;
                        a += wt * AT(B5, y, x + dx); b += wt * AT(A5, y, x + dx);
                    }
                    t5 = (b - a * px + (1 << 7)) >> 8;
                }
                acc += w0 * t5;
            }
            out[y * w + x] = port_iclip(px + ((acc + (1 << 10)) >> 11), 0, bitdepth_max);
        }
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) wr(p, y * v.sp + x, out[y * w + x], v.hbd);
#undef AT
    free(A3); free(B3); free(A5); free(B5); free(out);
}

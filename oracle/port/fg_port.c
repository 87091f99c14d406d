/* oracle/port — CPU restatement of film grain synthesis.  TEST INFRASTRUCTURE ONLY.
 *
 * fg.generate_grain_y / generate_grain_uv[*] (reference src/filmgrain_tmpl.c:50-154), fg.fgy_32x32xn (:169-276),
 * fg.fguv_32x32xn[*] (:278-413).  Grain templates are int arrays [73 + 1][82] here; the entry wrappers convert
 * to the reference's int8_t / int16_t element type. */
#include "port.h"
#include "fg_port.h"

enum { GW = 82, GH = 73 };

static unsigned lfsr_next(unsigned *state) {
    const unsigned r = *state;
    const unsigned bit = ((r >> 0) ^ (r >> 1) ^ (r >> 3) ^ (r >> 12)) & 1;
    *state = (r >> 1) | (bit << 15);
    return *state;
}
static int rnd_bits(unsigned *state, int bits) { return (int) ((lfsr_next(state) >> (16 - bits)) & ((1u << bits) - 1)); }
static int round2(int x, int shift) { return (x + ((1 << shift) >> 1)) >> shift; }

/* pl 0: luma template; pl 1 / 2: chroma template filtered against `luma` (subsampled by subx / suby) */
void port_generate_grain(int grain[][GW], const int luma[][GW], const PortFilmGrain *d, const int pl, const int subx, const int suby,
                         const int bitdepth_max)
{
    int bd = 0;
    while (bitdepth_max >> bd) bd++;
    const int bd8 = bd - 8;
    const int W = (pl && subx) ? 44 : GW, H = (pl && suby) ? 38 : GH;
    unsigned seed = d->seed;
    if (pl) seed ^= pl == 2 ? 0x49d8 : 0xb524;
    const int shift = 4 - bd8 + d->grain_scale_shift;
    const int ctr = 128 << bd8, gmin = -ctr, gmax = ctr - 1;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            grain[y][x] = round2(av1_gaussian_sequence[rnd_bits(&seed, 11)], shift);
    /* auto-regression in raster order over the interior (3-sample border untouched) */
    const int lag = d->ar_coeff_lag;
    const int8_t *const coef = pl ? d->ar_coeffs_uv[pl - 1] : d->ar_coeffs_y;
    for (int y = 3; y < H; y++)
        for (int x = 3; x < W - 3; x++) {
            int sum = 0, k = 0;
            for (int dy = -lag; dy <= 0; dy++)
                for (int dx = -lag; dx <= lag; dx++) {
                    if (!dx && !dy) break;
                    sum += coef[k++] * grain[y + dy][x + dx];
                }
            if (pl && d->num_y_points) {
                /* the last coefficient weighs the co-located luma grain (averaged over the subsampling footprint) */
                int l = 0;
                const int lx = ((x - 3) << subx) + 3, ly = ((y - 3) << suby) + 3;
                for (int i = 0; i <= suby; i++) for (int j = 0; j <= subx; j++) l += luma[ly + i][lx + j];
                sum += round2(l, subx + suby) * coef[k];
            }
            grain[y][x] = port_iclip(grain[y][x] + round2(sum, (int) d->ar_coeff_shift), gmin, gmax);
        }
}

/* sample_lut, :156-167 */
static int lut_at(const int lut[][GW], int randval, int subx, int suby, int bx, int by, int x, int y)
{
    const int offx = 3 + (2 >> subx) * (3 + (randval >> 4));
    const int offy = 3 + (2 >> suby) * (3 + (randval & 0xF));
    return lut[offy + y + (32 >> suby) * by][offx + x + (32 >> subx) * bx];
}

/* one row of 32x32 blocks of plane pl (0 luma: luma_row unused).  Pixel access through typed getters. */
void port_fg_row(const int pl, void *const dst_row, const void *const src_row, const ptrdiff_t stride, const PortFilmGrain *d, const int pw,
                 const uint8_t *const scaling, const int lut[][GW], const int bh, const int row_num, const void *const luma_row,
                 const ptrdiff_t luma_stride, const int subx, const int suby, const int is_id, const int bitdepth_max)
{
    const int hbd = bitdepth_max > 255;
    int bd = 0;
    while (bitdepth_max >> bd) bd++;
    const int bd8 = bd - 8, uv = pl - 1;
    const ptrdiff_t sp = hbd ? stride / 2 : stride, lsp = hbd ? luma_stride / 2 : luma_stride;
    const int ctr = 128 << bd8, gmin = -ctr, gmax = ctr - 1;
    int minv = 0, maxv = bitdepth_max;
    if (d->clip_to_restricted_range) { minv = 16 << bd8; maxv = ((pl && !is_id) ? 240 : 235) << bd8; }
    const int rows = 1 + (d->overlap_flag && row_num > 0);
    unsigned seed[2];
    for (int i = 0; i < rows; i++) {
        seed[i] = d->seed;
        seed[i] ^= (((row_num - i) * 37 + 178) & 0xFF) << 8;
        seed[i] ^= (((row_num - i) * 173 + 105) & 0xFF);
    }
    const int bstep = 32 >> subx;
    int offs[2][2] = { { 0, 0 }, { 0, 0 } };       /* [this block, previous block][this row, row above] */
    for (int bx = 0; bx < pw; bx += bstep) {
        const int bw = port_imin(bstep, pw - bx);
        for (int i = 0; i < rows; i++) { offs[1][i] = offs[0][i]; offs[0][i] = rnd_bits(&seed[i], 8); }
        const int ystart = (d->overlap_flag && row_num) ? port_imin(2 >> suby, bh) : 0;
        const int xstart = (d->overlap_flag && bx) ? port_imin(2 >> subx, bw) : 0;
        for (int y = 0; y < bh; y++)
            for (int x = 0; x < bw; x++) {
                /* overlap weights: (27, 17), (17, 27) at full resolution, (23, 22) when subsampled */
                const int wxa = subx ? 23 : (x == 0 ? 27 : 17), wxb = subx ? 22 : (x == 0 ? 17 : 27);
                const int wya = suby ? 23 : (y == 0 ? 27 : 17), wyb = suby ? 22 : (y == 0 ? 17 : 27);
                int grain = lut_at(lut, offs[0][0], subx, suby, 0, 0, x, y);
                if (x < xstart) {
                    const int old = lut_at(lut, offs[1][0], subx, suby, 1, 0, x, y);
                    grain = port_iclip(round2(old * wxa + grain * wxb, 5), gmin, gmax);
                }
                if (y < ystart) {
                    int top = lut_at(lut, offs[0][1], subx, suby, 0, 1, x, y);
                    if (x < xstart) {
                        const int old = lut_at(lut, offs[1][1], subx, suby, 1, 1, x, y);
                        top = port_iclip(round2(old * wxa + top * wxb, 5), gmin, gmax);
                    }
                    grain = port_iclip(round2(top * wya + grain * wyb, 5), gmin, gmax);
                }
                const ptrdiff_t i = y * sp + bx + x;
                const int s0 = hbd ? ((const uint16_t *) src_row)[i] : ((const uint8_t *) src_row)[i];
                int val = s0;
                if (pl) {
                    const ptrdiff_t li = (ptrdiff_t) (y << suby) * lsp + ((bx + x) << subx);
                    int avg = hbd ? ((const uint16_t *) luma_row)[li] : ((const uint8_t *) luma_row)[li];
                    if (subx) avg = (avg + (hbd ? ((const uint16_t *) luma_row)[li + 1] : ((const uint8_t *) luma_row)[li + 1]) + 1) >> 1;
                    val = avg;
                    if (!d->chroma_scaling_from_luma) {
                        const int comb = avg * d->uv_luma_mult[uv] + s0 * d->uv_mult[uv];
                        val = port_iclip((comb >> 6) + d->uv_offset[uv] * (1 << bd8), 0, bitdepth_max);
                    }
                }
                const int noise = round2(scaling[val] * grain, d->scaling_shift);
                const int o = port_iclip(s0 + noise, minv, maxv);
                if (hbd) ((uint16_t *) dst_row)[i] = (uint16_t) o; else ((uint8_t *) dst_row)[i] = (uint8_t) o;
            }
    }
}

/* ---- the driver: dav1d_apply_grain (reference src/fg_apply_tmpl.c:41-241) on caller-provided planes.  Builds the grain
 * templates and the scaling tables, copies the planes that get no grain, then walks the picture in rows of 32 lines. */

/* generate_scaling (:41-95): piecewise-linear table through the (x, y) points; for more than 8 bits the 8-bit grid is
 * spread out and the gaps are filled by interpolation */
static void scaling_table(uint8_t *const lut, const int bd, const uint8_t pts[][2], const int num)
{
    const int shift = bd - 8, size = 1 << bd;
    if (!num) { memset(lut, 0, size); return; }
    memset(lut, pts[0][1], (size_t) pts[0][0] << shift);
    for (int i = 0; i + 1 < num; i++) {
        const int x0 = pts[i][0], y0 = pts[i][1], run = pts[i + 1][0] - x0, rise = pts[i + 1][1] - y0;
        const int slope = rise * ((0x10000 + (run >> 1)) / run);
        for (int k = 0, acc = 0x8000; k < run; k++, acc += slope) lut[(x0 + k) << shift] = (uint8_t) (y0 + (acc >> 16));
    }
    const int last = pts[num - 1][0] << shift;
    memset(lut + last, pts[num - 1][1], size - last);
    if (shift) {
        const int pad = 1 << shift, half = pad >> 1;
        for (int i = 0; i + 1 < num; i++) {
            const int from = pts[i][0] << shift, to = pts[i + 1][0] << shift;
            for (int x = 0; x < to - from; x += pad) {
                const int span = lut[from + x + pad] - lut[from + x];
                for (int n = 1, acc = half; n < pad; n++) { acc += span; lut[from + x + n] = (uint8_t) (lut[from + x] + (acc >> shift)); }
            }
        }
    }
}

int dav1d_port_apply_grain(const int bpc, const PortFilmGrain *const d, const int w, const int h, const int layout, const int is_id,
                           void *const out[3], void *const in[3], const ptrdiff_t y_stride, const ptrdiff_t uv_stride)
{
    const int bdmax = (1 << bpc) - 1, hbd = bpc > 8, bps = hbd ? 2 : 1;
    const int sx = layout != 3 && layout != 0, sy = layout == 1;
    static int grain[3][74][GW];
    memset(grain, 0, sizeof(grain));
    uint8_t *scaling = malloc((size_t) 3 << bpc);
    if (!scaling) return -1;
    port_generate_grain(grain[0], NULL, d, 0, 0, 0, bdmax);
    for (int pl = 0; pl < 2; pl++)
        if (layout && (d->num_uv_points[pl] || d->chroma_scaling_from_luma))
            port_generate_grain(grain[1 + pl], (const int (*)[GW]) grain[0], d, 1 + pl, sx, sy, bdmax);
    if (d->num_y_points || d->chroma_scaling_from_luma) scaling_table(scaling, bpc, d->y_points, d->num_y_points);
    for (int pl = 0; pl < 2; pl++)
        if (d->num_uv_points[pl]) scaling_table(scaling + ((size_t) (1 + pl) << bpc), bpc, d->uv_points[pl], d->num_uv_points[pl]);
    /* planes without grain are copied (:127-163) */
    if (!d->num_y_points) memcpy(out[0], in[0], (size_t) h * y_stride);
    if (layout && !d->chroma_scaling_from_luma)
        for (int pl = 0; pl < 2; pl++)
            if (!d->num_uv_points[pl]) memcpy(out[1 + pl], in[1 + pl], (size_t) ((h + sy) >> sy) * uv_stride);
    const int cpw = (w + sx) >> sx;
    for (int row = 0; row * 32 < h; row++) {
        const int lines = port_imin(h - row * 32, 32);
        uint8_t *const luma = (uint8_t *) in[0] + (size_t) row * 32 * y_stride;
        if (d->num_y_points)
            port_fg_row(0, (uint8_t *) out[0] + (size_t) row * 32 * y_stride, luma, y_stride, d, w, scaling, (const int (*)[GW]) grain[0], lines,
                        row, NULL, 0, 0, 0, is_id, bdmax);
        if (!layout || (!d->num_uv_points[0] && !d->num_uv_points[1] && !d->chroma_scaling_from_luma)) continue;
        const int bh = (lines + sy) >> sy;
        if (w & sx)         /* odd width: the luma row gets one replicated pixel so that the 2:1 average has a partner (:193-199) */
            for (int y = 0; y < bh; y++) {
                uint8_t *p = luma + (size_t) (y << sy) * y_stride;
                memcpy(p + (size_t) w * bps, p + (size_t) (w - 1) * bps, bps);
            }
        const size_t uv_off = (size_t) row * 32 * uv_stride >> sy;
        for (int pl = 0; pl < 2; pl++) {
            if (!d->chroma_scaling_from_luma && !d->num_uv_points[pl]) continue;
            const uint8_t *sc = scaling + (d->chroma_scaling_from_luma ? 0 : (size_t) (1 + pl) << bpc);
            port_fg_row(1 + pl, (uint8_t *) out[1 + pl] + uv_off, (const uint8_t *) in[1 + pl] + uv_off, uv_stride, d, cpw, sc,
                        (const int (*)[GW]) grain[1 + pl], bh, row, luma, y_stride, sx, sy, is_id, bdmax);
        }
    }
    free(scaling);
    return 0;
}

/* grain templates as int16 [3][73 + 1][82], the chroma ones only where dav1d_apply_grain would build them */
int dav1d_port_generate_grain(const int bpc, const PortFilmGrain *const d, const int layout, int16_t *const out)
{
    static int grain[3][74][GW];
    memset(grain, 0, sizeof(grain));
    const int sx = layout != 3 && layout != 0, sy = layout == 1, bdmax = (1 << bpc) - 1;
    port_generate_grain(grain[0], NULL, d, 0, 0, 0, bdmax);
    for (int pl = 0; pl < 2; pl++)
        if (layout && (d->num_uv_points[pl] || d->chroma_scaling_from_luma))
            port_generate_grain(grain[1 + pl], (const int (*)[GW]) grain[0], d, 1 + pl, sx, sy, bdmax);
    for (int i = 0; i < 3 * 74 * GW; i++) out[i] = (int16_t) ((int *) grain)[i];
    return 0;
}

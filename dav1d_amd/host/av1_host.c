/* Host-side AV1 geometry derived by rule; see av1_host.h.  Plain C99. */
#include "av1_host.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

uint8_t h_bs_dim[H_N_BS][4];
HostTx  h_tx[H_N_TX];
uint8_t h_max_tx_for_bs[H_N_BS][4];
uint8_t h_block_sizes[5][H_N_PARTITIONS][2];

static int ilog2(int v) { int l = 0; while (v > 1) { v >>= 1; l++; } return l; }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

int h_bs_from_dim(const int bw4, const int bh4) {
    for (int bs = 0; bs < H_N_BS; bs++)
        if (h_bs_dim[bs][0] == bw4 && h_bs_dim[bs][1] == bh4) return bs;
    return -1;
}

int h_tx_from_dim(const int w4, const int h4) {
    for (int tx = 0; tx < H_N_TX; tx++)
        if (h_tx[tx].w == w4 && h_tx[tx].h == h4) return tx;
    return -1;
}

static void tables_build(void) {
    /* enum BlockSize: widths descending from 128, for each width heights descending, aspect ratios up to 4:1 (AV1 spec 6.10.4;
     * the order is that of the reference's enum, src/levels.h:149-173) */
    int n = 0;
    for (int w = 32; w >= 1; w >>= 1)
        for (int h = 32; h >= 1; h >>= 1) {
            if (w > 4 * h || h > 4 * w) continue;
            if ((w == 32 || h == 32) && imin(w, h) < 16) continue;      /* 128-pixel sides only come as 128x128, 128x64, 64x128 */
            h_bs_dim[n][0] = (uint8_t) w; h_bs_dim[n][1] = (uint8_t) h;
            h_bs_dim[n][2] = (uint8_t) ilog2(w); h_bs_dim[n][3] = (uint8_t) ilog2(h);
            n++;
        }
    /* enum RectTxfmSize: squares, then 1:2 / 2:1 pairs ascending, then 1:4 / 4:1 pairs ascending (src/levels.h:44-78) */
    n = 0;
    for (int s = 1; s <= 16; s <<= 1) { h_tx[n].w = (uint8_t) s; h_tx[n].h = (uint8_t) s; n++; }
    for (int s = 1; s <= 8; s <<= 1) {
        h_tx[n].w = (uint8_t) s; h_tx[n].h = (uint8_t) (2 * s); n++;
        h_tx[n].w = (uint8_t) (2 * s); h_tx[n].h = (uint8_t) s; n++;
    }
    for (int s = 1; s <= 4; s <<= 1) {
        h_tx[n].w = (uint8_t) s; h_tx[n].h = (uint8_t) (4 * s); n++;
        h_tx[n].w = (uint8_t) (4 * s); h_tx[n].h = (uint8_t) s; n++;
    }
    for (int tx = 0; tx < H_N_TX; tx++) {
        HostTx *const t = &h_tx[tx];
        t->lw = (uint8_t) ilog2(t->w); t->lh = (uint8_t) ilog2(t->h);
        t->min = t->lw < t->lh ? t->lw : t->lh; t->max = t->lw > t->lh ? t->lw : t->lh;
    }
    for (int tx = 0; tx < H_N_TX; tx++) {
        /* Split_Tx_Size: the longer side is halved, both when square (AV1 spec 9.3) */
        const int w = h_tx[tx].w, h = h_tx[tx].h;
        const int sw = w >= h ? w >> 1 : w, sh = h >= w ? h >> 1 : h;
        h_tx[tx].sub = (uint8_t) (tx == H_TX_4X4 ? 0 : h_tx_from_dim(imax(sw, 1), imax(sh, 1)));
    }
    /* largest transform of a block per plane (Max_Tx_Size_Rect + get_tx_size_uv, AV1 spec 5.11.37) */
    for (int bs = 0; bs < H_N_BS; bs++)
        for (int pl = 0; pl < 4; pl++) {
            const int ss_hor = pl == 1 || pl == 2, ss_ver = pl == 1;
            int w = h_bs_dim[bs][0], h = h_bs_dim[bs][1];
            if (pl) { w = imax(w >> ss_hor, 1); h = imax(h >> ss_ver, 1); }
            w = imin(w, 16); h = imin(h, 16);
            if (pl && (w == 16 || h == 16)) {           /* no 64-point chroma transforms */
                if (w == 4) h = 8; else if (h == 4) w = 8; else w = h = 8;
            }
            while (w > 4 * h) w >>= 1;
            while (h > 4 * w) h >>= 1;
            h_max_tx_for_bs[bs][pl] = (uint8_t) h_tx_from_dim(w, h);
        }
    /* partition shapes per block level (AV1 spec 5.11.4 / 9.3 Partition_Subsize) */
    memset(h_block_sizes, 0, sizeof(h_block_sizes));
    for (int bl = 0; bl < 5; bl++) {
        const int s = 32 >> bl, hs = s >> 1, q = s >> 2;
#define BSZ(w, h) ((uint8_t) imax(h_bs_from_dim(w, h), 0))
        h_block_sizes[bl][H_PART_NONE][0] = BSZ(s, s);
        h_block_sizes[bl][H_PART_H][0] = BSZ(s, hs);
        h_block_sizes[bl][H_PART_V][0] = BSZ(hs, s);
        if (bl == H_BL_8X8) { h_block_sizes[bl][H_PART_SPLIT][0] = BSZ(1, 1); continue; }
        h_block_sizes[bl][H_PART_T_TOP_SPLIT][0] = BSZ(hs, hs);    h_block_sizes[bl][H_PART_T_TOP_SPLIT][1] = BSZ(s, hs);
        h_block_sizes[bl][H_PART_T_BOTTOM_SPLIT][0] = BSZ(s, hs);  h_block_sizes[bl][H_PART_T_BOTTOM_SPLIT][1] = BSZ(hs, hs);
        h_block_sizes[bl][H_PART_T_LEFT_SPLIT][0] = BSZ(hs, hs);   h_block_sizes[bl][H_PART_T_LEFT_SPLIT][1] = BSZ(hs, s);
        h_block_sizes[bl][H_PART_T_RIGHT_SPLIT][0] = BSZ(hs, s);   h_block_sizes[bl][H_PART_T_RIGHT_SPLIT][1] = BSZ(hs, hs);
        if (bl != H_BL_128X128) {
            h_block_sizes[bl][H_PART_H4][0] = BSZ(s, q);
            h_block_sizes[bl][H_PART_V4][0] = BSZ(q, s);
        }
#undef BSZ
    }
}

/* --------------------------------------------------------------------------------------------- wedge / inter-intra masks */

static HostMasks g_masks;

/* one row of a master mask: 0 left of the 8-pixel transition centred on `ctr`, 64 right of it (AV1 spec 7.11.3.11) */
static int border_at(const uint8_t tbl[8], const int ctr, const int x) {
    const int k = x - (ctr - 4);
    return k < 0 ? 0 : k >= 8 ? 64 : tbl[k];
}

enum { W_HOR, W_VERT, W_OBL27, W_OBL63, W_OBL117, W_OBL153 };

static int master_at(const int dir, const int x, const int y) {
    static const uint8_t odd[8] = { 1, 2, 6, 18, 37, 53, 60, 63 }, even[8] = { 1, 4, 11, 27, 46, 58, 62, 63 },
                         vert[8] = { 0, 2, 7, 21, 43, 57, 62, 64 };
    switch (dir) {
    case W_VERT:   return border_at(vert, 32, x);
    case W_HOR:    return border_at(vert, 32, y);
    case W_OBL63:  return border_at((y & 1) ? odd : even, 48 - (y >> 1) - (y & 1), x);
    case W_OBL27:  return master_at(W_OBL63, y, x);
    case W_OBL117: return master_at(W_OBL63, 63 - x, y);
    default:       return master_at(W_OBL27, 63 - x, y);
    }
}

static void masks_build(void) {
    /* Wedge_Codebook (AV1 spec 7.11.3.11): { direction, x offset, y offset } in eighths of the block, per shape class */
    static const uint8_t cb_hgtw[16][3] = {
        { W_OBL27, 4, 4 }, { W_OBL63, 4, 4 }, { W_OBL117, 4, 4 }, { W_OBL153, 4, 4 }, { W_HOR, 4, 2 }, { W_HOR, 4, 4 },
        { W_HOR, 4, 6 }, { W_VERT, 4, 4 }, { W_OBL27, 4, 2 }, { W_OBL27, 4, 6 }, { W_OBL153, 4, 2 }, { W_OBL153, 4, 6 },
        { W_OBL63, 2, 4 }, { W_OBL63, 6, 4 }, { W_OBL117, 2, 4 }, { W_OBL117, 6, 4 } };
    static const uint8_t cb_hltw[16][3] = {
        { W_OBL27, 4, 4 }, { W_OBL63, 4, 4 }, { W_OBL117, 4, 4 }, { W_OBL153, 4, 4 }, { W_VERT, 2, 4 }, { W_VERT, 4, 4 },
        { W_VERT, 6, 4 }, { W_HOR, 4, 4 }, { W_OBL27, 4, 2 }, { W_OBL27, 4, 6 }, { W_OBL153, 4, 2 }, { W_OBL153, 4, 6 },
        { W_OBL63, 2, 4 }, { W_OBL63, 6, 4 }, { W_OBL117, 2, 4 }, { W_OBL117, 6, 4 } };
    static const uint8_t cb_heqw[16][3] = {
        { W_OBL27, 4, 4 }, { W_OBL63, 4, 4 }, { W_OBL117, 4, 4 }, { W_OBL153, 4, 4 }, { W_HOR, 4, 2 }, { W_HOR, 4, 6 },
        { W_VERT, 2, 4 }, { W_VERT, 6, 4 }, { W_OBL27, 4, 2 }, { W_OBL27, 4, 6 }, { W_OBL153, 4, 2 }, { W_OBL153, 4, 6 },
        { W_OBL63, 2, 4 }, { W_OBL63, 6, 4 }, { W_OBL117, 2, 4 }, { W_OBL117, 6, 4 } };
    /* Ii_Weights_1d sampled for a 32-pixel side (AV1 spec 7.11.3.13) */
    static const uint8_t ii_w[32] = { 60, 52, 45, 39, 34, 30, 26, 22, 19, 17, 15, 13, 11, 10, 8, 7,
                                      6, 6, 5, 4, 4, 3, 3, 2, 2, 2, 2, 1, 1, 1, 1, 1 };
    HostMasks *const m = &g_masks;
    size_t cap = 1 << 20, pos = 0;
    m->blob = (uint8_t *) malloc(cap);
    memset(m->wedge, 0xff, sizeof(m->wedge));
    memset(m->ii, 0xff, sizeof(m->ii));
    /* ---- wedge: the nine block sizes 32x32 .. 8x8 (both sides 8 .. 32) */
    for (int bs = H_BS_32x32; bs <= H_BS_8x8; bs++) {
        const int w = h_bs_dim[bs][0] * 4, h = h_bs_dim[bs][1] * 4;
        if (w < 8 || h < 8 || w > 32 || h > 32) continue;
        const uint8_t (*const cb)[3] = h > w ? cb_hgtw : h < w ? cb_hltw : cb_heqw;
        for (int n = 0; n < 16; n++) {
            const int dir = cb[n][0], x0 = 32 - (w * cb[n][1] >> 3), y0 = 32 - (h * cb[n][2] >> 3);
            /* the sign is flipped when the average of the mask's first row and first column is below one half */
            int sum = 0;
            for (int x = 0; x < w; x++) sum += master_at(dir, x0 + x, y0);
            for (int y = 1; y < h; y++) sum += master_at(dir, x0, y0 + y);
            const int flip = (sum + (w + h - 1) / 2) / (w + h - 1) < 32;
            uint8_t *const luma = m->blob + pos;
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    const int v = master_at(dir, x0 + x, y0 + y);
                    luma[y * w + x] = (uint8_t) (flip ? 64 - v : v);
                }
            m->wedge[0][bs - H_BS_32x32][0][n] = m->wedge[0][bs - H_BS_32x32][1][n] = (uint32_t) pos;
            pos += (size_t) w * h;
            /* chroma resolutions: the mean of the covered luma samples, rounded down for sign 1 */
            for (int c = 1; c < 3; c++) {
                const int ss_ver = c == 2, cw = w >> 1, ch = h >> ss_ver;
                for (int sign = 0; sign < 2; sign++) {
                    uint8_t *const o = m->blob + pos;
                    for (int y = 0; y < ch; y++)
                        for (int x = 0; x < cw; x++) {
                            const uint8_t *const l = luma + (y << ss_ver) * w + 2 * x;
                            int s = l[0] + l[1] + 1;
                            if (ss_ver) s += l[w] + l[w + 1] + 1;
                            o[y * cw + x] = (uint8_t) ((s - sign) >> (1 + ss_ver));
                        }
                    m->wedge[c][bs - H_BS_32x32][sign][n] = (uint32_t) pos;
                    pos += (size_t) cw * ch;
                }
            }
        }
    }
    /* ---- inter-intra: DC = one half everywhere; the others fall off from the predicted edge, scaled to the longer side */
    const size_t dc_off = pos;
    memset(m->blob + pos, 32, 32 * 32);
    pos += 32 * 32;
    for (int bs = H_BS_32x32; bs <= H_BS_8x8; bs++) {
        const int w = h_bs_dim[bs][0] * 4, h = h_bs_dim[bs][1] * 4;
        if (w < 8 || h < 8 || w > 32 || h > 32 || w > 2 * h || h > 2 * w) continue;      /* inter-intra: 8x8 .. 32x32, up to 2:1 */
        for (int c = 0; c < 3; c++) {
            const int ss_hor = c != 0, ss_ver = c == 2, pw = w >> ss_hor, ph = h >> ss_ver;
            const int step = 32 / (pw > ph ? pw : ph);
            m->ii[c][bs - H_BS_32x32][0] = (uint32_t) dc_off;
            for (int mode = 1; mode < 4; mode++) {
                uint8_t *const o = m->blob + pos;
                for (int y = 0; y < ph; y++)
                    for (int x = 0; x < pw; x++)
                        o[y * pw + x] = ii_w[(mode == 1 ? y : mode == 2 ? x : (x < y ? x : y)) * step];
                m->ii[c][bs - H_BS_32x32][mode] = (uint32_t) pos;
                pos += (size_t) pw * ph;
            }
        }
    }
    m->size = (pos + 255) & ~(size_t) 255;
    memset(m->blob + pos, 0, m->size - pos);
    (void) cap;
}

static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_all(void) { tables_build(); masks_build(); }
void h_tables_init(void) { pthread_once(&g_once, init_all); }
const HostMasks *h_masks(void) { h_tables_init(); return &g_masks; }

/* --------------------------------------------------------------------------------------------- warp set-up */

static int iclip(const int v, const int lo, const int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* round to a multiple of 64 after saturating to 16 bits (AV1 spec 7.11.3.6, "the shear parameters are rounded") */
static int shear_round(const int v) {
    const int cv = iclip(v, INT16_MIN, INT16_MAX);
    const int a = ((cv < 0 ? -cv : cv) + 32) >> 6;
    return (cv < 0 ? -a : a) * 64;
}

/* Div_Lut of the AV1 spec (7.11.3.7): 2^14 * 256 / (256 + i), rounded */
static int div_lut(const int i) { return ((1 << 22) + ((256 + i) >> 1)) / (256 + i); }

int h_shear_params(Dav1dHipWarpParams *const wm) {
    const int32_t *const mat = wm->matrix;
    if (mat[2] <= 0) return 1;
    wm->u.p.alpha = (int16_t) shear_round(mat[2] - 0x10000);
    wm->u.p.beta = (int16_t) shear_round(mat[3]);
    /* reciprocal of mat[2]: leading bit + an 8-bit fraction looked up */
    const unsigned d = (unsigned) mat[2];
    int shift = 31 - __builtin_clz(d);
    const int e = (int) (d - (1u << shift));
    const int f = shift > 8 ? (e + (1 << (shift - 9))) >> (shift - 8) : e << (8 - shift);
    shift += 14;
    const int y = div_lut(f);
    const int64_t v1 = ((int64_t) mat[4] * 0x10000) * y;
    const int64_t rnd = ((int64_t) 1 << shift) >> 1;
    const int g = (int) (((v1 < 0 ? -v1 : v1) + rnd) >> shift);
    wm->u.p.gamma = (int16_t) shear_round(v1 < 0 ? -g : g);
    const int64_t v2 = ((int64_t) mat[3] * mat[4]) * y;
    const int dd = (int) (((v2 < 0 ? -v2 : v2) + rnd) >> shift);
    wm->u.p.delta = (int16_t) shear_round(mat[5] - (v2 < 0 ? -dd : dd) - 0x10000);
    return (4 * abs(wm->u.p.alpha) + 7 * abs(wm->u.p.beta) >= 0x10000) || (4 * abs(wm->u.p.gamma) + 4 * abs(wm->u.p.delta) >= 0x10000);
}

int h_block_warp(Dav1dHipWarpParams *const wm, const int16_t matrix[4], const int16_t mv2d[2], const int bw4, const int bh4,
                 const int bx4, const int by4)
{
    memset(wm, 0, sizeof(*wm));
    if (matrix[0] == INT16_MIN) { wm->type = H_WM_IDENTITY; return wm->type; }      /* pass 1 found no valid model */
    wm->type = H_WM_AFFINE;
    int32_t *const mat = wm->matrix;
    mat[2] = matrix[0] + 0x10000;
    mat[3] = matrix[1];
    mat[4] = matrix[2];
    mat[5] = matrix[3] + 0x10000;
    /* translation such that the block centre moves by mv2d (AV1 spec 7.10.4.2 / 7.11.3.8) */
    const int isuy = by4 * 4 + 2 * bh4 - 1, isux = bx4 * 4 + 2 * bw4 - 1;
    mat[0] = iclip(mv2d[1] * 0x2000 - (isux * (mat[2] - 0x10000) + isuy * mat[3]), -0x800000, 0x7fffff);
    mat[1] = iclip(mv2d[0] * 0x2000 - (isux * mat[4] + isuy * (mat[5] - 0x10000)), -0x800000, 0x7fffff);
    (void) h_shear_params(wm);
    return wm->type;
}

/* Synthetic pass-1 output: what dav1d's entropy-decoding pass would leave behind for one frame, without a bitstream.
 *
 * No AV1 streams and no encoder exist in the build or GPU environment, so the tests and bench.py
 * need another producer of the hand-off arrays the pass-2 lister consumes: Av1Block records at block origins, cbi / cf in
 * tile decode order, palettes and packed palette indices — laid out exactly as pass 1 of the reference leaves them
 * (src/decode.c:683-2115 decides the syntax elements, dav1d_read_coef_blocks src/recon_tmpl.c:824-936 the order of the
 * coefficient arrays).  Block decisions are drawn from a seeded generator under the legality rules of the AV1 syntax (which
 * tool may appear on which block size), so that every combination the reconstruction can meet does occur.
 * This is an input generator like dav1d_amd/synth.py, not part of the decode path and not part of the product library: TEST
 * INFRASTRUCTURE, built into tests/synth/libdav1d_synth.so.  Plain C99. */
#include "../../dav1d_amd/host/av1_host.h"
#include "dav1d_synth.h"
#include <errno.h>
#include <stdlib.h>
#include <string.h>

#define __device__
#include "../../dav1d_amd/csrc/av1_scan_dev.h"
#undef __device__

typedef struct Rng { uint32_t s[4]; } Rng;
static uint32_t rnd(Rng *r) {
    uint32_t t = r->s[3];
    const uint32_t s0 = r->s[0];
    r->s[3] = r->s[2]; r->s[2] = r->s[1]; r->s[1] = s0;
    t ^= t << 11; t ^= t >> 8;
    return r->s[0] = t ^ s0 ^ (s0 >> 19);
}
static int rnd_n(Rng *r, const int n) { return (int) (rnd(r) % (uint32_t) n); }
static int pct(Rng *r, const int p) { return p > 0 && rnd_n(r, 100) < p; }
static int rnd_range(Rng *r, const int lo, const int hi) { return lo + rnd_n(r, hi - lo + 1); }

static int imin(const int a, const int b) { return a < b ? a : b; }
static int imax(const int a, const int b) { return a > b ? a : b; }

typedef struct Gen {
    const Dav1dHipFrameDesc *d;
    const Dav1dSynthParams *sp;
    Dav1dHipAv1Block *b;
    int16_t *cbi;
    uint8_t *cf, *pal, *pal_idx;
    Rng rng;
    int ss_hor, ss_ver, bw, bh, hbd, csz, bdmax;
    size_t cur_cbi, cur_cf, cur_pal_idx;             /* cursors of the tile being generated */
    size_t end_cbi, end_cf, end_pal_idx;             /* capacity checks */
    int col_start, col_end, row_start, row_end;
    int cur_lossless;                                /* the block being generated lies in a lossless segment: 4x4 WHT only */
    int err;
} Gen;

static const unsigned cfl_allowed = (1u << H_BS_32x32) | (1u << H_BS_32x16) | (1u << H_BS_32x8) | (1u << H_BS_16x32) | (1u << H_BS_16x16) |
    (1u << H_BS_16x8) | (1u << H_BS_16x4) | (1u << H_BS_8x32) | (1u << H_BS_8x16) | (1u << H_BS_8x8) | (1u << H_BS_8x4) |
    (1u << H_BS_4x16) | (1u << H_BS_4x8) | (1u << H_BS_4x4);                  /* chroma up to 32x32 (AV1 spec 5.11.22) */
static const unsigned wedge_allowed = (1u << H_BS_32x32) | (1u << H_BS_32x16) | (1u << H_BS_32x8) | (1u << H_BS_16x32) | (1u << H_BS_16x16) |
    (1u << H_BS_16x8) | (1u << H_BS_8x32) | (1u << H_BS_8x16) | (1u << H_BS_8x8);       /* Wedge_Bits != 0 */
static const unsigned interintra_allowed = (1u << H_BS_32x32) | (1u << H_BS_32x16) | (1u << H_BS_16x32) | (1u << H_BS_16x16) |
    (1u << H_BS_16x8) | (1u << H_BS_8x16) | (1u << H_BS_8x8);

/* legal (size, type) pairs of the inverse transform: the 64-point sizes only know DCT, the 32-point ones DCT and identity,
 * 16x16 has no 1-D ADST variants (AV1 spec 5.11.47 get_tx_set / Tx_Type_In_Set) */
static int pick_txtp(Gen *g, const int tx, const int intra) {
    const HostTx *t = &h_tx[tx];
    if (!pct(&g->rng, g->sp->alt_txtp_pct)) return 0;
    if (t->max == 4) return 0;
    if (t->max == 3) return (!intra && pct(&g->rng, 50)) ? 9 : 0;
    if (t->w == 4 && t->h == 4) return rnd_n(&g->rng, 12);
    return rnd_n(&g->rng, 16);
}

/* one transform block's worth of cbi + cf: eob class, then small values on the scan positions 0 .. eob */
static void gen_tx(Gen *g, const int tx, const int intra) {
    const HostTx *t = &h_tx[tx];
    const int sw = imin(t->w, 8) * 4, sh = imin(t->h, 8) * 4, ncoef = sw * sh;
    const size_t bytes = (size_t) ncoef * g->csz;
    if (g->cur_cbi + 1 > g->end_cbi || g->cur_cf + bytes > g->end_cf) { g->err = -ENOSPC; return; }
    int eob, txtp = g->cur_lossless ? 16 /* WHT_WHT, src/recon_tmpl.c:347-360 */ : pick_txtp(g, tx, intra);
    const int cls = rnd_n(&g->rng, 100);
    if (cls < g->sp->eob_none_pct) eob = -1;
    else if (cls < g->sp->eob_none_pct + 30) eob = 0;
    else if (cls < g->sp->eob_none_pct + 70) eob = rnd_range(&g->rng, 1, imax(1, ncoef / 4 - 1));
    else eob = rnd_range(&g->rng, 1, ncoef - 1);
    if (eob >= 0) {
        /* coefficient index of scan position i: the zig-zag for 2-D types, slab order for the types that are 1-D
         * horizontally (H_DCT 11, H_ADST 13, H_FLIPADST 15), column-interleaved for the vertical ones (10, 12, 14) —
         * the orders decode_coefs() writes in (src/recon_tmpl.c:458-520, 548-575) */
        const int v_cls = txtp == 10 || txtp == 12 || txtp == 14, h_cls = txtp == 11 || txtp == 13 || txtp == 15;
        const int lsw = __builtin_ctz(sw);
        uint8_t *const slab = g->cf + g->cur_cf;
        for (int i = 0; i <= eob; i++) {
            const int pos = h_cls ? i : v_cls ? (i & (sw - 1)) * sh + (i >> lsw) : av1_scans[av1_scan_off[tx] + i];
            /* magnitudes fall off along the scan; the last one is never zero */
            const int mag = i == 0 ? (g->bdmax + 1) * 4 : imax(2, (g->bdmax + 1) >> (1 + (i > 4) + (i > 16) + (i > 64)));
            int v = rnd_range(&g->rng, -mag, mag);
            if (!v && (i == eob || pct(&g->rng, 50))) v = 1 + rnd_n(&g->rng, 3);
            if (g->hbd) ((int32_t *) slab)[pos] = v; else ((int16_t *) slab)[pos] = (int16_t) v;
        }
    }
    g->cbi[g->cur_cbi++] = (int16_t) (eob * 32 + txtp);
    g->cur_cf += bytes;
}

/* read_tx_tree(), src/decode.c:119-167: which transform blocks of an inter block split further */
static void gen_tx_split(Gen *g, uint16_t masks[2], const int from, const int depth, const int x_off, const int y_off, const int bx, const int by) {
    const HostTx *t = &h_tx[from];
    int is_split = 0;
    if (depth < 2 && from > H_TX_4X4) {
        is_split = pct(&g->rng, g->sp->tx_split_pct);
        if (is_split) masks[depth] |= (uint16_t) (1u << (y_off * 4 + x_off));
    }
    if (is_split && t->max > 1) {
        const int sub = t->sub, sw = h_tx[sub].w, sh = h_tx[sub].h;
        gen_tx_split(g, masks, sub, depth + 1, x_off * 2 + 0, y_off * 2 + 0, bx, by);
        if (t->w >= t->h && bx + sw < g->bw) gen_tx_split(g, masks, sub, depth + 1, x_off * 2 + 1, y_off * 2 + 0, bx + sw, by);
        if (t->h >= t->w && by + sh < g->bh) {
            gen_tx_split(g, masks, sub, depth + 1, x_off * 2 + 0, y_off * 2 + 1, bx, by + sh);
            if (t->w >= t->h && bx + sw < g->bw) gen_tx_split(g, masks, sub, depth + 1, x_off * 2 + 1, y_off * 2 + 1, bx + sw, by + sh);
        }
    }
}

/* read_coef_tree() in pass 1: the leaves in tree order */
static void gen_coef_tree(Gen *g, const Dav1dHipAv1Block *b, const int tx, const int depth, const int x_off, const int y_off, const int bx, const int by) {
    const HostTx *t = &h_tx[tx];
    const unsigned split = depth == 0 ? b->u.p.tx_split0 : b->u.p.tx_split1;
    if (depth < 2 && split && (split & (1u << (y_off * 4 + x_off)))) {
        const int sub = t->sub, sw = h_tx[sub].w, sh = h_tx[sub].h;
        gen_coef_tree(g, b, sub, depth + 1, x_off * 2 + 0, y_off * 2 + 0, bx, by);
        if (t->w >= t->h && bx + sw < g->bw) gen_coef_tree(g, b, sub, depth + 1, x_off * 2 + 1, y_off * 2 + 0, bx + sw, by);
        if (t->h >= t->w && by + sh < g->bh) {
            gen_coef_tree(g, b, sub, depth + 1, x_off * 2 + 0, y_off * 2 + 1, bx, by + sh);
            if (t->w >= t->h && bx + sw < g->bw) gen_coef_tree(g, b, sub, depth + 1, x_off * 2 + 1, y_off * 2 + 1, bx + sw, by + sh);
        }
    } else {
        gen_tx(g, tx, 0);
    }
}

/* dav1d_read_coef_blocks(), src/recon_tmpl.c:824-936: the order pass 1 fills cbi / cf in */
static void gen_coefs(Gen *g, const Dav1dHipAv1Block *b, const int bs, const int bx, const int by) {
    const int ss_hor = g->ss_hor, ss_ver = g->ss_ver;
    const int bw4 = h_bs_dim[bs][0], bh4 = h_bs_dim[bs][1];
    const int has_chroma = g->d->layout != DAV1D_HIP_LAYOUT_I400 && (bw4 > ss_hor || (bx & 1)) && (bh4 > ss_ver || (by & 1));
    if (b->skip) return;
    const int w4 = imin(bw4, g->bw - bx), h4 = imin(bh4, g->bh - by);
    const int cw4 = (w4 + ss_hor) >> ss_hor, ch4 = (h4 + ss_ver) >> ss_ver;
    const HostTx *uv_t = &h_tx[b->uvtx], *t = &h_tx[b->intra ? b->u.i.tx : b->u.p.max_ytx];
    for (int init_y = 0; init_y < h4; init_y += 16) {
        const int sub_h4 = imin(h4, 16 + init_y);
        for (int init_x = 0; init_x < w4; init_x += 16) {
            const int sub_w4 = imin(w4, init_x + 16);
            int y_off = !!init_y;
            for (int y = init_y; y < sub_h4; y += t->h, y_off++) {
                int x_off = !!init_x;
                for (int x = init_x; x < sub_w4; x += t->w, x_off++) {
                    if (!b->intra) gen_coef_tree(g, b, b->u.p.max_ytx, 0, x_off, y_off, bx + x, by + y);
                    else gen_tx(g, b->u.i.tx, 1);
                }
            }
            if (!has_chroma) continue;
            const int sub_ch4 = imin(ch4, (init_y + 16) >> ss_ver), sub_cw4 = imin(cw4, (init_x + 16) >> ss_hor);
            for (int pl = 0; pl < 2; pl++)
                for (int y = init_y >> ss_ver; y < sub_ch4; y += uv_t->h)
                    for (int x = init_x >> ss_hor; x < sub_cw4; x += uv_t->w) gen_tx(g, b->uvtx, b->intra);
        }
    }
}

static void gen_pal(Gen *g, const int bx, const int by, const int pl0, const int n_pl, const int pal_sz, const int w4, const int h4) {
    const size_t idx = (size_t) ((by >> 1) + (bx & 1)) * (size_t) (g->d->b4_stride >> 1) + (size_t) ((bx >> 1) + (by & 1));
    for (int pl = pl0; pl < pl0 + n_pl; pl++)
        for (int i = 0; i < 8; i++) {
            const int v = rnd_n(&g->rng, g->bdmax + 1);
            if (g->hbd) ((uint16_t *) g->pal)[(idx * 3 + pl) * 8 + i] = (uint16_t) v;
            else g->pal[(idx * 3 + pl) * 8 + i] = (uint8_t) v;
        }
    const size_t bytes = (size_t) w4 * h4 * 8;
    if (g->cur_pal_idx + bytes > g->end_pal_idx) { g->err = -ENOSPC; return; }
    for (size_t i = 0; i < bytes; i++) g->pal_idx[g->cur_pal_idx + i] = (uint8_t) (rnd_n(&g->rng, pal_sz) | (rnd_n(&g->rng, pal_sz) << 4));
    g->cur_pal_idx += bytes;
}

static int is_directional(const int m) { return m >= H_VERT_PRED && m <= H_VERT_LEFT_PRED; }

/* decode_b() of pass 1 with the symbol decoder replaced by the generator (src/decode.c:808-1960) */
static void gen_block(Gen *g, const int bl, const int bs, const int bp, const int bx, const int by) {
    const Dav1dSynthParams *sp = g->sp;
    Dav1dHipAv1Block *b = &g->b[(size_t) by * g->d->b4_stride + bx];
    const int ss_hor = g->ss_hor, ss_ver = g->ss_ver, layout = g->d->layout;
    const uint8_t *b_dim = h_bs_dim[bs];
    const int bw4 = b_dim[0], bh4 = b_dim[1];
    const int cbw4 = (bw4 + ss_hor) >> ss_hor, cbh4 = (bh4 + ss_ver) >> ss_ver;
    const int has_chroma = layout != DAV1D_HIP_LAYOUT_I400 && (bw4 > ss_hor || (bx & 1)) && (bh4 > ss_ver || (by & 1));
    memset(b, 0, sizeof(*b));
    b->bl = (uint8_t) bl; b->bs = (uint8_t) bs; b->bp = (uint8_t) bp;
    b->skip = (uint8_t) pct(&g->rng, sp->skip_pct);
    b->intra = (uint8_t) (!g->d->is_inter || pct(&g->rng, sp->intra_pct));
    /* segmentation (src/decode.c:808-870): nothing of pass 2 depends on the segment; the deblocking levels and the lossless rule of
     * the mask builders (src/decode.c:1216-1226, 1882-1900) do */
    if (sp->n_segs > 1) b->seg_id = (uint8_t) rnd_n(&g->rng, imin(sp->n_segs, 8));
    const int lossless = g->d->lossless[b->seg_id];
    g->cur_lossless = lossless;
    /* skip_mode (src/decode.c:872-886, 1399-1404): two fixed references averaged, no residual */
    const int skip_mode = g->d->is_inter && !b->intra && imin(bw4, bh4) > 1 && imax(1, imin(sp->n_refs, 7)) > 1 && pct(&g->rng, sp->skip_mode_pct);
    if (skip_mode) { b->skip_mode = 1; b->skip = 1; }
    int bc_dx = 0, bc_dy = 0;
    if (!g->d->is_inter && sp->intrabc_pct && imax(bw4, bh4) <= 16 && pct(&g->rng, sp->intrabc_pct)) {
        /* Intra block copy: a source for the block — and for the chroma of the whole 8x8 a 4-wide / 4-high block carries, one more
         * pixel to the right and below for the bilinear taps of half positions — inside the tile, in the superblock rows above this
         * one or at least 256 pixels to the left of this superblock: decoded before this block whatever the partition order
         * (the stream-level rule, is_mv_valid, is narrower).  Vectors are whole pixels, at most 4095 of them. */
        const int sb4 = g->d->sb128 ? 32 : 16;
        const int ex0 = (bx & ~ss_hor) * 4, ey0 = (by & ~ss_ver) * 4;
        const int ew = (((bx + bw4) * 4 + 7 * ss_hor) & ~(7 * ss_hor)) - ex0 + 1, eh = (((by + bh4) * 4 + 7 * ss_ver) & ~(7 * ss_ver)) - ey0 + 1;
        const int tx0 = g->col_start * 4, tx1 = imin(g->col_end, g->bw) * 4, ty0 = g->row_start * 4, ty1 = imin(g->row_end, g->bh) * 4;
        const int sb_top = (by & ~(sb4 - 1)) * 4, sb_left = (bx & ~(sb4 - 1)) * 4, sb_bot = imin(sb_top + sb4 * 4, ty1);
        int sx = -1, sy = -1;
        for (int tries = 0; tries < 8 && sx < 0; tries++) {
            if (rnd_n(&g->rng, 2) && sb_top - ty0 >= eh && tx1 - tx0 >= ew) {              /* above */
                sy = rnd_range(&g->rng, imax(ty0, ey0 - 4000), sb_top - eh);
                sx = rnd_range(&g->rng, imax(tx0, ex0 - 4000), imin(tx1 - ew, ex0 + 4000));
            } else if (sb_left - 256 - tx0 >= ew && sb_bot - ty0 >= eh) {                  /* to the left */
                sx = rnd_range(&g->rng, imax(tx0, ex0 - 4000), sb_left - 256 - ew);
                sy = rnd_range(&g->rng, imax(ty0, ey0 - 4000), imin(sb_bot - eh, ey0 + 4000));
            }
            if (sx >= 0 && (sx > sb_left - 256 - ew && sy > sb_top - eh)) sx = sy = -1;         /* ranges that came out empty */
        }
        if (sx >= 0) { b->intra = 0; bc_dx = sx - ex0; bc_dy = sy - ey0; }
    }
    if (b->intra) {
        b->u.i.y_mode = (uint8_t) rnd_n(&g->rng, 13);
        if (pct(&g->rng, 30)) b->u.i.y_mode = H_DC_PRED;
        if (b_dim[2] + b_dim[3] >= 2 && is_directional(b->u.i.y_mode)) b->u.i.y_angle = (int8_t) rnd_range(&g->rng, -3, 3);
        if (has_chroma) {
            const int cfl_ok = lossless ? cbw4 == 1 && cbh4 == 1 : (cfl_allowed >> bs) & 1;       /* src/decode.c:1072-1074 */
            b->u.i.uv_mode = (uint8_t) rnd_n(&g->rng, 13);
            if (pct(&g->rng, 25)) b->u.i.uv_mode = H_DC_PRED;
            if (cfl_ok && pct(&g->rng, sp->cfl_pct)) {
                b->u.i.uv_mode = H_CFL_PRED;
                /* joint sign: not both zero */
                int su, sv;
                do { su = rnd_n(&g->rng, 3); sv = rnd_n(&g->rng, 3); } while (!su && !sv);
                b->u.i.cfl_alpha[0] = (int8_t) (su ? (su == 1 ? -1 : 1) * rnd_range(&g->rng, 1, 16) : 0);
                b->u.i.cfl_alpha[1] = (int8_t) (sv ? (sv == 1 ? -1 : 1) * rnd_range(&g->rng, 1, 16) : 0);
            } else if (b_dim[2] + b_dim[3] >= 2 && is_directional(b->u.i.uv_mode)) {
                b->u.i.uv_angle = (int8_t) rnd_range(&g->rng, -3, 3);
            }
        }
        if (sp->palette && g->pal && imax(bw4, bh4) <= 16 && bw4 + bh4 >= 4) {
            if (b->u.i.y_mode == H_DC_PRED && pct(&g->rng, sp->palette)) b->u.i.pal_sz[0] = (uint8_t) rnd_range(&g->rng, 2, 8);
            if (has_chroma && b->u.i.uv_mode == H_DC_PRED && pct(&g->rng, sp->palette)) b->u.i.pal_sz[1] = (uint8_t) rnd_range(&g->rng, 2, 8);
        }
        if (b->u.i.y_mode == H_DC_PRED && !b->u.i.pal_sz[0] && imax(b_dim[2], b_dim[3]) <= 3 && pct(&g->rng, sp->filter_intra_pct)) {
            b->u.i.y_mode = H_FILTER_PRED;
            b->u.i.y_angle = (int8_t) rnd_n(&g->rng, 5);
        }
        if (b->u.i.pal_sz[0]) gen_pal(g, bx, by, 0, 1, b->u.i.pal_sz[0], bw4, bh4);
        if (has_chroma && b->u.i.pal_sz[1]) gen_pal(g, bx, by, 1, 2, b->u.i.pal_sz[1], cbw4, cbh4);
        /* transform size: the largest of the block, or up to two levels below it (TX_MODE_SELECT) */
        int tx = h_max_tx_for_bs[bs][0];
        if (sp->tx_split_pct)
            for (int depth = imin(h_tx[tx].max, 2); depth > 0 && tx != H_TX_4X4; depth--)
                if (pct(&g->rng, sp->tx_split_pct)) tx = h_tx[tx].sub;
        b->u.i.tx = (uint8_t) tx;
        b->uvtx = h_max_tx_for_bs[bs][layout];
        if (lossless) b->u.i.tx = b->uvtx = H_TX_4X4;                       /* src/decode.c:1186-1188 */
    } else if (!g->d->is_inter) {
        /* intra block copy: reference "0" = the frame itself, one whole-pixel vector, bilinear, no inter tools (src/decode.c:1258-1330) */
        b->u.p.ref[0] = 0; b->u.p.ref[1] = -1;
        b->u.p.u.m.mv[0][0] = (int16_t) (bc_dy * 8);
        b->u.p.u.m.mv[0][1] = (int16_t) (bc_dx * 8);
        b->u.p.comp_type = H_COMP_INTER_NONE;
        b->u.p.inter_mode = 3;
        b->u.p.filter2d = 9;                                               /* FILTER_2D_BILINEAR */
        uint16_t masks[2] = { 0, 0 };
        b->u.p.max_ytx = h_max_tx_for_bs[bs][0];
        b->uvtx = h_max_tx_for_bs[bs][layout];
        if (!b->skip && (lossless || b->u.p.max_ytx == H_TX_4X4)) {         /* src/decode.c:456-459 */
            b->u.p.max_ytx = b->uvtx = H_TX_4X4;
        } else if (!b->skip && sp->tx_split_pct) {
            const HostTx *ytx = &h_tx[b->u.p.max_ytx];
            for (int y = 0, y_off = 0; y < bh4; y += ytx->h, y_off++)
                for (int x = 0, x_off = 0; x < bw4; x += ytx->w, x_off++)
                    gen_tx_split(g, masks, b->u.p.max_ytx, 0, x_off, y_off, bx + x, by + y);
        }
        b->u.p.tx_split0 = (uint8_t) masks[0];
        b->u.p.tx_split1 = masks[1];
    } else {
        const int n_refs = imax(1, imin(sp->n_refs, 7));
        int is_comp = skip_mode || (imin(bw4, bh4) > 1 && n_refs > 1 && pct(&g->rng, sp->compound_pct));
        b->u.p.ref[0] = (int8_t) rnd_n(&g->rng, n_refs);
        b->u.p.ref[1] = -1;
        for (int i = 0; i < 2; i++) {
            int my = rnd_range(&g->rng, -sp->mv_range, sp->mv_range), mx = rnd_range(&g->rng, -sp->mv_range, sp->mv_range);
            if (pct(&g->rng, sp->far_mv_pct)) { my *= 24; mx *= 24; }      /* far outside the picture: edge emulation */
            if (pct(&g->rng, 15)) my &= ~7;                                /* integer positions happen too */
            if (pct(&g->rng, 15)) mx &= ~7;
            b->u.p.u.m.mv[i][0] = (int16_t) imax(-16000, imin(16000, my));
            b->u.p.u.m.mv[i][1] = (int16_t) imax(-16000, imin(16000, mx));
        }
        b->u.p.inter_mode = 3;                                             /* NEWMV / NEWMV_NEWMV-like: no meaning in pass 2 */
        if (is_comp) {
            do { b->u.p.ref[1] = (int8_t) rnd_n(&g->rng, n_refs); } while (b->u.p.ref[1] == b->u.p.ref[0]);
            const int k = rnd_n(&g->rng, 100);
            b->u.p.comp_type = k < 40 ? H_COMP_INTER_AVG : k < 60 ? H_COMP_INTER_WEIGHTED_AVG : k < 80 ? H_COMP_INTER_SEG : H_COMP_INTER_WEDGE;
            if (!sp->masked_compound && b->u.p.comp_type >= H_COMP_INTER_SEG) b->u.p.comp_type = H_COMP_INTER_AVG;
            if (b->u.p.comp_type == H_COMP_INTER_WEDGE && !((wedge_allowed >> bs) & 1)) b->u.p.comp_type = H_COMP_INTER_SEG;
            if (skip_mode) b->u.p.comp_type = H_COMP_INTER_AVG;
            b->u.p.u.m.wedge_idx = (uint8_t) rnd_n(&g->rng, 16);
            b->u.p.u.m.mask_sign = (uint8_t) rnd_n(&g->rng, 2);
            b->u.p.inter_mode = 7;
            if (pct(&g->rng, sp->global_pct)) b->u.p.inter_mode = H_GLOBALMV_GLOBALMV;
        } else {
            b->u.p.comp_type = H_COMP_INTER_NONE;
            if (pct(&g->rng, sp->global_pct)) b->u.p.inter_mode = H_GLOBALMV;
            const int global_warp = b->u.p.inter_mode == H_GLOBALMV && g->d->gmv[b->u.p.ref[0]].type > H_WM_TRANSLATION;
            if (((interintra_allowed >> bs) & 1) && pct(&g->rng, sp->interintra_pct)) {
                b->u.p.u.m.interintra_mode = (uint8_t) rnd_n(&g->rng, 4);
                b->u.p.interintra_type = (uint8_t) (H_INTER_INTRA_BLEND + rnd_n(&g->rng, 2));
                b->u.p.u.m.wedge_idx = (uint8_t) rnd_n(&g->rng, 16);
            }
            if (!b->u.p.interintra_type && imin(bw4, bh4) >= 2 && !global_warp) {
                const int k = rnd_n(&g->rng, 100);
                /* warped motion needs a reference of the frame's own size (src/decode.c:1783-1785) */
                const int ref_same = !g->d->svc[b->u.p.ref[0]][0][0] && !g->d->svc[b->u.p.ref[0]][1][0];
                if (k < sp->obmc_pct) {
                    b->u.p.motion_mode = H_MM_OBMC;
                } else if (k < sp->obmc_pct + sp->warp_pct && ref_same) {
                    /* a local warp model: near-identity matrix with a valid shear, or "no valid model found" */
                    b->u.p.motion_mode = H_MM_WARP;
                    const int16_t mv2d[2] = { b->u.p.u.m.mv[0][0], b->u.p.u.m.mv[0][1] };
                    int16_t m[4];
                    int tries = 0;
                    Dav1dHipWarpParams wm;
                    do {
                        const int amp = 3000 >> imin(tries, 6);
                        for (int i = 0; i < 4; i++) m[i] = (int16_t) rnd_range(&g->rng, -amp, amp);
                        h_block_warp(&wm, m, mv2d, bw4, bh4, bx, by);
                    } while (h_shear_params(&wm) && ++tries < 16);
                    if (tries == 16 || pct(&g->rng, 10)) m[0] = INT16_MIN;
                    b->u.p.u.w.mv2d[0] = mv2d[0]; b->u.p.u.w.mv2d[1] = mv2d[1];
                    memcpy(b->u.p.u.w.matrix, m, sizeof(m));
                }
            }
        }
        b->u.p.filter2d = (uint8_t) (pct(&g->rng, 60) ? 0 : rnd_n(&g->rng, 10));
        /* read_vartx_tree(), src/decode.c:445-492 */
        uint16_t masks[2] = { 0, 0 };
        b->u.p.max_ytx = h_max_tx_for_bs[bs][0];
        b->uvtx = h_max_tx_for_bs[bs][layout];
        if (!b->skip && (lossless || b->u.p.max_ytx == H_TX_4X4)) {         /* src/decode.c:456-459 */
            b->u.p.max_ytx = b->uvtx = H_TX_4X4;
        } else if (!b->skip && sp->tx_split_pct) {
            const HostTx *ytx = &h_tx[b->u.p.max_ytx];
            for (int y = 0, y_off = 0; y < bh4; y += ytx->h, y_off++)
                for (int x = 0, x_off = 0; x < bw4; x += ytx->w, x_off++)
                    gen_tx_split(g, masks, b->u.p.max_ytx, 0, x_off, y_off, bx + x, by + y);
        }
        b->u.p.tx_split0 = (uint8_t) masks[0];
        b->u.p.tx_split1 = masks[1];
    }
    gen_coefs(g, b, bs, bx, by);
}

/* decode_sb() of pass 1: the partition tree (src/decode.c:2117-2375) */
static void gen_sb(Gen *g, const int bl, const int bx, const int by) {
    if (g->err) return;
    const int hsz = 16 >> bl;
    const int have_h = g->bw > bx + hsz, have_v = g->bh > by + hsz;
    const int no_tall = g->d->layout == DAV1D_HIP_LAYOUT_I422;     /* 4:2:2 has no partitions with tall chroma blocks */
    if (!have_h && !have_v) { gen_sb(g, bl + 1, bx, by); return; }
    const uint8_t (*sz)[2] = h_block_sizes[bl];
#define B(k, x, y, p) gen_block(g, bl, sz[p][k], p, x, y)
    if (have_h && have_v) {
        int bp;
        if (g->sp->fixed_bl >= 0) {
            bp = bl < g->sp->fixed_bl ? H_PART_SPLIT : H_PART_NONE;
        } else if (pct(&g->rng, g->sp->split_pct[bl])) {
            bp = H_PART_SPLIT;
        } else if (pct(&g->rng, g->sp->rect_pct)) {
            static const uint8_t cand[8] = { H_PART_H, H_PART_V, H_PART_T_TOP_SPLIT, H_PART_T_BOTTOM_SPLIT, H_PART_T_LEFT_SPLIT,
                                             H_PART_T_RIGHT_SPLIT, H_PART_H4, H_PART_V4 };
            for (;;) {
                bp = cand[rnd_n(&g->rng, 8)];
                if (bl == H_BL_8X8 && bp > H_PART_V) continue;
                if (bl == H_BL_128X128 && bp >= H_PART_H4) continue;
                if (no_tall && (bp == H_PART_V || bp == H_PART_V4 || bp == H_PART_T_LEFT_SPLIT || bp == H_PART_T_RIGHT_SPLIT)) continue;
                break;
            }
        } else {
            bp = H_PART_NONE;
        }
        switch (bp) {
        case H_PART_NONE: B(0, bx, by, bp); break;
        case H_PART_H: B(0, bx, by, bp); B(0, bx, by + hsz, bp); break;
        case H_PART_V: B(0, bx, by, bp); B(0, bx + hsz, by, bp); break;
        case H_PART_SPLIT:
            if (bl == H_BL_8X8) {
                gen_block(g, bl, H_BS_4x4, bp, bx, by); gen_block(g, bl, H_BS_4x4, bp, bx + 1, by);
                gen_block(g, bl, H_BS_4x4, bp, bx, by + 1); gen_block(g, bl, H_BS_4x4, bp, bx + 1, by + 1);
                if (g->sp->cf_align64) g->cur_cf = (g->cur_cf + 63) & ~(size_t) 63;     /* src/decode.c:2209-2218 */
            } else {
                gen_sb(g, bl + 1, bx, by); gen_sb(g, bl + 1, bx + hsz, by);
                gen_sb(g, bl + 1, bx, by + hsz); gen_sb(g, bl + 1, bx + hsz, by + hsz);
            }
            break;
        case H_PART_T_TOP_SPLIT: B(0, bx, by, bp); B(0, bx + hsz, by, bp); B(1, bx, by + hsz, bp); break;
        case H_PART_T_BOTTOM_SPLIT: B(0, bx, by, bp); B(1, bx, by + hsz, bp); B(1, bx + hsz, by + hsz, bp); break;
        case H_PART_T_LEFT_SPLIT: B(0, bx, by, bp); B(0, bx, by + hsz, bp); B(1, bx + hsz, by, bp); break;
        case H_PART_T_RIGHT_SPLIT: B(0, bx, by, bp); B(1, bx + hsz, by, bp); B(1, bx + hsz, by + hsz, bp); break;
        case H_PART_H4:
            B(0, bx, by, bp); B(0, bx, by + (hsz >> 1), bp); B(0, bx, by + hsz, bp);
            if (by + (hsz * 3 >> 1) < g->bh) B(0, bx, by + (hsz * 3 >> 1), bp);
            break;
        default: /* H_PART_V4 */
            B(0, bx, by, bp); B(0, bx + (hsz >> 1), by, bp); B(0, bx + hsz, by, bp);
            if (bx + (hsz * 3 >> 1) < g->bw) B(0, bx + (hsz * 3 >> 1), by, bp);
        }
    } else if (have_h) {            /* bottom frame edge: split, or one horizontal half */
        if (pct(&g->rng, 50)) { gen_sb(g, bl + 1, bx, by); gen_sb(g, bl + 1, bx + hsz, by); }
        else B(0, bx, by, H_PART_H);
    } else {                        /* right frame edge */
        if (no_tall || pct(&g->rng, 50)) { gen_sb(g, bl + 1, bx, by); gen_sb(g, bl + 1, bx, by + hsz); }
        else B(0, bx, by, H_PART_V);
    }
#undef B
}

/* Fills b / cbi / cf / pal / pal_idx (the arrays desc points to, sized as dav1d_decode_frame_init() sizes them, cf zeroed)
 * for the whole frame.  cf_bytes / cbi_entries / pal_idx_bytes: the capacities, for overflow checks. */
int dav1d_synth_frame(const Dav1dHipFrameDesc *d, const Dav1dSynthParams *sp, void *cf, size_t cf_bytes, size_t cbi_entries,
                          uint8_t *pal_idx, size_t pal_idx_bytes)
{
    if (!d || !sp || !d->b || !d->cbi || !cf || !d->tile_start_off) return -EINVAL;
    h_tables_init();
    Gen g;
    memset(&g, 0, sizeof(g));
    g.d = d; g.sp = sp;
    g.b = (Dav1dHipAv1Block *) d->b;          /* this IS the producer of the arrays the descriptor calls const */
    g.cbi = (int16_t *) d->cbi;
    g.cf = (uint8_t *) cf;
    g.pal = (uint8_t *) d->pal;
    g.pal_idx = pal_idx;
    g.ss_ver = d->layout == DAV1D_HIP_LAYOUT_I420;
    g.ss_hor = d->layout != DAV1D_HIP_LAYOUT_I444;
    g.bw = ((d->w + 7) >> 3) << 1; g.bh = ((d->h + 7) >> 3) << 1;
    g.hbd = d->bpc > 8;
    g.csz = g.hbd ? 4 : 2;
    g.bdmax = (1 << d->bpc) - 1;
    g.rng.s[0] = (uint32_t) sp->seed | 1; g.rng.s[1] = (uint32_t) (sp->seed >> 32) ^ 0x9e3779b9u; g.rng.s[2] = 0x85ebca6bu; g.rng.s[3] = 0xc2b2ae35u;
    for (int i = 0; i < 16; i++) rnd(&g.rng);
    static const uint8_t size_mul[4][2] = { { 4, 4 }, { 6, 5 }, { 8, 6 }, { 12, 8 } };
    const int sb_shift = d->sb128 ? 5 : 4, sb_step = 1 << sb_shift;
    const int n_tiles = d->n_tile_cols * d->n_tile_rows;
    for (int tr = 0; tr < d->n_tile_rows; tr++)
        for (int tc = 0; tc < d->n_tile_cols; tc++) {
            const int t = tr * d->n_tile_cols + tc;
            const size_t off = d->tile_start_off[t];
            const size_t nxt = t + 1 < n_tiles ? d->tile_start_off[t + 1] : (size_t) -1;
            g.cur_pal_idx = off * size_mul[d->layout][1] / 8;
            g.cur_cbi = off * size_mul[d->layout][0] / 64;
            g.cur_cf = (off * size_mul[d->layout][0]) >> !g.hbd;
            g.end_pal_idx = nxt == (size_t) -1 ? pal_idx_bytes : nxt * size_mul[d->layout][1] / 8;
            if (g.end_pal_idx > pal_idx_bytes) g.end_pal_idx = pal_idx_bytes;
            g.end_cbi = nxt == (size_t) -1 ? cbi_entries : nxt * size_mul[d->layout][0] / 64;
            g.end_cf = nxt == (size_t) -1 ? cf_bytes : (nxt * size_mul[d->layout][0]) >> !g.hbd;
            if (g.end_cbi > cbi_entries) g.end_cbi = cbi_entries;
            if (g.end_cf > cf_bytes) g.end_cf = cf_bytes;
            g.col_start = d->col_start_sb[tc] << sb_shift; g.col_end = imin(d->col_start_sb[tc + 1] << sb_shift, g.bw);
            g.row_start = d->row_start_sb[tr] << sb_shift; g.row_end = imin(d->row_start_sb[tr + 1] << sb_shift, g.bh);
            for (int by = g.row_start; by < g.row_end; by += sb_step)
                for (int bx = g.col_start; bx < g.col_end; bx += sb_step)
                    gen_sb(&g, d->sb128 ? H_BL_128X128 : H_BL_64X64, bx, by);
            if (g.err) return g.err;
        }
    return 0;
}

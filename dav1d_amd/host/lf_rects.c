/* Deblocking masks and levels from the pass-1 hand-off, host half: the block walk.
 *
 * The reference builds Av1Filter masks and the level cache block by block while it parses (dav1d_create_lf_mask_intra /
 * _inter, src/lf_mask.c:259-383, called from decode_b, src/decode.c:1216-1226, 1882-1900).  What those functions compute is a
 * property of the transform grid: along every 4-pixel unit of a transform (or block) edge the filter size class is the
 * smaller of the two transforms meeting there, capped at 16 pixels (luma) / 8 (chroma); the level cache holds, per 4x4, the
 * level of the block covering it.  So the walk here only RASTERISES: it emits one rectangle per transform block (a whole
 * skipped inter block counts as one: the reference leaves the edges inside it alone, src/lf_mask.c:108-141, 236-257) with
 * its size classes and levels, and the device (csrc/lfmask.hip) paints them into a cell map and reads the masks off it.
 * Plain C99, no HIP. */
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include "av1_host.h"
#include "lister_priv.h"

typedef struct RectOut { Dav1dHipLfRect *p; size_t n, cap; int err; } RectOut;
typedef struct LfWalk {
    const Dav1dHipFrameDesc *d;
    const uint8_t (*lflvl)[4][8][2];
    const uint8_t (*sb_lflvl)[8][4][8][2];       /* delta_lf: one table per superblock, raster order (NULL: the frame's table) */
    int sb_shift, sbw;
    RectOut *o;
    int w4, h4, bw, bh, ss_hor, ss_ver;
} LfWalk;

static int imin_(const int a, const int b) { return a < b ? a : b; }

static void push(RectOut *o, const int kind, const int x4, const int y4, const int w4, const int h4, const int cw, const int ch,
                 const int edges, const int l0, const int l1)
{
    if (o->err) return;
    if (o->n == o->cap) {
        const size_t cap = o->cap ? o->cap * 2 : 4096;
        Dav1dHipLfRect *p = (Dav1dHipLfRect *) realloc(o->p, cap * sizeof(*p));
        if (!p) { o->err = -ENOMEM; return; }
        o->p = p; o->cap = cap;
    }
    Dav1dHipLfRect *r = &o->p[o->n++];
    memset(r, 0, sizeof(*r));
    r->x4 = (uint16_t) x4; r->y4 = (uint16_t) y4; r->w4 = (uint8_t) w4; r->h4 = (uint8_t) h4;
    r->cls = (uint8_t) (cw | ch << 2 | edges << 4);
    r->kind = (uint8_t) kind;
    r->lvl[0] = (uint8_t) l0; r->lvl[1] = (uint8_t) l1;
}

/* the leaves of the transform tree of one inter block (decomp_tx, src/lf_mask.c:39-79) */
static void tx_leaves(const LfWalk *w, const Dav1dHipAv1Block *b, const int tx, const int depth, const int x_off, const int y_off,
                      const int x, const int y, const int bx, const int by, const int bw4, const int bh4, const int l0, const int l1)
{
    const HostTx *t = &h_tx[tx];
    if (x >= bw4 || y >= bh4) return;
    const unsigned split = depth == 0 ? b->u.p.tx_split0 : b->u.p.tx_split1;
    if (tx != H_TX_4X4 && depth < 2 && ((split >> (y_off * 4 + x_off)) & 1)) {
        const int sub = t->sub, hw = t->w >> 1, hh = t->h >> 1;
        tx_leaves(w, b, sub, depth + 1, x_off * 2, y_off * 2, x, y, bx, by, bw4, bh4, l0, l1);
        if (t->w >= t->h) tx_leaves(w, b, sub, depth + 1, x_off * 2 + 1, y_off * 2, x + hw, y, bx, by, bw4, bh4, l0, l1);
        if (t->h >= t->w) {
            tx_leaves(w, b, sub, depth + 1, x_off * 2, y_off * 2 + 1, x, y + hh, bx, by, bw4, bh4, l0, l1);
            if (t->w >= t->h) tx_leaves(w, b, sub, depth + 1, x_off * 2 + 1, y_off * 2 + 1, x + hw, y + hh, bx, by, bw4, bh4, l0, l1);
        }
    } else {
        push(w->o, DAV1D_HIP_LF_RECT_LUMA, bx + x, by + y, imin_(t->w, bw4 - x), imin_(t->h, bh4 - y), imin_(2, t->lw), imin_(2, t->lh), 3, l0, l1);
    }
}

static void tiles(const LfWalk *w, const int kind, const int tx, const int x4, const int y4, const int bw4, const int bh4, const int cap,
                  const int l0, const int l1)
{
    const HostTx *t = &h_tx[tx];
    for (int y = 0; y < bh4; y += t->h)
        for (int x = 0; x < bw4; x += t->w)
            push(w->o, kind, x4 + x, y4 + y, imin_(t->w, bw4 - x), imin_(t->h, bh4 - y), imin_(cap, t->lw), imin_(cap, t->lh), 3, l0, l1);
}

static void lf_block(const LfWalk *w, const int bs, const int bx, const int by) {
    const Dav1dHipFrameDesc *d = w->d;
    const Dav1dHipAv1Block *b = &d->b[(size_t) by * d->b4_stride + bx];
    const int bw4 = imin_(w->w4 - bx, h_bs_dim[bs][0]), bh4 = imin_(w->h4 - by, h_bs_dim[bs][1]);
    const int inter = !b->intra;
    /* which entry of the level table: [segment][plane][reference + 1][mode is not GLOBALMV], src/decode.c:1882-1890 */
    int ref = 0, nz = 0;
    if (inter) {
        const int is_comp = b->u.p.comp_type != H_COMP_INTER_NONE;
        ref = b->u.p.ref[0] + 1;
        nz = b->u.p.inter_mode != (is_comp ? H_GLOBALMV_GLOBALMV : H_GLOBALMV);
    }
    /* delta_lf: the levels of a block come from the table its superblock was parsed with (ts->lflvlmem, src/decode.c:1180-1206) */
    const uint8_t (*const tab)[4][8][2] = w->sb_lflvl ? w->sb_lflvl[(by >> w->sb_shift) * w->sbw + (bx >> w->sb_shift)] : w->lflvl;
    const uint8_t (*lv)[8][2] = tab[b->seg_id];
    const int l0 = lv[0][ref][nz], l1 = lv[1][ref][nz], l2 = lv[2][ref][nz], l3 = lv[3][ref][nz];
    if (bw4 > 0 && bh4 > 0) {
        if (!inter) tiles(w, DAV1D_HIP_LF_RECT_LUMA, b->u.i.tx, bx, by, bw4, bh4, 2, l0, l1);
        else if (b->skip) {
            /* a lossless segment: 4x4 whatever the record says (src/decode.c:1890-1893; a skipped block's max_ytx is the block's largest) */
            const HostTx *t = &h_tx[d->lossless[b->seg_id & 7] ? H_TX_4X4 : b->u.p.max_ytx];
            push(w->o, DAV1D_HIP_LF_RECT_LUMA, bx, by, bw4, bh4, imin_(2, t->lw), imin_(2, t->lh), 3, l0, l1);
        } else {
            const HostTx *t = &h_tx[b->u.p.max_ytx];
            for (int y = 0, yo = 0; y < bh4; y += t->h, yo++)
                for (int x = 0, xo = 0; x < bw4; x += t->w, xo++)
                    tx_leaves(w, b, b->u.p.max_ytx, 0, xo, yo, x, y, bx, by, bw4, bh4, l0, l1);
        }
    }
    /* blocks with coefficients, for CDEF (src/decode.c:1945-1956): unclipped, as the reference marks them */
    if (!b->skip) push(w->o, DAV1D_HIP_LF_RECT_NOSKIP, bx, by, h_bs_dim[bs][0], h_bs_dim[bs][1], 0, 0, 0, 0, 0);
    if (d->layout == DAV1D_HIP_LAYOUT_I400) return;
    const int ss_hor = w->ss_hor, ss_ver = w->ss_ver;
    const int has_chroma = (h_bs_dim[bs][0] > ss_hor || (bx & 1)) && (h_bs_dim[bs][1] > ss_ver || (by & 1));
    if (!has_chroma) return;
    const int cbw4 = imin_(((w->w4 + ss_hor) >> ss_hor) - (bx >> ss_hor), (h_bs_dim[bs][0] + ss_hor) >> ss_hor);
    const int cbh4 = imin_(((w->h4 + ss_ver) >> ss_ver) - (by >> ss_ver), (h_bs_dim[bs][1] + ss_ver) >> ss_ver);
    if (cbw4 <= 0 || cbh4 <= 0) return;
    if (inter && b->skip) {
        const HostTx *t = &h_tx[d->lossless[b->seg_id & 7] ? H_TX_4X4 : b->uvtx];
        push(w->o, DAV1D_HIP_LF_RECT_CHROMA, bx >> ss_hor, by >> ss_ver, cbw4, cbh4, imin_(1, t->lw), imin_(1, t->lh), 3, l2, l3);
    } else {
        tiles(w, DAV1D_HIP_LF_RECT_CHROMA, b->uvtx, bx >> ss_hor, by >> ss_ver, cbw4, cbh4, 1, l2, l3);
    }
}

/* decode_sb() with pass == 2, src/decode.c:2117-2375: the partition tree only */
static void walk(const LfWalk *w, const int bl, const int bx, const int by) {
    if (w->o->err) return;
    const int hsz = 16 >> bl;
    const int have_h = w->bw > bx + hsz, have_v = w->bh > by + hsz;
    if (!have_h && !have_v) { walk(w, bl + 1, bx, by); return; }
    const Dav1dHipAv1Block *b = &w->d->b[(size_t) by * w->d->b4_stride + bx];
    const uint8_t (*sz)[2] = h_block_sizes[bl];
    if (have_h && have_v) {
        const int bp = b->bl == bl ? b->bp : H_PART_SPLIT;
        switch (bp) {
        case H_PART_NONE: lf_block(w, sz[bp][0], bx, by); break;
        case H_PART_H: lf_block(w, sz[bp][0], bx, by); lf_block(w, sz[bp][0], bx, by + hsz); break;
        case H_PART_V: lf_block(w, sz[bp][0], bx, by); lf_block(w, sz[bp][0], bx + hsz, by); break;
        case H_PART_SPLIT:
            if (bl == H_BL_8X8) {
                lf_block(w, H_BS_4x4, bx, by); lf_block(w, H_BS_4x4, bx + 1, by);
                lf_block(w, H_BS_4x4, bx, by + 1); lf_block(w, H_BS_4x4, bx + 1, by + 1);
            } else {
                walk(w, bl + 1, bx, by); walk(w, bl + 1, bx + hsz, by); walk(w, bl + 1, bx, by + hsz); walk(w, bl + 1, bx + hsz, by + hsz);
            }
            break;
        case H_PART_T_TOP_SPLIT:
            lf_block(w, sz[bp][0], bx, by); lf_block(w, sz[bp][0], bx + hsz, by); lf_block(w, sz[bp][1], bx, by + hsz); break;
        case H_PART_T_BOTTOM_SPLIT:
            lf_block(w, sz[bp][0], bx, by); lf_block(w, sz[bp][1], bx, by + hsz); lf_block(w, sz[bp][1], bx + hsz, by + hsz); break;
        case H_PART_T_LEFT_SPLIT:
            lf_block(w, sz[bp][0], bx, by); lf_block(w, sz[bp][0], bx, by + hsz); lf_block(w, sz[bp][1], bx + hsz, by); break;
        case H_PART_T_RIGHT_SPLIT:
            lf_block(w, sz[bp][0], bx, by); lf_block(w, sz[bp][1], bx + hsz, by); lf_block(w, sz[bp][1], bx + hsz, by + hsz); break;
        case H_PART_H4:
            for (int k = 0; k < 4; k++) if (by + (hsz >> 1) * k < w->bh) lf_block(w, sz[bp][0], bx, by + (hsz >> 1) * k);
            break;
        case H_PART_V4:
            for (int k = 0; k < 4; k++) if (bx + (hsz >> 1) * k < w->bw) lf_block(w, sz[bp][0], bx + (hsz >> 1) * k, by);
            break;
        default: w->o->err = -EINVAL;
        }
    } else if (have_h) {
        if (b->bl != bl) { walk(w, bl + 1, bx, by); walk(w, bl + 1, bx + hsz, by); }
        else lf_block(w, sz[H_PART_H][0], bx, by);
    } else {
        if (b->bl != bl) { walk(w, bl + 1, bx, by); walk(w, bl + 1, bx, by + hsz); }
        else lf_block(w, sz[H_PART_V][0], bx, by);
    }
}

/* every block of the frame -> rectangles (malloc'ed array in *out, the caller frees it) */
int dav1d_hip_lf_rects(const Dav1dHipFrameDesc *d, const uint8_t lflvl[8][4][8][2], Dav1dHipLfRect **out, size_t *n) {
    return dav1d_hip_lf_rects_sb(d, lflvl, NULL, out, n);
}

int dav1d_hip_lf_rects_sb(const Dav1dHipFrameDesc *d, const uint8_t lflvl[8][4][8][2], const uint8_t (*sb_lflvl)[8][4][8][2],
                          Dav1dHipLfRect **out, size_t *n) {
    if (!d || !lflvl || !out || !n || !d->b || d->layout < 0 || d->layout > 3) return -EINVAL;
    h_tables_init();
    RectOut o;
    memset(&o, 0, sizeof(o));
    LfWalk w;
    w.d = d; w.lflvl = lflvl; w.o = &o;
    w.sb_lflvl = sb_lflvl;
    w.sb_shift = d->sb128 ? 5 : 4;
    w.sbw = ((((d->w + 7) >> 3) << 1) + (1 << w.sb_shift) - 1) >> w.sb_shift;
    w.w4 = (d->w + 3) >> 2; w.h4 = (d->h + 3) >> 2;
    w.bw = ((d->w + 7) >> 3) << 1; w.bh = ((d->h + 7) >> 3) << 1;
    w.ss_hor = d->layout != DAV1D_HIP_LAYOUT_I444; w.ss_ver = d->layout == DAV1D_HIP_LAYOUT_I420;
    const int sb4 = d->sb128 ? 32 : 16, root = d->sb128 ? H_BL_128X128 : H_BL_64X64;
    for (int by = 0; by < w.bh; by += sb4)
        for (int bx = 0; bx < w.bw; bx += sb4)
            walk(&w, root, bx, by);
    if (o.err) { free(o.p); return o.err; }
    *out = o.p; *n = o.n;
    return 0;
}

void dav1d_hip_lf_rects_free(Dav1dHipLfRect *p) { free(p); }

/* The drivers of the in-loop filters as producers of task records; see include/dav1d_hip.h (Dav1dHipFilterDesc).
 * One record per DSP call the reference would make: loop_filter_sb (src/lf_apply_tmpl.c:176-311), the 8x8 units of
 * dav1d_cdef_brow that are filtered (src/cdef_apply_tmpl.c:149-290), the unit stripes of lr_stripe (src/lr_apply_tmpl.c:36-97).
 * The filters run out of place on the device (every stage reads the finished picture of the stage before), so the line
 * buffers and pixel back-ups of the reference drivers have no counterpart here: only WHICH calls happen, with WHAT
 * parameters, is restated.  Plain C99. */
#include "av1_host.h"
#include "lister_priv.h"
#include "../csrc/cdef_rows.h"
#include <errno.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define AV1_TABLE_QUAL static const
#include "../csrc/av1_tables.h"

#define VEC(T) struct { T *p; size_t n, cap; }
/* out of memory: the push lands in a scratch element and the entry point returns -ENOMEM (see lister.c) */
static __thread int fl_oom;
static __thread uint64_t fl_sink[16];
#define VPUSH(v, T) (((v).n == (v).cap && fl_grow((void **) &(v).p, &(v).cap, sizeof(T))) ? (T *) (void *) fl_sink : &(v).p[(v).n++])
static int fl_grow(void **p, size_t *cap, const size_t esz) {
    const size_t nc = *cap ? *cap * 2 : 256;
    void *q = realloc(*p, nc * esz);
    if (!q) { fl_oom = 1; return 1; }
    *p = q; *cap = nc;
    return 0;
}
static int imin(const int a, const int b) { return a < b ? a : b; }

typedef struct FOut {
    VEC(Dav1dHipLfTask) lf;
    VEC(Dav1dHipCdefTask) cdef;
    VEC(Dav1dHipLrTask) lr;
} FOut;

static void push_lf(FOut *o, const int plane, const int dir, const int comp, const uint32_t dst_off, const uint32_t lvl_off, const uint32_t m0,
                    const uint32_t m1, const uint32_t m2)
{
    if (!(m0 | m1 | m2)) return;
    Dav1dHipLfTask *k = VPUSH(o->lf, Dav1dHipLfTask);
    memset(k, 0, sizeof(*k));
    k->dst_off = dst_off; k->lvl_off = lvl_off;
    k->vmask[0] = m0; k->vmask[1] = m1; k->vmask[2] = m2;
    k->plane = (uint8_t) plane; k->dir = (uint8_t) dir; k->lvl_comp = (uint8_t) comp;
}

/* dav1d_loopfilter_sbrow_cols + _rows, src/lf_apply_tmpl.c:313-466 */
static void list_deblock(const ListerGeo *g, const Dav1dHipFilterDesc *fd, FOut *o, const int sby) {
    if (!fd->lf_level_y[0] && !fd->lf_level_y[1]) return;
    const int is_sb64 = !g->sb128;
    const int starty4 = (sby & is_sb64) << 4, sbsz = 32 >> is_sb64, sbl2 = 5 - is_sb64;
    const int halign = (g->bh + 31) & ~31;
    const int ss_ver = g->ss_ver, ss_hor = g->ss_hor, has_chroma = g->layout != DAV1D_HIP_LAYOUT_I400;
    const int w4 = (g->w + 3) >> 2, h4 = (g->h + 3) >> 2;
    const int endy4 = starty4 + imin(h4 - sby * sbsz, sbsz);
    const int uv_endy4 = (endy4 + ss_ver) >> ss_ver;
    const int sb128w = (g->bw + 31) >> 5;
    const Dav1dHipAv1Filter *const lflvl = fd->lf_mask + (size_t) (sby >> !g->sb128) * sb128w;
    const int do_uv = has_chroma && (fd->lf_level_u || fd->lf_level_v);
    /* which tile row starts at this superblock row (f->lf.start_of_tile_row, src/decode.c:2765-2770) */
    int start_of_tile_row = 0;
    for (int tr = 1; tr < g->n_tile_rows; tr++) if (g->row_start_sb[tr] == sby) start_of_tile_row = tr;

    for (int x = 0; x < sb128w; x++) {
        const Dav1dHipAv1Filter *const m = &lflvl[x];
        const int w = imin(32, w4 - x * 32), cw = (w + ss_hor) >> ss_hor;
        /* ---- edges between columns: one call per 4-pixel column, units running down the superblock row */
        for (int pl = 0; pl < (do_uv ? 3 : 1); pl++) {
            const int sv = pl ? ss_ver : 0, sh = pl ? ss_hor : 0;
            const int n = pl ? cw : w, y0 = starty4 >> sv, y1 = pl ? uv_endy4 : endy4;
            const int half = 16 >> sv;                       /* units per mask half */
            for (int xx = 0; xx < n; xx++) {
                if (!x && !xx) continue;                    /* the frame's left edge */
                uint32_t hm[3] = { 0, 0, 0 };
                const int nk = pl ? 2 : 3;
                for (int k = 0; k < nk; k++) {
                    const uint32_t lo = pl ? m->filter_uv[0][xx][k][0] : m->filter_y[0][xx][k][0];
                    const uint32_t hi = pl ? m->filter_uv[0][xx][k][1] : m->filter_y[0][xx][k][1];
                    hm[k] = y0 ? hi : (y1 > half ? lo | hi << half : lo);
                }
                /* a tile column starts here: no edge is wider than the transforms left of it allow (:330-369) */
                for (int tc = 1; tc < g->n_tile_cols; tc++) {
                    const int sbx = g->col_start_sb[tc];
                    if ((sbx << sbl2) >= g->bw) break;
                    const int bx4 = (sbx & is_sb64) ? 16 : 0;
                    if ((sbx >> is_sb64) != x || (bx4 >> sh) != xx) continue;
                    const uint8_t *lpf = pl ? fd->tx_lpf_right_edge[1] + (size_t) (halign >> ss_ver) * (tc - 1) + ((size_t) sby << (sbl2 - ss_ver))
                                            : fd->tx_lpf_right_edge[0] + (size_t) halign * (tc - 1) + ((size_t) sby << sbl2);
                    for (int y = y0; y < y1; y++) {
                        const uint32_t bit = 1u << (y - y0);
                        const int idx = pl ? !!(hm[1] & bit) : 2 * !!(hm[2] & bit) + !!(hm[1] & bit);
                        hm[0] &= ~bit; hm[1] &= ~bit; hm[2] &= ~bit;
                        hm[imin(idx, lpf[y - y0])] |= bit;
                    }
                }
                const int py = (sby * sbsz * 4) >> sv, px = ((x * 128) >> sh) + xx * 4;
                const uint32_t lvl_off = (uint32_t) ((size_t) ((sby * sbsz) >> sv) * g->b4_stride + (size_t) ((x * 32) >> sh) + xx);
                push_lf(o, pl, 0, pl ? 1 + pl : 0, (uint32_t) ((size_t) py * g->stride[pl] + px), lvl_off, hm[0], hm[1], hm[2]);
            }
        }
        /* ---- edges between rows: one call per 4-pixel row of this 128-pixel column, units running across */
        for (int pl = 0; pl < (do_uv ? 3 : 1); pl++) {
            const int sv = pl ? ss_ver : 0, sh = pl ? ss_hor : 0;
            const int y0 = starty4 >> sv, y1 = pl ? uv_endy4 : endy4, n = pl ? cw : w;
            const int half = 16 >> sh;
            for (int y = y0; y < y1; y++) {
                if (!sby && !y) continue;                   /* the frame's top edge */
                uint32_t vm[3] = { 0, 0, 0 };
                const int nk = pl ? 2 : 3;
                for (int k = 0; k < nk; k++) {
                    const uint32_t lo = pl ? m->filter_uv[1][y][k][0] : m->filter_y[1][y][k][0];
                    const uint32_t hi = pl ? m->filter_uv[1][y][k][1] : m->filter_y[1][y][k][1];
                    vm[k] = lo | hi << half;
                }
                /* a tile row starts here: clamp to the transforms above, kept in the above context of the tile row before (:371-398) */
                if (start_of_tile_row && y == y0) {
                    const uint8_t *a = (pl ? fd->a_tx_lpf_uv : fd->a_tx_lpf_y) + fd->a_stride * ((size_t) sb128w * (start_of_tile_row - 1) + x);
                    for (int i = 0; i < n; i++) {
                        const uint32_t bit = 1u << i;
                        const int idx = pl ? !!(vm[1] & bit) : 2 * !!(vm[2] & bit) + !!(vm[1] & bit);
                        vm[0] &= ~bit; vm[1] &= ~bit; vm[2] &= ~bit;
                        vm[imin(idx, a[i])] |= bit;
                    }
                }
                const int py = ((sby * sbsz * 4) >> sv) + (y - y0) * 4, px = (x * 128) >> sh;
                const uint32_t lvl_off = (uint32_t) ((size_t) (((sby * sbsz) >> sv) + (y - y0)) * g->b4_stride + (size_t) ((x * 32) >> sh));
                push_lf(o, pl, 1, pl ? 1 + pl : 1, (uint32_t) ((size_t) py * g->stride[pl] + px), lvl_off, vm[0], vm[1], vm[2]);
            }
        }
    }
}

/* dav1d_cdef_brow, src/cdef_apply_tmpl.c:97-308: which 8x8 units are filtered, with which strengths and edges */
static void list_cdef(const ListerGeo *g, const Dav1dHipFilterDesc *fd, FOut *o, const int sby) {
    if (!fd->cdef_enabled) return;
    const int sbsz = g->sb128 ? 32 : 16, bd8 = g->bpc - 8;
    const int sb128w = (g->bw + 31) >> 5;
    const int start = sby * sbsz, end = imin(start + sbsz, g->bh);
    /* the frame's table of unit rows (NULL: the frame wants unit records) */
    int rows_stride = 0;
    size_t n_units = 0;
    Dav1dHipCdefRow *const rows = dav1d_hip_frame_cdef_rows(g->frame, &rows_stride);
    for (int by = start; by < end; by += 2) {
        const Dav1dHipAv1Filter *const row = fd->lf_mask + (size_t) (by >> 5) * sb128w;
        const int by_idx = (by & 30) >> 1;
        /* The last unit row of a superblock row's own band (dav1d_filter_sbrow_cdef leaves the 8 rows under it to the next superblock
         * row: src/recon_tmpl.c:2027-2051) reads its two bottom rows from the lines backup_lpf() saved, and backup_lpf stores the
         * picture's last row twice when it is the first of the two (src/lf_apply_tmpl.c:77-97, n_lines).  For a unit row of dav1d's own
         * walk that never coincides: such a row needs by + 4 < bh and h == 4 by + 9 (or + 10 for 4:2:0 chroma), but bh = 2 ceil(h / 8)
         * = by + 4 then — the row belongs to the picture's last superblock row, which has no rows below it at all.  The unit records
         * keep the flags (DAV1D_HIP_CDEF_BOT_REP_*, tests/test_cdef.py sets them) for callers with other walks; this lister sets none. */
        const int rep = 0;
        for (int sbx = 0; sbx * 16 < g->bw; sbx++) {
            const Dav1dHipAv1Filter *const m = &row[sbx >> 1];
            const int cdef_idx = m->cdef_idx[((by & 16) >> 3) + (sbx & 1)];
            if (cdef_idx == -1 || (!fd->cdef_y_strength[cdef_idx] && !fd->cdef_uv_strength[cdef_idx])) continue;
            const uint32_t noskip = (uint32_t) m->noskip_mask[by_idx][1] << 16 | m->noskip_mask[by_idx][0];
            const int y_lvl = fd->cdef_y_strength[cdef_idx], uv_lvl = fd->cdef_uv_strength[cdef_idx];
            int y_sec = y_lvl & 3, uv_sec = uv_lvl & 3;
            y_sec += y_sec == 3; uv_sec += uv_sec == 3;
            if (rows) {
                /* one record for the eight units: the device cuts it into unit records (cdef.hip cdef_expand_kernel) */
                unsigned mask = 0;
                for (int u = 0, bx = sbx * 16; u < 8 && bx < g->bw; u++, bx += 2)
                    if (noskip & (3u << (bx & 30))) mask |= 1u << u;
                if (!mask) continue;
                Dav1dHipCdefRow *k = &rows[(size_t) (by >> 1) * rows_stride + sbx];
                k->y_pri = (uint8_t) ((y_lvl >> 2) << bd8); k->y_sec = (uint8_t) (y_sec << bd8);
                k->uv_pri = (uint8_t) ((uv_lvl >> 2) << bd8); k->uv_sec = (uint8_t) (uv_sec << bd8);
                k->flags = (uint8_t) rep;
                k->pad = 0;
                k->mask = (uint8_t) mask;
                n_units += (size_t) __builtin_popcount(mask);
                continue;
            }
            for (int bx = sbx * 16; bx < imin((sbx + 1) * 16, g->bw); bx += 2) {
                if (!(noskip & (3u << (bx & 30)))) continue;
                Dav1dHipCdefTask *k = VPUSH(o->cdef, Dav1dHipCdefTask);
                memset(k, 0, sizeof(*k));
                k->bx = (uint16_t) (bx >> 1); k->by = (uint16_t) (by >> 1);
                k->y_pri = (uint8_t) ((y_lvl >> 2) << bd8); k->y_sec = (uint8_t) (y_sec << bd8);
                k->uv_pri = (uint8_t) ((uv_lvl >> 2) << bd8); k->uv_sec = (uint8_t) (uv_sec << bd8);
                k->flags = (uint8_t) rep;
                k->edges = (uint8_t) ((bx > 0 ? DAV1D_HIP_CDEF_HAVE_LEFT : 0) | (bx + 2 < g->bw ? DAV1D_HIP_CDEF_HAVE_RIGHT : 0) |
                                      (by > 0 ? DAV1D_HIP_CDEF_HAVE_TOP : 0) | (by + 2 < g->bh ? DAV1D_HIP_CDEF_HAVE_BOTTOM : 0));
            }
        }
    }
    if (rows && n_units) dav1d_hip_frame_cdef_rows_add(g->frame, n_units);
}

/* lr_stripe, src/lr_apply_tmpl.c:36-97 */
static void list_lr_unit(const ListerGeo *g, FOut *o, const int x, int y, const int plane, const int unit_w, const int row_h,
                         const Dav1dHipRestorationUnit *lr, int edges, const int sbh)
{
    const int ss_ver = plane && g->ss_ver;
    const int sby = (y + (y ? 8 << ss_ver : 0)) >> (6 - ss_ver + g->sb128);
    int stripe_h = imin((64 - 8 * !y) >> ss_ver, row_h - y);
    Dav1dHipLrTask t;
    memset(&t, 0, sizeof(t));
    t.plane = (uint8_t) plane;
    if (lr->type == 2) {
        int16_t (*const f)[8] = t.filter;
        f[0][0] = f[0][6] = lr->filter_h[0]; f[0][1] = f[0][5] = lr->filter_h[1]; f[0][2] = f[0][4] = lr->filter_h[2];
        f[0][3] = (int16_t) (-(f[0][0] + f[0][1] + f[0][2]) * 2 + (g->bpc > 8 ? 128 : 0));      /* :59-66: the +128 is folded in above 8 bits */
        f[1][0] = f[1][6] = lr->filter_v[0]; f[1][1] = f[1][5] = lr->filter_v[1]; f[1][2] = f[1][4] = lr->filter_v[2];
        f[1][3] = (int16_t) (128 - (f[1][0] + f[1][1] + f[1][2]) * 2);
        t.type = (f[0][0] | f[1][0]) ? DAV1D_HIP_LR_WIENER7 : DAV1D_HIP_LR_WIENER5;
    } else {
        const int idx = lr->type - 3;
        const int s0 = av1_sgr_params[idx * 2], s1 = av1_sgr_params[idx * 2 + 1];
        t.filter[0][0] = (int16_t) s0; t.filter[0][1] = (int16_t) s1;
        t.filter[0][2] = lr->sgr_weights[0];
        t.filter[0][3] = (int16_t) (128 - (lr->sgr_weights[0] + lr->sgr_weights[1]));
        t.type = (uint8_t) (DAV1D_HIP_LR_SGR_5X5 + !!s0 + !!s1 * 2 - 1);
    }
    while (y + stripe_h <= row_h) {
        if (sby + 1 != sbh || y + stripe_h != row_h) edges |= DAV1D_HIP_LR_HAVE_BOTTOM; else edges &= ~DAV1D_HIP_LR_HAVE_BOTTOM;
        Dav1dHipLrTask *k = VPUSH(o->lr, Dav1dHipLrTask);
        *k = t;
        k->x = (uint16_t) x; k->y = (uint16_t) y; k->w = (uint16_t) unit_w; k->h = (uint16_t) stripe_h;
        k->edges = (uint8_t) edges;
        y += stripe_h;
        edges |= DAV1D_HIP_LR_HAVE_TOP;
        stripe_h = imin(64 >> ss_ver, row_h - y);
        if (stripe_h == 0) break;
    }
}

/* lr_sbrow + dav1d_lr_sbrow, src/lr_apply_tmpl.c:99-202 */
static void list_lr(const ListerGeo *g, const Dav1dHipFilterDesc *fd, FOut *o, const int sby) {
    const int sbh = (g->bh + g->sb_step - 1) / g->sb_step;
    const int not_last = sby + 1 < sbh, offset_y = 8 * !!sby;
    const int sr_w = fd->sr_w > 0 ? fd->sr_w : g->w;            /* restoration works on the upscaled frame */
    const int sr_sb128w = (sr_w + 127) >> 7;
    for (int plane = 0; plane < (g->layout == DAV1D_HIP_LAYOUT_I400 ? 1 : 3); plane++) {
        if (!fd->lr_type[plane]) continue;
        const int ss_ver = plane && g->ss_ver, ss_hor = plane && g->ss_hor;
        const int h = (g->h + ss_ver) >> ss_ver, w = (sr_w + ss_hor) >> ss_hor;
        const int next_row_y = (sby + 1) << ((6 - ss_ver) + g->sb128);
        const int row_h = imin(next_row_y - (8 >> ss_ver) * not_last, h);
        const int y = (sby << ((6 - ss_ver) + g->sb128)) - (offset_y >> ss_ver);
        const int unit_size_log2 = fd->lr_unit_size[!!plane], unit_size = 1 << unit_size_log2;
        const int half_unit_size = unit_size >> 1, max_unit_size = unit_size + half_unit_size;
        const int row_y = y + ((8 >> ss_ver) * !!y);
        const int shift_hor = 7 - ss_hor;
        int edges = (y > 0 ? DAV1D_HIP_LR_HAVE_TOP : 0) | DAV1D_HIP_LR_HAVE_RIGHT;
        int aligned_unit_pos = row_y & ~(unit_size - 1);
        if (aligned_unit_pos && aligned_unit_pos + half_unit_size > h) aligned_unit_pos -= unit_size;
        aligned_unit_pos <<= ss_ver;
        const int sb_idx = (aligned_unit_pos >> 7) * sr_sb128w;
        const int unit_idx = ((aligned_unit_pos >> 6) & 1) << 1;
        const Dav1dHipRestorationUnit *lr = &fd->lr_mask[sb_idx].lr[plane][unit_idx];
        int x = 0;
        for (; x + max_unit_size <= w; edges |= DAV1D_HIP_LR_HAVE_LEFT) {
            const int next_x = x + unit_size;
            const int next_u_idx = unit_idx + ((next_x >> (shift_hor - 1)) & 1);
            const Dav1dHipRestorationUnit *nxt = &fd->lr_mask[sb_idx + (next_x >> shift_hor)].lr[plane][next_u_idx];
            if (lr->type) list_lr_unit(g, o, x, y, plane, unit_size, row_h, lr, edges, sbh);
            x = next_x;
            lr = nxt;
        }
        if (lr->type) list_lr_unit(g, o, x, y, plane, w - x, row_h, lr, edges & ~DAV1D_HIP_LR_HAVE_RIGHT, sbh);
    }
}

/* parts: 1 = deblocking, 2 = CDEF, 4 = restoration — the three lists of a superblock row are independent of each other */
static int filter_sbrow_parts(Dav1dHipLister *l, const Dav1dHipFilterDesc *fd, const int sby, const int parts) {
    if (!l || !fd || sby < 0) return -EINVAL;
    ListerGeo g;
    dav1d_hip_lister_geo(l, &g);
    const int sbh = (g.bh + g.sb_step - 1) / g.sb_step;
    if (sby >= sbh) return -EINVAL;
    if ((fd->lf_level_y[0] || fd->lf_level_y[1] || fd->cdef_enabled) && !fd->lf_mask) return -EINVAL;
    if ((fd->lr_type[0] || fd->lr_type[1] || fd->lr_type[2]) && !fd->lr_mask) return -EINVAL;
    if (g.n_tile_cols > 1 && (fd->lf_level_y[0] || fd->lf_level_y[1]) && (!fd->tx_lpf_right_edge[0] || !fd->tx_lpf_right_edge[1])) return -EINVAL;
    if (g.n_tile_rows > 1 && (fd->lf_level_y[0] || fd->lf_level_y[1]) && (!fd->a_tx_lpf_y || !fd->a_tx_lpf_uv)) return -EINVAL;
    FOut o;
    memset(&o, 0, sizeof(o));
    fl_oom = 0;
    if (parts & 1) list_deblock(&g, fd, &o, sby);
    if (parts & 2) list_cdef(&g, fd, &o, sby);
    if (parts & 4) list_lr(&g, fd, &o, sby);
    if (fl_oom) { free(o.lf.p); free(o.cdef.p); free(o.lr.p); return -ENOMEM; }
    if (!o.lf.n && !o.cdef.n && !o.lr.n) { free(o.lf.p); free(o.cdef.p); free(o.lr.p); return 0; }
    /* the arrays go to the frame as they are (it frees them): no copy, no growing vector under the frame's lock */
    const int rc = dav1d_hip_frame_submit_filter_owned(g.frame, o.lf.p, o.lf.n, o.cdef.p, o.cdef.n, o.lr.p, o.lr.n);
    if (rc) { free(o.lf.p); free(o.cdef.p); free(o.lr.p); }
    return rc;
}

int dav1d_hip_lister_filter_sbrow(Dav1dHipLister *l, const Dav1dHipFilterDesc *fd, const int sby) { return filter_sbrow_parts(l, fd, sby, 7); }

int dav1d_hip_lister_filter_units(const Dav1dHipLister *l) {
    ListerGeo g;
    dav1d_hip_lister_geo(l, &g);
    return 3 * ((g.bh + g.sb_step - 1) / g.sb_step);
}
int dav1d_hip_lister_filter_unit(Dav1dHipLister *l, const Dav1dHipFilterDesc *fd, const int unit) { return filter_sbrow_parts(l, fd, unit / 3, 1 << (unit % 3)); }

/* Every superblock row of the frame's filter tasks on n_threads threads of the library (rows handed out under a mutex) — the
 * counterpart of dav1d_hip_lister_run for callers without a thread pool of their own. */
typedef struct FRunAll { Dav1dHipLister *l; const Dav1dHipFilterDesc *fd; pthread_mutex_t mtx; int next, n, err; } FRunAll;
static void *frun_worker(void *arg) {
    FRunAll *r = (FRunAll *) arg;
    for (;;) {
        pthread_mutex_lock(&r->mtx);
        const int k = r->err ? r->n : r->next++;
        pthread_mutex_unlock(&r->mtx);
        if (k >= r->n) break;
        /* a unit of work is one of the three lists of a row: an 8K frame has 34 superblock rows, too few for the threads at hand */
        const int rc = filter_sbrow_parts(r->l, r->fd, k / 3, 1 << (k % 3));
        if (rc) { pthread_mutex_lock(&r->mtx); if (!r->err) r->err = rc; pthread_mutex_unlock(&r->mtx); }
    }
    return NULL;
}
int dav1d_hip_lister_filter_run(Dav1dHipLister *l, const Dav1dHipFilterDesc *fd, int n_threads) {
    if (!l || !fd || n_threads < 1) return -EINVAL;
    ListerGeo g;
    dav1d_hip_lister_geo(l, &g);
    FRunAll r;
    r.l = l; r.fd = fd; r.next = 0; r.err = 0;
    r.n = 3 * ((g.bh + g.sb_step - 1) / g.sb_step);
    if (n_threads > r.n) n_threads = r.n;
    if (n_threads > 256) n_threads = 256;
    pthread_mutex_init(&r.mtx, NULL);
    dav1d_hip_host_pool_run(frun_worker, &r, n_threads);
    pthread_mutex_destroy(&r.mtx);
    return r.err;
}

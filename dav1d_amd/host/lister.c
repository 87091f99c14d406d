/* Pass-2 lister: the walk of dav1d's reconstruction pass over the per-frame hand-off arrays, emitting flat task records.
 *
 * What the reference does per tile-sbrow in pass 2 (dav1d_decode_tile_sbrow -> decode_sb -> decode_b -> f->bd_fn.recon_b_intra /
 * recon_b_inter; src/decode.c:2594-2635, 2117-2375, 706-806; src/recon_tmpl.c:1176-1985) is restated here as a producer of
 * Dav1dHipMcTask / CompTask / WarpTask / McScaledTask / IpredTask / ItxTask records, one per DSP call the reference would make,
 * submitted to a Dav1dHipFrame.  Differences in formulation, none in meaning:
 *   - neighbour state (the reference's BlockContext a / l and the refmvs rows, src/env.h, src/decode.c:727-800) is one map per
 *     frame: for every 4x4 cell on a block's bottom row / right column, the index of that block's Av1Block;
 *   - the intra edge tree (src/intra_edge.c) is evaluated on the fly from (top has right, left has bottom) of the node;
 *   - scratch buffers the reference reuses per thread (compinter, lap, interintra, seg_mask) become offsets of one arena
 *     per frame, because every block of the frame is in flight at once;
 *   - a block that reads pixels of the current frame (intra prediction, the intra half of inter-intra) gets a wavefront step:
 *     1 + the largest step among the cells its edges reach; everything else is step 0 (no dependency inside the frame).
 * Plain C99, no HIP: the frame's submit calls are the only way out. */
#include "av1_host.h"
#include "lister_priv.h"
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __device__
#include "../csrc/av1_scan_dev.h"      /* dav1d_scans, flattened (generated from the reference build's tables by tools/gen_tables.py) */
#undef __device__

/* internal entry points of the frame driver (csrc/frame.hip) */
int dav1d_hip_frame_picture(const Dav1dHipFrame *f, Dav1dHipPicture *out);
int dav1d_hip_frame_set_sb_deps(Dav1dHipFrame *f, uint32_t first, size_t n, const uint8_t *mask);
int dav1d_hip_frame_submit_tile_sbrow_own(Dav1dHipFrame *f, const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                                          const Dav1dHipItxTask *itx, size_t n_itx, const uint16_t *itx_dep);

#define VEC(T) struct { T *p; size_t n, cap; }
/* A vector that cannot grow hands out a scratch element and raises the calling thread's flag: the walk goes on writing into
 * the scratch (nothing is read back from a pushed element), and the entry point that started it returns -ENOMEM (SURVEY 8b:
 * library errors are negative errno, never an abort of the host process). */
static __thread int v_oom;
static __thread uint64_t v_sink[16];          /* >= the largest task record */
/* (the scratch is reached through a function of its own: as an operand of the conditional below, the thread-local address was
 * looked up — a call of __tls_get_addr in a shared library — on EVERY push, 4 % of the walk) */
static __attribute__((noinline, cold)) void *vsink(void) { return v_sink; }
#define VPUSH(v, T) (((v).n == (v).cap && vgrow((void **) &(v).p, &(v).cap, sizeof(T))) ? (T *) vsink() : &(v).p[(v).n++])
static int vgrow(void **p, size_t *cap, const size_t esz) {
    const size_t nc = *cap ? *cap * 2 : 256;
    void *q = realloc(*p, nc * esz);
    if (!q) { v_oom = 1; return 1; }
    *p = q; *cap = nc;
    return 0;
}

typedef struct TileCursor {
    size_t cbi;        /* next entry of f->frame_thread.cbi */
    size_t cf;         /* next BYTE of f->frame_thread.cf */
    size_t pal_idx;    /* next byte of f->frame_thread.pal_idx */
    int next_sby;
    uint64_t win[2][2];           /* [prep, mask][next, end]: the tile's window of the shared arenas (a tile is walked by one thread at a time) */
} TileCursor;

struct Dav1dHipLister {
    Dav1dHipFrameDesc d;
    Dav1dHipFrame *frame;
    int ss_hor, ss_ver, bw, bh, sb_step, hbd, csz /* bytes per coef */, psz /* bytes per pixel */;
    int stride[3];                /* picture strides in pixels */
    uint32_t *owner;              /* [4x4 cell] -> by * b4_stride + bx of the block whose bottom row / right column it is */
    size_t map_bytes;             /* of one step map (the owner map is twice that) */
    uint16_t *step[3];            /* [4x4 cell of the plane] -> wavefront step of the transform block that wrote it */
    int step_stride[3];
    TileCursor *tiles;
    int cf_align64;
    uint8_t *sb_dep;              /* [superblock, raster] -> neighbouring superblocks its intra blocks read intra pixels of (bit 0 left, 1 top-left,
                                     2 top, 3 top-right): the levels of the superblock wavefront (csrc/intra_sb.hip) */
    int sbw;
    /* The shared counters live on cache lines of their own: every walking thread reads the fields above for every block, and a
     * counter bumped on the same line sent that line around all of them (the walk of an 8K frame took 9 ms on 32 threads and
     * 22 ms on one).  The arena cursors are bumped once per WINDOW a tile-sbrow draws, not once per compound block. */
    char pad0[64];
    uint64_t arena_bytes;         /* prep arena cursor (atomic) */
    char pad1[56];
    uint64_t mask_bytes;          /* mask arena cursor (atomic), starts behind the constant masks */
    char pad2[56];
    uint32_t max_step;            /* atomic */
    char pad3[60];
};

/* one tile-sbrow's worth of output */
typedef struct Out {
    VEC(Dav1dHipMcTask) mc;
    VEC(Dav1dHipCompTask) comp;
    VEC(Dav1dHipWarpTask) warp;
    VEC(Dav1dHipMcScaledTask) scaled;
    VEC(Dav1dHipItxTask) itx;                         /* step 0 */
    VEC(uint16_t) itx_dep;                            /* per `itx` entry: the launches that predict under it (see Walk.bdep) */
    VEC(Dav1dHipIpredTask) ipred; VEC(uint16_t) ipred_step;
    VEC(Dav1dHipCompTask) blend;  VEC(uint16_t) blend_step;
    VEC(Dav1dHipItxTask) sitx;    VEC(uint16_t) sitx_step;
    VEC(Dav1dHipPackRec) prec; size_t npack;                  /* packing lister: the tile-sbrow's blocks to pack, npack values in all */
} Out;

typedef struct Walk {
    Dav1dHipLister *l;
    Out *o;
    TileCursor *cur;
    unsigned seen_step;                               /* largest step this walk has reported to the lister */
    int col_start, col_end, row_start, row_end;       /* tile, 4-pixel units */
    int err;
    /* what the latest dep_step() found in OTHER superblocks: the highest step among the intra-written cells it read there and which
     * neighbours they lie in (bit 0 left, 1 top-left, 2 top, 3 top-right); new_ipred() stamps it on the task it makes */
    unsigned xs_step, xs_mask;
    /* What the walk knows about an inter block while it lists it and the chunk preparation (csrc/chunk.hip) would otherwise have to
     * find again through maps of 4x4 cells: bdep[pl] = the launches that write the block's pixels of plane pl before its residuals
     * (bit b: the prediction launch of tile shape b, bit 15: the compound / blend launch); cand[pl] = 1 + index in Out.mc of the ONE
     * prediction (a PUT, or the first of the two PREPs of an averaged pair) that covers the block of plane pl, to be run in one wave
     * with a transform block of exactly its square size (recon.hip) — emit_tx() writes 1 + that block's index into the prediction's
     * `pad`.  An averaged pair names its first PREP in Dav1dHipCompTask.mask_off (1 + index; the second follows it). */
    int hint_on;
    unsigned bdep[3];
    size_t cand[3];
    uint32_t cand_off[3];
    int cand_dim[3];
} Walk;



/* bytes of the prep (which = 0) / mask (1) arena for this walk: from its tile's window, which is refilled from the shared cursor
 * 128 KB at a time (what a tile leaves unused at the end of the frame is less than that) */
static uint64_t arena_alloc(Walk *w, const int which, uint64_t *cursor, const uint64_t bytes) {
    const uint64_t need = (bytes + 31) & ~(uint64_t) 31;
    uint64_t (*win)[2] = w->cur->win;
    if (win[which][0] + need > win[which][1]) {
        const uint64_t grab = need > (1u << 17) ? need : (1u << 17);
        win[which][0] = __atomic_fetch_add(cursor, grab, __ATOMIC_RELAXED);
        win[which][1] = win[which][0] + grab;
    }
    const uint64_t at = win[which][0];
    win[which][0] += need;
    return at;
}

static int imin(const int a, const int b) { return a < b ? a : b; }
static int imax(const int a, const int b) { return a > b ? a : b; }
static int iclip(const int v, const int lo, const int hi) { return v < lo ? lo : v > hi ? hi : v; }
/* bin of the <= 64x16 tiles a prediction block is cut into (csrc/capi.hip push_tiles): 3 * class(w) + class(h) */
static int tile_bin(const int w_px, const int h_px) {
    const int tw = imin(w_px, 64), th = imin(h_px, 16);
    return (tw <= 4 ? 0 : tw <= 8 ? 1 : tw <= 16 ? 2 : tw <= 32 ? 3 : 4) * 3 + (th <= 4 ? 0 : th <= 8 ? 1 : 2);
}

static void note_step(Dav1dHipLister *l, const uint32_t s) {
    uint32_t cur = __atomic_load_n(&l->max_step, __ATOMIC_RELAXED);
    while (s > cur && !__atomic_compare_exchange_n(&l->max_step, &cur, s, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

/* ------------------------------------------------------------------------------------------------ wavefront steps */

/* 1 + the largest step among the cells an intra prediction of (x4, y4, tw, th) in plane `pl` can read: the row above from the
 * top-left corner to the end of the top-right extension, the column to the left down to the end of the bottom-left extension
 * (dav1d_prepare_intra_edges, src/ipred_prepare_tmpl.c:118-201).  Cells not decoded yet hold 0. */
static unsigned dep_step(const Walk *w, const int pl, const int x4, const int y4, const int tw, const int th) {
    const Dav1dHipLister *l = w->l;
    const int sh = pl ? l->ss_hor : 0, sv = pl ? l->ss_ver : 0;
    const int x_lo = w->col_start >> sh, x_hi = (w->col_end + sh) >> sh, y_lo = w->row_start >> sv, y_hi = (w->row_end + sv) >> sv;
    const uint16_t *m = l->step[pl];
    const int st = l->step_stride[pl];
    unsigned s = 0;
    if (y4 > y_lo)
        for (int x = imax(x4 - 1, x_lo); x < imin(x4 + 2 * tw, x_hi); x++) s = m[(y4 - 1) * st + x] > s ? m[(y4 - 1) * st + x] : s;
    if (x4 > x_lo)
        for (int y = y4; y < imin(y4 + 2 * th, y_hi); y++) s = m[y * st + x4 - 1] > s ? m[y * st + x4 - 1] : s;
    /* a block on the top row or the left column of its superblock: which neighbouring superblocks hold the intra-written cells (step
     * > 0) among those just scanned.  Cells of superblocks that are decoded later (right of the top-right corner inside the
     * superblock's own row, below the bottom-left corner) hold 0. */
    const int cx = l->sb_step >> sh, cy = l->sb_step >> sv;          /* cells of this plane per superblock */
    const int on_top = y4 % cy == 0, on_left = x4 % cx == 0;
    unsigned xs = 0, bits = 0;
    if (s && (on_top || on_left)) {
        if (y4 > y_lo)
            for (int x = imax(x4 - 1, x_lo); x < imin(x4 + 2 * tw, x_hi); x++) {
                const unsigned v = m[(y4 - 1) * st + x];
                if (v) {
                    const int dx = x / cx - x4 / cx;
                    if (on_top) { bits |= dx < 0 ? 2 : dx == 0 ? 4 : 8; if (v > xs) xs = v; }
                    else if (dx < 0) { bits |= 1; if (v > xs) xs = v; }
                }
            }
        if (x4 > x_lo && on_left)
            for (int y = y4; y < imin(y4 + 2 * th, y_hi); y++) {
                const unsigned v = m[y * st + x4 - 1];
                if (v && y / cy == y4 / cy) { bits |= 1; if (v > xs) xs = v; }
            }
        if (bits) l->sb_dep[(size_t) (y4 / cy) * l->sbw + x4 / cx] |= (uint8_t) bits;
    }
    ((Walk *) w)->xs_step = xs; ((Walk *) w)->xs_mask = bits;
    return s + 1;
}

static void set_step(const Walk *w, const int pl, const int x4, const int y4, const int tw, const int th, const unsigned s) {
    const Dav1dHipLister *l = w->l;
    const int sh = pl ? l->ss_hor : 0, sv = pl ? l->ss_ver : 0;
    const int x_hi = imin(x4 + tw, (l->bw + sh) >> sh), y_hi = imin(y4 + th, (l->bh + sv) >> sv);
    uint16_t *m = l->step[pl];
    const int st = l->step_stride[pl];
    /* steps travel as uint16_t (the cell map, the *_step vectors, dav1d_hip_frame_submit_intra_sorted): a wavefront deeper than
     * that — no legal frame size gets there with tiles of at most 4096 x 2304 luma samples — is refused, not wrapped */
    if (s > 65535) { ((Walk *) w)->err = -ERANGE; return; }
    for (int y = y4; y < y_hi; y++)
        for (int x = x4; x < x_hi; x++) m[y * st + x] = (uint16_t) s;
    if (s > ((Walk *) w)->seen_step) { ((Walk *) w)->seen_step = s; note_step(w->l, s); }
}

/* ------------------------------------------------------------------------------------------------ neighbours */

static const Dav1dHipAv1Block *block_at(const Walk *w, const int bx, const int by) {
    const Dav1dHipLister *l = w->l;
    return &l->d.b[l->owner[(size_t) by * l->d.b4_stride + bx]];
}

static int is_smooth(const int m) { return m == H_SMOOTH_PRED || m == H_SMOOTH_H_PRED || m == H_SMOOTH_V_PRED; }

/* sm_flag(t->a, bx4) | sm_flag(&t->l, by4), src/ipred_prepare.h:95-100: the luma mode of the intra block above / to the left */
static int sm_flag_y(const Walk *w, const int bx, const int by) {
    int f = 0;
    if (by > w->row_start) { const Dav1dHipAv1Block *a = block_at(w, bx, by - 1); f |= a->intra && is_smooth(a->u.i.y_mode); }
    if (bx > w->col_start) { const Dav1dHipAv1Block *b = block_at(w, bx - 1, by); f |= b->intra && is_smooth(b->u.i.y_mode); }
    return f;
}

/* sm_uv_flag(): the chroma mode of the block that carries the chroma above / to the left — the one covering the odd luma
 * cell of that chroma cell (blocks without chroma do not touch the context, src/decode.c:738-742, 795-798) */
static int sm_flag_uv(const Walk *w, const int bx, const int by) {
    const Dav1dHipLister *l = w->l;
    const int cx = bx >> l->ss_hor, cy = by >> l->ss_ver;
    int f = 0;
    if (cy > (w->row_start >> l->ss_ver)) {
        const Dav1dHipAv1Block *a = block_at(w, imin((cx << l->ss_hor) + l->ss_hor, l->bw - 1), ((cy - 1) << l->ss_ver) + l->ss_ver);
        f |= a->intra && is_smooth(a->u.i.uv_mode);
    }
    if (cx > (w->col_start >> l->ss_hor)) {
        const Dav1dHipAv1Block *b = block_at(w, ((cx - 1) << l->ss_hor) + l->ss_hor, imin((cy << l->ss_ver) + l->ss_ver, l->bh - 1));
        f |= b->intra && is_smooth(b->u.i.uv_mode);
    }
    return f;
}

/* ------------------------------------------------------------------------------------------------ residuals */

static uint32_t dst_off(const Dav1dHipLister *l, const int pl, const int x_px, const int y_px) {
    return (uint32_t) ((size_t) y_px * l->stride[pl] + x_px);
}

/* Packing (Dav1dHipFrameDesc.cf): the eob + 1 values of one block, in the order decode_coefs() produced them — scan position i
 * sits at dav1d_scans[tx][i] for the 2-D transform classes, at i for the horizontal 1-D classes, column-interleaved for the
 * vertical ones (src/recon_tmpl.c:458-520, 548-575) — move to the frame's value arena; where they were becomes zero, as
 * the reference's inverse transform leaves its slab (src/itx_tmpl.c:60,108).  The walk only NOTES the block (pack_note: returns the offset of
 * its first value among the tile-sbrow's); the values move when the row is handed in (dav1d_hip_pack_run: straight into the arena, on the
 * library's preparation threads where the frame has them — a third of the walk's time, and one copy of every value, used to go here). */
static uint32_t pack_note(Walk *w, const int tx, const int txtp, const int eob, const size_t cf_byte) {
    Out *o = w->o;
    Dav1dHipPackRec *r = VPUSH(o->prec, Dav1dHipPackRec);
    r->cf_off = (uint32_t) (cf_byte / (size_t) w->l->csz); r->n = (uint16_t) (eob + 1); r->tx = (uint8_t) tx; r->txtp = (uint8_t) txtp;
    const uint32_t at = (uint32_t) o->npack;
    o->npack += (size_t) eob + 1;
    return at;
}
void dav1d_hip_pack_run(const Dav1dHipPackRec *recs, const size_t n_recs, void *const cf, void *const dstv, const int csz) {
    h_tables_init();
    size_t at = 0;
    for (size_t k = 0; k < n_recs; k++) {
        const Dav1dHipPackRec *const r = &recs[k];
        const size_t n = r->n;
        const HostTx *t = &h_tx[r->tx];
        const int sw = imin(t->w, 8) * 4, sh = imin(t->h, 8) * 4, lsw = sw == 4 ? 2 : sw == 8 ? 3 : sw == 16 ? 4 : 5;      /* HostTx.w / h: 4-pixel units */
        /* TxfmType: H_DCT 10, V_DCT 11, H_ADST 12, V_ADST 13, H_FLIPADST 14, V_FLIPADST 15 (src/levels.h:80-100); the transposed storage
         * of the coefficient slab makes the V_ kinds the row-contiguous ones (src/recon_tmpl.c:458-496) */
        const int txtp = r->txtp;
        const int cls = (txtp == 11 || txtp == 13 || txtp == 15) ? 1 : (txtp == 10 || txtp == 12 || txtp == 14) ? 2 : 0;
        const uint16_t *scan = av1_scans + av1_scan_off[r->tx];
#define PACK_LOOP(T) do { \
        T *src = (T *) cf + r->cf_off, *dst = (T *) dstv + at; \
        if (cls == 0) for (size_t i = 0; i < n; i++) { const int rc = scan[i]; dst[i] = src[rc]; src[rc] = 0; } \
        else if (cls == 1) { memcpy(dst, src, n * sizeof(T)); memset(src, 0, n * sizeof(T)); } \
        else for (size_t i = 0; i < n; i++) { const int rc = ((int) i & (sw - 1)) * sh + ((int) i >> lsw); dst[i] = src[rc]; src[rc] = 0; } \
    } while (0)
        if (csz == 4) PACK_LOOP(int32_t); else PACK_LOOP(int16_t);
#undef PACK_LOOP
        at += n;
    }
}

/* one transform block: the next cbi entry, its slab, the task (src/recon_tmpl.c:796-816, 1292-1330, 1924-1970) */
static void emit_tx(Walk *w, const int pl, const int tx, const int x_px, const int y_px, const unsigned step) {
    Dav1dHipLister *l = w->l;
    const HostTx *t = &h_tx[tx];
    const int cbi = l->d.cbi[w->cur->cbi++];
    const size_t cf = w->cur->cf;
    w->cur->cf += (size_t) imin(t->w, 8) * imin(t->h, 8) * 16 * l->csz;
    const int eob = cbi >> 5, txtp = cbi & 0x1f;
    if (eob < 0) return;
    Dav1dHipItxTask *k = step ? VPUSH(w->o->sitx, Dav1dHipItxTask) : VPUSH(w->o->itx, Dav1dHipItxTask);
    memset(k, 0, sizeof(*k));
    k->dst_off = dst_off(l, pl, x_px, y_px);
    k->cf_off = (uint32_t) (cf / l->csz);
    if (l->d.cf) { k->cf_off = pack_note(w, tx, txtp, eob, cf); k->flags = DAV1D_HIP_ITX_PACKED; }
    k->eob = (int16_t) eob;
    k->tx = (uint8_t) tx;
    k->txtp = (uint8_t) txtp;
    k->plane = (uint8_t) pl;
    if (step) { *VPUSH(w->o->sitx_step, uint16_t) = (uint16_t) step; return; }
    *VPUSH(w->o->itx_dep, uint16_t) = (uint16_t) (w->hint_on ? w->bdep[pl] : 0xffff);
    if (w->hint_on && w->cand[pl]) {
        if (tx <= 4 && (4 << tx) == w->cand_dim[pl] && k->dst_off == w->cand_off[pl] && w->o->itx.n && k == &w->o->itx.p[w->o->itx.n - 1])
            w->o->mc.p[w->cand[pl] - 1].pad = (uint32_t) w->o->itx.n;          /* 1 + index of k */
        w->cand[pl] = 0;
    }
}

/* read_coef_tree(), src/recon_tmpl.c:731-822: the luma transform blocks of one inter block in tree order */
static void tx_tree(Walk *w, const Dav1dHipAv1Block *b, const int tx, const int depth, const int x_off, const int y_off,
                    const int bx, const int by, const unsigned step)
{
    const Dav1dHipLister *l = w->l;
    const HostTx *t = &h_tx[tx];
    const unsigned split = depth == 0 ? b->u.p.tx_split0 : b->u.p.tx_split1;
    if (depth < 2 && split && (split & (1u << (y_off * 4 + x_off)))) {
        const int sub = t->sub;
        const int sw = h_tx[sub].w, sh = h_tx[sub].h;
        tx_tree(w, b, sub, depth + 1, x_off * 2 + 0, y_off * 2 + 0, bx, by, step);
        if (t->w >= t->h && bx + sw < l->bw) tx_tree(w, b, sub, depth + 1, x_off * 2 + 1, y_off * 2 + 0, bx + sw, by, step);
        if (t->h >= t->w && by + sh < l->bh) {
            tx_tree(w, b, sub, depth + 1, x_off * 2 + 0, y_off * 2 + 1, bx, by + sh, step);
            if (t->w >= t->h && bx + sw < l->bw) tx_tree(w, b, sub, depth + 1, x_off * 2 + 1, y_off * 2 + 1, bx + sw, by + sh, step);
        }
    } else {
        emit_tx(w, 0, tx, bx * 4, by * 4, step);
    }
}

/* ------------------------------------------------------------------------------------------------ intra blocks */

static Dav1dHipIpredTask *new_ipred(Walk *w, const unsigned step) {
    Dav1dHipIpredTask *k = VPUSH(w->o->ipred, Dav1dHipIpredTask);
    memset(k, 0, sizeof(*k));
    *VPUSH(w->o->ipred_step, uint16_t) = (uint16_t) step;
    /* where the prediction reaches into other superblocks (Dav1dHipIpredTask.pal of the kinds that carry no palette: include/
     * dav1d_hip.h): the superblock route then waits for exactly these units of its neighbours instead of for whole superblocks.
     * (A palette prediction reads no neighbour and overwrites the fields with its colours.) */
    k->pal[6] = (uint16_t) (0x8000u | w->xs_mask);
    k->pal[7] = (uint16_t) w->xs_step;
    w->xs_step = w->xs_mask = 0;
    return k;
}

static void copy_pal(Dav1dHipIpredTask *k, const Dav1dHipLister *l, const int bx, const int by, const int pl) {
    /* f->frame_thread.pal[((by >> 1) + (bx & 1)) * (b4_stride >> 1) + ((bx >> 1) + (by & 1))][pl], src/recon_tmpl.c:1218-1221 */
    const size_t idx = (size_t) ((by >> 1) + (bx & 1)) * (size_t) (l->d.b4_stride >> 1) + (size_t) ((bx >> 1) + (by & 1));
    if (l->hbd) memcpy(k->pal, (const uint16_t *) l->d.pal + (idx * 3 + pl) * 8, 16);
    else for (int i = 0; i < 8; i++) k->pal[i] = ((const uint8_t *) l->d.pal)[(idx * 3 + pl) * 8 + i];
}

/* recon_b_intra(), src/recon_tmpl.c:1176-1555, the frame-threading (pass 2) branches */
static void list_intra(Walk *w, const int bs, const int edge_flags, const Dav1dHipAv1Block *b, const int bx, const int by) {
    Dav1dHipLister *l = w->l;
    const int ss_hor = l->ss_hor, ss_ver = l->ss_ver, layout = l->d.layout;
    const int bw4 = h_bs_dim[bs][0], bh4 = h_bs_dim[bs][1];
    const int w4 = imin(bw4, l->bw - bx), h4 = imin(bh4, l->bh - by);
    const int cw4 = (w4 + ss_hor) >> ss_hor, ch4 = (h4 + ss_ver) >> ss_ver;
    const int cbw4 = (bw4 + ss_hor) >> ss_hor, cbh4 = (bh4 + ss_ver) >> ss_ver;
    const int has_chroma = layout != DAV1D_HIP_LAYOUT_I400 && (bw4 > ss_hor || (bx & 1)) && (bh4 > ss_ver || (by & 1));
    const HostTx *t_dim = &h_tx[b->u.i.tx], *uv_t_dim = &h_tx[b->uvtx];
    const int ief = l->d.intra_edge_filter ? 16 : 0;
    unsigned luma_step = 1;

    for (int init_y = 0; init_y < h4; init_y += 16) {
        const int sub_h4 = imin(h4, 16 + init_y);
        const int sub_ch4 = imin(ch4, (init_y + 16) >> ss_ver);
        for (int init_x = 0; init_x < w4; init_x += 16) {
            if (b->u.i.pal_sz[0]) {
                Dav1dHipIpredTask *k = new_ipred(w, 1);
                k->kind = DAV1D_HIP_IPRED_PAL;
                k->dst_off = dst_off(l, 0, bx * 4, by * 4);
                k->aux_off = (uint32_t) w->cur->pal_idx;
                w->cur->pal_idx += (size_t) bw4 * bh4 * 8;
                k->x4 = (uint16_t) bx; k->y4 = (uint16_t) by;
                k->tw = (uint8_t) bw4; k->th = (uint8_t) bh4;
                copy_pal(k, l, bx, by, 0);
                set_step(w, 0, bx, by, bw4, bh4, 1);
            }
            const int sm = sm_flag_y(w, bx, by) ? 32 : 0;
            const int sb_has_tr = init_x + 16 < w4 ? 1 : init_y ? 0 : edge_flags & H_EDGE_I444_TR;
            const int sb_has_bl = init_x ? 0 : init_y + 16 < h4 ? 1 : edge_flags & H_EDGE_I444_BL;
            const int sub_w4 = imin(w4, init_x + 16);
            for (int y = init_y; y < sub_h4; y += t_dim->h)
                for (int x = init_x; x < sub_w4; x += t_dim->w) {
                    const int tbx = bx + x, tby = by + y;
                    unsigned s = 1;
                    if (!b->u.i.pal_sz[0]) {
                        const int tr = !(((y > init_y || !sb_has_tr) && (x + t_dim->w >= sub_w4)));
                        const int blf = !((x > init_x || (!sb_has_bl && y + t_dim->h >= sub_h4)));
                        s = dep_step(w, 0, tbx, tby, t_dim->w, t_dim->h);
                        Dav1dHipIpredTask *k = new_ipred(w, s);
                        k->kind = DAV1D_HIP_IPRED_PRED;
                        k->dst_off = dst_off(l, 0, tbx * 4, tby * 4);
                        k->x4 = (uint16_t) tbx; k->y4 = (uint16_t) tby;
                        k->w4 = (uint16_t) w->col_end; k->h4 = (uint16_t) w->row_end;
                        k->tw = t_dim->w; k->th = t_dim->h;
                        k->mode = b->u.i.y_mode;
                        k->angle = b->u.i.y_angle;
                        k->flags = (uint8_t) ((tbx > w->col_start ? 1 : 0) | (tby > w->row_start ? 2 : 0) | (tr ? 4 : 0) | (blf ? 8 : 0) | ief | sm);
                        k->max_w = (uint16_t) (4 * l->bw - 4 * tbx);
                        k->max_h = (uint16_t) (4 * l->bh - 4 * tby);
                        set_step(w, 0, tbx, tby, t_dim->w, t_dim->h, s);
                    }
                    if (s > luma_step) luma_step = s;
                    if (!b->skip) emit_tx(w, 0, b->u.i.tx, tbx * 4, tby * 4, s);
                }
            if (!has_chroma) continue;

            const int cx0 = bx >> ss_hor, cy0 = by >> ss_ver;
            const int xstart = w->col_start >> ss_hor, ystart = w->row_start >> ss_ver;
            unsigned cfl_step[2] = { 0, 0 };
            if (b->u.i.uv_mode == H_CFL_PRED) {
                /* cfl_ac over the whole chroma block, then cfl_pred per plane with a non-zero alpha (:1367-1393) */
                const int furthest_r = ((cw4 << ss_hor) + t_dim->w - 1) & ~(t_dim->w - 1);
                const int furthest_b = ((ch4 << ss_ver) + t_dim->h - 1) & ~(t_dim->h - 1);
                for (int pl = 0; pl < 2; pl++) {
                    if (!b->u.i.cfl_alpha[pl]) continue;
                    unsigned s = dep_step(w, 1 + pl, cx0, cy0, uv_t_dim->w, uv_t_dim->h);
                    if (s <= luma_step) s = luma_step + 1;          /* reads the reconstructed luma of this block */
                    /* ... and, for a block that carries the chroma of its 8x8 from an odd row / column, the luma of the block above /
                     * to the left as well (cfl_ac starts at bx & ~ss_hor, by & ~ss_ver, src/recon_tmpl.c:1367-1381).  Its own luma
                     * prediction normally waits for those cells anyway; a palette block's does not (no edges). */
                    {
                        const uint16_t *m = l->step[0];
                        const int st = l->step_stride[0];
                        for (int yy = by & ~ss_ver; yy < by + h4; yy++)
                            for (int xx = bx & ~ss_hor; xx < bx + w4; xx++)
                                if (m[yy * st + xx] >= s) s = m[yy * st + xx] + 1u;
                    }
                    cfl_step[pl] = s;
                    Dav1dHipIpredTask *k = new_ipred(w, s);
                    k->kind = DAV1D_HIP_IPRED_CFL;
                    k->plane = (uint8_t) (1 + pl);
                    k->dst_off = dst_off(l, 1 + pl, cx0 * 4, cy0 * 4);
                    k->aux_off = dst_off(l, 0, 4 * (bx & ~ss_hor), 4 * (by & ~ss_ver));
                    k->x4 = (uint16_t) cx0; k->y4 = (uint16_t) cy0;
                    k->w4 = (uint16_t) (w->col_end >> ss_hor); k->h4 = (uint16_t) (w->row_end >> ss_ver);
                    k->tw = (uint8_t) cbw4; k->th = (uint8_t) cbh4;
                    k->angle = b->u.i.cfl_alpha[pl];
                    k->flags = (uint8_t) ((cx0 > xstart ? 1 : 0) | (cy0 > ystart ? 2 : 0));
                    k->max_w = (uint16_t) (cbw4 - (furthest_r >> ss_hor));
                    k->max_h = (uint16_t) (cbh4 - (furthest_b >> ss_ver));
                    set_step(w, 1 + pl, cx0, cy0, cbw4, cbh4, s);
                }
            } else if (b->u.i.pal_sz[1]) {
                for (int pl = 0; pl < 2; pl++) {
                    Dav1dHipIpredTask *k = new_ipred(w, 1);
                    k->kind = DAV1D_HIP_IPRED_PAL;
                    k->plane = (uint8_t) (1 + pl);
                    k->dst_off = dst_off(l, 1 + pl, cx0 * 4, cy0 * 4);
                    k->aux_off = (uint32_t) w->cur->pal_idx;
                    k->x4 = (uint16_t) cx0; k->y4 = (uint16_t) cy0;
                    k->tw = (uint8_t) cbw4; k->th = (uint8_t) cbh4;
                    copy_pal(k, l, bx, by, 1 + pl);
                    set_step(w, 1 + pl, cx0, cy0, cbw4, cbh4, 1);
                }
                w->cur->pal_idx += (size_t) cbw4 * cbh4 * 8;
            }

            const int sm_uv = sm_flag_uv(w, bx, by) ? 32 : 0;
            const int tr_bit = H_EDGE_I420_TR >> (layout - 1), bl_bit = H_EDGE_I420_BL >> (layout - 1);
            const int uv_sb_has_tr = ((init_x + 16) >> ss_hor) < cw4 ? 1 : init_y ? 0 : edge_flags & tr_bit;
            const int uv_sb_has_bl = init_x ? 0 : ((init_y + 16) >> ss_ver) < ch4 ? 1 : edge_flags & bl_bit;
            const int sub_cw4 = imin(cw4, (init_x + 16) >> ss_hor);
            for (int pl = 0; pl < 2; pl++)
                for (int y = init_y >> ss_ver; y < sub_ch4; y += uv_t_dim->h)
                    for (int x = init_x >> ss_hor; x < sub_cw4; x += uv_t_dim->w) {
                        /* luma position the reference's t->bx / t->by hold here: block origin + the chroma offset scaled up */
                        const int lbx = bx + (x << ss_hor), lby = by + (y << ss_ver);
                        const int tcx = lbx >> ss_hor, tcy = lby >> ss_ver;
                        unsigned s = 1;
                        if (b->u.i.uv_mode == H_CFL_PRED && b->u.i.cfl_alpha[pl]) {
                            s = cfl_step[pl];
                        } else if (!b->u.i.pal_sz[1]) {
                            const int tr = !((y > (init_y >> ss_ver) || !uv_sb_has_tr) && (x + uv_t_dim->w >= sub_cw4));
                            const int blf = !(x > (init_x >> ss_hor) || (!uv_sb_has_bl && y + uv_t_dim->h >= sub_ch4));
                            s = dep_step(w, 1 + pl, tcx, tcy, uv_t_dim->w, uv_t_dim->h);
                            Dav1dHipIpredTask *k = new_ipred(w, s);
                            k->kind = DAV1D_HIP_IPRED_PRED;
                            k->plane = (uint8_t) (1 + pl);
                            k->dst_off = dst_off(l, 1 + pl, tcx * 4, tcy * 4);
                            k->x4 = (uint16_t) tcx; k->y4 = (uint16_t) tcy;
                            k->w4 = (uint16_t) (w->col_end >> ss_hor); k->h4 = (uint16_t) (w->row_end >> ss_ver);
                            k->tw = uv_t_dim->w; k->th = uv_t_dim->h;
                            k->mode = b->u.i.uv_mode == H_CFL_PRED ? H_DC_PRED : b->u.i.uv_mode;
                            k->angle = b->u.i.uv_angle;
                            k->flags = (uint8_t) ((tcx > xstart ? 1 : 0) | (tcy > ystart ? 2 : 0) | (tr ? 4 : 0) | (blf ? 8 : 0) | ief | sm_uv);
                            k->max_w = (uint16_t) ((4 * l->bw + ss_hor - 4 * (lbx & ~ss_hor)) >> ss_hor);
                            k->max_h = (uint16_t) ((4 * l->bh + ss_ver - 4 * (lby & ~ss_ver)) >> ss_ver);
                            set_step(w, 1 + pl, tcx, tcy, uv_t_dim->w, uv_t_dim->h, s);
                        }
                        if (!b->skip) emit_tx(w, 1 + pl, b->uvtx, tcx * 4, tcy * 4, s);
                    }
        }
    }
}

/* ------------------------------------------------------------------------------------------------ inter blocks */

typedef struct Mv { int y, x; } Mv;

/* mc(), src/recon_tmpl.c:938-1050: one prediction of bw4 x bh4 luma units (already doubled by the caller where the reference
 * passes `bw4 << (bw4 == ss_hor)`), block position (bx, by) in luma 4-pixel units.  kind PUT: off = pixel offset in the
 * picture plane; PREP / PUT_TMP: off = element offset in the arena. */
static void emit_mc(Walk *w, const int kind, const uint32_t off, const int bw4, const int bh4, const int bx, const int by, const int pl,
                    const Mv mv, const int ref, const int filter_2d)
{
    Dav1dHipLister *l = w->l;
    const int ss_ver = pl && l->ss_ver, ss_hor = pl && l->ss_hor;
    const int h_mul = 4 >> ss_hor, v_mul = 4 >> ss_ver;
    if (!l->d.svc[ref][0][0] && !l->d.svc[ref][1][0]) {
        Dav1dHipMcTask *k = VPUSH(w->o->mc, Dav1dHipMcTask);
        memset(k, 0, sizeof(*k));
        k->dst_off = off;
        k->src_x = bx * h_mul + (mv.x >> (3 + ss_hor));
        k->src_y = by * v_mul + (mv.y >> (3 + ss_ver));
        k->w = (uint8_t) (bw4 * h_mul); k->h = (uint8_t) (bh4 * v_mul);
        k->mx = (uint8_t) ((mv.x & (15 >> !ss_hor)) << !ss_hor);
        k->my = (uint8_t) ((mv.y & (15 >> !ss_ver)) << !ss_ver);
        k->filter_2d = (uint8_t) filter_2d;
        k->kind = (uint8_t) kind;
        k->plane = (uint8_t) pl;
        k->ref = (uint8_t) ref;
        if (w->hint_on && kind == DAV1D_HIP_MC_PUT) {
            w->bdep[pl] |= 1u << tile_bin(k->w, k->h);
            w->cand[pl] = k->w == k->h && k->w >= 4 ? w->o->mc.n : 0;
            w->cand_off[pl] = off; w->cand_dim[pl] = k->w;
        }
    } else {
        /* scaled reference, :990-1047 */
        const int orig_pos_y = (by * v_mul << 4) + mv.y * (1 << !ss_ver);
        const int orig_pos_x = (bx * h_mul << 4) + mv.x * (1 << !ss_hor);
        int pos[2];
        const int orig[2] = { orig_pos_x, orig_pos_y };
        for (int i = 0; i < 2; i++) {
            const int scale = l->d.svc[ref][i][0];
            const int64_t tmp = (int64_t) orig[i] * scale + (int64_t) (scale - 0x4000) * 8;
            const int a = (int) (((tmp < 0 ? -tmp : tmp) + 128) >> 8);
            pos[i] = (tmp < 0 ? -a : a) + 32;
        }
        Dav1dHipMcScaledTask *k = VPUSH(w->o->scaled, Dav1dHipMcScaledTask);
        memset(k, 0, sizeof(*k));
        k->dst_off = off;
        k->src_x = pos[0] >> 10; k->src_y = pos[1] >> 10;
        k->mx = (int16_t) (pos[0] & 0x3ff); k->my = (int16_t) (pos[1] & 0x3ff);
        k->dx = (int16_t) l->d.svc[ref][0][1]; k->dy = (int16_t) l->d.svc[ref][1][1];
        k->w = (uint8_t) (bw4 * h_mul); k->h = (uint8_t) (bh4 * v_mul);
        k->filter_2d = (uint8_t) filter_2d;
        k->kind = (uint8_t) kind;
        k->plane = (uint8_t) pl;
        k->ref = (uint8_t) ref;
    }
}

/* warp_affine(), src/recon_tmpl.c:1115-1174: one record per 8x8 */
static void emit_warp(Walk *w, const int kind, const uint32_t off, const int stride, const int bs, const int bx, const int by, const int pl,
                      const int ref, const Dav1dHipWarpParams *wmp)
{
    Dav1dHipLister *l = w->l;
    const int ss_ver = pl && l->ss_ver, ss_hor = pl && l->ss_hor;
    const int h_mul = 4 >> ss_hor, v_mul = 4 >> ss_ver;
    const int32_t *mat = wmp->matrix;
    for (int y = 0; y < h_bs_dim[bs][1] * v_mul; y += 8) {
        const int src_y = by * 4 + ((y + 4) << ss_ver);
        const int64_t mat3_y = (int64_t) mat[3] * src_y + mat[0];
        const int64_t mat5_y = (int64_t) mat[5] * src_y + mat[1];
        for (int x = 0; x < h_bs_dim[bs][0] * h_mul; x += 8) {
            const int src_x = bx * 4 + ((x + 4) << ss_hor);
            const int64_t mvx = ((int64_t) mat[2] * src_x + mat3_y) >> ss_hor;
            const int64_t mvy = ((int64_t) mat[4] * src_x + mat5_y) >> ss_ver;
            Dav1dHipWarpTask *k = VPUSH(w->o->warp, Dav1dHipWarpTask);
            memset(k, 0, sizeof(*k));
            k->dst_off = off + (uint32_t) (y * stride + x);
            k->src_x = (int) (mvx >> 16) - 4;
            k->src_y = (int) (mvy >> 16) - 4;
            k->mx = (((int) mvx & 0xffff) - wmp->u.p.alpha * 4 - wmp->u.p.beta * 7) & ~0x3f;
            k->my = (((int) mvy & 0xffff) - wmp->u.p.gamma * 4 - wmp->u.p.delta * 4) & ~0x3f;
            memcpy(k->abcd, wmp->u.abcd, sizeof(k->abcd));
            k->tmp_stride = (uint16_t) stride;
            k->kind = (uint8_t) kind;
            k->plane = (uint8_t) pl;
            k->ref = (uint8_t) ref;
        }
    }
}

static Dav1dHipCompTask *new_comp(Walk *w, const int kind, const int pl, const uint32_t doff, const int cw, const int chh) {
    Dav1dHipCompTask *k = VPUSH(w->o->comp, Dav1dHipCompTask);
    memset(k, 0, sizeof(*k));
    k->dst_off = doff;
    k->w = (uint8_t) cw; k->h = (uint8_t) chh;
    k->kind = (uint8_t) kind;
    k->plane = (uint8_t) pl;
    return k;
}

static Mv mv_of(const int16_t m[2]) { Mv r = { m[0], m[1] }; return r; }

/* obmc(), src/recon_tmpl.c:1052-1112 */
static void list_obmc(Walk *w, const uint32_t doff, const int bs, const int pl, const int bx, const int by, const int w4, const int h4) {
    Dav1dHipLister *l = w->l;
    const int ss_ver = pl && l->ss_ver, ss_hor = pl && l->ss_hor;
    const int h_mul = 4 >> ss_hor, v_mul = 4 >> ss_ver;
    const uint8_t *b_dim = h_bs_dim[bs];
    if (by > w->row_start && (!pl || b_dim[0] * h_mul + b_dim[1] * v_mul >= 16)) {
        for (int i = 0, x = 0; x < w4 && i < imin(b_dim[2], 4); ) {
            const Dav1dHipAv1Block *a = block_at(w, bx + x + 1, by - 1);      /* odd cells only */
            const int step4 = iclip(h_bs_dim[a->bs][0], 2, 16);
            if (!a->intra) {
                const int ow4 = imin(step4, b_dim[0]), oh4 = imin(b_dim[1], 16) >> 1;
                const int lw = ow4 * h_mul, lh = ((oh4 * 3 + 3) >> 2) * v_mul;
                const uint64_t ab = arena_alloc(w, 0, &l->arena_bytes, (uint64_t) lw * lh * l->psz);
                emit_mc(w, DAV1D_HIP_MC_PUT_TMP, (uint32_t) (ab / l->psz), ow4, (oh4 * 3 + 3) >> 2, bx + x, by, pl, mv_of(a->u.p.u.m.mv[0]),
                        a->u.p.ref[0], a->u.p.filter2d);
                Dav1dHipCompTask *k = new_comp(w, DAV1D_HIP_COMP_BLEND_H, pl, doff + (uint32_t) (x * h_mul), h_mul * ow4, v_mul * oh4);
                k->tmp1_off = (uint32_t) (ab / l->psz);
                w->cand[pl] = 0; w->bdep[pl] |= 1u << 15;       /* prediction, blends, THEN the residual */
                i++;
            }
            x += step4;
        }
    }
    if (bx > w->col_start)
        for (int i = 0, y = 0; y < h4 && i < imin(b_dim[3], 4); ) {
            const Dav1dHipAv1Block *lf = block_at(w, bx - 1, by + y + 1);
            const int step4 = iclip(h_bs_dim[lf->bs][1], 2, 16);
            if (!lf->intra) {
                const int ow4 = imin(b_dim[0], 16) >> 1, oh4 = imin(step4, b_dim[1]);
                const int lw = ow4 * h_mul, lh = oh4 * v_mul;
                const uint64_t ab = arena_alloc(w, 0, &l->arena_bytes, (uint64_t) lw * lh * l->psz);
                emit_mc(w, DAV1D_HIP_MC_PUT_TMP, (uint32_t) (ab / l->psz), ow4, oh4, bx, by + y, pl, mv_of(lf->u.p.u.m.mv[0]),
                        lf->u.p.ref[0], lf->u.p.filter2d);
                Dav1dHipCompTask *k = new_comp(w, DAV1D_HIP_COMP_BLEND_V, pl, doff + (uint32_t) (y * v_mul * l->stride[pl]), h_mul * ow4, v_mul * oh4);
                k->tmp1_off = (uint32_t) (ab / l->psz);
                w->cand[pl] = 0; w->bdep[pl] |= 1u << 15;
                i++;
            }
            y += step4;
        }
}

/* the intra half of an inter-intra block + its blend (src/recon_tmpl.c:1606-1630, 1751-1784); returns the wavefront step */
static unsigned list_interintra(Walk *w, const int bs, const Dav1dHipAv1Block *b, const int pl, const int bx, const int by) {
    Dav1dHipLister *l = w->l;
    const int ss_ver = pl && l->ss_ver, ss_hor = pl && l->ss_hor;
    const int bw4 = h_bs_dim[bs][0], bh4 = h_bs_dim[bs][1];
    const int pw4 = pl ? (bw4 + ss_hor) >> ss_hor : bw4, ph4 = pl ? (bh4 + ss_ver) >> ss_ver : bh4;
    const int x4 = bx >> ss_hor, y4 = by >> ss_ver;
    const unsigned s = dep_step(w, pl, x4, y4, pw4, ph4);
    const uint64_t ab = arena_alloc(w, 0, &l->arena_bytes, (uint64_t) pw4 * ph4 * 16 * l->psz);
    Dav1dHipIpredTask *k = new_ipred(w, s);
    k->kind = DAV1D_HIP_IPRED_PRED_TMP;
    k->plane = (uint8_t) pl;
    k->dst_off = dst_off(l, pl, x4 * 4, y4 * 4);
    k->aux_off = (uint32_t) (ab / l->psz);
    k->x4 = (uint16_t) x4; k->y4 = (uint16_t) y4;
    k->w4 = (uint16_t) (w->col_end >> ss_hor); k->h4 = (uint16_t) (w->row_end >> ss_ver);
    k->tw = (uint8_t) pw4; k->th = (uint8_t) ph4;
    k->mode = b->u.p.u.m.interintra_mode == 3 /* II_SMOOTH_PRED */ ? H_SMOOTH_PRED : b->u.p.u.m.interintra_mode;
    k->flags = (uint8_t) ((x4 > (w->col_start >> ss_hor) ? 1 : 0) | (y4 > (w->row_start >> ss_ver) ? 2 : 0));
    const HostMasks *hm = h_masks();
    const int c = !pl ? 0 : l->d.layout == DAV1D_HIP_LAYOUT_I400 ? 0 : DAV1D_HIP_LAYOUT_I444 - l->d.layout;
    Dav1dHipCompTask *q = VPUSH(w->o->blend, Dav1dHipCompTask);
    memset(q, 0, sizeof(*q));
    *VPUSH(w->o->blend_step, uint16_t) = (uint16_t) s;
    q->dst_off = k->dst_off;
    q->tmp1_off = (uint32_t) (ab / l->psz);
    q->w = (uint8_t) (pw4 * 4); q->h = (uint8_t) (ph4 * 4);
    q->kind = DAV1D_HIP_COMP_BLEND;
    q->plane = (uint8_t) pl;
    q->mask_off = b->u.p.interintra_type == H_INTER_INTRA_BLEND ? hm->ii[c][bs - H_BS_32x32][b->u.p.u.m.interintra_mode]
                                                                : hm->wedge[c][bs - H_BS_32x32][0][b->u.p.u.m.wedge_idx];
    set_step(w, pl, x4, y4, pw4, ph4, s);
    return s;
}

static void list_inter_residuals(Walk *w, int bs, const Dav1dHipAv1Block *b, int bx, int by, const unsigned step[3]);

/* recon_b_inter(), src/recon_tmpl.c:1557-1985, for inter frames */
static void list_inter(Walk *w, const int bs, const Dav1dHipAv1Block *b, const int bx, const int by) {
    Dav1dHipLister *l = w->l;
    const int ss_hor = l->ss_hor, ss_ver = l->ss_ver, layout = l->d.layout;
    const uint8_t *b_dim = h_bs_dim[bs];
    const int bw4 = b_dim[0], bh4 = b_dim[1];
    const int w4 = imin(bw4, l->bw - bx), h4 = imin(bh4, l->bh - by);
    const int has_chroma = layout != DAV1D_HIP_LAYOUT_I400 && (bw4 > ss_hor || (bx & 1)) && (bh4 > ss_ver || (by & 1));
    const int chr_layout_idx = layout == DAV1D_HIP_LAYOUT_I400 ? 0 : DAV1D_HIP_LAYOUT_I444 - layout;
    const int cbh4 = (bh4 + ss_ver) >> ss_ver, cbw4 = (bw4 + ss_hor) >> ss_hor;
    const uint32_t ydst = dst_off(l, 0, bx * 4, by * 4);
    const uint32_t uvdst = has_chroma ? dst_off(l, 1, 4 * (bx >> ss_hor), 4 * (by >> ss_ver)) : 0;
    const int filter_2d = b->u.p.filter2d;
    unsigned step[3] = { 0, 0, 0 };                     /* wavefront step of the residuals per plane */
    w->hint_on = 1;
    for (int pl = 0; pl < 3; pl++) { w->bdep[pl] = 0; w->cand[pl] = 0; }

    if (b->u.p.comp_type == H_COMP_INTER_NONE) {
        const int ref = b->u.p.ref[0];
        Dav1dHipWarpParams warpmv;
        const Dav1dHipWarpParams *wmp = NULL;
        if (b->u.p.motion_mode == H_MM_WARP) {
            /* t->warpmv as decode_b() rebuilds it in pass 2, src/decode.c:743-775 */
            h_block_warp(&warpmv, b->u.p.u.w.matrix, b->u.p.u.w.mv2d, bw4, bh4, bx, by);
            if (warpmv.type > H_WM_TRANSLATION) wmp = &warpmv;
        }
        if (!wmp && b->u.p.inter_mode == H_GLOBALMV && l->d.gmv_warp_allowed[ref]) wmp = &l->d.gmv[ref];
        const Mv mv0 = mv_of(b->u.p.u.m.mv[0]);
        if (imin(bw4, bh4) > 1 && wmp) {
            emit_warp(w, DAV1D_HIP_MC_PUT, ydst, l->stride[0], bs, bx, by, 0, ref, wmp);
        } else {
            emit_mc(w, DAV1D_HIP_MC_PUT, ydst, bw4, bh4, bx, by, 0, mv0, ref, filter_2d);
            if (b->u.p.motion_mode == H_MM_OBMC) list_obmc(w, ydst, bs, 0, bx, by, w4, h4);
        }
        if (b->u.p.interintra_type) step[0] = list_interintra(w, bs, b, 0, bx, by);

        if (has_chroma) {
            /* sub8x8: a 4-pixel-wide / -high luma block carries the chroma of its 8x8 and predicts each quarter with the
             * motion of the luma block above it (:1632-1712) — when all of them are inter */
            int is_sub8x8 = bw4 == ss_hor || bh4 == ss_ver;
            const Dav1dHipAv1Block *nl = NULL, *nt = NULL, *ntl = NULL;
            if (is_sub8x8) {
                if (bw4 == 1) { nl = &l->d.b[(size_t) by * l->d.b4_stride + bx - 1]; is_sub8x8 &= !nl->intra; }
                if (bh4 == ss_ver) { nt = &l->d.b[(size_t) (by - 1) * l->d.b4_stride + bx]; is_sub8x8 &= !nt->intra; }
                if (bw4 == 1 && bh4 == ss_ver) { ntl = &l->d.b[(size_t) (by - 1) * l->d.b4_stride + bx - 1]; is_sub8x8 &= !ntl->intra; }
            }
            if (is_sub8x8) {
                int h_off = 0, v_off = 0;
                if (bw4 == 1 && bh4 == ss_ver) {
                    for (int pl = 1; pl < 3; pl++)
                        emit_mc(w, DAV1D_HIP_MC_PUT, uvdst, bw4, bh4, bx - 1, by - 1, pl, mv_of(ntl->u.p.u.m.mv[0]), ntl->u.p.ref[0], ntl->u.p.filter2d);
                    v_off = 2 * l->stride[1];
                    h_off = 2;
                }
                if (bw4 == 1) {
                    for (int pl = 1; pl < 3; pl++)
                        emit_mc(w, DAV1D_HIP_MC_PUT, uvdst + (uint32_t) v_off, bw4, bh4, bx - 1, by, pl, mv_of(nl->u.p.u.m.mv[0]), nl->u.p.ref[0], nl->u.p.filter2d);
                    h_off = 2;
                }
                if (bh4 == ss_ver) {
                    for (int pl = 1; pl < 3; pl++)
                        emit_mc(w, DAV1D_HIP_MC_PUT, uvdst + (uint32_t) h_off, bw4, bh4, bx, by - 1, pl, mv_of(nt->u.p.u.m.mv[0]), nt->u.p.ref[0], nt->u.p.filter2d);
                    v_off = 2 * l->stride[1];
                }
                for (int pl = 1; pl < 3; pl++)
                    emit_mc(w, DAV1D_HIP_MC_PUT, uvdst + (uint32_t) (h_off + v_off), bw4, bh4, bx, by, pl, mv0, ref, filter_2d);
            } else {
                if (imin(cbw4, cbh4) > 1 && wmp) {
                    for (int pl = 1; pl < 3; pl++) emit_warp(w, DAV1D_HIP_MC_PUT, uvdst, l->stride[1], bs, bx, by, pl, ref, wmp);
                } else {
                    for (int pl = 1; pl < 3; pl++) {
                        emit_mc(w, DAV1D_HIP_MC_PUT, uvdst, bw4 << (bw4 == ss_hor), bh4 << (bh4 == ss_ver), bx & ~ss_hor, by & ~ss_ver, pl, mv0, ref, filter_2d);
                        if (b->u.p.motion_mode == H_MM_OBMC) list_obmc(w, uvdst, bs, pl, bx, by, w4, h4);
                    }
                }
                if (b->u.p.interintra_type)
                    for (int pl = 1; pl < 3; pl++) step[pl] = list_interintra(w, bs, b, pl, bx, by);
            }
        }
    } else {
        /* compound: two int16 predictions per plane, then avg / w_avg / mask / w_mask (:1786-1893) */
        const HostMasks *hm = h_masks();
        uint32_t mask_off = 0;
        const int sign = b->u.p.u.m.mask_sign;
        for (int pl = 0; pl < (has_chroma ? 3 : 1); pl++) {
            const int sh = pl && ss_hor, sv = pl && ss_ver;
            const int pw = bw4 * 4 >> sh, ph = bh4 * 4 >> sv;
            uint32_t tmp[2];
            const size_t n0 = w->o->mc.n;
            for (int i = 0; i < 2; i++) {
                const int ref = b->u.p.ref[i];
                const uint64_t ab = arena_alloc(w, 0, &l->arena_bytes, (uint64_t) pw * ph * 2);
                tmp[i] = (uint32_t) (ab / 2);
                if (b->u.p.inter_mode == H_GLOBALMV_GLOBALMV && l->d.gmv_warp_allowed[ref] && (!pl || imin(cbw4, cbh4) > 1))
                    emit_warp(w, DAV1D_HIP_MC_PREP, tmp[i], pw, bs, bx, by, pl, ref, &l->d.gmv[ref]);
                else
                    emit_mc(w, DAV1D_HIP_MC_PREP, tmp[i], bw4, bh4, bx, by, pl, mv_of(b->u.p.u.m.mv[i]), ref, filter_2d);
            }
            const uint32_t doff = pl ? uvdst : ydst;
            /* both halves plain translations of unscaled references: the pair can be predicted twice and averaged in registers */
            const int plain = w->o->mc.n == n0 + 2;
            Dav1dHipCompTask *k;
            switch (b->u.p.comp_type) {
            case H_COMP_INTER_AVG:
            case H_COMP_INTER_WEIGHTED_AVG:
                if (b->u.p.comp_type == H_COMP_INTER_AVG) {
                    k = new_comp(w, DAV1D_HIP_COMP_AVG, pl, doff, pw, ph);
                } else {
                    k = new_comp(w, DAV1D_HIP_COMP_WAVG, pl, doff, pw, ph);
                    k->arg = (int8_t) l->d.jnt_weights[b->u.p.ref[0]][b->u.p.ref[1]];
                }
                k->tmp1_off = tmp[0]; k->tmp2_off = tmp[1];
                if (plain) {
                    k->mask_off = (uint32_t) n0 + 1;
                    w->bdep[pl] |= 1u << tile_bin(pw, ph);
                    w->cand[pl] = pw == ph && pw >= 4 ? n0 + 1 : 0;
                    w->cand_off[pl] = doff; w->cand_dim[pl] = pw;
                } else {
                    w->bdep[pl] |= 1u << 15;
                }
                break;
            case H_COMP_INTER_SEG:
                if (!pl) {
                    const int mw = bw4 * 4 >> (chr_layout_idx != 0), mh = bh4 * 4 >> (chr_layout_idx == 2);
                    mask_off = (uint32_t) arena_alloc(w, 1, &l->mask_bytes, (uint64_t) mw * mh);
                    k = new_comp(w, DAV1D_HIP_COMP_WMASK, 0, doff, pw, ph);
                    k->arg = (int8_t) sign;
                    k->ss = (uint8_t) chr_layout_idx;
                } else {
                    k = new_comp(w, DAV1D_HIP_COMP_MASK, pl, doff, pw, ph);
                }
                k->tmp1_off = tmp[sign]; k->tmp2_off = tmp[!sign];
                k->mask_off = mask_off;
                w->bdep[pl] |= 1u << 15;
                break;
            default: /* H_COMP_INTER_WEDGE */
                k = new_comp(w, DAV1D_HIP_COMP_MASK, pl, doff, pw, ph);
                k->tmp1_off = tmp[sign]; k->tmp2_off = tmp[!sign];
                k->mask_off = pl ? hm->wedge[chr_layout_idx][bs - H_BS_32x32][sign][b->u.p.u.m.wedge_idx]
                                 : hm->wedge[0][bs - H_BS_32x32][0][b->u.p.u.m.wedge_idx];
                w->bdep[pl] |= 1u << 15;
                break;
            }
        }
    }

    list_inter_residuals(w, bs, b, bx, by, step);
    w->hint_on = 0;
}

/* residuals of an inter (or intra-block-copy) block: per 64x64 of the block, the luma transform tree, then both chroma planes
 * (src/recon_tmpl.c:1916-1981); step[pl] = wavefront step of the plane's residuals */
static void list_inter_residuals(Walk *w, const int bs, const Dav1dHipAv1Block *b, const int bx, const int by, const unsigned step[3]) {
    Dav1dHipLister *l = w->l;
    const int ss_hor = l->ss_hor, ss_ver = l->ss_ver, layout = l->d.layout;
    const uint8_t *b_dim = h_bs_dim[bs];
    const int bw4 = b_dim[0], bh4 = b_dim[1];
    const int w4 = imin(bw4, l->bw - bx), h4 = imin(bh4, l->bh - by);
    const int has_chroma = layout != DAV1D_HIP_LAYOUT_I400 && (bw4 > ss_hor || (bx & 1)) && (bh4 > ss_ver || (by & 1));
    if (b->skip) return;
    const int cw4 = (w4 + ss_hor) >> ss_hor, ch4 = (h4 + ss_ver) >> ss_ver;
    const HostTx *uvtx = &h_tx[b->uvtx], *ytx = &h_tx[b->u.p.max_ytx];
    for (int init_y = 0; init_y < bh4; init_y += 16)
        for (int init_x = 0; init_x < bw4; init_x += 16) {
            int y_off = !!init_y;
            for (int y = init_y; y < imin(h4, init_y + 16); y += ytx->h, y_off++) {
                int x_off = !!init_x;
                for (int x = init_x; x < imin(w4, init_x + 16); x += ytx->w, x_off++)
                    tx_tree(w, b, b->u.p.max_ytx, 0, x_off, y_off, bx + x, by + y, step[0]);
            }
            if (has_chroma)
                for (int pl = 0; pl < 2; pl++)
                    for (int y = init_y >> ss_ver; y < imin(ch4, (init_y + 16) >> ss_ver); y += uvtx->h)
                        for (int x = init_x >> ss_hor; x < imin(cw4, (init_x + 16) >> ss_hor); x += uvtx->w)
                            emit_tx(w, 1 + pl, b->uvtx, 4 * ((bx >> ss_hor) + x), 4 * ((by >> ss_ver) + y), step[1 + pl]);
        }
}

/* Intra block copy: recon_b_inter() on key / intra-only frames (src/recon_tmpl.c:1583-1597).  The prediction is mc() from the
 * frame's own reconstruction with the bilinear filter — luma at integer positions; chroma of a subsampled layout lands on half
 * positions for odd vectors, and a 4-wide / 4-high luma block predicts the chroma of its whole 8x8 with its own vector
 * (`bw4 << (bw4 == ss_hor)` from `bx & ~ss_hor`).  The pixels it reads were written by blocks earlier in decode order of the same
 * tile: the block's wavefront step is 1 + the largest step among the cells of the source rectangles, its residuals run in the same
 * step after the copy, and its own cells carry that step for whoever predicts from them next. */
static unsigned src_step(const Walk *w, const int pl, const int x_px, const int y_px, const int w_px, const int h_px) {
    const Dav1dHipLister *l = w->l;
    const int sh = pl ? l->ss_hor : 0, sv = pl ? l->ss_ver : 0;
    const int cw = (l->bw + sh) >> sh, ch = (l->bh + sv) >> sv;                     /* cells of the plane */
    /* ... of the TILE: decode_b keeps the source rectangle inside it (src/decode.c:1290-1336), and the cell behind the rectangle's last
     * column / row (the bilinear taps' pixel more, never read at the integer vectors an intra block copy has) would be the first of the
     * tile to the right / below when the rectangle ends at the tile's edge — cells another thread lists, or has not cleared yet (the
     * maps are recycled uncleared): a step drawn from there differs from run to run, and -ERANGE when the stale cell held 0xffff */
    const int tx_lo = w->col_start >> sh, tx_hi = imin((w->col_end + sh) >> sh, cw), ty_lo = w->row_start >> sv, ty_hi = imin((w->row_end + sv) >> sv, ch);
    const int x0 = iclip(x_px >> 2, tx_lo, tx_hi - 1), x1 = iclip((x_px + w_px) >> 2, tx_lo, tx_hi - 1);    /* one pixel more: the bilinear taps */
    const int y0 = iclip(y_px >> 2, ty_lo, ty_hi - 1), y1 = iclip((y_px + h_px) >> 2, ty_lo, ty_hi - 1);
    const uint16_t *m = l->step[pl];
    const int st = l->step_stride[pl];
    unsigned s = 0;
    for (int y = y0; y <= y1; y++)
        for (int x = x0; x <= x1; x++) s = m[y * st + x] > s ? m[y * st + x] : s;
    return s;
}

static void list_intrabc(Walk *w, const int bs, const Dav1dHipAv1Block *b, const int bx, const int by) {
    Dav1dHipLister *l = w->l;
    const int ss_hor = l->ss_hor, ss_ver = l->ss_ver, layout = l->d.layout;
    const uint8_t *b_dim = h_bs_dim[bs];
    const int bw4 = b_dim[0], bh4 = b_dim[1];
    const int has_chroma = layout != DAV1D_HIP_LAYOUT_I400 && (bw4 > ss_hor || (bx & 1)) && (bh4 > ss_ver || (by & 1));
    const Mv mv = mv_of(b->u.p.u.m.mv[0]);
    const size_t n0 = w->o->mc.n;
    /* the three predictions, emitted like any mc() call (reference 0 stands for the frame itself), then moved to the stepped list */
    emit_mc(w, DAV1D_HIP_MC_PUT, dst_off(l, 0, bx * 4, by * 4), bw4, bh4, bx, by, 0, mv, 0, 9 /* FILTER_2D_BILINEAR */);
    if (has_chroma)
        for (int pl = 1; pl < 3; pl++)
            emit_mc(w, DAV1D_HIP_MC_PUT, dst_off(l, 1, 4 * (bx >> ss_hor), 4 * (by >> ss_ver)), bw4 << (bw4 == ss_hor), bh4 << (bh4 == ss_ver),
                    bx & ~ss_hor, by & ~ss_ver, pl, mv, 0, 9);
    if (w->o->mc.n - n0 != (size_t) (has_chroma ? 3 : 1)) { w->err = -EINVAL; return; }       /* a scaled "reference 0": not here */
    unsigned s = 0;
    for (size_t i = n0; i < w->o->mc.n; i++) {
        const Dav1dHipMcTask *k = &w->o->mc.p[i];
        const int sh = k->plane ? ss_hor : 0, sv = k->plane ? ss_ver : 0;
        /* decode_b keeps the source inside the tile with the tile's right edge rounded UP to the block's width and the bottom at the
         * end of the superblock row (src/decode.c:1296-1336), so a window may leave the coded area to the right / below: the
         * reference emulates the edge of the f->bw * 4 x f->bh * 4 area then (mc() with refp == &f->sr_cur, src/recon_tmpl.c:
         * 960-978), the stepped copies clamp their coordinates to the same area (dav1d_hip_frame_submit_step_copy).  The cells the
         * clamped window reads are what the step waits for (src_step clips likewise). */
        (void) sh; (void) sv;
        const unsigned q = src_step(w, k->plane, k->src_x, k->src_y, k->w, k->h);
        if (q > s) s = q;
    }
    s++;
    /* the copies join the predictions of step s (DAV1D_HIP_IPRED_COPY), in pieces of up to 64x64: on every route of the intra wavefront
     * they run where an intra prediction of that step would — the superblock route included, which waits for the superblocks under the
     * source window to have passed step s - 1 */
    for (size_t i = n0; i < w->o->mc.n; i++) {
        const Dav1dHipMcTask m = w->o->mc.p[i];
        const int pl = m.plane, ssh = pl ? ss_hor : 0, ssv = pl ? ss_ver : 0;
        const int px0 = (int) (m.dst_off % (uint32_t) l->stride[pl]), py0 = (int) (m.dst_off / (uint32_t) l->stride[pl]);
        if ((m.w & 3) || (m.h & 3) || (px0 & 3) || (py0 & 3) || m.src_x < -32768 || m.src_x > 32767 - 128 || m.src_y < -32768 || m.src_y > 32767 - 128) { w->err = -EINVAL; return; }
        for (int oy = 0; oy < m.h; oy += 64)
            for (int ox = 0; ox < m.w; ox += 64) {
                Dav1dHipIpredTask *k = new_ipred(w, s);
                k->kind = DAV1D_HIP_IPRED_COPY;
                k->plane = (uint8_t) pl;
                k->dst_off = m.dst_off + (uint32_t) oy * (uint32_t) l->stride[pl] + (uint32_t) ox;
                k->x4 = (uint16_t) ((px0 + ox) >> 2); k->y4 = (uint16_t) ((py0 + oy) >> 2);
                k->w4 = (uint16_t) (w->col_end >> ssh); k->h4 = (uint16_t) (w->row_end >> ssv);
                k->tw = (uint8_t) (imin(64, m.w - ox) >> 2); k->th = (uint8_t) (imin(64, m.h - oy) >> 2);
                k->pal[0] = (uint16_t) (int16_t) (m.src_x + ox); k->pal[1] = (uint16_t) (int16_t) (m.src_y + oy);
                k->pal[2] = (uint16_t) (m.mx | m.my << 8);
                k->pal[6] = 0x8000u; k->pal[7] = (uint16_t) (s - 1);
            }
    }
    w->o->mc.n = n0;
    const unsigned step[3] = { s, s, s };
    list_inter_residuals(w, bs, b, bx, by, step);
    set_step(w, 0, bx, by, bw4, bh4, s);
    if (has_chroma) {
        const int cbw4 = (bw4 + ss_hor) >> ss_hor, cbh4 = (bh4 + ss_ver) >> ss_ver;
        for (int pl = 1; pl < 3; pl++) set_step(w, pl, bx >> ss_hor, by >> ss_ver, cbw4, cbh4, s);
    }
}

/* ------------------------------------------------------------------------------------------------ the walk */

/* decode_b(), pass-2 branch (src/decode.c:706-806) */
static void list_block(Walk *w, const int bs, const int edge_flags, const int bx, const int by) {
    Dav1dHipLister *l = w->l;
    const size_t bi = (size_t) by * l->d.b4_stride + bx;
    const Dav1dHipAv1Block *b = &l->d.b[bi];
    if (b->intra) list_intra(w, bs, edge_flags, b, bx, by);
    else if (l->d.is_inter) list_inter(w, bs, b, bx, by);
    else list_intrabc(w, bs, b, bx, by);         /* an inter-coded block of a key / intra-only frame: intra block copy */
    if (w->err) return;
    /* what later blocks need to know about this one: its identity along the bottom row and the right column */
    const int bw4 = h_bs_dim[bs][0], bh4 = h_bs_dim[bs][1];
    const int xe = imin(bx + bw4, l->bw), ye = imin(by + bh4, l->bh);
    uint32_t *o = l->owner;
    for (int x = bx; x < xe; x++) o[(size_t) (ye - 1) * l->d.b4_stride + x] = (uint32_t) bi;
    for (int y = by; y < ye - 1; y++) o[(size_t) y * l->d.b4_stride + xe - 1] = (uint32_t) bi;
}

/* edge availability of a node of the partition tree from (top has right, left has bottom), src/intra_edge.c:57-100 */
typedef struct Node { int o, h[2], v[2], h4, v4, split[3]; } Node;
static Node node_flags(const int bl, const int tr, const int lb) {
    Node n;
    const int e = (tr ? H_EDGE_ALL_TR : 0) | (lb ? H_EDGE_ALL_BL : 0);
    n.o = e;
    n.h[0] = e | H_EDGE_ALL_BL;
    n.v[0] = e | H_EDGE_ALL_TR;
    n.h4 = n.v4 = 0;
    n.split[0] = n.split[1] = n.split[2] = 0;
    if (bl == H_BL_8X8) {
        n.h[1] = e & (H_EDGE_ALL_BL | H_EDGE_I420_TR);
        n.v[1] = e & (H_EDGE_ALL_TR | H_EDGE_I420_BL | H_EDGE_I422_BL);
        n.split[0] = (e & H_EDGE_ALL_TR) | H_EDGE_I422_BL;
        n.split[1] = e | H_EDGE_I444_TR;
        n.split[2] = e & (H_EDGE_I420_TR | H_EDGE_I420_BL | H_EDGE_I422_BL);
    } else {
        n.h[1] = e & H_EDGE_ALL_BL;
        n.v[1] = e & H_EDGE_ALL_TR;
        n.h4 = H_EDGE_ALL_BL | (bl == H_BL_16X16 ? e & H_EDGE_I420_TR : 0);
        n.v4 = H_EDGE_ALL_TR | (bl == H_BL_16X16 ? e & (H_EDGE_I420_BL | H_EDGE_I422_BL) : 0);
    }
    return n;
}
/* child i of a split node: (top has right, left has bottom) */
static int child_tr(const int i, const int tr) { return !(i == 3 || (i == 1 && !tr)); }
static int child_lb(const int i, const int lb) { return i == 0 || (i == 2 && lb); }

/* decode_sb() with pass == 2, src/decode.c:2117-2375 */
static void walk_sb(Walk *w, const int bl, const int bx, const int by, const int tr, const int lb) {
    Dav1dHipLister *l = w->l;
    if (w->err) return;
    const int hsz = 16 >> bl;
    const int have_h_split = l->bw > bx + hsz, have_v_split = l->bh > by + hsz;
    if (!have_h_split && !have_v_split) { walk_sb(w, bl + 1, bx, by, child_tr(0, tr), child_lb(0, lb)); return; }
    const Dav1dHipAv1Block *b = &l->d.b[(size_t) by * l->d.b4_stride + bx];
    const Node n = node_flags(bl, tr, lb);
    const uint8_t (*sz)[2] = h_block_sizes[bl];
#define CHILD(i, x, y) walk_sb(w, bl + 1, x, y, child_tr(i, tr), child_lb(i, lb))
    if (have_h_split && have_v_split) {
        const int bp = b->bl == bl ? b->bp : H_PART_SPLIT;
        switch (bp) {
        case H_PART_NONE: list_block(w, sz[bp][0], n.o, bx, by); break;
        case H_PART_H:
            list_block(w, sz[bp][0], n.h[0], bx, by);
            list_block(w, sz[bp][0], n.h[1], bx, by + hsz);
            break;
        case H_PART_V:
            list_block(w, sz[bp][0], n.v[0], bx, by);
            list_block(w, sz[bp][0], n.v[1], bx + hsz, by);
            break;
        case H_PART_SPLIT:
            if (bl == H_BL_8X8) {
                list_block(w, H_BS_4x4, H_EDGE_ALL_TR | H_EDGE_ALL_BL, bx, by);
                list_block(w, H_BS_4x4, n.split[0], bx + 1, by);
                list_block(w, H_BS_4x4, n.split[1], bx, by + 1);
                list_block(w, H_BS_4x4, n.split[2], bx + 1, by + 1);
                /* an x86-64 producer realigns its coefficient cursor here (src/decode.c:2209-2218) */
                if (l->cf_align64) w->cur->cf = (w->cur->cf + 63) & ~(size_t) 63;
            } else {
                CHILD(0, bx, by); CHILD(1, bx + hsz, by); CHILD(2, bx, by + hsz); CHILD(3, bx + hsz, by + hsz);
            }
            break;
        case H_PART_T_TOP_SPLIT:
            list_block(w, sz[bp][0], H_EDGE_ALL_TR | H_EDGE_ALL_BL, bx, by);
            list_block(w, sz[bp][0], n.v[1], bx + hsz, by);
            list_block(w, sz[bp][1], n.h[1], bx, by + hsz);
            break;
        case H_PART_T_BOTTOM_SPLIT:
            list_block(w, sz[bp][0], n.h[0], bx, by);
            list_block(w, sz[bp][1], n.v[0], bx, by + hsz);
            list_block(w, sz[bp][1], 0, bx + hsz, by + hsz);
            break;
        case H_PART_T_LEFT_SPLIT:
            list_block(w, sz[bp][0], H_EDGE_ALL_TR | H_EDGE_ALL_BL, bx, by);
            list_block(w, sz[bp][0], n.h[1], bx, by + hsz);
            list_block(w, sz[bp][1], n.v[1], bx + hsz, by);
            break;
        case H_PART_T_RIGHT_SPLIT:
            list_block(w, sz[bp][0], n.v[0], bx, by);
            list_block(w, sz[bp][1], n.h[0], bx + hsz, by);
            list_block(w, sz[bp][1], 0, bx + hsz, by + hsz);
            break;
        case H_PART_H4:
            list_block(w, sz[bp][0], n.h[0], bx, by);
            list_block(w, sz[bp][0], n.h4, bx, by + (hsz >> 1));
            list_block(w, sz[bp][0], H_EDGE_ALL_BL, bx, by + hsz);
            if (by + (hsz * 3 >> 1) < l->bh) list_block(w, sz[bp][0], n.h[1], bx, by + (hsz * 3 >> 1));
            break;
        case H_PART_V4:
            list_block(w, sz[bp][0], n.v[0], bx, by);
            list_block(w, sz[bp][0], n.v4, bx + (hsz >> 1), by);
            list_block(w, sz[bp][0], H_EDGE_ALL_TR, bx + hsz, by);
            if (bx + (hsz * 3 >> 1) < l->bw) list_block(w, sz[bp][0], n.v[1], bx + (hsz * 3 >> 1), by);
            break;
        default:
            if (getenv("DAV1D_HIP_TRACE_LISTER"))
                fprintf(stderr, "lister: block at (%d, %d) level %d: bl %d bs %d bp %d intra %d (frame %dx%d, bw %d bh %d, b4_stride %d)\n", bx, by, bl, b->bl, b->bs, b->bp, b->intra,
                        l->d.w, l->d.h, l->bw, l->bh, (int) l->d.b4_stride);
            w->err = -EINVAL;
        }
    } else if (have_h_split) {
        if (b->bl != bl) { CHILD(0, bx, by); CHILD(1, bx + hsz, by); }
        else list_block(w, sz[H_PART_H][0], n.h[0], bx, by);
    } else {
        if (b->bl != bl) { CHILD(0, bx, by); CHILD(2, bx, by + hsz); }
        else list_block(w, sz[H_PART_V][0], n.v[0], bx, by);
    }
#undef CHILD
}

/* ------------------------------------------------------------------------------------------------ entry points */

/* the cell maps of finished listers, kept for the next frame of the size (a few per process) */
static struct { pthread_mutex_t m; struct { void *p; size_t bytes; } e[16]; int n; } map_pool = { PTHREAD_MUTEX_INITIALIZER, { { NULL, 0 } }, 0 };
static void *map_get(const size_t bytes) {
    void *p = NULL;
    pthread_mutex_lock(&map_pool.m);
    for (int i = 0; i < map_pool.n; i++)
        if (map_pool.e[i].bytes == bytes) { p = map_pool.e[i].p; map_pool.e[i] = map_pool.e[--map_pool.n]; break; }
    pthread_mutex_unlock(&map_pool.m);
    return p ? p : malloc(bytes);
}
static void map_put(void *p, const size_t bytes) {
    if (!p) return;
    pthread_mutex_lock(&map_pool.m);
    if (map_pool.n < 16) { map_pool.e[map_pool.n].p = p; map_pool.e[map_pool.n].bytes = bytes; map_pool.n++; p = NULL; }
    pthread_mutex_unlock(&map_pool.m);
    free(p);
}

int dav1d_hip_lister_create(Dav1dHipLister **out, const Dav1dHipFrameDesc *d, Dav1dHipFrame *frame) {
    if (!out || !d || !frame || !d->b || !d->cbi || !d->tile_start_off) return -EINVAL;
    *out = NULL;
    if (d->layout < 0 || d->layout > 3 || (d->bpc != 8 && d->bpc != 10 && d->bpc != 12) || d->w < 1 || d->h < 1) return -EINVAL;
    if (d->n_tile_cols < 1 || d->n_tile_cols > 64 || d->n_tile_rows < 1 || d->n_tile_rows > 64) return -EINVAL;
    h_tables_init();
    Dav1dHipPicture cur;
    int rc = dav1d_hip_frame_picture(frame, &cur);
    if (rc) return rc;
    if (cur.bpc != d->bpc || cur.layout != d->layout || cur.p[0].w != d->w || cur.p[0].h != d->h) return -EINVAL;
    Dav1dHipLister *l = (Dav1dHipLister *) calloc(1, sizeof(*l));
    if (!l) return -ENOMEM;
    __atomic_fetch_add(&dav1d_hip_live[2], 1, __ATOMIC_RELAXED);      /* dav1d_hip_live_objects: listers alive */
    l->d = *d;
    if (!d->is_inter) {
        /* key / intra-only frames have no references: whatever the caller's f->svc / refp still hold from the frame context's last
         * inter frame is not looked at (an intra block copy predicts from the frame itself, never scaled: src/recon_tmpl.c:1584-1597) */
        memset(l->d.svc, 0, sizeof(l->d.svc));
        memset(l->d.gmv_warp_allowed, 0, sizeof(l->d.gmv_warp_allowed));
    }
    l->frame = frame;
    l->ss_ver = d->layout == DAV1D_HIP_LAYOUT_I420;
    l->ss_hor = d->layout != DAV1D_HIP_LAYOUT_I444;
    l->bw = ((d->w + 7) >> 3) << 1; l->bh = ((d->h + 7) >> 3) << 1;      /* f->bw / f->bh: whole 8x8s, src/decode.c:3543-3544 */
    l->sb_step = d->sb128 ? 32 : 16;
    l->sbw = (d->w + l->sb_step * 4 - 1) / (l->sb_step * 4);
    l->sb_dep = (uint8_t *) calloc((size_t) l->sbw * (size_t) ((d->h + l->sb_step * 4 - 1) / (l->sb_step * 4)) + 1, 1);
    l->hbd = d->bpc > 8;
    l->csz = l->hbd ? 4 : 2;
    l->psz = l->hbd ? 2 : 1;
    l->cf_align64 = d->cf_align64;
    for (int p = 0; p < 3; p++) l->stride[p] = cur.p[p].data ? (int) (cur.p[p].stride / l->psz) : 0;
    /* the cell maps: recycled from frame to frame, NOT cleared here — 20 MB for an 8K frame, cleared (or faulted in page by page) on the
     * thread that begins the frame while the listing threads wait for it; a tile's first tile-sbrow clears the tile's share instead */
    const size_t rows = (size_t) ((l->bh + 31) & ~31);
    l->map_bytes = rows * (size_t) d->b4_stride * sizeof(uint16_t);
    l->owner = (uint32_t *) map_get(2 * l->map_bytes);
    for (int p = 0; p < 3; p++) {
        l->step_stride[p] = (int) d->b4_stride;
        l->step[p] = (uint16_t *) map_get(l->map_bytes);
    }
    /* (test hook: the recycled maps arrive full of 0xffff instead of whatever they held — a step drawn from a cell outside the share its
     * tile has cleared then fails with -ERANGE instead of passing unnoticed, tests/test_lister.py) */
    if (getenv("DAV1D_HIP_LISTER_POISON"))
        for (int p = 0; p < 3; p++) if (l->step[p]) memset(l->step[p], 0xff, l->map_bytes);
    const int n_tiles = d->n_tile_cols * d->n_tile_rows;
    l->tiles = (TileCursor *) calloc((size_t) n_tiles, sizeof(TileCursor));
    if (!l->owner || !l->step[0] || !l->step[1] || !l->step[2] || !l->tiles || !l->sb_dep) { dav1d_hip_lister_destroy(l); return -ENOMEM; }
    /* setup_tile(), src/decode.c:2438-2452: where a tile's share of cbi / cf / pal_idx starts */
    static const uint8_t size_mul[4][2] = { { 4, 4 }, { 6, 5 }, { 8, 6 }, { 12, 8 } };
    for (int t = 0; t < n_tiles; t++) {
        const size_t off = d->tile_start_off[t];
        l->tiles[t].pal_idx = off * size_mul[d->layout][1] / 8;
        l->tiles[t].cbi = off * size_mul[d->layout][0] / 64;
        l->tiles[t].cf = (off * size_mul[d->layout][0]) >> !l->hbd;
        l->tiles[t].next_sby = d->row_start_sb[t / d->n_tile_cols];
    }
    l->mask_bytes = h_masks()->size;
    /* the frame learns its tiles: intra blocks then run superblock by superblock (csrc/intra_sb.hip).  A frame that already holds intra
     * submissions (a second lister on one frame) keeps what it has. */
    (void) dav1d_hip_frame_set_tiling(frame, d->sb128, d->n_tile_cols, d->col_start_sb, d->n_tile_rows, d->row_start_sb);
    *out = l;
    return 0;
}

#ifdef LISTER_PROF
void dav1d_hip_lister_prof(void);
#endif
void dav1d_hip_lister_destroy(Dav1dHipLister *l) {
    if (!l) return;
    __atomic_fetch_sub(&dav1d_hip_live[2], 1, __ATOMIC_RELAXED);
#ifdef LISTER_PROF
    dav1d_hip_lister_prof();
    { extern void dav1d_hip_chunk_prof(void); dav1d_hip_chunk_prof(); }
#endif
    map_put(l->owner, 2 * l->map_bytes);
    for (int p = 0; p < 3; p++) map_put(l->step[p], l->map_bytes);
    free(l->tiles);
    free(l->sb_dep);
    free(l);
}

void dav1d_hip_lister_geo(const Dav1dHipLister *l, ListerGeo *g) {
    g->frame = l->frame;
    g->w = l->d.w; g->h = l->d.h; g->layout = l->d.layout; g->bpc = l->d.bpc; g->sb128 = l->d.sb128;
    g->ss_hor = l->ss_hor; g->ss_ver = l->ss_ver; g->bw = l->bw; g->bh = l->bh; g->sb_step = l->sb_step;
    for (int p = 0; p < 3; p++) g->stride[p] = l->stride[p];
    g->b4_stride = l->d.b4_stride;
    g->n_tile_cols = l->d.n_tile_cols; g->n_tile_rows = l->d.n_tile_rows;
    g->col_start_sb = l->d.col_start_sb; g->row_start_sb = l->d.row_start_sb;
}

size_t dav1d_hip_lister_prep_elems(const Dav1dHipLister *l) { return l ? (size_t) ((l->arena_bytes + 1) / 2 + 64) : 0; }
size_t dav1d_hip_lister_mask_bytes(const Dav1dHipLister *l) { return l ? (size_t) l->mask_bytes + 64 : 0; }
size_t dav1d_hip_lister_steps(const Dav1dHipLister *l) { return l ? l->max_step : 0; }
/* test aid: where a mask sits in the constant blob; which 0 = wedge[c][bs - BS_32x32][sign][idx], 1 = inter-intra[c][bs - BS_32x32][idx] */
long dav1d_hip_lister_mask_offset(const int which, const int c, const int bs, const int sign, const int idx) {
    const HostMasks *m = h_masks();
    if (c < 0 || c > 2 || bs < H_BS_32x32 || bs > H_BS_8x8 || sign < 0 || sign > 1 || idx < 0 || idx > (which ? 3 : 15)) return -1;
    const uint32_t o = which ? m->ii[c][bs - H_BS_32x32][idx] : m->wedge[c][bs - H_BS_32x32][sign][idx];
    return o == 0xffffffffu ? -1 : (long) o;
}
/* test aid: the derived geometry tables, flattened: h_bs_dim[22][4], h_tx[19] as { w, h, lw, lh, min, max, sub }, h_max_tx_for_bs[22][4],
 * h_block_sizes[5][10][2] */
void dav1d_hip_lister_tables(uint8_t *out) {
    h_tables_init();
    memcpy(out, h_bs_dim, sizeof(h_bs_dim)); out += sizeof(h_bs_dim);
    for (int i = 0; i < H_N_TX; i++) { memcpy(out, &h_tx[i], 7); out += 7; }
    memcpy(out, h_max_tx_for_bs, sizeof(h_max_tx_for_bs)); out += sizeof(h_max_tx_for_bs);
    memcpy(out, h_block_sizes, sizeof(h_block_sizes));
}
/* test aid: the warp set-up of a MM_WARP block (h_block_warp) */
int dav1d_hip_lister_block_warp(Dav1dHipWarpParams *wm, const int16_t *matrix, const int16_t *mv2d, int bw4, int bh4, int bx4, int by4) {
    h_block_warp(wm, matrix, mv2d, bw4, bh4, bx4, by4);
    return h_shear_params(wm);
}
const uint8_t *dav1d_hip_lister_const_masks(size_t *bytes) { const HostMasks *m = h_masks(); if (bytes) *bytes = m->size; return m->blob; }

#ifdef LISTER_PROF
#include <time.h>
#include <stdio.h>
static uint64_t prof_ns[4];
static uint64_t prof_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (uint64_t) ts.tv_sec * 1000000000ull + ts.tv_nsec; }
#define PROF_T(v) const uint64_t v = prof_now()
#define PROF_ADD(i, d) __atomic_fetch_add(&prof_ns[i], (d), __ATOMIC_RELAXED)
void dav1d_hip_lister_prof(void) { fprintf(stderr, "lister: walk %.2f ms, submit %.2f ms\n", prof_ns[0] * 1e-6, prof_ns[1] * 1e-6); prof_ns[0] = prof_ns[1] = 0; }
#else
#define PROF_T(v)
#define PROF_ADD(i, d)
#endif

/* stable counting sort of stepped records by step, then one submit per step */
static int submit_steps(Dav1dHipLister *l, Out *o) {
    unsigned hi = 0;
    for (size_t i = 0; i < o->ipred_step.n; i++) if (o->ipred_step.p[i] > hi) hi = o->ipred_step.p[i];
    for (size_t i = 0; i < o->blend_step.n; i++) if (o->blend_step.p[i] > hi) hi = o->blend_step.p[i];
    for (size_t i = 0; i < o->sitx_step.n; i++) if (o->sitx_step.p[i] > hi) hi = o->sitx_step.p[i];
    if (!hi) return 0;
    size_t *cnt = (size_t *) calloc(3 * ((size_t) hi + 2), sizeof(size_t));
    Dav1dHipIpredTask *ip = (Dav1dHipIpredTask *) malloc((o->ipred.n + 1) * sizeof(*ip));
    Dav1dHipCompTask *bl = (Dav1dHipCompTask *) malloc((o->blend.n + 1) * sizeof(*bl));
    Dav1dHipItxTask *ix = (Dav1dHipItxTask *) malloc((o->sitx.n + 1) * sizeof(*ix));
    if (!cnt || !ip || !bl || !ix) { free(cnt); free(ip); free(bl); free(ix); return -ENOMEM; }
    size_t *ci = cnt, *cb = cnt + hi + 2, *cx = cnt + 2 * ((size_t) hi + 2);
    for (size_t i = 0; i < o->ipred_step.n; i++) ci[o->ipred_step.p[i] + 1]++;
    for (size_t i = 0; i < o->blend_step.n; i++) cb[o->blend_step.p[i] + 1]++;
    for (size_t i = 0; i < o->sitx_step.n; i++) cx[o->sitx_step.p[i] + 1]++;
    for (unsigned s = 0; s <= hi; s++) { ci[s + 1] += ci[s]; cb[s + 1] += cb[s]; cx[s + 1] += cx[s]; }
    /* ci[s] = first slot of step s; filling advances it, so afterwards ci[s] = end of step s = start of step s + 1 */
    for (size_t i = 0; i < o->ipred.n; i++) ip[ci[o->ipred_step.p[i]]++] = o->ipred.p[i];
    for (size_t i = 0; i < o->blend.n; i++) bl[cb[o->blend_step.p[i]]++] = o->blend.p[i];
    for (size_t i = 0; i < o->sitx.n; i++) ix[cx[o->sitx_step.p[i]]++] = o->sitx.p[i];
    /* after the fill, ci[s] / cb[s] / cx[s] = end of step s: exactly what the one-call submission takes */
    const int rc = dav1d_hip_frame_submit_intra_sorted(l->frame, (size_t) hi + 1, ip, ci, ix, cx, bl, cb);
    free(cnt); free(ip); free(bl); free(ix);
    return rc;
}

/* A thread's task vectors live as long as the thread (dav1d's workers, the library's pool): a tile-sbrow starts with the capacity the
 * last one grew to instead of growing a dozen vectors from nothing — 1,800 times per 8K frame, each growth step of the larger ones
 * a trip to the kernel's address-space lock that every other listing thread waits on. */
static pthread_key_t out_key;
static pthread_once_t out_once = PTHREAD_ONCE_INIT;
static __thread Out *out_tls;
static void out_free(void *p) {
    Out *o = (Out *) p;
    if (!o) return;
    free(o->prec.p);
    free(o->mc.p); free(o->comp.p); free(o->warp.p); free(o->scaled.p); free(o->itx.p); free(o->itx_dep.p);
    free(o->ipred.p); free(o->ipred_step.p); free(o->blend.p); free(o->blend_step.p); free(o->sitx.p); free(o->sitx_step.p);
    free(o);
}
static void out_key_make(void) { (void) pthread_key_create(&out_key, out_free); }
static Out *out_get(void) {
    Out *o = out_tls;
    if (!o) {
        pthread_once(&out_once, out_key_make);
        o = (Out *) calloc(1, sizeof(*o));
        if (!o) return NULL;
        out_tls = o;
        (void) pthread_setspecific(out_key, o);
    }
    o->mc.n = o->comp.n = o->warp.n = o->scaled.n = o->itx.n = o->itx_dep.n = o->ipred.n = o->ipred_step.n = o->blend.n = o->blend_step.n = 0;
    o->sitx.n = o->sitx_step.n = 0;
    o->npack = 0; o->prec.n = 0;
    return o;
}

int dav1d_hip_lister_tile_sbrow(Dav1dHipLister *l, const int tile_row, const int tile_col, const int sby) {
    if (!l || tile_row < 0 || tile_row >= l->d.n_tile_rows || tile_col < 0 || tile_col >= l->d.n_tile_cols) return -EINVAL;
    TileCursor *cur = &l->tiles[tile_row * l->d.n_tile_cols + tile_col];
    if (sby != cur->next_sby || sby >= l->d.row_start_sb[tile_row + 1]) return -EINVAL;      /* top to bottom inside a tile */
    const int sb_shift = l->d.sb128 ? 5 : 4;
    Out *const op = out_get();
    if (!op) return -ENOMEM;
    Walk w;
    w.l = l; w.o = op; w.cur = cur; w.err = 0;
    w.seen_step = 0; w.xs_step = w.xs_mask = 0;
    w.hint_on = 0;
    for (int pl = 0; pl < 3; pl++) { w.bdep[pl] = 0; w.cand[pl] = 0; w.cand_off[pl] = 0; w.cand_dim[pl] = 0; }
    w.col_start = l->d.col_start_sb[tile_col] << sb_shift;
    w.col_end = imin(l->d.col_start_sb[tile_col + 1] << sb_shift, l->bw);
    w.row_start = l->d.row_start_sb[tile_row] << sb_shift;
    w.row_end = imin(l->d.row_start_sb[tile_row + 1] << sb_shift, l->bh);
    const int by = sby << sb_shift;
    if (sby == l->d.row_start_sb[tile_row]) {
        /* the tile's share of the cell maps: zero = "final before the wavefront starts" (what an inter block's cells stay at), and
         * dep_step() looks at cells of rows that are listed later (the bottom-left extension of an edge) */
        /* ... of THIS tile only: the rows of the tile below belong to whoever lists that tile's first superblock row, which may
         * have happened already (tile-sbrows of different tiles come in any order); only the last tile row owns the padding rows */
        const int y_end = tile_row + 1 < l->d.n_tile_rows ? imin(l->d.row_start_sb[tile_row + 1] << sb_shift, (l->bh + 31) & ~31) : (l->bh + 31) & ~31;
        const size_t x0 = (size_t) w.col_start, nx = (size_t) (imin(l->d.col_start_sb[tile_col + 1] << sb_shift, (int) l->d.b4_stride) - w.col_start);
        for (int y = w.row_start; y < y_end; y++) memset(l->owner + (size_t) y * l->d.b4_stride + x0, 0, nx * sizeof(uint32_t));
        for (int p = 0; p < 3; p++) {            /* the chroma maps count in cells of their plane */
            const int sh = p ? l->ss_hor : 0, sv = p ? l->ss_ver : 0;
            const size_t px0 = x0 >> sh, pnx = ((x0 + nx + sh) >> sh) - px0;
            for (int y = w.row_start >> sv; y < (y_end + sv) >> sv; y++) memset(l->step[p] + (size_t) y * l->step_stride[p] + px0, 0, pnx * sizeof(uint16_t));
        }
    }
    v_oom = 0;
    PROF_T(t0);
    for (int bx = w.col_start; bx < w.col_end && !w.err && !v_oom; bx += l->sb_step)
        walk_sb(&w, l->d.sb128 ? H_BL_128X128 : H_BL_64X64, bx, by, 1, 0);
    int rc = v_oom ? -ENOMEM : w.err;
    void *pack_dst = NULL;
    if (!rc && l->d.cf && (op->itx.n || op->sitx.n)) {
        /* the tile-sbrow's values get their place in the frame's coefficient arena; the tasks counted from the start of the row */
        uint32_t base = 0;
        rc = dav1d_hip_frame_reserve_coefs(l->frame, op->npack, &base, &pack_dst);
        for (size_t i = 0; i < op->itx.n; i++) op->itx.p[i].cf_off += base;
        for (size_t i = 0; i < op->sitx.n; i++) op->sitx.p[i].cf_off += base;
    }
    PROF_T(t1);
    if (!rc) rc = dav1d_hip_frame_submit_tile_sbrow_packing(l->frame, op->mc.p, op->mc.n, op->comp.p, op->comp.n, op->itx.p, op->itx.n, op->itx_dep.n == op->itx.n ? op->itx_dep.p : NULL,
                                                            op->prec.p, pack_dst ? op->prec.n : 0, (void *) l->d.cf, pack_dst);
    PROF_T(t2);
    PROF_ADD(0, t1 - t0); PROF_ADD(1, t2 - t1);
    if (!rc && op->warp.n) rc = dav1d_hip_frame_submit_warp(l->frame, op->warp.p, op->warp.n);
    if (!rc && op->scaled.n) rc = dav1d_hip_frame_submit_scaled(l->frame, op->scaled.p, op->scaled.n);
    if (!rc && (op->ipred.n || op->sitx.n)) {
        /* what this row of superblocks reads of its neighbours (dep_step): a frame without a tiling has no use for it */
        const int sbx0 = w.col_start / l->sb_step, sbx1 = (w.col_end + l->sb_step - 1) / l->sb_step;
        (void) dav1d_hip_frame_set_sb_deps(l->frame, (uint32_t) (sby * l->sbw + sbx0), (size_t) (sbx1 - sbx0), l->sb_dep + (size_t) sby * l->sbw + sbx0);
    }
    if (!rc) rc = submit_steps(l, op);
    if (!rc) cur->next_sby = sby + 1;
    if (rc && getenv("DAV1D_HIP_TRACE_LISTER"))
        fprintf(stderr, "lister: tile (%d, %d) sby %d: rc %d (walk err %d, oom %d; %zu mc %zu comp %zu warp %zu scaled %zu itx %zu ipred %zu sitx)\n", tile_row, tile_col, sby, rc,
                w.err, v_oom, op->mc.n, op->comp.n, op->warp.n, op->scaled.n, op->itx.n, op->ipred.n, op->sitx.n);
    return rc;
}

/* ---- the library's host threads */
static struct HostPool {
    pthread_mutex_t m;
    pthread_cond_t work, done, idle;
    void *(*fn)(void *);
    void *arg;
    int created, want, claimed, finished, busy;
    unsigned gen;
} h_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, NULL, 0, 0, 0, 0, 0, 0 };

static void *pool_thread(void *unused) {
    (void) unused;
    unsigned seen = 0;
    pthread_mutex_lock(&h_pool.m);
    for (;;) {
        while (h_pool.gen == seen || h_pool.claimed >= h_pool.want) { seen = h_pool.gen; pthread_cond_wait(&h_pool.work, &h_pool.m); }
        seen = h_pool.gen;
        h_pool.claimed++;
        void *(*fn)(void *) = h_pool.fn;
        void *arg = h_pool.arg;
        pthread_mutex_unlock(&h_pool.m);
        fn(arg);
        pthread_mutex_lock(&h_pool.m);
        if (++h_pool.finished == h_pool.want) pthread_cond_signal(&h_pool.done);
    }
    return NULL;
}

void dav1d_hip_host_pool_run(void *(*fn)(void *), void *arg, int n) {
    if (n > 1) {
        pthread_mutex_lock(&h_pool.m);
        while (h_pool.busy) pthread_cond_wait(&h_pool.idle, &h_pool.m);
        h_pool.busy = 1;
        while (h_pool.created < n - 1) {
            pthread_t t;
            pthread_attr_t at;
            pthread_attr_init(&at);
            pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
            const int rc = pthread_create(&t, &at, pool_thread, NULL);
            pthread_attr_destroy(&at);
            if (rc) break;
            h_pool.created++;
        }
        h_pool.fn = fn; h_pool.arg = arg;
        h_pool.want = h_pool.created < n - 1 ? h_pool.created : n - 1;
        h_pool.claimed = h_pool.finished = 0;
        h_pool.gen++;
        pthread_cond_broadcast(&h_pool.work);
        pthread_mutex_unlock(&h_pool.m);
    }
    fn(arg);
    if (n > 1) {
        pthread_mutex_lock(&h_pool.m);
        while (h_pool.finished < h_pool.want) pthread_cond_wait(&h_pool.done, &h_pool.m);
        h_pool.want = 0;
        h_pool.busy = 0;
        pthread_cond_signal(&h_pool.idle);
        pthread_mutex_unlock(&h_pool.m);
    }
}

/* Every tile of the frame on n_threads threads of the library: tiles are handed out in raster order under a mutex, a thread walks
 * its tile's superblock rows top to bottom (dav1d_hip_lister_tile_sbrow).  A caller with a thread pool of its own — dav1d's task
 * threads — calls dav1d_hip_lister_tile_sbrow itself; this is for callers without one (and for timing the host side without a
 * foreign runtime's locks in the way). */
typedef struct RunAll { Dav1dHipLister *l; const Dav1dHipFilterDesc *fd; pthread_mutex_t mtx; int next, n_filter, err; } RunAll;
static void *run_worker(void *arg) {
    RunAll *r = (RunAll *) arg;
    const int n_tiles = r->l->d.n_tile_cols * r->l->d.n_tile_rows;
    for (;;) {
        pthread_mutex_lock(&r->mtx);
        const int k = r->err ? n_tiles + r->n_filter : r->next++;
        pthread_mutex_unlock(&r->mtx);
        if (k >= n_tiles + r->n_filter) break;
        if (k >= n_tiles) {
            /* the filter lists of the frame (dav1d_hip_lister_run_frame): short units that fill the threads' time once the tiles are handed out */
            const int rc = dav1d_hip_lister_filter_unit(r->l, r->fd, k - n_tiles);
            if (rc) { pthread_mutex_lock(&r->mtx); if (!r->err) r->err = rc; pthread_mutex_unlock(&r->mtx); }
            continue;
        }
        const int tr = k / r->l->d.n_tile_cols, tc = k % r->l->d.n_tile_cols;
        int rc = 0;
        for (int sby = r->l->d.row_start_sb[tr]; sby < r->l->d.row_start_sb[tr + 1] && !rc; sby++) rc = dav1d_hip_lister_tile_sbrow(r->l, tr, tc, sby);
        if (rc) { pthread_mutex_lock(&r->mtx); if (!r->err) r->err = rc; pthread_mutex_unlock(&r->mtx); }
    }
    return NULL;
}
int dav1d_hip_lister_run(Dav1dHipLister *l, int n_threads) {
    if (!l || n_threads < 1) return -EINVAL;
    const int n_tiles = l->d.n_tile_cols * l->d.n_tile_rows;
    if (n_threads > n_tiles) n_threads = n_tiles;
    if (n_threads > 256) n_threads = 256;
    RunAll r;
    r.l = l; r.fd = NULL; r.next = 0; r.n_filter = 0; r.err = 0;
    pthread_mutex_init(&r.mtx, NULL);
    dav1d_hip_host_pool_run(run_worker, &r, n_threads);
    pthread_mutex_destroy(&r.mtx);
    if (!r.err) (void) dav1d_hip_frame_flush(l->frame);         /* every tile is in: the lists start their way to the device */
    return r.err;
}
/* dav1d_hip_lister_run and dav1d_hip_lister_filter_run as ONE job: the tiles first, the filter lists (three per superblock row) behind
 * them in the same queue — a thread that is through with its tiles takes filter lists instead of waiting for the slowest tile. */
int dav1d_hip_lister_run_frame(Dav1dHipLister *l, const Dav1dHipFilterDesc *fd, int n_threads) {
    if (!l || !fd || n_threads < 1) return -EINVAL;
    RunAll r;
    r.l = l; r.fd = fd; r.next = 0; r.err = 0;
    r.n_filter = dav1d_hip_lister_filter_units(l);
    const int n_units = l->d.n_tile_cols * l->d.n_tile_rows + r.n_filter;
    if (n_threads > n_units) n_threads = n_units;
    if (n_threads > 256) n_threads = 256;
    pthread_mutex_init(&r.mtx, NULL);
    dav1d_hip_host_pool_run(run_worker, &r, n_threads);
    pthread_mutex_destroy(&r.mtx);
    if (!r.err) (void) dav1d_hip_frame_flush(l->frame);
    return r.err;
}

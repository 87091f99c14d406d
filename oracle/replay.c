/* CPU replay of flat task lists through an oracle's DSP function pointers.
 * TEST / BASELINE INFRASTRUCTURE ONLY (used by tests/ and bench.py's cpu_baseline leg).
 *
 * The same POD task lists that go to the HIP backend (include/dav1d_hip.h) are walked
 * here on the host, one task at a time, calling the oracle's functions with the
 * arguments the reference drivers would pass:
 *   - itx:  dsp->itx.itxfm_add[tx][txtp](dst, stride, coeff, eob)   (reference
 *           src/recon_tmpl.c:811-816)
 *   - mc:   emu_edge when the window leaves the visible plane, then mc / mct
 *           (reference src/recon_tmpl.c:960-989)
 *   - comp: avg / w_avg / mask / w_mask (reference src/recon_tmpl.c:1802-1826)
 * The function pointers come from oracle/_ref (the reference's own C code) or from
 * oracle/port via their *_dsp_entry() getters, so this file contains no codec math.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include "../include/dav1d_hip.h"

typedef void *(*entry_fn)(int bpc, const char *family, int i, int j);

typedef struct ReplayPlanes {
    void *data[3];
    ptrdiff_t stride[3];   /* bytes */
    int w[3], h[3];        /* visible */
} ReplayPlanes;

static const uint8_t tx_w[19] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
static const uint8_t tx_h[19] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };

typedef void (*itx8_fn)(uint8_t *, ptrdiff_t, int16_t *, int);
typedef void (*itx16_fn)(uint16_t *, ptrdiff_t, int32_t *, int, int);

int dav1d_replay_itx(entry_fn entry, int bpc, const ReplayPlanes *dst, const Dav1dHipItxTask *t, size_t n, void *coef)
{
    void *tab[19][17];
    memset(tab, 0, sizeof(tab));
    const int bdmax = (1 << bpc) - 1;
    for (size_t i = 0; i < n; i++) {
        const Dav1dHipItxTask *k = &t[i];
        if (!tab[k->tx][k->txtp]) tab[k->tx][k->txtp] = entry(bpc, "itxfm_add", k->tx, k->txtp);
        if (!tab[k->tx][k->txtp]) return -1;
        if (bpc == 8)
            ((itx8_fn) tab[k->tx][k->txtp])((uint8_t *) dst->data[k->plane] + k->dst_off, dst->stride[k->plane],
                                             (int16_t *) coef + k->cf_off, k->eob);
        else
            ((itx16_fn) tab[k->tx][k->txtp])((uint16_t *) dst->data[k->plane] + k->dst_off, dst->stride[k->plane],
                                              (int32_t *) coef + k->cf_off, k->eob, bdmax);
    }
    (void) tx_w; (void) tx_h;
    return 0;
}

typedef void (*emu_fn)(intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, void *, ptrdiff_t, const void *, ptrdiff_t);
typedef void (*mc8_fn)(uint8_t *, ptrdiff_t, const uint8_t *, ptrdiff_t, int, int, int, int);
typedef void (*mc16_fn)(uint16_t *, ptrdiff_t, const uint16_t *, ptrdiff_t, int, int, int, int, int);
typedef void (*mct8_fn)(int16_t *, const uint8_t *, ptrdiff_t, int, int, int, int);
typedef void (*mct16_fn)(int16_t *, const uint16_t *, ptrdiff_t, int, int, int, int, int);

int dav1d_replay_mc(entry_fn entry, int bpc, const ReplayPlanes *dst, const ReplayPlanes *refs,
                    const Dav1dHipMcTask *t, size_t n, int16_t *prep)
{
    void *put[10], *prp[10];
    for (int f = 0; f < 10; f++) { put[f] = entry(bpc, "mc", f, 0); prp[f] = entry(bpc, "mct", f, 0); }
    const emu_fn emu = (emu_fn) entry(bpc, "emu_edge", 0, 0);
    if (!emu) return -1;
    const int bps = bpc > 8 ? 2 : 1, bdmax = (1 << bpc) - 1;
    static __thread uint16_t emu_buf[192 * 192];
    for (size_t i = 0; i < n; i++) {
        const Dav1dHipMcTask *k = &t[i];
        const ReplayPlanes *rp = &refs[k->ref];
        const int pl = k->plane, w = k->w, h = k->h, mx = k->mx, my = k->my;
        const int dx = k->src_x, dy = k->src_y;
        const uint8_t *src;
        ptrdiff_t ss = rp->stride[pl];
        /* reference src/recon_tmpl.c:967-981 */
        if (dx < !!mx * 3 || dy < !!my * 3 || dx + w + !!mx * 4 > rp->w[pl] || dy + h + !!my * 4 > rp->h[pl]) {
            emu(w + !!mx * 7, h + !!my * 7, rp->w[pl], rp->h[pl], dx - !!mx * 3, dy - !!my * 3,
                emu_buf, 192 * bps, rp->data[pl], ss);
            src = (const uint8_t *) emu_buf + (192 * !!my * 3 + !!mx * 3) * bps;
            ss = 192 * bps;
        } else {
            src = (const uint8_t *) rp->data[pl] + dy * ss + (ptrdiff_t) dx * bps;
        }
        if (k->kind == DAV1D_HIP_MC_PUT) {
            uint8_t *d = (uint8_t *) dst->data[pl] + (size_t) k->dst_off * bps;
            if (bpc == 8) ((mc8_fn) put[k->filter_2d])(d, dst->stride[pl], src, ss, w, h, mx, my);
            else ((mc16_fn) put[k->filter_2d])((uint16_t *) d, dst->stride[pl], (const uint16_t *) src, ss, w, h, mx, my, bdmax);
        } else {
            if (bpc == 8) ((mct8_fn) prp[k->filter_2d])(prep + k->dst_off, src, ss, w, h, mx, my);
            else ((mct16_fn) prp[k->filter_2d])(prep + k->dst_off, (const uint16_t *) src, ss, w, h, mx, my, bdmax);
        }
    }
    return 0;
}

typedef void (*avg8_fn)(uint8_t *, ptrdiff_t, const int16_t *, const int16_t *, int, int);
typedef void (*avg16_fn)(uint16_t *, ptrdiff_t, const int16_t *, const int16_t *, int, int, int);
typedef void (*wavg8_fn)(uint8_t *, ptrdiff_t, const int16_t *, const int16_t *, int, int, int);
typedef void (*wavg16_fn)(uint16_t *, ptrdiff_t, const int16_t *, const int16_t *, int, int, int, int);
typedef void (*mask8_fn)(uint8_t *, ptrdiff_t, const int16_t *, const int16_t *, int, int, const uint8_t *);
typedef void (*mask16_fn)(uint16_t *, ptrdiff_t, const int16_t *, const int16_t *, int, int, const uint8_t *, int);
typedef void (*wmask8_fn)(uint8_t *, ptrdiff_t, const int16_t *, const int16_t *, int, int, uint8_t *, int);
typedef void (*wmask16_fn)(uint16_t *, ptrdiff_t, const int16_t *, const int16_t *, int, int, uint8_t *, int, int);

int dav1d_replay_comp(entry_fn entry, int bpc, const ReplayPlanes *dst, const Dav1dHipCompTask *t, size_t n,
                      const int16_t *prep, uint8_t *mask)
{
    void *f_avg = entry(bpc, "avg", 0, 0), *f_wavg = entry(bpc, "w_avg", 0, 0), *f_mask = entry(bpc, "mask", 0, 0);
    void *f_wm[3] = { entry(bpc, "w_mask", 0, 0), entry(bpc, "w_mask", 1, 0), entry(bpc, "w_mask", 2, 0) };
    const int bps = bpc > 8 ? 2 : 1, bdmax = (1 << bpc) - 1;
    for (size_t i = 0; i < n; i++) {
        const Dav1dHipCompTask *k = &t[i];
        uint8_t *d = (uint8_t *) dst->data[k->plane] + (size_t) k->dst_off * bps;
        const ptrdiff_t ds = dst->stride[k->plane];
        const int16_t *a = prep + k->tmp1_off, *b = prep + k->tmp2_off;
        switch (k->kind) {
        case DAV1D_HIP_COMP_AVG:
            if (bpc == 8) ((avg8_fn) f_avg)(d, ds, a, b, k->w, k->h);
            else ((avg16_fn) f_avg)((uint16_t *) d, ds, a, b, k->w, k->h, bdmax);
            break;
        case DAV1D_HIP_COMP_WAVG:
            if (bpc == 8) ((wavg8_fn) f_wavg)(d, ds, a, b, k->w, k->h, k->arg);
            else ((wavg16_fn) f_wavg)((uint16_t *) d, ds, a, b, k->w, k->h, k->arg, bdmax);
            break;
        case DAV1D_HIP_COMP_MASK:
            if (bpc == 8) ((mask8_fn) f_mask)(d, ds, a, b, k->w, k->h, mask + k->mask_off);
            else ((mask16_fn) f_mask)((uint16_t *) d, ds, a, b, k->w, k->h, mask + k->mask_off, bdmax);
            break;
        default:
            if (bpc == 8) ((wmask8_fn) f_wm[k->ss])(d, ds, a, b, k->w, k->h, mask + k->mask_off, k->arg);
            else ((wmask16_fn) f_wm[k->ss])((uint16_t *) d, ds, a, b, k->w, k->h, mask + k->mask_off, k->arg, bdmax);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ post filters
 * The same marshalling for the in-loop filters: every task becomes the DSP call (or calls) the reference drivers
 * would issue, on host planes.  Only driver-side glue is restated here (which rows go into `left` / `top` / `lpf`,
 * the variance adjustment of the CDEF primary strength); the filters themselves are the oracle's. */
#include <stdlib.h>

typedef void (*lf8_fn)(uint8_t *, ptrdiff_t, const uint32_t *, const uint8_t (*)[4], ptrdiff_t, const void *, int);
typedef void (*lf16_fn)(uint16_t *, ptrdiff_t, const uint32_t *, const uint8_t (*)[4], ptrdiff_t, const void *, int, int);

/* dsp->lf.loop_filter_sb as dav1d_loopfilter_sbrow_cols / _rows call it (reference src/lf_apply_tmpl.c:176-311): every
 * column-edge task, then every row-edge task.  lut = Av1FilterLUT { e[64], i[64], sharp[2] }. */
int dav1d_replay_lf(entry_fn entry, int bpc, const ReplayPlanes *dst, const Dav1dHipLfTask *t, size_t n, const uint8_t *lvl,
                    ptrdiff_t b4_stride, const void *lut)
{
    const int bdmax = (1 << bpc) - 1, bps = bpc > 8 ? 2 : 1;
    for (int dir = 0; dir < 2; dir++)
        for (size_t i = 0; i < n; i++) {
            const Dav1dHipLfTask *k = &t[i];
            if (k->dir != dir) continue;
            void *fn = entry(bpc, "loop_filter_sb", k->plane != 0, dir);
            if (!fn) return -1;
            uint8_t *p = (uint8_t *) dst->data[k->plane] + (size_t) k->dst_off * bps;
            const uint8_t (*l)[4] = (const uint8_t (*)[4]) (lvl + (size_t) k->lvl_off * 4 + k->lvl_comp);
            if (bpc == 8) ((lf8_fn) fn)(p, dst->stride[k->plane], k->vmask, l, b4_stride, lut, 32);
            else ((lf16_fn) fn)((uint16_t *) p, dst->stride[k->plane], k->vmask, l, b4_stride, lut, 32, bdmax);
        }
    return 0;
}

typedef int (*cdir8_fn)(const uint8_t *, ptrdiff_t, unsigned *);
typedef int (*cdir16_fn)(const uint16_t *, ptrdiff_t, unsigned *, int);
typedef void (*cfb8_fn)(uint8_t *, ptrdiff_t, const void *, const uint8_t *, const uint8_t *, int, int, int, int, int);
typedef void (*cfb16_fn)(uint16_t *, ptrdiff_t, const void *, const uint16_t *, const uint16_t *, int, int, int, int, int, int);

static void cdef_block(void *fn, int bpc, const ReplayPlanes *src, const ReplayPlanes *dst, int pl, int px0, int py0, int w, int h,
                       int pri, int sec, int dir, int damping, int edges, int rep_bot)
{
    const int bps = bpc > 8 ? 2 : 1, bdmax = (1 << bpc) - 1;
    const ptrdiff_t st = src->stride[pl];
    const uint8_t *s = (const uint8_t *) src->data[pl];
    uint8_t left[8 * 2 * 2];
    memset(left, 0, sizeof(left));
    if (px0 >= 2) for (int y = 0; y < h; y++) memcpy(left + y * 2 * bps, s + (py0 + y) * st + (px0 - 2) * bps, 2 * bps);
    /* rows above / below come from the unfiltered picture; when an edge flag is clear the pointer is never read */
    const uint8_t *top = s + (py0 >= 2 ? py0 - 2 : 0) * st + px0 * bps;
    const uint8_t *bot = s + (py0 + h < src->h[pl] ? py0 + h : src->h[pl] - 1) * st + px0 * bps;
    uint8_t *bot2 = NULL;              /* DAV1D_HIP_CDEF_BOT_REP_*: the first row below, twice (what backup_lpf saved), rows dst-stride apart */
    if (rep_bot) {
        const ptrdiff_t ds = dst->stride[pl];
        const int lo = px0 >= 2 ? 2 : 0;
        bot2 = malloc((size_t) ds + (size_t) (w + 4) * bps + 64);
        if (bot2) {
            for (int r = 0; r < 2; r++) memcpy(bot2 + r * ds + (2 - lo) * bps, s + (py0 + h) * st + (px0 - lo) * bps, (size_t) (w + 2 + lo) * bps);
            bot = bot2 + 2 * bps;
        }
    }
    uint8_t *blk = (uint8_t *) dst->data[pl] + py0 * dst->stride[pl] + px0 * bps;
    if (bpc == 8) ((cfb8_fn) fn)(blk, dst->stride[pl], left, top, bot, pri, sec, dir, damping, edges);
    else ((cfb16_fn) fn)((uint16_t *) blk, dst->stride[pl], left, (const uint16_t *) top, (const uint16_t *) bot, pri, sec, dir, damping, edges, bdmax);
    free(bot2);
}

/* One 8x8 unit of dav1d_cdef_brow (reference src/cdef_apply_tmpl.c:149-290), out of place: `dst` starts as a copy of
 * `src`; direction search and variance adjustment (adjust_strength, :91-95) as the driver does them; the chroma
 * direction remap of 4:2:2 (:115-117).  src and dst must share strides. */
int dav1d_replay_cdef(entry_fn entry, int bpc, int layout, const ReplayPlanes *src, const ReplayPlanes *dst, const Dav1dHipCdefTask *t,
                      size_t n, int damping)
{
    const int ss_ver = layout == 1, ss_hor = layout != 3 && layout != 0;
    void *dirfn = entry(bpc, "cdef_dir", 0, 0), *fby = entry(bpc, "cdef_fb", 0, 0), *fbuv = layout ? entry(bpc, "cdef_fb", 3 - layout, 0) : NULL;
    if (!dirfn || !fby) return -1;
    static const uint8_t uv422[8] = { 7, 0, 2, 4, 5, 6, 6, 6 };
    for (size_t i = 0; i < n; i++) {
        const Dav1dHipCdefTask *k = &t[i];
        if (k->flags & 1) continue;                          /* raw single-call tasks are not part of the frame flow */
        const int x0 = k->bx * 8, y0 = k->by * 8;
        int dir = 0;
        unsigned var = 0;
        if (k->y_pri || k->uv_pri) {
            const uint8_t *blk = (const uint8_t *) src->data[0] + y0 * src->stride[0] + x0 * (bpc > 8 ? 2 : 1);
            dir = bpc == 8 ? ((cdir8_fn) dirfn)(blk, src->stride[0], &var) : ((cdir16_fn) dirfn)((const uint16_t *) blk, src->stride[0], &var, (1 << bpc) - 1);
        }
        if (k->y_pri) {
            int adj = 0;
            if (var) {
                int lg = 0;
                for (unsigned v = var >> 6; v > 1; v >>= 1) lg++;
                const int idx = (var >> 6) ? (lg < 12 ? lg : 12) : 0;
                adj = (k->y_pri * (4 + idx) + 8) >> 4;
            }
            if (adj || k->y_sec) cdef_block(fby, bpc, src, dst, 0, x0, y0, 8, 8, adj, k->y_sec, dir, damping, k->edges, k->flags & DAV1D_HIP_CDEF_BOT_REP_Y);
        } else if (k->y_sec) {
            cdef_block(fby, bpc, src, dst, 0, x0, y0, 8, 8, 0, k->y_sec, 0, damping, k->edges, k->flags & DAV1D_HIP_CDEF_BOT_REP_Y);
        }
        if (fbuv && (k->uv_pri || k->uv_sec)) {
            const int uvdir = k->uv_pri ? (layout == 2 ? uv422[dir] : dir) : 0;
            for (int pl = 1; pl <= 2; pl++)
                cdef_block(fbuv, bpc, src, dst, pl, x0 >> ss_hor, y0 >> ss_ver, 8 >> ss_hor, 8 >> ss_ver, k->uv_pri, k->uv_sec, uvdir,
                           damping - 1, k->edges, k->flags & DAV1D_HIP_CDEF_BOT_REP_UV);
        }
    }
    return 0;
}

typedef void (*lr8_fn)(uint8_t *, ptrdiff_t, const void *, const uint8_t *, int, int, const void *, int);
typedef void (*lr16_fn)(uint16_t *, ptrdiff_t, const void *, const uint16_t *, int, int, const void *, int, int);

/* One looprestorationfilter_fn call per task as lr_stripe() issues it (reference src/lr_apply_tmpl.c:36-97), out of place:
 * `dst` starts as a copy of `src` (the tasks are walked in the order given, which must be raster order so that a unit's
 * right neighbour is still unfiltered); `left` = the 4 columns left of the unit from `src`; the lpf buffer = rows y-2,
 * y-1 at rows 0, 1 and rows y+h, y+h+1 at rows 6, 7 of an 8-row scratch with the plane's stride, from `lpf`. */
int dav1d_replay_lr(entry_fn entry, int bpc, const ReplayPlanes *src, const ReplayPlanes *lpf, const ReplayPlanes *dst,
                    const Dav1dHipLrTask *t, size_t n)
{
    const int bps = bpc > 8 ? 2 : 1, bdmax = (1 << bpc) - 1;
    ptrdiff_t max_stride = 0;
    for (int pl = 0; pl < 3; pl++) if (src->data[pl] && src->stride[pl] > max_stride) max_stride = src->stride[pl];
    uint8_t *rows = malloc((size_t) 8 * max_stride + 64);
    if (!rows) return -2;
    for (size_t i = 0; i < n; i++) {
        const Dav1dHipLrTask *k = &t[i];
        const int pl = k->plane;
        const ptrdiff_t st = dst->stride[pl];
        if (src->stride[pl] != st || lpf->stride[pl] != st) { free(rows); return -3; }
        void *fn = k->type <= DAV1D_HIP_LR_WIENER5 ? entry(bpc, "wiener", k->type, 0) : entry(bpc, "sgr", k->type - DAV1D_HIP_LR_SGR_5X5, 0);
        if (!fn) { free(rows); return -1; }
        uint8_t left[64 * 4 * 2];
        memset(left, 0, sizeof(left));
        if (k->x >= 4) for (int y = 0; y < k->h; y++) memcpy(left + y * 4 * bps, (const uint8_t *) src->data[pl] + (k->y + y) * st + (k->x - 4) * bps, 4 * bps);
        const uint8_t *lp = (const uint8_t *) lpf->data[pl];
        memset(rows, 0, (size_t) 8 * st);
        if (k->y >= 2) { memcpy(rows, lp + (k->y - 2) * st, st); memcpy(rows + st, lp + (k->y - 1) * st, st); }
        if (k->y + k->h + 1 < lpf->h[pl] + 8) {          /* planes are allocated with padding rows */
            memcpy(rows + 6 * st, lp + (k->y + k->h) * st, st);
            /* backup_lpf() stores the picture's last row twice where it is the first of the two (src/lf_apply_tmpl.c:77-97) */
            memcpy(rows + 7 * st, lp + (k->y + k->h + 1 < lpf->h[pl] ? k->y + k->h + 1 : lpf->h[pl] - 1) * st, st);
        }
        union { int16_t filter[2][8]; struct { uint32_t s0, s1; int16_t w0, w1; } sgr; } prm;
        memset(&prm, 0, sizeof(prm));
        if (k->type <= DAV1D_HIP_LR_WIENER5) memcpy(prm.filter, k->filter, sizeof(prm.filter));
        else { prm.sgr.s0 = (uint16_t) k->filter[0][0]; prm.sgr.s1 = (uint16_t) k->filter[0][1]; prm.sgr.w0 = k->filter[0][2]; prm.sgr.w1 = k->filter[0][3]; }
        uint8_t *p = (uint8_t *) dst->data[pl] + k->y * st + k->x * bps;
        if (bpc == 8) ((lr8_fn) fn)(p, st, left, rows + k->x * bps, k->w, k->h, &prm, k->edges);
        else ((lr16_fn) fn)((uint16_t *) p, st, left, (const uint16_t *) (rows + k->x * bps), k->w, k->h, &prm, k->edges, bdmax);
    }
    free(rows);
    return 0;
}

/* ---- the same three lists over several host threads: the "all host cores" leg of bench.py's cpu_baseline
 * (SURVEY.md 8d).  Phases run in the order of the single-thread replay (mc, compound, itx) with a barrier in between;
 * inside a phase the threads pull chunks of consecutive tasks from a shared counter.  That is only valid when the tasks
 * of one phase write disjoint rectangles (true for the synthetic inter frames: no mask / blend records) — the way
 * dav1d's own workers run tile-sbrows of one pass concurrently (reference src/thread_task.c:733-851). */
#include <pthread.h>

typedef struct ReplayMt {
    entry_fn entry; int bpc;
    const ReplayPlanes *dst, *refs;
    const Dav1dHipMcTask *mc; size_t n_mc;
    const Dav1dHipCompTask *comp; size_t n_comp;
    const Dav1dHipItxTask *itx; size_t n_itx;
    int16_t *prep; void *coef;
    size_t next[3], chunk;
    int rc;
    int n_thr;          /* threads that really run; fixed before `go` */
    int go;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    pthread_barrier_t bar;
} ReplayMt;

static void *replay_worker(void *arg)
{
    ReplayMt *const m = arg;
    const size_t n[3] = { m->n_mc, m->n_comp, m->n_itx };
    uint8_t mask[16] = { 0 };
    pthread_mutex_lock(&m->mu);
    while (!m->go) pthread_cond_wait(&m->cv, &m->mu);
    pthread_mutex_unlock(&m->mu);
    for (int ph = 0; ph < 3; ph++) {
        for (;;) {
            const size_t lo = __atomic_fetch_add(&m->next[ph], m->chunk, __ATOMIC_RELAXED);
            if (lo >= n[ph]) break;
            const size_t cnt = n[ph] - lo < m->chunk ? n[ph] - lo : m->chunk;
            int rc = ph == 0 ? dav1d_replay_mc(m->entry, m->bpc, m->dst, m->refs, m->mc + lo, cnt, m->prep)
                   : ph == 1 ? dav1d_replay_comp(m->entry, m->bpc, m->dst, m->comp + lo, cnt, m->prep, mask)
                             : dav1d_replay_itx(m->entry, m->bpc, m->dst, m->itx + lo, cnt, m->coef);
            if (rc) __atomic_store_n(&m->rc, rc, __ATOMIC_RELAXED);
        }
        pthread_barrier_wait(&m->bar);       /* a phase reads what the previous one wrote */
    }
    return NULL;
}

int dav1d_replay_recon_mt(entry_fn entry, int bpc, const ReplayPlanes *dst, const ReplayPlanes *refs,
                          const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                          const Dav1dHipItxTask *itx, size_t n_itx, int16_t *prep, void *coef, int n_threads)
{
    enum { MAX_THR = 1024 };
    if (n_threads < 1 || n_threads > MAX_THR) return -22;
    for (size_t i = 0; i < n_comp; i++)
        if (comp[i].kind >= DAV1D_HIP_COMP_MASK) return -22;        /* mask outputs / blends are order dependent */
    if (!entry(bpc, "mc", 0, 0)) return -1;                          /* the oracle fills its tables on the first call: do that here */
    ReplayMt m;
    memset(&m, 0, sizeof(m));
    m.entry = entry; m.bpc = bpc; m.dst = dst; m.refs = refs;
    m.mc = mc; m.n_mc = n_mc; m.comp = comp; m.n_comp = n_comp; m.itx = itx; m.n_itx = n_itx;
    m.prep = prep; m.coef = coef;
    /* chunks of consecutive tasks: small enough that every thread gets some tens of them, large enough to amortise the lookup
     * of the function pointers at the head of each replay call */
    m.chunk = (n_mc + n_itx) / ((size_t) n_threads * 32) + 1;
    if (m.chunk > 256) m.chunk = 256;
    if (m.chunk < 16) m.chunk = 16;
    pthread_mutex_init(&m.mu, NULL);
    pthread_cond_init(&m.cv, NULL);
    pthread_t th[MAX_THR];
    int started = 0;
    for (; started < n_threads - 1; started++)
        if (pthread_create(&th[started], NULL, replay_worker, &m)) break;
    m.n_thr = started + 1;                /* fewer than asked for if the host refused some */
    pthread_barrier_init(&m.bar, NULL, (unsigned) m.n_thr);
    pthread_mutex_lock(&m.mu);
    m.go = 1;
    pthread_cond_broadcast(&m.cv);
    pthread_mutex_unlock(&m.mu);
    replay_worker(&m);
    for (int k = 0; k < started; k++) pthread_join(th[k], NULL);
    pthread_barrier_destroy(&m.bar);
    pthread_cond_destroy(&m.cv);
    pthread_mutex_destroy(&m.mu);
    return m.rc ? m.rc : m.n_thr;         /* > 0: the number of threads that ran */
}

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

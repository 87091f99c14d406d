/* oracle/port — function-pointer table with the reference's DSP signatures (src/itx.h:37-40,
 * src/mc.h:38-122), looked up by (bpc, family, i, j) exactly like oracle/ref_shim.c does for the
 * reference build, so tests and oracle/replay.c can use either oracle interchangeably.
 * TEST INFRASTRUCTURE ONLY. */
#include "port.h"

#define FOR_TP(X, tx) X(tx, 0) X(tx, 1) X(tx, 2) X(tx, 3) X(tx, 4) X(tx, 5) X(tx, 6) X(tx, 7) X(tx, 8) X(tx, 9) \
                      X(tx, 10) X(tx, 11) X(tx, 12) X(tx, 13) X(tx, 14) X(tx, 15) X(tx, 16)
#define FOR_TX(X) FOR_TP(X, 0) FOR_TP(X, 1) FOR_TP(X, 2) FOR_TP(X, 3) FOR_TP(X, 4) FOR_TP(X, 5) FOR_TP(X, 6) FOR_TP(X, 7) \
                  FOR_TP(X, 8) FOR_TP(X, 9) FOR_TP(X, 10) FOR_TP(X, 11) FOR_TP(X, 12) FOR_TP(X, 13) FOR_TP(X, 14) \
                  FOR_TP(X, 15) FOR_TP(X, 16) FOR_TP(X, 17) FOR_TP(X, 18)

#define ITX_W(tx, tp) \
    static void itx8_##tx##_##tp(uint8_t *d, ptrdiff_t s, int16_t *c, int e) { port_inv_txfm_add(d, s, c, e, tx, tp, 255); } \
    static void itx16_##tx##_##tp(uint16_t *d, ptrdiff_t s, int32_t *c, int e, int bm) { port_inv_txfm_add(d, s, c, e, tx, tp, bm); }
FOR_TX(ITX_W)
#define E8(tx, tp) [tx][tp] = (void *) itx8_##tx##_##tp,
#define E16(tx, tp) [tx][tp] = (void *) itx16_##tx##_##tp,
static void *const itx8_tab[19][17] = { FOR_TX(E8) };
static void *const itx16_tab[19][17] = { FOR_TX(E16) };

#define FOR_F(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9)
#define MC_W(f) \
    static void mc8_##f(uint8_t *d, ptrdiff_t ds, const uint8_t *s, ptrdiff_t ss, int w, int h, int mx, int my) { port_mc(d, ds, NULL, s, ss, w, h, mx, my, f, 255); } \
    static void mc16_##f(uint16_t *d, ptrdiff_t ds, const uint16_t *s, ptrdiff_t ss, int w, int h, int mx, int my, int bm) { port_mc(d, ds, NULL, s, ss, w, h, mx, my, f, bm); } \
    static void mct8_##f(int16_t *t, const uint8_t *s, ptrdiff_t ss, int w, int h, int mx, int my) { port_mc(NULL, 0, t, s, ss, w, h, mx, my, f, 255); } \
    static void mct16_##f(int16_t *t, const uint16_t *s, ptrdiff_t ss, int w, int h, int mx, int my, int bm) { port_mc(NULL, 0, t, s, ss, w, h, mx, my, f, bm); }
FOR_F(MC_W)
#define M8(f) (void *) mc8_##f,
#define M16(f) (void *) mc16_##f,
#define T8(f) (void *) mct8_##f,
#define T16(f) (void *) mct16_##f,
static void *const mc8_tab[10] = { FOR_F(M8) }, *const mc16_tab[10] = { FOR_F(M16) };
static void *const mct8_tab[10] = { FOR_F(T8) }, *const mct16_tab[10] = { FOR_F(T16) };

static void avg8(uint8_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h) { port_comp(0, 0, d, ds, a, b, w, h, 0, NULL, NULL, 255); }
static void avg16(uint16_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, int bm) { port_comp(0, 0, d, ds, a, b, w, h, 0, NULL, NULL, bm); }
static void wavg8(uint8_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, int wt) { port_comp(1, 0, d, ds, a, b, w, h, wt, NULL, NULL, 255); }
static void wavg16(uint16_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, int wt, int bm) { port_comp(1, 0, d, ds, a, b, w, h, wt, NULL, NULL, bm); }
static void mask8(uint8_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, const uint8_t *m) { port_comp(2, 0, d, ds, a, b, w, h, 0, m, NULL, 255); }
static void mask16(uint16_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, const uint8_t *m, int bm) { port_comp(2, 0, d, ds, a, b, w, h, 0, m, NULL, bm); }
#define WM(ss) \
    static void wmask8_##ss(uint8_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, uint8_t *m, int sign) { port_comp(3, ss, d, ds, a, b, w, h, sign, NULL, m, 255); } \
    static void wmask16_##ss(uint16_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, uint8_t *m, int sign, int bm) { port_comp(3, ss, d, ds, a, b, w, h, sign, NULL, m, bm); }
WM(0) WM(1) WM(2)

static void emu8(intptr_t bw, intptr_t bh, intptr_t iw, intptr_t ih, intptr_t x, intptr_t y, void *d, ptrdiff_t ds, const void *r, ptrdiff_t rs) { port_emu_edge(bw, bh, iw, ih, x, y, d, ds, r, rs, 0); }
static void emu16(intptr_t bw, intptr_t bh, intptr_t iw, intptr_t ih, intptr_t x, intptr_t y, void *d, ptrdiff_t ds, const void *r, ptrdiff_t rs) { port_emu_edge(bw, bh, iw, ih, x, y, d, ds, r, rs, 1); }
static void blend8(void *d, ptrdiff_t ds, const void *t, int w, int h, const uint8_t *m) { port_blend(0, d, ds, t, w, h, m, 0); }
static void blend16(void *d, ptrdiff_t ds, const void *t, int w, int h, const uint8_t *m) { port_blend(0, d, ds, t, w, h, m, 1); }
static void blendv8(void *d, ptrdiff_t ds, const void *t, int w, int h) { port_blend(1, d, ds, t, w, h, NULL, 0); }
static void blendv16(void *d, ptrdiff_t ds, const void *t, int w, int h) { port_blend(1, d, ds, t, w, h, NULL, 1); }
static void blendh8(void *d, ptrdiff_t ds, const void *t, int w, int h) { port_blend(2, d, ds, t, w, h, NULL, 0); }
static void blendh16(void *d, ptrdiff_t ds, const void *t, int w, int h) { port_blend(2, d, ds, t, w, h, NULL, 1); }

static int legal_itx(int tx, int tp) {           /* reference src/itx_tmpl.c:160-178 */
    static const uint8_t w[19] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
    static const uint8_t h[19] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };
    if (tx < 0 || tx > 18 || tp < 0 || tp > 16) return 0;
    if (tp == 16) return tx == 0;
    const int m = w[tx] > h[tx] ? w[tx] : h[tx];
    if (m == 64) return tp == 0;
    if (m == 32) return tp == 0 || tp == 9;
    if (w[tx] == 16 && h[tx] == 16) return tp <= 11;
    return 1;
}

void *dav1d_port_dsp_entry(const int bpc, const char *const family, const int i, const int j) {
    const int hbd = bpc > 8;
#define F(n) (!strcmp(family, n))
    if (F("itxfm_add")) return legal_itx(i, j) ? (hbd ? itx16_tab[i][j] : itx8_tab[i][j]) : NULL;
    if (F("mc")) return i >= 0 && i < 10 ? (hbd ? mc16_tab[i] : mc8_tab[i]) : NULL;
    if (F("mct")) return i >= 0 && i < 10 ? (hbd ? mct16_tab[i] : mct8_tab[i]) : NULL;
    if (F("avg")) return hbd ? (void *) avg16 : (void *) avg8;
    if (F("w_avg")) return hbd ? (void *) wavg16 : (void *) wavg8;
    if (F("mask")) return hbd ? (void *) mask16 : (void *) mask8;
    if (F("w_mask")) {
        if (i == 0) return hbd ? (void *) wmask16_0 : (void *) wmask8_0;
        if (i == 1) return hbd ? (void *) wmask16_1 : (void *) wmask8_1;
        if (i == 2) return hbd ? (void *) wmask16_2 : (void *) wmask8_2;
        return NULL;
    }
    if (F("emu_edge")) return hbd ? (void *) emu16 : (void *) emu8;
    if (F("blend")) return hbd ? (void *) blend16 : (void *) blend8;
    if (F("blend_v")) return hbd ? (void *) blendv16 : (void *) blendv8;
    if (F("blend_h")) return hbd ? (void *) blendh16 : (void *) blendh8;
#undef F
    return NULL;
}

/* oracle/port — function-pointer table with the reference's DSP signatures (src/itx.h:37-40,
 * src/mc.h:38-122), looked up by (bpc, family, i, j) exactly like oracle/ref_shim.c does for the
 * reference build, so tests and oracle/replay.c can use either oracle interchangeably.  Families: itx, mc, ipred,
 * loop filter, cdef, loop restoration, film grain (signatures: src/ipred.h, src/loopfilter.h, src/cdef.h,
 * src/looprestoration.h, src/filmgrain.h).
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

/* ---- warp, scaled mc, resize (src/mc.h:60-122) */
static void warp8(void *d, ptrdiff_t ds, const void *s, ptrdiff_t ss, const int16_t *abcd, int mx, int my) { port_warp8x8(d, ds, NULL, 0, s, ss, abcd, mx, my, 255); }
static void warp16(void *d, ptrdiff_t ds, const void *s, ptrdiff_t ss, const int16_t *abcd, int mx, int my, int bm) { port_warp8x8(d, ds, NULL, 0, s, ss, abcd, mx, my, bm); }
static void warpt8(int16_t *t, ptrdiff_t ts, const void *s, ptrdiff_t ss, const int16_t *abcd, int mx, int my) { port_warp8x8(NULL, 0, t, ts, s, ss, abcd, mx, my, 255); }
static void warpt16(int16_t *t, ptrdiff_t ts, const void *s, ptrdiff_t ss, const int16_t *abcd, int mx, int my, int bm) { port_warp8x8(NULL, 0, t, ts, s, ss, abcd, mx, my, bm); }
#define MCS_W(f) \
    static void mcs8_##f(void *d, ptrdiff_t ds, const void *s, ptrdiff_t ss, int w, int h, int mx, int my, int dx, int dy) { port_mc_scaled(d, ds, NULL, s, ss, w, h, mx, my, dx, dy, f, 255); } \
    static void mcs16_##f(void *d, ptrdiff_t ds, const void *s, ptrdiff_t ss, int w, int h, int mx, int my, int dx, int dy, int bm) { port_mc_scaled(d, ds, NULL, s, ss, w, h, mx, my, dx, dy, f, bm); } \
    static void mcts8_##f(int16_t *t, const void *s, ptrdiff_t ss, int w, int h, int mx, int my, int dx, int dy) { port_mc_scaled(NULL, 0, t, s, ss, w, h, mx, my, dx, dy, f, 255); } \
    static void mcts16_##f(int16_t *t, const void *s, ptrdiff_t ss, int w, int h, int mx, int my, int dx, int dy, int bm) { port_mc_scaled(NULL, 0, t, s, ss, w, h, mx, my, dx, dy, f, bm); }
FOR_F(MCS_W)
#define S8(f) (void *) mcs8_##f,
#define S16(f) (void *) mcs16_##f,
#define ST8(f) (void *) mcts8_##f,
#define ST16(f) (void *) mcts16_##f,
static void *const mcs8_tab[10] = { FOR_F(S8) }, *const mcs16_tab[10] = { FOR_F(S16) };
static void *const mcts8_tab[10] = { FOR_F(ST8) }, *const mcts16_tab[10] = { FOR_F(ST16) };
static void resize8(void *d, ptrdiff_t ds, const void *s, ptrdiff_t ss, int dw, int h, int sw, int dx, int mx0) { port_resize(d, ds, s, ss, dw, h, sw, dx, mx0, 255); }
static void resize16(void *d, ptrdiff_t ds, const void *s, ptrdiff_t ss, int dw, int h, int sw, int dx, int mx0, int bm) { port_resize(d, ds, s, ss, dw, h, sw, dx, mx0, bm); }

/* ---- loop filter: loop_filter_sb[plane != 0][dir] (src/loopfilter.h:38-53); lut = Av1FilterLUT { e[64], i[64], .. } */
#define LF_W(c, d) \
    static void lf8_##c##d(void *p, ptrdiff_t s, const uint32_t *m, const uint8_t (*l)[4], ptrdiff_t ls, const uint8_t *lut, int n) { (void) n; port_loop_filter_sb(c, d, p, s, m, l, ls, lut, 255); } \
    static void lf16_##c##d(void *p, ptrdiff_t s, const uint32_t *m, const uint8_t (*l)[4], ptrdiff_t ls, const uint8_t *lut, int n, int bm) { (void) n; port_loop_filter_sb(c, d, p, s, m, l, ls, lut, bm); }
LF_W(0, 0) LF_W(0, 1) LF_W(1, 0) LF_W(1, 1)

/* ---- cdef (src/cdef.h:44-67) */
static int cdir8(const void *p, ptrdiff_t s, unsigned *v) { return port_cdef_dir(p, s, v, 255); }
static int cdir16(const void *p, ptrdiff_t s, unsigned *v, int bm) { return port_cdef_dir(p, s, v, bm); }
#define CDEF_W(n, w, h) \
    static void cfb8_##n(void *d, ptrdiff_t s, const void *l, const void *t, const void *b, int pri, int sec, int dir, int damp, int e) { port_cdef_fb(w, h, d, s, l, t, b, pri, sec, dir, damp, e, 255); } \
    static void cfb16_##n(void *d, ptrdiff_t s, const void *l, const void *t, const void *b, int pri, int sec, int dir, int damp, int e, int bm) { port_cdef_fb(w, h, d, s, l, t, b, pri, sec, dir, damp, e, bm); }
CDEF_W(0, 8, 8) CDEF_W(1, 4, 8) CDEF_W(2, 4, 4)

/* ---- loop restoration (src/looprestoration.h:49-75): params = union { int16_t filter[2][8]; struct { u32 s0, s1; i16 w0, w1; } sgr; } */
typedef union { int16_t filter[2][8]; struct { uint32_t s0, s1; int16_t w0, w1; } sgr; } LrParams;
static void wien8(void *p, ptrdiff_t s, const void *l, const void *lpf, int w, int h, const LrParams *prm, int e) { port_wiener(p, s, l, lpf, w, h, prm->filter, e, 255); }
static void wien16(void *p, ptrdiff_t s, const void *l, const void *lpf, int w, int h, const LrParams *prm, int e, int bm) { port_wiener(p, s, l, lpf, w, h, prm->filter, e, bm); }
#define SGR_W(t) \
    static void sgr8_##t(void *p, ptrdiff_t s, const void *l, const void *lpf, int w, int h, const LrParams *prm, int e) { port_sgr(t, p, s, l, lpf, w, h, prm->sgr.s0, prm->sgr.s1, prm->sgr.w0, prm->sgr.w1, e, 255); } \
    static void sgr16_##t(void *p, ptrdiff_t s, const void *l, const void *lpf, int w, int h, const LrParams *prm, int e, int bm) { port_sgr(t, p, s, l, lpf, w, h, prm->sgr.s0, prm->sgr.s1, prm->sgr.w0, prm->sgr.w1, e, bm); }
SGR_W(0) SGR_W(1) SGR_W(2)

/* ---- intra prediction (src/ipred.h:44-90) */
#define FOR_M(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#define IP_W(m) \
    static void ip8_##m(void *d, ptrdiff_t s, const void *tl, int w, int h, int a, int mw, int mh) { port_intra_pred(m, d, s, tl, w, h, a, mw, mh, 255); } \
    static void ip16_##m(void *d, ptrdiff_t s, const void *tl, int w, int h, int a, int mw, int mh, int bm) { port_intra_pred(m, d, s, tl, w, h, a, mw, mh, bm); }
FOR_M(IP_W)
#define I8(m) (void *) ip8_##m,
#define I16(m) (void *) ip16_##m,
static void *const ip8_tab[14] = { FOR_M(I8) }, *const ip16_tab[14] = { FOR_M(I16) };
#define CFLAC_W(l) \
    static void cac8_##l(int16_t *ac, const void *y, ptrdiff_t s, int wp, int hp, int cw, int ch) { port_cfl_ac(l, ac, y, s, wp, hp, cw, ch, 0); } \
    static void cac16_##l(int16_t *ac, const void *y, ptrdiff_t s, int wp, int hp, int cw, int ch) { port_cfl_ac(l, ac, y, s, wp, hp, cw, ch, 1); }
CFLAC_W(0) CFLAC_W(1) CFLAC_W(2)
#define CFLP_W(m) \
    static void cpr8_##m(void *d, ptrdiff_t s, const void *tl, int w, int h, const int16_t *ac, int alpha) { port_cfl_pred(m, d, s, tl, w, h, ac, alpha, 255); } \
    static void cpr16_##m(void *d, ptrdiff_t s, const void *tl, int w, int h, const int16_t *ac, int alpha, int bm) { port_cfl_pred(m, d, s, tl, w, h, ac, alpha, bm); }
CFLP_W(0) CFLP_W(3) CFLP_W(4) CFLP_W(5)
static void pal8(void *d, ptrdiff_t s, const void *pal, const uint8_t *idx, int w, int h) { port_pal_pred(d, s, pal, idx, w, h, 0); }
static void pal16(void *d, ptrdiff_t s, const void *pal, const uint8_t *idx, int w, int h) { port_pal_pred(d, s, pal, idx, w, h, 1); }

/* ---- film grain (src/filmgrain.h:46-80): grain entries are int8_t at 8 bpc, int16_t above */
#include "fg_port.h"
void port_generate_grain(int grain[][82], const int luma[][82], const PortFilmGrain *d, int pl, int subx, int suby, int bitdepth_max);
void port_fg_row(int pl, void *dst_row, const void *src_row, ptrdiff_t stride, const PortFilmGrain *d, int pw, const uint8_t *scaling,
                 const int lut[][82], int bh, int row_num, const void *luma_row, ptrdiff_t luma_stride, int subx, int suby, int is_id,
                 int bitdepth_max);
#define FG_W(B, entry, BMDECL, BM) \
    static void load_##B(int (*o)[82], const entry (*in)[82]) { for (int y = 0; y < 73; y++) for (int x = 0; x < 82; x++) o[y][x] = in[y][x]; } \
    static void ggy##B(entry (*buf)[82], const PortFilmGrain *d BMDECL) { \
        int g[74][82]; memset(g, 0, sizeof(g)); port_generate_grain(g, NULL, d, 0, 0, 0, BM); \
        for (int y = 0; y < 73; y++) for (int x = 0; x < 82; x++) buf[y][x] = (entry) g[y][x]; } \
    static void gguv##B(int layout, entry (*buf)[82], const entry (*by)[82], const PortFilmGrain *d, intptr_t uv BMDECL) { \
        int g[74][82], l[74][82]; memset(g, 0, sizeof(g)); memset(l, 0, sizeof(l)); load_##B(l, by); \
        const int sx = layout < 2, sy = layout == 0; \
        port_generate_grain(g, (const int (*)[82]) l, d, 1 + (int) uv, sx, sy, BM); \
        for (int y = 0; y < (sy ? 38 : 73); y++) for (int x = 0; x < (sx ? 44 : 82); x++) buf[y][x] = (entry) g[y][x]; } \
    static void gguv##B##_0(entry (*b)[82], const entry (*by)[82], const PortFilmGrain *d, intptr_t uv BMDECL) { gguv##B(0, b, by, d, uv FG_PASS_##B); } \
    static void gguv##B##_1(entry (*b)[82], const entry (*by)[82], const PortFilmGrain *d, intptr_t uv BMDECL) { gguv##B(1, b, by, d, uv FG_PASS_##B); } \
    static void gguv##B##_2(entry (*b)[82], const entry (*by)[82], const PortFilmGrain *d, intptr_t uv BMDECL) { gguv##B(2, b, by, d, uv FG_PASS_##B); } \
    static void fgy##B(void *dst, const void *src, ptrdiff_t st, const PortFilmGrain *d, size_t pw, const uint8_t *sc, const entry (*lut)[82], \
                       int bh, int row BMDECL) { \
        int l[74][82]; memset(l, 0, sizeof(l)); load_##B(l, lut); \
        port_fg_row(0, dst, src, st, d, (int) pw, sc, (const int (*)[82]) l, bh, row, NULL, 0, 0, 0, 0, BM); } \
    static void fguv##B(int layout, void *dst, const void *src, ptrdiff_t st, const PortFilmGrain *d, size_t pw, const uint8_t *sc, \
                        const entry (*lut)[82], int bh, int row, const void *luma, ptrdiff_t ls, int uv, int is_id BMDECL) { \
        int l[74][82]; memset(l, 0, sizeof(l)); load_##B(l, lut); \
        port_fg_row(1 + uv, dst, src, st, d, (int) pw, sc, (const int (*)[82]) l, bh, row, luma, ls, layout < 2, layout == 0, is_id, BM); } \
    static void fguv##B##_0(void *dst, const void *src, ptrdiff_t st, const PortFilmGrain *d, size_t pw, const uint8_t *sc, const entry (*lut)[82], \
                            int bh, int row, const void *luma, ptrdiff_t ls, int uv, int is_id BMDECL) { fguv##B(0, dst, src, st, d, pw, sc, lut, bh, row, luma, ls, uv, is_id FG_PASS_##B); } \
    static void fguv##B##_1(void *dst, const void *src, ptrdiff_t st, const PortFilmGrain *d, size_t pw, const uint8_t *sc, const entry (*lut)[82], \
                            int bh, int row, const void *luma, ptrdiff_t ls, int uv, int is_id BMDECL) { fguv##B(1, dst, src, st, d, pw, sc, lut, bh, row, luma, ls, uv, is_id FG_PASS_##B); } \
    static void fguv##B##_2(void *dst, const void *src, ptrdiff_t st, const PortFilmGrain *d, size_t pw, const uint8_t *sc, const entry (*lut)[82], \
                            int bh, int row, const void *luma, ptrdiff_t ls, int uv, int is_id BMDECL) { fguv##B(2, dst, src, st, d, pw, sc, lut, bh, row, luma, ls, uv, is_id FG_PASS_##B); }
#define FG_PASS_8
#define FG_PASS_16 , bm
#define FG_NOBM
#define FG_BM , int bm
FG_W(8, int8_t, FG_NOBM, 255)
FG_W(16, int16_t, FG_BM, bm)

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
    if (F("mc_scaled")) return i >= 0 && i < 10 ? (hbd ? mcs16_tab[i] : mcs8_tab[i]) : NULL;
    if (F("mct_scaled")) return i >= 0 && i < 10 ? (hbd ? mcts16_tab[i] : mcts8_tab[i]) : NULL;
    if (F("warp8x8")) return hbd ? (void *) warp16 : (void *) warp8;
    if (F("warp8x8t")) return hbd ? (void *) warpt16 : (void *) warpt8;
    if (F("resize")) return hbd ? (void *) resize16 : (void *) resize8;
    if (F("loop_filter_sb")) {
        static void *const t8[2][2] = { { (void *) lf8_00, (void *) lf8_01 }, { (void *) lf8_10, (void *) lf8_11 } };
        static void *const t16[2][2] = { { (void *) lf16_00, (void *) lf16_01 }, { (void *) lf16_10, (void *) lf16_11 } };
        return (unsigned) i < 2 && (unsigned) j < 2 ? (hbd ? t16[i][j] : t8[i][j]) : NULL;
    }
    if (F("cdef_dir")) return hbd ? (void *) cdir16 : (void *) cdir8;
    if (F("cdef_fb")) {
        static void *const t8[3] = { (void *) cfb8_0, (void *) cfb8_1, (void *) cfb8_2 }, *const t16[3] = { (void *) cfb16_0, (void *) cfb16_1, (void *) cfb16_2 };
        return (unsigned) i < 3 ? (hbd ? t16[i] : t8[i]) : NULL;
    }
    if (F("wiener")) return (unsigned) i < 2 ? (hbd ? (void *) wien16 : (void *) wien8) : NULL;
    if (F("sgr")) {
        static void *const t8[3] = { (void *) sgr8_0, (void *) sgr8_1, (void *) sgr8_2 }, *const t16[3] = { (void *) sgr16_0, (void *) sgr16_1, (void *) sgr16_2 };
        return (unsigned) i < 3 ? (hbd ? t16[i] : t8[i]) : NULL;
    }
    if (F("intra_pred")) return (unsigned) i < 14 ? (hbd ? ip16_tab[i] : ip8_tab[i]) : NULL;
    if (F("cfl_ac")) {
        static void *const t8[3] = { (void *) cac8_0, (void *) cac8_1, (void *) cac8_2 }, *const t16[3] = { (void *) cac16_0, (void *) cac16_1, (void *) cac16_2 };
        return (unsigned) i < 3 ? (hbd ? t16[i] : t8[i]) : NULL;
    }
    if (F("cfl_pred")) {
        static void *const t8[6] = { (void *) cpr8_0, NULL, NULL, (void *) cpr8_3, (void *) cpr8_4, (void *) cpr8_5 };
        static void *const t16[6] = { (void *) cpr16_0, NULL, NULL, (void *) cpr16_3, (void *) cpr16_4, (void *) cpr16_5 };
        return (unsigned) i < 6 ? (hbd ? t16[i] : t8[i]) : NULL;
    }
    if (F("pal_pred")) return hbd ? (void *) pal16 : (void *) pal8;
    if (F("generate_grain_y")) return hbd ? (void *) ggy16 : (void *) ggy8;
    if (F("generate_grain_uv")) {
        static void *const t8[3] = { (void *) gguv8_0, (void *) gguv8_1, (void *) gguv8_2 }, *const t16[3] = { (void *) gguv16_0, (void *) gguv16_1, (void *) gguv16_2 };
        return (unsigned) i < 3 ? (hbd ? t16[i] : t8[i]) : NULL;
    }
    if (F("fgy_32x32xn")) return hbd ? (void *) fgy16 : (void *) fgy8;
    if (F("fguv_32x32xn")) {
        static void *const t8[3] = { (void *) fguv8_0, (void *) fguv8_1, (void *) fguv8_2 }, *const t16[3] = { (void *) fguv16_0, (void *) fguv16_1, (void *) fguv16_2 };
        return (unsigned) i < 3 ? (hbd ? t16[i] : t8[i]) : NULL;
    }
#undef F
    return NULL;
}

/* oracle/port — CPU restatement of the inverse transforms.  TEST INFRASTRUCTURE ONLY:
 * nothing in dav1d_amd/ includes, links or loads this; tests/ and bench.py's cpu_baseline may.
 *
 * Follows the reference's inv_txfm_add_c (src/itx_tmpl.c:43-124), its WHT variant
 * (:184-203) and the 1-D kernels of src/itx_1d.c:66-1017, but the flow graphs are evaluated
 * by run-time loops over the AV1 recursive structure (spec 7.13.2: even/odd split,
 * bit-reversed first-stage rotations, alternating add/sub and rotation stages) instead of the
 * reference's hand-unrolled code.  Value rules that make it bit-identical: every rotation is
 * Round2(ka*a + kb*b, 12) computed exactly in 64 bits; only add/sub outputs are clamped. */
#include "port.h"

static const int cos128[65] = {
    4096, 4095, 4091, 4085, 4076, 4065, 4052, 4036, 4017, 3996, 3973, 3948, 3920,
    3889, 3857, 3822, 3784, 3745, 3703, 3659, 3612, 3564, 3513, 3461, 3406, 3349,
    3290, 3229, 3166, 3102, 3035, 2967, 2896, 2824, 2751, 2675, 2598, 2520, 2440,
    2359, 2276, 2191, 2106, 2019, 1931, 1842, 1751, 1660, 1567, 1474, 1380, 1285,
    1189, 1092,  995,  897,  799,  700,  601,  501,  401,  301,  201,  101,    0,
};

static int ilog2i(int v) { int r = 0; while (v > 1) { v >>= 1; r++; } return r; }
static int brev(int v, int bits) { int r = 0; for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i); return r; }
static int rot(int a, int b, int ka, int kb) { return (int) (((int64_t) a * ka + (int64_t) b * kb + 2048) >> 12); }
static int rot45(int v) { return (int) (((int64_t) v * 181 + 128) >> 8); }   /* 2896/4096 == 181/256 */

/* inverse DCT-II of size n (4..64), reference src/itx_1d.c:66-780 */
static void idct(const int *in, int *out, int n, int lo, int hi) {
    if (n == 2) {
        out[0] = rot45(in[0] + in[1]);
        out[1] = rot45(in[0] - in[1]);
        return;
    }
    const int m = n / 2;
    int ev[32], e[32], t[32];
    for (int i = 0; i < m; i++) ev[i] = in[2 * i];
    idct(ev, e, m, lo, hi);
    const int bits_a = ilog2i(m / 2 > 0 ? m / 2 : 1);
    for (int i = 0; i < m / 2; i++) {
        const int k = 4 * brev(i, bits_a) + 1;
        const int a = k * 64 / n;
        const int c = cos128[a], s = cos128[64 - a];
        const int x = in[k], y = in[n - k];
        t[i] = rot(x, y, s, -c);
        t[m - 1 - i] = rot(x, y, c, s);
    }
    for (int g = 2; g <= m / 2; g *= 2) {
        for (int G = 0; G < m / g; G++)
            for (int i = 0; i < g / 2; i++) {
                const int p = G * g + i, q = G * g + g - 1 - i;
                const int u = t[p], v = t[q];
                if (G & 1) { t[p] = port_iclip(v - u, lo, hi); t[q] = port_iclip(v + u, lo, hi); }
                else       { t[p] = port_iclip(u + v, lo, hi); t[q] = port_iclip(u - v, lo, hi); }
            }
        const int nblk = m / (4 * g) > 0 ? m / (4 * g) : 1;
        const int bits_r = ilog2i(nblk);
        for (int j = 0; j < m / 2; j++) {
            const int blk = j / (2 * g), o = j % (2 * g);
            if (o < g / 2 || o >= 3 * g / 2) continue;
            const int mm = m - 1 - j;
            const int a = (4 * brev(blk, bits_r) + 1) * 64 * g / m;
            const int u = t[j], v = t[mm];
            if (a == 32) {
                t[j] = rot45(v - u);
                t[mm] = rot45(v + u);
            } else {
                const int c = cos128[a], s = cos128[64 - a];
                if (o < g) { t[j] = rot(v, u, s, -c);  t[mm] = rot(v, u, c, s); }
                else       { t[j] = rot(v, u, -c, -s); t[mm] = rot(v, u, s, -c); }
            }
        }
    }
    for (int i = 0; i < m; i++) {
        const int u = e[i], v = t[m - 1 - i];
        out[i] = port_iclip(u + v, lo, hi);
        out[n - 1 - i] = port_iclip(u - v, lo, hi);
    }
}

/* inverse ADST4, reference src/itx_1d.c:782-802 (sinpi constants 1321 2482 3344 3803) */
static void iadst4(const int *in, int *out) {
    const int64_t a = in[0], b = in[1], c = in[2], d = in[3];
    out[0] = (int) ((1321 * a + 3344 * b + 3803 * c + 2482 * d + 2048) >> 12);
    out[1] = (int) ((2482 * a + 3344 * b - 1321 * c - 3803 * d + 2048) >> 12);
    out[2] = (int) ((209 * (a - c + d) + 128) >> 8);
    out[3] = (int) ((3803 * a - 3344 * b + 2482 * c - 1321 * d + 2048) >> 12);
}

/* inverse ADST8 / ADST16, reference src/itx_1d.c:804-955 */
static void iadst(const int *in, int *out, int n, int lo, int hi) {
    int t[16];
    for (int i = 0; i < n / 2; i++) {
        const int a = (4 * i + 1) * 32 / n;
        const int c = cos128[a], s = cos128[64 - a];
        const int x = in[n - 1 - 2 * i], y = in[2 * i];
        t[2 * i]     = rot(x, y, c, s);
        t[2 * i + 1] = rot(x, y, s, -c);
    }
    for (int d = n / 2; d >= 4; d /= 2) {
        for (int b = 0; b < n; b += 2 * d)
            for (int i = 0; i < d; i++) {
                const int u = t[b + i], v = t[b + i + d];
                t[b + i] = port_iclip(u + v, lo, hi);
                t[b + i + d] = port_iclip(u - v, lo, hi);
            }
        for (int b = d; b < n; b += 2 * d)
            for (int p = 0; p < d / 2; p++) {
                const int u = t[b + 2 * p], v = t[b + 2 * p + 1];
                const int half = d / 4, pp = p % half;
                const int ang = (4 * pp + 1) * 64 / d;
                const int c = cos128[ang], s = cos128[64 - ang];
                if (p < half) { t[b + 2 * p] = rot(u, v, c, s);  t[b + 2 * p + 1] = rot(u, v, s, -c); }
                else          { t[b + 2 * p] = rot(v, u, c, -s); t[b + 2 * p + 1] = rot(v, u, s, c); }
            }
    }
    for (int g = 0; g < n / 4; g++) {
        const int b = 4 * g;
        const int s0 = port_iclip(t[b] + t[b + 2], lo, hi), s1 = port_iclip(t[b + 1] + t[b + 3], lo, hi);
        const int d0 = port_iclip(t[b] - t[b + 2], lo, hi), d1 = port_iclip(t[b + 1] - t[b + 3], lo, hi);
        const int slot = n == 8 ? (g == 0 ? 0 : 1) : (g == 0 ? 0 : g == 1 ? 3 : g == 2 ? 1 : 2);
        const int r0 = rot45(d0 + d1), r1 = rot45(d0 - d1);
        if (!(slot & 1)) {
            out[slot] = s0;              out[n - 1 - slot] = -s1;
            out[n / 2 - 1 - slot] = -r0; out[n / 2 + slot] = r1;
        } else {
            out[slot] = -s0;             out[n - 1 - slot] = s1;
            out[n / 2 - 1 - slot] = r0;  out[n / 2 + slot] = -r1;
        }
    }
}

/* identity transforms, reference src/itx_1d.c:976-1017 */
static void iidentity(const int *in, int *out, int n) {
    for (int i = 0; i < n; i++) {
        const int v = in[i];
        if (n == 4)       out[i] = v + ((v * 1697 + 2048) >> 12);
        else if (n == 8)  out[i] = 2 * v;
        else if (n == 16) out[i] = 2 * v + ((v * 1697 + 1024) >> 11);
        else              out[i] = 4 * v;
    }
}

/* WHT4, reference src/itx_1d.c:1066-1082 */
static void iwht4(const int *in, int *out) {
    const int s = in[0] + in[1], d = in[2] - in[3];
    const int m = (s - d) >> 1;
    const int p = m - in[3], q = m - in[1];
    out[0] = s - p; out[1] = p; out[2] = q; out[3] = d + q;
}

enum { K_DCT, K_ADST, K_IDENTITY, K_FLIPADST };

static void tx1d(int kind, const int *in, int *out, int n, int lo, int hi) {
    int tmp[64];
    switch (kind) {
    case K_DCT: idct(in, out, n, lo, hi); break;
    case K_IDENTITY: iidentity(in, out, n); break;
    default:
        if (n == 4) iadst4(in, tmp); else iadst(in, tmp, n, lo, hi);
        for (int i = 0; i < n; i++) out[i] = kind == K_FLIPADST ? tmp[n - 1 - i] : tmp[i];
    }
}

static const uint8_t tx_w[19] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
static const uint8_t tx_h[19] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };
static const uint8_t tx_shift[19] = { 0, 1, 2, 2, 2, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2 };   /* src/itx_tmpl.c:160-178 */
/* itxfm_add table index -> (first = horizontal, second = vertical) 1-D kind; the table entry [A_B]
 * runs B horizontally and A vertically (src/itx_tmpl.c:233-262 + src/itx_1d.c:1043-1060) */
static const uint8_t kinds[16][2] = {
    { K_DCT, K_DCT }, { K_DCT, K_ADST }, { K_ADST, K_DCT }, { K_ADST, K_ADST }, { K_DCT, K_FLIPADST }, { K_FLIPADST, K_DCT },
    { K_FLIPADST, K_FLIPADST }, { K_FLIPADST, K_ADST }, { K_ADST, K_FLIPADST }, { K_IDENTITY, K_IDENTITY },
    { K_IDENTITY, K_DCT }, { K_DCT, K_IDENTITY }, { K_IDENTITY, K_ADST }, { K_ADST, K_IDENTITY },
    { K_IDENTITY, K_FLIPADST }, { K_FLIPADST, K_IDENTITY },
};

/* generic over bit depth: pixels / coefs are accessed through the hbd flag */
void port_inv_txfm_add(void *dst, ptrdiff_t stride, void *coeff, int eob, int tx, int txtp, int bitdepth_max)
{
    const int hbd = bitdepth_max > 255;
    const int w = tx_w[tx], h = tx_h[tx], shift = tx_shift[tx];
    const int sw = w < 32 ? w : 32, sh = h < 32 ? h : 32;
    const int rect2 = w * 2 == h || h * 2 == w;
    const int rnd = (1 << shift) >> 1;
    uint8_t *d8 = dst;
#define CF(i) (hbd ? ((int32_t *) coeff)[i] : ((int16_t *) coeff)[i])
#define CF_ZERO(i) do { if (hbd) ((int32_t *) coeff)[i] = 0; else ((int16_t *) coeff)[i] = 0; } while (0)
#define PX_ADD(x, y, v) do { \
        if (hbd) { uint16_t *p = (uint16_t *) (d8 + (y) * stride) + (x); *p = (uint16_t) port_iclip(*p + (v), 0, bitdepth_max); } \
        else { uint8_t *p = d8 + (y) * stride + (x); *p = (uint8_t) port_iclip(*p + (v), 0, 255); } } while (0)
    int tmp[64 * 64];
    if (txtp == 16) {                              /* WHT_WHT 4x4, src/itx_tmpl.c:184-203 */
        int in[4], out[4];
        for (int y = 0; y < 4; y++) {
            for (int x = 0; x < 4; x++) in[x] = CF(y + x * 4) >> 2;
            iwht4(in, out);
            for (int x = 0; x < 4; x++) tmp[y * 4 + x] = out[x];
        }
        for (int i = 0; i < 16; i++) CF_ZERO(i);
        for (int x = 0; x < 4; x++) {
            for (int y = 0; y < 4; y++) in[y] = tmp[y * 4 + x];
            iwht4(in, out);
            for (int y = 0; y < 4; y++) PX_ADD(x, y, out[y]);
        }
        return;
    }
    if (txtp == 0 && eob < 1) {                    /* dc-only, src/itx_tmpl.c:58-70 */
        int dc = CF(0);
        CF_ZERO(0);
        if (rect2) dc = (dc * 181 + 128) >> 8;
        dc = (dc * 181 + 128) >> 8;
        dc = (dc + rnd) >> shift;
        dc = (dc * 181 + 128 + 2048) >> 12;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) PX_ADD(x, y, dc);
        return;
    }
    const int row_min = hbd ? (int) ((unsigned) ~bitdepth_max << 7) : -32768, row_max = ~row_min;
    const int col_min = hbd ? (int) ((unsigned) ~bitdepth_max << 5) : -32768, col_max = ~col_min;
    /* all sh rows are transformed: rows past the last non-zero one are zero in, zero out, which is what the
     * reference's last_nonzero_col shortcut + memset produces (src/itx_tmpl.c:88-106) */
    for (int y = 0; y < sh; y++) {
        int in[64], out[64];
        for (int x = 0; x < w; x++) {
            int v = x < sw ? CF(y + x * sh) : 0;
            if (rect2) v = (v * 181 + 128) >> 8;
            in[x] = v;
        }
        tx1d(kinds[txtp][0], in, out, w, row_min, row_max);
        for (int x = 0; x < w; x++) tmp[y * w + x] = port_iclip((out[x] + rnd) >> shift, col_min, col_max);
    }
    for (int i = 0; i < sw * sh; i++) CF_ZERO(i);
    for (int x = 0; x < w; x++) {
        int in[64], out[64];
        for (int y = 0; y < h; y++) in[y] = y < sh ? tmp[y * w + x] : 0;
        tx1d(kinds[txtp][1], in, out, h, col_min, col_max);
        for (int y = 0; y < h; y++) PX_ADD(x, y, (out[y] + 8) >> 4);
    }
#undef CF
#undef CF_ZERO
#undef PX_ADD
}

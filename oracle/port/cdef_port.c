/* oracle/port — CPU restatement of CDEF.  TEST INFRASTRUCTURE ONLY.
 *
 * cdef.dir = cdef_find_dir_c (reference src/cdef_tmpl.c:239-319): direction of an 8x8 block and its contrast.
 * cdef.fb[*] = cdef_filter_block_c (:103-237): the constrained directional filter over an 8x8 / 4x8 / 4x4 block whose
 * 2-sample frame comes from `left`, `top`, `bottom` (and the block's own right neighbours) where `edges` says so. */
#include "port.h"
#include <limits.h>

static int rd(const void *p, ptrdiff_t i, int hbd) { return hbd ? ((const uint16_t *) p)[i] : ((const uint8_t *) p)[i]; }

int port_cdef_dir(const void *const img, const ptrdiff_t stride, unsigned *const var, const int bitdepth_max)
{
    const int hbd = bitdepth_max > 255;
    int bd = 0;
    while (bitdepth_max >> bd) bd++;
    const ptrdiff_t sp = hbd ? stride / 2 : stride;
    /* line sums along the 8 directions: two diagonals (15 lines), horizontal / vertical (8), four "alt" slopes (11) */
    int diag[2][15] = { { 0 } }, hv[2][8] = { { 0 } }, alt[4][11] = { { 0 } };
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            const int px = (rd(img, y * sp + x, hbd) >> (bd - 8)) - 128;
            diag[0][y + x] += px;
            alt[0][y + (x >> 1)] += px;
            hv[0][y] += px;
            alt[1][3 + y - (x >> 1)] += px;
            diag[1][7 + y - x] += px;
            alt[2][3 - (y >> 1) + x] += px;
            hv[1][x] += px;
            alt[3][(y >> 1) + x] += px;
        }
    /* cost of a direction = sum of squared line sums, each weighted by 840 / (number of pixels on the line) */
    static const int w[8] = { 0, 840, 420, 280, 210, 168, 140, 120 };      /* index = pixels on the line */
    unsigned cost[8] = { 0 };
    for (int d = 0; d < 2; d++) {
        for (int i = 0; i < 15; i++) {
            const int len = i < 8 ? i + 1 : 15 - i;
            cost[d * 4] += (unsigned) (diag[d][i] * diag[d][i]) * (len == 8 ? 105 : w[len]);
        }
        for (int i = 0; i < 8; i++) cost[2 + d * 4] += (unsigned) (hv[d][i] * hv[d][i]) * 105;
    }
    for (int a = 0; a < 4; a++)
        for (int i = 0; i < 11; i++) {
            const int len = i < 3 ? 2 * (i + 1) : i > 7 ? 2 * (11 - i) : 8;
            cost[1 + 2 * a] += (unsigned) (alt[a][i] * alt[a][i]) * (len == 8 ? 105 : w[len]);
        }
    int best = 0;
    for (int d = 1; d < 8; d++) if (cost[d] > cost[best]) best = d;
    *var = (cost[best] - cost[best ^ 4]) >> 10;
    return best;
}

/* constrain(), src/cdef_tmpl.c:37-42 */
static int constrain(const int diff, const int threshold, const int shift)
{
    const int ad = diff < 0 ? -diff : diff;
    const int lim = port_imax(0, threshold - (ad >> shift));
    const int m = port_imin(ad, lim);
    return diff < 0 ? -m : m;
}
static int ulog2(unsigned v) { int r = 0; while (v >>= 1) r++; return r; }

void port_cdef_fb(const int w, const int h, void *const dst, const ptrdiff_t stride, const void *const left, const void *const top,
                  const void *const bottom, const int pri, const int sec, const int dir, const int damping, const int edges,
                  const int bitdepth_max)
{
    enum { HL = 1, HR = 2, HT = 4, HB = 8, NONE = INT_MIN };
    const int hbd = bitdepth_max > 255;
    int bd = 0;
    while (bitdepth_max >> bd) bd++;
    const ptrdiff_t sp = hbd ? stride / 2 : stride;
    /* the block with a 2-sample frame; samples outside the picture are marked absent */
    int win[12][12];
    for (int y = -2; y < h + 2; y++)
        for (int x = -2; x < w + 2; x++) {
            int v = NONE;
            const int inx = (x >= 0 || (edges & HL)) && (x < w || (edges & HR));
            if (y < 0) { if ((edges & HT) && inx) v = rd(top, (y + 2) * sp + x, hbd); }
            else if (y >= h) { if ((edges & HB) && inx) v = rd(bottom, (y - h) * sp + x, hbd); }
            else if (x < 0) { if (edges & HL) v = rd(left, y * 2 + (x + 2), hbd); }
            else if (inx) v = rd(dst, y * sp + x, hbd);
            win[y + 2][x + 2] = v;
        }
    /* tap offsets (dy, dx) at distance 1 and 2 along direction d = av1_cdef_directions on a 12-wide grid (:116-145) */
    const int bd8 = bd - 8;
    const int pri_tap0 = 4 - ((pri >> bd8) & 1);          /* primary taps {4, 2} or {3, 3} */
    const int pri_shift = pri ? port_imax(0, damping - ulog2(pri)) : 0;
    const int sec_shift = sec ? damping - ulog2(sec) : 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int px = win[y + 2][x + 2];
            int sum = 0, mx = px, mn = px;
            for (int k = 0; k < 2; k++) {
                if (pri) {
                    const int off = av1_cdef_directions[(dir + 2) * 2 + k];       /* offset on a 12-wide grid */
                    const int tap = k ? (pri_tap0 == 4 ? 2 : 3) : pri_tap0;
                    for (int sgn = -1; sgn <= 1; sgn += 2) {
                        const int o = sgn * off, idx = (y + 2) * 12 + (x + 2) + o;
                        const int p = win[idx / 12][idx % 12];
                        if (p == NONE) continue;
                        sum += tap * constrain(p - px, pri, pri_shift);
                        mx = port_imax(mx, p); mn = port_imin(mn, p);
                    }
                }
                if (sec) {
                    const int tap = 2 - k;
                    for (int s2 = 0; s2 < 2; s2++) {
                        const int off = av1_cdef_directions[(dir + (s2 ? 0 : 4)) * 2 + k];
                        for (int sgn = -1; sgn <= 1; sgn += 2) {
                            const int idx = (y + 2) * 12 + (x + 2) + sgn * off;
                            const int p = win[idx / 12][idx % 12];
                            if (p == NONE) continue;
                            sum += tap * constrain(p - px, sec, sec_shift);
                            mx = port_imax(mx, p); mn = port_imin(mn, p);
                        }
                    }
                }
            }
            int v = px + ((sum - (sum < 0) + 8) >> 4);
            /* the clamp to the local range only exists in the combined primary + secondary path (:147-181) */
            if (pri && sec) v = port_iclip(v, mn, mx);
            if (hbd) ((uint16_t *) dst)[y * sp + x] = (uint16_t) v; else ((uint8_t *) dst)[y * sp + x] = (uint8_t) v;
        }
}

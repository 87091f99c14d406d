/* oracle/port — CPU restatement of the deblocking filter.  TEST INFRASTRUCTURE ONLY.
 *
 * One call = dsp->lf.loop_filter_sb[plane != 0][dir] (reference src/loopfilter_tmpl.c:163-243): a line of up to 32
 * edge units of 4 samples; per unit the level picks E / I / H, the masks pick the filter width, and the edge
 * filter itself is loop_filter() (:37-161).  Written over an explicit sample vector s[-8..7] across the edge; the
 * wide filters use the AV1 specification's windowed form (7.14.6.4) instead of the reference's unrolled sums. */
#include "port.h"

static int absdiff(int a, int b) { return a > b ? a - b : b - a; }

/* one line across the edge: s[0] is q0, s[-1] is p0; returns with s[] holding the filtered samples */
static void filter_line(int *const s, const int wd, const int E, const int I, const int H, const int bd8, const int px_max)
{
    const int F = 1 << bd8;                 /* flatness threshold */
#define P(i) s[-1 - (i)]
#define Q(i) s[(i)]
    /* filter mask, src/loopfilter_tmpl.c:60-78 */
    int on = absdiff(P(1), P(0)) <= I && absdiff(Q(1), Q(0)) <= I && absdiff(P(0), Q(0)) * 2 + (absdiff(P(1), Q(1)) >> 1) <= E;
    if (wd > 4) on = on && absdiff(P(2), P(1)) <= I && absdiff(Q(2), Q(1)) <= I;
    if (wd > 6) on = on && absdiff(P(3), P(2)) <= I && absdiff(Q(3), Q(2)) <= I;
    if (!on) return;
    int flat_in = 0, flat_out = 0;
    if (wd >= 6) {                          /* :86-91 */
        flat_in = absdiff(P(2), P(0)) <= F && absdiff(P(1), P(0)) <= F && absdiff(Q(1), Q(0)) <= F && absdiff(Q(2), Q(0)) <= F;
        if (wd >= 8) flat_in = flat_in && absdiff(P(3), P(0)) <= F && absdiff(Q(3), Q(0)) <= F;
    }
    if (wd >= 16)                           /* :80-84 */
        flat_out = absdiff(P(6), P(0)) <= F && absdiff(P(5), P(0)) <= F && absdiff(P(4), P(0)) <= F &&
                   absdiff(Q(4), Q(0)) <= F && absdiff(Q(5), Q(0)) <= F && absdiff(Q(6), Q(0)) <= F;
    int in[16];
    for (int i = 0; i < 16; i++) in[i] = s[i - 8];
#define T(i) in[(i) + 8]                    /* tap i of the unfiltered line, clamped to the 7 samples each side that exist */
    /* The three smoothing filters (:93-134) are one formula: output i = -n .. n-1 sums the taps i-n .. i+n of the
     * unfiltered line, positions clamped to the n+1 samples that exist on each side, the middle 1 + 2*n2 taps counted
     * twice; (n, n2, shift) = (6, 1, 4) for the 16-wide, (3, 0, 3) for the 8-wide, (2, 1, 3) for the 6-wide filter. */
    int n = 0, n2 = 0, sh = 0;
    if (wd >= 16 && flat_out && flat_in) { n = 6; n2 = 1; sh = 4; }
    else if (wd >= 8 && flat_in) { n = 3; n2 = 0; sh = 3; }
    else if (wd == 6 && flat_in) { n = 2; n2 = 1; sh = 3; }
    if (n) {
        for (int i = -n; i < n; i++) {
            int acc = (1 << sh) >> 1;
            for (int j = -n; j <= n; j++) acc += T(port_iclip(i + j, -(n + 1), n)) * ((j < 0 ? -j : j) <= n2 ? 2 : 1);
            s[i] = acc >> sh;
        }
    } else {
        /* narrow filter, :135-158 */
        const int hev = absdiff(P(1), P(0)) > H || absdiff(Q(1), Q(0)) > H;
        const int lo = -128 * (1 << bd8), hi = 128 * (1 << bd8) - 1;
        int f = hev ? port_iclip(P(1) - Q(1), lo, hi) : 0;
        f = port_iclip(3 * (Q(0) - P(0)) + f, lo, hi);
        const int f1 = port_imin(f + 4, hi) >> 3, f2 = port_imin(f + 3, hi) >> 3;
        const int p0 = P(0), q0 = Q(0), p1 = P(1), q1 = Q(1);
        s[-1] = port_iclip(p0 + f2, 0, px_max);
        s[0] = port_iclip(q0 - f1, 0, px_max);
        if (!hev) {
            const int g = (f1 + 1) >> 1;
            s[-2] = port_iclip(p1 + g, 0, px_max);
            s[1] = port_iclip(q1 - g, 0, px_max);
        }
    }
#undef T
#undef P
#undef Q
}

void port_loop_filter_sb(const int chroma, const int dir, void *const dst, const ptrdiff_t stride, const uint32_t *const vmask,
                         const uint8_t (*l)[4], const ptrdiff_t b4_stride, const uint8_t *const lut, const int bitdepth_max)
{
    const int hbd = bitdepth_max > 255;
    int bd = 0;
    while (bitdepth_max >> bd) bd++;
    const int bd8 = bd - 8;
    const ptrdiff_t sp = hbd ? stride / 2 : stride;
    const ptrdiff_t along = dir ? 1 : sp, across = dir ? sp : 1;      /* dir 0: units run down, samples across x */
    const uint32_t vm = vmask[0] | vmask[1] | (chroma ? 0 : vmask[2]);
    for (int u = 0; u < 32; u++) {
        if (!((vm >> u) & 1)) continue;
        const uint8_t (*lu)[4] = dir ? l + u : l + u * b4_stride;
        int L = lu[0][0];
        if (!L) L = dir ? lu[-b4_stride][0] : lu[-1][0];
        if (!L) continue;
        const int H = (L >> 4) << bd8, E = lut[L] << bd8, I = lut[64 + L] << bd8;
        int wd;
        if (chroma) wd = ((vmask[1] >> u) & 1) ? 6 : 4;
        else wd = ((vmask[2] >> u) & 1) ? 16 : ((vmask[1] >> u) & 1) ? 8 : 4;
        const int reach = wd == 16 ? 7 : wd == 8 ? 4 : wd == 6 ? 3 : 2;
        for (int i = 0; i < 4; i++) {
            const ptrdiff_t base = (4 * u + i) * along;
            int buf[16] = { 0 };
            int *const s = buf + 8;
            for (int k = -reach; k < reach; k++)
                s[k] = hbd ? ((const uint16_t *) dst)[base + k * across] : ((const uint8_t *) dst)[base + k * across];
            filter_line(s, wd, E, I, H, bd8, bitdepth_max);
            for (int k = -reach; k < reach; k++) {        /* untouched samples are written back unchanged */
                if (hbd) ((uint16_t *) dst)[base + k * across] = (uint16_t) s[k];
                else ((uint8_t *) dst)[base + k * across] = (uint8_t) s[k];
            }
        }
    }
}

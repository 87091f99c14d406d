// 1-D inverse transforms of AV1, evaluated per lane on register arrays.
//
// Arithmetic contract = dav1d's C path, src/itx_1d.c:66-1017 (DCT4..64, ADST4..16,
// identity4..32) and :1066 (WHT4).  The reference spells every butterfly out by hand;
// here the same flow graph is *generated* at compile time from its recursive structure
// (AV1 spec 7.13.2: even/odd split, bit-reversed first-stage rotations, alternating
// clamped Hadamard stages and mid-block rotations).  Two value-level rules make the
// result bit-identical to the reference:
//   * every rotation output is Round2(ka*a + kb*b, 12) evaluated exactly (the
//     reference's "(k - 4096)" forms, itx_1d.c:36-63, are exact re-associations of it);
//   * only Hadamard (add/sub) outputs are clamped to [lo, hi]; rotations are not.
// All loops below have compile-time trip counts and are fully unrolled, so the arrays
// live in VGPRs and every constant becomes an immediate.
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"

namespace itx1d {

// Cos128 of the AV1 specification: round(4096 * cos(i * pi / 128)), i = 0..64.
__device__ static constexpr int kCos128[65] = {
    4096, 4095, 4091, 4085, 4076, 4065, 4052, 4036, 4017, 3996, 3973, 3948, 3920,
    3889, 3857, 3822, 3784, 3745, 3703, 3659, 3612, 3564, 3513, 3461, 3406, 3349,
    3290, 3229, 3166, 3102, 3035, 2967, 2896, 2824, 2751, 2675, 2598, 2520, 2440,
    2359, 2276, 2191, 2106, 2019, 1931, 1842, 1751, 1660, 1567, 1474, 1380, 1285,
    1189, 1092,  995,  897,  799,  700,  601,  501,  401,  301,  201,  101,    0,
};

__device__ constexpr int ilog2c(int v) { int r = 0; while (v > 1) { v >>= 1; r++; } return r; }
__device__ constexpr int brev(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return dv::clamp3(v, lo, hi); }

// Round2(ka*a + kb*b, 12) for |ka|,|kb| <= 4096 without leaving 32 bits when
// |a|,|b| < 2^19: multipliers above 2048 in magnitude are folded by +-4096 and the
// folded part is re-added outside the shift (exact, since 4096*x is a multiple of 2^12).
__device__ __forceinline__ int rot(int a, int b, int ka, int kb) {
    const int qa = ka > 2048 ? 1 : ka < -2048 ? -1 : 0;
    const int qb = kb > 2048 ? 1 : kb < -2048 ? -1 : 0;
    const int fa = ka - 4096 * qa, fb = kb - 4096 * qb;
    // the operands are far inside 24 bits (coefficients are clamped to bitdepth + 8 bits on the way in, Hadamard outputs to the same
    // range, a rotation adds one bit): the full-rate 24-bit multiply-add gives the low 32 bits of the same products; a 32-bit
    // v_mul_lo_u32 takes four times as long, and these multiplies are most of what a transform is
    int acc = 2048;
    if (fb) acc = dv::mad_i24k(b, fb, acc);
    if (fa) acc = dv::mad_i24k(a, fa, acc);
    return (acc >> 12) + qa * a + qb * b;
}
// Round2(v * 2896, 12) == (v * 181 + 128) >> 8 (2896 = 181 * 16)
__device__ __forceinline__ int rot45(int v) { return dv::mad_i24k(v, 181, 128) >> 8; }

// ---------------------------------------------------------------- inverse DCT-II
template <int N>
__device__ __forceinline__ void idct(const int *in, int *out, const int lo, const int hi) {
    if constexpr (N == 2) {
        out[0] = rot45(in[0] + in[1]);
        out[1] = rot45(in[0] - in[1]);
    } else {
        constexpr int M = N / 2;
        int ev[M], e[M], t[M];
#pragma unroll
        for (int i = 0; i < M; i++) ev[i] = in[2 * i];
        idct<M>(ev, e, lo, hi);

        // first stage: rotations of (in[k], in[N-k]), k = 1, 5, 9, ... in bit-reversed order
        constexpr int bitsA = ilog2c(M / 2);
#pragma unroll
        for (int i = 0; i < M / 2; i++) {
            const int k = 4 * brev(i, bitsA) + 1;
            const int a = k * 64 / N;
            const int c = kCos128[a], s = kCos128[64 - a];
            const int x = in[k], y = in[N - k];
            t[i] = rot(x, y, s, -c);
            t[M - 1 - i] = rot(x, y, c, s);
        }
        // alternating clamped Hadamard stages over groups of g and rotations of the
        // middle g elements of every 2g-block against their mirror images
#pragma unroll
        for (int g = 2; g <= M / 2; g *= 2) {
#pragma unroll
            for (int G = 0; G < M / g; G++) {
#pragma unroll
                for (int i = 0; i < g / 2; i++) {
                    const int p = G * g + i, q = G * g + g - 1 - i;
                    const int u = t[p], v = t[q];
                    if (G & 1) { t[p] = clampi(v - u, lo, hi); t[q] = clampi(v + u, lo, hi); }
                    else       { t[p] = clampi(u + v, lo, hi); t[q] = clampi(u - v, lo, hi); }
                }
            }
            const int nblk = M / (4 * g) > 0 ? M / (4 * g) : 1;
            const int bitsR = ilog2c(nblk);
#pragma unroll
            for (int j = 0; j < M / 2; j++) {
                const int blk = j / (2 * g), o = j % (2 * g);
                if (o < g / 2 || o >= 3 * g / 2) continue;
                const int m = M - 1 - j;
                const int a = (4 * brev(blk, bitsR) + 1) * 64 * g / M;
                const int u = t[j], v = t[m];
                if (a == 32) {
                    t[j] = rot45(v - u);
                    t[m] = rot45(v + u);
                } else {
                    const int c = kCos128[a], s = kCos128[64 - a];
                    if (o < g) { t[j] = rot(v, u, s, -c);  t[m] = rot(v, u, c, s); }
                    else       { t[j] = rot(v, u, -c, -s); t[m] = rot(v, u, s, -c); }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < M; i++) {
            const int u = e[i], v = t[M - 1 - i];
            out[i] = clampi(u + v, lo, hi);
            out[N - 1 - i] = clampi(u - v, lo, hi);
        }
    }
}

// ---------------------------------------------------------------- inverse ADST
// N = 4: the sinpi form (itx_1d.c:782-802); no clamps.
__device__ __forceinline__ void iadst4(const int *in, int *out) {
    const int a = in[0], b = in[1], c = in[2], d = in[3];
    // Round2(1321 a + 3344 b + 3803 c + 2482 d, 12) etc.; multipliers above 2048 are
    // folded by 4096 as in rot() so that four 19-bit operands stay inside 32 bits
    using dv::mad_i24k;
    out[0] = (mad_i24k(a, 1321, mad_i24k(b, 3344 - 4096, mad_i24k(c, 3803 - 4096, mad_i24k(d, 2482 - 4096, 2048)))) >> 12)
             + b + c + d;
    out[1] = (mad_i24k(a, 2482 - 4096, mad_i24k(b, 3344 - 4096, mad_i24k(c, -1321, mad_i24k(d, -(3803 - 4096), 2048)))) >> 12)
             + a + b - d;
    out[2] = mad_i24k(a - c + d, 209, 128) >> 8;
    out[3] = (mad_i24k(a, 3803 - 4096, mad_i24k(b, -(3344 - 4096), mad_i24k(c, 2482 - 4096, mad_i24k(d, -1321, 2048)))) >> 12)
             + a - b + c;
}

// N = 8, 16 (itx_1d.c:804-955): first-stage rotations by (4i+1)pi/(4N) of the pairs
// (in[N-1-2i], in[2i]); then log2(N)-1 rounds of {clamped add/sub at distance d,
// rotation of the upper half of every 2d-group}, and a final pi/4 stage; alternating
// output signs and the bit-reversal-like output order of the AV1 ADST flow graph.
template <int N>
__device__ __forceinline__ void iadst(const int *in, int *out, const int lo, const int hi) {
    static_assert(N == 8 || N == 16, "adst size");
    int t[N];
#pragma unroll
    for (int i = 0; i < N / 2; i++) {
        const int a = (4 * i + 1) * 32 / N;           // (4i+1) pi / (4N) on the pi/128 grid
        const int c = kCos128[a], s = kCos128[64 - a];
        const int x = in[N - 1 - 2 * i], y = in[2 * i];
        t[2 * i]     = rot(x, y, c, s);
        t[2 * i + 1] = rot(x, y, s, -c);
    }
#pragma unroll
    for (int d = N / 2; d >= 4; d /= 2) {
        // add/sub at distance d inside every 2d-group
#pragma unroll
        for (int b = 0; b < N; b += 2 * d) {
#pragma unroll
            for (int i = 0; i < d; i++) {
                const int u = t[b + i], v = t[b + i + d];
                t[b + i] = clampi(u + v, lo, hi);
                t[b + i + d] = clampi(u - v, lo, hi);
            }
        }
        // rotate pairs in the upper half of every 2d-group; the pair index inside the
        // half selects the angle (4pp+1) pi / (2d); the second half of the pairs uses the
        // mirrored form
#pragma unroll
        for (int b = d; b < N; b += 2 * d) {
#pragma unroll
            for (int p = 0; p < d / 2; p++) {
                const int u = t[b + 2 * p], v = t[b + 2 * p + 1];
                const int half = d / 4;                   // pairs per direction
                const int pp = p % half;
                const int ang = (4 * pp + 1) * 64 / d;    // (4pp+1) pi / (2d) on the pi/128 grid
                const int c = kCos128[ang], s = kCos128[64 - ang];
                if (p < half) { t[b + 2 * p] = rot(u, v, c, s);  t[b + 2 * p + 1] = rot(u, v, s, -c); }
                else          { t[b + 2 * p] = rot(v, u, c, -s); t[b + 2 * p + 1] = rot(v, u, s, c); }
            }
        }
    }
    // distance-2 add/sub producing half of the outputs directly, then the pi/4 stage
    // output index / sign pattern per group of four (g = 0 .. N/4-1)
#pragma unroll
    for (int g = 0; g < N / 4; g++) {
        const int b = 4 * g;
        const int s0 = clampi(t[b] + t[b + 2], lo, hi);
        const int s1 = clampi(t[b + 1] + t[b + 3], lo, hi);
        const int d0 = clampi(t[b] - t[b + 2], lo, hi);
        const int d1 = clampi(t[b + 1] - t[b + 3], lo, hi);
        // group -> output slot: gray-code like order of the ADST graph
        const int slot = (N == 8) ? (g == 0 ? 0 : 1)
                                  : (g == 0 ? 0 : g == 1 ? 3 : g == 2 ? 1 : 2);
        const bool neg_first = slot & 1;               // sign alternates with the slot
        const int r0 = rot45(d0 + d1), r1 = rot45(d0 - d1);
        if (!neg_first) {
            out[slot] = s0;              out[N - 1 - slot] = -s1;
            out[N / 2 - 1 - slot] = -r0; out[N / 2 + slot] = r1;
        } else {
            out[slot] = -s0;             out[N - 1 - slot] = s1;
            out[N / 2 - 1 - slot] = r0;  out[N / 2 + slot] = -r1;
        }
    }
}

// ---------------------------------------------------------------- identity (itx_1d.c:976-1017)
template <int N>
__device__ __forceinline__ void iidentity(const int *in, int *out) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int v = in[i];
        if (N == 4)       out[i] = v + (dv::mad_i24k(v, 1697, 2048) >> 12);
        else if (N == 8)  out[i] = 2 * v;
        else if (N == 16) out[i] = 2 * v + (dv::mad_i24k(v, 1697, 1024) >> 11);
        else              out[i] = 4 * v;
    }
}

// ---------------------------------------------------------------- WHT4 (itx_1d.c:1066-1082)
__device__ __forceinline__ void iwht4(const int *in, int *out) {
    const int s = in[0] + in[1];
    const int d = in[2] - in[3];
    const int m = (s - d) >> 1;
    const int p = m - in[3], q = m - in[1];
    out[0] = s - p; out[1] = p; out[2] = q; out[3] = d + q;
}

} // namespace itx1d

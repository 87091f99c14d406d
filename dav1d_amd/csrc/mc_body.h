// The motion-compensation work of one wave (gather -> horizontal -> vertical -> combine), shared by the per-shape kernels of
// mc.hip and by the fused prediction + residual kernels of recon.hip.  See mc.hip for the description of the mapping.
#pragma once
#include "common.h"
#include "capi.h"
#include "av1_tables.h"
#include <type_traits>

namespace {

struct RefSet { DevPlanes r[8]; };
// 1 / 0: every reference of the call is handed over as its tiled twin / as raster planes; -1: a mix, which no kernel variant reads
inline int refs_tiled(const DevPlanes *refs, const int n_refs) {
    int t = 0;
    for (int i = 0; i < n_refs; i++) t += refs[i].tiled != 0;
    return t == 0 ? 0 : t == n_refs ? 1 : -1;
}
static_assert(sizeof(DevPlanes) == 64 && sizeof(RefSet) == 512, "the waves copy the table to LDS dword by dword");

struct __attribute__((packed, aligned(2))) U64u { uint32_t a, b; };   // 2-byte aligned 8-byte global load
struct __attribute__((packed, aligned(1))) U64b { uint32_t a, b; };   // byte-aligned (8 bpc rows)
struct __attribute__((packed, aligned(2))) U128u { uint32_t a, b, c, d; };   // 2-byte aligned 16-byte global load

// taps of one direction packed for v_dot2: ev[k] = (f[2k], f[2k+1]), od[k] = (f[2k-1], f[2k]) with f[-1] = f[8] = 0
struct Taps { uint32_t ev[4]; uint32_t od[5]; };

__device__ __forceinline__ Taps load_taps(const int set, const int m) {
    // av1_mc_taps_packed[set 0..5 | 6 = bilinear][phase 0..15, 0 = unit tap][ev0..3, od0..4]
    const uint32_t *p = &av1_mc_taps_packed[(set * 16 + m) * 9];
    Taps t;
#ifdef DV_KO_TAPS
    for (int k = 0; k < 4; k++) t.ev[k] = (uint32_t) (set + m + k);
    for (int k = 0; k < 5; k++) t.od[k] = (uint32_t) (set - m + k);
    if (set != 12345) return t;
#endif
#pragma unroll
    for (int k = 0; k < 4; k++) t.ev[k] = p[k];
#pragma unroll
    for (int k = 0; k < 5; k++) t.od[k] = p[4 + k];
    return t;
}

constexpr int mc_cmin(int a, int b) { return a < b ? a : b; }
// window row stride in pixels: TW + 8 columns rounded up to whole 16-byte chunks.  4-wide tiles: blocks of width 4 are only ever
// filtered with the 4-tap sets, bilinear or the unit tap (reference src/mc_tmpl.c GET_H_FILTER: w > 4 ? type : 3 + (type & 1);
// the host builds the tiles by the same rule, capi.hip), whose taps 0, 1, 6 and 7 are zero: the 4 outputs of a row reach the 7
// columns src_x - 1 .. src_x + 5, so the window starts at src_x - 2 and is one 16-byte piece per row
constexpr int mc_win_stride(int tw) { return tw == 4 ? 8 : (tw + 8 + 7) & ~7; }
// The same for a reference stored as 8x8 tiles (TILED, see "tiled twin" below): the window is made of whole tile rows — aligned
// 8-pixel pieces — so it starts at the piece that holds the first column the taps reach (src_x - 3; 4-wide tiles src_x - 1) and
// holds one piece more than the span needs: TW + 7 (7) columns at any of 8 offsets
// (16-wide tiles: 8 pixels of padding.  A row pair of 2 x 32 pixels is exactly the 32 banks of the LDS, so the twelve row pairs the lanes
// of a horizontal pass read side by side all sat on the same banks — 56 % of the 16x16 pairs' LDS cycles were bank conflicts,
// profiles/r06/lds_conflicts.txt; with 40 the pairs step through four bank groups, which is what three times 32 dwords need anyway)
constexpr int mc_win_stride_tiled(int tw) { return tw == 4 ? 16 : tw == 16 ? 40 : tw + 16; }
// pieces of 8 pixels a row of the tiled window is fetched in (the stride may hold padding on top)
constexpr int mc_win_pieces_tiled(int tw) { return tw == 4 ? 2 : (tw + 16) / 8; }

// LDS bytes one wave needs for tile shape (TW, TH): window + row-pair intermediate + the reference table (the tile records pass through
// the window buffer before anything is gathered).  The LDS is handed out in pieces of 1,280 bytes (tools/calib/residency.hip: 5.9 KB hold
// 25 workgroups on a CU, not 27): the 4x4 launch's 6,432 bytes were six pieces, 21 waves per CU; without the records five, 25
// Rows of the window a tile of TH rows KEEPS in LDS, and the first of them.  The bodies below index a window of TH + 8 rows (the reach of
// eight vertical taps).  Tiles of 4 rows only ever meet the 4-tap, bilinear or unit sets vertically (blocks of height <= 4, reference
// src/mc_tmpl.c GET_V_FILTER): rows 2 .. 8 of that window.  They keep 8 rows (2 .. 9, whole row pairs) and hang their window two rows
// higher than it is stored: the rows they never store or read with a non-zero tap — 0, 1, 10, 11 — lie in the NEIGHBOURING tiles' rows of
// the shared buffer.  The 4x4 kernel's LDS per wave 8,896 -> 6,464 bytes: 4.25 -> 6 waves per SIMD.
constexpr int mc_win_rows(int th) { return th == 4 ? 8 : th + 8; }
constexpr int mc_win_row0(int th) { return th == 4 ? 2 : 0; }
template <int TW, int TH, bool TILED = false>
constexpr int mc_lds_bytes() {
    constexpr int LPT = mc_cmin(64, TW * TH / 4), G = 64 / LPT, WS = TILED ? mc_win_stride_tiled(TW) : mc_win_stride(TW);
    constexpr int WRA = mc_win_rows(TH), WR0 = mc_win_row0(TH), NPRA = WRA / 2;
    // window rows (+ WR0 rows in front: the first tile's rows 0, 1 are addressable), row pairs of the intermediate (4-row tiles: + a pair at either end)
    return (WR0 + G * WRA) * WS * 2 + (G * NPRA + (TH == 4 ? 2 : 0)) * TW * 4 + (G > 1 ? (int) sizeof(RefSet) : 0);
}

// ---- the shape of a tile class and what the stages of one prediction share
template <int TW_, int TH_, typename pixel, bool TILED>
struct McShape {
    static constexpr int TW = TW_, TH = TH_;
    static constexpr int NS = TW / 4;                          // 4-pixel strips per row
    static constexpr int LPT = mc_cmin(64, TW * TH / 4);       // lanes per tile
    static constexpr int G = 64 / LPT;                         // tiles side by side in a wave
    static constexpr int R = TW * TH / 4 / LPT;                // output strips per lane (1, 2 or 4)
    static constexpr int WS = TILED ? mc_win_stride_tiled(TW) : mc_win_stride(TW);    // window row stride (int16)
    static constexpr int WR = TH + 8;                          // window rows ADDRESSED (TH+7 used, +1 so row pairs are complete)
    static constexpr int WRA = mc_win_rows(TH), WR0 = mc_win_row0(TH);   // ... of which rows WR0 .. WR0 + WRA - 1 are this tile's own in LDS (see mc_win_rows)
    static constexpr int WRG = TH == 4 ? WRA : WR - 1;         // rows the raster gathers fetch and store: WR0 .. WR0 + WRG - 1
    static constexpr int NCH = TILED ? mc_win_pieces_tiled(TW) : (WS + 7) / 8;   // 8-pixel (16-byte) chunks fetched per window row
    static constexpr int NPR = WR / 2;                         // row pairs of the intermediate (addressed)
    static constexpr int NPRA = WRA / 2, PR0 = WR0 / 2;        // ... kept per tile, first kept
    static constexpr int NLD = (WRG * NCH + LPT - 1) / LPT;    // window loads per lane
    static constexpr bool NARROW = TW == 4;                    // window columns start at src_x - 2 instead of src_x - 4, taps 2 .. 5 only
    static constexpr bool HBD = sizeof(pixel) == 2;
};

// one prediction (one reference of one tile): what its filters are and which part of the window they reach
struct McPred {
    int fbits;
    int row_lo, row_hi;       // window rows the vertical taps can reach: the others only ever meet zero taps, so they are neither
                              // fetched nor filtered (6-tap regular, 4-tap smooth / small blocks, 2-tap bilinear, 1-tap full-pel)
    int tx0, xa, toff;        // TILED: the window starts at the aligned piece that holds its first tap column; `toff` = that column's offset in the piece
};
template <int TW, int TH>
__device__ __forceinline__ McPred mc_pred_of(const McRef rf) {
    constexpr bool NARROW = TW == 4;
    McPred p;
    p.fbits = rf.fh == 6 ? 4 : 6;
    const int vspan = rf.vspan;
    // (4-row tiles keep rows 2 .. 9 only, mc_win_rows: their filters reach rows 2 .. 8; the clamp is for a caller that breaks the rule)
    p.row_lo = TH == 4 ? dv::imax(vspan & 15, 2) : vspan & 15;
    p.row_hi = TH == 4 ? dv::imin(TH - 1 + (vspan >> 4), 9) : TH - 1 + (vspan >> 4);
    p.tx0 = rf.src_x - (NARROW ? 1 : 3); p.xa = p.tx0 & ~7; p.toff = p.tx0 & 7;
    return p;
}

// ---- 1. gather the window of one prediction into `win` (this tile's rows of the wave's LDS window buffer); l = this lane's index in its tile
template <int TW, int TH, typename pixel, bool TILED>
__device__ __forceinline__ void mc_gather(const McRef rf, const McPred &pd, const pixel *src, const int rs, const int rw, const int rh,
                                          int16_t *const win, const int l)
{
    typedef McShape<TW, TH, pixel, TILED> S;
    constexpr int NS = S::NS, LPT = S::LPT, R = S::R, WS = S::WS, WR = S::WR, WR0 = S::WR0, WRG = S::WRG, NCH = S::NCH, NPR = S::NPR, NLD = S::NLD;
    constexpr bool NARROW = S::NARROW, HBD = S::HBD;
    (void) NS; (void) LPT; (void) R; (void) WS; (void) WR; (void) WR0; (void) WRG; (void) NCH; (void) NPR; (void) NLD; (void) NARROW; (void) HBD;
    const int row_lo = pd.row_lo, row_hi = pd.row_hi, xa = pd.xa;
    if constexpr (TILED) {
        // columns the horizontal taps reach (the span table again, origin src_x - 3) and rows the vertical ones do: the window
        // is "inside" when those are — the rest of it is neither fetched nor met by a non-zero tap
        const int c_first = rf.src_x - 3 + (rf.hspan & 15), c_last = rf.src_x - 3 + TW - 2 + (rf.hspan >> 4);
        const int y0 = rf.src_y - 3;
        const bool interior = c_first >= 0 && c_last < rw && y0 + row_lo >= 0 && y0 + row_hi <= rh;
        if (interior) {
            // Lanes in a grid of rows x pieces, both powers of two (no division): lane -> (row lr of its row group, piece
            // column lp), the loops step over row groups and piece columns.  Consecutive lanes walk down the rows of one
            // tile column: 8 of them share a 128-byte line.  Tiles of 4 rows only ever meet the 4-tap / bilinear / unit
            // sets vertically (blocks of height <= 4, reference GET_V_FILTER): rows 2 .. TH + 4 of the window.
            constexpr int R0 = TH == 4 ? 2 : 0, NR = TH == 4 ? TH + 3 : WR - 1;
            constexpr int RG = NR <= 8 ? 8 : NR <= 16 ? 16 : 32;                 // lanes of a row group
            constexpr int RPAR = mc_cmin(LPT, RG), NRI = (NR + RPAR - 1) / RPAR;
            constexpr int PG = LPT >= RG ? LPT / RG : 1, NPI = (NCH + PG - 1) / PG;
            typedef typename std::conditional<HBD, uint4, uint2>::type piece_t;
            const int lr = l & (RPAR - 1), lp = PG > 1 ? l / RG : 0;
            piece_t ld[NRI][NPI];
            bool ok[NRI][NPI];
#pragma unroll
            for (int ri = 0; ri < NRI; ri++) {
                const int wr = R0 + lr + ri * RPAR, y = y0 + wr;
                const bool row_ok = lr + ri * RPAR < NR && wr >= row_lo && wr < row_hi;
                const pixel *prow = src + (dv::mul_i24(y & ~7, rs) + ((y & 7) << 3));
#pragma unroll
                for (int pi = 0; pi < NPI; pi++) {
                    const int pc = lp + pi * PG, x = xa + 8 * pc;
                    ok[ri][pi] = row_ok && pc < NCH && x + 7 >= c_first && x <= c_last;
#ifdef DV_KO_GATHER      // (knock-out variant builds, tools/knockout.sh: timing only, the pixels are wrong)
                    if (ok[ri][pi]) { ld[ri][pi].x = (unsigned) (size_t) prow; ld[ri][pi].y = (unsigned) x; if constexpr (HBD) { ld[ri][pi].z = 0; ld[ri][pi].w = 1; } }
#else
                    if (ok[ri][pi]) ld[ri][pi] = *reinterpret_cast<const piece_t *>(prow + (x << 3));
#endif
                }
            }
            // pieces that were not fetched are not stored either: what they would hold only ever meets zero taps
#pragma unroll
            for (int ri = 0; ri < NRI; ri++)
#pragma unroll
                for (int pi = 0; pi < NPI; pi++) {
                    if (!ok[ri][pi]) continue;
                    const int wr = R0 + lr + ri * RPAR, pc = lp + pi * PG;
                    uint4 v;
                    if constexpr (HBD) v = ld[ri][pi];
                    else v = make_uint4(__builtin_amdgcn_perm(0u, ld[ri][pi].x, 0x0c010c00u), __builtin_amdgcn_perm(0u, ld[ri][pi].x, 0x0c030c02u),
                                        __builtin_amdgcn_perm(0u, ld[ri][pi].y, 0x0c010c00u), __builtin_amdgcn_perm(0u, ld[ri][pi].y, 0x0c030c02u));
                    *reinterpret_cast<uint4 *>(win + wr * WS + 8 * pc) = v;
                }
        } else {
            // edge emulation through the tile addressing: per-pixel clamped fetch, 8 independent loads in flight per lane
            for (int i0 = l; i0 < WRG * WS; i0 += 8 * LPT) {
                pixel v[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int i = dv::imin(i0 + e * LPT, WRG * WS - 1);
                    const int wq = dv::div_small<WS>(i), wr = WR0 + wq;
                    const int sy = dv::iclip(y0 + wr, 0, rh - 1);
                    const int sx = dv::iclip(xa + (i - wq * WS), 0, rw - 1);
                    v[e] = src[dv::mul_i24(sy & ~7, rs) + ((sx >> 3) << 6) + ((sy & 7) << 3) + (sx & 7)];
                }
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int i = i0 + e * LPT;
                    if (i < WRG * WS) win[WR0 * WS + i] = (int16_t) v[e];
                }
            }
        }
    } else {
    const int x0 = rf.src_x - (NARROW ? 2 : 4), y0 = rf.src_y - 3;
    const bool interior = x0 >= 0 && y0 + WR0 >= 0 && x0 + NCH * 8 <= rw && y0 + WR0 + WRG <= rh;
    if (interior) {
        // 16-byte (8-pixel) loads, rows at arbitrary 2-byte alignment; all of a lane's loads are
        // issued before the first LDS write
        const pixel *base = src + y0 * rs + x0;
        uint4 ld[NLD];
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int i = dv::imin(l + k * LPT, WRG * NCH - 1);
            const int wq = dv::div_small<NCH>(i), wr = WR0 + wq;      // row and 8-pixel piece of the window; the offset stays in 32 bits
            const pixel *p = base + (dv::mul_i24(wr, rs) + 8 * (i - wq * NCH));
            ld[k] = make_uint4(0, 0, 0, 0);
            if (wr < row_lo || wr >= row_hi) continue;
            if (HBD) {
                const U128u v = *reinterpret_cast<const U128u *>(p);
                ld[k] = make_uint4(v.a, v.b, v.c, v.d);
            } else {
                const U64b v = *reinterpret_cast<const U64b *>(p);
                // bytes -> 16-bit lanes: one byte permute per pair of pixels (selector 0x0c = a zero byte)
                ld[k] = make_uint4(__builtin_amdgcn_perm(0u, v.a, 0x0c010c00u), __builtin_amdgcn_perm(0u, v.a, 0x0c030c02u),
                                   __builtin_amdgcn_perm(0u, v.b, 0x0c010c00u), __builtin_amdgcn_perm(0u, v.b, 0x0c030c02u));
            }
        }
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int i = l + k * LPT;
            if (i >= WRG * NCH) continue;
            const int wq = dv::div_small<NCH>(i), wr = WR0 + wq;
            int16_t *const wp = win + wr * WS + 8 * (i - wq * NCH);
            if (WS % 8 == 0) {
                *reinterpret_cast<uint4 *>(wp) = ld[k];
            } else {        // 12-column rows: 8-byte stores, the last chunk keeps only its first half
                *reinterpret_cast<uint2 *>(wp) = make_uint2(ld[k].x, ld[k].y);
                if (8 * (i - wq * NCH) + 8 <= WS) *reinterpret_cast<uint2 *>(wp + 4) = make_uint2(ld[k].z, ld[k].w);
            }
        }
    } else {
        // edge emulation: per-pixel clamped fetch, 8 independent loads in flight per lane
        for (int i0 = l; i0 < WRG * WS; i0 += 8 * LPT) {
            pixel v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int i = dv::imin(i0 + e * LPT, WRG * WS - 1);
                const int sy = dv::iclip(y0 + WR0 + i / WS, 0, rh - 1);
                const int sx = dv::iclip(x0 + i % WS, 0, rw - 1);
                v[e] = src[sy * rs + sx];
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int i = i0 + e * LPT;
                if (i < WRG * WS) win[WR0 * WS + i] = (int16_t) v[e];
            }
        }
    }
    }
}

// ---- 2. horizontal pass: item = (row pair, strip) -> mid2[pair][4 cols] = (even row, odd row)
template <int TW, int TH, typename pixel, bool TILED>
__device__ __forceinline__ void mc_hpass(const McPred &pd, const Taps &fh, const int16_t *const win, uint32_t *const mid, const int l,
                                         const int ib, const int bias, const bool as_prep)
{
    typedef McShape<TW, TH, pixel, TILED> S;
    constexpr int NS = S::NS, LPT = S::LPT, R = S::R, WS = S::WS, WR = S::WR, WR0 = S::WR0, WRG = S::WRG, NCH = S::NCH, NPR = S::NPR, NLD = S::NLD;
    constexpr bool NARROW = S::NARROW, HBD = S::HBD;
    (void) NS; (void) LPT; (void) R; (void) WS; (void) WR; (void) WR0; (void) WRG; (void) NCH; (void) NPR; (void) NLD; (void) NARROW; (void) HBD;
    const int fbits = pd.fbits, row_lo = pd.row_lo, row_hi = pd.row_hi, toff = pd.toff;
    (void) toff; (void) bias; (void) as_prep;
    // One formula for the four cases of the reference (src/mc_tmpl.c:129-187 put, :246-305 prep, :434-586 bilinear): the intermediate is
    // (sum + rnd) >> (fbits - ib) whether there is a horizontal filter or not — without one the tap table holds the unit tap at the
    // filters' scale (64; bilinear 16), the sum is the pixel times the scale and the shift leaves pixel << ib exactly — and the vertical
    // pass (mc_vpass) finishes every sample the same way.  The reference's H-only and V-only forms are these with the other direction's
    // unit tap: ((s + r1) >> (6 - ib) + r2) >> ib == (s + 32 + r1) >> 6, and sums of pixel << ib have no low bits to round.
    const int sh1 = fbits - ib;
    const int rnd1 = (1 << sh1) >> 1;
    // the four sums of one 4-pixel strip of one window row (rounding offset included)
    auto row_sums = [&](const int row, const int s, int (&o)[4]) {
        const uint2 *wp = reinterpret_cast<const uint2 *>(win + row * WS + 4 * s);
        int s0 = rnd1, s1 = rnd1, s2 = rnd1, s3 = rnd1;
        if constexpr (TILED) {
            // Output x of this strip sums f[k] * p[c + k] from window column c = toff + 4 s + x on.  c even: the pixel
            // pairs of the row's dwords meet the tap pairs (f0, f1) (f2, f3) .. = ev[]; c odd: (0, f0) (f1, f2) .. = od[]
            // one dword earlier.  Whether the strip's first column is even is a property of the tile (lists group
            // tiles by it so that a wave rarely holds both kinds).
            const uint32_t *dw = reinterpret_cast<const uint32_t *>(win + row * WS) + ((toff >> 1) + 2 * s);
            if constexpr (NARROW) {
                // taps 2 .. 5 only: the sums start one tap pair (two columns) into the 8-tap layout
                const uint32_t d0 = dw[0], d1 = dw[1], d2 = dw[2], d3 = dw[3];
                if (!(toff & 1)) {
                    s0 = dv::dot2(d0, fh.ev[1], dv::dot2(d1, fh.ev[2], s0));
                    s1 = dv::dot2(d0, fh.od[1], dv::dot2(d1, fh.od[2], dv::dot2(d2, fh.od[3], s1)));
                    s2 = dv::dot2(d1, fh.ev[1], dv::dot2(d2, fh.ev[2], s2));
                    s3 = dv::dot2(d1, fh.od[1], dv::dot2(d2, fh.od[2], dv::dot2(d3, fh.od[3], s3)));
                } else {
                    s0 = dv::dot2(d0, fh.od[1], dv::dot2(d1, fh.od[2], dv::dot2(d2, fh.od[3], s0)));
                    s1 = dv::dot2(d1, fh.ev[1], dv::dot2(d2, fh.ev[2], s1));
                    s2 = dv::dot2(d1, fh.od[1], dv::dot2(d2, fh.od[2], dv::dot2(d3, fh.od[3], s2)));
                    s3 = dv::dot2(d2, fh.ev[1], dv::dot2(d3, fh.ev[2], s3));
                }
            } else {
                const uint32_t d[6] = { dw[0], dw[1], dw[2], dw[3], dw[4], dw[5] };
                if (!(toff & 1)) {
#pragma unroll
                    for (int k = 0; k < 4; k++) { s0 = dv::dot2(d[k], fh.ev[k], s0); s2 = dv::dot2(d[k + 1], fh.ev[k], s2); }
#pragma unroll
                    for (int k = 0; k < 5; k++) { s1 = dv::dot2(d[k], fh.od[k], s1); s3 = dv::dot2(d[k + 1], fh.od[k], s3); }
                } else {
#pragma unroll
                    for (int k = 0; k < 5; k++) { s0 = dv::dot2(d[k], fh.od[k], s0); s2 = dv::dot2(d[k + 1], fh.od[k], s2); }
#pragma unroll
                    for (int k = 0; k < 4; k++) { s1 = dv::dot2(d[k + 1], fh.ev[k], s1); s3 = dv::dot2(d[k + 2], fh.ev[k], s3); }
                }
            }
        } else if constexpr (NARROW) {
            // out x sums f[k] * p[x - 1 + k] over k = 2 .. 5, p[] = the 8 pixels of the row: the tap pairs (f1, f2) (f3, f4)
            // (f5, f6) and (f2, f3) (f4, f5) of the 8-tap layout meet pixel pairs two columns further left
            const uint2 a = wp[0], b = wp[1];
            const uint32_t d[4] = { a.x, a.y, b.x, b.y };
#pragma unroll
            for (int k = 0; k < 3; k++) { s0 = dv::dot2(d[k], fh.od[k + 1], s0); s2 = dv::dot2(d[k + 1], fh.od[k + 1], s2); }
#pragma unroll
            for (int k = 0; k < 2; k++) { s1 = dv::dot2(d[k + 1], fh.ev[k + 1], s1); s3 = dv::dot2(d[k + 2], fh.ev[k + 1], s3); }
        } else {
            const uint2 a = wp[0], b = wp[1], c = wp[2];
            const uint32_t d[6] = { a.x, a.y, b.x, b.y, c.x, c.y };
            // out x sums f[k] * p[x + 1 + k], p[] = the 12 pixels of d[]; the rounding offset seeds the sum
#pragma unroll
            for (int k = 0; k < 5; k++) { s0 = dv::dot2(d[k], fh.od[k], s0); s2 = dv::dot2(d[k + 1], fh.od[k], s2); }
#pragma unroll
            for (int k = 0; k < 4; k++) { s1 = dv::dot2(d[k + 1], fh.ev[k], s1); s3 = dv::dot2(d[k + 2], fh.ev[k], s3); }
        }
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
    };
    // Items of (row pair, strip) dealt over the tile's lanes.  Where the last round would have items for exactly half of the lanes (32x16
    // tiles: 96 items, 64 lanes) it is dealt as SINGLE rows over all of them instead — the same rows in half the time; the row's
    // four values go into their halves of the pair-interleaved intermediate as 16-bit stores.
    constexpr int NIT = NPR * NS, FULL = NIT / LPT * LPT;
    constexpr bool SPLIT = (NIT - FULL) * 2 == LPT;
#pragma unroll
    for (int it0 = 0; it0 < (SPLIT ? FULL : NIT); it0 += LPT) {
        const int it = it0 + l;
        if (it >= NIT) break;
        const int pr = it / NS, s = it % NS;
        if (2 * pr + 1 < row_lo || 2 * pr >= row_hi) continue;
        int o[2][4];
        row_sums(2 * pr, s, o[0]);
        row_sums(2 * pr + 1, s, o[1]);
        // intermediate rounding, reference src/mc_tmpl.c:150-152 (8-tap) / 462-464 (bilinear)
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
            for (int x = 0; x < 4; x++) o[e][x] >>= sh1;
        uint4 m;
        m.x = dv::pack2(o[0][0], o[1][0]);
        m.y = dv::pack2(o[0][1], o[1][1]);
        m.z = dv::pack2(o[0][2], o[1][2]);
        m.w = dv::pack2(o[0][3], o[1][3]);
        *reinterpret_cast<uint4 *>(mid + pr * TW + 4 * s) = m;
    }
    if constexpr (SPLIT) {
        const int it = FULL + (l >> 1), pr = it / NS, s = it % NS, row = 2 * pr + (l & 1);
        if (row >= row_lo && row < row_hi) {
            int o[4];
            row_sums(row, s, o);
            int16_t *const mp = reinterpret_cast<int16_t *>(mid + pr * TW + 4 * s) + (l & 1);
#pragma unroll
            for (int x = 0; x < 4; x++) mp[2 * x] = (int16_t) (o[x] >> sh1);
        }
    }
}

// ---- 3. vertical pass: item = (output row, strip), R items per lane -> q[R][4]
template <int TW, int TH, typename pixel, bool TILED>
__device__ __forceinline__ void mc_vpass(const McPred &pd, const Taps &fv, const uint32_t *const mid, const int l,
                                         const int ib, const int bias, const bool as_prep, int (&q)[McShape<TW, TH, pixel, TILED>::R][4])
{
    typedef McShape<TW, TH, pixel, TILED> S;
    constexpr int NS = S::NS, LPT = S::LPT, R = S::R, WS = S::WS, WR = S::WR, WR0 = S::WR0, WRG = S::WRG, NCH = S::NCH, NPR = S::NPR, NLD = S::NLD;
    constexpr bool NARROW = S::NARROW, HBD = S::HBD;
    (void) NS; (void) LPT; (void) R; (void) WS; (void) WR; (void) WR0; (void) WRG; (void) NCH; (void) NPR; (void) NLD; (void) NARROW; (void) HBD;
    const int fbits = pd.fbits;
    // pixels: (sum + rnd) >> (fbits + ib), src/mc_tmpl.c:157-159, 176-178; intermediates of a compound: (sum + rnd) >> fbits, less the bias, :272-277, 294-299
    const int sh2 = as_prep ? fbits : fbits + ib, vb = as_prep ? bias : 0;
    const int rnd2 = (1 << sh2) >> 1;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int it = r * LPT + l;
        const int vr = it / NS, vs = it % NS;
        // rows vr .. vr+7 of the window = pairs j0 .. j0+4; odd vr starts in the middle of a pair
        const int j0 = vr >> 1;
        const bool odd = vr & 1;
        int sum[4] = { rnd2, rnd2, rnd2, rnd2 };
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const uint32_t g = odd ? fv.od[k] : (k < 4 ? fv.ev[k] : 0u);
            const int j = dv::imin(j0 + k, NPR - 1);     // the 5th pair of an even row is weight 0
            const uint4 m = *reinterpret_cast<const uint4 *>(mid + j * TW + 4 * vs);
            sum[0] = dv::dot2(m.x, g, sum[0]);
            sum[1] = dv::dot2(m.y, g, sum[1]);
            sum[2] = dv::dot2(m.z, g, sum[2]);
            sum[3] = dv::dot2(m.w, g, sum[3]);
        }
#pragma unroll
        for (int x = 0; x < 4; x++) q[r][x] = (sum[x] >> sh2) - vb;
    }
}

// One wave's worth of tiles of shape (TW, TH): tiles[t0 .. t0 + nt), nt <= 64 / LPT.  `smem` = mc_lds_bytes<TW, TH>() of LDS.
// TO_LDS (fused prediction + residual kernels): pixels of PUT / AVG / WAVG tiles go to pred_s instead of the picture — block
// (tile index - pred_tile0) >> pred_tpb_log2 of the wave, pred_w x pred_h pixels each, row stride pred_w, pred_w x (pred_h + 1) pixels from
// one block to the next (itx_tile_stride, itx_body.h).
//
// TILED (the "tiled twin" of a reference picture, written by dav1d_hip_picture_retile / the frame's last stage): refs.r[].data
// point to planes of the same size and stride whose pixels are stored as 8x8 tiles, 64 consecutive pixels each (128 bytes at
// 10 / 12 bits: one memory line), tile (tx, ty) at pixel offset ty * 8 * stride + tx * 64, row r of the tile at + 8 r.  A window
// of (TW + 7) x (TH + 7) pixels at an arbitrary position then touches (1 + (TW + 6) / 8) x (1 + (TH + 6) / 8) lines instead of one
// or two per ROW (a raster row of 48 bytes costs 1.4 lines: measured, profiles/r02_calib_fetch_size.txt) — for an 8x8 block 7.6
// lines instead of 18.  The gather fetches whole tile rows (aligned 16-byte pieces; 8 consecutive lanes = the 8 rows of one tile =
// one line), and the horizontal pass picks its taps by the parity of the window's offset inside the piece instead of shifting data.
// TWIN: pixels that go to the picture (PUT / AVG / WAVG tiles outside the paired kernels) go to the planes of its tiled twin as well
// (`twin`: same strides; every 4-pixel strip lies inside one tile row).
template <int TW, int TH, typename pixel, bool TO_LDS = false, bool TILED = false, bool TWIN = false>
__device__ __forceinline__ void mc_body(const DevPlanes &dst, const RefSet &refs, const McTile *__restrict__ tiles, const int t0, const int nt,
                                        int16_t *__restrict__ prep, const int bitdepth_max, uint4 *smem,
                                        pixel *pred_s = nullptr, const int pred_tile0 = 0, const int pred_tpb_log2 = 0,
                                        const int pred_w = 0, const int pred_h = 0, const DevPlanes &twin = DevPlanes())
{
    constexpr int NS = TW / 4;                          // 4-pixel strips per row
    constexpr int LPT = mc_cmin(64, TW * TH / 4);       // lanes per tile
    constexpr int G = 64 / LPT;                         // tiles side by side in a wave
    constexpr int R = TW * TH / 4 / LPT;                // output strips per lane (1, 2 or 4)
    constexpr int WS = TILED ? mc_win_stride_tiled(TW) : mc_win_stride(TW);    // window row stride (int16)
    constexpr int WR = TH + 8;                          // window rows ADDRESSED (TH+7 used, +1 so row pairs are complete)
    constexpr int WRA = mc_win_rows(TH), WR0 = mc_win_row0(TH);   // ... of which rows WR0 .. WR0 + WRA - 1 are this tile's own in LDS (see mc_win_rows)
    constexpr int WRG = TH == 4 ? WRA : WR - 1;         // rows the raster gathers fetch and store: WR0 .. WR0 + WRG - 1
    constexpr int NCH = TILED ? mc_win_pieces_tiled(TW) : (WS + 7) / 8;   // 8-pixel (16-byte) chunks fetched per window row
    constexpr int NPR = WR / 2;                         // row pairs of the intermediate (addressed)
    constexpr int NPRA = WRA / 2, PR0 = WR0 / 2;        // ... kept per tile, first kept
    constexpr int NLD = (WRG * NCH + LPT - 1) / LPT;    // window loads per lane
    constexpr bool NARROW = TW == 4;                        // window columns start at src_x - 2 instead of src_x - 4, taps 2 .. 5 only
    constexpr bool HBD = sizeof(pixel) == 2;
    // phase slots of this tile shape (DV_PHASES builds): 0 records, 1 window gather (first reference), 2 horizontal, 3 vertical,
    // 4 / 5 / 6 the same of the second reference, 7 combine + store, 8 whole body, 9 bodies counted
    constexpr int PH = ((TW == 4 ? 0 : TW == 8 ? 1 : TW == 16 ? 2 : TW == 32 ? 3 : 4) * 3 + (TH == 4 ? 0 : TH == 8 ? 1 : 2)) * 16 + (TO_LDS ? 256 : 0);
    DV_PHASE_BEGIN();
    int dv_second_ = 0;

    int16_t *const win_s = reinterpret_cast<int16_t *>(smem);
    constexpr int SL = TH == 4 ? TW : 0;                // 4-row tiles: one pair of slack in front of and behind the intermediates
    uint32_t *const mid_s = reinterpret_cast<uint32_t *>(win_s + (WR0 + G * WRA) * WS) + SL;

    const int lane = dv::lane_id();       // the body belongs to one wave (recon.hip runs several side by side in a workgroup)
    // G == 1: the whole wave works on one tile, so the record, the taps and all the control flow
    // derived from them are wave-uniform (scalar loads, SGPRs, s_cbranch instead of exec masking)
    const int sub = G == 1 ? 0 : lane / LPT, l = G == 1 ? lane : lane % LPT;
    const int ti = t0 + sub;
    const bool live = sub < nt;

    McTile t;
    if (G == 1) {
        t = tiles[__builtin_amdgcn_readfirstlane(live ? ti : 0)];
    } else {
        // the G records of the wave come in with one coalesced sweep and are handed to their lanes through LDS
        constexpr int RW = sizeof(McTile) / 4;
        static_assert(G * (int) sizeof(McTile) <= G * WRA * WS * 2, "the records fit the window buffer");
        uint32_t *const rec_s = reinterpret_cast<uint32_t *>(win_s);      // (dead until the gather: every lane has its record in registers first)
        const uint32_t *recs = reinterpret_cast<const uint32_t *>(tiles + t0);
        const int nw = nt * RW;
        for (int i = lane; i < nw; i += 64) rec_s[i] = recs[i];
        // the reference plane table goes to LDS in the same round trip: the lanes index it by their own tile's reference,
        // and a second, dependent trip to the kernel arguments would sit in front of every window fetch
        uint32_t *const ref_s = mid_s + G * NPRA * TW + SL;
        const uint32_t *rsrc = reinterpret_cast<const uint32_t *>(&refs);
#pragma unroll
        for (int i = 0; i < (int) sizeof(RefSet) / 4; i += 64) ref_s[i + lane] = rsrc[i + lane];
        dv::wave_sync();
        const uint32_t *rp = rec_s + (live ? sub : 0) * RW;
        uint32_t *tw_ = reinterpret_cast<uint32_t *>(&t);
#pragma unroll
        for (int i = 0; i < RW; i++) tw_[i] = rp[i];
        dv::wave_sync();                    // (the gather writes where the records lie)
    }

    // row r of the tile's window at win + r * WS, pair j of its intermediate at mid + j * TW: the tile's own rows start WR0 rows in
    int16_t *const win = win_s + sub * WRA * WS;                       // (= stored row WR0 of tile `sub` minus WR0 rows, WR0 rows of slack in front)
    uint32_t *const mid = mid_s + sub * NPRA * TW - PR0 * TW;
    DV_PHASE(PH + 0);

    const int ib = HBD ? 14 - (32 - __clz(bitdepth_max)) : 4;   // intermediate_bits
    const int bias = HBD ? 8192 : 0;                            // PREP_BIAS
#ifdef DV_KO_SECOND
    const bool compound = false;
#else
    const bool compound = live && (t.kind == MCT_AVG || t.kind == MCT_WAVG);
#endif
    const bool as_prep = t.kind != MCT_PUT && t.kind != MCT_PUT_TMP;   // PREP and both inputs of a compound tile

    int acc0[R][4], q[R][4];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int x = 0; x < 4; x++) acc0[r][x] = q[r][x] = 0;

    // one prediction (gather -> h -> v) of this lane's strips into q[]; a lambda invoked once or
    // twice rather than a loop over t.r[] so that the record is never indexed dynamically
    auto predict = [&](const McRef rf, const bool act) {
        const McPred pd = mc_pred_of<TW, TH>(rf);
        const Taps fh = load_taps(rf.fh, rf.mx), fv = load_taps(rf.fv, rf.my);
        if (act) {
            const pixel *src;
            int rs, rw, rh;
            if (G == 1) {
                const DevPlanes &rp = refs.r[rf.ref];
                src = reinterpret_cast<const pixel *>(rp.data[t.plane]);
                rs = rp.stride[t.plane]; rw = rp.w[t.plane]; rh = rp.h[t.plane];
            } else {
                constexpr int RW = sizeof(McTile) / 4, DW = sizeof(DevPlanes) / 4;
                const uint32_t *rt = mid_s + G * NPRA * TW + SL + rf.ref * DW;      // == ref_s above
                const uint32_t *pdat = rt + 2 * t.plane;
                src = reinterpret_cast<const pixel *>((uint64_t) pdat[0] | ((uint64_t) pdat[1] << 32));
                rs = (int) rt[6 + t.plane]; rw = (int) rt[9 + t.plane]; rh = (int) rt[12 + t.plane];
            }
            mc_gather<TW, TH, pixel, TILED>(rf, pd, src, rs, rw, rh, win, l);
        }
        dv::wave_sync();
        DV_PHASE(PH + 1 + 3 * dv_second_);
#ifdef DV_KO_HV
        if (act && bitdepth_max == 12345) {
#else
        if (act) {
#endif
            mc_hpass<TW, TH, pixel, TILED>(pd, fh, win, mid, l, ib, bias, as_prep);
        }
        dv::wave_sync();
        DV_PHASE(PH + 2 + 3 * dv_second_);
#ifdef DV_KO_HV
        if (act && bitdepth_max == 12345) {
#else
        if (act) {
#endif
            mc_vpass<TW, TH, pixel, TILED>(pd, fv, mid, l, ib, bias, as_prep, q);
        }
        DV_PHASE(PH + 3 + 3 * dv_second_);
    };

    predict(t.r[0], live);
    // the second prediction of the compound tiles: entered by the whole wave (lanes of single-reference tiles idle through
    // it) so that every lane passes the same LDS hand-off points, whatever follows this body
    if (__any(compound)) {
        if (compound) {
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int x = 0; x < 4; x++) acc0[r][x] = q[r][x];
        }
        dv::wave_sync();                        // the second gather overwrites win / mid
        dv_second_ = 1;
        predict(t.r[1], compound);
    }

    // ---- combine + store
    int tbx = 0, tby = 0;          // TWIN: the block's position in its plane
    const int tstride = t.plane == 0 ? dst.stride[0] : t.plane == 1 ? dst.stride[1] : dst.stride[2];
    if (TWIN && live && t.kind != MCT_PREP && t.kind != MCT_PUT_TMP) dv::off_to_xy(t.dst_off, tstride, tbx, tby);
    if (live) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int it = r * LPT + l;
            const int vr = it / NS, vs = it % NS;
            if (vr >= t.h) continue;
            int o[4];
            if (t.kind == MCT_AVG) {
#pragma unroll
                for (int x = 0; x < 4; x++) o[x] = (acc0[r][x] + q[r][x] + (1 << ib) + bias * 2) >> (ib + 1);          // avg_c
            } else if (t.kind == MCT_WAVG) {
#pragma unroll
                for (int x = 0; x < 4; x++)
                    o[x] = dv::mad_i24(acc0[r][x], t.weight, dv::mad_i24(q[r][x], 16 - t.weight, (8 << ib) + bias * 16)) >> (ib + 4);  // w_avg_c
            } else {
#pragma unroll
                for (int x = 0; x < 4; x++) o[x] = q[r][x];
            }
            const int nvalid = dv::imin(4, t.w - 4 * vs);
            if (t.kind != MCT_PREP) {
#pragma unroll
                for (int x = 0; x < 4; x++) o[x] = dv::clamp3(o[x], 0, bitdepth_max);
                // PUT_TMP: pixels into the scratch arena, row stride = block width (the reference's `lap` buffer of obmc())
                pixel *d = t.kind == MCT_PUT_TMP
                    ? reinterpret_cast<pixel *>(prep) + t.dst_off + (t.oy + vr) * t.bw + t.ox + 4 * vs
                    : TO_LDS ? pred_s + ((ti - pred_tile0) >> pred_tpb_log2) * (pred_w * (pred_h + 1)) + (t.oy + vr) * pred_w + t.ox + 4 * vs
                    : reinterpret_cast<pixel *>(dst.data[t.plane]) + t.dst_off + (t.oy + vr) * dst.stride[t.plane] + t.ox + 4 * vs;
                // TWIN with twin.tiled == 2: the picture lives in its twin only (DAV1D_HIP_TWIN_ONLY) — nothing goes to the raster planes
                const bool to_raster = !(TWIN && !TO_LDS && twin.tiled == 2 && t.kind != MCT_PUT_TMP);
                if (!to_raster) {
                } else if (nvalid == 4) {
                    if (HBD) *reinterpret_cast<uint2 *>(d) = make_uint2(dv::pack2(o[0], o[1]), dv::pack2(o[2], o[3]));
                    else *reinterpret_cast<uint32_t *>(d) = (uint32_t) o[0] | ((uint32_t) o[1] << 8) | ((uint32_t) o[2] << 16) | ((uint32_t) o[3] << 24);
                } else if (nvalid > 0) {
                    for (int x = 0; x < nvalid; x++) d[x] = (pixel) o[x];
                }
                if (TWIN && !TO_LDS && t.kind != MCT_PUT_TMP && nvalid > 0) {
                    const int X = tbx + t.ox + 4 * vs, Y = tby + t.oy + vr;
                    pixel *const td = reinterpret_cast<pixel *>(t.plane == 0 ? twin.data[0] : t.plane == 1 ? twin.data[1] : twin.data[2]) +
                                      (dv::mul_i24(Y & ~7, tstride) + ((X >> 3) << 6) + ((Y & 7) << 3) + (X & 7));
                    if (nvalid == 4) {
                        if (HBD) *reinterpret_cast<uint2 *>(td) = make_uint2(dv::pack2(o[0], o[1]), dv::pack2(o[2], o[3]));
                        else *reinterpret_cast<uint32_t *>(td) = (uint32_t) o[0] | ((uint32_t) o[1] << 8) | ((uint32_t) o[2] << 16) | ((uint32_t) o[3] << 24);
                    } else {
                        for (int x = 0; x < nvalid; x++) td[x] = (pixel) o[x];
                    }
                }
            } else {
                int16_t *d = prep + t.dst_off + (t.oy + vr) * t.bw + t.ox + 4 * vs;
                if (nvalid == 4) *reinterpret_cast<uint2 *>(d) = make_uint2(dv::pack2(o[0], o[1]), dv::pack2(o[2], o[3]));
                else if (nvalid > 0) for (int x = 0; x < nvalid; x++) d[x] = (int16_t) o[x];
            }
        }
    }
    DV_PHASE(PH + 7);
    DV_PHASE_WAVE(PH + 8);
}


} // namespace

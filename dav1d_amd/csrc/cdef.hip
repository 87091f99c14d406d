// Batched CDEF (constrained directional enhancement filter) for gfx950.
//
// Contract per task = what the reference driver dav1d_cdef_brow does for one 8x8 luma
// unit (src/cdef_apply_tmpl.c:149-290): dsp->cdef.dir on the luma block
// (cdef_find_dir_c, src/cdef_tmpl.c:239-319), adjust_strength (src/cdef_apply_tmpl.c:91-95),
// dsp->cdef.fb[0] on luma and dsp->cdef.fb[uv_idx] on both chroma blocks
// (cdef_filter_block_c, src/cdef_tmpl.c:103-237).
//
// The reference filters in place and therefore keeps backups of pre-filter rows / columns
// (cdef_line, lr_bak); here the filter is out of place: every unit reads its (w+4) x (h+4)
// neighbourhood from the immutable pre-CDEF picture `src` and writes `dst`, so all units of
// a frame are independent.  Mapping: one wave per unit, one lane per luma pixel; the window
// goes to LDS as int16 with the INT16_MIN sentinel where the frame edge cuts it off.
#include <type_traits>
#include "common.h"
#include "capi.h"
#include "cdef_rows.h"
#include "av1_tables.h"

namespace {

__device__ __forceinline__ int ulog2(unsigned v) { return 31 - __clz((int) v); }

// constrain() of src/cdef_tmpl.c:56-62 = clamp(diff, -lim, lim) with lim = max(0, threshold - (|diff| >> shift))
__device__ __forceinline__ int constrain(const int p, const int px, const int threshold, const int shift) {
    const int diff = p - px;
    const int adiff = dv::imax(diff, -diff);
    const int lim = dv::sub_floor0(threshold, adiff >> shift);
    return dv::med3(diff, -lim, lim);
}

// filters pixel (x, y) of a block whose padded window sits in tmp (12-wide rows, origin at tmp[2*12+2])
__device__ __forceinline__ int cdef_px(const int16_t *tmp, const int x, const int y, const int pri, const int sec,
                                       const int dir, const int damping, const int bitdepth_min_8)
{
    const int16_t *c = tmp + (y + 2) * 12 + x + 2;
    const int px = c[0];
    int sum = 0, mx = px;
    unsigned mn = (unsigned) px;
    const int8_t *dirs = &av1_cdef_directions[dir * 2];     // [12][2], entries dir .. dir+4 are used
    if (pri) {
        const int pri_tap = 4 - ((pri >> bitdepth_min_8) & 1);
        const int pri_shift = dv::imax(0, damping - ulog2(pri));
        int tap = pri_tap;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int off = dirs[2 * 2 + k];
            const int p0 = c[off], p1 = c[-off];
            sum += tap * constrain(p0, px, pri, pri_shift);
            sum += tap * constrain(p1, px, pri, pri_shift);
            tap = (tap & 3) | 2;
            mn = min(mn, (unsigned) p0); mx = dv::imax(mx, p0);
            mn = min(mn, (unsigned) p1); mx = dv::imax(mx, p1);
        }
    }
    if (sec) {
        const int sec_shift = damping - ulog2(sec);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int off2 = dirs[4 * 2 + k], off3 = dirs[0 * 2 + k];
            const int s0 = c[off2], s1 = c[-off2], s2 = c[off3], s3 = c[-off3];
            const int tap = 2 - k;
            sum += tap * constrain(s0, px, sec, sec_shift);
            sum += tap * constrain(s1, px, sec, sec_shift);
            sum += tap * constrain(s2, px, sec, sec_shift);
            sum += tap * constrain(s3, px, sec, sec_shift);
            mn = min(mn, (unsigned) s0); mx = dv::imax(mx, s0);
            mn = min(mn, (unsigned) s1); mx = dv::imax(mx, s1);
            mn = min(mn, (unsigned) s2); mx = dv::imax(mx, s2);
            mn = min(mn, (unsigned) s3); mx = dv::imax(mx, s3);
        }
    }
    const int v = px + ((sum - (sum < 0) + 8) >> 4);
    // only the primary+secondary path clamps to the local range (src/cdef_tmpl.c:165 vs :185,209)
    return (pri && sec) ? dv::iclip(v, (int) mn, mx) : v;
}

// The same pixel by TWO neighbouring lanes (lane ^ 1): lane `half` takes the taps of distance half + 1 (one primary pair, two
// secondary pairs: 6 of the 12), the partial sum and the range travel across the pair, both lanes end up with the result.
// Used where a pass would leave half the wave idle (the 32 chroma pixels of a 4:2:0 unit).
__device__ __forceinline__ int cdef_px_pair(const int16_t *tmp, const int x, const int y, const int half, const int pri, const int sec,
                                            const int dir, const int damping, const int bitdepth_min_8)
{
    const int16_t *c = tmp + (y + 2) * 12 + x + 2;
    const int px = c[0];
    int sum = 0, mx = px;
    unsigned mn = (unsigned) px;
    const int8_t *dirs = &av1_cdef_directions[dir * 2];
    if (pri) {
        const int pri_tap = 4 - ((pri >> bitdepth_min_8) & 1);
        const int pri_shift = dv::imax(0, damping - ulog2(pri));
        const int tap = half ? ((pri_tap & 3) | 2) : pri_tap;
        const int off = half ? dirs[2 * 2 + 1] : dirs[2 * 2 + 0];
        const int p0 = c[off], p1 = c[-off];
        sum += tap * constrain(p0, px, pri, pri_shift);
        sum += tap * constrain(p1, px, pri, pri_shift);
        mn = min(mn, (unsigned) p0); mx = dv::imax(mx, p0);
        mn = min(mn, (unsigned) p1); mx = dv::imax(mx, p1);
    }
    if (sec) {
        const int sec_shift = damping - ulog2(sec);
        const int off2 = half ? dirs[4 * 2 + 1] : dirs[4 * 2 + 0], off3 = half ? dirs[0 * 2 + 1] : dirs[0 * 2 + 0];
        const int s0 = c[off2], s1 = c[-off2], s2 = c[off3], s3 = c[-off3];
        const int tap = 2 - half;
        sum += tap * constrain(s0, px, sec, sec_shift);
        sum += tap * constrain(s1, px, sec, sec_shift);
        sum += tap * constrain(s2, px, sec, sec_shift);
        sum += tap * constrain(s3, px, sec, sec_shift);
        mn = min(mn, (unsigned) s0); mx = dv::imax(mx, s0);
        mn = min(mn, (unsigned) s1); mx = dv::imax(mx, s1);
        mn = min(mn, (unsigned) s2); mx = dv::imax(mx, s2);
        mn = min(mn, (unsigned) s3); mx = dv::imax(mx, s3);
    }
    sum += __shfl_xor(sum, 1);
    mn = min(mn, (unsigned) __shfl_xor((int) mn, 1));
    mx = dv::imax(mx, __shfl_xor(mx, 1));
    const int v = px + ((sum - (sum < 0) + 8) >> 4);
    return (pri && sec) ? dv::iclip(v, (int) mn, mx) : v;
}

// The (w+4) x (h+4) neighbourhood of a block goes to LDS in 12-wide rows (the layout cdef_px and the direction tables
// expect), INT16_MIN where `edges` says the neighbour does not exist.  WW = w + 4 is a compile-time constant, so the
// entry -> (row, column) split is a shift or a multiply, and only the entries of the window are visited: 64 for a 4x4
// block (one pass), 96 for 4x8, 144 for 8x8.
template <typename pixel, int WW>
__device__ __forceinline__ void load_window(int16_t *tmp, const pixel *src, const int stride, const int x0, const int y0,
                                            const int h, const int edges, const int lane, const bool rep_bot = false)
{
    constexpr int w = WW - 4;
    const int n = WW * (h + 4);
    for (int i = lane; i < n; i += 64) {
        const int yy = i / WW - 2, xx = i % WW - 2;
        const bool avail = (yy >= 0 || (edges & 4)) && (yy < h || (edges & 8)) && (xx >= 0 || (edges & 1)) && (xx < w || (edges & 2));
        int v = -32768;
        // rep_bot: the second row below the block is the first one once more (DAV1D_HIP_CDEF_BOT_REP_*)
        if (avail) v = src[(y0 + (rep_bot && yy == h + 1 ? h : yy)) * stride + x0 + xx];
        tmp[(yy + 2) * 12 + xx + 2] = (int16_t) v;
    }
}

// The same window when every neighbour exists and pixels are 16 bits wide: rows of WW * 2 bytes fetched as 8-byte pieces
// (4-byte aligned in memory: block origins are multiples of 4 pixels, the window starts 2 pixels to the left), one piece
// per lane, lanes [lane0, lane0 + (WW / 4) * (h + 4)).
struct __attribute__((packed, aligned(4))) CdefU64 { uint32_t a, b; };
template <int WW>
__device__ __forceinline__ void load_window_fast(int16_t *tmp, const uint16_t *src, const int stride, const int x0, const int y0,
                                                 const int h, const int item)
{
    constexpr int NP = WW / 4;
    if (item < 0 || item >= NP * (h + 4)) return;
    const int row = item / NP, part = item % NP;
    const CdefU64 v = *reinterpret_cast<const CdefU64 *>(src + (y0 - 2 + row) * stride + x0 - 2 + 4 * part);
    *reinterpret_cast<uint2 *>(tmp + row * 12 + 4 * part) = make_uint2(v.a, v.b);
}

template <typename pixel>
__global__ __launch_bounds__(64) void cdef_kernel(const DevPlanes dst, const DevPlanes src, const Dav1dHipCdefTask *__restrict__ tasks,
                                                  const int n, const int damping, const int layout, const int bitdepth_max,
                                                  uint32_t *__restrict__ dirvar, const int raw_only)
{
    __shared__ __attribute__((aligned(8))) int16_t tmp[144], tmp2[144];
    __shared__ int16_t pdir[64];
    constexpr bool HBD = sizeof(pixel) == 2;

    const int ti = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x);
    if (ti >= n) return;
    const int lane = threadIdx.x;
    const int tis = __builtin_amdgcn_readfirstlane(ti);
    const Dav1dHipCdefTask t = tasks[tis];
    if (raw_only && !(t.flags & 1)) return;
    const int bitdepth_min_8 = (32 - __clz(bitdepth_max)) - 8;
    const int edges = t.edges;

    // ---- luma window
    const bool raw_ = t.flags & 1;
    const int lpl = raw_ ? t.plane : 0;                  // raw calls may address any plane / 8x8, 4x8, 4x4 blocks
    const int lw = raw_ ? (t.flags & 2 ? 4 : 8) : 8, lh = raw_ ? (t.flags & 4 ? 4 : 8) : 8;
    const pixel *sy = reinterpret_cast<const pixel *>(src.data[lpl]);
    const int x0 = raw_ ? t.bx : t.bx * 8, y0 = raw_ ? t.by : t.by * 8;   // raw: pixel coordinates
    const bool rep_y = !raw_ && (t.flags & DAV1D_HIP_CDEF_BOT_REP_Y), rep_uv = !raw_ && (t.flags & DAV1D_HIP_CDEF_BOT_REP_UV);
    if (HBD && edges == 15 && lw == 8 && !rep_y) load_window_fast<12>(tmp, reinterpret_cast<const uint16_t *>(sy), src.stride[lpl], x0, y0, lh, lane);
    else if (lw == 8) load_window<pixel, 12>(tmp, sy, src.stride[lpl], x0, y0, lh, edges, lane, rep_y);
    else load_window<pixel, 8>(tmp, sy, src.stride[lpl], x0, y0, lh, edges, lane);
    dv::wave_sync();

    // ---- direction search (only when a primary strength is in play, src/cdef_apply_tmpl.c:208-212)
    // cdef_find_dir_c (src/cdef_tmpl.c:239-319) squares 90 partial sums of the block along 8 directions.  Here every
    // 16-lane row of the wave owns one of the eight partial-sum arrays (two passes of four), lane r of a row gathers
    // element r from the window in LDS (at most 8 steps of one or two pixels), squares and weighs it, and a butterfly
    // over the row adds the array's cost.  No LDS atomics, no serial tail.
    int dir = 0;
    unsigned var = 0;
    const bool raw = t.flags & 1;      // DSP-level call: explicit dir / strengths, luma path only, no adjust
    if (!raw && (t.y_pri || t.uv_pri || dirvar)) {
        // the samples the search sums, (px >> (bitdepth - 8)) - 128, once per pixel instead of once per gather
        {
            const int x = lane & 7, y = lane >> 3;
            pdir[lane] = (int16_t) ((tmp[(y + 2) * 12 + x + 2] >> bitdepth_min_8) - 128);
        }
        dv::wave_sync();
        const int r = lane & 15, g = lane >> 4;
        unsigned cost[8];
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            // pixel of step k: x = cx + sx * k + hx * (k >> 1), y = k (hv[0]: y = r); `two`: also x + 1
            //   pass 0: diag[0] (idx y + x), diag[1] (7 + y - x), alt[0] (y + (x >> 1)), alt[1] (3 + y - (x >> 1))
            //   pass 1: alt[2] (3 - (y >> 1) + x), alt[3] ((y >> 1) + x), hv[0] (y), hv[1] (x)
            int cx, sx, hx;
            bool two = false, rows = false;
            if (pass == 0) {
                cx = g == 0 ? r : g == 1 ? 7 - r : g == 2 ? 2 * r : 6 - 2 * r;
                sx = g == 0 ? -1 : g == 1 ? 1 : g == 2 ? -2 : 2;
                hx = 0;
                two = g >= 2;
            } else {
                cx = g == 0 ? r - 3 : g == 1 ? r : g == 2 ? 0 : r;
                sx = g == 2 ? 1 : 0;
                hx = g == 0 ? 1 : g == 1 ? -1 : 0;
                rows = g == 2;
            }
            int p = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int x = cx + sx * k + hx * (k >> 1), y = rows ? r : k;
                if ((unsigned) x < 8u && y < 8) {
                    const int16_t *c = pdir + y * 8 + x;
                    p += c[0];
                    if (two) p += c[1];
                }
            }
            // weights: 105 for the full-length lines, 840 / (k + 1) for the short ones (div_table, src/cdef_tmpl.c:279)
            const bool is_diag = pass == 0 && g < 2, is_hv = pass == 1 && g >= 2;
            int short_k = -1;                                   // index into div_table, -1 = weight 105
            if (is_diag) short_k = r < 7 ? r : r > 7 ? 14 - r : -1;
            else if (!is_hv) short_k = r < 3 ? 2 * r + 1 : r > 7 ? 2 * (10 - r) + 1 : -1;
            const unsigned long long div_lo = 840ull | 420ull << 16 | 280ull << 32 | 210ull << 48, div_hi = 168ull | 140ull << 16 | 120ull << 32;
            const unsigned wgt = short_k < 0 ? 105u : (unsigned) ((short_k < 4 ? div_lo >> (16 * short_k) : div_hi >> (16 * (short_k - 4))) & 0xffff);
            unsigned c = (unsigned) (p * p) * wgt;
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) c += (unsigned) __shfl_xor((int) c, m);
            // direction numbering: diag[0] 0, alt[0] 1, hv[0] 2, alt[1] 3, diag[1] 4, alt[2] 5, hv[1] 6, alt[3] 7
            const unsigned c0 = (unsigned) __shfl((int) c, 0), c1 = (unsigned) __shfl((int) c, 16);
            const unsigned c2 = (unsigned) __shfl((int) c, 32), c3 = (unsigned) __shfl((int) c, 48);
            if (pass == 0) { cost[0] = c0; cost[4] = c1; cost[1] = c2; cost[3] = c3; }
            else           { cost[5] = c0; cost[7] = c1; cost[2] = c2; cost[6] = c3; }
        }
        unsigned best = cost[0];
#pragma unroll
        for (int k = 1; k < 8; k++) if (cost[k] > best) { best = cost[k]; dir = k; }
        unsigned opp = cost[4];                                 // cost[dir ^ 4], picked by compares (never indexed)
#pragma unroll
        for (int k = 1; k < 8; k++) opp = dir == k ? cost[k ^ 4] : opp;
        var = (best - opp) >> 10;
        if (dirvar && lane == 0) dirvar[tis] = (uint32_t) dir | (var << 3);
    }

    // ---- luma filter
    {
        int pri = 0, sec = t.y_sec, d = 0;
        bool run = false;
        if (raw) {
            pri = t.y_pri; d = t.dir; run = pri || sec;
        } else if (t.y_pri) {
            // adjust_strength, src/cdef_apply_tmpl.c:91-95
            int adj = 0;
            if (var) {
                const int i = (var >> 6) ? dv::imin(ulog2(var >> 6), 12) : 0;
                adj = (t.y_pri * (4 + i) + 8) >> 4;
            }
            pri = adj; d = dir;
            run = adj || sec;
        } else if (sec) {
            run = true;
        }
        const int x = lane & (lw - 1), y = lane >> (lw == 4 ? 2 : 3);     // lw is 4 or 8
        if (run && y < lh) {
            const int v = cdef_px(tmp, x, y, pri, sec, d, damping, bitdepth_min_8);
            reinterpret_cast<pixel *>(dst.data[lpl])[(y0 + y) * dst.stride[lpl] + x0 + x] = (pixel) v;
        }
    }

    // ---- chroma: both planes in one pass, U on the first w*h lanes, V on the next w*h (4:4:4: two passes of 64)
    if (!raw && (t.uv_pri || t.uv_sec) && layout != DAV1D_HIP_LAYOUT_I400) {
        const int ss_ver = layout == DAV1D_HIP_LAYOUT_I420, ss_hor = layout != DAV1D_HIP_LAYOUT_I444;
        const int w = 8 >> ss_hor, h = 8 >> ss_ver;
        // 4:2:2 remaps the direction (src/cdef_apply_tmpl.c:115-117)
        const unsigned uv422 = 0x66654207u;   // nibbles 7,0,2,4,5,6,6,6 for dir 0..7
        int uvdir = 0;
        if (t.uv_pri) uvdir = layout == DAV1D_HIP_LAYOUT_I422 ? (int) ((uv422 >> (4 * dir)) & 15) : dir;
        const int cx0 = x0 >> ss_hor, cy0 = y0 >> ss_ver;
        dv::wave_sync();
        if (HBD && edges == 15 && w == 4 && !rep_uv) {
            // 4-wide chroma: 2 pieces x (h + 4) rows per plane, U on lanes 0 .. 31, V on lanes 32 .. 63
            load_window_fast<8>(lane < 32 ? tmp : tmp2, reinterpret_cast<const uint16_t *>(src.data[lane < 32 ? 1 : 2]),
                                src.stride[lane < 32 ? 1 : 2], cx0, cy0, h, lane & 31);
        } else if (w == 8) {
            load_window<pixel, 12>(tmp, reinterpret_cast<const pixel *>(src.data[1]), src.stride[1], cx0, cy0, h, edges, lane, rep_uv);
            load_window<pixel, 12>(tmp2, reinterpret_cast<const pixel *>(src.data[2]), src.stride[2], cx0, cy0, h, edges, lane, rep_uv);
        } else {
            load_window<pixel, 8>(tmp, reinterpret_cast<const pixel *>(src.data[1]), src.stride[1], cx0, cy0, h, edges, lane, rep_uv);
            load_window<pixel, 8>(tmp2, reinterpret_cast<const pixel *>(src.data[2]), src.stride[2], cx0, cy0, h, edges, lane, rep_uv);
        }
        dv::wave_sync();
        const int npx = w * h;                        // 16, 32 or 64 pixels per plane
        if (2 * npx == 32) {
            // 4:2:0: 32 chroma pixels, two lanes each (see cdef_px_pair) — one full-width pass instead of a half-empty one
            const int i = lane >> 1, half = lane & 1;
            const int pl = 1 + (i >= 16), k = i & 15;
            const int x = k & 3, y = k >> 2;
            const int v = cdef_px_pair(pl == 1 ? tmp : tmp2, x, y, half, t.uv_pri, t.uv_sec, uvdir, damping - 1, bitdepth_min_8);
            if (!half) reinterpret_cast<pixel *>(dst.data[pl])[(cy0 + y) * dst.stride[pl] + cx0 + x] = (pixel) v;
        } else
        for (int i = lane; i < 2 * npx; i += 64) {
            const int pl = 1 + (i >= npx), k = i - (pl - 1) * npx;
            const int x = k & (w - 1), y = k >> (3 - ss_hor);
            const int v = cdef_px(pl == 1 ? tmp : tmp2, x, y, t.uv_pri, t.uv_sec, uvdir, damping - 1, bitdepth_min_8);
            reinterpret_cast<pixel *>(dst.data[pl])[(cy0 + y) * dst.stride[pl] + cx0 + x] = (pixel) v;
        }
    }
}


// =====================================================================================================================
// Strip kernel: one wave = up to 16 horizontally adjacent 8x8 units of one unit row (a CdefGroup), every plane.
//
// * One window per plane in LDS shared by the 16 units (128 + 2 * 8 columns x (8 + 4) rows for luma), loaded as aligned
//   W-pixel pieces; a piece is either inside the picture or — where the group's edge flags say so — the INT16_MIN sentinel.
// * Direction search (cdef_find_dir_c, src/cdef_tmpl.c:239-319) for the 16 units at once: the 90 partial sums of a unit are
//   a 0/1 incidence matrix (which pixel feeds which line of which direction) times its 64 samples, and the samples
//   (px >> (bitdepth - 8)) - 128 are int8, so seven v_mfma_i32_16x16x64_i8 (one per 16-line tile: two diagonals, four
//   "alt" families, rows + columns) give all 16 x 90 sums exactly; squaring, weighting by 840 / line length and the
//   arg-max stay on the VALU, four lines of one unit per lane.
// * Filter (cdef_filter_block_c, src/cdef_tmpl.c:103-237) on pixel PAIRS in packed 16-bit arithmetic (the sums fit: |sum|
//   <= 12 * 240 + 12 * 64): lane l filters rows [q * H / 4, (q + 1) * H / 4) of unit u = l >> 2, q = l & 3; strengths,
//   direction and shifts are per-lane values, so units with different parameters share the wave.
// =====================================================================================================================

struct alignas(16) CdefDirTab {
    uint8_t m[7][16][64];       // [tile][line][pixel y * 8 + x]
    uint16_t w[7][16];          // 840 / (pixels on the line), 0 for the unused lines of a tile
};
// line of pixel (x, y) in tile t: the index expressions of cdef_find_dir_c (src/cdef_tmpl.c:255-270)
constexpr int cdef_line(const int t, const int x, const int y) {
    return t == 0 ? y + x : t == 1 ? 7 + y - x : t == 2 ? y + (x >> 1) : t == 3 ? 3 + y - (x >> 1) :
           t == 4 ? 3 - (y >> 1) + x : t == 5 ? (y >> 1) + x : y;
}
constexpr CdefDirTab make_cdef_dir_tab() {
    CdefDirTab d = {};
    for (int t = 0; t < 7; t++) {
        int cnt[16] = {};
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) {
                const int i = cdef_line(t, x, y);
                d.m[t][i][y * 8 + x] = 1; cnt[i]++;
                if (t == 6) { d.m[t][8 + x][y * 8 + x] = 1; cnt[8 + x]++; }      // tile 6: rows 0..7 = hv[0], 8..15 = hv[1]
            }
        for (int i = 0; i < 16; i++) d.w[t][i] = cnt[i] ? (uint16_t) (840 / cnt[i]) : 0;
    }
    return d;
}
__device__ const CdefDirTab cdef_dir_tab = make_cdef_dir_tab();
// (dy, dx) of the two taps of a direction, from the 12-wide offsets of av1_cdef_directions (entries 2 .. 9 are directions 0 .. 7)
constexpr uint32_t cdef_dir_yx(const int dir) {
    uint32_t v = 0;
    for (int k = 0; k < 2; k++) {
        const int off = av1_cdef_directions[(dir + 2) * 2 + k];
        const int dy = (off + 6 + 120) / 12 - 10, dx = off - 12 * dy;
        v |= (uint32_t) ((dy & 0xff) | (dx & 0xff) << 8) << (16 * k);
    }
    return v;
}
__device__ const uint32_t cdef_dir_yx_tab[8] = { cdef_dir_yx(0), cdef_dir_yx(1), cdef_dir_yx(2), cdef_dir_yx(3),
                                                 cdef_dir_yx(4), cdef_dir_yx(5), cdef_dir_yx(6), cdef_dir_yx(7) };

template <typename pixel, int W>
__device__ __forceinline__ void cdef_load_piece(uint16_t *dstp, const pixel *srcp, const bool ok) {
    constexpr bool HBD = sizeof(pixel) == 2;
    if (W == 8) {
        uint4 v = make_uint4(0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u);
        if (ok) {
            if (HBD) v = *reinterpret_cast<const uint4 *>(srcp);
            else {
                const uint2 b = *reinterpret_cast<const uint2 *>(srcp);
                v = make_uint4((b.x & 0xff) | (b.x & 0xff00) << 8, (b.x >> 16 & 0xff) | (b.x >> 24) << 16,
                               (b.y & 0xff) | (b.y & 0xff00) << 8, (b.y >> 16 & 0xff) | (b.y >> 24) << 16);
            }
        }
        *reinterpret_cast<uint4 *>(dstp) = v;
    } else {
        uint2 v = make_uint2(0x80008000u, 0x80008000u);
        if (ok) {
            if (HBD) v = *reinterpret_cast<const uint2 *>(srcp);
            else {
                const uint32_t b = *reinterpret_cast<const uint32_t *>(srcp);
                v = make_uint2((b & 0xff) | (b & 0xff00) << 8, (b >> 16 & 0xff) | (b >> 24) << 16);
            }
        }
        *reinterpret_cast<uint2 *>(dstp) = v;
    }
}

// Window of one plane: rows y0 - 2 .. y0 + H + 1, 18 pieces of W pixels per row (piece p = picture columns x0 + (p - 1) * W ..),
// of which 0 .. span + 1 are filled.  Unit u's pixels start at window column (u + 1) * W.  Lanes 0 .. 53 take three rows of 18
// pieces per pass: a lane keeps its piece and its row within the pass, only the row advances.
template <typename pixel, int W, int H>
__device__ __forceinline__ void cdef_load_window(uint16_t *win, const pixel *src, const int stride, const int x0, const int y0,
                                                 const int span, const int edges, const int lane, const bool rep_bot)
{
    constexpr int NPR = 18;
    const int r3 = lane / NPR, p = lane - r3 * NPR;
    if (lane >= 3 * NPR || p > span + 1) return;
    const bool col_ok = (p > 0 || (edges & 1)) && (p <= span || (edges & 2));
    const pixel *sp = src + (y0 + r3 - 2) * stride + x0 + (p - 1) * W;
    uint16_t *wp = win + (r3 * NPR + p) * W;
#pragma unroll
    for (int it = 0; it < (H + 4 + 2) / 3; it++) {
        const int row = it * 3 + r3;
        if ((H + 4) % 3 && row >= H + 4) break;
        const bool ok = col_ok && (row >= 2 || (edges & 4)) && (row < H + 2 || (edges & 8));
        // rep_bot: the window's last row (the second below the units) is the one before it once more (DAV1D_HIP_CDEF_BOT_REP_*)
        cdef_load_piece<pixel, W>(wp + it * 3 * NPR * W, sp + (it * 3 - (rep_bot && row == H + 3)) * stride, ok);
    }
}

struct CdefTapSet { uint32_t thr2, sh2, tap2[2]; int off[2]; };      // one strength: threshold, shift, the two tap weights and offsets

// 12 taps on RPL rows of W pixels (W / 2 packed pairs per row); `base` = window index (pixels) of the lane's first pixel.
// Taps outside, rows inside: the address and the odd-column shift of a tap are worked out once for all rows.
template <int W, int RPL>
__device__ __forceinline__ void cdef_filter_rows(const uint16_t *win, const int base, const bool any_pri, const bool any_sec,
                                                 const CdefTapSet pri, const CdefTapSet sec0, const CdefTapSet sec1,
                                                 const bool clamp_range, uint32_t (&out)[RPL][W / 2])
{
    constexpr int NP = W / 2, WS = 18 * W;
    const uint32_t *w32 = reinterpret_cast<const uint32_t *>(win);
    uint32_t px[RPL][NP], sum[RPL][NP], mn[RPL][NP], mx[RPL][NP];
#pragma unroll
    for (int r = 0; r < RPL; r++)
#pragma unroll
        for (int j = 0; j < NP; j++) { px[r][j] = w32[(base >> 1) + r * (WS / 2) + j]; sum[r][j] = 0; mn[r][j] = mx[r][j] = px[r][j]; }
    auto tap = [&](const CdefTapSet &ts, const int k, const int idx) {
        const uint32_t *q = w32 + (idx >> 1);
        const uint32_t shv = (uint32_t) (idx & 1) << 4;
#pragma unroll
        for (int r = 0; r < RPL; r++) {
            uint32_t d[NP + 1];
#pragma unroll
            for (int j = 0; j <= NP; j++) d[j] = q[r * (WS / 2) + j];
#pragma unroll
            for (int j = 0; j < NP; j++) {
                const uint32_t p = __builtin_amdgcn_alignbit(d[j + 1], d[j], shv);
                // constrain(), src/cdef_tmpl.c:56-62, two pixels at a time
                const uint32_t d1 = dv::pk_sub(p, px[r][j]), d2 = dv::pk_sub(px[r][j], p);
                const uint32_t ad = dv::pk_max_i16(d1, d2);
                const uint32_t lim = dv::pk_sub_u16_sat(ts.thr2, dv::pk_lshr(ad, ts.sh2));
                const uint32_t c = dv::pk_min_i16(dv::pk_max_i16(d1, dv::pk_sub(0u, lim)), lim);
                sum[r][j] = dv::pk_mad(c, ts.tap2[k], sum[r][j]);
                mn[r][j] = dv::pk_min_u16(mn[r][j], p);
                mx[r][j] = dv::pk_max_i16(mx[r][j], p);
            }
        }
    };
    if (any_pri) { tap(pri, 0, base + pri.off[0]); tap(pri, 0, base - pri.off[0]); tap(pri, 1, base + pri.off[1]); tap(pri, 1, base - pri.off[1]); }
    if (any_sec) {
        tap(sec0, 0, base + sec0.off[0]); tap(sec0, 0, base - sec0.off[0]); tap(sec1, 0, base + sec1.off[0]); tap(sec1, 0, base - sec1.off[0]);
        tap(sec0, 1, base + sec0.off[1]); tap(sec0, 1, base - sec0.off[1]); tap(sec1, 1, base + sec1.off[1]); tap(sec1, 1, base - sec1.off[1]);
    }
#pragma unroll
    for (int r = 0; r < RPL; r++)
#pragma unroll
        for (int j = 0; j < NP; j++) {
            // px + ((sum - (sum < 0) + 8) >> 4)
            uint32_t s = dv::pk_add(sum[r][j], dv::pk_ashr(sum[r][j], dv::rep2(15)));
            s = dv::pk_ashr(dv::pk_add(s, dv::rep2(8)), dv::rep2(4));
            const uint32_t v = dv::pk_add(px[r][j], s);
            // only the primary + secondary path clamps to the local range (src/cdef_tmpl.c:165 vs :185,209)
            out[r][j] = clamp_range ? dv::pk_min_i16(dv::pk_max_i16(v, mn[r][j]), mx[r][j]) : v;
        }
}

template <int WS>
__device__ __forceinline__ CdefTapSet cdef_tapset(const int strength, const int shift, const int tap0, const int tap1, const uint32_t yx) {
    CdefTapSet t;
    t.thr2 = dv::rep2(strength); t.sh2 = dv::rep2(shift); t.tap2[0] = dv::rep2(tap0); t.tap2[1] = dv::rep2(tap1);
#pragma unroll
    for (int k = 0; k < 2; k++) t.off[k] = (int) (int8_t) (yx >> (16 * k)) * WS + (int) (int8_t) (yx >> (16 * k + 8));
    return t;
}

// filter + store of one plane for the lane's (unit u, quarter q)
template <typename pixel, int W, int H>
__device__ __forceinline__ void cdef_plane(const uint16_t *win, const uint32_t *dir_yx, pixel *dstp, const int stride, const int x0, const int y0,
                                           const int u, const int q, const int pri, const int sec, const int dir, const int damping,
                                           const int bitdepth_min_8, const bool run, const bool present)
{
    constexpr int RPL = H / 4, NP = W / 2, WS = 18 * W;
    const bool any_pri = __any(run && pri), any_sec = __any(run && sec);
    const int pri_tap = 4 - ((pri >> bitdepth_min_8) & 1);
    const CdefTapSet tp = cdef_tapset<WS>(pri, pri ? dv::imax(0, damping - ulog2(pri)) : 0, pri_tap, (pri_tap & 3) | 2, dir_yx[dir]);
    const int sec_shift = sec ? damping - ulog2(sec) : 0;
    const CdefTapSet ts0 = cdef_tapset<WS>(sec, sec_shift, 2, 1, dir_yx[(dir + 2) & 7]);
    const CdefTapSet ts1 = cdef_tapset<WS>(sec, sec_shift, 2, 1, dir_yx[(dir + 6) & 7]);
    uint32_t out[RPL][NP];
    const int row0 = q * RPL;
    cdef_filter_rows<W, RPL>(win, (row0 + 2) * WS + (u + 1) * W, any_pri, any_sec, tp, ts0, ts1, pri && sec, out);
    // a listed unit that does not filter this plane has both strengths 0 here: its sums are 0 and `out` is the window's centre, so
    // every listed unit is WRITTEN and dst needs no copy of src underneath (frame.hip fills the unlisted units, cdef_fill_unlisted)
    if (!present) return;
#pragma unroll
    for (int r = 0; r < RPL; r++) {
        pixel *d = dstp + (y0 + row0 + r) * stride + x0 + u * W;
        if (sizeof(pixel) == 2) {
            if (W == 8) *reinterpret_cast<uint4 *>(d) = make_uint4(out[r][0], out[r][1], out[r][NP - 2], out[r][NP - 1]);
            else *reinterpret_cast<uint2 *>(d) = make_uint2(out[r][0], out[r][1]);
        } else {
            // low bytes of the four halves of two registers
            const uint32_t lo = __builtin_amdgcn_perm(out[r][1], out[r][0], 0x06040200u);
            if (W == 8) *reinterpret_cast<uint2 *>(d) = make_uint2(lo, __builtin_amdgcn_perm(out[r][NP - 1], out[r][NP - 2], 0x06040200u));
            else *reinterpret_cast<uint32_t *>(d) = lo;
        }
    }
}

// CW x CH = chroma unit (8x8 4:4:4, 4x8 4:2:2, 4x4 4:2:0; CW = 0: no chroma).  The three windows have LDS of their own, so
// every load of the wave is in flight before the first wait.
template <typename pixel, int CW, int CH>
__global__ __launch_bounds__(64) void cdef_strip_kernel(const DevPlanes dst, const DevPlanes src, const Dav1dHipCdefTask *__restrict__ tasks,
                                                        const CdefGroup *__restrict__ groups, const int n_groups, const int damping,
                                                        const int layout, const int bitdepth_max, uint32_t *__restrict__ dirvar)
{
    constexpr int CWIN = CW ? (CH + 4) * 18 * CW : 8;
    __shared__ __attribute__((aligned(16))) uint16_t win[12 * 18 * 8], cwin[2][CWIN];
    __shared__ uint32_t traw[16][2], upar[16][2], dir_yx[8];

    const int gi = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x);
    if (gi >= n_groups) return;
    const int lane = threadIdx.x;
    const CdefGroup g = groups[__builtin_amdgcn_readfirstlane(gi)];
    if (!g.n) return;                 // (the slots of cdef_expand_kernel that hold no unit)
    const int bitdepth_min_8 = (32 - __clz(bitdepth_max)) - 8;
    const int edges = g.edges, span = g.span;
    const int x0 = g.bx0 * 8, y0 = g.by * 8;

    if (lane < 16) { traw[lane][0] = 0; traw[lane][1] = 0; }
    if (lane < 8) dir_yx[lane] = cdef_dir_yx_tab[lane];
    dv::wave_sync();
    bool uv = false;
    if (lane < g.n) {
        const Dav1dHipCdefTask t = tasks[g.first + lane];
        const int slot = t.bx - g.bx0;
        traw[slot][0] = (uint32_t) t.y_pri | (uint32_t) t.y_sec << 8 | (uint32_t) t.uv_pri << 16 | (uint32_t) t.uv_sec << 24;
        traw[slot][1] = 0x100u | (uint32_t) lane;                       // present, rank in the group
        uv = t.uv_pri || t.uv_sec;
    }
    const bool any_uv = CW && __any(uv);
    constexpr int ss_hor = CW == 4, ss_ver = CH == 4;
    const int cx0 = x0 >> ss_hor, cy0 = y0 >> ss_ver;
    const bool rep_y = g.pad & DAV1D_HIP_CDEF_BOT_REP_Y, rep_uv = g.pad & DAV1D_HIP_CDEF_BOT_REP_UV;      // (the group carries its units' flags)
    cdef_load_window<pixel, 8, 8>(win, reinterpret_cast<const pixel *>(src.data[0]), src.stride[0], x0, y0, span, edges, lane, rep_y);
    if (CW && any_uv) {
        cdef_load_window<pixel, CW ? CW : 8, CW ? CH : 8>(cwin[0], reinterpret_cast<const pixel *>(src.data[1]), src.stride[1], cx0, cy0, span, edges, lane, rep_uv);
        cdef_load_window<pixel, CW ? CW : 8, CW ? CH : 8>(cwin[1], reinterpret_cast<const pixel *>(src.data[2]), src.stride[2], cx0, cy0, span, edges, lane, rep_uv);
    }
    dv::wave_sync();

    // ---- direction search for unit n = lane & 15 (every lane of the four 16-lane rows ends up with the unit's eight costs)
    {
        const int n = lane & 15, gq = lane >> 4;
        const uint32_t raw0 = traw[n][0], raw1 = traw[n][1];
        const int y_pri = raw0 & 0xff, y_sec = raw0 >> 8 & 0xff, uv_pri = raw0 >> 16 & 0xff, uv_sec = raw0 >> 24;
        const bool present = raw1 >> 8 & 1;
        int dir = 0;
        unsigned var = 0;
        if (__any(present && (y_pri || uv_pri)) || dirvar) {
            // B operand: samples 16 * gq .. + 15 of unit n = its rows 2 * gq, 2 * gq + 1, as int8
            uint32_t bq[4];
            const uint32_t sh2 = dv::rep2(bitdepth_min_8);
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const uint4 v = *reinterpret_cast<const uint4 *>(win + (2 * gq + r + 2) * 144 + (n + 1) * 8);
                bq[2 * r + 0] = __builtin_amdgcn_perm(dv::pk_lshr(v.y, sh2), dv::pk_lshr(v.x, sh2), 0x06040200u) ^ 0x80808080u;
                bq[2 * r + 1] = __builtin_amdgcn_perm(dv::pk_lshr(v.w, sh2), dv::pk_lshr(v.z, sh2), 0x06040200u) ^ 0x80808080u;
            }
            const uint4 b = make_uint4(bq[0], bq[1], bq[2], bq[3]);
            unsigned cost[8];
#pragma unroll
            for (int t = 0; t < 7; t++) {
                const uint4 a = *reinterpret_cast<const uint4 *>(&cdef_dir_tab.m[t][n][16 * gq]);
                int acc[4] = { 0, 0, 0, 0 };
                dv::mfma_i32_16x16x64_i8(a, b, acc);
                const uint2 wq = *reinterpret_cast<const uint2 *>(&cdef_dir_tab.w[t][4 * gq]);
                // |sum| <= 16 * 128 and weight <= 840: 24-bit multiplies (full rate) are exact here
                unsigned c = dv::mul_u24((unsigned) dv::mul_i24(acc[0], acc[0]), wq.x & 0xffff) + dv::mul_u24((unsigned) dv::mul_i24(acc[1], acc[1]), wq.x >> 16) +
                             dv::mul_u24((unsigned) dv::mul_i24(acc[2], acc[2]), wq.y & 0xffff) + dv::mul_u24((unsigned) dv::mul_i24(acc[3], acc[3]), wq.y >> 16);
                c += (unsigned) __shfl_xor((int) c, 16);
                if (t < 6) {
                    c += (unsigned) __shfl_xor((int) c, 32);
                    // direction numbering: diag[0] 0, alt[0] 1, hv[0] 2, alt[1] 3, diag[1] 4, alt[2] 5, hv[1] 6, alt[3] 7
                    cost[t == 0 ? 0 : t == 1 ? 4 : t == 2 ? 1 : t == 3 ? 3 : t == 4 ? 5 : 7] = c;
                } else {
                    const unsigned o = (unsigned) __shfl_xor((int) c, 32);
                    cost[2] = gq < 2 ? c : o;
                    cost[6] = gq < 2 ? o : c;
                }
            }
            unsigned best = cost[0];
#pragma unroll
            for (int k = 1; k < 8; k++) if (cost[k] > best) { best = cost[k]; dir = k; }
            unsigned opp = cost[4];                                 // cost[dir ^ 4], picked by compares (never indexed)
#pragma unroll
            for (int k = 1; k < 8; k++) opp = dir == k ? cost[k ^ 4] : opp;
            var = (best - opp) >> 10;
            if (dirvar && lane < 16 && present) dirvar[g.first + (raw1 & 0xff)] = (uint32_t) dir | (var << 3);
        }
        if (lane < 16) {
            // luma: adjust_strength, src/cdef_apply_tmpl.c:91-95; direction 0 without a primary strength (:218-230)
            int pri = 0, d = 0;
            bool run = false;
            if (y_pri) {
                if (var) {
                    const int i = (var >> 6) ? dv::imin(ulog2(var >> 6), 12) : 0;
                    pri = (y_pri * (4 + i) + 8) >> 4;
                }
                d = dir;
                run = pri || y_sec;
            } else if (y_sec) run = true;
            // chroma: 4:2:2 remaps the direction (src/cdef_apply_tmpl.c:115-117)
            const unsigned uv422 = 0x66654207u;   // nibbles 7,0,2,4,5,6,6,6 for dir 0..7
            int uvdir = 0;
            if (uv_pri) uvdir = layout == DAV1D_HIP_LAYOUT_I422 ? (int) ((uv422 >> (4 * dir)) & 15) : dir;
            const uint32_t pb = (uint32_t) present << 25;
            upar[n][0] = (uint32_t) (run ? pri : 0) | (uint32_t) (run ? y_sec : 0) << 8 | (uint32_t) d << 16 | (uint32_t) (present && run) << 24 | pb;
            upar[n][1] = (uint32_t) uv_pri | (uint32_t) uv_sec << 8 | (uint32_t) uvdir << 16 | (uint32_t) (present && (uv_pri || uv_sec)) << 24 | pb;
        }
    }
    dv::wave_sync();

    const int u = lane >> 2, q = lane & 3;
    const uint32_t p0 = upar[u][0], p1 = upar[u][1];
    cdef_plane<pixel, 8, 8>(win, dir_yx, reinterpret_cast<pixel *>(dst.data[0]), dst.stride[0], x0, y0, u, q, p0 & 0xff, p0 >> 8 & 0xff,
                            p0 >> 16 & 0xff, damping, bitdepth_min_8, p0 >> 24 & 1, p0 >> 25 & 1);
    if (!CW) return;
    if (!any_uv) {
        // no unit of the strip filters chroma (no window was loaded): the listed units' chroma blocks go across as they are
        if (!(p0 >> 25 & 1)) return;
        constexpr int CWc = CW ? CW : 8, RPL = (CW ? CH : 8) / 4;
        typedef typename std::conditional<CWc * sizeof(pixel) == 16, uint4, typename std::conditional<CWc * sizeof(pixel) == 8, uint2, uint32_t>::type>::type piece;
#pragma unroll
        for (int pl = 1; pl < 3; pl++)
#pragma unroll
            for (int r = 0; r < RPL; r++) {
                const ptrdiff_t so = (ptrdiff_t) (cy0 + q * RPL + r) * src.stride[pl] + cx0 + u * CWc, dof = (ptrdiff_t) (cy0 + q * RPL + r) * dst.stride[pl] + cx0 + u * CWc;
                *reinterpret_cast<piece *>(reinterpret_cast<pixel *>(dst.data[pl]) + dof) = *reinterpret_cast<const piece *>(reinterpret_cast<const pixel *>(src.data[pl]) + so);
            }
        return;
    }
    const int pri = p1 & 0xff, sec = p1 >> 8 & 0xff, d = p1 >> 16 & 0xff;
    const bool run = p1 >> 24 & 1;
#pragma unroll
    for (int pl = 1; pl < 3; pl++)
        cdef_plane<pixel, CW ? CW : 8, CW ? CH : 8>(cwin[pl - 1], dir_yx, reinterpret_cast<pixel *>(dst.data[pl]), dst.stride[pl], cx0, cy0, u, q,
                                                      pri, sec, d, damping - 1, bitdepth_min_8, run, p1 >> 25 & 1);
}

// The units CDEF does NOT list (skipped blocks, zero strengths) keep the deblocked pixels: instead of copying the whole picture under
// the filtered units (100 MB for an 8K frame), the listed units are marked in a bitmap of 8x8 units and only the others are copied.
__global__ __launch_bounds__(256) void cdef_mark_kernel(const Dav1dHipCdefTask *__restrict__ tasks, const int n, uint32_t *__restrict__ bitmap, const int w8) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Dav1dHipCdefTask t = tasks[i];
    if (t.flags & DAV1D_HIP_CDEF_RAW) return;
    const int idx = t.by * w8 + t.bx;
    atomicOr(&bitmap[idx >> 5], 1u << (idx & 31));
}

// 256 threads = 32 units of a unit row x 8 pixel rows
template <typename pixel>
__global__ __launch_bounds__(256) void cdef_fill_unlisted_kernel(const DevPlanes dst, const DevPlanes src, const uint32_t *__restrict__ bitmap,
                                                                 const int w8, const int layout)
{
    const int bx = blockIdx.x * 32 + (threadIdx.x & 31), by = blockIdx.y, r = threadIdx.x >> 5;
    if (bx >= w8) return;
    const int idx = by * w8 + bx;
    if (bitmap[idx >> 5] >> (idx & 31) & 1) return;
    typedef typename std::conditional<sizeof(pixel) == 2, uint4, uint2>::type row8;
    typedef typename std::conditional<sizeof(pixel) == 2, uint2, uint32_t>::type row4;
    {
        const ptrdiff_t so = (ptrdiff_t) (by * 8 + r) * src.stride[0] + bx * 8, dof = (ptrdiff_t) (by * 8 + r) * dst.stride[0] + bx * 8;
        *reinterpret_cast<row8 *>(reinterpret_cast<pixel *>(dst.data[0]) + dof) = *reinterpret_cast<const row8 *>(reinterpret_cast<const pixel *>(src.data[0]) + so);
    }
    if (layout == DAV1D_HIP_LAYOUT_I400) return;
    const int ss_hor = layout != DAV1D_HIP_LAYOUT_I444, ss_ver = layout == DAV1D_HIP_LAYOUT_I420;
    if (r >= 8 >> ss_ver) return;
#pragma unroll
    for (int pl = 1; pl < 3; pl++) {
        const int y = by * (8 >> ss_ver) + r, x = bx * (8 >> ss_hor);
        const pixel *s = reinterpret_cast<const pixel *>(src.data[pl]) + (ptrdiff_t) y * src.stride[pl] + x;
        pixel *d = reinterpret_cast<pixel *>(dst.data[pl]) + (ptrdiff_t) y * dst.stride[pl] + x;
        if (ss_hor) *reinterpret_cast<row4 *>(d) = *reinterpret_cast<const row4 *>(s);
        else *reinterpret_cast<row8 *>(d) = *reinterpret_cast<const row8 *>(s);
    }
}

} // namespace

extern "C" int dav1d_hip_launch_cdef(const DevPlanes *dst, const DevPlanes *src, int bpc, int layout,
                                     const Dav1dHipCdefTask *tasks, int n, int damping, uint32_t *dirvar, int raw_only, void *stream)
{
    if (n <= 0) return 0;
    const int bitdepth_max = (1 << bpc) - 1;
    if (bpc == 8)
        hipLaunchKernelGGL((cdef_kernel<uint8_t>), dim3(n), dim3(64), 0, (hipStream_t) stream, *dst, *src, tasks, n, damping, layout, bitdepth_max, dirvar, raw_only);
    else
        hipLaunchKernelGGL((cdef_kernel<uint16_t>), dim3(n), dim3(64), 0, (hipStream_t) stream, *dst, *src, tasks, n, damping, layout, bitdepth_max, dirvar, raw_only);
    return hip_rc(hipGetLastError());
}

template <typename pixel>
static void launch_strips(const DevPlanes *dst, const DevPlanes *src, int layout, const Dav1dHipCdefTask *tasks, const CdefGroup *groups,
                          int n_groups, int damping, int bitdepth_max, uint32_t *dirvar, hipStream_t stream)
{
#define STRIPS(CW, CH) hipLaunchKernelGGL((cdef_strip_kernel<pixel, CW, CH>), dim3(n_groups), dim3(64), 0, stream, *dst, *src, tasks, groups, \
                                          n_groups, damping, layout, bitdepth_max, dirvar)
    if (layout == DAV1D_HIP_LAYOUT_I420) STRIPS(4, 4);
    else if (layout == DAV1D_HIP_LAYOUT_I422) STRIPS(4, 8);
    else if (layout == DAV1D_HIP_LAYOUT_I444) STRIPS(8, 8);
    else STRIPS(0, 0);
#undef STRIPS
}

extern "C" int dav1d_hip_launch_cdef_groups(const DevPlanes *dst, const DevPlanes *src, int bpc, int layout, const Dav1dHipCdefTask *tasks,
                                            const CdefGroup *groups, int n_groups, int damping, uint32_t *dirvar, void *stream)
{
    if (n_groups <= 0) return 0;
    const int bitdepth_max = (1 << bpc) - 1;
    if (bpc == 8) launch_strips<uint8_t>(dst, src, layout, tasks, groups, n_groups, damping, bitdepth_max, dirvar, (hipStream_t) stream);
    else launch_strips<uint16_t>(dst, src, layout, tasks, groups, n_groups, damping, bitdepth_max, dirvar, (hipStream_t) stream);
    return hip_rc(hipGetLastError());
}

// The unit records of a frame, made on the device from one record per unit row of a 64-pixel column (cdef_rows.h): thread g = by8 * P + p
// (P = pairs of 64-pixel columns) takes the up to 16 listed units of columns 2p, 2p + 1 in unit row by8 and writes them to tasks[16 g ..]
// with groups[g] over them (n = 0: nothing there) — the arrays the host used to build, group and upload (518,400 records = 8 MB for
// an 8K frame).  bitmap (or nullptr): the listed units are marked for cdef_fill_unlisted_kernel.
__global__ __launch_bounds__(256) void cdef_expand_kernel(const Dav1dHipCdefRow *__restrict__ rows, const int w64, const int h8, const int bw4, const int bh4,
                                                          const int w8, Dav1dHipCdefTask *__restrict__ tasks, CdefGroup *__restrict__ groups,
                                                          uint32_t *__restrict__ bitmap)
{
    const int P = (w64 + 1) >> 1;
    const int gi = blockIdx.x * 256 + threadIdx.x;
    if (gi >= P * h8) return;
    const int by = gi / P, px = gi - by * P;
    Dav1dHipCdefRow r[2];
    r[0] = rows[(size_t) by * w64 + 2 * px];
    if (2 * px + 1 < w64) r[1] = rows[(size_t) by * w64 + 2 * px + 1]; else r[1].mask = 0;
    const uint32_t m = (uint32_t) r[0].mask | (uint32_t) r[1].mask << 8;
    CdefGroup g;
    g.first = (uint32_t) gi * 16u; g.by = (uint16_t) by; g.n = (uint8_t) __builtin_popcount(m);
    g.bx0 = 0; g.span = 0; g.edges = 0; g.pad = 0;
    if (m) {
        const int first = __builtin_ctz(m), last = 31 - __builtin_clz(m);
        g.bx0 = (uint16_t) (16 * px + first); g.span = (uint8_t) (last - first + 1);
        const int tb = (by > 0 ? DAV1D_HIP_CDEF_HAVE_TOP : 0) | (2 * by + 2 < bh4 ? DAV1D_HIP_CDEF_HAVE_BOTTOM : 0);
        g.pad = r[first >> 3].flags;
        g.edges = (uint8_t) (tb | (g.bx0 > 0 ? DAV1D_HIP_CDEF_HAVE_LEFT : 0) | (2 * (16 * px + last) + 2 < bw4 ? DAV1D_HIP_CDEF_HAVE_RIGHT : 0));
        Dav1dHipCdefTask *t = tasks + (size_t) gi * 16;
        for (uint32_t mm = m; mm; mm &= mm - 1) {
            const int k = __builtin_ctz(mm), bx = 16 * px + k;
            const Dav1dHipCdefRow &q = r[k >> 3];
            Dav1dHipCdefTask o;
            o.bx = (uint16_t) bx; o.by = (uint16_t) by;
            o.y_pri = q.y_pri; o.y_sec = q.y_sec; o.uv_pri = q.uv_pri; o.uv_sec = q.uv_sec;
            o.edges = (uint8_t) (tb | (bx > 0 ? DAV1D_HIP_CDEF_HAVE_LEFT : 0) | (2 * bx + 2 < bw4 ? DAV1D_HIP_CDEF_HAVE_RIGHT : 0));
            o.flags = q.flags;
            o.plane = 0; o.dir = 0; o.pad[0] = o.pad[1] = o.pad[2] = o.pad[3] = 0;
            *t++ = o;
            if (bitmap) { const int idx = by * w8 + bx; atomicOr(&bitmap[idx >> 5], 1u << (idx & 31)); }
        }
    }
    groups[gi] = g;
}

extern "C" int dav1d_hip_launch_cdef_expand(const void *rows, int w64, int h8, int bw4, int bh4, int w8, Dav1dHipCdefTask *tasks, void *groups,
                                            uint32_t *bitmap, void *stream)
{
    const int n = ((w64 + 1) >> 1) * h8;
    if (n <= 0) return 0;
    if (bitmap && hipMemsetAsync(bitmap, 0, (((size_t) w8 * h8 + 31) / 32) * 4, (hipStream_t) stream) != hipSuccess) return hip_rc(hipGetLastError());
    hipLaunchKernelGGL(cdef_expand_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t) stream, (const Dav1dHipCdefRow *) rows, w64, h8, bw4, bh4, w8,
                       tasks, (CdefGroup *) groups, bitmap);
    return hip_rc(hipGetLastError());
}

// bitmap: DEVICE, (w8 * h8 + 31) / 32 words, cleared here (tasks == nullptr: marked already, by cdef_expand_kernel).  The same alignment conditions as the strip kernel (dav1d_hip_cdef_strip_ok).
extern "C" int dav1d_hip_launch_cdef_fill_unlisted(const DevPlanes *dst, const DevPlanes *src, int bpc, int layout, const Dav1dHipCdefTask *tasks,
                                                   int n, uint32_t *bitmap, int w8, int h8, void *stream)
{
    if (w8 <= 0 || h8 <= 0) return 0;
    const size_t words = ((size_t) w8 * h8 + 31) / 32;
    if (tasks && hipMemsetAsync(bitmap, 0, words * 4, (hipStream_t) stream) != hipSuccess) return hip_rc(hipGetLastError());
    if (tasks && n > 0) hipLaunchKernelGGL(cdef_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t) stream, tasks, n, bitmap, w8);
    const dim3 grid((w8 + 31) / 32, h8);
    if (bpc == 8) hipLaunchKernelGGL((cdef_fill_unlisted_kernel<uint8_t>), grid, dim3(256), 0, (hipStream_t) stream, *dst, *src, bitmap, w8, layout);
    else hipLaunchKernelGGL((cdef_fill_unlisted_kernel<uint16_t>), grid, dim3(256), 0, (hipStream_t) stream, *dst, *src, bitmap, w8, layout);
    return hip_rc(hipGetLastError());
}

// The strip kernel moves whole 16-byte (luma, 4:4:4 chroma) / 8-byte pieces: planes and strides have to be aligned for it
// (pictures from dav1d_hip_picture_alloc are: 64-byte strides and plane starts).
bool dav1d_hip_cdef_strip_ok(const DevPlanes *dst, const DevPlanes *src, int bpc) {
    const int bps = bpc > 8 ? 2 : 1;
    for (int p = 0; p < 3; p++) {
        if (!src->data[p]) continue;
        if (((uintptr_t) src->data[p] | (uintptr_t) dst->data[p]) & 15) return false;
        if (((size_t) src->stride[p] * bps | (size_t) dst->stride[p] * bps) & 15) return false;
    }
    return true;
}

size_t dav1d_hip_cdef_make_groups(const Dav1dHipCdefTask *tasks, size_t n, size_t base, std::vector<CdefGroup> &out) {
    size_t n_raw = 0;
    bool open = false;
    CdefGroup g = {};
    int last_bx = 0, last_edges = 0;
    for (size_t i = 0; i < n; i++) {
        const Dav1dHipCdefTask &t = tasks[i];
        if (t.flags & 1) { n_raw++; if (open) { out.push_back(g); open = false; } continue; }
        // a unit joins the open group when it lies further right in the same row within 16 units of the group's first one, shares
        // the rows above / below, has a left neighbour, and the unit before it has a right neighbour
        const uint8_t rep = t.flags & (DAV1D_HIP_CDEF_BOT_REP_Y | DAV1D_HIP_CDEF_BOT_REP_UV);
        const bool joins = open && t.by == g.by && t.bx > last_bx && t.bx - g.bx0 < 16 && g.n < 16 && (t.edges & 12) == (g.edges & 12) &&
                           (t.edges & 1) && (last_edges & 2) && rep == g.pad;
        if (!joins) {
            if (open) out.push_back(g);
            g.first = (uint32_t) (base + i); g.bx0 = t.bx; g.by = t.by; g.n = 0; g.pad = rep;
            g.edges = (uint8_t) (t.edges & 13);
            open = true;
        }
        g.n++;
        g.span = (uint8_t) (t.bx - g.bx0 + 1);
        g.edges = (uint8_t) ((g.edges & 13) | (t.edges & 2));
        last_bx = t.bx; last_edges = t.edges;
    }
    if (open) out.push_back(g);
    return n_raw;
}

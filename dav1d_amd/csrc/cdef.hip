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
#include "common.h"
#include "capi.h"
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
                                            const int h, const int edges, const int lane)
{
    constexpr int w = WW - 4;
    const int n = WW * (h + 4);
    for (int i = lane; i < n; i += 64) {
        const int yy = i / WW - 2, xx = i % WW - 2;
        const bool avail = (yy >= 0 || (edges & 4)) && (yy < h || (edges & 8)) && (xx >= 0 || (edges & 1)) && (xx < w || (edges & 2));
        int v = -32768;
        if (avail) v = src[(y0 + yy) * stride + x0 + xx];
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
                                                  uint32_t *__restrict__ dirvar)
{
    __shared__ __attribute__((aligned(8))) int16_t tmp[144], tmp2[144];
    __shared__ int16_t pdir[64];
    constexpr bool HBD = sizeof(pixel) == 2;

    const int ti = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x);
    if (ti >= n) return;
    const int lane = threadIdx.x;
    const int tis = __builtin_amdgcn_readfirstlane(ti);
    const Dav1dHipCdefTask t = tasks[tis];
    const int bitdepth_min_8 = (32 - __clz(bitdepth_max)) - 8;
    const int edges = t.edges;

    // ---- luma window
    const bool raw_ = t.flags & 1;
    const int lpl = raw_ ? t.plane : 0;                  // raw calls may address any plane / 8x8, 4x8, 4x4 blocks
    const int lw = raw_ ? (t.flags & 2 ? 4 : 8) : 8, lh = raw_ ? (t.flags & 4 ? 4 : 8) : 8;
    const pixel *sy = reinterpret_cast<const pixel *>(src.data[lpl]);
    const int x0 = raw_ ? t.bx : t.bx * 8, y0 = raw_ ? t.by : t.by * 8;   // raw: pixel coordinates
    if (HBD && edges == 15 && lw == 8) load_window_fast<12>(tmp, reinterpret_cast<const uint16_t *>(sy), src.stride[lpl], x0, y0, lh, lane);
    else if (lw == 8) load_window<pixel, 12>(tmp, sy, src.stride[lpl], x0, y0, lh, edges, lane);
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
        if (HBD && edges == 15 && w == 4) {
            // 4-wide chroma: 2 pieces x (h + 4) rows per plane, U on lanes 0 .. 31, V on lanes 32 .. 63
            load_window_fast<8>(lane < 32 ? tmp : tmp2, reinterpret_cast<const uint16_t *>(src.data[lane < 32 ? 1 : 2]),
                                src.stride[lane < 32 ? 1 : 2], cx0, cy0, h, lane & 31);
        } else if (w == 8) {
            load_window<pixel, 12>(tmp, reinterpret_cast<const pixel *>(src.data[1]), src.stride[1], cx0, cy0, h, edges, lane);
            load_window<pixel, 12>(tmp2, reinterpret_cast<const pixel *>(src.data[2]), src.stride[2], cx0, cy0, h, edges, lane);
        } else {
            load_window<pixel, 8>(tmp, reinterpret_cast<const pixel *>(src.data[1]), src.stride[1], cx0, cy0, h, edges, lane);
            load_window<pixel, 8>(tmp2, reinterpret_cast<const pixel *>(src.data[2]), src.stride[2], cx0, cy0, h, edges, lane);
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

} // namespace

extern "C" int dav1d_hip_launch_cdef(const DevPlanes *dst, const DevPlanes *src, int bpc, int layout,
                                     const Dav1dHipCdefTask *tasks, int n, int damping, uint32_t *dirvar, void *stream)
{
    if (n <= 0) return 0;
    const int bitdepth_max = (1 << bpc) - 1;
    if (bpc == 8)
        hipLaunchKernelGGL((cdef_kernel<uint8_t>), dim3(n), dim3(64), 0, (hipStream_t) stream, *dst, *src, tasks, n, damping, layout, bitdepth_max, dirvar);
    else
        hipLaunchKernelGGL((cdef_kernel<uint16_t>), dim3(n), dim3(64), 0, (hipStream_t) stream, *dst, *src, tasks, n, damping, layout, bitdepth_max, dirvar);
    return hip_rc(hipGetLastError());
}

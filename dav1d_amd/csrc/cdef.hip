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

__device__ __forceinline__ int constrain(const int diff, const int threshold, const int shift) {
    const int adiff = diff < 0 ? -diff : diff;
    const int v = dv::imin(adiff, dv::imax(0, threshold - (adiff >> shift)));
    return diff < 0 ? -v : v;
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
            sum += tap * constrain(p0 - px, pri, pri_shift);
            sum += tap * constrain(p1 - px, pri, pri_shift);
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
            sum += tap * constrain(s0 - px, sec, sec_shift);
            sum += tap * constrain(s1 - px, sec, sec_shift);
            sum += tap * constrain(s2 - px, sec, sec_shift);
            sum += tap * constrain(s3 - px, sec, sec_shift);
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

template <typename pixel>
__global__ __launch_bounds__(64) void cdef_kernel(const DevPlanes dst, const DevPlanes src, const Dav1dHipCdefTask *__restrict__ tasks,
                                                  const int n, const int damping, const int layout, const int bitdepth_max,
                                                  uint32_t *__restrict__ dirvar)
{
    __shared__ int16_t tmp[144], tmp2[144];
    __shared__ int psum[2 * 8 + 2 * 15 + 4 * 11];    // hv[2][8], diag[2][15], alt[4][11]
    __shared__ unsigned cost_s[8];

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
    if (lw == 8) load_window<pixel, 12>(tmp, sy, src.stride[lpl], x0, y0, lh, edges, lane);
    else load_window<pixel, 8>(tmp, sy, src.stride[lpl], x0, y0, lh, edges, lane);
    for (int i = lane; i < 90; i += 64) psum[i] = 0;
    dv::wave_sync();

    // ---- direction search (only when a primary strength is in play, src/cdef_apply_tmpl.c:208-212)
    int dir = 0;
    unsigned var = 0;
    const bool raw = t.flags & 1;      // DSP-level call: explicit dir / strengths, luma path only, no adjust
    if (!raw && (t.y_pri || t.uv_pri || dirvar)) {
        const int x = lane & 7, y = lane >> 3;
        const int px = (tmp[(y + 2) * 12 + x + 2] >> bitdepth_min_8) - 128;
        int *hv = psum, *diag = psum + 16, *alt = psum + 46;
        atomicAdd(&diag[0 * 15 + y + x], px);
        atomicAdd(&alt[0 * 11 + y + (x >> 1)], px);
        atomicAdd(&hv[0 * 8 + y], px);
        atomicAdd(&alt[1 * 11 + 3 + y - (x >> 1)], px);
        atomicAdd(&diag[1 * 15 + 7 + y - x], px);
        atomicAdd(&alt[2 * 11 + 3 - (y >> 1) + x], px);
        atomicAdd(&hv[1 * 8 + x], px);
        atomicAdd(&alt[3 * 11 + (y >> 1) + x], px);
        dv::wave_sync();
        if (lane < 8) {
            static const unsigned short div_table[7] = { 840, 420, 280, 210, 168, 140, 120 };
            unsigned c = 0;
            const int nn = lane;
            if (nn == 2 || nn == 6) {
                const int *p = hv + (nn == 6) * 8;
                for (int k = 0; k < 8; k++) c += p[k] * p[k];
                c *= 105;
            } else if (nn == 0 || nn == 4) {
                const int *p = diag + (nn == 4) * 15;
                for (int k = 0; k < 7; k++) c += (p[k] * p[k] + p[14 - k] * p[14 - k]) * div_table[k];
                c += p[7] * p[7] * 105;
            } else {
                const int *p = alt + (nn >> 1) * 11;
                for (int m = 0; m < 5; m++) c += p[3 + m] * p[3 + m];
                c *= 105;
                for (int m = 0; m < 3; m++) c += (p[m] * p[m] + p[10 - m] * p[10 - m]) * div_table[2 * m + 1];
            }
            cost_s[nn] = c;
        }
        dv::wave_sync();
        unsigned best = cost_s[0];
        for (int k = 1; k < 8; k++) if (cost_s[k] > best) { best = cost_s[k]; dir = k; }
        var = (best - cost_s[dir ^ 4]) >> 10;
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
        if (w == 8) {
            load_window<pixel, 12>(tmp, reinterpret_cast<const pixel *>(src.data[1]), src.stride[1], cx0, cy0, h, edges, lane);
            load_window<pixel, 12>(tmp2, reinterpret_cast<const pixel *>(src.data[2]), src.stride[2], cx0, cy0, h, edges, lane);
        } else {
            load_window<pixel, 8>(tmp, reinterpret_cast<const pixel *>(src.data[1]), src.stride[1], cx0, cy0, h, edges, lane);
            load_window<pixel, 8>(tmp2, reinterpret_cast<const pixel *>(src.data[2]), src.stride[2], cx0, cy0, h, edges, lane);
        }
        dv::wave_sync();
        const int npx = w * h;                        // 16, 32 or 64 pixels per plane
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

// Batched loop restoration for gfx950 (Wiener).
//
// Contract per task = one call of dsp->lr.wiener[*] (wiener_c, reference
// src/looprestoration_tmpl.c:44-387) on one restoration-unit stripe (w <= 384, h <= 64) with the
// arguments lr_stripe() builds (src/lr_apply_tmpl.c:36-97).
//
// The reference filters in place and therefore keeps a 4-pixel `left` backup and the `lpf` stripe
// rows saved from the deblocked picture (src/lf_apply_tmpl.c:41-102).  Here the filter is out of
// place: `src` is the immutable loop-restoration input (CDEF output), `lpf` the immutable deblocked
// (pre-CDEF) picture that supplies the 2 rows above / below a stripe, `dst` the output, so every
// task is independent.  Mapping: one lane per output column, 64 columns per wave; a lane walks
// down the h + 6 virtual rows, filters each horizontally (7 taps straight from L1-resident
// pixels) and keeps the last 7 results in registers for the vertical filter.
#include "common.h"
#include "capi.h"

namespace {

template <typename pixel>
__global__ __launch_bounds__(64) void wiener_kernel(const DevPlanes dst, const DevPlanes src, const DevPlanes lpf,
                                                    const Dav1dHipLrTask *__restrict__ tasks, const int n, const int bitdepth_max)
{
    constexpr bool HBD = sizeof(pixel) == 2;
    const int ti = blockIdx.y;
    if (ti >= n) return;
    const Dav1dHipLrTask t = tasks[__builtin_amdgcn_readfirstlane(ti)];
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (blockIdx.x * 64 >= t.w) return;
    const bool active = x < t.w;
    const int xc = active ? x : t.w - 1;

    const int bitdepth = 32 - __clz(bitdepth_max);
    const int round_bits_h = 3 + (bitdepth == 12) * 2, rounding_off_h = 1 << (round_bits_h - 1);
    const int clip_limit = 1 << (bitdepth + 1 + 7 - round_bits_h);
    const int round_bits_v = 11 - (bitdepth == 12) * 2, rounding_off_v = 1 << (round_bits_v - 1);
    const int round_offset = 1 << (bitdepth + (round_bits_v - 1));

    const int pl = t.plane, w = t.w, h = t.h, edges = t.edges;
    const pixel *const s = reinterpret_cast<const pixel *>(src.data[pl]);
    const pixel *const l = reinterpret_cast<const pixel *>(lpf.data[pl]);
    const int ss = src.stride[pl], ls = lpf.stride[pl];
    // the reference only reaches its "two rows below" code for stripes of at least 4 (with rows above) or 6 rows
    // (src/looprestoration_tmpl.c:274-355); shorter stripes replicate their last row
    const bool use_bottom = (edges & 8) && h >= ((edges & 4) ? 4 : 6);

    // columns of the 7 taps, with the edge rules of wiener_filter_h (:47-161)
    int col[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        int c = xc + i - 3;
        if (c < 0 && !(edges & 1)) c = 0;
        if (c >= w && !(edges & 2)) c = w - 1;
        col[i] = t.x + c;
    }
    int fh[7], fv[7];
#pragma unroll
    for (int i = 0; i < 7; i++) { fh[i] = t.filter[0][i]; fv[i] = t.filter[1][i]; }

    int win[7];     // horizontally filtered rows r-6 .. r
#pragma unroll
    for (int i = 0; i < 7; i++) win[i] = 0;
    pixel *d = reinterpret_cast<pixel *>(dst.data[pl]) + t.y * dst.stride[pl] + t.x + xc;

    for (int r = -3; r < h + 3; r++) {
        // which picture row feeds virtual row r
        const pixel *row;
        if (r < 0) {
            if (edges & 4) row = l + (t.y - (r == -1 ? 1 : 2)) * ls;
            else row = s + t.y * ss;
        } else if (r >= h) {
            if (use_bottom) row = l + (t.y + h + (r == h ? 0 : 1)) * ls;
            else row = s + (t.y + h - 1) * ss;
        } else {
            row = s + (t.y + r) * ss;
        }
        int sum = 1 << (bitdepth + 6);
        if (!HBD) sum += row[t.x + xc] * 128;
#pragma unroll
        for (int i = 0; i < 7; i++) sum += row[col[i]] * fh[i];
        sum = dv::iclip((sum + rounding_off_h) >> round_bits_h, 0, clip_limit - 1);
#pragma unroll
        for (int i = 0; i < 6; i++) win[i] = win[i + 1];
        win[6] = sum;
        if (r >= 3) {
            int v = -round_offset;
#pragma unroll
            for (int k = 0; k < 7; k++) v += win[k] * fv[k];
            if (active) d[(r - 3) * dst.stride[pl]] = (pixel) dv::iclip((v + rounding_off_v) >> round_bits_v, 0, bitdepth_max);
        }
    }
}

} // namespace

extern "C" int dav1d_hip_launch_wiener(const DevPlanes *dst, const DevPlanes *src, const DevPlanes *lpf, int bpc,
                                       const Dav1dHipLrTask *tasks, int n, void *stream)
{
    if (n <= 0) return 0;
    const int bitdepth_max = (1 << bpc) - 1;
    const dim3 grid(6, n);          // up to 384 columns per unit
    if (bpc == 8)
        hipLaunchKernelGGL((wiener_kernel<uint8_t>), grid, dim3(64), 0, (hipStream_t) stream, *dst, *src, *lpf, tasks, n, bitdepth_max);
    else
        hipLaunchKernelGGL((wiener_kernel<uint16_t>), grid, dim3(64), 0, (hipStream_t) stream, *dst, *src, *lpf, tasks, n, bitdepth_max);
    return hip_rc(hipGetLastError());
}

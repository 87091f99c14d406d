// Batched deblocking filter for gfx950.
//
// Contract per task = one call of the reference's dsp->lf.loop_filter_sb[plane_type][dir]
// (src/loopfilter_tmpl.c:163-257 -> loop_filter, :37-161) with the arguments the sbrow
// drivers build (src/lf_apply_tmpl.c:176-311): a line of up to 32 edge units (4 pixels each)
// along one superblock column / row, three (luma) or two (chroma) bitmasks selecting the
// filter width per unit, the per-4x4 level array, and the E / I limit LUT.
//
// All edges of one direction are independent of each other (an edge modifies at most half of the
// smaller adjacent transform block), so a pass = one launch over every task of that direction;
// vertical edges of the whole frame run before horizontal ones, which is equivalent to the
// reference's sbrow order (src/thread_task.c:783-798).  Mapping: one lane per edge unit,
// 32 lanes per task, two tasks per wave; a lane filters its 4 lines in registers and stores
// exactly the pixels the reference stores.
#include "common.h"
#include "capi.h"

namespace {

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// One line across the edge.  `stb` = distance (in pixels) between p/q taps.  Follows loop_filter()
// (src/loopfilter_tmpl.c:47-160) statement by statement.
template <typename pixel>
__device__ __forceinline__ void lf_line(pixel *dst, const int stb, const int wd, const int E, const int I, const int H,
                                        const int F, const int bitdepth_min_8, const int bitdepth_max)
{
    int p6 = 0, p5 = 0, p4 = 0, p3 = 0, p2 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0;
    const int p1 = dst[stb * -2], p0 = dst[stb * -1], q0 = dst[0], q1 = dst[stb];
    bool fm = iabs(p1 - p0) <= I && iabs(q1 - q0) <= I && iabs(p0 - q0) * 2 + (iabs(p1 - q1) >> 1) <= E;
    if (wd > 4) {
        p2 = dst[stb * -3]; q2 = dst[stb * 2];
        fm &= iabs(p2 - p1) <= I && iabs(q2 - q1) <= I;
        if (wd > 6) {
            p3 = dst[stb * -4]; q3 = dst[stb * 3];
            fm &= iabs(p3 - p2) <= I && iabs(q3 - q2) <= I;
        }
    }
    if (!fm) return;
    bool flat8out = false, flat8in = false;
    if (wd >= 16) {
        p6 = dst[stb * -7]; p5 = dst[stb * -6]; p4 = dst[stb * -5];
        q4 = dst[stb * 4]; q5 = dst[stb * 5]; q6 = dst[stb * 6];
        flat8out = iabs(p6 - p0) <= F && iabs(p5 - p0) <= F && iabs(p4 - p0) <= F &&
                   iabs(q4 - q0) <= F && iabs(q5 - q0) <= F && iabs(q6 - q0) <= F;
    }
    if (wd >= 6) flat8in = iabs(p2 - p0) <= F && iabs(p1 - p0) <= F && iabs(q1 - q0) <= F && iabs(q2 - q0) <= F;
    if (wd >= 8) flat8in = flat8in && iabs(p3 - p0) <= F && iabs(q3 - q0) <= F;

    if (wd >= 16 && flat8out && flat8in) {
        dst[stb * -6] = (pixel) ((p6 + p6 + p6 + p6 + p6 + p6 * 2 + p5 * 2 + p4 * 2 + p3 + p2 + p1 + p0 + q0 + 8) >> 4);
        dst[stb * -5] = (pixel) ((p6 + p6 + p6 + p6 + p6 + p5 * 2 + p4 * 2 + p3 * 2 + p2 + p1 + p0 + q0 + q1 + 8) >> 4);
        dst[stb * -4] = (pixel) ((p6 + p6 + p6 + p6 + p5 + p4 * 2 + p3 * 2 + p2 * 2 + p1 + p0 + q0 + q1 + q2 + 8) >> 4);
        dst[stb * -3] = (pixel) ((p6 + p6 + p6 + p5 + p4 + p3 * 2 + p2 * 2 + p1 * 2 + p0 + q0 + q1 + q2 + q3 + 8) >> 4);
        dst[stb * -2] = (pixel) ((p6 + p6 + p5 + p4 + p3 + p2 * 2 + p1 * 2 + p0 * 2 + q0 + q1 + q2 + q3 + q4 + 8) >> 4);
        dst[stb * -1] = (pixel) ((p6 + p5 + p4 + p3 + p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1 + q2 + q3 + q4 + q5 + 8) >> 4);
        dst[stb * +0] = (pixel) ((p5 + p4 + p3 + p2 + p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2 + q3 + q4 + q5 + q6 + 8) >> 4);
        dst[stb * +1] = (pixel) ((p4 + p3 + p2 + p1 + p0 + q0 * 2 + q1 * 2 + q2 * 2 + q3 + q4 + q5 + q6 + q6 + 8) >> 4);
        dst[stb * +2] = (pixel) ((p3 + p2 + p1 + p0 + q0 + q1 * 2 + q2 * 2 + q3 * 2 + q4 + q5 + q6 + q6 + q6 + 8) >> 4);
        dst[stb * +3] = (pixel) ((p2 + p1 + p0 + q0 + q1 + q2 * 2 + q3 * 2 + q4 * 2 + q5 + q6 + q6 + q6 + q6 + 8) >> 4);
        dst[stb * +4] = (pixel) ((p1 + p0 + q0 + q1 + q2 + q3 * 2 + q4 * 2 + q5 * 2 + q6 + q6 + q6 + q6 + q6 + 8) >> 4);
        dst[stb * +5] = (pixel) ((p0 + q0 + q1 + q2 + q3 + q4 * 2 + q5 * 2 + q6 * 2 + q6 + q6 + q6 + q6 + q6 + 8) >> 4);
    } else if (wd >= 8 && flat8in) {
        dst[stb * -3] = (pixel) ((p3 + p3 + p3 + 2 * p2 + p1 + p0 + q0 + 4) >> 3);
        dst[stb * -2] = (pixel) ((p3 + p3 + p2 + 2 * p1 + p0 + q0 + q1 + 4) >> 3);
        dst[stb * -1] = (pixel) ((p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2 + 4) >> 3);
        dst[stb * +0] = (pixel) ((p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3 + 4) >> 3);
        dst[stb * +1] = (pixel) ((p1 + p0 + q0 + 2 * q1 + q2 + q3 + q3 + 4) >> 3);
        dst[stb * +2] = (pixel) ((p0 + q0 + q1 + 2 * q2 + q3 + q3 + q3 + 4) >> 3);
    } else if (wd == 6 && flat8in) {
        dst[stb * -2] = (pixel) ((p2 + 2 * p2 + 2 * p1 + 2 * p0 + q0 + 4) >> 3);
        dst[stb * -1] = (pixel) ((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
        dst[stb * +0] = (pixel) ((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
        dst[stb * +1] = (pixel) ((p0 + 2 * q0 + 2 * q1 + 2 * q2 + q2 + 4) >> 3);
    } else {
        const bool hev = iabs(p1 - p0) > H || iabs(q1 - q0) > H;
        const int lo = -128 * (1 << bitdepth_min_8), hi = 128 * (1 << bitdepth_min_8) - 1;
        if (hev) {
            int f = dv::iclip(p1 - q1, lo, hi);
            f = dv::iclip(3 * (q0 - p0) + f, lo, hi);
            const int f1 = dv::imin(f + 4, hi) >> 3, f2 = dv::imin(f + 3, hi) >> 3;
            dst[stb * -1] = (pixel) dv::iclip(p0 + f2, 0, bitdepth_max);
            dst[0] = (pixel) dv::iclip(q0 - f1, 0, bitdepth_max);
        } else {
            int f = dv::iclip(3 * (q0 - p0), lo, hi);
            const int f1 = dv::imin(f + 4, hi) >> 3, f2 = dv::imin(f + 3, hi) >> 3;
            dst[stb * -1] = (pixel) dv::iclip(p0 + f2, 0, bitdepth_max);
            dst[0] = (pixel) dv::iclip(q0 - f1, 0, bitdepth_max);
            f = (f1 + 1) >> 1;
            dst[stb * -2] = (pixel) dv::iclip(p1 + f, 0, bitdepth_max);
            dst[stb] = (pixel) dv::iclip(q1 - f, 0, bitdepth_max);
        }
    }
}

struct LfLut { uint8_t e[64], i[64]; };     // Av1FilterLUT (src/lf_mask.h:36-40) without the sharp[] helper

template <typename pixel>
__global__ __launch_bounds__(64) void lf_kernel(const DevPlanes dst, const Dav1dHipLfTask *__restrict__ tasks, const int n,
                                                const uint8_t *__restrict__ lvl, const int b4_stride, const LfLut lut,
                                                const int bitdepth_max)
{
    const int lane = threadIdx.x;
    const int ti = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x) * 2 + (lane >> 5);
    if (ti >= n) return;
    const int u = lane & 31;
    const Dav1dHipLfTask t = tasks[ti];
    const unsigned bit = 1u << u;
    const bool luma = t.plane == 0;
    const unsigned vm = t.vmask[0] | t.vmask[1] | (luma ? t.vmask[2] : 0u);
    if (!(vm & bit)) return;
    // level of this unit, else of the neighbour on the other side of the edge (src/loopfilter_tmpl.c:175,199)
    const uint8_t *l = lvl + ((size_t) t.lvl_off + (size_t) u * (t.dir ? 1 : b4_stride)) * 4 + t.lvl_comp;
    int L = l[0];
    if (!L) L = t.dir ? l[-4 * b4_stride] : l[-4];
    if (!L) return;
    const int bitdepth_min_8 = (32 - __clz(bitdepth_max)) - 8;
    const int H = (L >> 4) << bitdepth_min_8, E = lut.e[L] << bitdepth_min_8, I = lut.i[L] << bitdepth_min_8;
    const int F = 1 << bitdepth_min_8;
    int wd;
    if (luma) wd = 4 << ((t.vmask[2] & bit) ? 2 : ((t.vmask[1] & bit) ? 1 : 0));
    else wd = 4 + 2 * ((t.vmask[1] & bit) ? 1 : 0);
    const int stride = dst.stride[t.plane];
    pixel *d = reinterpret_cast<pixel *>(dst.data[t.plane]) + t.dst_off;
    // dir 0: edge between columns, units run down the column (4 rows each), taps along x
    // dir 1: edge between rows, units run along the row (4 columns each), taps along y
    const int sta = t.dir ? 1 : stride, stb = t.dir ? stride : 1;
    d += (size_t) u * 4 * sta;
#pragma unroll 1
    for (int i = 0; i < 4; i++) lf_line<pixel>(d + i * sta, stb, wd, E, I, H, F, bitdepth_min_8, bitdepth_max);
}

} // namespace

extern "C" int dav1d_hip_launch_lf(const DevPlanes *dst, int bpc, const Dav1dHipLfTask *tasks, int n, const uint8_t *lvl,
                                   int b4_stride, const uint8_t *lut_e, const uint8_t *lut_i, void *stream)
{
    if (n <= 0) return 0;
    LfLut lut;
    for (int k = 0; k < 64; k++) { lut.e[k] = lut_e[k]; lut.i[k] = lut_i[k]; }
    const int bitdepth_max = (1 << bpc) - 1;
    const int grid = (n + 1) / 2;
    if (bpc == 8)
        hipLaunchKernelGGL((lf_kernel<uint8_t>), dim3(grid), dim3(64), 0, (hipStream_t) stream, *dst, tasks, n, lvl, b4_stride, lut, bitdepth_max);
    else
        hipLaunchKernelGGL((lf_kernel<uint16_t>), dim3(grid), dim3(64), 0, (hipStream_t) stream, *dst, tasks, n, lvl, b4_stride, lut, bitdepth_max);
    return hip_rc(hipGetLastError());
}

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

// How lf_line reaches the pixels of its line: tap k is k pixels from the edge (k < 0: the p side).
// Strided: straight to memory, taps `stb` pixels apart (horizontal edges: the taps of a line sit in different rows).
template <typename pixel>
struct LfStrided {
    pixel *dst; int stb;
    __device__ __forceinline__ int get(const int k) const { return dst[stb * k]; }
    __device__ __forceinline__ void set(const int k, const int v) { dst[stb * k] = (pixel) v; }
    __device__ __forceinline__ void commit(int) {}
};
// Window (vertical edges: the taps of a line are neighbours in memory): the 8 (filters up to 8 wide) or 16 pixels around
// the edge come in with one or two wide loads and live in registers; commit(n) writes back the n pixels on either side of
// the edge — exactly the span that belongs to this edge alone for the filter that ran (2: the narrow filters, 4: the 8-wide
// flat filter, whose neighbours are at least 8 pixels away, 8: the 16-wide one), never pixels a neighbouring edge may be
// filtering at the same time.
struct __attribute__((packed, aligned(2))) LfW16 { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(2))) LfW8 { uint32_t a, b; };
struct __attribute__((packed, aligned(1))) LfB8 { uint32_t a, b; };
struct __attribute__((packed, aligned(1))) LfB4 { uint32_t a; };
template <typename pixel>
struct LfWindow {
    pixel *dst;              // pixel 0 = first pixel on the q side
    int win[16];             // taps -8 .. 7
    __device__ __forceinline__ LfWindow(pixel *d, const bool wide) : dst(d) {
#pragma unroll
        for (int k = 0; k < 16; k++) win[k] = 0;
        if (sizeof(pixel) == 2) {
            if (wide) {
                const LfW16 lo = *reinterpret_cast<const LfW16 *>(d - 8), hi = *reinterpret_cast<const LfW16 *>(d);
                const uint32_t w[8] = { lo.a, lo.b, lo.c, lo.d, hi.a, hi.b, hi.c, hi.d };
#pragma unroll
                for (int k = 0; k < 8; k++) { win[2 * k] = (int) (w[k] & 0xffff); win[2 * k + 1] = (int) (w[k] >> 16); }
            } else {
                const LfW16 m = *reinterpret_cast<const LfW16 *>(d - 4);
                const uint32_t w[4] = { m.a, m.b, m.c, m.d };
#pragma unroll
                for (int k = 0; k < 4; k++) { win[4 + 2 * k] = (int) (w[k] & 0xffff); win[5 + 2 * k] = (int) (w[k] >> 16); }
            }
        } else {
            if (wide) {
                const LfB8 lo = *reinterpret_cast<const LfB8 *>(d - 8), hi = *reinterpret_cast<const LfB8 *>(d);
                const uint32_t w[4] = { lo.a, lo.b, hi.a, hi.b };
#pragma unroll
                for (int k = 0; k < 16; k++) win[k] = (int) ((w[k >> 2] >> (8 * (k & 3))) & 0xff);
            } else {
                const LfB8 m = *reinterpret_cast<const LfB8 *>(d - 4);
                const uint32_t w[2] = { m.a, m.b };
#pragma unroll
                for (int k = 0; k < 8; k++) win[4 + k] = (int) ((w[k >> 2] >> (8 * (k & 3))) & 0xff);
            }
        }
    }
    __device__ __forceinline__ int get(const int k) const { return win[k + 8]; }
    __device__ __forceinline__ void set(const int k, const int v) { win[k + 8] = v; }
    __device__ __forceinline__ void commit(const int n) {
        if (sizeof(pixel) == 2) {
            if (n == 2) {
                LfW8 o = { (uint32_t) win[6] | (uint32_t) win[7] << 16, (uint32_t) win[8] | (uint32_t) win[9] << 16 };
                *reinterpret_cast<LfW8 *>(dst - 2) = o;
            } else if (n == 4) {
                LfW16 o = { (uint32_t) win[4] | (uint32_t) win[5] << 16, (uint32_t) win[6] | (uint32_t) win[7] << 16,
                            (uint32_t) win[8] | (uint32_t) win[9] << 16, (uint32_t) win[10] | (uint32_t) win[11] << 16 };
                *reinterpret_cast<LfW16 *>(dst - 4) = o;
            } else {
                LfW16 lo = { (uint32_t) win[0] | (uint32_t) win[1] << 16, (uint32_t) win[2] | (uint32_t) win[3] << 16,
                             (uint32_t) win[4] | (uint32_t) win[5] << 16, (uint32_t) win[6] | (uint32_t) win[7] << 16 };
                LfW16 hi = { (uint32_t) win[8] | (uint32_t) win[9] << 16, (uint32_t) win[10] | (uint32_t) win[11] << 16,
                             (uint32_t) win[12] | (uint32_t) win[13] << 16, (uint32_t) win[14] | (uint32_t) win[15] << 16 };
                *reinterpret_cast<LfW16 *>(dst - 8) = lo;
                *reinterpret_cast<LfW16 *>(dst) = hi;
            }
        } else {
            auto b4 = [&](const int k) { return (uint32_t) win[k] | (uint32_t) win[k + 1] << 8 | (uint32_t) win[k + 2] << 16 | (uint32_t) win[k + 3] << 24; };
            if (n == 2) { LfB4 o = { b4(6) }; *reinterpret_cast<LfB4 *>(dst - 2) = o; }
            else if (n == 4) { LfB8 o = { b4(4), b4(8) }; *reinterpret_cast<LfB8 *>(dst - 4) = o; }
            else { LfB8 lo = { b4(0), b4(4) }, hi = { b4(8), b4(12) }; *reinterpret_cast<LfB8 *>(dst - 8) = lo; *reinterpret_cast<LfB8 *>(dst) = hi; }
        }
    }
};

// One line across the edge.  `stb` = distance (in pixels) between p/q taps.  Follows loop_filter()
// (src/loopfilter_tmpl.c:47-160) statement by statement.
template <typename IO>
__device__ __forceinline__ void lf_line(IO &io, const int wd, const int E, const int I, const int H,
                                        const int F, const int bitdepth_min_8, const int bitdepth_max)
{
    int p6 = 0, p5 = 0, p4 = 0, p3 = 0, p2 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0;
    const int p1 = io.get(-2), p0 = io.get(-1), q0 = io.get(0), q1 = io.get(1);
    bool fm = iabs(p1 - p0) <= I && iabs(q1 - q0) <= I && iabs(p0 - q0) * 2 + (iabs(p1 - q1) >> 1) <= E;
    if (wd > 4) {
        p2 = io.get(-3); q2 = io.get(2);
        fm &= iabs(p2 - p1) <= I && iabs(q2 - q1) <= I;
        if (wd > 6) {
            p3 = io.get(-4); q3 = io.get(3);
            fm &= iabs(p3 - p2) <= I && iabs(q3 - q2) <= I;
        }
    }
    if (!fm) return;
    bool flat8out = false, flat8in = false;
    if (wd >= 16) {
        p6 = io.get(-7); p5 = io.get(-6); p4 = io.get(-5);
        q4 = io.get(4); q5 = io.get(5); q6 = io.get(6);
        flat8out = iabs(p6 - p0) <= F && iabs(p5 - p0) <= F && iabs(p4 - p0) <= F &&
                   iabs(q4 - q0) <= F && iabs(q5 - q0) <= F && iabs(q6 - q0) <= F;
    }
    if (wd >= 6) flat8in = iabs(p2 - p0) <= F && iabs(p1 - p0) <= F && iabs(q1 - q0) <= F && iabs(q2 - q0) <= F;
    if (wd >= 8) flat8in = flat8in && iabs(p3 - p0) <= F && iabs(q3 - q0) <= F;

    if (wd >= 16 && flat8out && flat8in) {
        io.set(-6, ((p6 + p6 + p6 + p6 + p6 + p6 * 2 + p5 * 2 + p4 * 2 + p3 + p2 + p1 + p0 + q0 + 8) >> 4));
        io.set(-5, ((p6 + p6 + p6 + p6 + p6 + p5 * 2 + p4 * 2 + p3 * 2 + p2 + p1 + p0 + q0 + q1 + 8) >> 4));
        io.set(-4, ((p6 + p6 + p6 + p6 + p5 + p4 * 2 + p3 * 2 + p2 * 2 + p1 + p0 + q0 + q1 + q2 + 8) >> 4));
        io.set(-3, ((p6 + p6 + p6 + p5 + p4 + p3 * 2 + p2 * 2 + p1 * 2 + p0 + q0 + q1 + q2 + q3 + 8) >> 4));
        io.set(-2, ((p6 + p6 + p5 + p4 + p3 + p2 * 2 + p1 * 2 + p0 * 2 + q0 + q1 + q2 + q3 + q4 + 8) >> 4));
        io.set(-1, ((p6 + p5 + p4 + p3 + p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1 + q2 + q3 + q4 + q5 + 8) >> 4));
        io.set(0, ((p5 + p4 + p3 + p2 + p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2 + q3 + q4 + q5 + q6 + 8) >> 4));
        io.set(1, ((p4 + p3 + p2 + p1 + p0 + q0 * 2 + q1 * 2 + q2 * 2 + q3 + q4 + q5 + q6 + q6 + 8) >> 4));
        io.set(2, ((p3 + p2 + p1 + p0 + q0 + q1 * 2 + q2 * 2 + q3 * 2 + q4 + q5 + q6 + q6 + q6 + 8) >> 4));
        io.set(3, ((p2 + p1 + p0 + q0 + q1 + q2 * 2 + q3 * 2 + q4 * 2 + q5 + q6 + q6 + q6 + q6 + 8) >> 4));
        io.set(4, ((p1 + p0 + q0 + q1 + q2 + q3 * 2 + q4 * 2 + q5 * 2 + q6 + q6 + q6 + q6 + q6 + 8) >> 4));
        io.set(5, ((p0 + q0 + q1 + q2 + q3 + q4 * 2 + q5 * 2 + q6 * 2 + q6 + q6 + q6 + q6 + q6 + 8) >> 4));
        io.commit(8);
    } else if (wd >= 8 && flat8in) {
        io.set(-3, ((p3 + p3 + p3 + 2 * p2 + p1 + p0 + q0 + 4) >> 3));
        io.set(-2, ((p3 + p3 + p2 + 2 * p1 + p0 + q0 + q1 + 4) >> 3));
        io.set(-1, ((p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2 + 4) >> 3));
        io.set(0, ((p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3 + 4) >> 3));
        io.set(1, ((p1 + p0 + q0 + 2 * q1 + q2 + q3 + q3 + 4) >> 3));
        io.set(2, ((p0 + q0 + q1 + 2 * q2 + q3 + q3 + q3 + 4) >> 3));
        io.commit(4);
    } else if (wd == 6 && flat8in) {
        io.set(-2, ((p2 + 2 * p2 + 2 * p1 + 2 * p0 + q0 + 4) >> 3));
        io.set(-1, ((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3));
        io.set(0, ((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3));
        io.set(1, ((p0 + 2 * q0 + 2 * q1 + 2 * q2 + q2 + 4) >> 3));
        io.commit(2);
    } else {
        const bool hev = iabs(p1 - p0) > H || iabs(q1 - q0) > H;
        const int lo = -128 * (1 << bitdepth_min_8), hi = 128 * (1 << bitdepth_min_8) - 1;
        if (hev) {
            int f = dv::iclip(p1 - q1, lo, hi);
            f = dv::iclip(3 * (q0 - p0) + f, lo, hi);
            const int f1 = dv::imin(f + 4, hi) >> 3, f2 = dv::imin(f + 3, hi) >> 3;
            io.set(-1, dv::iclip(p0 + f2, 0, bitdepth_max));
            io.set(0, dv::iclip(q0 - f1, 0, bitdepth_max));
            io.commit(2);
        } else {
            int f = dv::iclip(3 * (q0 - p0), lo, hi);
            const int f1 = dv::imin(f + 4, hi) >> 3, f2 = dv::imin(f + 3, hi) >> 3;
            io.set(-1, dv::iclip(p0 + f2, 0, bitdepth_max));
            io.set(0, dv::iclip(q0 - f1, 0, bitdepth_max));
            f = (f1 + 1) >> 1;
            io.set(-2, dv::iclip(p1 + f, 0, bitdepth_max));
            io.set(1, dv::iclip(q1 - f, 0, bitdepth_max));
            io.commit(2);
        }
    }
}

struct LfLut { uint8_t e[64], i[64]; };     // Av1FilterLUT (src/lf_mask.h:36-40) without the sharp[] helper

// DIR 1 (edges between rows): one lane per edge unit, 32 lanes per task, two tasks per wave; a lane walks its 4 lines
// (neighbouring lanes sit 4 pixels apart in the same rows, so their 2-byte accesses coalesce).
// DIR 0 (edges between columns): one lane per LINE — a wave takes 16 units = 64 consecutive rows of a task, two waves per
// task — and the taps of a line, neighbours in memory, come and go in wide pieces (LfWindow): with one lane per unit every
// 2-byte tap access of a wave touched 64 different rows, 26 such instructions per line.
template <typename pixel, int DIR>
__global__ __launch_bounds__(64) void lf_kernel(const DevPlanes dst, const Dav1dHipLfTask *__restrict__ tasks, const int n,
                                                const uint8_t *__restrict__ lvl, const int b4_stride, const LfLut lut,
                                                const int bitdepth_max)
{
    const int lane = threadIdx.x;
    const int wid = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x);
    const int ti = DIR ? wid * 2 + (lane >> 5) : wid >> 1;
    if (ti >= n) return;
    const int u = DIR ? lane & 31 : (wid & 1) * 16 + (lane >> 2);
    const Dav1dHipLfTask t = tasks[DIR ? ti : __builtin_amdgcn_readfirstlane(ti)];
    const unsigned bit = 1u << u;
    const bool luma = t.plane == 0;
    const unsigned vm = t.vmask[0] | t.vmask[1] | (luma ? t.vmask[2] : 0u);
    if (!(vm & bit)) return;
    // level of this unit, else of the neighbour on the other side of the edge (src/loopfilter_tmpl.c:175,199)
    const uint8_t *l = lvl + ((size_t) t.lvl_off + (size_t) u * (DIR ? 1 : b4_stride)) * 4 + t.lvl_comp;
    int L = l[0];
    if (!L) L = DIR ? l[-4 * b4_stride] : l[-4];
    if (!L) return;
    const int bitdepth_min_8 = (32 - __clz(bitdepth_max)) - 8;
    const int H = (L >> 4) << bitdepth_min_8, E = lut.e[L] << bitdepth_min_8, I = lut.i[L] << bitdepth_min_8;
    const int F = 1 << bitdepth_min_8;
    int wd;
    if (luma) wd = 4 << ((t.vmask[2] & bit) ? 2 : ((t.vmask[1] & bit) ? 1 : 0));
    else wd = 4 + 2 * ((t.vmask[1] & bit) ? 1 : 0);
    const int stride = dst.stride[t.plane];
    pixel *d = reinterpret_cast<pixel *>(dst.data[t.plane]) + t.dst_off;
    if (DIR) {
        // units run along the row (4 columns each), taps along y
        d += (size_t) u * 4;
#pragma unroll 1
        for (int i = 0; i < 4; i++) {
            LfStrided<pixel> io = { d + i, stride };
            lf_line(io, wd, E, I, H, F, bitdepth_min_8, bitdepth_max);
        }
    } else {
        // units run down the column (4 rows each), taps along x; this lane: line (lane & 3) of unit u
        d += ((size_t) u * 4 + (lane & 3)) * stride;
        LfWindow<pixel> io(d, wd == 16);
        lf_line(io, wd, E, I, H, F, bitdepth_min_8, bitdepth_max);
    }
}

} // namespace

// tasks[] (device): n tasks, all of direction `dir`
extern "C" int dav1d_hip_launch_lf(const DevPlanes *dst, int bpc, int dir, const Dav1dHipLfTask *tasks, int n, const uint8_t *lvl,
                                   int b4_stride, const uint8_t *lut_e, const uint8_t *lut_i, void *stream)
{
    if (n <= 0) return 0;
    LfLut lut;
    for (int k = 0; k < 64; k++) { lut.e[k] = lut_e[k]; lut.i[k] = lut_i[k]; }
    const int bitdepth_max = (1 << bpc) - 1;
    hipStream_t st = (hipStream_t) stream;
    if (dir) {
        const int grid = (n + 1) / 2;                    // two tasks per wave
        if (bpc == 8) hipLaunchKernelGGL((lf_kernel<uint8_t, 1>), dim3(grid), dim3(64), 0, st, *dst, tasks, n, lvl, b4_stride, lut, bitdepth_max);
        else hipLaunchKernelGGL((lf_kernel<uint16_t, 1>), dim3(grid), dim3(64), 0, st, *dst, tasks, n, lvl, b4_stride, lut, bitdepth_max);
    } else {
        const int grid = n * 2;                          // two waves per task
        if (bpc == 8) hipLaunchKernelGGL((lf_kernel<uint8_t, 0>), dim3(grid), dim3(64), 0, st, *dst, tasks, n, lvl, b4_stride, lut, bitdepth_max);
        else hipLaunchKernelGGL((lf_kernel<uint16_t, 0>), dim3(grid), dim3(64), 0, st, *dst, tasks, n, lvl, b4_stride, lut, bitdepth_max);
    }
    return hip_rc(hipGetLastError());
}

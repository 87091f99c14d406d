// The rarer motion-compensation entries for gfx950: warp8x8 / warp8x8t, the scaled put / prep
// variants, super-resolution resize and stand-alone emu_edge.
//
// Contracts: warp_affine_8x8_c / warp_affine_8x8t_c (reference src/mc_tmpl.c:799-866) as driven by
// warp_affine() (src/recon_tmpl.c:1115-1174); put/prep_8tap_scaled_c and put/prep_bilin_scaled_c
// (src/mc_tmpl.c:189-244, 307-357, 491-626) as driven by the scaled branch of mc()
// (src/recon_tmpl.c:990-1047); resize_c (src/mc_tmpl.c:918-944); emu_edge_c (:868-916).
// These are per-pixel gathers with position-dependent filters, so the mapping is one lane per output
// pixel reading its taps straight from (L1-resident) memory; edge emulation is a coordinate clamp.
#include "common.h"
#include "capi.h"
#include "av1_tables.h"
#include <type_traits>

namespace {

struct RefSet { DevPlanes r[8]; };

template <typename pixel>
__device__ __forceinline__ int ref_px(const pixel *p, const int stride, const int w, const int h, const int x, const int y) {
    return p[dv::iclip(y, 0, h - 1) * stride + dv::iclip(x, 0, w - 1)];
}

// ------------------------------------------------------------------------------------------ warp
template <typename pixel>
__global__ __launch_bounds__(64) void warp_kernel(const DevPlanes dst, const RefSet refs, const Dav1dHipWarpTask *__restrict__ tasks,
                                                  const int n, int16_t *__restrict__ prep, const int bitdepth_max)
{
    __shared__ int16_t mid[15 * 8];
    const int ti = blockIdx.x;
    if (ti >= n) return;
    const Dav1dHipWarpTask t = tasks[__builtin_amdgcn_readfirstlane(ti)];
    constexpr bool HBD = sizeof(pixel) == 2;
    const int ib = HBD ? 14 - (32 - __clz(bitdepth_max)) : 4;
    const DevPlanes &rp = refs.r[t.ref];
    const pixel *src = reinterpret_cast<const pixel *>(rp.data[t.plane]);
    const int rs = rp.stride[t.plane], rw = rp.w[t.plane], rh = rp.h[t.plane];
    const int lane = threadIdx.x;
    // 15 rows x 8 columns of horizontally filtered samples (src/mc_tmpl.c:808-820)
    for (int i = lane; i < 15 * 8; i += 64) {
        const int y = i >> 3, x = i & 7;
        const int tmx = t.mx + y * t.abcd[1] + x * t.abcd[0];
        const int8_t *f = &av1_mc_warp_filter[(64 + ((tmx + 512) >> 10)) * 8];
        int s = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) s += f[k] * ref_px(src, rs, rw, rh, t.src_x + x + k - 3, t.src_y + y - 3);
        mid[i] = (int16_t) ((s + ((1 << (7 - ib)) >> 1)) >> (7 - ib));
    }
    dv::wave_sync();
    {
        const int y = lane >> 3, x = lane & 7;
        const int tmy = t.my + y * t.abcd[3] + x * t.abcd[2];
        const int8_t *f = &av1_mc_warp_filter[(64 + ((tmy + 512) >> 10)) * 8];
        int s = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) s += f[k] * mid[(y + k) * 8 + x];
        if (t.kind == DAV1D_HIP_MC_PUT) {
            const int v = (s + ((1 << (7 + ib)) >> 1)) >> (7 + ib);
            reinterpret_cast<pixel *>(dst.data[t.plane])[t.dst_off + y * dst.stride[t.plane] + x] = (pixel) dv::iclip(v, 0, bitdepth_max);
        } else {
            prep[t.dst_off + y * t.tmp_stride + x] = (int16_t) (((s + 64) >> 7) - (HBD ? 8192 : 0));
        }
    }
}

// ---------------------------------------------------------------------------------- scaled put / prep
template <typename pixel>
__global__ __launch_bounds__(64) void mc_scaled_kernel(const DevPlanes dst, const RefSet refs, const Dav1dHipMcScaledTask *__restrict__ tasks,
                                                       const int n, int16_t *__restrict__ prep, const int bitdepth_max)
{
    const int ti = blockIdx.x;
    if (ti >= n) return;
    const Dav1dHipMcScaledTask t = tasks[__builtin_amdgcn_readfirstlane(ti)];
    constexpr bool HBD = sizeof(pixel) == 2;
    const int ib = HBD ? 14 - (32 - __clz(bitdepth_max)) : 4;
    const int bias = HBD ? 8192 : 0;
    const DevPlanes &rp = refs.r[t.ref];
    const pixel *src = reinterpret_cast<const pixel *>(rp.data[t.plane]);
    const int rs = rp.stride[t.plane], rw = rp.w[t.plane], rh = rp.h[t.plane];
    const bool bilin = t.filter_2d == 9;
    const bool as_put = t.kind != DAV1D_HIP_MC_PREP;
    // enum Filter2d -> (h, v) 8-tap families, 4-tap rows for w <= 4 / h <= 4 (src/mc_tmpl.c:115-123)
    const unsigned long long ht = 0x111222000ull, vt = 0x210210210ull;      // nibble f of each = type of Filter2d f
    const int h_type = (int) (ht >> (4 * t.filter_2d)) & 15, v_type = (int) (vt >> (4 * t.filter_2d)) & 15;
    const int hset = t.w > 4 ? h_type : 3 + (h_type & 1), vset = t.h > 4 ? v_type : 3 + (v_type & 1);
    for (int i = threadIdx.x; i < t.w * t.h; i += 64) {
        const int y = i / t.w, x = i % t.w;
        const int px = t.mx + x * t.dx, py = t.my + y * t.dy;
        const int ioff = px >> 10, fxi = (px & 0x3ff) >> 6, src_y = py >> 10, fyi = (py & 0x3ff) >> 6;
        int v;
        if (bilin) {
            // mid[r][x] = FILTER_BILIN_RND(src, ioff, fxi, 1, 4 - ib) on source rows src_y, src_y + 1
            int m[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int a = ref_px(src, rs, rw, rh, t.src_x + ioff, t.src_y + src_y + r);
                const int b = ref_px(src, rs, rw, rh, t.src_x + ioff + 1, t.src_y + src_y + r);
                m[r] = (int16_t) ((16 * a + fxi * (b - a) + ((1 << (4 - ib)) >> 1)) >> (4 - ib));
            }
            const int s = 16 * m[0] + fyi * (m[1] - m[0]);
            v = as_put ? (s + ((1 << (4 + ib)) >> 1)) >> (4 + ib) : ((s + 8) >> 4) - bias;
        } else {
            const int8_t *fh = fxi ? &av1_mc_subpel_filters[(hset * 15 + fxi - 1) * 8] : nullptr;
            const int8_t *fv = fyi ? &av1_mc_subpel_filters[(vset * 15 + fyi - 1) * 8] : nullptr;
            int mid[8];
            // the vertical filter spans source rows src_y - 3 .. src_y + 4 (mid_ptrs[0..7])
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int sy = t.src_y + src_y + r - 3;
                if (!fv && r != 3) { mid[r] = 0; continue; }
                int s;
                if (fh) {
                    s = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) s += fh[k] * ref_px(src, rs, rw, rh, t.src_x + ioff + k - 3, sy);
                    s = (s + ((1 << (6 - ib)) >> 1)) >> (6 - ib);
                } else {
                    s = ref_px(src, rs, rw, rh, t.src_x + ioff, sy) << ib;
                }
                mid[r] = (int16_t) s;
            }
            if (fv) {
                int s = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) s += fv[k] * mid[k];
                v = as_put ? (s + ((1 << (6 + ib)) >> 1)) >> (6 + ib) : ((s + 32) >> 6) - bias;
            } else {
                v = as_put ? (mid[3] + ((1 << ib) >> 1)) >> ib : mid[3] - bias;
            }
        }
        if (t.kind == DAV1D_HIP_MC_PUT)
            reinterpret_cast<pixel *>(dst.data[t.plane])[t.dst_off + y * dst.stride[t.plane] + x] = (pixel) dv::iclip(v, 0, bitdepth_max);
        else if (t.kind == DAV1D_HIP_MC_PUT_TMP)      // the `lap` prediction of obmc() from a scaled reference: pixels into the scratch arena
            reinterpret_cast<pixel *>(prep)[t.dst_off + y * t.w + x] = (pixel) dv::iclip(v, 0, bitdepth_max);
        else
            prep[t.dst_off + y * t.w + x] = (int16_t) v;
    }
}

// ------------------------------------------------------------------------------------------ resize
template <typename pixel>
__global__ __launch_bounds__(256) void resize_kernel(const DevPlanes dst, const DevPlanes src, const int plane, const int dst_w, const int y0, const int h,
                                                     const int src_w, const int dx, const int mx0, const int bitdepth_max)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = y0 + blockIdx.y;
    if (x >= dst_w) return;
    // position after x steps of "mx += dx; src_x += mx >> 14; mx &= 0x3fff" starting from (mx0, -1)
    const long long pos = (long long) mx0 + (long long) x * dx;
    const int src_x = -1 + (int) (pos >> 14), mx = (int) (pos & 0x3fff);
    const int8_t *F = &av1_resize_filter[(mx >> 8) * 8];
    const pixel *s = reinterpret_cast<const pixel *>(src.data[plane]) + y * src.stride[plane];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) sum += F[k] * s[dv::iclip(src_x + k - 3, 0, src_w - 1)];
    reinterpret_cast<pixel *>(dst.data[plane])[y * dst.stride[plane] + x] = (pixel) dv::iclip((-sum + 64) >> 7, 0, bitdepth_max);
}

// ---------------------------------------------------------------------------------------- emu_edge
template <typename pixel>
__global__ __launch_bounds__(256) void emu_edge_kernel(pixel *dst, const int dst_stride, const pixel *ref, const int ref_stride,
                                                       const int bw, const int bh, const int iw, const int ih, const int x0, const int y0)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < bw * bh; i += gridDim.x * 256) {
        const int y = i / bw, x = i % bw;
        dst[y * dst_stride + x] = ref[dv::iclip(y0 + y, 0, ih - 1) * ref_stride + dv::iclip(x0 + x, 0, iw - 1)];
    }
}

RefSet make_refs(const DevPlanes *refs, int n_refs) {
    RefSet rs;
    for (int i = 0; i < 8; i++) rs.r[i] = refs[i < n_refs ? i : 0];
    return rs;
}

} // namespace

extern "C" int dav1d_hip_launch_warp(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc, const Dav1dHipWarpTask *tasks, int n,
                                     int16_t *prep, void *stream)
{
    if (n <= 0) return 0;
    const int bm = (1 << bpc) - 1;
    if (bpc == 8) hipLaunchKernelGGL((warp_kernel<uint8_t>), dim3(n), dim3(64), 0, (hipStream_t) stream, *dst, make_refs(refs, n_refs), tasks, n, prep, bm);
    else hipLaunchKernelGGL((warp_kernel<uint16_t>), dim3(n), dim3(64), 0, (hipStream_t) stream, *dst, make_refs(refs, n_refs), tasks, n, prep, bm);
    return hip_rc(hipGetLastError());
}

extern "C" int dav1d_hip_launch_mc_scaled(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc,
                                          const Dav1dHipMcScaledTask *tasks, int n, int16_t *prep, void *stream)
{
    if (n <= 0) return 0;
    const int bm = (1 << bpc) - 1;
    if (bpc == 8) hipLaunchKernelGGL((mc_scaled_kernel<uint8_t>), dim3(n), dim3(64), 0, (hipStream_t) stream, *dst, make_refs(refs, n_refs), tasks, n, prep, bm);
    else hipLaunchKernelGGL((mc_scaled_kernel<uint16_t>), dim3(n), dim3(64), 0, (hipStream_t) stream, *dst, make_refs(refs, n_refs), tasks, n, prep, bm);
    return hip_rc(hipGetLastError());
}

extern "C" int dav1d_hip_launch_resize(const DevPlanes *dst, const DevPlanes *src, int bpc, int plane, int dst_w, int y0, int h, int src_w,
                                       int dx, int mx0, void *stream)
{
    const int bm = (1 << bpc) - 1;
    const dim3 grid((dst_w + 255) / 256, h);
    if (bpc == 8) hipLaunchKernelGGL((resize_kernel<uint8_t>), grid, dim3(256), 0, (hipStream_t) stream, *dst, *src, plane, dst_w, y0, h, src_w, dx, mx0, bm);
    else hipLaunchKernelGGL((resize_kernel<uint16_t>), grid, dim3(256), 0, (hipStream_t) stream, *dst, *src, plane, dst_w, y0, h, src_w, dx, mx0, bm);
    return hip_rc(hipGetLastError());
}

extern "C" int dav1d_hip_launch_emu_edge(void *dst, ptrdiff_t dst_stride, const void *ref, ptrdiff_t ref_stride, int bw, int bh,
                                         int iw, int ih, int x, int y, int bpc, void *stream)
{
    const int blocks = (bw * bh + 255) / 256;
    if (bpc == 8) hipLaunchKernelGGL((emu_edge_kernel<uint8_t>), dim3(blocks), dim3(256), 0, (hipStream_t) stream, (uint8_t *) dst, (int) dst_stride,
                                     (const uint8_t *) ref, (int) ref_stride, bw, bh, iw, ih, x, y);
    else hipLaunchKernelGGL((emu_edge_kernel<uint16_t>), dim3(blocks), dim3(256), 0, (hipStream_t) stream, (uint16_t *) dst, (int) (dst_stride / 2),
                            (const uint16_t *) ref, (int) (ref_stride / 2), bw, bh, iw, ih, x, y);
    return hip_rc(hipGetLastError());
}


// ------------------------------------------------------------------------------------------ tiled twin
// Raster plane -> 8x8 tiles of 64 consecutive pixels (Dav1dHipPicture.twin; what reads it: mc_body.h, TILED).  One wave moves 8
// rows x 64 pixels: lane l reads the 8 pixels (r = l >> 3, c = l & 7) of its row segment — the 8 lanes of a row read 128 (64)
// contiguous bytes — and writes row r of tile c: the wave's stores cover 8 whole tiles, 1 KB (512 bytes) of contiguous memory.
namespace {
// UNTILE: the other way (twin -> raster rows, for the rows [ty0 * 8, ...) the grid covers), same mapping
template <typename pixel, bool UNTILE>
__global__ __launch_bounds__(64) void retile_kernel(const pixel *__restrict__ src, pixel *__restrict__ twin, const int stride, const int h,
                                                    const int n_xg, const int ty0)
{
    typedef typename std::conditional<sizeof(pixel) == 2, uint4, uint2>::type piece_t;
    constexpr int ROWS = 8;                   // tile rows (of 8 picture rows) per wave: 8 KB (4 KB) in flight per wave
    const int g = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x);
    const int tyg = g / n_xg, xg = g - tyg * n_xg;
    const int lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
    const int x = xg * 64 + c * 8;
    if (x >= stride) return;
    const int n_ty = (h + 7) >> 3;
    piece_t v[ROWS];
    if (UNTILE) {
        // `src` = the twin, `twin` = the raster plane; rows at or below h are not the picture's (a caller-wrapped plane need not have them)
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
            const int ty = dv::imin(ty0 + tyg * ROWS + k, n_ty - 1);
            v[k] = *reinterpret_cast<const piece_t *>(src + (size_t) ty * 8 * stride + (size_t) (x >> 3) * 64 + r * 8);
        }
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
            const int y = (ty0 + tyg * ROWS + k) * 8 + r;
            if (y < h) *reinterpret_cast<piece_t *>(twin + (size_t) y * stride + x) = v[k];
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const int ty = tyg * ROWS + k;
        // rows below the plane (a caller-wrapped picture need not have them): the last row again
        const int y = dv::imin(ty * 8 + r, h - 1);
        v[k] = *reinterpret_cast<const piece_t *>(src + (size_t) y * stride + x);
    }
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const int ty = tyg * ROWS + k;
        if (ty < n_ty) *reinterpret_cast<piece_t *>(twin + (size_t) ty * 8 * stride + (size_t) (x >> 3) * 64 + r * 8) = v[k];
    }
}
} // namespace

extern "C" int dav1d_hip_launch_retile(const DevPlanes *src, void *const twin[3], int bpc, void *stream) {
    for (int pl = 0; pl < 3; pl++) {
        if (!src->data[pl]) continue;
        if (!twin[pl] || src->stride[pl] % 8 || src->h[pl] <= 0) return -EINVAL;
        const int n_xg = (src->stride[pl] + 63) / 64, n_ty = ((src->h[pl] + 7) / 8 + 7) / 8;      // 8 tile rows per wave
        const dim3 grid((unsigned) n_xg * (unsigned) n_ty), wave(64);
        if (bpc == 8)
            hipLaunchKernelGGL((retile_kernel<uint8_t, false>), grid, wave, 0, (hipStream_t) stream, (const uint8_t *) src->data[pl], (uint8_t *) twin[pl],
                               src->stride[pl], src->h[pl], n_xg, 0);
        else
            hipLaunchKernelGGL((retile_kernel<uint16_t, false>), grid, wave, 0, (hipStream_t) stream, (const uint16_t *) src->data[pl], (uint16_t *) twin[pl],
                               src->stride[pl], src->h[pl], n_xg, 0);
    }
    return hip_rc(hipGetLastError());
}

// Twin -> raster planes (`dst`: the raster planes with their strides and heights), rows [row0[pl], row1[pl]) of each plane widened to
// whole tile rows; plane_mask: bit pl = untile plane pl.  What a picture that lives in its twin only (DAV1D_HIP_TWIN_ONLY) goes
// through before anything that reads raster planes — the fetch to the host first of all (src/picture.c:46-63's layout at the output only).
extern "C" int dav1d_hip_launch_untile(const DevPlanes *dst, void *const twin[3], int bpc, const int row0[3], const int row1[3], int plane_mask, void *stream) {
    for (int pl = 0; pl < 3; pl++) {
        if (!dst->data[pl] || !(plane_mask >> pl & 1)) continue;
        if (!twin[pl] || dst->stride[pl] % 8 || dst->h[pl] <= 0) return -EINVAL;
        const int h = dst->h[pl];
        const int r0 = row0 ? (row0[pl] < 0 ? 0 : row0[pl]) : 0, r1 = row1 ? (row1[pl] > h ? h : row1[pl]) : h;
        if (r1 <= r0) continue;
        const int ty0 = r0 >> 3, nty = ((r1 + 7) >> 3) - ty0;
        const int n_xg = (dst->stride[pl] + 63) / 64, n_tyg = (nty + 7) / 8;
        const dim3 grid((unsigned) n_xg * (unsigned) n_tyg), wave(64);
        if (bpc == 8)
            hipLaunchKernelGGL((retile_kernel<uint8_t, true>), grid, wave, 0, (hipStream_t) stream, (const uint8_t *) twin[pl], (uint8_t *) dst->data[pl],
                               dst->stride[pl], h, n_xg, ty0);
        else
            hipLaunchKernelGGL((retile_kernel<uint16_t, true>), grid, wave, 0, (hipStream_t) stream, (const uint16_t *) twin[pl], (uint16_t *) dst->data[pl],
                               dst->stride[pl], h, n_xg, ty0);
    }
    return hip_rc(hipGetLastError());
}

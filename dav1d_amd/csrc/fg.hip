// Film grain synthesis for gfx950.
//
// Contract = the reference's dav1d_apply_grain (src/fg_apply_tmpl.c:97-241) over its DSP
// entries: generate_grain_y / generate_grain_uv (src/filmgrain_tmpl.c:50-154) build the
// 73x82 (chroma: 38x44 when subsampled) grain templates, fgy_32x32xn / fguv_32x32xn
// (:169-413) add scaled grain per 32x32 block with the 2-pixel overlap blends.
//
// Grain templates (one workgroup, one wave per plane): the 16-bit LFSR is linear, so the
// state at the start of every template row is obtained by applying the "advance one row"
// GF(2) matrix (built by stepping the 16 basis states) row after row; rows then draw their
// Gaussian samples in parallel.  The auto-regressive filter runs as a wavefront: lane = row,
// each row trails the row above by lag + 1 samples.
// Application: one wave per 32x32 luma block (and its chroma blocks); the per-block random
// offsets are the k-th outputs of the per-row LFSR, recomputed by every wave (<= 241 steps).
#include "common.h"
#include "capi.h"
#include <type_traits>
#include "av1_tables.h"
#include <string.h>

namespace {

enum { GW = 82, GH = 73, SGW = 44, SGH = 38 };

__device__ __forceinline__ unsigned lfsr_step(const unsigned r) {
    const unsigned bit = ((r >> 0) ^ (r >> 1) ^ (r >> 3) ^ (r >> 12)) & 1;
    return (r >> 1) | (bit << 15);
}
__device__ __forceinline__ int round2(const int x, const int shift) { return (x + ((1 << shift) >> 1)) >> shift; }

struct FgParams {                 // the scalar part of Dav1dFilmGrainData the kernels need
    unsigned seed;
    int num_y_points, chroma_scaling_from_luma, num_uv_points[2];
    int scaling_shift, ar_coeff_lag, ar_coeff_shift, grain_scale_shift;
    int uv_mult[2], uv_luma_mult[2], uv_offset[2];
    int overlap_flag, clip_to_restricted_range;
    int8_t ar_coeffs_y[24];
    int8_t ar_coeffs_uv[2][28];
};

// The auto-regressive pass of one template as a wavefront: lane = row, row y trails row y-1 by LAG+1 columns, so everything a
// sample needs is final when its turn comes (src/filmgrain_tmpl.c:72-91, 123-153).  LAG is a compile-time constant so that the
// tap loops unroll and the coefficients live in scalar registers; a lane keeps its own last LAG results and a sliding
// (2 LAG + 1)-wide window of each of the LAG rows above in registers, so a step reads LAG new samples from LDS, not
// 2 LAG (LAG + 1).
template <int LAG>
__device__ void ar_filter(int16_t *lut, const int16_t *lut_y, const FgParams &p, const int pl, const int subx, const int suby,
                          const int W, const int H, const int grain_min, const int grain_max, const int lane)
{
    constexpr int pad = 3, NC = 2 * LAG * (LAG + 1), WW = 2 * LAG + 1;
    int coef[NC + 1];
    {
        const int8_t *c = pl ? p.ar_coeffs_uv[pl - 1] : p.ar_coeffs_y;
#pragma unroll
        for (int k = 0; k <= NC; k++) coef[k] = c[k];
    }
    const bool with_luma = pl && p.num_y_points;
    for (int y0 = pad; y0 < H; y0 += 64) {
        const int rows = dv::imin(64, H - y0);
        const int y = y0 + lane;
        const int steps = (W - 2 * pad) + (rows - 1) * (LAG + 1);
        int win[LAG > 0 ? LAG : 1][WW], own[LAG > 0 ? LAG : 1];
#pragma unroll
        for (int k = 0; k < (LAG > 0 ? LAG : 1); k++) {
            own[k] = 0;
#pragma unroll
            for (int j = 0; j < WW; j++) win[k][j] = 0;
        }
        for (int s = 0; s < steps; s++) {
            const int x = pad + s - lane * (LAG + 1);
            if (lane < rows && x >= pad && x < W - pad) {
                if (x == pad) {
                    // first sample of the row: fill the windows (columns 0 .. 2 LAG of the rows above, 0 .. LAG-1 of this row)
#pragma unroll
                    for (int k = 0; k < LAG; k++) {
                        own[k] = lut[y * GW + x - LAG + k];
#pragma unroll
                        for (int j = 0; j < WW; j++) win[k][j] = lut[(y - LAG + k) * GW + x - LAG + j];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < LAG; k++) {
#pragma unroll
                        for (int j = 0; j + 1 < WW; j++) win[k][j] = win[k][j + 1];
                        win[k][WW - 1] = lut[(y - LAG + k) * GW + x + LAG];
                    }
                }
                // one partial sum per row: independent multiply-add chains instead of one 24-deep chain (a lone wave has
                // nothing else to hide the ALU latency behind)
                int part[LAG + 1];
#pragma unroll
                for (int k = 0; k < LAG; k++) {
                    part[k] = 0;
#pragma unroll
                    for (int j = 0; j < WW; j++) part[k] += coef[k * WW + j] * win[k][j];
                }
                part[LAG] = 0;
#pragma unroll
                for (int j = 0; j < LAG; j++) part[LAG] += coef[LAG * WW + j] * own[j];
                int sum = 0;
#pragma unroll
                for (int k = 0; k <= LAG; k++) sum += part[k];
                if (with_luma) {
                    int luma = 0;
                    const int lx = ((x - pad) << subx) + pad, ly = ((y - pad) << suby) + pad;
                    for (int i = 0; i <= suby; i++)
                        for (int j = 0; j <= subx; j++) luma += lut_y[(ly + i) * GW + lx + j];
                    sum += round2(luma, subx + suby) * coef[NC];
                }
                const int v = dv::iclip(lut[y * GW + x] + round2(sum, p.ar_coeff_shift), grain_min, grain_max);
                lut[y * GW + x] = (int16_t) v;
#pragma unroll
                for (int j = 0; j + 1 < LAG; j++) own[j] = own[j + 1];
                if (LAG > 0) own[LAG - 1] = v;
            }
            dv::wave_sync();
        }
    }
}

// one wave builds template `pl` (0 luma, 1/2 chroma) in `lut` (int16 [74][82])
__device__ void gen_template(int16_t *lut, const int16_t *lut_y, const FgParams &p, const int pl, const int subx, const int suby,
                             const int bitdepth_min_8, uint16_t *cols, uint16_t *rowstart, const int16_t *gauss, const int lane)
{
    const int W = (pl && subx) ? SGW : GW, H = (pl && suby) ? SGH : GH;
    const int shift = 4 - bitdepth_min_8 + p.grain_scale_shift;
    const int grain_ctr = 128 << bitdepth_min_8, grain_min = -grain_ctr, grain_max = grain_ctr - 1;
    unsigned seed = p.seed;
    if (pl) seed ^= pl == 2 ? 0x49d8 : 0xb524;
    // ---- "advance W steps" matrix, columns = images of the basis states
    if (lane < 16) {
        unsigned s = 1u << lane;
        for (int i = 0; i < W; i++) s = lfsr_step(s);
        cols[lane] = (uint16_t) s;
    }
    dv::wave_sync();
    if (lane == 0) {
        unsigned s = seed;
        for (int y = 0; y < H; y++) {
            rowstart[y] = (uint16_t) s;
            unsigned n = 0;
            for (int b = 0; b < 16; b++) if (s & (1u << b)) n ^= cols[b];
            s = n;
        }
    }
    dv::wave_sync();
    for (int y = lane; y < H; y += 64) {
        unsigned s = rowstart[y];
        for (int x = 0; x < W; x++) {
            s = lfsr_step(s);
            lut[y * GW + x] = (int16_t) round2(gauss[(s >> 5) & 2047], shift);
        }
    }
    dv::wave_sync();
    // ---- auto-regressive filter as a wavefront over rows (src/filmgrain_tmpl.c:72-91, 123-153); the lag is a
    // compile-time constant of ar_filter so that the tap loops unroll and the coefficients stay in scalar registers
    switch (p.ar_coeff_lag) {
    case 0: ar_filter<0>(lut, lut_y, p, pl, subx, suby, W, H, grain_min, grain_max, lane); break;
    case 1: ar_filter<1>(lut, lut_y, p, pl, subx, suby, W, H, grain_min, grain_max, lane); break;
    case 2: ar_filter<2>(lut, lut_y, p, pl, subx, suby, W, H, grain_min, grain_max, lane); break;
    default: ar_filter<3>(lut, lut_y, p, pl, subx, suby, W, H, grain_min, grain_max, lane); break;
    }
}

// part 3: like 1 / 2 with the plane taken from blockIdx.x and skipped when dav1d_apply_grain would not build it: both
// chroma templates side by side, after a part -1 launch finished the luma template
__global__ __launch_bounds__(64) void fg_gen_kernel(int16_t *luts, const FgParams p, const int layout, const int bitdepth_min_8, int part)
{
    __shared__ uint16_t cols[16], rowstart[GH];
    __shared__ int16_t t_y[(GH + 1) * GW], t_c[(GH + 1) * GW];     // templates are built in LDS, then copied out
    __shared__ int16_t gauss[2048];                                 // the Gaussian table: random lookups, so LDS not memory
    const int lane = threadIdx.x;
    const int subx = layout != DAV1D_HIP_LAYOUT_I444, suby = layout == DAV1D_HIP_LAYOUT_I420;
    if (part == 3) {
        part = 1 + blockIdx.x;
        if (layout == DAV1D_HIP_LAYOUT_I400 || !(p.num_uv_points[part - 1] || p.chroma_scaling_from_luma)) return;
    }
    for (int i = lane; i < 2048; i += 64) gauss[i] = av1_gaussian_sequence[i];
    for (int i = lane; i < (GH + 1) * GW; i += 64) t_y[i] = t_c[i] = 0;
    dv::wave_sync();
    if (part > 0) {
        for (int i = lane; i < (GH + 1) * GW; i += 64) t_y[i] = luts[i];
    } else {
        gen_template(t_y, t_y, p, 0, subx, suby, bitdepth_min_8, cols, rowstart, gauss, lane);
        dv::wave_sync();
        for (int i = lane; i < (GH + 1) * GW; i += 64) luts[i] = t_y[i];
    }
    if (part < 0) return;
    for (int pl = 1; pl <= 2; pl++) {
        dv::wave_sync();
        if (part ? pl == part : (layout != DAV1D_HIP_LAYOUT_I400 && (p.num_uv_points[pl - 1] || p.chroma_scaling_from_luma))) {
            gen_template(t_c, t_y, p, pl, subx, suby, bitdepth_min_8, cols, rowstart, gauss, lane);
            dv::wave_sync();
            for (int i = lane; i < (GH + 1) * GW; i += 64) luts[pl * (GH + 1) * GW + i] = t_c[i];
        }
    }
}

// sample_lut, src/filmgrain_tmpl.c:156-167
__device__ __forceinline__ int sample_lut(const int16_t *lut, const int randval, const int subx, const int suby,
                                          const int bx, const int by, const int x, const int y)
{
    const int offx = 3 + (2 >> subx) * (3 + (randval >> 4));
    const int offy = 3 + (2 >> suby) * (3 + (randval & 0xF));
    return lut[(offy + y + (32 >> suby) * by) * GW + offx + x + (32 >> subx) * bx];
}

// Per-block random offsets (src/filmgrain_tmpl.c:192-214): block bx of block row r uses the (bx+1)-th output of an LFSR
// seeded from r.  One lane walks one row's sequence once; the blocks then look their offsets up instead of each
// re-running the generator from the start of the row.
__global__ __launch_bounds__(64) void fg_offsets_kernel(uint8_t *__restrict__ offs, const unsigned seed, const int row0, const int nrows,
                                                        const int nblk)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= nrows) return;
    const int row_num = row0 + r;
    unsigned s = seed;
    s ^= ((row_num * 37 + 178) & 0xFF) << 8;
    s ^= (row_num * 173 + 105) & 0xFF;
    for (int k = 0; k < nblk; k++) {
        s = lfsr_step(s);
        offs[r * nblk + k] = (uint8_t) ((s >> 8) & 0xFF);
    }
}

template <typename pixel>
__global__ __launch_bounds__(256) void fg_apply_kernel(const DevPlanes dst, const DevPlanes src, const int16_t *__restrict__ luts,
                                                      const uint8_t *__restrict__ scaling, const int scaling_size, const FgParams p,
                                                      const int layout, const int is_id, const int bitdepth_max,
                                                      const int row_base, const int only_pl,
                                                      const uint8_t *__restrict__ offs, const int offs_row0, const int offs_stride)
{
    // row_base / only_pl: the table-level entries run one block row of one plane on pictures that hold just that row
    // four waves = four neighbouring blocks of a block row: they share ONE copy of the scaling table in LDS (a copy per 32x32 block
    // was twice the block's own pixels in table traffic)
    const int bxi = (int) blockIdx.x * 4 + ((int) threadIdx.x >> 6), row_num = row_base + blockIdx.y, pl = only_pl < 0 ? (int) blockIdx.z : only_pl;
    const int prow = blockIdx.y;          // block row inside the pictures
    const int lane = threadIdx.x & 63;
    const int bitdepth_min_8 = (32 - __clz(bitdepth_max)) - 8;
    const int grain_ctr = 128 << bitdepth_min_8, grain_min = -grain_ctr, grain_max = grain_ctr - 1;
    const int sx = pl && layout != DAV1D_HIP_LAYOUT_I444, sy = pl && layout == DAV1D_HIP_LAYOUT_I420;
    if (pl == 0 && !p.num_y_points) return;
    if (pl && !(p.num_uv_points[pl - 1] || p.chroma_scaling_from_luma)) return;
    const int uv = pl - 1;

    int min_value = 0, max_value = bitdepth_max;
    if (p.clip_to_restricted_range) {
        min_value = 16 << bitdepth_min_8;
        max_value = ((pl && !is_id) ? 240 : 235) << bitdepth_min_8;
    }
    // plane geometry
    const int pw = pl ? (src.w[0] + sx) >> sx : src.w[0];          // cpw / out->p.w
    const int ph_luma = src.h[0];
    const int bh = pl ? (dv::imin(ph_luma - prow * 32, 32) + sy) >> sy : dv::imin(ph_luma - prow * 32, 32);
    const int bstep = 32 >> sx;
    const int bx = bxi * bstep;
    if (bh <= 0) return;                                  // (the whole workgroup)
    const bool active = bx < pw;                          // a wave past the end of the row still helps with the table
    const int bw = active ? dv::imin(bstep, pw - bx) : 0;

    // per-block random offsets: k-th output of the row LFSRs (src/filmgrain_tmpl.c:192-214)
    // the k-th outputs of the row LFSRs were tabulated by fg_offsets_kernel: [row][block]
    const bool two_rows = p.overlap_flag && row_num > 0;
    const uint8_t *orow = offs + (row_num - offs_row0) * offs_stride;
    const int bxo = active ? bxi : 0;
    const int off_cur = orow[bxo], off_left = bxo ? orow[bxo - 1] : 0;
    const int off_cur_up = two_rows ? orow[bxo - offs_stride] : 0, off_left_up = (two_rows && bxo) ? orow[bxo - 1 - offs_stride] : 0;
    const int ystart = (p.overlap_flag && row_num) ? dv::imin(2 >> sy, bh) : 0;
    const int xstart = (p.overlap_flag && bxi) ? dv::imin(2 >> sx, bw) : 0;
    // overlap weights: full resolution {27,17},{17,27}; subsampled {23,22}
    const int16_t *lut = luts + pl * (GH + 1) * GW;
    const uint8_t *sc = scaling + (size_t) ((pl && !p.chroma_scaling_from_luma) ? pl : 0) * scaling_size;

    const pixel *__restrict__ const sp = reinterpret_cast<const pixel *>(src.data[pl]);
    pixel *__restrict__ const dp = reinterpret_cast<pixel *>(dst.data[pl]);
    const pixel *__restrict__ const lp = reinterpret_cast<const pixel *>(src.data[0]);
    const int y0 = pl ? (prow * 32) >> sy : prow * 32;

    // the scaling table moves to LDS once per block: the per-pixel lookup then costs an LDS read instead of a second
    // dependent trip to memory
    __shared__ __attribute__((aligned(16))) uint8_t sc_s[4096];
    for (int i = (int) threadIdx.x * 16; i < scaling_size; i += 256 * 16) *reinterpret_cast<uint4 *>(sc_s + i) = *reinterpret_cast<const uint4 *>(sc + i);
    __syncthreads();
    if (!active) return;

    // lane = (group of four columns, row): 8 rows of 32 or 16 rows of 16 pixels per pass; a lane fetches its four source pixels
    // (and, for chroma, the eight luma pixels under them) with one load each and stores four pixels at once — round 1 moved one
    // pixel per lane and instruction (0.16 ms per 8K frame, latency-bound)
    const int qpr = bstep >> 2, rpp = 64 / qpr;        // lanes per row, rows per pass
    const int x0 = (lane % qpr) * 4, yl = lane / qpr;
    const int ss = src.stride[pl], ds = dst.stride[pl], ls = src.stride[0];
    constexpr bool HBD = sizeof(pixel) == 2;
    // every pass's wide loads are issued before the first pass computes (at most four passes: 32 rows, at least 8 per pass): a wave is
    // one trip to memory long instead of one per pass
    typedef typename std::conditional<HBD, uint2, uint32_t>::type quad_t;
    quad_t pre_s[4], pre_la[4], pre_lb[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int y = q * rpp + yl;
        pre_s[q] = quad_t(); pre_la[q] = quad_t(); pre_lb[q] = quad_t();
        if (q * rpp >= bh || y >= bh || x0 + 4 > bw) continue;
        pre_s[q] = *reinterpret_cast<const quad_t *>(sp + (y0 + y) * ss + bx + x0);
        if (pl) {
            const int lx = (bx + x0) << sx, ly = (prow * 32) + (y << sy);
            if (lx + (4 << sx) <= src.w[0]) {
                pre_la[q] = *reinterpret_cast<const quad_t *>(lp + ly * ls + lx);
                if (sx) pre_lb[q] = *reinterpret_cast<const quad_t *>(lp + ly * ls + lx + 4);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int yb = q * rpp;
        if (yb >= bh) continue;
        const int y = yb + yl;
        if (y >= bh || x0 >= bw) continue;
        const bool whole = x0 + 4 <= bw;
        int s0[4], lum[4] = { 0, 0, 0, 0 };
        const pixel *srow = sp + (y0 + y) * ss + bx + x0;
        if (whole) {
            if constexpr (HBD) { const uint2 v = pre_s[q]; s0[0] = v.x & 0xffff; s0[1] = v.x >> 16; s0[2] = v.y & 0xffff; s0[3] = v.y >> 16; }
            else { const uint32_t v = pre_s[q]; s0[0] = v & 0xff; s0[1] = v >> 8 & 0xff; s0[2] = v >> 16 & 0xff; s0[3] = v >> 24; }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) s0[k] = x0 + k < bw ? (int) srow[k] : 0;
        }
        if (pl) {
            // luma co-located average; the reference extends the luma row by one pixel for odd widths (src/fg_apply_tmpl.c:193-199)
            const int lx = (bx + x0) << sx, ly = (prow * 32) + (y << sy);
            const pixel *lrow = lp + ly * ls;
            if (whole && lx + (4 << sx) <= src.w[0]) {
                int l8[8];
                if constexpr (HBD) {
                    const uint2 a = pre_la[q];
                    l8[0] = a.x & 0xffff; l8[1] = a.x >> 16; l8[2] = a.y & 0xffff; l8[3] = a.y >> 16;
                    if (sx) { const uint2 b = pre_lb[q]; l8[4] = b.x & 0xffff; l8[5] = b.x >> 16; l8[6] = b.y & 0xffff; l8[7] = b.y >> 16; }
                } else {
                    const uint32_t a = pre_la[q];
                    l8[0] = a & 0xff; l8[1] = a >> 8 & 0xff; l8[2] = a >> 16 & 0xff; l8[3] = a >> 24;
                    if (sx) { const uint32_t b = pre_lb[q]; l8[4] = b & 0xff; l8[5] = b >> 8 & 0xff; l8[6] = b >> 16 & 0xff; l8[7] = b >> 24; }
                }
#pragma unroll
                for (int k = 0; k < 4; k++) lum[k] = sx ? (l8[2 * k] + l8[2 * k + 1] + 1) >> 1 : l8[k];
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (x0 + k >= bw) continue;
                    const int lxk = (bx + x0 + k) << sx;
                    int avg = lrow[dv::imin(lxk, src.w[0] - 1)];
                    if (sx) avg = (avg + lrow[dv::imin(lxk + 1, src.w[0] - 1)] + 1) >> 1;
                    lum[k] = avg;
                }
            }
        }
        int res[4];
        const int wya = sy ? 23 : (y == 0 ? 27 : 17), wyb = sy ? 22 : (y == 0 ? 17 : 27);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = x0 + k;
            res[k] = 0;
            if (x >= bw) continue;
            const int wxa = sx ? 23 : (x == 0 ? 27 : 17), wxb = sx ? 22 : (x == 0 ? 17 : 27);
            int grain = sample_lut(lut, off_cur, sx, sy, 0, 0, x, y);
            if (x < xstart) {
                const int old = sample_lut(lut, off_left, sx, sy, 1, 0, x, y);
                grain = dv::iclip(round2(old * wxa + grain * wxb, 5), grain_min, grain_max);
            }
            if (y < ystart) {
                int top = sample_lut(lut, off_cur_up, sx, sy, 0, 1, x, y);
                if (x < xstart) {
                    const int old = sample_lut(lut, off_left_up, sx, sy, 1, 1, x, y);
                    top = dv::iclip(round2(old * wxa + top * wxb, 5), grain_min, grain_max);
                }
                grain = dv::iclip(round2(top * wya + grain * wyb, 5), grain_min, grain_max);
            }
            int val = s0[k];
            if (pl) {
                val = lum[k];
                if (!p.chroma_scaling_from_luma) {
                    const int combined = lum[k] * p.uv_luma_mult[uv] + s0[k] * p.uv_mult[uv];
                    val = dv::iclip((combined >> 6) + (p.uv_offset[uv] * (1 << bitdepth_min_8)), 0, bitdepth_max);
                }
            }
            const int noise = round2(sc_s[val] * grain, p.scaling_shift);
            res[k] = dv::iclip(s0[k] + noise, min_value, max_value);
        }
        pixel *drow = dp + (y0 + y) * ds + bx + x0;
        if (whole) {
            if (HBD) *reinterpret_cast<uint2 *>(drow) = make_uint2((uint32_t) res[0] | (uint32_t) res[1] << 16, (uint32_t) res[2] | (uint32_t) res[3] << 16);
            else *reinterpret_cast<uint32_t *>(drow) = (uint32_t) res[0] | (uint32_t) res[1] << 8 | (uint32_t) res[2] << 16 | (uint32_t) res[3] << 24;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) if (x0 + k < bw) drow[k] = (pixel) res[k];
        }
    }
}

FgParams make_params(const Dav1dHipFilmGrainData *d) {
    FgParams p;
    memset(&p, 0, sizeof(p));
    p.seed = d->seed; p.num_y_points = d->num_y_points; p.chroma_scaling_from_luma = d->chroma_scaling_from_luma;
    p.scaling_shift = d->scaling_shift; p.ar_coeff_lag = d->ar_coeff_lag; p.ar_coeff_shift = (int) d->ar_coeff_shift;
    p.grain_scale_shift = d->grain_scale_shift; p.overlap_flag = d->overlap_flag; p.clip_to_restricted_range = d->clip_to_restricted_range;
    for (int i = 0; i < 2; i++) {
        p.num_uv_points[i] = d->num_uv_points[i];
        p.uv_mult[i] = d->uv_mult[i]; p.uv_luma_mult[i] = d->uv_luma_mult[i]; p.uv_offset[i] = d->uv_offset[i];
        memcpy(p.ar_coeffs_uv[i], d->ar_coeffs_uv[i], 28);
    }
    memcpy(p.ar_coeffs_y, d->ar_coeffs_y, 24);
    return p;
}

} // namespace

extern "C" int dav1d_hip_launch_fg_gen(int16_t *luts, const Dav1dHipFilmGrainData *data, int bpc, int layout, void *stream)
{
    // luma first, then both chroma templates side by side (they filter against the finished luma template)
    hipLaunchKernelGGL(fg_gen_kernel, dim3(1), dim3(64), 0, (hipStream_t) stream, luts, make_params(data), layout, bpc - 8, -1);
    if (layout != DAV1D_HIP_LAYOUT_I400)
        hipLaunchKernelGGL(fg_gen_kernel, dim3(2), dim3(64), 0, (hipStream_t) stream, luts, make_params(data), layout, bpc - 8, 3);
    return hip_rc(hipGetLastError());
}

extern "C" int dav1d_hip_launch_fg_gen_part(int16_t *luts, const Dav1dHipFilmGrainData *data, int bpc, int layout, int part, void *stream)
{
    hipLaunchKernelGGL(fg_gen_kernel, dim3(1), dim3(64), 0, (hipStream_t) stream, luts, make_params(data), layout, bpc - 8, part);
    return hip_rc(hipGetLastError());
}

// offs: device scratch of at least ceil(w / 32) * ceil(h / 32) bytes for the offset table
extern "C" int dav1d_hip_launch_fg_apply(const DevPlanes *dst, const DevPlanes *src, const int16_t *luts, const uint8_t *scaling,
                                         int scaling_size, const Dav1dHipFilmGrainData *data, int bpc, int layout, int is_id,
                                         uint8_t *offs, void *stream)
{
    const int bitdepth_max = (1 << bpc) - 1;
    const int nbx = (src->w[0] + 31) / 32;
    const dim3 grid((nbx + 3) / 4, (src->h[0] + 31) / 32, layout == DAV1D_HIP_LAYOUT_I400 ? 1 : 3);        // four blocks of a row per workgroup
    const FgParams p = make_params(data);
    const int offs_row0 = 0, offs_stride = nbx;
    hipLaunchKernelGGL(fg_offsets_kernel, dim3((grid.y + 63) / 64), dim3(64), 0, (hipStream_t) stream, offs, p.seed, 0, (int) grid.y, nbx);
    if (bpc == 8)
        hipLaunchKernelGGL((fg_apply_kernel<uint8_t>), grid, dim3(256), 0, (hipStream_t) stream, *dst, *src, luts, scaling, scaling_size, p, layout, is_id, bitdepth_max, 0, -1, offs, offs_row0, offs_stride);
    else
        hipLaunchKernelGGL((fg_apply_kernel<uint16_t>), grid, dim3(256), 0, (hipStream_t) stream, *dst, *src, luts, scaling, scaling_size, p, layout, is_id, bitdepth_max, 0, -1, offs, offs_row0, offs_stride);
    return hip_rc(hipGetLastError());
}

// one block row (`row_num`) of one plane, the pictures holding only that row
extern "C" int dav1d_hip_launch_fg_apply_rows(const DevPlanes *dst, const DevPlanes *src, const int16_t *luts, const uint8_t *scaling,
                                              int scaling_size, const Dav1dHipFilmGrainData *data, int bpc, int layout, int is_id,
                                              int row_num, int pl, uint8_t *offs, void *stream)
{
    const int bitdepth_max = (1 << bpc) - 1;
    const int nbx = (src->w[0] + 31) / 32;
    const dim3 grid((nbx + 3) / 4, 1, 1);
    const FgParams p = make_params(data);
    // two table rows: the row above (its offsets feed the vertical overlap) and this one
    const int offs_row0 = row_num - 1, offs_stride = nbx;
    hipLaunchKernelGGL(fg_offsets_kernel, dim3(1), dim3(64), 0, (hipStream_t) stream, offs, p.seed, offs_row0, 2, nbx);
    if (bpc == 8)
        hipLaunchKernelGGL((fg_apply_kernel<uint8_t>), grid, dim3(256), 0, (hipStream_t) stream, *dst, *src, luts, scaling, scaling_size, p, layout, is_id, bitdepth_max, row_num, pl, offs, offs_row0, offs_stride);
    else
        hipLaunchKernelGGL((fg_apply_kernel<uint16_t>), grid, dim3(256), 0, (hipStream_t) stream, *dst, *src, luts, scaling, scaling_size, p, layout, is_id, bitdepth_max, row_num, pl, offs, offs_row0, offs_stride);
    return hip_rc(hipGetLastError());
}

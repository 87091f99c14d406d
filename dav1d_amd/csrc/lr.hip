// Batched loop restoration for gfx950 (Wiener + self-guided).
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
#include "av1_tables.h"
#include <vector>
#include <algorithm>

namespace {

enum { LR_SEG = 64 };        // output rows per wave: the whole stripe (16-row segments were measured 20 % slower: 6 extra
                             // rows of horizontal filtering per segment outweigh the shorter chains).  Also measured and
                             // dropped: fetching rows two iterations ahead (+6 %), one unaligned 8-pixel fetch per lane and
                             // row instead of seven 1-pixel ones (+60 %: neighbouring lanes read overlapping 16-byte pieces)

// this wave's pixels are out (written back as far as the device's memory: the per-XCD L2s do not see each other's lines otherwise), its
// band's counter goes up, and whoever completes the band tells the host
// (The pixels of a signalling launch leave through agent-scope stores — written through this XCD's L2 — so that "out" only takes waiting
// for the wave's own stores.  A release fence here instead writes back EVERY dirty line of the XCD's L2, once per wave: measured at 8K
// that was most of the 0.2 ms a frame with a listener cost more than one without.)
__device__ __forceinline__ void band_done(const BandSignal &sg, const int band) {
    if (!sg.cnt) return;
    dv::stores_done();
    if ((threadIdx.x & 63) == 0) {
        const unsigned old = atomicAdd(sg.cnt + band, 1u);
        if (old + 1 == sg.target[band]) {
#ifdef DAV1D_HIP_EMU
            sg.host_flags[band] = sg.seq;
#else
            __hip_atomic_store(sg.host_flags + band, sg.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
        }
    }
}

// SIG: the launch tells the host band by band where it is (frame.hip frame_lr_banded): its pixels leave write-through
template <typename pixel, bool SIG>
__global__ __launch_bounds__(64) void wiener_kernel(const DevPlanes dst, const DevPlanes src, const DevPlanes lpf,
                                                    const Dav1dHipLrTask *__restrict__ tasks, const int n, const int bitdepth_max, const BandSignal sig)
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

    // a wave filters SEG output rows: rows seg0 .. seg1-1 need the horizontally filtered virtual rows seg0-3 .. seg1+2, so
    // splitting a stripe costs 6 extra rows per segment and buys shorter serial chains and SEG-fold more waves
    const int seg0 = blockIdx.z * LR_SEG, seg1 = dv::imin(seg0 + LR_SEG, h);
    if (seg0 >= h) return;
    for (int r = seg0 - 3; r < seg1 + 3; r++) {
        // which picture row feeds virtual row r
        const pixel *row;
        if (r < 0) {
            if (edges & 4) row = l + (t.y - (r == -1 ? 1 : 2)) * ls;
            else row = s + t.y * ss;
        } else if (r >= h) {
            // (a row beyond the plane's last is that last row once more: backup_lpf's n_lines, src/lf_apply_tmpl.c:77-97)
            if (use_bottom) row = l + dv::imin(t.y + h + (r == h ? 0 : 1), lpf.h[pl] - 1) * ls;
            else row = s + (t.y + h - 1) * ss;
        } else {
            row = s + (t.y + r) * ss;
        }
        int sum = 1 << (bitdepth + 6);
        if (!HBD) sum += row[t.x + xc] * 128;
#pragma unroll
        for (int i = 0; i < 7; i++) sum += row[col[i]] * fh[i];
        sum = dv::clamp3((sum + rounding_off_h) >> round_bits_h, 0, clip_limit - 1);
#pragma unroll
        for (int i = 0; i < 6; i++) win[i] = win[i + 1];
        win[6] = sum;
        if (r >= seg0 + 3) {
            int v = -round_offset;
#pragma unroll
            for (int k = 0; k < 7; k++) v = dv::mad_i24(win[k], fv[k], v);       // win < 2^16 (clip_limit), 16-bit taps: the full-rate multiplier
            const pixel px_out = (pixel) dv::clamp3((v + rounding_off_v) >> round_bits_v, 0, bitdepth_max);
            if (active) { if (SIG) dv::st_coherent(d + (r - 3) * dst.stride[pl], px_out); else d[(r - 3) * dst.stride[pl]] = px_out; }
        }
    }
    if (SIG) band_done(sig, t.pad);        // (the library's device copy carries the band in the record's spare byte)
}

// ---------------------------------------------------------------------------------------------
// Self-guided restoration (sgr_5x5_c / sgr_3x3_c / sgr_mix_c, reference
// src/looprestoration_tmpl.c:389-1363).  The reference streams rows through rotating pointer sets;
// what it computes is: box sums (3x3 and / or 5x5) over a virtual image V whose rows < 0 / >= h come
// from the two lpf rows (one more replica for 5x5) or replicate the stripe's first / last row, the
// (A, B) surfaces of sgr_calc_row_ab on columns -1 .. w (5x5: odd rows only), and the 4/3- resp.
// 6/5-weighted neighbourhoods of sgr_finish_filter_row1 / sgr_finish_filter2.
// Mapping: one wave per 62 output columns of a task; lane = one column of the (A, B) surfaces
// (62 outputs + 1 halo column each side).  A lane walks down the virtual rows keeping the
// horizontal box sums of the last 3 / 5 rows in registers; finished (A, B) rows go to a small
// LDS ring so that the output stage can read the neighbouring columns.
struct AB { int a; int b; };

__device__ __forceinline__ AB calc_ab(const int sumsq, const int sum, const int s, const int bitdepth_min_8, const int n, const int one_by_x,
                                      const uint8_t *x_by_x /* LDS copy of av1_sgr_x_by_x */)
{
    // sgr_calc_row_ab, src/looprestoration_tmpl.c:505-523.  Ranges (12-bit worst case): a <= 25 * 255^2 < 2^21, b <= 25 * 255 < 2^13,
    // x <= 255, sum <= 25 * 4095 < 2^17: those products take the full-rate 24-bit multiplier; p * s and (x * sum) * one_by_x need
    // the 32-bit one (they wrap mod 2^32 exactly as the reference's unsigned arithmetic does).
    const int a = (sumsq + ((1 << (2 * bitdepth_min_8)) >> 1)) >> (2 * bitdepth_min_8);
    const int b = (sum + ((1 << bitdepth_min_8) >> 1)) >> bitdepth_min_8;
    const unsigned p = (unsigned) dv::imax((int) dv::mul_u24((unsigned) a, (unsigned) n) - (int) dv::mul_u24((unsigned) b, (unsigned) b), 0);
    const unsigned z = (p * (unsigned) s + (1u << 19)) >> 20;
    const unsigned x = x_by_x[z < 255 ? z : 255];
    AB r;
    r.a = (int) ((dv::mul_u24(x, (unsigned) sum) * (unsigned) one_by_x + (1 << 11)) >> 12);
    r.b = (int) x;
    return r;
}

// One wave = 64 consecutive "virtual columns" of a ROW of units (tasks of the same plane, y and height, any x): a unit of width w
// takes w + 2 virtual columns (its (A, B) columns -1 .. w), the units of a row follow each other without gaps, and consecutive
// waves overlap by two lanes, so that every virtual column is an interior lane (1 .. 62) of exactly one wave.  Round 1 cut
// every unit into waves of its own: a 64-pixel unit needs 66 (A, B) columns = two waves, the second one with 2 of 64 lanes at
// work; a row of sixty such units now takes 64 waves instead of 120.  The unit a lane works for — its position, width, edge
// flags, filter type and strengths — is per-lane data; what the waves of a row share is the plane, y and the height.
struct SgrWave { uint32_t first, end; int32_t v0; uint32_t pad; };      // tasks [first, end) = the row; lane 0 is virtual column v0 of task `first`

template <typename pixel, bool SIG>
__global__ __launch_bounds__(64) void sgr_kernel(const DevPlanes dst, const DevPlanes src, const DevPlanes lpf,
                                                 const Dav1dHipLrTask *__restrict__ tasks, const SgrWave *__restrict__ waves, const int n_waves,
                                                 const int bitdepth_max, const BandSignal sig)
{
    __shared__ int a3[4][64], b3[4][64], a5[2][64], b5[2][64];
    __shared__ __attribute__((aligned(4))) uint8_t x_by_x[256];
    if ((int) blockIdx.x >= n_waves) return;
    const SgrWave wv = waves[blockIdx.x];
    const int lane = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++) x_by_x[4 * lane + k] = av1_sgr_x_by_x[4 * lane + k];      // first read comes after the loop's first wave_sync
    // which unit of the row this lane works for
    uint32_t ti = wv.first;
    int v = wv.v0 + lane;
    bool valid = true;
    for (int it = 0; it < 64; it++) {
        if (ti >= wv.end) { valid = false; break; }
        const int wt = tasks[ti].w;
        if (v < wt + 2) break;
        v -= wt + 2;
        ti++;
    }
    if (ti >= wv.end) { valid = false; ti = wv.end - 1; }
    const Dav1dHipLrTask t = tasks[ti];
    const Dav1dHipLrTask t0 = tasks[wv.first];                     // the row's plane, y, height (the same for every unit of it)
    const int c = v - 1;                                                // (A, B) column of this lane inside its unit, -1 .. w
    const int bitdepth_min_8 = (32 - __clz(bitdepth_max)) - 8;
    const int pl = __builtin_amdgcn_readfirstlane((int) t0.plane), h = __builtin_amdgcn_readfirstlane((int) t0.h),
              ty = __builtin_amdgcn_readfirstlane((int) t0.y);
    const int w = t.w, edges = t.edges;
    const bool do5 = valid && t.type != DAV1D_HIP_LR_SGR_3X3, do3 = valid && t.type != DAV1D_HIP_LR_SGR_5X5;
    const bool any5 = __any(do5), any3 = __any(do3);
    const int s0 = t.filter[0][0], s1 = t.filter[0][1], w0 = t.filter[0][2], w1 = t.filter[0][3];
    const pixel *const s = reinterpret_cast<const pixel *>(src.data[pl]);
    const pixel *const l = reinterpret_cast<const pixel *>(lpf.data[pl]);
    const int ss = src.stride[pl], ls = lpf.stride[pl];
    // rows below the stripe: the reference only gets to them for long enough (5x5 / mix: even) stripes
    bool use_bottom = (edges & 8) != 0;
    if (t.type != DAV1D_HIP_LR_SGR_3X3) use_bottom = use_bottom && !(h & 1) && h >= ((edges & 4) ? 4 : 6);
    else use_bottom = use_bottom && h >= 3;

    int col[5];                                         // picture columns of c-2 .. c+2 with the edge rules of sgr_box*_row_h
#pragma unroll
    for (int i = 0; i < 5; i++) {
        int cc = c + i - 2;
        if (cc < 0 && !(edges & 1)) cc = 0;
        if (cc >= w && !(edges & 2)) cc = w - 1;
        cc = dv::iclip(cc, -3, w + 2);                  // halo lanes beyond column w are never used
        col[i] = t.x + cc;
    }
    int s3[3] = { 0, 0, 0 }, q3[3] = { 0, 0, 0 }, s5[5] = { 0, 0, 0, 0, 0 }, q5[5] = { 0, 0, 0, 0, 0 };
    const bool out_lane = valid && lane >= 1 && lane <= 62 && c >= 0 && c < w;
    pixel *const d = reinterpret_cast<pixel *>(dst.data[pl]) + ty * dst.stride[pl] + t.x + c;

    const int rs0 = 0, rs1 = h;
    for (int r = rs0 - 3; r < rs1 + 3; r++) {
        // ---- horizontal box sums of virtual row r (which picture row feeds it depends on the lane's unit: its edge flags)
        const pixel *row;
        if (r < 0) {
            if (edges & 4) row = l + (ty - (r == -1 ? 1 : 2)) * ls;
            else row = s + ty * ss;
        } else if (r >= h) {
            if (use_bottom) row = l + dv::imin(ty + h + (r == h ? 0 : 1), lpf.h[pl] - 1) * ls;
            else row = s + (ty + h - 1) * ss;
        } else row = s + (ty + r) * ss;
        const int p0 = row[col[0]], p1 = row[col[1]], p2 = row[col[2]], p3 = row[col[3]], p4 = row[col[4]];
#pragma unroll
        for (int i = 0; i < 2; i++) { s3[i] = s3[i + 1]; q3[i] = q3[i + 1]; }
#pragma unroll
        for (int i = 0; i < 4; i++) { s5[i] = s5[i + 1]; q5[i] = q5[i + 1]; }
        s3[2] = p1 + p2 + p3;
        q3[2] = p1 * p1 + p2 * p2 + p3 * p3;
        s5[4] = s3[2] + p0 + p4;
        q5[4] = q3[2] + p0 * p0 + p4 * p4;
        // ---- (A, B) rows that just became complete
        if (any3 && r >= rs0 && r <= rs1 + 1) {       // row j = r - 1 of the 3x3 surface (rows rs0-1 .. rs1 feed this segment)
            const AB ab = calc_ab(q3[0] + q3[1] + q3[2], s3[0] + s3[1] + s3[2], s1, bitdepth_min_8, 9, 455, x_by_x);
            a3[(r - 1) & 3][lane] = ab.a; b3[(r - 1) & 3][lane] = ab.b;
        }
        if (any5 && r >= rs0 + 1 && ((r - 2) & 1)) {    // row j = r - 2 (odd) of the 5x5 surface
            const AB ab = calc_ab(q5[0] + q5[1] + q5[2] + q5[3] + q5[4], s5[0] + s5[1] + s5[2] + s5[3] + s5[4], s0, bitdepth_min_8, 25, 164, x_by_x);
            a5[((r - 2) >> 1) & 1][lane] = ab.a; b5[((r - 2) >> 1) & 1][lane] = ab.b;
        }
        dv::wave_sync();
        // ---- output row y = r - 3
        const int y = r - 3;
        if (y >= rs0 && y < rs1 && out_lane) {
            const int px = s[(ty + y) * ss + t.x + c];
            int acc = 0;
            if (do3) {
                const int *A0 = a3[(y - 1) & 3], *A1 = a3[y & 3], *A2 = a3[(y + 1) & 3];
                const int *B0 = b3[(y - 1) & 3], *B1 = b3[y & 3], *B2 = b3[(y + 1) & 3];
                const int i = lane;
                const int a = (B1[i] + B1[i - 1] + B1[i + 1] + B0[i] + B2[i]) * 4 + (B0[i - 1] + B2[i - 1] + B0[i + 1] + B2[i + 1]) * 3;
                const int b = (A1[i] + A1[i - 1] + A1[i + 1] + A0[i] + A2[i]) * 4 + (A0[i - 1] + A2[i - 1] + A0[i + 1] + A2[i + 1]) * 3;
                // a <= 32 * 255, px < 2^12, |(b - a * px) >> 9| < 2^17, 8-bit weights: every product fits the 24-bit multiplier
                acc = dv::mad_i24(w1, (b - (int) dv::mul_u24((unsigned) a, (unsigned) px) + (1 << 8)) >> 9, acc);
            }
            if (do5) {
                const int i = lane;
                int t5;
                if (!(y & 1)) {
                    const int k0 = ((y - 1) >> 1) & 1, k1 = ((y + 1) >> 1) & 1;
                    const int a = (b5[k0][i] + b5[k1][i]) * 6 + (b5[k0][i - 1] + b5[k1][i - 1] + b5[k0][i + 1] + b5[k1][i + 1]) * 5;
                    const int b = (a5[k0][i] + a5[k1][i]) * 6 + (a5[k0][i - 1] + a5[k1][i - 1] + a5[k0][i + 1] + a5[k1][i + 1]) * 5;
                    t5 = (b - (int) dv::mul_u24((unsigned) a, (unsigned) px) + (1 << 8)) >> 9;
                } else {
                    const int k = (y >> 1) & 1;
                    const int a = b5[k][i] * 6 + (b5[k][i - 1] + b5[k][i + 1]) * 5;
                    const int b = a5[k][i] * 6 + (a5[k][i - 1] + a5[k][i + 1]) * 5;
                    t5 = (b - (int) dv::mul_u24((unsigned) a, (unsigned) px) + (1 << 7)) >> 8;
                }
                acc = dv::mad_i24(w0, t5, acc);
            }
            const pixel px_out = (pixel) dv::clamp3(px + ((acc + (1 << 10)) >> 11), 0, bitdepth_max);
            if (SIG) dv::st_coherent(d + y * dst.stride[pl], px_out); else d[y * dst.stride[pl]] = px_out;
        }
        dv::wave_sync();
    }
    if (SIG) band_done(sig, (int) wv.pad);
}

} // namespace

// max_w: width of the widest unit of the batch (<= 384): the grid covers that many columns, not 384 for everybody
extern "C" int dav1d_hip_launch_wiener_sig(const DevPlanes *dst, const DevPlanes *src, const DevPlanes *lpf, int bpc,
                                           const Dav1dHipLrTask *tasks, int n, int max_w, const BandSignal *sig, void *stream)
{
    if (n <= 0) return 0;
    const int bitdepth_max = (1 << bpc) - 1;
    const BandSignal sg = sig ? *sig : BandSignal{ nullptr, nullptr, nullptr, 0 };
    const dim3 grid(((max_w < 1 ? 1 : max_w > 384 ? 384 : max_w) + 63) / 64, n, (64 + LR_SEG - 1) / LR_SEG);   // 64 columns x 64 rows per wave
#define WIENER(P, S) hipLaunchKernelGGL((wiener_kernel<P, S>), grid, dim3(64), 0, (hipStream_t) stream, *dst, *src, *lpf, tasks, n, bitdepth_max, sg)
    if (bpc == 8) { if (sg.cnt) WIENER(uint8_t, true); else WIENER(uint8_t, false); }
    else { if (sg.cnt) WIENER(uint16_t, true); else WIENER(uint16_t, false); }
#undef WIENER
    return hip_rc(hipGetLastError());
}
extern "C" int dav1d_hip_launch_wiener(const DevPlanes *dst, const DevPlanes *src, const DevPlanes *lpf, int bpc,
                                       const Dav1dHipLrTask *tasks, int n, int max_w, void *stream)
{
    return dav1d_hip_launch_wiener_sig(dst, src, lpf, bpc, tasks, n, max_w, nullptr, stream);
}

// tasks: DEVICE, the self-guided tasks sorted into rows (dav1d_hip_sgr_make_rows); waves: DEVICE wave descriptors
extern "C" int dav1d_hip_launch_sgr_sig(const DevPlanes *dst, const DevPlanes *src, const DevPlanes *lpf, int bpc,
                                        const Dav1dHipLrTask *tasks, const void *waves, int n_waves, const BandSignal *sig, void *stream)
{
    if (n_waves <= 0) return 0;
    const int bitdepth_max = (1 << bpc) - 1;
    const BandSignal sg = sig ? *sig : BandSignal{ nullptr, nullptr, nullptr, 0 };
#define SGR(P, S) hipLaunchKernelGGL((sgr_kernel<P, S>), dim3(n_waves), dim3(64), 0, (hipStream_t) stream, *dst, *src, *lpf, tasks, \
                                     (const SgrWave *) waves, n_waves, bitdepth_max, sg)
    if (bpc == 8) { if (sg.cnt) SGR(uint8_t, true); else SGR(uint8_t, false); }
    else { if (sg.cnt) SGR(uint16_t, true); else SGR(uint16_t, false); }
#undef SGR
    return hip_rc(hipGetLastError());
}
extern "C" int dav1d_hip_launch_sgr(const DevPlanes *dst, const DevPlanes *src, const DevPlanes *lpf, int bpc,
                                    const Dav1dHipLrTask *tasks, const void *waves, int n_waves, void *stream)
{
    return dav1d_hip_launch_sgr_sig(dst, src, lpf, bpc, tasks, waves, n_waves, nullptr, stream);
}

// Self-guided tasks -> rows (same plane, y, height; sorted by x, in place) and the waves that cover them, appended to `waves` as
// 16-byte descriptors { first task, end task, virtual column of lane 0 inside the first task, 0 }; task indices count from tasks[0].
void dav1d_hip_sgr_make_rows(Dav1dHipLrTask *tasks, size_t n, std::vector<uint32_t> &waves) {
    std::stable_sort(tasks, tasks + n, [](const Dav1dHipLrTask &a, const Dav1dHipLrTask &b) {
        if (a.plane != b.plane) return a.plane < b.plane;
        if (a.y != b.y) return a.y < b.y;
        if (a.h != b.h) return a.h < b.h;
        return a.x < b.x;
    });
    for (size_t r0 = 0; r0 < n;) {
        size_t r1 = r0 + 1;
        while (r1 < n && tasks[r1].plane == tasks[r0].plane && tasks[r1].y == tasks[r0].y && tasks[r1].h == tasks[r0].h) r1++;
        long total = 0;
        for (size_t i = r0; i < r1; i++) total += tasks[i].w + 2;
        size_t ti = r0;
        long start = 0;                 // virtual position where task ti begins
        for (long p0 = 0; p0 + 2 < total; p0 += 62) {
            while (p0 >= start + tasks[ti].w + 2) { start += tasks[ti].w + 2; ti++; }
            waves.push_back((uint32_t) ti); waves.push_back((uint32_t) r1); waves.push_back((uint32_t) (int32_t) (p0 - start)); waves.push_back(0);
        }
        r0 = r1;
    }
}

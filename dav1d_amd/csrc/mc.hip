// Batched motion compensation (put / prep, 8-tap + bilinear, unscaled) for gfx950.
//
// Contract per task = reference put_8tap_c / prep_8tap_c / put_bilin_c / prep_bilin_c
// (src/mc_tmpl.c:129-187, 246-305, 434-489, 516-586) applied to the window that the
// reference driver mc() (src/recon_tmpl.c:938-989) would hand them, including its
// emu_edge step (src/mc_tmpl.c:868-916), which here is a per-pixel coordinate clamp.
//
// Mapping: the host splits every prediction block into tiles of at most 16x16 and bins
// them by tile shape (TW, TH) in {4,8,16}^2.  A tile owns LPT = max(TW*TH/4, 16) lanes,
// 64/LPT tiles share a wave.  Per tile:
//   1. the (TH+7) x (TW+8) source window is gathered (clamped coordinates) into LDS as
//      int16, column 4 of the window = column 0 of the tile, so that every 4-pixel strip
//      starts on an 8-byte LDS boundary;
//   2. horizontal pass: one work item = one row x one 4-pixel strip, three ds_read_b64
//      (12 pixels) -> 4 filtered values -> one ds_write_b64 into the int16 `mid` tile;
//   3. vertical pass: one lane = one row x one 4-pixel strip, eight ds_read_b64 of mid
//      -> 4 outputs -> one 8-byte (4-byte @8bpc) global store.
// Bilinear is the same machinery with taps {0,0,0,16-m,m,0,0,0} and 4 instead of 6
// bits of filter precision.
#include "common.h"
#include "capi.h"
#include "av1_tables.h"

namespace {

struct RefSet { DevPlanes r[8]; };

// 4 consecutive int16 held as two dwords
struct S4 { uint32_t a, b; };
__device__ __forceinline__ int s16lo(uint32_t v) { return (int) (int16_t) (v & 0xffff); }
__device__ __forceinline__ int s16hi(uint32_t v) { return (int) v >> 16; }

__device__ __forceinline__ void load_taps(int set, int m, int *f) {
    // set 0..5 = row of av1_mc_subpel_filters, 6 = bilinear
    if (set == 6) {
#pragma unroll
        for (int i = 0; i < 8; i++) f[i] = 0;
        f[3] = 16 - m;
        f[4] = m;
    } else {
        const int8_t *p = &av1_mc_subpel_filters[(set * 15 + (m - 1)) * 8];
#pragma unroll
        for (int i = 0; i < 8; i++) f[i] = p[i];
    }
}

template <int TW, int TH, typename pixel>
__global__ __launch_bounds__(64) void mc_kernel(const DevPlanes dst, const RefSet refs, const McTile *__restrict__ tiles,
                                                const int n, int16_t *__restrict__ prep, const int bitdepth_max)
{
    constexpr int LPT = (TW * TH / 4) > 16 ? (TW * TH / 4) : 16;
    constexpr int G = 64 / LPT;
    constexpr int WS = TW + 8;          // window row stride (int16)
    constexpr int WR = TH + 7;          // window rows
    constexpr int NS = TW / 4;          // 4-pixel strips per row
    constexpr bool HBD = sizeof(pixel) == 2;

    __shared__ __attribute__((aligned(16))) int16_t win_s[G * WR * WS];
    __shared__ __attribute__((aligned(16))) int16_t mid_s[G * WR * TW];

    const int lane = threadIdx.x;
    const int sub = lane / LPT, l = lane % LPT;
    const int ti = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x) * G + sub;
    const bool live = ti < n;

    McTile t;
    if (live) t = tiles[ti];
    else { t.dst_off = 0; t.src_x = t.src_y = 0; t.w = t.h = 0; t.mx = t.my = 0; t.fh = t.fv = 0; t.kind = 0; t.plane = 0; t.ref = 0; t.bw = 0; t.ox = t.oy = 0; }

    int16_t *const win = win_s + sub * WR * WS;
    int16_t *const mid = mid_s + sub * WR * TW;

    int ib;   // intermediate_bits
    if (HBD) ib = 14 - (32 - __clz(bitdepth_max)); else ib = 4;
    const int bias = HBD ? 8192 : 0;
    const bool bilin = t.fh == 6;
    const int fbits = bilin ? 4 : 6;

    // ---- 1. gather the window
    if (live) {
        const DevPlanes &rp = refs.r[t.ref];
        const pixel *src = reinterpret_cast<const pixel *>(rp.data[t.plane]);
        const int rs = rp.stride[t.plane], rw = rp.w[t.plane], rh = rp.h[t.plane];
        // only the rows / columns this tile can reference
        const int rows = t.h + 7, cols = t.w + 8;
        for (int i = l; i < rows * cols; i += LPT) {
            const int ry = i / cols, cx = i - ry * cols;
            const int sy = dv::iclip(t.src_y - 3 + ry, 0, rh - 1);
            const int sx = dv::iclip(t.src_x - 4 + cx, 0, rw - 1);
            win[ry * WS + cx] = (int16_t) src[sy * rs + sx];
        }
    }
    __syncthreads();

    const bool has_h = t.mx != 0, has_v = t.my != 0;
    const int ns = (t.w + 3) >> 2;      // strips actually present (w = 2 -> 1 partial strip)

    // ---- 2. horizontal pass into mid (rows 0..h+6 when a vertical pass follows, else the h output rows)
    if (live) {
        int fh[8];
        if (has_h) load_taps(t.fh, t.mx, fh);
        const int r0 = has_v ? 0 : 3, nr = has_v ? t.h + 7 : t.h;
        // h-only rounding (reference src/mc_tmpl.c:135-136,165-171 / 452-461)
        const int sh1 = fbits - ib;
        const int rnd1 = (1 << sh1) >> 1;
        for (int i = l; i < nr * ns; i += LPT) {
            const int r = r0 + i / ns, s = i % ns;
            const uint32_t *wp = reinterpret_cast<const uint32_t *>(win + r * WS + 4 * s);
            uint32_t d[6];
#pragma unroll
            for (int k = 0; k < 6; k++) d[k] = wp[k];
            int o[4];
            if (has_h) {
                int p[12];
#pragma unroll
                for (int k = 0; k < 6; k++) { p[2 * k] = s16lo(d[k]); p[2 * k + 1] = s16hi(d[k]); }
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    int acc = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) acc += fh[k] * p[x + 1 + k];
                    o[x] = acc;
                }
                if (has_v) {
#pragma unroll
                    for (int x = 0; x < 4; x++) o[x] = (o[x] + rnd1) >> sh1;
                }
            } else {
                // no horizontal filter: mid carries the plain pixels (columns 4..7 of the strip read)
                o[0] = s16lo(d[2]); o[1] = s16hi(d[2]); o[2] = s16lo(d[3]); o[3] = s16hi(d[3]);
            }
            uint32_t *mp = reinterpret_cast<uint32_t *>(mid + r * TW + 4 * s);
            if (has_v || !has_h) {
                mp[0] = dv::pack2(o[0], o[1]);
                mp[1] = dv::pack2(o[2], o[3]);
            } else {
                // horizontal-only: finish here, keep full precision in registers via a second
                // packed buffer is not needed -- write the final values (they fit int16)
                int q[4];
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    if (t.kind == DAV1D_HIP_MC_PUT) {
                        if (bilin) {
                            const int px = (o[x] + rnd1) >> sh1;
                            q[x] = dv::iclip((px + ((1 << ib) >> 1)) >> ib, 0, bitdepth_max);
                        } else {
                            q[x] = dv::iclip((o[x] + 32 + rnd1) >> 6, 0, bitdepth_max);
                        }
                    } else {
                        q[x] = ((o[x] + rnd1) >> sh1) - bias;
                    }
                }
                mp[0] = dv::pack2(q[0], q[1]);
                mp[1] = dv::pack2(q[2], q[3]);
            }
        }
    }
    __syncthreads();

    // ---- 3. vertical pass / output: lane = (row, strip)
    if (live) {
        int fv[8];
        if (has_v) load_taps(t.fv, t.my, fv);
        for (int i = l; i < t.h * ns; i += LPT) {
            const int r = i / ns, s = i % ns;
            int q[4];
            if (has_v) {
                int acc[4] = { 0, 0, 0, 0 };
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t *mp = reinterpret_cast<const uint32_t *>(mid + (r + k) * TW + 4 * s);
                    const uint32_t a = mp[0], b = mp[1];
                    acc[0] += fv[k] * s16lo(a); acc[1] += fv[k] * s16hi(a);
                    acc[2] += fv[k] * s16lo(b); acc[3] += fv[k] * s16hi(b);
                }
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    if (t.kind == DAV1D_HIP_MC_PUT) {
                        const int sh = has_h ? fbits + ib : fbits;
                        q[x] = dv::iclip((acc[x] + ((1 << sh) >> 1)) >> sh, 0, bitdepth_max);
                    } else {
                        const int sh = has_h ? fbits : fbits - ib;
                        q[x] = ((acc[x] + ((1 << sh) >> 1)) >> sh) - bias;
                    }
                }
            } else {
                const uint32_t *mp = reinterpret_cast<const uint32_t *>(mid + (r + 3) * TW + 4 * s);
                const uint32_t a = mp[0], b = mp[1];
                q[0] = s16lo(a); q[1] = s16hi(a); q[2] = s16lo(b); q[3] = s16hi(b);
                if (!has_h && t.kind == DAV1D_HIP_MC_PREP) {
#pragma unroll
                    for (int x = 0; x < 4; x++) q[x] = (q[x] << ib) - bias;
                }
            }
            const int nvalid = dv::imin(4, t.w - 4 * s);
            if (t.kind == DAV1D_HIP_MC_PUT) {
                pixel *d = reinterpret_cast<pixel *>(dst.data[t.plane]) + t.dst_off + (t.oy + r) * dst.stride[t.plane] + t.ox + 4 * s;
                if (nvalid == 4) {
                    if (HBD) {
                        uint2 v; v.x = dv::pack2(q[0], q[1]); v.y = dv::pack2(q[2], q[3]);
                        *reinterpret_cast<uint2 *>(d) = v;
                    } else {
                        *reinterpret_cast<uint32_t *>(d) = (uint32_t) q[0] | ((uint32_t) q[1] << 8) |
                                                           ((uint32_t) q[2] << 16) | ((uint32_t) q[3] << 24);
                    }
                } else {
                    for (int x = 0; x < nvalid; x++) d[x] = (pixel) q[x];
                }
            } else {
                int16_t *d = prep + t.dst_off + (t.oy + r) * t.bw + t.ox + 4 * s;
                if (nvalid == 4) {
                    uint2 v; v.x = dv::pack2(q[0], q[1]); v.y = dv::pack2(q[2], q[3]);
                    *reinterpret_cast<uint2 *>(d) = v;
                } else {
                    for (int x = 0; x < nvalid; x++) d[x] = (int16_t) q[x];
                }
            }
        }
    }
}

template <typename pixel>
hipError_t launch_cls(const int cls, const DevPlanes &dst, const RefSet &refs, const McTile *tiles, const int n,
                      int16_t *prep, const int bitdepth_max, hipStream_t stream)
{
#define CASE(C, TW, TH) case C: { \
        constexpr int lpt = (TW * TH / 4) > 16 ? (TW * TH / 4) : 16; \
        constexpr int g = 64 / lpt; \
        hipLaunchKernelGGL((mc_kernel<TW, TH, pixel>), dim3((n + g - 1) / g), dim3(64), 0, stream, \
                           dst, refs, tiles, n, prep, bitdepth_max); \
        break; }
    switch (cls) {
        CASE(0, 4, 4) CASE(1, 4, 8) CASE(2, 4, 16)
        CASE(3, 8, 4) CASE(4, 8, 8) CASE(5, 8, 16)
        CASE(6, 16, 4) CASE(7, 16, 8) CASE(8, 16, 16)
        default: return hipErrorInvalidValue;
    }
#undef CASE
    return hipGetLastError();
}

} // namespace

extern "C" int dav1d_hip_launch_mc_bin(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc, int cls,
                                       const McTile *tiles, int n, int16_t *prep, void *stream)
{
    if (n <= 0) return 0;
    RefSet rs;
    for (int i = 0; i < 8; i++) rs.r[i] = refs[i < n_refs ? i : 0];
    const int bitdepth_max = (1 << bpc) - 1;
    hipError_t e;
    if (bpc == 8) e = launch_cls<uint8_t>(cls, *dst, rs, tiles, n, prep, bitdepth_max, (hipStream_t) stream);
    else          e = launch_cls<uint16_t>(cls, *dst, rs, tiles, n, prep, bitdepth_max, (hipStream_t) stream);
    return hip_rc(e);
}

// Batched motion compensation (put / prep / fused compound avg, 8-tap + bilinear,
// unscaled) for gfx950.
//
// Contract per task = reference put_8tap_c / prep_8tap_c / put_bilin_c / prep_bilin_c
// (src/mc_tmpl.c:129-187, 246-305, 434-489, 516-586) applied to the window that the
// reference driver mc() (src/recon_tmpl.c:938-989) would hand them, including its
// emu_edge step (src/mc_tmpl.c:868-916), which here is a per-pixel coordinate clamp;
// fused tiles additionally apply avg_c / w_avg_c (src/mc_tmpl.c:628-660) to the two
// prep results while they are still in registers.
//
// Mapping: the host cuts every prediction block into tiles of at most 16x16 and bins
// them by tile shape (TW, TH) in {4,8,16}^2.  A tile owns LPT = TW*TH/4 lanes (one
// lane per 4-pixel output strip), 64/LPT tiles share a wave (16x16: 1, 8x8: 4, 4x4: 16).
// Per tile and reference:
//   1. gather: the (TH+7) x (TW+8) window goes to LDS as int16, tile column 0 at window
//      column 4 so every 4-pixel strip is 8-byte aligned.  Interior windows are fetched
//      with 8-byte (4-pixel) loads; windows touching the picture edge fall back to
//      per-pixel clamped loads (== emu_edge).
//   2. horizontal pass: one work item = TWO rows x one 4-pixel strip: 2 x 3 ds_read_b64,
//      v_dot2 on packed pixel pairs (even outputs use the taps packed (f0,f1)(f2,f3)..,
//      odd outputs the taps packed (0,f0)(f1,f2)..(f7,0): no re-alignment of the data),
//      one ds_write_b128 of the ROW-PAIR-INTERLEAVED intermediate mid2[row/2][col] =
//      (row even, row odd).
//   3. vertical pass: one lane = one output row x one 4-pixel strip: 5 ds_read_b128 of
//      mid2, v_dot2 against the taps packed for the row's parity, round/clip, 8-byte store.
// "No filter" in a direction is the unit tap with zero shift, bilinear is the taps
// (16-m, m) with 4 instead of 6 bits of precision, so all variants share one code path.
#include "common.h"
#include "capi.h"
#include "av1_tables.h"

namespace {

struct RefSet { DevPlanes r[8]; };

struct __attribute__((packed, aligned(2))) U64u { uint32_t a, b; };   // 2-byte aligned 8-byte global load
struct __attribute__((packed, aligned(1))) U32u { uint32_t a; };

// taps of one direction packed for v_dot2: ev[k] = (f[2k], f[2k+1]), od[k] = (f[2k-1], f[2k]) with f[-1] = f[8] = 0
struct Taps { uint32_t ev[4]; uint32_t od[5]; };

__device__ __forceinline__ Taps load_taps(const int set, const int m) {
    // av1_mc_taps_packed[set 0..5 | 6 = bilinear][phase 0..15, 0 = unit tap][ev0..3, od0..4]
    const uint32_t *p = &av1_mc_taps_packed[(set * 16 + m) * 9];
    Taps t;
#pragma unroll
    for (int k = 0; k < 4; k++) t.ev[k] = p[k];
#pragma unroll
    for (int k = 0; k < 5; k++) t.od[k] = p[4 + k];
    return t;
}

template <int TW, int TH, typename pixel>
__global__ __launch_bounds__(64) void mc_kernel(const DevPlanes dst, const RefSet refs, const McTile *__restrict__ tiles,
                                                const int n, int16_t *__restrict__ prep, const int bitdepth_max)
{
    constexpr int LPT = TW * TH / 4;    // lanes per tile = output strips per tile
    constexpr int G = 64 / LPT;         // tiles per wave
    constexpr int WS = TW + 8;          // window row stride (int16)
    constexpr int WR = TH + 8;          // window rows held (TH+7 used, +1 so row pairs are complete)
    constexpr int NS = TW / 4;          // 4-pixel strips per row
    constexpr int NCH = WS / 4;         // 4-pixel chunks per window row
    constexpr int NPR = WR / 2;         // row pairs of the intermediate
    constexpr bool HBD = sizeof(pixel) == 2;

    __shared__ __attribute__((aligned(16))) int16_t win_s[G * WR * WS];
    __shared__ __attribute__((aligned(16))) uint32_t mid_s[G * NPR * TW];

    const int lane = threadIdx.x;
    // G == 1: the whole wave works on one tile, so the record, the taps and all the control flow
    // derived from them are wave-uniform (scalar loads, SGPRs, s_cbranch instead of exec masking)
    const int sub = G == 1 ? 0 : lane / LPT, l = G == 1 ? lane : lane % LPT;
    const int ti = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x) * G + sub;
    const bool live = ti < n;

    McTile t;
    if (G == 1) {
        // one tile per wave: pull the record through readfirstlane so the compiler keeps it (and the
        // taps, strides, branch conditions derived from it) in SGPRs
        static_assert(sizeof(McTile) == 44, "McTile layout");
        const uint32_t *tp = reinterpret_cast<const uint32_t *>(tiles + (live ? ti : 0));
        uint32_t raw[11];
#pragma unroll
        for (int k = 0; k < 11; k++) raw[k] = (uint32_t) __builtin_amdgcn_readfirstlane((int) tp[k]);
        __builtin_memcpy(&t, raw, sizeof(t));
    } else if (live) t = tiles[ti];
    else {
        t.dst_off = 0; t.w = t.h = 0; t.kind = 0; t.plane = 0; t.bw = 0; t.ox = t.oy = 0; t.weight = 0;
#pragma unroll
        for (int k = 0; k < 2; k++) { t.r[k].src_x = t.r[k].src_y = 0; t.r[k].mx = t.r[k].my = 0; t.r[k].fh = t.r[k].fv = 0; t.r[k].ref = 0; }
    }

    int16_t *const win = win_s + sub * WR * WS;
    uint32_t *const mid = mid_s + sub * NPR * TW;

    const int ib = HBD ? 14 - (32 - __clz(bitdepth_max)) : 4;   // intermediate_bits
    const int bias = HBD ? 8192 : 0;                            // PREP_BIAS
    const bool compound = t.kind >= MCT_AVG;
    const bool as_prep = t.kind != MCT_PUT;                     // PREP and both inputs of a compound tile

    // output strip of this lane in the vertical pass
    const int vr = l / NS, vs = l % NS;
    int acc0[4] = { 0, 0, 0, 0 };   // first prediction of a compound tile
    int q[4] = { 0, 0, 0, 0 };

    // one prediction (gather -> h -> v) of this lane's strip into out[]; a lambda invoked once or
    // twice rather than a loop over t.r[] so that the record is never indexed dynamically
    auto predict = [&](const McRef rf, int (&out)[4]) {
        const bool has_h = rf.mx != 0, has_v = rf.my != 0;
        const int fbits = rf.fh == 6 ? 4 : 6;

        // ---- 1. gather the window
        if (live) {
            const DevPlanes &rp = refs.r[rf.ref];
            const pixel *src = reinterpret_cast<const pixel *>(rp.data[t.plane]);
            const int rs = rp.stride[t.plane], rw = rp.w[t.plane], rh = rp.h[t.plane];
            const int x0 = rf.src_x - 4, y0 = rf.src_y - 3;
            const bool interior = x0 >= 0 && y0 >= 0 && x0 + WS <= rw && y0 + WR - 1 <= rh;
            if (interior) {
                const pixel *base = src + y0 * rs + x0;
#pragma unroll
                for (int i = l; i < (WR - 1) * NCH; i += LPT) {
                    const int ry = i / NCH, ch = i % NCH;
                    uint32_t lo, hi;
                    if (HBD) {
                        const U64u v = *reinterpret_cast<const U64u *>(base + ry * rs + 4 * ch);
                        lo = v.a; hi = v.b;
                    } else {
                        const uint32_t v = reinterpret_cast<const U32u *>(base + ry * rs + 4 * ch)->a;
                        lo = (v & 0xff) | ((v & 0xff00) << 8);
                        hi = ((v >> 16) & 0xff) | ((v >> 8) & 0xff0000);
                    }
                    *reinterpret_cast<uint2 *>(win + ry * WS + 4 * ch) = make_uint2(lo, hi);
                }
            } else {
                // edge emulation: per-pixel clamped fetch, 8 independent loads in flight per lane
                for (int i0 = l; i0 < (WR - 1) * WS; i0 += 8 * LPT) {
                    pixel v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int i = dv::imin(i0 + u * LPT, (WR - 1) * WS - 1);
                        const int ry = i / WS, cx = i % WS;
                        const int sy = dv::iclip(y0 + ry, 0, rh - 1);
                        const int sx = dv::iclip(x0 + cx, 0, rw - 1);
                        v[u] = src[sy * rs + sx];
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int i = i0 + u * LPT;
                        if (i < (WR - 1) * WS) win[i] = (int16_t) v[u];
                    }
                }
            }
        }
        __syncthreads();

        // ---- 2. horizontal pass: item = (row pair, strip) -> mid2[pair][4 cols] = (even row, odd row)
        if (live) {
            const Taps fh = load_taps(rf.fh, rf.mx);
            const int sh1 = has_h ? fbits - ib : 0;
            const int rnd1 = (1 << sh1) >> 1;
            for (int it = l; it < NPR * NS; it += LPT) {
                const int pr = it / NS, s = it % NS;
                int o[2][4];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const uint2 *wp = reinterpret_cast<const uint2 *>(win + (2 * pr + e) * WS + 4 * s);
                    const uint2 a = wp[0], b = wp[1], c = wp[2];
                    const uint32_t d[6] = { a.x, a.y, b.x, b.y, c.x, c.y };
                    // out x sums f[k] * p[x + 1 + k], p[] = the 12 pixels of d[]
                    int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
                    for (int k = 0; k < 5; k++) { s0 = dv::dot2(d[k], fh.od[k], s0); s2 = dv::dot2(d[k + 1], fh.od[k], s2); }
#pragma unroll
                    for (int k = 0; k < 4; k++) { s1 = dv::dot2(d[k + 1], fh.ev[k], s1); s3 = dv::dot2(d[k + 2], fh.ev[k], s3); }
                    o[e][0] = s0; o[e][1] = s1; o[e][2] = s2; o[e][3] = s3;
                }
                if (has_v) {
                    // intermediate rounding, reference src/mc_tmpl.c:150-152 (8-tap) / 462-464 (bilinear)
#pragma unroll
                    for (int e = 0; e < 2; e++)
#pragma unroll
                        for (int x = 0; x < 4; x++) o[e][x] = (o[e][x] + rnd1) >> sh1;
                } else {
                    // no vertical filter: finish the sample here (the vertical pass is then the unit tap)
#pragma unroll
                    for (int e = 0; e < 2; e++)
#pragma unroll
                        for (int x = 0; x < 4; x++) {
                            int v = o[e][x];
                            if (!as_prep) {
                                if (has_h) {
                                    if (fbits == 4) v = (((v + rnd1) >> sh1) + ((1 << ib) >> 1)) >> ib;   // src/mc_tmpl.c:467-476
                                    else            v = (v + 32 + rnd1) >> 6;                              // src/mc_tmpl.c:165-171
                                }
                            } else {
                                v = has_h ? ((v + rnd1) >> sh1) - bias : (v << ib) - bias;                 // :283-291 / :61-72
                            }
                            o[e][x] = v;
                        }
                }
                uint4 m;
                m.x = dv::pack2(o[0][0], o[1][0]);
                m.y = dv::pack2(o[0][1], o[1][1]);
                m.z = dv::pack2(o[0][2], o[1][2]);
                m.w = dv::pack2(o[0][3], o[1][3]);
                *reinterpret_cast<uint4 *>(mid + pr * TW + 4 * s) = m;
            }
        }
        __syncthreads();

        // ---- 3. vertical pass: lane = (output row vr, strip vs)
        if (live) {
            const Taps fv = load_taps(rf.fv, rf.my);
            // rows vr .. vr+7 of the window = pairs j0 .. j0+4; odd vr starts in the middle of a pair
            const int j0 = vr >> 1;
            const bool odd = vr & 1;
            uint32_t g[5];
#pragma unroll
            for (int k = 0; k < 5; k++) g[k] = odd ? fv.od[k] : (k < 4 ? fv.ev[k] : 0u);
            int sum[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int j = dv::imin(j0 + k, NPR - 1);     // the 5th pair of an even row is weight 0
                const uint4 m = *reinterpret_cast<const uint4 *>(mid + j * TW + 4 * vs);
                sum[0] = dv::dot2(m.x, g[k], sum[0]);
                sum[1] = dv::dot2(m.y, g[k], sum[1]);
                sum[2] = dv::dot2(m.z, g[k], sum[2]);
                sum[3] = dv::dot2(m.w, g[k], sum[3]);
            }
            int sh2, vb;
            if (!has_v) { sh2 = 0; vb = 0; }
            else if (!as_prep) { sh2 = has_h ? fbits + ib : fbits; vb = 0; }          // src/mc_tmpl.c:157-159,176-178
            else { sh2 = has_h ? fbits : fbits - ib; vb = bias; }                     // :272-277, :294-299
            const int rnd2 = (1 << sh2) >> 1;
#pragma unroll
            for (int x = 0; x < 4; x++) out[x] = ((sum[x] + rnd2) >> sh2) - vb;
        }
    };

    predict(t.r[0], q);
    if (compound) {
#pragma unroll
        for (int x = 0; x < 4; x++) acc0[x] = q[x];
        __syncthreads();                        // the second gather overwrites win / mid
        predict(t.r[1], q);
    }

    // ---- combine + store
    if (live && vr < t.h) {
        if (t.kind == MCT_AVG) {
#pragma unroll
            for (int x = 0; x < 4; x++) q[x] = (acc0[x] + q[x] + (1 << ib) + bias * 2) >> (ib + 1);          // avg_c
        } else if (t.kind == MCT_WAVG) {
#pragma unroll
            for (int x = 0; x < 4; x++)
                q[x] = (acc0[x] * t.weight + q[x] * (16 - t.weight) + (8 << ib) + bias * 16) >> (ib + 4);  // w_avg_c
        }
        const int nvalid = dv::imin(4, t.w - 4 * vs);
        if (t.kind != MCT_PREP) {
#pragma unroll
            for (int x = 0; x < 4; x++) q[x] = dv::iclip(q[x], 0, bitdepth_max);
            pixel *d = reinterpret_cast<pixel *>(dst.data[t.plane]) + t.dst_off + (t.oy + vr) * dst.stride[t.plane] + t.ox + 4 * vs;
            if (nvalid == 4) {
                if (HBD) *reinterpret_cast<uint2 *>(d) = make_uint2(dv::pack2(q[0], q[1]), dv::pack2(q[2], q[3]));
                else *reinterpret_cast<uint32_t *>(d) = (uint32_t) q[0] | ((uint32_t) q[1] << 8) | ((uint32_t) q[2] << 16) | ((uint32_t) q[3] << 24);
            } else if (nvalid > 0) {
                for (int x = 0; x < nvalid; x++) d[x] = (pixel) q[x];
            }
        } else {
            int16_t *d = prep + t.dst_off + (t.oy + vr) * t.bw + t.ox + 4 * vs;
            if (nvalid == 4) *reinterpret_cast<uint2 *>(d) = make_uint2(dv::pack2(q[0], q[1]), dv::pack2(q[2], q[3]));
            else if (nvalid > 0) for (int x = 0; x < nvalid; x++) d[x] = (int16_t) q[x];
        }
    }
}

template <typename pixel>
hipError_t launch_cls(const int cls, const DevPlanes &dst, const RefSet &refs, const McTile *tiles, const int n,
                      int16_t *prep, const int bitdepth_max, hipStream_t stream)
{
#define CASE(C, TW, TH) case C: { \
        constexpr int g = 64 / (TW * TH / 4); \
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

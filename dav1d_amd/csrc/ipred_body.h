// The intra prediction of one transform block by one wave (edge preparation + predictor), shared by ipred.hip and by the
// fused prediction + residual kernel of the intra wavefront (intra_pair.hip).  See ipred.hip for the description.
#pragma once
#include "common.h"
#include "capi.h"
#include "av1_tables.h"

namespace {

enum { M_DC = 0, M_VERT, M_HOR, M_LEFT_DC, M_TOP_DC, M_DC_128, M_Z1, M_Z2, M_Z3, M_SMOOTH, M_SMOOTH_V, M_SMOOTH_H, M_PAETH, M_FILTER };
enum { EC = 160, ESZ = 336 };       // centre / size of the LDS edge arrays (indices -144 .. +175 used at most)

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// get_filter_strength / get_upsample, src/ipred_tmpl.c:327-359, 386-388
__device__ __forceinline__ int filter_strength(const int wh, const int angle, const int is_sm) {
    if (is_sm) {
        if (wh <= 8) { if (angle >= 64) return 2; if (angle >= 40) return 1; }
        else if (wh <= 16) { if (angle >= 48) return 2; if (angle >= 20) return 1; }
        else if (wh <= 24) { if (angle >= 4) return 3; }
        else return 3;
    } else {
        if (wh <= 8) { if (angle >= 56) return 1; }
        else if (wh <= 16) { if (angle >= 40) return 1; }
        else if (wh <= 24) { if (angle >= 32) return 3; if (angle >= 16) return 2; if (angle >= 8) return 1; }
        else if (wh <= 32) { if (angle >= 32) return 3; if (angle >= 4) return 2; return 1; }
        else return 3;
    }
    return 0;
}
__device__ __forceinline__ int get_upsample(const int wh, const int angle, const int is_sm) { return angle < 40 && wh <= (16 >> is_sm); }

// filter_edge, src/ipred_tmpl.c:361-384: element i of the output
__device__ __forceinline__ int filter_edge_at(const int16_t *in, const int i, const int lim_from, const int lim_to,
                                              const int from, const int to, const int strength) {
    if (i < lim_from || i >= lim_to) return in[dv::iclip(i, from, to - 1)];
    const int k0 = strength == 3 ? 2 : 0, k1 = strength == 1 ? 4 : strength == 2 ? 5 : 4, k2 = strength == 1 ? 8 : strength == 2 ? 6 : 4;
    const int s = in[dv::iclip(i - 2, from, to - 1)] * k0 + in[dv::iclip(i - 1, from, to - 1)] * k1 + in[dv::iclip(i, from, to - 1)] * k2 +
                  in[dv::iclip(i + 1, from, to - 1)] * k1 + in[dv::iclip(i + 2, from, to - 1)] * k0;
    return (s + 8) >> 4;
}
// upsample_edge, src/ipred_tmpl.c:390-405: element j of the output (2*hsz - 1 elements)
__device__ __forceinline__ int upsample_edge_at(const int16_t *in, const int j, const int from, const int to, const int bitdepth_max) {
    const int i = j >> 1;
    if (!(j & 1)) return in[dv::iclip(i, from, to - 1)];
    const int s = -in[dv::iclip(i - 1, from, to - 1)] + 9 * in[dv::iclip(i, from, to - 1)] + 9 * in[dv::iclip(i + 1, from, to - 1)] -
                  in[dv::iclip(i + 2, from, to - 1)];
    return dv::iclip((s + 8) >> 4, 0, bitdepth_max);
}

constexpr int IPRED_PARTS = 4;

// t: the block's task; split / part: this workgroup predicts part `part` of IPRED_PARTS of the block's pixels (split) or all
// of them; e1, e2 (ESZ int16 each), blk (32 x 32 int16): LDS of the wave; o / ostride: where the predicted pixels go — the
// block's place in the picture, or an LDS tile (row stride = block width) when a residual is added by the same wave.
// COH 1: the pixels read from the picture (edges, CfL's luma) were written by other workgroups of the SAME launch (intra_flow.hip);
// COH 2: `dst` describes images kept in LDS (intra_sb.hip)
template <typename pixel, int COH = 0>
__device__ __forceinline__ void ipred_body(const DevPlanes &dst, const Dav1dHipIpredTask &t, const int part, const bool split,
                                           uint8_t *aux, const int layout, const int bitdepth_max,
                                           int16_t *e1, int16_t *e2, int16_t *blk, pixel *const o, const int ostride)
{
    const uint8_t *const pal_idx = aux;
    const int lane = threadIdx.x & 63;        // the body belongs to one wave (the intra kernels run four side by side in a workgroup)
    constexpr bool HBD = sizeof(pixel) == 2;
    const int bitdepth = 32 - __clz(bitdepth_max);
    const int stride = dst.stride[t.plane];
    const dv::PxRead<pixel, COH> d = { reinterpret_cast<const pixel *>(dst.data[t.plane]) + t.dst_off };
    const int w = t.tw * 4, h = t.th * 4;
    const int lw = __builtin_ctz(w);     // block sides are powers of two: pixel i of the block is (i >> lw, i & (w - 1))
    int16_t *const E = e1 + EC;          // E[k] == topleft_out[k]
    const int i_lo = split ? part * (w * h / IPRED_PARTS) : 0, i_hi = split ? i_lo + w * h / IPRED_PARTS : w * h;

    // ---------------------------------------------------------------- palette
    if (t.kind == DAV1D_HIP_IPRED_PAL) {
        // t.pal[] = 8 colours, indices packed two per byte (src/ipred_tmpl.c:717-730)
        const uint8_t *idx = pal_idx + t.aux_off;
        for (int i = (i_lo >> 1) + lane; i < (i_hi >> 1); i += 64) {
            const int y = i >> (lw - 1), x = (i & ((w >> 1) - 1)) * 2;
            const int v = idx[i];
            // the palette sits in scalar registers: picked by compares, never indexed (indexing would spill it to scratch)
            int c0 = t.pal[0], c1 = t.pal[0];
#pragma unroll
            for (int k = 1; k < 8; k++) { c0 = (v & 7) == k ? (int) t.pal[k] : c0; c1 = (v >> 4) == k ? (int) t.pal[k] : c1; }
            o[y * ostride + x] = (pixel) c0;
            o[y * ostride + x + 1] = (pixel) c1;
        }
        return;
    }

    // ---------------------------------------------------------------- intra block copy: put_bilin_c (src/mc_tmpl.c:434-489) on the frame itself
    if (t.kind == DAV1D_HIP_IPRED_COPY) {
        const int ss_hor = t.plane && layout != DAV1D_HIP_LAYOUT_I444, ss_ver = t.plane && layout == DAV1D_HIP_LAYOUT_I420;
        // mc() bounds the source by the coded area in whole 8x8 luma blocks (f->bw * 4 >> ss_hor, src/recon_tmpl.c:960-978)
        const int iw = ((dst.w[0] + 7) & ~7) >> ss_hor, ih = ((dst.h[0] + 7) & ~7) >> ss_ver;
        const int sx = (int) (int16_t) t.pal[0], sy = (int) (int16_t) t.pal[1], mx = t.pal[2] & 15, my = (t.pal[2] >> 8) & 15;
        const dv::PxRead<pixel, COH> P = { reinterpret_cast<const pixel *>(dst.data[t.plane]) };
        const int ib = HBD ? 14 - bitdepth : 4;                 // get_intermediate_bits
        auto px = [&](const int xx, const int yy) -> int { return (int) P[dv::iclip(yy, 0, ih - 1) * stride + dv::iclip(xx, 0, iw - 1)]; };
        for (int i = i_lo + lane; i < i_hi; i += 64) {
            const int y = i >> lw, x = i & (w - 1);
            const int X = sx + x, Y = sy + y;
            int v;
            if (mx) {
                const int a0 = px(X, Y), a1 = px(X + 1, Y);
                const int m0 = (16 * a0 + mx * (a1 - a0) + ((1 << (4 - ib)) >> 1)) >> (4 - ib);
                if (my) {
                    const int b0 = px(X, Y + 1), b1 = px(X + 1, Y + 1);
                    const int m1 = (16 * b0 + mx * (b1 - b0) + ((1 << (4 - ib)) >> 1)) >> (4 - ib);
                    v = (16 * m0 + my * (m1 - m0) + ((1 << (4 + ib)) >> 1)) >> (4 + ib);
                } else {
                    v = (m0 + ((1 << ib) >> 1)) >> ib;
                }
            } else if (my) {
                const int a0 = px(X, Y), b0 = px(X, Y + 1);
                v = (16 * a0 + my * (b0 - a0) + 8) >> 4;
            } else {
                v = px(X, Y);
            }
            o[y * ostride + x] = (pixel) dv::iclip(v, 0, bitdepth_max);
        }
        return;
    }

    // ---------------------------------------------------------------- mode mapping (prepare_intra_edges, :89-116)
    const bool have_left = t.flags & 1, have_top = t.flags & 2;
    bool edge_filter = t.flags & 16;
    int is_sm = (t.flags >> 5) & 1;
    int mode, angle = 0;
    // DSP-level kinds: the caller did prepare_intra_edges, `mode` is the table index and the edge array sits in `aux`
    const bool dsp_edge = t.kind == DAV1D_HIP_IPRED_DSP || t.kind == DAV1D_HIP_IPRED_DSP_CFL_PRED;
    const bool is_cfl = t.kind == DAV1D_HIP_IPRED_CFL || t.kind == DAV1D_HIP_IPRED_DSP_CFL_AC || t.kind == DAV1D_HIP_IPRED_DSP_CFL_PRED;
    if (dsp_edge) {
        mode = t.mode;
        const int raw = t.pal[0];                     // the `angle` argument with its flag bits (src/ipred_prepare.h:92-93)
        angle = raw & 511; is_sm = (raw >> 9) & 1; edge_filter = (raw >> 10) & 1;
    } else if (t.kind == DAV1D_HIP_IPRED_DSP_CFL_AC) {
        mode = M_DC_128;
    } else if (t.kind == DAV1D_HIP_IPRED_CFL) {
        mode = have_left ? (have_top ? M_DC : M_LEFT_DC) : (have_top ? M_TOP_DC : M_DC_128);
    } else if (t.mode >= 1 && t.mode <= 8) {
        const int base = t.mode == 1 ? 90 : t.mode == 2 ? 180 : t.mode == 3 ? 45 : t.mode == 4 ? 135 : t.mode == 5 ? 113 :
                         t.mode == 6 ? 157 : t.mode == 7 ? 203 : 67;
        angle = base + 3 * t.angle;
        if (angle <= 90) mode = (angle < 90 && have_top) ? M_Z1 : M_VERT;
        else if (angle < 180) mode = M_Z2;
        else mode = (angle > 180 && have_left) ? M_Z3 : M_HOR;
    } else if (t.mode == 0) {
        mode = have_left ? (have_top ? M_DC : M_LEFT_DC) : (have_top ? M_TOP_DC : M_DC_128);
    } else if (t.mode == 12) {
        mode = have_left ? (have_top ? M_PAETH : M_HOR) : (have_top ? M_VERT : M_DC_128);
    } else if (t.mode == 13) {
        mode = M_FILTER;
        angle = t.angle;
    } else {
        mode = t.mode;      // SMOOTH, SMOOTH_V, SMOOTH_H keep their index (9..11)
    }
    // needs_{left, top, topleft, topright, bottomleft}, src/ipred_prepare_tmpl.c:50-73
    const bool n_left = mode == M_DC || mode == M_HOR || mode == M_LEFT_DC || mode == M_Z2 || mode == M_Z3 || (mode >= M_SMOOTH && mode <= M_FILTER);
    const bool n_top = mode == M_DC || mode == M_VERT || mode == M_TOP_DC || mode == M_Z1 || mode == M_Z2 || (mode >= M_SMOOTH && mode <= M_FILTER);
    const bool n_tl = mode == M_Z1 || mode == M_Z2 || mode == M_Z3 || mode == M_PAETH || mode == M_FILTER;
    const bool n_tr = mode == M_Z1, n_bl = mode == M_Z3;

    // ---------------------------------------------------------------- edge gathering (:118-201)
    const dv::PxRead<pixel, COH> dtop = d - stride;
    if (dsp_edge) {
        const pixel *const edge = reinterpret_cast<const pixel *>(aux) + t.aux_off;
        const int m = dv::imin(w, h);
        for (int k = -(h + m) + lane; k <= w + m; k += 64) E[k] = (int16_t) edge[k];
    } else if (t.kind == DAV1D_HIP_IPRED_DSP_CFL_AC) {
    } else {
    if (n_left) {
        const int sz = h;
        if (have_left) {
            const int px_have = dv::imin(sz, (t.h4 - t.y4) << 2);
            for (int i = lane; i < sz; i += 64) E[-1 - i] = (int16_t) d[dv::imin(i, px_have - 1) * stride - 1];
        } else {
            const int v = have_top ? (int) dtop[0] : ((1 << bitdepth) >> 1) + 1;
            for (int i = lane; i < sz; i += 64) E[-1 - i] = (int16_t) v;
        }
        if (n_bl) {
            const bool have_bl = (!have_left || t.y4 + t.th >= t.h4) ? false : (t.flags & 8) != 0;
            if (have_bl) {
                const int px_have = dv::imin(sz, (t.h4 - t.y4 - t.th) << 2);
                for (int i = lane; i < sz; i += 64) E[-sz - 1 - i] = (int16_t) d[(sz + dv::imin(i, px_have - 1)) * stride - 1];
            } else {
                // replicate left[0] = topleft_out[-sz]
                int v;
                if (have_left) v = d[dv::imin(sz - 1, dv::imin(sz, (t.h4 - t.y4) << 2) - 1) * stride - 1];
                else v = have_top ? (int) dtop[0] : ((1 << bitdepth) >> 1) + 1;
                for (int i = lane; i < sz; i += 64) E[-sz - 1 - i] = (int16_t) v;
            }
        }
    }
    if (n_top) {
        const int sz = w;
        if (have_top) {
            const int px_have = dv::imin(sz, (t.w4 - t.x4) << 2);
            for (int i = lane; i < sz; i += 64) E[1 + i] = (int16_t) dtop[dv::imin(i, px_have - 1)];
        } else {
            const int v = have_left ? (int) d[-1] : ((1 << bitdepth) >> 1) - 1;
            for (int i = lane; i < sz; i += 64) E[1 + i] = (int16_t) v;
        }
        if (n_tr) {
            const bool have_tr = (!have_top || t.x4 + t.tw >= t.w4) ? false : (t.flags & 4) != 0;
            if (have_tr) {
                const int px_have = dv::imin(sz, (t.w4 - t.x4 - t.tw) << 2);
                for (int i = lane; i < sz; i += 64) E[1 + sz + i] = (int16_t) dtop[sz + dv::imin(i, px_have - 1)];
            } else {
                int v;      // top[sz - 1]
                if (have_top) v = dtop[dv::imin(sz - 1, dv::imin(sz, (t.w4 - t.x4) << 2) - 1)];
                else v = have_left ? (int) d[-1] : ((1 << bitdepth) >> 1) - 1;
                for (int i = lane; i < sz; i += 64) E[1 + sz + i] = (int16_t) v;
            }
        }
    }
    dv::wave_sync();
    if (n_tl && lane == 0) {
        int v;
        if (have_left) v = have_top ? dtop[-1] : d[-1];
        else v = have_top ? (int) dtop[0] : (1 << bitdepth) >> 1;
        if (mode == M_Z2 && t.tw + t.th >= 6 && edge_filter) v = ((E[-1] + E[1]) * 5 + v * 6 + 8) >> 4;
        E[0] = (int16_t) v;
    }
    }
    dv::wave_sync();

    // ---------------------------------------------------------------- CfL: ac from luma, then dc + alpha * ac
    if (is_cfl) {
        int16_t *const ac_mem = reinterpret_cast<int16_t *>(aux) + ((uint32_t) t.pal[1] | ((uint32_t) t.pal[2] << 16));
        const int ss_hor = layout != DAV1D_HIP_LAYOUT_I444, ss_ver = layout == DAV1D_HIP_LAYOUT_I420;
        const dv::PxRead<pixel, COH> ypx = { reinterpret_cast<const pixel *>(dst.data[0]) + t.aux_off };
        const int ys = dst.stride[0];
        const int w_pad = t.max_w, h_pad = t.max_h;           // CfL tasks reuse the fields for w_pad / h_pad (4-px units)
        const int wv = w - 4 * w_pad, hv = h - 4 * h_pad;
        int part = 0;
        if (t.kind == DAV1D_HIP_IPRED_DSP_CFL_PRED) {
            for (int i = lane; i < w * h; i += 64) blk[i] = ac_mem[i];
        } else
        for (int i = lane; i < w * h; i += 64) {
            const int y = i >> lw, x = i & (w - 1);
            const int xs = dv::imin(x, wv - 1), yy = dv::imin(y, hv - 1);     // padding replicates the last visible column / row
            const dv::PxRead<pixel, COH> p = ypx + ((yy << ss_ver) * ys + (xs << ss_hor));
            int s = p[0];
            if (ss_hor) s += p[1];
            if (ss_ver) { s += p[ys]; if (ss_hor) s += p[ys + 1]; }
            const int v = s << (1 + !ss_ver + !ss_hor);
            blk[i] = (int16_t) v;
            part += v;
        }
        const int log2sz = __builtin_ctz(w) + __builtin_ctz(h);
        const int mean = t.kind == DAV1D_HIP_IPRED_DSP_CFL_PRED ? 0 : (wave_sum(part) + ((1 << log2sz) >> 1)) >> log2sz;
        if (t.kind == DAV1D_HIP_IPRED_DSP_CFL_AC) {
            for (int i = lane; i < w * h; i += 64) ac_mem[i] = (int16_t) (blk[i] - mean);
            return;
        }
        int dc;
        if (mode == M_DC_128) dc = (bitdepth_max + 1) >> 1;
        else {
            int s = 0;
            if (mode != M_LEFT_DC) for (int i = lane; i < w; i += 64) s += E[1 + i];
            if (mode != M_TOP_DC) for (int i = lane; i < h; i += 64) s += E[-1 - i];
            s = wave_sum(s);
            if (mode == M_TOP_DC) dc = (s + (w >> 1)) >> __builtin_ctz(w);
            else if (mode == M_LEFT_DC) dc = (s + (h >> 1)) >> __builtin_ctz(h);
            else {
                unsigned u = (unsigned) (s + ((w + h) >> 1)) >> __builtin_ctz(w + h);
                if (w != h) { u *= (w > h * 2 || h > w * 2) ? (HBD ? 0x6667u : 0x3334u) : (HBD ? 0xAAABu : 0x5556u); u >>= HBD ? 17 : 16; }
                dc = (int) u;
            }
        }
        const int alpha = t.angle;
        for (int i = lane; i < w * h; i += 64) {
            const int diff = alpha * (blk[i] - mean);
            const int ad = diff < 0 ? -diff : diff;
            const int m = (ad + 32) >> 6;
            o[(i >> lw) * ostride + (i & (w - 1))] = (pixel) dv::iclip(dc + (diff < 0 ? -m : m), 0, bitdepth_max);
        }
        return;
    }

    // ---------------------------------------------------------------- the 14 predictors
    if (mode <= M_DC_128 && mode != M_VERT && mode != M_HOR) {
        int dc;
        if (mode == M_DC_128) dc = (bitdepth_max + 1) >> 1;
        else {
            int s = 0;
            if (mode != M_LEFT_DC) for (int i = lane; i < w; i += 64) s += E[1 + i];
            if (mode != M_TOP_DC) for (int i = lane; i < h; i += 64) s += E[-1 - i];
            s = wave_sum(s);
            if (mode == M_TOP_DC) dc = (s + (w >> 1)) >> __builtin_ctz(w);
            else if (mode == M_LEFT_DC) dc = (s + (h >> 1)) >> __builtin_ctz(h);
            else {      // dc_gen, src/ipred_tmpl.c:150-166
                unsigned u = (unsigned) (s + ((w + h) >> 1)) >> __builtin_ctz(w + h);
                if (w != h) { u *= (w > h * 2 || h > w * 2) ? (HBD ? 0x6667u : 0x3334u) : (HBD ? 0xAAABu : 0x5556u); u >>= HBD ? 17 : 16; }
                dc = (int) u;
            }
        }
        for (int i = i_lo + lane; i < i_hi; i += 64) o[(i >> lw) * ostride + (i & (w - 1))] = (pixel) dc;
        return;
    }

    int dx = 0, dy = 0, max_base = 0, ups_a = 0, ups_l = 0, z3_filtered = 0;
    int16_t *const F = e2 + EC;
    if (mode == M_Z1) {
        // src/ipred_tmpl.c:407-456
        dx = av1_dr_intra_derivative[angle >> 1];
        ups_a = edge_filter ? get_upsample(w + h, 90 - angle, is_sm) : 0;
        if (ups_a) {
            for (int j = lane; j < 2 * (w + h) - 1; j += 64) F[j] = (int16_t) upsample_edge_at(E + 1, j, -1, w + dv::imin(w, h), bitdepth_max);
            max_base = 2 * (w + h) - 2;
            dx <<= 1;
        } else {
            const int fs = edge_filter ? filter_strength(w + h, 90 - angle, is_sm) : 0;
            if (fs) {
                for (int j = lane; j < w + h; j += 64) F[j] = (int16_t) filter_edge_at(E + 1, j, 0, w + h, -1, w + dv::imin(w, h), fs);
                max_base = w + h - 1;
            } else {
                for (int j = lane; j < w + dv::imin(w, h); j += 64) F[j] = E[1 + j];
                max_base = w + dv::imin(w, h) - 1;
            }
        }
    } else if (mode == M_Z2) {
        // src/ipred_tmpl.c:458-541; F[] plays the reference's `topleft` (edge + 64)
        dy = av1_dr_intra_derivative[(angle - 90) >> 1];
        dx = av1_dr_intra_derivative[(180 - angle) >> 1];
        ups_l = edge_filter ? get_upsample(w + h, 180 - angle, is_sm) : 0;
        ups_a = edge_filter ? get_upsample(w + h, angle - 90, is_sm) : 0;
        if (ups_a) {
            for (int j = lane; j < 2 * (w + 1) - 1; j += 64) F[j] = (int16_t) upsample_edge_at(E, j, 0, w + 1, bitdepth_max);
            dx <<= 1;
        } else {
            const int fs = edge_filter ? filter_strength(w + h, angle - 90, is_sm) : 0;
            for (int j = lane; j < w; j += 64) F[1 + j] = fs ? (int16_t) filter_edge_at(E + 1, j, 0, t.max_w, -1, w, fs) : E[1 + j];
        }
        if (ups_l) {
            for (int j = lane; j < 2 * (h + 1) - 1; j += 64) F[-2 * h + j] = (int16_t) upsample_edge_at(E - h, j, 0, h + 1, bitdepth_max);
            dy <<= 1;
        } else {
            const int fs = edge_filter ? filter_strength(w + h, 180 - angle, is_sm) : 0;
            for (int j = lane; j < h; j += 64) F[-h + j] = fs ? (int16_t) filter_edge_at(E - h, j, h - t.max_h, h, 0, h + 1, fs) : E[-h + j];
        }
        dv::wave_sync();
        if (lane == 0) F[0] = E[0];
    } else if (mode == M_Z3) {
        // src/ipred_tmpl.c:543-598; F[k] holds left_out[k]
        dy = av1_dr_intra_derivative[(270 - angle) >> 1];
        ups_l = edge_filter ? get_upsample(w + h, angle - 180, is_sm) : 0;
        if (ups_l) {
            for (int j = lane; j < 2 * (w + h) - 1; j += 64)
                F[j] = (int16_t) upsample_edge_at(E - (w + h), j, dv::imax(w - h, 0), w + h + 1, bitdepth_max);
            max_base = 2 * (w + h) - 2;
            dy <<= 1;
        } else {
            const int fs = edge_filter ? filter_strength(w + h, angle - 180, is_sm) : 0;
            if (fs) {
                for (int j = lane; j < w + h; j += 64)
                    F[j] = (int16_t) filter_edge_at(E - (w + h), j, 0, w + h, dv::imax(w - h, 0), w + h + 1, fs);
                max_base = w + h - 1;
                z3_filtered = 1;
            } else {
                max_base = h + dv::imin(w, h) - 1;
            }
        }
    } else if (mode == M_FILTER) {
        // src/ipred_tmpl.c:616-655 (x86 tap layout of the oracle build: src/tables.c:751-763).  Sub-block (bx, by) =
        // 4x2 pixels needs its left, top and top-left neighbours: anti-diagonal bx + by = s is independent.
        const int8_t *flt = &av1_filter_intra_taps[(angle & 511) * 64];
        const int nbx = w >> 2, nby = h >> 1;
        for (int s = 0; s < nbx + nby - 1; s++) {
            for (int k = lane; k < nby; k += 64) {
                const int by = k, bx = s - k;
                if (bx < 0 || bx >= nbx) continue;
                const int x = bx * 4, y = by * 2;
                int p[7];
                // p0 = topleft, p1..p4 = top, p5, p6 = left
#pragma unroll
                for (int q = 0; q < 5; q++) p[q] = y ? blk[(y - 1) * w + x - 1 + q] : E[x + q];
                if (y && !x) p[0] = E[-y];
                p[5] = x ? blk[y * w + x - 1] : E[-1 - y];
                p[6] = x ? blk[(y + 1) * w + x - 1] : E[-2 - y];
#pragma unroll
                for (int yy = 0; yy < 2; yy++)
#pragma unroll
                    for (int xx = 0; xx < 4; xx++) {
                        const int8_t *f = flt + (yy * 4 + xx) * 2;
                        const int acc = f[0] * p[0] + f[1] * p[1] + f[16] * p[2] + f[17] * p[3] + f[32] * p[4] + f[33] * p[5] + f[48] * p[6];
                        blk[(y + yy) * w + x + xx] = (int16_t) dv::iclip((acc + 8) >> 4, 0, bitdepth_max);
                    }
            }
            dv::wave_sync();
        }
        for (int i = lane; i < w * h; i += 64) o[(i >> lw) * ostride + (i & (w - 1))] = (pixel) blk[i];
        return;
    }
    dv::wave_sync();

    const int right = E[w], bottom = E[-h], tl = E[0];
    for (int i = i_lo + lane; i < i_hi; i += 64) {
        const int y = i >> lw, x = i & (w - 1);
        int v;
        switch (mode) {
        case M_VERT: v = E[1 + x]; break;
        case M_HOR: v = E[-1 - y]; break;
        case M_PAETH: {
            const int left = E[-1 - y], top = E[1 + x];
            const int base = left + top - tl;
            const int ld = base > left ? base - left : left - base, td = base > top ? base - top : top - base;
            const int tld = base > tl ? base - tl : tl - base;
            v = (ld <= td && ld <= tld) ? left : (td <= tld ? top : tl);
            break;
        }
        case M_SMOOTH: {
            const int wv = av1_sm_weights[h + y], wh = av1_sm_weights[w + x];
            v = (wv * E[1 + x] + (256 - wv) * bottom + wh * E[-1 - y] + (256 - wh) * right + 256) >> 9;
            break;
        }
        case M_SMOOTH_V: { const int wv = av1_sm_weights[h + y]; v = (wv * E[1 + x] + (256 - wv) * bottom + 128) >> 8; break; }
        case M_SMOOTH_H: { const int wh = av1_sm_weights[w + x]; v = (wh * E[-1 - y] + (256 - wh) * right + 128) >> 8; break; }
        case M_Z1: {
            const int xpos = dx * (y + 1), frac = xpos & 0x3E;
            const int base = (xpos >> 6) + x * (1 + ups_a);
            v = base < max_base ? (F[base] * (64 - frac) + F[base + 1] * frac + 32) >> 6 : F[max_base];
            break;
        }
        case M_Z2: {
            const int xpos = ((1 + ups_a) << 6) - dx * (y + 1);
            const int base_x = (xpos >> 6) + x * (1 + ups_a), frac_x = xpos & 0x3E;
            if (base_x >= 0) {
                v = (F[base_x] * (64 - frac_x) + F[base_x + 1] * frac_x + 32) >> 6;
            } else {
                const int ypos = (y << (6 + ups_l)) - dy * (x + 1);
                const int base_y = ypos >> 6, frac_y = ypos & 0x3E;
                const int16_t *left = F - (1 + ups_l);
                v = (left[-base_y] * (64 - frac_y) + left[-(base_y + 1)] * frac_y + 32) >> 6;
            }
            break;
        }
        default: {      // M_Z3
            const int ypos = dy * (x + 1), frac = ypos & 0x3E;
            const int base = (ypos >> 6) + y * (1 + ups_l);
            // left[-k]: unfiltered = topleft_in[-1 - k]; filtered = left_out[w+h-1-k]; upsampled = left_out[2(w+h)-2-k]
            const int16_t *left = ups_l ? F + 2 * (w + h) - 2 : z3_filtered ? F + w + h - 1 : E - 1;
            v = base < max_base ? (left[-base] * (64 - frac) + left[-(base + 1)] * frac + 32) >> 6 : left[-max_base];
            break;
        }
        }
        o[y * ostride + x] = (pixel) v;
    }
}

} // namespace

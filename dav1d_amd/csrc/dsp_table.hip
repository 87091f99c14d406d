// Kernel-level drop-in: function pointers with the reference's DSP signatures
// (src/itx.h:37-40, src/mc.h:38-122) that run ONE call through the batched HIP
// kernels: stage the host rectangles on the device, launch, copy back.  This is the
// plug a maintainer gets by calling dav1d_hip_dsp_init_{8,16}bpc after the reference's
// own dav1d_*_dsp_init (src/decode.c:3387-3415); it is meant for parity runs through
// unmodified call sites, not for speed (the batched API is the fast path).
// There is no CPU fallback: without a usable device init returns -ENODEV and the
// table stays zeroed.
#include "dsp_stage.h"

std::mutex &dsp_stage::mutex() { static std::mutex m; return m; }

namespace {

using namespace dsp_stage;
#define g_mtx (dsp_stage::mutex())

const uint8_t tw[19] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
const uint8_t th[19] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };

template <typename pixel, typename coef>
void itx_call(int tx, int txtp, pixel *dst, ptrdiff_t stride, coef *coeff, int eob, int bpc) {
    std::lock_guard<std::mutex> lk(g_mtx);
    const int w = tw[tx], h = th[tx], sw = w < 32 ? w : 32, sh = h < 32 ? h : 32;
    Stage s(1 << 20);
    Dav1dHipPicture pic = scratch_pic(s, w, h, bpc);
    coef *dcf = (coef *) s.take(sizeof(coef) * sw * sh);
    up2d(s.c, pic.p[0].data, pic.p[0].stride, dst, stride, w * sizeof(pixel), h);
    hipMemcpyAsync(dcf, coeff, sizeof(coef) * sw * sh, hipMemcpyHostToDevice, s.c->stream);
    Dav1dHipItxTask t;
    memset(&t, 0, sizeof(t));
    t.eob = (int16_t) eob; t.tx = tx; t.txtp = txtp;
    if (dav1d_hip_itx_add_batch(s.c, &pic, &t, 1, dcf)) abort();
    down2d(s.c, dst, stride, pic.p[0].data, pic.p[0].stride, w * sizeof(pixel), h);
    hipMemcpyAsync(coeff, dcf, sizeof(coef) * sw * sh, hipMemcpyDeviceToHost, s.c->stream);
    hipStreamSynchronize(s.c->stream);
}

template <int TX, int TXTP> void itx8(uint8_t *d, ptrdiff_t st, int16_t *cf, int eob) { itx_call<uint8_t, int16_t>(TX, TXTP, d, st, cf, eob, 8); }
template <int TX, int TXTP> void itx16(uint16_t *d, ptrdiff_t st, int32_t *cf, int eob, int bdmax) {
    itx_call<uint16_t, int32_t>(TX, TXTP, d, st, cf, eob, bpc_of(bdmax));
}

// put (tmp == NULL) or prep (dst == NULL) of one block
template <typename pixel>
void mc_call(int filter_2d, pixel *dst, ptrdiff_t dst_stride, int16_t *tmp, const pixel *src, ptrdiff_t src_stride,
             int w, int h, int mx, int my, int bpc) {
    std::lock_guard<std::mutex> lk(g_mtx);
    Stage s(1 << 20);
    // stage exactly the rows / columns the reference function reads
    const int l = mx ? 3 : 0, r = mx ? 4 : 0, t = my ? 3 : 0, b = my ? 4 : 0;
    if (filter_2d == 9) { /* bilinear reads +1 only */ }
    const int rw = w + 8, rh = h + 7;
    Dav1dHipPicture ref = scratch_pic(s, rw, rh, bpc);
    const int bl = filter_2d == 9 ? 0 : l, br = filter_2d == 9 ? (mx ? 1 : 0) : r;
    const int bt = filter_2d == 9 ? 0 : t, bb = filter_2d == 9 ? (my ? 1 : 0) : b;
    const ptrdiff_t sp = src_stride / (ptrdiff_t) sizeof(pixel);
    up2d(s.c, (pixel *) ref.p[0].data + (3 - bt) * (ref.p[0].stride / (ptrdiff_t) sizeof(pixel)) + (4 - bl), ref.p[0].stride,
         src - bt * sp - bl, src_stride, (size_t) (w + bl + br) * sizeof(pixel), h + bt + bb);
    Dav1dHipMcTask k;
    memset(&k, 0, sizeof(k));
    k.src_x = 4; k.src_y = 3; k.w = w; k.h = h; k.mx = mx; k.my = my; k.filter_2d = filter_2d;
    if (dst) {
        Dav1dHipPicture out = scratch_pic(s, w, h, bpc);
        k.kind = DAV1D_HIP_MC_PUT;
        if (dav1d_hip_mc_batch(s.c, &out, &ref, 1, &k, 1, nullptr)) abort();
        down2d(s.c, dst, dst_stride, out.p[0].data, out.p[0].stride, w * sizeof(pixel), h);
    } else {
        int16_t *dt = (int16_t *) s.take(sizeof(int16_t) * w * h);
        Dav1dHipPicture out = scratch_pic(s, 4, 4, bpc);   // unused dst
        k.kind = DAV1D_HIP_MC_PREP;
        if (dav1d_hip_mc_batch(s.c, &out, &ref, 1, &k, 1, dt)) abort();
        hipMemcpyAsync(tmp, dt, sizeof(int16_t) * w * h, hipMemcpyDeviceToHost, s.c->stream);
    }
    hipStreamSynchronize(s.c->stream);
}

template <int F> void mc8(uint8_t *d, ptrdiff_t ds, const uint8_t *s, ptrdiff_t ss, int w, int h, int mx, int my) {
    mc_call<uint8_t>(F, d, ds, nullptr, s, ss, w, h, mx, my, 8);
}
template <int F> void mc16(uint16_t *d, ptrdiff_t ds, const uint16_t *s, ptrdiff_t ss, int w, int h, int mx, int my, int bdmax) {
    mc_call<uint16_t>(F, d, ds, nullptr, s, ss, w, h, mx, my, bpc_of(bdmax));
}
template <int F> void mct8(int16_t *t, const uint8_t *s, ptrdiff_t ss, int w, int h, int mx, int my) {
    mc_call<uint8_t>(F, nullptr, 0, t, s, ss, w, h, mx, my, 8);
}
template <int F> void mct16(int16_t *t, const uint16_t *s, ptrdiff_t ss, int w, int h, int mx, int my, int bdmax) {
    mc_call<uint16_t>(F, nullptr, 0, t, s, ss, w, h, mx, my, bpc_of(bdmax));
}

template <typename pixel>
void comp_call(int kind, int ss, pixel *dst, ptrdiff_t dst_stride, const int16_t *tmp1, const int16_t *tmp2,
               int w, int h, int arg, const uint8_t *mask_in, uint8_t *mask_out, int bpc) {
    std::lock_guard<std::mutex> lk(g_mtx);
    Stage s(1 << 20);
    Dav1dHipPicture out = scratch_pic(s, w, h, bpc);
    int16_t *dp = (int16_t *) s.take(sizeof(int16_t) * w * h * 2);
    uint8_t *dm = (uint8_t *) s.take((size_t) w * h);
    hipMemcpyAsync(dp, tmp1, sizeof(int16_t) * w * h, hipMemcpyHostToDevice, s.c->stream);
    hipMemcpyAsync(dp + w * h, tmp2, sizeof(int16_t) * w * h, hipMemcpyHostToDevice, s.c->stream);
    size_t mbytes = 0;
    if (kind == DAV1D_HIP_COMP_MASK) {
        mbytes = (size_t) w * h;
        hipMemcpyAsync(dm, mask_in, mbytes, hipMemcpyHostToDevice, s.c->stream);
    } else if (kind == DAV1D_HIP_COMP_WMASK) {
        mbytes = (size_t) (w >> (ss > 0)) * (h >> (ss == 2));
    }
    Dav1dHipCompTask k;
    memset(&k, 0, sizeof(k));
    k.tmp1_off = 0; k.tmp2_off = w * h; k.w = w; k.h = h; k.kind = kind; k.arg = (int8_t) arg; k.ss = ss;
    if (dav1d_hip_comp_batch(s.c, &out, &k, 1, dp, dm)) abort();
    down2d(s.c, dst, dst_stride, out.p[0].data, out.p[0].stride, w * sizeof(pixel), h);
    if (kind == DAV1D_HIP_COMP_WMASK) hipMemcpyAsync(mask_out, dm, mbytes, hipMemcpyDeviceToHost, s.c->stream);
    hipStreamSynchronize(s.c->stream);
}

void avg8(uint8_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h) { comp_call<uint8_t>(0, 0, d, ds, a, b, w, h, 0, nullptr, nullptr, 8); }
void avg16(uint16_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, int bm) { comp_call<uint16_t>(0, 0, d, ds, a, b, w, h, 0, nullptr, nullptr, bpc_of(bm)); }
void w_avg8(uint8_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, int wt) { comp_call<uint8_t>(1, 0, d, ds, a, b, w, h, wt, nullptr, nullptr, 8); }
void w_avg16(uint16_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, int wt, int bm) { comp_call<uint16_t>(1, 0, d, ds, a, b, w, h, wt, nullptr, nullptr, bpc_of(bm)); }
void mask8(uint8_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, const uint8_t *m) { comp_call<uint8_t>(2, 0, d, ds, a, b, w, h, 0, m, nullptr, 8); }
void mask16(uint16_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, const uint8_t *m, int bm) { comp_call<uint16_t>(2, 0, d, ds, a, b, w, h, 0, m, nullptr, bpc_of(bm)); }
template <int SS> void w_mask8(uint8_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, uint8_t *m, int sign) { comp_call<uint8_t>(3, SS, d, ds, a, b, w, h, sign, nullptr, m, 8); }
template <int SS> void w_mask16(uint16_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, uint8_t *m, int sign, int bm) { comp_call<uint16_t>(3, SS, d, ds, a, b, w, h, sign, nullptr, m, bpc_of(bm)); }

// ---- blend / blend_v / blend_h
template <typename pixel>
void blend_call(int kind, pixel *dst, ptrdiff_t dst_stride, const pixel *tmp, int w, int h, const uint8_t *mask, int bpc) {
    std::lock_guard<std::mutex> lk(g_mtx);
    Stage s(1 << 20);
    Dav1dHipPicture out = scratch_pic(s, w, h, bpc);
    pixel *dt = (pixel *) s.take(sizeof(pixel) * w * h);
    uint8_t *dm = (uint8_t *) s.take((size_t) w * h);
    up2d(s.c, out.p[0].data, out.p[0].stride, dst, dst_stride, w * sizeof(pixel), h);
    hipMemcpyAsync(dt, tmp, sizeof(pixel) * w * h, hipMemcpyHostToDevice, s.c->stream);
    if (mask) hipMemcpyAsync(dm, mask, (size_t) w * h, hipMemcpyHostToDevice, s.c->stream);
    Dav1dHipCompTask k;
    memset(&k, 0, sizeof(k));
    k.w = w; k.h = h; k.kind = kind;
    if (dav1d_hip_comp_batch(s.c, &out, &k, 1, (const int16_t *) dt, dm)) abort();
    down2d(s.c, dst, dst_stride, out.p[0].data, out.p[0].stride, w * sizeof(pixel), h);
    s.sync();
}
void blend8(uint8_t *d, ptrdiff_t ds, const uint8_t *t, int w, int h, const uint8_t *m) { blend_call<uint8_t>(DAV1D_HIP_COMP_BLEND, d, ds, t, w, h, m, 8); }
void blend16(uint16_t *d, ptrdiff_t ds, const uint16_t *t, int w, int h, const uint8_t *m) { blend_call<uint16_t>(DAV1D_HIP_COMP_BLEND, d, ds, t, w, h, m, 10); }
void blend_v8(uint8_t *d, ptrdiff_t ds, const uint8_t *t, int w, int h) { blend_call<uint8_t>(DAV1D_HIP_COMP_BLEND_V, d, ds, t, w, h, nullptr, 8); }
void blend_v16(uint16_t *d, ptrdiff_t ds, const uint16_t *t, int w, int h) { blend_call<uint16_t>(DAV1D_HIP_COMP_BLEND_V, d, ds, t, w, h, nullptr, 10); }
void blend_h8(uint8_t *d, ptrdiff_t ds, const uint8_t *t, int w, int h) { blend_call<uint8_t>(DAV1D_HIP_COMP_BLEND_H, d, ds, t, w, h, nullptr, 8); }
void blend_h16(uint16_t *d, ptrdiff_t ds, const uint16_t *t, int w, int h) { blend_call<uint16_t>(DAV1D_HIP_COMP_BLEND_H, d, ds, t, w, h, nullptr, 10); }

// ---- warp8x8 (dst != NULL) / warp8x8t (tmp != NULL): the 15x15 window around src
template <typename pixel>
void warp_call(pixel *dst, ptrdiff_t dst_stride, int16_t *tmp, ptrdiff_t tmp_stride, const pixel *src, ptrdiff_t src_stride,
               const int16_t *abcd, int mx, int my, int bpc) {
    std::lock_guard<std::mutex> lk(g_mtx);
    Stage s(1 << 20);
    Dav1dHipPicture ref = scratch_pic(s, 15, 15, bpc);
    up2d(s.c, ref.p[0].data, ref.p[0].stride, src - 3 * (src_stride / (ptrdiff_t) sizeof(pixel)) - 3, src_stride, 15 * sizeof(pixel), 15);
    Dav1dHipPicture out = scratch_pic(s, 8, 8, bpc);
    int16_t *dt = (int16_t *) s.take(sizeof(int16_t) * 64);
    Dav1dHipWarpTask k;
    memset(&k, 0, sizeof(k));
    k.src_x = 3; k.src_y = 3; k.mx = mx; k.my = my; k.tmp_stride = 8;
    memcpy(k.abcd, abcd, sizeof(k.abcd));
    k.kind = dst ? DAV1D_HIP_MC_PUT : DAV1D_HIP_MC_PREP;
    if (dav1d_hip_warp_batch(s.c, &out, &ref, 1, &k, 1, dt)) abort();
    if (dst) down2d(s.c, dst, dst_stride, out.p[0].data, out.p[0].stride, 8 * sizeof(pixel), 8);
    else down2d(s.c, tmp, tmp_stride * (ptrdiff_t) sizeof(int16_t), dt, 8 * sizeof(int16_t), 8 * sizeof(int16_t), 8);
    s.sync();
}
void warp8(uint8_t *d, ptrdiff_t ds, const uint8_t *s, ptrdiff_t ss, const int16_t *abcd, int mx, int my) { warp_call<uint8_t>(d, ds, nullptr, 0, s, ss, abcd, mx, my, 8); }
void warp16(uint16_t *d, ptrdiff_t ds, const uint16_t *s, ptrdiff_t ss, const int16_t *abcd, int mx, int my, int bm) { warp_call<uint16_t>(d, ds, nullptr, 0, s, ss, abcd, mx, my, bpc_of(bm)); }
void warpt8(int16_t *t, ptrdiff_t ts, const uint8_t *s, ptrdiff_t ss, const int16_t *abcd, int mx, int my) { warp_call<uint8_t>(nullptr, 0, t, ts, s, ss, abcd, mx, my, 8); }
void warpt16(int16_t *t, ptrdiff_t ts, const uint16_t *s, ptrdiff_t ss, const int16_t *abcd, int mx, int my, int bm) { warp_call<uint16_t>(nullptr, 0, t, ts, s, ss, abcd, mx, my, bpc_of(bm)); }

// ---- mc_scaled / mct_scaled: the window the reference functions read (src/mc_tmpl.c:189-244, 491-531)
template <typename pixel>
void scaled_call(int filter_2d, pixel *dst, ptrdiff_t dst_stride, int16_t *tmp, const pixel *src, ptrdiff_t src_stride,
                 int w, int h, int mx, int my, int dx, int dy, int bpc) {
    std::lock_guard<std::mutex> lk(g_mtx);
    Stage s(4 << 20);
    const bool bilin = filter_2d == 9;
    const int lo = bilin ? 0 : 3, hi = bilin ? 1 : 4;
    const int cols = ((mx + (w - 1) * dx) >> 10) + 1 + lo + hi, rows = ((my + (h - 1) * dy) >> 10) + 1 + lo + hi;
    Dav1dHipPicture ref = scratch_pic(s, cols, rows, bpc);
    up2d(s.c, ref.p[0].data, ref.p[0].stride, src - lo * (src_stride / (ptrdiff_t) sizeof(pixel)) - lo, src_stride, cols * sizeof(pixel), rows);
    Dav1dHipPicture out = scratch_pic(s, w, h, bpc);
    int16_t *dt = (int16_t *) s.take(sizeof(int16_t) * w * h);
    Dav1dHipMcScaledTask k;
    memset(&k, 0, sizeof(k));
    k.src_x = lo; k.src_y = lo; k.mx = mx; k.my = my; k.dx = dx; k.dy = dy; k.w = w; k.h = h; k.filter_2d = filter_2d;
    k.kind = dst ? DAV1D_HIP_MC_PUT : DAV1D_HIP_MC_PREP;
    if (dav1d_hip_mc_scaled_batch(s.c, &out, &ref, 1, &k, 1, dt)) abort();
    if (dst) down2d(s.c, dst, dst_stride, out.p[0].data, out.p[0].stride, w * sizeof(pixel), h);
    else hipMemcpyAsync(tmp, dt, sizeof(int16_t) * w * h, hipMemcpyDeviceToHost, s.c->stream);
    s.sync();
}
template <int F> void mcs8(uint8_t *d, ptrdiff_t ds, const uint8_t *s, ptrdiff_t ss, int w, int h, int mx, int my, int dx, int dy) {
    scaled_call<uint8_t>(F, d, ds, nullptr, s, ss, w, h, mx, my, dx, dy, 8);
}
template <int F> void mcs16(uint16_t *d, ptrdiff_t ds, const uint16_t *s, ptrdiff_t ss, int w, int h, int mx, int my, int dx, int dy, int bm) {
    scaled_call<uint16_t>(F, d, ds, nullptr, s, ss, w, h, mx, my, dx, dy, bpc_of(bm));
}
template <int F> void mcts8(int16_t *t, const uint8_t *s, ptrdiff_t ss, int w, int h, int mx, int my, int dx, int dy) {
    scaled_call<uint8_t>(F, nullptr, 0, t, s, ss, w, h, mx, my, dx, dy, 8);
}
template <int F> void mcts16(int16_t *t, const uint16_t *s, ptrdiff_t ss, int w, int h, int mx, int my, int dx, int dy, int bm) {
    scaled_call<uint16_t>(F, nullptr, 0, t, s, ss, w, h, mx, my, dx, dy, bpc_of(bm));
}

// ---- emu_edge: only the part of the plane the clamped window can touch is staged
template <typename pixel>
void emu_edge_call(intptr_t bw, intptr_t bh, intptr_t iw, intptr_t ih, intptr_t x, intptr_t y, pixel *dst, ptrdiff_t dst_stride,
                   const pixel *ref, ptrdiff_t ref_stride, int bpc) {
    std::lock_guard<std::mutex> lk(g_mtx);
    auto clip = [](intptr_t v, intptr_t lo, intptr_t hi) { return v < lo ? lo : v > hi ? hi : v; };
    const intptr_t xa = clip(x, 0, iw - 1), xb = clip(x + bw - 1, 0, iw - 1), ya = clip(y, 0, ih - 1), yb = clip(y + bh - 1, 0, ih - 1);
    const int sw = (int) (xb - xa + 1), sh = (int) (yb - ya + 1);
    Stage s((size_t) (sw + 16) * sh * sizeof(pixel) + (size_t) (bw + 16) * bh * sizeof(pixel) + (1 << 16));
    Dav1dHipPicture in = scratch_pic(s, sw, sh, bpc), out = scratch_pic(s, (int) bw, (int) bh, bpc);
    up2d(s.c, in.p[0].data, in.p[0].stride, ref + ya * (ref_stride / (ptrdiff_t) sizeof(pixel)) + xa, ref_stride, sw * sizeof(pixel), sh);
    if (dav1d_hip_emu_edge(s.c, bpc, bw, bh, sw, sh, x - xa, y - ya, out.p[0].data, out.p[0].stride, in.p[0].data, in.p[0].stride)) abort();
    down2d(s.c, dst, dst_stride, out.p[0].data, out.p[0].stride, bw * sizeof(pixel), (int) bh);
    s.sync();
}
void emu8(intptr_t bw, intptr_t bh, intptr_t iw, intptr_t ih, intptr_t x, intptr_t y, uint8_t *d, ptrdiff_t ds, const uint8_t *r, ptrdiff_t rs) {
    emu_edge_call<uint8_t>(bw, bh, iw, ih, x, y, d, ds, r, rs, 8);
}
void emu16(intptr_t bw, intptr_t bh, intptr_t iw, intptr_t ih, intptr_t x, intptr_t y, uint16_t *d, ptrdiff_t ds, const uint16_t *r, ptrdiff_t rs) {
    emu_edge_call<uint16_t>(bw, bh, iw, ih, x, y, d, ds, r, rs, 10);
}

// ---- resize
template <typename pixel>
void resize_call(pixel *dst, ptrdiff_t dst_stride, const pixel *src, ptrdiff_t src_stride, int dst_w, int h, int src_w, int dx, int mx0, int bpc) {
    std::lock_guard<std::mutex> lk(g_mtx);
    Stage s((size_t) (dst_w + src_w + 32) * h * sizeof(pixel) + (1 << 16));
    Dav1dHipPicture in = scratch_pic(s, src_w, h, bpc), out = scratch_pic(s, dst_w, h, bpc);
    up2d(s.c, in.p[0].data, in.p[0].stride, src, src_stride, src_w * sizeof(pixel), h);
    if (dav1d_hip_resize(s.c, &out, &in, 0, dst_w, 0, h, src_w, dx, mx0)) abort();
    down2d(s.c, dst, dst_stride, out.p[0].data, out.p[0].stride, dst_w * sizeof(pixel), h);
    s.sync();
}
void resize8(uint8_t *d, ptrdiff_t ds, const uint8_t *s, ptrdiff_t ss, int dw, int h, int sw, int dx, int mx) { resize_call<uint8_t>(d, ds, s, ss, dw, h, sw, dx, mx, 8); }
void resize16(uint16_t *d, ptrdiff_t ds, const uint16_t *s, ptrdiff_t ss, int dw, int h, int sw, int dx, int mx, int bm) { resize_call<uint16_t>(d, ds, s, ss, dw, h, sw, dx, mx, bpc_of(bm)); }

bool legal(int tx, int txtp) {
    if (txtp == 16) return tx == 0;
    const int w = tw[tx], h = th[tx], mx = w > h ? w : h;
    if (mx == 64) return txtp == 0;
    if (mx == 32) return txtp == 0 || txtp == 9;
    if (w == 16 && h == 16) return txtp <= 11;
    return true;
}

template <int TX, int TXTP = 0>
struct FillItx {
    static void run(Dav1dHipInvTxfmDSPContext8 *c8, Dav1dHipInvTxfmDSPContext16 *c16) {
        if (legal(TX, TXTP)) {
            if (c8) c8->itxfm_add[TX][TXTP] = itx8<TX, TXTP>;
            if (c16) c16->itxfm_add[TX][TXTP] = itx16<TX, TXTP>;
        }
        if constexpr (TXTP + 1 < 17) FillItx<TX, TXTP + 1>::run(c8, c16);
        else if constexpr (TX + 1 < 19) FillItx<TX + 1, 0>::run(c8, c16);
    }
};

template <int F = 0>
struct FillMc {
    static void run(Dav1dHipMCDSPContext8 *c8, Dav1dHipMCDSPContext16 *c16) {
        if (c8) { c8->mc[F] = mc8<F>; c8->mct[F] = mct8<F>; c8->mc_scaled[F] = mcs8<F>; c8->mct_scaled[F] = mcts8<F>; }
        if (c16) { c16->mc[F] = mc16<F>; c16->mct[F] = mct16<F>; c16->mc_scaled[F] = mcs16<F>; c16->mct_scaled[F] = mcts16<F>; }
        if constexpr (F + 1 < 10) FillMc<F + 1>::run(c8, c16);
    }
};

} // namespace

extern "C" int dav1d_hip_dsp_init_8bpc(Dav1dHipDSPContext8 *c) {
    if (!c) return -EINVAL;
    memset(c, 0, sizeof(*c));
    if (!dav1d_hip_default_context()) return -ENODEV;
    FillItx<0, 0>::run(&c->itx, nullptr);
    FillMc<0>::run(&c->mc, nullptr);
    c->mc.avg = avg8; c->mc.w_avg = w_avg8; c->mc.mask = mask8;
    c->mc.w_mask[0] = w_mask8<0>; c->mc.w_mask[1] = w_mask8<1>; c->mc.w_mask[2] = w_mask8<2>;
    c->mc.blend = blend8; c->mc.blend_v = blend_v8; c->mc.blend_h = blend_h8;
    c->mc.warp8x8 = warp8; c->mc.warp8x8t = warpt8; c->mc.emu_edge = emu8; c->mc.resize = resize8;
    dav1d_hip_dsp_fill_post_8(c);
    return 0;
}

extern "C" int dav1d_hip_dsp_init_16bpc(Dav1dHipDSPContext16 *c, int bpc) {
    if (!c || (bpc != 10 && bpc != 12)) return -EINVAL;
    memset(c, 0, sizeof(*c));
    if (!dav1d_hip_default_context()) return -ENODEV;
    FillItx<0, 0>::run(nullptr, &c->itx);
    FillMc<0>::run(nullptr, &c->mc);
    c->mc.avg = avg16; c->mc.w_avg = w_avg16; c->mc.mask = mask16;
    c->mc.w_mask[0] = w_mask16<0>; c->mc.w_mask[1] = w_mask16<1>; c->mc.w_mask[2] = w_mask16<2>;
    c->mc.blend = blend16; c->mc.blend_v = blend_v16; c->mc.blend_h = blend_h16;
    c->mc.warp8x8 = warp16; c->mc.warp8x8t = warpt16; c->mc.emu_edge = emu16; c->mc.resize = resize16;
    dav1d_hip_dsp_fill_post_16(c);
    return 0;
}

// Kernel-level drop-in: function pointers with the reference's DSP signatures
// (src/itx.h:37-40, src/mc.h:38-122) that run ONE call through the batched HIP
// kernels: stage the host rectangles on the device, launch, copy back.  This is the
// plug a maintainer gets by calling dav1d_hip_dsp_init_{8,16}bpc after the reference's
// own dav1d_*_dsp_init (src/decode.c:3387-3415); it is meant for parity runs through
// unmodified call sites, not for speed (the batched API is the fast path).
// There is no CPU fallback: without a usable device init returns -ENODEV and the
// table stays zeroed.
#include "capi.h"
#include <string.h>
#include <mutex>

namespace {

std::mutex g_mtx;   // the single-call wrappers share one scratch arena

struct Stage {
    Dav1dHipContext *c;
    uint8_t *base;
    size_t used, cap;
    bool ok;
    explicit Stage(size_t bytes) : c(dav1d_hip_default_context()), base(nullptr), used(0), cap(bytes), ok(false) {
        void *p = nullptr;
        if (c && !dav1d_hip_scratch(c, bytes, &p)) { base = (uint8_t *) p; ok = true; }
    }
    void *take(size_t bytes) {
        used = (used + 255) & ~(size_t) 255;
        void *p = base + used;
        used += bytes;
        if (used > cap) ok = false;
        return p;
    }
};

// a device picture with one plane of w x h pixels living in the scratch arena
Dav1dHipPicture scratch_pic(Stage &s, int w, int h, int bpc) {
    Dav1dHipPicture p;
    memset(&p, 0, sizeof(p));
    const int bps = bpc > 8 ? 2 : 1;
    p.bpc = bpc;
    p.layout = DAV1D_HIP_LAYOUT_I400;
    p.p[0].stride = (ptrdiff_t) ((w * bps + 15) & ~15);
    p.p[0].w = w;
    p.p[0].h = h;
    p.p[0].data = s.take((size_t) p.p[0].stride * h);
    return p;
}

void up2d(Dav1dHipContext *c, void *dev, ptrdiff_t dstride, const void *host, ptrdiff_t hstride, size_t row_bytes, int rows) {
    hipMemcpy2DAsync(dev, dstride, host, hstride, row_bytes, rows, hipMemcpyHostToDevice, c->stream);
}
void down2d(Dav1dHipContext *c, void *host, ptrdiff_t hstride, const void *dev, ptrdiff_t dstride, size_t row_bytes, int rows) {
    hipMemcpy2DAsync(host, hstride, dev, dstride, row_bytes, rows, hipMemcpyDeviceToHost, c->stream);
}

const uint8_t tw[19] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
const uint8_t th[19] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };

template <typename pixel, typename coef>
void itx_call(int tx, int txtp, pixel *dst, ptrdiff_t stride, coef *coeff, int eob, int bpc) {
    std::lock_guard<std::mutex> lk(g_mtx);
    const int w = tw[tx], h = th[tx], sw = w < 32 ? w : 32, sh = h < 32 ? h : 32;
    Stage s(1 << 20);
    if (!s.ok) abort();
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
    itx_call<uint16_t, int32_t>(TX, TXTP, d, st, cf, eob, bdmax == 0x3ff ? 10 : 12);
}

// put (tmp == NULL) or prep (dst == NULL) of one block
template <typename pixel>
void mc_call(int filter_2d, pixel *dst, ptrdiff_t dst_stride, int16_t *tmp, const pixel *src, ptrdiff_t src_stride,
             int w, int h, int mx, int my, int bpc) {
    std::lock_guard<std::mutex> lk(g_mtx);
    Stage s(1 << 20);
    if (!s.ok) abort();
    // stage exactly the rows / columns the reference function reads
    const int l = mx ? 3 : 0, r = mx ? 4 : 0, t = my ? 3 : 0, b = my ? 4 : 0;
    if (filter_2d == 9) { /* bilinear reads +1 only */ }
    const int rw = w + 8, rh = h + 7;
    Dav1dHipPicture ref = scratch_pic(s, rw, rh, bpc);
    hipMemsetAsync(ref.p[0].data, 0, (size_t) ref.p[0].stride * rh, s.c->stream);
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
    mc_call<uint16_t>(F, d, ds, nullptr, s, ss, w, h, mx, my, bdmax == 0x3ff ? 10 : 12);
}
template <int F> void mct8(int16_t *t, const uint8_t *s, ptrdiff_t ss, int w, int h, int mx, int my) {
    mc_call<uint8_t>(F, nullptr, 0, t, s, ss, w, h, mx, my, 8);
}
template <int F> void mct16(int16_t *t, const uint16_t *s, ptrdiff_t ss, int w, int h, int mx, int my, int bdmax) {
    mc_call<uint16_t>(F, nullptr, 0, t, s, ss, w, h, mx, my, bdmax == 0x3ff ? 10 : 12);
}

template <typename pixel>
void comp_call(int kind, int ss, pixel *dst, ptrdiff_t dst_stride, const int16_t *tmp1, const int16_t *tmp2,
               int w, int h, int arg, const uint8_t *mask_in, uint8_t *mask_out, int bpc) {
    std::lock_guard<std::mutex> lk(g_mtx);
    Stage s(1 << 20);
    if (!s.ok) abort();
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
void avg16(uint16_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, int bm) { comp_call<uint16_t>(0, 0, d, ds, a, b, w, h, 0, nullptr, nullptr, bm == 0x3ff ? 10 : 12); }
void w_avg8(uint8_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, int wt) { comp_call<uint8_t>(1, 0, d, ds, a, b, w, h, wt, nullptr, nullptr, 8); }
void w_avg16(uint16_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, int wt, int bm) { comp_call<uint16_t>(1, 0, d, ds, a, b, w, h, wt, nullptr, nullptr, bm == 0x3ff ? 10 : 12); }
void mask8(uint8_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, const uint8_t *m) { comp_call<uint8_t>(2, 0, d, ds, a, b, w, h, 0, m, nullptr, 8); }
void mask16(uint16_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, const uint8_t *m, int bm) { comp_call<uint16_t>(2, 0, d, ds, a, b, w, h, 0, m, nullptr, bm == 0x3ff ? 10 : 12); }
template <int SS> void w_mask8(uint8_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, uint8_t *m, int sign) { comp_call<uint8_t>(3, SS, d, ds, a, b, w, h, sign, nullptr, m, 8); }
template <int SS> void w_mask16(uint16_t *d, ptrdiff_t ds, const int16_t *a, const int16_t *b, int w, int h, uint8_t *m, int sign, int bm) { comp_call<uint16_t>(3, SS, d, ds, a, b, w, h, sign, nullptr, m, bm == 0x3ff ? 10 : 12); }

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
        if (c8) { c8->mc[F] = mc8<F>; c8->mct[F] = mct8<F>; }
        if (c16) { c16->mc[F] = mc16<F>; c16->mct[F] = mct16<F>; }
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
    return 0;
}

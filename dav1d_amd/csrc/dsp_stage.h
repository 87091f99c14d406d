// Staging helpers shared by the reference-signature wrappers (dsp_table*.hip): every wrapper copies the host
// rectangles a DSP entry touches into small device pictures carved out of the default context's scratch arena,
// runs the batched kernel on one task, and copies the result back.
#pragma once
#include "capi.h"
#include <string.h>
#include <mutex>

namespace dsp_stage {

std::mutex &mutex();            // the single-call wrappers share one scratch arena

struct Stage {
    Dav1dHipContext *c;
    uint8_t *base;
    size_t used, cap;
    bool ok;
    explicit Stage(size_t bytes) : c(dav1d_hip_default_context()), base(nullptr), used(0), cap(bytes), ok(false) {
        void *p = nullptr;
        if (c && !dav1d_hip_scratch(c, bytes, &p)) { base = (uint8_t *) p; ok = true; }
        if (!ok) abort();       // a void DSP entry cannot report failure; there is no CPU fallback to take instead
    }
    void *take(size_t bytes) {
        used = (used + 255) & ~(size_t) 255;
        void *p = base + used;
        used += bytes;
        if (used > cap) abort();
        return p;
    }
    void *take_zero(size_t bytes) {
        void *p = take(bytes);
        hipMemsetAsync(p, 0, bytes, c->stream);
        return p;
    }
    void sync() { hipStreamSynchronize(c->stream); }
};

inline Dav1dHipPicture empty_pic(int bpc, int layout) {
    Dav1dHipPicture p;
    memset(&p, 0, sizeof(p));
    p.bpc = bpc;
    p.layout = layout;
    return p;
}
// adds a zero-filled plane of w x h pixels living in the scratch arena
inline void add_plane(Stage &s, Dav1dHipPicture &p, int pl, int w, int h) {
    const int bps = p.bpc > 8 ? 2 : 1;
    p.p[pl].stride = (ptrdiff_t) ((w * bps + 15) & ~15);
    p.p[pl].w = w;
    p.p[pl].h = h;
    p.p[pl].data = s.take_zero((size_t) p.p[pl].stride * h);
}
inline Dav1dHipPicture scratch_pic(Stage &s, int w, int h, int bpc) {
    Dav1dHipPicture p = empty_pic(bpc, DAV1D_HIP_LAYOUT_I400);
    add_plane(s, p, 0, w, h);
    return p;
}
template <typename pixel> inline pixel *px(const Dav1dHipPicture &p, int pl, int x, int y) {
    return (pixel *) ((uint8_t *) p.p[pl].data + (ptrdiff_t) y * p.p[pl].stride) + x;
}
inline void up2d(Dav1dHipContext *c, void *dev, ptrdiff_t dstride, const void *host, ptrdiff_t hstride, size_t row_bytes, int rows) {
    if (row_bytes && rows > 0) hipMemcpy2DAsync(dev, dstride, host, hstride, row_bytes, rows, hipMemcpyHostToDevice, c->stream);
}
inline void down2d(Dav1dHipContext *c, void *host, ptrdiff_t hstride, const void *dev, ptrdiff_t dstride, size_t row_bytes, int rows) {
    if (row_bytes && rows > 0) hipMemcpy2DAsync(host, hstride, dev, dstride, row_bytes, rows, hipMemcpyDeviceToHost, c->stream);
}
inline int bpc_of(int bitdepth_max) { return bitdepth_max == 0xff ? 8 : bitdepth_max == 0x3ff ? 10 : 12; }

} // namespace dsp_stage

// filled by dsp_table_post.hip: fg, ipred, lf, cdef, lr
void dav1d_hip_dsp_fill_post_8(Dav1dHipDSPContext8 *c);
void dav1d_hip_dsp_fill_post_16(Dav1dHipDSPContext16 *c);

// Batched compound combination: avg / w_avg / mask / w_mask for gfx950.
//
// Contract per task = reference avg_c, w_avg_c, mask_c, w_mask_c
// (src/mc_tmpl.c:628-681, 724-794).  Pure streaming: two int16 predictions in, pixels
// out (+ the w_mask segmentation mask at chroma resolution).  One lane handles a 2x2
// pixel quad so that the 4:2:0 mask sample falls out of a single lane.
#include "common.h"
#include "capi.h"
#include "av1_tables.h"

namespace {

template <typename pixel>
__global__ __launch_bounds__(64) void comp_kernel(const DevPlanes dst, const Dav1dHipCompTask *__restrict__ tasks, const int n,
                                                  const int16_t *__restrict__ prep, uint8_t *__restrict__ mask,
                                                  const int bitdepth_max)
{
    constexpr bool HBD = sizeof(pixel) == 2;
    const int ti = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x);
    if (ti >= n) return;
    const Dav1dHipCompTask t = tasks[ti];
    const int bitdepth = 32 - __clz(bitdepth_max);
    const int ib = HBD ? 14 - bitdepth : 4;
    const int bias = HBD ? 8192 : 0;
    const int w = t.w, h = t.h, qw = w >> 1;
    pixel *const d0 = reinterpret_cast<pixel *>(dst.data[t.plane]) + t.dst_off;
    const int stride = dst.stride[t.plane];
    const int16_t *const t1 = prep + t.tmp1_off, *const t2 = prep + t.tmp2_off;

    if (t.kind >= DAV1D_HIP_COMP_BLEND) {
        // blend_c / blend_v_c / blend_h_c, src/mc_tmpl.c:682-722
        const pixel *const tmp = reinterpret_cast<const pixel *>(prep) + t.tmp1_off;
        const int ww = t.kind == DAV1D_HIP_COMP_BLEND_V ? (w * 3) >> 2 : w, hh = t.kind == DAV1D_HIP_COMP_BLEND_H ? (h * 3) >> 2 : h;
        for (int i = threadIdx.x; i < ww * hh; i += 64) {
            const int y = i / ww, x = i - y * ww;
            const int m = t.kind == DAV1D_HIP_COMP_BLEND ? mask[t.mask_off + y * w + x]
                        : t.kind == DAV1D_HIP_COMP_BLEND_V ? av1_obmc_masks[w + x] : av1_obmc_masks[h + y];
            const int a = d0[y * stride + x], b = tmp[y * w + x];
            d0[y * stride + x] = (pixel) ((a * (64 - m) + b * m + 32) >> 6);
        }
        return;
    }

    for (int i = threadIdx.x; i < qw * (h >> 1); i += 64) {
        const int qy = i / qw, qx = i - qy * qw;
        int m4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = 2 * qy + (k >> 1), x = 2 * qx + (k & 1);
            const int a = t1[y * w + x], b = t2[y * w + x];
            int v;
            if (t.kind == DAV1D_HIP_COMP_AVG) {
                v = (a + b + (1 << ib) + bias * 2) >> (ib + 1);
            } else if (t.kind == DAV1D_HIP_COMP_WAVG) {
                v = (a * t.arg + b * (16 - t.arg) + (8 << ib) + bias * 16) >> (ib + 4);
            } else if (t.kind == DAV1D_HIP_COMP_MASK) {
                const int m = mask[t.mask_off + y * w + x];
                v = (a * m + b * (64 - m) + (32 << ib) + bias * 64) >> (ib + 6);
            } else {
                const int mask_sh = bitdepth + ib - 4;
                const int diff = a - b;
                const int ad = diff < 0 ? -diff : diff;
                const int m = dv::imin(38 + ((ad + (1 << (mask_sh - 5))) >> mask_sh), 64);
                m4[k] = m;
                v = (diff * m + b * 64 + (32 << ib) + bias * 64) >> (ib + 6);
            }
            d0[y * stride + x] = (pixel) dv::iclip(v, 0, bitdepth_max);
        }
        if (t.kind == DAV1D_HIP_COMP_WMASK) {
            uint8_t *mo = mask + t.mask_off;
            const int sign = t.arg;
            if (t.ss == 0) {            // 4:4:4: full resolution
                mo[(2 * qy) * w + 2 * qx] = m4[0];
                mo[(2 * qy) * w + 2 * qx + 1] = m4[1];
                mo[(2 * qy + 1) * w + 2 * qx] = m4[2];
                mo[(2 * qy + 1) * w + 2 * qx + 1] = m4[3];
            } else if (t.ss == 1) {     // 4:2:2: horizontal pairs
                mo[(2 * qy) * qw + qx] = (m4[0] + m4[1] + 1 - sign) >> 1;
                mo[(2 * qy + 1) * qw + qx] = (m4[2] + m4[3] + 1 - sign) >> 1;
            } else {                    // 4:2:0: 2x2
                mo[qy * qw + qx] = (m4[0] + m4[1] + m4[2] + m4[3] + 2 - sign) >> 2;
            }
        }
    }
}

} // namespace

extern "C" int dav1d_hip_launch_comp(const DevPlanes *dst, int bpc, const Dav1dHipCompTask *tasks, int n,
                                     const int16_t *prep, uint8_t *mask, void *stream)
{
    if (n <= 0) return 0;
    const int bitdepth_max = (1 << bpc) - 1;
    if (bpc == 8)
        hipLaunchKernelGGL((comp_kernel<uint8_t>), dim3(n), dim3(64), 0, (hipStream_t) stream,
                           *dst, tasks, n, prep, mask, bitdepth_max);
    else
        hipLaunchKernelGGL((comp_kernel<uint16_t>), dim3(n), dim3(64), 0, (hipStream_t) stream,
                           *dst, tasks, n, prep, mask, bitdepth_max);
    return hip_rc(hipGetLastError());
}

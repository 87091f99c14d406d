// The intra wavefront of a frame as ONE launch (gfx950).
//
// The reference reconstructs an intra transform block as prepare_intra_edges + intra_pred, then itxfm_add on the same
// pixels (src/recon_tmpl.c:1207-1360), block after block in decode order; what a block reads (left / top / top-right /
// bottom-left edge pixels, CfL's luma) comes from blocks reconstructed before it.  The lister turns that order into steps:
// a block's step is one more than the largest step among the cells its edges read.  Round 1 ran a step as up to three
// dependent launches (4x4 / 8x8 pairs, the other predictions, the other residuals): 66 launches for the 46 steps of the
// benchmark's inter frame, several thousand for a key frame, each paying a launch-to-finish latency of 5 - 20 us.
//
// Here the whole pass is a list of UNITS sorted by step, one unit = prediction + residual of one transform block (or only
// one of the two), worked off by a grid of one-wave workgroups inside a single launch:
//   * the units are dealt to the waves round-robin (wave w works on units w, w + G, w + 2 G, ...; every wave of the launch has to
//     be resident, which the launch function guarantees by sizing the grid from the occupancy query), and a unit starts once
//     the group before it — the units of the previous step — has finished: every group has eight completion counters in
//     cache lines of their own, a finishing wave bumps one of them, a waiting wave sums the eight.  One shared "done" word and
//     one shared ticket word, as the first version had, cap the launch at what ONE address takes in atomics (about 25 ns
//     each: 37 ms for the 615 K units of an 8K key frame, whatever the number of waves);
//   * the prediction goes to an LDS tile, the residual is added from there (ipred_body.h + itx_body.h, the bodies of the
//     stand-alone kernels), the pixels leave with agent-scope stores, and once those are acknowledged the wave bumps its
//     counter.  Edge pixels are read with agent-scope loads: per-XCD L2s and per-CU L1s are not coherent for plain accesses
//     within a launch (MI355X_MICROARCH.md, "inter-workgroup visibility").
// A step boundary then costs a counter hand-off (about a microsecond) instead of a kernel boundary plus the fill and drain
// of a small grid.
#include "ipred_body.h"
#include "itx_body.h"
#include "capi.h"
#include <type_traits>

namespace {

constexpr int flow_itx_lds_of(int tx) {
    return (64 / cmax(cmin(tx_h(tx), 32), tx_w(tx))) * cmin(tx_h(tx), 32) * (tx_w(tx) + 1);
}
constexpr int flow_itx_lds_max(int tx = 0) { return tx == 19 ? 0 : cmax(flow_itx_lds_of(tx), flow_itx_lds_max(tx + 1)); }

enum { FLOW_SPIN_LIMIT = 1 << 20 };       // polls before a wave gives up and raises the error word (a bug, never a normal run)

template <typename pixel, typename coef>
__global__ __launch_bounds__(64, 4) void intra_flow_kernel(const DevPlanes dst, const IntraUnit *__restrict__ units, const int n_units,
                                                           uint8_t *aux, coef *__restrict__ cf, const int layout, const int bitdepth_max,
                                                           uint32_t *ctr /* word 0: error; from word 32: FLOW_SUB counters per group */,
                                                           const int mode)
{
    __shared__ int16_t e1[ESZ], e2[ESZ];
    __shared__ int16_t blk[32 * 32];
    // the predicted tile and the transform's slabs share their LDS: the transform body has the tile in registers before it stores
    // its first slab chunk (itx_body.h, PRED_LDS)
    constexpr int TILE_B = 64 * 64 * (int) sizeof(pixel), ITX_B = flow_itx_lds_max() * 4;
    __shared__ uint4 smem[(cmax(TILE_B, ITX_B) + 15) / 16];
    pixel *const tile = reinterpret_cast<pixel *>(smem);
    int *const smem_itx = reinterpret_cast<int *>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x, n_waves = gridDim.x;
    uint32_t *const cnt = ctr + 32;

    for (int ui = wave; ui < n_units; ui += n_waves) {
        const IntraUnit *const up = units + ui;
        const IntraUnit u = *up;
        // the wave's next record and the first lines of this unit's coefficients set off now; the values are claimed after the
        // prediction, so the trips to memory run under the wait for the step and the prediction
        const bool has_pred = u.has & 1, has_tx = u.has & 2;
        const int keep0 = dv::fetch_begin(units + (ui + n_waves < n_units ? ui + n_waves : ui));
        const int nb = has_tx ? ((int) u.t.rsv[0] | (int) u.t.rsv[1] << 8) * (int) sizeof(coef) : 0;
        const int keep1 = dv::fetch_begin(has_tx ? reinterpret_cast<const char *>(cf + u.t.cf_off) + (lane * 64 < nb ? lane * 64 : 0)
                                                 : reinterpret_cast<const char *>(units + ui));
        if (u.prev_n) {
            // the group before this one is through when its counters add up to its size (mode bit 1: fences as well; A/B aid)
            const uint32_t *const prev = cnt + (size_t) (u.grp - 1) * (FLOW_SUB * FLOW_SUB_STRIDE);
            int spins = 0;
            for (;;) {
                unsigned v = lane < FLOW_SUB ? dv::ld_coherent(prev + lane * FLOW_SUB_STRIDE) : 0u;
#pragma unroll
                for (int m = 1; m < FLOW_SUB; m <<= 1) v += (unsigned) __shfl_xor((int) v, m);
                v = (unsigned) __builtin_amdgcn_readfirstlane((int) v);
                if (v >= u.prev_n) break;
                if (2 * v < u.prev_n) dv::nap_long();           // far from done: leave the lines alone for a while
                dv::nap();
                // never met in a run that works: the wave raises the error word and LEAVES — reconstructing a unit whose edges are not there
                // yet would put wrong pixels into the picture; without its completions the other waves give up the same way and the
                // frame comes back -EIO
                if (++spins > FLOW_SPIN_LIMIT) { if (lane == 0) atomicAdd(&ctr[0], 1u); return; }
            }
            if (mode & 2) dv::fence_acquire_agent();
        }
        const int plane = has_pred ? u.p.plane : u.t.plane;
        const uint32_t dst_off = has_pred ? u.p.dst_off : u.t.dst_off;
        const int w = has_pred ? u.p.tw * 4 : tx_w(u.t.tx), h = has_pred ? u.p.th * 4 : tx_h(u.t.tx);
        pixel *const d = reinterpret_cast<pixel *>(dst.data[plane]) + dst_off;
        const int stride = dst.stride[plane];
        if (has_pred) {
            ipred_body<pixel, true>(dst, u.p, 0, false, aux, layout, bitdepth_max, e1, e2, blk, tile, w);
        } else {
            // a residual on its own (the blocks of a palette block, ...): the pixels it is added to come from the picture
            for (int i = lane; i < w * h; i += 64) tile[i] = dv::ld_coherent(d + (i / w) * stride + (i % w));
        }
        dv::fetch_end(keep0);
        dv::fetch_end(keep1);
        dv::wave_sync();
        if (has_tx) {
#define CASE(T) case T: itx_body<T, pixel, coef, true, true>(dst, &up->t, 1, cf, bitdepth_max, 0, smem_itx, tile); break;
            switch (u.t.tx) {
                CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9)
                CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18)
            }
#undef CASE
        }
        dv::wave_sync();
        // the reconstructed tile leaves the LDS four pixels per store (blocks are at least four pixels wide and four-pixel aligned)
        {
            typedef typename std::conditional<sizeof(pixel) == 2, uint64_t, uint32_t>::type quad;
            const int wq = w >> 2;
            for (int i = lane; i < wq * h; i += 64) {
                const int y = i / wq, x = (i - y * wq) * 4;
                dv::st_coherent(reinterpret_cast<quad *>(d + y * stride + x), *reinterpret_cast<const quad *>(tile + y * w + x));
            }
        }
        dv::stores_done();
        if (mode & 2) dv::fence_release_agent();
        dv::wave_sync();                 // the LDS is free for the next unit
        if (lane == 0) atomicAdd(cnt + ((size_t) u.grp * FLOW_SUB + (wave & (FLOW_SUB - 1))) * FLOW_SUB_STRIDE, 1u);
        // a convergence point after the lane-0 region (the first version of this kernel, with a lane-0 ticket draw at the top of
        // the loop, was threaded by the compiler from one such region into the next and never ended)
        dv::wave_sync();
    }
}

#ifndef DAV1D_HIP_EMU
// per device: the grid the kernel may have (8 / 16 bpc) and the end of the most recent dataflow launch
enum { FLOW_MAX_DEVICES = 64 };
struct FlowDevice { std::mutex mtx; int cap[2] = { 0, 0 }; hipEvent_t done = nullptr; bool any = false; };
FlowDevice flow_devices[FLOW_MAX_DEVICES];
#endif

} // namespace

extern "C" int dav1d_hip_launch_intra_flow(const DevPlanes *dst, int bpc, int layout, const IntraUnit *units, int n_units, uint8_t *aux,
                                           void *coef, uint32_t *ctr, int n_waves, int mode, void *stream)
{
    if (n_units <= 0) return 0;
    const int bitdepth_max = (1 << bpc) - 1;
    // every wave of the grid has to be resident (units are dealt round-robin, a unit waits for units other waves hold): never
    // more waves than the device runs at once for this kernel
    int grid = n_units < n_waves ? n_units : n_waves;
#ifdef DAV1D_HIP_EMU
    grid = 1;                               // the emulator runs workgroups one after the other
#else
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FLOW_MAX_DEVICES) return -EIO;
    FlowDevice &fd = flow_devices[dev];
    std::lock_guard<std::mutex> lk(fd.mtx);
    if (!fd.cap[bpc != 8]) {                // asked once per device and kernel, not on every launch
        int per_cu = 0, cus = 0;
        const void *fn = bpc == 8 ? (const void *) intra_flow_kernel<uint8_t, int16_t> : (const void *) intra_flow_kernel<uint16_t, int32_t>;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, 0) != hipSuccess || per_cu < 1 || cus < 1) return -EIO;
        // one block per CU less than the query says (MI355X_MICROARCH.md: the API over-reports by one near SGPR limits)
        fd.cap[bpc != 8] = cus * (per_cu > 1 ? per_cu - 1 : 1);
    }
    if (grid > fd.cap[bpc != 8]) grid = fd.cap[bpc != 8];
    // The occupancy figure is that of an EMPTY device.  Other kernels only delay the residency of this grid (they end without
    // waiting for it), but a second dataflow launch — another context's key frame, frames in flight — could hold the slots this one
    // needs while waiting for slots this one holds.  So the dataflow launches of a device form a chain: each waits for the one
    // before it, whatever stream and context it came from (a device-side wait, the host does not block).
    if (!fd.done && hipEventCreateWithFlags(&fd.done, hipEventDisableTiming) != hipSuccess) return -EIO;
    if (fd.any && hipStreamWaitEvent((hipStream_t) stream, fd.done, 0) != hipSuccess) return -EIO;
#endif
    if (bpc == 8)
        hipLaunchKernelGGL((intra_flow_kernel<uint8_t, int16_t>), dim3(grid), dim3(64), 0, (hipStream_t) stream, *dst, units, n_units, aux,
                           (int16_t *) coef, layout, bitdepth_max, ctr, mode);
    else
        hipLaunchKernelGGL((intra_flow_kernel<uint16_t, int32_t>), dim3(grid), dim3(64), 0, (hipStream_t) stream, *dst, units, n_units, aux,
                           (int32_t *) coef, layout, bitdepth_max, ctr, mode);
    const int rc = hip_rc(hipGetLastError());
#ifndef DAV1D_HIP_EMU
    if (!rc) {
        if (hipEventRecord(fd.done, (hipStream_t) stream) != hipSuccess) return -EIO;
        fd.any = true;
    }
#endif
    return rc;
}

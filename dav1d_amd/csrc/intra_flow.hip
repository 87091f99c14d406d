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
//   * a wave draws the next unit with an atomic ticket, loads its task records, and waits until `done` (the count of
//     finished units) has reached the unit's `need` = the number of units in earlier steps.  Tickets are handed out in
//     list order, so a waiting wave only ever waits for units held by waves that are already running: no residency
//     requirement, no deadlock, and a grid of any size works;
//   * the prediction goes to an LDS tile, the residual is added from there (ipred_body.h + itx_body.h, the bodies of the
//     stand-alone kernels), the pixels leave with agent-scope stores, and once those are acknowledged the wave bumps
//     `done`.  Edge pixels are read with agent-scope loads: per-XCD L2s and per-CU L1s are not coherent for plain accesses
//     within a launch (MI355X_MICROARCH.md, "inter-workgroup visibility").
// A step boundary then costs a counter hand-off (about a microsecond) instead of a kernel boundary plus the fill and drain
// of a small grid.
#include "ipred_body.h"
#include "itx_body.h"
#include "capi.h"

namespace {

constexpr int flow_itx_lds_of(int tx) {
    return (64 / cmax(cmin(tx_h(tx), 32), tx_w(tx))) * cmin(tx_h(tx), 32) * (tx_w(tx) + 1);
}
constexpr int flow_itx_lds_max(int tx = 0) { return tx == 19 ? 0 : cmax(flow_itx_lds_of(tx), flow_itx_lds_max(tx + 1)); }

enum { FLOW_TICKET = 0, FLOW_DONE = 32, FLOW_ERROR = 64 };   // == capi.hip
enum { FLOW_SPIN_LIMIT = 1 << 20 };       // polls before a wave gives up and raises ctr[2] (a bug, never a normal run)

template <typename pixel, typename coef>
__global__ __launch_bounds__(64, 4) void intra_flow_kernel(const DevPlanes dst, const IntraUnit *__restrict__ units, const int n_units,
                                                           uint8_t *aux, coef *__restrict__ cf, const int layout, const int bitdepth_max,
                                                           uint32_t *ctr /* words 0 / 32 / 64 (lines of their own): next ticket, done, error */,
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
    const int lane = threadIdx.x;

    // The next ticket is drawn when the unit's last stores are on their way, so that the round trip of the draw hides behind
    // the wait for their acknowledgement; drawing earlier would park a unit behind a busy wave while other waves idle
    // (measured on an 8K key frame: 94 ms against 77).
    auto draw = [&]() {
        int t = 0;
        if (lane == 0) t = (int) atomicAdd(&ctr[FLOW_TICKET], 1u);
        return t;                       // lane 0's value; readfirstlane where it is used
    };
    int ticket = __builtin_amdgcn_readfirstlane(draw());
    while (ticket < n_units) {
        const IntraUnit *const up = units + ticket;
        const IntraUnit u = *up;
        const bool has_pred = u.has & 1, has_tx = u.has & 2;
        if (has_tx) {
            // ... and the first lines of this unit's coefficients
            const int nb = ((int) u.t.rsv[0] | (int) u.t.rsv[1] << 8) * (int) sizeof(coef);
            if (lane * 64 < nb) dv::touch(reinterpret_cast<const char *>(cf + u.t.cf_off) + lane * 64);
        }
        if (u.need) {
            // mode bit 0: poll through the atomic unit; bit 1: agent-scope fences around the hand-off as well
            int spins = 0;
            for (;;) {
                unsigned v = 0;
                if (mode & 1) v = (unsigned) __builtin_amdgcn_readfirstlane((int) dv::ld_rmw(&ctr[FLOW_DONE]));
                else v = dv::ld_coherent(&ctr[FLOW_DONE]);
                if (v >= u.need) break;
                // a wave that is steps ahead of the front leaves the counter's line alone for a while: thousands of waves polling
                // one word slow every hand-off down (measured: 52 us per step on an 8K key frame with 2048 pollers, 22 with 256)
                const unsigned gap = u.need - v;
                if (gap > 1024) dv::nap_long();
                if (gap > 128) dv::nap_long();
                dv::nap();
                if (++spins > FLOW_SPIN_LIMIT) { if (lane == 0) atomicAdd(&ctr[FLOW_ERROR], 1u); break; }
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
        dv::wave_sync();
        if (has_tx) {
#define CASE(T) case T: itx_body<T, pixel, coef, true, true>(dst, &up->t, 1, cf, bitdepth_max, 0, smem_itx, tile); break;
            switch (u.t.tx) {
                CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9)
                CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18)
            }
#undef CASE
        } else {
            for (int i = lane; i < w * h; i += 64) dv::st_coherent(d + (i / w) * stride + (i % w), tile[i]);
        }
        const int next_raw = draw();
        dv::stores_done();
        if (mode & 2) dv::fence_release_agent();
        const int next_ticket = __builtin_amdgcn_readfirstlane(next_raw);
        dv::touch(units + (next_ticket < n_units ? next_ticket : n_units));            // the next record, on its way
        dv::wave_sync();                 // the LDS is free for the next unit
        if (lane == 0) atomicAdd(&ctr[FLOW_DONE], 1u);
        // a convergence point between this lane-0 region and the next one (the ticket draw at the top of the loop): without it
        // the compiler threads lane 0 from here straight into that region and the wave falls apart (observed: the kernel of the
        // first version never ended)
        dv::wave_sync();
        ticket = next_ticket;
    }
}

} // namespace

extern "C" int dav1d_hip_launch_intra_flow(const DevPlanes *dst, int bpc, int layout, const IntraUnit *units, int n_units, uint8_t *aux,
                                           void *coef, uint32_t *ctr, int max_groups, int mode, void *stream)
{
    if (n_units <= 0) return 0;
    const int bitdepth_max = (1 << bpc) - 1;
    const int grid = n_units < max_groups ? n_units : max_groups;
    if (bpc == 8)
        hipLaunchKernelGGL((intra_flow_kernel<uint8_t, int16_t>), dim3(grid), dim3(64), 0, (hipStream_t) stream, *dst, units, n_units, aux,
                           (int16_t *) coef, layout, bitdepth_max, ctr, mode);
    else
        hipLaunchKernelGGL((intra_flow_kernel<uint16_t, int32_t>), dim3(grid), dim3(64), 0, (hipStream_t) stream, *dst, units, n_units, aux,
                           (int32_t *) coef, layout, bitdepth_max, ctr, mode);
    return hip_rc(hipGetLastError());
}

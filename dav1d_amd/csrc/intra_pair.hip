// Intra prediction and residual of a small block in one wave (gfx950).
//
// The reference reconstructs an intra transform block as prepare_intra_edges + intra_pred into the picture, then
// itxfm_add on the same pixels (src/recon_tmpl.c:1207-1360).  The wavefront steps of an intra frame are chains of small
// dependent launches, each paying its own launch-to-finish latency; for the 4x4 and 8x8 blocks — all there is in the
// second half of the wavefront of a superblock — the two launches of a step become one: the wave predicts the block into
// LDS (ipred_body.h, the body of ipred.hip) and adds the residual from there (itx_body.h, the body of itx.hip).
#include "ipred_body.h"
#include "itx_body.h"
#include "capi.h"

namespace {

// pairs[i] = (prediction task, transform task) of block i; both describe the same rectangle, tx = TX_4X4 or TX_8X8
template <typename pixel, typename coef>
__global__ __launch_bounds__(64) void intra_pair_kernel(const DevPlanes dst, const Dav1dHipIpredTask *__restrict__ preds,
                                                        const Dav1dHipItxTask *__restrict__ txs, const int n, uint8_t *aux,
                                                        coef *__restrict__ cf, const int layout, const int bitdepth_max)
{
    __shared__ int16_t e1[ESZ], e2[ESZ];
    __shared__ int16_t blk[32 * 32];
    __shared__ __attribute__((aligned(16))) int smem_itx[cmax(itx_lds_ints<0>(), itx_lds_ints<1>())];
    __shared__ __attribute__((aligned(16))) pixel pred[8 * 8];
    const int bi = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x);
    if (bi >= n) return;
    const int bs = __builtin_amdgcn_readfirstlane(bi);
    const Dav1dHipIpredTask t = preds[bs];
    // the residual's record and the first lines of its coefficients set off now: their trip to memory overlaps the prediction's
    // (record -> edge pixels -> predictor) instead of following it; a step of the wavefront is this chain of latencies
    const Dav1dHipItxTask tt = txs[bs];
    const int nb = ((int) tt.rsv[0] | (int) tt.rsv[1] << 8) * (int) sizeof(coef);
    const int lane64 = (int) threadIdx.x * 64 < nb ? (int) threadIdx.x * 64 : 0;
    const int keep = dv::fetch_begin(reinterpret_cast<const char *>(cf + tt.cf_off) + lane64);
    const int w = t.tw * 4;
    ipred_body<pixel>(dst, t, 0, false, aux, layout, bitdepth_max, e1, e2, blk, pred, w);
    dv::fetch_end(keep);
    dv::wave_sync();
    // the transform body packs several blocks into a wave; here it gets a list of one: lanes past the first block idle
    if (w == 4) itx_body<0, pixel, coef, true>(dst, txs + bs, 1, cf, bitdepth_max, 0, smem_itx, pred);
    else        itx_body<1, pixel, coef, true>(dst, txs + bs, 1, cf, bitdepth_max, 0, smem_itx, pred);
}

// The first two launches of a wavefront step in one: workgroups [0, n_items) predict the blocks that are not paired (what
// ipred_kernel does, ipred.hip), workgroups [n_items, n_items + n_pairs) run the 4x4 / 8x8 pairs.  The two sets are independent
// and both launches are latency chains of a few microseconds: side by side they cost the longer of the two, not the sum.
template <typename pixel, typename coef>
__global__ __launch_bounds__(64) void intra_step_kernel(const DevPlanes dst, const Dav1dHipIpredTask *__restrict__ tasks, const int n, const int n_big,
                                                        const int n_items, const Dav1dHipIpredTask *__restrict__ preds,
                                                        const Dav1dHipItxTask *__restrict__ txs, const int n_pairs, uint8_t *aux, void *tmp,
                                                        coef *__restrict__ cf, const int layout, const int bitdepth_max)
{
    __shared__ int16_t e1[ESZ], e2[ESZ];
    __shared__ int16_t blk[32 * 32];
    __shared__ __attribute__((aligned(16))) int smem_itx[cmax(itx_lds_ints<0>(), itx_lds_ints<1>())];
    __shared__ __attribute__((aligned(16))) pixel pred[8 * 8];
    if ((int) blockIdx.x < n_items) {
        // == ipred_kernel: the first n_big tasks get IPRED_PARTS workgroups each
        const int b = blockIdx.x;
        const bool many = b < n_big * IPRED_PARTS;
        const int part = many ? b % IPRED_PARTS : 0;
        const int ti = many ? b / IPRED_PARTS : b - n_big * (IPRED_PARTS - 1);
        if (ti >= n) return;
        const Dav1dHipIpredTask t = tasks[__builtin_amdgcn_readfirstlane(ti)];
        const bool to_tmp = t.kind == DAV1D_HIP_IPRED_PRED_TMP;
        pixel *const d = to_tmp ? reinterpret_cast<pixel *>(tmp) + t.aux_off : reinterpret_cast<pixel *>(dst.data[t.plane]) + t.dst_off;
        ipred_body<pixel>(dst, t, part, many, aux, layout, bitdepth_max, e1, e2, blk, d, to_tmp ? t.tw * 4 : dst.stride[t.plane]);
        return;
    }
    const int bs = __builtin_amdgcn_readfirstlane((int) blockIdx.x - n_items);
    if (bs >= n_pairs) return;
    const Dav1dHipIpredTask t = preds[bs];
    const Dav1dHipItxTask tt = txs[bs];
    const int nb = ((int) tt.rsv[0] | (int) tt.rsv[1] << 8) * (int) sizeof(coef);
    const int lane64 = (int) threadIdx.x * 64 < nb ? (int) threadIdx.x * 64 : 0;
    const int keep = dv::fetch_begin(reinterpret_cast<const char *>(cf + tt.cf_off) + lane64);
    const int w = t.tw * 4;
    ipred_body<pixel>(dst, t, 0, false, aux, layout, bitdepth_max, e1, e2, blk, pred, w);
    dv::fetch_end(keep);
    dv::wave_sync();
    if (w == 4) itx_body<0, pixel, coef, true>(dst, txs + bs, 1, cf, bitdepth_max, 0, smem_itx, pred);
    else        itx_body<1, pixel, coef, true>(dst, txs + bs, 1, cf, bitdepth_max, 0, smem_itx, pred);
}

} // namespace

extern "C" int dav1d_hip_launch_intra_step(const DevPlanes *dst, int bpc, int layout, const Dav1dHipIpredTask *tasks, int n, int n_big,
                                           const Dav1dHipIpredTask *preds, const Dav1dHipItxTask *txs, int n_pairs, uint8_t *aux, void *tmp,
                                           void *coef, void *stream)
{
    if (n < 0 || n_pairs < 0 || n_big < 0 || n_big > n) return -22;
    const int n_items = n + n_big * (IPRED_PARTS - 1), grid = n_items + n_pairs;
    if (grid <= 0) return 0;
    const int bitdepth_max = (1 << bpc) - 1;
    if (bpc == 8)
        hipLaunchKernelGGL((intra_step_kernel<uint8_t, int16_t>), dim3(grid), dim3(64), 0, (hipStream_t) stream, *dst, tasks, n, n_big, n_items, preds, txs,
                           n_pairs, aux, tmp, (int16_t *) coef, layout, bitdepth_max);
    else
        hipLaunchKernelGGL((intra_step_kernel<uint16_t, int32_t>), dim3(grid), dim3(64), 0, (hipStream_t) stream, *dst, tasks, n, n_big, n_items, preds, txs,
                           n_pairs, aux, tmp, (int32_t *) coef, layout, bitdepth_max);
    return hip_rc(hipGetLastError());
}

extern "C" int dav1d_hip_launch_intra_pairs(const DevPlanes *dst, int bpc, int layout, const Dav1dHipIpredTask *preds,
                                            const Dav1dHipItxTask *txs, int n, uint8_t *aux, void *coef, void *stream)
{
    if (n <= 0) return 0;
    const int bitdepth_max = (1 << bpc) - 1;
    if (bpc == 8)
        hipLaunchKernelGGL((intra_pair_kernel<uint8_t, int16_t>), dim3(n), dim3(64), 0, (hipStream_t) stream, *dst, preds, txs, n, aux,
                           (int16_t *) coef, layout, bitdepth_max);
    else
        hipLaunchKernelGGL((intra_pair_kernel<uint16_t, int32_t>), dim3(n), dim3(64), 0, (hipStream_t) stream, *dst, preds, txs, n, aux,
                           (int32_t *) coef, layout, bitdepth_max);
    return hip_rc(hipGetLastError());
}

// Batched intra prediction for gfx950.
//
// Contract per task = what the reference driver does for one transform block
// (src/recon_tmpl.c:1239-1290, chroma :1420-1520): dav1d_prepare_intra_edges
// (src/ipred_prepare_tmpl.c:75-204) gathers / extends the edge pixels around the block and
// maps the bitstream mode to a DSP entry, then dsp->ipred.intra_pred[m] (src/ipred_tmpl.c:93-655)
// predicts the block; CfL tasks run cfl_ac + cfl_pred (:657-715, :71-84), palette tasks pal_pred
// (:717-730).  A batch holds tasks that do not depend on each other (the caller orders batches
// along the decode wavefront); reconstruction happens before the in-loop filters, so the row above
// a superblock row is simply read from the picture.
//
// Mapping: one wave per task, the task record in SGPRs.  The edge array (topleft[-2h .. 2w]) is
// assembled in LDS, directional modes materialise their filtered / upsampled edge in a second LDS
// array, then every lane produces pixels i = lane, lane + 64, ... of the block.  Filter-intra runs
// its 4x2 sub-blocks along anti-diagonals through an LDS copy of the block.
#include "ipred_body.h"

namespace {

template <typename pixel>
__global__ __launch_bounds__(64) void ipred_kernel(const DevPlanes dst, const Dav1dHipIpredTask *__restrict__ tasks, const int n,
                                                   const int n_big, uint8_t *aux, void *tmp, const int layout, const int bitdepth_max)
{
    __shared__ int16_t e1[ESZ], e2[ESZ];
    __shared__ int16_t blk[32 * 32];

    // The first n_big tasks of the batch (the host puts them there: blocks of 1024 pixels or more without a serial
    // predictor) get IPRED_PARTS workgroups each — every one prepares the edge for itself and writes its share of the
    // rows; a lone wave walking 4096 pixels is what the wavefront steps of an intra frame would wait for.  The rest: one
    // workgroup per block.
    const int b = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x);
    const bool many = b < n_big * IPRED_PARTS;
    const int part = many ? b % IPRED_PARTS : 0;
    const int ti = many ? b / IPRED_PARTS : b - n_big * (IPRED_PARTS - 1);
    if (ti >= n) return;
    const Dav1dHipIpredTask t = tasks[__builtin_amdgcn_readfirstlane(ti)];
    // PRED_TMP (the intra half of an inter-intra block): same edges, the prediction goes to the scratch arena, row stride = width
    const bool to_tmp = t.kind == DAV1D_HIP_IPRED_PRED_TMP;
    pixel *const d = to_tmp ? reinterpret_cast<pixel *>(tmp) + t.aux_off : reinterpret_cast<pixel *>(dst.data[t.plane]) + t.dst_off;
    ipred_body<pixel>(dst, t, part, many, aux, layout, bitdepth_max, e1, e2, blk, d, to_tmp ? t.tw * 4 : dst.stride[t.plane]);
}

} // namespace

extern "C" int dav1d_hip_launch_ipred(const DevPlanes *dst, int bpc, int layout, const Dav1dHipIpredTask *tasks, int n, int n_big,
                                      uint8_t *pal_idx, void *tmp, void *stream)
{
    if (n <= 0) return 0;
    if (n_big < 0 || n_big > n) return -22;
    const int grid = n + n_big * (IPRED_PARTS - 1);
    const int bitdepth_max = (1 << bpc) - 1;
    if (bpc == 8)
        hipLaunchKernelGGL((ipred_kernel<uint8_t>), dim3(grid), dim3(64), 0, (hipStream_t) stream, *dst, tasks, n, n_big, pal_idx, tmp, layout, bitdepth_max);
    else
        hipLaunchKernelGGL((ipred_kernel<uint16_t>), dim3(grid), dim3(64), 0, (hipStream_t) stream, *dst, tasks, n, n_big, pal_idx, tmp, layout, bitdepth_max);
    return hip_rc(hipGetLastError());
}

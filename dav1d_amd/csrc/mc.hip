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
// Mapping: the host cuts every prediction block into tiles of at most 64x16 (a strip of
// one block: same motion vector, same taps, long contiguous source rows) and bins them by
// tile shape (TW, TH), TW in {4..64}, TH in {4,8,16}.  A tile owns LPT = min(64, TW*TH/4)
// lanes (each lane produces 1, 2 or 4 output strips of 4 pixels), 64/LPT tiles share a wave
// (64x16: one tile, 4 strips per lane; 16x16: 1; 8x8: 4; 4x4: 16).
// Per tile and reference:
//   1. gather: the (TH+7) x (TW+8) window goes to LDS as int16, tile column 0 at window
//      column 4 so every 4-pixel strip is 8-byte aligned.  Interior windows are fetched
//      with 16-byte (8-pixel) loads at arbitrary 2-byte alignment, all issued before the
//      first is consumed; windows touching the picture edge fall back to per-pixel clamped
//      loads (== emu_edge).
//   2. horizontal pass: one work item = TWO rows x one 4-pixel strip: 2 x 3 ds_read_b64,
//      v_dot2 on packed pixel pairs (even outputs use the taps packed (f0,f1)(f2,f3)..,
//      odd outputs the taps packed (0,f0)(f1,f2)..(f7,0): no re-alignment of the data),
//      one ds_write_b128 of the ROW-PAIR-INTERLEAVED intermediate mid2[row/2][col] =
//      (row even, row odd).
//   3. vertical pass: one lane = one output row x one 4-pixel strip: 5 ds_read_b128 of
//      mid2, v_dot2 against the taps packed for the row's parity, round/clip, 8-byte store.
// "No filter" in a direction is the unit tap with zero shift, bilinear is the taps
// (16-m, m) with 4 instead of 6 bits of precision, so all variants share one code path.
#define DV_UNIT mc        // (names this unit's phase accessor in -DDV_PHASES variant builds, common.h)
#include "mc_body.h"

DV_PHASE_DEFINE(DV_UNIT)
namespace {

// one tile shape per launch: workgroup b of the XCD-chunked order handles tiles [b * G, b * G + G)
// TILED: the references are read through their tiled twins (refs.r[].data point there; mc_body.h)
// MC_WAVES waves to a workgroup, each with a group of tiles and LDS of its own (they never meet): a launch of an 8K frame's 4x4 tiles is
// 20,000 groups, and what bounds it is how fast workgroups are handed out, not how many fit
#ifndef MC_WAVES
#define MC_WAVES 1
#endif
template <int TW, int TH, typename pixel, bool TILED>
__global__ __launch_bounds__(64 * MC_WAVES) void mc_kernel(const DevPlanes dst, const RefSet refs, const McTile *__restrict__ tiles,
                                                const int n, int16_t *__restrict__ prep, const int bitdepth_max)
{
    constexpr int G = 64 / mc_cmin(64, TW * TH / 4);
    constexpr int SMEM_V = (mc_lds_bytes<TW, TH, TILED>() + 15) / 16;
    __shared__ uint4 smem[MC_WAVES][SMEM_V];
    const int wv = MC_WAVES == 1 ? 0 : __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    const int t0 = ((int) dv::xcd_chunk_id(blockIdx.x, gridDim.x) * MC_WAVES + wv) * G;
    if (t0 >= n) return;
    mc_body<TW, TH, pixel, false, TILED>(dst, refs, tiles, t0, dv::imin(G, n - t0), prep, bitdepth_max, smem[wv]);
}

// TILED references and the picture's own tiled twin written along with the raster planes (mc_body.h, TWIN)
template <int TW, int TH, typename pixel>
__global__ __launch_bounds__(64) void mc_twin_kernel(const DevPlanes dst, const RefSet refs, const McTile *__restrict__ tiles,
                                                     const int n, int16_t *__restrict__ prep, const int bitdepth_max, const DevPlanes twin)
{
    constexpr int G = 64 / mc_cmin(64, TW * TH / 4);
    __shared__ uint4 smem[(mc_lds_bytes<TW, TH, true>() + 15) / 16];
    const int t0 = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x) * G;
    if (t0 >= n) return;
    mc_body<TW, TH, pixel, false, true, true>(dst, refs, tiles, t0, dv::imin(G, n - t0), prep, bitdepth_max, smem, nullptr, 0, 0, 0, 0, twin);
}

// every tile shape in one launch: the tiles are ordered by where they read (all shapes interleaved, see
// mc_list_from_bins) and cut into wave-sized groups of one shape, so that the lines one shape pulls into an
// XCD's L2 are still there when its neighbours of other shapes need them
constexpr int mc_cmax(int a, int b) { return a > b ? a : b; }
constexpr int MC_LDS_BIG =      // shapes at least 16 wide (classes 6 .. 14): one tile per wave, small LDS / VGPR footprints
    mc_cmax(mc_cmax(mc_cmax(mc_lds_bytes<16, 4>(), mc_lds_bytes<16, 8>()), mc_cmax(mc_lds_bytes<16, 16>(), mc_lds_bytes<32, 4>())),
    mc_cmax(mc_cmax(mc_lds_bytes<32, 8>(), mc_lds_bytes<32, 16>()), mc_cmax(mc_cmax(mc_lds_bytes<64, 4>(), mc_lds_bytes<64, 8>()), mc_lds_bytes<64, 16>())));
constexpr int MC_LDS_MAX =
    mc_cmax(mc_cmax(mc_cmax(mc_lds_bytes<4, 4>(), mc_lds_bytes<4, 8>()), mc_cmax(mc_lds_bytes<4, 16>(), mc_lds_bytes<8, 4>())),
    mc_cmax(mc_cmax(mc_lds_bytes<8, 8>(), mc_lds_bytes<8, 16>()), MC_LDS_BIG));

// SMALL = false: the kernel only knows the shapes that are at least 16 wide
template <typename pixel, bool SMALL>
__global__ __launch_bounds__(64) void mc_all_kernel(const DevPlanes dst, const RefSet refs, const McTile *__restrict__ tiles,
                                                    const McGroup *__restrict__ groups, const int n_groups,
                                                    int16_t *__restrict__ prep, const int bitdepth_max)
{
    __shared__ uint4 smem[((SMALL ? MC_LDS_MAX : MC_LDS_BIG) + 15) / 16];
    const int gi = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x);
    if (gi >= n_groups) return;
    const McGroup g = groups[__builtin_amdgcn_readfirstlane(gi)];
    const int t0 = __builtin_amdgcn_readfirstlane((int) g.start), nt = __builtin_amdgcn_readfirstlane((int) g.n);
#define CASE(C, TW, TH) case C: mc_body<TW, TH, pixel>(dst, refs, tiles, t0, nt, prep, bitdepth_max, smem); break;
    const int cls = __builtin_amdgcn_readfirstlane((int) g.cls);
    if (SMALL && cls < 6) {
        switch (cls) {
            CASE(0, 4, 4) CASE(1, 4, 8) CASE(2, 4, 16)
            CASE(3, 8, 4) CASE(4, 8, 8) CASE(5, 8, 16)
        }
        return;
    }
    switch (cls) {
        CASE(6, 16, 4) CASE(7, 16, 8) CASE(8, 16, 16)
        CASE(9, 32, 4) CASE(10, 32, 8) CASE(11, 32, 16)
        CASE(12, 64, 4) CASE(13, 64, 8) CASE(14, 64, 16)
    }
#undef CASE
}

template <typename pixel, bool TILED>
hipError_t launch_cls(const int cls, const DevPlanes &dst, const RefSet &refs, const McTile *tiles, const int n,
                      int16_t *prep, const int bitdepth_max, hipStream_t stream)
{
#define CASE(C, TW, TH) case C: { \
        constexpr int lpt = mc_cmin(64, TW * TH / 4); \
        constexpr int g = 64 / lpt; \
        hipLaunchKernelGGL((mc_kernel<TW, TH, pixel, TILED>), dim3(((n + g - 1) / g + MC_WAVES - 1) / MC_WAVES), dim3(64 * MC_WAVES), 0, stream, \
                           dst, refs, tiles, n, prep, bitdepth_max); \
        break; }
    // class = 3 * wclass + hclass, widths 4 8 16 32 64, heights 4 8 16
    switch (cls) {
        CASE(0, 4, 4) CASE(1, 4, 8) CASE(2, 4, 16)
        CASE(3, 8, 4) CASE(4, 8, 8) CASE(5, 8, 16)
        CASE(6, 16, 4) CASE(7, 16, 8) CASE(8, 16, 16)
        CASE(9, 32, 4) CASE(10, 32, 8) CASE(11, 32, 16)
        CASE(12, 64, 4) CASE(13, 64, 8) CASE(14, 64, 16)
        default: return hipErrorInvalidValue;
    }
#undef CASE
    return hipGetLastError();
}

template <typename pixel>
hipError_t launch_cls_twin(const int cls, const DevPlanes &dst, const RefSet &refs, const McTile *tiles, const int n,
                           int16_t *prep, const int bitdepth_max, const DevPlanes &twin, hipStream_t stream)
{
#define CASE(C, TW, TH) case C: { \
        constexpr int lpt = mc_cmin(64, TW * TH / 4); \
        constexpr int g = 64 / lpt; \
        hipLaunchKernelGGL((mc_twin_kernel<TW, TH, pixel>), dim3((n + g - 1) / g), dim3(64), 0, stream, \
                           dst, refs, tiles, n, prep, bitdepth_max, twin); \
        break; }
    switch (cls) {
        CASE(0, 4, 4) CASE(1, 4, 8) CASE(2, 4, 16)
        CASE(3, 8, 4) CASE(4, 8, 8) CASE(5, 8, 16)
        CASE(6, 16, 4) CASE(7, 16, 8) CASE(8, 16, 16)
        CASE(9, 32, 4) CASE(10, 32, 8) CASE(11, 32, 16)
        CASE(12, 64, 4) CASE(13, 64, 8) CASE(14, 64, 16)
        default: return hipErrorInvalidValue;
    }
#undef CASE
    return hipGetLastError();
}

} // namespace

// with the tiled twin of dst written along (tiled references only: the twin form exists for the TILED kernels)
extern "C" int dav1d_hip_launch_mc_bin_twin(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc, int cls,
                                            const McTile *tiles, int n, int16_t *prep, const DevPlanes *dst_twin, void *stream)
{
    if (n <= 0) return 0;
    if (!dst_twin || refs_tiled(refs, n_refs) != 1) return -EINVAL;
    RefSet rs;
    for (int i = 0; i < 8; i++) rs.r[i] = refs[i < n_refs ? i : 0];
    const int bitdepth_max = (1 << bpc) - 1;
    hipError_t e;
#ifdef DV_LEAN
    if (bpc == 8) return -ENOTSUP;
#else
    if (bpc == 8) e = launch_cls_twin<uint8_t>(cls, *dst, rs, tiles, n, prep, bitdepth_max, *dst_twin, (hipStream_t) stream);
    else
#endif
                  e = launch_cls_twin<uint16_t>(cls, *dst, rs, tiles, n, prep, bitdepth_max, *dst_twin, (hipStream_t) stream);
    return hip_rc(e);
}

namespace {
} // namespace

extern "C" int dav1d_hip_launch_mc_bin(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc, int cls,
                                       const McTile *tiles, int n, int16_t *prep, void *stream)
{
    if (n <= 0) return 0;
    RefSet rs;
    for (int i = 0; i < 8; i++) rs.r[i] = refs[i < n_refs ? i : 0];
    const int bitdepth_max = (1 << bpc) - 1;
    const int tiled = refs_tiled(refs, n_refs);
    if (tiled < 0) return -EINVAL;
#ifdef DV_LEAN
    return -ENOTSUP;
#else
    hipError_t e;
    if (bpc == 8) e = tiled ? launch_cls<uint8_t, true>(cls, *dst, rs, tiles, n, prep, bitdepth_max, (hipStream_t) stream)
                            : launch_cls<uint8_t, false>(cls, *dst, rs, tiles, n, prep, bitdepth_max, (hipStream_t) stream);
    else          e = tiled ? launch_cls<uint16_t, true>(cls, *dst, rs, tiles, n, prep, bitdepth_max, (hipStream_t) stream)
                            : launch_cls<uint16_t, false>(cls, *dst, rs, tiles, n, prep, bitdepth_max, (hipStream_t) stream);
    return hip_rc(e);
#endif
}

// with_small = 0: the group list only holds shapes that are at least 16 wide
extern "C" int dav1d_hip_launch_mc_all(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc, const McTile *tiles,
                                       const McGroup *groups, int n_groups, int with_small, int16_t *prep, void *stream)
{
    if (n_groups <= 0) return 0;
    if (refs_tiled(refs, n_refs) != 0) return -EINVAL;       // the all-shapes launch (an experiment, off by default) reads raster planes
#ifdef DV_LEAN
    return -ENOTSUP;
#else
    RefSet rs;
    for (int i = 0; i < 8; i++) rs.r[i] = refs[i < n_refs ? i : 0];
    const int bitdepth_max = (1 << bpc) - 1;
    const dim3 grid(n_groups), wave(64);
    hipStream_t st = (hipStream_t) stream;
    if (bpc == 8 && with_small) hipLaunchKernelGGL((mc_all_kernel<uint8_t, true>), grid, wave, 0, st, *dst, rs, tiles, groups, n_groups, prep, bitdepth_max);
    else if (bpc == 8) hipLaunchKernelGGL((mc_all_kernel<uint8_t, false>), grid, wave, 0, st, *dst, rs, tiles, groups, n_groups, prep, bitdepth_max);
    else if (with_small) hipLaunchKernelGGL((mc_all_kernel<uint16_t, true>), grid, wave, 0, st, *dst, rs, tiles, groups, n_groups, prep, bitdepth_max);
    else hipLaunchKernelGGL((mc_all_kernel<uint16_t, false>), grid, wave, 0, st, *dst, rs, tiles, groups, n_groups, prep, bitdepth_max);
    return hip_rc(hipGetLastError());
#endif
}

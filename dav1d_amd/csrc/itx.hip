// Batched inverse transform + add (itxfm_add) for gfx950.
//
// Contract per task = reference inv_txfm_add_c (src/itx_tmpl.c:43-124) and
// inv_txfm_add_wht_wht_4x4_c (:184-203): dc-only shortcut, rect2 pre-scale, row
// pass on c[x] = coeff[y + x*sh], intermediate round/clip, column pass, add to dst
// with pixel clip, and the coefficient slab zeroed.
//
// Mapping: every lane evaluates complete 1-D transforms in registers (itx1d.h).
// A W x H block owns LPB = max(min(H,32), W) lanes: in the row pass lane r
// transforms row r, in the column pass lane c transforms column c; the transpose in
// between goes through LDS with a padded row stride (conflict-free both ways).
// 64/LPB blocks share one wave, so 4x4 blocks run 16 to a wave and 64x64 one.
// The slab is fetched with 16-byte loads into LDS (zeroed with 16-byte stores in the same
// sweep) together with the destination pixels the column pass will need much later; the
// row pass reads it transposed, and the same LDS region then becomes the transpose buffer.
#define DV_UNIT itx        // (names this unit's phase accessor in -DDV_PHASES variant builds, common.h)
#include "itx_body.h"
#include "capi.h"
#include <string.h>

DV_PHASE_DEFINE(DV_UNIT)
namespace {

template <int TX, typename pixel, typename coef>
__global__ __launch_bounds__(64) void itx_add_kernel(const DevPlanes dst, const Dav1dHipItxTask *__restrict__ tasks,
                                                     const int n, coef *__restrict__ cf, const int bitdepth_max)
{
    __shared__ __attribute__((aligned(16))) int tmp_s[itx_lds_ints<TX>()];
    itx_body<TX, pixel, coef>(dst, tasks, n, cf, bitdepth_max, (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x), tmp_s);
}

// The same with the sums leaving through an LDS tile in row pieces (tile_write_out, itx_body.h) — to the raster planes and, when
// `twin` has planes, to the picture's tiled twin.  The tile shares its memory with the transpose buffer: every lane has its column in
// registers before the first sum is written (itx_body's COH form).
template <int TX, typename pixel, typename coef>
__global__ __launch_bounds__(64) void itx_add_wide_kernel(const DevPlanes dst, const Dav1dHipItxTask *__restrict__ tasks,
                                                          const int n, coef *__restrict__ cf, const int bitdepth_max, const DevPlanes twin)
{
    constexpr int W = tx_w(TX), H = tx_h(TX), LPB = cmax(cmin(H, 32), W), BPW = 64 / LPB;
    static_assert(BPW * itx_tile_stride(W, H) * (int) sizeof(pixel) <= itx_lds_ints<TX>() * 4, "the tile fits the transpose buffer");
    __shared__ __attribute__((aligned(16))) int tmp_s[itx_lds_ints<TX>()];
    pixel *const tile = reinterpret_cast<pixel *>(tmp_s);
    const int group = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x);
    const int block0 = group * BPW;
    if (block0 >= n) return;
    DV_PHASE_BEGIN();
    // twin.tiled == 2: the picture lives in its twin only — the predicted pixels are read from there and the sums go back there
    // (`twin` carries the strides and sizes of the raster planes next to the twin's data pointers; a DevPlanes put together here from the
    // two kernel arguments would live in scratch memory)
    const bool twin_only = twin.tiled == 2;
    uint32_t toff = 0;
    int tpl = 0;
    if (twin_only) itx_body<TX, pixel, coef, false, true>(twin, tasks, n, cf, bitdepth_max, group, tmp_s, tile, true, &toff, &tpl);
    else itx_body<TX, pixel, coef, false, true>(dst, tasks, n, cf, bitdepth_max, group, tmp_s, tile, false, &toff, &tpl);
    dv::wave_sync();
    DV_PHASE(512 + TX * 16 + 6);        // (the body's own marks, then the wait for the sums in the tile)
    tile_write_out<W, H, BPW, pixel>(tile, tasks + block0, dv::imin(BPW, n - block0), dst, twin, twin.data[0] != nullptr, !twin_only, toff, tpl);
    DV_PHASE(512 + TX * 16 + 7);        // tile_write_out
}

// Every transform size in one launch, for the short lists of an intra wavefront step (a few hundred blocks of up to
// five sizes: one launch instead of five keeps the dependent chain of the step short).  tasks[] is the size-binned
// list; seg.off[b] = first task of size b, seg.grp[b] = first workgroup of size b.  LDS / registers are those of the
// hungriest size, which does not matter for lists that cannot fill the chip anyway.
struct ItxSegments { int off[20]; int grp[20]; };
constexpr int itx_lds_ints_of(int tx) {
    return (64 / cmax(cmin(tx_h(tx), 32), tx_w(tx))) * cmin(tx_h(tx), 32) * (tx_w(tx) + 1);
}
constexpr int itx_lds_max(int tx = 0) { return tx == 19 ? 0 : cmax(itx_lds_ints_of(tx), itx_lds_max(tx + 1)); }
static_assert(itx_lds_ints_of(7) == itx_lds_ints<7>() && itx_lds_ints_of(4) == itx_lds_ints<4>(), "LDS sizing");

template <typename pixel, typename coef>
__global__ __launch_bounds__(64) void itx_multi_kernel(const DevPlanes dst, const Dav1dHipItxTask *__restrict__ tasks,
                                                       const ItxSegments seg, coef *__restrict__ cf, const int bitdepth_max)
{
    __shared__ __attribute__((aligned(16))) int tmp_s[itx_lds_max()];
    const int g = blockIdx.x;
    int b = 0;
#pragma unroll
    for (int k = 1; k < 19; k++) b = g >= seg.grp[k] ? k : b;
    const int first = seg.off[b], cnt = seg.off[b + 1] - first, group = g - seg.grp[b];
#define CASE(T) case T: itx_body<T, pixel, coef>(dst, tasks + first, cnt, cf, bitdepth_max, group, tmp_s); break;
    switch (b) {
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9)
        CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18)
    }
#undef CASE
}

template <typename pixel, typename coef>
hipError_t launch_tx_wide(const int tx, const DevPlanes &dst, const Dav1dHipItxTask *tasks, const int n,
                          coef *cf, const int bitdepth_max, const DevPlanes &twin, hipStream_t stream)
{
#define CASE(T) case T: { \
        constexpr int lpb = cmax(cmin(tx_h(T), 32), tx_w(T)); \
        constexpr int bpw = 64 / lpb; \
        const int grid = (n + bpw - 1) / bpw; \
        hipLaunchKernelGGL((itx_add_wide_kernel<T, pixel, coef>), dim3(grid), dim3(64), 0, stream, \
                           dst, tasks, n, cf, bitdepth_max, twin); \
        break; }
    switch (tx) {
#ifdef DV_LEAN      // (variant builds that only hold what the 10-bit tiled step launches, see recon.hip)
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4)
#else
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9)
        CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18)
#endif
        default: return hipErrorInvalidValue;
    }
#undef CASE
    return hipGetLastError();
}

template <typename pixel, typename coef>
hipError_t launch_tx(const int tx, const DevPlanes &dst, const Dav1dHipItxTask *tasks, const int n,
                     coef *cf, const int bitdepth_max, hipStream_t stream)
{
#define CASE(T) case T: { \
        constexpr int lpb = cmax(cmin(tx_h(T), 32), tx_w(T)); \
        constexpr int bpw = 64 / lpb; \
        const int grid = (n + bpw - 1) / bpw; \
        hipLaunchKernelGGL((itx_add_kernel<T, pixel, coef>), dim3(grid), dim3(64), 0, stream, \
                           dst, tasks, n, cf, bitdepth_max); \
        break; }
    switch (tx) {
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9)
        CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18)
        default: return hipErrorInvalidValue;
    }
#undef CASE
    return hipGetLastError();
}

} // namespace

// tasks[] (device) = a whole size-binned list, off[b] .. off[b + 1] = the tasks of size b: one launch for all of them
extern "C" int dav1d_hip_launch_itx_all(const DevPlanes *dst, int bpc, const Dav1dHipItxTask *tasks, const size_t *off,
                                        void *coef, void *stream)
{
    ItxSegments seg;
    int groups = 0;
    for (int b = 0; b < 19; b++) {
        const int lpb = cmax(cmin(tx_h(b), 32), tx_w(b)), bpw = 64 / lpb;
        seg.off[b] = (int) off[b];
        seg.grp[b] = groups;
        groups += ((int) (off[b + 1] - off[b]) + bpw - 1) / bpw;
    }
    seg.off[19] = (int) off[19];
    seg.grp[19] = groups;
    if (!groups) return 0;
#ifdef DV_LEAN
    return -ENOTSUP;
#else
    const int bitdepth_max = (1 << bpc) - 1;
    if (bpc == 8)
        hipLaunchKernelGGL((itx_multi_kernel<uint8_t, int16_t>), dim3(groups), dim3(64), 0, (hipStream_t) stream,
                           *dst, tasks, seg, (int16_t *) coef, bitdepth_max);
    else
        hipLaunchKernelGGL((itx_multi_kernel<uint16_t, int32_t>), dim3(groups), dim3(64), 0, (hipStream_t) stream,
                           *dst, tasks, seg, (int32_t *) coef, bitdepth_max);
    return hipGetLastError() == hipSuccess ? 0 : -5;
#endif
}

// tasks[] (device) holds the tasks of ONE tx size; offsets are managed by capi.
extern "C" int dav1d_hip_launch_itx_bin(const DevPlanes *dst, int bpc, int tx, const Dav1dHipItxTask *tasks,
                                        int n, void *coef, void *stream)
{
    if (n <= 0) return 0;
#ifdef DV_LEAN
    return -ENOTSUP;
#else
    const int bitdepth_max = (1 << bpc) - 1;
    hipError_t e;
    if (bpc == 8) e = launch_tx<uint8_t, int16_t>(tx, *dst, tasks, n, (int16_t *) coef, bitdepth_max, (hipStream_t) stream);
    else          e = launch_tx<uint16_t, int32_t>(tx, *dst, tasks, n, (int32_t *) coef, bitdepth_max, (hipStream_t) stream);
    return e == hipSuccess ? 0 : -5;
#endif
}

// the same with wide stores (wide != 0) and, optionally, the tiled twin of dst written along (dst_twin != NULL, needs wide)
extern "C" int dav1d_hip_launch_itx_bin_out(const DevPlanes *dst, int bpc, int tx, const Dav1dHipItxTask *tasks, int n, void *coef, int wide,
                                            const DevPlanes *dst_twin, void *stream)
{
    if (!wide) return dst_twin ? -EINVAL : dav1d_hip_launch_itx_bin(dst, bpc, tx, tasks, n, coef, stream);
    if (n <= 0) return 0;
    const int bitdepth_max = (1 << bpc) - 1;
    DevPlanes twin;
    memset(&twin, 0, sizeof(twin));
    if (dst_twin) twin = *dst_twin;
    hipError_t e;
#ifdef DV_LEAN
    if (bpc == 8) return -ENOTSUP;
#else
    if (bpc == 8) e = launch_tx_wide<uint8_t, int16_t>(tx, *dst, tasks, n, (int16_t *) coef, bitdepth_max, twin, (hipStream_t) stream);
    else
#endif
                  e = launch_tx_wide<uint16_t, int32_t>(tx, *dst, tasks, n, (int32_t *) coef, bitdepth_max, twin, (hipStream_t) stream);
    return e == hipSuccess ? 0 : -5;
}

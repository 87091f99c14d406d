// Prediction and residual of a block in one wave (gfx950).
//
// The reference reconstructs an inter block as mc() into the picture followed by itxfm_add() on the same pixels
// (src/recon_tmpl.c:1557-2000).  Run as two launches, every predicted pixel is written to HBM once and read back once by
// the residual launch — 4 of the 16 bytes per sample the pair moves.  Where a transform block covers exactly one prediction
// block (same rectangle; the recon list pairs them up), the wave that owns the transform block first runs the
// motion-compensation body for the block's tiles with the LDS as destination, then the inverse-transform body with the LDS
// as the source of the pixels it adds to: the picture is written once.  Both bodies are the ones of mc.hip / itx.hip
// (mc_body.h, itx_body.h), bit for bit.
//
// One kernel per square transform size (4x4 .. 64x64): a wave takes as many blocks as the transform body packs into a wave
// (16, 8, 4, 2, 1) and predicts them tile group by tile group (tiles of <= 64x16, as in mc.hip).
#define DV_UNIT recon        // (names this unit's phase accessor in -DDV_PHASES variant builds, common.h)
#include "mc_body.h"
#include "itx_body.h"
#include <string.h>

DV_PHASE_DEFINE(DV_UNIT)
namespace {

#ifndef RECON_WAVES
#define RECON_WAVES 7      // (7 waves per SIMD asked of the register allocator: 72 VGPRs; measured on the 16x16 pairs 74.7 -> 70.1 us, profiles/r05)
#endif
#ifndef RECON_WAVES_8
#define RECON_WAVES_8 7
#endif
#ifndef RECON_WAVES_32
#define RECON_WAVES_32 7   // (the 32x32 pairs: 90 registers whatever is asked, and the LDS holds 4.5 waves per SIMD)
#endif

constexpr int rc_log2(int v) { return v <= 1 ? 0 : 1 + rc_log2(v >> 1); }

// Waves per workgroup of the cooperative form: the prediction body takes G tiles per call, a wave's blocks have BPW * TPB of them.
// Where that is more than one call (8x8: 2; 16x16, 32x32, 64x64: 4) the calls are independent chains of record -> window fetch ->
// two filter passes; the cooperative form gives every call a wave of its own (the fetches of all of a group's tiles in flight
// together, the transform one fetch latency after launch instead of four) and the other waves end at the barrier.  Measured on
// MI355X: with enough groups to fill the chip several times over (an 8K frame: 6,500 - 19,000 groups per size) the one-wave
// form is 5 - 10 % faster, the SIMDs are busy either way and the extra waves cost LDS and a barrier; with few groups (a 4K frame:
// 1,600 - 4,800, one or two per SIMD) a kernel lasts as long as its longest wave and the cooperative form shortens that.
// The launch picks by group count (option recon_coop_below).
template <int CLS> constexpr int recon_waves() {
    constexpr int W = 4 << CLS, TW = mc_cmin(W, 64), TH = mc_cmin(W, 16), TPB = (W / TW) * (W / TH);
    constexpr int LPB = cmax(cmin(W, 32), W), BPW = 64 / LPB, G = 64 / mc_cmin(64, TW * TH / 4);
    return (BPW * TPB + G - 1) / G;
}

// WIDE: the reconstructed blocks leave through the LDS tile in row pieces of up to 16 bytes (tile_write_out, itx_body.h) instead of
// two bytes per lane and row — and go to the picture's tiled twin as well when `twin` has planes (twin.data[0] != nullptr).
template <int CLS, typename pixel, typename coef, bool COOP, bool TILED, bool WIDE>
__global__ __launch_bounds__(COOP ? 64 * recon_waves<CLS>() : 64, COOP ? 1 : CLS == 1 ? RECON_WAVES_8 : CLS == 3 ? RECON_WAVES_32 : RECON_WAVES)
void recon_fused_kernel(const DevPlanes dst, const RefSet refs, const McTile *__restrict__ tiles,
                        const Dav1dHipItxTask *__restrict__ tasks, const int n_blocks,
                        int16_t *__restrict__ prep, coef *__restrict__ cf, const int bitdepth_max, const DevPlanes twin)
{
    constexpr int TX = CLS;                                  // TX_4X4 .. TX_64X64
    constexpr int W = 4 << CLS;
    constexpr int TW = mc_cmin(W, 64), TH = mc_cmin(W, 16);  // tile shape of mc.hip for a W x W block
    constexpr int TPB = (W / TW) * (W / TH);                 // tiles per block: 1, 1, 1, 2, 4
    constexpr int LPB = cmax(cmin(W, 32), W), BPW = 64 / LPB;   // blocks per wave of the transform body
    constexpr int G = 64 / mc_cmin(64, TW * TH / 4);         // tiles per call of the prediction body
    constexpr int NW = COOP ? recon_waves<CLS>() : 1;
    // One LDS buffer, used twice: [prediction windows + intermediates, one set per wave | the predicted blocks] while the blocks
    // are predicted, then [slabs / transpose buffer of the transform] — the transform body has the predicted pixels in registers
    // before it stores its first slab chunk, so the regions may overlap.  Fewer LDS bytes per wave = more resident waves per CU,
    // which is what these kernels are short of (16x16: 8320 -> 4352 bytes, 5 -> 6 waves per SIMD, 89 -> 76 us per 8K frame).
    constexpr int MC_B = (mc_lds_bytes<TW, TH, TILED>() + 15) / 16 * 16, PRED_B = BPW * itx_tile_stride(W, W) * (int) sizeof(pixel);
    constexpr int ITX_B = itx_lds_ints<TX>() * 4;
    constexpr int LDS_B = cmax(NW * MC_B + PRED_B, ITX_B);
    __shared__ uint4 smem[(LDS_B + 15) / 16];
    const int wave = NW == 1 ? 0 : (int) (threadIdx.x >> 6);
    uint4 *const smem_mc = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(smem) + wave * MC_B);
    pixel *const pred = reinterpret_cast<pixel *>(reinterpret_cast<char *>(smem) + NW * MC_B);
    int *const smem_itx = reinterpret_cast<int *>(smem);

    const int group = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x);
    const int block0 = group * BPW;
    if (block0 >= n_blocks) return;
    // phase slots of the paired kernel of this size (DV_PHASES builds): 0 every prediction of the wave's blocks, 1 the transform body,
    // 2 tile_write_out, 3 whole wave, 4 waves counted (the bodies' own slots: mc_body.h 256 + ..., itx_body.h 512 + ... + 8)
    DV_PHASE_BEGIN();
    const int nb = dv::imin(BPW, n_blocks - block0);
    const int tile0 = block0 * TPB, ntile = nb * TPB;
#ifdef DV_KO_NOMC
    if (bitdepth_max == 12345)
#endif
    if constexpr (NW == 1) {
        for (int c = 0; c < ntile; c += G) {
            mc_body<TW, TH, pixel, true, TILED>(dst, refs, tiles, tile0 + c, dv::imin(G, ntile - c), prep, bitdepth_max, smem_mc,
                                         pred, tile0, rc_log2(TPB), W, W);
            dv::wave_sync();
        }
    } else {
        const int c = wave * G;
        if (c < ntile)
            mc_body<TW, TH, pixel, true, TILED>(dst, refs, tiles, tile0 + c, dv::imin(G, ntile - c), prep, bitdepth_max, smem_mc,
                                         pred, tile0, rc_log2(TPB), W, W);
        __syncthreads();
        if (wave) return;
    }
    DV_PHASE(768 + CLS * 16 + 0);
    if constexpr (WIDE) {
        uint32_t toff = 0;
        int tpl = 0;
#ifdef DV_KO_NOITX
        if (bitdepth_max == 12345)
#endif
        itx_body<TX, pixel, coef, true, true>(dst, tasks, n_blocks, cf, bitdepth_max, group, smem_itx, pred, false, &toff, &tpl);
        dv::wave_sync();
        DV_PHASE(768 + CLS * 16 + 1);
#ifdef DV_KO_NOOUT
        if (bitdepth_max == 12345)
#endif
        tile_write_out<W, W, BPW, pixel>(pred, tasks + block0, nb, dst, twin, twin.data[0] != nullptr, twin.tiled != 2, toff, tpl);
        DV_PHASE(768 + CLS * 16 + 2);
    } else {
        itx_body<TX, pixel, coef, true>(dst, tasks, n_blocks, cf, bitdepth_max, group, smem_itx, pred);
        DV_PHASE(768 + CLS * 16 + 1);
    }
    DV_PHASE_WAVE(768 + CLS * 16 + 3);
}

template <int CLS, typename pixel, typename coef, bool TILED, bool WIDE>
void launch_cls(const DevPlanes &dst, const RefSet &refs, const McTile *tiles, const Dav1dHipItxTask *tasks, const int n,
                int16_t *prep, coef *cf, const int bitdepth_max, const int coop_below, const DevPlanes &twin, hipStream_t stream)
{
    constexpr int W = 4 << CLS, LPB = cmax(cmin(W, 32), W), BPW = 64 / LPB;
    const int groups = (n + BPW - 1) / BPW;
#ifndef DV_LEAN      // (DV_LEAN: variant builds of tools/build_variant.py that only hold what the 10-bit tiled step launches — a tenth of the compile time)
    if (recon_waves<CLS>() > 1 && groups < coop_below)
        hipLaunchKernelGGL((recon_fused_kernel<CLS, pixel, coef, true, TILED, WIDE>), dim3(groups), dim3(64 * recon_waves<CLS>()), 0, stream,
                           dst, refs, tiles, tasks, n, prep, cf, bitdepth_max, twin);
    else
#endif
        hipLaunchKernelGGL((recon_fused_kernel<CLS, pixel, coef, false, TILED, WIDE>), dim3(groups), dim3(64), 0, stream,
                           dst, refs, tiles, tasks, n, prep, cf, bitdepth_max, twin);
}

template <typename pixel, typename coef, bool TILED, bool WIDE>
hipError_t launch_any(const int cls, const DevPlanes &dst, const RefSet &refs, const McTile *tiles, const Dav1dHipItxTask *tasks,
                      const int n, int16_t *prep, coef *cf, const int bitdepth_max, const int coop_below, const DevPlanes &twin, hipStream_t stream)
{
    switch (cls) {
    case 0: launch_cls<0, pixel, coef, TILED, WIDE>(dst, refs, tiles, tasks, n, prep, cf, bitdepth_max, coop_below, twin, stream); break;
    case 1: launch_cls<1, pixel, coef, TILED, WIDE>(dst, refs, tiles, tasks, n, prep, cf, bitdepth_max, coop_below, twin, stream); break;
    case 2: launch_cls<2, pixel, coef, TILED, WIDE>(dst, refs, tiles, tasks, n, prep, cf, bitdepth_max, coop_below, twin, stream); break;
    case 3: launch_cls<3, pixel, coef, TILED, WIDE>(dst, refs, tiles, tasks, n, prep, cf, bitdepth_max, coop_below, twin, stream); break;
    case 4: launch_cls<4, pixel, coef, TILED, WIDE>(dst, refs, tiles, tasks, n, prep, cf, bitdepth_max, coop_below, twin, stream); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <typename pixel, typename coef>
hipError_t launch_variant(const bool tiled, const bool wide, const int cls, const DevPlanes &dst, const RefSet &refs, const McTile *tiles,
                          const Dav1dHipItxTask *tasks, const int n, int16_t *prep, coef *cf, const int bitdepth_max, const int coop_below,
                          const DevPlanes &twin, hipStream_t st)
{
#ifdef DV_LEAN
    if (!tiled || !wide || sizeof(pixel) != 2) return hipErrorInvalidValue;
    if constexpr (sizeof(pixel) == 2) return launch_any<pixel, coef, true, true>(cls, dst, refs, tiles, tasks, n, prep, cf, bitdepth_max, coop_below, twin, st);
    else return hipErrorInvalidValue;
#else
    if (tiled) return wide ? launch_any<pixel, coef, true, true>(cls, dst, refs, tiles, tasks, n, prep, cf, bitdepth_max, coop_below, twin, st)
                           : launch_any<pixel, coef, true, false>(cls, dst, refs, tiles, tasks, n, prep, cf, bitdepth_max, coop_below, twin, st);
    return wide ? launch_any<pixel, coef, false, true>(cls, dst, refs, tiles, tasks, n, prep, cf, bitdepth_max, coop_below, twin, st)
                : launch_any<pixel, coef, false, false>(cls, dst, refs, tiles, tasks, n, prep, cf, bitdepth_max, coop_below, twin, st);
#endif
}

} // namespace

// tiles[] / tasks[] (device): the n blocks of ONE square transform size cls = 0 (4x4) .. 4 (64x64); block i owns
// tasks[i] and the (1, 1, 1, 2, 4) tiles starting at tiles[i * tiles_per_block].  wide: the blocks leave through tile_write_out
// (planes and strides must be 16-byte aligned: the caller checks); dst_twin (with wide; may be NULL): the planes of dst's tiled twin,
// written along with the raster planes.
extern "C" int dav1d_hip_launch_recon_fused_out(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc, int cls, const McTile *tiles,
                                                const Dav1dHipItxTask *tasks, int n, int16_t *prep, void *coef, int coop_below, int wide,
                                                const DevPlanes *dst_twin, void *stream)
{
    if (n <= 0) return 0;
    if (dst_twin && !wide) return -EINVAL;
    RefSet rs;
    for (int i = 0; i < 8; i++) rs.r[i] = refs[i < n_refs ? i : 0];
    const int bitdepth_max = (1 << bpc) - 1;
    const int tiled = refs_tiled(refs, n_refs);
    if (tiled < 0) return -EINVAL;
    DevPlanes twin;
    memset(&twin, 0, sizeof(twin));
    if (dst_twin) twin = *dst_twin;
    hipStream_t st = (hipStream_t) stream;
    hipError_t e;
    if (bpc == 8) e = launch_variant<uint8_t, int16_t>(tiled, wide & 1, cls, *dst, rs, tiles, tasks, n, prep, (int16_t *) coef, bitdepth_max, coop_below, twin, st);
    else          e = launch_variant<uint16_t, int32_t>(tiled, wide & 1, cls, *dst, rs, tiles, tasks, n, prep, (int32_t *) coef, bitdepth_max, coop_below, twin, st);
    return hip_rc(e);
}

extern "C" int dav1d_hip_launch_recon_fused(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc, int cls, const McTile *tiles,
                                            const Dav1dHipItxTask *tasks, int n, int16_t *prep, void *coef, int coop_below, void *stream)
{
    return dav1d_hip_launch_recon_fused_out(dst, refs, n_refs, bpc, cls, tiles, tasks, n, prep, coef, coop_below, 0, nullptr, stream);
}

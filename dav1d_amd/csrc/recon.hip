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
__global__ __launch_bounds__(COOP ? 64 * recon_waves<CLS>() : 64, COOP ? 1 : CLS == 1 ? RECON_WAVES_8 : RECON_WAVES)
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
    constexpr int MC_B = (mc_lds_bytes<TW, TH, TILED>() + 15) / 16 * 16, PRED_B = BPW * W * W * (int) sizeof(pixel);
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
        itx_body<TX, pixel, coef, true, true>(dst, tasks, n_blocks, cf, bitdepth_max, group, smem_itx, pred, false, &toff, &tpl);
        dv::wave_sync();
        DV_PHASE(768 + CLS * 16 + 1);
        tile_write_out<W, W, BPW, pixel>(pred, tasks + block0, nb, dst, twin, twin.data[0] != nullptr, twin.tiled != 2, toff, tpl);
        DV_PHASE(768 + CLS * 16 + 2);
    } else {
        itx_body<TX, pixel, coef, true>(dst, tasks, n_blocks, cf, bitdepth_max, group, smem_itx, pred);
        DV_PHASE(768 + CLS * 16 + 1);
    }
    DV_PHASE_WAVE(768 + CLS * 16 + 3);
}

// ---- the pipelined form (tiled references, 10 / 12 bits, block sizes whose tiles a whole wave works on: 16x16 and up)
//
// Measured on the kernel above (profiles/r06/knockouts.txt): with every global memory access AND all of the arithmetic taken out, a wave
// of the paired kernels still lives 40 % of its time — what it does then is wait for its records, four times in a row, one tile after
// the other; with the accesses back in, every tile adds a window trip and the transform a coefficient trip, each behind the one before:
// ten dependent trips to memory for a wave of four 16x16 blocks.  Here a wave makes TWO: (1) every record it owns — its tiles' in one
// sweep into LDS, its transform blocks' into registers; (2) everything those records name — the window of every tile by LDS-DMA
// (mc_gather_dma: no registers, no waiting) and the coefficient prefixes into registers.  Then it computes: horizontal and vertical
// pass tile by tile out of the windows that are all there, the second window of a compound tile fetched into the first one's place as
// soon as the horizontal pass has read it (it arrives under the tiles that follow), the transform on coefficients that landed long ago.
template <int CLS> constexpr int piped_lds_bytes() {
    constexpr int W = 4 << CLS, TW = mc_cmin(W, 64), TH = mc_cmin(W, 16), TPB = (W / TW) * (W / TH);
    constexpr int LPB = cmax(cmin(W, 32), W), BPW = 64 / LPB, NT = BPW * TPB;
    constexpr int WS = mc_win_stride_tiled(TW), WIN_B = (TH + 8) * WS * 2, MID_B = (TH + 8) / 2 * TW * 4, REC_B = (NT * (int) sizeof(McTile) + 15) / 16 * 16;
    return cmax(REC_B + NT * WIN_B + MID_B + BPW * W * W * 2, REC_B + itx_lds_ints<CLS>() * 4);
}
#ifndef RECON_PIPE_WAVES
#define RECON_PIPE_WAVES 5
#endif
template <int CLS, typename pixel, typename coef>
__global__ __launch_bounds__(64, RECON_PIPE_WAVES)
void recon_piped_kernel(const DevPlanes dst, const RefSet refs, const McTile *__restrict__ tiles,
                        const Dav1dHipItxTask *__restrict__ tasks, const int n_blocks,
                        coef *__restrict__ cf, const int bitdepth_max, const DevPlanes twin)
{
    constexpr int TX = CLS, W = 4 << CLS;
    constexpr int TW = mc_cmin(W, 64), TH = mc_cmin(W, 16), TPB = (W / TW) * (W / TH);
    constexpr int LPB = cmax(cmin(W, 32), W), BPW = 64 / LPB, NT = BPW * TPB;
    typedef McShape<TW, TH, pixel, true> S;
    static_assert(S::G == 1 && S::HBD, "tiles a whole wave works on, 16-bit pixels");
    constexpr int WS = S::WS, WIN_B = S::WR * WS * 2, MID_B = S::NPR * TW * 4, REC_B = (NT * (int) sizeof(McTile) + 15) / 16 * 16;
    constexpr int RW = sizeof(McTile) / 4, R = S::R, NS = S::NS;
    constexpr int TPB_LOG2 = rc_log2(TPB);
    __shared__ uint4 smem[(piped_lds_bytes<CLS>() + 15) / 16];
    char *const base = reinterpret_cast<char *>(smem);
    uint32_t *const rec_s = reinterpret_cast<uint32_t *>(base);
    char *const win_s = base + REC_B;
    uint32_t *const mid = reinterpret_cast<uint32_t *>(win_s + NT * WIN_B);
    pixel *const pred = reinterpret_cast<pixel *>(win_s + NT * WIN_B + MID_B);
    int *const smem_itx = reinterpret_cast<int *>(base + REC_B);

    const int lane = threadIdx.x & 63;
    const int group = (int) dv::xcd_chunk_id(blockIdx.x, gridDim.x);
    const int block0 = group * BPW;
    if (block0 >= n_blocks) return;
    const int nb = dv::imin(BPW, n_blocks - block0);
    const int tile0 = block0 * TPB, ntile = nb * TPB;

    // ---- trip 1: the records
    ItxPre<TX, coef> pre;
    itx_prefetch_task<TX, coef>(pre, tasks, n_blocks, group);
    {
        const uint32_t *recs = reinterpret_cast<const uint32_t *>(tiles + tile0);
        const int nw = ntile * RW;
        for (int i = lane; i < nw; i += 64) rec_s[i] = recs[i];
    }
    dv::wave_sync();
    // ---- trip 2: everything they name
    itx_prefetch_coefs<TX, coef>(pre, cf, n_blocks, group);
    auto record = [&](const int c) {
        McTile t;
        uint32_t *tw_ = reinterpret_cast<uint32_t *>(&t);
#pragma unroll
        for (int i = 0; i < RW; i++) tw_[i] = (uint32_t) __builtin_amdgcn_readfirstlane((int) rec_s[c * RW + i]);
        return t;
    };
    // one reference of one tile: its plane, and whether its window lies inside it
    auto plane_of = [&](const McTile &t, const McRef &rf, const pixel *&src, int &rs, int &rw, int &rh) {
        const DevPlanes &rp = refs.r[rf.ref];
        src = reinterpret_cast<const pixel *>(rp.data[t.plane]);
        rs = rp.stride[t.plane]; rw = rp.w[t.plane]; rh = rp.h[t.plane];
    };
    auto fetch = [&](const McTile &t, const McRef &rf, const int c) {
        const McPred pd = mc_pred_of<TW, TH>(rf);
        const pixel *src; int rs, rw, rh;
        plane_of(t, rf, src, rs, rw, rh);
        if (mc_window_inside<TW, TH>(rf, pd, rw, rh)) mc_gather_dma<TW, TH, pixel>(rf, pd, src, rs, reinterpret_cast<int16_t *>(win_s + c * WIN_B), lane);
    };
#pragma unroll
    for (int c = 0; c < NT; c++)
        if (c < ntile) { const McTile t = record(c); fetch(t, t.r[0], c); }

    const int ib = 14 - (32 - __clz(bitdepth_max));       // intermediate_bits
    const int bias = 8192;                                 // PREP_BIAS
    // one prediction of tile c out of its window: the strips of this lane in q[]
    auto predict = [&](const McTile &t, const McRef &rf, const int c, const bool as_prep, const bool refetch, int (&q)[R][4]) {
        const McPred pd = mc_pred_of<TW, TH>(rf);
        const Taps fh = load_taps(rf.fh, rf.mx), fv = load_taps(rf.fv, rf.my);
        int16_t *const win = reinterpret_cast<int16_t *>(win_s + c * WIN_B);
        const pixel *src; int rs, rw, rh;
        plane_of(t, rf, src, rs, rw, rh);
        if (!mc_window_inside<TW, TH>(rf, pd, rw, rh)) {      // (edge emulation: the clamped gather, here and now)
            mc_gather<TW, TH, pixel, true>(rf, pd, src, rs, rw, rh, win, lane);
            dv::wave_sync();
        }
#ifndef DV_KO_HV
        mc_hpass<TW, TH, pixel, true>(pd, fh, win, mid, lane, ib, bias, as_prep);
#endif
        dv::wave_sync();
        if (refetch) fetch(t, t.r[1], c);                     // the window has been read: the second reference's takes its place
#ifndef DV_KO_HV
        mc_vpass<TW, TH, pixel, true>(pd, fv, mid, lane, ib, bias, as_prep, q);
#endif
    };
    auto pred_at = [&](const McTile &t, const int c, const int vr, const int vs) {
        return pred + (c >> TPB_LOG2) * (W * W) + (t.oy + vr) * W + t.ox + 4 * vs;
    };
    dv::glds_wait();
    dv::wave_sync();
    bool any_two = false;
    for (int c = 0; c < ntile; c++) {
        const McTile t = record(c);
#ifdef DV_KO_SECOND
        const bool two = false;
#else
        const bool two = t.kind == MCT_AVG || t.kind == MCT_WAVG;
#endif
        any_two |= two;
        int q[R][4];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int x = 0; x < 4; x++) q[r][x] = 0;
        predict(t, t.r[0], c, two, two, q);
        // single reference: the pixels; compound: the first reference's intermediate (16 bits as well), combined below
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int it = r * S::LPT + lane, vr = it / NS, vs = it % NS;
            int o[4];
#pragma unroll
            for (int x = 0; x < 4; x++) o[x] = two ? q[r][x] : dv::clamp3(q[r][x], 0, bitdepth_max);
            *reinterpret_cast<uint2 *>(pred_at(t, c, vr, vs)) = make_uint2(dv::pack2(o[0], o[1]), dv::pack2(o[2], o[3]));
        }
        dv::wave_sync();                                      // (the next tile's horizontal pass overwrites mid)
    }
    if (any_two) {
        dv::glds_wait();                                      // the second windows: issued one horizontal pass after their tile began
        dv::wave_sync();
        for (int c = 0; c < ntile; c++) {
            const McTile t = record(c);
            if (!(t.kind == MCT_AVG || t.kind == MCT_WAVG)) continue;
            int q[R][4];
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int x = 0; x < 4; x++) q[r][x] = 0;
            predict(t, t.r[1], c, true, false, q);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int it = r * S::LPT + lane, vr = it / NS, vs = it % NS;
                pixel *const d = pred_at(t, c, vr, vs);
                const uint2 a = *reinterpret_cast<const uint2 *>(d);
                const int a0[4] = { (int) (int16_t) (a.x & 0xffff), (int) (int16_t) (a.x >> 16), (int) (int16_t) (a.y & 0xffff), (int) (int16_t) (a.y >> 16) };
                int o[4];
                if (t.kind == MCT_AVG) {
#pragma unroll
                    for (int x = 0; x < 4; x++) o[x] = (a0[x] + q[r][x] + (1 << ib) + bias * 2) >> (ib + 1);          // avg_c
                } else {
#pragma unroll
                    for (int x = 0; x < 4; x++)
                        o[x] = dv::mad_i24(a0[x], t.weight, dv::mad_i24(q[r][x], 16 - t.weight, (8 << ib) + bias * 16)) >> (ib + 4);  // w_avg_c
                }
#pragma unroll
                for (int x = 0; x < 4; x++) o[x] = dv::clamp3(o[x], 0, bitdepth_max);
                *reinterpret_cast<uint2 *>(d) = make_uint2(dv::pack2(o[0], o[1]), dv::pack2(o[2], o[3]));
            }
            dv::wave_sync();
        }
    }
    dv::wave_sync();
    // ---- the residual, on coefficients that have been in registers since trip 2, and out
    uint32_t toff = 0;
    int tpl = 0;
    itx_body<TX, pixel, coef, true, true, true>(dst, tasks, n_blocks, cf, bitdepth_max, group, smem_itx, pred, false, &toff, &tpl, &pre);
    dv::wave_sync();
    tile_write_out<W, W, BPW, pixel>(pred, tasks + block0, nb, dst, twin, twin.data[0] != nullptr, twin.tiled != 2, toff, tpl);
}

template <int CLS, typename pixel, typename coef, bool TILED, bool WIDE, bool PIPED = false>
void launch_cls(const DevPlanes &dst, const RefSet &refs, const McTile *tiles, const Dav1dHipItxTask *tasks, const int n,
                int16_t *prep, coef *cf, const int bitdepth_max, const int coop_below, const DevPlanes &twin, hipStream_t stream)
{
    constexpr int W = 4 << CLS, LPB = cmax(cmin(W, 32), W), BPW = 64 / LPB;
    const int groups = (n + BPW - 1) / BPW;
    if constexpr (PIPED && TILED && WIDE && sizeof(pixel) == 2 && (CLS == 2 || CLS == 3)) {
        hipLaunchKernelGGL((recon_piped_kernel<CLS, pixel, coef>), dim3(groups), dim3(64), 0, stream, dst, refs, tiles, tasks, n, cf, bitdepth_max, twin);
        return;
    }
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

// the pipelined form of the sizes that have one (16x16, 32x32; tiled references, 16-bit pixels, wide stores)
template <typename pixel, typename coef>
hipError_t launch_piped(const int cls, const DevPlanes &dst, const RefSet &refs, const McTile *tiles, const Dav1dHipItxTask *tasks,
                        const int n, int16_t *prep, coef *cf, const int bitdepth_max, const int coop_below, const DevPlanes &twin, hipStream_t stream)
{
    if (cls == 2) launch_cls<2, pixel, coef, true, true, true>(dst, refs, tiles, tasks, n, prep, cf, bitdepth_max, coop_below, twin, stream);
    else launch_cls<3, pixel, coef, true, true, true>(dst, refs, tiles, tasks, n, prep, cf, bitdepth_max, coop_below, twin, stream);
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
    // wide bit 1: the pipelined form where one exists (recon_piped_kernel)
    if (bpc > 8 && tiled && (wide & 1) && (wide & 2) && (cls == 2 || cls == 3))
        e = launch_piped<uint16_t, int32_t>(cls, *dst, rs, tiles, tasks, n, prep, (int32_t *) coef, bitdepth_max, coop_below, twin, st);
    else if (bpc == 8) e = launch_variant<uint8_t, int16_t>(tiled, wide & 1, cls, *dst, rs, tiles, tasks, n, prep, (int16_t *) coef, bitdepth_max, coop_below, twin, st);
    else          e = launch_variant<uint16_t, int32_t>(tiled, wide & 1, cls, *dst, rs, tiles, tasks, n, prep, (int32_t *) coef, bitdepth_max, coop_below, twin, st);
    return hip_rc(e);
}

extern "C" int dav1d_hip_launch_recon_fused(const DevPlanes *dst, const DevPlanes *refs, int n_refs, int bpc, int cls, const McTile *tiles,
                                            const Dav1dHipItxTask *tasks, int n, int16_t *prep, void *coef, int coop_below, void *stream)
{
    return dav1d_hip_launch_recon_fused_out(dst, refs, n_refs, bpc, cls, tiles, tasks, n, prep, coef, coop_below, 0, nullptr, stream);
}

// The intra wavefront superblock by superblock (gfx950).
//
// What the reference does with an intra block — prepare_intra_edges + intra_pred, then itxfm_add on the same pixels, transform
// block after transform block in decode order (src/recon_tmpl.c:1207-1360) — has two scales of dependency.  INSIDE a superblock
// a transform block reads what its neighbours of the same superblock wrote a moment ago (a 64x64 superblock of 4x4 blocks is a
// chain of 46 such steps); BETWEEN superblocks only the left, top-left, top and top-right neighbours of the same tile are ever
// read (the reference decodes a tile's superblocks in raster order and edge availability stops at the tile,
// src/decode.c:2117-2375, src/ipred_prepare_tmpl.c:82-116).  The step-granular routes (a launch per step, intra_pair.hip; one
// launch with a counter hand-off per step, intra_flow.hip) pay a device-wide hand-off — a kernel boundary, or a coherent trip
// to memory and back — for every one of the several hundred steps of a key frame's tiles.
//
// Here ONE WORKGROUP owns a superblock: its waves work the superblock's units (prediction + residual of one transform block)
// off step by step with a workgroup barrier in between, the pixels handed over through the XCD's L2 (plain stores,
// acknowledged, then loads that bypass the CU's L1).  Superblocks are sorted into LEVELS — level(sb) = 1 + the highest level
// among those of its four neighbours whose intra pixels its blocks read (the lister records that per superblock) — 8 + 2 x 10
// levels for a tile of 8 x 10 superblocks instead of several hundred steps, all tiles side by side; an inter frame's scattered
// intra superblocks are a few levels.  The levels run as ONE launch: workgroups in level order, a superblock waiting for the
// flags of the neighbours it reads (release / acquire fences at superblock granularity) — or, on request, as a launch per level.
// Measured on MI355X: an 8K key frame 27.4 ms as one dataflow launch -> 12.5 ms as 12 level launches -> 10.3 ms as
// one launch of superblocks.
#define DV_UNIT intra_sb   // (names this unit's phase accessor in -DDV_PHASES variant builds, common.h)
#include "ipred_body.h"
#include "itx_body.h"
#include "capi.h"
#include <type_traits>
#include <algorithm>
#include <atomic>
#include <string.h>

// workgroups of four waves a CU is asked to hold (register budget 512 / (4 x this) per lane; the LDS of a workgroup allows three)
#ifndef SB_MIN_BLOCKS
#define SB_MIN_BLOCKS 2
#endif
#ifndef SB_MIN_WAVES_8
#define SB_MIN_WAVES_8 2      // (waves per SIMD asked of the register allocator for the eight- and one-wave forms; see SB_MIN_BLOCKS)
#endif
// the next unit's coefficients touched while the current one is worked on: 1 = at the start of the current unit (ahead of its own edge
// loads), 2 = behind its prediction, 0 = not at all
#ifndef SB_PREFETCH
#define SB_PREFETCH 1
#endif

DV_PHASE_DEFINE(DV_UNIT)
namespace {

constexpr int sb_itx_lds_of(int tx) {
    return (64 / cmax(cmin(tx_h(tx), 32), tx_w(tx))) * cmin(tx_h(tx), 32) * (tx_w(tx) + 1);
}
constexpr int sb_itx_lds_max(int tx = 0) { return tx == 19 ? 0 : cmax(sb_itx_lds_of(tx), sb_itx_lds_max(tx + 1)); }
enum { SB_SPIN_LIMIT = 1 << 18 };        // naps of ~3 µs before a waiting workgroup gives up (a bug, never a normal run)

// the unit the same wave works on after `u` (host: dav1d_hip_sbw_emit)
// (one wave per superblock: the records in their order — no chain to follow)
template <int NW> __device__ __forceinline__ uint32_t sb_next(const IntraUnit &u, const uint32_t at, const uint32_t n_units) {
    return NW == 1 ? (at + 1 < n_units ? at + 1 : SB_NONE) : NW == 4 ? u.pad2 : u.prev_n;
}

// records [r.first, r.first + r.n): the superblock's header, then its units sorted by (step, predictions first); grp = the unit's
// group, groups are separated by workgroup barriers.
template <typename pixel, typename coef, int SB_WAVES>
__global__ __launch_bounds__(SB_WAVES * 64, SB_WAVES == 4 ? SB_MIN_BLOCKS : SB_MIN_WAVES_8) void intra_sb_kernel(const DevPlanes dst, const IntraUnit *__restrict__ units,
                                                                     const SbRegion *__restrict__ regions, uint8_t *aux,
                                                                     const uint8_t *__restrict__ mask /* inter-intra masks (may be nullptr without such units) */,
                                                                     coef *__restrict__ cf, const int layout, const int bitdepth_max,
                                                                     uint32_t *flags /* ONE launch for every level: a word per superblock (+ the error word behind them), or nullptr */,
                                                                     const int n_regions, const uint32_t *__restrict__ where /* superblock (raster) -> its region; with flags */,
                                                                     const int sbw, const int sb_log2, const int fail_at /* test hook: this workgroup gives up at once; -1 */,
                                                                     uint8_t *done /* a byte per record, or nullptr: with flags, set behind every reconstructed unit; without, the units to skip */)
{
    __shared__ int16_t e1_s[SB_WAVES][ESZ], e2_s[SB_WAVES][ESZ];
    __shared__ int16_t blk_s[SB_WAVES][32 * 32];
    // the predicted tile and the transform's slabs share their LDS (itx_body.h, PRED_LDS)
    constexpr int TILE_B = 64 * 64 * (int) sizeof(pixel), ITX_B = sb_itx_lds_max() * 4;
    constexpr int SMEM_V = (cmax(TILE_B, ITX_B) + 15) / 16;
    __shared__ uint4 smem_s[SB_WAVES][SMEM_V];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    int16_t *const e1 = e1_s[wv], *const e2 = e2_s[wv], *const blk = blk_s[wv];
    pixel *const tile = reinterpret_cast<pixel *>(smem_s[wv]);
    int *const smem_itx = reinterpret_cast<int *>(smem_s[wv]);

    // phase slots of a wave of this kernel (DV_PHASES builds, tools/intra_phase_probe.py): 900 start-up (region, records), 901 waiting for
    // neighbouring superblocks, 902 prediction (edge gather, filter, predict, blend), 903 transform, 904 store, 905 end of a group (own stores
    // acknowledged + workgroup barrier + publication), 906 whole wave, 907 waves, 908 units, 909 groups, 910 idle turns (groups in which
    // the wave had no unit)
    DV_PHASE_BEGIN();
    const SbRegion r = regions[blockIdx.x];
    if (flags && (int) blockIdx.x == fail_at) {
        // (option intra_sb_fail_at: what a workgroup does whose wait ran out — its superblock untouched, its flag telling the others)
        if (threadIdx.x == 0) { atomicAdd(flags + n_regions, 1u); dv::st_coherent(flags + blockIdx.x, 2u); }
        return;
    }
    // FINE (one launch, every prediction stamped by the lister): nobody waits for whole superblocks.  flags[sb] = progress << 2 | state
    // (state 1 finished, 2 gave up; progress P = every unit of a step < P has its pixels out).  A unit whose prediction reads intra pixels
    // of other superblocks waits, itself, until those have published a progress above the step it needs; units on a superblock's right /
    // bottom border store coherently and the superblock publishes behind their groups.
    const bool fine = flags && reinterpret_cast<const uint32_t *>(units + r.first)[15] == 1u;
    // give_up: the whole-superblock form's verdict (set by thread 0 before a barrier: uniform).  give_up2[g & 1]: the per-block form's,
    // raised by a wave that waits in vain DURING group g and read by every wave behind the barrier that ends group g — a wave that is
    // already waiting in group g + 1 raises the OTHER word, so all waves of the workgroup take the same way out (with one
    // word a fast wave could raise it between a slow wave's barrier and its read)
    __shared__ int give_up;
    __shared__ int give_up2[2];
    if (flags && fine) {
        if (threadIdx.x == 0) { give_up = 0; give_up2[0] = give_up2[1] = 0; }
        __syncthreads();
    } else if (flags) {
        // Every level in one launch: the superblocks this one reads from (r.dep, all earlier in the array) have to be through.
        // Workgroups are dispatched in the order of their index, so whoever is waited for is running or done — no workgroup waits for
        // one that cannot start.  The neighbours' pixels were stored plainly and written back by the release fence at their end;
        // here one lane polls, then an acquire fence, and the edge reads are loads that bypass the L1 anyway.
        if (threadIdx.x == 0) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (r.dep[k] == SB_NONE || !ok) continue;
                int spins = 0;
                for (;;) {
                    const uint32_t v = dv::ld_coherent(flags + r.dep[k]) & 3u;
                    if (v == 1u) break;
                    if (v == 2u || ++spins > SB_SPIN_LIMIT) { ok = false; break; }       // a neighbour gave up, or never came: so does this one
                    dv::nap_long();
                }
            }
            if (!ok) {
                // never in a sound run.  The superblock is NOT reconstructed from neighbours that are not there; its flag says so, which
                // lets everything behind it leave at once instead of timing out one after the other, and the frame comes back -EIO
                atomicAdd(flags + n_regions, 1u);
                dv::st_coherent(flags + blockIdx.x, 2u);
            }
            give_up = !ok;
            give_up2[0] = give_up2[1] = 0;
            dv::fence_acquire_agent();
        }
        __syncthreads();
        if (give_up) return;
    }
    const IntraUnit *const ru = units + r.first, *const us = ru + 1;       // header record, then the units
    const uint32_t *const hdr = reinterpret_cast<const uint32_t *>(ru);
    const uint32_t n_groups = (uint32_t) __builtin_amdgcn_readfirstlane((int) hdr[0]);
    // this wave's chain of units (sb_next: the host linked them), two records ahead: while unit `u` is worked on, the record of the
    // next one (un) is in registers — its coefficients set off — and the one after that (u2) is on its way
    const uint32_t n_units = r.n - 1;
    uint32_t ci = SB_WAVES == 1 ? (n_units ? 0u : SB_NONE) : (uint32_t) __builtin_amdgcn_readfirstlane((int) hdr[(SB_WAVES == 4 ? 2 : 6) + wv]);
    IntraUnit u, un;
    if (ci != SB_NONE) u = us[ci];
    uint32_t ni = ci != SB_NONE ? (uint32_t) __builtin_amdgcn_readfirstlane((int) sb_next<SB_WAVES>(u, ci, n_units)) : SB_NONE;
    if (ni != SB_NONE) un = us[ni];
    DV_PHASE(900);
    for (uint32_t g = 0; g < n_groups; g++) {
        uint32_t pub = 0;                    // (wave 0 holds the first unit of every group: it knows what to publish behind it)
        int dv_units_ = 0;
        while (ci != SB_NONE && ((uint32_t) __builtin_amdgcn_readfirstlane((int) u.grp) & 0xffffu) == g) {
            dv_units_++;
            const IntraUnit *const up = us + ci;
            pub = (uint32_t) __builtin_amdgcn_readfirstlane((int) u.grp) >> 16;
            uint32_t n2 = SB_NONE;
            IntraUnit u2;
            int keepn = 0;
            if (ni != SB_NONE) {
                n2 = (uint32_t) __builtin_amdgcn_readfirstlane((int) sb_next<SB_WAVES>(un, ni, n_units));
                if (n2 != SB_NONE) u2 = us[n2];
                if (SB_PREFETCH == 1 && (un.has & 2)) {
                    const int nbn = ((int) un.t.rsv[0] | (int) un.t.rsv[1] << 8) * (int) sizeof(coef);
                    keepn = dv::fetch_begin(reinterpret_cast<const char *>(cf + un.t.cf_off) + (lane * 64 < nbn ? lane * 64 : 0));
                }
            }
            bool skip = false;
            if (fine && (u.has & 1) && u.p.kind != DAV1D_HIP_IPRED_PAL) {
                // the neighbouring superblocks this prediction reads intra pixels of: wait (one lane) until each has published a progress
                // above the step it needs, or has finished
                const uint32_t nmask = (uint32_t) __builtin_amdgcn_readfirstlane((int) u.p.pal[6]) & 15u;
                const uint32_t nstep = (uint32_t) __builtin_amdgcn_readfirstlane((int) u.p.pal[7]);
                if (nmask) {
                    int bad = 0;
                    if (lane == 0) {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (!(nmask >> k & 1) || r.dep[k] == SB_NONE || bad) continue;
                            int spins = 0;
                            for (;;) {
                                const uint32_t v = dv::ld_coherent(flags + r.dep[k]);
                                if ((v & 3u) == 1u || (v >> 2) > nstep) break;
                                if ((v & 3u) == 2u || ++spins > (SB_SPIN_LIMIT << 3) || ((volatile int *) give_up2)[0] || ((volatile int *) give_up2)[1]) { bad = 1; break; }
                                dv::nap();
                            }
                        }
                        if (bad) ((volatile int *) give_up2)[g & 1] = 1;
                    }
                    bad = __shfl(bad, 0);
                    skip = bad != 0;
                    dv::fence_acquire_agent();
                }
            }
            if (flags && where && (u.has & 1) && u.p.kind == DAV1D_HIP_IPRED_COPY && !skip) {
                // an intra block copy: the superblocks under its source window (any earlier ones of the tile, not just the four neighbours)
                // have to be THROUGH: a superblock writes only its border units through to memory as it goes, what a copy reads from its
                // inside is out of the XCD's L2 with the release at its end (progress beyond pal[7] would not be enough)
                const int pl = u.p.plane, ssh = pl && layout != DAV1D_HIP_LAYOUT_I444, ssv = pl && layout == DAV1D_HIP_LAYOUT_I420;
                const int iw = ((dst.w[0] + 7) & ~7) >> ssh, ih = ((dst.h[0] + 7) & ~7) >> ssv;
                const int sx = (int) (int16_t) u.p.pal[0], sy = (int) (int16_t) u.p.pal[1];
                const int bx0 = (dv::iclip(sx, 0, iw - 1) << ssh) >> sb_log2, bx1 = (dv::iclip(sx + u.p.tw * 4 - 1 + ((u.p.pal[2] & 15) != 0), 0, iw - 1) << ssh) >> sb_log2;
                const int by0 = (dv::iclip(sy, 0, ih - 1) << ssv) >> sb_log2, by1 = (dv::iclip(sy + u.p.th * 4 - 1 + ((u.p.pal[2] >> 8) != 0), 0, ih - 1) << ssv) >> sb_log2;
                int bad = 0;
                if (lane == 0) {
                    for (int k = 0; k < 4 && !bad; k++) {
                        if ((k & 1 && bx1 == bx0) || (k & 2 && by1 == by0)) continue;
                        const uint32_t at = where[(k & 2 ? by1 : by0) * sbw + (k & 1 ? bx1 : bx0)];
                        if (at == SB_NONE || at >= blockIdx.x) continue;        // (nothing written there by this pass; never itself or a later one)
                        int spins = 0;
                        for (;;) {
                            const uint32_t v = dv::ld_coherent(flags + at);
                            if ((v & 3u) == 1u) break;
                            if ((v & 3u) == 2u || ++spins > (SB_SPIN_LIMIT << 3) || ((volatile int *) give_up2)[0] || ((volatile int *) give_up2)[1]) { bad = 1; break; }
                            dv::nap();
                        }
                    }
                    if (bad) ((volatile int *) give_up2)[g & 1] = 1;
                }
                bad = __shfl(bad, 0);
                skip = bad != 0;
                dv::fence_acquire_agent();
            }
            DV_PHASE(901);
            // the launches per level that follow a one-launch pass in which workgroups gave up (frame.hip): what that pass finished stays
            if (!flags && done && done[r.first + 1 + ci]) skip = true;
            if (skip) {          // never in a sound run: the unit is NOT reconstructed from pixels that are not there
                dv::fetch_end(keepn);
                ci = ni; u = un; ni = n2; un = u2;
                continue;
            }
            const bool has_pred = u.has & 1, has_tx = u.has & 2;
            const int plane = has_pred ? u.p.plane : u.t.plane;
            const uint32_t dst_off = has_pred ? u.p.dst_off : u.t.dst_off;
            const int w = has_pred ? u.p.tw * 4 : tx_w(u.t.tx), h = has_pred ? u.p.th * 4 : tx_h(u.t.tx);
            pixel *const d = reinterpret_cast<pixel *>(dst.data[plane]) + dst_off;
            const int stride = dst.stride[plane];
            if (has_pred) {
                ipred_body<pixel, true>(dst, u.p, 0, false, aux, layout, bitdepth_max, e1, e2, blk, tile, w);
                if (u.has & 4) {
                    // the intra half of an inter-intra block (src/recon_tmpl.c:1606-1630, 1751-1784): blended into the inter prediction, which
                    // an earlier launch left in the picture (blend_c, src/mc_tmpl.c:682-695); the mask is that of the whole block
                    dv::wave_sync();
                    const uint8_t *const m = mask + u.p.aux_off;
                    for (int i = lane; i < w * h; i += 64) {
                        const int a = d[(i / w) * stride + (i % w)], b = tile[i], mm = m[i];
                        tile[i] = (pixel) ((a * (64 - mm) + b * mm + 32) >> 6);
                    }
                }
            } else {
                // a residual on its own (the blocks of a palette block, ...): the pixels it is added to come from the picture
                for (int i = lane; i < w * h; i += 64) tile[i] = dv::ld_coherent(d + (i / w) * stride + (i % w));
            }
            if (SB_PREFETCH == 2 && ni != SB_NONE && (un.has & 2)) {
                const int nbn = ((int) un.t.rsv[0] | (int) un.t.rsv[1] << 8) * (int) sizeof(coef);
                keepn = dv::fetch_begin(reinterpret_cast<const char *>(cf + un.t.cf_off) + (lane * 64 < nbn ? lane * 64 : 0));
            }
            dv::wave_sync();
            DV_PHASE(902);
            if (has_tx) {
#define CASE(T) case T: itx_body<T, pixel, coef, true, true>(dst, &up->t, 1, cf, bitdepth_max, 0, smem_itx, tile); break;
                switch (u.t.tx) {
                    CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9)
                    CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18)
                }
#undef CASE
            }
            dv::wave_sync();
            DV_PHASE(903);
            // the reconstructed tile leaves the LDS four pixels per store (blocks are at least four pixels wide and four-pixel
            // aligned); plain stores: the line stays in this XCD's L2, where the next group of the workgroup reads it
            {
                typedef typename std::conditional<sizeof(pixel) == 2, uint64_t, uint32_t>::type quad;
                const int wq = w >> 2;
                // (border units of the FINE form: agent-scope stores, written through to where the other XCDs' agent-scope loads look)
                const bool coh = fine && (u.has & 8);
                for (int i = lane; i < wq * h; i += 64) {
                    const int y = i / wq, x = (i - y * wq) * 4;
                    const quad v = *reinterpret_cast<const quad *>(tile + y * w + x);
                    if (coh) dv::st_coherent(reinterpret_cast<quad *>(d + y * stride + x), v);
                    else *reinterpret_cast<quad *>(d + y * stride + x) = v;
                }
            }
            // one launch for every level: the unit is marked, so that, should a workgroup of this launch give up waiting, the
            // launches per level that then finish the frame run exactly the units that are still to do (a residual is added once)
            if (flags && done && lane == 0) done[r.first + 1 + ci] = 1;
            dv::fetch_end(keepn);
            dv::wave_sync();                 // the LDS is free for the wave's next unit
            DV_PHASE(904);
            ci = ni; u = un; ni = n2; un = u2;
        }
        DV_PHASE_COUNT(908, dv_units_);
        DV_PHASE_COUNT(909, 1);
        DV_PHASE_COUNT(910, dv_units_ == 0);
        dv::stores_done();                   // this wave's pixels have reached the L2 (the coherent ones: memory) ...
        __syncthreads();                     // ... and so have the other waves': the next group may read them
        if (flags) {
            if (((volatile int *) give_up2)[g & 1]) {
                if (threadIdx.x == 0) { atomicAdd(flags + n_regions, 1u); dv::st_coherent(flags + blockIdx.x, 2u); }
                return;
            }
            if (fine && threadIdx.x == 0 && pub) dv::st_coherent(flags + blockIdx.x, pub << 2);
        }
        DV_PHASE(905);
    }
    DV_PHASE_WAVE(906);
    if (flags && threadIdx.x == 0) {
        dv::fence_release_agent();           // the XCD's L2 writes the superblock's pixels back: other XCDs may read them
        dv::st_coherent(flags + blockIdx.x, 0xfffffffdu);       // progress: everything; state: finished
    }
}

// ---- the same with the superblock's pixels RESIDENT IN LDS (4:2:0 / 4:0:0).
//
// The workgroup keeps an image of its superblock per plane: the pixels as they are in the picture when the launch starts (the
// inter blocks of the frame are there already), one column to the left, one row above that reaches a superblock's width further
// to the right (the top-right extension of the blocks on the superblock's top row).  Element (x, y) of the superblock, x in
// [-PAD, 2W), y in [-1, H), sits at (y + 1) * S + x + PAD with S = 2W + PAD (PAD = the pixels of a 16-byte chunk): rows start 16-byte aligned, the image is filled with
// 16-byte loads.  The prediction body (ipred_body.h) is handed a DevPlanes that describes the IMAGES (their generic addresses;
// dv::PxRead mode 2 turns them back into LDS reads) and a task whose offsets are the block's place in the image.  A finished
// unit goes to the image (for its neighbours) and to the picture (stores nobody waits for): a step of the superblock then
// costs LDS round trips and a workgroup barrier instead of trips to the L2 and back.  Measured: no faster than the L2 hand-off
// (what a unit costs is its own chain of LDS round trips and arithmetic, not the edge reads); an option, a launch per level.
template <typename pixel, int SBL2>
struct SbImage {
    static constexpr int W = 1 << SBL2, H = 1 << SBL2, CW = W >> 1, CH = H >> 1;
    static constexpr int PAD = 16 / (int) sizeof(pixel);                 // pixels left of the superblock: one 16-byte chunk (it holds column -1)
    static constexpr int SY = 2 * W + PAD, SC = 2 * CW + PAD;
    static constexpr int NY = (H + 1) * SY, NC = (CH + 1) * SC;
};

template <typename pixel, int SBL2>
__device__ __forceinline__ void sb_image_load(pixel *img, const int S, const int W, const int H, const pixel *plane, const int stride,
                                               const int px0, const int py0, const int rows_ok /* rows of the plane that exist */)
{
    constexpr int PPC = 16 / (int) sizeof(pixel);         // pixels per 16-byte chunk
    // rows -1 .. H - 1, chunks from x = -PPC (holds column -1) to 2W (a decoder only ever reads right of the superblock in the row above
    // it; the other rows are filled as well — pixels as they are when the launch starts): a chunk is loaded when it lies inside the
    // plane's allocation (x in [0, stride), y in [0, rows_ok)); what is not is never read by a block whose edge flags are right
    const int cpr = (2 * W) / PPC + 1;                     // chunks per row incl. the one left of the superblock
    for (int i = threadIdx.x; i < (H + 1) * cpr; i += blockDim.x) {
        const int r = i / cpr, c = i - r * cpr;
        const int y = r - 1, x = (c - 1) * PPC;
        const int gx = px0 + x, gy = py0 + y;
        if (gx < 0 || gy < 0 || gx + PPC > stride || gy >= rows_ok) continue;
        *reinterpret_cast<uint4 *>(img + (y + 1) * S + x + PPC) = *reinterpret_cast<const uint4 *>(plane + (size_t) gy * stride + gx);
    }
}

template <typename pixel, typename coef, int SBL2, int NW>
__global__ __launch_bounds__(NW * 64, SBL2 == 6 ? 2 : 1) void intra_sbl_kernel(const DevPlanes dst, const IntraUnit *__restrict__ units,
                                                            const SbRegion *__restrict__ regions, uint8_t *aux,
                                                            coef *__restrict__ cf, const int layout, const int bitdepth_max)
{
    typedef SbImage<pixel, SBL2> G;
    __shared__ int16_t e1_s[NW][ESZ], e2_s[NW][ESZ];
    __shared__ int16_t blk_s[NW][32 * 32];
    constexpr int TILE_B = 64 * 64 * (int) sizeof(pixel), ITX_B = sb_itx_lds_max() * 4;
    constexpr int SMEM_V = (cmax(TILE_B, ITX_B) + 15) / 16;
    __shared__ uint4 smem_s[NW][SMEM_V];
    __shared__ __attribute__((aligned(16))) pixel img_y[G::NY];
    __shared__ __attribute__((aligned(16))) pixel img_u[G::NC], img_v[G::NC];
    __shared__ DevPlanes img;                      // the images as the prediction body sees "the picture"
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    int16_t *const e1 = e1_s[wv], *const e2 = e2_s[wv], *const blk = blk_s[wv];
    pixel *const tile = reinterpret_cast<pixel *>(smem_s[wv]);
    int *const smem_itx = reinterpret_cast<int *>(smem_s[wv]);

    const SbRegion r = regions[blockIdx.x];
    const IntraUnit *const ru = units + r.first;
    const bool chroma = layout != DAV1D_HIP_LAYOUT_I400;
    const int x0 = r.x0, y0 = r.y0;
    // rows that exist: the picture's height rounded up to whole 8x8 blocks (what the frame's blocks may write)
    sb_image_load<pixel, SBL2>(img_y, G::SY, G::W, G::H, reinterpret_cast<const pixel *>(dst.data[0]), dst.stride[0], x0, y0, (dst.h[0] + 7) & ~7);
    if (chroma) {
        const int ch = ((dst.h[0] + 7) & ~7) >> 1;
        sb_image_load<pixel, SBL2>(img_u, G::SC, G::CW, G::CH, reinterpret_cast<const pixel *>(dst.data[1]), dst.stride[1], x0 >> 1, y0 >> 1, ch);
        sb_image_load<pixel, SBL2>(img_v, G::SC, G::CW, G::CH, reinterpret_cast<const pixel *>(dst.data[2]), dst.stride[2], x0 >> 1, y0 >> 1, ch);
    }
    if (threadIdx.x == 0) {
        img.data[0] = img_y; img.data[1] = img_u; img.data[2] = img_v;
        img.stride[0] = G::SY; img.stride[1] = img.stride[2] = G::SC;
        for (int p = 0; p < 3; p++) { img.w[p] = dst.w[p]; img.h[p] = dst.h[p]; }
        img.tiled = 0;
    }
    __syncthreads();

    const IntraUnit *const us = ru + 1;                                    // (ru[0] is the header record)
    const uint32_t *const hdr = reinterpret_cast<const uint32_t *>(ru);
    const uint32_t n_groups = (uint32_t) __builtin_amdgcn_readfirstlane((int) hdr[0]);
    // this wave's chain of units, two records ahead (see intra_sb_kernel)
    uint32_t ci = (uint32_t) __builtin_amdgcn_readfirstlane((int) hdr[(NW == 4 ? 2 : 6) + wv]);
    IntraUnit u, un;
    if (ci != SB_NONE) u = us[ci];
    uint32_t ni = ci != SB_NONE ? (uint32_t) __builtin_amdgcn_readfirstlane((int) sb_next<NW>(u, ci, 0u)) : SB_NONE;
    if (ni != SB_NONE) un = us[ni];
    for (uint32_t g = 0; g < n_groups; g++) {
        while (ci != SB_NONE && ((uint32_t) __builtin_amdgcn_readfirstlane((int) u.grp) & 0xffffu) == g) {
            const IntraUnit *const up = us + ci;
            uint32_t n2 = SB_NONE;
            IntraUnit u2;
            int keepn = 0;
            if (ni != SB_NONE) {
                n2 = (uint32_t) __builtin_amdgcn_readfirstlane((int) sb_next<NW>(un, ni, 0u));
                if (n2 != SB_NONE) u2 = us[n2];
                if (SB_PREFETCH == 1 && (un.has & 2)) {
                    const int nbn = ((int) un.t.rsv[0] | (int) un.t.rsv[1] << 8) * (int) sizeof(coef);
                    keepn = dv::fetch_begin(reinterpret_cast<const char *>(cf + un.t.cf_off) + (lane * 64 < nbn ? lane * 64 : 0));
                }
            }
            const bool has_pred = u.has & 1, has_tx = u.has & 2;
            const int plane = has_pred ? u.p.plane : u.t.plane;
            const uint32_t dst_off = has_pred ? u.p.dst_off : u.t.dst_off;
            const int w = has_pred ? u.p.tw * 4 : tx_w(u.t.tx), h = has_pred ? u.p.th * 4 : tx_h(u.t.tx);
            // the unit's place in its plane (host: need = y << 16 | x) and in the image of that plane
            const int ux = (int) (u.need & 0xffff), uy = (int) (u.need >> 16);
            const int lx = ux - (plane ? x0 >> 1 : x0), ly = uy - (plane ? y0 >> 1 : y0);
            const int S = plane ? G::SC : G::SY;
            pixel *const im = (plane == 0 ? img_y : plane == 1 ? img_u : img_v) + (ly + 1) * S + lx + G::PAD;
            if (has_pred) {
                Dav1dHipIpredTask tp = u.p;
                tp.dst_off = (uint32_t) ((ly + 1) * S + lx + G::PAD);
                if (tp.kind == DAV1D_HIP_IPRED_CFL)          // the co-located luma block (host/lister.c: the block's own luma origin)
                    tp.aux_off = (uint32_t) (((uy << 1) - y0 + 1) * G::SY + ((ux << 1) - x0) + G::PAD);
                ipred_body<pixel, 2>(img, tp, 0, false, aux, layout, bitdepth_max, e1, e2, blk, tile, w);
            } else {
                // a residual on its own (the blocks of a palette block, ...): the pixels it is added to come from the image
                for (int i = lane; i < w * h; i += 64) tile[i] = im[(i / w) * S + (i % w)];
            }
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
            // the reconstructed tile goes to the image and to the picture, four pixels per store
            {
                typedef typename std::conditional<sizeof(pixel) == 2, uint64_t, uint32_t>::type quad;
                pixel *const d = reinterpret_cast<pixel *>(plane == 0 ? dst.data[0] : plane == 1 ? dst.data[1] : dst.data[2]) + dst_off;
                const int stride = plane == 0 ? dst.stride[0] : plane == 1 ? dst.stride[1] : dst.stride[2];
                const int wq = w >> 2;
                for (int i = lane; i < wq * h; i += 64) {
                    const int y = i / wq, x = (i - y * wq) * 4;
                    const quad v = *reinterpret_cast<const quad *>(tile + y * w + x);
                    *reinterpret_cast<quad *>(im + y * S + x) = v;
                    *reinterpret_cast<quad *>(d + y * stride + x) = v;
                }
            }
            dv::fetch_end(keepn);
            dv::wave_sync();                 // the wave's LDS is free for its next unit
            ci = ni; u = un; ni = n2; un = u2;
        }
        __syncthreads();                     // the group's pixels are in the image
    }
}

} // namespace

// flags != nullptr: regions[0 .. n_regions) are EVERY level of the frame in level order and run as one launch (L2 hand-off form only;
// flags = n_regions + 1 zeroed words, the last one counts workgroups that gave up waiting).
// waves: workgroup size in waves: 1 (a superblock's units one after the other by ONE wave: no workgroup barrier, eight superblocks per CU in
// flight — for superblocks with a handful of units, the isolated intra blocks of an inter frame), 4 or 8 (0: the form's own choice).  lds: the LDS-resident form where it exists (4:2:0 / 4:0:0 pictures).
// option intra_sb_fail_at (process-wide; tests): the workgroup of this index gives up at once in the one-launch form, as if its wait had run out
static std::atomic<int> g_sbw_fail_at{-1};
void dav1d_hip_sbw_set_fail_at(int at) { g_sbw_fail_at.store(at); }

extern "C" int dav1d_hip_launch_intra_sb(const DevPlanes *dst, int bpc, int layout, const IntraUnit *units, const SbRegion *regions, int n_regions,
                                         uint8_t *aux, const uint8_t *mask, void *coef, int waves, int sb_log2, int lds, uint32_t *flags, void *stream,
                                         const uint32_t *where, int sbw, uint8_t *done)
{
    if (n_regions <= 0) return 0;
    const int bitdepth_max = (1 << bpc) - 1;
    if (lds && !flags && (layout == DAV1D_HIP_LAYOUT_I420 || layout == DAV1D_HIP_LAYOUT_I400) && (sb_log2 == 6 || sb_log2 == 7)) {
        // 128-pixel superblocks: the image leaves room for four waves' worth of scratch (160 KB of LDS per CU).  64-pixel ones: four
        // waves and two workgroups per CU (waves = 0 or 4: a superblock's steps are mostly narrower than four units, two superblocks
        // in flight keep the CU busier), or eight waves and one workgroup (waves = 8)
#define SBL_LAUNCH(P, Cf, L2, NW) hipLaunchKernelGGL((intra_sbl_kernel<P, Cf, L2, NW>), dim3(n_regions), dim3(NW * 64), 0, (hipStream_t) stream, *dst, units, \
                                                     regions, aux, (Cf *) coef, layout, bitdepth_max)
        if (bpc == 8) {
            if (sb_log2 == 7) SBL_LAUNCH(uint8_t, int16_t, 7, 4);
            else if (waves >= 8) SBL_LAUNCH(uint8_t, int16_t, 6, 8);
            else SBL_LAUNCH(uint8_t, int16_t, 6, 4);
        } else {
            if (sb_log2 == 7) SBL_LAUNCH(uint16_t, int32_t, 7, 4);
            else if (waves >= 8) SBL_LAUNCH(uint16_t, int32_t, 6, 8);
            else SBL_LAUNCH(uint16_t, int32_t, 6, 4);
        }
#undef SBL_LAUNCH
        return hip_rc(hipGetLastError());
    }
#define SB_LAUNCH(P, Cf, NW) hipLaunchKernelGGL((intra_sb_kernel<P, Cf, NW>), dim3(n_regions), dim3(NW * 64), 0, (hipStream_t) stream, *dst, units, regions, \
                                                aux, mask, (Cf *) coef, layout, bitdepth_max, flags, n_regions, where, sbw, sb_log2, flags ? g_sbw_fail_at.load() : -1, done)
    if (bpc == 8) { if (waves == 1) SB_LAUNCH(uint8_t, int16_t, 1); else if (waves == 4) SB_LAUNCH(uint8_t, int16_t, 4); else SB_LAUNCH(uint8_t, int16_t, 8); }
    else { if (waves == 1) SB_LAUNCH(uint16_t, int32_t, 1); else if (waves == 4) SB_LAUNCH(uint16_t, int32_t, 4); else SB_LAUNCH(uint16_t, int32_t, 8); }
#undef SB_LAUNCH
    return hip_rc(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------- host side

// Units sorted by step (dav1d_hip_intra_units_build: per step the units with a prediction [.. ua_end[s]), then the residuals on
// their own [.. ub_end[s])) -> one run of records per superblock met: a HEADER record, then the superblock's units sorted by (step,
// kind).  A unit carries the dense index of its group (grp) and the index of the unit the same wave works on next — units of a
// group are dealt to the waves round-robin — for workgroups of four (pad2) and eight waves (prev_n); the header (read as 32-bit
// words) holds the number of groups [0], 0 in the place of `has` [1] and the first unit of every wave, [2 .. 5] for four waves,
// [6 .. 13] for eight.  With the chains a wave has the record after next in flight and the coefficients of its next unit on their
// way while it works (intra_sb_kernel).  strides: of the picture's planes in pixels.  prepare() says how many records there will
// be (st.n_records = units + superblocks), emit() writes them to `out` (the frame's pinned unit arena, or any array).
int dav1d_hip_sbw_prepare(std::vector<IntraUnit> &units, const std::vector<uint32_t> &ua_end, const std::vector<uint32_t> &ub_end,
                          const SbTiling &tl, const int strides[3], const int ss_hor, const int ss_ver, SbSort &st)
{
    st.parts.clear(); st.pos.clear(); st.key.clear(); st.n_records = 0; st.copy_deps.clear();
    const size_t n = units.size();
    if (!n) return 0;
    if (tl.sb_log2 != 6 && tl.sb_log2 != 7) return -EINVAL;
    const int sbw = tl.sbw;
    std::vector<uint32_t> sb(n);
    st.key.resize(n); st.pos.resize(n);
    uint32_t lo = 0xffffffffu, hi = 0;
    size_t s = 0;
    for (size_t i = 0; i < n; i++) {
        while (s < ub_end.size() && i >= ub_end[s]) s++;
        if (s >= ub_end.size()) return -EINVAL;
        const IntraUnit &u = units[i];
        int x, y, pl;
        if (u.has & 1) { pl = u.p.plane; x = u.p.x4 * 4; y = u.p.y4 * 4; }
        else { pl = u.t.plane; if (pl > 2 || strides[pl] <= 0) return -EINVAL; x = (int) (u.t.dst_off % (uint32_t) strides[pl]); y = (int) (u.t.dst_off / (uint32_t) strides[pl]); }
        if (x > 0xffff || y > 0xffff) return -EINVAL;
        units[i].need = (uint32_t) y << 16 | (uint32_t) x;          // the unit's place in its plane (the LDS-resident kernel)
        const int uw = (u.has & 1) ? u.p.tw * 4 : tx_w(u.t.tx), uh = (u.has & 1) ? u.p.th * 4 : tx_h(u.t.tx);
        int x1 = x + uw, y1 = y + uh;
        if (pl) { x <<= ss_hor; y <<= ss_ver; x1 <<= ss_hor; y1 <<= ss_ver; }
        const int sx = x >> tl.sb_log2, sy = y >> tl.sb_log2;
        if (sx >= sbw || sy >= tl.sbh) return -EINVAL;
        // a unit that touches its superblock's last column or last row writes pixels the superblocks to the right / below may read:
        // its stores go out coherently and the superblock publishes its progress behind its group (intra_sb_kernel)
        if (x1 >= (sx + 1) << tl.sb_log2 || y1 >= (sy + 1) << tl.sb_log2) units[i].has |= 8; else units[i].has &= ~8u;
        sb[i] = (uint32_t) (sy * sbw + sx);
        if ((u.has & 1) && u.p.kind == DAV1D_HIP_IPRED_COPY) {
            // the superblocks under the copy's source window (the same arithmetic as the kernel's wait): whom this superblock's level
            // has to lie above
            const int ssh = pl ? ss_hor : 0, ssv = pl ? ss_ver : 0;
            const int iw = (tl.sbw << tl.sb_log2) >> ssh, ih = (tl.sbh << tl.sb_log2) >> ssv;      // (an upper bound of the coded area is enough here)
            const int cx = (int) (int16_t) u.p.pal[0], cy = (int) (int16_t) u.p.pal[1];
            const int bx0 = (std::min(std::max(cx, 0), iw - 1) << ssh) >> tl.sb_log2, bx1 = (std::min(std::max(cx + u.p.tw * 4 - 1 + ((u.p.pal[2] & 15) != 0), 0), iw - 1) << ssh) >> tl.sb_log2;
            const int by0 = (std::min(std::max(cy, 0), ih - 1) << ssv) >> tl.sb_log2, by1 = (std::min(std::max(cy + u.p.th * 4 - 1 + ((u.p.pal[2] >> 8) != 0), 0), ih - 1) << ssv) >> tl.sb_log2;
            for (int k = 0; k < 4; k++) {
                if ((k & 1 && bx1 == bx0) || (k & 2 && by1 == by0)) continue;
                const uint32_t src = (uint32_t) ((k & 2 ? by1 : by0) * sbw + (k & 1 ? bx1 : bx0));
                const uint64_t pr = (uint64_t) sb[i] << 32 | src;
                if (src != sb[i] && (st.copy_deps.empty() || st.copy_deps.back() != pr)) st.copy_deps.push_back(pr);
            }
        }
        st.key[i] = (uint32_t) s * 2 + (i >= ua_end[s]);
        lo = std::min(lo, sb[i]); hi = std::max(hi, sb[i]);
    }
    // stable counting sort by superblock (the units of a submission sit in a short run of superblock numbers); superblock j of
    // the submission starts j records later than its units alone would: its header and those of the superblocks before it
    const size_t span = (size_t) hi - lo + 1;
    std::vector<uint32_t> cnt(span + 1, 0);
    for (size_t i = 0; i < n; i++) cnt[sb[i] - lo + 1]++;
    for (size_t k = 0; k < span; k++) cnt[k + 1] += cnt[k];
    std::vector<uint32_t> hdrs(span, 0);
    for (size_t k = 0, j = 0; k < span; k++) {
        if (cnt[k] == cnt[k + 1]) continue;
        st.parts.push_back({ (uint32_t) (lo + k), (uint32_t) (cnt[k] + j), cnt[k + 1] - cnt[k] });
        hdrs[k] = (uint32_t) ++j;
    }
    {
        std::vector<uint32_t> at(cnt.begin(), cnt.end() - 1);
        for (size_t i = 0; i < n; i++) st.pos[i] = at[sb[i] - lo]++ + hdrs[sb[i] - lo];
    }
    st.n_records = n + st.parts.size();
    return 0;
}

// option intra_sb_fine (process-wide; A/B aid): 0 = superblocks wait for whole neighbouring superblocks even when the lister's stamps are there
static std::atomic<int> g_sbw_fine{1};
void dav1d_hip_sbw_set_fine(int on) { g_sbw_fine.store(on); }

void dav1d_hip_sbw_emit(const std::vector<IntraUnit> &units, const SbSort &st, IntraUnit *out)
{
    const size_t n = units.size();
    std::vector<uint32_t> okey(st.n_records, 0);
    for (size_t i = 0; i < n; i++) { out[st.pos[i]] = units[i]; okey[st.pos[i]] = st.key[i]; }
    for (const SbPart &p : st.parts) {
        IntraUnit *const hdr = out + p.first, *const us = hdr + 1;
        const uint32_t *const k = okey.data() + p.first + 1;
        memset(hdr, 0, sizeof(*hdr));
        uint32_t *const hw = reinterpret_cast<uint32_t *>(hdr);
        uint32_t last4[4], last8[8];
        for (int w = 0; w < 4; w++) { last4[w] = SB_NONE; hw[2 + w] = SB_NONE; }
        for (int w = 0; w < 8; w++) { last8[w] = SB_NONE; hw[6 + w] = SB_NONE; }
        uint32_t groups = 0;
        bool fine = true;           // every prediction of the superblock says where it reaches into other superblocks (the lister's stamp)
        for (uint32_t i = 0; i < p.n; i++)
            if ((us[i].has & 1) && us[i].p.kind != DAV1D_HIP_IPRED_PAL && !(us[i].p.pal[6] & 0x8000)) fine = false;
        for (uint32_t i = 0; i < p.n; groups++) {
            uint32_t j = i + 1;
            while (j < p.n && k[j] == k[i]) j++;
            // what the superblock publishes when this group is through: "every unit of a step below <the next group's step> is done" —
            // only behind groups that hold a unit on the superblock's right / bottom border (nobody reads the others from outside)
            bool border = false;
            for (uint32_t q = i; q < j; q++) border = border || (us[q].has & 8);
            const uint32_t pub = border && j < p.n ? std::min<uint32_t>(k[j] >> 1, 0xffffu) : 0;
            for (uint32_t q = i; q < j; q++) {
                us[q].grp = groups | pub << 16; us[q].prev_n = SB_NONE; us[q].pad2 = SB_NONE;
                const uint32_t w4 = (q - i) & 3, w8 = (q - i) & 7;
                if (last4[w4] == SB_NONE) hw[2 + w4] = q; else us[last4[w4]].pad2 = q;
                if (last8[w8] == SB_NONE) hw[6 + w8] = q; else us[last8[w8]].prev_n = q;
                last4[w4] = q; last8[w8] = q;
            }
            i = j;
        }
        hw[0] = groups;
        hw[15] = fine && groups < 0xffffu && g_sbw_fine.load() ? 1u : 0u;
    }
}

// level[k] of parts' superblocks: 0 where no neighbour (left, top-left, top, top-right, inside the same tile) holds units, else one
// more than the highest of theirs.  sbs: the superblock numbers (any order, each once); out: the level of each.  Returns the number
// of levels.
// dep: per superblock of the frame, which neighbours its blocks' edges reach into (bit 0 left, 1 top-left, 2 top, 3 top-right), or
// nullptr: all four.
int dav1d_hip_sbw_levels(const SbTiling &tl, const uint32_t *sbs, size_t n, const uint8_t *dep, std::vector<int> &level_of_sb, std::vector<uint32_t> &level,
                         const std::vector<uint64_t> *extra)
{
    const int sbw = tl.sbw, sbh = tl.sbh;
    level_of_sb.assign((size_t) sbw * sbh, -1);
    for (size_t i = 0; i < n; i++) {
        if (sbs[i] >= (uint32_t) (sbw * sbh)) return -EINVAL;
        level_of_sb[sbs[i]] = 0;
    }
    int top = 0;
    for (int tr = 0; tr < tl.n_rows; tr++)
        for (int tc = 0; tc < tl.n_cols; tc++) {
            const int x0 = tl.col_start[tc], x1 = std::min<int>(tl.col_start[tc + 1], sbw);
            const int y0 = tl.row_start[tr], y1 = std::min<int>(tl.row_start[tr + 1], sbh);
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) {
                    int &lv = level_of_sb[(size_t) y * sbw + x];
                    if (lv < 0) continue;
                    int m = -1;
                    const int dm = dep ? dep[(size_t) y * sbw + x] : 15;
                    if (x > x0 && (dm & 1)) m = std::max(m, level_of_sb[(size_t) y * sbw + x - 1]);
                    if (y > y0) {
                        if (dm & 4) m = std::max(m, level_of_sb[(size_t) (y - 1) * sbw + x]);
                        if (x > x0 && (dm & 2)) m = std::max(m, level_of_sb[(size_t) (y - 1) * sbw + x - 1]);
                        if (x + 1 < x1 && (dm & 8)) m = std::max(m, level_of_sb[(size_t) (y - 1) * sbw + x + 1]);
                    }
                    if (extra) {
                        // sources of intra block copies: earlier superblocks of the same tile (decode order), so their levels stand
                        const uint64_t me = (uint64_t) ((size_t) y * sbw + x) << 32;
                        for (auto it = std::lower_bound(extra->begin(), extra->end(), me); it != extra->end() && (*it >> 32) == (me >> 32); ++it) {
                            const uint32_t src = (uint32_t) *it;
                            if (src < (uint32_t) (sbw * sbh)) m = std::max(m, level_of_sb[src]);
                        }
                    }
                    lv = m + 1;
                    top = std::max(top, lv + 1);
                }
        }
    level.resize(n);
    for (size_t i = 0; i < n; i++) level[i] = (uint32_t) level_of_sb[sbs[i]];
    return top;
}

int dav1d_hip_sb_tiling_make(SbTiling *tl, int w, int h, int sb128, int n_cols, const uint16_t *col_start_sb, int n_rows, const uint16_t *row_start_sb)
{
    if (!tl || w < 1 || h < 1 || n_cols < 1 || n_cols > 64 || n_rows < 1 || n_rows > 64 || !col_start_sb || !row_start_sb) return -EINVAL;
    tl->sb_log2 = sb128 ? 7 : 6;
    tl->sbw = (w + (1 << tl->sb_log2) - 1) >> tl->sb_log2;
    tl->sbh = (h + (1 << tl->sb_log2) - 1) >> tl->sb_log2;
    tl->n_cols = n_cols; tl->n_rows = n_rows;
    for (int i = 0; i <= n_cols; i++) { if (i && col_start_sb[i] <= col_start_sb[i - 1]) return -EINVAL; tl->col_start[i] = col_start_sb[i]; }
    for (int i = 0; i <= n_rows; i++) { if (i && row_start_sb[i] <= row_start_sb[i - 1]) return -EINVAL; tl->row_start[i] = row_start_sb[i]; }
    if (tl->col_start[0] || tl->row_start[0] || tl->col_start[n_cols] < tl->sbw || tl->row_start[n_rows] < tl->sbh) return -EINVAL;
    return 0;
}

// The launches of a frame: every superblock that holds units, sorted by level.  A superblock met in two arrays (never by the
// listers: a superblock belongs to one tile-sbrow) is refused.
int dav1d_hip_sbw_plan(const SbTiling &tl, const std::vector<const std::vector<SbPart> *> &parts, const std::vector<size_t> &base, const uint8_t *dep,
                       SbPlan &plan, const std::vector<uint64_t> *extra)
{
    plan.regions.clear(); plan.level_start.clear();
    if (parts.size() != base.size()) return -EINVAL;
    std::vector<uint32_t> sbs;
    std::vector<SbRegion> reg;
    for (size_t k = 0; k < parts.size(); k++)
        for (const SbPart &p : *parts[k]) {
            if (base[k] + p.first + p.n + 1 > 0xffffffffu) return -ENOTSUP;
            sbs.push_back(p.sb);
            reg.push_back({ (uint32_t) (base[k] + p.first), p.n + 1, (uint16_t) ((p.sb % (uint32_t) tl.sbw) << tl.sb_log2), (uint16_t) ((p.sb / (uint32_t) tl.sbw) << tl.sb_log2), 0, { SB_NONE, SB_NONE, SB_NONE, SB_NONE } });
        }
    std::vector<int> level_of_sb;
    std::vector<uint32_t> level;
    {
        std::vector<uint32_t> seen(sbs);
        std::sort(seen.begin(), seen.end());
        if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) return -EINVAL;
    }
    const int nl = dav1d_hip_sbw_levels(tl, sbs.data(), sbs.size(), dep, level_of_sb, level, extra);
    if (nl < 0) return nl;
    plan.level_start.assign((size_t) nl + 1, 0);
    for (size_t i = 0; i < level.size(); i++) plan.level_start[level[i] + 1]++;
    for (int l = 0; l < nl; l++) plan.level_start[l + 1] += plan.level_start[l];
    plan.regions.resize(reg.size());
    // within a level the superblocks with the most units first: the launch ends with its longest workgroup, which should not be
    // the last one to start
    std::vector<uint32_t> order(reg.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t) i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return level[a] != level[b] ? level[a] < level[b] : reg[a].n > reg[b].n; });
    for (size_t i = 0; i < order.size(); i++) plan.regions[i] = reg[order[i]];
    // whom a superblock waits for when the levels run as ONE launch (intra_sb_kernel, flags): the positions, in this order, of the
    // neighbours that set its level — the same rule as dav1d_hip_sbw_levels, so each of them sits earlier in the array
    std::vector<uint32_t> &where = plan.where;
    where.assign((size_t) tl.sbw * tl.sbh, SB_NONE);
    for (size_t i = 0; i < order.size(); i++) where[sbs[order[i]]] = (uint32_t) i;
    if (extra)          // a copy's source has to sit earlier in the array (the wait inside one launch relies on it)
        for (const uint64_t pr : *extra) {
            const uint32_t a = where[pr >> 32], b = (uint32_t) pr < where.size() ? where[(uint32_t) pr] : SB_NONE;
            if (a != SB_NONE && b != SB_NONE && b >= a) return -EINVAL;
        }
    std::vector<int> tile_x0(tl.sbw, 0), tile_x1(tl.sbw, 0), tile_y0(tl.sbh, 0);
    for (int tc = 0; tc < tl.n_cols; tc++)
        for (int x = tl.col_start[tc]; x < std::min<int>(tl.col_start[tc + 1], tl.sbw); x++) { tile_x0[x] = tl.col_start[tc]; tile_x1[x] = std::min<int>(tl.col_start[tc + 1], tl.sbw); }
    for (int tr = 0; tr < tl.n_rows; tr++)
        for (int y = tl.row_start[tr]; y < std::min<int>(tl.row_start[tr + 1], tl.sbh); y++) tile_y0[y] = tl.row_start[tr];
    for (size_t i = 0; i < order.size(); i++) {
        SbRegion &r = plan.regions[i];
        const uint32_t sb = sbs[order[i]];
        const int x = (int) (sb % (uint32_t) tl.sbw), y = (int) (sb / (uint32_t) tl.sbw);
        const int dm = dep ? dep[sb] : 15;
        for (int k = 0; k < 4; k++) r.dep[k] = SB_NONE;
        if (x > tile_x0[x] && (dm & 1)) r.dep[0] = where[sb - 1];
        if (y > tile_y0[y]) {
            if (x > tile_x0[x] && (dm & 2)) r.dep[1] = where[sb - tl.sbw - 1];
            if (dm & 4) r.dep[2] = where[sb - tl.sbw];
            if (x + 1 < tile_x1[x] && (dm & 8)) r.dep[3] = where[sb - tl.sbw + 1];
        }
        for (int k = 0; k < 4; k++) if (r.dep[k] != SB_NONE && r.dep[k] >= i) return -EINVAL;      // (cannot happen: levels rise along every dependency)
    }
    return 0;
}

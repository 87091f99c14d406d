// The inverse-transform work of one wave (slab -> row pass -> transpose -> column pass -> add), shared by the kernels of
// itx.hip and by the fused prediction + residual kernels of recon.hip.  See itx.hip for the description of the mapping.
#pragma once
#include "common.h"
#include "itx1d.h"
#include "av1_scan_dev.h"
#include <type_traits>

namespace {

enum { K_DCT = 0, K_ADST = 1, K_IDENTITY = 2, K_FLIPADST = 3, K_WHT = 4 };

__host__ __device__ constexpr int tx_w(int tx) {
    constexpr int w[19] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
    return w[tx];
}
__host__ __device__ constexpr int tx_h(int tx) {
    constexpr int h[19] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };
    return h[tx];
}
// intermediate shift per size (reference src/itx_tmpl.c:160-178)
__host__ __device__ constexpr int tx_shift(int tx) {
    constexpr int s[19] = { 0, 1, 2, 2, 2, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2 };
    return s[tx];
}
__host__ __device__ constexpr int cmin(int a, int b) { return a < b ? a : b; }
__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }

// itxfm_add table index -> 1-D kind of the first (horizontal) and second (vertical)
// pass.  The reference's table entry [A_B] is the function whose internal type is
// B_A (src/itx_tmpl.c:233-262), and dav1d_tx1d_types[internal] = {first, second}
// (src/itx_1d.c:1043-1060): net effect first = B, second = A; V_x = {IDENTITY, x},
// H_x = {x, IDENTITY}.
__device__ __forceinline__ void txtp_kinds(const int txtp, int &first, int &second) {
    // packed 2 bits per entry: first | second << 2
    //            DCT_DCT ADST_DCT DCT_ADST ADST_ADST FLIPADST_DCT DCT_FLIPADST FLIPADST_FLIPADST ADST_FLIPADST
    // first        D       D        A        A         D            F            F                 F
    // second       D       A        D        A         F            D            F                 A
    //            FLIPADST_ADST IDTX V_DCT H_DCT V_ADST H_ADST V_FLIPADST H_FLIPADST
    // first        A             I    I     D     I      A      I          F
    // second       F             I    D     I     A      I      F          I
    // one 64-bit constant, four bits per entry: first | second << 2 (a chain of sixteen compares and selects costs a 4x4 wave a sixth of
    // its vector instructions)
    constexpr unsigned long long tab =
        (0ull | 0ull << 2) << 0 | (0ull | 1ull << 2) << 4 | (1ull | 0ull << 2) << 8 | (1ull | 1ull << 2) << 12 |
        (0ull | 3ull << 2) << 16 | (3ull | 0ull << 2) << 20 | (3ull | 3ull << 2) << 24 | (3ull | 1ull << 2) << 28 |
        (1ull | 3ull << 2) << 32 | (2ull | 2ull << 2) << 36 | (2ull | 0ull << 2) << 40 | (0ull | 2ull << 2) << 44 |
        (2ull | 1ull << 2) << 48 | (1ull | 2ull << 2) << 52 | (2ull | 3ull << 2) << 56 | (3ull | 2ull << 2) << 60;
    const unsigned v = txtp < 16 ? (unsigned) (tab >> (4 * txtp)) & 15u : 0u;
    first = v & 3;
    second = v >> 2;
}

// Runs the 1-D transform `kind` and hands the result to `done`.  Every kind finishes inside its own branch: merging
// the branches' output arrays instead makes the compiler keep part of them in scratch memory.
template <int N, typename F>
__device__ __forceinline__ void tx1d(const int kind, const int *in, const int lo, const int hi, F &&done) {
#ifdef DV_KO_TX
    if (lo != 12345) { done(in); return; }
#endif
    if constexpr (N == 64) {
        int out[N];
        itx1d::idct<64>(in, out, lo, hi);
        done(out);
    } else if constexpr (N == 32) {
        if (kind == K_IDENTITY) { int out[N]; itx1d::iidentity<32>(in, out); done(out); }
        else { int out[N]; itx1d::idct<32>(in, out, lo, hi); done(out); }
    } else {
        if (kind == K_DCT) { int out[N]; itx1d::idct<N>(in, out, lo, hi); done(out); }
        else if (kind == K_IDENTITY) { int out[N]; itx1d::iidentity<N>(in, out); done(out); }
        else {
            int out[N];
            if constexpr (N == 4) itx1d::iadst4(in, out);
            else itx1d::iadst<N>(in, out, lo, hi);
            done(out);
        }
    }
}

// Pixels from one block to the next in an LDS tile of W x H blocks ([block][row][W pixels]: the predicted blocks of the paired kernels, the
// finished ones on their way out): one row more than the block has.  A lane of the transform reads and writes COLUMNS of its block;
// with the blocks exactly W x H pixels apart — a multiple of the 32 banks for every size — the lanes of the 2 .. 16 blocks of a wave met
// on the same banks in every one of those accesses.
__host__ __device__ constexpr int itx_tile_stride(int w, int h) { return w * (h + 1); }

// LDS ints one wave needs for transform size TX (all of its blocks)
template <int TX>
constexpr int itx_lds_ints() {
    constexpr int W = tx_w(TX), H = tx_h(TX), SH = cmin(H, 32), LPB = cmax(SH, W);
    return (64 / LPB) * SH * (W + 1);
}

// What a wave can fetch for its transform blocks before it needs them (recon.hip's pipelined form issues these loads at the top of the
// wave, under the prediction of the same blocks): the task record, then the part of the coefficient slab the scan reached (or coeff[0]
// of a dc-only block).  itx_body<..., PRE = true> takes them from here instead of loading.
template <int TX, typename coef>
struct ItxPre {
    static constexpr int W = tx_w(TX), H = tx_h(TX), SW = cmin(W, 32), SH = cmin(H, 32), LPB = cmax(SH, W), BPW = 64 / LPB;
    static constexpr int NCH = SW * SH * (int) sizeof(coef) / 16, NV = (NCH + LPB - 1) / LPB;
    Dav1dHipItxTask t;
    int4 v[NV];
    int dc;
};
template <int TX, typename coef>
__device__ __forceinline__ void itx_prefetch_task(ItxPre<TX, coef> &pre, const Dav1dHipItxTask *__restrict__ tasks, const int n, const int group) {
    typedef ItxPre<TX, coef> P;
    const int lane = dv::lane_id();
    const int sub = P::BPW == 1 ? 0 : lane / P::LPB;
    const int ti = group * P::BPW + sub;
    const bool live = ti < n;
    if (P::BPW == 1) pre.t = tasks[__builtin_amdgcn_readfirstlane(live ? ti : 0)];
    else pre.t = tasks[live ? ti : 0];
}
template <int TX, typename coef>
__device__ __forceinline__ void itx_prefetch_coefs(ItxPre<TX, coef> &pre, const coef *__restrict__ cf, const int n, const int group) {
    typedef ItxPre<TX, coef> P;
    const int lane = dv::lane_id();
    const int sub = P::BPW == 1 ? 0 : lane / P::LPB, l = P::BPW == 1 ? lane : lane % P::LPB;
    const bool live = group * P::BPW + sub < n;
    const Dav1dHipItxTask &t = pre.t;
    const bool dconly = live && t.txtp == 0 && t.eob < 1;
    const bool full = live && !dconly;
    const coef *const gcf = cf + t.cf_off;
    const bool packed = t.flags & DAV1D_HIP_ITX_PACKED;
    const int nch = packed ? 0 : (((int) t.rsv[0] | ((int) t.rsv[1] << 8)) * (int) sizeof(coef) + 15) >> 4;
    const int4 *g4 = reinterpret_cast<const int4 *>(gcf);
    pre.dc = 0;
#pragma unroll
    for (int k = 0; k < P::NV; k++) {
        pre.v[k] = make_int4(0, 0, 0, 0);
        if (full && l + k * P::LPB < nch) pre.v[k] = g4[l + k * P::LPB];
    }
    if (dconly && l == 0) pre.dc = gcf[0];
}

// The wave `group` of the blocks of ONE transform size: blocks [group * BPW, group * BPW + BPW) of tasks[0 .. n).
// PRED_LDS (fused prediction + residual kernels): the pixels the residual is added to come from pred_s (block `sub` of the
// wave, W x H, row stride W) instead of the picture; the sum still goes to the picture.
// COH (with PRED_LDS): the result goes back to the LDS tile instead of the picture; the caller (intra_flow.hip) writes it out.
// task_off / task_plane (optional): this lane's copy of its block's dst_off / plane (lane b * LPB holds block b's), for tile_write_out.
// tsrc (with COH, without PRED_LDS): `dst` holds the planes of the picture's tiled twin (8x8 tiles of 64 consecutive pixels, mc_body.h)
// and the pixels the residual is added to are read from there — a frame whose pictures live in the twin only (DAV1D_HIP_TWIN_ONLY).
// PRE: the record and the coefficients were fetched ahead of time (`pre`, ItxPre above).
template <int TX, typename pixel, typename coef, bool PRED_LDS = false, bool COH = false, bool PRE = false>
__device__ __forceinline__ void itx_body(const DevPlanes &dst, const Dav1dHipItxTask *__restrict__ tasks,
                                         const int n, coef *__restrict__ cf, const int bitdepth_max, const int group, int *tmp_s,
                                         const pixel *pred_s = nullptr, const bool tsrc = false, uint32_t *task_off = nullptr, int *task_plane = nullptr,
                                         const ItxPre<TX, coef> *pre = nullptr)
{
    constexpr int W = tx_w(TX), H = tx_h(TX);
    constexpr int SW = cmin(W, 32), SH = cmin(H, 32);
    constexpr int LPB = cmax(SH, W);          // lanes per block
    constexpr int BPW = 64 / LPB;             // blocks per wave
    constexpr int TS = W + 1;                 // padded row stride of the transpose buffer
    constexpr int SHIFT = tx_shift(TX);
    constexpr bool RECT2 = (W * 2 == H) || (H * 2 == W);
    constexpr bool HBD = sizeof(pixel) == 2;

    constexpr int NCH = SW * SH * (int) sizeof(coef) / 16;   // 16-byte chunks per slab
    // one LDS region per block, used twice: first the raw slab (landing zone of the 16-byte
    // loads), then, once every lane holds its row in registers, the transposed intermediate
    static_assert(BPW * SH * TS == itx_lds_ints<TX>(), "LDS sizing");

    // phase slots of this transform size (DV_PHASES builds): 0 record + slab and destination loads landed, 1 rows in registers, 2 row pass,
    // 3 column pass + add + store, 4 whole body, 5 bodies counted
    constexpr int PH = 512 + TX * 16 + (PRED_LDS ? 8 : 0);
    DV_PHASE_BEGIN();
    const int lane = dv::lane_id();       // the body belongs to one wave (recon.hip runs several side by side in a workgroup)
    const int sub = BPW == 1 ? 0 : lane / LPB, l = BPW == 1 ? lane : lane % LPB;
    const int ti = group * BPW + sub;
    const bool live = ti < n;

    Dav1dHipItxTask t;
    if (PRE) t = pre->t;
    else if (BPW == 1) t = tasks[__builtin_amdgcn_readfirstlane(live ? ti : 0)];   // one block per wave: record in SGPRs
    else t = tasks[live ? ti : 0];

    // (tile_write_out wants the blocks' positions: the lanes that hold a block's record hand them over, no second trip to memory)
    if (task_off) *task_off = t.dst_off;
    if (task_plane) *task_plane = t.plane;
    const bool wht = TX == 0 && t.txtp == 16;
    const bool dconly = live && t.txtp == 0 && t.eob < 1;
    const bool full = live && !dconly;
    int k1 = 0, k2 = 0;
    txtp_kinds(t.txtp, k1, k2);

    coef *const gcf = cf + t.cf_off;
    int *const tmp = tmp_s + sub * SH * TS;
    pixel *const d = reinterpret_cast<pixel *>(dst.data[t.plane]) + t.dst_off + l;
    const int stride = dst.stride[t.plane];
    // tsrc: this lane's column of the block in the tiled plane — pixel (X, Y) lives at (Y & ~7) * stride + (X >> 3) * 64 + (Y & 7) * 8 + (X & 7)
    int tbx = 0, tby = 0;
    if (!PRED_LDS && COH && tsrc && live) dv::off_to_xy(t.dst_off, stride, tbx, tby);
    const pixel *const tcol = reinterpret_cast<const pixel *>(dst.data[t.plane]) + ((((tbx + l) >> 3) << 6) + ((tbx + l) & 7));
    auto load_dst = [&](pixel (&px)[H]) {
        if (!PRED_LDS && COH && tsrc) {
#pragma unroll
            for (int y = 0; y < H; y++) px[y] = tcol[dv::mul_i24((tby + y) & ~7, stride) + (((tby + y) & 7) << 3)];
        } else {
#pragma unroll
            for (int y = 0; y < H; y++) px[y] = d[y * stride];
        }
    };

    // ---- issue every global load up front: the row's coefficients (lane r = row r reads
    // coeff[r + x*SH], consecutive lanes -> consecutive addresses) and, for the column pass later,
    // this lane's column of destination pixels.  dc-only blocks touch coeff[0] only.
    int in[W];
    pixel dpx[H];
    int dc = 0;
    const bool row_lane = full && l < SH;
    if (PRED_LDS) {
        // the predicted pixels leave the LDS before anything of this body is stored there: recon.hip lets the two regions overlap
        if (live && l < W) {
#pragma unroll
            for (int y = 0; y < H; y++) dpx[y] = pred_s[sub * itx_tile_stride(W, H) + y * W + l];
        }
        dv::wave_sync();
    }
    static_assert(SW * SH * (int) sizeof(coef) <= SH * TS * (int) sizeof(int), "slab fits the transpose buffer");
    if (full) {
        // 16-byte loads of the contiguous slab, zeroed in the same sweep (src/itx_tmpl.c:108)
        const int4 *g4 = reinterpret_cast<const int4 *>(gcf);
        int4 *z4 = reinterpret_cast<int4 *>(gcf);
        int4 *s4 = reinterpret_cast<int4 *>(tmp);
        // only the part of the slab the scan can have reached (t.pad = prefix length in coefficients, filled in by the
        // host from the eob); the rest is zero in memory by contract and is zero-filled in LDS without being fetched
        const bool packed = t.flags & DAV1D_HIP_ITX_PACKED;
        const int nch = packed ? 0 : (((int) t.rsv[0] | ((int) t.rsv[1] << 8)) * (int) sizeof(coef) + 15) >> 4;
        int4 v[(NCH + LPB - 1) / LPB];
#pragma unroll
        for (int k = 0; k < (NCH + LPB - 1) / LPB; k++) {
            v[k] = make_int4(0, 0, 0, 0);
#ifdef DV_KO_COEF
            if (l + k * LPB < nch) v[k] = make_int4(l, k, 0, 0);
#else
            if constexpr (PRE) v[k] = pre->v[k];
            else if (l + k * LPB < nch) v[k] = g4[l + k * LPB];
#endif
        }
        if (!PRED_LDS && l < W) load_dst(dpx);
#pragma unroll
        for (int k = 0; k < (NCH + LPB - 1) / LPB; k++) {
            if (l + k * LPB < NCH) s4[l + k * LPB] = v[k];
#ifndef DV_KO_COEF
            if (l + k * LPB < nch) z4[l + k * LPB] = make_int4(0, 0, 0, 0);
#endif
        }
    } else if (dconly) {
        if (l == 0) { dc = PRE ? pre->dc : gcf[0]; if (!(t.flags & DAV1D_HIP_ITX_PACKED)) gcf[0] = 0; }            // src/itx_tmpl.c:59-60
        if (!PRED_LDS && l < W) load_dst(dpx);
    }
    dc = __shfl(dc, sub * LPB);
    dv::wave_sync();
    if (full && (t.flags & DAV1D_HIP_ITX_PACKED)) {
        // sparse wire format: eob + 1 values in decode order are scattered to their slab positions (the slab in LDS was
        // just zero-filled); positions as decode_coefs derives them (reference src/recon_tmpl.c:458-496, 548-575)
        coef *const slab = reinterpret_cast<coef *>(tmp);
        const int cls = (t.txtp == 11 || t.txtp == 13 || t.txtp == 15) ? 1 : (t.txtp == 10 || t.txtp == 12 || t.txtp == 14) ? 2 : 0;
        const uint16_t *const scan = av1_scans + av1_scan_off[TX];
        constexpr int LSW = SW == 4 ? 2 : SW == 8 ? 3 : SW == 16 ? 4 : 5;
        for (int i = l; i <= t.eob; i += LPB) {
            const int rc = cls == 0 ? (int) scan[i] : cls == 1 ? i : ((i & (SW - 1)) * SH + (i >> LSW));
            slab[rc] = gcf[i];
        }
    }
    dv::wave_sync();     // (every lane of the wave passes the two points above and this one: blocks of a wave differ in their paths)
    DV_PHASE(PH + 0);
    if (row_lane) {
        const coef *slab = reinterpret_cast<const coef *>(tmp);
#pragma unroll
        for (int x = 0; x < SW; x++) in[x] = slab[x * SH + l];
    }
    dv::wave_sync();     // every row is in registers: the region becomes the transpose buffer
    DV_PHASE(PH + 1);

    int row_min, row_max, col_min, col_max;
    if (HBD) {
        row_min = (int) ((unsigned) ~bitdepth_max << 7);
        col_min = (int) ((unsigned) ~bitdepth_max << 5);
    } else {
        row_min = col_min = -32768;
    }
    row_max = ~row_min;
    col_max = ~col_min;

    // ---- first pass: lane r = row r, W-point transform along x
    if (row_lane) {
        int out[W];
#pragma unroll
        for (int x = 0; x < W; x++) {
            int v = x < SW ? in[x] : 0;
            if (RECT2 && x < SW) v = dv::mad_i24k(v, 181, 128) >> 8;
            in[x] = v;
        }
        if (TX == 0 && wht) {
#pragma unroll
            for (int x = 0; x < W; x++) in[x] >>= 2;
            if constexpr (W == 4) itx1d::iwht4(in, out);
#pragma unroll
            for (int x = 0; x < W; x++) tmp[l * TS + x] = out[x];
        } else {
            const int rnd = (1 << SHIFT) >> 1;
            const bool flip = k1 == K_FLIPADST;
            tx1d<W>(k1, in, row_min, row_max, [&](const int *res) {
#pragma unroll
                for (int x = 0; x < W; x++) {
                    const int xo = flip ? W - 1 - x : x;
                    tmp[l * TS + xo] = dv::clamp3((res[x] + rnd) >> SHIFT, col_min, col_max);
                }
            });
        }
    }
    dv::wave_sync();
    DV_PHASE(PH + 2);

    // ---- second pass: lane c = column c, H-point transform along y, add to dst.  COH: the sums go back to the LDS tile the
    // prediction came from (the caller writes it out with wide coherent stores); the tile shares its memory with tmp, so every
    // lane has its column of tmp in registers before the first lane writes
    int cin[H];
    if (live && l < W && !dconly) {
#pragma unroll
        for (int y = 0; y < H; y++) cin[y] = y < SH ? tmp[y * TS + l] : 0;
    }
    if (COH) dv::wave_sync();
    pixel *const o = COH ? const_cast<pixel *>(pred_s) + sub * itx_tile_stride(W, H) + l : d;
    const int ostride = COH ? W : stride;
    if (live && l < W) {
        if (dconly) {
            if (RECT2) dc = (dc * 181 + 128) >> 8;
            dc = (dc * 181 + 128) >> 8;
            dc = (dc + ((1 << SHIFT) >> 1)) >> SHIFT;
            dc = (dc * 181 + 128 + 2048) >> 12;
#pragma unroll
            for (int y = 0; y < H; y++)
                o[y * ostride] = ((pixel) dv::clamp3((int) dpx[y] + dc, 0, bitdepth_max));
        } else {
            int out[H];
            if (TX == 0 && wht) {
                if constexpr (H == 4) itx1d::iwht4(cin, out);
#pragma unroll
                for (int y = 0; y < H; y++)
                    o[y * ostride] = ((pixel) dv::clamp3((int) dpx[y] + out[y], 0, bitdepth_max));
            } else {
                tx1d<H>(k2, cin, col_min, col_max, [&](const int *res) {
                    if (k2 == K_FLIPADST) {
#pragma unroll
                        for (int y = 0; y < H; y++)
                            o[y * ostride] = ((pixel) dv::clamp3((int) dpx[y] + ((res[H - 1 - y] + 8) >> 4), 0, bitdepth_max));
                    } else {
#pragma unroll
                        for (int y = 0; y < H; y++)
                            o[y * ostride] = ((pixel) dv::clamp3((int) dpx[y] + ((res[y] + 8) >> 4), 0, bitdepth_max));
                    }
                });
            }
        }
    }
    DV_PHASE(PH + 3);
    DV_PHASE_WAVE(PH + 4);
}


// ---- reconstructed blocks from an LDS tile to the picture in wide pieces (and, optionally, to the picture's tiled twin)
//
// The column pass leaves a lane with a COLUMN of its block: written from there, a W x H block costs H two-byte stores per lane.
// Through the tile (itx_body's COH form: the sums go back to LDS, [block][row][W pixels], blocks itx_tile_stride apart) every lane instead takes row pieces of
// up to 8 pixels — 16 bytes at 10 / 12 bits — and stores each once to the raster plane and, when the picture has a tiled twin
// (Dav1dHipPicture.twin: 8x8 tiles of 64 consecutive pixels, mc_body.h), once to the twin: a piece is a whole tile row there, an
// 8x8 block one 128-byte line.  raster = false: the twin only (twin.tiled == 2 at the kernels: the picture lives in its twin,
// DAV1D_HIP_TWIN_ONLY).

// task_off / task_plane: what itx_body handed back (lane b * (64 / BPW) holds block b's record: the positions come from there by wave
// shuffles — a fresh load of tasks[] here was a whole trip to memory at the end of every wave, 10 % of a paired 8x8 wave's cycles and a
// quarter of a 4x4 residual wave's: profiles/r05/phases.jsonl); without them (task_plane < 0) the records are read from tasks[].
template <int W, int H, int BPW, typename pixel>
__device__ __forceinline__ void tile_write_out(const pixel *tile, const Dav1dHipItxTask *__restrict__ tasks, const int nb,
                                               const DevPlanes &dst, const DevPlanes &twin, const bool has_twin, const bool raster = true,
                                               const uint32_t task_off = 0, const int task_plane = -1)
{
    constexpr int CP = W < 8 ? W : 8;                   // pixels per piece: inside one row of one 8x8 tile
    constexpr int CPR = W / CP, PER_BLOCK = H * CPR, NCHK = BPW * PER_BLOCK;
    constexpr int BYTES = CP * (int) sizeof(pixel);
    typedef typename std::conditional<BYTES == 16, uint4, typename std::conditional<BYTES == 8, uint2, uint32_t>::type>::type piece_t;
    static_assert(BYTES == 4 || BYTES == 8 || BYTES == 16, "piece size");
    const int lane = dv::lane_id();
    // Where a lane's pieces go: the position of their block.  With the records handed over by the transform body (task_plane >= 0: every lane
    // of a block's lane group holds the block's record) no LDS permute is needed where the lanes that store a block are the lanes that
    // transformed it (OWN: 4x4, 8x8), where a round of 64 pieces belongs to one block (ONE: 32x32) or to two (TWO: 16x16: two scalar reads and
    // a select) — a wave's last instructions are these stores, and the three permutes per round in front of them were a twentieth of a
    // 32x32 pair wave's life (profiles/r06: 69.0 -> 65.5 us).  Otherwise lane b works out block b's position and the others ask it.
    constexpr int LPBX = 64 / BPW;                                       // lanes of a block in the transform body
    constexpr bool OWN = BPW > 1 && PER_BLOCK == LPBX;                   // (then NCHK == 64: one round)
    constexpr bool ONE = BPW > 1 && PER_BLOCK % 64 == 0;
    constexpr bool TWO = BPW > 1 && PER_BLOCK == 32;
    const bool direct = task_plane >= 0 && (OWN || ONE || TWO);
    int bx = 0, by = 0, bpl = 0;
    if (direct) {
        if constexpr (OWN) {
            bpl = task_plane;
            dv::off_to_xy(task_off, bpl == 0 ? dst.stride[0] : bpl == 1 ? dst.stride[1] : dst.stride[2], bx, by);
        }
    } else if (task_plane >= 0) {
        // every lane of a block's group holds the block's record: lane b takes it from the first lane of group b
        const uint32_t off = BPW == 1 ? task_off : (uint32_t) __shfl((int) task_off, (lane & (BPW - 1)) * LPBX);
        bpl = BPW == 1 ? task_plane : __shfl(task_plane, (lane & (BPW - 1)) * LPBX);
        if (lane < nb) dv::off_to_xy(off, bpl == 0 ? dst.stride[0] : bpl == 1 ? dst.stride[1] : dst.stride[2], bx, by);
    } else if (lane < nb) {
        const Dav1dHipItxTask t = tasks[lane];
        bpl = t.plane;
        dv::off_to_xy(t.dst_off, bpl == 0 ? dst.stride[0] : bpl == 1 ? dst.stride[1] : dst.stride[2], bx, by);
    }
#pragma unroll
    for (int i0 = 0; i0 < NCHK; i0 += 64) {
        const int i = i0 + lane;
        const int b = BPW == 1 ? 0 : (i / PER_BLOCK) & (BPW - 1), rem = i - (i / PER_BLOCK) * PER_BLOCK;
        const int y = rem / CPR, c = rem - y * CPR;
        int x0, y0, pl;
        if constexpr (BPW == 1) { x0 = __builtin_amdgcn_readfirstlane(bx); y0 = __builtin_amdgcn_readfirstlane(by); pl = __builtin_amdgcn_readfirstlane(bpl); }
        else if (direct && OWN) { x0 = bx; y0 = by; pl = bpl; }
        else if (direct && ONE) {
            // (the record of the round's block, out of the first lane of its group in the transform body)
            const int bl = ((i0 / PER_BLOCK) & (BPW - 1)) * LPBX;
            pl = dv::readlane(task_plane, bl);
            dv::off_to_xy((uint32_t) dv::readlane((int) task_off, bl), pl == 0 ? dst.stride[0] : pl == 1 ? dst.stride[1] : dst.stride[2], x0, y0);
        } else if (direct && TWO) {
            const int bl = ((i0 / PER_BLOCK) & (BPW - 1)) * LPBX;      // the round's first block; lanes 32 .. 63 store the next one
            const bool hi = lane >= 32;
            pl = hi ? dv::readlane(task_plane, bl + LPBX) : dv::readlane(task_plane, bl);
            const uint32_t off = (uint32_t) (hi ? dv::readlane((int) task_off, bl + LPBX) : dv::readlane((int) task_off, bl));
            dv::off_to_xy(off, pl == 0 ? dst.stride[0] : pl == 1 ? dst.stride[1] : dst.stride[2], x0, y0);
        } else { x0 = __shfl(bx, b); y0 = __shfl(by, b); pl = __shfl(bpl, b); }
        if (i >= NCHK || i / PER_BLOCK >= nb) continue;
        const piece_t v = *reinterpret_cast<const piece_t *>(tile + b * itx_tile_stride(W, H) + y * W + c * CP);
#ifdef DV_KO_WRITE
        if (*reinterpret_cast<const uint32_t *>(&v) != 0xfeedbeefu) continue;
#endif
        const int X = x0 + c * CP, Y = y0 + y;
        const int stride = pl == 0 ? dst.stride[0] : pl == 1 ? dst.stride[1] : dst.stride[2];
        pixel *const base = reinterpret_cast<pixel *>(pl == 0 ? dst.data[0] : pl == 1 ? dst.data[1] : dst.data[2]);
        if (raster) *reinterpret_cast<piece_t *>(base + (dv::mul_i24(Y, stride) + X)) = v;
        if (has_twin) {      // (kernel arguments are never indexed by a run-time value nor have their address taken: either sends them to scratch memory)
            pixel *const tb = reinterpret_cast<pixel *>(pl == 0 ? twin.data[0] : pl == 1 ? twin.data[1] : twin.data[2]);
            *reinterpret_cast<piece_t *>(tb + (dv::mul_i24(Y & ~7, stride) + ((X >> 3) << 6) + ((Y & 7) << 3) + (X & 7))) = v;
        }
    }
}

} // namespace

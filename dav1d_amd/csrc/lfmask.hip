// Deblocking masks and the level cache on the device (gfx950).
//
// What dav1d_create_lf_mask_intra / _inter (reference src/lf_mask.c:259-383) leave behind block by block during pass 1 —
// Av1Filter.filter_y / filter_uv (which 4-pixel units of which column / row edge get which filter size), the level cache
// (uint8_t[4] per 4x4), noskip_mask (src/decode.c:1945-1956) and the transform-size contexts at tile edges that the sbrow
// drivers use for their fix-ups (lf.tx_lpf_right_edge, the above contexts; src/decode.c:2730-2740, src/lf_apply_tmpl.c:313-466)
// — is a function of the transform grid.  The host walk (host/lf_rects.c) turns the Av1Block array into rectangles, one per
// transform block; here
//   paint : one thread per rectangle writes its size classes, its edge flags and its levels into a per-4x4 cell map,
//   masks : one thread per (128x128 superblock, plane type, direction, line) reads a line of the map and assembles the mask
//           words — the size class along an edge is the smaller of the two cells meeting there, a cell at a tile's first
//           column / row meets the reset context (which never lowers its class, src/decode.c:2401-2402),
//   edges : the contexts at the right column of every tile column and the bottom row of every tile row.
// No word is written by two threads except noskip_mask (atomic OR).  HBM traffic: one byte per 4x4 cell and plane type
// written and read about twice, 4 bytes of levels per cell written: ~15 MB for an 8K frame.
#include "common.h"
#include "capi.h"
#include <string.h>
#include <vector>

namespace {

struct LfGeo {
    int w4, h4;                 // luma cells of the frame
    int cw4, ch4;               // chroma cells
    int mapw, maph;             // cell map pitch / rows (both plane types)
    int b4_stride;              // level cache pitch
    int sb128w, sb128h;
    int ss_hor, ss_ver, has_chroma;
    int n_tile_cols, n_tile_rows, align_h;
};

__global__ __launch_bounds__(256) void lf_paint_kernel(const Dav1dHipLfRect *__restrict__ rects, const int n, const LfGeo g,
                                                       uint8_t *__restrict__ map_y, uint8_t *__restrict__ map_c, uint8_t *__restrict__ level,
                                                       Dav1dHipAv1Filter *__restrict__ masks)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Dav1dHipLfRect r = rects[i];
    if (r.kind == DAV1D_HIP_LF_RECT_NOSKIP) {
        Dav1dHipAv1Filter *m = masks + (size_t) (r.y4 >> 5) * g.sb128w + (r.x4 >> 5);
        const unsigned bits = ((0xffffffffu >> (32 - r.w4)) << (r.x4 & 15)) & 0xffffu;
        const unsigned word = ((r.x4 & 16) ? bits << 16 : bits) | (r.w4 == 32 ? bits << 16 : 0u);
        uint32_t *rows = reinterpret_cast<uint32_t *>(&m->noskip_mask[0][0]);         // [16] rows of two 16-bit halves
        for (int y = 0; y < r.h4; y += 2) atomicOr(&rows[((r.y4 & 31) >> 1) + (y >> 1)], word);
        return;
    }
    const bool chroma = r.kind == DAV1D_HIP_LF_RECT_CHROMA;
    uint8_t *const map = chroma ? map_c : map_y;
    const int base = 0x80 | (r.cls & 15);
    for (int y = 0; y < r.h4; y++)
        for (int x = 0; x < r.w4; x++) {
            const size_t cell = (size_t) (r.y4 + y) * g.mapw + r.x4 + x;
            map[cell] = (uint8_t) (base | ((r.cls & 16) && !x ? 16 : 0) | ((r.cls & 32) && !y ? 32 : 0));
            uint8_t *lv = level + ((size_t) (r.y4 + y) * g.b4_stride + r.x4 + x) * 4 + (chroma ? 2 : 0);
            lv[0] = r.lvl[0]; lv[1] = r.lvl[1];
        }
}

// tile_x[x4] / tile_y[y4] (luma 4x4 units): 1 where a tile column / row starts
__global__ __launch_bounds__(128) void lf_masks_kernel(const LfGeo g, const uint8_t *__restrict__ map_y, const uint8_t *__restrict__ map_c,
                                                       const uint8_t *__restrict__ tile_x, const uint8_t *__restrict__ tile_y,
                                                       Dav1dHipAv1Filter *__restrict__ masks)
{
    const int sb = blockIdx.x, sbx = sb % g.sb128w, sby = sb / g.sb128w;
    const int tid = threadIdx.x, pt = tid >> 6, dir = (tid >> 5) & 1, line = tid & 31;
    if (pt && !g.has_chroma) return;
    const int ssx = pt ? g.ss_hor : 0, ssy = pt ? g.ss_ver : 0;
    const int cols = 32 >> ssx, rows = 32 >> ssy, pw4 = pt ? g.cw4 : g.w4, ph4 = pt ? g.ch4 : g.h4;
    const uint8_t *const map = pt ? map_c : map_y;
    unsigned word[3][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 } };
    if (line >= (dir ? rows : cols)) return;
    const int n = dir ? cols : rows;                   // units along the line
    const int half = 16 >> (dir ? ssx : ssy);          // units per 16-bit half (src/lf_mask.c:226-231)
    for (int k = 0; k < n; k++) {
        const int x = sbx * cols + (dir ? k : line), y = sby * rows + (dir ? line : k);
        if (x >= pw4 || y >= ph4) continue;
        const int c = map[(size_t) y * g.mapw + x];
        if (!(c & 0x80) || !(c & (dir ? 32 : 16))) continue;
        int cls = dir ? (c >> 2) & 3 : c & 3;
        const bool tile_start = dir ? tile_y[y << ssy] : tile_x[x << ssx];
        if (!tile_start) {
            const int o = dir ? map[(size_t) (y - 1) * g.mapw + x] : map[(size_t) y * g.mapw + x - 1];
            if (o & 0x80) cls = dv::imin(cls, dir ? (o >> 2) & 3 : o & 3);
        }
        const int s = k >= half;
        word[cls][s] |= 1u << (k - s * half);
    }
    Dav1dHipAv1Filter *m = masks + sb;
    if (!pt) {
#pragma unroll
        for (int c = 0; c < 3; c++) { m->filter_y[dir][line][c][0] = (uint16_t) word[c][0]; m->filter_y[dir][line][c][1] = (uint16_t) word[c][1]; }
    } else {
#pragma unroll
        for (int c = 0; c < 2; c++) { m->filter_uv[dir][line][c][0] = (uint16_t) word[c][0]; m->filter_uv[dir][line][c][1] = (uint16_t) word[c][1]; }
    }
}

struct LfTiles { uint16_t col_end4[64], row_end4[64]; };        // luma 4x4 units, clipped to the frame

__global__ __launch_bounds__(256) void lf_edges_kernel(const LfGeo g, const LfTiles t, const uint8_t *__restrict__ map_y,
                                                       const uint8_t *__restrict__ map_c, uint8_t *__restrict__ right_y, uint8_t *__restrict__ right_uv,
                                                       uint8_t *__restrict__ a_y, uint8_t *__restrict__ a_uv)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int what = blockIdx.y;            // 0: right edge luma, 1: right edge chroma, 2: above luma, 3: above chroma
    const int pt = what & 1;
    if (pt && !g.has_chroma) return;
    const int ssx = pt ? g.ss_hor : 0, ssy = pt ? g.ss_ver : 0;
    const uint8_t *const map = pt ? map_c : map_y;
    const int pw4 = pt ? g.cw4 : g.w4, ph4 = pt ? g.ch4 : g.h4;
    const int reset = pt ? 1 : 2;
    if (what < 2) {
        const int rows = g.align_h >> ssy, tc = i / rows, y = i % rows;
        if (tc >= g.n_tile_cols) return;
        int v = reset;
        const int x = (t.col_end4[tc] >> ssx) - 1;
        if (y < ph4 && x >= 0 && x < pw4) { const int c = map[(size_t) y * g.mapw + x]; if (c & 0x80) v = c & 3; }
        (pt ? right_uv : right_y)[(size_t) rows * tc + y] = (uint8_t) v;
    } else {
        const int cols = g.sb128w * (32 >> ssx), tr = i / cols, x = i % cols;
        if (tr >= g.n_tile_rows) return;
        int v = reset;
        const int y = (t.row_end4[tr] >> ssy) - 1;
        if (x < pw4 && y >= 0 && y < ph4) { const int c = map[(size_t) y * g.mapw + x]; if (c & 0x80) v = (c >> 2) & 3; }
        const int per = 32 >> ssx;
        (pt ? a_uv : a_y)[((size_t) tr * g.sb128w + x / per) * 32 + x % per] = (uint8_t) v;
    }
}

} // namespace

extern "C" {

// rects: HOST array from dav1d_hip_lf_rects().  masks_out: HOST, sb128w * sb128h entries (filter_y, filter_uv and noskip_mask
// are written, cdef_idx is zeroed: it comes from the bitstream).  level_dev: DEVICE, (sb128h * 32) * b4_stride * 4 bytes, every
// cell of a block gets its four levels.  right_edge[2]: HOST, align_h * n_tile_cols (luma) / (align_h >> ss_ver) * n_tile_cols
// bytes == f->lf.tx_lpf_right_edge.  a_y / a_uv: HOST, 32 bytes per (tile row, sb128 column) == the tx_lpf_y / tx_lpf_uv
// members of the pass-1 above contexts (a_stride 32 in Dav1dHipFilterDesc).
int dav1d_hip_lf_masks_build(Dav1dHipContext *c, const Dav1dHipFrameDesc *d, const Dav1dHipLfRect *rects, size_t n_rects,
                             Dav1dHipAv1Filter *masks_out, uint8_t *level_dev, uint8_t *right_edge[2], uint8_t *a_y, uint8_t *a_uv)
{
    if (!c || !d || (!rects && n_rects) || !masks_out || !level_dev || !right_edge || !a_y || !a_uv) return -EINVAL;
    if (d->n_tile_cols < 1 || d->n_tile_cols > 64 || d->n_tile_rows < 1 || d->n_tile_rows > 64 || n_rects > 0x7fffffff) return -EINVAL;
    LfGeo g;
    g.w4 = (d->w + 3) >> 2; g.h4 = (d->h + 3) >> 2;
    g.ss_hor = d->layout != DAV1D_HIP_LAYOUT_I444; g.ss_ver = d->layout == DAV1D_HIP_LAYOUT_I420;
    g.has_chroma = d->layout != DAV1D_HIP_LAYOUT_I400;
    g.cw4 = (g.w4 + g.ss_hor) >> g.ss_hor; g.ch4 = (g.h4 + g.ss_ver) >> g.ss_ver;
    const int bw = ((d->w + 7) >> 3) << 1, bh = ((d->h + 7) >> 3) << 1;
    g.sb128w = (bw + 31) >> 5; g.sb128h = (bh + 31) >> 5;
    g.mapw = g.sb128w * 32; g.maph = g.sb128h * 32;
    g.b4_stride = (int) d->b4_stride;
    g.n_tile_cols = d->n_tile_cols; g.n_tile_rows = d->n_tile_rows;
    g.align_h = (bh + 31) & ~31;
    if (g.b4_stride < g.mapw) return -EINVAL;
    const int sb4 = d->sb128 ? 32 : 16;
    LfTiles t;
    memset(&t, 0, sizeof(t));
    std::vector<uint8_t> tile_xy((size_t) g.mapw + g.maph, 0);
    for (int k = 0; k < d->n_tile_cols; k++) {
        const int s = d->col_start_sb[k] * sb4, e = d->col_start_sb[k + 1] * sb4;
        if (s < g.mapw) tile_xy[s] = 1;
        t.col_end4[k] = (uint16_t) (e < bw ? e : bw);
    }
    for (int k = 0; k < d->n_tile_rows; k++) {
        const int s = d->row_start_sb[k] * sb4, e = d->row_start_sb[k + 1] * sb4;
        if (s < g.maph) tile_xy[(size_t) g.mapw + s] = 1;
        t.row_end4[k] = (uint16_t) (e < bh ? e : bh);
    }
    const size_t n_sb = (size_t) g.sb128w * g.sb128h, map_b = (size_t) g.mapw * g.maph;
    const size_t right_b = (size_t) g.align_h * g.n_tile_cols, a_b = (size_t) g.n_tile_rows * g.sb128w * 32;
    // one allocation: rects | masks | maps | tile flags | edge outputs
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t) 255; return o; };
    const size_t o_rects = take(n_rects * sizeof(Dav1dHipLfRect)), o_masks = take(n_sb * sizeof(Dav1dHipAv1Filter)), o_my = take(map_b),
                 o_mc = take(map_b), o_tile = take(tile_xy.size()), o_ry = take(right_b), o_ruv = take(right_b), o_ay = take(a_b), o_auv = take(a_b);
    uint8_t *dev = nullptr;
    if (hipMalloc((void **) &dev, off + 256) != hipSuccess) return -ENOMEM;
    hipStream_t s = c->stream;
    int rc = 0;
    if (hipMemsetAsync(dev + o_masks, 0, off - o_masks, s) != hipSuccess) rc = -EIO;
    if (!rc && n_rects) rc = dav1d_hip_upload(c, dev + o_rects, rects, n_rects * sizeof(Dav1dHipLfRect));
    if (!rc) rc = dav1d_hip_upload(c, dev + o_tile, tile_xy.data(), tile_xy.size());
    Dav1dHipAv1Filter *d_masks = reinterpret_cast<Dav1dHipAv1Filter *>(dev + o_masks);
    if (!rc && n_rects) {
        hipLaunchKernelGGL(lf_paint_kernel, dim3((unsigned) ((n_rects + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const Dav1dHipLfRect *>(dev + o_rects), (int) n_rects, g, dev + o_my, dev + o_mc, level_dev, d_masks);
        rc = hip_rc(hipGetLastError());
    }
    if (!rc) {
        hipLaunchKernelGGL(lf_masks_kernel, dim3((unsigned) n_sb), dim3(128), 0, s, g, dev + o_my, dev + o_mc, dev + o_tile, dev + o_tile + g.mapw, d_masks);
        rc = hip_rc(hipGetLastError());
    }
    if (!rc) {
        const size_t most = right_b > a_b ? right_b : a_b;
        hipLaunchKernelGGL(lf_edges_kernel, dim3((unsigned) ((most + 255) / 256), 4), dim3(256), 0, s, g, t, dev + o_my, dev + o_mc,
                           dev + o_ry, dev + o_ruv, dev + o_ay, dev + o_auv);
        rc = hip_rc(hipGetLastError());
    }
    if (!rc) rc = dav1d_hip_download(c, masks_out, d_masks, n_sb * sizeof(Dav1dHipAv1Filter));
    if (!rc) rc = dav1d_hip_download(c, right_edge[0], dev + o_ry, right_b);
    if (!rc) rc = dav1d_hip_download(c, right_edge[1], dev + o_ruv, (size_t) (g.align_h >> g.ss_ver) * g.n_tile_cols);
    if (!rc) rc = dav1d_hip_download(c, a_y, dev + o_ay, a_b);
    if (!rc) rc = dav1d_hip_download(c, a_uv, dev + o_auv, a_b);
    (void) hipStreamSynchronize(s);
    (void) hipFree(dev);
    return rc;
}

} // extern "C"

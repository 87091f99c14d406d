// Per-tile-sbrow preparation of the reconstruction lists, done on the SUBMITTING thread.
//
// Round 1 built the device lists of a frame inside dav1d_hip_frame_end(): validation, fusing compound pairs, cutting blocks
// into tiles, binning, ordering, pairing, the dependency map and the upload, single-threaded, 0.16 us per task — 0.2 s for an
// 8K frame that the device reconstructs in 0.3 ms.  Here the same work happens per chunk (the tasks of one
// dav1d_hip_frame_submit_tile_sbrow() call) on whatever worker thread submits it — dav1d's pass-2 workers run tile-sbrows in
// parallel (src/thread_task.c:733-752) — the finished chunk goes to pinned memory and is uploaded at once on a copy stream,
// and frame_end() only has to (1) add up the per-bin counts, (2) launch one gather kernel that lines the chunks' segments up
// into contiguous per-bin arrays, (3) launch the frame.  Everything a chunk needs to know about other chunks is nothing: the
// tasks of a tile-sbrow write a rectangle of their own, so fusing, pairing and the prediction -> residual dependencies are
// local to it.
#include "capi.h"
#include "lists.h"
#include "chunk.h"
#include "av1_scan_prefix.h"
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <unordered_map>
#include <new>

uint64_t dav1d_hip_mc_geo_sig(const DevPlanes *rp, int n_refs);       // capi.hip: the key mc_regroup() remembers a grouping by

namespace {

// ------------------------------------------------------------------ pinned slabs, recycled through the context

// capacities are powers of two (>= 256 KB): one free list per size, so that a 256 KB chunk blob never walks off with the 64 MB twin
// of an arena (a fresh hipHostMalloc of that size is a millisecond-class stall) and a request costs no search — with best fit over
// ONE list, the 1,800 requests of an 8K frame each scanned the few thousand slabs of the pool under its lock
static int slab_class(size_t cap) { int k = 0; while (((size_t) 1 << k) < cap) k++; return k < 47 ? k : 47; }

uint8_t *slab_get(Dav1dHipContext *c, size_t bytes, size_t *cap) {
    size_t want = 1 << 18;
    while (want < bytes) want <<= 1;
    {
        std::lock_guard<std::mutex> lk(c->pool_mtx);
        // the exact class, else the next two up (a slab four times too large is still a better deal than a new allocation)
        for (int k = slab_class(want), e = std::min(k + 3, 48); k < e; k++) {
            std::vector<Dav1dHipContext::Slab> &v = c->free_slabs[k];
            if (v.empty()) continue;
            const Dav1dHipContext::Slab s = v.back();
            v.pop_back();
            *cap = s.cap;
            return s.host;
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, want, 0) != hipSuccess) return nullptr;
    *cap = want;
    return (uint8_t *) p;
}

void slab_put(Dav1dHipContext *c, uint8_t *host, size_t cap) {
    if (!host) return;
    std::lock_guard<std::mutex> lk(c->pool_mtx);
    c->free_slabs[slab_class(cap)].push_back({ host, cap });
}

// where a tile reads: (reference, plane, 64-row band, x) in 31 bits
inline uint32_t src_key(const McTile &t) {
    const McRef &r = t.r[0];
    const uint32_t y = (uint32_t) (r.src_y + 4096) & 0xffff, x = (uint32_t) (r.src_x + 4096) & 0xffff;
    return ((uint32_t) (r.ref & 7) << 28) | ((uint32_t) (t.plane & 3) << 26) | ((y >> 6) << 16) | x;
}

// v ordered by key[i] (ties: original order); key and position travel as one 64-bit word through the sort
template <typename T> void sort_by_key(std::vector<T> &v, const std::vector<uint32_t> &key) {
    const size_t n = v.size();
    if (n < 2) return;
    static thread_local std::vector<uint64_t> o;
    static thread_local std::vector<T> t;
    o.resize(n); t.resize(n);
    bool sorted = true;
    for (size_t i = 0; i < n; i++) { o[i] = (uint64_t) key[i] << 32 | (uint32_t) i; sorted &= !i || key[i] >= key[i - 1]; }
    if (sorted) return;
    std::sort(o.begin(), o.end());
    for (size_t i = 0; i < n; i++) t[i] = v[(uint32_t) o[i]];
    for (size_t i = 0; i < n; i++) v[i] = t[i];
}

// inside consecutive windows of `win` elements: stable counting sort by a one-byte key
template <typename T> void group_in_windows(std::vector<T> &v, const std::vector<uint8_t> &key, size_t win) {
    const size_t n = v.size();
    if (n < 2 || !win) return;
    static thread_local std::vector<T> t;
    t.resize(std::min(win, n));
    for (size_t lo = 0; lo < n; lo += win) {
        const size_t m = std::min(win, n - lo);
        uint32_t cnt[257] = { 0 };
        bool same = true;
        for (size_t i = 0; i < m; i++) { cnt[key[lo + i] + 1]++; same &= key[lo + i] == key[lo]; }
        if (same) continue;
        for (int k = 0; k < 256; k++) cnt[k + 1] += cnt[k];
        for (size_t i = 0; i < m; i++) t[cnt[key[lo + i]]++] = v[lo + i];
        for (size_t i = 0; i < m; i++) v[lo + i] = t[i];
    }
}

// pixel offset -> (x, y) of a plane: two hardware divisions per conversion were a quarter of the chunk preparation (a few
// conversions per task, 1.6 M tasks per 8K frame); the quotient from a double product is exact after one correction step
struct PlaneDiv {
    uint32_t d;
    double inv;
    void set(int stride) { d = (uint32_t) (stride > 0 ? stride : 1); inv = 1.0 / (double) d; }
    inline void xy(uint32_t off, int &x, int &y) const {
        uint32_t q = (uint32_t) ((double) off * inv);
        int64_t r = (int64_t) off - (int64_t) q * d;
        if (r < 0) { q--; r += d; } else if (r >= (int64_t) d) { q++; r -= d; }
        x = (int) r; y = (int) q;
    }
};

// open-addressing map uint32 -> uint32 for the PREP producers of a chunk (keys: arena offsets)
struct FlatMap {
    std::vector<uint32_t> k, v;
    uint32_t mask;
    void reset(size_t n) { size_t c = 16; while (c < 2 * n + 2) c <<= 1; k.assign(c, 0xffffffffu); v.assign(c, 0); mask = (uint32_t) c - 1; }
    uint32_t *slot(uint32_t key, bool insert) {
        uint32_t h = (key * 2654435761u) & mask;
        while (k[h] != 0xffffffffu && k[h] != key) h = (h + 1) & mask;
        if (k[h] == 0xffffffffu) { if (!insert) return nullptr; k[h] = key; }
        return &v[h];
    }
};

// a plane-local map of 4x4 cells over the bounding box of a chunk's destination rectangles
struct CellMap {
    int x0, y0, w, h, stride;       // in cells; stride of the PLANE in pixels
    PlaneDiv dv;
    std::vector<uint16_t> writers;  // bit b: the prediction launch of tile shape b writes the cell; bit 15: the compound / blend launch
    std::vector<uint8_t> blend;     // a blend task writes the cell
    std::vector<uint32_t> tx_at;    // 1 + index of the (pairable) square transform block whose top-left cell this is
    size_t cell_of(uint32_t off) const {
        int px, py;
        dv.xy(off, px, py);
        return (size_t) ((py >> 2) - y0) * w + ((px >> 2) - x0);
    }
    bool empty() const { return w <= 0 || h <= 0; }
    template <typename F> void each(uint32_t off, int bw, int bh, F f) {
        int px, py;
        dv.xy(off, px, py);
        for (int cy = py >> 2; cy <= (py + bh - 1) >> 2; cy++)
            for (int cx = px >> 2; cx <= (px + bw - 1) >> 2; cx++) {
                if (cx < x0 || cy < y0 || cx >= x0 + w || cy >= y0 + h) continue;
                f((size_t) (cy - y0) * w + (cx - x0));
            }
    }
};

// What a chunk preparation needs besides its output, kept per thread: dav1d's workers (and the library's pool threads) prepare
// a tile-sbrow after the other, and two dozen vectors allocated, faulted in page by page and handed back for each of the 1,800
// chunks of an 8K frame were a third of the preparation time.
struct ChunkScratch {
    CellMap cm[3];
    std::vector<char> taken, fused_prep;
    std::vector<McTile> p_tiles[5], bins[MC_BINS], pt_sorted[5];
    std::vector<uint32_t> p_itx[5], ord;
    std::vector<Dav1dHipCompTask> rest, c_first, c_second;
    std::vector<Dav1dHipItxTask> ibins[19], pk_sorted[5];
    std::vector<uint8_t> gk;
    std::vector<uint32_t> sk;
    FlatMap producer, readers;
};

} // namespace

void Dav1dHipChunk::release(Dav1dHipContext *c) {
    slab_put(c, host, cap);
    host = nullptr;
}


// ------------------------------------------------------------------ the library's own lister: no maps, no intermediate vectors
//
// dav1d_hip_chunk_build() below rediscovers, through maps of 4x4 cells and hash tables of arena offsets, what the walk that made the
// records knew when it made them: which transform block covers exactly one prediction, which two PREP blocks one average consumes,
// which launches write under a residual.  host/lister.c writes those three facts into the records (Walk.bdep / cand, lister.c) and the
// preparation becomes two passes over them: count, then cut the tiles straight into the pinned blob.  Measured on the C2 8K frame
// (923 K prediction, 106 K compound, 622 K transform records): 90 -> [see DESIGN.md 5] CPU-ms per frame.
namespace {
struct PairedBlock { uint32_t src, itx; };           // src: index of the PUT task, or 0x80000000 | index of the compound task
struct HintScratch {
    std::vector<uint8_t> taken, role, gk;
    std::vector<PairedBlock> pb[5], tmp;
};

inline int n_tiles(const int w, const int h) { return ((w + 63) >> 6) * ((h + 15) >> 4); }
inline int bin_of(const int w, const int h) { return tile_dim_class(w < 64 ? w : 64) * 3 + tile_dim_class(h < 16 ? h : 16); }

// mc_ref_of (capi.hip) in line: a million calls per 8K frame
inline McRef ref_of(const Dav1dHipMcTask &t) {
    McRef r;
    r.src_x = t.src_x; r.src_y = t.src_y;
    r.mx = t.mx; r.my = t.my; r.ref = t.ref; r.pad = 0;
    if (t.filter_2d == 9) {
        r.fh = r.fv = 6;
    } else {
        // enum Filter2d -> (h type, v type), 4-tap rows for w <= 4 / h <= 4 (see mc_ref_of)
        const int h_type = (0x15a80u >> (2 * t.filter_2d)) & 3, v_type = (0x24924u >> (2 * t.filter_2d)) & 3;
        r.fh = (uint8_t) (t.w > 4 ? h_type : 3 + (h_type & 1));
        r.fv = (uint8_t) (t.h > 4 ? v_type : 3 + (v_type & 1));
    }
    r.vspan = av1_mc_tap_span_host[r.fv * 16 + r.my];
    r.hspan = av1_mc_tap_span_host[r.fh * 16 + r.mx];
    return r;
}

inline McTile *cut_tiles(McTile *dst, const Dav1dHipMcTask &t, const int kind, const uint32_t dst_off, const Dav1dHipMcTask *second, const int weight) {
    McTile m;
    m.dst_off = dst_off;
    m.kind = (uint8_t) kind; m.plane = t.plane; m.bw = t.w; m.weight = (int8_t) weight;
    const McRef r0 = ref_of(t), r1 = second ? ref_of(*second) : r0;
    const int tw = t.w < 64 ? t.w : 64, th = t.h < 16 ? t.h : 16;
    for (int oy = 0; oy < t.h; oy += th)
        for (int ox = 0; ox < t.w; ox += tw) {
            m.w = (uint8_t) tw; m.h = (uint8_t) std::min(th, t.h - oy); m.ox = (uint8_t) ox; m.oy = (uint8_t) oy;
            m.r[0] = r0; m.r[0].src_x += ox; m.r[0].src_y += oy;
            m.r[1] = r1; m.r[1].src_x += ox; m.r[1].src_y += oy;
            *dst++ = m;
        }
    return dst;
}
} // namespace

static int chunk_build_hinted(Dav1dHipContext *c, Dav1dHipChunk **out, const Dav1dHipMcTask *mc, const size_t n_mc, const Dav1dHipCompTask *comp,
                              const size_t n_comp, const Dav1dHipItxTask *itx, const size_t n_itx, const uint16_t *itx_dep,
                              Dav1dHipChunkPlace place_blob, void *cookie)
{
    static thread_local HintScratch hs_tls;
    HintScratch &hs = hs_tls;                  // (one trip through the thread-local lookup, not one per use)
    const int fuse_mask = recon_fuse_mask(c);
    std::vector<uint8_t> &taken = hs.taken, &role = hs.role;
    taken.assign(n_itx, 0);
    role.assign(n_mc, 0);                      // 1: a PREP an averaged pair consumes (its tiles are cut with the pair); 2: a PUT that runs with
                                               // its transform block; 3: a PREP of an averaged pair that does
    for (int k = 0; k < 5; k++) hs.pb[k].clear();
    size_t cnt[CK_N] = { 0 };
    uint64_t order = ~0ull;
    int max_ref = 0;
    bool any_wmask = false;
    // the transform block a prediction runs in one wave with: named by the lister, taken when the size class is switched on
    auto partner = [&](const Dav1dHipMcTask &t, const int w, const int h, const uint32_t off) -> long {
        const uint32_t j = t.pad;
        if (!j || j > n_itx || w != h) return -1;
        const Dav1dHipItxTask &x = itx[j - 1];
        if (x.tx > 4 || (4 << x.tx) != w || x.dst_off != off || x.plane != t.plane || !(fuse_mask >> x.tx & 1) || taken[j - 1]) return -1;
        return (long) j - 1;
    };
    // ---- pass 1: what goes where
    for (size_t i = 0; i < n_comp; i++) {
        const Dav1dHipCompTask &k = comp[i];
        if (k.kind <= DAV1D_HIP_COMP_WAVG && k.mask_off) {
            const size_t a = (size_t) k.mask_off - 1;
            if (a + 1 >= n_mc || mc[a].kind != DAV1D_HIP_MC_PREP || mc[a + 1].kind != DAV1D_HIP_MC_PREP || mc[a].w != k.w || mc[a].h != k.h ||
                mc[a + 1].w != k.w || mc[a + 1].h != k.h || role[a] || role[a + 1]) return -EINVAL;
            role[a] = role[a + 1] = 1;
            const long j = partner(mc[a], k.w, k.h, k.dst_off);
            if (j >= 0) {
                taken[j] = 1;
                role[a] = role[a + 1] = 3;
                hs.pb[itx[j].tx].push_back({ 0x80000000u | (uint32_t) i, (uint32_t) j });
            } else {
                cnt[CK_MC + bin_of(k.w, k.h)] += (size_t) n_tiles(k.w, k.h);
            }
        } else {
            any_wmask |= k.kind == DAV1D_HIP_COMP_WMASK;
        }
    }
    for (size_t i = 0; i < n_mc; i++) {
        const Dav1dHipMcTask &t = mc[i];
        max_ref = std::max(max_ref, (int) t.ref);
        if (role[i]) continue;
        if (t.kind == DAV1D_HIP_MC_PUT) {
            order = std::min(order, (uint64_t) t.plane << 40 | t.dst_off);
            const long j = partner(t, t.w, t.h, t.dst_off);
            if (j >= 0) {
                taken[j] = 1;
                role[i] = 2;
                hs.pb[itx[j].tx].push_back({ (uint32_t) i, (uint32_t) j });
                continue;
            }
        }
        cnt[CK_MC + bin_of(t.w, t.h)] += (size_t) n_tiles(t.w, t.h);
    }
    Dav1dHipChunk *ck = new (std::nothrow) Dav1dHipChunk();
    if (!ck) return -ENOMEM;
    memset(ck->seg, 0, sizeof(ck->seg));
    memset(ck->dep, 0, sizeof(ck->dep));
    ck->host = nullptr; ck->cap = ck->used = 0; ck->dev_off = 0; ck->uploaded = false;
    for (size_t i = 0; i < n_itx; i++) {
        const Dav1dHipItxTask &t = itx[i];
        order = std::min(order, (uint64_t) t.plane << 40 | t.dst_off);
        if (taken[i]) continue;
        ck->dep[t.tx] |= itx_dep[i];
        cnt[CK_ITX + t.tx]++;
    }
    ck->wide_ok = true;         // the lister's blocks lie on AV1's grid: a transform block starts at a multiple of its width (or of 8)
    ck->order = order;
    ck->max_ref = max_ref;
    // compound / blend tasks that stay tasks: BLEND_V and the MASK tasks that read a mask a W_MASK task of the chunk writes go second
    std::unordered_map<uint32_t, char> wmask_out;
    if (any_wmask) for (size_t i = 0; i < n_comp; i++) if (comp[i].kind == DAV1D_HIP_COMP_WMASK) wmask_out[comp[i].mask_off] = 1;
    auto second = [&](const Dav1dHipCompTask &t) { return t.kind == DAV1D_HIP_COMP_BLEND_V || (t.kind == DAV1D_HIP_COMP_MASK && any_wmask && wmask_out.count(t.mask_off)); };
    for (size_t i = 0; i < n_comp; i++) {
        const Dav1dHipCompTask &k = comp[i];
        if (k.kind <= DAV1D_HIP_COMP_WAVG && k.mask_off) continue;
        cnt[CK_COMP + (second(k) ? 1 : 0)]++;
    }
    // paired blocks inside windows of 128 waves by code path (transform kinds, prediction kind, column parity of the source): the
    // blocks of a wave share their branch of recon_fused_kernel
    for (int k = 0; k < 5; k++) {
        std::vector<PairedBlock> &v = hs.pb[k];
        const int tpb = k < 3 ? 1 : k == 3 ? 2 : 4, bpw = k == 0 ? 16 : k == 1 ? 8 : k == 2 ? 4 : k == 3 ? 2 : 1;
        cnt[CK_PTILE + k] = v.size() * tpb;
        cnt[CK_PTASK + k] = v.size();
        if (v.size() < 2) continue;
        hs.gk.resize(v.size());
        for (size_t i = 0; i < v.size(); i++) {
            const bool two = v[i].src >> 31;
            const Dav1dHipCompTask *q = two ? &comp[v[i].src & 0x7fffffffu] : nullptr;
            const Dav1dHipMcTask &t0 = two ? mc[q->mask_off - 1] : mc[v[i].src];
            hs.gk[i] = (uint8_t) ((itx_path_key(itx[v[i].itx]) * 3 + (!two ? 0 : q->kind == DAV1D_HIP_COMP_AVG ? 1 : 2)) * 2 + (t0.src_x & 1));
        }
        group_in_windows(v, hs.gk, (size_t) 128 * bpw);
    }
    // ---- the blob: [mc bins][itx bins][paired tiles][paired residuals][comp first][comp second], 16-byte aligned segments
    size_t total = 0;
    auto place = [&](int id, size_t esz) { ck->seg[id].off = (uint32_t) total; ck->seg[id].n = (uint32_t) cnt[id]; total += (cnt[id] * esz + 15) & ~(size_t) 15; };
    for (int b = 0; b < MC_BINS; b++) place(CK_MC + b, sizeof(McTile));
    for (int b = 0; b < 19; b++) place(CK_ITX + b, sizeof(Dav1dHipItxTask));
    for (int k = 0; k < 5; k++) place(CK_PTILE + k, sizeof(McTile));
    for (int k = 0; k < 5; k++) place(CK_PTASK + k, sizeof(Dav1dHipItxTask));
    place(CK_COMP, sizeof(Dav1dHipCompTask));
    place(CK_COMP + 1, sizeof(Dav1dHipCompTask));
    ck->used = total;
    if (!total) { *out = ck; return 0; }
    uint8_t *dst = place_blob ? place_blob(cookie, total, &ck->dev_off) : nullptr;
    if (dst) {
        ck->uploaded = true;                 // part of the twin: goes up with it
    } else {
        dst = ck->host = slab_get(c, total, &ck->cap);
        if (!ck->host) { delete ck; return -ENOMEM; }
    }
    // ---- pass 2: cut and copy
    McTile *mcur[MC_BINS];
    for (int b = 0; b < MC_BINS; b++) mcur[b] = reinterpret_cast<McTile *>(dst + ck->seg[CK_MC + b].off);
    Dav1dHipCompTask *ccur[2] = { reinterpret_cast<Dav1dHipCompTask *>(dst + ck->seg[CK_COMP].off), reinterpret_cast<Dav1dHipCompTask *>(dst + ck->seg[CK_COMP + 1].off) };
    for (size_t i = 0; i < n_comp; i++) {
        const Dav1dHipCompTask &k = comp[i];
        if (k.kind <= DAV1D_HIP_COMP_WAVG && k.mask_off) {
            const size_t a = (size_t) k.mask_off - 1;
            if (role[a] == 3) continue;      // paired: below
            McTile *&p = mcur[bin_of(k.w, k.h)];
            p = cut_tiles(p, mc[a], k.kind == DAV1D_HIP_COMP_AVG ? MCT_AVG : MCT_WAVG, k.dst_off, &mc[a + 1], k.arg);
        } else {
            Dav1dHipCompTask *&p = ccur[second(k) ? 1 : 0];
            *p++ = k;
        }
    }
    for (size_t i = 0; i < n_mc; i++) {
        const Dav1dHipMcTask &t = mc[i];
        if (role[i]) continue;
        McTile *&p = mcur[bin_of(t.w, t.h)];
        p = cut_tiles(p, t, t.kind == DAV1D_HIP_MC_PUT ? MCT_PUT : t.kind == DAV1D_HIP_MC_PREP ? MCT_PREP : MCT_PUT_TMP, t.dst_off, nullptr, 0);
    }
    Dav1dHipItxTask *icur[19];
    for (int b = 0; b < 19; b++) icur[b] = reinterpret_cast<Dav1dHipItxTask *>(dst + ck->seg[CK_ITX + b].off);
    for (size_t i = 0; i < n_itx; i++)
        if (!taken[i]) {
            Dav1dHipItxTask *p = icur[itx[i].tx]++;
            *p = itx[i];
            itx_fill_prefix(*p);
        }
    for (int k = 0; k < 5; k++) {
        McTile *pt = reinterpret_cast<McTile *>(dst + ck->seg[CK_PTILE + k].off);
        Dav1dHipItxTask *pk = reinterpret_cast<Dav1dHipItxTask *>(dst + ck->seg[CK_PTASK + k].off);
        for (const PairedBlock &b : hs.pb[k]) {
            if (b.src >> 31) {
                const Dav1dHipCompTask &q = comp[b.src & 0x7fffffffu];
                const size_t a = (size_t) q.mask_off - 1;
                pt = cut_tiles(pt, mc[a], q.kind == DAV1D_HIP_COMP_AVG ? MCT_AVG : MCT_WAVG, q.dst_off, &mc[a + 1], q.arg);
            } else {
                pt = cut_tiles(pt, mc[b.src], MCT_PUT, mc[b.src].dst_off, nullptr, 0);
            }
            *pk = itx[b.itx];
            itx_fill_prefix(*pk);
            pk++;
        }
    }
    *out = ck;
    return 0;
}

// Everything dav1d_hip_recon_list_create() does for a whole frame, for the tasks of one tile-sbrow.
#ifdef CHUNK_PROF
#include <time.h>
#include <stdio.h>
static uint64_t cprof[16], cprof_n[4];
static uint64_t cnow() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (uint64_t) ts.tv_sec * 1000000000ull + ts.tv_nsec; }
#define P(i) { const uint64_t t_ = cnow(); __atomic_fetch_add(&cprof[i], t_ - tp_, __ATOMIC_RELAXED); tp_ = t_; }
extern "C" void dav1d_hip_chunk_prof() { for (int i = 0; i < 14; i++) { fprintf(stderr, "%d:%.1f ", i, cprof[i] * 1e-6); cprof[i] = 0; } fprintf(stderr, " mc %llu comp %llu itx %llu\n", (unsigned long long) cprof_n[0], (unsigned long long) cprof_n[1], (unsigned long long) cprof_n[2]); cprof_n[0] = cprof_n[1] = cprof_n[2] = 0; }
#else
#define P(i)
#endif
int dav1d_hip_chunk_build(Dav1dHipContext *c, Dav1dHipChunk **out, const Dav1dHipPicture *geom, const Dav1dHipPicture *refs, int n_refs,
                          const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                          const Dav1dHipItxTask *itx, size_t n_itx, Dav1dHipChunkPlace place_blob, void *cookie, bool trusted, const uint16_t *itx_dep)
{
    // records of the library's own lister carry what the maps below would find (chunk_build_hinted); option chunk_order = 1 and
    // chunk_hints = 0 keep the general preparation
    if (trusted && (itx_dep || !n_itx) && !c->chunk_order && c->chunk_hints) {
        return chunk_build_hinted(c, out, mc, n_mc, comp, n_comp, itx, n_itx, itx_dep, place_blob, cookie);
    }
#ifdef CHUNK_PROF
    uint64_t tp_ = cnow();
    cprof_n[0] += n_mc; cprof_n[1] += n_comp; cprof_n[2] += n_itx;
#endif
    *out = nullptr;
    const int bps = geom->bpc > 8 ? 2 : 1;
    int stride[3];
    for (int p = 0; p < 3; p++) stride[p] = geom->p[p].data ? (int) (geom->p[p].stride / bps) : 0;
    // ---- validation (what the list creators check); the library's own lister is not asked for its papers
    if (!trusted) {
        for (size_t i = 0; i < n_mc; i++) if (!mc_task_valid(mc[i]) || !stride[mc[i].plane] || (mc[i].kind != DAV1D_HIP_MC_PUT_TMP && mc[i].kind > DAV1D_HIP_MC_PREP)) return -EINVAL;
        for (size_t i = 0; i < n_comp; i++) {
            const Dav1dHipCompTask &t = comp[i];
            if (t.kind > 6 || t.plane > 2 || !stride[t.plane] || t.ss > 2 || t.w > 128 || t.h > 128 || t.w < 2 || t.h < 2 ||
                (t.kind <= 3 && (t.w < 4 || t.h < 4 || (t.w & 1) || (t.h & 1)))) return -EINVAL;
        }
        for (size_t i = 0; i < n_itx; i++) if (!itx_task_ok(itx[i]) || !stride[itx[i].plane]) return -EINVAL;
    }
    P(0);
    static const uint8_t tx_w[19] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
    static const uint8_t tx_h[19] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };

    // ---- bounding boxes of what the chunk writes, per plane, in 4x4 cells
    static thread_local ChunkScratch scr;
    CellMap (&cm)[3] = scr.cm;
    for (int p = 0; p < 3; p++) { cm[p].blend.clear(); cm[p].tx_at.clear(); cm[p].x0 = cm[p].y0 = 1 << 30; cm[p].w = cm[p].h = 0; cm[p].stride = stride[p]; cm[p].dv.set(stride[p]); }
    int x1[3] = { -1, -1, -1 }, y1[3] = { -1, -1, -1 };
    auto grow = [&](int p, uint32_t off, int bw, int bh) {
        int px, py;
        cm[p].dv.xy(off, px, py);
        cm[p].x0 = std::min(cm[p].x0, px >> 2); cm[p].y0 = std::min(cm[p].y0, py >> 2);
        x1[p] = std::max(x1[p], (px + bw - 1) >> 2); y1[p] = std::max(y1[p], (py + bh - 1) >> 2);
    };
    for (size_t i = 0; i < n_mc; i++) if (mc[i].kind == DAV1D_HIP_MC_PUT) grow(mc[i].plane, mc[i].dst_off, mc[i].w, mc[i].h);
    for (size_t i = 0; i < n_comp; i++) grow(comp[i].plane, comp[i].dst_off, comp[i].w, comp[i].h);
    for (size_t i = 0; i < n_itx; i++) grow(itx[i].plane, itx[i].dst_off, tx_w[itx[i].tx], tx_h[itx[i].tx]);
    bool any_blend = false;
    for (size_t i = 0; i < n_comp && !any_blend; i++) any_blend = comp[i].kind >= DAV1D_HIP_COMP_BLEND;
    for (int p = 0; p < 3; p++) {
        if (x1[p] < 0) { cm[p].w = cm[p].h = 0; continue; }
        cm[p].w = x1[p] - cm[p].x0 + 1; cm[p].h = y1[p] - cm[p].y0 + 1;
        cm[p].writers.assign((size_t) cm[p].w * cm[p].h, 0);
        if (any_blend) cm[p].blend.assign((size_t) cm[p].w * cm[p].h, 0);
    }
    if (any_blend)
        for (size_t i = 0; i < n_comp; i++)
            if (comp[i].kind >= DAV1D_HIP_COMP_BLEND) {
                CellMap &m = cm[comp[i].plane];
                m.each(comp[i].dst_off, comp[i].w, comp[i].h, [&](size_t k) { m.blend[k] = 1; });
            }

    P(1);
    // ---- pairing: a square transform block that covers exactly one prediction block runs with it in one wave (recon.hip)
    const int fuse_mask = recon_fuse_mask(c);
    std::vector<char> &taken = scr.taken;
    taken.assign(n_itx, 0);
    if (fuse_mask) {
        for (int p = 0; p < 3; p++) if (!cm[p].empty()) cm[p].tx_at.assign((size_t) cm[p].w * cm[p].h, 0);
        for (size_t i = 0; i < n_itx; i++)
            if (itx[i].tx <= 4 && (fuse_mask >> itx[i].tx & 1)) {
                CellMap &m = cm[itx[i].plane];
                int px, py;
                m.dv.xy(itx[i].dst_off, px, py);
                if (!((px | py) & 3)) m.tx_at[(size_t) ((py >> 2) - m.y0) * m.w + ((px >> 2) - m.x0)] = (uint32_t) i + 1;
            }
    }
    std::vector<McTile> (&p_tiles)[5] = scr.p_tiles;
    std::vector<uint32_t> (&p_itx)[5] = scr.p_itx;
    for (int k = 0; k < 5; k++) { p_tiles[k].clear(); p_itx[k].clear(); }
    auto find_pair = [&](int plane, uint32_t off, int bw, int bh) -> long {
        if (!fuse_mask || bw != bh || bw < 4) return -1;
        int px, py;
        cm[plane].dv.xy(off, px, py);
        if ((px | py) & 3) return -1;
        const uint32_t q = cm[plane].tx_at[(size_t) ((py >> 2) - cm[plane].y0) * cm[plane].w + ((px >> 2) - cm[plane].x0)];
        if (!q || taken[q - 1]) return -1;
        const Dav1dHipItxTask &t = itx[q - 1];
        if (!(t.tx <= 4 && (4 << t.tx) == bw) || t.dst_off != off) return -1;
        if (any_blend) {           // OBMC: prediction, blends, THEN the residual (src/recon_tmpl.c:1052-1112)
            bool hit = false;
            CellMap &m = cm[plane];
            m.each(off, bw, bh, [&](size_t k) { hit |= m.blend[k] != 0; });
            if (hit) return -1;
        }
        return (long) q - 1;
    };

    P(2);
    // ---- compound pairs whose two PREP blocks nothing else reads are predicted twice and combined in registers
    size_t n_prep = 0;
    for (size_t i = 0; i < n_mc; i++) n_prep += mc[i].kind == DAV1D_HIP_MC_PREP;
    FlatMap &producer = scr.producer, &readers = scr.readers;      // arena offset -> 1 + task index / number of readers
    producer.reset(n_comp ? n_prep : 0); readers.reset(n_comp ? 2 * n_comp : 0);
    if (n_comp) {
        for (size_t i = 0; i < n_mc; i++) if (mc[i].kind == DAV1D_HIP_MC_PREP) *producer.slot(mc[i].dst_off, true) = (uint32_t) i + 1;
        for (size_t i = 0; i < n_comp; i++)
            if (comp[i].kind <= DAV1D_HIP_COMP_WMASK) { ++*readers.slot(comp[i].tmp1_off, true); ++*readers.slot(comp[i].tmp2_off, true); }
    }
    P(12);
    std::vector<char> &fused_prep = scr.fused_prep;
    fused_prep.assign(n_mc, 0);
    std::vector<Dav1dHipCompTask> &rest = scr.rest;
    rest.clear();
    std::vector<McTile> (&bins)[MC_BINS] = scr.bins;
    for (int b = 0; b < MC_BINS; b++) bins[b].clear();
    P(13);
    {
        size_t est[MC_BINS] = { 0 };
        for (size_t i = 0; i < n_mc; i++) {
            const int tw = mc[i].w < 64 ? mc[i].w : 64, th = mc[i].h < 16 ? mc[i].h : 16;
            est[tile_dim_class(tw) * 3 + tile_dim_class(th)] += (size_t) ((mc[i].w + 63) >> 6) * ((mc[i].h + 15) >> 4);      // a reserve() hint
        }
        for (int b = 0; b < MC_BINS; b++) bins[b].reserve(est[b]);
        rest.reserve(n_comp / 4 + 16);
    }
    P(10);
    for (size_t i = 0; i < n_comp; i++) {
        const Dav1dHipCompTask &k = comp[i];
        bool fuse = k.kind == DAV1D_HIP_COMP_AVG || k.kind == DAV1D_HIP_COMP_WAVG;
        uint32_t a = 0, b = 0;
        if (fuse) {
            const uint32_t *pa = producer.slot(k.tmp1_off, false), *pb = producer.slot(k.tmp2_off, false);
            fuse = pa && pb && *pa && *pb && k.tmp1_off != k.tmp2_off && *readers.slot(k.tmp1_off, false) == 1 && *readers.slot(k.tmp2_off, false) == 1;
            if (fuse) {
                a = *pa - 1; b = *pb - 1;
                fuse = mc[a].w == k.w && mc[a].h == k.h && mc[b].w == k.w && mc[b].h == k.h && mc[a].plane == k.plane && mc[b].plane == k.plane;
            }
        }
        if (fuse) {
            const long j = find_pair(k.plane, k.dst_off, k.w, k.h);
            if (j >= 0) { taken[j] = 1; p_itx[itx[j].tx].push_back((uint32_t) j); }
            push_tiles(bins, mc[a], k.kind == DAV1D_HIP_COMP_AVG ? MCT_AVG : MCT_WAVG, k.dst_off, &mc[b], k.arg, j >= 0 ? &p_tiles[itx[j].tx] : nullptr);
            fused_prep[a] = fused_prep[b] = 1;
        } else {
            rest.push_back(k);
        }
    }
    P(11);
    for (size_t i = 0; i < n_mc; i++)
        if (!fused_prep[i]) {
            const long j = mc[i].kind == DAV1D_HIP_MC_PUT ? find_pair(mc[i].plane, mc[i].dst_off, mc[i].w, mc[i].h) : -1;
            if (j >= 0) { taken[j] = 1; p_itx[itx[j].tx].push_back((uint32_t) j); }
            push_tiles(bins, mc[i], mc[i].kind == DAV1D_HIP_MC_PUT ? MCT_PUT : mc[i].kind == DAV1D_HIP_MC_PREP ? MCT_PREP : MCT_PUT_TMP,
                       mc[i].dst_off, nullptr, 0, j >= 0 ? &p_tiles[itx[j].tx] : nullptr);
        }

    P(3);
    Dav1dHipChunk *ck = new (std::nothrow) Dav1dHipChunk();
    if (!ck) return -ENOMEM;
    memset(ck->seg, 0, sizeof(ck->seg));
    memset(ck->dep, 0, sizeof(ck->dep));
    ck->max_ref = 0; ck->host = nullptr; ck->cap = ck->used = 0; ck->dev_off = 0; ck->uploaded = false;
    ck->order = ~0ull;

    // ---- who writes which cell, then: which prediction launches each residual size has to wait for
    for (int b = 0; b < MC_BINS; b++)
        for (const McTile &t : bins[b])
            if (t.kind == MCT_PUT || t.kind == MCT_AVG || t.kind == MCT_WAVG) {
                CellMap &m = cm[t.plane];
                m.each(t.dst_off + (uint32_t) t.oy * (uint32_t) stride[t.plane] + t.ox, t.w, t.h, [&](size_t k) { m.writers[k] |= (uint16_t) (1u << b); });
            }
    for (const Dav1dHipCompTask &k : rest) { CellMap &m = cm[k.plane]; m.each(k.dst_off, k.w, k.h, [&](size_t q) { m.writers[q] |= 1u << 15; }); }
    std::vector<Dav1dHipItxTask> (&ibins)[19] = scr.ibins;
    for (int b = 0; b < 19; b++) ibins[b].clear();
    ck->wide_ok = true;
    for (size_t i = 0; i < n_itx && ck->wide_ok; i++)      // (as dav1d_hip_recon_list_create: may the blocks leave in row pieces of up to 8 pixels?)
        ck->wide_ok = stride[itx[i].plane] > 0 && (int) (itx[i].dst_off % (uint32_t) stride[itx[i].plane]) % std::min((int) tx_w[itx[i].tx], 8) == 0;
    for (size_t i = 0; i < n_itx; i++) {
        const Dav1dHipItxTask &t = itx[i];
        ck->order = std::min(ck->order, (uint64_t) t.plane << 40 | t.dst_off);
        if (taken[i]) continue;
        CellMap &m = cm[t.plane];
        uint16_t d = 0;
        m.each(t.dst_off, tx_w[t.tx], tx_h[t.tx], [&](size_t k) { d |= m.writers[k]; });
        ck->dep[t.tx] |= d;
        ibins[t.tx].push_back(t);
        itx_fill_prefix(ibins[t.tx].back());
    }
    for (size_t i = 0; i < n_mc; i++) if (mc[i].kind == DAV1D_HIP_MC_PUT) ck->order = std::min(ck->order, (uint64_t) mc[i].plane << 40 | mc[i].dst_off);

    P(4);
    // ---- order inside the chunk.  Predictions: by where they read (reference, plane, 64-row band, x), then inside windows of
    // 128 waves' worth by (leaves the reference plane, kind) so that the tiles of a wave share their code path.  Residuals:
    // inside windows by code path (dc-only, 1-D kinds).  Speed only: the tasks of a chunk write disjoint pixels.
    DevPlanes rp[8];
    for (int i = 0; i < n_refs; i++) rp[i] = dev_planes(&refs[i]);
    // Option chunk_order (0 by default since round 3): 1 = the orders below, which make the launches a few per cent faster (tiles of a wave on
    // one code path, neighbours in the reference next to each other); 0 = decode order, which a chunk — one row of superblocks walked left
    // to right — mostly is already.  The route from pass 1 to pixels is bound by HOST time per frame: measured on
    // MI355X at 8K, the orders cost 8.5 of 102 CPU-ms per frame and return 0.06 ms of a 1.9 ms dav1d_hip_frame_end.
    const bool ordered = c->chunk_order != 0;
    static const int mc_win_env = getenv("DAV1D_HIP_MC_GROUP_WINDOW") ? atoi(getenv("DAV1D_HIP_MC_GROUP_WINDOW")) : 128;
    static const int itx_win_env = getenv("DAV1D_HIP_ITX_SORT_WINDOW") ? atoi(getenv("DAV1D_HIP_ITX_SORT_WINDOW")) : 128;
    const int mc_win = ordered ? mc_win_env : 0, itx_win = ordered ? itx_win_env : 0;
    for (int b = 0; b < MC_BINS; b++) {
        std::vector<McTile> &v = bins[b];
        if (v.empty()) continue;
        // DAV1D_HIP_CHUNK_SORT: 1 = by where the tiles read (reference, plane, 64-row band, x); 2 = by (reference, plane) only, keeping
        // the decode order inside (a chunk is one row of superblocks: decode order already runs left to right); 0 = decode order
        // (measured on MI355X, 8K 10-bit: mode 1 makes the frame 2.5 % faster on the device and the listing 20 % slower on the host)
        static const int sort_mode_env = getenv("DAV1D_HIP_CHUNK_SORT") ? atoi(getenv("DAV1D_HIP_CHUNK_SORT")) : 2;
        const int sort_mode = ordered ? sort_mode_env : 0;
        if (sort_mode == 1) {
            std::vector<uint32_t> &sk = scr.sk;
            sk.resize(v.size());
            for (size_t i = 0; i < v.size(); i++) sk[i] = src_key(v[i]);
            sort_by_key(v, sk);
        } else if (sort_mode == 2) {
            std::vector<uint8_t> &gk = scr.gk;
            gk.resize(v.size());
            for (size_t i = 0; i < v.size(); i++) gk[i] = (uint8_t) ((v[i].r[0].ref & 7) * 3 + v[i].plane);
            group_in_windows(v, gk, v.size());
        }
        const int tw = 4 << (b / 3), th = 4 << (b % 3);
        const int lanes = tw * th / 4 < 64 ? tw * th / 4 : 64;
        if (mc_win > 0 && 64 / lanes >= 2 && n_refs) {
            const int ws = tw == 4 ? 12 : (tw + 8 + 7) & ~7, ext_x = (ws + 7) / 8 * 8, ext_y = th + 7;
            auto key = [&](const McTile &t) -> int {
                const bool two = t.kind == MCT_AVG || t.kind == MCT_WAVG;
                bool edge = false;
                for (int k = 0; k < (two ? 2 : 1); k++) {
                    const McRef &r = t.r[k];
                    if (r.ref >= n_refs) continue;
                    const int xx = r.src_x - 4, yy = r.src_y - 3;
                    edge |= xx < 0 || yy < 0 || xx + ext_x > rp[r.ref].w[t.plane] || yy + ext_y > rp[r.ref].h[t.plane];
                }
                return (edge ? 16 : 0) | t.kind << 1 | (t.r[0].src_x & 1);      // the column parity: the tiled form of the horizontal pass (mc_body.h)
            };
            std::vector<uint8_t> &gk = scr.gk;
            gk.resize(v.size());
            for (size_t i = 0; i < v.size(); i++) gk[i] = (uint8_t) key(v[i]);
            group_in_windows(v, gk, (size_t) mc_win * (size_t) (64 / lanes));
        }
        for (const McTile &t : v) {
            const bool two = t.kind == MCT_AVG || t.kind == MCT_WAVG;
            ck->max_ref = std::max(ck->max_ref, std::max((int) t.r[0].ref, two ? (int) t.r[1].ref : 0));
        }
    }
    P(5);
    if (itx_win > 0)
        for (int b = 0; b < 19; b++) {
            std::vector<Dav1dHipItxTask> &v = ibins[b];
            const int lanes = std::max(std::min((int) tx_h[b], 32), (int) tx_w[b]);
            std::vector<uint8_t> &gk = scr.gk;
            gk.resize(v.size());
            for (size_t i = 0; i < v.size(); i++) gk[i] = (uint8_t) itx_path_key(v[i]);
            group_in_windows(v, gk, (size_t) itx_win * (size_t) std::max(1, 64 / lanes));
        }
    P(6);
    // paired blocks: by where the first tile reads, then by (transform code path, prediction kind) inside windows
    std::vector<McTile> (&pt_sorted)[5] = scr.pt_sorted;
    std::vector<Dav1dHipItxTask> (&pk_sorted)[5] = scr.pk_sorted;
    for (int k = 0; k < 5; k++) { pt_sorted[k].clear(); pk_sorted[k].clear(); }
    for (int k = 0; k < 5; k++) {
        const size_t nblk = p_itx[k].size();
        if (!nblk) continue;
        const int tpb = k < 3 ? 1 : k == 3 ? 2 : 4, bpw = k == 0 ? 16 : k == 1 ? 8 : k == 2 ? 4 : k == 3 ? 2 : 1;
        if (p_tiles[k].size() != nblk * tpb) { delete ck; return -EINVAL; }
        std::vector<uint32_t> &ord = scr.ord;
        ord.resize(nblk);
        for (size_t i = 0; i < nblk; i++) ord[i] = (uint32_t) i;
        {
            // DAV1D_HIP_PAIR_SORT: 1 = by where the first tile reads (reference, plane, 64-row band, x); 2 (default) = by (reference, plane)
            // only — a chunk is one row of superblocks listed in decode order, which already runs from left to right
            static const int pair_sort_env = getenv("DAV1D_HIP_PAIR_SORT") ? atoi(getenv("DAV1D_HIP_PAIR_SORT")) : 2;
            const int pair_sort = ordered ? pair_sort_env : 0;
            if (pair_sort == 1) {
                std::vector<uint32_t> &sk = scr.sk;
                sk.resize(nblk);
                for (size_t i = 0; i < nblk; i++) sk[i] = src_key(p_tiles[k][i * tpb]);
                sort_by_key(ord, sk);
            } else if (pair_sort == 2) {
                std::vector<uint8_t> &rk = scr.gk;
                rk.resize(nblk);
                for (size_t i = 0; i < nblk; i++) { const McTile &t0 = p_tiles[k][i * tpb]; rk[i] = (uint8_t) ((t0.r[0].ref & 7) * 3 + t0.plane); }
                group_in_windows(ord, rk, nblk);
            }
            std::vector<uint8_t> &gk = scr.gk;
            gk.resize(nblk);
            for (size_t i = 0; i < nblk; i++) {
                const int kind = p_tiles[k][(size_t) ord[i] * tpb].kind;
                gk[i] = (uint8_t) ((itx_path_key(itx[p_itx[k][ord[i]]]) * 3 + (kind == MCT_AVG ? 1 : kind == MCT_WAVG ? 2 : 0)) * 2 +
                                   (p_tiles[k][(size_t) ord[i] * tpb].r[0].src_x & 1));
            }
            group_in_windows(ord, gk, (size_t) 128 * bpw);
        }
        pt_sorted[k].resize(nblk * tpb);
        pk_sorted[k].resize(nblk);
        for (size_t i = 0; i < nblk; i++) {
            for (int j = 0; j < tpb; j++) {
                const McTile &t = pt_sorted[k][i * tpb + j] = p_tiles[k][(size_t) ord[i] * tpb + j];
                const bool two = t.kind == MCT_AVG || t.kind == MCT_WAVG;
                ck->max_ref = std::max(ck->max_ref, std::max((int) t.r[0].ref, two ? (int) t.r[1].ref : 0));
            }
            pk_sorted[k][i] = itx[p_itx[k][ord[i]]];
            itx_fill_prefix(pk_sorted[k][i]);
        }
    }
    P(7);
    // compound / blend tasks: BLEND_V and the MASK tasks that read a mask a W_MASK task of the chunk writes go second
    std::vector<Dav1dHipCompTask> &c_first = scr.c_first, &c_second = scr.c_second;
    c_first.clear(); c_second.clear();
    if (!rest.empty()) {
        std::unordered_map<uint32_t, char> wmask_out;
        for (const Dav1dHipCompTask &t : rest) if (t.kind == DAV1D_HIP_COMP_WMASK) wmask_out[t.mask_off] = 1;
        for (const Dav1dHipCompTask &t : rest)
            ((t.kind == DAV1D_HIP_COMP_BLEND_V || (t.kind == DAV1D_HIP_COMP_MASK && wmask_out.count(t.mask_off))) ? c_second : c_first).push_back(t);
    }

    P(8);
    // ---- one pinned blob: [mc bins][itx bins][paired tiles][paired residuals][comp first][comp second], 16-byte aligned segments
    size_t total = 0;
    auto place = [&](int id, size_t n, size_t esz) { ck->seg[id].off = (uint32_t) total; ck->seg[id].n = (uint32_t) n; total += (n * esz + 15) & ~(size_t) 15; };
    for (int b = 0; b < MC_BINS; b++) place(CK_MC + b, bins[b].size(), sizeof(McTile));
    for (int b = 0; b < 19; b++) place(CK_ITX + b, ibins[b].size(), sizeof(Dav1dHipItxTask));
    for (int k = 0; k < 5; k++) place(CK_PTILE + k, pt_sorted[k].size(), sizeof(McTile));
    for (int k = 0; k < 5; k++) place(CK_PTASK + k, pk_sorted[k].size(), sizeof(Dav1dHipItxTask));
    place(CK_COMP, c_first.size(), sizeof(Dav1dHipCompTask));
    place(CK_COMP + 1, c_second.size(), sizeof(Dav1dHipCompTask));
    ck->used = total;
    if (total) {
        uint8_t *dst = place_blob ? place_blob(cookie, total, &ck->dev_off) : nullptr;
        if (dst) {
            ck->uploaded = true;                 // part of the twin: goes up with it
        } else {
            dst = ck->host = slab_get(c, total, &ck->cap);
            if (!ck->host) { delete ck; return -ENOMEM; }
        }
        auto put = [&](int id, const void *src, size_t esz) { if (ck->seg[id].n) memcpy(dst + ck->seg[id].off, src, (size_t) ck->seg[id].n * esz); };
        for (int b = 0; b < MC_BINS; b++) put(CK_MC + b, bins[b].data(), sizeof(McTile));
        for (int b = 0; b < 19; b++) put(CK_ITX + b, ibins[b].data(), sizeof(Dav1dHipItxTask));
        for (int k = 0; k < 5; k++) put(CK_PTILE + k, pt_sorted[k].data(), sizeof(McTile));
        for (int k = 0; k < 5; k++) put(CK_PTASK + k, pk_sorted[k].data(), sizeof(Dav1dHipItxTask));
        put(CK_COMP, c_first.data(), sizeof(Dav1dHipCompTask));
        put(CK_COMP + 1, c_second.data(), sizeof(Dav1dHipCompTask));
    }
    P(9);
    *out = ck;
    return 0;
}

// the pinned slabs for other users of the library (frame.hip: the merged units of a dataflow launch)
uint8_t *dav1d_hip_slab_get(Dav1dHipContext *c, size_t bytes, size_t *cap) { return slab_get(c, bytes, cap); }
void dav1d_hip_slab_put(Dav1dHipContext *c, uint8_t *host, size_t cap) { slab_put(c, host, cap); }

// ------------------------------------------------------------------ gather: chunk segments -> contiguous per-bin arrays

namespace {
struct GatherSeg { uint32_t src, dst, words, pad; };          // byte offsets in the chunk arena / the gathered arena, length in 32-bit words

__global__ __launch_bounds__(256) void gather_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const GatherSeg *__restrict__ segs, const int n)
{
    const int i = blockIdx.x;
    if (i >= n) return;
    const GatherSeg s = segs[i];
    const uint32_t *const a = reinterpret_cast<const uint32_t *>(src + s.src);
    uint32_t *const d = reinterpret_cast<uint32_t *>(dst + s.dst);
    for (uint32_t k = threadIdx.x; k < s.words; k += 256) d[k] = a[k];
}
} // namespace

static int ensure_dev(uint8_t **p, size_t *cap, size_t want, hipStream_t sync_on) {
    if (*cap >= want) return 0;
    if (*p) { (void) hipStreamSynchronize(sync_on); (void) hipFree(*p); *p = nullptr; *cap = 0; }
    size_t n = 1 << 22;
    while (n < want) n <<= 1;
    if (hipMalloc((void **) p, n) != hipSuccess) return -ENOMEM;
    *cap = n;
    return 0;
}

int dav1d_hip_chunks_grow_arena(Dav1dHipContext *c, uint8_t **arena, size_t *arena_cap, size_t need, bool *regrown) {
    *regrown = false;
    if (need + 256 <= *arena_cap) return 0;
    (void) hipStreamSynchronize(c->copy_stream);
    const int rc = ensure_dev(arena, arena_cap, need + 256, c->stream);
    if (!rc) *regrown = true;
    return rc;
}

int dav1d_hip_chunks_send_late(Dav1dHipContext *c, std::vector<Dav1dHipChunk *> &chunks, uint8_t *arena, size_t arena_cap, bool regrown) {
    for (Dav1dHipChunk *ck : chunks) {
        if (!ck->used || !ck->host || (ck->uploaded && !regrown)) continue;
        if (ck->dev_off + ck->used > arena_cap) return -EINVAL;
        const int rc = hip_rc(hipMemcpyAsync(arena + ck->dev_off, ck->host, ck->used, hipMemcpyHostToDevice, c->copy_stream));
        if (rc) return rc;
        ck->uploaded = true;
    }
    return 0;
}

// Uploads the chunks (one copy each, on the context's copy stream), lines their segments up per bin with one gather launch and
// fills the caller-provided list objects with views of the gathered arrays.  The lists own nothing (never destroy them).
int dav1d_hip_chunks_to_recon_list(Dav1dHipContext *c, std::vector<Dav1dHipChunk *> &chunks, uint8_t **arena, size_t *arena_cap,
                                   const Dav1dHipPicture *refs, int n_refs,
                                   Dav1dHipReconList *l, Dav1dHipInterList *il, Dav1dHipMcList *ml, Dav1dHipCompList *cl, Dav1dHipItxList *xl)
{
    std::sort(chunks.begin(), chunks.end(), [](const Dav1dHipChunk *a, const Dav1dHipChunk *b) { return a->order < b->order; });
    size_t esz[CK_N];
    for (int a = 0; a < CK_N; a++) esz[a] = sizeof(Dav1dHipItxTask);
    for (int b = 0; b < MC_BINS; b++) esz[CK_MC + b] = sizeof(McTile);
    for (int k = 0; k < 5; k++) esz[CK_PTILE + k] = sizeof(McTile);
    esz[CK_COMP] = esz[CK_COMP + 1] = sizeof(Dav1dHipCompTask);
    size_t cnt[CK_N] = { 0 }, src_total = 0;
    bool all_up = true;
    for (Dav1dHipChunk *ck : chunks) {
        for (int a = 0; a < CK_N; a++) cnt[a] += ck->seg[a].n;
        src_total += (ck->used + 255) & ~(size_t) 255;
        all_up &= ck->uploaded || !ck->used;
    }
    c->arena_hint = std::max(c->arena_hint, src_total);
    int rc = 0;
    if (!all_up) return -EINVAL;                 // dav1d_hip_chunks_send_late comes first
    (void) arena; (void) arena_cap;
    // gathered layout: the 15 prediction bins back to back, the 19 residual bins back to back, every paired array on its own,
    // the two compound runs back to back; each group starts 256-byte aligned
    size_t aoff[CK_N], end = 0;
    auto group = [&](int first, int n) {
        end = (end + 255) & ~(size_t) 255;
        for (int a = first; a < first + n; a++) { aoff[a] = end; end += cnt[a] * esz[a]; }
    };
    group(CK_MC, MC_BINS);
    group(CK_ITX, 19);
    for (int k = 0; k < 5; k++) group(CK_PTILE + k, 1);
    for (int k = 0; k < 5; k++) group(CK_PTASK + k, 1);
    group(CK_COMP, 2);
    rc = ensure_dev(&c->gather_dev, &c->gather_cap, end + 256, c->stream);
    if (rc) return rc;
    size_t n_seg = 0;
    for (Dav1dHipChunk *ck : chunks) for (int a = 0; a < CK_N; a++) n_seg += ck->seg[a].n != 0;
    size_t tab_cap = 0;
    uint8_t *tab_host = n_seg ? slab_get(c, n_seg * sizeof(GatherSeg), &tab_cap) : nullptr;
    if (n_seg && !tab_host) return -ENOMEM;
    rc = ensure_dev(&c->segtab_dev, &c->segtab_cap, n_seg * sizeof(GatherSeg) + 256, c->stream);
    if (rc) { slab_put(c, tab_host, tab_cap); return rc; }
    GatherSeg *tab = reinterpret_cast<GatherSeg *>(tab_host);
    size_t run[CK_N] = { 0 }, k = 0;
    for (Dav1dHipChunk *ck : chunks)
        for (int a = 0; a < CK_N; a++)
            if (ck->seg[a].n) {
                tab[k].src = (uint32_t) (ck->dev_off + ck->seg[a].off);
                tab[k].dst = (uint32_t) (aoff[a] + run[a] * esz[a]);
                tab[k].words = (uint32_t) (ck->seg[a].n * esz[a] / 4);
                tab[k].pad = 0;
                run[a] += ck->seg[a].n;
                k++;
            }
    // the chunks went up on the copy stream when they were submitted; the gather runs behind them
    if (!rc && n_seg) rc = hip_rc(hipMemcpyAsync(c->segtab_dev, tab_host, n_seg * sizeof(GatherSeg), hipMemcpyHostToDevice, c->copy_stream));
    if (!rc) rc = hip_rc(hipEventRecord(c->ev_copy, c->copy_stream));
    if (!rc) rc = hip_rc(hipStreamWaitEvent(c->stream, c->ev_copy, 0));
    if (!rc && n_seg) {
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned) n_seg), dim3(256), 0, c->stream, *arena, c->gather_dev,
                           reinterpret_cast<const GatherSeg *>(c->segtab_dev), (int) n_seg);
        rc = hip_rc(hipGetLastError());
    }
    c->pending_slab = tab_host; c->pending_slab_cap = tab_cap;      // recycled once the frame has synchronised
    if (rc) return rc;

    // ---- the list views
    memset(static_cast<void *>(ml), 0, sizeof(*ml));
    ml->dev = reinterpret_cast<McTile *>(c->gather_dev + aoff[CK_MC]);
    for (int b = 0; b < MC_BINS; b++) ml->off[b + 1] = ml->off[b] + cnt[CK_MC + b];
    ml->n = ml->off[MC_BINS];
    DevPlanes rp[8];
    for (int i = 0; i < n_refs; i++) rp[i] = dev_planes(&refs[i]);
    ml->geo_sig = dav1d_hip_mc_geo_sig(rp, n_refs);      // the chunks grouped their tiles for this geometry already
    cl->dev = reinterpret_cast<Dav1dHipCompTask *>(c->gather_dev + aoff[CK_COMP]);
    cl->n_first = cnt[CK_COMP];
    cl->n = cnt[CK_COMP] + cnt[CK_COMP + 1];
    memset(static_cast<void *>(xl), 0, sizeof(*xl));
    xl->dev = reinterpret_cast<Dav1dHipItxTask *>(c->gather_dev + aoff[CK_ITX]);
    for (int b = 0; b < 19; b++) xl->off[b + 1] = xl->off[b] + cnt[CK_ITX + b];
    xl->n = xl->off[19];
    il->mc = ml; il->comp = cl; il->n_fused = 0;
    for (int p = 0; p < 3; p++) il->cell_stride[p] = il->stride_px[p] = 0;
    l->inter = il; l->itx = xl;
    l->f_max_ref = 0;
    l->wide_ok = true;
    for (int b = 0; b < 19; b++) l->dep[b] = 0;
    for (Dav1dHipChunk *ck : chunks) {
        l->wide_ok = l->wide_ok && ck->wide_ok;
        for (int b = 0; b < 19; b++) l->dep[b] |= ck->dep[b];
        ml->max_ref = std::max(ml->max_ref, ck->max_ref);
        l->f_max_ref = std::max(l->f_max_ref, ck->max_ref);
    }
    for (int q = 0; q < 5; q++) {
        l->f_n[q] = cnt[CK_PTASK + q];
        l->f_tiles[q] = cnt[CK_PTILE + q] ? reinterpret_cast<McTile *>(c->gather_dev + aoff[CK_PTILE + q]) : nullptr;
        l->f_tasks[q] = cnt[CK_PTASK + q] ? reinterpret_cast<Dav1dHipItxTask *>(c->gather_dev + aoff[CK_PTASK + q]) : nullptr;
    }
    for (int p = 0; p < 3; p++) l->stride_px[p] = 0;        // the frame's own picture: no geometry to re-check at run time
    return 0;
}

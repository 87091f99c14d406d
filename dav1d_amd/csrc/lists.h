// Device-resident task lists of the C ABI (host side): shared by capi.hip (lists made from whole task arrays) and chunk.hip
// (lists assembled from per-tile-sbrow chunks that were preprocessed on the submitting threads).
#pragma once
#include "capi.h"
#include <vector>

#define MC_BINS 15

struct Dav1dHipItxList {
    Dav1dHipItxTask *dev;
    size_t n;
    size_t off[20];   // bin b occupies [off[b], off[b+1])
};

struct Dav1dHipMcList {
    McTile *dev;      // tiles bin by bin (one launch per tile shape)
    size_t n;
    size_t off[16];   // 15 tile-shape bins: 3 * class(w in 4..64) + class(h in 4..16)
    McTile *dev_all;  // the same tiles, all shapes interleaved in source order (one launch for everything)
    McGroup *groups;
    size_t n_groups, n_fused;
    int max_ref;      // highest reference index any tile uses: checked against n_refs at run time
    McTile *host;     // host copy of `dev` in source order: regrouped per reference geometry at run time
    uint64_t geo_sig; // geometry the device copy is grouped for (0 = not yet)
};

struct Dav1dHipCompList {
    Dav1dHipCompTask *dev;
    size_t n;
    size_t n_first;   // tasks [0, n_first) run in the first launch, the BLEND_V tasks after them in a second one
};

struct Dav1dHipInterList {
    Dav1dHipMcList *mc;
    Dav1dHipCompList *comp;
    size_t n_fused;
    // with a picture geometry (recon lists): which launches write each 4x4 cell of a plane — bit b = the mc launch of tile
    // shape b, bit 15 = the compound / blend launch
    std::vector<uint16_t> writers[3];
    int cell_stride[3], stride_px[3];
};

struct Dav1dHipReconList {
    Dav1dHipInterList *inter;  // predictions that have no residual of their own shape (and everything when pairing is off)
    Dav1dHipItxList *itx;      // residuals without a prediction of their own shape
    uint16_t dep[19];          // per transform size: bits of the launches (see Dav1dHipInterList::writers) it has to wait for
    int stride_px[3];
    // paired blocks, per square size class 4x4 .. 64x64: tiles (1, 1, 1, 2, 4 per block) and transform tasks, device resident
    McTile *f_tiles[5];
    Dav1dHipItxTask *f_tasks[5];
    size_t f_n[5];
    int f_max_ref;
    bool wide_ok;              // every transform block starts at a multiple of min(its width, 8) pixels: the blocks may leave through
                               // tile_write_out (itx_body.h) in aligned row pieces.  Always so for AV1 geometry; checked because lists are an API
};

bool itx_task_ok(const Dav1dHipItxTask &t);
void itx_fill_prefix(Dav1dHipItxTask &t);
int itx_path_key(const Dav1dHipItxTask &t);
int mc_task_valid(const Dav1dHipMcTask &t);
McRef mc_ref_of(const Dav1dHipMcTask &t);
void push_tiles(std::vector<McTile> *bins, const Dav1dHipMcTask &t, int kind, uint32_t dst_off, const Dav1dHipMcTask *second, int weight,
                std::vector<McTile> *single = nullptr);
int recon_fuse_mask(const Dav1dHipContext *c);
int tile_dim_class(int v);
uint8_t *dav1d_hip_slab_get(Dav1dHipContext *c, size_t bytes, size_t *cap);      // chunk.hip: pinned host memory, recycled through the context
void dav1d_hip_slab_put(Dav1dHipContext *c, uint8_t *host, size_t cap);

// One preprocessed tile-sbrow of reconstruction work (chunk.hip).
#pragma once
#include "capi.h"
#include "lists.h"
#include <vector>

// array ids inside a chunk / the gathered arena
enum { CK_MC = 0, CK_ITX = 15, CK_PTILE = 34, CK_PTASK = 39, CK_COMP = 44, CK_N = 46 };

struct Dav1dHipChunk {
    struct Seg { uint32_t off, n; } seg[CK_N];   // byte offset inside the blob, element count
    uint16_t dep[19];                            // per residual size: the prediction launches its blocks wait for
    int max_ref;
    uint8_t *host;                               // pinned blob (a slab of the context's pool)
    size_t cap, used;
    size_t dev_off;                              // where the blob sits in the frame's chunk arena
    bool uploaded;                               // ... once it has been sent there (at submit time when the arena had room)
    uint64_t order;                              // first destination position: chunks are lined up in picture order
    void release(Dav1dHipContext *c);
};

int dav1d_hip_chunk_build(Dav1dHipContext *c, Dav1dHipChunk **out, const Dav1dHipPicture *geom, const Dav1dHipPicture *refs, int n_refs,
                          const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                          const Dav1dHipItxTask *itx, size_t n_itx);
int dav1d_hip_chunks_to_recon_list(Dav1dHipContext *c, std::vector<Dav1dHipChunk *> &chunks, uint8_t **arena, size_t *arena_cap,
                                   const Dav1dHipPicture *refs, int n_refs,
                                   Dav1dHipReconList *l, Dav1dHipInterList *il, Dav1dHipMcList *ml, Dav1dHipCompList *cl, Dav1dHipItxList *xl);

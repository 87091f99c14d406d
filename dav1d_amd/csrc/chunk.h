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
    uint8_t *host;                               // pinned blob (a slab of the context's pool) — NULL when the blob was written straight into
                                                 // the pinned twin of the frame's arena (the usual case)
    size_t cap, used;
    size_t dev_off;                              // where the blob sits in the frame's chunk arena (drawn when the chunk was built)
    bool uploaded;                               // it is in the twin, or has been sent on its own
    bool wide_ok;                                // its transform blocks may leave in row pieces of up to 8 pixels (Dav1dHipReconList::wide_ok)
    uint64_t order;                              // first destination position: chunks are lined up in picture order
    void release(Dav1dHipContext *c);
};

// place(cookie, bytes, &dev_off): draws the blob's place in the frame's arena; returns where to write it (inside the arena's pinned twin)
// or NULL when the twin has no room — the blob then goes to a slab of its own and is sent at frame end
typedef uint8_t *(*Dav1dHipChunkPlace)(void *cookie, size_t bytes, size_t *dev_off);
int dav1d_hip_chunk_build(Dav1dHipContext *c, Dav1dHipChunk **out, const Dav1dHipPicture *geom, const Dav1dHipPicture *refs, int n_refs,
                          const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                          const Dav1dHipItxTask *itx, size_t n_itx, Dav1dHipChunkPlace place, void *cookie, bool trusted = false,
                          const uint16_t *itx_dep = nullptr);
// The frame's arena grown to `need` bytes if it is smaller (contents lost: *regrown = true, the caller sends its twin again).  Then, AFTER
// the twin has gone (a late chunk may start inside the range the twin covers: the twin's bytes there are not the chunk's), every chunk
// that lives in a slab of its own is sent to its place (copy stream; `regrown`: also those that had been sent before).
int dav1d_hip_chunks_grow_arena(Dav1dHipContext *c, uint8_t **arena, size_t *arena_cap, size_t need, bool *regrown);
int dav1d_hip_chunks_send_late(Dav1dHipContext *c, std::vector<Dav1dHipChunk *> &chunks, uint8_t *arena, size_t arena_cap, bool regrown);
int dav1d_hip_chunks_to_recon_list(Dav1dHipContext *c, std::vector<Dav1dHipChunk *> &chunks, uint8_t **arena, size_t *arena_cap,
                                   const Dav1dHipPicture *refs, int n_refs,
                                   Dav1dHipReconList *l, Dav1dHipInterList *il, Dav1dHipMcList *ml, Dav1dHipCompList *cl, Dav1dHipItxList *xl);

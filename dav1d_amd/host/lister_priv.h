/* What the filter lister needs to know about a Dav1dHipLister (dav1d_amd/host/lister.c). */
#ifndef DAV1D_HIP_LISTER_PRIV_H
#define DAV1D_HIP_LISTER_PRIV_H
#include "dav1d_hip.h"

typedef struct ListerGeo {
    Dav1dHipFrame *frame;
    int w, h, layout, bpc, sb128, ss_hor, ss_ver, bw, bh, sb_step;
    int stride[3];              /* picture strides in pixels */
    ptrdiff_t b4_stride;
    int n_tile_cols, n_tile_rows;
    const uint16_t *col_start_sb, *row_start_sb;
} ListerGeo;
void dav1d_hip_lister_geo(const Dav1dHipLister *l, ListerGeo *out);
extern long long dav1d_hip_live[8];      /* csrc/capi.hip: objects alive by kind (dav1d_hip_live_objects) */
/* frame.hip: filter tasks handed over as malloc'ed arrays the frame frees (no copy under the frame's lock) */
int dav1d_hip_frame_submit_filter_owned(Dav1dHipFrame *f, Dav1dHipLfTask *lf, size_t n_lf, Dav1dHipCdefTask *cdef, size_t n_cdef,
                                        Dav1dHipLrTask *lr, size_t n_lr);
/* frame.hip: the frame's dense table of CDEF unit rows ([by8 * stride + 64-pixel column], cdef_rows.h; zeroed) for the filter lister to
 * fill in — every entry by one thread — or NULL when the frame takes unit records (banded post filters, unaligned planes, option
 * cdef_rows 0); _add: so many units were marked in it. */
struct Dav1dHipCdefRow;
struct Dav1dHipCdefRow *dav1d_hip_frame_cdef_rows(Dav1dHipFrame *f, int *stride);
void dav1d_hip_frame_cdef_rows_add(Dav1dHipFrame *f, size_t n_units);
/* One transform block's values to pack (lister.c pack_note / dav1d_hip_pack_run): the block's slab in the hand-off's coefficient array (cf_off, in
 * coefficients), how many values in decode order (eob + 1), the transform and its kind (which say where value i sits in the slab). */
typedef struct Dav1dHipPackRec { uint32_t cf_off; uint16_t n; uint8_t tx, txtp; } Dav1dHipPackRec;
/* lister.c: the blocks of recs[] packed one behind the other into dst (csz = 2 or 4 bytes per value), their slabs in `cf` zeroed */
void dav1d_hip_pack_run(const Dav1dHipPackRec *recs, size_t n, void *cf, void *dst, int csz);
/* frame.hip: room for n values in the frame's packed-coefficient arena: *base = its offset in values, *dst = where the caller (or the preparation job
 * it hands the row to) writes them */
int dav1d_hip_frame_reserve_coefs(Dav1dHipFrame *f, size_t n, uint32_t *base, void **dst);
/* frame.hip: a tile-sbrow's records (as dav1d_hip_frame_submit_tile_sbrow_own) together with its packing work: recs are packed from cf into dst — here, or
 * on the library's preparation threads with the chunk preparation (option prep_async) */
int dav1d_hip_frame_submit_tile_sbrow_packing(Dav1dHipFrame *f, const Dav1dHipMcTask *mc, size_t n_mc, const Dav1dHipCompTask *comp, size_t n_comp,
                                              const Dav1dHipItxTask *itx, size_t n_itx, const uint16_t *itx_dep,
                                              const Dav1dHipPackRec *recs, size_t n_recs, void *cf, void *dst);
/* lister.c: fn(arg) on n threads at once — the caller and n - 1 threads of a pool the library keeps (created on first use, parked on
 * a condition variable between jobs; one job at a time per process).  Starting 63 threads per frame took the caller a millisecond. */
void dav1d_hip_host_pool_run(void *(*fn)(void *), void *arg, int n);
/* filter_lister.c: the filter tasks of a frame as independent units of work (a superblock row's deblocking, CDEF or restoration list) */
int dav1d_hip_lister_filter_units(const Dav1dHipLister *l);
int dav1d_hip_lister_filter_unit(Dav1dHipLister *l, const Dav1dHipFilterDesc *fd, int unit);
#endif

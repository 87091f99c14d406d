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
#endif

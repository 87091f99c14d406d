/* Host-side AV1 geometry shared by the pass-2 lister, the filter lister and the synthetic frame generator.
 *
 * Everything here is derived by rule from the AV1 specification (block / transform size enumerations, partition shapes,
 * transform splitting, wedge and inter-intra masks, the warp shear set-up) instead of being carried as literal tables;
 * tests/test_host_tables.py checks every derived value against the tables of the reference build
 * (dav1d_block_dimensions, dav1d_txfm_dimensions, dav1d_max_txfm_size_for_bs, dav1d_block_sizes: reference
 * src/tables.c:56-240; dav1d_masks: src/wedge.c; dav1d_get_shear_params / dav1d_set_affine_mv2d: src/warpmv.c:79-147).
 * Plain C99, no HIP. */
#ifndef DAV1D_HIP_AV1_HOST_H
#define DAV1D_HIP_AV1_HOST_H

#include <stddef.h>
#include <stdint.h>
#include "dav1d_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* numeric values == the reference's enums (src/levels.h) because they are what Av1Block stores */
enum { H_BL_128X128, H_BL_64X64, H_BL_32X32, H_BL_16X16, H_BL_8X8 };
enum { H_PART_NONE, H_PART_H, H_PART_V, H_PART_SPLIT, H_PART_T_TOP_SPLIT, H_PART_T_BOTTOM_SPLIT, H_PART_T_LEFT_SPLIT,
       H_PART_T_RIGHT_SPLIT, H_PART_H4, H_PART_V4, H_N_PARTITIONS };
enum { H_BS_128x128, H_BS_128x64, H_BS_64x128, H_BS_64x64, H_BS_64x32, H_BS_64x16, H_BS_32x64, H_BS_32x32, H_BS_32x16,
       H_BS_32x8, H_BS_16x64, H_BS_16x32, H_BS_16x16, H_BS_16x8, H_BS_16x4, H_BS_8x32, H_BS_8x16, H_BS_8x8, H_BS_8x4,
       H_BS_4x16, H_BS_4x8, H_BS_4x4, H_N_BS };
enum { H_TX_4X4, H_TX_8X8, H_TX_16X16, H_TX_32X32, H_TX_64X64, H_RTX_4X8, H_RTX_8X4, H_RTX_8X16, H_RTX_16X8, H_RTX_16X32,
       H_RTX_32X16, H_RTX_32X64, H_RTX_64X32, H_RTX_4X16, H_RTX_16X4, H_RTX_8X32, H_RTX_32X8, H_RTX_16X64, H_RTX_64X16, H_N_TX };
enum { H_DC_PRED, H_VERT_PRED, H_HOR_PRED, H_DIAG_DOWN_LEFT_PRED, H_DIAG_DOWN_RIGHT_PRED, H_VERT_RIGHT_PRED, H_HOR_DOWN_PRED,
       H_HOR_UP_PRED, H_VERT_LEFT_PRED, H_SMOOTH_PRED, H_SMOOTH_V_PRED, H_SMOOTH_H_PRED, H_PAETH_PRED, H_CFL_PRED = 13,
       H_FILTER_PRED = 13 };
enum { H_COMP_INTER_NONE, H_COMP_INTER_WEIGHTED_AVG, H_COMP_INTER_AVG, H_COMP_INTER_SEG, H_COMP_INTER_WEDGE };
enum { H_INTER_INTRA_NONE, H_INTER_INTRA_BLEND, H_INTER_INTRA_WEDGE };
enum { H_MM_TRANSLATION, H_MM_OBMC, H_MM_WARP };
enum { H_GLOBALMV = 2, H_GLOBALMV_GLOBALMV = 6 };
enum { H_FILTER_2D_BILINEAR = 9 };
enum { H_WM_IDENTITY, H_WM_TRANSLATION, H_WM_ROT_ZOOM, H_WM_AFFINE };
/* edge availability bits == enum EdgeFlags, reference src/intra_edge.h:33-52 */
enum { H_EDGE_I444_TR = 1, H_EDGE_I422_TR = 2, H_EDGE_I420_TR = 4, H_EDGE_I444_BL = 8, H_EDGE_I422_BL = 16, H_EDGE_I420_BL = 32,
       H_EDGE_ALL_TR = 7, H_EDGE_ALL_BL = 56 };

typedef struct HostTx { uint8_t w, h, lw, lh, min, max, sub; } HostTx;     /* w, h in 4-pixel units */

extern uint8_t h_bs_dim[H_N_BS][4];          /* bw4, bh4, log2 of both */
extern HostTx  h_tx[H_N_TX];
extern uint8_t h_max_tx_for_bs[H_N_BS][4];   /* [bs][0 = luma, 1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4 chroma] */
extern uint8_t h_block_sizes[5][H_N_PARTITIONS][2];
void h_tables_init(void);                    /* idempotent, thread-safe after the first call returned */
int h_bs_from_dim(int bw4, int bh4);         /* -1 when no such block size */
int h_tx_from_dim(int w4, int h4);

/* Wedge and inter-intra masks as one constant blob (to be placed at the start of the mask arena). */
typedef struct HostMasks {
    uint8_t *blob;
    size_t size;
    uint32_t wedge[3][11][2][16];  /* [0 = 4:4:4, 1 = 4:2:2, 2 = 4:2:0][bs - BS_32x32 (.. BS_8x8)][sign][wedge_idx] -> byte offset */
    uint32_t ii[3][11][4];         /* [layout idx][bs - BS_32x32][II mode: DC, VERT, HOR, SMOOTH] -> byte offset */
} HostMasks;
const HostMasks *h_masks(void);

/* t->warpmv of a MM_WARP block from the Av1Block fields (reference src/decode.c:746-757): returns the warp type */
int h_block_warp(Dav1dHipWarpParams *wm, const int16_t matrix[4], const int16_t mv2d[2] /* y, x */, int bw4, int bh4, int bx4, int by4);
int h_shear_params(Dav1dHipWarpParams *wm); /* != 0: not a valid shear */

#ifdef __cplusplus
}
#endif
#endif

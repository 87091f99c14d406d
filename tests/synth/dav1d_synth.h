/* Generator of synthetic pass-1 output.  TEST INFRASTRUCTURE: built into tests/synth/libdav1d_synth.so (oracle/Makefile, target
 * `synth`), loaded by tests/, bench.py's generators and the chain mode of oracle/ref_hooked.c — never part of, linked into or
 * exported by the product library dav1d_amd/libdav1d_hip.so (round 3 had it there). */
#ifndef DAV1D_SYNTH_H
#define DAV1D_SYNTH_H
#include "dav1d_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Input generator (tests / bench.py; not part of the decode path): fills the hand-off arrays `desc` points to the way pass 1 of
 * dav1d would have — block decisions drawn from a seeded generator under the legality rules of the AV1 syntax, coefficients on
 * the scan positions up to each block's eob — because no AV1 streams or encoders exist in the build / GPU environment.  The
 * arrays must be sized as dav1d_decode_frame_init() sizes them (src/decode.c:2839-2895) and cf zeroed.  Percentages 0..100. */
typedef struct Dav1dSynthParams {
    uint64_t seed;
    int intra_pct, skip_pct;                 /* intra blocks on inter frames; skip (no residual) blocks */
    int compound_pct, masked_compound;       /* two-reference blocks; allow COMP_INTER_SEG / WEDGE among them */
    int global_pct;                          /* GLOBALMV blocks (warped when desc->gmv_warp_allowed[ref]) */
    int interintra_pct, obmc_pct, warp_pct;  /* single-reference tools */
    int cfl_pct, palette, filter_intra_pct;  /* intra tools (palette: percentage among eligible blocks, needs desc->pal) */
    int tx_split_pct, alt_txtp_pct;          /* transform splitting per tree node; non-DCT_DCT transform types */
    int eob_none_pct;                        /* transform blocks without coefficients (eob = -1) */
    int mv_range, far_mv_pct;                /* |mv| in 1/8 pel; vectors pointing far outside the picture (edge emulation) */
    int n_refs;                              /* references in use, 1..7 */
    int split_pct[5], rect_pct;              /* per block level 128 .. 8: split; among the rest: a non-square partition */
    int fixed_bl;                            /* >= 0: every block is the square of that level (0 = 128x128 .. 4 = 8x8, 5 = 4x4) */
    int cf_align64;                          /* == Dav1dHipFrameDesc.cf_align64 */
    int intrabc_pct;                         /* key / intra-only frames: blocks (up to 64x64) coded as intra block copies where a source
                                                rectangle exists in the tile's superblock rows above or 256 pixels to the left */
    int n_segs;                              /* segmentation: seg_id drawn from 0 .. n_segs - 1 per block (0 / 1: every block in segment 0) */
    int skip_mode_pct;                       /* inter frames: blocks coded with skip_mode (two fixed references averaged, no residual,
                                                src/decode.c:1399-1404) */
} Dav1dSynthParams;
int dav1d_synth_frame(const Dav1dHipFrameDesc *desc, const Dav1dSynthParams *sp, void *cf, size_t cf_bytes,
                                        size_t cbi_entries, uint8_t *pal_idx, size_t pal_idx_bytes);

#ifdef __cplusplus
}
#endif
#endif

/* oracle/port — plain-C restatement of the reference's DSP functions for the hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle/Makefile): used as a checker by tests/ and as an optional
 * CPU baseline by bench.py; never included, linked or loaded by dav1d_amd/. */
#ifndef ORACLE_PORT_H
#define ORACLE_PORT_H
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int port_iclip(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static inline int port_imin(int a, int b) { return a < b ? a : b; }
static inline int port_imax(int a, int b) { return a > b ? a : b; }

/* AV1 constant tables (generated spec data shared with the kernels: dav1d_amd/csrc/av1_tables.h) */
#define AV1_TABLE_QUAL static const
#include "../../dav1d_amd/csrc/av1_tables.h"

void port_inv_txfm_add(void *dst, ptrdiff_t stride, void *coeff, int eob, int tx, int txtp, int bitdepth_max);
void port_mc(void *dst, ptrdiff_t dst_stride, int16_t *tmp, const void *src, ptrdiff_t src_stride,
             int w, int h, int mx, int my, int filter_2d, int bitdepth_max);
void port_comp(int kind, int ss, void *dst, ptrdiff_t dst_stride, const int16_t *tmp1, const int16_t *tmp2,
               int w, int h, int arg, const uint8_t *mask_in, uint8_t *mask_out, int bitdepth_max);
void port_emu_edge(intptr_t bw, intptr_t bh, intptr_t iw, intptr_t ih, intptr_t x, intptr_t y,
                   void *dst, ptrdiff_t dst_stride, const void *ref, ptrdiff_t ref_stride, int hbd);
void port_blend(int dir, void *dst, ptrdiff_t dst_stride, const void *tmp, int w, int h, const uint8_t *mask, int hbd);
void port_warp8x8(void *dst, ptrdiff_t dst_stride, int16_t *tmp, ptrdiff_t tmp_stride, const void *src, ptrdiff_t src_stride,
                  const int16_t *abcd, int mx, int my, int bitdepth_max);
void port_mc_scaled(void *dst, ptrdiff_t dst_stride, int16_t *tmp, const void *src, ptrdiff_t src_stride, int w, int h,
                    int mx, int my, int dx, int dy, int filter_2d, int bitdepth_max);
void port_resize(void *dst, ptrdiff_t dst_stride, const void *src, ptrdiff_t src_stride, int dst_w, int h, int src_w, int dx, int mx0,
                 int bitdepth_max);
void port_loop_filter_sb(int chroma, int dir, void *dst, ptrdiff_t stride, const uint32_t *vmask, const uint8_t (*l)[4],
                         ptrdiff_t b4_stride, const uint8_t *lut, int bitdepth_max);
int port_cdef_dir(const void *img, ptrdiff_t stride, unsigned *var, int bitdepth_max);
void port_cdef_fb(int w, int h, void *dst, ptrdiff_t stride, const void *left, const void *top, const void *bottom,
                  int pri, int sec, int dir, int damping, int edges, int bitdepth_max);
void port_wiener(void *p, ptrdiff_t stride, const void *left, const void *lpf, int w, int h, const int16_t filter[2][8], int edges,
                 int bitdepth_max);
void port_sgr(int type, void *p, ptrdiff_t stride, const void *left, const void *lpf, int w, int h, unsigned s0, unsigned s1,
              int w0, int w1, int edges, int bitdepth_max);
void port_intra_pred(int mode, void *dst, ptrdiff_t stride, const void *topleft, int w, int h, int angle, int max_w, int max_h,
                     int bitdepth_max);
void port_cfl_ac(int layout, int16_t *ac, const void *ypx, ptrdiff_t stride, int w_pad, int h_pad, int cw, int ch, int hbd);
void port_cfl_pred(int mode, void *dst, ptrdiff_t stride, const void *topleft, int w, int h, const int16_t *ac, int alpha,
                   int bitdepth_max);
void port_pal_pred(void *dst, ptrdiff_t stride, const void *pal, const uint8_t *idx, int w, int h, int hbd);
#endif

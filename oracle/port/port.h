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
#endif

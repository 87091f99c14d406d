/* One unit row (8 luma rows) of one 64-pixel column of CDEF, as the filter lister writes it and the device expands it into unit
 * records (cdef.hip cdef_expand_kernel): what dav1d_cdef_brow decides per 64x64 — the strengths of its cdef_idx — and per 8x8 — the
 * skip bits (src/cdef_apply_tmpl.c:149-175).  The table of a frame is dense: entry [by8 * w64 + sbx], mask 0 = nothing to filter there.
 * Plain C: shared by host/filter_lister.c and csrc/. */
#ifndef DAV1D_HIP_CDEF_ROWS_H
#define DAV1D_HIP_CDEF_ROWS_H
#include <stdint.h>
typedef struct Dav1dHipCdefRow {
    uint8_t y_pri, y_sec, uv_pri, uv_sec;   /* as Dav1dHipCdefTask */
    uint8_t mask;                           /* bit u: the 8x8 unit bx8 = 8 * sbx + u is filtered */
    uint8_t flags;                          /* DAV1D_HIP_CDEF_BOT_REP_* of the row */
    uint16_t pad;
} Dav1dHipCdefRow;
#endif

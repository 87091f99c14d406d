/* Film grain parameters: same members, order and types as Dav1dFilmGrainData (reference include/dav1d/headers.h:315-333). */
#ifndef ORACLE_FG_PORT_H
#define ORACLE_FG_PORT_H
#include <stdint.h>
typedef struct PortFilmGrain {
    unsigned seed;
    int num_y_points;
    uint8_t y_points[14][2];
    int chroma_scaling_from_luma;
    int num_uv_points[2];
    uint8_t uv_points[2][10][2];
    int scaling_shift;
    int ar_coeff_lag;
    int8_t ar_coeffs_y[24];
    int8_t ar_coeffs_uv[2][25 + 3];
    uint64_t ar_coeff_shift;
    int grain_scale_shift;
    int uv_mult[2];
    int uv_luma_mult[2];
    int uv_offset[2];
    int overlap_flag;
    int clip_to_restricted_range;
} PortFilmGrain;
#endif

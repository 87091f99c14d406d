// Shared device-side definitions of the reconstruction kernels (gfx950 / wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/dav1d_hip.h"

// Planes of one picture as the kernels see them (strides in PIXELS, i.e. the
// reference's PXSTRIDE(), include/common/bitdepth.h:53,78-81).
struct DevPlanes {
    void *data[3];
    int stride[3];
    int w[3], h[3];
};

#define WAVE 64

namespace dv {

__device__ __forceinline__ int iclip(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// a.lo*b.lo + a.hi*b.hi + c on packed int16 pairs (v_dot2c_i32_i16 on gfx950)
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c) {
#ifdef DAV1D_HIP_EMU
    return __builtin_amdgcn_sdot2(a, b, c, false);
#else
    typedef short s2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b), c, false);
#endif
}
// single-instruction forms the compiler does not find on its own for runtime bounds
__device__ __forceinline__ int med3(int v, int lo, int hi) {           // clamp(v, lo, hi) for lo <= hi
#ifdef DAV1D_HIP_EMU
    return v < lo ? lo : v > hi ? hi : v;
#else
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
#endif
}
__device__ __forceinline__ int sub_floor0(int a, int b) {               // max(0, a - b) for a, b >= 0
#ifdef DAV1D_HIP_EMU
    return a > b ? a - b : 0;
#else
    int r;
    asm("v_sub_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}
__device__ __forceinline__ uint32_t pack2(int lo, int hi) { return (uint32_t) (lo & 0xffff) | ((uint32_t) hi << 16); }

// LDS hand-off between the lanes of ONE wave (no other wave reads the data): order the
// accesses and let the wave's outstanding LDS operations land; no s_barrier involved, so
// waves of a workgroup never wait for each other.
__device__ __forceinline__ void wave_sync() {
#ifdef DAV1D_HIP_EMU
    emu_wave_sync();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// Launch-order -> work-order remap.  Workgroup b is observed to land on XCD b % 8
// (MI355X_MICROARCH.md "Workgroup dispatch"); giving XCD k the k-th contiguous
// eighth of the (raster-ordered) task list keeps neighbouring blocks of a picture,
// which share 128-byte lines, inside one XCD's L2.  Speed only, never correctness.
__device__ __forceinline__ unsigned xcd_chunk_id(unsigned b, unsigned nb) {
    const unsigned per = nb >> 3;             // full groups of 8
    if (b >= per * 8) return b;               // ragged tail keeps launch order
    return (b & 7) * per + (b >> 3);
}

} // namespace dv

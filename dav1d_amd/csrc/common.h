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
    int tiled;          // data[] are the planes of the picture's tiled twin (8x8 tiles, mc_body.h); only ever set on references
};
static_assert(sizeof(DevPlanes) == 64, "kernel argument layout");

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

// ---- two 16-bit values per 32-bit register (v_pk_* on gfx950); `2` = one value per half
#ifdef DAV1D_HIP_EMU
#define DV_PK2(expr_lo, expr_hi) ((uint32_t) ((expr_lo) & 0xffff) | ((uint32_t) ((expr_hi) & 0xffff) << 16))
#define DV_LO(v) ((int) (int16_t) ((v) & 0xffff))
#define DV_HI(v) ((int) (int16_t) ((v) >> 16))
#define DV_ULO(v) ((int) ((v) & 0xffff))
#define DV_UHI(v) ((int) ((v) >> 16))
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return DV_PK2(DV_LO(a) + DV_LO(b), DV_HI(a) + DV_HI(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return DV_PK2(DV_LO(a) - DV_LO(b), DV_HI(a) - DV_HI(b)); }
__device__ __forceinline__ uint32_t pk_sub_u16_sat(uint32_t a, uint32_t b) {      // max(0, a - b) on unsigned halves
    return DV_PK2(DV_ULO(a) > DV_ULO(b) ? DV_ULO(a) - DV_ULO(b) : 0, DV_UHI(a) > DV_UHI(b) ? DV_UHI(a) - DV_UHI(b) : 0);
}
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) { return DV_PK2(imin(DV_LO(a), DV_LO(b)), imin(DV_HI(a), DV_HI(b))); }
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) { return DV_PK2(imax(DV_LO(a), DV_LO(b)), imax(DV_HI(a), DV_HI(b))); }
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) { return DV_PK2(imin(DV_ULO(a), DV_ULO(b)), imin(DV_UHI(a), DV_UHI(b))); }
__device__ __forceinline__ uint32_t pk_lshr(uint32_t v, uint32_t sh2) { return DV_PK2(DV_ULO(v) >> (sh2 & 15), DV_UHI(v) >> ((sh2 >> 16) & 15)); }
__device__ __forceinline__ uint32_t pk_ashr(uint32_t v, uint32_t sh2) { return DV_PK2(DV_LO(v) >> (sh2 & 15), DV_HI(v) >> ((sh2 >> 16) & 15)); }
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) {  // a * b + c, low 16 bits per half
    return DV_PK2(DV_LO(a) * DV_LO(b) + DV_LO(c), DV_HI(a) * DV_HI(b) + DV_HI(c));
}
#else
typedef short pk_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u2 __attribute__((ext_vector_type(2)));
#define DV_S2(v) __builtin_bit_cast(dv::pk_s2, (uint32_t) (v))
#define DV_U2(v) __builtin_bit_cast(dv::pk_u2, (uint32_t) (v))
#define DV_R(v) __builtin_bit_cast(uint32_t, (v))
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return DV_R(DV_S2(a) + DV_S2(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return DV_R(DV_S2(a) - DV_S2(b)); }
__device__ __forceinline__ uint32_t pk_sub_u16_sat(uint32_t a, uint32_t b) { return DV_R(__builtin_elementwise_sub_sat(DV_U2(a), DV_U2(b))); }
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) { return DV_R(__builtin_elementwise_min(DV_S2(a), DV_S2(b))); }
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) { return DV_R(__builtin_elementwise_max(DV_S2(a), DV_S2(b))); }
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) { return DV_R(__builtin_elementwise_min(DV_U2(a), DV_U2(b))); }
__device__ __forceinline__ uint32_t pk_lshr(uint32_t v, uint32_t sh2) { return DV_R(DV_U2(v) >> DV_U2(sh2)); }
__device__ __forceinline__ uint32_t pk_ashr(uint32_t v, uint32_t sh2) { return DV_R(DV_S2(v) >> DV_S2(sh2)); }
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) { return DV_R(DV_S2(a) * DV_S2(b) + DV_S2(c)); }
#endif
// clamp with lo <= hi as one instruction (v_med3_i32); the compiler only forms it when both bounds are literals, and the
// select form of iclip() above costs three
__device__ __forceinline__ int clamp3(int v, int lo, int hi) {
#ifdef DAV1D_HIP_EMU
    if (lo > hi) __builtin_trap();
    return v < lo ? lo : v > hi ? hi : v;
#else
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
#endif
}
// full-rate 24-bit multiplies (v_mul_i32_i24 / v_mul_u32_u24): exact when both operands fit 24 bits
__device__ __forceinline__ int mul_i24(int a, int b) {
#ifdef DAV1D_HIP_EMU
    if (a < -(1 << 23) || a >= (1 << 23) || b < -(1 << 23) || b >= (1 << 23)) __builtin_trap();
    return (int) ((unsigned) a * (unsigned) b);
#else
    int r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}
// a * b + c, all three per-lane values
__device__ __forceinline__ int mad_i24(int a, int b, int c) {
#ifdef DAV1D_HIP_EMU
    if (a < -(1 << 23) || a >= (1 << 23) || b < -(1 << 23) || b >= (1 << 23)) __builtin_trap();
    return (int) ((unsigned) a * (unsigned) b + (unsigned) c);
#else
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#endif
}
// i / D for a compile-time D <= 128 and 0 <= i with i * D < 2^18 (D <= 64: i < 4096) without the 32-bit multiplier: with
// M = floor(2^18 / D) + 1, i * M / 2^18 exceeds i / D by at most i / 2^18 < 1 / D, which cannot carry into the quotient
template <int D> __device__ __forceinline__ int div_small(int i) {
    static_assert(D >= 1 && D <= 128, "divisor");
#ifdef DAV1D_HIP_EMU
    if (i < 0 || i * D >= (1 << 18)) __builtin_trap();
    return i / D;
#else
    if constexpr ((D & (D - 1)) == 0) return i >> (31 - __builtin_clz(D));
    else {
        unsigned r;
        asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(i), "v"((1u << 18) / D + 1));
        return (int) (r >> 18);
    }
#endif
}
// a * k + c on the same multiplier (v_mad_i32_i24), k a compile-time constant (it travels in an SGPR: anything that is not
// wave-uniform must not be passed here).  The emulated build checks the 24-bit range of the operands instead of wrapping them.
__device__ __forceinline__ int mad_i24k(int a, int k, int c) {
#ifdef DAV1D_HIP_EMU
    if (a < -(1 << 23) || a >= (1 << 23) || k < -(1 << 23) || k >= (1 << 23)) __builtin_trap();
    return (int) ((unsigned) a * (unsigned) k + (unsigned) c);
#else
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c));
    return r;
#endif
}
__device__ __forceinline__ unsigned mul_u24(unsigned a, unsigned b) {
#ifdef DAV1D_HIP_EMU
    if (a >= (1u << 24) || b >= (1u << 24)) __builtin_trap();
    return a * b;
#else
    unsigned r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}
__device__ __forceinline__ uint32_t rep2(int v) {                     // the same value in both halves
#ifdef DAV1D_HIP_EMU
    return (uint32_t) (v & 0xffff) * 0x10001u;
#else
    return __builtin_amdgcn_perm((uint32_t) v, (uint32_t) v, 0x05040100u);
#endif
}

// D = A x B + C on the matrix cores, int8 operands: A 16 x 64, B 64 x 16, C / D 16 x 16 int32 (v_mfma_i32_16x16x64_i8).
// Lane l supplies 16 bytes of row l & 15 of A and 16 bytes of column l & 15 of B, both for the SAME 16 values of k (chosen
// by l >> 4 and the byte position), and gets D[4 * (l >> 4) + r][l & 15] in c[r].
__device__ __forceinline__ void mfma_i32_16x16x64_i8(const uint4 a, const uint4 b, int c[4]) {
#ifdef DAV1D_HIP_EMU
    emu_mfma_i32_16x16x64_i8(&a.x, &b.x, c);
#else
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i acc = { c[0], c[1], c[2], c[3] };
    const v4i va = { (int) a.x, (int) a.y, (int) a.z, (int) a.w }, vb = { (int) b.x, (int) b.y, (int) b.z, (int) b.w };
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(va, vb, acc, 0, 0, 0);
    c[0] = acc[0]; c[1] = acc[1]; c[2] = acc[2]; c[3] = acc[3];
#endif
}

// ---- data handed from one workgroup to another INSIDE a launch (intra_flow.hip).  A CU's vector L1 is never refreshed by
// another CU's stores and the per-XCD L2s are not coherent with each other for ordinary accesses; relaxed atomics at agent
// scope (global_load / global_store ... sc1) are: the producer stores its pixels with st_coherent, waits for the stores
// (stores_done) and bumps a counter; the consumer polls the counter and reads the pixels with ld_coherent.
template <typename T>
__device__ __forceinline__ T ld_coherent(const T *p) {
#ifdef DAV1D_HIP_EMU
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
template <typename T>
__device__ __forceinline__ void st_coherent(T *p, const T v) {
#ifdef DAV1D_HIP_EMU
    *p = v;
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void stores_done() {       // every vector memory operation of this wave has been acknowledged
#ifndef DAV1D_HIP_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ unsigned ld_rmw(unsigned *p) {          // the value as the atomic unit sees it (fetch-add of 0)
#ifdef DAV1D_HIP_EMU
    return *p;
#else
    return __hip_atomic_fetch_add(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void fence_release_agent() {
#ifndef DAV1D_HIP_EMU
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ void fence_acquire_agent() {
#ifndef DAV1D_HIP_EMU
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
__device__ __forceinline__ void nap_long() {
#ifndef DAV1D_HIP_EMU
    __builtin_amdgcn_s_sleep(127);
#endif
}
__device__ __forceinline__ void nap() {
#ifndef DAV1D_HIP_EMU
    __builtin_amdgcn_s_sleep(2);
#endif
}
// Prefetch in two halves: fetch_begin issues the load of the line at p, fetch_end (any time later) is where the compiler has to
// have the value — put work between them and the trip to memory runs under it.  (A single "touch" whose value is consumed on the
// spot makes the wave wait right there: measured, no gain.)
__device__ __forceinline__ int fetch_begin(const void *p) {
#ifndef DAV1D_HIP_EMU
    return *reinterpret_cast<const volatile int *>(p);
#else
    (void) p;
    return 0;
#endif
}
__device__ __forceinline__ void fetch_end(const int v) {
#ifndef DAV1D_HIP_EMU
    asm volatile("" :: "v"(v));
#else
    (void) v;
#endif
}
__device__ __forceinline__ void touch(const void *p) { fetch_end(fetch_begin(p)); }
// picture pixels behind operator[]: plain loads (COH 0), coherent ones when another workgroup of the same launch wrote them (1), or — the
// "picture" being an image kept in LDS whose generic address travels in a DevPlanes (intra_sb.hip) — LDS reads (2): through the generic
// pointer the compiler would issue flat loads, which wait for the vector memory path as well as for the LDS
template <typename pixel, int COH>
struct PxRead {
    const pixel *p;
    __device__ __forceinline__ pixel operator[](const int i) const {
#ifndef DAV1D_HIP_EMU
        if constexpr (COH == 2) return ((__attribute__((address_space(3))) const pixel *) p)[i];
#endif
        if constexpr (COH == 1) return ld_coherent(p + i);
        return p[i];
    }
    __device__ __forceinline__ PxRead operator+(const int k) const { return PxRead{ p + k }; }
    __device__ __forceinline__ PxRead operator-(const int k) const { return PxRead{ p - k }; }
};

// x, y of a pixel offset in a plane (off < 2^28, stride < 2^15) without the integer divider: one float quotient, corrected
__device__ __forceinline__ void off_to_xy(const uint32_t off, const int stride, int &x, int &y) {
    int q = (int) ((float) off / (float) stride);
    int r = (int) off - q * stride;
    if (r < 0) { q--; r += stride; }
    if (r < 0) { q--; r += stride; }
    if (r >= stride) { q++; r -= stride; }
    if (r >= stride) { q++; r -= stride; }
    x = r; y = q;
}

// This lane's index in its wave
__device__ __forceinline__ int lane_id() { return (int) (threadIdx.x & 63); }

// the value lane `lane` (the same for every lane that asks: a loop counter of an unrolled loop, a scalar) holds, as a scalar
__device__ __forceinline__ int readlane(const int v, const int lane) {
#ifdef DAV1D_HIP_EMU
    return __shfl(v, lane);
#else
    return __builtin_amdgcn_readlane(v, lane);
#endif
}

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

// ---- where a wave's time goes (variant builds only: -DDV_PHASES, tools/build_variant.py).  DV_PHASE_BEGIN() reads the shader clock,
// DV_PHASE(slot) adds the cycles since the previous mark to g_phase[slot] (lane 0 of the wave, one atomic) and DV_PHASE_WAVE(slot) the
// cycles since DV_PHASE_BEGIN.  The marks sit behind the wave_sync that ends a phase (reading the clock also waits for the wave's LDS and
// scalar-memory operations, not for its vector-memory ones), so a slot holds issue + stall of that phase.  Each translation unit has its own g_phase[] and an accessor (dav1d_hip_debug_phases_<unit>).
#if defined(DV_PHASES) && !defined(DAV1D_HIP_EMU)
// (every wave adds into its own word — slot x (workgroup mod DV_PHASE_SPREAD) — : thousands of waves adding into ONE word made the
// instrumented kernels thirty times slower than the plain ones; the accessor sums the words of a slot)
#define DV_PHASE_SLOTS 1024
#define DV_PHASE_SPREAD 8192
namespace { __device__ unsigned long long g_phase[DV_PHASE_SLOTS * DV_PHASE_SPREAD]; }
#define DV_PHASE_DEFINE(unit) DV_PHASE_DEFINE_(unit)
#define DV_PHASE_DEFINE_(unit) \
    extern "C" __attribute__((visibility("default"))) int dav1d_hip_debug_phases_##unit(unsigned long long *out, int reset) { \
        if (out) { \
            unsigned long long *h = new unsigned long long[(size_t) DV_PHASE_SLOTS * DV_PHASE_SPREAD]; \
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(g_phase)) != hipSuccess) { delete[] h; return -5; } \
            for (int s_ = 0; s_ < DV_PHASE_SLOTS; s_++) { unsigned long long t_ = 0; for (int k_ = 0; k_ < DV_PHASE_SPREAD; k_++) t_ += h[(size_t) s_ * DV_PHASE_SPREAD + k_]; out[s_] = t_; } \
            delete[] h; \
        } \
        if (reset) { void *p_ = nullptr; if (hipGetSymbolAddress(&p_, HIP_SYMBOL(g_phase)) != hipSuccess || hipMemset(p_, 0, sizeof(g_phase)) != hipSuccess) return -5; } \
        return 0; }
#define DV_PHASE_AT(slot) g_phase[(size_t) (slot) * DV_PHASE_SPREAD + (blockIdx.x & (DV_PHASE_SPREAD - 1))]
#define DV_PHASE_BEGIN() unsigned long long dv_t0_ = __builtin_amdgcn_s_memtime(), dv_t_ = dv_t0_
#define DV_PHASE(slot) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
        if ((threadIdx.x & 63) == 0) atomicAdd(&DV_PHASE_AT(slot), n_ - dv_t_); dv_t_ = n_; } while (0)
#define DV_PHASE_WAVE(slot) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
        if ((threadIdx.x & 63) == 0) { atomicAdd(&DV_PHASE_AT(slot), n_ - dv_t0_); atomicAdd(&DV_PHASE_AT((slot) + 1), 1ull); } } while (0)
#define DV_PHASE_COUNT(slot, n) do { if ((threadIdx.x & 63) == 0) atomicAdd(&DV_PHASE_AT(slot), (unsigned long long) (n)); } while (0)
#else
#define DV_PHASE_DEFINE(unit)
#define DV_PHASE_BEGIN() do { } while (0)
#define DV_PHASE(slot) do { } while (0)
#define DV_PHASE_WAVE(slot) do { } while (0)
#define DV_PHASE_COUNT(slot, n) do { } while (0)
#endif

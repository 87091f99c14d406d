// SIMT emulator for development WITHOUT a GPU.  TEST INFRASTRUCTURE ONLY.
//
// This header shadows <hip/hip_runtime.h> when the kernel sources under
// dav1d_amd/csrc are compiled with g++ (-Itests/emu first on the include path)
// into tests/emu/libdav1d_hip_emu.so.  Every thread of a workgroup runs as a
// ucontext fiber on one OS thread; __syncthreads() and the wave-level cross-lane
// operations are rendezvous points of the fiber scheduler (emu_rt.cpp).  The
// point is to run the *unmodified* kernel source (indexing, LDS staging, lane
// mapping) against the oracle on the CPU-only build container.  It is never
// linked into, loaded by or substituted for the product library
// (dav1d_amd/libdav1d_hip.so), bench.py or smoke().
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <functional>
#include <algorithm>

#define DAV1D_HIP_EMU 1

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3 { unsigned x, y, z; };

struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline int4 make_int4(int x, int y, int z, int w) { int4 v = { x, y, z, w }; return v; }
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 v = { x, y }; return v; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 v = { x, y, z, w }; return v; }

extern uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__
#endif

typedef int hipError_t;
typedef struct emu_stream_st *hipStream_t;
typedef struct emu_event_st *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorPeerAccessAlreadyEnabled = 704, hipErrorNotSupported = 801, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost,
                     hipMemcpyDeviceToDevice, hipMemcpyDefault };

void emu_syncthreads();
void emu_wave_sync();            // rendezvous of the live lanes of one 64-lane wave
uint64_t *emu_wave_slots();      // 64 x 64-bit exchange slots of the calling wave
void emu_launch(const std::function<void()> &body, dim3 grid, dim3 block);

static inline void __syncthreads() { emu_syncthreads(); }

static inline int emu_lane() {
    return (int) ((threadIdx.x + threadIdx.y * blockDim.x +
                   threadIdx.z * blockDim.x * blockDim.y) & 63);
}

template <typename T> static inline T emu_exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "exchange width");
    uint64_t *s = emu_wave_slots();
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    s[emu_lane()] = raw;
    emu_wave_sync();
    raw = s[src_lane & 63];
    T out;
    memcpy(&out, &raw, sizeof(T));
    emu_wave_sync();
    return out;
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    const int lane = emu_lane();
    return emu_exchange(v, (lane & ~(width - 1)) | (src & (width - 1)));
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    return emu_exchange(v, emu_lane() ^ mask);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    const int lane = emu_lane();
    const int src = lane + (int) d;
    return emu_exchange(v, ((src & ~(width - 1)) == (lane & ~(width - 1))) ? src : lane);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    const int lane = emu_lane();
    const int src = lane - (int) d;
    return emu_exchange(v, (src >= 0 && (src & ~(width - 1)) == (lane & ~(width - 1))) ? src : lane);
}
static inline unsigned long long __ballot(int pred) {
    uint64_t *s = emu_wave_slots();
    s[emu_lane()] = pred ? 1 : 0;
    emu_wave_sync();
    unsigned long long m = 0;
    // lanes that already exited keep a 0 in their slot (cleared at wave start)
    for (int i = 0; i < 64; i++) m |= (unsigned long long) (s[i] & 1) << i;
    emu_wave_sync();
    s[emu_lane()] = 0;
    return m;
}
static inline int atomicAdd(int *p, int v) { const int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
static inline unsigned atomicOr(unsigned *p, unsigned v) { const unsigned o = *p; *p = o | v; return o; }
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) { return __ballot(!pred) == 0; }
static inline int __builtin_amdgcn_readfirstlane(int v) {
    // emulation assumes wave-uniform control flow at the call site: lowest live lane
    uint64_t *s = emu_wave_slots();
    s[emu_lane()] = ((uint64_t) 1 << 63) | (uint32_t) v;
    emu_wave_sync();
    int out = v;
    for (int i = 0; i < 64; i++) if (s[i] >> 63) { out = (int) (uint32_t) s[i]; break; }
    emu_wave_sync();
    s[emu_lane()] = 0;
    return out;
}
static inline int __builtin_amdgcn_sdot2(uint32_t a, uint32_t b, int c, bool) {
    return c + (int16_t) (a & 0xffff) * (int16_t) (b & 0xffff) +
               (int16_t) (a >> 16) * (int16_t) (b >> 16);
}
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
    return (uint32_t) ((((uint64_t) hi << 32) | lo) >> (sh & 31));
}
static inline uint32_t __builtin_amdgcn_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
    // v_perm_b32: result byte i = byte sel[i] of the 8 bytes { s1 (0..3), s0 (4..7) }; selector 12 = 0x00, 13 and above = 0xff
    // (8 .. 11, the sign replications, are not used here)
    const uint64_t both = ((uint64_t) s0 << 32) | s1;
    uint32_t out = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned k = (sel >> (8 * i)) & 0xff;
        const uint32_t byte = k < 8 ? (uint32_t) ((both >> (8 * k)) & 0xff) : k == 12 ? 0u : k >= 13 ? 0xffu : (abort(), 0u);
        out |= byte << (8 * i);
    }
    return out;
}
// v_mfma_i32_16x16x64_i8 as dv::mfma_i32_16x16x64_i8 describes it: every lane publishes its operand bytes, then computes its four
// results from the operands of the lanes that hold the matching row of A / column of B.
static inline void emu_mfma_i32_16x16x64_i8(const uint32_t *a, const uint32_t *b, int *c) {
    uint32_t A[64][4], B[64][4];
    uint64_t *s = emu_wave_slots();
    for (int k = 0; k < 4; k++) {
        s[emu_lane()] = ((uint64_t) b[k] << 32) | a[k];
        emu_wave_sync();
        for (int i = 0; i < 64; i++) { A[i][k] = (uint32_t) s[i]; B[i][k] = (uint32_t) (s[i] >> 32); }
        emu_wave_sync();
    }
    s[emu_lane()] = 0;
    const int l = emu_lane(), col = l & 15;
    for (int r = 0; r < 4; r++) {
        const int row = 4 * (l >> 4) + r;
        int acc = c[r];
        for (int g = 0; g < 4; g++)
            for (int k = 0; k < 16; k++)
                acc += (int) (int8_t) (A[row + 16 * g][k >> 2] >> (8 * (k & 3))) * (int) (int8_t) (B[col + 16 * g][k >> 2] >> (8 * (k & 3)));
        c[r] = acc;
    }
}
static inline int __mul24(int a, int b) { return (int) ((int64_t) ((a << 8) >> 8) * ((b << 8) >> 8)); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (unsigned) ((uint64_t) (a & 0xffffff) * (b & 0xffffff)); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned) v) : 32; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long) v); }
using std::min;
using std::max;

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu_launch([&]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block))

// Devices: $DAV1D_EMU_DEVICES of them (default 1), the current one a property of the calling thread as in HIP.  With more than one,
// every allocation remembers the device it was made on (emu_rt.cpp): hipPointerGetAttributes tells, hipMemcpyPeerAsync insists on it,
// and the library's own checks (a frame's pictures live on the frame's device) have something to check — kernel arguments are opaque,
// a launch itself checks nothing.
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
enum { hipHostMallocDefault = 0 };
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags = 0);
static inline hipError_t hipHostFree(void *p) { return hipFree(p); }
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipGetDeviceCount(int *n);
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void *devicePointer; void *hostPointer; };
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p);
hipError_t hipMemcpyPeerAsync(void *d, int d_dev, const void *s, int s_dev, size_t n, hipStream_t = 0);
static inline hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = 0) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = 0) {
    for (size_t y = 0; y < h; y++) memcpy((char *) d + y * dp, (const char *) s + y * sp, w);
    return hipSuccess;
}
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }      /* launches run to completion inside the launch call */
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = 0; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = 0) { return hipSuccess; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipEventReleaseToDevice = 0x40000000 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = 0; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = 0; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
// stream capture: the emulator runs launches on the spot and cannot record them
typedef struct emu_graph_st *hipGraph_t;
typedef struct emu_graph_exec_st *hipGraphExec_t;
typedef struct emu_graph_node_st *hipGraphNode_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g) { *g = 0; return hipErrorNotSupported; }
static inline hipError_t hipGraphGetNodes(hipGraph_t, hipGraphNode_t *, size_t *n) { *n = 0; return hipErrorNotSupported; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t *, hipGraph_t, hipGraphNode_t *, char *, size_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// How many 128-byte memory lines per nanosecond does an MI355X deliver to scattered requests, and does the answer depend on how much of
// each line is used?  ("re-run the 60-lines/ns calibration with 4 / 32 / 128 useful bytes per scattered request".)
// Every wave issues rounds of U independent 4-byte loads per lane; the lanes of a wave share a line in groups of `lanes per line`
// (1: 4 useful bytes per line, 8: 32 bytes, 32: 128 bytes = the whole line), lines drawn at random from a 2 GiB buffer (far larger than
// the 256 MB last-level cache).  hipcc --offload-arch=gfx950 -O2 tools/calib/line_rate.hip -o tools/calib/line_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int LPL, int U>
__global__ __launch_bounds__(64) void scatter(const uint32_t *__restrict__ buf, const uint32_t n_lines, const int rounds, uint32_t *__restrict__ out) {
    const uint32_t lane = threadIdx.x, grp = lane / LPL, within = lane % LPL;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t line = mix((blockIdx.x * 977u + r) * 64u * U + grp * U + u) % n_lines;
            v[u] = buf[(size_t) line * 32 + within];
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <int LPL>
static void run(const uint32_t *buf, uint32_t n_lines, uint32_t *out, int waves, int rounds) {
    constexpr int U = 8;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((scatter<LPL, U>), dim3(waves), dim3(64), 0, 0, buf, n_lines, 2, out);        // warm
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((scatter<LPL, U>), dim3(waves), dim3(64), 0, 0, buf, n_lines, rounds, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double lines = (double) waves * rounds * U * (64 / LPL), useful = lines * LPL * 4;
    printf("%3d useful bytes per line (%2d lanes per line): %8.1f us, %6.1f lines / ns = %6.2f TB/s of lines, %6.2f TB/s useful (%d waves x %d rounds x %d loads)\n",
           LPL * 4, LPL, ms * 1e3, lines / (ms * 1e6), lines * 128 / (ms * 1e9), useful / (ms * 1e9), waves, rounds, U);
}

int main() {
    const size_t bytes = (size_t) 2 << 30;
    uint32_t *buf = nullptr, *out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 1 << 22) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    const uint32_t n_lines = (uint32_t) (bytes / 128);
    for (int waves : { 19824, 79296 }) {
        run<1>(buf, n_lines, out, waves, 16);
        run<8>(buf, n_lines, out, waves, 64);
        run<32>(buf, n_lines, out, waves, 256);
    }
    return 0;
}

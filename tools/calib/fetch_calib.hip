// FETCH_SIZE calibration on an mc-shaped access pattern (VERDICT r1 #4a): how many HBM bytes does the counter report for
//   stream : every lane reads 16 contiguous bytes, the whole buffer once (the case the guide's x2 correction is stated for),
//   rows   : "prediction windows": W-byte rows (W = 46: 23 pixels of 16 bits, the 8-tap window of a 16-wide block) at 2-byte
//            aligned, otherwise arbitrary positions, 23 rows per window, every window in a region of its own (no line is shared
//            between two windows, nothing is re-read), read as 2-byte elements by consecutive lanes like the gather of mc_body.h.
// The program prints the bytes each kernel touches at 2-byte, 32-byte, 64-byte and 128-byte granularity; run it under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE   and compare (tools/calib/README in profiles/r02_calib.txt).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/calib/fetch_calib tools/calib/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void stream_kernel(const uint4 *__restrict__ p, size_t n, uint32_t *out) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t) gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *out = acc;
}

// one wave per window: rows * row_elems 2-byte elements, lane-consecutive along a row
__global__ void rows_kernel(const uint16_t *__restrict__ p, const uint32_t *__restrict__ start /* element index per window */, int n_win,
                            int rows, int row_elems, int pitch_elems, uint32_t *out)
{
    const int w = blockIdx.x;
    if (w >= n_win) return;
    const uint16_t *s = p + start[w];
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < rows * row_elems; i += 64) acc ^= s[(size_t) (i / row_elems) * pitch_elems + i % row_elems];
    if (acc == 0x12345678u) *out = acc;
}

// the same windows read the way the main gather path of mc_body.h reads them: 8-byte pieces at 4-byte alignment, 6 per 48-byte row
struct __attribute__((packed, aligned(4))) U64 { uint32_t a, b; };
__global__ void rows8_kernel(const uint16_t *__restrict__ p, const uint32_t *__restrict__ start, int n_win, int rows, int pitch_elems, uint32_t *out)
{
    const int w = blockIdx.x;
    if (w >= n_win) return;
    const uint16_t *s = p + (start[w] & ~1u);                       // 4-byte aligned
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < rows * 6; i += 64) {
        const U64 v = *reinterpret_cast<const U64 *>(s + (size_t) (i / 6) * pitch_elems + (i % 6) * 4);
        acc ^= v.a ^ v.b;
    }
    if (acc == 0x12345678u) *out = acc;
}

int main() {
    const size_t bytes = 512u << 20;
    uint8_t *buf; uint32_t *out;
    hipMalloc((void **) &buf, bytes); hipMalloc((void **) &out, 4);
    hipMemset(buf, 1, bytes);
    // windows: region of 32 KB each (23 rows * 1024-byte pitch fits), start at a pseudo-random even byte offset within the first 512 bytes
    const int rows = 23, row_elems = 23, pitch_elems = 512;      // 46-byte rows, 1024-byte pitch
    const int n_win = (int) (bytes / (32 << 10));
    std::vector<uint32_t> start(n_win);
    uint64_t tot2 = 0, tot32 = 0, tot64 = 0, tot128 = 0;
    uint32_t rng = 12345;
    for (int w = 0; w < n_win; w++) {
        rng = rng * 1664525u + 1013904223u;
        const uint32_t off = ((rng >> 8) % 256) * 2;                 // even byte offset 0 .. 510
        start[w] = (uint32_t) (((size_t) w * (32 << 10) + off) / 2);
        for (int r = 0; r < rows; r++) {
            const uint64_t a = (uint64_t) w * (32 << 10) + off + (uint64_t) r * pitch_elems * 2, b = a + row_elems * 2 - 1;
            tot2 += row_elems * 2;
            tot32 += (b / 32 - a / 32 + 1) * 32; tot64 += (b / 64 - a / 64 + 1) * 64; tot128 += (b / 128 - a / 128 + 1) * 128;
        }
    }
    uint32_t *d_start; hipMalloc((void **) &d_start, n_win * 4);
    hipMemcpy(d_start, start.data(), n_win * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, 0, (const uint4 *) buf, bytes / 16, out);
        hipLaunchKernelGGL(rows_kernel, dim3(n_win), dim3(64), 0, 0, (const uint16_t *) buf, d_start, n_win, rows, row_elems, pitch_elems, out);
        hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, 0, (const uint4 *) buf, bytes / 16, out);      // evict
        hipLaunchKernelGGL(rows8_kernel, dim3(n_win), dim3(64), 0, 0, (const uint16_t *) buf, d_start, n_win, rows, pitch_elems, out);
    }
    hipDeviceSynchronize();
    printf("stream_kernel: %llu bytes read once (16 B per lane)\n", (unsigned long long) bytes);
    printf("rows_kernel: %d windows x %d rows x %d bytes: touched %llu B at 2-byte, %llu B at 32-byte, %llu B at 64-byte, %llu B at 128-byte granularity\n",
           n_win, rows, row_elems * 2, (unsigned long long) tot2, (unsigned long long) tot32, (unsigned long long) tot64, (unsigned long long) tot128);
    printf("rows8_kernel: the same windows as 48-byte rows in 8-byte pieces (about the same sectors)\n");
    return 0;
}

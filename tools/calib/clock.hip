// What clock do the SIMDs run at under load?  A kernel that keeps every SIMD busy with integer vector work (v_dot2 / v_mad like the
// reconstruction kernels) for a fixed number of instructions and reads both counters around it: s_memtime (shader clock) and
// s_memrealtime (constant 100 MHz).  Prints MHz for a light and a heavy launch and the instruction rate per SIMD.
// build: hipcc --offload-arch=gfx950 -O2 -o clock clock.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_spin(unsigned long long *out, int iters, int seed) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int a = threadIdx.x + seed, b = blockIdx.x, c = 3, d = 5;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            a = a * 3 + b; b = b * 5 + c; c = c * 7 + d; d = d * 9 + a;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (a + b + c + d == 0x12345) out[0] = 0;
}
int main() {
    unsigned long long *d, *h;
    const int maxwg = 1 << 16;
    hipMalloc(&d, maxwg * 16); h = (unsigned long long *) malloc(maxwg * 16);
    for (int wgs : { 256, 1024, 4096, 8192 }) for (int threads : { 64, 256 }) {
        const int iters = 2000;
        hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(threads), 0, 0, d, iters, 1);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(threads), 0, 0, d, iters, 2);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, wgs * 16, hipMemcpyDeviceToHost);
        double st = 0, sr = 0;
        for (int i = 0; i < wgs; i++) { st += h[2 * i]; sr += h[2 * i + 1]; }
        const double waves = (double) wgs * threads / 64, insts = waves * iters * 64.0;      // 64 v_mad per iteration
        printf("%5d workgroups x %3d threads: kernel %.1f us, memtime / memrealtime = %.3f (x 100 MHz = %.0f MHz if memtime is the shader clock), "
               "VALU instructions per SIMD per us: %.0f (a SIMD at f MHz issues f / 4 per us)\n",
               wgs, threads, ms * 1e3, st / sr, st / sr * 100, insts / 1024 / (ms * 1e3));
    }
    return 0;
}

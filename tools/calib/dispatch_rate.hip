// How fast does the MI355X hand out workgroups?  Empty kernels of N workgroups (64 / 256 threads, with and without LDS, a few VGPRs
// or many), timed with events: the floor under every launch of the recon step that is made of ~20,000 one-wave workgroups.
// build: hipcc --offload-arch=gfx950 -O2 -o dispatch_rate dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDS, int REGS>
__global__ void k_empty(int *out, int n) {
    __shared__ int s[LDS ? LDS / 4 : 1];
    int acc = 0;
    if (LDS) { s[threadIdx.x] = threadIdx.x; acc = s[(threadIdx.x + 1) & 63]; }
    if (REGS) {
        int v[REGS ? REGS : 1];
#pragma unroll
        for (int i = 0; i < REGS; i++) v[i] = acc + i * n;
#pragma unroll
        for (int i = 0; i < REGS; i++) acc ^= v[i] * (i + 3);
    }
    if (n < 0) out[blockIdx.x] = acc;       // never
}
// a wave that lives ~us: a dependent chain of global loads
__global__ void k_chain(const int *p, int *out, int n, int hops) {
    int i = (blockIdx.x * 64 + threadIdx.x) % n;
    for (int h = 0; h < hops; h++) i = p[i];
    if (i == -1) out[0] = i;
}
template <typename F> float timeit(F f, int reps = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; r++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}
int main() {
    int *d; hipMalloc(&d, 1 << 26);
    const int N = 19824;
    printf("empty, 64 thr, no LDS      : %.1f us for %d workgroups\n", timeit([&] { hipLaunchKernelGGL((k_empty<0, 0>), dim3(N), dim3(64), 0, 0, d, 1); }), N);
    printf("empty, 64 thr, 6 KB LDS    : %.1f us\n", timeit([&] { hipLaunchKernelGGL((k_empty<6144, 0>), dim3(N), dim3(64), 0, 0, d, 1); }));
    printf("empty, 64 thr, 6 KB, 64 reg: %.1f us\n", timeit([&] { hipLaunchKernelGGL((k_empty<6144, 64>), dim3(N), dim3(64), 0, 0, d, 1); }));
    printf("empty, 256 thr, 24 KB LDS  : %.1f us for %d workgroups\n", timeit([&] { hipLaunchKernelGGL((k_empty<24576, 0>), dim3(N / 4), dim3(256), 0, 0, d, 1); }), N / 4);
    printf("empty, 64 thr, no LDS x4   : %.1f us for %d workgroups\n", timeit([&] { hipLaunchKernelGGL((k_empty<0, 0>), dim3(4 * N), dim3(64), 0, 0, d, 1); }), 4 * N);
    // chains: n ints forming a random-ish permutation over 64 MB / 4
    const int n = 1 << 24;
    int *h = (int *) malloc(n * 4);
    for (long i = 0; i < n; i++) h[i] = (int) ((i * 2654435761u + 12345u) % n);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    int *o; hipMalloc(&o, 64);
    for (int hops : { 1, 2, 4, 8 })
        printf("chain of %d dependent loads, %d one-wave workgroups: %.1f us\n", hops, N, timeit([&] { hipLaunchKernelGGL(k_chain, dim3(N), dim3(64), 0, 0, d, o, n, hops); }));
    return 0;
}

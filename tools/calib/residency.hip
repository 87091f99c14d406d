// How many one-wave workgroups does a CU really hold at once?  Every workgroup bumps a counter of its own CU (XCC_ID / SE / CU from the
// hardware id registers) on entry, spins for a while, notes the highest count it saw, and drops the counter on exit.  LDS bytes and a
// register budget like the paired reconstruction kernels' (launch bounds 64 x 7: 72 VGPRs).
// build: hipcc --offload-arch=gfx950 -O2 -o residency residency.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int LDS, int MINW>
__global__ __launch_bounds__(64, MINW) void k_res(int *cnt, int *maxseen, int *simd_cnt, int *simd_max, int spin) {
    __shared__ int s[LDS / 4];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, simd = (hw >> 4) & 3;
    const int id = (int) (((xcc & 15) * 8 + se) * 2 + sh) * 16 + cu;
    s[threadIdx.x] = id;
    int seen = 0, sseen = 0;
    if (threadIdx.x == 0) { seen = atomicAdd(&cnt[id], 1) + 1; sseen = atomicAdd(&simd_cnt[id * 4 + simd], 1) + 1; }
    int acc = s[(threadIdx.x + 1) & 63];
    for (int i = 0; i < spin; i++) { acc = acc * 3 + i; if (threadIdx.x == 0 && (i & 63) == 0) { int c = atomicAdd(&cnt[id], 0); seen = c > seen ? c : seen; c = atomicAdd(&simd_cnt[id * 4 + simd], 0); sseen = c > sseen ? c : sseen; } }
    if (threadIdx.x == 0) { atomicMax(&maxseen[id], seen); atomicMax(&simd_max[id * 4 + simd], sseen); atomicSub(&cnt[id], 1); atomicSub(&simd_cnt[id * 4 + simd], 1); }
    if (acc == 0x7fffffff) cnt[0] = 0;
}
template <int LDS, int MINW> void run(const char *what, int n) {
    const int NID = 16 * 8 * 2 * 16;
    int *d; hipMalloc(&d, NID * 10 * 4); hipMemset(d, 0, NID * 10 * 4);
    hipLaunchKernelGGL((k_res<LDS, MINW>), dim3(n), dim3(64), 0, 0, d, d + NID, d + 2 * NID, d + 6 * NID, 20000);
    hipDeviceSynchronize();
    int *h = (int *) malloc(NID * 10 * 4); hipMemcpy(h, d, NID * 10 * 4, hipMemcpyDeviceToHost);
    int cus = 0, mx = 0, mn = 1 << 30; long sum = 0; int smx = 0; long ssum = 0; int simds = 0;
    for (int i = 0; i < NID; i++) if (h[NID + i]) { cus++; sum += h[NID + i]; mx = h[NID + i] > mx ? h[NID + i] : mx; mn = h[NID + i] < mn ? h[NID + i] : mn; }
    for (int i = 0; i < NID * 4; i++) if (h[6 * NID + i]) { simds++; ssum += h[6 * NID + i]; smx = h[6 * NID + i] > smx ? h[6 * NID + i] : smx; }
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_res<LDS, MINW>, 64, 0);
    printf("%-34s %6d workgroups: %d CUs seen, most workgroups at once on a CU: min %d avg %.1f max %d; per SIMD (%d seen): avg %.1f max %d; the runtime's occupancy figure: %d per CU\n",
           what, n, cus, mn, (double) sum / cus, mx, simds, (double) ssum / simds, smx, occ);
    hipFree(d); free(h);
}
int main() {
    run<256, 1>("256 B LDS, no register bound", 20000);
    run<4736, 7>("4.7 KB LDS, 7 waves per SIMD asked", 14425);
    run<5936, 7>("5.9 KB LDS, 7 waves asked", 19464);
    run<8576, 5>("8.6 KB LDS, 5 waves asked", 6540);
    run<8576, 5>("8.6 KB LDS, 5 waves asked", 20000);
    run<16384, 2>("16 KB LDS", 20000);
    return 0;
}

// The two data-parallel members of the reference's Dav1dRefmvsDSPContext on the device (gfx950): splat_mv and save_tmvs
// (reference src/refmvs.c:763-803, 914-923).  SURVEY §8(f)#4.
//
// The reference keeps refmvs_block rows in a ring of 35 rows per tile row and hands the functions an array of row pointers; here
// the map is one frame-level array r[y4 * stride4 + x4] of 12-byte records (== refmvs_block), so a batch of blocks is splatted
// by one launch and the temporal vectors of a whole frame (or any rectangle of 8x8 units) are saved by another.
//   splat : one wave per block, lanes over its 4x4 cells (3 dwords each);
//   save  : one thread per 8x8 unit.  The reference walks a row block by block and copies the candidate of each block's first
//           unit to all of its units; every unit of a block holds the same record after splat_mv, so each unit reading ITS OWN
//           cell (row 2 * y8, column 2 * x8 + 1 — the cell the reference samples) gives the same rp.
#include "common.h"
#include "capi.h"

namespace {

struct RefSign { uint8_t s[8]; };

__global__ __launch_bounds__(64) void splat_mv_kernel(uint32_t *__restrict__ r, const int stride4, const Dav1dHipSplatTask *__restrict__ tasks, const int n)
{
    const int ti = blockIdx.x;
    if (ti >= n) return;
    const Dav1dHipSplatTask t = tasks[ti];
    const int cells = t.bw4 * t.bh4;
    for (int i = threadIdx.x; i < cells; i += 64) {
        const int y = i / t.bw4, x = i - y * t.bw4;
        uint32_t *d = r + ((size_t) (t.by4 + y) * stride4 + t.bx4 + x) * 3;
        d[0] = t.rmv[0]; d[1] = t.rmv[1]; d[2] = t.rmv[2];
    }
}

__device__ __forceinline__ int iabs(const int v) { return v < 0 ? -v : v; }

__global__ __launch_bounds__(256) void save_tmvs_kernel(uint8_t *__restrict__ rp, const int rp_stride, const uint32_t *__restrict__ r, const int stride4,
                                                        const RefSign sign, const int col_start8, const int col_end8, const int row_start8,
                                                        const int row_end8)
{
    const int x = col_start8 + blockIdx.x * 256 + threadIdx.x, y = row_start8 + blockIdx.y;
    if (x >= col_end8 || y >= row_end8) return;
    const uint32_t *c = r + ((size_t) (2 * y) * stride4 + 2 * x + 1) * 3;
    const uint32_t mv0 = c[0], mv1 = c[1], tail = c[2];
    const int ref0 = (int) (int8_t) (tail & 0xff), ref1 = (int) (int8_t) (tail >> 8 & 0xff);
    uint32_t mv = 0;
    int ref = 0;
    // the second reference first, then the first (src/refmvs.c:775-795); a vector qualifies when its reference lies in the
    // past per ref_sign and both components are below 4096 in magnitude
    auto small = [](const uint32_t m) { return (iabs((int) (int16_t) (m & 0xffff)) | iabs((int) (int16_t) (m >> 16))) < 4096; };
    if (ref1 > 0 && sign.s[ref1 - 1] && small(mv1)) { mv = mv1; ref = ref1; }
    else if (ref0 > 0 && sign.s[ref0 - 1] && small(mv0)) { mv = mv0; ref = ref0; }
    uint8_t *o = rp + ((size_t) y * rp_stride + x) * 5;          // refmvs_temporal_block: mv (y, x as int16), ref; 5 bytes, packed
    o[0] = (uint8_t) mv; o[1] = (uint8_t) (mv >> 8); o[2] = (uint8_t) (mv >> 16); o[3] = (uint8_t) (mv >> 24); o[4] = (uint8_t) ref;
}

} // namespace

extern "C" {

// r: DEVICE map of 12-byte refmvs_block records, stride4 records per 4x4 row.  tasks: HOST.
int dav1d_hip_refmvs_splat_batch(Dav1dHipContext *c, void *r, ptrdiff_t stride4, const Dav1dHipSplatTask *tasks, size_t n) {
    if (!c || !r || stride4 <= 0 || (!tasks && n) || n > 0x7fffffff) return -EINVAL;
    if (!n) return 0;
    for (size_t i = 0; i < n; i++) if (!tasks[i].bw4 || !tasks[i].bh4 || tasks[i].bw4 > 32 || tasks[i].bh4 > 32 || tasks[i].bx4 + tasks[i].bw4 > stride4) return -EINVAL;
    // the task array goes through the context's pool of task buffers (no allocation per call once the pool is warm) and the launch is
    // left on the context's stream: whoever reads the map next does so on that stream, or calls dav1d_hip_sync
    TaskBuf buf(c, n * sizeof(Dav1dHipSplatTask));
    Dav1dHipSplatTask *const dev = reinterpret_cast<Dav1dHipSplatTask *>(buf.p);
    if (!dev) return -ENOMEM;
    int rc = dav1d_hip_upload(c, dev, tasks, n * sizeof(*dev));
    if (!rc) {
        hipLaunchKernelGGL(splat_mv_kernel, dim3((unsigned) n), dim3(64), 0, c->stream, (uint32_t *) r, (int) stride4, dev, (int) n);
        rc = hip_rc(hipGetLastError());
    }
    // the buffer returns to the pool when this scope ends: the next user of it uploads on the same stream, behind the launch
    return rc;
}

// rp: DEVICE array of 5-byte refmvs_temporal_block records, rp_stride records per 8x8 row (the reference: ((width + 127) & ~127) >> 3);
// ref_sign == the reference's argument (rf->mfmv_sign); the rectangle is in 8x8 units, as the reference's arguments are.
int dav1d_hip_refmvs_save_tmvs(Dav1dHipContext *c, void *rp, ptrdiff_t rp_stride, const void *r, ptrdiff_t stride4, const uint8_t ref_sign[7],
                               int col_start8, int col_end8, int row_start8, int row_end8) {
    if (!c || !rp || !r || !ref_sign || rp_stride <= 0 || stride4 <= 0 || col_start8 < 0 || row_start8 < 0) return -EINVAL;
    if (col_end8 <= col_start8 || row_end8 <= row_start8) return 0;
    if (col_end8 > rp_stride || 2 * col_end8 > stride4 || row_end8 - row_start8 > 65535) return -EINVAL;
    RefSign s;
    for (int i = 0; i < 7; i++) s.s[i] = ref_sign[i];
    s.s[7] = 0;
    hipLaunchKernelGGL(save_tmvs_kernel, dim3((unsigned) ((col_end8 - col_start8 + 255) / 256), (unsigned) (row_end8 - row_start8)), dim3(256), 0, c->stream,
                       (uint8_t *) rp, (int) rp_stride, (const uint32_t *) r, (int) stride4, s, col_start8, col_end8, row_start8, row_end8);
    return hip_rc(hipGetLastError());       // stream-ordered: no host wait (dav1d_hip_sync, or a download on the context's stream)
}

} // extern "C"

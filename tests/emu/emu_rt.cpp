// Fiber scheduler behind tests/emu/hip/hip_runtime.h.  TEST INFRASTRUCTURE ONLY.
#include "hip/hip_runtime.h"
#include <ucontext.h>
#include <mutex>
#include <vector>

uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {
enum State { RUN, WAIT_BLOCK, WAIT_WAVE, DONE };
struct Fiber {
    ucontext_t ctx;
    char *stack;
    State st;
    uint3 tid;
    int wave;
};
constexpr size_t kStack = 256 * 1024;
std::vector<Fiber> fibers;
std::vector<char *> stack_pool;
ucontext_t sched_ctx;
int cur = -1;
const std::function<void()> *cur_body;
uint64_t wave_slots[16][64];

void trampoline() {
    (*cur_body)();
    fibers[cur].st = DONE;
    swapcontext(&fibers[cur].ctx, &sched_ctx);
}
void yield_to_sched() { swapcontext(&fibers[cur].ctx, &sched_ctx); }

void run_block(const std::function<void()> &body, dim3 block) {
    const int n = (int) (block.x * block.y * block.z);
    if (n > 1024) { fprintf(stderr, "emu: block too large\n"); abort(); }
    fibers.resize(n);
    while ((int) stack_pool.size() < n) stack_pool.push_back((char *) malloc(kStack));
    memset(wave_slots, 0, sizeof(wave_slots));
    cur_body = &body;
    for (int i = 0; i < n; i++) {
        Fiber &f = fibers[i];
        f.stack = stack_pool[i];
        f.st = RUN;
        f.tid.x = i % block.x;
        f.tid.y = (i / block.x) % block.y;
        f.tid.z = i / (block.x * block.y);
        f.wave = i / 64;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = &sched_ctx;
        makecontext(&f.ctx, trampoline, 0);
    }
    int live = n;
    while (live > 0) {
        bool progressed = false;
        for (int i = 0; i < n; i++) {
            if (fibers[i].st != RUN) continue;
            cur = i;
            threadIdx = fibers[i].tid;
            swapcontext(&sched_ctx, &fibers[i].ctx);
            progressed = true;
            if (fibers[i].st == DONE) live--;
        }
        // release wave rendezvous: all live lanes of a wave are waiting on the wave op
        const int nw = (n + 63) / 64;
        for (int w = 0; w < nw; w++) {
            int waiting = 0, alive = 0;
            for (int i = w * 64; i < std::min(n, w * 64 + 64); i++) {
                if (fibers[i].st != DONE) alive++;
                if (fibers[i].st == WAIT_WAVE) waiting++;
            }
            if (alive && waiting == alive) {
                for (int i = w * 64; i < std::min(n, w * 64 + 64); i++)
                    if (fibers[i].st == WAIT_WAVE) fibers[i].st = RUN;
                progressed = true;
            }
        }
        // release block barrier: every live thread is waiting on it
        int waiting = 0, alive = 0;
        for (int i = 0; i < n; i++) {
            if (fibers[i].st != DONE) alive++;
            if (fibers[i].st == WAIT_BLOCK) waiting++;
        }
        if (alive && waiting == alive) {
            for (int i = 0; i < n; i++) if (fibers[i].st == WAIT_BLOCK) fibers[i].st = RUN;
            progressed = true;
        }
        if (!progressed && live > 0) {
            fprintf(stderr, "emu: deadlock (divergent barrier or cross-lane op): "
                            "%d live, %d at block barrier\n", alive, waiting);
            abort();
        }
    }
}
} // namespace

void emu_syncthreads() { fibers[cur].st = WAIT_BLOCK; yield_to_sched(); }
void emu_wave_sync() { fibers[cur].st = WAIT_WAVE; yield_to_sched(); }
uint64_t *emu_wave_slots() { return wave_slots[fibers[cur].wave]; }

// one launch at a time: the scheduler's state (and threadIdx / blockIdx) is global, and host threads with contexts of their own do
// launch side by side (a frame ending on one thread, film grain going onto an output picture on another)
static std::mutex launch_mtx;

void emu_launch(const std::function<void()> &body, dim3 grid, dim3 block) {
    std::lock_guard<std::mutex> lk(launch_mtx);
    blockDim = block;
    gridDim = grid;
    for (unsigned z = 0; z < grid.z; z++)
        for (unsigned y = 0; y < grid.y; y++)
            for (unsigned x = 0; x < grid.x; x++) {
                blockIdx.x = x; blockIdx.y = y; blockIdx.z = z;
                run_block(body, block);
            }
}

// ---- devices and the allocations made on them (see hip_runtime.h)
#include <map>
namespace {
int n_devices() {
    static const int n = []() { const char *e = getenv("DAV1D_EMU_DEVICES"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v > 16 ? 16 : v; }();
    return n;
}
thread_local int cur_device = 0;
struct Alloc { size_t n; int dev; };
std::mutex alloc_mtx;
std::map<uintptr_t, Alloc> allocs;       // only kept with more than one device
const Alloc *find_alloc(const void *p, uintptr_t *base) {
    const uintptr_t a = (uintptr_t) p;
    auto it = allocs.upper_bound(a);
    if (it == allocs.begin()) return nullptr;
    --it;
    if (a >= it->first + it->second.n) return nullptr;
    if (base) *base = it->first;
    return &it->second;
}
hipError_t alloc_on(void **p, size_t n, int dev) {
    *p = malloc(n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    if (n_devices() > 1) { std::lock_guard<std::mutex> lk(alloc_mtx); allocs[(uintptr_t) *p] = Alloc{ n ? n : 1, dev }; }
    return hipSuccess;
}
}
hipError_t hipSetDevice(int d) { if (d < 0 || d >= n_devices()) return hipErrorInvalidValue; cur_device = d; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = cur_device; return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = n_devices(); return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { return alloc_on(p, n, cur_device); }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return alloc_on(p, n, -1); }
hipError_t hipFree(void *p) {
    if (p && n_devices() > 1) { std::lock_guard<std::mutex> lk(alloc_mtx); allocs.erase((uintptr_t) p); }
    free(p);
    return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p) {
    a->type = hipMemoryTypeUnregistered; a->device = 0; a->devicePointer = a->hostPointer = nullptr;
    if (n_devices() == 1) { a->type = hipMemoryTypeDevice; a->devicePointer = const_cast<void *>(p); return hipSuccess; }      // nothing is tracked
    std::lock_guard<std::mutex> lk(alloc_mtx);
    const Alloc *al = find_alloc(p, nullptr);
    if (!al) return hipErrorInvalidValue;
    if (al->dev < 0) { a->type = hipMemoryTypeHost; a->hostPointer = const_cast<void *>(p); }
    else { a->type = hipMemoryTypeDevice; a->device = al->dev; a->devicePointer = const_cast<void *>(p); }
    return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void *d, int d_dev, const void *s, int s_dev, size_t n, hipStream_t) {
    if (n_devices() > 1) {
        std::lock_guard<std::mutex> lk(alloc_mtx);
        uintptr_t bd = 0, bs = 0;
        const Alloc *ad = find_alloc(d, &bd), *as = find_alloc(s, &bs);
        if (!ad || !as || ad->dev != d_dev || as->dev != s_dev) return hipErrorInvalidValue;
        if ((uintptr_t) d + n > bd + ad->n || (uintptr_t) s + n > bs + as->n) return hipErrorInvalidValue;
    }
    memcpy(d, s, n);
    return hipSuccess;
}

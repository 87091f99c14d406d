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

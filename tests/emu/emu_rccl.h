// A stand-in for the handful of RCCL entry points csrc/peer.hip uses, for the SIMT-emulated build: ranks are processes of ONE host
// that meet in a POSIX shared-memory segment named after the unique id.  TEST INFRASTRUCTURE ONLY (tests/test_dist.py runs the C
// peer entry points on two CPU ranks through it); the GPU build resolves the real librccl.so.
#pragma once
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#include <atomic>

typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclSystemError = 2 };
typedef int ncclDataType_t;
enum { ncclUint8 = 1 };
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;

struct EmuNcclShared {
    std::atomic<uint32_t> arrived[16];        // per collective phase: ranks that have posted
    std::atomic<uint64_t> seq[8];             // per rank: collectives entered
    std::atomic<uint64_t> done[8];            // per rank: collectives left
    size_t slot_bytes;
    // then 8 slots of slot_bytes
};
struct EmuNcclComm { EmuNcclShared *sh; uint8_t *slots; int rank, world; size_t slot_bytes; char name[64]; uint64_t n; bool in_group; };
typedef EmuNcclComm *ncclComm_t;
enum { EMU_NCCL_SLOT = 64u << 20 };

static inline ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof(*id));
    struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof(id->internal), "/dav1d_emu_nccl_%d_%ld_%ld", (int) getpid(), (long) ts.tv_sec, (long) ts.tv_nsec);
    return ncclSuccess;
}
static inline ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > 8) return ncclSystemError;
    EmuNcclComm *c = new EmuNcclComm();
    c->rank = rank; c->world = nranks; c->slot_bytes = EMU_NCCL_SLOT; c->n = 0; c->in_group = false;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    const size_t total = 4096 + (size_t) nranks * c->slot_bytes;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t) total)) { delete c; return ncclSystemError; }
    void *p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = (EmuNcclShared *) p; c->slots = (uint8_t *) p + 4096;
    *out = c;
    return ncclSuccess;
}
static inline ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    munmap(c->sh, 4096 + (size_t) c->world * c->slot_bytes);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return ncclSuccess;
}
// every rank posts `bytes` into its slot, waits for all, then reads what it needs, then waits for all to have read
static inline void emu_nccl_post(ncclComm_t c, const void *src, size_t bytes) {
    if (src && bytes) memcpy(c->slots + (size_t) c->rank * c->slot_bytes, src, bytes);
    c->n++;
    c->sh->seq[c->rank].store(c->n, std::memory_order_release);
    for (int r = 0; r < c->world; r++) while (c->sh->seq[r].load(std::memory_order_acquire) < c->n) usleep(50);
}
static inline void emu_nccl_leave(ncclComm_t c) {
    c->sh->done[c->rank].store(c->n, std::memory_order_release);
    for (int r = 0; r < c->world; r++) while (c->sh->done[r].load(std::memory_order_acquire) < c->n) usleep(50);
}
static inline ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t, ncclComm_t c, void *) {
    if (count > c->slot_bytes) return ncclSystemError;
    emu_nccl_post(c, send, count);
    for (int r = 0; r < c->world; r++) memcpy((uint8_t *) recv + (size_t) r * count, c->slots + (size_t) r * c->slot_bytes, count);
    emu_nccl_leave(c);
    return ncclSuccess;
}
static inline ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t, int root, ncclComm_t c, void *) {
    // in pieces of one slot
    for (size_t off = 0; off < count; off += c->slot_bytes) {
        const size_t n = count - off < c->slot_bytes ? count - off : c->slot_bytes;
        emu_nccl_post(c, c->rank == root ? (const uint8_t *) send + off : nullptr, c->rank == root ? n : 0);
        if (c->rank != root || recv != send) memcpy((uint8_t *) recv + off, c->slots + (size_t) root * c->slot_bytes, n);
        emu_nccl_leave(c);
    }
    return ncclSuccess;
}
// Point-to-point inside a group (the halo exchange): sends are parked until ncclGroupEnd, which every rank of the communicator
// enters (the callers here do: a rank without a neighbour on one side simply has fewer operations): each rank writes its messages
// into its own slot, one segment per destination, all meet, each reads the segments addressed to it, all meet again.
struct EmuP2P { int peer; void *ptr; size_t n; bool send; };
static thread_local EmuP2P emu_p2p[16];
static thread_local int emu_p2p_n;
static thread_local ncclComm_t emu_p2p_comm;
static inline ncclResult_t ncclGroupStart(void) { emu_p2p_n = 0; emu_p2p_comm = nullptr; return ncclSuccess; }
static inline ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t, int peer, ncclComm_t c, void *) {
    if (emu_p2p_n >= 16) return ncclSystemError;
    emu_p2p[emu_p2p_n++] = EmuP2P{ peer, const_cast<void *>(buf), count, true };
    emu_p2p_comm = c;
    return ncclSuccess;
}
static inline ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t, int peer, ncclComm_t c, void *) {
    if (emu_p2p_n >= 16) return ncclSystemError;
    emu_p2p[emu_p2p_n++] = EmuP2P{ peer, buf, count, false };
    emu_p2p_comm = c;
    return ncclSuccess;
}
static inline ncclResult_t emu_nccl_group_end(ncclComm_t c) {
    const size_t seg = c->slot_bytes / 8;
    for (int i = 0; i < emu_p2p_n; i++)
        if (emu_p2p[i].send) {
            if (emu_p2p[i].n > seg) return ncclSystemError;
            memcpy(c->slots + (size_t) c->rank * c->slot_bytes + (size_t) emu_p2p[i].peer * seg, emu_p2p[i].ptr, emu_p2p[i].n);
        }
    emu_nccl_post(c, nullptr, 0);
    for (int i = 0; i < emu_p2p_n; i++)
        if (!emu_p2p[i].send) memcpy(emu_p2p[i].ptr, c->slots + (size_t) emu_p2p[i].peer * c->slot_bytes + (size_t) c->rank * seg, emu_p2p[i].n);
    emu_nccl_leave(c);
    emu_p2p_n = 0;
    return ncclSuccess;
}

// Peer-push all-gather over HIP IPC: a provider of lg_comm::all_gather that needs no collective library (include/ligero_hip.h,
// "A second provider of lg_comm::all_gather").  SURVEY 8(e) step 2: the coefficient rows of a coset-sharded commit are
// all-gathered in place -- every rank holds block `rank` of `world` equal blocks and needs the rest.  Here every rank WRITES its
// block into the same place of every peer's buffer:
//
//   all_gather(device_buf, bytes_per_rank, stream)                        (same call, same order, on every rank)
//     0. the allocation holding device_buf is mapped into the peers on first use: {ipc handle, offset} all-gathered over the
//        caller's out-of-band channel, hipIpcOpenMemHandle per peer
//     1. record ev_ready on `stream`      -- whatever read the buffer's old contents was queued before this call
//        host barrier                     -- every rank has recorded (an event must be recorded before a peer's wait is queued)
//     2. per peer p: stream waits ev_ready[p]; hipMemcpyAsync(peer_buf[p] + rank * B, device_buf + rank * B, B) on `stream`
//        record ev_pushed on `stream`; host barrier
//     3. per peer p: stream waits ev_pushed[p]      -- the peers' blocks are in place for everything queued after the call
//
// The host never waits for the device; it meets the other hosts twice per exchange.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

// the library is built with -fvisibility=hidden: only what the public header declares is exported
#pragma GCC visibility push(default)
#include "../../include/ligero_hip.h"
#pragma GCC visibility pop

struct lg_push_comm {
    int device = 0;
    uint32_t world = 1, rank = 0;
    lg_push_bootstrap boot{};
    hipEvent_t ev_ready = nullptr, ev_pushed = nullptr;              // mine (interprocess)
    std::vector<hipEvent_t> peer_ready, peer_pushed;                 // the peers', opened (own slot unused)
    struct Mapping { uint8_t* base; size_t size; std::vector<uint8_t*> peer; };   // one allocation of mine and where each peer's twin is mapped here
    std::vector<Mapping> maps;
    char err[256] = {0};
};

namespace {

struct Hello { hipIpcEventHandle_t ready, pushed; };
struct BufHello { hipIpcMemHandle_t mem; uint64_t offset, size; };

int fail(lg_push_comm* pc, hipError_t e, const char* what) {
    snprintf(pc->err, sizeof(pc->err), "%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return -1;
}
#define PC_HIP(pc, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(pc, e_, #call); } while (0)

int boot_fail(lg_push_comm* pc, const char* what, int rc) {
    snprintf(pc->err, sizeof(pc->err), "bootstrap %s returned %d", what, rc);
    return -1;
}

// the mapping of the allocation that holds [p, p + bytes), made on first use (collective)
int mapping_of(lg_push_comm* pc, uint8_t* p, size_t bytes, lg_push_comm::Mapping** out) {
    for (auto& m : pc->maps)
        if (p >= m.base && p + bytes <= m.base + m.size) { *out = &m; return 0; }
    void* base = nullptr;
    size_t size = 0;
    PC_HIP(pc, hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &size, p));
    if (p + bytes > static_cast<uint8_t*>(base) + size) { snprintf(pc->err, sizeof(pc->err), "the exchanged range leaves its allocation"); return -1; }
    BufHello mine;
    memset(&mine, 0, sizeof(mine));
    PC_HIP(pc, hipIpcGetMemHandle(&mine.mem, base));
    mine.offset = 0;
    mine.size = size;
    std::vector<BufHello> all(pc->world);
    if (const int rc = pc->boot.all_gather_host(pc->boot.user, &mine, all.data(), sizeof(BufHello)); rc != 0) return boot_fail(pc, "all_gather_host", rc);
    lg_push_comm::Mapping m;
    m.base = static_cast<uint8_t*>(base);
    m.size = size;
    m.peer.assign(pc->world, nullptr);
    for (uint32_t r = 0; r < pc->world; r++) {
        if (r == pc->rank) continue;
        if (all[r].size < (size_t)(p - m.base) + bytes) { snprintf(pc->err, sizeof(pc->err), "rank %u's buffer of this role is smaller than the exchanged range", r); return -1; }
        void* q = nullptr;
        PC_HIP(pc, hipIpcOpenMemHandle(&q, all[r].mem, hipIpcMemLazyEnablePeerAccess));
        m.peer[r] = static_cast<uint8_t*>(q) + all[r].offset;
    }
    pc->maps.push_back(std::move(m));
    *out = &pc->maps.back();
    return 0;
}

int push_all_gather(void* user, void* device_buf, uint64_t bytes_per_rank, void* stream) {
    lg_push_comm* pc = static_cast<lg_push_comm*>(user);
    if (!pc || !device_buf) return -1;
    if (pc->world == 1) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipSetDevice(pc->device) != hipSuccess) return fail(pc, hipGetLastError(), "hipSetDevice");
    uint8_t* buf = static_cast<uint8_t*>(device_buf);
    lg_push_comm::Mapping* m = nullptr;
    if (const int rc = mapping_of(pc, buf, (size_t)bytes_per_rank * pc->world, &m); rc != 0) return rc;
    const size_t off = (size_t)(buf - m->base) + (size_t)pc->rank * bytes_per_rank;
    PC_HIP(pc, hipEventRecord(pc->ev_ready, s));
    if (const int rc = pc->boot.barrier(pc->boot.user); rc != 0) return boot_fail(pc, "barrier", rc);
    for (uint32_t r = 0; r < pc->world; r++) {
        if (r == pc->rank) continue;
        PC_HIP(pc, hipStreamWaitEvent(s, pc->peer_ready[r], 0));
        if (bytes_per_rank) PC_HIP(pc, hipMemcpyAsync(m->peer[r] + off, m->base + off, bytes_per_rank, hipMemcpyDeviceToDevice, s));
    }
    PC_HIP(pc, hipEventRecord(pc->ev_pushed, s));
    if (const int rc = pc->boot.barrier(pc->boot.user); rc != 0) return boot_fail(pc, "barrier", rc);
    for (uint32_t r = 0; r < pc->world; r++)
        if (r != pc->rank) PC_HIP(pc, hipStreamWaitEvent(s, pc->peer_pushed[r], 0));
    return 0;
}

}  // namespace

extern "C" {

static char g_push_create_err[256] = "";

const char* lg_push_comm_last_error(const lg_push_comm* pc) { return pc ? pc->err : g_push_create_err; }

void lg_push_comm_destroy(lg_push_comm* pc) {
    if (!pc) return;
    (void)hipSetDevice(pc->device);
    if (pc->world > 1 && pc->boot.barrier) (void)pc->boot.barrier(pc->boot.user);      // every rank is done pushing and waiting
    for (auto& m : pc->maps)
        for (uint32_t r = 0; r < pc->world; r++)
            if (r != pc->rank && m.peer[r]) (void)hipIpcCloseMemHandle(m.peer[r]);
    for (uint32_t r = 0; r < pc->world && r < pc->peer_ready.size(); r++) {
        if (r == pc->rank) continue;
        if (pc->peer_ready[r]) (void)hipEventDestroy(pc->peer_ready[r]);
        if (pc->peer_pushed[r]) (void)hipEventDestroy(pc->peer_pushed[r]);
    }
    if (pc->world > 1 && pc->boot.barrier) (void)pc->boot.barrier(pc->boot.user);      // ... and has unmapped, before anyone frees
    if (pc->ev_ready) (void)hipEventDestroy(pc->ev_ready);
    if (pc->ev_pushed) (void)hipEventDestroy(pc->ev_pushed);
    delete pc;
}

int lg_push_comm_create(lg_push_comm** out, int device, uint32_t world, uint32_t rank, const lg_push_bootstrap* boot) {
    if (!out || world == 0 || rank >= world || (world > 1 && (!boot || !boot->all_gather_host || !boot->barrier))) return LG_ERR_BAD_ARG;
    *out = nullptr;
    lg_push_comm* pc = new (std::nothrow) lg_push_comm();
    if (!pc) return LG_ERR_OOM;
    pc->device = device; pc->world = world; pc->rank = rank;
    if (boot) pc->boot = *boot;
    auto body = [&]() -> int {
        PC_HIP(pc, hipSetDevice(device));
        if (world == 1) return 0;
        PC_HIP(pc, hipEventCreateWithFlags(&pc->ev_ready, hipEventDisableTiming | hipEventInterprocess));
        PC_HIP(pc, hipEventCreateWithFlags(&pc->ev_pushed, hipEventDisableTiming | hipEventInterprocess));
        Hello mine;
        memset(&mine, 0, sizeof(mine));
        PC_HIP(pc, hipIpcGetEventHandle(&mine.ready, pc->ev_ready));
        PC_HIP(pc, hipIpcGetEventHandle(&mine.pushed, pc->ev_pushed));
        std::vector<Hello> all(world);
        if (const int rc = pc->boot.all_gather_host(pc->boot.user, &mine, all.data(), sizeof(Hello)); rc != 0) return boot_fail(pc, "all_gather_host", rc);
        pc->peer_ready.assign(world, nullptr);
        pc->peer_pushed.assign(world, nullptr);
        for (uint32_t r = 0; r < world; r++) {
            if (r == rank) continue;
            PC_HIP(pc, hipIpcOpenEventHandle(&pc->peer_ready[r], all[r].ready));
            PC_HIP(pc, hipIpcOpenEventHandle(&pc->peer_pushed[r], all[r].pushed));
        }
        return 0;
    };
    if (body() != 0) {
        snprintf(g_push_create_err, sizeof(g_push_create_err), "%s", pc->err);
        pc->boot.barrier = nullptr;       // the peers may not be there to meet: release without the barriers
        lg_push_comm_destroy(pc);
        return LG_ERR_HIP;
    }
    *out = pc;
    return LG_OK;
}

int lg_push_comm_bind(lg_push_comm* pc, lg_comm* comm, uint32_t flags) {
    if (!pc || !comm) return LG_ERR_BAD_ARG;
    comm->world = pc->world;
    comm->rank = pc->rank;
    comm->flags = flags;
    comm->user = pc;
    comm->all_gather = push_all_gather;
    return LG_OK;
}

}  // extern "C"

// Collective layer of the multi-GPU path (include/openmm_hip_comm.h): RCCL over xGMI, and a host-staged callback
// transport used by tests.  One process per GPU; every call is made by all ranks in the same order.
//
// RCCL is bound at run time (dlopen of librccl.so.1 -- PyTorch-ROCm ships the same SONAME, so inside a torch process the
// copy that is already mapped is the one that is used) and all traffic is enqueued on the caller's stream:
//   all-gather   -> ncclAllGather (in place)
//   all-to-all   -> ncclGroupStart, size x (ncclSend, ncclRecv), ncclGroupEnd: every peer pair has an xGMI link of its own,
//                   so the size-1 transfers of a rank proceed concurrently
//   ring         -> one group with two sends and two receives (two links)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <vector>
#include "../../../include/openmm_hip_comm.h"

#ifndef OMMHIP_EMU
#include <dlfcn.h>
#include <rccl/rccl.h>
#endif

struct ommhip_comm {
    int rank = 0, size = 1;
    bool rccl = false;
    // "alone" transport (diagnostics): this rank of `size` exists, its peers do not -- every collective returns at once and moves nothing
    bool alone = false;
    // callback transport
    ommhip_host_all_gather_fn fn = nullptr;
    void* user = nullptr;
    std::vector<char> hostSend, hostRecv;
    // rccl transport
    void* nccl = nullptr;
    void* smallDev = nullptr;          // staging for ommhip_comm_all_gather_host
    size_t smallBytes = 0;
};

namespace {

#ifndef OMMHIP_EMU
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclAllGather) allGather = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGroupStart) groupStart = nullptr;
    decltype(&ncclGroupEnd) groupEnd = nullptr;
    decltype(&ncclCommSplit) commSplit = nullptr;      // optional
    bool ok = false;
};

RcclApi& rccl_api() {
    static RcclApi api;
    if (api.handle != nullptr || api.ok) return api;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (api.handle != nullptr) break;
    }
    if (api.handle == nullptr) { fprintf(stderr, "HIP platform: cannot open librccl: %s\n", dlerror()); return api; }
#define OMM_BIND(field, sym) api.field = (decltype(api.field)) dlsym(api.handle, sym)
    OMM_BIND(getUniqueId, "ncclGetUniqueId"); OMM_BIND(commInitRank, "ncclCommInitRank"); OMM_BIND(commDestroy, "ncclCommDestroy");
    OMM_BIND(allGather, "ncclAllGather"); OMM_BIND(send, "ncclSend"); OMM_BIND(recv, "ncclRecv");
    OMM_BIND(groupStart, "ncclGroupStart"); OMM_BIND(groupEnd, "ncclGroupEnd"); OMM_BIND(commSplit, "ncclCommSplit");
#undef OMM_BIND
    api.ok = api.getUniqueId && api.commInitRank && api.commDestroy && api.allGather && api.send && api.recv && api.groupStart && api.groupEnd;
    if (!api.ok) fprintf(stderr, "HIP platform: librccl lacks a required symbol\n");
    return api;
}
inline int nccl_rc(ncclResult_t r) { return r == ncclSuccess ? 0 : 1000 + (int) r; }
#define NCCL_TRY(call) do { int rc__ = nccl_rc(call); if (rc__ != 0) return rc__; } while (0)
#endif

// Diagnostics of the callback transport (OMMHIP_COMM_DIAG=1, bench.py --serialize-ranks): wall time spent inside collectives,
// counted from the moment the device is idle -- staging copies, the callback and whatever the callback waits for.  A rank's wall
// time minus this is its step without communication.
double g_commSeconds = 0.0;
struct CommDiag {
    bool on;
    std::chrono::steady_clock::time_point t0;
    CommDiag() {
        static const bool enabled = getenv("OMMHIP_COMM_DIAG") != nullptr;
        on = enabled;
        if (on) { hipDeviceSynchronize(); t0 = std::chrono::steady_clock::now(); }
    }
    ~CommDiag() { if (on) g_commSeconds += 1e-9 * (double) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};

// host staging of the callback transport: device -> pinned-less host vector (blocking), callback, host -> device
int stage_down(ommhip_comm* c, const void* src_d, size_t bytes, size_t offset, hipStream_t st) {
    if (c->hostSend.size() < offset + bytes) c->hostSend.resize(offset + bytes);
    hipError_t e = hipMemcpyAsync(c->hostSend.data() + offset, src_d, bytes, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return (int) e;
    return (int) hipStreamSynchronize(st);
}
int stage_up(void* dst_d, const char* src, size_t bytes, hipStream_t st) {
    hipError_t e = hipMemcpyAsync(dst_d, src, bytes, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return (int) e;
    return (int) hipStreamSynchronize(st);
}

// forces a neighbour computed on this rank's atoms (ommhip_comm_halo_return): component-major staging -> the SoA force buffer
__global__ void k_add_returned(long long* force, int paddedSlots, int first, int count, const long long* staging) {
    const size_t g = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 3 * (size_t) count) return;
    const size_t k = g / count, s = g % count;
    force[k * paddedSlots + first + s] += staging[g];
}

}  // namespace

extern "C" {

int ommhip_comm_unique_id(char* hex) {
#ifdef OMMHIP_EMU
    (void) hex;
    return 1;
#else
    RcclApi& api = rccl_api();
    if (!api.ok) return 1;
    ncclUniqueId id;
    NCCL_TRY(api.getUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    const unsigned char* b = (const unsigned char*) &id;
    for (int i = 0; i < 128; i++) snprintf(hex + 2 * i, 3, "%02x", b[i]);
    hex[256] = 0;
    return 0;
#endif
}

int ommhip_comm_create_rccl(const char* id_hex, int rank, int size, ommhip_comm** comm) {
#ifdef OMMHIP_EMU
    (void) id_hex; (void) rank; (void) size; (void) comm;
    return 1;
#else
    RcclApi& api = rccl_api();
    if (!api.ok || id_hex == nullptr || strlen(id_hex) != 256 || rank < 0 || rank >= size) return 1;
    ncclUniqueId id;
    unsigned char* b = (unsigned char*) &id;
    for (int i = 0; i < 128; i++) {
        unsigned v = 0;
        if (sscanf(id_hex + 2 * i, "%2x", &v) != 1) return 1;
        b[i] = (unsigned char) v;
    }
    ncclComm_t nc;
    NCCL_TRY(api.commInitRank(&nc, size, id, rank));
    ommhip_comm* c = new ommhip_comm();
    c->rank = rank; c->size = size; c->rccl = true; c->nccl = (void*) nc;
    *comm = c;
    return 0;
#endif
}

int ommhip_comm_create_callback(ommhip_host_all_gather_fn fn, void* user, int rank, int size, ommhip_comm** comm) {
    if (fn == nullptr || rank < 0 || rank >= size) return 1;
    ommhip_comm* c = new ommhip_comm();
    c->rank = rank; c->size = size; c->rccl = false; c->fn = fn; c->user = user;
    *comm = c;
    return 0;
}

int ommhip_comm_create_alone(int rank, int size, ommhip_comm** comm) {
    if (rank < 0 || rank >= size) return 1;
    ommhip_comm* c = new ommhip_comm();
    c->rank = rank; c->size = size; c->alone = true;
    *comm = c;
    return 0;
}

int ommhip_comm_duplicate(ommhip_comm* comm, ommhip_comm** copy) {
    ommhip_comm* c = new ommhip_comm();
    c->rank = comm->rank; c->size = comm->size; c->rccl = comm->rccl; c->fn = comm->fn; c->user = comm->user; c->alone = comm->alone;
#ifndef OMMHIP_EMU
    if (comm->rccl) {
        RcclApi& api = rccl_api();
        if (api.commSplit == nullptr) { delete c; return 1; }
        ncclComm_t nc;
        int rc = nccl_rc(api.commSplit((ncclComm_t) comm->nccl, 0, comm->rank, &nc, nullptr));
        if (rc != 0) { delete c; return rc; }
        c->nccl = (void*) nc;
    }
#endif
    *copy = c;
    return 0;
}

int ommhip_comm_destroy(ommhip_comm* comm) {
    if (comm == nullptr) return 0;
#ifndef OMMHIP_EMU
    if (comm->rccl && comm->nccl != nullptr) rccl_api().commDestroy((ncclComm_t) comm->nccl);
#endif
    if (comm->smallDev != nullptr) hipFree(comm->smallDev);
    delete comm;
    return 0;
}

int ommhip_comm_rank(const ommhip_comm* comm) { return comm->rank; }
int ommhip_comm_size(const ommhip_comm* comm) { return comm->size; }
const char* ommhip_comm_transport(const ommhip_comm* comm) { return comm->alone ? "alone" : (comm->rccl ? "rccl" : "callback"); }

int ommhip_comm_all_gather(ommhip_comm* c, void* buffer_d, size_t bytes, void* stream) {
    if (bytes == 0 || c->alone) return 0;
    hipStream_t st = (hipStream_t) stream;
    char* buf = (char*) buffer_d;
#ifndef OMMHIP_EMU
    if (c->rccl) {
        // xGMI is point-to-point: every pair of the node's GPUs has a link of its own.  A ring all-gather pushes (N-1)/N of the
        // whole buffer through each link, hop after hop; sending this rank's part straight to its N-1 peers (one grouped
        // send/recv, the primitive the all-to-all uses) moves 1/N of it over each of the N-1 links at the same time, one hop.
        // OPENMM_HIP_ALLGATHER=ring selects ncclAllGather (also used with one rank, where it exercises the real transport).
        static const bool ring = getenv("OPENMM_HIP_ALLGATHER") != nullptr && strcmp(getenv("OPENMM_HIP_ALLGATHER"), "ring") == 0;
        RcclApi& api = rccl_api();
        if (ring || c->size == 1) {
            NCCL_TRY(api.allGather(buf + (size_t) c->rank * bytes, buf, bytes, ncclChar, (ncclComm_t) c->nccl, st));
            return 0;
        }
        NCCL_TRY(api.groupStart());
        for (int p = 0; p < c->size; p++) {
            if (p == c->rank) continue;
            NCCL_TRY(api.send(buf + (size_t) c->rank * bytes, bytes, ncclChar, p, (ncclComm_t) c->nccl, st));
            NCCL_TRY(api.recv(buf + (size_t) p * bytes, bytes, ncclChar, p, (ncclComm_t) c->nccl, st));
        }
        NCCL_TRY(api.groupEnd());
        return 0;
    }
#endif
    if (c->size == 1) return 0;
    CommDiag diag;
    int rc = stage_down(c, buf + (size_t) c->rank * bytes, bytes, 0, st);
    if (rc != 0) return rc;
    c->hostRecv.resize((size_t) c->size * bytes);
    if (c->fn(c->user, c->hostSend.data(), c->hostRecv.data(), bytes) != 0) return 1;
    return stage_up(buf, c->hostRecv.data(), (size_t) c->size * bytes, st);
}

int ommhip_comm_all_to_all(ommhip_comm* c, const void* send_d, void* recv_d, size_t bytes, void* stream) {
    hipStream_t st = (hipStream_t) stream;
    if (bytes == 0 || c->alone) return 0;
    const char* s = (const char*) send_d;
    char* r = (char*) recv_d;
#ifndef OMMHIP_EMU
    if (c->rccl) {
        RcclApi& api = rccl_api();
        NCCL_TRY(api.groupStart());
        for (int p = 0; p < c->size; p++) {
            NCCL_TRY(api.send(s + (size_t) p * bytes, bytes, ncclChar, p, (ncclComm_t) c->nccl, st));
            NCCL_TRY(api.recv(r + (size_t) p * bytes, bytes, ncclChar, p, (ncclComm_t) c->nccl, st));
        }
        NCCL_TRY(api.groupEnd());
        return 0;
    }
#endif
    if (c->size == 1) return (int) hipMemcpyAsync(recv_d, send_d, bytes, hipMemcpyDeviceToDevice, st);
    // host transport: gather everybody's whole send buffer, keep the chunk addressed to this rank
    CommDiag diag;
    const size_t all = (size_t) c->size * bytes;
    int rc = stage_down(c, s, all, 0, st);
    if (rc != 0) return rc;
    c->hostRecv.resize((size_t) c->size * all);
    if (c->fn(c->user, c->hostSend.data(), c->hostRecv.data(), all) != 0) return 1;
    std::vector<char> mine(all);
    for (int p = 0; p < c->size; p++) memcpy(mine.data() + (size_t) p * bytes, c->hostRecv.data() + (size_t) p * all + (size_t) c->rank * bytes, bytes);
    return stage_up(r, mine.data(), all, st);
}

int ommhip_comm_ring_exchange(ommhip_comm* c, const void* send_down_d, void* recv_from_up_d, size_t bytes_down,
                              const void* send_up_d, void* recv_from_down_d, size_t bytes_up, void* stream) {
    hipStream_t st = (hipStream_t) stream;
    if (c->alone) return 0;
    if (c->size == 1 && !c->rccl) {
        // the only slab is its own neighbour on both sides
        if (bytes_down > 0) { hipError_t e = hipMemcpyAsync(recv_from_up_d, send_down_d, bytes_down, hipMemcpyDeviceToDevice, st); if (e != hipSuccess) return (int) e; }
        if (bytes_up > 0) { hipError_t e = hipMemcpyAsync(recv_from_down_d, send_up_d, bytes_up, hipMemcpyDeviceToDevice, st); if (e != hipSuccess) return (int) e; }
        return 0;
    }
    const int down = (c->rank + c->size - 1) % c->size, up = (c->rank + 1) % c->size;
#ifndef OMMHIP_EMU
    if (c->rccl) {
        RcclApi& api = rccl_api();
        NCCL_TRY(api.groupStart());
        if (bytes_down > 0) {
            NCCL_TRY(api.send(send_down_d, bytes_down, ncclChar, down, (ncclComm_t) c->nccl, st));
            NCCL_TRY(api.recv(recv_from_up_d, bytes_down, ncclChar, up, (ncclComm_t) c->nccl, st));
        }
        if (bytes_up > 0) {
            NCCL_TRY(api.send(send_up_d, bytes_up, ncclChar, up, (ncclComm_t) c->nccl, st));
            NCCL_TRY(api.recv(recv_from_down_d, bytes_up, ncclChar, down, (ncclComm_t) c->nccl, st));
        }
        NCCL_TRY(api.groupEnd());
        return 0;
    }
#endif
    CommDiag diag;
    const size_t rec = bytes_down + bytes_up;
    int rc = 0;
    if (bytes_down > 0) rc = stage_down(c, send_down_d, bytes_down, 0, st);
    if (rc == 0 && bytes_up > 0) rc = stage_down(c, send_up_d, bytes_up, bytes_down, st);
    if (rc != 0) return rc;
    if (c->hostSend.size() < rec) c->hostSend.resize(rec);
    c->hostRecv.resize((size_t) c->size * rec);
    if (c->fn(c->user, c->hostSend.data(), c->hostRecv.data(), rec) != 0) return 1;
    if (bytes_down > 0) rc = stage_up(recv_from_up_d, c->hostRecv.data() + (size_t) up * rec, bytes_down, st);
    if (rc == 0 && bytes_up > 0) rc = stage_up(recv_from_down_d, c->hostRecv.data() + (size_t) down * rec + bytes_down, bytes_up, st);
    return rc;
}

int ommhip_comm_halo_exchange(ommhip_comm* c, void* buffer_d, const ommhip_halo_plan* plan, void* stream) {
    if (c->size > OMMHIP_MAX_RANKS) return 1;
    if (c->alone) return 0;
    // One rank: the only slab holds everything already.  Over RCCL the group is issued all the same -- the rank is its own neighbour on
    // both sides, every section lands on itself -- so that a one-rank run walks the call pattern of the real thing (tests; the platform
    // never exchanges halos with one rank).
    if (c->size == 1 && !c->rccl) return 0;
    hipStream_t st = (hipStream_t) stream;
    char* buf = (char*) buffer_d;
    const int me = c->rank, down = (me + c->size - 1) % c->size, up = (me + 1) % c->size;
    char* mine = buf + (size_t) me * plan->rank_stride;
#ifndef OMMHIP_EMU
    if (c->rccl) {
        RcclApi& api = rccl_api();
        ncclComm_t nc = (ncclComm_t) c->nccl;
        NCCL_TRY(api.groupStart());
        // Sends and receives between one pair of ranks are matched in the order they are issued: with two ranks `down` and `up` are the
        // same peer, and both sides issue "down section" before "up section".
        if (plan->down_bytes[me] > 0) NCCL_TRY(api.send(mine + plan->down_offset[me], plan->down_bytes[me], ncclChar, down, nc, st));
        if (plan->down_bytes[up] > 0) NCCL_TRY(api.recv(buf + (size_t) up * plan->rank_stride + plan->down_offset[up], plan->down_bytes[up], ncclChar, up, nc, st));
        if (plan->up_bytes[me] > 0) NCCL_TRY(api.send(mine + plan->up_offset[me], plan->up_bytes[me], ncclChar, up, nc, st));
        if (plan->up_bytes[down] > 0) NCCL_TRY(api.recv(buf + (size_t) down * plan->rank_stride + plan->up_offset[down], plan->up_bytes[down], ncclChar, down, nc, st));
        if (plan->trailer_bytes > 0)
            for (int p = 0; p < c->size; p++) {
                if (p == me) continue;
                NCCL_TRY(api.send(mine + plan->trailer_offset, plan->trailer_bytes, ncclChar, p, nc, st));
                NCCL_TRY(api.recv(buf + (size_t) p * plan->rank_stride + plan->trailer_offset, plan->trailer_bytes, ncclChar, p, nc, st));
            }
        NCCL_TRY(api.groupEnd());
        return 0;
    }
#endif
    // host transport: records of one size (the largest section sizes of the plan) -- [down section | up section | trailer]
    CommDiag diag;
    size_t maxDown = 0, maxUp = 0;
    for (int r = 0; r < c->size; r++) { if (plan->down_bytes[r] > maxDown) maxDown = plan->down_bytes[r]; if (plan->up_bytes[r] > maxUp) maxUp = plan->up_bytes[r]; }
    const size_t rec = maxDown + maxUp + plan->trailer_bytes;
    if (rec == 0) return 0;
    if (c->hostSend.size() < rec) c->hostSend.resize(rec);
    int rc = 0;
    if (plan->down_bytes[me] > 0) rc = stage_down(c, mine + plan->down_offset[me], plan->down_bytes[me], 0, st);
    if (rc == 0 && plan->up_bytes[me] > 0) rc = stage_down(c, mine + plan->up_offset[me], plan->up_bytes[me], maxDown, st);
    if (rc == 0 && plan->trailer_bytes > 0) rc = stage_down(c, mine + plan->trailer_offset, plan->trailer_bytes, maxDown + maxUp, st);
    if (rc != 0) return rc;
    c->hostRecv.resize((size_t) c->size * rec);
    if (c->fn(c->user, c->hostSend.data(), c->hostRecv.data(), rec) != 0) return 1;
    if (plan->down_bytes[up] > 0) rc = stage_up(buf + (size_t) up * plan->rank_stride + plan->down_offset[up], c->hostRecv.data() + (size_t) up * rec, plan->down_bytes[up], st);
    if (rc == 0 && plan->up_bytes[down] > 0)
        rc = stage_up(buf + (size_t) down * plan->rank_stride + plan->up_offset[down], c->hostRecv.data() + (size_t) down * rec + maxDown, plan->up_bytes[down], st);
    for (int p = 0; p < c->size && rc == 0 && plan->trailer_bytes > 0; p++)
        if (p != me) rc = stage_up(buf + (size_t) p * plan->rank_stride + plan->trailer_offset, c->hostRecv.data() + (size_t) p * rec + maxDown + maxUp, plan->trailer_bytes, st);
    return rc;
}

int ommhip_comm_halo_return(ommhip_comm* c, long long* force_d, int padded_slots, const ommhip_halo_return_plan* plan, long long* staging_d, void* stream) {
    if (c->size > OMMHIP_MAX_RANKS) return 1;
    if (c->alone) return 0;
    if (c->size == 1 && !c->rccl) return 0;         // (one rank over RCCL: the section goes to itself and is added once more -- tests only)
    hipStream_t st = (hipStream_t) stream;
    const int me = c->rank, down = (me + c->size - 1) % c->size, up = (me + 1) % c->size;
    const int sendFirst = plan->first_slot[down], sendCount = plan->num_slots[down], recvFirst = plan->first_slot[me], recvCount = plan->num_slots[me];
    if (sendFirst < 0 || recvFirst < 0 || sendFirst + sendCount > padded_slots || recvFirst + recvCount > padded_slots) return 1;
    bool received = false;
#ifndef OMMHIP_EMU
    if (c->rccl) {
        RcclApi& api = rccl_api();
        ncclComm_t nc = (ncclComm_t) c->nccl;
        NCCL_TRY(api.groupStart());
        // (two ranks: `down` and `up` are the same peer -- its sends and this rank's receives are matched in the order of issue, component by component)
        for (int k = 0; k < 3; k++) {
            if (sendCount > 0) NCCL_TRY(api.send(force_d + (size_t) k * padded_slots + sendFirst, sizeof(long long) * (size_t) sendCount, ncclChar, down, nc, st));
            if (recvCount > 0) NCCL_TRY(api.recv(staging_d + (size_t) k * recvCount, sizeof(long long) * (size_t) recvCount, ncclChar, up, nc, st));
        }
        NCCL_TRY(api.groupEnd());
        received = true;
    }
#endif
    if (!received) {
        // host transport: every rank contributes one record of 3 x (largest section) elements; a rank keeps its upper neighbour's
        CommDiag diag;
        int maxCount = 0;
        for (int r = 0; r < c->size; r++) if (plan->num_slots[r] > maxCount) maxCount = plan->num_slots[r];
        const size_t part = sizeof(long long) * (size_t) maxCount, rec = 3 * part;
        if (rec == 0) return 0;
        if (c->hostSend.size() < rec) c->hostSend.resize(rec);
        int rc = 0;
        for (int k = 0; k < 3 && rc == 0 && sendCount > 0; k++)
            rc = stage_down(c, force_d + (size_t) k * padded_slots + sendFirst, sizeof(long long) * (size_t) sendCount, (size_t) k * part, st);
        if (rc != 0) return rc;
        c->hostRecv.resize((size_t) c->size * rec);
        if (c->fn(c->user, c->hostSend.data(), c->hostRecv.data(), rec) != 0) return 1;
        for (int k = 0; k < 3 && rc == 0 && recvCount > 0; k++)
            rc = stage_up(staging_d + (size_t) k * recvCount, c->hostRecv.data() + (size_t) up * rec + (size_t) k * part, sizeof(long long) * (size_t) recvCount, st);
        if (rc != 0) return rc;
    }
    if (recvCount > 0) hipLaunchKernelGGL(k_add_returned, dim3((unsigned) ((3 * (size_t) recvCount + 255) / 256)), dim3(256), 0, st, force_d, padded_slots, recvFirst, recvCount, staging_d);
    return (int) hipGetLastError();
}

int ommhip_comm_all_gather_host(ommhip_comm* c, const void* send, void* recv, size_t bytes, void* stream) {
    if (c->size == 1 && !c->rccl) { memcpy(recv, send, bytes); return 0; }
    if (c->alone) { for (int r = 0; r < c->size; r++) memcpy((char*) recv + (size_t) r * bytes, send, bytes); return 0; }
#ifndef OMMHIP_EMU
    if (c->rccl) {
        hipStream_t st = (hipStream_t) stream;
        const size_t all = (size_t) c->size * bytes;
        if (c->smallBytes < all) {
            if (c->smallDev != nullptr) hipFree(c->smallDev);
            hipError_t e = hipMalloc(&c->smallDev, all);
            if (e != hipSuccess) return (int) e;
            c->smallBytes = all;
        }
        char* dev = (char*) c->smallDev;
        hipError_t e = hipMemcpyAsync(dev + (size_t) c->rank * bytes, send, bytes, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return (int) e;
        NCCL_TRY(rccl_api().allGather(dev + (size_t) c->rank * bytes, dev, bytes, ncclChar, (ncclComm_t) c->nccl, st));
        e = hipMemcpyAsync(recv, dev, all, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return (int) e;
        return (int) hipStreamSynchronize(st);
    }
#endif
    (void) stream;
    CommDiag diag;
    return c->fn(c->user, send, recv, bytes) != 0 ? 1 : 0;
}

double ommhip_comm_diag_seconds(int reset) {
    const double v = g_commSeconds;
    if (reset) g_commSeconds = 0.0;
    return v;
}

}

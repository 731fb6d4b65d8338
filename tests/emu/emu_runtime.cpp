// tests/emu/emu_runtime.cpp -- TEST INFRASTRUCTURE ONLY (see tests/emu/hip/hip_runtime.h).
//
// Cooperative-fiber SIMT engine: one fiber per work-item, workgroups run one after another,
// wave64 collectives and __syncthreads() are rendezvous points.  Fibers never preempt each
// other, so "atomics" are plain read-modify-write.  This catches indexing / algorithm /
// host-plumbing bugs; it cannot catch data races or memory-model bugs -- those are what the
// `-m gpu` tests on real hardware are for.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

namespace emu {

thread_local ThreadCtx* cur = nullptr;

// ---- minimal x86-64 context switch (callee-saved registers only) ----
extern "C" void emu_switch(void** saveSp, void* newSp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

static const size_t STACK_BYTES = 256 * 1024;

struct WaveState {
    uint64_t slots[64];
    int arrived = 0;
    int alive = 0;
    unsigned gen = 0;
    unsigned long long aliveMask = 0;
};

struct FiberState {
    ThreadCtx ctx;
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = true;
};

struct Engine {
    std::vector<FiberState> fibers;
    std::vector<WaveState> waves;
    void* schedSp = nullptr;
    int current = -1;
    int blockArrived = 0, blockAlive = 0;
    unsigned blockGen = 0;
    void (*entry)(void*) = nullptr;
    void* arg = nullptr;
};
static thread_local Engine* eng = nullptr;

static void yield_to_scheduler() {
    FiberState& f = eng->fibers[eng->current];
    emu_switch(&f.sp, eng->schedSp);
}

static void fiber_main() {
    Engine* e = eng;
    FiberState& f = e->fibers[e->current];
    e->entry(e->arg);
    f.done = true;
    WaveState& w = e->waves[f.ctx.wave];
    w.alive--;
    w.aliveMask &= ~(1ull << f.ctx.lane);
    e->blockAlive--;
    // A finished lane may complete a rendezvous the others are waiting on.
    if (w.alive > 0 && w.arrived == w.alive) { w.arrived = 0; w.gen++; }
    if (e->blockAlive > 0 && e->blockArrived == e->blockAlive) { e->blockArrived = 0; e->blockGen++; }
    emu_switch(&f.sp, e->schedSp);
    abort();  // never resumed
}

void wave_barrier() {
    FiberState& f = eng->fibers[eng->current];
    WaveState& w = eng->waves[f.ctx.wave];
    unsigned myGen = w.gen;
    w.arrived++;
    if (w.arrived == w.alive) { w.arrived = 0; w.gen++; return; }
    while (w.gen == myGen) yield_to_scheduler();
}

void block_barrier() {
    Engine* e = eng;
    unsigned myGen = e->blockGen;
    e->blockArrived++;
    if (e->blockArrived == e->blockAlive) { e->blockArrived = 0; e->blockGen++; return; }
    while (e->blockGen == myGen) yield_to_scheduler();
}

uint64_t* wave_slots() { return eng->waves[eng->fibers[eng->current].ctx.wave].slots; }
unsigned long long wave_alive_mask() { return eng->waves[eng->fibers[eng->current].ctx.wave].aliveMask; }

static void prepare_fiber(FiberState& f) {
    if (f.stack == nullptr) f.stack = (char*) aligned_alloc(64, STACK_BYTES);
    // Initial frame: six callee-saved registers + return address into fiber_main.  At entry to a
    // function (rsp+8) must be 16-byte aligned, i.e. rsp % 16 == 8 right after the `ret`.
    uintptr_t top = ((uintptr_t) (f.stack + STACK_BYTES)) & ~(uintptr_t) 15;
    void** sp = (void**) (top - 8);      // slot that "ret" leaves behind: keeps alignment (top-8 after ret => %16==8)
    *(--sp) = (void*) &fiber_main;       // return address popped by ret
    for (int i = 0; i < 6; i++) *(--sp) = nullptr;
    f.sp = sp;
    f.done = false;
}

void run_grid(dim3 grid, dim3 block, void (*entry)(void*), void* arg) {
    Engine local;
    Engine* saved = eng;
    static thread_local std::vector<FiberState>* stackPool = nullptr;
    if (stackPool == nullptr) stackPool = new std::vector<FiberState>();
    eng = &local;
    local.entry = entry;
    local.arg = arg;
    const unsigned nthreads = block.x * block.y * block.z;
    const unsigned nwaves = (nthreads + 63) / 64;
    if (stackPool->size() < nthreads) stackPool->resize(nthreads);
    local.fibers.swap(*stackPool);
    local.waves.resize(nwaves);
    ThreadCtx* savedCur = cur;
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        for (unsigned w = 0; w < nwaves; w++) { local.waves[w] = WaveState(); }
        local.blockArrived = 0; local.blockAlive = nthreads; local.blockGen = 0;
        for (unsigned t = 0; t < nthreads; t++) {
            FiberState& f = local.fibers[t];
            f.ctx.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            f.ctx.bid = dim3(bx, by, bz);
            f.ctx.bdim = block;
            f.ctx.gdim = grid;
            f.ctx.lane = t & 63;
            f.ctx.wave = t >> 6;
            WaveState& w = local.waves[t >> 6];
            w.alive++;
            w.aliveMask |= 1ull << (t & 63);
            prepare_fiber(f);
        }
        unsigned remaining = nthreads;
        while (remaining > 0) {
            unsigned progressed = 0;
            for (unsigned t = 0; t < nthreads; t++) {
                FiberState& f = local.fibers[t];
                if (f.done) continue;
                local.current = (int) t;
                cur = &f.ctx;
                emu_switch(&local.schedSp, f.sp);
                if (f.done) { remaining--; }
                progressed++;
            }
            if (progressed == 0) break;
        }
    }
    cur = savedCur;
    local.fibers.swap(*stackPool);
    eng = saved;
}

}  // namespace emu

// ---------------------------------------------------------------- fake runtime API
struct emuStream { int dummy; };
struct emuEvent { std::chrono::steady_clock::time_point t; };

extern "C" {
hipError_t hipMalloc(void** p, size_t n) { *p = n ? aligned_alloc(256, (n + 255) / 256 * 256) : nullptr; if (*p) memset(*p, 0xCD, n); return hipSuccess; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = aligned_alloc(256, (n + 255) / 256 * 256); memset(*p, 0, n); return hipSuccess; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new emuStream(); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new emuStream(); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = new emuStream(); return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emuEvent(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p)); strcpy(p->name, "hip-emu (CPU fibers)"); strcpy(p->gcnArchName, "emu"); p->multiProcessorCount = 4; p->totalGlobalMem = 1ull << 34; p->clockRate = 1000000;
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "emu"; }
}

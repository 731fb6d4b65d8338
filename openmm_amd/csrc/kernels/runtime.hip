// C-ABI veneer over the HIP runtime (device memory, streams, events) -- see include/openmm_hip_kernels.h.
// The host-side plugin never includes hip_runtime.h; everything it needs goes through these calls.
#include <cstdint>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../../include/openmm_hip_kernels.h"

extern "C" {

int ommhip_device_count(int* count) { return (int) hipGetDeviceCount(count); }
int ommhip_set_device(int device) { return (int) hipSetDevice(device); }
int ommhip_device_info(int device, char* name, int name_len, int* num_cus, size_t* total_mem) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return (int) e;
    if (name != nullptr && name_len > 0) {
        // (some driver stacks leave the marketing name empty: the architecture, CU count and memory then say what the device is)
        if (prop.name[0] != 0) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
        else snprintf(name, name_len, "AMD GPU, %d CUs, %.0f GB (%s)", prop.multiProcessorCount, (double) prop.totalGlobalMem / 1073741824.0, prop.gcnArchName);
    }
    if (num_cus != nullptr) *num_cus = prop.multiProcessorCount;
    if (total_mem != nullptr) *total_mem = prop.totalGlobalMem;
    return 0;
}
int ommhip_malloc(void** ptr_d, size_t bytes) { return (int) hipMalloc(ptr_d, bytes > 0 ? bytes : 16); }
int ommhip_free(void* ptr_d) { return ptr_d == nullptr ? 0 : (int) hipFree(ptr_d); }
int ommhip_host_malloc(void** ptr, size_t bytes) { return (int) hipHostMalloc(ptr, bytes > 0 ? bytes : 16, hipHostMallocDefault); }
int ommhip_host_free(void* ptr) { return ptr == nullptr ? 0 : (int) hipHostFree(ptr); }
int ommhip_memcpy_h2d(void* dst_d, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    return (int) hipMemcpyAsync(dst_d, src, bytes, hipMemcpyHostToDevice, (hipStream_t) stream);
}
int ommhip_memcpy_d2h(void* dst, const void* src_d, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    return (int) hipMemcpyAsync(dst, src_d, bytes, hipMemcpyDeviceToHost, (hipStream_t) stream);
}
int ommhip_memcpy_d2d(void* dst_d, const void* src_d, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    return (int) hipMemcpyAsync(dst_d, src_d, bytes, hipMemcpyDeviceToDevice, (hipStream_t) stream);
}
int ommhip_memset(void* dst_d, int value, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    return (int) hipMemsetAsync(dst_d, value, bytes, (hipStream_t) stream);
}
// A/B knob OPENMM_HIP_PME_CUS=n: the high-priority (reciprocal-space) stream is created on the first n bits of the CU mask and the ordinary
// streams on the rest, so that the two streams' kernels run on disjoint compute units instead of queueing behind each other's resident
// wavefronts (hipExtStreamCreateWithCUMask; consecutive mask bits go round the XCDs, so both sets spread over all eight).
static int cu_split() {
    static const int n = getenv("OPENMM_HIP_PME_CUS") != nullptr ? atoi(getenv("OPENMM_HIP_PME_CUS")) : 0;
    return n;
}
static int create_masked(void** stream, bool side) {
#ifndef OMMHIP_EMU
    int dev = 0; hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount, n = cu_split();
    if (n <= 0 || n >= cus) return 1;
    uint32_t mask[16] = {0};
    for (int i = 0; i < cus && i < 512; i++)
        if ((i < n) == side) mask[i / 32] |= 1u << (i % 32);
    return (int) hipExtStreamCreateWithCUMask((hipStream_t*) stream, (uint32_t) ((cus + 31) / 32), mask);
#else
    return 1;
#endif
}
int ommhip_stream_create(void** stream) {
    if (cu_split() > 0 && create_masked(stream, false) == 0) return 0;
    return (int) hipStreamCreateWithFlags((hipStream_t*) stream, hipStreamNonBlocking);
}
int ommhip_stream_create_priority(void** stream, int high_priority) {
    if (high_priority && cu_split() > 0 && create_masked(stream, true) == 0) return 0;
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);           // numerically lower = higher priority
    return (int) hipStreamCreateWithPriority((hipStream_t*) stream, hipStreamNonBlocking, high_priority ? greatest : least);
}
int ommhip_event_create_untimed(void** event) { return (int) hipEventCreateWithFlags((hipEvent_t*) event, hipEventDisableTiming); }
int ommhip_stream_destroy(void* stream) { return (int) hipStreamDestroy((hipStream_t) stream); }
int ommhip_stream_sync(void* stream) { return (int) hipStreamSynchronize((hipStream_t) stream); }
int ommhip_device_sync(int device) {
    if (device >= 0) { const hipError_t e = hipSetDevice(device); if (e != hipSuccess) return (int) e; }
    return (int) hipDeviceSynchronize();
}
int ommhip_event_create(void** event) { return (int) hipEventCreate((hipEvent_t*) event); }
int ommhip_event_destroy(void* event) { return (int) hipEventDestroy((hipEvent_t) event); }
int ommhip_event_record(void* event, void* stream) { return (int) hipEventRecord((hipEvent_t) event, (hipStream_t) stream); }
int ommhip_event_sync(void* event) { return (int) hipEventSynchronize((hipEvent_t) event); }
int ommhip_event_elapsed_ms(void* start, void* stop, float* ms) { return (int) hipEventElapsedTime(ms, (hipEvent_t) start, (hipEvent_t) stop); }
int ommhip_stream_wait_event(void* stream, void* event) { return (int) hipStreamWaitEvent((hipStream_t) stream, (hipEvent_t) event, 0); }
size_t ommhip_struct_size(int which) {
    switch (which) {
        case 0: return sizeof(ommhip_neighbor_list);
        case 1: return sizeof(ommhip_nonbonded_params);
        case 2: return sizeof(ommhip_pme);
        case 3: return sizeof(ommhip_term_batch);
        case 4: return sizeof(ommhip_integrator_state);
        case 5: return sizeof(ommhip_step_units);
        case 6: return sizeof(ommhip_ccma);
        case 7: return sizeof(ommhip_valence_list);
        case 8: return sizeof(ommhip_vm_instruction);
        case 9: return sizeof(ommhip_vm_step);
        case 10: return sizeof(ommhip_vm_state);
        case 11: return sizeof(ommhip_vm_bonds);
    }
    return 0;
}
const char* ommhip_error_string(int code) {
    if (code >= 1000) {                      // 1000 + ncclResult_t (include/openmm_hip_comm.h)
        static thread_local char buf[64];
        snprintf(buf, sizeof(buf), "RCCL error %d", code - 1000);
        return buf;
    }
    return hipGetErrorString((hipError_t) code);
}

}

// ------------------------------------------------------------------------------------------------
// Opt-in per-kernel timing with HIP events recorded on the stream the kernel is launched on
// (used by bench.py for the roofline figure; off by default, zero cost when off).
// ------------------------------------------------------------------------------------------------
#include <vector>
namespace {
struct ProfileTimer {
    std::vector<hipEvent_t> start, stop;
    size_t used = 0;
    double totalMs = 0;
    long long calls = 0;
    unsigned seq = 0;      // launches seen; every profileEnabled-th one is timed
    bool open = false;
    bool lastTaken = false;      // did the last ommhip_profile_take of this timer hand out an event pair?
};
ProfileTimer timers[OMMHIP_PROFILE_NUM_TIMERS];
int profileEnabled = 0;
unsigned profileMask = ~0u;      // timers that record (bit per timer)
void profile_drain(ProfileTimer& t) {
    for (size_t i = 0; i < t.used; i++) {
        float ms = 0;
        if (hipEventSynchronize(t.stop[i]) == hipSuccess && hipEventElapsedTime(&ms, t.start[i], t.stop[i]) == hipSuccess) {
            t.totalMs += ms;
            t.calls++;
        }
    }
    t.used = 0;
}
}  // namespace

extern "C" {
int ommhip_profile_enable(int enabled) { profileEnabled = enabled < 0 ? 0 : enabled; profileMask = ~0u; return 0; }   /* n > 1: time every n-th launch */
int ommhip_profile_enable_timers(int every, unsigned mask, int reserve) {
    // only the timers of `mask` record, and `reserve` event pairs per timer are created now rather than at their first use
    // (an event pair costs more to create than a small kernel takes: a 20-step timed region should not pay for that)
    profileEnabled = every < 0 ? 0 : every; profileMask = mask;
    for (int i = 0; i < OMMHIP_PROFILE_NUM_TIMERS; i++) {
        if (((mask >> i) & 1u) == 0) continue;
        while ((int) timers[i].start.size() < reserve) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return 1;
            timers[i].start.push_back(a); timers[i].stop.push_back(b);
        }
    }
    return 0;
}
int ommhip_profile_reset() {
    for (int i = 0; i < OMMHIP_PROFILE_NUM_TIMERS; i++) { profile_drain(timers[i]); timers[i].totalMs = 0; timers[i].calls = 0; }
    return 0;
}
int ommhip_profile_begin(int timer, void* stream) {
    if (!profileEnabled || timer < 0 || timer >= OMMHIP_PROFILE_NUM_TIMERS || ((profileMask >> timer) & 1u) == 0) return 0;
    ProfileTimer& t = timers[timer];
    t.open = (t.seq++ % (unsigned) profileEnabled) == 0;
    if (!t.open) return 0;
    if (t.used == 4096) profile_drain(t);
    if (t.used == t.start.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return 0;
        t.start.push_back(a); t.stop.push_back(b);
    }
    return (int) hipEventRecord(t.start[t.used], (hipStream_t) stream);
}
int ommhip_profile_end(int timer, void* stream) {
    if (!profileEnabled || timer < 0 || timer >= OMMHIP_PROFILE_NUM_TIMERS) return 0;
    ProfileTimer& t = timers[timer];
    if (!t.open || t.used >= t.stop.size()) return 0;
    t.open = false;
    hipError_t e = hipEventRecord(t.stop[t.used], (hipStream_t) stream);
    t.used++;
    return (int) e;
}
int ommhip_profile_take(int timer, void** start_event, void** stop_event) {
    return ommhip_profile_take_if(timer, -1, start_event, stop_event);
}
/* with_timer >= 0: no sampling decision of its own -- a pair is handed out exactly when `with_timer` handed one out last (the per-launch
 * timers of a group of launches follow the timer that brackets the group) */
int ommhip_profile_take_if(int timer, int with_timer, void** start_event, void** stop_event) {
    *start_event = nullptr; *stop_event = nullptr;
    if (!profileEnabled || timer < 0 || timer >= OMMHIP_PROFILE_NUM_TIMERS) return 0;
    if (((profileMask >> timer) & 1u) == 0) return 0;           // (a following timer needs its own bit too: every event pair on a launch costs)
    ProfileTimer& t = timers[timer];
    if (with_timer >= 0) { if (with_timer >= OMMHIP_PROFILE_NUM_TIMERS || !timers[with_timer].lastTaken) return 0; }
    else {
        t.lastTaken = (t.seq++ % (unsigned) profileEnabled) == 0;
        if (!t.lastTaken) return 0;
    }
    if (t.used == 4096) profile_drain(t);
    if (t.used == t.start.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return 0;
        t.start.push_back(a); t.stop.push_back(b);
    }
    *start_event = t.start[t.used]; *stop_event = t.stop[t.used];
    t.used++;
    return 0;
}
int ommhip_profile_collect(int timer, long long* calls, double* total_ms) {
    if (timer < 0 || timer >= OMMHIP_PROFILE_NUM_TIMERS) return 1;
    profile_drain(timers[timer]);
    *calls = timers[timer].calls;
    *total_ms = timers[timer].totalMs;
    return 0;
}
}

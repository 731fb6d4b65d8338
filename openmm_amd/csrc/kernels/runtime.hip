// C-ABI veneer over the HIP runtime (device memory, streams, events) -- see include/openmm_hip_kernels.h.
// The host-side plugin never includes hip_runtime.h; everything it needs goes through these calls.
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
        snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
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
int ommhip_stream_create(void** stream) { return (int) hipStreamCreateWithFlags((hipStream_t*) stream, hipStreamNonBlocking); }
int ommhip_stream_destroy(void* stream) { return (int) hipStreamDestroy((hipStream_t) stream); }
int ommhip_stream_sync(void* stream) { return (int) hipStreamSynchronize((hipStream_t) stream); }
int ommhip_event_create(void** event) { return (int) hipEventCreate((hipEvent_t*) event); }
int ommhip_event_destroy(void* event) { return (int) hipEventDestroy((hipEvent_t) event); }
int ommhip_event_record(void* event, void* stream) { return (int) hipEventRecord((hipEvent_t) event, (hipStream_t) stream); }
int ommhip_event_sync(void* event) { return (int) hipEventSynchronize((hipEvent_t) event); }
int ommhip_event_elapsed_ms(void* start, void* stop, float* ms) { return (int) hipEventElapsedTime(ms, (hipEvent_t) start, (hipEvent_t) stop); }
int ommhip_stream_wait_event(void* stream, void* event) { return (int) hipStreamWaitEvent((hipStream_t) stream, (hipEvent_t) event, 0); }
const char* ommhip_error_string(int code) { return hipGetErrorString((hipError_t) code); }

}

// Microbenchmark: throughput of global atomics on MI355X by memory scope and operand type.
// Addresses mimic the force-accumulation pattern: each wave instruction adds to 64 consecutive slots of a 0.6 MB buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SCOPE, typename T>
__global__ void k(T* buf, int n, int iters, int stridePattern) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned h = wave * 2654435761u;
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        int base = (h >> 8) % (n - 64 * stridePattern);
        T* p = buf + base + lane * stridePattern;
        if (SCOPE == 0) __hip_atomic_fetch_add(p, (T) 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (SCOPE == 1) __hip_atomic_fetch_add(p, (T) 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (SCOPE == 2) __hip_atomic_fetch_add(p, (T) 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else *p += (T) 1;   // plain RMW (racy) as an upper bound
    }
}
__global__ void xcc(int* out) { if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }
template <int SCOPE, typename T> void run(const char* name, int stride) {
    int n = 80000; T* buf; hipMalloc(&buf, n * sizeof(T)); hipMemset(buf, 0, n * sizeof(T));
    int blocks = 2048, iters = 64;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<SCOPE, T><<<blocks, 256>>>(buf, n, iters, stride); hipDeviceSynchronize();
    hipEventRecord(a); k<SCOPE, T><<<blocks, 256>>>(buf, n, iters, stride); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double) blocks * 256 * iters;
    printf("%-28s stride %d: %8.1f us  %7.2f G atomics/s\n", name, stride, ms * 1e3, ops / ms / 1e6);
    hipFree(buf);
}
int main() {
    int* d; hipMalloc(&d, 64 * 4); xcc<<<64, 64>>>(d); std::vector<int> h(64); hipMemcpy(h.data(), d, 256, hipMemcpyDeviceToHost);
    printf("xcc ids of blocks 0..15:"); for (int i = 0; i < 16; i++) printf(" %d", h[i]); printf("\n");
    for (int stride : {1, 7}) {
        run<0, unsigned long long>("u64 agent", stride);
        run<1, unsigned long long>("u64 workgroup", stride);
        run<2, unsigned long long>("u64 wavefront", stride);
        run<3, unsigned long long>("u64 plain (racy)", stride);
        run<0, float>("f32 agent", stride);
        run<1, float>("f32 workgroup", stride);
        run<3, float>("f32 plain (racy)", stride);
        run<0, unsigned>("u32 agent", stride);
        run<1, unsigned>("u32 workgroup", stride);
    }
    return 0;
}

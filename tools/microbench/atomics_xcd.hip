// Microbenchmark: device-scope atomics on ONE buffer against workgroup-scope atomics on a copy per XCD (selected by the hardware's
// XCC_ID), at the sizes of the 1M-atom step: a 192^3 float grid (charge spreading: 16 lanes per 64-byte line, 4 lines per wave
// instruction, lines picked inside a 16^3 brick around a per-workgroup origin) and a 3 x 1M int64 force buffer (64 consecutive slots
// per wave instruction).  A workgroup-scope atomic is performed in the XCD's own L2; it is only correct if every writer of a copy
// sits on that XCD, which picking the copy by XCC_ID guarantees.  The sums are checked after a kernel boundary.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/atomics_xcd.hip -o /tmp/atomics_xcd && /tmp/atomics_xcd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }      // hwreg(HW_REG_XCC_ID, 0, 4)

template <int MODE>      // 0 agent scope, one buffer; 1 workgroup scope, copy of this XCD
__global__ __launch_bounds__(256) void grid_lines(float* grid, size_t copyStride, int n, int bricksPerWg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* g = grid + (MODE == 1 ? (size_t) xcc_id() * copyStride : 0);
    unsigned h = blockIdx.x * 2654435761u + 12345u;
    for (int b = 0; b < bricksPerWg; b++) {
        h = h * 1664525u + 1013904223u;
        const int ox = (h >> 4) % (n - 16), oy = (h >> 12) % (n - 16), oz = ((h >> 20) % (n - 16)) & ~15;
        // the workgroup flushes a 16 x 16 x 16 brick: 256 rows of 16 floats (one 64-byte line each), 4 rows per wave instruction
        for (int r = wave * 4 + (lane >> 4); r < 256; r += 16) {
            float* p = g + ((size_t) (ox + (r >> 4)) * n + oy + (r & 15)) * n + oz + (lane & 15);
            if (MODE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void force_rows(unsigned long long* f, size_t copyStride, int slots, int iters) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned long long* g = f + (MODE == 1 ? (size_t) xcc_id() * copyStride : 0);
    unsigned h = wave * 2654435761u + 99u;
    // a wave works in a neighbourhood of ~4000 slots (its i-block's j atoms), as the pair kernel does
    const int home = (int) (((unsigned long long) wave * 2654435761ull) % (unsigned) (slots - 8192));
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        const int base = home + (int) ((h >> 8) % 4096u) / 64 * 64;
        for (int c = 0; c < 3; c++) {
            unsigned long long* p = g + (size_t) c * slots + base + lane;
            if (MODE == 0) __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

template <typename T> __global__ void reduce_copies(T* buf, size_t copyStride, size_t n, int copies) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T s = buf[i];
    for (int c = 1; c < copies; c++) s += buf[i + c * copyStride];
    buf[i] = s;
}
template <typename T> __global__ void total(const T* buf, size_t n, double* out) {
    double s = 0;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) s += (double) buf[i];
    atomicAdd(out, s);
}
__global__ void xcc(int* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

template <typename F> float timed(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3f;
}

int main() {
    int* d; hipMalloc(&d, 64 * 4); xcc<<<64, 64>>>(d); std::vector<int> h(64); hipMemcpy(h.data(), d, 256, hipMemcpyDeviceToHost);
    printf("xcc ids of workgroups 0..15:"); for (int i = 0; i < 16; i++) printf(" %d", h[i]); printf("\n");
    double* dsum; hipMalloc(&dsum, 8);
    {   // ---- grid: 30 798 bricks (one per 32-atom block of the 1M box)
        const int n = 192; const size_t cells = (size_t) n * n * n;
        float* grid; hipMalloc(&grid, cells * 4 * 8);
        const int wgs = 30798, bricks = 1;
        for (int mode = 0; mode < 2; mode++) {
            hipMemset(grid, 0, cells * 4 * 8); hipDeviceSynchronize();
            float us = 0;
            for (int rep = 0; rep < 3; rep++) us = timed([&] { if (mode == 0) grid_lines<0><<<wgs, 256>>>(grid, cells, n, bricks); else grid_lines<1><<<wgs, 256>>>(grid, cells, n, bricks); });
            float usR = 0;
            if (mode == 1) usR = timed([&] { reduce_copies<float><<<(unsigned) ((cells + 255) / 256), 256>>>(grid, cells, cells, 8); });
            hipMemset(dsum, 0, 8); total<float><<<1024, 256>>>(grid, cells, dsum); double s; hipMemcpy(&s, dsum, 8, hipMemcpyDeviceToHost);
            printf("grid  %-34s %8.1f us per launch (+ reduce %6.1f us)   sum %.0f (expected %.0f)\n", mode == 0 ? "agent scope, one grid" : "workgroup scope, grid per XCD", us, usR,
                   s, 3.0 * wgs * bricks * 4096);
        }
        hipFree(grid);
    }
    {   // ---- forces: 356 000 rows x 2 (j forces + i forces as the same pattern) x 3 components
        const int slots = 985600; const size_t n = (size_t) 3 * slots;
        unsigned long long* f; hipMalloc(&f, n * 8 * 8);
        const int waves = 78000 * 4, iters = 2;        // ~ rows of the 1M list, one j-row add and one i-block add each
        for (int mode = 0; mode < 2; mode++) {
            hipMemset(f, 0, n * 8 * 8); hipDeviceSynchronize();
            float us = 0;
            for (int rep = 0; rep < 3; rep++) us = timed([&] { if (mode == 0) force_rows<0><<<waves / 4, 256>>>(f, n, slots, iters); else force_rows<1><<<waves / 4, 256>>>(f, n, slots, iters); });
            float usR = 0;
            if (mode == 1) usR = timed([&] { reduce_copies<unsigned long long><<<(unsigned) ((n + 255) / 256), 256>>>(f, n, n, 8); });
            hipMemset(dsum, 0, 8); total<unsigned long long><<<1024, 256>>>(f, n, dsum); double s; hipMemcpy(&s, dsum, 8, hipMemcpyDeviceToHost);
            printf("force %-34s %8.1f us per launch (+ reduce %6.1f us)   sum %.0f (expected %.0f)\n", mode == 0 ? "agent scope, one buffer" : "workgroup scope, buffer per XCD", us, usR,
                   s, 3.0 * waves * iters * 3 * 64);
        }
        hipFree(f);
    }
    return 0;
}

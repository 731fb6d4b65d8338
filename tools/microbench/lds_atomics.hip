// Microbenchmark: throughput of LDS atomics (ds_add_f32 / ds_add_u32 / plain read-modify-write) per CU on MI355X,
// with the address pattern of the PME spread brick (a 5x5x5 stencil per wave instruction at a random offset).
// Build: hipcc --offload-arch=gfx950 -O3 lds_atomics.hip -o build/microbench_lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#define CELLS (16 * 16 * 17)
template <int MODE, typename T>
__global__ __launch_bounds__(256) void k(T* out, int iters) {
    __shared__ T brick[CELLS];
    for (int i = threadIdx.x; i < CELLS; i += 256) brick[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane;                       // first 64 points of the 125-point stencil
    const int ix = pt / 25, iy = (pt / 5) % 5, iz = pt % 5;
    unsigned h = (blockIdx.x * 4 + wave) * 2654435761u;
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        const int ox = (h >> 8) % 11, oy = (h >> 12) % 11, oz = (h >> 16) % 11;
        T* p = &brick[((ox + ix) * 16 + oy + iy) * 17 + oz + iz];
        if (MODE == 0) atomicAdd(p, (T) 1);
        else if (MODE == 1) *p += (T) 1;        // racy upper bound
        else if (MODE == 2) { T v = *p; if (v == (T) -5) out[0] = v; }   // read only
    }
    __syncthreads();
    T s = 0;
    for (int i = threadIdx.x; i < CELLS; i += 256) s += brick[i];
    if (s == (T) -1) out[0] = s;
}
template <int MODE, typename T> void run(const char* name) {
    T* d; hipMalloc(&d, 64);
    const int blocks = 256 * 3, iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE, T><<<blocks, 256>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(a); k<MODE, T><<<blocks, 256>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double) blocks * 256 * iters;
    printf("%-26s %8.1f us  %8.2f G lane-ops/s  = %.2f lane-ops/clk/CU at 2.4 GHz, 256 CUs\n", name, ms * 1e3, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4);
    hipFree(d);
}
int main() {
    run<0, float>("ds_add_f32 (atomic)");
    run<0, unsigned>("ds_add_u32 (atomic)");
    run<0, int>("ds_add_i32 (atomic)");
    run<1, float>("f32 plain rmw (racy)");
    run<2, float>("f32 read only");
    return 0;
}

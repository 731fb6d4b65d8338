// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A CPU stand-in for <hip/hip_runtime.h>, found first on the include path when the HIP
// sources under openmm_amd/csrc/kernels are compiled with g++ for the "emu" build
// (tests/emu/Makefile).  It executes every workgroup as a set of cooperative fibers (one per
// work-item), with wave64 cross-lane operations and __syncthreads() implemented as fiber
// rendezvous, so the plugin's host logic and the kernels' indexing/algorithms can be
// exercised in a container that has no GPU.  It is never linked into the product libraries
// (libopenmm_hip_kernels.so / libOpenMMHIP.so), which are built by hipcc for gfx950 only.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <tuple>
#include <utility>

#define OMMHIP_EMU 1

// ---------------------------------------------------------------- qualifiers
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
// (thread_local: one copy per HOST thread -- the ranks of a device-list Context run their kernels on threads of one process, tests/hip/TestHipParallel.cpp)
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __restrict__ __restrict

// ---------------------------------------------------------------- vector types
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct double3 { double x, y, z; };
struct alignas(32) double4 { double x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return {x, y}; }
static inline double3 make_double3(double x, double y, double z) { return {x, y, z}; }
static inline double4 make_double4(double x, double y, double z, double w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int3 make_int3(int x, int y, int z) { return {x, y, z}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---------------------------------------------------------------- runtime API subset
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600 };
typedef struct emuStream* hipStream_t;
typedef struct emuEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipEventDefault = 0 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; int clockRate; };

extern "C" {
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int priority);
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
const char* hipGetErrorString(hipError_t e);
}

// ---------------------------------------------------------------- SIMT execution engine
namespace emu {
struct Fiber;
struct ThreadCtx {
    dim3 tid, bid, bdim, gdim;
    int lane;      // lane within the wave
    int wave;      // wave within the block
};
extern thread_local ThreadCtx* cur;
// Rendezvous of all live fibers of the current wave / block.
void wave_barrier();
void block_barrier();
// 64 8-byte exchange slots of the current wave.
uint64_t* wave_slots();
unsigned long long wave_alive_mask();
void run_grid(dim3 grid, dim3 block, void (*entry)(void*), void* arg);

template <class T> static inline uint64_t to_bits(T v) {
    static_assert(sizeof(T) <= 8, "shuffle payload too large");
    uint64_t b = 0; std::memcpy(&b, &v, sizeof(T)); return b;
}
template <class T> static inline T from_bits(uint64_t b) { T v; std::memcpy(&v, &b, sizeof(T)); return v; }

template <class T> static inline T shfl(T v, int srcLane) {
    uint64_t* s = wave_slots();
    s[cur->lane] = to_bits(v);
    wave_barrier();
    T r = from_bits<T>(s[srcLane & 63]);
    wave_barrier();
    return r;
}
static inline unsigned long long ballot(int pred) {
    uint64_t* s = wave_slots();
    s[cur->lane] = pred ? 1 : 0;
    wave_barrier();
    unsigned long long m = 0, alive = wave_alive_mask();
    for (int i = 0; i < 64; i++) if (((alive >> i) & 1) && s[i] == 1) m |= (1ull << i);
    wave_barrier();
    return m;
}
}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::cur->bid)
#define blockDim (emu::cur->bdim)
#define gridDim (emu::cur->gdim)
static const int warpSize = 64;

static inline void __syncthreads() { emu::block_barrier(); }
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    int lane = emu::cur->lane;
    int base = lane & ~(width - 1);
    return emu::shfl(v, base + (src & (width - 1)));
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) { return __shfl(v, (emu::cur->lane ^ mask), width); }
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int lane = emu::cur->lane; int l = lane & (width - 1);
    int src = (l + (int) d < width) ? lane + (int) d : lane;
    return emu::shfl(v, src);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int lane = emu::cur->lane; int l = lane & (width - 1);
    int src = (l >= (int) d) ? lane - (int) d : lane;
    return emu::shfl(v, src);
}
static inline unsigned long long __ballot(int pred) { return emu::ballot(pred); }
static inline int __any(int pred) { return emu::ballot(pred) != 0; }
static inline int __all(int pred) { return emu::ballot(!pred) == 0; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned) v); }
static inline int __clzll(long long v) { return v == 0 ? 64 : __builtin_clzll((unsigned long long) v); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// amdgcn builtins used by the kernels (emulated)
template <class T> static inline T emu_readfirstlane(T v) {
    // first *active* lane: all fibers of the wave that are alive participate
    uint64_t* s = emu::wave_slots();
    s[emu::cur->lane] = emu::to_bits(v);
    emu::wave_barrier();
    unsigned long long alive = emu::wave_alive_mask();
    T r = emu::from_bits<T>(s[__builtin_ctzll(alive)]);
    emu::wave_barrier();
    return r;
}
#define __builtin_amdgcn_readfirstlane(x) emu_readfirstlane((int) (x))
#define __builtin_amdgcn_readlane(x, l) emu::shfl((int) (x), (l))
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
#define __builtin_amdgcn_sched_barrier(x) ((void) 0)
#define __builtin_amdgcn_mbcnt_lo(m, c) ((c) + __builtin_popcount((unsigned) (m) & (unsigned) (((1ull << (emu::cur->lane < 32 ? emu::cur->lane : 32)) - 1))))
#define __builtin_amdgcn_mbcnt_hi(m, c) ((c) + (emu::cur->lane > 32 ? __builtin_popcount((unsigned) (m) & (unsigned) ((1ull << (emu::cur->lane - 32)) - 1)) : 0))
#define __builtin_amdgcn_s_sleep(x) ((void) 0)
#define __builtin_amdgcn_wavefrontsize() 64

// ---------------------------------------------------------------- atomics (fibers never preempt each other)
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, long long v) { unsigned long long o = *p; *p = o + (unsigned long long) v; return o; }
static inline float atomicAdd(float* p, double v) { float o = *p; *p = o + (float) v; return o; }
template <class T> static inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; *p = std::max(o, v); return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; *p = std::min(o, v); return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

// ---------------------------------------------------------------- math
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline int __float2int_rn(float x) { return (int) nearbyintf(x); }
static inline double rsqrt(double x) { return 1.0 / sqrt(x); }
#define __expf(x) expf(x)
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frsqrt_rn(float x) { return 1.0f / sqrtf(x); }
static inline float __fsqrt_rn(float x) { return sqrtf(x); }
static inline float __saturatef(float x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }
static inline double __longlong_as_double(long long v) { double d; std::memcpy(&d, &v, 8); return d; }
static inline long long __double_as_longlong(double v) { long long d; std::memcpy(&d, &v, 8); return d; }
static inline float __int_as_float(int v) { float d; std::memcpy(&d, &v, 4); return d; }
static inline int __float_as_int(float v) { int d; std::memcpy(&d, &v, 4); return d; }
static inline unsigned __float_as_uint(float v) { unsigned d; std::memcpy(&d, &v, 4); return d; }
static inline float __uint_as_float(unsigned v) { float d; std::memcpy(&d, &v, 4); return d; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned) (((unsigned long long) a * b) >> 32); }
using std::min;
using std::max;

// ---------------------------------------------------------------- kernel launch
namespace emu {
template <class F, class... Args> struct LaunchPack {
    F f; std::tuple<Args...> args;
};
template <class F, class Tup, size_t... I> static inline void call_with(F f, Tup& t, std::index_sequence<I...>) { f(std::get<I>(t)...); }
}  // namespace emu
template <class... KArgs, class... Args>
static inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t /*shmem*/, hipStream_t /*stream*/, Args... args) {
    std::tuple<KArgs...> tup(static_cast<KArgs>(args)...);
    struct Pack { void (*k)(KArgs...); std::tuple<KArgs...>* t; } pack{kernel, &tup};
    emu::run_grid(grid, block, [](void* p) {
        Pack* pk = (Pack*) p;
        emu::call_with(pk->k, *pk->t, std::index_sequence_for<KArgs...>{});
    }, &pack);
}

// device clock (diagnostics only)
static inline long long clock64() { return 0; }

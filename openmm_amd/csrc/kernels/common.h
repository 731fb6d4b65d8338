// Shared device-side helpers for the MI355X (gfx950, wave64) kernels of the OpenMM "HIP" platform.
//
// Conventions used by every kernel in this directory
//   * Atoms live in two index spaces: "atom" = the System's particle index (state arrays pos/vel),
//     and "slot" = position in the spatially sorted, 32-atom-blocked order used by the nonbonded /
//     PME kernels (posq, sigmaEps, force).  atomOfSlot[slot] / slotOfAtom[atom] convert.
//   * Forces are accumulated as 64-bit fixed point (value * 2^32) in SoA layout
//     force[0..P) = x, force[P..2P) = y, force[2P..3P) = z with P = paddedAtoms, indexed by slot.
//     Integer accumulation makes the sum independent of the order of the atomics.
//   * A wavefront is 64 lanes.  Block sizes are multiples of 64.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>

#define OMM_TILE 32            // atoms per i-block
#define OMM_ROW 64             // j-atoms per neighbour-list row (= one wavefront)
#define OMM_CHUNK_ROWS 2       // rows per work chunk (all rows of a chunk share one i-block)
#define OMM_FORCE_SCALE 4294967296.0   // 2^32

// 1/(4 pi eps0) in kJ nm/(mol e^2) from the CODATA-2018 constants of platforms/reference/include/SimTKOpenMMRealType.h:74-89
#define OMM_ONE_4PI_EPS0_D 138.93545764438198
#define OMM_ONE_4PI_EPS0 138.93545764438198f

typedef long long omm_fixed;   // 64-bit fixed-point force component

namespace omm {

// Does the list builder keep the per-step pruned rows of this list (ommhip_neighbor_list::chunk_info_inner), and the pair kernel walk
// them?  One rule for both sides.  (A template so that this header need not know the struct; OPENMM_HIP_NO_PRUNE is an A/B knob.)
template <class NL>
inline bool list_is_pruned(const NL* nl) {
    static const bool off = getenv("OPENMM_HIP_NO_PRUNE") != nullptr;
    return !off && nl->cutoff > 0 && nl->pbc != 2 && nl->chunk_info_inner != nullptr && nl->row_j_inner != nullptr && nl->row_mask_inner != nullptr &&
           nl->block_runs != nullptr && nl->posq_rel != nullptr && nl->posq_ref_inner != nullptr;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// e^x through the hardware 2^x unit.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

__device__ __forceinline__ omm_fixed to_fixed(float v) { return (omm_fixed) ((double) v * OMM_FORCE_SCALE); }
__device__ __forceinline__ omm_fixed to_fixed(double v) { return (omm_fixed) (v * OMM_FORCE_SCALE); }
__device__ __forceinline__ double from_fixed(omm_fixed v) { return (double) v * (1.0 / OMM_FORCE_SCALE); }

__device__ __forceinline__ void atomic_add_fixed(omm_fixed* p, omm_fixed v) {
    atomicAdd((unsigned long long*) p, (unsigned long long) v);
}

__device__ __forceinline__ void add_force(omm_fixed* __restrict__ force, int paddedAtoms, int slot, float fx, float fy, float fz) {
    atomic_add_fixed(force + slot, to_fixed(fx));
    atomic_add_fixed(force + slot + paddedAtoms, to_fixed(fy));
    atomic_add_fixed(force + slot + 2 * paddedAtoms, to_fixed(fz));
}
__device__ __forceinline__ void add_force(omm_fixed* __restrict__ force, int paddedAtoms, int slot, double fx, double fy, double fz) {
    atomic_add_fixed(force + slot, to_fixed(fx));
    atomic_add_fixed(force + slot + paddedAtoms, to_fixed(fy));
    atomic_add_fixed(force + slot + 2 * paddedAtoms, to_fixed(fz));
}

// Two floats per lane: arithmetic on v2f compiles to the packed FP32 instructions (v_pk_add/mul/fma_f32), which issue at
// the rate of their scalar counterparts -- the only way to the FP32 peak of the CDNA3/4 vector ALUs.
#ifdef OMMHIP_EMU
struct v2f { float x, y; };
static inline v2f operator+(v2f a, v2f b) { return {a.x + b.x, a.y + b.y}; }
static inline v2f operator-(v2f a, v2f b) { return {a.x - b.x, a.y - b.y}; }
static inline v2f operator*(v2f a, v2f b) { return {a.x * b.x, a.y * b.y}; }
static inline v2f operator-(v2f a) { return {-a.x, -a.y}; }
#else
typedef float v2f __attribute__((ext_vector_type(2)));
#endif
__device__ __forceinline__ v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ v2f bc2(float a) { return mk2(a, a); }
// a * b + c with ONE rounding per element (v_pk_fma_f32): written out where a packed and a scalar code path must agree bit for bit
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) {
#ifdef OMMHIP_EMU
    return mk2(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y));
#else
    return __builtin_elementwise_fma(a, b, c);
#endif
}
// Squared length of a separation vector.  The pair kernel decides "inside the cutoff" on this number in its packed loops and
// re-derives the same decision in its (scalar) cutoff-edge path: explicit fused multiply-adds in a fixed order, so that the
// two agree bit for bit whatever the compiler's contraction choices are at either site.
__device__ __forceinline__ float r2_of(float dx, float dy, float dz) { return fmaf(dx, dx, fmaf(dy, dy, dz * dz)); }
__device__ __forceinline__ v2f r2_of(v2f dx, v2f dy, v2f dz) { return fma2(dx, dx, fma2(dy, dy, dz * dz)); }

// lanes of the wavefront for which the predicate holds, as a scalar (SGPR pair): accumulating such masks costs no vector instruction
__device__ __forceinline__ unsigned long long wave_ballot(bool p) {
#ifdef OMMHIP_EMU
    return __ballot(p ? 1 : 0);
#else
    return __builtin_amdgcn_ballot_w64(p);
#endif
}

// Wave-wide sum; every lane of the wave must call it.  Result valid in all lanes.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ int wave_min_int(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(v, m); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m));
    return v;
}

// acc |= in & ~lo on the scalar unit, as ONE dependency chain: written as plain C the compiler re-associates the ORs of an unrolled
// loop into a tree, keeps dozens of 64-bit masks alive and spills SGPRs through v_writelane / v_readlane inside the pair loops
__device__ __forceinline__ void mask_accumulate(unsigned long long& acc, unsigned long long in, unsigned long long lo) {
#ifdef OMMHIP_EMU
    acc |= in & ~lo;
#else
    const unsigned long long t = in & ~lo;
    asm("s_or_b64 %0, %0, %1" : "+s"(acc) : "s"(t));
#endif
}

// Number of set bits of `mask` below this lane.
__device__ __forceinline__ int lane_prefix_count(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned) (mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) mask, 0));
}

// Periodic box in reduced form a=(ax,0,0) b=(bx,by,0) c=(cx,cy,cz)  (openmmapi ContextImpl.cpp:267-275).
struct Box {
    float ax, bx, by, cx, cy, cz;
    float invAx, invBy, invCz;
    float axLo, byLo, czLo;     // box length minus its float value: ax + axLo is the edge to ~1e-14 nm (image shifts of block origins)
};

struct BoxD {
    double ax, bx, by, cx, cy, cz;
};

// Minimum-image displacement.  TRICLINIC=false assumes bx=cx=cy=0.
template <bool TRICLINIC>
__device__ __forceinline__ void min_image(float& dx, float& dy, float& dz, const Box& b) {
    if (TRICLINIC) {
        // platforms/reference/src/SimTKReference/ReferenceForce.cpp getDeltaRPeriodic (triclinic branch):
        // subtract multiples of c, then b, then a.
        // (explicit fused multiply-adds: the pair kernel's cutoff-edge path repeats this reduction and must land on the same bits)
        float s = rintf(dz * b.invCz);
        dx = fmaf(-s, b.cx, dx); dy = fmaf(-s, b.cy, dy); dz = fmaf(-s, b.cz, dz);
        s = rintf(dy * b.invBy);
        dx = fmaf(-s, b.bx, dx); dy = fmaf(-s, b.by, dy);
        s = rintf(dx * b.invAx);
        dx = fmaf(-s, b.ax, dx);
    }
    else {
        dx = fmaf(-rintf(dx * b.invAx), b.ax, dx);
        dy = fmaf(-rintf(dy * b.invBy), b.by, dy);
        dz = fmaf(-rintf(dz * b.invCz), b.cz, dz);
    }
}

// words of the neighbour-list state array (ommhip_neighbor_list::state)
//   REBUILD      request flag: set by the displacement check or by the host, cleared by the last builder workgroup
//   NUM_CHUNKS   chunks of the current list (published by the last builder workgroup; read by the pair kernel)
//   OVERFLOW     sticky: a rebuild needed more chunks than allocated (host grows the arrays and clears it)
//   ALLOC        working allocation counter of a rebuild in flight (zero at rest)
enum { ST_REBUILD = 0, ST_NUM_CHUNKS = 1, ST_OVERFLOW = 2, ST_BLOCKS_DONE = 3, ST_REBUILD_COUNT = 4, ST_ALLOC = 5,
       ST_FROZEN = 6 /* integrate.hip */,
       ST_NUM_CHUNKS_INNER = 7, ST_ALLOC_INNER = 8, ST_PRUNE_DONE = 9, ST_NO_PRUNE = 10, ST_PRUNE_REQUEST = 11 };

inline Box make_box(const double* bv) {
    // bv = {ax, bx, by, cx, cy, cz}
    Box b;
    b.ax = (float) bv[0]; b.bx = (float) bv[1]; b.by = (float) bv[2]; b.cx = (float) bv[3]; b.cy = (float) bv[4]; b.cz = (float) bv[5];
    b.invAx = (float) (1.0 / bv[0]); b.invBy = (float) (1.0 / bv[2]); b.invCz = (float) (1.0 / bv[5]);
    b.axLo = (float) (bv[0] - (double) b.ax); b.byLo = (float) (bv[2] - (double) b.by); b.czLo = (float) (bv[5] - (double) b.cz);
    return b;
}

__device__ __forceinline__ void min_image_d(double& dx, double& dy, double& dz, const BoxD& b) {
    double s = rint(dz / b.cz);
    dx -= s * b.cx; dy -= s * b.cy; dz -= s * b.cz;
    s = rint(dy / b.by);
    dx -= s * b.bx; dy -= s * b.by;
    s = rint(dx / b.ax);
    dx -= s * b.ax;
}

}  // namespace omm

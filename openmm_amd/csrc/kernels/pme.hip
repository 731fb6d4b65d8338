// Reciprocal-space PME for MI355X (gfx950): B-spline charge spreading, hand-written 3-D FFT,
// convolution and force interpolation; plus the classic Ewald k-sum used for small systems.
//
// Replaces (behaviourally) platforms/reference/src/SimTKReference/ReferencePME.cpp:
//   pme_update_grid_index_and_fraction :206-266, pme_update_bsplines :274-327,
//   pme_grid_spread_charge :330-405, fftpack_exec_3d (fftpack.cpp) via pme_exec :760-803,
//   pme_reciprocal_convolution :409-514, pme_grid_interpolate_force :617-713,
// and the Ewald k-sum of ReferenceLJCoulombIxn.cpp:272-367.
//
// Data layout in HBM
//   real grid    float  [nx][ny][nz]            (charge density, then potential)
//   half-complex float2 [nx][ny][nz/2+1]        (forward transform of the real grid)
//   eterm        float  [nx][ny][nz/2+1]        influence function, rebuilt only when the box changes
// The transform is unnormalised in both directions, exactly like fftpack in the reference, so the
// influence function carries all constants.
#include "common.h"
#include "../../../include/openmm_hip_kernels.h"
#include "../../../include/openmm_hip_comm.h"
#include <cstdlib>

using namespace omm;

namespace {

#define PME_ORDER 5
#define FFT_MAX_LDS 2048      // complex elements per ping-pong buffer (2 x 16 KB + 8 KB twiddles: 4 workgroups per CU)
#define FFT_THREADS 256
#define FFT_MAX_RADICES 12

struct RecipBox {   // reciprocal box vectors (rows), ReferencePME.cpp:196-204 (invert_box_vectors)
    float r00, r10, r11, r20, r21, r22;
};

struct PmeArgs {
    int paddedAtoms, nx, ny, nz, debug;
    RecipBox recip;
    const float4* posq;
    float* grid;
    omm_fixed* force;
    // folded Ewald exclusion correction (interpolation launch only)
    const int* exclStart; const int* exclAtoms; const int* atomOfSlot;
    const double4* pos; const double* charge;
    BoxD boxd;
    double alpha;
    int exclPeriodic, includeEnergy, energySlots;
    double* energyBuffer;
    // Slab decomposition (DD kernels only): this rank holds the x planes [planeLo - haloLo, planeLo + planeCount + haloHi) of
    // the real grid as local planes 0 .. gridPlanes-1; charges are spread onto the planeCount own planes only.
    int planeLo, planeCount, haloLo, gridPlanes;
    int ownSlot0, ownSlot1;
    int* ddError;
    const float4* blockCenter; const float4* blockHalf;
    float detScale;          // > 0: the grid is accumulated as int32 fixed point with this scale (deterministic sums), converted afterwards
    // tile spreading (spread_mode 2): tiles of TILE^3 cells over the spread range (x: the own planes), ntx * nty * ntz of them
    int* tileCount; int* tileBlocks; int tileCap, ntx, nty, ntz, numBlocks;
    float tileScale;         // fixed-point scale of the LDS accumulation (a power of two)
    int xcdBlocks;           // interpolation: workgroups per XCD when the launch is placed XCD-aware (0: plain order)
    // DD spreading over the ranges of slots with current positions only: range r = slots [activeBegin[r], activeEnd[r]), its workgroups are
    // [activeGroup0[r], activeGroup0[r + 1]) of the launch (set by launch_spread for its group size); numActive = 0: every slot
    int numActive, activeBegin[4], activeEnd[4], activeGroup0[5];
};

// one grid accumulation: float atomic, or -- for bit-reproducible sums -- an integer atomic on the same word
__device__ __forceinline__ void grid_add(const PmeArgs& a, size_t index, float v) {
    if (a.detScale > 0.f) atomicAdd((int*) &a.grid[index], __float2int_rn(v * a.detScale));
    else atomicAdd(&a.grid[index], v);
}

__global__ void pme_fixed_to_float(float* __restrict__ grid, size_t n, float invScale) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) grid[i] = (float) ((const int*) grid)[i] * invScale;
}

// local plane of global x plane gx for spreading (own planes only; -1 = not mine) and for interpolation (own + halo planes)
template <bool DD> __device__ __forceinline__ int spread_plane(const PmeArgs& a, int gx) {
    if (!DD) return gx;
    int d = gx - a.planeLo;
    if (d < 0) d += a.nx;
    return d < a.planeCount ? d + a.haloLo : -1;
}
template <bool DD> __device__ __forceinline__ int gather_plane(const PmeArgs& a, int gx) {
    if (!DD) return gx;
    int d = gx - (a.planeLo - a.haloLo);
    if (d < 0) d += a.nx;
    if (d >= a.nx) d -= a.nx;
    return d < a.gridPlanes ? d : -1;
}

// Grid index and B-spline weights of one coordinate.  ReferencePME.cpp:259-264 and :274-327 (order 5).
__device__ __forceinline__ void bspline(float t, int n, int& index, float (&theta)[PME_ORDER], float (&dtheta)[PME_ORDER]) {
    t = (t - floorf(t)) * n;
    int ti = (int) t;
    float dr = t - ti;
    index = ti % n;
    theta[PME_ORDER - 1] = 0.f;
    theta[1] = dr;
    theta[0] = 1.f - dr;
#pragma unroll
    for (int k = 3; k < PME_ORDER; k++) {
        float div = 1.f / (k - 1.f);
        theta[k - 1] = div * dr * theta[k - 2];
#pragma unroll
        for (int l = 1; l < k - 1; l++)
            theta[k - l - 1] = div * ((dr + l) * theta[k - l - 2] + (k - l - dr) * theta[k - l - 1]);
        theta[0] = div * (1.f - dr) * theta[0];
    }
    dtheta[0] = -theta[0];
#pragma unroll
    for (int k = 1; k < PME_ORDER; k++) dtheta[k] = theta[k - 1] - theta[k];
    const float div = 1.f / (PME_ORDER - 1);
    theta[PME_ORDER - 1] = div * dr * theta[PME_ORDER - 2];
#pragma unroll
    for (int l = 1; l < PME_ORDER - 1; l++)
        theta[PME_ORDER - l - 1] = div * ((dr + l) * theta[PME_ORDER - l - 2] + (PME_ORDER - l - dr) * theta[PME_ORDER - l - 1]);
    theta[0] = div * (1.f - dr) * theta[0];
}

__device__ __forceinline__ void atom_splines(const PmeArgs& a, float4 p, int (&idx)[3], float (&th)[3][PME_ORDER], float (&dth)[3][PME_ORDER]) {
    // fractional coordinates t_d = sum_k coord[k]*recip[k][d]           (ReferencePME.cpp:256-258)
    float tx = p.x * a.recip.r00 + p.y * a.recip.r10 + p.z * a.recip.r20;
    float ty = p.y * a.recip.r11 + p.z * a.recip.r21;
    float tz = p.z * a.recip.r22;
    bspline(tx, a.nx, idx[0], th[0], dth[0]);
    bspline(ty, a.ny, idx[1], th[1], dth[1]);
    bspline(tz, a.nz, idx[2], th[2], dth[2]);
}

// 8 lanes per atom; lane `sub` handles stencil points sub, sub+8, ... < 125.
__global__ __launch_bounds__(256) void pme_spread(PmeArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int slot = t >> 3, sub = t & 7;
    if (slot >= a.paddedAtoms) return;
    const float4 p = a.posq[slot];
    if (p.w == 0.f) return;
    int idx[3]; float th[3][PME_ORDER], dth[3][PME_ORDER];
    atom_splines(a, p, idx, th, dth);
    for (int pt = sub; pt < PME_ORDER * PME_ORDER * PME_ORDER; pt += 8) {
        const int ix = pt / 25, iy = (pt / 5) % 5, iz = pt % 5;
        int gx = idx[0] + ix; gx -= gx >= a.nx ? a.nx : 0;
        int gy = idx[1] + iy; gy -= gy >= a.ny ? a.ny : 0;
        int gz = idx[2] + iz; gz -= gz >= a.nz ? a.nz : 0;
        // dynamic indexing of the small theta arrays is resolved with selects after unrolling
        float wx = th[0][0], wy = th[1][0], wz = th[2][0];
#pragma unroll
        for (int k = 1; k < PME_ORDER; k++) { wx = ix == k ? th[0][k] : wx; wy = iy == k ? th[1][k] : wy; wz = iz == k ? th[2][k] : wz; }
        grid_add(a, ((size_t) gx * a.ny + gy) * a.nz + gz, p.w * wx * wy * wz);
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-staged spreading.  A workgroup takes SPREAD_ATOMS consecutive slots (spatially compact after the
// Hilbert sort), accumulates their 125-point stencils into a BRICK^3 sub-grid held in LDS with LDS
// atomics, and flushes the touched part of the brick to HBM with z-contiguous (coalesced) atomics.
// Device-scope float atomics resolve at the memory side on MI355X, one transaction per touched
// 64-byte line, so turning 125 scattered atomics per atom into a few hundred line-coalesced ones per
// workgroup is what makes this stage cheap.  Atoms whose stencil does not fit the brick (a group
// that straddles more than BRICK-5 cells) fall back to direct global atomics.
// ------------------------------------------------------------------------------------------------
// One workgroup = one 32-atom block of the spatial sort (the unit whose bounding box the neighbour list also uses):
// its atoms span ~7 grid cells, so stencils (+5) fit a 16^3 brick; 8 threads share the 125 points of an atom.
#define SPREAD_ATOMS 32
#define BRICK 16
#define BRICK_ZS (BRICK + 1)      // z stride of the LDS brick: an odd row length spreads a 5x5x5 stencil over the LDS banks
#define BRICK_WORDS (BRICK * BRICK * BRICK_ZS)

__device__ __forceinline__ int wrap_rel(int d, int n) {      // d in (-n, n) -> [-n/2, n/2)
    if (d >= (n + 1) / 2) d -= n;
    if (d < -(n / 2)) d += n;
    return d;
}

// LDS of one spread workgroup.  The brick accumulates in 32-bit fixed point: LDS integer atomics run ~9x faster than LDS
// float atomics on this chip (tools/microbench/lds_atomics.hip: 2.9 vs 0.33 lane-ops/clk/CU), and the sum is order independent.
struct SpreadShared {
    int touches;
    int brick[BRICK_WORDS];
    float brickScale;
    float th[SPREAD_ATOMS][3][PME_ORDER];
    int baseIdx[SPREAD_ATOMS][3];
    float charge[SPREAD_ATOMS];
    int ref[3];
    int minRel[3];
};

template <bool DD>
__device__ __forceinline__ void pme_spread_body(const PmeArgs& a, const int block, SpreadShared& sh) {
    // a block without a bounding box: empty, or (halo mode) one this rank holds no current positions for -- whatever its slots contain
    if (DD && a.blockHalf[block].x < 0.f) return;
    if (DD && a.recip.r10 == 0.f && a.recip.r20 == 0.f) {
        // rectangular box: blocks whose bounding box (+ stencil) lies clear of this rank's planes leave at once
        const float cx = a.blockCenter[block].x, hx = a.blockHalf[block].x;
        const float lo = (cx - hx) * a.recip.r00 * a.nx - 1.f, hi = (cx + hx) * a.recip.r00 * a.nx + (float) PME_ORDER;
        if (hi - lo < (float) a.nx) {
            // distance (in planes, periodic) from the block's plane interval [lo, hi] to the own interval
            const float ownLo = (float) a.planeLo, ownHi = (float) (a.planeLo + a.planeCount);
            float shift = floorf((0.5f * (lo + hi) - 0.5f * (ownLo + ownHi)) / (float) a.nx + 0.5f) * (float) a.nx;
            if (lo - shift > ownHi || hi - shift < ownLo) return;
        }
    }
    int* const brick = sh.brick;
    float& brickScale = sh.brickScale;
    float (&th)[SPREAD_ATOMS][3][PME_ORDER] = sh.th;
    int (&baseIdx)[SPREAD_ATOMS][3] = sh.baseIdx;
    float* const charge = sh.charge;
    int* const ref = sh.ref;
    int* const minRel = sh.minRel;
    const int t = threadIdx.x;
    const int slot0 = block * SPREAD_ATOMS;
    if (t < 3) { minRel[t] = 1 << 30; ref[t] = -1; }
    if (t == 0) sh.touches = DD ? 0 : 1;
    for (int i = t; i < BRICK_WORDS; i += 256) brick[i] = 0;
    // splines: thread (atom, dimension)
    if (t < 4 * SPREAD_ATOMS) {
        const int atom = t >> 2, d = t & 3;
        const int slot = slot0 + atom;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot < a.paddedAtoms) p = a.posq[slot];
        if (d == 3) charge[atom] = p.w;
        else if (p.w != 0.f) {
            // fractional coordinate along dimension d (ReferencePME.cpp:256-258)
            const float frac = d == 0 ? p.x * a.recip.r00 + p.y * a.recip.r10 + p.z * a.recip.r20
                             : (d == 1 ? p.y * a.recip.r11 + p.z * a.recip.r21 : p.z * a.recip.r22);
            const int nd = d == 0 ? a.nx : (d == 1 ? a.ny : a.nz);
            int idx; float theta[PME_ORDER], dtheta[PME_ORDER];
            bspline(frac, nd, idx, theta, dtheta);
            baseIdx[atom][d] = idx;
#pragma unroll
            for (int k = 0; k < PME_ORDER; k++) th[atom][d][k] = theta[k];
        }
    }
    __syncthreads();
    // reference cell = base index of the first charged atom of the group (wave 0 holds one atom per lane)
    if (t < 64) {
        const unsigned long long charged = __ballot(t < SPREAD_ATOMS && charge[t < SPREAD_ATOMS ? t : 0] != 0.f);
        if (charged != 0 && t == __ffsll((long long) charged) - 1) { ref[0] = baseIdx[t][0]; ref[1] = baseIdx[t][1]; ref[2] = baseIdx[t][2]; }
        // fixed-point scale: a power of two such that 32 atoms of the largest charge stacked on one cell stay below 2^30
        const float qmax = wave_max(t < SPREAD_ATOMS ? fabsf(charge[t]) : 0.f);
        if (t == 0) brickScale = exp2f(floorf(30.f - log2f(fmaxf(SPREAD_ATOMS * qmax, 1e-20f))));
    }
    __syncthreads();
    if (ref[0] < 0) return;                                   // no charged atom in this group
    const int n[3] = {a.nx, a.ny, a.nz};
    if (t < SPREAD_ATOMS && charge[t] != 0.f) {
#pragma unroll
        for (int d = 0; d < 3; d++) atomicMin(&minRel[d], wrap_rel(baseIdx[t][d] - ref[d], n[d]));
        if (DD) {
            bool touch = false;
#pragma unroll
            for (int k = 0; k < PME_ORDER; k++) { int gx = baseIdx[t][0] + k; gx -= gx >= a.nx ? a.nx : 0; touch = touch || spread_plane<DD>(a, gx) >= 0; }
            if (touch) sh.touches = 1;
        }
    }
    __syncthreads();
    if (DD && sh.touches == 0) return;                         // nothing of this block lands on the rank's planes
    // ---- accumulate: each wavefront takes 8 atoms, one at a time; its lanes are the stencil points (l and l + 64), so
    //      one LDS-atomic instruction never hits the same cell twice (neighbouring atoms -- a water's O, H, H -- share
    //      most of their cells, and same-address lanes serialise)
    if (!(a.debug & 1)) {
        const int lane = t & 63, wave = t >> 6;
        const int ptA = lane, ptB = lane + 64;
        const int ixA = ptA / 25, iyA = (ptA / 5) % 5, izA = ptA % 5;
        const int ixB = ptB / 25, iyB = (ptB / 5) % 5, izB = ptB % 5;
        const bool hasB = ptB < PME_ORDER * PME_ORDER * PME_ORDER;
        const float scale = brickScale;
        for (int k = 0; k < SPREAD_ATOMS / 4; k++) {
            const int atom = wave * (SPREAD_ATOMS / 4) + k;
            const float q = charge[atom];
            if (q == 0.f) continue;                             // wave-uniform
            int off[3];
            bool fits = true;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                off[d] = wrap_rel(baseIdx[atom][d] - ref[d], n[d]) - minRel[d];
                fits = fits && off[d] + PME_ORDER <= BRICK && BRICK <= n[d];
            }
            const float vA = q * th[atom][0][ixA] * th[atom][1][iyA] * th[atom][2][izA];
            const float vB = hasB ? q * th[atom][0][ixB] * th[atom][1][iyB] * th[atom][2][izB] : 0.f;
            if (fits) {
                atomicAdd(&brick[((off[0] + ixA) * BRICK + off[1] + iyA) * BRICK_ZS + off[2] + izA], __float2int_rn(vA * scale));
                if (hasB) atomicAdd(&brick[((off[0] + ixB) * BRICK + off[1] + iyB) * BRICK_ZS + off[2] + izB], __float2int_rn(vB * scale));
            }
            else {
                int gx = baseIdx[atom][0] + ixA; gx -= gx >= a.nx ? a.nx : 0;
                int gy = baseIdx[atom][1] + iyA; gy -= gy >= a.ny ? a.ny : 0;
                int gz = baseIdx[atom][2] + izA; gz -= gz >= a.nz ? a.nz : 0;
                gx = spread_plane<DD>(a, gx);
                if (gx >= 0) grid_add(a, ((size_t) gx * a.ny + gy) * a.nz + gz, vA);
                if (hasB) {
                    gx = baseIdx[atom][0] + ixB; gx -= gx >= a.nx ? a.nx : 0;
                    gy = baseIdx[atom][1] + iyB; gy -= gy >= a.ny ? a.ny : 0;
                    gz = baseIdx[atom][2] + izB; gz -= gz >= a.nz ? a.nz : 0;
                    gx = spread_plane<DD>(a, gx);
                    if (gx >= 0) grid_add(a, ((size_t) gx * a.ny + gy) * a.nz + gz, vB);
                }
            }
        }
    }
    __syncthreads();
    // ---- flush the brick: consecutive threads -> consecutive z -> coalesced atomics
    if (a.debug & 2) return;
    int org[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { org[d] = (ref[d] + minRel[d]) % n[d]; if (org[d] < 0) org[d] += n[d]; }
    const float invScale = 1.f / brickScale;
    for (int i = t; i < BRICK_WORDS; i += 256) {
        const int fixed = brick[i];
        if (fixed != 0) {
            const float v = (float) fixed * invScale;
            int gx = org[0] + i / (BRICK * BRICK_ZS); gx -= gx >= a.nx ? a.nx : 0;
            int gy = org[1] + (i / BRICK_ZS) % BRICK; gy -= gy >= a.ny ? a.ny : 0;
            int gz = org[2] + i % BRICK_ZS; gz -= gz >= a.nz ? a.nz : 0;
            gx = spread_plane<DD>(a, gx);
            if (gx >= 0) grid_add(a, ((size_t) gx * a.ny + gy) * a.nz + gz, v);
        }
    }
}

template <bool DD>
__global__ __launch_bounds__(256) void pme_spread_lds(PmeArgs a) {
    __shared__ SpreadShared sh;
    pme_spread_body<DD>(a, blockIdx.x, sh);
}

// ------------------------------------------------------------------------------------------------
// The same with G consecutive 32-atom blocks per workgroup and a B^3 brick (round 5; systems whose spreading is a launch of its own).
// What the brick scheme pays for is the flush: a grid cell receives one global atomic transaction from every brick that covers it --
// with one block per brick ~17 per cell on the 192^3 grid of the 1M-atom box (4.3 M line transactions for 30 798 blocks), which IS the
// kernel's time (profiles/r05b_atomics_by_scope.txt).  G consecutive blocks of the Hilbert order are a compact blob of G times the volume
// but (G^(1/3) x edge + stencil)^3 cells: 128 atoms touch ~15^3 cells where four separate bricks flush 4 x 11^3.  Atoms whose stencil
// leaves the brick (a blob longer than B - 5 cells along some axis) go through global atomics one by one, as before.
// ------------------------------------------------------------------------------------------------
template <int G, int B>
struct SpreadSharedG {
    int touches;
    int brick[B * B * (B + 1)];
    float brickScale;
    float th[G * SPREAD_ATOMS][3][PME_ORDER];
    int baseIdx[G * SPREAD_ATOMS][3];
    float charge[G * SPREAD_ATOMS];
    int ref[3];
    int minRel[3];
};

template <bool DD, int G, int B, int THREADS>
__global__ __launch_bounds__(THREADS) void pme_spread_group(PmeArgs a) {
    constexpr int ATOMS = G * SPREAD_ATOMS, ZS = B + 1, WORDS = B * B * ZS, WAVES = THREADS / 64;
    static_assert(ATOMS % WAVES == 0, "every wavefront takes the same number of atoms");
    __shared__ SpreadSharedG<G, B> sh;
    const int t = threadIdx.x;
    int slot0 = blockIdx.x * ATOMS, slotEnd = a.paddedAtoms;
    if (DD && a.numActive > 0) {
        // the launch covers the ranges this rank has positions for: which range is this workgroup's, and where in it
        // (compile-time indices only: a run-time index into an array of the by-value argument struct moves the whole struct to private memory)
        const int g = (int) blockIdx.x;
        int begin = a.activeBegin[0], first = 0, end = a.activeEnd[0];
#pragma unroll
        for (int r = 1; r < 4; r++)
            if (r < a.numActive && g >= a.activeGroup0[r]) { begin = a.activeBegin[r]; first = a.activeGroup0[r]; end = a.activeEnd[r]; }
        slot0 = begin + (g - first) * ATOMS;
        slotEnd = end;
    }
    if (t < 3) { sh.minRel[t] = 1 << 30; sh.ref[t] = -1; }
    if (t == 0) sh.touches = DD ? 0 : 1;
    for (int i = t; i < WORDS; i += THREADS) sh.brick[i] = 0;
    // splines: (atom, dimension) pairs; an atom of a block this rank holds no current positions for (halo mode) counts as uncharged
    for (int w = t; w < 4 * ATOMS; w += THREADS) {
        const int atom = w >> 2, d = w & 3;
        const int slot = slot0 + atom;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot < slotEnd && !(DD && a.blockHalf[slot / SPREAD_ATOMS].x < 0.f)) p = a.posq[slot];
        if (d == 3) sh.charge[atom] = p.w;
        else if (p.w != 0.f) {
            const float frac = d == 0 ? p.x * a.recip.r00 + p.y * a.recip.r10 + p.z * a.recip.r20
                             : (d == 1 ? p.y * a.recip.r11 + p.z * a.recip.r21 : p.z * a.recip.r22);
            const int nd = d == 0 ? a.nx : (d == 1 ? a.ny : a.nz);
            int idx; float theta[PME_ORDER], dtheta[PME_ORDER];
            bspline(frac, nd, idx, theta, dtheta);
            sh.baseIdx[atom][d] = idx;
#pragma unroll
            for (int k = 0; k < PME_ORDER; k++) sh.th[atom][d][k] = theta[k];
        }
    }
    __syncthreads();
    // reference cell = base index of the first charged atom; fixed-point scale from the largest charge (no more than SPREAD_ATOMS atoms of
    // it can pile their largest weight, 0.6^3, onto one cell: the scale of the one-block brick holds for G blocks as well)
    if (t < 64) {
        int first = ATOMS;
        float qmax = 0.f;
        for (int i = t; i < ATOMS; i += 64) { const float q = sh.charge[i]; if (q != 0.f && i < first) first = i; qmax = fmaxf(qmax, fabsf(q)); }
        first = wave_min_int(first);
        qmax = wave_max(qmax);
        if (t == 0) {
            if (first < ATOMS) { sh.ref[0] = sh.baseIdx[first][0]; sh.ref[1] = sh.baseIdx[first][1]; sh.ref[2] = sh.baseIdx[first][2]; }
            sh.brickScale = exp2f(floorf(30.f - log2f(fmaxf(SPREAD_ATOMS * qmax, 1e-20f))));
        }
    }
    __syncthreads();
    if (sh.ref[0] < 0) return;
    const int n[3] = {a.nx, a.ny, a.nz};
    for (int i = t; i < ATOMS; i += THREADS) {
        if (sh.charge[i] == 0.f) continue;
        if (DD) {
            // an atom none of whose five stencil planes is this rank's takes no part (a group at a slab boundary: about half of its atoms):
            // no LDS accumulation for it, and the brick shrinks to the atoms that count
            bool touch = false;
#pragma unroll
            for (int k = 0; k < PME_ORDER; k++) { int gx = sh.baseIdx[i][0] + k; gx -= gx >= a.nx ? a.nx : 0; touch = touch || spread_plane<DD>(a, gx) >= 0; }
            if (touch) sh.touches = 1;
            else { sh.charge[i] = 0.f; continue; }
        }
#pragma unroll
        for (int d = 0; d < 3; d++) atomicMin(&sh.minRel[d], wrap_rel(sh.baseIdx[i][d] - sh.ref[d], n[d]));
    }
    __syncthreads();
    if (DD && sh.touches == 0) return;
    {
        const int lane = t & 63, wave = t >> 6;
        const int ptA = lane, ptB = lane + 64;
        const int ixA = ptA / 25, iyA = (ptA / 5) % 5, izA = ptA % 5;
        const int ixB = ptB / 25, iyB = (ptB / 5) % 5, izB = ptB % 5;
        const bool hasB = ptB < PME_ORDER * PME_ORDER * PME_ORDER;
        const float scale = sh.brickScale;
        for (int k = 0; k < ATOMS / WAVES; k++) {
            const int atom = wave * (ATOMS / WAVES) + k;
            const float q = sh.charge[atom];
            if (q == 0.f) continue;                             // wave-uniform
            int off[3];
            bool fits = true;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                off[d] = wrap_rel(sh.baseIdx[atom][d] - sh.ref[d], n[d]) - sh.minRel[d];
                fits = fits && off[d] + PME_ORDER <= B && B <= n[d];
            }
            const float vA = q * sh.th[atom][0][ixA] * sh.th[atom][1][iyA] * sh.th[atom][2][izA];
            const float vB = hasB ? q * sh.th[atom][0][ixB] * sh.th[atom][1][iyB] * sh.th[atom][2][izB] : 0.f;
            if (fits) {
                atomicAdd(&sh.brick[((off[0] + ixA) * B + off[1] + iyA) * ZS + off[2] + izA], __float2int_rn(vA * scale));
                if (hasB) atomicAdd(&sh.brick[((off[0] + ixB) * B + off[1] + iyB) * ZS + off[2] + izB], __float2int_rn(vB * scale));
            }
            else {
                int gx = sh.baseIdx[atom][0] + ixA; gx -= gx >= a.nx ? a.nx : 0;
                int gy = sh.baseIdx[atom][1] + iyA; gy -= gy >= a.ny ? a.ny : 0;
                int gz = sh.baseIdx[atom][2] + izA; gz -= gz >= a.nz ? a.nz : 0;
                gx = spread_plane<DD>(a, gx);
                if (gx >= 0) grid_add(a, ((size_t) gx * a.ny + gy) * a.nz + gz, vA);
                if (hasB) {
                    gx = sh.baseIdx[atom][0] + ixB; gx -= gx >= a.nx ? a.nx : 0;
                    gy = sh.baseIdx[atom][1] + iyB; gy -= gy >= a.ny ? a.ny : 0;
                    gz = sh.baseIdx[atom][2] + izB; gz -= gz >= a.nz ? a.nz : 0;
                    gx = spread_plane<DD>(a, gx);
                    if (gx >= 0) grid_add(a, ((size_t) gx * a.ny + gy) * a.nz + gz, vB);
                }
            }
        }
    }
    __syncthreads();
    int org[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { org[d] = (sh.ref[d] + sh.minRel[d]) % n[d]; if (org[d] < 0) org[d] += n[d]; }
    const float invScale = 1.f / sh.brickScale;
    for (int i = t; i < WORDS; i += THREADS) {
        const int fixed = sh.brick[i];
        if (fixed != 0) {
            const float v = (float) fixed * invScale;
            int gx = org[0] + i / (B * ZS); gx -= gx >= a.nx ? a.nx : 0;
            int gy = org[1] + (i / ZS) % B; gy -= gy >= a.ny ? a.ny : 0;
            int gz = org[2] + i % ZS; gz -= gz >= a.nz ? a.nz : 0;
            gx = spread_plane<DD>(a, gx);
            if (gx >= 0) grid_add(a, ((size_t) gx * a.ny + gy) * a.nz + gz, v);
        }
    }
}

// blocks per brick of the stand-alone spreading launch: 1 (the 16^3 brick of the fused front launch), 4 (20^3) or 6 (22^3);
// OPENMM_HIP_SPREAD_GROUP overrides the default
static int spread_group_blocks(int padded_atoms) {
    static const int env = getenv("OPENMM_HIP_SPREAD_GROUP") != nullptr ? atoi(getenv("OPENMM_HIP_SPREAD_GROUP")) : -1;
    if (env >= 0) return env;
    return padded_atoms >= 200000 ? 2 : 1;          // 92 k atoms: one and two blocks within 0.3 % (profiles/r11/r11s_ab_misc.txt), 985 k: -2.5 %
}
template <bool DD>
static void launch_spread(const PmeArgs& pain, int padded_atoms, hipStream_t st) {
    PmeArgs pa = pain;
    const int g = spread_group_blocks(padded_atoms);          // 1, 2, 3, 4, 6; 14 / 18: four / eight blocks with 512-thread workgroups (A/B)
    int blocks = (padded_atoms + SPREAD_ATOMS - 1) / SPREAD_ATOMS;
    const bool bigEnough = pa.nx >= 24 && pa.ny >= 24 && pa.nz >= 24;
    static const bool allSlots = getenv("OPENMM_HIP_DD_SPREAD_ALL_SLOTS") != nullptr;          // A/B knob: the launch over every slot of the box (rounds 2-5)
    if (DD && pa.numActive > 0 && bigEnough && g == 2 && !allSlots) {
        // decomposed, halo mode: workgroups for the slot ranges with current positions only (a third of the box at eight ranks)
        pa.activeGroup0[0] = 0;
        for (int r = 0; r < pa.numActive; r++) pa.activeGroup0[r + 1] = pa.activeGroup0[r] + ((pa.activeEnd[r] - pa.activeBegin[r]) / SPREAD_ATOMS + 1) / 2;
        hipLaunchKernelGGL((pme_spread_group<DD, 2, 18, 256>), dim3(pa.activeGroup0[pa.numActive] > 0 ? pa.activeGroup0[pa.numActive] : 1), dim3(256), 0, st, pa);
        return;
    }
    pa.numActive = 0;
    if (!bigEnough || g <= 1) hipLaunchKernelGGL(pme_spread_lds<DD>, dim3(blocks), dim3(256), 0, st, pa);
    else if (g == 2) hipLaunchKernelGGL((pme_spread_group<DD, 2, 18, 256>), dim3((blocks + 1) / 2), dim3(256), 0, st, pa);
    else if (g == 3) hipLaunchKernelGGL((pme_spread_group<DD, 3, 19, 192>), dim3((blocks + 2) / 3), dim3(192), 0, st, pa);
    else if (g == 4) hipLaunchKernelGGL((pme_spread_group<DD, 4, 20, 256>), dim3((blocks + 3) / 4), dim3(256), 0, st, pa);
    else if (g == 6) hipLaunchKernelGGL((pme_spread_group<DD, 6, 22, 256>), dim3((blocks + 5) / 6), dim3(256), 0, st, pa);
    else if (g == 8) hipLaunchKernelGGL((pme_spread_group<DD, 8, 22, 1024>), dim3((blocks + 7) / 8), dim3(1024), 0, st, pa);
    else if (g == 16) hipLaunchKernelGGL((pme_spread_group<DD, 16, 26, 1024>), dim3((blocks + 15) / 16), dim3(1024), 0, st, pa);
    else if (g == 88) hipLaunchKernelGGL((pme_spread_group<DD, 8, 22, 512>), dim3((blocks + 7) / 8), dim3(512), 0, st, pa);
    else if (g == 12) hipLaunchKernelGGL((pme_spread_group<DD, 2, 18, 512>), dim3((blocks + 1) / 2), dim3(512), 0, st, pa);
    else if (g == 14) hipLaunchKernelGGL((pme_spread_group<DD, 4, 20, 512>), dim3((blocks + 3) / 4), dim3(512), 0, st, pa);
    else hipLaunchKernelGGL((pme_spread_group<DD, 2, 18, 256>), dim3((blocks + 1) / 2), dim3(256), 0, st, pa);
}


// ------------------------------------------------------------------------------------------------
// Tile spreading (spread_mode 2, systems too large for the fused front launch).  The brick scheme above turns an atom's 125
// scattered atomics into line-coalesced ones, but every grid cell still receives ~17 global atomic transactions (one per
// block whose brick covers it): 4.3 M line transactions for 30 798 blocks on a 192^3 grid.  Here the roles are swapped: ONE
// workgroup owns a tile of TILE^3 cells, accumulates -- in LDS, 32-bit fixed point, integer LDS atomics -- every stencil
// point that falls into it from the atoms of the 32-atom blocks whose bounding box plus stencil reaches the tile, and writes
// the tile once with plain stores.  No global atomics, no zeroed grid, and sums that do not depend on the order of arrival.
// pme_bin_tiles (one thread per block) fills the per-tile block lists every evaluation from the boxes nl_prepare keeps
// current; a tile whose list overflows scans all blocks itself.  Rectangular boxes only (a block's cell range is then a box).
// ------------------------------------------------------------------------------------------------
#define TILE 16
#define TILE_ZS (TILE + 1)
#define TILE_WORDS (TILE * TILE * TILE_ZS)
#define TILE_ATOMS 64         // atoms (two blocks) staged per round

// cells [lo, hi] (unwrapped) that the stencils of block b's atoms can touch along one axis
__device__ __forceinline__ void block_cell_range(float c, float h, float recip, int n, int& lo, int& hi) {
    lo = (int) floorf((c - h) * recip * (float) n) - 1;
    hi = (int) floorf((c + h) * recip * (float) n) + PME_ORDER;
}
// does the periodic image set of the cell interval [lo, hi] meet the tile's cells [t0, t1] (both in 0 .. n-1)?
__device__ __forceinline__ bool range_meets_tile(int lo, int hi, int t0, int t1, int n) {
    if (hi - lo + 1 >= n) return true;
    // some k with lo <= t1 + k n and t0 + k n <= hi
    const int kLo = (int) ceilf((float) (lo - t1) / (float) n), kHi = (int) floorf((float) (hi - t0) / (float) n);
    return kHi >= kLo;
}
template <bool DD> __device__ __forceinline__ int tile_x_origin(const PmeArgs& a, int tx) { return (DD ? a.planeLo : 0) + tx * TILE; }
template <bool DD> __device__ __forceinline__ int tile_x_end(const PmeArgs& a) { return DD ? a.planeLo + a.planeCount : a.nx; }

template <bool DD>
__global__ __launch_bounds__(256) void pme_bin_tiles(PmeArgs a) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= a.numBlocks) return;
    const float4 c = a.blockCenter[b], h = a.blockHalf[b];
    if (h.x < 0.f) return;                                     // empty block
    int lo[3], hi[3];
    block_cell_range(c.x, h.x, a.recip.r00, a.nx, lo[0], hi[0]);
    block_cell_range(c.y, h.y, a.recip.r11, a.ny, lo[1], hi[1]);
    block_cell_range(c.z, h.z, a.recip.r22, a.nz, lo[2], hi[2]);
    unsigned long long mx = 0, my = 0, mz = 0;
    const int xEnd = tile_x_end<DD>(a);
    for (int t = 0; t < a.ntx; t++) { const int t0 = tile_x_origin<DD>(a, t), t1 = min(t0 + TILE, xEnd) - 1; if (range_meets_tile(lo[0], hi[0], t0, t1, a.nx)) mx |= 1ull << t; }
    for (int t = 0; t < a.nty; t++) { const int t0 = t * TILE, t1 = min(t0 + TILE, a.ny) - 1; if (range_meets_tile(lo[1], hi[1], t0, t1, a.ny)) my |= 1ull << t; }
    for (int t = 0; t < a.ntz; t++) { const int t0 = t * TILE, t1 = min(t0 + TILE, a.nz) - 1; if (range_meets_tile(lo[2], hi[2], t0, t1, a.nz)) mz |= 1ull << t; }
    for (unsigned long long ix = mx; ix != 0; ix &= ix - 1) {
        const int tx = __ffsll((long long) ix) - 1;
        for (unsigned long long iy = my; iy != 0; iy &= iy - 1) {
            const int ty = __ffsll((long long) iy) - 1;
            for (unsigned long long iz = mz; iz != 0; iz &= iz - 1) {
                const int tz = __ffsll((long long) iz) - 1;
                const int tile = (tx * a.nty + ty) * a.ntz + tz;
                const int pos = atomicAdd(&a.tileCount[tile], 1);
                if (pos < a.tileCap) a.tileBlocks[(size_t) tile * a.tileCap + pos] = b;
            }
        }
    }
}

struct TileShared {
    int cells[TILE_WORDS];
    float th[TILE_ATOMS][3][PME_ORDER];
    int baseIdx[TILE_ATOMS][3];
    float charge[TILE_ATOMS];
};

template <bool DD>
__global__ __launch_bounds__(256) void pme_spread_tiles(PmeArgs a) {
    __shared__ TileShared sh;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tile = blockIdx.x;
    const int tz = tile % a.ntz, ty = (tile / a.ntz) % a.nty, tx = tile / (a.ntz * a.nty);
    const int org[3] = {tile_x_origin<DD>(a, tx), ty * TILE, tz * TILE};
    const int ext[3] = {min(TILE, tile_x_end<DD>(a) - org[0]), min(TILE, a.ny - org[1]), min(TILE, a.nz - org[2])};
    const int n[3] = {a.nx, a.ny, a.nz};
    for (int i = t; i < TILE_WORDS; i += 256) sh.cells[i] = 0;
    const int listed = a.tileCount[tile];
    const bool overflow = listed > a.tileCap;                  // more blocks than the list holds: look at every block
    const int numCand = overflow ? a.numBlocks : listed;
    const int ptA = lane, ptB = lane + 64;
    const int ixA = ptA / 25, iyA = (ptA / 5) % 5, izA = ptA % 5;
    const int ixB = ptB / 25, iyB = (ptB / 5) % 5, izB = ptB % 5;
    const bool hasB = ptB < PME_ORDER * PME_ORDER * PME_ORDER;
    for (int c0 = 0; c0 < numCand; c0 += TILE_ATOMS / 32) {
        __syncthreads();                                       // the previous round's tables are done with (and the cells are zeroed)
        // ---- splines: thread (atom, dimension); d = 3 carries the charge
        {
            const int atom = t >> 2, d = t & 3;
            const int ci = c0 + (atom >> 5);
            int blk = -1;
            if (ci < numCand) blk = overflow ? ci : a.tileBlocks[(size_t) tile * a.tileCap + ci];
            float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
            if (blk >= 0) p = a.posq[blk * 32 + (atom & 31)];
            if (d == 3) sh.charge[atom] = p.w;
            else if (p.w != 0.f) {
                const float frac = d == 0 ? p.x * a.recip.r00 : (d == 1 ? p.y * a.recip.r11 : p.z * a.recip.r22);
                int idx; float theta[PME_ORDER], dtheta[PME_ORDER];
                bspline(frac, n[d], idx, theta, dtheta);
                sh.baseIdx[atom][d] = idx;
#pragma unroll
                for (int k = 0; k < PME_ORDER; k++) sh.th[atom][d][k] = theta[k];
            }
        }
        __syncthreads();
        // ---- accumulate: each wavefront takes 16 atoms, one at a time; its lanes are the stencil points (l and l + 64)
        for (int k = 0; k < TILE_ATOMS / 4; k++) {
            const int atom = wave * (TILE_ATOMS / 4) + k;
            const float q = sh.charge[atom];
            if (q == 0.f) continue;                            // wave-uniform
            int rel[3];
            bool reaches = true;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                rel[d] = wrap_rel(sh.baseIdx[atom][d] - org[d], n[d]);        // base cell relative to the tile's origin, nearest image
                reaches = reaches && rel[d] > -PME_ORDER && rel[d] < ext[d];
            }
            if (!reaches) continue;                            // wave-uniform: the atom's stencil misses this tile
            const int cxA = rel[0] + ixA, cyA = rel[1] + iyA, czA = rel[2] + izA;
            if (cxA >= 0 && cxA < ext[0] && cyA >= 0 && cyA < ext[1] && czA >= 0 && czA < ext[2])
                atomicAdd(&sh.cells[(cxA * TILE + cyA) * TILE_ZS + czA], __float2int_rn(q * sh.th[atom][0][ixA] * sh.th[atom][1][iyA] * sh.th[atom][2][izA] * a.tileScale));
            if (hasB) {
                const int cxB = rel[0] + ixB, cyB = rel[1] + iyB, czB = rel[2] + izB;
                if (cxB >= 0 && cxB < ext[0] && cyB >= 0 && cyB < ext[1] && czB >= 0 && czB < ext[2])
                    atomicAdd(&sh.cells[(cxB * TILE + cyB) * TILE_ZS + czB], __float2int_rn(q * sh.th[atom][0][ixB] * sh.th[atom][1][iyB] * sh.th[atom][2][izB] * a.tileScale));
            }
        }
    }
    __syncthreads();
    // ---- write the tile: consecutive threads -> consecutive z
    const float invScale = 1.f / a.tileScale;
    for (int i = t; i < TILE * TILE * TILE; i += 256) {
        const int cz = i % TILE, cy = (i / TILE) % TILE, cx = i / (TILE * TILE);
        if (cx < ext[0] && cy < ext[1] && cz < ext[2]) {
            int gx = org[0] + cx; gx -= gx >= a.nx ? a.nx : 0;
            const int px = spread_plane<DD>(a, gx);
            a.grid[((size_t) px * a.ny + org[1] + cy) * a.nz + org[2] + cz] = (float) sh.cells[(cx * TILE + cy) * TILE_ZS + cz] * invScale;
        }
    }
}

// Launches the binning and the tile kernel when the evaluation qualifies; false = use the brick kernel.
template <bool DD>
static bool launch_tile_spread(const ommhip_pme* pme, PmeArgs pa, const void* block_center_d, const void* block_half_d, hipStream_t st) {
    if (pme->spread_mode != 2 || pme->tile_count == nullptr || pme->tile_blocks == nullptr || block_center_d == nullptr || block_half_d == nullptr) return false;
    if (pme->deterministic || pme->max_charge <= 0.0) return false;
    if (pa.recip.r10 != 0.f || pa.recip.r20 != 0.f || pa.recip.r21 != 0.f) return false;
    const int xCells = DD ? pa.planeCount : pa.nx;
    if (pa.nx < 2 * TILE || pa.ny < 2 * TILE || pa.nz < 2 * TILE) return false;
    pa.ntx = (xCells + TILE - 1) / TILE; pa.nty = (pa.ny + TILE - 1) / TILE; pa.ntz = (pa.nz + TILE - 1) / TILE;
    const long long tiles = (long long) pa.ntx * pa.nty * pa.ntz;
    if (pa.ntx > 64 || pa.nty > 64 || pa.ntz > 64 || tiles > pme->max_tiles || pme->tile_cap < 1) return false;
    pa.tileCount = pme->tile_count; pa.tileBlocks = pme->tile_blocks; pa.tileCap = pme->tile_cap;
    pa.blockCenter = (const float4*) block_center_d; pa.blockHalf = (const float4*) block_half_d;
    pa.numBlocks = pa.paddedAtoms / 32;
    // fixed-point scale: a power of two such that a cell's sum stays below 2^30 as long as sum |q w| <= 64 max|q|.  A cell collects
    // from the atoms of the 5^3 cells below it (17 atoms in water at 0.11 nm spacing, ~60 at diamond density and 0.15 nm), each
    // with a weight product of at most 0.22: <= 14 max|q| for any condensed-phase system -- the brick kernel's bound of 32 atoms
    // stacked on one cell had the same form.  Resolution: max|q| 2^-24 per contribution, as float32 accumulation would give.
    pa.tileScale = exp2f(floorf(30.f - log2f(64.f * (float) pme->max_charge)));
    hipMemsetAsync(pa.tileCount, 0, sizeof(int) * (size_t) tiles, st);
    hipLaunchKernelGGL(pme_bin_tiles<DD>, dim3((pa.numBlocks + 255) / 256), dim3(256), 0, st, pa);
    hipLaunchKernelGGL(pme_spread_tiles<DD>, dim3((unsigned) tiles), dim3(256), 0, st, pa);
    return true;
}

template <bool DD>
__global__ __launch_bounds__(256) void pme_interpolate(PmeArgs a) {
    // XCD-aware placement (a.xcdBlocks > 0): workgroups go to the 8 XCDs round-robin by index, and every XCD has an L2 of its own; workgroup b
    // takes the (b / 8)-th slot group of the (b % 8)-th eighth of the slots, so that an L2 sees one compact eighth of the grid (3.5 MB of a
    // 192^3 grid) instead of all of it.  Same idea as the pair kernel's chunk placement.
    int wg = blockIdx.x;
    if (a.xcdBlocks > 0) { const int x = wg % 8, k = wg / 8; wg = k < a.xcdBlocks ? x * a.xcdBlocks + k : (int) gridDim.x; }
    const int t = wg * blockDim.x + threadIdx.x;
    int slot = (DD ? a.ownSlot0 : 0) + (t >> 3);
    const int sub = t & 7;
    const bool valid = slot < (DD ? a.ownSlot1 : a.paddedAtoms);
    if (!valid) slot = a.paddedAtoms - 1;
    const float4 p = a.posq[slot];
    float fx = 0.f, fy = 0.f, fz = 0.f;
    if (valid && p.w != 0.f) {
        int idx[3]; float th[3][PME_ORDER], dth[3][PME_ORDER];
        atom_splines(a, p, idx, th, dth);
        // The 16 gathers of this lane (points sub, sub + 8, ...) are issued back to back and consumed afterwards: a rolled
        // loop waits for one L2 round trip per point, which was most of this kernel's time.
        constexpr int NPT = (PME_ORDER * PME_ORDER * PME_ORDER + 7) / 8;
        float g[NPT];
#pragma unroll
        for (int i = 0; i < NPT; i++) {
            const int pt = min(sub + 8 * i, PME_ORDER * PME_ORDER * PME_ORDER - 1);
            const int ix = pt / 25, iy = (pt / 5) % 5, iz = pt % 5;
            int gx = idx[0] + ix; gx -= gx >= a.nx ? a.nx : 0;
            int gy = idx[1] + iy; gy -= gy >= a.ny ? a.ny : 0;
            int gz = idx[2] + iz; gz -= gz >= a.nz ? a.nz : 0;
            gx = gather_plane<DD>(a, gx);
            if (DD && gx < 0) { *a.ddError = 1; gx = 0; }          // the atom drifted out of the planes this rank holds: flagged, the host re-sorts
            g[i] = a.grid[((size_t) gx * a.ny + gy) * a.nz + gz];
        }
#pragma unroll
        for (int i = 0; i < NPT; i++) {
            const int pt = sub + 8 * i;
            const int ix = pt / 25, iy = (pt / 5) % 5, iz = pt % 5;
            float wx = th[0][0], wy = th[1][0], wz = th[2][0], dx = dth[0][0], dy = dth[1][0], dz = dth[2][0];
#pragma unroll
            for (int k = 1; k < PME_ORDER; k++) {
                wx = ix == k ? th[0][k] : wx; wy = iy == k ? th[1][k] : wy; wz = iz == k ? th[2][k] : wz;
                dx = ix == k ? dth[0][k] : dx; dy = iy == k ? dth[1][k] : dy; dz = iz == k ? dth[2][k] : dz;
            }
            const float gi = pt < PME_ORDER * PME_ORDER * PME_ORDER ? g[i] : 0.f;
            fx += dx * wy * wz * gi;
            fy += wx * dy * wz * gi;
            fz += wx * wy * dz * gi;
        }
    }
    // Ewald exclusion correction of this atom (ReferenceLJCoulombIxn.cpp:462-523): lane `sub` takes partners sub, sub+8, ...
    // Separations come from the double-precision positions; erf/exp run in single precision like the pair kernel.
    float ex = 0.f, ey = 0.f, ez = 0.f;
    double exclEnergy = 0.0;
    if (a.exclStart != nullptr) {
        const int atom = valid ? a.atomOfSlot[slot] : -1;
        if (atom >= 0) {
            const int e1 = a.exclStart[atom + 1];
            int e = a.exclStart[atom] + sub;
            if (e < e1) {
                const double4 pi = a.pos[atom];
                const double qi = OMM_ONE_4PI_EPS0_D * a.charge[atom];
                for (; e < e1; e += 8) {
                    const int j = a.exclAtoms[e];
                    const double4 pj = a.pos[j];
                    double ddx = pj.x - pi.x, ddy = pj.y - pi.y, ddz = pj.z - pi.z;
                    if (a.exclPeriodic) min_image_d(ddx, ddy, ddz, a.boxd);
                    const float dx = (float) ddx, dy = (float) ddy, dz = (float) ddz;
                    const double qqd = qi * a.charge[j];
                    const float qq = (float) qqd;
                    const float r2 = dx * dx + dy * dy + dz * dz;
                    const float invR = rsqrtf(r2), r = r2 * invR;
                    const float ar = (float) a.alpha * r;
                    const float erfAr = erff(ar);
                    if (erfAr > 1e-6f) {
                        const float s = qq * invR * invR * invR * (erfAr - 2.0f * ar * expf(-ar * ar) * 0.56418958354775628695f);
                        ex += s * dx; ey += s * dy; ez += s * dz;
                        exclEnergy -= 0.5 * qqd * (double) (invR * erfAr);         // every pair is visited from both ends
                    }
                    else
                        exclEnergy -= 0.5 * a.alpha * 1.12837916709551257390 * qqd;
                }
            }
        }
    }
    // reduce the 8 lanes of this atom
    fx += __shfl_xor(fx, 1); fy += __shfl_xor(fy, 1); fz += __shfl_xor(fz, 1);
    fx += __shfl_xor(fx, 2); fy += __shfl_xor(fy, 2); fz += __shfl_xor(fz, 2);
    fx += __shfl_xor(fx, 4); fy += __shfl_xor(fy, 4); fz += __shfl_xor(fz, 4);
    if (a.exclStart != nullptr) {
        ex += __shfl_xor(ex, 1); ey += __shfl_xor(ey, 1); ez += __shfl_xor(ez, 1);
        ex += __shfl_xor(ex, 2); ey += __shfl_xor(ey, 2); ez += __shfl_xor(ez, 2);
        ex += __shfl_xor(ex, 4); ey += __shfl_xor(ey, 4); ez += __shfl_xor(ez, 4);
    }
    if (valid && sub == 0 && p.w != 0.f) {          // an uncharged atom has no exclusion correction either
        // ReferencePME.cpp:709-711
        const float q = p.w;
        const float gx = fx * a.nx, gy = fy * a.ny, gz = fz * a.nz;
        add_force(a.force, a.paddedAtoms, slot,
                  ex - q * (gx * a.recip.r00),
                  ey - q * (gx * a.recip.r10 + gy * a.recip.r11),
                  ez - q * (gx * a.recip.r20 + gy * a.recip.r21 + gz * a.recip.r22));
    }
    if (a.exclStart != nullptr && a.includeEnergy) {
        exclEnergy = wave_sum(exclEnergy);
        if ((threadIdx.x & 63) == 0) atomicAdd(&a.energyBuffer[(blockIdx.x * 4 + (threadIdx.x >> 6)) % a.energySlots], exclEnergy);
    }
}

// ------------------------------------------------------------------------------------------------
// Interpolation with FIVE lanes per atom (round 5): lane z of an atom walks the 25 (x, y) rows of the stencil at its own z offset, so the
// x and y weights are indexed by the (unrolled) loop counters and only the z weight is picked per lane, once.  The eight-lane kernel above
// picks all six weights of every point with compare / select chains (registers cannot be indexed by a lane-dependent value): ~40 VALU
// instructions per stencil point against 3 here -- it was bound by exactly that (1M atoms: 105 M wavefront instructions, 4 cycles each,
// = 170 us of its 235).  12 atoms per wavefront (lanes 60-63 idle); the five lanes of an atom read five consecutive floats of a z row.
// Same arithmetic as ReferencePME.cpp:617-713 up to the order of the sums.
// ------------------------------------------------------------------------------------------------
#define INTERP_LANES 5
#define INTERP_ATOMS_PER_WAVE 12
// register budget: 99 VGPRs (four wavefronts per SIMD) left alone; asked for six the compiler fits 68 without spilling (seven per SIMD):
// 177 -> 171 us at 1M atoms, 23.0 -> 22.0 at 92 k (`profiles/r11/r11n_ab_interpolate_occupancy.txt`)
#ifndef OMM_INTERP_WAVES
#define OMM_INTERP_WAVES 6
#endif
#ifdef OMMHIP_EMU
#define OMM_INTERP_ATTR
#else
#define OMM_INTERP_ATTR __attribute__((amdgpu_waves_per_eu(OMM_INTERP_WAVES)))
#endif
template <bool DD>
__global__ __launch_bounds__(256) OMM_INTERP_ATTR void pme_interpolate_z(PmeArgs a) {
    int wg = blockIdx.x;
    if (a.xcdBlocks > 0) { const int x = wg % 8, k = wg / 8; wg = k < a.xcdBlocks ? x * a.xcdBlocks + k : (int) gridDim.x; }
    const int lane = threadIdx.x & 63;
    const int wave = wg * (int) (blockDim.x >> 6) + (int) (threadIdx.x >> 6);
    const int sub = lane % INTERP_LANES, atomInWave = lane / INTERP_LANES;
    int slot = (DD ? a.ownSlot0 : 0) + wave * INTERP_ATOMS_PER_WAVE + atomInWave;
    const bool valid = atomInWave < INTERP_ATOMS_PER_WAVE && slot < (DD ? a.ownSlot1 : a.paddedAtoms) && wg < (int) gridDim.x;
    if (!valid) slot = a.paddedAtoms - 1;
    const float4 p = a.posq[slot];
    float fx = 0.f, fy = 0.f, fz = 0.f;
    if (valid && p.w != 0.f) {
        int idx[3]; float th[3][PME_ORDER], dth[3][PME_ORDER];
        atom_splines(a, p, idx, th, dth);
        float wz = th[2][0], dz = dth[2][0];
#pragma unroll
        for (int k = 1; k < PME_ORDER; k++) { wz = sub == k ? th[2][k] : wz; dz = sub == k ? dth[2][k] : dz; }
        int gz = idx[2] + sub; gz -= gz >= a.nz ? a.nz : 0;
        size_t rowX[PME_ORDER]; int rowY[PME_ORDER];
#pragma unroll
        for (int k = 0; k < PME_ORDER; k++) {
            int gx = idx[0] + k; gx -= gx >= a.nx ? a.nx : 0;
            gx = gather_plane<DD>(a, gx);
            if (DD && gx < 0) { *a.ddError = 1; gx = 0; }          // the atom drifted out of the planes this rank holds: flagged, the host re-sorts
            rowX[k] = (size_t) gx * a.ny;
            int gy = idx[1] + k; gy -= gy >= a.ny ? a.ny : 0;
            rowY[k] = gy;
        }
        // all 25 gathers of this lane are issued back to back and consumed afterwards
        float g[PME_ORDER][PME_ORDER];
#pragma unroll
        for (int ix = 0; ix < PME_ORDER; ix++)
#pragma unroll
            for (int iy = 0; iy < PME_ORDER; iy++) g[ix][iy] = a.grid[(rowX[ix] + rowY[iy]) * a.nz + gz];
        float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
        for (int ix = 0; ix < PME_ORDER; ix++) {
            float r = 0.f, rd = 0.f;          // sum over y of theta_y g and of dtheta_y g
#pragma unroll
            for (int iy = 0; iy < PME_ORDER; iy++) { r += th[1][iy] * g[ix][iy]; rd += dth[1][iy] * g[ix][iy]; }
            sx += dth[0][ix] * r; sy += th[0][ix] * rd; sz += th[0][ix] * r;
        }
        fx = sx * wz; fy = sy * wz; fz = sz * dz;
    }
    // Ewald exclusion correction of this atom (ReferenceLJCoulombIxn.cpp:462-523): lane `sub` takes partners sub, sub + 5, ...
    float ex = 0.f, ey = 0.f, ez = 0.f;
    double exclEnergy = 0.0;
    if (a.exclStart != nullptr) {
        const int atom = valid ? a.atomOfSlot[slot] : -1;
        if (atom >= 0) {
            const int e1 = a.exclStart[atom + 1];
            int e = a.exclStart[atom] + sub;
            if (e < e1) {
                const double4 pi = a.pos[atom];
                const double qi = OMM_ONE_4PI_EPS0_D * a.charge[atom];
                for (; e < e1; e += INTERP_LANES) {
                    const int j = a.exclAtoms[e];
                    const double4 pj = a.pos[j];
                    double ddx = pj.x - pi.x, ddy = pj.y - pi.y, ddz = pj.z - pi.z;
                    if (a.exclPeriodic) min_image_d(ddx, ddy, ddz, a.boxd);
                    const float dx = (float) ddx, dy = (float) ddy, dz = (float) ddz;
                    const double qqd = qi * a.charge[j];
                    const float qq = (float) qqd;
                    const float r2 = dx * dx + dy * dy + dz * dz;
                    const float invR = rsqrtf(r2), r = r2 * invR;
                    const float ar = (float) a.alpha * r;
                    const float erfAr = erff(ar);
                    if (erfAr > 1e-6f) {
                        const float sc = qq * invR * invR * invR * (erfAr - 2.0f * ar * expf(-ar * ar) * 0.56418958354775628695f);
                        ex += sc * dx; ey += sc * dy; ez += sc * dz;
                        exclEnergy -= 0.5 * qqd * (double) (invR * erfAr);         // every pair is visited from both ends
                    }
                    else
                        exclEnergy -= 0.5 * a.alpha * 1.12837916709551257390 * qqd;
                }
            }
        }
    }
    // the first lane of an atom collects the other four (lanes 60-63 hold zeros)
    float tx = fx, ty = fy, tz = fz, ux = ex, uy = ey, uz = ez;
#pragma unroll
    for (int k = 1; k < INTERP_LANES; k++) {
        tx += __shfl_down(fx, k); ty += __shfl_down(fy, k); tz += __shfl_down(fz, k);
        if (a.exclStart != nullptr) { ux += __shfl_down(ex, k); uy += __shfl_down(ey, k); uz += __shfl_down(ez, k); }
    }
    if (valid && sub == 0 && p.w != 0.f) {          // an uncharged atom has no exclusion correction either
        // ReferencePME.cpp:709-711
        const float q = p.w;
        const float gx = tx * a.nx, gy = ty * a.ny, gzf = tz * a.nz;
        add_force(a.force, a.paddedAtoms, slot,
                  ux - q * (gx * a.recip.r00),
                  uy - q * (gx * a.recip.r10 + gy * a.recip.r11),
                  uz - q * (gx * a.recip.r20 + gy * a.recip.r21 + gzf * a.recip.r22));
    }
    if (a.exclStart != nullptr && a.includeEnergy) {
        exclEnergy = wave_sum(exclEnergy);
        if ((threadIdx.x & 63) == 0) atomicAdd(&a.energyBuffer[(blockIdx.x * 4 + (threadIdx.x >> 6)) % a.energySlots], exclEnergy);
    }
}

// which interpolation kernel: five lanes per atom (default) or the eight-lane kernel of rounds 1-4 (OPENMM_HIP_INTERPOLATE_LANES=8, A/B)
static bool interpolate_by_z() {
    static const bool old = getenv("OPENMM_HIP_INTERPOLATE_LANES") != nullptr && atoi(getenv("OPENMM_HIP_INTERPOLATE_LANES")) == 8;
    return !old;
}
template <bool DD>
static void launch_interpolate(PmeArgs& pa, int slots, bool xcdPlacement, hipStream_t st) {
    const int perBlock = interpolate_by_z() ? 4 * INTERP_ATOMS_PER_WAVE : 32;
    int blocks = (slots + perBlock - 1) / perBlock;
    if (xcdPlacement && blocks >= 2048) { pa.xcdBlocks = (blocks + 7) / 8; blocks = pa.xcdBlocks * 8; }      // large systems: the grid no longer fits one L2
    if (blocks > 0) {
        if (interpolate_by_z()) hipLaunchKernelGGL(pme_interpolate_z<DD>, dim3(blocks), dim3(256), 0, st, pa);
        else hipLaunchKernelGGL(pme_interpolate<DD>, dim3(blocks), dim3(256), 0, st, pa);
    }
    pa.xcdBlocks = 0;
}

// ------------------------------------------------------------------------------------------------
// Influence function  eterm(kx,ky,kz) = ONE_4PI_EPS0 exp(-pi^2 m^2/alpha^2) / (pi V m^2 bx by bz)
// on the half grid (ReferencePME.cpp:409-514), evaluated in double and stored as float.
// ------------------------------------------------------------------------------------------------
struct EtermArgs {
    int nx, ny, nz, nzc;
    int y0, nyl;            // rows [y0, y0 + nyl) of the influence function are built (slab decomposition; whole grid: 0, ny)
    int dispersion;         // LJPME's dispersion grid: ReferencePME.cpp:518-614
    double r00, r10, r11, r20, r21, r22, alpha, volume;
    const double* modX; const double* modY; const double* modZ;
    float* eterm;
};

__global__ void pme_build_eterm(EtermArgs a) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t) a.nx * a.nyl * a.nzc;
    if (i >= total) return;
    const int kz = (int) (i % a.nzc), ky = a.y0 + (int) ((i / a.nzc) % a.nyl), kx = (int) (i / ((size_t) a.nzc * a.nyl));
    if (!a.dispersion && kx == 0 && ky == 0 && kz == 0) { a.eterm[i] = 0.f; return; }
    const int mx = kx < (a.nx + 1) / 2 ? kx : kx - a.nx;
    const int my = ky < (a.ny + 1) / 2 ? ky : ky - a.ny;
    const int mz = kz < (a.nz + 1) / 2 ? kz : kz - a.nz;
    const double mhx = mx * a.r00;
    const double mhy = mx * a.r10 + my * a.r11;
    const double mhz = mx * a.r20 + my * a.r21 + mz * a.r22;
    const double m2 = mhx * mhx + mhy * mhy + mhz * mhz;
    const double pi = 3.14159265358979323846;
    if (a.dispersion) {
        // dpme_reciprocal_convolution: every frequency contributes, m = 0 included
        const double boxfactor = -2.0 * pi * sqrt(pi) / (6.0 * a.volume);
        const double m = sqrt(m2), b = pi * m / a.alpha;
        const double term = 2.0 * pi * pi * pi * sqrt(pi) * erfc(b) * m * m2 + exp(-b * b) * (a.alpha * a.alpha * a.alpha - 2.0 * a.alpha * pi * pi * m2);
        a.eterm[i] = (float) (term * boxfactor / (a.modX[kx] * a.modY[ky] * a.modZ[kz]));
        return;
    }
    const double denom = m2 * (pi * a.volume * a.modX[kx]) * a.modY[ky] * a.modZ[kz];
    a.eterm[i] = (float) (OMM_ONE_4PI_EPS0_D * exp(-(pi * pi / (a.alpha * a.alpha)) * m2) / denom);
}

// ------------------------------------------------------------------------------------------------
// Mixed-radix Stockham FFT over lines staged in LDS.
// A workgroup transforms `B` lines of length n at once; element e of line l lives at lds[e*B + l].
// ------------------------------------------------------------------------------------------------
struct FftPlan {
    int n;
    int numRadices;
    unsigned long long radices;      // 4 bits per pass, first pass in the low bits: read with shifts (a table indexed by the pass
                                     // number would be a memory load -- and a wait for everything in flight -- in every pass)
};

struct FftArgs {
    FftPlan plan;
    int B;                 // lines per workgroup
    int numInner;          // lines are indexed (outer, inner); tiles of B consecutive inner lines
    int numOuter;
    long long inOuterStride, inInnerStride, inElemStride;     // in elements of the input type
    long long outOuterStride, outInnerStride, outElemStride;
    int mode;              // 0 c2c, 1 r2c (real in, first n/2+1 out), 2 c2r (n/2+1 in, real out), 3 c2c forward * eterm then backward
    int sign;              // -1 forward, +1 backward (ignored for mode 3)
    const float2* twiddle; // exp(-2 pi i k/n), k = 0..n-1
    const void* in;
    void* out;
    const float* eterm;    // mode 3: same indexing as `in`
    double* energyBuffer;  // mode 3 with energy
    int energySlots;
    int nzFull;            // mode 3: full z dimension (for the Hermitian weights); inner index = kz
    // slab decomposition, y pass only (outer = local x plane, inner = kz, element = ky): the side flagged here addresses the
    // transpose-ready layout [ky / remapNyl][x local of remapNxl][ky % remapNyl][kz] instead of the strides above
    int remapIn, remapOut, remapNxl, remapNyl;
    unsigned long long* diag;      // diagnostics (OMMHIP_FFT_DIAG=1): clock ticks of workgroup 0's thread 0 per phase, summed over launches
};

// (offset of element e of line `inner` of plane `outer` in the transpose-ready layout: (((e / Nyl) * Nxl + outer) * Nyl + e % Nyl) * numInner + inner)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// Complex numbers as v2f (re, im): an addition is ONE packed instruction (v_pk_add_f32), a multiplication by i a swizzle with a
// sign (source modifiers of the packed instruction that consumes it), a complex product a packed multiply and a packed fma.
// The radix passes are bound by VALU issue, and packed FP32 issues two results per lane per slot.
__device__ __forceinline__ v2f swp(v2f v) { return mk2(v.y, v.x); }
__device__ __forceinline__ v2f cmulp(v2f a, v2f w) { return bc2(w.x) * a + bc2(w.y) * mk2(-a.y, a.x); }
__device__ __forceinline__ v2f rot(v2f v, float fs) { return mk2(-fs, fs) * swp(v); }          // v * (i fs)

// Radix-R butterflies on R values held in registers: v[] -> o[], the DFT of length R with the sign of the pass (fs = -1 forward,
// +1 backward).  Split-radix style forms instead of the O(R^2) sum: radix 4 and 8 need no / two real constants, the odd radices
// pair v[k] with v[R-k] (cosine part on the sums, sine part on the differences).
template <int R> __device__ __forceinline__ void butterfly(const v2f (&v)[R], v2f (&o)[R], float fs);

template <> __device__ __forceinline__ void butterfly<2>(const v2f (&v)[2], v2f (&o)[2], float) {
    o[0] = v[0] + v[1]; o[1] = v[0] - v[1];
}
template <> __device__ __forceinline__ void butterfly<4>(const v2f (&v)[4], v2f (&o)[4], float fs) {
    const v2f t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3], t3 = rot(v[1] - v[3], fs);
    o[0] = t0 + t2; o[2] = t0 - t2; o[1] = t1 + t3; o[3] = t1 - t3;
}
template <> __device__ __forceinline__ void butterfly<8>(const v2f (&v)[8], v2f (&o)[8], float fs) {
    // two radix-4 transforms (even and odd inputs), odd outputs twisted by W8^k = exp(i fs pi k / 4)
    const v2f e[4] = {v[0], v[2], v[4], v[6]}, d[4] = {v[1], v[3], v[5], v[7]};
    v2f E[4], D[4];
    butterfly<4>(e, E, fs);
    butterfly<4>(d, D, fs);
    const float h = 0.70710678118654752440f;
    const v2f w1 = bc2(h) * (D[1] + rot(D[1], fs));          // D1 * (1 + i fs) / sqrt 2
    const v2f w2 = rot(D[2], fs);
    const v2f w3 = bc2(h) * (rot(D[3], fs) - D[3]);          // D3 * (-1 + i fs) / sqrt 2
    o[0] = E[0] + D[0]; o[4] = E[0] - D[0];
    o[1] = E[1] + w1;   o[5] = E[1] - w1;
    o[2] = E[2] + w2;   o[6] = E[2] - w2;
    o[3] = E[3] + w3;   o[7] = E[3] - w3;
}
template <> __device__ __forceinline__ void butterfly<3>(const v2f (&v)[3], v2f (&o)[3], float fs) {
    const v2f t = v[1] + v[2], d = v[1] - v[2];
    const v2f m = v[0] - bc2(0.5f) * t;
    const v2f n = rot(bc2(0.86602540378443864676f) * d, fs);
    o[0] = v[0] + t; o[1] = m + n; o[2] = m - n;
}
template <> __device__ __forceinline__ void butterfly<5>(const v2f (&v)[5], v2f (&o)[5], float fs) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f, s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const v2f t1 = v[1] + v[4], t2 = v[2] + v[3], d1 = v[1] - v[4], d2 = v[2] - v[3];
    const v2f m1 = v[0] + bc2(c1) * t1 + bc2(c2) * t2;
    const v2f m2 = v[0] + bc2(c2) * t1 + bc2(c1) * t2;
    const v2f n1 = rot(bc2(s1) * d1 + bc2(s2) * d2, fs);
    const v2f n2 = rot(bc2(s2) * d1 - bc2(s1) * d2, fs);
    o[0] = v[0] + t1 + t2;
    o[1] = m1 + n1; o[4] = m1 - n1; o[2] = m2 + n2; o[3] = m2 - n2;
}
template <> __device__ __forceinline__ void butterfly<7>(const v2f (&v)[7], v2f (&o)[7], float fs) {
    const float c1 = 0.62348980185873353053f, c2 = -0.22252093395631440429f, c3 = -0.90096886790241912624f;
    const float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f, s3 = 0.43388373911755812048f;
    const v2f t1 = v[1] + v[6], t2 = v[2] + v[5], t3 = v[3] + v[4];
    const v2f d1 = v[1] - v[6], d2 = v[2] - v[5], d3 = v[3] - v[4];
    // cos(2 pi j k / 7) for j, k = 1..3: row j = (c_j, c_2j, c_3j) with c4 = c3, c6 = c1, c9 = c2; sines likewise with s4 = -s3, s6 = -s1, s9 = s2
    const v2f m1 = v[0] + bc2(c1) * t1 + bc2(c2) * t2 + bc2(c3) * t3;
    const v2f m2 = v[0] + bc2(c2) * t1 + bc2(c3) * t2 + bc2(c1) * t3;
    const v2f m3 = v[0] + bc2(c3) * t1 + bc2(c1) * t2 + bc2(c2) * t3;
    const v2f n1 = rot(bc2(s1) * d1 + bc2(s2) * d2 + bc2(s3) * d3, fs);
    const v2f n2 = rot(bc2(s2) * d1 - bc2(s3) * d2 - bc2(s1) * d3, fs);
    const v2f n3 = rot(bc2(s3) * d1 - bc2(s1) * d2 + bc2(s2) * d3, fs);
    o[0] = v[0] + t1 + t2 + t3;
    o[1] = m1 + n1; o[6] = m1 - n1; o[2] = m2 + n2; o[5] = m2 - n2; o[3] = m3 + n3; o[4] = m3 - n3;
}

// One Stockham pass of radix R over all B lines.  `sign` selects forward (-1) or backward (+1).
// THREADS = workgroup size as a compile-time constant: blockDim.x is a 16-bit load from the dispatch packet, i.e. a wait for
// every global read in flight, in every pass.
// FIRST: the pass with Ns = 1, whose twiddle factors are all 1.
template <int R, int THREADS, bool FIRST>
// Element e of line l lives at e*BP + l*LS (BP = element stride, LS = line stride).
__device__ __forceinline__ void fft_pass(const float2* __restrict__ src, float2* __restrict__ dst, int n, int Ns, int B, int BP, int sign,
                                         const float2* __restrict__ tw, int LS) {
    // `tw` is the LDS copy of the twiddle table exp(-2 pi i k/n).
    const int butterflies = n / R;
    const float fsign = (float) -sign;             // table holds exp(-i...), i.e. forward
    const float fs = (float) sign;
    const int twStep = n / (Ns * R);               // twiddle index increment per r
    const bool pow2B = (B & (B - 1)) == 0;
    const int shiftB = 31 - __clz(B);
    const bool pow2Ns = (Ns & (Ns - 1)) == 0;      // radix-2/4/8 passes come first: shifts instead of a division
    const int shiftNs = 31 - __clz(Ns);
    for (int idx = threadIdx.x; idx < butterflies * B; idx += THREADS) {
        int line, j;
        if (pow2B) { line = idx & (B - 1); j = idx >> shiftB; } else { j = idx / B; line = idx - j * B; }
        int q, k;
        if (FIRST) { q = j; k = 0; }
        else if (pow2Ns) { q = j >> shiftNs; k = j & (Ns - 1); }
        else { q = j / Ns; k = j - q * Ns; }
        v2f v[R];
        { const float2 x = src[j * BP + line * LS]; v[0] = mk2(x.x, x.y); }
#pragma unroll
        for (int r = 1; r < R; r++) {
            const float2 x = src[(j + r * butterflies) * BP + line * LS];
            if (FIRST) v[r] = mk2(x.x, x.y);
            else {
                const float2 w = tw[k * r * twStep];         // < n because k < Ns and r < R
                v[r] = cmulp(mk2(x.x, x.y), mk2(w.x, w.y * fsign));
            }
        }
        v2f o[R];
        butterfly<R>(v, o, fs);
        const int j0 = q * Ns * R + k;
#pragma unroll
        for (int p = 0; p < R; p++) dst[(j0 + p * Ns) * BP + line * LS] = make_float2(o[p].x, o[p].y);
    }
}

// Runs all passes; returns the buffer that holds the result.
template <int THREADS>
__device__ __forceinline__ float2* fft_lines(const FftPlan& plan, float2* bufA, float2* bufB, int B, int BP, int sign, const float2* tw, int LS = 1) {
    float2* src = bufA;
    float2* dst = bufB;
    int Ns = 1;
    for (int s = 0; s < plan.numRadices; s++) {
        const int R = (int) ((plan.radices >> (4 * s)) & 15ull);
        if (s == 0) {
            switch (R) {
                case 2: fft_pass<2, THREADS, true>(src, dst, plan.n, Ns, B, BP, sign, tw, LS); break;
                case 3: fft_pass<3, THREADS, true>(src, dst, plan.n, Ns, B, BP, sign, tw, LS); break;
                case 4: fft_pass<4, THREADS, true>(src, dst, plan.n, Ns, B, BP, sign, tw, LS); break;
                case 5: fft_pass<5, THREADS, true>(src, dst, plan.n, Ns, B, BP, sign, tw, LS); break;
                case 7: fft_pass<7, THREADS, true>(src, dst, plan.n, Ns, B, BP, sign, tw, LS); break;
                default: fft_pass<8, THREADS, true>(src, dst, plan.n, Ns, B, BP, sign, tw, LS); break;
            }
        }
        else {
            switch (R) {
                case 2: fft_pass<2, THREADS, false>(src, dst, plan.n, Ns, B, BP, sign, tw, LS); break;
                case 3: fft_pass<3, THREADS, false>(src, dst, plan.n, Ns, B, BP, sign, tw, LS); break;
                case 4: fft_pass<4, THREADS, false>(src, dst, plan.n, Ns, B, BP, sign, tw, LS); break;
                case 5: fft_pass<5, THREADS, false>(src, dst, plan.n, Ns, B, BP, sign, tw, LS); break;
                case 7: fft_pass<7, THREADS, false>(src, dst, plan.n, Ns, B, BP, sign, tw, LS); break;
                default: fft_pass<8, THREADS, false>(src, dst, plan.n, Ns, B, BP, sign, tw, LS); break;
            }
        }
        Ns *= R;
        __syncthreads();
        float2* t = src; src = dst; dst = t;
    }
    return src;
}

struct FftShared {
    float2 bufA[FFT_MAX_LDS];
    float2 bufB[FFT_MAX_LDS];
    double energyPartial[8];
    float2 twS[FFT_MAX_LDS / 2];          // n <= FFT_MAX_LDS/2 (ommhip_fft_supported_size)
};

// THREADS = workgroup size (256 as a kernel of its own; fused launches may differ); `block` = first tile of this workgroup,
// `tileStride` > 0: the workgroup goes on with tiles block + tileStride, block + 2 tileStride, ... < numTiles, and the global
// reads of the next tile are issued before the passes of the current one (they land in the registers the current tile has
// just left for LDS), so memory stays busy while the butterflies run.  A line-pass workgroup alone keeps ~12 KB in flight
// for a third of its life; at 4 workgroups per CU that is a fifth of what 8 TB/s needs.
template <int THREADS, int MODE, bool REMAP_IN, bool REMAP_OUT>
__device__ __forceinline__ void fft_body_mode(const FftArgs& a, const int block, FftShared& sh, const int tileStride, const int numTiles) {
    float2* const bufA = sh.bufA;
    float2* const bufB = sh.bufB;
    double* const energyPartial = sh.energyPartial;
    float2* const twS = sh.twS;
    const int n = a.plan.n, B = a.B, BP = a.B + 1;   // LDS line stride B+1: conflict-free for both staging orders
    for (int i = threadIdx.x; i < n; i += THREADS) twS[i] = a.twiddle[i];
    const int tilesPerOuter = (a.numInner + B - 1) / B;
    const int nIn = MODE == 2 ? n / 2 + 1 : n;
    const int nOut = MODE == 1 ? n / 2 + 1 : n;
    const bool elemFastIn = a.inElemStride == 1;
    const bool elemFastOut = a.outElemStride == 1;
    // ---- load: all global reads of this thread are issued back to back before anything is consumed; a rolled loop would
    //      wait for one round trip per element
    constexpr int MAXLD = FFT_MAX_LDS / THREADS;          // (B + 1) * n <= FFT_MAX_LDS
    float2 ld[MAXLD];
    unsigned ldZero = 0;          // bit it: ld[it] belongs to a line beyond the end and is staged as zero
    float etv[MAXLD];
    // What a thread reads and writes is the same in every tile up to the tile's origin: LDS position, line within the tile and
    // memory offset relative to the origin are computed once (the divisions of the index split are a third of a tile's
    // instructions otherwise).  pack = LDS position | line << 16, or -1 for "no element".
    int ldPack[MAXLD], stPack[MAXLD];
    unsigned ldRel[MAXLD], stRel[MAXLD];
#pragma unroll
    for (int it = 0; it < MAXLD; it++) {
        const int idx = threadIdx.x + it * THREADS;
        int line, e;
        if (elemFastIn) { e = idx % nIn; line = idx / nIn; } else { line = idx % B; e = idx / B; }
        ldPack[it] = idx < nIn * B ? (e * BP + line) | (line << 16) : -1;
        if (REMAP_IN) { const int q = e / a.remapNyl, yl = e - q * a.remapNyl; ldRel[it] = (unsigned) ((q * a.remapNxl * a.remapNyl + yl) * a.numInner + line); }
        else ldRel[it] = (unsigned) (line * a.inInnerStride + e * a.inElemStride);
        if (elemFastOut) { e = idx % nOut; line = idx / nOut; } else { line = idx % B; e = idx / B; }
        stPack[it] = idx < nOut * B ? (e * BP + line) | (line << 16) : -1;
        if (REMAP_OUT) { const int q = e / a.remapNyl, yl = e - q * a.remapNyl; stRel[it] = (unsigned) ((q * a.remapNxl * a.remapNyl + yl) * a.numInner + line); }
        else stRel[it] = (unsigned) (line * a.outInnerStride + e * a.outElemStride);
    }
    auto originIn = [&](int outer, int inner0) -> long long {
        return REMAP_IN ? (long long) outer * a.remapNyl * a.numInner + inner0 : outer * a.inOuterStride + inner0 * a.inInnerStride;
    };
    auto originOut = [&](int outer, int inner0) -> long long {
        return REMAP_OUT ? (long long) outer * a.remapNyl * a.numInner + inner0 : outer * a.outOuterStride + inner0 * a.outInnerStride;
    };
    // Every read is unconditional (lanes without an element read element 0 and discard it): no branches, so the compiler can
    // issue all of them before the first wait.
    auto loadTile = [&](int tile) {
        const int outer = tile / tilesPerOuter, inner0 = (tile % tilesPerOuter) * B;
        const long long origin = originIn(outer, inner0);
        ldZero = 0;
#pragma unroll
        for (int it = 0; it < MAXLD; it++) {
            const bool valid = ldPack[it] >= 0 && inner0 + (ldPack[it] >> 16) < a.numInner;
            const long long off = valid ? origin + ldRel[it] : 0;
            float2 v;
            if (MODE == 1) v = make_float2(((const float*) a.in)[off], 0.f);
            else v = ((const float2*) a.in)[off];
            ld[it] = v;                                   // not looked at here: a use would make the compiler wait for it
            ldZero |= (valid ? 0u : 1u) << it;
        }
    };
    // influence function of the tile being transformed (mode 3; same indexing as the input): consumed after the forward passes,
    // so it is requested at their start
    auto loadEterm = [&](int tile) {
        const int outer = tile / tilesPerOuter, inner0 = (tile % tilesPerOuter) * B;
        const long long origin = originIn(outer, inner0);
#pragma unroll
        for (int it = 0; it < MAXLD; it++) {
            const bool valid = ldPack[it] >= 0 && inner0 + (ldPack[it] >> 16) < a.numInner;
            etv[it] = a.eterm[valid ? origin + ldRel[it] : 0];      // lanes beyond the end never use it
        }
    };
    const bool diagOn = a.diag != nullptr && threadIdx.x == 0 && block == 0;
    long long dT0 = diagOn ? clock64() : 0, dStage = 0, dPass = 0, dStore = 0, dIssue = 0;
    loadTile(block);
    __builtin_amdgcn_sched_barrier(0);          // all reads of the first tile are issued before its staging starts
    for (int tile = block; ; ) {
        const int outer = tile / tilesPerOuter, inner0 = (tile % tilesPerOuter) * B;
        long long dA = diagOn ? clock64() : 0;
#pragma unroll
        for (int it = 0; it < MAXLD; it++) {
            if (ldPack[it] >= 0) {
                const float2 v = (ldZero >> it) & 1u ? make_float2(0.f, 0.f) : ld[it];
                const int pos = ldPack[it] & 0xFFFF, line = ldPack[it] >> 16;
                bufA[pos] = v;
                if (MODE == 2) {
                    const int e = (pos - line) / BP;
                    if (e > 0 && e < n - e) bufA[(n - e) * BP + line] = make_float2(v.x, -v.y);   // Hermitian completion
                }
            }
        }
        __syncthreads();
        long long dB = diagOn ? clock64() : 0;
        const int next = tile + tileStride;
        const bool more = tileStride > 0 && next < numTiles;
        if (MODE == 3) loadEterm(tile);
        if (more) loadTile(next);
        long long dC = diagOn ? clock64() : 0;
        float2* res;
        if (MODE == 3) {
            res = fft_lines<THREADS>(a.plan, bufA, bufB, B, BP, -1, twS);
            double energy = 0;
#pragma unroll
            for (int it = 0; it < MAXLD; it++) {
                if (ldPack[it] >= 0) {
                    const int pos = ldPack[it] & 0xFFFF, line = ldPack[it] >> 16;
                    float2 v = res[pos];
                    const float et = etv[it];
                    const int kz = inner0 + line;
                    if (a.energyBuffer != nullptr) {
                        const float wgt = (kz == 0 || 2 * kz == a.nzFull) ? 1.f : 2.f;
                        energy += (double) (wgt * et * (v.x * v.x + v.y * v.y));
                    }
                    res[pos] = make_float2(v.x * et, v.y * et);
                }
            }
            if (a.energyBuffer != nullptr) {
                energy = wave_sum(energy);
                if ((threadIdx.x & 63) == 0) energyPartial[threadIdx.x >> 6] = energy;
            }
            __syncthreads();
            if (a.energyBuffer != nullptr && threadIdx.x == 0) {
                double e = 0;
                for (int w = 0; w < THREADS / 64; w++) e += energyPartial[w];
                atomicAdd(&a.energyBuffer[tile % a.energySlots], 0.5 * e);
            }
            float2* other = res == bufA ? bufB : bufA;
            res = fft_lines<THREADS>(a.plan, res, other, B, BP, +1, twS);
        }
        else {
            res = fft_lines<THREADS>(a.plan, bufA, bufB, B, BP, a.sign, twS);
        }
        // ---- store
        long long dD = diagOn ? clock64() : 0;
        const long long outOrigin = originOut(outer, inner0);
#pragma unroll
        for (int it = 0; it < MAXLD; it++) {
            if (stPack[it] >= 0 && inner0 + (stPack[it] >> 16) < a.numInner) {
                const float2 v = res[stPack[it] & 0xFFFF];
                if (MODE == 2) ((float*) a.out)[outOrigin + stRel[it]] = v.x;
                else ((float2*) a.out)[outOrigin + stRel[it]] = v;
            }
        }
        if (diagOn) { const long long dE = clock64(); dStage += dB - dA; dIssue += dC - dB; dPass += dD - dC; dStore += dE - dD; }
        if (!more) break;
        tile = next;
        __syncthreads();          // the next tile is staged into bufA, which the last pass may still be read from
    }
    if (diagOn) {
        unsigned long long* d = a.diag + 8 * MODE;
        atomicAdd(&d[0], 1ull); atomicAdd(&d[1], (unsigned long long) (clock64() - dT0)); atomicAdd(&d[2], (unsigned long long) dStage);
        atomicAdd(&d[3], (unsigned long long) dIssue); atomicAdd(&d[4], (unsigned long long) dPass); atomicAdd(&d[5], (unsigned long long) dStore);
    }
}

// Tiles per launch of the line-pass kernel: every workgroup resident at once (4 per CU), each with the same number of tiles
static int fft_grid(int numTiles) {
    static const char* forced = getenv("OMMHIP_FFT_RESIDENT_WORKGROUPS");      // tests: a few workgroups walk through many tiles
    const int resident = forced != nullptr && atoi(forced) > 0 ? atoi(forced) : 1024;
    const int rounds = (numTiles + resident - 1) / resident;
    return (numTiles + rounds - 1) / rounds;
}

// One kernel per (mode, remap) combination: each gets its own register allocation (the six bodies in one kernel needed 145 VGPRs
// and spilled 180 SGPRs).
template <int MODE, bool REMAP_IN, bool REMAP_OUT>
__global__ __launch_bounds__(FFT_THREADS) void fft_kernel(FftArgs a) {
    __shared__ FftShared sh;
    fft_body_mode<FFT_THREADS, MODE, REMAP_IN, REMAP_OUT>(a, blockIdx.x, sh, gridDim.x, a.numOuter * ((a.numInner + a.B - 1) / a.B));
}

static unsigned long long* g_fftDiag = nullptr;
extern "C" int ommhip_fft_diag(unsigned long long* out32) {      // diagnostics only (tools/diag_fft_phases.py); not part of the product ABI
    if (g_fftDiag == nullptr) return 1;
    hipDeviceSynchronize();
    return (int) hipMemcpy(out32, g_fftDiag, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}

static void launch_fft(const FftArgs& fin, hipStream_t st) {
    FftArgs f = fin;
    static const bool diag = getenv("OMMHIP_FFT_DIAG") != nullptr;
    if (diag && g_fftDiag == nullptr) { hipMalloc((void**) &g_fftDiag, 32 * sizeof(unsigned long long)); hipMemset(g_fftDiag, 0, 32 * sizeof(unsigned long long)); }
    f.diag = g_fftDiag;
    const dim3 grid(fft_grid(f.numOuter * ((f.numInner + f.B - 1) / f.B))), block(FFT_THREADS);
    if (f.mode == 1) hipLaunchKernelGGL((fft_kernel<1, false, false>), grid, block, 0, st, f);
    else if (f.mode == 2) hipLaunchKernelGGL((fft_kernel<2, false, false>), grid, block, 0, st, f);
    else if (f.mode == 3) hipLaunchKernelGGL((fft_kernel<3, false, false>), grid, block, 0, st, f);
    else if (f.remapIn) hipLaunchKernelGGL((fft_kernel<0, true, false>), grid, block, 0, st, f);
    else if (f.remapOut) hipLaunchKernelGGL((fft_kernel<0, false, true>), grid, block, 0, st, f);
    else hipLaunchKernelGGL((fft_kernel<0, false, false>), grid, block, 0, st, f);
}

// ------------------------------------------------------------------------------------------------
// Fused plane transform: one workgroup does the z and the y transform of one x-plane entirely in LDS,
// so the (y,z) half of the 3-D transform touches HBM once instead of twice and costs one launch
// instead of two.  forward: real [ny][nz] -> z r2c -> y c2c -> complex [ny][nz/2+1];  backward: the
// reverse.  Used when a plane fits the LDS budget (PLANE_MAX complex elements per buffer).
// LDS layout of a plane: element (y, kz) at kz*(ny+1) + y; the z transform sees lines = y (stride 1),
// elements = z (stride ny+1); the y transform sees lines = kz (stride ny+1), elements = y (stride 1).
// ------------------------------------------------------------------------------------------------
#define PLANE_MAX 9472        // complex elements per LDS buffer: 2 x 74 KB + 4 KB twiddles of the 160 KB per CU (planes up to 96 x 96)
#define PLANE_THREADS 512     // one plane has ~400 butterflies per pass: 512 threads finish a pass in one sweep

struct PlaneArgs {
    FftPlan planY, planZ, planZh;      // planZh: nz / 2 points (large planes, fft_bigplane_kernel)
    int ny, nz, forward;
    const float2* twY; const float2* twZ;
    float* real;           // [nx][ny][nz]
    float2* cplx;          // [nx][ny][nz/2+1]
    // slab decomposition: the workgroup's plane is local plane `block` of nxl; its complex rows live in transpose-ready order
    // [dest rank q = ky / nyl][x local][ky % nyl][kz], so that the all-to-all moves one contiguous chunk per rank.  nyl = 0: plain.
    int nxl, nyl;
};

__device__ __forceinline__ size_t plane_cplx_index(const PlaneArgs& a, int x, int ky, int kz, int nzc) {
    if (a.nyl == 0) return ((size_t) x * a.ny + ky) * nzc + kz;
    const int q = ky / a.nyl, yl = ky - q * a.nyl;
    return (((size_t) q * a.nxl + x) * a.nyl + yl) * nzc + kz;
}

template <int CAP>
struct PlaneShared {
    float2 bufA[CAP];
    float2 bufB[CAP];
    float2 twYs[256];
    float2 twZs[256];
};

// THREADS = workgroup size, CAP = complex elements per LDS buffer (nz * (ny + 1) <= CAP), `block` = x plane of this workgroup
template <int THREADS, int CAP>
__device__ __forceinline__ void fft_plane_body(const PlaneArgs& a, const int block, PlaneShared<CAP>& sh) {
    float2* const bufA = sh.bufA;
    float2* const bufB = sh.bufB;
    float2* const twYs = sh.twYs;
    float2* const twZs = sh.twZs;
    const int ny = a.ny, nz = a.nz, nzc = nz / 2 + 1, S = ny + 1;
    const int x = block;
    for (int i = threadIdx.x; i < ny; i += THREADS) twYs[i] = a.twY[i];
    for (int i = threadIdx.x; i < nz; i += THREADS) twZs[i] = a.twZ[i];
    if (a.forward) {
        const float* in = a.real + (size_t) x * ny * nz;
        constexpr int MAXLD = (CAP + THREADS - 1) / THREADS;       // all reads of the plane in flight at once
        float ld[MAXLD];
#pragma unroll
        for (int it = 0; it < MAXLD; it++) {
            const int idx = threadIdx.x + it * THREADS;
            ld[it] = idx < ny * nz ? in[idx] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < MAXLD; it++) {
            const int idx = threadIdx.x + it * THREADS;
            if (idx < ny * nz) {
                const int y = idx / nz, z = idx % nz;
                bufA[z * S + y] = make_float2(ld[it], 0.f);
            }
        }
        __syncthreads();
        float2* r1 = fft_lines<THREADS>(a.planZ, bufA, bufB, ny, S, -1, twZs, 1);            // lines = y, elements = z
        float2* other = r1 == bufA ? bufB : bufA;
        float2* r2 = fft_lines<THREADS>(a.planY, r1, other, nzc, 1, -1, twYs, S);            // lines = kz < nzc, elements = y
        for (int idx = threadIdx.x; idx < ny * nzc; idx += THREADS) {
            const int ky = idx / nzc, kz = idx % nzc;
            a.cplx[plane_cplx_index(a, x, ky, kz, nzc)] = r2[kz * S + ky];
        }
    }
    else {
        constexpr int MAXLD = (CAP + THREADS - 1) / THREADS;
        float2 ld[MAXLD];
#pragma unroll
        for (int it = 0; it < MAXLD; it++) {
            const int idx = threadIdx.x + it * THREADS;
            ld[it] = idx < ny * nzc ? a.cplx[plane_cplx_index(a, x, idx / nzc, idx % nzc, nzc)] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < MAXLD; it++) {
            const int idx = threadIdx.x + it * THREADS;
            if (idx < ny * nzc) {
                const int ky = idx / nzc, kz = idx % nzc;
                bufA[kz * S + ky] = ld[it];
            }
        }
        __syncthreads();
        float2* r1 = fft_lines<THREADS>(a.planY, bufA, bufB, nzc, 1, +1, twYs, S);           // backward y on the half plane
        // Hermitian completion along z: element kz' = nz - kz is the conjugate of kz (for every y)
        for (int idx = threadIdx.x; idx < ny * (nz - nzc); idx += THREADS) {
            const int y = idx % ny, kz = nzc + idx / ny;
            const float2 v = r1[(nz - kz) * S + y];
            r1[kz * S + y] = make_float2(v.x, -v.y);
        }
        __syncthreads();
        float2* other = r1 == bufA ? bufB : bufA;
        float2* r2 = fft_lines<THREADS>(a.planZ, r1, other, ny, S, +1, twZs, 1);
        float* out = a.real + (size_t) x * ny * nz;
        for (int idx = threadIdx.x; idx < ny * nz; idx += THREADS) {
            const int y = idx / nz, z = idx % nz;
            out[idx] = r2[z * S + y].x;
        }
    }
}

__global__ __launch_bounds__(PLANE_THREADS) void fft_plane_kernel(PlaneArgs a) {
    __shared__ PlaneShared<PLANE_MAX> sh;
    fft_plane_body<PLANE_THREADS, PLANE_MAX>(a, blockIdx.x, sh);
}

// ------------------------------------------------------------------------------------------------
// Large planes (beyond PLANE_MAX, up to 192 x 192): the same fused (y, z) transform by ONE 1024-thread workgroup that takes a
// whole CU's LDS.  Two things make a 192 x 192 plane fit into 160 KB: the z transform of the real rows runs as a complex
// transform of HALF the length on (even, odd) pairs, followed by the usual split into the nz/2 + 1 Hermitian coefficients
// (backward: the inverse recombination first), and the passes work IN PLACE -- every thread reads the inputs of all its
// butterflies into registers, the workgroup meets at a barrier, then the outputs go to their (Stockham-permuted) places --
// so there is one plane buffer instead of two.  With the line passes the (y, z) half of a 192^3 transform is two launches
// and 2 x 171 MB of HBM traffic each way (36.8 + 36.8 us forward, 46.1 + 43.8 us backward on MI355X); here it is one launch
// that reads the plane once and writes it once.  Layout as in the small kernel: element (y, kz) at kz * (ny + 1) + y.
// ------------------------------------------------------------------------------------------------
#define BIGPLANE_THREADS 1024
#define BIGPLANE_CAP 19456    // complex elements of the plane buffer: (nz/2 + 1) * (ny + 1) <= CAP (152 KB + 4 KB of twiddles)
#define BIGPLANE_PTS 20       // points per thread and pass: BIGPLANE_THREADS * BIGPLANE_PTS >= BIGPLANE_CAP

#ifdef OMMHIP_EMU
#define SCHED_FENCE()
#define OPAQUE_S(x)
#else
#define OPAQUE_S(x) asm volatile("" : "+s"(x))
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

struct BigPlaneShared {
    float2 buf[BIGPLANE_CAP];
    float2 twYs[256];         // exp(-2 pi i k / ny)
    float2 twZh[128];         // exp(-2 pi i k / (nz/2)): the half-length transform's table
    float2 twZs[132];         // exp(-2 pi i k / nz), k <= nz/2: the split / recombination factors
};

// One radix-R pass over B lines of n points, in place: element e of line l at e*BP + l*LS.
template <int R, bool FIRST>
__device__ __forceinline__ void fft_pass_inplace(float2* __restrict__ buf, int n, int Ns, int B, int BP, int LS, int sign, const float2* __restrict__ tw) {
    constexpr int MAXB = (BIGPLANE_PTS + R - 1) / R;
    OPAQUE_S(B);                // or the (idx / B, idx % B) of every b would be computed once for all passes and kept in registers throughout
    const int butterflies = n / R, total = butterflies * B;
    const float fsign = (float) -sign, fs = (float) sign;
    const int twStep = n / (Ns * R);
    v2f o[MAXB][R];
#pragma unroll
    for (int b = 0; b < MAXB; b++) {
        const int idx = threadIdx.x + b * BIGPLANE_THREADS;
        if (idx < total) {
            const int j = idx / B, line = idx - j * B;
            const int k = FIRST ? 0 : j % Ns;
            v2f v[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float2 x = buf[(j + r * butterflies) * BP + line * LS];
                if (FIRST || r == 0) v[r] = mk2(x.x, x.y);
                else {
                    const float2 w = tw[k * r * twStep];
                    v[r] = cmulp(mk2(x.x, x.y), mk2(w.x, w.y * fsign));
                }
            }
            butterfly<R>(v, o[b], fs);
        }
        SCHED_FENCE();          // one butterfly's reads and arithmetic at a time: interleaved, the unrolled iterations need more registers than 16 waves per CU leave
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < MAXB; b++) {
        const int idx = threadIdx.x + b * BIGPLANE_THREADS;
        if (idx < total) {
            const int j = idx / B, line = idx - j * B;
            const int q = FIRST ? j : j / Ns, k = FIRST ? 0 : j - q * Ns;
            const int j0 = q * Ns * R + k;
#pragma unroll
            for (int p = 0; p < R; p++) buf[(j0 + p * Ns) * BP + line * LS] = make_float2(o[b][p].x, o[b][p].y);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void fft_lines_inplace(const FftPlan& plan, float2* buf, int B, int BP, int LS, int sign, const float2* tw) {
    int Ns = 1;
    for (int s = 0; s < plan.numRadices; s++) {
        const int R = (int) ((plan.radices >> (4 * s)) & 15ull);
        if (s == 0) {
            switch (R) {
                case 2: fft_pass_inplace<2, true>(buf, plan.n, Ns, B, BP, LS, sign, tw); break;
                case 3: fft_pass_inplace<3, true>(buf, plan.n, Ns, B, BP, LS, sign, tw); break;
                case 4: fft_pass_inplace<4, true>(buf, plan.n, Ns, B, BP, LS, sign, tw); break;
                case 5: fft_pass_inplace<5, true>(buf, plan.n, Ns, B, BP, LS, sign, tw); break;
                case 7: fft_pass_inplace<7, true>(buf, plan.n, Ns, B, BP, LS, sign, tw); break;
                default: fft_pass_inplace<8, true>(buf, plan.n, Ns, B, BP, LS, sign, tw); break;
            }
        }
        else {
            switch (R) {
                case 2: fft_pass_inplace<2, false>(buf, plan.n, Ns, B, BP, LS, sign, tw); break;
                case 3: fft_pass_inplace<3, false>(buf, plan.n, Ns, B, BP, LS, sign, tw); break;
                case 4: fft_pass_inplace<4, false>(buf, plan.n, Ns, B, BP, LS, sign, tw); break;
                case 5: fft_pass_inplace<5, false>(buf, plan.n, Ns, B, BP, LS, sign, tw); break;
                case 7: fft_pass_inplace<7, false>(buf, plan.n, Ns, B, BP, LS, sign, tw); break;
                default: fft_pass_inplace<8, false>(buf, plan.n, Ns, B, BP, LS, sign, tw); break;
            }
        }
        Ns *= R;
    }
}

__device__ __forceinline__ void fft_bigplane_body(const PlaneArgs& a, const int block, BigPlaneShared& sh) {
    float2* const buf = sh.buf;
    const int ny = a.ny, nz = a.nz, nh = nz / 2, nzc = nh + 1, S = ny + 1;
    const int x = block, t = threadIdx.x;
    constexpr int MAXLD = BIGPLANE_CAP / BIGPLANE_THREADS;
    for (int i = t; i < ny; i += BIGPLANE_THREADS) sh.twYs[i] = a.twY[i];
    for (int i = t; i < nh; i += BIGPLANE_THREADS) sh.twZh[i] = a.twZ[2 * i];
    for (int i = t; i <= nh; i += BIGPLANE_THREADS) sh.twZs[i] = a.twZ[i];
    const int pairs = nh / 2 + 1;                               // (k, nh - k), k = 0 .. nh/2
    if (a.forward) {
        // rows of real numbers as rows of nh complex numbers (even, odd): line y, element m at m*S + y
        const float2* in = (const float2*) (a.real + (size_t) x * ny * nz);
        float2 ld[MAXLD];
#pragma unroll
        for (int it = 0; it < MAXLD; it++) {
            const int idx = t + it * BIGPLANE_THREADS;
            ld[it] = idx < ny * nh ? in[idx] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < MAXLD; it++) {
            const int idx = t + it * BIGPLANE_THREADS;
            if (idx < ny * nh) { const int y = idx / nh, m = idx - y * nh; buf[m * S + y] = ld[it]; }
        }
        __syncthreads();
        fft_lines_inplace(a.planZh, buf, ny, S, 1, -1, sh.twZh);
        // split: Z[k] = Ev[k] + i Od[k] (transforms of the even and the odd samples), X[k] = Ev[k] + w^k Od[k], X[nh-k] = conj(Ev[k] - w^k Od[k])
        for (int idx = t; idx < pairs * ny; idx += BIGPLANE_THREADS) {
            const int k = idx / ny, y = idx - k * ny;
            if (k == 0) {
                const float2 z = buf[y];
                buf[y] = make_float2(z.x + z.y, 0.f);
                buf[nh * S + y] = make_float2(z.x - z.y, 0.f);
            }
            else {
                const float2 p = buf[k * S + y], q = buf[(nh - k) * S + y], w = sh.twZs[k];
                const float evx = 0.5f * (p.x + q.x), evy = 0.5f * (p.y - q.y);        // (p + conj q) / 2
                const float odx = 0.5f * (p.y + q.y), ody = -0.5f * (p.x - q.x);       // (p - conj q) / (2i)
                const float tx = w.x * odx - w.y * ody, ty = w.x * ody + w.y * odx;
                buf[k * S + y] = make_float2(evx + tx, evy + ty);
                if (nh - k != k) buf[(nh - k) * S + y] = make_float2(evx - tx, -(evy - ty));
            }
        }
        __syncthreads();
        fft_lines_inplace(a.planY, buf, nzc, 1, S, -1, sh.twYs);                       // lines = kz, elements = y
        for (int idx = t; idx < ny * nzc; idx += BIGPLANE_THREADS) {
            const int ky = idx / nzc, kz = idx - ky * nzc;
            a.cplx[plane_cplx_index(a, x, ky, kz, nzc)] = buf[kz * S + ky];
        }
    }
    else {
        float2 ld[MAXLD];
#pragma unroll
        for (int it = 0; it < MAXLD; it++) {
            const int idx = t + it * BIGPLANE_THREADS;
            ld[it] = idx < ny * nzc ? a.cplx[plane_cplx_index(a, x, idx / nzc, idx % nzc, nzc)] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < MAXLD; it++) {
            const int idx = t + it * BIGPLANE_THREADS;
            if (idx < ny * nzc) { const int ky = idx / nzc, kz = idx - ky * nzc; buf[kz * S + ky] = ld[it]; }
        }
        __syncthreads();
        fft_lines_inplace(a.planY, buf, nzc, 1, S, +1, sh.twYs);
        // recombination: Z[k] = E[k] + i O[k] with E[k] = X[k] + conj X[nh-k], O[k] = conj(w)^k (X[k] - conj X[nh-k]); Z[nh-k] = conj(E[k] - i O[k]).
        // The imaginary parts of X[0] and X[nh] would only add to the imaginary part of the real-space result: dropped, as taking .x does.
        for (int idx = t; idx < pairs * ny; idx += BIGPLANE_THREADS) {
            const int k = idx / ny, y = idx - k * ny;
            if (k == 0) {
                const float p = buf[y].x, q = buf[nh * S + y].x;
                buf[y] = make_float2(p + q, p - q);
            }
            else {
                const float2 p = buf[k * S + y], q = buf[(nh - k) * S + y], w = sh.twZs[k];
                const float ex = p.x + q.x, ey = p.y - q.y;                            // p + conj q
                const float dx = p.x - q.x, dy = p.y + q.y;                            // p - conj q
                const float ox = w.x * dx + w.y * dy, oy = w.x * dy - w.y * dx;        // conj(w) * (p - conj q)
                buf[k * S + y] = make_float2(ex - oy, ey + ox);                        // E + i O
                if (nh - k != k) buf[(nh - k) * S + y] = make_float2(ex + oy, -(ey - ox));   // conj(E - i O)
            }
        }
        __syncthreads();
        fft_lines_inplace(a.planZh, buf, ny, S, 1, +1, sh.twZh);
        float2* out = (float2*) (a.real + (size_t) x * ny * nz);
        for (int idx = t; idx < ny * nh; idx += BIGPLANE_THREADS) {
            const int y = idx / nh, m = idx - y * nh;
            out[idx] = buf[m * S + y];
        }
    }
}

__global__ __launch_bounds__(BIGPLANE_THREADS) void fft_bigplane_kernel(PlaneArgs a) {
    __shared__ BigPlaneShared sh;
    fft_bigplane_body(a, blockIdx.x, sh);
}

FftPlan make_plan(int n) {
    FftPlan p;
    p.n = n; p.numRadices = 0; p.radices = 0;
    int m = n;
    const int cand[6] = {8, 4, 2, 3, 5, 7};
    // power-of-two part first (as 8s, then a 4 or 2), then odd primes
    for (int c = 0; c < 6; c++)
        while (m % cand[c] == 0 && p.numRadices < FFT_MAX_RADICES) { p.radices |= (unsigned long long) cand[c] << (4 * p.numRadices++); m /= cand[c]; }
    if (m != 1) p.n = -1;   // unsupported size
    return p;
}

int lines_per_group(int n) {
    int b = FFT_MAX_LDS / n - 1;     // (b+1)*n elements of LDS per buffer
    if (b > 16) b = 16;
    if (b < 1) b = 1;
    int p = 1;                       // a power of two: line / butterfly indices by shift and mask, aligned segments in memory
    while (2 * p <= b) p *= 2;
    return p;
}

// which fused plane kernel takes the (y, z) half of this grid: 1 the small one (two LDS buffers), 2 the large one (one workgroup
// per CU, in place), 0 neither (line passes)
int plane_kernel_kind(const ommhip_pme* pme, int planes) {
    const int ny = pme->ny, nz = pme->nz;
    if (pme->fft_mode == 1 || ny > 256 || nz > 256) return 0;
    if (nz * (ny + 1) <= PLANE_MAX) return 1;
    static const bool noBig = getenv("OPENMM_HIP_NO_BIG_PLANE") != nullptr;            // A/B knob
    if (noBig || nz % 2 != 0 || (nz / 2 + 1) * (ny + 1) > BIGPLANE_CAP || make_plan(nz / 2).n != nz / 2) return 0;
    // a workgroup of the large kernel takes ~50 us whatever the number of planes: a thin slab (8 ranks of a 192^3 grid: 24 planes) is
    // served faster by the two line-pass launches (measured with the ranks serialised on one GPU: 0.59-0.66 against 0.62-0.70 ms per step)
    if (planes < 64 && pme->fft_mode != 2) return 0;
    return 2;
}

PlaneArgs make_plane_args(const ommhip_pme* pme, bool forward) {
    PlaneArgs p;
    p.planY = make_plan(pme->ny); p.planZ = make_plan(pme->nz); p.planZh = make_plan(pme->nz / 2); p.ny = pme->ny; p.nz = pme->nz; p.forward = forward ? 1 : 0;
    p.twY = (const float2*) pme->twiddle_y; p.twZ = (const float2*) pme->twiddle_z;
    p.real = (float*) pme->grid_real; p.cplx = (float2*) pme->grid_complex;
    p.nxl = pme->nx; p.nyl = 0;
    return p;
}

// x pass of the reciprocal-space chain: forward, multiply by the influence function (+ energy), backward
FftArgs make_xconv_args(const ommhip_pme* pme, double* energy_buffer_d, int energy_slots, int include_energy) {
    const int nx = pme->nx, ny = pme->ny, nz = pme->nz, nzc = nz / 2 + 1;
    float2* cgrid = (float2*) pme->grid_complex;
    FftArgs f;
    f.diag = nullptr;
    f.remapIn = f.remapOut = 0; f.remapNxl = f.remapNyl = 1;
    f.nzFull = nz;
    f.plan = make_plan(nx); f.B = lines_per_group(nx); f.numOuter = ny; f.numInner = nzc;
    f.inOuterStride = nzc; f.inInnerStride = 1; f.inElemStride = (long long) ny * nzc;
    f.outOuterStride = nzc; f.outInnerStride = 1; f.outElemStride = (long long) ny * nzc;
    f.mode = 3; f.sign = -1; f.twiddle = (const float2*) pme->twiddle_x; f.in = cgrid; f.out = cgrid;
    f.eterm = (const float*) pme->eterm; f.energyBuffer = include_energy ? energy_buffer_d : nullptr; f.energySlots = energy_slots;
    return f;
}

// The (y,z) half of the 3-D transform: fused plane kernel when a plane fits in LDS, two line passes otherwise.
void launch_yz(const ommhip_pme* pme, bool forward, hipStream_t st) {
    const int nx = pme->nx, ny = pme->ny, nz = pme->nz, nzc = nz / 2 + 1;
    float2* cgrid = (float2*) pme->grid_complex;
    if (const int kind = plane_kernel_kind(pme, nx)) {
        PlaneArgs p = make_plane_args(pme, forward);
        p.cplx = cgrid;
        if (kind == 1) hipLaunchKernelGGL(fft_plane_kernel, dim3(nx), dim3(PLANE_THREADS), 0, st, p);
        else hipLaunchKernelGGL(fft_bigplane_kernel, dim3(nx), dim3(BIGPLANE_THREADS), 0, st, p);
        return;
    }
    FftArgs f;
    f.diag = nullptr;
    f.remapIn = f.remapOut = 0; f.remapNxl = f.remapNyl = 1;
    f.eterm = nullptr; f.energyBuffer = nullptr; f.energySlots = 1; f.nzFull = nz;
    auto zpass = [&]() {
        f.plan = make_plan(nz); f.B = lines_per_group(nz); f.numOuter = 1; f.numInner = nx * ny;
        f.inOuterStride = 0; f.outOuterStride = 0; f.inElemStride = 1; f.outElemStride = 1;
        f.twiddle = (const float2*) pme->twiddle_z;
        if (forward) { f.inInnerStride = nz; f.outInnerStride = nzc; f.mode = 1; f.sign = -1; f.in = pme->grid_real; f.out = cgrid; }
        else { f.inInnerStride = nzc; f.outInnerStride = nz; f.mode = 2; f.sign = +1; f.in = cgrid; f.out = pme->grid_real; }
        launch_fft(f, st);
    };
    auto ypass = [&]() {
        f.plan = make_plan(ny); f.B = lines_per_group(ny); f.numOuter = nx; f.numInner = nzc;
        f.inOuterStride = (long long) ny * nzc; f.inInnerStride = 1; f.inElemStride = nzc;
        f.outOuterStride = f.inOuterStride; f.outInnerStride = 1; f.outElemStride = nzc;
        f.mode = 0; f.sign = forward ? -1 : +1; f.twiddle = (const float2*) pme->twiddle_y; f.in = cgrid; f.out = cgrid;
        launch_fft(f, st);
    };
    if (forward) { zpass(); ypass(); }
    else { ypass(); zpass(); }
}

}  // namespace

extern "C" int ommhip_fft_supported_size(int n) {
    if (n < 2 || 2 * n > FFT_MAX_LDS) return 0;
    return make_plan(n).n == n ? 1 : 0;
}

extern "C" int ommhip_pme_build_eterm(const ommhip_pme* pme, void* stream) {
    EtermArgs a;
    a.nx = pme->nx; a.ny = pme->ny; a.nz = pme->nz; a.nzc = pme->nz / 2 + 1;
    const double* b = pme->box;
    const double det = b[0] * b[2] * b[5];
    // ReferencePME.cpp:196-204
    a.r00 = b[2] * b[5] / det;
    a.r10 = -b[1] * b[5] / det; a.r11 = b[0] * b[5] / det;
    a.r20 = (b[1] * b[4] - b[2] * b[3]) / det; a.r21 = -b[0] * b[4] / det; a.r22 = b[0] * b[2] / det;
    a.alpha = pme->alpha; a.volume = det;
    a.modX = pme->moduli_x; a.modY = pme->moduli_y; a.modZ = pme->moduli_z;
    a.eterm = (float*) pme->eterm;
    a.y0 = 0; a.nyl = a.ny; a.dispersion = pme->dispersion;
    if (pme->dd_ranks > 1) { a.nyl = a.ny / pme->dd_ranks; a.y0 = pme->dd_rank * a.nyl; }
    const size_t total = (size_t) a.nx * a.nyl * a.nzc;
    hipLaunchKernelGGL(pme_build_eterm, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, (hipStream_t) stream, a);
    return (int) hipGetLastError();
}

static PmeArgs make_pme_args(const ommhip_pme* pme, const void* posq_d, int padded_atoms, long long* force_d,
                             double* energy_buffer_d, int energy_slots, int include_energy) {
    const int nx = pme->nx, ny = pme->ny, nz = pme->nz;
    PmeArgs pa;
    pa.paddedAtoms = padded_atoms; pa.nx = nx; pa.ny = ny; pa.nz = nz;
    static const int spreadDebug = getenv("OPENMM_HIP_DEBUG_SPREAD") != nullptr ? atoi(getenv("OPENMM_HIP_DEBUG_SPREAD")) : 0;   // profiling only
    pa.debug = spreadDebug;
    const double* b = pme->box;
    const double det = b[0] * b[2] * b[5];
    pa.recip.r00 = (float) (b[2] * b[5] / det);
    pa.recip.r10 = (float) (-b[1] * b[5] / det); pa.recip.r11 = (float) (b[0] * b[5] / det);
    pa.recip.r20 = (float) ((b[1] * b[4] - b[2] * b[3]) / det); pa.recip.r21 = (float) (-b[0] * b[4] / det); pa.recip.r22 = (float) (b[0] * b[2] / det);
    pa.posq = (const float4*) posq_d; pa.grid = (float*) pme->grid_real; pa.force = force_d;
    pa.exclStart = pme->excl_start; pa.exclAtoms = pme->excl_atoms; pa.atomOfSlot = pme->atom_of_slot;
    pa.pos = (const double4*) pme->pos; pa.charge = pme->charge;
    pa.boxd.ax = b[0]; pa.boxd.bx = b[1]; pa.boxd.by = b[2]; pa.boxd.cx = b[3]; pa.boxd.cy = b[4]; pa.boxd.cz = b[5];
    pa.alpha = pme->alpha; pa.exclPeriodic = pme->excl_periodic;
    pa.includeEnergy = include_energy; pa.energySlots = energy_slots; pa.energyBuffer = energy_buffer_d;
    pa.planeLo = 0; pa.planeCount = nx; pa.haloLo = 0; pa.gridPlanes = nx; pa.ownSlot0 = 0; pa.ownSlot1 = padded_atoms;
    pa.ddError = nullptr; pa.blockCenter = nullptr; pa.blockHalf = nullptr;
    pa.detScale = pme->deterministic && pme->max_charge > 0 ? (float) (2147483648.0 / (64.0 * pme->max_charge)) : 0.f;
    pa.xcdBlocks = 0;
    pa.numActive = 0;
    for (int r = 0; r < 4; r++) { pa.activeBegin[r] = pa.activeEnd[r] = 0; pa.activeGroup0[r] = 0; }
    pa.activeGroup0[4] = 0;
    return pa;
}

extern "C" int ommhip_pme_reciprocal(const ommhip_pme* pme, const void* posq_d, int padded_atoms, long long* force_d,
                                     double* energy_buffer_d, int energy_slots, int include_energy, void* stream) {
    hipStream_t st = (hipStream_t) stream;
    const int nx = pme->nx, ny = pme->ny, nz = pme->nz, nzc = nz / 2 + 1;
    PmeArgs pa = make_pme_args(pme, posq_d, padded_atoms, force_d, energy_buffer_d, energy_slots, include_energy);
    float2* cgrid = (float2*) pme->grid_complex;

    const int spreadBlocks = (padded_atoms * 8 + 255) / 256;
    if (pme->phases != OMMHIP_PME_AFTER_SPREAD && pme->phases != OMMHIP_PME_INTERPOLATE_ONLY) {
        ommhip_profile_begin(OMMHIP_TIMER_PME_SPREAD, stream);
        if (!launch_tile_spread<false>(pme, pa, pme->block_center, pme->block_half, st)) {      // tiles write every cell: no cleared grid needed
            if (!pme->grid_precleared) hipMemsetAsync(pa.grid, 0, sizeof(float) * (size_t) nx * ny * nz, st);
            if (pme->spread_mode == 1)
                hipLaunchKernelGGL(pme_spread, dim3(spreadBlocks), dim3(256), 0, st, pa);             // direct global atomics (reference variant)
            else
                launch_spread<false>(pa, padded_atoms, st);
        }
        if (pa.detScale > 0.f) {
            const size_t n = (size_t) nx * ny * nz;
            hipLaunchKernelGGL(pme_fixed_to_float, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, pa.grid, n, 1.f / pa.detScale);
        }
        ommhip_profile_end(OMMHIP_TIMER_PME_SPREAD, stream);
    }
    if (pme->phases == OMMHIP_PME_SPREAD_ONLY) return (int) hipGetLastError();
    if (pme->phases != OMMHIP_PME_INTERPOLATE_ONLY) {
    ommhip_profile_begin(OMMHIP_TIMER_PME_FFT, stream);

    FftArgs f;
    f.diag = nullptr;
    f.remapIn = f.remapOut = 0; f.remapNxl = f.remapNyl = 1;
    f.eterm = nullptr; f.energyBuffer = nullptr; f.energySlots = 1; f.nzFull = nz;
    // ---- forward z (r2c) and y
    launch_yz(pme, true, st);
    // ---- x: forward, multiply by the influence function (+ energy), backward -- one pass over HBM
    f.plan = make_plan(nx); f.B = lines_per_group(nx); f.numOuter = ny; f.numInner = nzc;
    f.inOuterStride = nzc; f.inInnerStride = 1; f.inElemStride = (long long) ny * nzc;
    f.outOuterStride = nzc; f.outInnerStride = 1; f.outElemStride = (long long) ny * nzc;
    f.mode = 3; f.sign = -1; f.twiddle = (const float2*) pme->twiddle_x; f.in = cgrid; f.out = cgrid;
    f.eterm = (const float*) pme->eterm; f.energyBuffer = include_energy ? energy_buffer_d : nullptr; f.energySlots = energy_slots;
    launch_fft(f, st);
    // ---- backward y and z (c2r)
    launch_yz(pme, false, st);

    ommhip_profile_end(OMMHIP_TIMER_PME_FFT, stream);
    }
    ommhip_profile_begin(OMMHIP_TIMER_PME_INTERPOLATE, stream);
    {
        static const bool noXcd = getenv("OPENMM_HIP_XCD_INTERPOLATE") == nullptr;          // opt-in: measured neutral at 1M atoms, -1.7 % at 92 k (docs/EXPERIMENTS.md)
        launch_interpolate<false>(pa, padded_atoms, !noXcd, st);
    }
    ommhip_profile_end(OMMHIP_TIMER_PME_INTERPOLATE, stream);
    return (int) hipGetLastError();
}

extern "C" int ommhip_pme_convolve(const ommhip_pme* pme, void* stream) {
    hipStream_t st = (hipStream_t) stream;
    const int nx = pme->nx, ny = pme->ny, nz = pme->nz, nzc = nz / 2 + 1;
    float2* cgrid = (float2*) pme->grid_complex;
    FftArgs f;
    f.diag = nullptr;
    f.remapIn = f.remapOut = 0; f.remapNxl = f.remapNyl = 1;
    f.eterm = nullptr; f.energyBuffer = nullptr; f.energySlots = 1; f.nzFull = nz;
    launch_yz(pme, true, st);
    f.plan = make_plan(nx); f.B = lines_per_group(nx); f.numOuter = ny; f.numInner = nzc;
    f.inOuterStride = nzc; f.inInnerStride = 1; f.inElemStride = (long long) ny * nzc;
    f.outOuterStride = nzc; f.outInnerStride = 1; f.outElemStride = (long long) ny * nzc;
    f.mode = 3; f.sign = -1; f.twiddle = (const float2*) pme->twiddle_x; f.in = cgrid; f.out = cgrid;
    f.eterm = (const float*) pme->eterm;
    launch_fft(f, st);
    launch_yz(pme, false, st);
    return (int) hipGetLastError();
}

// The same for TWO grids of one shape in the same three launches (blockIdx.y picks the grid): the AMOEBA solver convolves the potentials of
// two sets of induced dipoles per iteration, and each of these launches fills a fraction of the chip (openmm_hip_amoeba.h).
namespace {
__global__ __launch_bounds__(PLANE_THREADS) void fft_plane_kernel2(PlaneArgs a, PlaneArgs b) {
    __shared__ PlaneShared<PLANE_MAX> sh;
    fft_plane_body<PLANE_THREADS, PLANE_MAX>(blockIdx.y == 0 ? a : b, blockIdx.x, sh);
}
__global__ __launch_bounds__(FFT_THREADS) void fft_kernel2_xconv(FftArgs a, FftArgs b) {
    __shared__ FftShared sh;
    const FftArgs& f = blockIdx.y == 0 ? a : b;
    fft_body_mode<FFT_THREADS, 3, false, false>(f, blockIdx.x, sh, gridDim.x, f.numOuter * ((f.numInner + f.B - 1) / f.B));
}
}  // namespace

extern "C" int ommhip_pme_convolve2(const ommhip_pme* pme, const ommhip_pme* pme2, void* stream) {
    hipStream_t st = (hipStream_t) stream;
    if (pme->nx != pme2->nx || pme->ny != pme2->ny || pme->nz != pme2->nz || plane_kernel_kind(pme, pme->nx) != 1 || plane_kernel_kind(pme2, pme2->nx) != 1) return -1;
    const int nx = pme->nx, ny = pme->ny, nz = pme->nz, nzc = nz / 2 + 1;
    FftArgs f[2];
    PlaneArgs fwd[2], bwd[2];
    const ommhip_pme* both[2] = {pme, pme2};
    for (int g = 0; g < 2; g++) {
        float2* cgrid = (float2*) both[g]->grid_complex;
        fwd[g] = make_plane_args(both[g], true); fwd[g].cplx = cgrid;
        bwd[g] = make_plane_args(both[g], false); bwd[g].cplx = cgrid;
        FftArgs& x = f[g];
        x.diag = nullptr;
        x.remapIn = x.remapOut = 0; x.remapNxl = x.remapNyl = 1;
        x.energyBuffer = nullptr; x.energySlots = 1; x.nzFull = nz;
        x.plan = make_plan(nx); x.B = lines_per_group(nx); x.numOuter = ny; x.numInner = nzc;
        x.inOuterStride = nzc; x.inInnerStride = 1; x.inElemStride = (long long) ny * nzc;
        x.outOuterStride = nzc; x.outInnerStride = 1; x.outElemStride = (long long) ny * nzc;
        x.mode = 3; x.sign = -1; x.twiddle = (const float2*) both[g]->twiddle_x; x.in = cgrid; x.out = cgrid;
        x.eterm = (const float*) both[g]->eterm;
    }
    hipLaunchKernelGGL(fft_plane_kernel2, dim3(nx, 2), dim3(PLANE_THREADS), 0, st, fwd[0], fwd[1]);
    hipLaunchKernelGGL(fft_kernel2_xconv, dim3(fft_grid(f[0].numOuter * ((f[0].numInner + f[0].B - 1) / f[0].B)), 2), dim3(FFT_THREADS), 0, st, f[0], f[1]);
    hipLaunchKernelGGL(fft_plane_kernel2, dim3(nx, 2), dim3(PLANE_THREADS), 0, st, bwd[0], bwd[1]);
    return (int) hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// One rank of a slab-decomposed reciprocal-space evaluation.  Same kernels as above; what changes is which
// planes / rows a launch covers and where the complex data sits, so that each of the two transposes is ONE all-to-all of
// contiguous chunks:
//   A = grid_complex   [dest rank q][x local][y local of q][kz]   written by the forward plane / y transforms
//   B = grid_complex2  [x (all)][y local][kz]                      = A's chunks as they arrive, source-rank major
// ------------------------------------------------------------------------------------------------
namespace {
void launch_yz_dd(const ommhip_pme* pme, bool forward, float* realOwn, hipStream_t st) {
    const int R = pme->dd_ranks, nx = pme->nx, ny = pme->ny, nz = pme->nz, nzc = nz / 2 + 1, nxl = nx / R, nyl = ny / R;
    float2* A = (float2*) pme->grid_complex;
    float2* B = (float2*) pme->grid_complex2;
    if (const int kind = plane_kernel_kind(pme, nxl)) {
        PlaneArgs p = make_plane_args(pme, forward);
        p.real = realOwn; p.cplx = A; p.nxl = nxl; p.nyl = nyl;
        if (kind == 1) hipLaunchKernelGGL(fft_plane_kernel, dim3(nxl), dim3(PLANE_THREADS), 0, st, p);
        else hipLaunchKernelGGL(fft_bigplane_kernel, dim3(nxl), dim3(BIGPLANE_THREADS), 0, st, p);
        return;
    }
    FftArgs f;
    f.diag = nullptr;
    f.remapIn = f.remapOut = 0; f.remapNxl = nxl; f.remapNyl = nyl;
    f.eterm = nullptr; f.energyBuffer = nullptr; f.energySlots = 1; f.nzFull = nz;
    auto zpass = [&]() {      // real planes <-> B in plain [x local][y][kz] order
        f.plan = make_plan(nz); f.B = lines_per_group(nz); f.numOuter = 1; f.numInner = nxl * ny;
        f.inOuterStride = 0; f.outOuterStride = 0; f.inElemStride = 1; f.outElemStride = 1; f.remapIn = f.remapOut = 0;
        f.twiddle = (const float2*) pme->twiddle_z;
        if (forward) { f.inInnerStride = nz; f.outInnerStride = nzc; f.mode = 1; f.sign = -1; f.in = realOwn; f.out = B; }
        else { f.inInnerStride = nzc; f.outInnerStride = nz; f.mode = 2; f.sign = +1; f.in = B; f.out = realOwn; }
        launch_fft(f, st);
    };
    auto ypass = [&]() {      // B (plain) <-> A (transpose-ready)
        f.plan = make_plan(ny); f.B = lines_per_group(ny); f.numOuter = nxl; f.numInner = nzc;
        f.inOuterStride = (long long) ny * nzc; f.inInnerStride = 1; f.inElemStride = nzc;
        f.outOuterStride = f.inOuterStride; f.outInnerStride = 1; f.outElemStride = nzc;
        f.mode = 0; f.sign = forward ? -1 : +1; f.twiddle = (const float2*) pme->twiddle_y;
        if (forward) { f.in = B; f.out = A; f.remapIn = 0; f.remapOut = 1; }
        else { f.in = A; f.out = B; f.remapIn = 1; f.remapOut = 0; }
        launch_fft(f, st);
    };
    if (forward) { zpass(); ypass(); }
    else { ypass(); zpass(); }
}
}  // namespace

extern "C" int ommhip_pme_reciprocal_dd(const ommhip_pme* pme, const void* posq_d, int padded_atoms, int own_slot0, int own_slot1, const void* block_center_d,
                                        const void* block_half_d, long long* force_d, double* energy_buffer_d, int energy_slots, int include_energy, void* stream) {
    hipStream_t st = (hipStream_t) stream;
    const int R = pme->dd_ranks, rank = pme->dd_rank, D = pme->dd_halo;
    const int nx = pme->nx, ny = pme->ny, nz = pme->nz, nzc = nz / 2 + 1;
    if (R < 1 || nx % R != 0 || ny % R != 0 || pme->comm == nullptr || pme->grid_complex2 == nullptr || pme->dd_error == nullptr) return 1;
    const int nxl = nx / R, nyl = ny / R;
    if (D < 0 || D + 4 > nxl) return 1;
    ommhip_comm* comm = (ommhip_comm*) pme->comm;
    PmeArgs pa = make_pme_args(pme, posq_d, padded_atoms, force_d, energy_buffer_d, energy_slots, include_energy);
    pa.planeLo = rank * nxl; pa.planeCount = nxl; pa.haloLo = D; pa.gridPlanes = nxl + 2 * D + 4;
    pa.ownSlot0 = own_slot0; pa.ownSlot1 = own_slot1; pa.ddError = pme->dd_error;
    if (pme->dd_num_active_ranges > 0 && pme->dd_num_active_ranges <= 4) {
        pa.numActive = pme->dd_num_active_ranges;
        for (int r = 0; r < pa.numActive; r++) { pa.activeBegin[r] = pme->dd_active_range[2 * r]; pa.activeEnd[r] = pme->dd_active_range[2 * r + 1]; }
    }
    pa.blockCenter = (const float4*) block_center_d; pa.blockHalf = (const float4*) block_half_d;
    float* real = (float*) pme->grid_real;
    float* realOwn = real + (size_t) D * ny * nz;
    const size_t planeBytes = sizeof(float) * (size_t) ny * nz;

    if (pme->phases != OMMHIP_PME_AFTER_SPREAD && pme->phases != OMMHIP_PME_INTERPOLATE_ONLY) {
        if (pa.blockCenter == nullptr || pa.blockHalf == nullptr) return 1;
        ommhip_profile_begin(OMMHIP_TIMER_PME_SPREAD, stream);
        if (!launch_tile_spread<true>(pme, pa, block_center_d, block_half_d, st)) {
            if (!pme->grid_precleared) hipMemsetAsync(real, 0, planeBytes * (size_t) pa.gridPlanes, st);
            launch_spread<true>(pa, padded_atoms, st);
        }
        ommhip_profile_end(OMMHIP_TIMER_PME_SPREAD, stream);
    }
    if (pme->phases == OMMHIP_PME_SPREAD_ONLY) return (int) hipGetLastError();
    if (pme->phases != OMMHIP_PME_INTERPOLATE_ONLY) {
        ommhip_profile_begin(OMMHIP_TIMER_PME_FFT, stream);
        launch_yz_dd(pme, true, realOwn, st);
        const size_t pairBytes = sizeof(float2) * (size_t) nxl * nyl * nzc;
        int rc = ommhip_comm_all_to_all(comm, pme->grid_complex, pme->grid_complex2, pairBytes, stream);
        if (rc != 0) return rc;
        // x transform * influence function * inverse x transform on this rank's rows, in place in B = [x][y local][kz]
        FftArgs f;
        f.diag = nullptr;
        f.remapIn = f.remapOut = 0; f.remapNxl = f.remapNyl = 1;
        f.nzFull = nz;
        f.plan = make_plan(nx); f.B = lines_per_group(nx); f.numOuter = nyl; f.numInner = nzc;
        f.inOuterStride = nzc; f.inInnerStride = 1; f.inElemStride = (long long) nyl * nzc;
        f.outOuterStride = nzc; f.outInnerStride = 1; f.outElemStride = (long long) nyl * nzc;
        f.mode = 3; f.sign = -1; f.twiddle = (const float2*) pme->twiddle_x; f.in = pme->grid_complex2; f.out = pme->grid_complex2;
        f.eterm = (const float*) pme->eterm; f.energyBuffer = include_energy ? energy_buffer_d : nullptr; f.energySlots = energy_slots;
        launch_fft(f, st);
        rc = ommhip_comm_all_to_all(comm, pme->grid_complex2, pme->grid_complex, pairBytes, stream);
        if (rc != 0) return rc;
        launch_yz_dd(pme, false, realOwn, st);
        // potential planes the neighbours' atoms (and mine, beyond my slab) interpolate from
        rc = ommhip_comm_ring_exchange(comm, realOwn, realOwn + (size_t) nxl * ny * nz, planeBytes * (size_t) (D + 4),
                                       realOwn + (size_t) (nxl - D) * ny * nz, real, planeBytes * (size_t) D, stream);
        if (rc != 0) return rc;
        ommhip_profile_end(OMMHIP_TIMER_PME_FFT, stream);
    }
    ommhip_profile_begin(OMMHIP_TIMER_PME_INTERPOLATE, stream);
    const int ownSlots = own_slot1 - own_slot0;
    if (ownSlots > 0) launch_interpolate<true>(pa, ownSlots, false, st);
    ommhip_profile_end(OMMHIP_TIMER_PME_INTERPOLATE, stream);
    return (int) hipGetLastError();
}

// Raw 3-D transforms for unit tests (pattern of platforms/cuda/tests/TestCudaFFT3D.cpp:52-108).
extern "C" int ommhip_fft3d_r2c_c2r(const ommhip_pme* pme, int forward, void* stream) {
    hipStream_t st = (hipStream_t) stream;
    const int nx = pme->nx, ny = pme->ny, nz = pme->nz, nzc = nz / 2 + 1;
    float2* cgrid = (float2*) pme->grid_complex;
    FftArgs f;
    f.diag = nullptr;
    f.remapIn = f.remapOut = 0; f.remapNxl = f.remapNyl = 1;
    f.eterm = nullptr; f.energyBuffer = nullptr; f.energySlots = 1; f.nzFull = nz;
    auto zpass = [&](bool fwd) {
        f.plan = make_plan(nz); f.B = lines_per_group(nz); f.numOuter = 1; f.numInner = nx * ny;
        f.inOuterStride = 0; f.outOuterStride = 0; f.inElemStride = 1; f.outElemStride = 1;
        f.twiddle = (const float2*) pme->twiddle_z;
        if (fwd) { f.inInnerStride = nz; f.outInnerStride = nzc; f.mode = 1; f.sign = -1; f.in = pme->grid_real; f.out = cgrid; }
        else { f.inInnerStride = nzc; f.outInnerStride = nz; f.mode = 2; f.sign = +1; f.in = cgrid; f.out = pme->grid_real; }
        launch_fft(f, st);
    };
    auto ypass = [&](int sign) {
        f.plan = make_plan(ny); f.B = lines_per_group(ny); f.numOuter = nx; f.numInner = nzc;
        f.inOuterStride = (long long) ny * nzc; f.inInnerStride = 1; f.inElemStride = nzc;
        f.outOuterStride = f.inOuterStride; f.outInnerStride = 1; f.outElemStride = nzc;
        f.mode = 0; f.sign = sign; f.twiddle = (const float2*) pme->twiddle_y; f.in = cgrid; f.out = cgrid;
        launch_fft(f, st);
    };
    auto xpass = [&](int sign) {
        f.plan = make_plan(nx); f.B = lines_per_group(nx); f.numOuter = ny; f.numInner = nzc;
        f.inOuterStride = nzc; f.inInnerStride = 1; f.inElemStride = (long long) ny * nzc;
        f.outOuterStride = nzc; f.outInnerStride = 1; f.outElemStride = (long long) ny * nzc;
        f.mode = 0; f.sign = sign; f.twiddle = (const float2*) pme->twiddle_x; f.in = cgrid; f.out = cgrid;
        launch_fft(f, st);
    };
    (void) zpass; (void) ypass;
    if (forward) { launch_yz(pme, true, st); xpass(-1); }
    else { xpass(+1); launch_yz(pme, false, st); }
    return (int) hipGetLastError();
}

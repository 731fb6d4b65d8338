// Direct-space NonbondedForce for MI355X (gfx950): the pair kernel (the neighbour list is built by neighbor.hip).
//
// Replaces (behaviourally) the Reference path
//   platforms/reference/src/ReferenceKernels.cpp:967-1014   (ReferenceCalcNonbondedForceKernel::execute)
//   platforms/reference/src/SimTKReference/ReferenceNeighborList.cpp:221-259 (neighbour list)
//   platforms/reference/src/SimTKReference/ReferenceLJCoulombIxn.cpp:379-457 (Ewald direct sum)
//   platforms/reference/src/SimTKReference/ReferenceLJCoulombIxn.cpp:543-639 (cutoff / no-cutoff pair ixn)
//
// Formulation (designed for wave64, not a warp-32 tiling):
//   * atoms are spatially sorted into blocks of 32 ("i-blocks");
//   * the neighbour list is a set of ROWS: one i-block X and 64 individually selected j-atoms
//     (slot index + a 32-bit mask saying which of X's 32 atoms interact with that j);
//   * one wavefront processes a row with lane = j-atom.  The 32 i-atoms are wave-uniform, so
//     their position/charge/LJ data arrive through *scalar* loads (s_load_dwordx16) and cost no
//     vector registers, no LDS traffic and no cross-lane shuffles in the inner loop;
//   * forces on j accumulate in 3 VGPRs, forces on the 32 i-atoms in 96 VGPRs that are
//     transpose-reduced across the wave once per chunk of rows;
//   * results are added to the 64-bit fixed-point force buffer with atomics.
//   Diagonal blocks, exclusions and padding atoms are all expressed through the row masks, so
//   there is a single code path; rows whose 64 masks are all-ones take a mask-free inner loop.
#include "common.h"
#include "erfc_coeffs.h"
#include <cmath>
#include <cstdlib>
#include "../../../include/openmm_hip_kernels.h"

using namespace omm;

namespace {

// ================================================================================================
// Pair kernel
// ================================================================================================
#define OMM_EWPOLY_DEGREE 11
struct NbArgs {
    int paddedAtoms, maxChunks, energySlots, debugFlags;
    int xcdAware;             // XCD-aware placement of the work units (ChunkSchedule)
    int ljHeadSplit;          // use the split loops for i-blocks whose atoms from OMM_LJ_HEAD on have epsilon = 0
    int ownSlot0, ownSlot1;   // domain decomposition: forces on j atoms outside [ownSlot0, ownSlot1) are dropped (their owner evaluates the pair too)
    int keepSlot0, keepSlot1; // ... unless they lie in [keepSlot0, keepSlot1): half-shell evaluation, the lower neighbour's section -- force and energy of the pair are this rank's to compute
    float cutoff2, alpha, krf, crf, switchDist, invSwitchWidth;
    // Cutoff edge (posqLo != null): a pair is "inside" for the packed loops when r^2 < cutoff2 = rc^2 (1 + d), which includes every
    // pair whose float separation could be the rounding of a double-precision separation inside the cutoff; pairs in the band
    // [cutoff2Lo, cutoff2) = rc^2 (1 -+ d) raise a wave-uniform flag, and the rare path behind it (fix_edge_pairs) re-decides each of
    // them from the hi + lo coordinates in double -- the decision the Reference platform takes -- and takes the pair out again when
    // it lies outside.  Without posqLo: cutoff2 = cutoff2Lo = rc^2 and the float separation decides.
    float cutoff2Lo;
    double cutoff2d;
    BoxD boxd;
    const float4* posqLo;     // low parts of the block-relative coordinates (ommhip_neighbor_list::posq_rel_lo), or null
    float ewPoly[OMM_EWPOLY_DEGREE + 1];   // METHOD & 8: alpha^3 g((u + 1) zmax / 2) as a polynomial in u, highest power first (see ewald_poly_for)
    float ewPolyScale;                     // u = r^2 * ewPolyScale - 1
    float dispAlpha2, invCut6, dispShift;   // LJPME (METHOD & 4): alpha_d^2, 1/rc^6, (1 - exp(-x)(1 + x + x^2/2)) / rc^6 at x = (alpha_d rc)^2
    Box box;
    const float4* posq;       // block-relative coordinates + charge (ommhip_neighbor_list::posq_rel): position minus blockCenter of its block
    const float2* sigEps;
    const int* state;
    const int2* chunkInfo;
    const int* rowJ;
    const unsigned* rowMask;
    // the pruned list (ommhip_neighbor_list::chunk_info_inner), walked instead of the rows above unless state[ST_NO_PRUNE] is set; null: none
    const int2* chunkInfoInner;
    const int* rowJInner;
    const unsigned* rowMaskInner;
    const float4* blockCenter;   // per i-block bounding box (neighbor.hip); blockHalf.w = 1 when the block's atoms are image-coherent
    const float4* blockHalf;
    float cutoff;
    omm_fixed* force;
    double* energyBuffer;     // one slot per workgroup
};

#define OMM_LJ_HEAD 12        // slots of a block that may hold atoms with Lennard-Jones parameters when the rest has none (32 water atoms: 10-11 oxygens)

// EDGE = false: called from the cutoff-edge path itself (divergent code: no wave-wide mask arithmetic there)
template <int METHOD, int PBC, bool ENERGY, bool MASKED, bool EDGE = true>
__device__ __forceinline__ void pair_ixn(const NbArgs& a, const float4 pi, const float2 sei, const float4 pj, const float2 sej, const float qjK,
                                         bool bit, float& fix, float& fiy, float& fiz, float& fjx, float& fjy, float& fjz, float& energy,
                                         unsigned long long& edge) {
    float dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;
    if (PBC == 1) min_image<false>(dx, dy, dz, a.box);
    if (PBC == 2) min_image<true>(dx, dy, dz, a.box);
    const float r2 = r2_of(dx, dy, dz);
    bool in = r2 < a.cutoff2;
    if (EDGE) mask_accumulate(edge, wave_ballot(in), wave_ballot(r2 < a.cutoff2Lo));        // scalar mask arithmetic: one extra vector compare per pair
    if (MASKED) in = in && bit;
    const float invR = fast_rsqrt(r2);
    const float r = r2 * invR;
    const float invR2 = invR * invR;
    const float sig = sei.x + sej.x;
    const float eps = sei.y * sej.y;
    float s2 = sig * invR; s2 *= s2;
    const float s6 = s2 * s2 * s2;
    float ljF = eps * (12.f * s6 - 6.f) * s6;          // dE/dr * (-r)
    float ljE = eps * (s6 - 1.f) * s6;
    float dispF = 0.f;                                  // LJPME: force coefficient of the multiplicative C6 term (multiplies the separation vector)
    if (METHOD & 4) {
        // ReferenceLJCoulombIxn.cpp:407-435: the Lorentz-Berthelot LJ stays as it is; the part of the geometric-mean C6 term that
        // reciprocal space does not cover is taken out again, plus a shift that makes the energy continuous at the cutoff
        const float c6 = (8.f * sei.x * sei.x * sei.x * sei.y) * (8.f * sej.x * sej.x * sej.x * sej.y);
        const float x = a.dispAlpha2 * r2, ex = fast_exp(-x);
        const float invR6 = invR2 * invR2 * invR2;
        dispF = 6.f * c6 * invR6 * invR2 * (1.f - ex * (1.f + x + 0.5f * x * x + x * x * x * (1.f / 6.f)));
        if (ENERGY || (METHOD & 2)) {
            float sc2 = sig * sig; const float sc6 = sc2 * sc2 * sc2 * a.invCut6;
            ljE += c6 * invR6 * (1.f - ex * (1.f + x + 0.5f * x * x)) + eps * (1.f - sc6) * sc6 - c6 * a.dispShift;
        }
    }
    if (METHOD & 2) {
        // ReferenceLJCoulombIxn.cpp:388-392,437-440 / :587-594,615-618; t = 0 below the switching
        // distance gives sw = 1, dsw = 0, so no branch is needed.
        const float t = fmaxf(0.f, (r - a.switchDist) * a.invSwitchWidth);
        const float sw = 1.f + t * t * t * (-10.f + t * (15.f - t * 6.f));
        const float dsw = t * t * (-30.f + t * (60.f - t * 30.f)) * a.invSwitchWidth;
        ljF = sw * ljF - ljE * dsw * r;
        ljE *= sw;
    }
    const float qq = pi.w * qjK;
    float cF, cE;
    if (METHOD & 1) {
        // erfc(alpha r) + 2 alpha r exp(-alpha^2 r^2)/sqrt(pi)      (ReferenceLJCoulombIxn.cpp:396-399)
        const float ar = a.alpha * r;
        const float ex = fast_exp(-ar * ar);
        const float t = fast_rcp(1.f + OMM_ERFC_P * ar);
        const float c[OMM_ERFC_DEGREE + 1] = OMM_ERFC_COEFFS;
        float poly = c[0];
#pragma unroll
        for (int n = 1; n <= OMM_ERFC_DEGREE; n++) poly = poly * t + c[n];
        const float erfcv = ex * t * poly;
        cE = qq * invR * erfcv;
        cF = qq * invR * (erfcv + ar * ex * 1.12837916709551257390f);
    }
    else {
        // reaction field (krf = crf = 0 for NoCutoff)          (ReferenceLJCoulombIxn.cpp:611-624)
        cF = qq * (invR - 2.f * a.krf * r2);
        cE = qq * (invR + a.krf * r2 - a.crf);
    }
    float dEdR = (ljF + cF) * invR2 + dispF;
    dEdR = in ? dEdR : 0.f;
    fjx += dEdR * dx; fjy += dEdR * dy; fjz += dEdR * dz;
    fix -= dEdR * dx; fiy -= dEdR * dy; fiz -= dEdR * dz;
    if (ENERGY) energy += in ? (ljE + cE) : 0.f;
}

// ------------------------------------------------------------------------------------------------
// Two i atoms per call in packed FP32 (v_pk_add/mul/fma_f32: two floats per lane per instruction at the issue rate of
// one) -- the vector ALUs of CDNA3/4 only reach their FP32 peak through these.  Same arithmetic as pair_ixn; the three
// transcendentals per pair (rsq, rcp, exp) stay scalar.  Used on the single-image path, where no image search is needed.
// ------------------------------------------------------------------------------------------------
// wave-uniform copy of lane k's value (v_readlane_b32): i-atom data held one atom per lane, no memory access in the loop
__device__ __forceinline__ float rl(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }
__device__ __forceinline__ float4 rl4(float4 v, int k) { return make_float4(rl(v.x, k), rl(v.y, k), rl(v.z, k), rl(v.w, k)); }
__device__ __forceinline__ float2 rl2(float2 v, int k) { return make_float2(rl(v.x, k), rl(v.y, k)); }

// NOLJ: both i atoms have epsilon = 0 (wave-uniform, decided by the caller from the i-block's parameters): the whole
// Lennard-Jones part -- a fifth of the packed arithmetic -- is left out; its terms would all carry the factor eps_i eps_j = 0.
// Two i atoms as the packed loops take them: (atom k, atom k + 1) per component.
struct IPair { v2f x, y, z, q, sig, eps; };

template <int METHOD, bool ENERGY, bool MASKED, bool NOLJ = false>
__device__ __forceinline__ void pair_ixn2(const NbArgs& a, const IPair& ip,
                                          const float4 pj, const float2 sej, const float qjK, bool bit0, bool bit1,
                                          v2f& fix, v2f& fiy, v2f& fiz, v2f& fjx, v2f& fjy, v2f& fjz, v2f& energy, unsigned long long& edge) {
    const v2f dx = bc2(pj.x) - ip.x, dy = bc2(pj.y) - ip.y, dz = bc2(pj.z) - ip.z;
    const v2f r2 = r2_of(dx, dy, dz);
    bool in0 = r2.x < a.cutoff2, in1 = r2.y < a.cutoff2;
    mask_accumulate(edge, wave_ballot(in0), wave_ballot(r2.x < a.cutoff2Lo));
    mask_accumulate(edge, wave_ballot(in1), wave_ballot(r2.y < a.cutoff2Lo));
    if (MASKED) { in0 = in0 && bit0; in1 = in1 && bit1; }
    const v2f invR = mk2(fast_rsqrt(r2.x), fast_rsqrt(r2.y));
    const v2f r = r2 * invR;
    const v2f invR2 = invR * invR;
    v2f ljF = bc2(0.f), ljE = bc2(0.f), dispF = bc2(0.f);
    if (!NOLJ) {
        const v2f sig = ip.sig + bc2(sej.x);
        const v2f eps = ip.eps * bc2(sej.y);
        v2f s2 = sig * invR; s2 = s2 * s2;
        const v2f s6 = s2 * s2 * s2;
        ljF = eps * (bc2(12.f) * s6 - bc2(6.f)) * s6;
        ljE = eps * (s6 - bc2(1.f)) * s6;
        if (METHOD & 4) {
            const v2f c6 = (bc2(8.f) * ip.sig * ip.sig * ip.sig * ip.eps) * bc2(8.f * sej.x * sej.x * sej.x * sej.y);
            const v2f x = bc2(a.dispAlpha2) * r2;
            const v2f argd = -x * bc2(1.44269504088896340736f);
            const v2f ex = mk2(__builtin_amdgcn_exp2f(argd.x), __builtin_amdgcn_exp2f(argd.y));
            const v2f invR6 = invR2 * invR2 * invR2;
            dispF = bc2(6.f) * c6 * invR6 * invR2 * (bc2(1.f) - ex * (bc2(1.f) + x + bc2(0.5f) * x * x + x * x * x * bc2(1.f / 6.f)));
            if (ENERGY || (METHOD & 2)) {
                const v2f sc2 = sig * sig; const v2f sc6 = sc2 * sc2 * sc2 * bc2(a.invCut6);
                ljE = ljE + c6 * invR6 * (bc2(1.f) - ex * (bc2(1.f) + x + bc2(0.5f) * x * x)) + eps * (bc2(1.f) - sc6) * sc6 - c6 * bc2(a.dispShift);
            }
        }
        if (METHOD & 2) {
            v2f t = (r - bc2(a.switchDist)) * bc2(a.invSwitchWidth);
            t = mk2(fmaxf(0.f, t.x), fmaxf(0.f, t.y));
            const v2f sw = bc2(1.f) + t * t * t * (bc2(-10.f) + t * (bc2(15.f) - t * bc2(6.f)));
            const v2f dsw = t * t * (bc2(-30.f) + t * (bc2(60.f) - t * bc2(30.f))) * bc2(a.invSwitchWidth);
            ljF = sw * ljF - ljE * dsw * r;
            ljE = ljE * sw;
        }
    }
    const v2f qq = ip.q * bc2(qjK);
    if ((METHOD & 8) && !ENERGY) {
        // Real-space Ewald force without exp and rcp (forces only): the bracket of ReferenceLJCoulombIxn.cpp:396-399 over r^3 is
        //   [erfc(ar) + 2 ar exp(-(ar)^2) / sqrt(pi)] / r^3 = 1 / r^3 - alpha^3 g(z),   z = (alpha r)^2,
        //   g(z) = [erf(sqrt z) - 2 sqrt(z / pi) exp(-z)] / z^(3/2)   (entire in z, g(0) = 4 / (3 sqrt pi)),
        // and alpha^3 g is a degree-11 polynomial in z on [0, (alpha cutoff)^2] to 1e-7 of g(0) (fitted for the run's alpha and
        // cutoff on the host, make_nb_args).  The difference of two numbers of the size of 1/r^3 is good to 6e-8 / r^3 --
        // 1e-7 of the pair's own Coulomb force at contact, 1e-4 kJ/mol/nm at the cutoff, where the force itself is tiny.
        const v2f u = r2 * bc2(a.ewPolyScale) - bc2(1.f);
        v2f poly = bc2(a.ewPoly[0]);
#pragma unroll
        for (int n = 1; n <= OMM_EWPOLY_DEGREE; n++) poly = poly * u + bc2(a.ewPoly[n]);
        const v2f coul = qq * (invR2 * invR - poly);
        v2f dEdR = NOLJ ? coul : ljF * invR2 + coul;
        dEdR = mk2(in0 ? dEdR.x : 0.f, in1 ? dEdR.y : 0.f);
        fjx = fjx + dEdR * dx; fjy = fjy + dEdR * dy; fjz = fjz + dEdR * dz;
        fix = fix - dEdR * dx; fiy = fiy - dEdR * dy; fiz = fiz - dEdR * dz;
        return;
    }
    v2f cF, cE;
    if (METHOD & 1) {
        const v2f ar = bc2(a.alpha) * r;
        const v2f arg = -(ar * ar) * bc2(1.44269504088896340736f);
        const v2f ex = mk2(__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y));
        const v2f den = bc2(1.f) + bc2(OMM_ERFC_P) * ar;
        const v2f t = mk2(fast_rcp(den.x), fast_rcp(den.y));
        const float c[OMM_ERFC_DEGREE + 1] = OMM_ERFC_COEFFS;
        v2f poly = bc2(c[0]);
#pragma unroll
        for (int n = 1; n <= OMM_ERFC_DEGREE; n++) poly = poly * t + bc2(c[n]);
        const v2f erfcv = ex * t * poly;
        cE = qq * invR * erfcv;
        cF = qq * invR * (erfcv + ar * ex * bc2(1.12837916709551257390f));
    }
    else {
        cF = qq * (invR - bc2(2.f * a.krf) * r2);
        cE = qq * (invR + bc2(a.krf) * r2 - bc2(a.crf));
    }
    v2f dEdR = NOLJ ? cF * invR2 : (ljF + cF) * invR2 + dispF;
    dEdR = mk2(in0 ? dEdR.x : 0.f, in1 ? dEdR.y : 0.f);
    fjx = fjx + dEdR * dx; fjy = fjy + dEdR * dy; fjz = fjz + dEdR * dz;
    fix = fix - dEdR * dx; fiy = fiy - dEdR * dy; fiz = fiz - dEdR * dz;
    if (ENERGY) { const v2f e = NOLJ ? cE : ljE + cE; energy = energy + mk2(in0 ? e.x : 0.f, in1 ? e.y : 0.f); }
}

// Where the 32 i atoms of a block come from inside the pair loops.  OMM_I_FROM_LANES = 1: lane k holds atom k in six
// VGPRs (one coalesced vector load per chunk, together with the row gathers) and the loops fetch an atom with
// v_readlane_b32 -- no memory access in the loops.  0: wave-uniform scalar loads (s_load_dwordx16) from posqI / sigEpsI
// inside the loops: no VALU work, but 768 bytes do not fit the SGPR file, so every row re-loads them in pieces and waits
// for each piece -- measured on the 1M-atom box (rocprofv3 SQ counters, profiles/r03b_*): waves parked on s_waitcnt for
// 40-50 % of their cycles with the VALU active 30 %.
#ifndef OMM_I_FROM_LANES
#define OMM_I_FROM_LANES 1
#endif
// OMM_I_FROM_LANES = 2: the block's atoms as six arrays of 32 floats in LDS (x, y, z, q, sigma, epsilon; one copy per wavefront, written
// once per chunk), read with ds_read_b64 at compile-time offsets -- every lane reads the same address (a broadcast), the two floats land in
// a register pair a packed instruction takes as it is.  The LDS pipe is otherwise idle in this kernel; the lane form spends 8-12 VALU
// issue slots per pair of atoms on v_readlane_b32, a fifth of the loop (the list builder reads its i atoms this way: neighbor.hip).
#define OMM_I_LDS_FLOATS (6 * OMM_TILE)
struct IAtoms {
    const float4* __restrict__ ip; const float2* __restrict__ ise;     // memory form
    float4 pLane; float2 seLane;                                       // lane form
    const float* lds;                                                  // LDS form
    // k is a compile-time constant after unrolling: v_readlane_b32 with an immediate lane
    __device__ __forceinline__ float4 posq(int k) const {
        if (OMM_I_FROM_LANES == 2) return make_float4(lds[k], lds[OMM_TILE + k], lds[2 * OMM_TILE + k], lds[3 * OMM_TILE + k]);
        return OMM_I_FROM_LANES ? rl4(pLane, k) : ip[k];
    }
    __device__ __forceinline__ float2 sigEps(int k) const {
        if (OMM_I_FROM_LANES == 2) return make_float2(lds[4 * OMM_TILE + k], lds[5 * OMM_TILE + k]);
        return OMM_I_FROM_LANES ? rl2(seLane, k) : ise[k];
    }
    template <bool NOLJ>
    __device__ __forceinline__ IPair pair(int k) const {
        IPair r;
        if (OMM_I_FROM_LANES == 2) {
            const v2f* p = (const v2f*) lds;
            r.x = p[(0 * OMM_TILE + k) / 2]; r.y = p[(1 * OMM_TILE + k) / 2]; r.z = p[(2 * OMM_TILE + k) / 2]; r.q = p[(3 * OMM_TILE + k) / 2];
            if (!NOLJ) { r.sig = p[(4 * OMM_TILE + k) / 2]; r.eps = p[(5 * OMM_TILE + k) / 2]; }
            else { r.sig = bc2(0.f); r.eps = bc2(0.f); }
        }
        else {
            const float4 p0 = posq(k), p1 = posq(k + 1);
            r.x = mk2(p0.x, p1.x); r.y = mk2(p0.y, p1.y); r.z = mk2(p0.z, p1.z); r.q = mk2(p0.w, p1.w);
            if (!NOLJ) { const float2 s0 = sigEps(k), s1 = sigEps(k + 1); r.sig = mk2(s0.x, s1.x); r.eps = mk2(s0.y, s1.y); }
            else { r.sig = bc2(0.f); r.eps = bc2(0.f); }
        }
        return r;
    }
};

// i atoms [K0, K1) of a block against the j atom of this lane, two per call (single-image path).
template <int METHOD, bool ENERGY, bool MASKED, bool NOLJ, int K0, int K1>
__device__ __forceinline__ void row_pairs2(const NbArgs& a, const IAtoms& ia, const float4 pj, const float2 sej,
                                           const float qjK, const unsigned m, float (&fix)[OMM_TILE], float (&fiy)[OMM_TILE], float (&fiz)[OMM_TILE],
                                           v2f& fj2x, v2f& fj2y, v2f& fj2z, v2f& energy2, unsigned long long& edge) {
#pragma unroll
    for (int k = K0; k < K1; k += 2) {
#if OMM_I_FROM_LANES == 2 && !defined(OMMHIP_EMU)
        // keep the LDS reads of an iteration inside it: hoisted freely, the 16 x 6 register pairs of a row's i atoms are all requested up
        // front and the kernel -- two registers short of its budget as it is -- spills them
        __builtin_amdgcn_sched_barrier(0);
#endif
        v2f ax = mk2(fix[k], fix[k + 1]), ay = mk2(fiy[k], fiy[k + 1]), az = mk2(fiz[k], fiz[k + 1]);
        pair_ixn2<METHOD, ENERGY, MASKED, NOLJ>(a, ia.template pair<NOLJ>(k), pj, sej, qjK, MASKED ? ((m >> k) & 1u) != 0 : true,
                                                MASKED ? ((m >> (k + 1)) & 1u) != 0 : true, ax, ay, az, fj2x, fj2y, fj2z, energy2, edge);
        fix[k] = ax.x; fix[k + 1] = ax.y; fiy[k] = ay.x; fiy[k + 1] = ay.y; fiz[k] = az.x; fiz[k + 1] = az.y;
    }
}
// ... and one per call with the image search per pair (general path)
template <int METHOD, int PBC, bool ENERGY, bool MASKED>
__device__ __forceinline__ void row_pairs1(const NbArgs& a, const IAtoms& ia, const float4 pj, const float2 sej, const float qjK, const unsigned m,
                                           float (&fix)[OMM_TILE], float (&fiy)[OMM_TILE], float (&fiz)[OMM_TILE], float& fjx, float& fjy, float& fjz, float& energy,
                                           unsigned long long& edge) {
#pragma unroll
    for (int k = 0; k < OMM_TILE; k++)
        pair_ixn<METHOD, PBC, ENERGY, MASKED>(a, ia.posq(k), ia.sigEps(k), pj, sej, qjK, MASKED ? ((m >> k) & 1u) != 0 : true, fix[k], fiy[k], fiz[k], fjx, fjy, fjz, energy, edge);
}

// The rare path behind the cutoff-edge flag of a row (see NbArgs::cutoff2Lo): some lane's j atom has a partner among the 32 i
// atoms whose float separation lies within the rounding band around the cutoff.  Walk through the i atoms once more (a real loop,
// wave-uniform k: this code runs for a handful of the millions of rows of a step), find those pairs with the arithmetic of the
// packed loops (r2_of: same bits), decide each from the double-precision separation -- block-relative hi + lo coordinates, the
// offset between the two block centres and the minimum image all in double -- and subtract the pair's force and energy again when
// it lies outside the cutoff there (the Reference platform includes r^2 <= rc^2, ReferenceNeighborList.cpp:195-197).
// pjShifted: the j atom as the loops saw it (in X's frame; image chosen when `single`).
template <int METHOD, int PBC, bool ENERGY>
__device__ __forceinline__ void fix_edge_pairs(const NbArgs& a, const float4 iLane, const float2 seLane, const int X, const int j, const float4 pjShifted,
                                                      const float4 cX, const float4 cY, const float2 sej, const float qjK, const unsigned m, const bool single,
                                                      float& fjx, float& fjy, float& fjz, float& energy) {
    for (int k = 0; k < OMM_TILE; k++) {
        const float4 pi = rl4(iLane, k);
        const float2 sei = rl2(seLane, k);
        float dx = pjShifted.x - pi.x, dy = pjShifted.y - pi.y, dz = pjShifted.z - pi.z;
        if (PBC == 1 && !single) min_image<false>(dx, dy, dz, a.box);
        if (PBC == 2) min_image<true>(dx, dy, dz, a.box);
        const float r2 = r2_of(dx, dy, dz);
        if (!(r2 < a.cutoff2 && !(r2 < a.cutoff2Lo) && ((m >> k) & 1u) != 0)) continue;
        const float4 pjr = a.posq[j], pjl = a.posqLo[j], pil = a.posqLo[X * OMM_TILE + k];
        double ex = ((double) pjr.x + (double) pjl.x) + ((double) cY.x - (double) cX.x) - ((double) pi.x + (double) pil.x);
        double ey = ((double) pjr.y + (double) pjl.y) + ((double) cY.y - (double) cX.y) - ((double) pi.y + (double) pil.y);
        double ez = ((double) pjr.z + (double) pjl.z) + ((double) cY.z - (double) cX.z) - ((double) pi.z + (double) pil.z);
        if (PBC != 0) min_image_d(ex, ey, ez, a.boxd);
        if (!(ex * ex + ey * ey + ez * ez > a.cutoff2d)) continue;          // inside in double as well: the loops were right
        float tix = 0.f, tiy = 0.f, tiz = 0.f, tjx = 0.f, tjy = 0.f, tjz = 0.f, te = 0.f;
        unsigned long long ignored = 0;
        if (PBC == 1 && !single) pair_ixn<METHOD & 7, 1, ENERGY, false, false>(a, pi, sei, pjShifted, sej, qjK, true, tix, tiy, tiz, tjx, tjy, tjz, te, ignored);
        else if (PBC == 2) pair_ixn<METHOD & 7, 2, ENERGY, false, false>(a, pi, sei, pjShifted, sej, qjK, true, tix, tiy, tiz, tjx, tjy, tjz, te, ignored);
        else pair_ixn<METHOD & 7, 0, ENERGY, false, false>(a, pi, sei, pjShifted, sej, qjK, true, tix, tiy, tiz, tjx, tjy, tjz, te, ignored);
        fjx -= tjx; fjy -= tjy; fjz -= tjz;
        if (ENERGY) energy -= te;
        add_force(a.force, a.paddedAtoms, X * OMM_TILE + k, -tix, -tiy, -tiz);
    }
}

// Two registers in, one out: lanes 0-31 get a[l] + a[l + 32], lanes 32-63 get b[l - 32] + b[l] -- gfx950's
// v_permlane32_swap exchanges the upper half of one register with the lower half of another, so one swap and one add
// halve two partial sums at once (a shuffle-based butterfly needs two selects, a cross-lane move and an add for that).
__device__ __forceinline__ float swap_add32(float a, float b, int lane) {
#ifdef OMMHIP_EMU
    const float xa = a + __shfl_xor(a, 32), xb = b + __shfl_xor(b, 32);
    return lane < 32 ? xa : xb;
#else
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
}
// The same one level down (v_permlane16_swap: odd 16-lane rows of one register <-> even rows of the other):
// even rows get a[row] + a[row + 1], odd rows get b[row - 1] + b[row].
__device__ __forceinline__ float swap_add16(float a, float b, int lane) {
#ifdef OMMHIP_EMU
    const float xa = a + __shfl_xor(a, 16), xb = b + __shfl_xor(b, 16);
    return (lane & 16) ? xb : xa;
#else
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
}

// Transpose-reduce: on entry every lane holds 32 partial sums v[0..32); on exit lane l holds the wave-wide total of
// v[l >> 1] (each total in two neighbouring lanes).  24 lane swaps + 8 cross-lane moves instead of 32 * 6 moves.
__device__ __forceinline__ float transpose_reduce32(float (&v)[OMM_TILE], int lane) {
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = swap_add32(v[k], v[k + 16], lane);        // halves: v[k] | v[k + 16]
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = swap_add16(v[k], v[k + 8], lane);          // rows: v[k], v[k + 8], v[k + 16], v[k + 24]
#pragma unroll
    for (int k = 0; k < 4; k++) { bool up = lane & 8; float send = up ? v[k] : v[k + 4]; float keep = up ? v[k + 4] : v[k]; v[k] = keep + __shfl_xor(send, 8); }
#pragma unroll
    for (int k = 0; k < 2; k++) { bool up = lane & 4; float send = up ? v[k] : v[k + 2]; float keep = up ? v[k + 2] : v[k]; v[k] = keep + __shfl_xor(send, 4); }
    { bool up = lane & 2; float send = up ? v[0] : v[1]; float keep = up ? v[1] : v[0]; v[0] = keep + __shfl_xor(send, 2); }
    return v[0] + __shfl_xor(v[0], 1);
}

// Test hook (ommhip_test_transpose_reduce): one wavefront, lane l starts with the 32 values in[32 l .. 32 l + 32) and ends with the
// wave-wide total of column l >> 1 -- what the pair kernel does with the forces on its 32 i atoms.  The product build runs the
// v_permlane32/16_swap form, the CPU emulator its shuffle twin: tests/test_gpu_kernels.py pins the former against plain sums.
__global__ __launch_bounds__(64) void k_test_transpose_reduce(const float* __restrict__ in, float* __restrict__ out) {
    const int lane = threadIdx.x;
    float v[OMM_TILE];
#pragma unroll
    for (int k = 0; k < OMM_TILE; k++) v[k] = in[(size_t) blockIdx.x * 64 * OMM_TILE + lane * OMM_TILE + k];
    out[(size_t) blockIdx.x * 64 + lane] = transpose_reduce32(v, lane);
}

// A wavefront's work units: of the part [fracLo, fracHi) / 64 of the list (fused launches split the list between several
// launches; its length is only known on the device), the units first, first + stride, ...
// xcd >= 0: XCD-aware placement.  Workgroups go to the 8 XCDs round-robin by index and every XCD has an L2 of its own, so
// the launch gives XCD x the x-th eighth of the units -- neighbouring chunks (same i-block, overlapping j atoms, which the
// Hilbert-sorted slot order keeps close in memory) then share an L2 instead of each L2 seeing the whole system.  `first`
// is the wavefront's rank among the wavefronts of its XCD, `stride` their number.
struct ChunkSchedule {
    int first, stride;
    int fracLo, fracHi;
    int xcd;
};
#define OMM_NUM_XCD 8

// One wavefront's share of the pair kernel.  A work unit is UNIT_ROWS rows of one chunk: the whole chunk in the kernel of
// its own (the i-block's atoms are set up and its forces reduced once per chunk), single rows where the unit's latency
// matters more than that overhead (fused launches).
template <int METHOD, int PBC, bool ENERGY, int UNIT_ROWS = OMM_CHUNK_ROWS>
__device__ __forceinline__ void nb_direct_body(const NbArgs& a, const float4* __restrict__ posqI, const float2* __restrict__ sigEpsI,
                                               const ChunkSchedule& sched, const int energySlot) {
    // posqI/sigEpsI alias a.posq/a.sigEps; passing them as separate __restrict__ kernel arguments
    // lets the compiler prove they are never written here and fetch the wave-uniform i-atom data
    // with scalar loads.
    const int lane = threadIdx.x & 63;
#if OMM_I_FROM_LANES == 2
    __shared__ float iLdsAll[4][OMM_I_LDS_FLOATS];          // one copy per wavefront of the workgroup (the fused launches run four)
    float* const iLds = iLdsAll[(threadIdx.x >> 6) & 3];
#else
    float* const iLds = nullptr;
#endif
    // wave-uniform choice of the list: the per-step pruned rows when the builder keeps them
    const bool pruned = a.rowJInner != nullptr && a.state[ST_NO_PRUNE] == 0;
    const int2* __restrict__ const chunkInfo = pruned ? a.chunkInfoInner : a.chunkInfo;
    const int* __restrict__ const rowJ = pruned ? a.rowJInner : a.rowJ;
    const unsigned* __restrict__ const rowMask = pruned ? a.rowMaskInner : a.rowMask;
    int numChunks = a.state[pruned ? ST_NUM_CHUNKS_INNER : ST_NUM_CHUNKS];
    if (numChunks > a.maxChunks) numChunks = a.maxChunks;
    double energyTotal = 0;
    constexpr int UNITS_PER_CHUNK = OMM_CHUNK_ROWS / UNIT_ROWS;
    const long long numUnits = (long long) numChunks * UNITS_PER_CHUNK;
    const int unitLo = (int) ((numUnits * sched.fracLo) >> 6), unitHi = (int) ((numUnits * sched.fracHi) >> 6);
    int uBegin = unitLo + sched.first, uEnd = unitHi;
    if (sched.xcd >= 0) {
        const int perXcd = (unitHi - unitLo + OMM_NUM_XCD - 1) / OMM_NUM_XCD;
        uBegin = unitLo + sched.xcd * perXcd + sched.first;
        uEnd = min(unitHi, unitLo + (sched.xcd + 1) * perXcd);
    }
    for (int u = uBegin; u < uEnd; u += sched.stride) {
        const int c = u / UNITS_PER_CHUNK;
        const int rowBase = (u % UNITS_PER_CHUNK) * UNIT_ROWS;
        // The row words are read unconditionally and before the chunk header is looked at (the arrays cover every row of
        // every chunk; rows a chunk does not use hold stale words that are discarded below): header, row indices and masks
        // come back in ONE memory round trip, the posq / (sigma, eps) gathers that depend on the indices are the second.
        int jWord[UNIT_ROWS]; unsigned mWord[UNIT_ROWS];
#pragma unroll
        for (int row = 0; row < UNIT_ROWS; row++) {
            const size_t r = ((size_t) c * OMM_CHUNK_ROWS + rowBase + row) * OMM_ROW + lane;
            jWord[row] = rowJ[r];
            mWord[row] = rowMask[r];
        }
        const int2 info = chunkInfo[c];
        const int X = __builtin_amdgcn_readfirstlane(info.x);
        const int nrows = __builtin_amdgcn_readfirstlane(info.y & 0xff) - rowBase;       // rows of this unit
        const int maskedBits = __builtin_amdgcn_readfirstlane(info.y >> 8) >> rowBase;
        if (nrows <= 0) continue;
        IAtoms ia;
        ia.ip = posqI + X * OMM_TILE; ia.ise = sigEpsI + X * OMM_TILE;
        ia.lds = iLds;
        if (OMM_I_FROM_LANES) { ia.pLane = a.posq[X * OMM_TILE + (lane & (OMM_TILE - 1))]; ia.seLane = a.sigEps[X * OMM_TILE + (lane & (OMM_TILE - 1))]; }
        if (OMM_I_FROM_LANES == 2 && lane < OMM_TILE) {
            // (the reads of the previous chunk are complete: a wavefront's LDS operations execute in order)
            iLds[0 * OMM_TILE + lane] = ia.pLane.x; iLds[1 * OMM_TILE + lane] = ia.pLane.y; iLds[2 * OMM_TILE + lane] = ia.pLane.z; iLds[3 * OMM_TILE + lane] = ia.pLane.w;
            iLds[4 * OMM_TILE + lane] = ia.seLane.x; iLds[5 * OMM_TILE + lane] = ia.seLane.y;
        }
        float fix[OMM_TILE], fiy[OMM_TILE], fiz[OMM_TILE];
#pragma unroll
        for (int k = 0; k < OMM_TILE; k++) { fix[k] = 0.f; fiy[k] = 0.f; fiz[k] = 0.f; }
        float energy = 0.f, eRowStart = 0.f;
        v2f energy2 = bc2(0.f);
        // All coordinates are relative to the centre of the atom's own block (posq_rel); a j atom is moved into the frame
        // of X by adding the offset between the two block centres.  Nothing of the size of the box enters the pair
        // arithmetic, so the separations are good to ~1e-7 nm in a 6 nm box and in a 60 nm box alike.
        // Single-image path (rectangular boxes): when the block is image-coherent and block + cutoff stay inside half a
        // box length on every axis, the image of j nearest to the block centre is the nearest image for every i atom
        // within the cutoff (any other pair only comes out farther), so the image search is done once per j, not per pair.
        const float4 cX = a.blockCenter[X];
        bool single = false;
        if (PBC == 1) {
            const float4 hX = a.blockHalf[X];
            single = hX.w != 0.f && hX.x + a.cutoff < 0.5f * a.box.ax && hX.y + a.cutoff < 0.5f * a.box.by && hX.z + a.cutoff < 0.5f * a.box.cz;
        }
        // wave-uniform: do the i atoms from OMM_LJ_HEAD on all have epsilon = 0?  (scalar compares on the block's parameters)
        bool ljFree = a.ljHeadSplit != 0;
        if (PBC == 1 && single && ljFree) {
            if (OMM_I_FROM_LANES) ljFree = ((unsigned) __ballot(ia.seLane.y != 0.f) >> OMM_LJ_HEAD) == 0u;       // lanes 12..31 of the lower half
            else {
#pragma unroll
                for (int k = OMM_LJ_HEAD; k < OMM_TILE; k++) ljFree = ljFree && ia.ise[k].y == 0.f;
            }
        }
        // All rows of the chunk are fetched before the first one is processed (index, then the gathers that depend on
        // it): the two memory round trips are paid once per chunk and the later rows arrive while the first is computed.
        int jRow[UNIT_ROWS]; unsigned mRow[UNIT_ROWS]; float4 pjRow[UNIT_ROWS]; float2 seRow[UNIT_ROWS]; float4 cjRow[UNIT_ROWS];
#pragma unroll
        for (int row = 0; row < UNIT_ROWS; row++) {
            jRow[row] = row < nrows ? jWord[row] : X * OMM_TILE;
            mRow[row] = row < nrows && ((maskedBits >> row) & 1) ? mWord[row] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int row = 0; row < UNIT_ROWS; row++) { pjRow[row] = a.posq[jRow[row]]; seRow[row] = a.sigEps[jRow[row]]; cjRow[row] = a.blockCenter[jRow[row] >> 5]; }
#pragma unroll
        for (int row = 0; row < UNIT_ROWS; row++) {
            if (row >= nrows) break;
            const int j = jRow[row];
            float4 pj = pjRow[row];
            const float4 cY = cjRow[row];
            const float2 sej = seRow[row];
            const float qjK = OMM_ONE_4PI_EPS0 * pj.w;
            float fjx = 0.f, fjy = 0.f, fjz = 0.f;
            const bool masked = (maskedBits >> row) & 1;
            const unsigned m = mRow[row];
            unsigned long long edge = 0;          // lanes with a pair in the rounding band around the cutoff (scalar register pair)
            if (PBC == 1 && single) {
                // offset of Y's centre from X's, in the nearest image.  cY - n L is formed first: the two are of similar size,
                // so the FMA is exact, the low part of the box edge restores what its float value lost, and the final
                // difference is between two numbers a few nm apart at most.
                // (the image is chosen per j atom -- Y may be a wide block some of whose atoms are nearest in another image)
                const float nx = rintf((cY.x - cX.x + pj.x) * a.box.invAx), ny = rintf((cY.y - cX.y + pj.y) * a.box.invBy), nz = rintf((cY.z - cX.z + pj.z) * a.box.invCz);
                const float ox = fmaf(-nx, a.box.axLo, fmaf(-nx, a.box.ax, cY.x)) - cX.x;
                const float oy = fmaf(-ny, a.box.byLo, fmaf(-ny, a.box.by, cY.y)) - cX.y;
                const float oz = fmaf(-nz, a.box.czLo, fmaf(-nz, a.box.cz, cY.z)) - cX.z;
                pj.x += ox; pj.y += oy; pj.z += oz;
                v2f fj2x = bc2(0.f), fj2y = bc2(0.f), fj2z = bc2(0.f);
                if (ljFree) {
                    // the block's atoms from OMM_LJ_HEAD on carry no Lennard-Jones parameters (water: the slot order puts the
                    // oxygens of a block first): two thirds of the row's pairs skip that part of the arithmetic
                    if (masked) {
                        row_pairs2<METHOD, ENERGY, true, false, 0, OMM_LJ_HEAD>(a, ia, pj, sej, qjK, m, fix, fiy, fiz, fj2x, fj2y, fj2z, energy2, edge);
                        row_pairs2<METHOD, ENERGY, true, true, OMM_LJ_HEAD, OMM_TILE>(a, ia, pj, sej, qjK, m, fix, fiy, fiz, fj2x, fj2y, fj2z, energy2, edge);
                    }
                    else {
                        row_pairs2<METHOD, ENERGY, false, false, 0, OMM_LJ_HEAD>(a, ia, pj, sej, qjK, m, fix, fiy, fiz, fj2x, fj2y, fj2z, energy2, edge);
                        row_pairs2<METHOD, ENERGY, false, true, OMM_LJ_HEAD, OMM_TILE>(a, ia, pj, sej, qjK, m, fix, fiy, fiz, fj2x, fj2y, fj2z, energy2, edge);
                    }
                }
                else if (masked) row_pairs2<METHOD, ENERGY, true, false, 0, OMM_TILE>(a, ia, pj, sej, qjK, m, fix, fiy, fiz, fj2x, fj2y, fj2z, energy2, edge);
                else row_pairs2<METHOD, ENERGY, false, false, 0, OMM_TILE>(a, ia, pj, sej, qjK, m, fix, fiy, fiz, fj2x, fj2y, fj2z, energy2, edge);
                fjx += fj2x.x + fj2x.y; fjy += fj2y.x + fj2y.y; fjz += fj2z.x + fj2z.y;
            }
            else {
                // general path: j in X's frame without an image shift; the pair code searches the image per pair
                pj.x += cY.x - cX.x; pj.y += cY.y - cX.y; pj.z += cY.z - cX.z;
                if (masked) row_pairs1<METHOD, PBC, ENERGY, true>(a, ia, pj, sej, qjK, m, fix, fiy, fiz, fjx, fjy, fjz, energy, edge);
                else row_pairs1<METHOD, PBC, ENERGY, false>(a, ia, pj, sej, qjK, m, fix, fiy, fiz, fjx, fjy, fjz, energy, edge);
            }
            if (edge != 0) {
                // wave-uniform and rare (a few rows per step): pairs within float rounding of the cutoff are re-decided in double
                const float4 iLane = OMM_I_FROM_LANES == 1 ? ia.pLane : a.posq[X * OMM_TILE + (lane & (OMM_TILE - 1))];
                const float2 seLane = OMM_I_FROM_LANES == 1 ? ia.seLane : a.sigEps[X * OMM_TILE + (lane & (OMM_TILE - 1))];
                fix_edge_pairs<METHOD, PBC, ENERGY>(a, iLane, seLane, X, j, pj, cX, cY, sej, qjK, m, PBC == 1 && single, fjx, fjy, fjz, energy);
            }
            const bool jOwned = (j >= a.ownSlot0 && j < a.ownSlot1) || (j >= a.keepSlot0 && j < a.keepSlot1);
            if (!(a.debugFlags & 1)) { if (jOwned) add_force(a.force, a.paddedAtoms, j, fjx, fjy, fjz); }
            else if (fjx == 12345.f) a.force[0] = 1;           // profiling knob: keep the arithmetic alive without the atomics
            if (ENERGY) {
                // a pair with a foreign j atom is also evaluated by j's owner: half of its energy belongs to this side
                const float eNow = energy + energy2.x + energy2.y;
                if (!jOwned) { energy = eRowStart + 0.5f * (eNow - eRowStart); energy2 = bc2(0.f); }
                eRowStart = energy + energy2.x + energy2.y;
            }
        }
        const float tx = transpose_reduce32(fix, lane);
        const float ty = transpose_reduce32(fiy, lane);
        const float tz = transpose_reduce32(fiz, lane);
        if ((lane & 1) == 0) {          // lane l holds the total of i atom l >> 1
            if (!(a.debugFlags & 2)) add_force(a.force, a.paddedAtoms, X * OMM_TILE + (lane >> 1), tx, ty, tz);
            else if (tx == 12345.f) a.force[0] = 1;
        }
        if (ENERGY) energyTotal += (double) energy + (double) energy2.x + (double) energy2.y;
    }
    if (ENERGY) {
        energyTotal = wave_sum(energyTotal);
        if (lane == 0 && energyTotal != 0.0) atomicAdd(&a.energyBuffer[energySlot % a.energySlots], energyTotal);
    }
}

template <int METHOD, int PBC, bool ENERGY>
__global__ __launch_bounds__(64, 2) void nb_direct(NbArgs a, const float4* __restrict__ posqI, const float2* __restrict__ sigEpsI) {
    ChunkSchedule sched = {(int) blockIdx.x, (int) gridDim.x, 0, 64, -1};
    if (a.xcdAware) {
        // one wavefront per workgroup: workgroup b runs on XCD b % 8 and is the (b / 8)-th of the gridDim.x / 8 there
        sched.xcd = blockIdx.x % OMM_NUM_XCD; sched.first = blockIdx.x / OMM_NUM_XCD; sched.stride = gridDim.x / OMM_NUM_XCD;
    }
    nb_direct_body<METHOD, PBC, ENERGY>(a, posqI, sigEpsI, sched, blockIdx.x);
}

template <int METHOD, int PBC>
void launch_direct2(bool energy, int grid, hipStream_t st, const NbArgs& a) {
    if (energy) hipLaunchKernelGGL((nb_direct<METHOD, PBC, true>), dim3(grid), dim3(64), 0, st, a, a.posq, a.sigEps);
    else hipLaunchKernelGGL((nb_direct<METHOD, PBC, false>), dim3(grid), dim3(64), 0, st, a, a.posq, a.sigEps);
}
template <int METHOD>
void launch_direct1(int pbc, bool energy, int grid, hipStream_t st, const NbArgs& a) {
    if (pbc == 0) launch_direct2<METHOD, 0>(energy, grid, st, a);
    else if (pbc == 1) launch_direct2<METHOD, 1>(energy, grid, st, a);
    else launch_direct2<METHOD, 2>(energy, grid, st, a);
}

}  // namespace

// alpha^3 g(z), z = (alpha r)^2, as a polynomial of degree OMM_EWPOLY_DEGREE in u = 2 z / zmax - 1 (see pair_ixn2, METHOD & 8):
// interpolation at the Chebyshev nodes (near-minimax for this entire function), converted to powers of u -- the series
// decays so fast that the power form is as well conditioned as the Chebyshev one (sum |c_k| = g(0) to seven digits).
// ok = the fit is good to 1.5e-7 g(0) over the whole range (alpha * cutoff up to ~2.9; Ewald tolerances down to ~1e-4).
struct EwaldPoly { double alpha = -1, cutoff = -1; bool ok = false; float coeff[OMM_EWPOLY_DEGREE + 1]; float scale = 0; double maxErr = 0; };
static double ewald_g(double z) {
    if (z < 1e-3) return 4.0 / (3.0 * sqrt(M_PI)) * (1.0 - 0.6 * z + 3.0 * z * z / 14.0 - z * z * z / 27.0);
    const double s = sqrt(z);
    return (erf(s) - 2.0 * s / sqrt(M_PI) * exp(-z)) / (z * s);
}
static const EwaldPoly& ewald_poly_for(double alpha, double cutoff) {
    static thread_local EwaldPoly cache;
    if (cache.alpha == alpha && cache.cutoff == cutoff) return cache;
    cache.alpha = alpha; cache.cutoff = cutoff; cache.ok = false;
    if (!(alpha > 0) || !(cutoff > 0)) return cache;
    constexpr int N = OMM_EWPOLY_DEGREE;
    const double zmax = alpha * alpha * cutoff * cutoff * 1.0001, a3 = alpha * alpha * alpha;
    double f[N + 1], x[N + 1], cheb[N + 1];
    for (int j = 0; j <= N; j++) { x[j] = cos(M_PI * (j + 0.5) / (N + 1)); f[j] = a3 * ewald_g(0.5 * (x[j] + 1.0) * zmax); }
    for (int k = 0; k <= N; k++) {
        double sum = 0;
        for (int j = 0; j <= N; j++) sum += f[j] * cos(M_PI * k * (j + 0.5) / (N + 1));
        cheb[k] = (k == 0 ? 1.0 : 2.0) * sum / (N + 1);
    }
    // Chebyshev series -> powers of u:  T_0 = 1, T_1 = u, T_{k+1} = 2 u T_k - T_{k-1}
    double mono[N + 1] = {0}, tPrev[N + 1] = {0}, tCur[N + 1] = {0}, tNext[N + 1];
    tPrev[0] = 1.0; tCur[1] = 1.0;
    for (int i = 0; i <= N; i++) mono[i] += cheb[0] * tPrev[i] + (N >= 1 ? cheb[1] * tCur[i] : 0.0);
    for (int k = 2; k <= N; k++) {
        for (int i = 0; i <= N; i++) tNext[i] = (i > 0 ? 2.0 * tCur[i - 1] : 0.0) - tPrev[i];
        for (int i = 0; i <= N; i++) { mono[i] += cheb[k] * tNext[i]; tPrev[i] = tCur[i]; tCur[i] = tNext[i]; }
    }
    for (int i = 0; i <= N; i++) cache.coeff[i] = (float) mono[N - i];          // highest power first (Horner)
    cache.scale = (float) (2.0 * alpha * alpha / zmax);
    // check the single-precision Horner form over the whole range
    double worst = 0;
    for (int i = 0; i <= 2000; i++) {
        const double r = cutoff * i / 2000.0;
        const float u = (float) (r * r) * cache.scale - 1.f;
        float p = cache.coeff[0];
        for (int n = 1; n <= N; n++) p = p * u + cache.coeff[n];
        worst = fmax(worst, fabs((double) p - a3 * ewald_g(alpha * alpha * r * r)));
    }
    cache.maxErr = worst / (a3 * ewald_g(0.0));
    cache.ok = cache.maxErr < 4e-7;          // fit error (<= 1.5e-7 up to alpha * cutoff = 2.9) + float rounding of the Horner steps
    return cache;
}

static NbArgs make_nb_args(const ommhip_neighbor_list* nl, const ommhip_nonbonded_params* p, const void* sig_eps,
                           long long* force, double* energy_buffer, int energy_slots) {
    NbArgs a;
    a.paddedAtoms = nl->padded_atoms; a.maxChunks = nl->max_chunks; a.energySlots = energy_slots;
    // OPENMM_HIP_DEBUG_SKIP_ATOMICS (profiling only, results are wrong): bit 0 drops the j-force atomics, bit 1 the i-force atomics
    static const int debugFlags = getenv("OPENMM_HIP_DEBUG_SKIP_ATOMICS") != nullptr ? atoi(getenv("OPENMM_HIP_DEBUG_SKIP_ATOMICS")) : 0;
    a.debugFlags = debugFlags;
    static const bool xcdAware = getenv("OPENMM_HIP_NO_XCD_PLACEMENT") == nullptr;          // A/B knob
    a.xcdAware = xcdAware ? 1 : 0;
    static const bool ljHeadSplit = getenv("OPENMM_HIP_NO_LJ_SPLIT") == nullptr;            // A/B knob
    a.ljHeadSplit = ljHeadSplit ? 1 : 0;
    a.ownSlot0 = 0; a.ownSlot1 = nl->padded_atoms;
    a.keepSlot0 = a.keepSlot1 = 0;
    if (nl->dd_mode != 0 && nl->owned_blocks > 0) {
        a.ownSlot0 = nl->first_block * OMM_TILE; a.ownSlot1 = (nl->first_block + nl->owned_blocks) * OMM_TILE;
        if (nl->dd_half_shell != 0) { a.keepSlot0 = nl->dd_eval_slot0; a.keepSlot1 = nl->dd_eval_slot1; }
    }
    a.cutoff2 = nl->cutoff > 0 ? (float) (nl->cutoff * nl->cutoff) : INFINITY;
    a.cutoff2Lo = a.cutoff2; a.cutoff2d = nl->cutoff > 0 ? nl->cutoff * nl->cutoff : INFINITY;
    a.posqLo = (const float4*) nl->posq_rel_lo;
    a.boxd.ax = nl->box[0]; a.boxd.bx = nl->box[1]; a.boxd.by = nl->box[2]; a.boxd.cx = nl->box[3]; a.boxd.cy = nl->box[4]; a.boxd.cz = nl->box[5];
    static const bool noEdge = getenv("OPENMM_HIP_NO_CUTOFF_EDGE") != nullptr;           // A/B knob: the float separation decides
    if (a.posqLo != nullptr && nl->cutoff > 0 && !noEdge) {
        // band of about +-1.2e-6 rc^2 around rc^2: the float r^2 of a pair at the cutoff is off by 2 r x (1e-7 nm of separation error) =
        // 2.2e-7 rc^2 at rc = 0.9 nm (block-relative coordinates, whatever the box), plus the rounding of r^2 itself; ~13 of DHFR's 3.5 M pairs fall into it
        const double c2 = nl->cutoff * nl->cutoff, band = 1e-6 * c2 + 2e-7 * nl->cutoff;
        a.cutoff2 = (float) (c2 + band); a.cutoff2Lo = (float) (c2 - band);
    }
    a.alpha = (float) p->ewald_alpha; a.krf = (float) p->krf; a.crf = (float) p->crf;
    a.ewPolyScale = 0.f;
    for (int i = 0; i <= OMM_EWPOLY_DEGREE; i++) a.ewPoly[i] = 0.f;
    if (p->ewald && nl->cutoff > 0) {
        const EwaldPoly& ep = ewald_poly_for(p->ewald_alpha, nl->cutoff);
        if (ep.ok) { a.ewPolyScale = ep.scale; for (int i = 0; i <= OMM_EWPOLY_DEGREE; i++) a.ewPoly[i] = ep.coeff[i]; }
    }
    a.switchDist = (float) p->switch_distance;
    a.dispAlpha2 = (float) (p->dispersion_alpha * p->dispersion_alpha);
    a.invCut6 = nl->cutoff > 0 ? (float) pow(nl->cutoff, -6.0) : 0.f;
    {
        const double xc = p->dispersion_alpha * p->dispersion_alpha * nl->cutoff * nl->cutoff;
        a.dispShift = nl->cutoff > 0 ? (float) ((1.0 - exp(-xc) * (1.0 + xc + 0.5 * xc * xc)) * pow(nl->cutoff, -6.0)) : 0.f;
    }
    a.invSwitchWidth = p->use_switch ? (float) (1.0 / (nl->cutoff - p->switch_distance)) : 0.f;
    a.box = make_box(nl->box);
    a.posq = (const float4*) nl->posq_rel; a.sigEps = (const float2*) sig_eps; a.state = nl->state;
    a.chunkInfo = (const int2*) nl->chunk_info; a.rowJ = nl->row_j; a.rowMask = nl->row_mask;
    const bool pruned = list_is_pruned(nl);
    a.chunkInfoInner = pruned ? (const int2*) nl->chunk_info_inner : nullptr; a.rowJInner = pruned ? nl->row_j_inner : nullptr; a.rowMaskInner = pruned ? nl->row_mask_inner : nullptr;
    a.blockCenter = (const float4*) nl->block_center; a.blockHalf = (const float4*) nl->block_half;
    a.cutoff = nl->cutoff > 0 ? (float) nl->cutoff : INFINITY;
    a.force = force; a.energyBuffer = energy_buffer;
    return a;
}

// METHOD 9 = plain Ewald/PME direct space with the polynomial form of the real-space force (forces only, rectangular box)
static bool use_ewald_poly(const NbArgs& a, const ommhip_neighbor_list* nl, const ommhip_nonbonded_params* p, int include_energy) {
    static const bool off = getenv("OPENMM_HIP_NO_EWALD_POLY") != nullptr;              // A/B knob
    return !off && a.ewPolyScale != 0.f && nl->pbc == 1 && p->ewald && !p->use_switch && !p->ljpme && include_energy == 0;
}

extern "C" int ommhip_test_transpose_reduce(const float* in_d, float* out_d, int num_waves, void* stream) {
    if (num_waves <= 0) return 1;
    hipLaunchKernelGGL(k_test_transpose_reduce, dim3(num_waves), dim3(64), 0, (hipStream_t) stream, in_d, out_d);
    return (int) hipGetLastError();
}

extern "C" int ommhip_nb_direct(const ommhip_neighbor_list* nl, const ommhip_nonbonded_params* p, const void* sig_eps,
                                long long* force, double* energy_buffer, int energy_slots, int include_energy, void* stream) {
    if (nl->posq_rel == nullptr) return 1;      // hipErrorInvalidValue: the pair kernel needs the block-relative coordinates
    NbArgs a = make_nb_args(nl, p, sig_eps, force, energy_buffer, energy_slots);
    // One workgroup (= one wavefront) per chunk: the list length is only known on the device, so the launch covers
    // the allocated capacity and surplus workgroups exit at once; the hardware dispatcher balances the rest.
    int grid = p->direct_grid > 0 ? p->direct_grid : nl->max_chunks;
    if (grid < 1) grid = 1;
    if (a.xcdAware) grid = (grid + OMM_NUM_XCD - 1) / OMM_NUM_XCD * OMM_NUM_XCD;       // the same number of wavefronts on every XCD
    hipStream_t st = (hipStream_t) stream;
    ommhip_profile_begin(OMMHIP_TIMER_NB_DIRECT, stream);
    if (p->ljpme && !p->ewald) return 1;
    if (use_ewald_poly(a, nl, p, include_energy)) hipLaunchKernelGGL((nb_direct<9, 1, false>), dim3(grid), dim3(64), 0, st, a, a.posq, a.sigEps);
    else switch ((p->ewald ? 1 : 0) | (p->use_switch ? 2 : 0) | (p->ljpme ? 4 : 0)) {
        case 0: launch_direct1<0>(nl->pbc, include_energy != 0, grid, st, a); break;
        case 1: launch_direct1<1>(nl->pbc, include_energy != 0, grid, st, a); break;
        case 2: launch_direct1<2>(nl->pbc, include_energy != 0, grid, st, a); break;
        case 3: launch_direct1<3>(nl->pbc, include_energy != 0, grid, st, a); break;
        case 5: launch_direct1<5>(nl->pbc, include_energy != 0, grid, st, a); break;
        default: launch_direct1<7>(nl->pbc, include_energy != 0, grid, st, a); break;
    }
    ommhip_profile_end(OMMHIP_TIMER_NB_DIRECT, stream);
    return (int) hipGetLastError();
}

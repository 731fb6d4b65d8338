// Per-term ("bonded-style") force kernels, one thread per term, double precision on the unwrapped
// double positions (atom order); results go to the fixed-point force buffer (slot order).
//
//   * NonbondedForce 1-4 exceptions        ReferenceLJCoulomb14.cpp (via ReferenceKernels.cpp:1000-1007)
//   * Ewald/PME exclusion correction       ReferenceLJCoulombIxn.cpp:462-523
//   * HarmonicBondForce                    ReferenceHarmonicBondIxn.cpp:73-112
//   * HarmonicAngleForce                   ReferenceAngleBondIxn.cpp:70-160
//   * PeriodicTorsionForce                 ReferenceProperDihedralBond.cpp:75-160, ReferenceBondIxn.cpp:115-205
//   * classic Ewald reciprocal k-sum       ReferenceLJCoulombIxn.cpp:272-367
//   * CMMotionRemover                      ReferenceKernels.cpp:2712-2740
//
// The work per step is tiny compared with the pair kernel (tens of thousands of terms), so these
// run in FP64 -- MI355X's FP64 vector rate makes that free and it removes a source of parity noise.
// At this size a launch costs more than the arithmetic (~4 us per dependent launch on the device
// timeline), so all term lists of one force evaluation go out in ONE launch: each list owns a
// contiguous range of workgroups, so every workgroup executes a single term kind (uniform branch).
#include "common.h"
#include "../../../include/openmm_hip_kernels.h"

using namespace omm;

namespace {

struct TermList {
    int kind, numTerms, periodic, firstBlock;
    int ownSlot0, ownSlot1;    // halo mode of a decomposed run (ownSlot1 > ownSlot0): terms without an owned atom are skipped, the energy counts where the first atom is owned
    int halfShell, evalSlot0, evalSlot1, upSlot0, upSlot1, rank, ranks, slotsPerRank;      // half-shell evaluation: ommhip_term_batch
    int* errorFlags;
    double alpha;
    const int* atoms;          // numTerms * atomsPerTerm
    const double* params;      // numTerms * paramsPerTerm
    const double* charge;      // per atom (exclusion correction)
};

struct TermArgs {
    int numLists, paddedAtoms, includeEnergy, energySlots;
    int debugSkipAtomics;     // profiling only (OPENMM_HIP_DEBUG_SKIP_TERM_ATOMICS, results are wrong): the arithmetic without the force atomics
    BoxD box;
    const double4* pos;
    const int* slotOfAtom;
    omm_fixed* force;
    double* energyBuffer;
    TermList list[OMMHIP_MAX_TERM_LISTS];
};

struct TermCtx {     // what one term evaluation needs
    const TermArgs& a;
    const TermList& l;
    __device__ TermCtx(const TermArgs& a_, const TermList& l_) : a(a_), l(l_) {}
    __device__ __forceinline__ double3 delta(int from, int to) const {
        double4 p = a.pos[from], q = a.pos[to];
        double dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
        if (l.periodic) min_image_d(dx, dy, dz, a.box);
        return make_double3(dx, dy, dz);
    }
    __device__ __forceinline__ void add(int atom, double fx, double fy, double fz) const {
        if (a.debugSkipAtomics) { if (fx == 12345.678) a.force[0] = 1; return; }
        add_force(a.force, a.paddedAtoms, a.slotOfAtom[atom], fx, fy, fz);
    }
};

__device__ __forceinline__ double dot3(double3 a, double3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double3 cross3(double3 a, double3 b) { return make_double3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

__device__ __forceinline__ double term_exception14(const TermCtx& c, int t) {
    const int i = c.l.atoms[2 * t], j = c.l.atoms[2 * t + 1];
    const double qq = c.l.params[3 * t], sig = c.l.params[3 * t + 1], eps4 = 4.0 * c.l.params[3 * t + 2];
    const double3 d = c.delta(i, j);                      // j - i
    const double invR = 1.0 / sqrt(dot3(d, d));
    double s2 = sig * invR; s2 *= s2;
    const double s6 = s2 * s2 * s2;
    const double dEdR = (eps4 * (12.0 * s6 - 6.0) * s6 + OMM_ONE_4PI_EPS0_D * qq * invR) * invR * invR;
    c.add(j, dEdR * d.x, dEdR * d.y, dEdR * d.z);
    c.add(i, -dEdR * d.x, -dEdR * d.y, -dEdR * d.z);
    return eps4 * (s6 - 1.0) * s6 + OMM_ONE_4PI_EPS0_D * qq * invR;
}

__device__ __forceinline__ double term_ewald_exclusion(const TermCtx& c, int t) {
    const int i = c.l.atoms[2 * t], j = c.l.atoms[2 * t + 1];
    const double qq = OMM_ONE_4PI_EPS0_D * c.l.charge[i] * c.l.charge[j];
    const double3 d = c.delta(i, j);
    // The separation is formed in double; erf/exp run in single precision like the direct-space kernel whose
    // contribution this term cancels (the double-precision libm versions cost ~10x more and dominate this launch).
    const float dx = (float) d.x, dy = (float) d.y, dz = (float) d.z;
    const float r2 = dx * dx + dy * dy + dz * dz;
    const float invR = rsqrtf(r2), r = r2 * invR;
    const float ar = (float) c.l.alpha * r;
    const float erfAr = erff(ar);
    if (erfAr > 1e-6f) {
        const double dEdR = qq * (double) (invR * invR * invR * (erfAr - 2.0f * ar * expf(-ar * ar) * 0.56418958354775628695f));
        c.add(j, -dEdR * d.x, -dEdR * d.y, -dEdR * d.z);
        c.add(i, dEdR * d.x, dEdR * d.y, dEdR * d.z);
        return -qq * (double) (invR * erfAr);
    }
    return -c.l.alpha * 1.12837916709551257390 * qq;
}

// LJPME: the dispersion grid also covers excluded pairs; this takes their share out again (ReferenceLJCoulombIxn.cpp:505-520).
// l.charge holds the per-atom C6 factors, l.alpha the dispersion alpha.
__device__ __forceinline__ double term_dispersion_exclusion(const TermCtx& c, int t) {
    const int i = c.l.atoms[2 * t], j = c.l.atoms[2 * t + 1];
    const double c6 = c.l.charge[i] * c.l.charge[j];
    const double3 d = c.delta(i, j);                      // j - i
    const double r2 = dot3(d, d), inv2 = 1.0 / r2, inv6 = inv2 * inv2 * inv2;
    const double x = c.l.alpha * c.l.alpha * r2, ex = exp(-x);
    const double coeff = 6.0 * c6 * inv6 * inv2 * (1.0 - ex * (1.0 + x + 0.5 * x * x + x * x * x / 6.0));
    c.add(j, coeff * d.x, coeff * d.y, coeff * d.z);
    c.add(i, -coeff * d.x, -coeff * d.y, -coeff * d.z);
    return c6 * inv6 * (1.0 - ex * (1.0 + x + 0.5 * x * x));
}

__device__ __forceinline__ double term_harmonic_bond(const TermCtx& c, int t) {
    const int i = c.l.atoms[2 * t], j = c.l.atoms[2 * t + 1];
    const double r0 = c.l.params[2 * t], k = c.l.params[2 * t + 1];
    const double3 d = c.delta(i, j);
    const double r = sqrt(dot3(d, d));
    const double dl = r - r0;
    const double dEdR = r > 0.0 ? k * dl / r : 0.0;
    c.add(i, dEdR * d.x, dEdR * d.y, dEdR * d.z);
    c.add(j, -dEdR * d.x, -dEdR * d.y, -dEdR * d.z);
    return 0.5 * k * dl * dl;
}

__device__ __forceinline__ double term_harmonic_angle(const TermCtx& c, int t) {
    const int ia = c.l.atoms[3 * t], ib = c.l.atoms[3 * t + 1], ic = c.l.atoms[3 * t + 2];
    const double theta0 = c.l.params[2 * t], k = c.l.params[2 * t + 1];
    const double3 d0 = c.delta(ia, ib);      // b - a
    const double3 d1 = c.delta(ic, ib);      // b - c
    const double3 p = cross3(d0, d1);
    double rp = sqrt(dot3(p, p));
    if (rp < 1.0e-06) rp = 1.0e-06;
    const double r20 = dot3(d0, d0), r21 = dot3(d1, d1);
    const double cosine = dot3(d0, d1) / sqrt(r20 * r21);
    const double angle = cosine >= 1.0 ? 0.0 : (cosine <= -1.0 ? 3.14159265358979323846 : acos(cosine));
    const double dth = angle - theta0;
    const double dEdR = k * dth;
    const double termA = dEdR / (r20 * rp), termC = -dEdR / (r21 * rp);
    double3 fa = cross3(d0, p), fc = cross3(d1, p);
    fa.x *= termA; fa.y *= termA; fa.z *= termA;
    fc.x *= termC; fc.y *= termC; fc.z *= termC;
    c.add(ia, fa.x, fa.y, fa.z);
    c.add(ic, fc.x, fc.y, fc.z);
    c.add(ib, -(fa.x + fc.x), -(fa.y + fc.y), -(fa.z + fc.z));
    return 0.5 * k * dth * dth;
}

__device__ __forceinline__ double term_periodic_torsion(const TermCtx& c, int t) {
    const int ia = c.l.atoms[4 * t], ib = c.l.atoms[4 * t + 1], ic = c.l.atoms[4 * t + 2], id = c.l.atoms[4 * t + 3];
    // params: OMMHIP_TORSION_SUBTERMS x (k, cos(phase), sin(phase), periodicity) -- every term on the same four atoms shares one
    // thread (force fields put up to four periodicities on a dihedral: one geometry, no same-address atomics between them);
    // unused sub-terms have k = 0.  The host takes the sine and cosine of the phase once.
    const double* par = c.l.params + 4 * OMMHIP_TORSION_SUBTERMS * t;
    const double3 v0 = c.delta(ib, ia);      // a - b
    const double3 v1 = c.delta(ib, ic);      // c - b
    const double3 v2 = c.delta(id, ic);      // c - d
    const double3 cp0 = cross3(v0, v1), cp1 = cross3(v1, v2);
    // The dihedral angle phi of ReferenceBondIxn.cpp:115-135 never has to be formed: with the plane normals cp0, cp1
    //   cos(phi) = cp0.cp1 / (|cp0| |cp1|),   sin(phi) = |v1| (v0.cp1) / (|cp0| |cp1|)   (sign as the reference's dot(v0, cp1) test),
    // cos(n phi), sin(n phi) follow by the angle-addition recurrence for the integer periodicity n, and the phase enters through
    // its own sine and cosine -- no inverse trigonometric function and no sin/cos call in double precision per term.
    const double n0 = dot3(cp0, cp0), n1 = dot3(cp1, cp1);
    const double invNorm = 1.0 / sqrt(n0 * n1);
    const double normBC = sqrt(dot3(v1, v1));
    const double cosPhi = dot3(cp0, cp1) * invNorm, sinPhi = normBC * dot3(v0, cp1) * invNorm;
    double dEdAngle = 0.0, energy = 0.0;
#pragma unroll
    for (int sub = 0; sub < OMMHIP_TORSION_SUBTERMS; sub++) {
        const double k = par[4 * sub];
        if (k == 0.0) continue;
        const double cosPhase = par[4 * sub + 1], sinPhase = par[4 * sub + 2];
        const int periodicity = (int) par[4 * sub + 3];
        double cn = 1.0, sn = 0.0;
        for (int i = 0; i < periodicity; i++) { const double cNext = cn * cosPhi - sn * sinPhi; sn = sn * cosPhi + cn * sinPhi; cn = cNext; }
        dEdAngle -= k * periodicity * (sn * cosPhase - cn * sinPhase);          // -k n sin(n phi - phase)
        energy += k * (1.0 + cn * cosPhase + sn * sinPhase);                    //  k (1 + cos(n phi - phase))
    }
    const double ff0 = (-dEdAngle * normBC) / n0;
    const double ff3 = (dEdAngle * normBC) / n1;
    const double ff1 = dot3(v0, v1) / dot3(v1, v1);
    const double ff2 = dot3(v2, v1) / dot3(v1, v1);
    const double3 f0 = make_double3(ff0 * cp0.x, ff0 * cp0.y, ff0 * cp0.z);
    const double3 f3 = make_double3(ff3 * cp1.x, ff3 * cp1.y, ff3 * cp1.z);
    const double3 s = make_double3(ff1 * f0.x - ff2 * f3.x, ff1 * f0.y - ff2 * f3.y, ff1 * f0.z - ff2 * f3.z);
    c.add(ia, f0.x, f0.y, f0.z);
    c.add(ib, -(f0.x - s.x), -(f0.y - s.y), -(f0.z - s.z));
    c.add(ic, -(f3.x + s.x), -(f3.y + s.y), -(f3.z + s.z));
    c.add(id, f3.x, f3.y, f3.z);
    return energy;
}

__device__ __forceinline__ void terms_body(const TermArgs& a, const int block, double (&partial)[4]) {
    // which list does this workgroup belong to?  (numLists <= OMMHIP_MAX_TERM_LISTS, wave-uniform)
    int li = 0;
#pragma unroll
    for (int i = 1; i < OMMHIP_MAX_TERM_LISTS; i++)
        if (i < a.numLists && block >= a.list[i].firstBlock) li = i;
    const TermList& l = a.list[li];
    const int t = (block - l.firstBlock) * 256 + threadIdx.x;
    double energy = 0;
    bool evaluate = t < l.numTerms, countEnergy = true;
    if (evaluate && l.ownSlot1 > l.ownSlot0) {
        const int perTerm = l.kind == OMMHIP_TERM_HARMONIC_ANGLE ? 3 : (l.kind == OMMHIP_TERM_PERIODIC_TORSION ? 4 : 2);
        bool any = false, all = true, handedUp = true;
        for (int k = 0; k < perTerm; k++) {
            const int slot = a.slotOfAtom[l.atoms[(size_t) t * perTerm + k]];
            const bool own = slot >= l.ownSlot0 && slot < l.ownSlot1;
            any = any || own;
            if (k == 0) countEnergy = own;
            if (l.halfShell) {
                const bool visible = own || (slot >= l.evalSlot0 && slot < l.evalSlot1);
                all = all && visible;
                // could the upper neighbour evaluate the term?  It holds my up section and its own atoms.
                if (own) handedUp = handedUp && slot >= l.upSlot0 && slot < l.upSlot1;
                else if (!visible) handedUp = handedUp && slot / l.slotsPerRank == (l.rank + 1) % l.ranks;
                else handedUp = false;                       // an atom of the lower neighbour in a term that also leaves my sight: three slabs
            }
        }
        evaluate = any;
        if (l.halfShell) {
            evaluate = any && all;
            countEnergy = evaluate;
            if (any && !all && !handedUp && l.errorFlags != nullptr) atomicOr(&l.errorFlags[1], 8);
        }
    }
    if (evaluate) {
        const TermCtx c(a, l);
        switch (l.kind) {
            case OMMHIP_TERM_EXCEPTION14: energy = term_exception14(c, t); break;
            case OMMHIP_TERM_EWALD_EXCLUSION: energy = term_ewald_exclusion(c, t); break;
            case OMMHIP_TERM_HARMONIC_BOND: energy = term_harmonic_bond(c, t); break;
            case OMMHIP_TERM_HARMONIC_ANGLE: energy = term_harmonic_angle(c, t); break;
            case OMMHIP_TERM_DISPERSION_EXCLUSION: energy = term_dispersion_exclusion(c, t); break;
            default: energy = term_periodic_torsion(c, t); break;
        }
    }
    if (a.includeEnergy) {
        if (!countEnergy) energy = 0;
        energy = wave_sum(energy);
        if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = energy;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&a.energyBuffer[block % a.energySlots], partial[0] + partial[1] + partial[2] + partial[3]);
    }
}

__global__ __launch_bounds__(256) void k_terms(TermArgs a) {
    __shared__ double partial[4];
    terms_body(a, blockIdx.x, partial);
}

// ------------------------------------------------------------------------------------------------
// Classic Ewald reciprocal sum (rectangular boxes; small systems / TestEwald).
// Pass 1: one workgroup per k-vector -> structure factor (cs, ss) and energy.
// Pass 2: one thread per atom sums over all k-vectors.
// k-vector enumeration order and the half-space convention follow ReferenceLJCoulombIxn.cpp:303-365.
// ------------------------------------------------------------------------------------------------
struct EwaldArgs {
    int numAtoms, paddedAtoms, kx, ky, kz, numK, includeEnergy, energySlots;
    double recipX, recipY, recipZ, alpha, coeff;     // coeff = ONE_4PI_EPS0*4*pi/V
    const double4* pos;
    const double* charge;
    const int* slotOfAtom;
    double2* structure;       // [numK] (cs, ss)
    omm_fixed* force;
    double* energyBuffer;
};

__device__ __forceinline__ bool ewald_kvec(const EwaldArgs& a, int index, int& rx, int& ry, int& rz) {
    // index enumerates rx in [0,kx), ry in (-ky,ky), rz in (-kz,kz); the half space keeps
    // (rx>0) or (rx==0 && ry>0) or (rx==0 && ry==0 && rz>0).
    const int ny = 2 * a.ky - 1, nz = 2 * a.kz - 1;
    rz = index % nz - (a.kz - 1);
    ry = (index / nz) % ny - (a.ky - 1);
    rx = index / (nz * ny);
    if (rx == 0 && (ry < 0 || (ry == 0 && rz <= 0))) return false;
    return true;
}

__global__ __launch_bounds__(256) void k_ewald_structure(EwaldArgs a) {
    __shared__ double pc[4], ps[4];
    int rx, ry, rz;
    const int kIndex = blockIdx.x;
    const bool valid = ewald_kvec(a, kIndex, rx, ry, rz);
    double cs = 0, ss = 0;
    if (valid) {
        const double kxv = rx * a.recipX, kyv = ry * a.recipY, kzv = rz * a.recipZ;
        for (int i = threadIdx.x; i < a.numAtoms; i += blockDim.x) {
            double4 p = a.pos[i];
            double ph = kxv * p.x + kyv * p.y + kzv * p.z;
            double s, c;
            sincos(ph, &s, &c);
            double q = a.charge[i];
            cs += q * c; ss += q * s;
        }
    }
    cs = wave_sum(cs); ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) { pc[threadIdx.x >> 6] = cs; ps[threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        cs = pc[0] + pc[1] + pc[2] + pc[3]; ss = ps[0] + ps[1] + ps[2] + ps[3];
        a.structure[kIndex] = make_double2(cs, ss);
        if (valid && a.includeEnergy) {
            const double kxv = rx * a.recipX, kyv = ry * a.recipY, kzv = rz * a.recipZ;
            const double k2 = kxv * kxv + kyv * kyv + kzv * kzv;
            const double ak = exp(-k2 / (4.0 * a.alpha * a.alpha)) / k2;
            atomicAdd(&a.energyBuffer[blockIdx.x % a.energySlots], a.coeff * ak * (cs * cs + ss * ss));
        }
    }
}

__global__ __launch_bounds__(256) void k_ewald_forces(EwaldArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.numAtoms) return;
    const double4 p = a.pos[i];
    const double q = a.charge[i];
    double fx = 0, fy = 0, fz = 0;
    for (int kIndex = 0; kIndex < a.numK; kIndex++) {
        int rx, ry, rz;
        if (!ewald_kvec(a, kIndex, rx, ry, rz)) continue;
        const double kxv = rx * a.recipX, kyv = ry * a.recipY, kzv = rz * a.recipZ;
        const double k2 = kxv * kxv + kyv * kyv + kzv * kzv;
        const double ak = exp(-k2 / (4.0 * a.alpha * a.alpha)) / k2;
        double s, c;
        sincos(kxv * p.x + kyv * p.y + kzv * p.z, &s, &c);
        const double2 sf = a.structure[kIndex];
        const double f = 2.0 * a.coeff * ak * q * (sf.x * s - sf.y * c);
        fx += f * kxv; fy += f * kyv; fz += f * kzv;
    }
    add_force(a.force, a.paddedAtoms, a.slotOfAtom[i], fx, fy, fz);
}

// ------------------------------------------------------------------------------------------------
// CMMotionRemover: two-stage momentum reduction, then subtraction.
// ------------------------------------------------------------------------------------------------
#define CM_BLOCKS 64
// Stage 1: CM_BLOCKS workgroups each reduce a strided slice of the atoms to (px, py, pz, mass).
__global__ __launch_bounds__(256) void k_cm_momentum(const double4* __restrict__ vel, int numAtoms, double* __restrict__ partial) {
    __shared__ double part[4][4];
    double px = 0, py = 0, pz = 0, m = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < numAtoms; i += 256 * CM_BLOCKS) {
        double4 v = vel[i];
        double mass = v.w == 0.0 ? 0.0 : 1.0 / v.w;
        px += mass * v.x; py += mass * v.y; pz += mass * v.z; m += mass;
    }
    px = wave_sum(px); py = wave_sum(py); pz = wave_sum(pz); m = wave_sum(m);
    if ((threadIdx.x & 63) == 0) { int w = threadIdx.x >> 6; part[w][0] = px; part[w][1] = py; part[w][2] = pz; part[w][3] = m; }
    __syncthreads();
    if (threadIdx.x < 4) partial[4 * blockIdx.x + threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

// Stage 2: every workgroup re-reduces the CM_BLOCKS partials (256 doubles) and subtracts the CM velocity.
__global__ __launch_bounds__(256) void k_cm_subtract(double4* __restrict__ vel, int numAtoms, const double* __restrict__ partial) {
    __shared__ double mom[4];
    if (threadIdx.x < 4) {
        double s = 0;
        for (int b = 0; b < CM_BLOCKS; b++) s += partial[4 * b + threadIdx.x];
        mom[threadIdx.x] = s;
    }
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numAtoms) return;
    double4 v = vel[i];
    if (v.w == 0.0) return;
    double invM = 1.0 / mom[3];
    v.x -= mom[0] * invM; v.y -= mom[1] * invM; v.z -= mom[2] * invM;
    vel[i] = v;
}

}  // namespace

// -> number of workgroups (0: nothing to do, -1: bad input)
static int make_term_args(TermArgs& a, int num_lists, const ommhip_term_batch* lists, const void* pos_d, const int* slot_of_atom_d, int padded_atoms,
                          const double box[6], long long* force_d, double* energy_buffer_d, int energy_slots, int include_energy) {
    if (num_lists > OMMHIP_MAX_TERM_LISTS) return -1;
    a.numLists = 0; a.paddedAtoms = padded_atoms; a.includeEnergy = include_energy; a.energySlots = energy_slots;
    { static const bool skip = getenv("OPENMM_HIP_DEBUG_SKIP_TERM_ATOMICS") != nullptr; a.debugSkipAtomics = skip ? 1 : 0; }
    a.box.ax = box[0]; a.box.bx = box[1]; a.box.by = box[2]; a.box.cx = box[3]; a.box.cy = box[4]; a.box.cz = box[5];
    a.pos = (const double4*) pos_d; a.slotOfAtom = slot_of_atom_d; a.force = force_d; a.energyBuffer = energy_buffer_d;
    int blocks = 0;
    for (int i = 0; i < num_lists; i++) {
        if (lists[i].terms.num_terms <= 0) continue;
        if (lists[i].kind < OMMHIP_TERM_EXCEPTION14 || lists[i].kind > OMMHIP_TERM_DISPERSION_EXCLUSION) return -1;
        TermList& l = a.list[a.numLists++];
        l.kind = lists[i].kind; l.numTerms = lists[i].terms.num_terms; l.periodic = lists[i].periodic; l.firstBlock = blocks; l.ownSlot0 = lists[i].own_slot0; l.ownSlot1 = lists[i].own_slot1;
        l.halfShell = lists[i].half_shell != 0 && lists[i].own_slot1 > lists[i].own_slot0 && lists[i].slots_per_rank > 0 && lists[i].ranks > 1 ? 1 : 0;
        l.evalSlot0 = lists[i].eval_slot0; l.evalSlot1 = lists[i].eval_slot1; l.upSlot0 = lists[i].up_slot0; l.upSlot1 = lists[i].up_slot1;
        l.rank = lists[i].rank; l.ranks = lists[i].ranks; l.slotsPerRank = lists[i].slots_per_rank; l.errorFlags = lists[i].error_flags;
        l.alpha = lists[i].alpha; l.atoms = lists[i].terms.atoms; l.params = lists[i].terms.params; l.charge = lists[i].charge;
        blocks += (l.numTerms + 255) / 256;
    }
    for (int i = a.numLists; i < OMMHIP_MAX_TERM_LISTS; i++) { a.list[i] = a.list[0]; a.list[i].numTerms = 0; a.list[i].firstBlock = blocks; }
    return blocks;
}

extern "C" int ommhip_term_forces_multi(int num_lists, const ommhip_term_batch* lists, const void* pos_d, const int* slot_of_atom_d, int padded_atoms,
                                        const double box[6], long long* force_d, double* energy_buffer_d, int energy_slots, int include_energy, void* stream) {
    TermArgs a;
    const int blocks = make_term_args(a, num_lists, lists, pos_d, slot_of_atom_d, padded_atoms, box, force_d, energy_buffer_d, energy_slots, include_energy);
    if (blocks < 0) return 1;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(k_terms, dim3(blocks), dim3(256), 0, (hipStream_t) stream, a);
    return (int) hipGetLastError();
}

extern "C" int ommhip_term_forces(int kind, const ommhip_term_list* terms, const void* pos_d, const int* slot_of_atom_d, int padded_atoms,
                                  const double box[6], int periodic, const double* charge_d, double alpha,
                                  long long* force_d, double* energy_buffer_d, int energy_slots, int include_energy, void* stream) {
    ommhip_term_batch b;
    b.kind = kind; b.terms = *terms; b.periodic = periodic; b.charge = charge_d; b.alpha = alpha;
    return ommhip_term_forces_multi(1, &b, pos_d, slot_of_atom_d, padded_atoms, box, force_d, energy_buffer_d, energy_slots, include_energy, stream);
}

extern "C" int ommhip_ewald_reciprocal(const void* pos_d, const double* charge_d, const int* slot_of_atom_d, int num_atoms, int padded_atoms,
                                       const double box[6], double alpha, int kmax_x, int kmax_y, int kmax_z, void* structure_d,
                                       long long* force_d, double* energy_buffer_d, int energy_slots, int include_energy, void* stream) {
    EwaldArgs a;
    a.numAtoms = num_atoms; a.paddedAtoms = padded_atoms; a.kx = kmax_x; a.ky = kmax_y; a.kz = kmax_z;
    a.numK = kmax_x * (2 * kmax_y - 1) * (2 * kmax_z - 1);
    a.includeEnergy = include_energy; a.energySlots = energy_slots;
    const double pi = 3.14159265358979323846;
    a.recipX = 2 * pi / box[0]; a.recipY = 2 * pi / box[2]; a.recipZ = 2 * pi / box[5];
    a.alpha = alpha; a.coeff = OMM_ONE_4PI_EPS0_D * 4 * pi / (box[0] * box[2] * box[5]);
    a.pos = (const double4*) pos_d; a.charge = charge_d; a.slotOfAtom = slot_of_atom_d;
    a.structure = (double2*) structure_d; a.force = force_d; a.energyBuffer = energy_buffer_d;
    hipStream_t st = (hipStream_t) stream;
    hipLaunchKernelGGL(k_ewald_structure, dim3(a.numK), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_ewald_forces, dim3((num_atoms + 255) / 256), dim3(256), 0, st, a);
    return (int) hipGetLastError();
}

extern "C" int ommhip_remove_cm_motion(void* vel_d, int num_atoms, double* scratch_d, void* stream) {
    // scratch_d must hold 4*64 doubles
    hipStream_t st = (hipStream_t) stream;
    hipLaunchKernelGGL(k_cm_momentum, dim3(CM_BLOCKS), dim3(256), 0, st, (const double4*) vel_d, num_atoms, scratch_d);
    hipLaunchKernelGGL(k_cm_subtract, dim3((num_atoms + 255) / 256), dim3(256), 0, st, (double4*) vel_d, num_atoms, (const double*) scratch_d);
    return (int) hipGetLastError();
}

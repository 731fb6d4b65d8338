// AMOEBA kernels of the OpenMM "HIP" platform (include/openmm_hip_amoeba.h): first native slice -- AmoebaVdwForce.
//
// Oracle: plugins/amoeba/platforms/reference/src/SimTKReference/AmoebaReferenceVdwForce.cpp (cited per function below).
// Double precision throughout: AMOEBA's validation numbers (plugins/amoeba/tests/TestAmoebaVdwForce.h, from Tinker) are held to
// 1e-4 and better, and this force is a small part of an AMOEBA step next to the multipoles.
//
// Formulation for wave64: one thread owns one interaction site i and walks through all sites j, 256 at a time staged in LDS
// (site, atom position, type, flag: 64 bytes per j; every lane reads the same j at the same time -- LDS broadcast).  Each pair is
// evaluated from both sides, so a thread only ever accumulates the force on its own site: no atomics in the loop, and the
// exclusion list of i -- ascending -- is consumed with one cursor while j ascends.  The site force is shared between the atom and
// its parent at the end (two fixed-point atomics per component).
#include "common.h"
#include "../../../include/openmm_hip_amoeba.h"
#include "amoeba_pairs.h"
#include <cstdlib>

using namespace omm;

namespace {

#define VDW_BLOCK 256
#define VDW_SPLIT 4            // lanes per atom in the list kernel (amoeba_multipole.hip: MP_SPLIT)

struct VdwArgs {
    int numAtoms, numTypes, paddedAtoms, alchemicalMethod, lennardJones, periodic, includeEnergy, energySlots;
    const int* parent; const double* reduction; const int* type;
    const double* sigma; const double* epsilon;
    const int* exclStart; const int* exclAtoms;
    const unsigned char* alchemical;
    double epsilonScale, softcore, cutoff2, taperCutoff, c3, c4, c5;
    BoxD box;
    const double4* pos;
    double4* reduced;
    const int* slotOfAtom;
    omm_fixed* force;
    double* energyBuffer;
    // scan positions: position g holds atom order[g] (the platform's slot order, -1: padding) or g itself
    const int* order; int numScan;
    const int* pairList; const int* pairCount; int listStride, listSubcap;      // pair lists (amoeba_pairs.h); nullptr: the scan over all atoms
    int mixed;                                                                  // the list kernel's pair arithmetic in float
    int listSkin;                                                               // the lists reach beyond the cutoff (Verlet skin): the pair kernel re-tests the atom distance
};

__device__ __forceinline__ int scan_atom(const VdwArgs& a, int g) { return g < a.numScan ? (a.order != nullptr ? a.order[g] : g) : -1; }

// AmoebaReferenceVdwForce::setReducedPositions (AmoebaReferenceVdwForce.cpp:173-189)
__global__ void k_vdw_reduce(VdwArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.numAtoms) return;
    const double4 x = a.pos[i];
    const double red = a.reduction[i];
    double4 r = x;
    if (red != 0.0) {
        const double4 p = a.pos[a.parent[i]];
        r.x = red * (x.x - p.x) + p.x; r.y = red * (x.y - p.y) + p.y; r.z = red * (x.z - p.z) + p.z;
    }
    a.reduced[i] = r;
}

// one pair, seen from site i: energy of the pair and dE/dr / r (the force on i is -that times (r_i - r_j))
// AmoebaReferenceVdwForce::calculatePairIxn (AmoebaReferenceVdwForce.cpp:106-171)
// T = float: the pair arithmetic of the list kernel in "mixed" mode (separation formed in double, sums kept in double)
template <class T>
__device__ __forceinline__ T vdw_pair_t(const VdwArgs& a, T r, T sigma, T epsilon, T softcore, T& dEdRoverR) {
    T energy, dEdR;
    if (a.lennardJones) {
        const T pp1 = sigma / r, pp2 = pp1 * pp1, pp3 = pp2 * pp1, pp6 = pp3 * pp3, pp12 = pp6 * pp6;
        energy = T(4) * epsilon * (pp12 - pp6);
        dEdR = T(-24) * epsilon * (T(2) * pp12 - pp6) / r;
    }
    else {
        const T dhal = T(0.07), ghal1 = T(1.12), dhal1 = T(1.07);
        const T rho = r / sigma, rho2 = rho * rho, rho6 = rho2 * rho2 * rho2;
        const T rhoplus = rho + dhal, rhodec2 = rhoplus * rhoplus, rhodec = rhodec2 * rhodec2 * rhodec2;
        const T s1 = T(1) / (softcore + rhodec * rhoplus);
        const T s2 = T(1) / (softcore + rho6 * rho + T(0.12));
        const T point72 = dhal1 * dhal1;
        const T t1 = dhal1 * point72 * point72 * point72 * s1;
        const T t2 = ghal1 * s2;
        const T t2min = t2 - T(2);
        const T dt1 = T(-7) * rhodec * t1 * s1;
        const T dt2 = T(-7) * rho6 * t2 * s2;
        energy = epsilon * t1 * t2min;
        dEdR = epsilon * (dt1 * t2min + t1 * dt2) / sigma;
    }
    if (a.periodic && r > (T) a.taperCutoff) {
        const T delta = r - (T) a.taperCutoff;
        const T taper = T(1) + delta * delta * delta * ((T) a.c3 + delta * ((T) a.c4 + delta * (T) a.c5));
        const T dtaper = delta * delta * (T(3) * (T) a.c3 + delta * (T(4) * (T) a.c4 + delta * T(5) * (T) a.c5));
        dEdR = energy * dtaper + dEdR * taper;
        energy *= taper;
    }
    dEdRoverR = dEdR / r;
    return energy;
}
__device__ __forceinline__ double vdw_pair(const VdwArgs& a, double r, double sigma, double epsilon, double softcore, double& dEdRoverR) {
    return vdw_pair_t<double>(a, r, sigma, epsilon, softcore, dEdRoverR);
}

__global__ __launch_bounds__(VDW_BLOCK) void k_vdw_pairs(VdwArgs a) {
    __shared__ double4 sSite[VDW_BLOCK];
    __shared__ double4 sAtom[VDW_BLOCK];
    __shared__ int sType[VDW_BLOCK];
    __shared__ unsigned char sAlch[VDW_BLOCK];
    __shared__ double sEnergy[VDW_BLOCK / 64];
    const int t = threadIdx.x;
    const int g = blockIdx.x * VDW_BLOCK + t, i = scan_atom(a, g);
    const bool active = i >= 0;
    const int ii = active ? i : 0;
    const double4 si = a.reduced[ii], xi = a.pos[ii];
    const int typeI = a.type[ii];
    const bool alchI = a.alchemical != nullptr && a.alchemical[ii] != 0;
    const int* exclPartner = a.exclAtoms;
    int cursor = a.exclStart[ii];
    const int exclEnd = a.exclStart[ii + 1];
    int nextExcl = cursor < exclEnd ? exclPartner[cursor] : 0x7fffffff;
    double fx = 0, fy = 0, fz = 0, energy = 0;
    for (int j0 = 0; j0 < a.numScan; j0 += VDW_BLOCK) {
        const int jl = scan_atom(a, j0 + t);
        __syncthreads();
        sType[t] = -1;
        if (jl >= 0) { sSite[t] = a.reduced[jl]; sAtom[t] = a.pos[jl]; sType[t] = a.type[jl]; sAlch[t] = a.alchemical != nullptr ? a.alchemical[jl] : 0; }
        __syncthreads();
        const int n = min(VDW_BLOCK, a.numScan - j0);
        if (!active) continue;
        for (int k = 0; k < n; k++) {
            const int j = j0 + k;
            // the exclusion list ascends, and so does j
            while (nextExcl < j) { cursor++; nextExcl = cursor < exclEnd ? exclPartner[cursor] : 0x7fffffff; }
            if (j == g || j == nextExcl || sType[k] < 0) continue;
            if (a.periodic) {
                // pair selection on the ATOM positions (the Reference's neighbour list is built on them)
                const double4 xj = sAtom[k];
                double dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
                min_image_d(dx, dy, dz, a.box);
                if (dx * dx + dy * dy + dz * dz > a.cutoff2) continue;
            }
            const double4 sj = sSite[k];
            double dx = si.x - sj.x, dy = si.y - sj.y, dz = si.z - sj.z;
            if (a.periodic) min_image_d(dx, dy, dz, a.box);
            const double r = sqrt(dx * dx + dy * dy + dz * dz);
            double sigma = a.sigma[typeI * a.numTypes + sType[k]], epsilon = a.epsilon[typeI * a.numTypes + sType[k]], softcore = 0.0;
            const bool alchJ = sAlch[k] != 0;
            if ((a.alchemicalMethod == 1 && alchI != alchJ) || (a.alchemicalMethod == 2 && (alchI || alchJ))) { epsilon *= a.epsilonScale; softcore = a.softcore; }
            double dEdRoverR;
            const double e = vdw_pair(a, r, sigma, epsilon, softcore, dEdRoverR);
            fx -= dEdRoverR * dx; fy -= dEdRoverR * dy; fz -= dEdRoverR * dz;
            energy += 0.5 * e;              // every pair is visited from both of its sites
        }
    }
    if (active) {
        // AmoebaReferenceVdwForce::addReducedForce (AmoebaReferenceVdwForce.cpp:92-104)
        const int p = a.parent[i];
        if (p == i) add_force(a.force, a.paddedAtoms, a.slotOfAtom[i], fx, fy, fz);
        else {
            const double red = a.reduction[i];
            add_force(a.force, a.paddedAtoms, a.slotOfAtom[i], fx * red, fy * red, fz * red);
            add_force(a.force, a.paddedAtoms, a.slotOfAtom[p], fx * (1.0 - red), fy * (1.0 - red), fz * (1.0 - red));
        }
    }
    if (a.includeEnergy) {
        energy = wave_sum(active ? energy : 0.0);
        if ((t & 63) == 0) sEnergy[t >> 6] = energy;
        __syncthreads();
        if (t == 0) {
            double e = 0;
            for (int w = 0; w < VDW_BLOCK / 64; w++) e += sEnergy[w];
            atomicAdd(&a.energyBuffer[blockIdx.x % a.energySlots], e);
        }
    }
}

// The same pair terms over the pair lists of amoeba_pairs.h (CutoffPeriodic): thread g owns the atom at scan position g and walks its own
// list -- partners within the cutoff by ATOM distance, exclusions left out by the builder -- so every iteration is a pair that counts.
__global__ __launch_bounds__(VDW_BLOCK) void k_vdw_pairs_list(VdwArgs a) {
    __shared__ double sEnergy[VDW_BLOCK / 64];
    const int t = threadIdx.x;
    const int g = (blockIdx.x * VDW_BLOCK + t) / VDW_SPLIT, q = t % VDW_SPLIT, i = scan_atom(a, g);      // VDW_SPLIT lanes share an atom's list
    const bool active = i >= 0;
    const int ii = active ? i : 0;
    const double4 si = a.reduced[ii], xi = a.pos[ii];
    const int typeI = a.type[ii];
    const bool alchI = a.alchemical != nullptr && a.alchemical[ii] != 0;
    double fx = 0, fy = 0, fz = 0, energy = 0;
    PlSpan span = {0, 0, 0, 0};
    if (active) span = pl_span(a.pairCount, a.listStride, g);
    for (int k = q; k < span.total; k += VDW_SPLIT) {
        const int j = scan_atom(a, pl_at(a.pairList, a.listStride, a.listSubcap, span, k, g) & PL_POS_MASK);
        const double4 sj = a.reduced[j];
        double dx = si.x - sj.x, dy = si.y - sj.y, dz = si.z - sj.z;
        min_image_d(dx, dy, dz, a.box);
        if (a.listSkin) {
            // a list with a skin: the pair counts when the ATOMS are within the cutoff (the rule the list was built by, amoeba_pairs.h)
            const double4 xj = a.pos[j];
            double ax = xi.x - xj.x, ay = xi.y - xj.y, az = xi.z - xj.z;
            min_image_d(ax, ay, az, a.box);
            if (ax * ax + ay * ay + az * az > a.cutoff2) continue;
        }
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        const int typeJ = a.type[j];
        double sigma = a.sigma[typeI * a.numTypes + typeJ], epsilon = a.epsilon[typeI * a.numTypes + typeJ], softcore = 0.0;
        const bool alchJ = a.alchemical != nullptr && a.alchemical[j] != 0;
        if ((a.alchemicalMethod == 1 && alchI != alchJ) || (a.alchemicalMethod == 2 && (alchI || alchJ))) { epsilon *= a.epsilonScale; softcore = a.softcore; }
        double dEdRoverR, e;
        if (a.mixed) { float d; e = (double) vdw_pair_t<float>(a, (float) r, (float) sigma, (float) epsilon, (float) softcore, d); dEdRoverR = (double) d; }
        else e = vdw_pair(a, r, sigma, epsilon, softcore, dEdRoverR);
        fx -= dEdRoverR * dx; fy -= dEdRoverR * dy; fz -= dEdRoverR * dz;
        energy += 0.5 * e;
    }
    fx += __shfl_xor(fx, 1); fy += __shfl_xor(fy, 1); fz += __shfl_xor(fz, 1); energy += __shfl_xor(energy, 1);
    fx += __shfl_xor(fx, 2); fy += __shfl_xor(fy, 2); fz += __shfl_xor(fz, 2); energy += __shfl_xor(energy, 2);
    if (q != 0) energy = 0.0;
    if (active && q == 0) {
        const int p = a.parent[i];
        if (p == i) add_force(a.force, a.paddedAtoms, a.slotOfAtom[i], fx, fy, fz);
        else {
            const double red = a.reduction[i];
            add_force(a.force, a.paddedAtoms, a.slotOfAtom[i], fx * red, fy * red, fz * red);
            add_force(a.force, a.paddedAtoms, a.slotOfAtom[p], fx * (1.0 - red), fy * (1.0 - red), fz * (1.0 - red));
        }
    }
    if (a.includeEnergy) {
        energy = wave_sum(active ? energy : 0.0);
        if ((t & 63) == 0) sEnergy[t >> 6] = energy;
        __syncthreads();
        if (t == 0) {
            double e = 0;
            for (int w = 0; w < VDW_BLOCK / 64; w++) e += sEnergy[w];
            atomicAdd(&a.energyBuffer[blockIdx.x % a.energySlots], e);
        }
    }
}

}  // namespace

extern "C" int ommhip_amoeba_vdw_forces(const ommhip_amoeba_vdw* v, const void* pos_d, const double box[6], const int* slot_of_atom_d, int padded_atoms,
                                        long long* force_d, double* energy_buffer_d, int energy_slots, int include_energy, void* stream) {
    if (v->num_atoms <= 0) return 0;
    if (v->reduced == nullptr || v->parent == nullptr || v->type == nullptr || v->excl_start == nullptr) return 1;
    VdwArgs a;
    a.mixed = 0; a.listSkin = 0;
    a.numAtoms = v->num_atoms; a.numTypes = v->num_types; a.paddedAtoms = padded_atoms; a.alchemicalMethod = v->alchemical_method;
    a.lennardJones = v->lennard_jones; a.periodic = v->periodic; a.includeEnergy = include_energy; a.energySlots = energy_slots;
    a.parent = v->parent; a.reduction = v->reduction; a.type = v->type; a.sigma = v->sigma; a.epsilon = v->epsilon;
    a.exclStart = v->excl_start; a.exclAtoms = v->excl_atoms; a.alchemical = v->alchemical;
    a.epsilonScale = v->epsilon_scale; a.softcore = v->softcore; a.cutoff2 = v->cutoff * v->cutoff; a.taperCutoff = v->taper_cutoff;
    a.c3 = v->taper_c3; a.c4 = v->taper_c4; a.c5 = v->taper_c5;
    a.box.ax = box[0]; a.box.bx = box[1]; a.box.by = box[2]; a.box.cx = box[3]; a.box.cy = box[4]; a.box.cz = box[5];
    a.pos = (const double4*) pos_d; a.reduced = (double4*) v->reduced; a.slotOfAtom = slot_of_atom_d;
    a.force = force_d; a.energyBuffer = energy_buffer_d;
    hipStream_t st = (hipStream_t) stream;
    const int blocks = (a.numAtoms + VDW_BLOCK - 1) / VDW_BLOCK;
    a.pairList = nullptr; a.pairCount = nullptr; a.listStride = 0;
    if (a.periodic && v->pair_list != nullptr && v->pair_count != nullptr && v->pair_overflow != nullptr && v->tile_bounds != nullptr && v->excl_pos != nullptr && v->pair_cap >= 4) {
        // CutoffPeriodic: pair lists (partners by atom distance, exclusions left out), then the pair terms over the lists
        hipLaunchKernelGGL(k_vdw_reduce, dim3(blocks), dim3(VDW_BLOCK), 0, st, a);
        a.order = nullptr; a.numScan = a.numAtoms;
        if (v->atom_of_slot != nullptr && padded_atoms >= a.numAtoms) { a.order = v->atom_of_slot; a.numScan = padded_atoms; }
        PairListArgs p;
        p.n = a.numAtoms; p.numScan = a.numScan; p.subcap = v->pair_cap / PL_PARTS; p.stride = a.numScan; p.excludeListed = 1;
        static const bool noTilesList = getenv("OPENMM_HIP_AMOEBA_NO_TILES") != nullptr;
        p.skipTiles = !noTilesList && box[1] == 0.0 && box[3] == 0.0 && box[4] == 0.0 ? 1 : 0;
        p.pos = a.pos; p.order = a.order; p.slotOfAtom = a.slotOfAtom; p.box = a.box; p.cutoff2 = a.cutoff2;
        const int tiles = (a.numScan + PL_BLOCK - 1) / PL_BLOCK;
        p.tileCenter = (double4*) v->tile_bounds; p.tileHalf = p.tileCenter + tiles;
        p.rowStart = a.exclStart; p.rowAtom = a.exclAtoms; p.rowPos = v->excl_pos; p.rowData = nullptr; p.rowDataIn = nullptr;
        p.list = v->pair_list; p.count = v->pair_count; p.overflow = v->pair_overflow;
        // Verlet skin: the list holds the partners within cutoff + skin and lives until an atom has moved by skin / 2; the pair kernel re-tests
        p.refPos = nullptr; p.state = nullptr; p.skinHalf2 = 0.0; p.forceRebuild = 1;
        a.listSkin = 0; a.mixed = v->mixed_precision != 0 ? 1 : 0;
        if (v->skin > 0.0 && v->ref_pos != nullptr && v->list_state != nullptr) {
            const double radius = v->cutoff + v->skin;
            p.cutoff2 = radius * radius; p.refPos = (double4*) v->ref_pos; p.state = v->list_state; p.skinHalf2 = 0.25 * v->skin * v->skin; p.forceRebuild = v->force_rebuild != 0;
            a.listSkin = 1;
        }
        const int rc = pl_launch(p, v->pair_needed, st, v->list_builds);
        if (rc != 0) return rc;
        a.pairList = v->pair_list; a.pairCount = v->pair_count; a.listStride = a.numScan; a.listSubcap = v->pair_cap / PL_PARTS;
        hipLaunchKernelGGL(k_vdw_pairs_list, dim3((unsigned) (((size_t) a.numScan * VDW_SPLIT + VDW_BLOCK - 1) / VDW_BLOCK)), dim3(VDW_BLOCK), 0, st, a);
        return (int) hipGetLastError();
    }
    // no lists (NoCutoff, or a caller without the work arrays): every thread scans all atoms, in atom order
    a.order = nullptr; a.numScan = a.numAtoms;
    hipLaunchKernelGGL(k_vdw_reduce, dim3(blocks), dim3(VDW_BLOCK), 0, st, a);
    hipLaunchKernelGGL(k_vdw_pairs, dim3((a.numScan + VDW_BLOCK - 1) / VDW_BLOCK), dim3(VDW_BLOCK), 0, st, a);
    return (int) hipGetLastError();
}

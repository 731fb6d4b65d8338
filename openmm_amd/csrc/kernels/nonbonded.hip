// Direct-space NonbondedForce for MI355X (gfx950): neighbour-list construction and the pair kernel.
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
#include "../../../include/openmm_hip_kernels.h"

using namespace omm;

namespace {

enum { ST_REBUILD = 0, ST_NUM_CHUNKS = 1, ST_OVERFLOW = 2, ST_BLOCKS_DONE = 3, ST_REBUILD_COUNT = 4 };

struct NlArgs {
    int numAtoms, paddedAtoms, numBlocks, maxChunks;
    int pbc;                 // 0 none, 1 orthorhombic, 2 triclinic
    float listCutoff2;       // (cutoff + padding)^2, +inf for NoCutoff
    float maxDisp2;          // (padding/2)^2
    Box box;
    const float4* posq;
    float4* posqRef;
    const int* atomOfSlot;
    const int* slotOfAtom;
    const int* exclStart;
    const int* exclAtoms;
    int* state;
    float4* blockCenter;
    float4* blockHalf;
    int2* chunkInfo;
    int* rowJ;
    unsigned* rowMask;
};

__device__ __forceinline__ void apply_pbc(int pbc, float& dx, float& dy, float& dz, const Box& b) {
    if (pbc == 1) min_image<false>(dx, dy, dz, b);
    else if (pbc == 2) min_image<true>(dx, dy, dz, b);
}

// ------------------------------------------------------------------------------------------------
// Per-step: did any atom move more than padding/2 since the list was built?  (one thread per slot)
// ------------------------------------------------------------------------------------------------
__global__ void nl_check_displacement(NlArgs a) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    bool moved = false;
    if (s < a.paddedAtoms && a.atomOfSlot[s] >= 0) {
        float4 p = a.posq[s], r = a.posqRef[s];
        float dx = p.x - r.x, dy = p.y - r.y, dz = p.z - r.z;
        apply_pbc(a.pbc, dx, dy, dz, a.box);
        moved = !(dx * dx + dy * dy + dz * dz <= a.maxDisp2);   // NaN counts as moved
    }
    if (__any(moved) && lane_id() == 0) atomicOr(&a.state[ST_REBUILD], 1);
}

// ------------------------------------------------------------------------------------------------
// Bounding boxes of the 32-atom blocks (two blocks per wavefront); also snapshots posq -> posqRef.
// ------------------------------------------------------------------------------------------------
__global__ void nl_block_bounds(NlArgs a) {
    if (a.state[ST_REBUILD] == 0) return;
    int s = blockIdx.x * blockDim.x + threadIdx.x;     // slot
    if (s == 0) { a.state[ST_NUM_CHUNKS] = 0; a.state[ST_OVERFLOW] = 0; }
    bool inRange = s < a.paddedAtoms;
    int sl = inRange ? s : a.paddedAtoms - 1;
    float4 p = a.posq[sl];
    bool valid = inRange && a.atomOfSlot[sl] >= 0;
    if (inRange) a.posqRef[sl] = p;
    // first atom of the block (always valid: every block holds at least one real atom)
    float4 p0 = make_float4(__shfl(p.x, 0, 32), __shfl(p.y, 0, 32), __shfl(p.z, 0, 32), 0.f);
    float dx = p.x - p0.x, dy = p.y - p0.y, dz = p.z - p0.z;
    apply_pbc(a.pbc, dx, dy, dz, a.box);
    if (!valid) { dx = dy = dz = 0; }
    float minx = dx, maxx = dx, miny = dy, maxy = dy, minz = dz, maxz = dz;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        minx = fminf(minx, __shfl_xor(minx, m)); maxx = fmaxf(maxx, __shfl_xor(maxx, m));
        miny = fminf(miny, __shfl_xor(miny, m)); maxy = fmaxf(maxy, __shfl_xor(maxy, m));
        minz = fminf(minz, __shfl_xor(minz, m)); maxz = fmaxf(maxz, __shfl_xor(maxz, m));
    }
    if (inRange && (s & 31) == 0) {
        int blk = s >> 5;
        a.blockCenter[blk] = make_float4(p0.x + 0.5f * (minx + maxx), p0.y + 0.5f * (miny + maxy), p0.z + 0.5f * (minz + maxz), 0.f);
        a.blockHalf[blk] = make_float4(0.5f * (maxx - minx), 0.5f * (maxy - miny), 0.5f * (maxz - minz), 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// Neighbour-list rows for i-block X = blockIdx.x (one wavefront per i-block).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void nl_find_interactions(NlArgs a) {
    if (a.state[ST_REBUILD] == 0) return;
    __shared__ float4 xPos[OMM_TILE];
    __shared__ int stageJ[2 * OMM_ROW];
    __shared__ unsigned stageM[2 * OMM_ROW];
    const int lane = threadIdx.x;
    const int X = blockIdx.x;
    const float R2 = a.listCutoff2;
    const float Rlist = sqrtf(R2);

    if (lane < OMM_TILE) xPos[lane] = a.posq[X * OMM_TILE + lane];
    bool iValid = lane < OMM_TILE && a.atomOfSlot[X * OMM_TILE + lane] >= 0;
    const unsigned iValidMask = (unsigned) __ballot(iValid);
    const float4 cX = a.blockCenter[X], hX = a.blockHalf[X];
    __syncthreads();

    int count = 0;           // staged entries (wave-uniform)
    int chunk = -1, rowsInChunk = 0, maskedBits = 0;

    auto flushRow = [&](int nvalid) {
        // Writes stage[0..64) as one row (entries >= nvalid are padding), then shifts the stage down.
        if (rowsInChunk == 0) {
            int c = 0;
            if (lane == 0) c = atomicAdd(&a.state[ST_NUM_CHUNKS], 1);
            chunk = __shfl(c, 0);
            maskedBits = 0;
        }
        int j = lane < nvalid ? stageJ[lane] : X * OMM_TILE;
        unsigned m = lane < nvalid ? stageM[lane] : 0u;
        bool masked = __any(m != 0xFFFFFFFFu);
        if (chunk < a.maxChunks) {
            size_t r = ((size_t) chunk * OMM_CHUNK_ROWS + rowsInChunk) * OMM_ROW + lane;
            a.rowJ[r] = j;
            a.rowMask[r] = m;
        }
        else if (lane == 0) atomicOr(&a.state[ST_OVERFLOW], 1);
        if (masked) maskedBits |= 1 << rowsInChunk;
        rowsInChunk++;
        if (rowsInChunk == OMM_CHUNK_ROWS) {
            if (lane == 0 && chunk < a.maxChunks) a.chunkInfo[chunk] = make_int2(X, rowsInChunk | (maskedBits << 8));
            rowsInChunk = 0;
        }
        __syncthreads();
        int j2 = stageJ[OMM_ROW + lane];
        unsigned m2 = stageM[OMM_ROW + lane];
        __syncthreads();
        stageJ[lane] = j2;
        stageM[lane] = m2;
        __syncthreads();
    };

    for (int ybase = X; ybase < a.numBlocks; ybase += 64) {
        // ---- block-level test: 64 candidate blocks at a time
        int Y = ybase + lane;
        bool cand = false;
        if (Y < a.numBlocks) {
            float4 cY = a.blockCenter[Y], hY = a.blockHalf[Y];
            float dx = cY.x - cX.x, dy = cY.y - cX.y, dz = cY.z - cX.z;
            apply_pbc(a.pbc, dx, dy, dz, a.box);
            dx = fmaxf(0.f, fabsf(dx) - hX.x - hY.x);
            dy = fmaxf(0.f, fabsf(dy) - hX.y - hY.y);
            dz = fmaxf(0.f, fabsf(dz) - hX.z - hY.z);
            cand = !(dx * dx + dy * dy + dz * dz >= R2);
            // Triclinic: the sequential image reduction only finds the nearest copy when it is less than
            // half a box width away; if that cannot be guaranteed, defer to the exact per-atom test.
            if (a.pbc == 2 && (0.5f * a.box.cz - hX.z - hY.z < Rlist || 0.5f * a.box.by - hX.y - hY.y < Rlist)) cand = true;
        }
        unsigned long long cm = __ballot(cand);
        // ---- atom-level test: two candidate blocks per pass (lanes 0-31 / 32-63)
        while (cm != 0) {
            int y0 = __ffsll((long long) cm) - 1; cm &= cm - 1;
            int y1 = -1;
            if (cm != 0) { y1 = __ffsll((long long) cm) - 1; cm &= cm - 1; }
            int ysel = lane < 32 ? y0 : y1;
            int Yc = ybase + ysel;
            int lj = lane & 31;
            int j = Yc * OMM_TILE + lj;
            bool ok = ysel >= 0;
            int atomJ = ok ? a.atomOfSlot[j] : -1;
            ok = ok && atomJ >= 0;
            unsigned mask = 0;
            if (ok) {
                float4 pj = a.posq[j];
                // distance to X's bounding box
                float dx = pj.x - cX.x, dy = pj.y - cX.y, dz = pj.z - cX.z;
                apply_pbc(a.pbc, dx, dy, dz, a.box);
                float bx = fmaxf(0.f, fabsf(dx) - hX.x), by = fmaxf(0.f, fabsf(dy) - hX.y), bz = fmaxf(0.f, fabsf(dz) - hX.z);
                bool near = !(bx * bx + by * by + bz * bz >= R2);
                if (a.pbc == 2 && (0.5f * a.box.cz - hX.z < Rlist || 0.5f * a.box.by - hX.y < Rlist)) near = true;
                if (near) {
                    // exact test against the 32 atoms of X, same metric as the pair kernel
                    bool any = false;
                    for (int k = 0; k < OMM_TILE; k++) {
                        float4 pi = xPos[k];
                        float ex = pj.x - pi.x, ey = pj.y - pi.y, ez = pj.z - pi.z;
                        apply_pbc(a.pbc, ex, ey, ez, a.box);
                        any = any || !(ex * ex + ey * ey + ez * ez >= R2);
                    }
                    if (any) {
                        mask = iValidMask;
                        if (Yc == X) mask &= (1u << lj) - 1u;        // diagonal block: each pair once, no self pair
                        for (int e = a.exclStart[atomJ]; e < a.exclStart[atomJ + 1]; e++) {
                            int s = a.slotOfAtom[a.exclAtoms[e]];
                            if ((s >> 5) == X) mask &= ~(1u << (s & 31));
                        }
                    }
                }
            }
            bool pass = mask != 0;
            unsigned long long pm = __ballot(pass);
            if (pass) {
                int pos = count + lane_prefix_count(pm);
                stageJ[pos] = j;
                stageM[pos] = mask;
            }
            count += __popcll(pm);
            __syncthreads();
            if (count >= OMM_ROW) {
                flushRow(OMM_ROW);
                count -= OMM_ROW;
            }
        }
    }
    if (count > 0) flushRow(count);
    if (rowsInChunk > 0 && lane == 0 && chunk < a.maxChunks)
        a.chunkInfo[chunk] = make_int2(X, rowsInChunk | (maskedBits << 8));

    // Last wave out clears the rebuild request.
    __syncthreads();
    if (lane == 0) {
        __threadfence();
        int done = atomicAdd(&a.state[ST_BLOCKS_DONE], 1);
        if (done == (int) gridDim.x - 1) {
            a.state[ST_BLOCKS_DONE] = 0;
            a.state[ST_REBUILD] = 0;
            atomicAdd(&a.state[ST_REBUILD_COUNT], 1);
        }
    }
}

// ================================================================================================
// Pair kernel
// ================================================================================================
struct NbArgs {
    int paddedAtoms, maxChunks;
    float cutoff2, alpha, krf, crf, switchDist, invSwitchWidth;
    Box box;
    const float4* posq;
    const float2* sigEps;
    const int* state;
    const int2* chunkInfo;
    const int* rowJ;
    const unsigned* rowMask;
    omm_fixed* force;
    double* energyBuffer;     // one slot per workgroup
};

template <int METHOD, int PBC, bool ENERGY, bool MASKED>
__device__ __forceinline__ void pair_ixn(const NbArgs& a, const float4 pi, const float2 sei, const float4 pj, const float2 sej, const float qjK,
                                         bool bit, float& fix, float& fiy, float& fiz, float& fjx, float& fjy, float& fjz, float& energy) {
    float dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;
    if (PBC == 1) min_image<false>(dx, dy, dz, a.box);
    if (PBC == 2) min_image<true>(dx, dy, dz, a.box);
    const float r2 = dx * dx + dy * dy + dz * dz;
    bool in = r2 < a.cutoff2;
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
    float dEdR = (ljF + cF) * invR2;
    dEdR = in ? dEdR : 0.f;
    fjx += dEdR * dx; fjy += dEdR * dy; fjz += dEdR * dz;
    fix -= dEdR * dx; fiy -= dEdR * dy; fiz -= dEdR * dz;
    if (ENERGY) energy += in ? (ljE + cE) : 0.f;
}

// Transpose-reduce: on entry every lane holds 32 partial sums v[0..32); on exit lane l holds the
// wave-wide total of v[l & 31].  63 cross-lane moves instead of 32*6.
__device__ __forceinline__ float transpose_reduce32(float (&v)[OMM_TILE], int lane) {
#pragma unroll
    for (int k = 0; k < 32; k++) v[k] += __shfl_xor(v[k], 32);
#pragma unroll
    for (int k = 0; k < 16; k++) { bool up = lane & 16; float send = up ? v[k] : v[k + 16]; float keep = up ? v[k + 16] : v[k]; v[k] = keep + __shfl_xor(send, 16); }
#pragma unroll
    for (int k = 0; k < 8; k++) { bool up = lane & 8; float send = up ? v[k] : v[k + 8]; float keep = up ? v[k + 8] : v[k]; v[k] = keep + __shfl_xor(send, 8); }
#pragma unroll
    for (int k = 0; k < 4; k++) { bool up = lane & 4; float send = up ? v[k] : v[k + 4]; float keep = up ? v[k + 4] : v[k]; v[k] = keep + __shfl_xor(send, 4); }
#pragma unroll
    for (int k = 0; k < 2; k++) { bool up = lane & 2; float send = up ? v[k] : v[k + 2]; float keep = up ? v[k + 2] : v[k]; v[k] = keep + __shfl_xor(send, 2); }
    { bool up = lane & 1; float send = up ? v[0] : v[1]; float keep = up ? v[1] : v[0]; v[0] = keep + __shfl_xor(send, 1); }
    return v[0];
}

template <int METHOD, int PBC, bool ENERGY>
__global__ __launch_bounds__(64) void nb_direct(NbArgs a, const float4* __restrict__ posqI, const float2* __restrict__ sigEpsI) {
    // posqI/sigEpsI alias a.posq/a.sigEps; passing them as separate __restrict__ kernel arguments
    // lets the compiler prove they are never written here and fetch the wave-uniform i-atom data
    // with scalar loads.
    const int lane = threadIdx.x;
    int numChunks = a.state[ST_NUM_CHUNKS];
    if (numChunks > a.maxChunks) numChunks = a.maxChunks;
    double energyTotal = 0;
    for (int c = blockIdx.x; c < numChunks; c += gridDim.x) {
        const int2 info = a.chunkInfo[c];
        const int X = __builtin_amdgcn_readfirstlane(info.x);
        const int nrows = __builtin_amdgcn_readfirstlane(info.y & 0xff);
        const int maskedBits = __builtin_amdgcn_readfirstlane(info.y >> 8);
        const float4* __restrict__ ip = posqI + X * OMM_TILE;
        const float2* __restrict__ ise = sigEpsI + X * OMM_TILE;
        float fix[OMM_TILE], fiy[OMM_TILE], fiz[OMM_TILE];
#pragma unroll
        for (int k = 0; k < OMM_TILE; k++) { fix[k] = 0.f; fiy[k] = 0.f; fiz[k] = 0.f; }
        float energy = 0.f;
        for (int row = 0; row < nrows; row++) {
            const size_t r = ((size_t) c * OMM_CHUNK_ROWS + row) * OMM_ROW + lane;
            const int j = a.rowJ[r];
            const float4 pj = a.posq[j];
            const float2 sej = a.sigEps[j];
            const float qjK = OMM_ONE_4PI_EPS0 * pj.w;
            float fjx = 0.f, fjy = 0.f, fjz = 0.f;
            if ((maskedBits >> row) & 1) {
                const unsigned m = a.rowMask[r];
#pragma unroll
                for (int k = 0; k < OMM_TILE; k++)
                    pair_ixn<METHOD, PBC, ENERGY, true>(a, ip[k], ise[k], pj, sej, qjK, (m >> k) & 1u, fix[k], fiy[k], fiz[k], fjx, fjy, fjz, energy);
            }
            else {
#pragma unroll
                for (int k = 0; k < OMM_TILE; k++)
                    pair_ixn<METHOD, PBC, ENERGY, false>(a, ip[k], ise[k], pj, sej, qjK, true, fix[k], fiy[k], fiz[k], fjx, fjy, fjz, energy);
            }
            add_force(a.force, a.paddedAtoms, j, fjx, fjy, fjz);
        }
        const float tx = transpose_reduce32(fix, lane);
        const float ty = transpose_reduce32(fiy, lane);
        const float tz = transpose_reduce32(fiz, lane);
        if (lane < OMM_TILE) add_force(a.force, a.paddedAtoms, X * OMM_TILE + lane, tx, ty, tz);
        if (ENERGY) energyTotal += (double) energy;
    }
    if (ENERGY) {
        energyTotal = wave_sum(energyTotal);
        if (lane == 0) a.energyBuffer[blockIdx.x] += energyTotal;
    }
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

Box make_box(const double* bv) {
    // bv = {ax, bx, by, cx, cy, cz}
    Box b;
    b.ax = (float) bv[0]; b.bx = (float) bv[1]; b.by = (float) bv[2]; b.cx = (float) bv[3]; b.cy = (float) bv[4]; b.cz = (float) bv[5];
    b.invAx = (float) (1.0 / bv[0]); b.invBy = (float) (1.0 / bv[2]); b.invCz = (float) (1.0 / bv[5]);
    return b;
}

NlArgs make_nl_args(const ommhip_neighbor_list* nl) {
    NlArgs a;
    a.numAtoms = nl->num_atoms; a.paddedAtoms = nl->padded_atoms; a.numBlocks = nl->padded_atoms / OMM_TILE; a.maxChunks = nl->max_chunks;
    a.pbc = nl->pbc;
    double rl = nl->cutoff + nl->padding;
    a.listCutoff2 = nl->cutoff > 0 ? (float) (rl * rl) : INFINITY;
    a.maxDisp2 = (float) (0.25 * nl->padding * nl->padding);
    a.box = make_box(nl->box);
    a.posq = (const float4*) nl->posq; a.posqRef = (float4*) nl->posq_ref;
    a.atomOfSlot = nl->atom_of_slot; a.slotOfAtom = nl->slot_of_atom;
    a.exclStart = nl->excl_start; a.exclAtoms = nl->excl_atoms;
    a.state = nl->state;
    a.blockCenter = (float4*) nl->block_center; a.blockHalf = (float4*) nl->block_half;
    a.chunkInfo = (int2*) nl->chunk_info; a.rowJ = nl->row_j; a.rowMask = nl->row_mask;
    return a;
}

}  // namespace

extern "C" int ommhip_nl_update(const ommhip_neighbor_list* nl, void* stream) {
    hipStream_t st = (hipStream_t) stream;
    NlArgs a = make_nl_args(nl);
    ommhip_profile_begin(OMMHIP_TIMER_NL_UPDATE, stream);
    if (nl->cutoff > 0)    // NoCutoff lists never go stale through motion
        hipLaunchKernelGGL(nl_check_displacement, dim3((a.paddedAtoms + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(nl_block_bounds, dim3((a.paddedAtoms + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(nl_find_interactions, dim3(a.numBlocks), dim3(64), 0, st, a);
    ommhip_profile_end(OMMHIP_TIMER_NL_UPDATE, stream);
    return (int) hipGetLastError();
}

extern "C" int ommhip_nb_direct(const ommhip_neighbor_list* nl, const ommhip_nonbonded_params* p, const void* sig_eps,
                                long long* force, double* energy_buffer, int energy_slots, int include_energy, void* stream) {
    NbArgs a;
    a.paddedAtoms = nl->padded_atoms; a.maxChunks = nl->max_chunks;
    a.cutoff2 = nl->cutoff > 0 ? (float) (nl->cutoff * nl->cutoff) : INFINITY;
    a.alpha = (float) p->ewald_alpha; a.krf = (float) p->krf; a.crf = (float) p->crf;
    a.switchDist = (float) p->switch_distance;
    a.invSwitchWidth = p->use_switch ? (float) (1.0 / (nl->cutoff - p->switch_distance)) : 0.f;
    a.box = make_box(nl->box);
    a.posq = (const float4*) nl->posq; a.sigEps = (const float2*) sig_eps; a.state = nl->state;
    a.chunkInfo = (const int2*) nl->chunk_info; a.rowJ = nl->row_j; a.rowMask = nl->row_mask;
    a.force = force; a.energyBuffer = energy_buffer;
    int grid = p->direct_grid > 0 ? p->direct_grid : 2048;
    if (include_energy && grid > energy_slots) grid = energy_slots;
    hipStream_t st = (hipStream_t) stream;
    ommhip_profile_begin(OMMHIP_TIMER_NB_DIRECT, stream);
    switch ((p->ewald ? 1 : 0) | (p->use_switch ? 2 : 0)) {
        case 0: launch_direct1<0>(nl->pbc, include_energy != 0, grid, st, a); break;
        case 1: launch_direct1<1>(nl->pbc, include_energy != 0, grid, st, a); break;
        case 2: launch_direct1<2>(nl->pbc, include_energy != 0, grid, st, a); break;
        default: launch_direct1<3>(nl->pbc, include_energy != 0, grid, st, a); break;
    }
    ommhip_profile_end(OMMHIP_TIMER_NB_DIRECT, stream);
    return (int) hipGetLastError();
}

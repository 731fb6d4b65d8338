// Neighbour-list construction for the direct-space kernel (MI355X / gfx950, wave64).
//
// Replaces (behaviourally) computeNeighborListVoxelHash -- platforms/reference/src/SimTKReference/ReferenceNeighborList.cpp:221-259 --
// which the Reference platform runs on every force evaluation.  Here the list is built with a padding and only
// rebuilt when an atom has moved more than padding/2, and the decision is taken ON THE DEVICE: the three kernels of
// ommhip_nl_update() are enqueued every step and return immediately unless state[ST_REBUILD] is set, so the step
// loop has no host round trip.
//
// Output format (consumed by nonbonded.hip): ROWS of 64 individually selected j-atoms for one 32-atom i-block X,
// each entry = j slot + a 32-bit mask of the i-atoms that interact with it (diagonal half, exclusions and padding
// atoms are folded into the mask); rows are grouped in CHUNKS of up to 4 rows of the same X.
//
// One workgroup of 4 wavefronts builds all rows of one i-block:
//   phase 1  block-level test, 64 candidate blocks per wave-instruction, survivors appended to an LDS list;
//   phase 2  atom-level test, two candidate blocks (64 atoms) per wavefront pass, four passes in flight; the
//            32 atoms of X are broadcast from registers with v_readlane (no LDS or memory traffic in the loop);
//            survivors (j, mask) are appended to an LDS list with one LDS atomic per wavefront;
//   flush    the LDS list is cut into 64-entry rows and written out; chunk indices come from one global atomic.
#include "common.h"
#include "../../../include/openmm_hip_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace omm;

namespace {

#ifndef NL_THREADS
#define NL_THREADS 256
#endif
#define NL_WAVES (NL_THREADS / 64)
#ifndef NL_LIST
#define NL_LIST 2048          // staged (j, mask) entries
#endif
#ifndef NL_FLUSH
#define NL_FLUSH 1024         // flush full rows once this many entries are staged
#endif
#ifndef NL_CAND
#define NL_CAND 1024          // candidate blocks per window
#endif
#ifndef NL_ROUND
#define NL_ROUND 4            // passes a wavefront runs between two workgroup-wide flush checks (NL_WAVES x NL_ROUND x 64 <= NL_LIST - NL_FLUSH)
#endif
#define NL_MAX_RUNS 8         // flushes of one i-block the pruning pass keeps track of (more: the pair kernel walks the unpruned rows)
#ifndef NL_BIG_HALF
#define NL_BIG_HALF 1.0       // blocks with a half extent above this fraction of the list cutoff are not binned
#endif
static_assert(NL_WAVES * NL_ROUND * 64 <= NL_LIST - NL_FLUSH, "a round of passes must fit behind the flush threshold");

struct NlArgs {
    int numAtoms, paddedAtoms, numBlocks, maxChunks;
    int firstBlock, ownedBlocks;      // the list is built for the i-blocks [firstBlock, firstBlock + ownedBlocks)
    int ddMode;                       // domain decomposition: partners are Y >= X plus the foreign blocks below firstBlock
    int ddHalfShell, evalBlock0, evalBlock1;   // ... or (half-shell evaluation): own blocks Y >= X plus the blocks [evalBlock0, evalBlock1) of the lower neighbour's section
    const uint4* posWire;             // DD: all positions as fixed-point box fractions, slot order (the all-gathered buffer)
    double4* posScatter;              // DD: atom-ordered positions, refreshed for foreign slots by nl_prepare
    long long* trace;                 // profiling (OPENMM_HIP_NL_TRACE): per workgroup start / end clock of a rebuild
    int xcdAware;                     // resident builder workgroups take the i-blocks of their XCD's eighth of the slot order
    int pbc;                 // 0 none, 1 orthorhombic, 2 triclinic
    float listCutoff2;       // (cutoff + padding)^2, +inf for NoCutoff
    float maxDisp2;          // (padding/2)^2
    Box box;
    const float4* posq;
    float4* posqRef;
    float4* posqRel;         // block-relative coordinates (position minus blockCenter of its block) + charge, or null
    // halo mode: the slot ranges with current wire records (own + the neighbours' sections); numActive = 0: all slots
    int numActive, activeBegin[4], activeEnd[4], activeTotal;
    const uint4* wireRef; const unsigned char* ddGuardAtom; unsigned ddWarn, ddMax; int* ddFlags; int ddRanks, ddSlotsPerRank, ddTrailerSlot;
    float4* posqRelLo;       // what the float rounding of posqRel left of the double-precision value (pair kernel's cutoff-edge path), or null
    const int* atomOfSlot;
    const int* slotOfAtom;
    const int* exclStart;
    const int* exclAtoms;
    const int2* exclBlockRange;   // per i-block: [lowest, highest] block holding an exclusion partner of its atoms (or null)
    const int* exclSlotStart;     // slot-keyed exclusion CSR (or null)
    const int* exclSlots;
    int* state;
    float4* blockCenter;
    float4* blockHalf;
    int2* chunkInfo;
    int* rowJ;
    unsigned* rowMask;
    // the pruned list (null: none) -- see ommhip_neighbor_list::chunk_info_inner
    int2* chunkInfoInner; int* rowJInner; unsigned* rowMaskInner; int* blockRuns;
    float cutoff, pruneCutoff2;       // the cutoff itself; (cutoff + inner padding)^2 with the margin of the prune test
    float4* posqRefInner; float maxDispInner2;      // positions at the last cut, (inner padding / 2)^2
    // cell-binned candidate search (large rectangular systems): blocks bucketed by the grid cell of their centre
    int cellMode, ncx, ncy, ncz;
    float cellInvX, cellInvY, cellInvZ;   // cells per nm
    int* cellStart;          // [ncells + 1] CSR over cells, followed by [ncells] fill cursors
    int* cellBlocks;         // [numBlocks] block indices sorted by cell, then [numBlocks] the list of oversized blocks
    float bigHalf;           // blocks with a half extent above this are not binned: every block tests them directly
    float4* cellBoxes;       // [2 * numBlocks] (centre, half extent) of those blocks, in the same order
    float* cellMeta;         // [0..2] largest half extent per axis among the binned blocks, [3] number of oversized blocks (int bits)
};

// g-th slot of the active ranges (halo mode), or g itself; false beyond the last one
__device__ __forceinline__ bool active_slot(const NlArgs& a, int g, int& s) {
    if (a.numActive == 0) { s = g; return g < a.paddedAtoms; }
    s = 0;
    bool found = false;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int len = r < a.numActive ? a.activeEnd[r] - a.activeBegin[r] : 0;
        if (!found && g < len) { s = a.activeBegin[r] + g; found = true; }
        g -= len;
    }
    return found;
}
// ... and the g-th BLOCK of the active ranges (they are whole numbers of blocks): the blocks a decomposed rank has positions for
__device__ __forceinline__ bool active_block(const NlArgs& a, int g, int& b) {
    if (a.numActive == 0) { b = g; return g < a.numBlocks; }
    int s;
    const bool ok = active_slot(a, g * OMM_TILE, s);
    b = s / OMM_TILE;
    return ok;
}

template <int PBC>
__device__ __forceinline__ void apply_pbc(float& dx, float& dy, float& dz, const Box& b) {
    if (PBC == 1) min_image<false>(dx, dy, dz, b);
    if (PBC == 2) min_image<true>(dx, dy, dz, b);
}
__device__ __forceinline__ void apply_pbc_rt(int pbc, float& dx, float& dy, float& dz, const Box& b) {
    if (pbc == 1) min_image<false>(dx, dy, dz, b);
    else if (pbc == 2) min_image<true>(dx, dy, dz, b);
}

// the same question for the pruned list: more than inner padding / 2 since it was last cut?
__device__ __forceinline__ void check_inner_displacement(const NlArgs& a, int s, bool valid, float4 p) {
    if (a.posqRefInner == nullptr) return;
    bool moved = false;
    if (valid) {
        const float4 r = a.posqRefInner[s];
        float dx = p.x - r.x, dy = p.y - r.y, dz = p.z - r.z;
        apply_pbc_rt(a.pbc, dx, dy, dz, a.box);
        moved = !(dx * dx + dy * dy + dz * dz <= a.maxDispInner2);
    }
    if (__any(moved) && lane_id() == 0) atomicOr(&a.state[ST_PRUNE_REQUEST], 1);
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
        apply_pbc_rt(a.pbc, dx, dy, dz, a.box);
        moved = !(dx * dx + dy * dy + dz * dz <= a.maxDisp2);   // NaN counts as moved
    }
    if (__any(moved) && lane_id() == 0) atomicOr(&a.state[ST_REBUILD], 1);
    check_inner_displacement(a, s, s < a.paddedAtoms && a.atomOfSlot[s] >= 0, s < a.paddedAtoms ? a.posq[s] : make_float4(0.f, 0.f, 0.f, 0.f));
}

// ------------------------------------------------------------------------------------------------
// Bounding boxes of the 32-atom blocks (two blocks per wavefront); also snapshots posq -> posqRef.
// ------------------------------------------------------------------------------------------------
__global__ void nl_block_bounds(NlArgs a) {
    const bool rebuild = a.state[ST_REBUILD] != 0;
    if (!rebuild && a.posqRel == nullptr) return;
    int s = blockIdx.x * blockDim.x + threadIdx.x;     // slot
    bool inRange = s < a.paddedAtoms;
    int sl = inRange ? s : a.paddedAtoms - 1;
    float4 p = a.posq[sl];
    bool valid = inRange && a.atomOfSlot[sl] >= 0;
    float4 center = a.blockCenter[sl >> 5];
    if (rebuild) {
        // first atom of the block (always valid: every block holds at least one real atom)
        float4 p0 = make_float4(__shfl(p.x, 0, 32), __shfl(p.y, 0, 32), __shfl(p.z, 0, 32), 0.f);
        float dx = p.x - p0.x, dy = p.y - p0.y, dz = p.z - p0.z;
        apply_pbc_rt(a.pbc, dx, dy, dz, a.box);
        if (!valid) { dx = dy = dz = 0; }
        float minx = dx, maxx = dx, miny = dy, maxy = dy, minz = dz, maxz = dz;
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) {
            minx = fminf(minx, __shfl_xor(minx, m)); maxx = fmaxf(maxx, __shfl_xor(maxx, m));
            miny = fminf(miny, __shfl_xor(miny, m)); maxy = fmaxf(maxy, __shfl_xor(maxy, m));
            minz = fminf(minz, __shfl_xor(minz, m)); maxz = fmaxf(maxz, __shfl_xor(maxz, m));
        }
        center = make_float4(p0.x + 0.5f * (minx + maxx), p0.y + 0.5f * (miny + maxy), p0.z + 0.5f * (minz + maxz), 0.f);
        if (inRange && (s & 31) == 0) {
            int blk = s >> 5;
            a.blockCenter[blk] = center;
            a.blockHalf[blk] = make_float4(0.5f * (maxx - minx), 0.5f * (maxy - miny), 0.5f * (maxz - minz), 0.f);
        }
    }
    // block-relative coordinates for the pair kernel (any origin works as long as both sides use the same one; between
    // rebuilds this entry keeps the centre of the last rebuild)
    if (inRange && a.posqRel != nullptr)
        a.posqRel[s] = valid ? make_float4(p.x - center.x, p.y - center.y, p.z - center.z, p.w) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (inRange && a.posqRelLo != nullptr) a.posqRelLo[s] = make_float4(0.f, 0.f, 0.f, 0.f);        // float positions in: nothing was lost
}

// ------------------------------------------------------------------------------------------------
// Large systems: bucket the i-blocks by the grid cell (edge >= list cutoff) of their bounding-box centre, so that
// nl_find_interactions only tests the blocks of nearby cells instead of all of them (which is O(blocks^2): 2.2 ms
// per rebuild at one million atoms).  One workgroup; runs only when a rebuild was requested.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int cell_coord(float x, float inv, int n) {
    float f = x * inv / (float) n;
    f -= floorf(f);
    int c = (int) (f * (float) n);
    return c >= n ? n - 1 : c;
}
__device__ __forceinline__ int cell_of(const NlArgs& a, float4 c) {
    return (cell_coord(c.x, a.cellInvX, a.ncx) * a.ncy + cell_coord(c.y, a.cellInvY, a.ncy)) * a.ncz + cell_coord(c.z, a.cellInvZ, a.ncz);
}

#define NL_BIN_CELLS 7168     // cells whose counters fit the binning workgroup's LDS (2 x 28 KB); the host never asks for more
#define NL_BIN_BATCH 4        // blocks per thread whose boxes are requested before any is processed

__global__ __launch_bounds__(1024) void nl_bin_blocks(NlArgs a) {
    if (a.state[ST_REBUILD] == 0) return;
    // Counting sort of the blocks by cell, ONE workgroup, counters and cursors in LDS: with the counters in global memory every
    // one of the ~100 dependent steps of a thread was an atomic round trip to L2 (180 us at 30 798 blocks).
    __shared__ int count[NL_BIN_CELLS + 1];      // count[c + 1] = blocks in cell c, then prefix-summed in place: the CSR start array
    __shared__ int cursor[NL_BIN_CELLS];
    __shared__ int partial[1024];
    __shared__ float hmax[3][16];
    __shared__ int numBig;
    const int t = threadIdx.x, ncells = a.ncx * a.ncy * a.ncz;
    for (int i = t; i <= ncells; i += 1024) count[i] = 0;
    for (int i = t; i < ncells; i += 1024) cursor[i] = 0;
    if (t == 0) numBig = 0;
    __syncthreads();
    // A handful of blocks have large bounding boxes (stragglers of the spatial sort); binning them would force every
    // block to search as far as the largest of them reaches.  They go to a short list that everybody scans instead.
    float hx = 0.f, hy = 0.f, hz = 0.f;
    // (a decomposed rank in halo mode has positions for its own blocks and its neighbours' sections only: the other blocks are not even looked at)
    const int numScan = a.numActive == 0 ? a.numBlocks : a.activeTotal / OMM_TILE;
    for (int b0 = 0; b0 < numScan; b0 += 1024 * NL_BIN_BATCH) {
        float4 h[NL_BIN_BATCH], c[NL_BIN_BATCH];
        int bb[NL_BIN_BATCH];
#pragma unroll
        for (int u = 0; u < NL_BIN_BATCH; u++) {
            active_block(a, min(b0 + u * 1024 + t, numScan - 1), bb[u]);
            h[u] = a.blockHalf[bb[u]]; c[u] = a.blockCenter[bb[u]];
        }
#pragma unroll
        for (int u = 0; u < NL_BIN_BATCH; u++) {
            const int b = bb[u];
            if (b0 + u * 1024 + t >= numScan) continue;
            if (h[u].x < 0.f) continue;                  // a block without atoms, or (halo mode) without current positions on this rank
            if (h[u].x > a.bigHalf || h[u].y > a.bigHalf || h[u].z > a.bigHalf) { a.cellBlocks[a.numBlocks + atomicAdd(&numBig, 1)] = b; continue; }
            atomicAdd(&count[cell_of(a, c[u]) + 1], 1);
            hx = fmaxf(hx, h[u].x); hy = fmaxf(hy, h[u].y); hz = fmaxf(hz, h[u].z);
        }
    }
    hx = wave_max(hx); hy = wave_max(hy); hz = wave_max(hz);
    if ((t & 63) == 0) { hmax[0][t >> 6] = hx; hmax[1][t >> 6] = hy; hmax[2][t >> 6] = hz; }
    __syncthreads();
    if (t < 3) {
        float m = 0.f;
        for (int w = 0; w < 16; w++) m = fmaxf(m, hmax[t][w]);
        a.cellMeta[t] = m;
    }
    if (t == 3) a.cellMeta[3] = __int_as_float(numBig);
    // exclusive prefix sum of the counts: each thread owns a contiguous run of cells
    const int per = (ncells + 1023) / 1024, c0 = t * per, c1 = min(ncells, c0 + per);
    int sum = 0;
    for (int c = c0; c < c1; c++) sum += count[c + 1];
    partial[t] = sum;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int i = 0; i < 1024; i++) { const int v = partial[i]; partial[i] = run; run += v; }
    }
    __syncthreads();
    int run = partial[t];
    for (int c = c0; c < c1; c++) { const int v = count[c + 1]; run += v; count[c + 1] = run; }
    __syncthreads();
    // after the scan count[c + 1] holds the END of cell c, i.e. count[] is the CSR start array (count[0] = 0)
    for (int i = t; i <= ncells; i += 1024) a.cellStart[i] = count[i];
    for (int b0 = 0; b0 < numScan; b0 += 1024 * NL_BIN_BATCH) {
        float4 h[NL_BIN_BATCH], c[NL_BIN_BATCH];
        int bb[NL_BIN_BATCH];
#pragma unroll
        for (int u = 0; u < NL_BIN_BATCH; u++) {
            active_block(a, min(b0 + u * 1024 + t, numScan - 1), bb[u]);
            h[u] = a.blockHalf[bb[u]]; c[u] = a.blockCenter[bb[u]];
        }
#pragma unroll
        for (int u = 0; u < NL_BIN_BATCH; u++) {
            const int b = bb[u];
            if (b0 + u * 1024 + t >= numScan) continue;
            if (h[u].x < 0.f || h[u].x > a.bigHalf || h[u].y > a.bigHalf || h[u].z > a.bigHalf) continue;
            const int cell = cell_of(a, c[u]);
            const int pos = count[cell] + atomicAdd(&cursor[cell], 1);
            a.cellBlocks[pos] = b;
            a.cellBoxes[2 * pos] = c[u];          // copies in cell order: the scan streams them
            a.cellBoxes[2 * pos + 1] = h[u];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Rows of i-block X = blockIdx.x.
// ------------------------------------------------------------------------------------------------
// LDS of one builder workgroup (a struct so that fused launches can alias it with the LDS of other work)
struct NlShared {
    int listJ[NL_LIST];
    unsigned listM[NL_LIST];
    int candY[NL_CAND];
    int rowMasked[NL_LIST / OMM_ROW];
    int candCount, listCount, chunkBase, candOverflow;
    int runCount, runBase[NL_MAX_RUNS], runRows[NL_MAX_RUNS];      // what this workgroup flushed: first chunk and rows of every flush
    // the i atoms of X relative to its centre, one array per component: two neighbouring atoms come back with one 64-bit
    // broadcast read, ready for the packed-FP32 exact test
    float ix[OMM_TILE], iy[OMM_TILE], iz[OMM_TILE];
};

// X = i-block of this workgroup, numWorkgroups = number of builder workgroups of the launch (for the hand-over at the end)
// May block Y be a partner of the owned block X?  One GPU: Y >= X (every block pair once).  Decomposed, pairs across a slab boundary on
// both sides: also the foreign blocks below the own range (those above are Y >= X).  Half-shell: own blocks Y >= X and the blocks of the lower
// neighbour's section only -- the pairs with the upper neighbour's atoms are that rank's, whatever of its blocks this rank holds for spreading.
__device__ __forceinline__ bool partner_allowed(const NlArgs& a, int X, int Y) {
    if (a.ddHalfShell) {
        const bool own = Y >= a.firstBlock && Y < a.firstBlock + a.ownedBlocks;
        return own ? Y >= X : (Y >= a.evalBlock0 && Y < a.evalBlock1);
    }
    return Y >= X || (a.ddMode && Y < a.firstBlock);
}

template <int PBC>
__device__ __forceinline__ void nl_build_body(const NlArgs& a, const int X, const int numWorkgroups, NlShared& sh) {
    int* const listJ = sh.listJ;
    unsigned* const listM = sh.listM;
    int* const candY = sh.candY;
    int* const rowMasked = sh.rowMasked;
    int& sCandCount = sh.candCount; int& sListCount = sh.listCount; int& sChunkBase = sh.chunkBase; int& sCandOverflow = sh.candOverflow;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float R2 = a.listCutoff2;
    const float Rlist = sqrtf(R2);
    const long long tStart = clock64();      // builder cost per i-block, kept in posqRef[..].w for diagnostics
    int candTotal = 0, entryTotal = 0;
    long long tPhase1 = tStart, tFlush = 0, tSetup = tStart, tRanges = tStart, tEntries = tStart;

    // atom (lane & 31) of X in every lane; broadcast later with v_readlane
    const float4 px = a.posq[X * OMM_TILE + (lane & 31)];
    if (t < OMM_TILE) a.posqRef[X * OMM_TILE + t] = px;          // reference positions of the displacement check
    if (a.ownedBlocks < a.numBlocks) {
        // A ranked list also holds j atoms of blocks this launch has no workgroup for, and the displacement check looks at
        // every slot: the workgroups share the snapshot of all of them.
        const int total = a.numActive == 0 ? a.paddedAtoms : a.activeTotal;
        for (int g = (X - a.firstBlock) * NL_THREADS + t; g < total; g += numWorkgroups * NL_THREADS) {
            int s;
            active_slot(a, g, s);
            if (s < a.firstBlock * OMM_TILE || s >= (a.firstBlock + a.ownedBlocks) * OMM_TILE) a.posqRef[s] = a.posq[s];
        }
    }
    const bool iValid = lane < OMM_TILE && a.atomOfSlot[X * OMM_TILE + lane] >= 0;
    const unsigned iValidMask = (unsigned) __ballot(iValid);
    const float4 cX = a.blockCenter[X], hX = a.blockHalf[X];
    const int2 exclRange = a.exclBlockRange != nullptr ? a.exclBlockRange[X] : make_int2(0, a.numBlocks);
    if (t == 0) { sCandCount = 0; sListCount = 0; sChunkBase = 0; sCandOverflow = 0; sh.runCount = 0; }
    // i atoms relative to the block centre.  When the block plus the list cutoff fits inside half a box length on
    // every axis, the image of j nearest to the centre is also the image nearest to every i atom within range, so
    // the exact test needs no per-pair image search (pairs beyond the list cutoff can only come out farther).
    float rx = px.x - cX.x, ry = px.y - cX.y, rz = px.z - cX.z;
    apply_pbc<PBC>(rx, ry, rz, a.box);
    const bool singleImage = PBC == 0 || (PBC == 1 && hX.x + Rlist < 0.5f * a.box.ax && hX.y + Rlist < 0.5f * a.box.by && hX.z + Rlist < 0.5f * a.box.cz);
    if (t < OMM_TILE) { sh.ix[t] = rx; sh.iy[t] = ry; sh.iz[t] = rz; }
    __syncthreads();
    tSetup = clock64();

    // Writes staged entries as rows.  final = false: only full rows, the remainder stays staged.
    auto flush = [&](bool final) {
        const long long tf0 = clock64();
        const int total = sListCount;
        const int nRows = final ? (total + OMM_ROW - 1) / OMM_ROW : total / OMM_ROW;
        const int nChunks = (nRows + OMM_CHUNK_ROWS - 1) / OMM_CHUNK_ROWS;
        if (nRows > 0) {
            if (t == 0) {
                sChunkBase = atomicAdd(&a.state[ST_ALLOC], nChunks);
                if (sh.runCount < NL_MAX_RUNS) { sh.runBase[sh.runCount] = sChunkBase; sh.runRows[sh.runCount] = nRows; }
                sh.runCount++;
            }
            __syncthreads();
            const int chunkBase = sChunkBase;
            for (int r = wave; r < nRows; r += NL_WAVES) {
                const int e = r * OMM_ROW + lane;
                const bool valid = e < total;
                const int j = valid ? listJ[e] : X * OMM_TILE;
                const unsigned m = valid ? listM[e] : 0u;
                const bool masked = __any(m != 0xFFFFFFFFu);
                const int chunk = chunkBase + r / OMM_CHUNK_ROWS;
                if (chunk < a.maxChunks) {
                    const size_t o = ((size_t) chunk * OMM_CHUNK_ROWS + (r % OMM_CHUNK_ROWS)) * OMM_ROW + lane;
                    a.rowJ[o] = j;
                    a.rowMask[o] = m;
                }
                else if (lane == 0) atomicOr(&a.state[ST_OVERFLOW], 1);
                if (lane == 0) rowMasked[r] = masked ? 1 : 0;
            }
            __syncthreads();
            for (int c = t; c < nChunks; c += NL_THREADS) {
                const int rowsIn = min(OMM_CHUNK_ROWS, nRows - OMM_CHUNK_ROWS * c);
                int bits = 0;
                for (int i = 0; i < rowsIn; i++) bits |= rowMasked[OMM_CHUNK_ROWS * c + i] << i;
                if (chunkBase + c < a.maxChunks) a.chunkInfo[chunkBase + c] = make_int2(X, rowsIn | (bits << 8));
            }
        }
        // keep the partial row staged
        const int rem = final ? 0 : total - nRows * OMM_ROW;
        int tj = 0; unsigned tm = 0;
        if (t < rem) { tj = listJ[nRows * OMM_ROW + t]; tm = listM[nRows * OMM_ROW + t]; }
        __syncthreads();
        if (t < rem) { listJ[t] = tj; listM[t] = tm; }
        if (t == 0) sListCount = rem;
        __syncthreads();
        entryTotal += total - rem;
        tFlush += clock64() - tf0;
    };

    // Block-level test of one candidate Y against X's bounding box.
    auto blockTest = [&](int Y) -> bool {
        const float4 cY = a.blockCenter[Y], hY = a.blockHalf[Y];
        float dx = cY.x - cX.x, dy = cY.y - cX.y, dz = cY.z - cX.z;
        apply_pbc<PBC>(dx, dy, dz, a.box);
        dx = fmaxf(0.f, fabsf(dx) - hX.x - hY.x);
        dy = fmaxf(0.f, fabsf(dy) - hX.y - hY.y);
        dz = fmaxf(0.f, fabsf(dz) - hX.z - hY.z);
        bool cand = !(dx * dx + dy * dy + dz * dz >= R2);
        // Triclinic: the sequential image reduction only finds the nearest copy when it is less than half a
        // box width away; if that cannot be guaranteed, defer to the exact per-atom test.
        if (PBC == 2 && (0.5f * a.box.cz - hX.z - hY.z < Rlist || 0.5f * a.box.by - hX.y - hY.y < Rlist)) cand = true;
        return cand;
    };
    auto appendCandidates = [&](bool cand, int Y) {
        const unsigned long long cm = __ballot(cand);
        if (cm != 0) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&sCandCount, __popcll(cm));
            base = __shfl(base, 0);
            if (base + __popcll(cm) > NL_CAND) { if (lane == 0) sCandOverflow = 1; }
            else if (cand) candY[base + lane_prefix_count(cm)] = Y;
        }
    };
    // ---- phase 1, cell mode: only the blocks whose centre cell lies within reach of X (PBC == 1 only)
    bool cellDone = false;
    if (PBC == 1 && a.cellMode) {
        const int n[3] = {a.ncx, a.ncy, a.ncz};
        const float inv[3] = {a.cellInvX, a.cellInvY, a.cellInvZ};
        const float c3[3] = {cX.x, cX.y, cX.z}, h3[3] = {hX.x, hX.y, hX.z};
        // A binned block Y lies inside its centre's cell grown by the largest binned half extent (cellMeta), so X can only
        // reach it if X's box comes within Rlist of that grown cell.  Work in unwrapped cell coordinates around X's centre:
        // per axis the cells whose grown extent is within Rlist of X's box, clamped to one period.
        int lo[3], cnt[3];
        float grow[3];          // hX + hmax + half a cell, per axis (nm)
#pragma unroll
        for (int d = 0; d < 3; d++) {
            grow[d] = h3[d] + a.cellMeta[d] + 0.5f / inv[d];
            const float u = c3[d] * inv[d] - 0.5f, w = (Rlist + grow[d]) * inv[d] + 0.01f;     // cell k qualifies if |k - u| <= w
            lo[d] = (int) ceilf(u - w);
            cnt[d] = (int) floorf(u + w) - lo[d] + 1;
            if (cnt[d] >= n[d]) { lo[d] = 0; cnt[d] = n[d]; }
        }
        // One thread per (x, y) column of the region: the cells of a column are consecutive in the cell-sorted block list, so
        // a column is one range of entries (two when it wraps around the box).  The z extent of each column is cut to what
        // the x/y gap leaves of Rlist.  Entries are first expanded into an LDS list (listJ is free during phase 1), then
        // tested by all threads in parallel: three dependent memory round trips per workgroup instead of two per cell.
        int* const entryList = listJ;
        int& sEntryCount = sh.listCount;          // phase 2 starts from zero again
        auto testEntry = [&](int e) {
            const int Y = a.cellBlocks[e];
            const float4 cY = a.cellBoxes[2 * e], hY = a.cellBoxes[2 * e + 1];
            float dx = cY.x - cX.x, dy = cY.y - cX.y, dz = cY.z - cX.z;
            apply_pbc<PBC>(dx, dy, dz, a.box);
            dx = fmaxf(0.f, fabsf(dx) - hX.x - hY.x);
            dy = fmaxf(0.f, fabsf(dy) - hX.y - hY.y);
            dz = fmaxf(0.f, fabsf(dz) - hX.z - hY.z);
            if (partner_allowed(a, X, Y) && !(dx * dx + dy * dy + dz * dz >= R2)) {
                const int pos = atomicAdd(&sCandCount, 1);
                if (pos < NL_CAND) candY[pos] = Y; else sCandOverflow = 1;
            }
        };
        const int numCols = cnt[0] * cnt[1];
        for (int col0 = 0; col0 < numCols; col0 += NL_THREADS) {
            const int col = col0 + t;
            int r0[2] = {0, 0}, r1[2] = {0, 0};          // entry ranges of this thread's column
            if (col < numCols) {
                const int kx = lo[0] + col / cnt[1], ky = lo[1] + col % cnt[1];
                // distance of the column's centre line from X's centre, nearest image (a span of the whole period starts at cell 0,
                // wherever X is)
                float ddx = (kx + 0.5f) / inv[0] - c3[0], ddy = (ky + 0.5f) / inv[1] - c3[1];
                ddx -= (n[0] / inv[0]) * rintf(ddx * inv[0] / n[0]);
                ddy -= (n[1] / inv[1]) * rintf(ddy * inv[1] / n[1]);
                const float gx = fmaxf(0.f, fabsf(ddx) - grow[0]);
                const float gy = fmaxf(0.f, fabsf(ddy) - grow[1]);
                const float rem = R2 - gx * gx - gy * gy;
                if (rem > 0.f) {
                    int zlo = lo[2], zn = cnt[2];
                    if (cnt[2] < n[2]) {
                        const float u = c3[2] * inv[2] - 0.5f, w = (sqrtf(rem) + grow[2]) * inv[2] + 0.01f;
                        zlo = max(lo[2], (int) ceilf(u - w));
                        zn = min(lo[2] + cnt[2] - 1, (int) floorf(u + w)) - zlo + 1;
                    }
                    if (zn > 0) {
                        const int xw = ((kx % n[0]) + n[0]) % n[0], yw = ((ky % n[1]) + n[1]) % n[1];
                        const int base = (xw * n[1] + yw) * n[2];
                        const int z0 = ((zlo % n[2]) + n[2]) % n[2];
                        const int first = min(zn, n[2] - z0);          // cells before the wrap
                        r0[0] = a.cellStart[base + z0]; r1[0] = a.cellStart[base + z0 + first];
                        if (first < zn) { r0[1] = a.cellStart[base]; r1[1] = a.cellStart[base + zn - first]; }
                    }
                }
            }
            const int len = (r1[0] - r0[0]) + (r1[1] - r0[1]);
            int slot = 0;
            if (len > 0) slot = atomicAdd(&sEntryCount, len);
            if (len > 0 && slot + len <= NL_LIST) {
                for (int e = r0[0]; e < r1[0]; e++) entryList[slot++] = e;
                for (int e = r0[1]; e < r1[1]; e++) entryList[slot++] = e;
            }
            else if (len > 0) {
                // more entries than the staging list holds (an oversized X reaching across the box): this thread's share one by
                // one, and the part of the list its range would have covered is marked empty
                for (int i = slot; i < NL_LIST; i++) entryList[i] = -1;
                for (int e = r0[0]; e < r1[0]; e++) testEntry(e);
                for (int e = r0[1]; e < r1[1]; e++) testEntry(e);
            }
            __syncthreads();
            tRanges = clock64();
            const int numEntries = min(sEntryCount, NL_LIST);
            for (int i = t; i < numEntries; i += NL_THREADS) {
                const int e = entryList[i];
                if (e >= 0) testEntry(e);
            }
            __syncthreads();
            tEntries = clock64();
            if (t == 0) sEntryCount = 0;
            __syncthreads();
        }
        // the oversized blocks, tested by everybody
        const int numBig = __float_as_int(a.cellMeta[3]);
        for (int i = t; i < numBig; i += NL_THREADS) {
            const int Y = a.cellBlocks[a.numBlocks + i];
            if (partner_allowed(a, X, Y) && blockTest(Y)) {
                const int pos = atomicAdd(&sCandCount, 1);
                if (pos < NL_CAND) candY[pos] = Y; else sCandOverflow = 1;
            }
        }
        __syncthreads();
        tPhase1 = clock64();
        if (sCandOverflow == 0) cellDone = true;
        else { __syncthreads(); if (t == 0) { sCandCount = 0; sCandOverflow = 0; } __syncthreads(); }   // absurdly fat block: scan everything
    }
    // candidate ranges: [X, numBlocks) and, on a decomposed run, the foreign blocks [0, firstBlock) as well (their owners
    // evaluate the same pairs for their own atoms)
    for (int range = 0; range < (a.ddMode ? 2 : 1); range++) {
    const int rangeBegin = range == 0 ? X : 0, rangeEnd = range == 0 ? a.numBlocks : a.firstBlock;
    for (int window = rangeBegin; window < rangeEnd; window += NL_CAND) {
        const int windowEnd = min(rangeEnd, window + NL_CAND);
        // ---- phase 1: block-level test
        if (!cellDone) {
            for (int yb = window + wave * 64; yb < windowEnd; yb += NL_THREADS) {
                const int Y = yb + lane;
                appendCandidates(Y < windowEnd && (!a.ddHalfShell || partner_allowed(a, X, Y)) && blockTest(Y), Y);
            }
        }
        __syncthreads();
        const int numCand = sCandCount;
        candTotal += numCand;
        const int numPasses = (numCand + 1) / 2;
        // ---- phase 2: atom-level test, one pass = two candidate blocks = 64 atoms per wavefront.  The wavefronts run
        //      NL_ROUND passes each on their own (appending through one LDS atomic per pass) before the workgroup meets
        //      again to flush full rows; the atom data of the next pass is fetched while the current one is tested.
        const int lj = lane & 31;
        auto fetch = [&](int pass, int& Yc, int& atomJ, float4& pj) {
            const int ci = 2 * pass + (lane >> 5);
            const bool ok = pass < numPasses && ci < numCand;
            Yc = ok ? candY[ci] : X;
            const int j = Yc * OMM_TILE + lj;
            atomJ = ok ? a.atomOfSlot[j] : -1;
            pj = a.posq[j];
        };
        for (int p0 = 0; p0 < numPasses; p0 += NL_WAVES * NL_ROUND) {
            int YcN, atomJN; float4 pjN;
            fetch(p0 + wave, YcN, atomJN, pjN);
            for (int r = 0; r < NL_ROUND; r++) {
                const int pass = p0 + r * NL_WAVES + wave;
                if (pass >= numPasses) break;                    // wave-uniform
                const int Yc = YcN, atomJ = atomJN;
                const float4 pj = pjN;
                if (r + 1 < NL_ROUND) fetch(pass + NL_WAVES, YcN, atomJN, pjN);
                const int j = Yc * OMM_TILE + lj;
                const bool ok = atomJ >= 0;
                // distance to X's bounding box
                float dx = pj.x - cX.x, dy = pj.y - cX.y, dz = pj.z - cX.z;
                apply_pbc<PBC>(dx, dy, dz, a.box);
                const float bx = fmaxf(0.f, fabsf(dx) - hX.x), by = fmaxf(0.f, fabsf(dy) - hX.y), bz = fmaxf(0.f, fabsf(dz) - hX.z);
                bool near = !(bx * bx + by * by + bz * bz >= R2);
                if (PBC == 2 && (0.5f * a.box.cz - hX.z < Rlist || 0.5f * a.box.by - hX.y < Rlist)) near = true;
                // exact test against the 32 atoms of X (same metric as the pair kernel); executed by every lane so
                // that the v_readlane broadcasts sit in convergent code.  Passes none of whose 64 atoms comes near X's box
                // (candidate blocks at the rim of the reach) skip it altogether -- a wave-uniform decision.
                if (!__any(ok && near)) continue;
                bool any = false;
                if (singleImage) {
                    // two i atoms per iteration in packed FP32; their coordinates are wave-uniform LDS reads (no VALU slot,
                    // unlike a v_readlane broadcast): 8 vector instructions per pair of atoms instead of 22
                    const v2f jx = bc2(dx), jy = bc2(dy), jz = bc2(dz);
#pragma unroll
                    for (int k = 0; k < OMM_TILE; k += 2) {
                        const v2f ex = jx - mk2(sh.ix[k], sh.ix[k + 1]), ey = jy - mk2(sh.iy[k], sh.iy[k + 1]), ez = jz - mk2(sh.iz[k], sh.iz[k + 1]);
                        const v2f r2 = ex * ex + ey * ey + ez * ez;
                        any = any || !(r2.x >= R2) || !(r2.y >= R2);
                    }
                }
                else {
#pragma unroll
                    for (int k = 0; k < OMM_TILE; k++) {
                        const float xi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px.x), k));
                        const float yi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px.y), k));
                        const float zi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px.z), k));
                        float ex = pj.x - xi, ey = pj.y - yi, ez = pj.z - zi;
                        apply_pbc<PBC>(ex, ey, ez, a.box);
                        any = any || !(ex * ex + ey * ey + ez * ez >= R2);
                    }
                }
                unsigned mask = 0;
                if (ok && near && any) {
                    mask = iValidMask;
                    if (Yc == X) mask &= (1u << lj) - 1u;        // diagonal block: each pair once, no self pair
                    // Only blocks inside X's exclusion-partner range can hold an excluded partner.
                    if (Yc >= exclRange.x && Yc <= exclRange.y) {
                        if (a.exclSlotStart != nullptr) {
                            // slot-keyed table: four independent loads in flight per trip
                            const int e1 = a.exclSlotStart[j + 1];
                            for (int e = a.exclSlotStart[j]; e < e1; e += 4) {
                                const int s0 = a.exclSlots[e];
                                const int s1 = e + 1 < e1 ? a.exclSlots[e + 1] : -1;
                                const int s2 = e + 2 < e1 ? a.exclSlots[e + 2] : -1;
                                const int s3 = e + 3 < e1 ? a.exclSlots[e + 3] : -1;
                                if ((s0 >> 5) == X) mask &= ~(1u << (s0 & 31));
                                if ((s1 >> 5) == X) mask &= ~(1u << (s1 & 31));
                                if ((s2 >> 5) == X) mask &= ~(1u << (s2 & 31));
                                if ((s3 >> 5) == X) mask &= ~(1u << (s3 & 31));
                            }
                        }
                        else
                            for (int e = a.exclStart[atomJ]; e < a.exclStart[atomJ + 1]; e++) {
                                const int s = a.slotOfAtom[a.exclAtoms[e]];
                                if ((s >> 5) == X) mask &= ~(1u << (s & 31));
                            }
                    }
                }
                const bool passes = mask != 0;
                const unsigned long long pm = __ballot(passes);
                if (pm != 0) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&sListCount, __popcll(pm));
                    base = __shfl(base, 0);
                    if (passes) {
                        const int pos = base + lane_prefix_count(pm);
                        listJ[pos] = j;
                        listM[pos] = mask;
                    }
                }
            }
            __syncthreads();
            if (sListCount >= NL_FLUSH) flush(false);
        }
        __syncthreads();
        if (t == 0) sCandCount = 0;
        __syncthreads();
        if (cellDone) break;
    }
    if (cellDone) break;
    }
    flush(true);
    if (a.blockRuns != nullptr && t <= NL_MAX_RUNS) {
        // the run directory of X, for the pruning passes of the steps that follow
        int* const dir = a.blockRuns + (size_t) X * (1 + 2 * NL_MAX_RUNS);
        if (t == 0) { dir[0] = sh.runCount; if (sh.runCount > NL_MAX_RUNS) atomicOr(&a.state[ST_NO_PRUNE], 1); }
        else if (t - 1 < min(sh.runCount, NL_MAX_RUNS)) { dir[2 * t - 1] = sh.runBase[t - 1]; dir[2 * t] = sh.runRows[t - 1]; }
    }

    // Last workgroup out clears the rebuild request.
    if (t == 0) {
        a.posqRef[X * OMM_TILE].w = (float) (clock64() - tStart);
        a.posqRef[X * OMM_TILE + 1].w = (float) candTotal;
        a.posqRef[X * OMM_TILE + 2].w = (float) (tPhase1 - tStart);
        a.posqRef[X * OMM_TILE + 3].w = (float) tFlush;
        a.posqRef[X * OMM_TILE + 4].w = (float) entryTotal;
        a.posqRef[X * OMM_TILE + 5].w = (float) (tSetup - tStart);
        a.posqRef[X * OMM_TILE + 6].w = (float) (tRanges - tStart);
        a.posqRef[X * OMM_TILE + 7].w = (float) (tEntries - tStart);
        // No __threadfence() here (OMM_NL_TAIL_FENCE=1 restores it for A/B): nothing inside this launch reads what the other
        // workgroups wrote -- the last one out only exchanges the allocation counter, and the workgroup barriers of flush()
        // order every allocation of this workgroup before the increment below; rows, reference positions and the published
        // state reach their consumers through the kernel boundary.  A fence per builder workgroup is an L2 write-back of
        // everything written on the XCD so far, 30 798 times per rebuild at a million atoms.
#if defined(OMM_NL_TAIL_FENCE) && OMM_NL_TAIL_FENCE
        __threadfence();
#endif
        const int done = atomicAdd(&a.state[ST_BLOCKS_DONE], 1);
        if (done == numWorkgroups - 1) {
            // publish the list length, return the working counter to zero, clear the request
            a.state[ST_NUM_CHUNKS] = atomicExch(&a.state[ST_ALLOC], 0);
            a.state[ST_BLOCKS_DONE] = 0;
            a.state[ST_REBUILD] = 0;
            a.state[ST_PRUNE_REQUEST] = 1;                 // new rows: the pruned list is cut again by the launch that follows
            atomicAdd(&a.state[ST_REBUILD_COUNT], 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The pruned list (every step, a launch of its own behind the builder's): of the entries an i-block's rows hold -- j atoms within
// cutoff + padding of an atom of X when the list was built -- those whose j atom lies within the cutoff of an atom of X NOW,
// packed into fresh rows.  The position of j in X's frame is formed exactly as the pair kernel forms it (block-relative
// coordinates plus the offset of the two block centres in the nearest image) and compared with the block-relative coordinates of
// X's atoms; the squared cutoff carries a relative margin of 1e-4, far above any rounding and above the band the cutoff-edge
// path re-decides.  Order of the entries is kept (row by row, lane by lane): the pruned rows are a deterministic function of
// the positions.
// One wavefront per i-block, no barrier after the first: pass 1 takes the rows four at a time (the two dependent memory round
// trips -- indices, then the gathers -- are paid once per four rows) and leaves each row's keep mask in the lane of its number;
// a wavefront scan places the rows, one atomic allocates the chunks, pass 2 re-reads the (cache-resident) rows and writes the
// survivors to their places.  Few registers, 1.5 kB of LDS per workgroup: the launch is bound by memory latency and needs the
// occupancy.
// ------------------------------------------------------------------------------------------------
#define NL_RUN_ROWS (NL_LIST / OMM_ROW)              // rows of one flush at most
#define NL_PRUNE_BATCH 4
#ifdef OMMHIP_EMU
#define OMM_WAVES_PER_EU(n)
#else
#define OMM_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#endif
template <int PBC>
__global__ __launch_bounds__(256) void nl_prune_rows(NlArgs a) {
    __shared__ float sIx[4][OMM_TILE], sIy[4][OMM_TILE], sIz[4][OMM_TILE];
    // (read by every workgroup before the last one out clears it: that one waits for all of them)
    if (a.state[ST_PRUNE_REQUEST] == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        // reference positions of the displacement check of this list: every slot the check looks at
        const int total = a.numActive == 0 ? a.paddedAtoms : a.activeTotal;
        for (int g = blockIdx.x * 256 + threadIdx.x; g < total; g += gridDim.x * 256) {
            int sl;
            active_slot(a, g, sl);
            a.posqRefInner[sl] = a.posq[sl];
        }
    }
    const int Xraw = a.firstBlock + blockIdx.x * 4 + wave;
    const bool live = Xraw < a.firstBlock + a.ownedBlocks;
    const int X = live ? Xraw : a.firstBlock;
    if (lane < OMM_TILE) { const float4 pi = a.posqRel[X * OMM_TILE + lane]; sIx[wave][lane] = pi.x; sIy[wave][lane] = pi.y; sIz[wave][lane] = pi.z; }
    __syncthreads();
    if (live) {
        const float* const ix = sIx[wave]; const float* const iy = sIy[wave]; const float* const iz = sIz[wave];
        const int* const dir = a.blockRuns + (size_t) X * (1 + 2 * NL_MAX_RUNS);
        const int numRuns = min(dir[0], NL_MAX_RUNS);
        const float4 cX = a.blockCenter[X], hX = a.blockHalf[X];
        // the cases in which one image per j atom serves all of X (the pair kernel's `single`); anything else is copied unpruned
        bool exact = PBC == 0;
        if (PBC == 1) exact = hX.w != 0.f && hX.x + a.cutoff < 0.5f * a.box.ax && hX.y + a.cutoff < 0.5f * a.box.by && hX.z + a.cutoff < 0.5f * a.box.cz;
        const float R2 = a.pruneCutoff2;
        for (int run = 0; run < numRuns; run++) {
            const int base = dir[1 + 2 * run];
            int nRows = min(dir[2 + 2 * run], NL_RUN_ROWS);
            if ((long long) base * OMM_CHUNK_ROWS + nRows > (long long) a.maxChunks * OMM_CHUNK_ROWS) nRows = max(0, (a.maxChunks - base) * OMM_CHUNK_ROWS);     // the list overflowed
            const size_t inFirst = (size_t) base * OMM_CHUNK_ROWS * OMM_ROW;
            // pass 1: keep masks; lane r ends up with those of row r
            unsigned long long myKeep = 0ull; bool myCut = false;
            for (int r0 = 0; r0 < nRows; r0 += NL_PRUNE_BATCH) {
                int jv[NL_PRUNE_BATCH]; unsigned mv[NL_PRUNE_BATCH]; float4 pj[NL_PRUNE_BATCH], cY[NL_PRUNE_BATCH];
#pragma unroll
                for (int q = 0; q < NL_PRUNE_BATCH; q++) {
                    const bool in = r0 + q < nRows;
                    const size_t o = inFirst + (size_t) (in ? r0 + q : r0) * OMM_ROW + lane;
                    jv[q] = a.rowJ[o]; mv[q] = in ? a.rowMask[o] : 0u;
                }
#pragma unroll
                for (int q = 0; q < NL_PRUNE_BATCH; q++) { pj[q] = a.posqRel[jv[q]]; cY[q] = a.blockCenter[jv[q] >> 5]; }
                // j in X's frame, two copies per lane for the packed arithmetic
                v2f jx[NL_PRUNE_BATCH], jy[NL_PRUNE_BATCH], jz[NL_PRUNE_BATCH];
#pragma unroll
                for (int q = 0; q < NL_PRUNE_BATCH; q++) {
                    float ox = cY[q].x - cX.x, oy = cY[q].y - cX.y, oz = cY[q].z - cX.z;
                    if (PBC == 1) {
                        const float nx = rintf((cY[q].x - cX.x + pj[q].x) * a.box.invAx), ny = rintf((cY[q].y - cX.y + pj[q].y) * a.box.invBy), nz = rintf((cY[q].z - cX.z + pj[q].z) * a.box.invCz);
                        ox = fmaf(-nx, a.box.axLo, fmaf(-nx, a.box.ax, cY[q].x)) - cX.x;
                        oy = fmaf(-ny, a.box.byLo, fmaf(-ny, a.box.by, cY[q].y)) - cX.y;
                        oz = fmaf(-nz, a.box.czLo, fmaf(-nz, a.box.cz, cY[q].z)) - cX.z;
                    }
                    jx[q] = bc2(pj[q].x + ox); jy[q] = bc2(pj[q].y + oy); jz[q] = bc2(pj[q].z + oz);
                }
                bool any[NL_PRUNE_BATCH];
#pragma unroll
                for (int q = 0; q < NL_PRUNE_BATCH; q++) any[q] = !exact;
                if (exact) {
                    // two atoms of X per iteration (wave-uniform LDS reads), against the four rows of the batch.  The barrier keeps the
                    // compiler from holding all 96 coordinates in registers across the loop over the batches (a third of the occupancy).
                    asm volatile("" ::: "memory");
#pragma unroll 4
                    for (int k = 0; k < OMM_TILE; k += 2) {
                        const v2f xi = mk2(ix[k], ix[k + 1]), yi = mk2(iy[k], iy[k + 1]), zi = mk2(iz[k], iz[k + 1]);
#pragma unroll
                        for (int q = 0; q < NL_PRUNE_BATCH; q++) {
                            const v2f ex = jx[q] - xi, ey = jy[q] - yi, ez = jz[q] - zi;
                            const v2f r2 = ex * ex + ey * ey + ez * ez;
                            any[q] = any[q] || !(r2.x >= R2) || !(r2.y >= R2);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < NL_PRUNE_BATCH; q++) {
                    if (r0 + q >= nRows) break;
                    const bool keep = mv[q] != 0u && any[q];
                    const unsigned long long kb = __ballot(keep);
                    const bool cut = __any(keep && mv[q] != 0xFFFFFFFFu);
                    if (lane == r0 + q) { myKeep = kb; myCut = cut; }
                }
            }
            // where each row's survivors go
            const int cnt = __popcll(myKeep);
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < NL_RUN_ROWS; d <<= 1) { const int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
            const int total = __shfl(incl, NL_RUN_ROWS - 1);
            const int excl = incl - cnt;                              // first position of row `lane`
            const int outRows = (total + OMM_ROW - 1) / OMM_ROW, outChunks = (outRows + OMM_CHUNK_ROWS - 1) / OMM_CHUNK_ROWS;
            if (outChunks == 0) continue;
            int outBase = 0;
            if (lane == 0) outBase = atomicAdd(&a.state[ST_ALLOC_INNER], outChunks);
            outBase = __shfl(outBase, 0);
            if (outBase + outChunks > a.maxChunks) continue;          // cannot happen unless the list as built overflowed
            const size_t outFirst = (size_t) outBase * OMM_CHUNK_ROWS * OMM_ROW;
            // pass 2: the entries go straight to their places (the rows are re-read, from cache, four at a time)
            for (int r0 = 0; r0 < nRows; r0 += NL_PRUNE_BATCH) {
                int jv[NL_PRUNE_BATCH]; unsigned mv[NL_PRUNE_BATCH];
#pragma unroll
                for (int q = 0; q < NL_PRUNE_BATCH; q++) {
                    const size_t o = inFirst + (size_t) (r0 + q < nRows ? r0 + q : r0) * OMM_ROW + lane;
                    jv[q] = a.rowJ[o]; mv[q] = a.rowMask[o];
                }
#pragma unroll
                for (int q = 0; q < NL_PRUNE_BATCH; q++) {
                    const int r = r0 + q;
                    if (r >= nRows) break;
                    const unsigned kbLo = (unsigned) __shfl((int) (unsigned) myKeep, r), kbHi = (unsigned) __shfl((int) (unsigned) (myKeep >> 32), r);
                    const unsigned long long kb = ((unsigned long long) kbHi << 32) | kbLo;
                    const int first = __shfl(excl, r);
                    if ((kb >> lane) & 1ull) {
                        const size_t w = outFirst + first + lane_prefix_count(kb);
                        a.rowJInner[w] = jv[q];
                        a.rowMaskInner[w] = mv[q];
                    }
                }
            }
            // the unused lanes of the last row
            {
                const int e = (outRows - 1) * OMM_ROW + lane;
                if (e >= total) { a.rowJInner[outFirst + e] = X * OMM_TILE; a.rowMaskInner[outFirst + e] = 0u; }
            }
            // chunk headers (lane c: chunk c): a row is flagged "masked" when a row of the run with partial masks among its
            // survivors overlaps it (a superset of the rows that really hold one: the flag only selects the code path that looks
            // at the masks)
            int bits = 0;
            const int rowsIn = min(OMM_CHUNK_ROWS, outRows - OMM_CHUNK_ROWS * lane);
            for (int r = 0; r < nRows; r++) {
                const int n = __shfl(cnt, r), pos = __shfl(excl, r);
                const bool cutR = __shfl((int) myCut, r) != 0;
                if (cutR && n > 0)
                    for (int i = 0; i < OMM_CHUNK_ROWS; i++) {
                        const int lo = (OMM_CHUNK_ROWS * lane + i) * OMM_ROW;
                        if (i < rowsIn && pos < lo + OMM_ROW && pos + n > lo) bits |= 1 << i;
                    }
            }
            if (lane < outChunks) {
                // the last row's padding lanes carry an empty mask
                if (OMM_CHUNK_ROWS * lane + rowsIn == outRows && (total & (OMM_ROW - 1)) != 0) bits |= 1 << (rowsIn - 1);
                a.chunkInfoInner[outBase + lane] = make_int2(X, rowsIn | (bits << 8));
            }
        }
    }
    // last workgroup out publishes the length of the pruned list and returns the working counters to zero
    __syncthreads();
    if (threadIdx.x == 0) {
        const int done = atomicAdd(&a.state[ST_PRUNE_DONE], 1);
        if (done == (int) gridDim.x - 1) {
            a.state[ST_NUM_CHUNKS_INNER] = atomicExch(&a.state[ST_ALLOC_INNER], 0);
            a.state[ST_PRUNE_DONE] = 0;
            a.state[ST_PRUNE_REQUEST] = 0;
        }
    }
}

// One workgroup of the per-step list launch: X's rows are rebuilt if a rebuild is due.
template <int PBC>
__device__ __forceinline__ void nl_find_body(const NlArgs& a, const int X, const int numWorkgroups, NlShared& sh) {
    if (a.state[ST_REBUILD] == 0) return;
    nl_build_body<PBC>(a, X, numWorkgroups, sh);
}

static void launch_prune(const NlArgs& a, hipStream_t st) {
    if (a.rowJInner == nullptr) return;
    const dim3 grid((a.ownedBlocks + 3) / 4);
    if (a.pbc == 0) hipLaunchKernelGGL(nl_prune_rows<0>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(nl_prune_rows<1>, grid, dim3(256), 0, st, a);
}

template <int PBC>
__global__ __launch_bounds__(NL_THREADS) void nl_find_interactions(NlArgs a) {
    __shared__ NlShared sh;
#ifndef OMMHIP_EMU
    const bool traced = a.trace != nullptr && a.state[ST_REBUILD] != 0;          // profiling (OPENMM_HIP_NL_TRACE): start and end of every workgroup
    if (traced && threadIdx.x == 0) a.trace[2 * blockIdx.x] = (long long) wall_clock64();
#endif
    nl_find_body<PBC>(a, a.firstBlock + blockIdx.x, gridDim.x, sh);
#ifndef OMMHIP_EMU
    if (traced && threadIdx.x == 0) a.trace[2 * blockIdx.x + 1] = (long long) wall_clock64();
#endif
}

// The same builder with resident workgroups that walk through the i-blocks (grid = a few workgroups per compute unit): at a million
// atoms the one-block-per-workgroup launch keeps ~3.4 workgroups per CU in flight although 7 fit (30 798 workgroups living 66 us each
// arrive at ~13 per microsecond: `OPENMM_HIP_NL_TRACE`).  Same-box A/B (`profiles/r05x_ab_resident_builder.txt`): 985 527 atoms 2.298 -> 2.22 ms per
// step with 7, 10 or 14 workgroups per CU, 92 224 atoms 0.2950 -> 0.2888.  OPENMM_HIP_NL_PERSISTENT=<per CU> (default 8, 0 = one workgroup per block).
// Register budget of the resident builder: left to itself the compiler takes 143 VGPRs for the loop around the builder's body (loop-invariant
// values kept across i-blocks) -- three wavefronts per SIMD, so three of the seven workgroups a CU's LDS would hold.  Asked for four
// wavefronts it fits 128 without spilling, for five it spills four (94): same-box, 985 527 atoms, one stream: 2.096 / 2.050 / 2.063 ms per step
// at 3 / 4 / 5 (`profiles/r11/r11m_ab_builder_occupancy.txt`; a rebuild 1.14 -> 0.87 ms).
#ifndef OMM_NL_RESIDENT_WAVES
#define OMM_NL_RESIDENT_WAVES 4
#endif
#ifdef OMMHIP_EMU
#define OMM_NL_RESIDENT_ATTR
#else
#define OMM_NL_RESIDENT_ATTR __attribute__((amdgpu_waves_per_eu(OMM_NL_RESIDENT_WAVES)))
#endif
template <int PBC>
__global__ __launch_bounds__(NL_THREADS) OMM_NL_RESIDENT_ATTR void nl_find_interactions_resident(NlArgs a) {
    __shared__ NlShared sh;
    if (a.state[ST_REBUILD] == 0) return;          // read once: the last block to finish clears the request, and by then no block is left
    // ONE call site of the builder's body for both ways of walking through the i-blocks (round 5): with a call in each branch the body was
    // inlined twice and the kernel took 144 VGPRs -- three wavefronts per SIMD, three workgroups per CU -- where one copy takes 95 (five).
    int first = blockIdx.x, stride = gridDim.x, base = 0, limit = a.ownedBlocks;
    if (a.xcdAware && gridDim.x % 8 == 0) {
        // workgroup w runs on XCD w % 8 (round-robin dispatch): it walks through the (w % 8)-th eighth of the i-blocks, whose candidate blocks
        // -- spatial neighbours, close in the slot order -- then stay in that XCD's L2
        limit = (a.ownedBlocks + 7) / 8;
        base = (int) (blockIdx.x % 8) * limit;
        first = blockIdx.x / 8; stride = gridDim.x / 8;
    }
    for (int k = first; k < limit; k += stride) {
        const int b = base + k;
        if (b < a.ownedBlocks) nl_build_body<PBC>(a, a.firstBlock + b, a.ownedBlocks, sh);          // block-uniform
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Fused per-step front end (one launch instead of three): double positions -> wrapped float posq,
// displacement check against posqRef, and the block bounding boxes (recomputed every step; they are
// only consumed when a rebuild follows, and recomputing them costs less than a conditional launch).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nl_prepare(NlArgs a, const double4* __restrict__ pos, const int4* __restrict__ wrap, BoxD boxd,
                                                  float4* __restrict__ posqOut, int checkDisplacement,
                                                  uint4* __restrict__ clearA, size_t clearNA, uint4* __restrict__ clearB, size_t clearNB) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;      // the grid covers the slots to convert (a multiple of 32) exactly
    // start-of-evaluation clears (force accumulator, PME charge grid) ride along: nothing in this launch reads them
    {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        const size_t stride = (size_t) gridDim.x * blockDim.x;
        for (size_t i = (size_t) g; i < clearNA + clearNB; i += stride) {
            if (i < clearNA) clearA[i] = z;
            else clearB[i - clearNA] = z;
        }
    }
    int s;                                                     // slot: g itself, or the g-th slot of the active ranges (halo mode)
    const bool inRange = active_slot(a, g, s);
    const int sl = inRange ? s : a.paddedAtoms - 1;
    if (g == 0 && a.ddFlags != nullptr && a.posWire != nullptr) {
        // some rank saw one of its atoms near the drift margin one step ago (the flag travelled in its trailer): every rank raises
        // the same word at the same evaluation, the hosts re-sort together
        int level = 0;                                         // 1: re-sort soon (off the step), 3: re-sort now
        for (int r = 0; r < a.ddRanks; r++) level |= (int) ((const double4*) (a.posWire + (size_t) r * a.ddSlotsPerRank + a.ddTrailerSlot))->w;
        if (level != 0) a.ddFlags[2] |= level;               // bit 2 (4): some rank's atom left the margin -- every rank ends the run at the same evaluation
        // This rank's list, as built last, fills 7/8 of its allocation: ask for a common re-sort NOW (level 3 travels in the trailer as
        // the drift levels do) -- the rebuild after a re-sort is verified by the host, which grows the allocation to 1.5 x the list.  An
        // overflow itself cannot be undone on a decomposed run (the other ranks have stepped on by the time they hear of it).
        if (a.state[ST_NUM_CHUNKS] > a.maxChunks / 8 * 7) atomicOr(&a.ddFlags[1], 3);
    }
    const int atom = a.atomOfSlot[sl];
    const bool valid = inRange && atom >= 0;
    float4 p = posqOut[sl];
    double xw = 0.0, yw = 0.0, zw = 0.0;               // the position in double: the float posq is its rounding
    if (valid) {
        if (a.posWire != nullptr) {
            // decomposed run: every slot -- own ones too, so that all ranks see the same numbers -- comes from the wire record,
            // already wrapped into the box: coefficients of the (reduced) box vectors, each in [0, 1)
            const uint4 u = a.posWire[sl];
            const double f = 1.0 / 4294967296.0;
            const double sx = (double) u.x * f, sy = (double) u.y * f, sz = (double) u.z * f;
            xw = sx * boxd.ax + sy * boxd.bx + sz * boxd.cx; yw = sy * boxd.by + sz * boxd.cy; zw = sz * boxd.cz;
            const bool own = sl >= a.firstBlock * OMM_TILE && sl < (a.firstBlock + a.ownedBlocks) * OMM_TILE;
            if (a.wireRef != nullptr && own && (a.ddGuardAtom == nullptr || a.ddGuardAtom[atom] != 0)) {
                // drift along the first box fraction (x in a rectangular box: the slabs are cut in it) since the re-sort; wrap-around
                // arithmetic of the 32-bit fractions = minimum image
                const int d = (int) (u.x - a.wireRef[sl].x);
                const unsigned ad = (unsigned) (d < 0 ? -d : d);
                if (ad > a.ddWarn) {
                    // [1]: what this rank tells the others through its trailer -- 1 near the margin, 3 when little of it is left
                    // (a re-sort that lags would come too late); [3]: the largest drift seen (diagnostics)
                    atomicOr(&a.ddFlags[1], ad - a.ddWarn > (a.ddMax - a.ddWarn) / 5 * 3 ? 3 : 1);
                    atomicMax(&a.ddFlags[3], (int) (ad >> 1));
                }
                if (ad > a.ddMax) { atomicOr(&a.ddFlags[1], 7); a.ddFlags[0] = 1; }      // [0]: it happened on THIS rank (diagnostics); the decision travels in the trailer
            }
            if (a.posScatter != nullptr && !own) {
                // atom-ordered copy of a foreign atom: the last known position moved by the minimum-image displacement
                double4 o = a.posScatter[atom];
                double dx = xw - o.x, dy = yw - o.y, dz = zw - o.z;
                { const double n = rint(dz / boxd.cz); dx -= n * boxd.cx; dy -= n * boxd.cy; dz -= n * boxd.cz; }
                { const double n = rint(dy / boxd.by); dx -= n * boxd.bx; dy -= n * boxd.by; }
                dx -= rint(dx / boxd.ax) * boxd.ax;
                o.x += dx; o.y += dy; o.z += dz;
                a.posScatter[atom] = o;
            }
        }
        else {
            const double4 x = pos[atom];
            const int4 w = wrap[atom];
            xw = x.x - (w.x * boxd.ax + w.y * boxd.bx + w.z * boxd.cx);
            yw = x.y - (w.y * boxd.by + w.z * boxd.cy);
            zw = x.z - (w.z * boxd.cz);
        }
        p.x = (float) xw; p.y = (float) yw; p.z = (float) zw;
    }
    else { p.x = 0.f; p.y = 0.f; p.z = 0.f; p.w = 0.f; }
    // displacement since the last rebuild
    bool moved = false;
    if (valid && checkDisplacement) {
        const float4 r = a.posqRef[sl];
        float dx = p.x - r.x, dy = p.y - r.y, dz = p.z - r.z;
        apply_pbc_rt(a.pbc, dx, dy, dz, a.box);
        moved = !(dx * dx + dy * dy + dz * dz <= a.maxDisp2);
    }
    if (__any(moved) && lane_id() == 0) atomicOr(&a.state[ST_REBUILD], 1);
    if (checkDisplacement) check_inner_displacement(a, sl, valid, p);
    // bounding box of the 32-atom block, relative to its first atom
    const float4 p0 = make_float4(__shfl(p.x, 0, 32), __shfl(p.y, 0, 32), __shfl(p.z, 0, 32), 0.f);
    float dx = p.x - p0.x, dy = p.y - p0.y, dz = p.z - p0.z;
    // Rectangular boxes: store every atom in the periodic image nearest to the first atom of its block, so the 32 atoms
    // of a block are mutually image-coherent (coordinates may leave [0, L) by a block width; every consumer either
    // reduces images itself or, like the pair kernel's single-image path, relies on exactly this coherence).  The shift
    // is applied to the double position, so posq stays the correctly rounded value of what posqRel is derived from.
    if (valid && a.pbc == 1) {
        xw -= (double) rintf(dx * a.box.invAx) * boxd.ax;
        yw -= (double) rintf(dy * a.box.invBy) * boxd.by;
        zw -= (double) rintf(dz * a.box.invCz) * boxd.cz;
        p.x = (float) xw; p.y = (float) yw; p.z = (float) zw;
        dx = p.x - p0.x; dy = p.y - p0.y; dz = p.z - p0.z;
    }
    else apply_pbc_rt(a.pbc, dx, dy, dz, a.box);
    if (!valid) { dx = dy = dz = 0; }
    if (inRange) posqOut[sl] = p;
    float minx = dx, maxx = dx, miny = dy, maxy = dy, minz = dz, maxz = dz;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        minx = fminf(minx, __shfl_xor(minx, m)); maxx = fmaxf(maxx, __shfl_xor(maxx, m));
        miny = fminf(miny, __shfl_xor(miny, m)); maxy = fmaxf(maxy, __shfl_xor(maxy, m));
        minz = fminf(minz, __shfl_xor(minz, m)); maxz = fmaxf(maxz, __shfl_xor(maxz, m));
    }
    // every lane of the block holds the same bounds, hence the same centre
    const float4 center = make_float4(p0.x + 0.5f * (minx + maxx), p0.y + 0.5f * (miny + maxy), p0.z + 0.5f * (minz + maxz), 0.f);
    // A block without atoms (padding at the end of a rank's slot range; padding slots are always trailing, so the first
    // slot decides) gets a hugely negative half extent: no block test against it can pass, from either side.
    const bool emptyBlock = __shfl((int) valid, 0, 32) == 0;
    if (inRange && (s & 31) == 0) {
        const int blk = s >> 5;
        a.blockCenter[blk] = center;
        // .w = 1: the block's atoms are image-coherent (written above); the pair kernel may then use one image per j atom
        a.blockHalf[blk] = emptyBlock ? make_float4(-1e30f, -1e30f, -1e30f, 0.f)
                                      : make_float4(0.5f * (maxx - minx), 0.5f * (maxy - miny), 0.5f * (maxz - minz), a.pbc == 1 ? 1.f : 0.f);
    }
    // Block-relative coordinates (double position minus the float centre, rounded once): what the pair kernel computes
    // with.  Their error is the rounding of a number below ~1 nm (6e-8 nm), independent of where in the box the block is.
    if (inRange && a.posqRel != nullptr) {
        const double rx = xw - (double) center.x, ry = yw - (double) center.y, rz = zw - (double) center.z;
        const float4 hi = valid ? make_float4((float) rx, (float) ry, (float) rz, p.w) : make_float4(0.f, 0.f, 0.f, 0.f);
        a.posqRel[sl] = hi;
        // ... and what that rounding dropped (<= 3e-8 nm): only the pair kernel's cutoff-edge path reads it
        if (a.posqRelLo != nullptr)
            a.posqRelLo[sl] = valid ? make_float4((float) (rx - (double) hi.x), (float) (ry - (double) hi.y), (float) (rz - (double) hi.z), 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

NlArgs make_nl_args(const ommhip_neighbor_list* nl) {
    NlArgs a;
    a.numAtoms = nl->num_atoms; a.paddedAtoms = nl->padded_atoms; a.numBlocks = nl->padded_atoms / OMM_TILE; a.maxChunks = nl->max_chunks;
    a.firstBlock = 0; a.ownedBlocks = a.numBlocks;
    if (nl->owned_blocks > 0 && nl->first_block >= 0 && nl->first_block + nl->owned_blocks <= a.numBlocks) { a.firstBlock = nl->first_block; a.ownedBlocks = nl->owned_blocks; }
    a.ddMode = nl->dd_mode != 0 && a.ownedBlocks < a.numBlocks ? 1 : 0;
    a.ddHalfShell = a.ddMode && nl->dd_half_shell != 0 ? 1 : 0; a.evalBlock0 = nl->dd_eval_slot0 / OMM_TILE; a.evalBlock1 = nl->dd_eval_slot1 / OMM_TILE;
    a.posWire = (const uint4*) nl->pos_wire; a.posScatter = (double4*) nl->pos_scatter; a.trace = nullptr;
    { static const bool nlXcd = getenv("OPENMM_HIP_NL_XCD") != nullptr; a.xcdAware = nlXcd ? 1 : 0; }          // opt-in: measured +0.7 % step time at 1M atoms (docs/EXPERIMENTS.md)
    a.numActive = 0; a.activeTotal = a.paddedAtoms;
    for (int r = 0; r < 4; r++) { a.activeBegin[r] = 0; a.activeEnd[r] = 0; }
    if (a.ddMode && nl->num_active_ranges > 0 && nl->num_active_ranges <= 4) {
        a.numActive = nl->num_active_ranges; a.activeTotal = 0;
        for (int r = 0; r < a.numActive; r++) { a.activeBegin[r] = nl->active_range[2 * r]; a.activeEnd[r] = nl->active_range[2 * r + 1]; a.activeTotal += a.activeEnd[r] - a.activeBegin[r]; }
    }
    a.ddGuardAtom = nl->dd_guard_atom;
    a.wireRef = (const uint4*) nl->wire_ref; a.ddWarn = nl->dd_warn; a.ddMax = nl->dd_max; a.ddFlags = nl->dd_flags;
    a.ddRanks = nl->dd_ranks; a.ddSlotsPerRank = nl->dd_slots_per_rank; a.ddTrailerSlot = nl->dd_trailer_slot;
    if (a.ddFlags == nullptr || !a.ddMode) { a.wireRef = nullptr; a.ddFlags = nullptr; }
    a.pbc = nl->pbc;
    double rl = nl->cutoff + nl->padding;
    a.listCutoff2 = nl->cutoff > 0 ? (float) (rl * rl) : INFINITY;
    a.maxDisp2 = (float) (0.25 * nl->padding * nl->padding);
    a.box = make_box(nl->box);
    a.posq = (const float4*) nl->posq; a.posqRef = (float4*) nl->posq_ref; a.posqRel = (float4*) nl->posq_rel; a.posqRelLo = (float4*) nl->posq_rel_lo;
    a.atomOfSlot = nl->atom_of_slot; a.slotOfAtom = nl->slot_of_atom;
    a.exclStart = nl->excl_start; a.exclAtoms = nl->excl_atoms; a.exclBlockRange = (const int2*) nl->excl_block_range;
    a.exclSlotStart = nl->excl_slot_start; a.exclSlots = nl->excl_slots;
    a.state = nl->state;
    a.blockCenter = (float4*) nl->block_center; a.blockHalf = (float4*) nl->block_half;
    a.chunkInfo = (int2*) nl->chunk_info; a.rowJ = nl->row_j; a.rowMask = nl->row_mask;
    a.chunkInfoInner = (int2*) nl->chunk_info_inner; a.rowJInner = nl->row_j_inner; a.rowMaskInner = nl->row_mask_inner; a.blockRuns = nl->block_runs;
    a.cutoff = (float) nl->cutoff; a.pruneCutoff2 = (float) ((nl->cutoff + nl->inner_padding) * (nl->cutoff + nl->inner_padding) * (1.0 + 1e-4));
    a.posqRefInner = (float4*) nl->posq_ref_inner; a.maxDispInner2 = (float) (0.25 * nl->inner_padding * nl->inner_padding);
    if (!list_is_pruned(nl)) {
        a.chunkInfoInner = nullptr; a.rowJInner = nullptr; a.rowMaskInner = nullptr; a.blockRuns = nullptr; a.posqRefInner = nullptr;
    }
    // cell mode: rectangular periodic boxes with enough blocks for the all-blocks scan to hurt
    a.cellMode = 0; a.bigHalf = 0.f; a.ncx = a.ncy = a.ncz = 1; a.cellInvX = a.cellInvY = a.cellInvZ = 0.f;
    a.cellStart = nl->cell_start; a.cellBlocks = nl->cell_blocks; a.cellMeta = nl->cell_meta; a.cellBoxes = (float4*) nl->cell_boxes;
    // measured on MI355X: 20 % off the rebuild at 30 798 blocks (1M atoms), but a loss at 3 072 blocks, where scanning all
    // blocks is only a dozen coalesced sweeps per workgroup
    const int minBlocks = nl->cell_min_blocks > 0 ? nl->cell_min_blocks : 16384;
    if (nl->cell_start != nullptr && nl->cell_boxes != nullptr && nl->max_cells > 0 && nl->pbc == 1 && nl->cutoff > 0 && a.numBlocks >= minBlocks) {
        int n[3];
        const double L[3] = {nl->box[0], nl->box[2], nl->box[5]};
        for (int d = 0; d < 3; d++) { n[d] = (int) floor(L[d] / rl); if (n[d] < 1) n[d] = 1; }
        while ((long long) n[0] * n[1] * n[2] > (nl->max_cells < NL_BIN_CELLS ? nl->max_cells : NL_BIN_CELLS)) {
            const int big = n[0] >= n[1] && n[0] >= n[2] ? 0 : (n[1] >= n[2] ? 1 : 2);
            n[big]--;
        }
        a.cellMode = 1; a.ncx = n[0]; a.ncy = n[1]; a.ncz = n[2];
        a.bigHalf = (float) (NL_BIG_HALF * rl);
        a.cellInvX = (float) (n[0] / L[0]); a.cellInvY = (float) (n[1] / L[1]); a.cellInvZ = (float) (n[2] / L[2]);
    }
    return a;
}

}  // namespace

static void launch_find(const NlArgs& ain, hipStream_t st) {
    NlArgs a = ain;
    a.trace = nullptr;
#ifndef OMMHIP_EMU
    static const bool tracing = getenv("OPENMM_HIP_NL_TRACE") != nullptr;
    static long long* traceBuf = nullptr;
    if (tracing && a.ownedBlocks <= 65536) {
        if (traceBuf == nullptr) hipMalloc((void**) &traceBuf, sizeof(long long) * 2 * 65536);
        hipMemsetAsync(traceBuf, 0, sizeof(long long) * 2 * a.ownedBlocks, st);
        a.trace = traceBuf;
    }
#endif
    if (a.cellMode) hipLaunchKernelGGL(nl_bin_blocks, dim3(1), dim3(1024), 0, st, a);
    static const int residentPerCu = getenv("OPENMM_HIP_NL_PERSISTENT") != nullptr ? atoi(getenv("OPENMM_HIP_NL_PERSISTENT")) : 8;     // 0: one workgroup per i-block, as before
    static int numCus = 0;
    if (residentPerCu > 0 && numCus == 0) { hipDeviceProp_t prop; int dev = 0; hipGetDevice(&dev); numCus = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256; }
    if (residentPerCu > 0 && a.ownedBlocks > residentPerCu * numCus && a.trace == nullptr) {
        const dim3 grid(residentPerCu * numCus);
        if (a.pbc == 0) hipLaunchKernelGGL(nl_find_interactions_resident<0>, grid, dim3(NL_THREADS), 0, st, a);
        else if (a.pbc == 1) hipLaunchKernelGGL(nl_find_interactions_resident<1>, grid, dim3(NL_THREADS), 0, st, a);
        else hipLaunchKernelGGL(nl_find_interactions_resident<2>, grid, dim3(NL_THREADS), 0, st, a);
    }
    else if (a.pbc == 0) hipLaunchKernelGGL(nl_find_interactions<0>, dim3(a.ownedBlocks), dim3(NL_THREADS), 0, st, a);
    else if (a.pbc == 1) hipLaunchKernelGGL(nl_find_interactions<1>, dim3(a.ownedBlocks), dim3(NL_THREADS), 0, st, a);
    else hipLaunchKernelGGL(nl_find_interactions<2>, dim3(a.ownedBlocks), dim3(NL_THREADS), 0, st, a);
#ifndef OMMHIP_EMU
    if (a.trace != nullptr) {
        // wall_clock64 ticks at 100 MHz.  Printed only for launches that rebuilt the list.
        const int n = a.ownedBlocks;
        std::vector<long long> h(2 * (size_t) n);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), a.trace, sizeof(long long) * 2 * n, hipMemcpyDeviceToHost);
        if (h[0] != 0) {
            long long t0 = h[0], t1 = h[1], life = 0;
            std::vector<long long> lives(n);
            for (int b = 0; b < n; b++) { t0 = std::min(t0, h[2 * b]); t1 = std::max(t1, h[2 * b + 1]); lives[b] = h[2 * b + 1] - h[2 * b]; life += lives[b]; }
            std::vector<long long> sorted(lives); std::sort(sorted.begin(), sorted.end());
            long long lastStart = 0; for (int b = 0; b < n; b++) lastStart = std::max(lastStart, h[2 * b] - t0);
            fprintf(stderr, "nl_find trace: %d workgroups, span %.1f us, last start at %.1f us, life mean %.1f median %.1f p99 %.1f max %.1f us\n", n, (t1 - t0) * 0.01, lastStart * 0.01,
                    life * 0.01 / n, sorted[n / 2] * 0.01, sorted[(size_t) n * 99 / 100] * 0.01, sorted[n - 1] * 0.01);
        }
    }
#endif
    launch_prune(a, st);
}

// Per-step entry of the platform: conversion + displacement check + bounds in one launch, then the
// (device-conditional) rebuild.  Two launches per step, no host synchronisation.
extern "C" int ommhip_nl_step(const ommhip_neighbor_list* nl, const void* pos_d, const void* wrap_d, void* stream) {
    return ommhip_nl_step_clear(nl, pos_d, wrap_d, nullptr, 0, nullptr, 0, stream);
}

extern "C" int ommhip_nl_step_clear(const ommhip_neighbor_list* nl, const void* pos_d, const void* wrap_d,
                                    void* clear_a_d, size_t a_bytes, void* clear_b_d, size_t b_bytes, void* stream) {
    int rc = ommhip_nl_prepare(nl, pos_d, wrap_d, clear_a_d, a_bytes, clear_b_d, b_bytes, stream);
    if (rc == 0) rc = ommhip_nl_rebuild_if_requested(nl, stream);
    return rc;
}

extern "C" int ommhip_nl_prepare(const ommhip_neighbor_list* nl, const void* pos_d, const void* wrap_d,
                                 void* clear_a_d, size_t a_bytes, void* clear_b_d, size_t b_bytes, void* stream) {
    if ((a_bytes | b_bytes) & 15) return 1;
    hipStream_t st = (hipStream_t) stream;
    NlArgs a = make_nl_args(nl);
    BoxD bd;
    bd.ax = nl->box[0]; bd.bx = nl->box[1]; bd.by = nl->box[2]; bd.cx = nl->box[3]; bd.cy = nl->box[4]; bd.cz = nl->box[5];
    hipLaunchKernelGGL(nl_prepare, dim3(((a.numActive == 0 ? a.paddedAtoms : a.activeTotal) + 255) / 256), dim3(256), 0, st, a, (const double4*) pos_d, (const int4*) wrap_d, bd,
                       (float4*) nl->posq, nl->cutoff > 0 ? 1 : 0,
                       (uint4*) clear_a_d, clear_a_d != nullptr ? a_bytes / 16 : 0, (uint4*) clear_b_d, clear_b_d != nullptr ? b_bytes / 16 : 0);
    return (int) hipGetLastError();
}

extern "C" int ommhip_nl_rebuild_if_requested(const ommhip_neighbor_list* nl, void* stream) {
    NlArgs a = make_nl_args(nl);
    ommhip_profile_begin(OMMHIP_TIMER_NL_UPDATE, stream);
    launch_find(a, (hipStream_t) stream);
    ommhip_profile_end(OMMHIP_TIMER_NL_UPDATE, stream);
    return (int) hipGetLastError();
}

extern "C" int ommhip_nl_update(const ommhip_neighbor_list* nl, void* stream) {
    hipStream_t st = (hipStream_t) stream;
    NlArgs a = make_nl_args(nl);
    ommhip_profile_begin(OMMHIP_TIMER_NL_UPDATE, stream);
    if (nl->cutoff > 0)    // NoCutoff lists never go stale through motion
        hipLaunchKernelGGL(nl_check_displacement, dim3((a.paddedAtoms + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(nl_block_bounds, dim3((a.paddedAtoms + 255) / 256), dim3(256), 0, st, a);
    launch_find(a, st);
    ommhip_profile_end(OMMHIP_TIMER_NL_UPDATE, stream);
    return (int) hipGetLastError();
}

// Per-atom pair lists of the AMOEBA kernels (amoeba.hip: vdW, amoeba_multipole.hip: multipoles): built for the cutoff plus a skin and
// kept until some atom has moved by half the skin (the device decides: pl_check_body in pl_prepare), the consumers re-testing the cutoff per pair.
//
// The AMOEBA pair terms are long (a multipole pair is three derivative chains with erfc / exp and ~1 000 double-precision operations), so
// what matters is that a wavefront only ever executes them for pairs inside the cutoff.  A scan "one thread per atom i, every thread
// of the wave looks at the same candidate j" -- the first version of these kernels -- runs the pair arithmetic whenever ANY of its 64
// lanes has j inside the cutoff: with a spatially sorted order that is nearly every candidate of the neighbouring tiles, at 2-4 % of
// the lanes.  Here the scan only TESTS distances (cheap) and writes, per atom, the list of its partners; the physics kernels then walk
// their own lists, lane by lane, each iteration a pair inside the cutoff on (nearly) every lane.  The geometry is fixed within an
// evaluation, so the list serves every kernel of the evaluation -- the induced-dipole field is evaluated ~10 times per step.
//
// Scan order: position g holds atom order[g] (the platform's slot order: Hilbert-sorted 32-atom blocks; -1 = padding), or g itself
// without an order.  Tiles of PL_BLOCK positions carry bounding boxes; the builder skips the tiles farther from its own than the cutoff
// (rectangular boxes).  An entry is  j's scan position | (1 + index of j in i's row of listed partners) << 24 :  the multipole kernels
// find the scale factors of a covalently related pair through that index, the vdW kernel leaves listed partners (exclusions) out.
// Layout: PL_PARTS sub-lists per position -- the candidate tiles are dealt out to PL_PARTS workgroups per tile of owners, or the
// builder would run on 2 wavefronts per 128 atoms --; entry k of sub-list p of position g at list[(p * subcap + k) * stride + g]
// (coalesced across the lanes of a wave), its length at count[p * stride + g].
#ifndef OMM_AMOEBA_PAIRS_H_
#define OMM_AMOEBA_PAIRS_H_

#include "common.h"
#include <cstdio>
#include <cstdlib>

namespace omm {

#define PL_BLOCK 128
#define PL_POS_MASK 0xFFFFFF
#define PL_MAX_ROW 126            // listed partners per atom that an entry can index
#define PL_PARTS 4
#define PL_FAR_CHUNK 4096          // tiles whose far / near flags are held in LDS at a time
#define PL_ROW_LDS 40              // listed partners per atom that the builder keeps in LDS (longer rows go on in global memory)

struct PairListArgs {
    int n, numScan, skipTiles, subcap, stride, excludeListed, debug;      // subcap: entries per sub-list
    const double4* pos; const int* order; const int* slotOfAtom;
    BoxD box; double cutoff2;
    double4* tileCenter; double4* tileHalf;
    const int* rowStart; const int* rowAtom;       // listed partners per atom (CSR by atom, ascending atom index)
    int* rowPos;                                   // work array: the same rows as scan positions, ascending
    double4* rowData; const double4* rowDataIn;    // optional payload carried along when the rows are re-sorted (multipole scale factors)
    int* list; int* count; int* overflow;
    long long* trace;                              // profiling (OPENMM_HIP_PL_DEBUG & 4): per workgroup start and end clock, hardware id
    // Verlet skin (optional: state == nullptr rebuilds at every call).  cutoff2 above is then the LIST radius squared, (cutoff + skin)^2.
    // state[0]: rebuild needed (pl_prepare: an atom moved more than skin / 2 since the positions in refPos, or the caller forces it),
    // state[1]: ticket of pl_finish, state[2]: rebuilds so far.
    double4* refPos; int* state; double skinHalf2; int forceRebuild;
};

__device__ __forceinline__ int pl_scan_atom(const PairListArgs& a, int g) { return g < a.numScan ? (a.order != nullptr ? a.order[g] : g) : -1; }

namespace {      // kernels with internal linkage: the header is included by two translation units

// rows of listed partners re-keyed to scan positions and sorted (insertion sort: rows hold the bonded neighbourhood of an atom)
__global__ void pl_sort_rows(PairListArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int b = a.rowStart[i], e = a.rowStart[i + 1];
    for (int c = b; c < e; c++) {
        const int partner = a.rowAtom[c];
        const int pos = a.order != nullptr ? a.slotOfAtom[partner] : partner;
        double4 v = make_double4(0, 0, 0, 0);
        if (a.rowData != nullptr) v = a.rowDataIn[c];
        int k = c;
        while (k > b && a.rowPos[k - 1] > pos) { a.rowPos[k] = a.rowPos[k - 1]; if (a.rowData != nullptr) a.rowData[k] = a.rowData[k - 1]; k--; }
        a.rowPos[k] = pos;
        if (a.rowData != nullptr) a.rowData[k] = v;
    }
}

// Does the list have to be rebuilt?  One thread per atom: moved by more than half the skin since the list was built (plain displacement:
// the atom-ordered positions are continuous between re-sorts; a jump by a box vector simply asks for a rebuild).
__device__ __forceinline__ void pl_check_body(const PairListArgs& a, int i) {
    if (i == 0 && a.forceRebuild) a.state[0] = 1;
    if (i >= a.n || a.forceRebuild) return;
    const double4 p = a.pos[i], r = a.refPos[i];
    const double dx = p.x - r.x, dy = p.y - r.y, dz = p.z - r.z;
    if (dx * dx + dy * dy + dz * dz > a.skinHalf2) a.state[0] = 1;
}

// After a rebuild that fitted: the positions it was made for, and the request taken back by the block that finishes last.
__global__ void pl_finish(PairListArgs a) {
    if (a.state[0] == 0) return;
    const bool fitted = *a.overflow == 0;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (fitted && i < a.n) a.refPos[i] = a.pos[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&a.state[1], 1) == (int) gridDim.x - 1) {
            a.state[1] = 0;
            __threadfence();
            if (fitted) { a.state[2]++; a.state[0] = 0; }       // a list that did not fit stays requested: the caller grows it and calls again
        }
    }
}

// bounding boxes of the tiles (nearest images relative to the tile's first atom; a tile without atoms gets half extents of -1e30)
// One launch in front of the builder (round 5; three before): workgroups [0, tiles) bound the tiles -- whether or not the list will be rebuilt:
// that is being decided by the workgroups behind them, which look at the displacements (pl_check_body); thread 0 clears the overflow word.
__global__ __launch_bounds__(PL_BLOCK) void pl_prepare(PairListArgs a, int tiles) {
    __shared__ double lo[3][PL_BLOCK], hi[3][PL_BLOCK];
    __shared__ int firstValid;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.overflow = 0;
    if ((int) blockIdx.x >= tiles) { if (a.state != nullptr) pl_check_body(a, ((int) blockIdx.x - tiles) * PL_BLOCK + (int) threadIdx.x); return; }
    const int t = threadIdx.x, g = blockIdx.x * PL_BLOCK + t, i = pl_scan_atom(a, g);
    if (t == 0) firstValid = PL_BLOCK;
    __syncthreads();
    if (i >= 0) atomicMin(&firstValid, t);
    __syncthreads();
    if (firstValid == PL_BLOCK) {
        if (t == 0) { a.tileCenter[blockIdx.x] = make_double4(0, 0, 0, 0); a.tileHalf[blockIdx.x] = make_double4(-1e30, -1e30, -1e30, 0); }
        return;
    }
    const double4 ref = a.pos[pl_scan_atom(a, blockIdx.x * PL_BLOCK + firstValid)];
    double d[3] = {0, 0, 0};
    if (i >= 0) {
        const double4 p = a.pos[i];
        d[0] = p.x - ref.x; d[1] = p.y - ref.y; d[2] = p.z - ref.z;
        min_image_d(d[0], d[1], d[2], a.box);
    }
    for (int k = 0; k < 3; k++) { lo[k][t] = d[k]; hi[k][t] = d[k]; }
    __syncthreads();
    for (int m = PL_BLOCK / 2; m >= 1; m >>= 1) {
        if (t < m) for (int k = 0; k < 3; k++) { lo[k][t] = fmin(lo[k][t], lo[k][t + m]); hi[k][t] = fmax(hi[k][t], hi[k][t + m]); }
        __syncthreads();
    }
    if (t == 0) {
        a.tileCenter[blockIdx.x] = make_double4(ref.x + 0.5 * (lo[0][0] + hi[0][0]), ref.y + 0.5 * (lo[1][0] + hi[1][0]), ref.z + 0.5 * (lo[2][0] + hi[2][0]), 0);
        a.tileHalf[blockIdx.x] = make_double4(0.5 * (hi[0][0] - lo[0][0]), 0.5 * (hi[1][0] - lo[1][0]), 0.5 * (hi[2][0] - lo[2][0]), 0);
    }
}

__device__ __forceinline__ bool pl_tiles_far(const PairListArgs& a, int ti, int tj) {
    if (!a.skipTiles) return false;
    const double4 ci = a.tileCenter[ti], hi = a.tileHalf[ti], cj = a.tileCenter[tj], hj = a.tileHalf[tj];
    double dx = cj.x - ci.x, dy = cj.y - ci.y, dz = cj.z - ci.z;
    dx -= rint(dx / a.box.ax) * a.box.ax; dy -= rint(dy / a.box.by) * a.box.by; dz -= rint(dz / a.box.cz) * a.box.cz;
    const double gx = fmax(fabs(dx) - hi.x - hj.x, 0.0), gy = fmax(fabs(dy) - hi.y - hj.y, 0.0), gz = fmax(fabs(dz) - hi.z - hj.z, 0.0);
    return gx * gx + gy * gy + gz * gz > a.cutoff2;
}

// One thread per scan position: partners within the cutoff (cutoff2 < 0: all atoms), in ascending scan position.
// The distance test runs once per candidate and thread, so it is kept to a subtraction and a comparison: where the workgroup's own
// tile (half extents h) satisfies h + cutoff < L / 2 on every axis -- rectangular boxes -- the staging thread moves each candidate
// to the periodic image nearest to the tile's centre c (one minimum-image reduction per staged atom, not one per pair), the own atom
// is expressed in the same frame, and the image of j nearest to c is then the only one that can lie within the cutoff of any atom
// of the tile.  Other boxes and oversized tiles reduce every pair (min_image_d).
__global__ __launch_bounds__(PL_BLOCK) void pl_build(PairListArgs a) {
    if (a.state != nullptr && a.state[0] == 0) return;             // the list of an earlier call is still good
#ifndef OMMHIP_EMU
    if (a.trace != nullptr && threadIdx.x == 0) { a.trace[3 * blockIdx.x] = (long long) wall_clock64(); a.trace[3 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long) (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf) << 32); }
#endif
    __shared__ double4 sj[PL_BLOCK];               // .w < 0: no atom at this position
    __shared__ unsigned sNear[PL_FAR_CHUNK / 32];  // bit per candidate tile of the current chunk: within reach of this workgroup's tile
    const int tileI = blockIdx.x / PL_PARTS, part = blockIdx.x % PL_PARTS;
    const int t = threadIdx.x, g = tileI * PL_BLOCK + t, i = pl_scan_atom(a, g);
    const bool active = i >= 0;
    const int ii = active ? i : 0;
    double4 xi = a.pos[ii];
    bool tileFrame = false;
    double4 c = make_double4(0, 0, 0, 0);
    if (a.skipTiles && a.cutoff2 >= 0.0) {
        const double4 h = a.tileHalf[tileI];
        const double rc = sqrt(a.cutoff2);
        c = a.tileCenter[tileI];
        tileFrame = h.x >= 0.0 && h.x + rc < 0.5 * a.box.ax && h.y + rc < 0.5 * a.box.by && h.z + rc < 0.5 * a.box.cz;       // block-uniform
        if (tileFrame) {
            double dx = xi.x - c.x, dy = xi.y - c.y, dz = xi.z - c.z;
            min_image_d(dx, dy, dz, a.box);
            xi.x = dx; xi.y = dy; xi.z = dz;                      // relative to the tile's centre
        }
    }
    const int rowBegin = a.rowStart != nullptr ? a.rowStart[ii] : 0, rowEnd = a.rowStart != nullptr ? a.rowStart[ii + 1] : 0;
    // The atom's listed partners, in LDS: a global load inside the candidate loop would have to wait for every list store issued before
    // it (loads and stores share one in-order counter on this chip) -- a memory round trip per partner and lane, serialised over the wave.
    __shared__ int sRow[PL_ROW_LDS][PL_BLOCK];
    for (int cc = 0; cc < PL_ROW_LDS && rowBegin + cc < rowEnd; cc++) sRow[cc][t] = a.rowPos[rowBegin + cc];
#define PL_PARTNER(cur) ((cur) < rowEnd ? ((cur) - rowBegin < PL_ROW_LDS ? sRow[(cur) - rowBegin][t] : a.rowPos[cur]) : 0x7fffffff)
    int cursor = rowBegin;
    int next = PL_PARTNER(cursor);
    int cnt = 0;                                   // partners found so far; what exceeds the sub-list's capacity is counted, not stored
    bool tagOver = false;                          // a listed partner beyond the 126 that an entry can index
    int* const myList = a.list + (size_t) part * a.subcap * a.stride + g;
    const int numTiles = (a.numScan + PL_BLOCK - 1) / PL_BLOCK;
    // A staged tile is four 32-slot blocks of the platform's spatial order -- compact clouds of ~0.7 nm.  Their bounding boxes (in the tile
    // frame) let a wavefront leave out a whole block that none of its 64 owners can reach: the tile-against-tile test above passes ~6 000
    // candidates per owner for the ~400 inside a 0.95 nm list radius, the block test brings that to ~1 500 (round 5).
    // In the tile frame the tests run in FLOAT on structure-of-arrays copies (coordinates relative to the tile centre are a few nm: 2e-7 nm
    // of rounding), two candidates per packed instruction, against the list radius widened by 1e-5 -- the list is a superset of the double
    // one by a hair, and every consumer re-tests the cutoff in double anyway.  An empty position holds 1e30: it fails the test by itself.
    __shared__ __attribute__((aligned(16))) float sxf[PL_BLOCK], syf[PL_BLOCK], szf[PL_BLOCK];
    __shared__ float sbLo[PL_BLOCK / 32][3], sbHi[PL_BLOCK / 32][3];
    const float xf = (float) xi.x, yf = (float) xi.y, zf = (float) xi.z;
    const float radius2f = (float) (a.cutoff2 * (1.0 + 1e-5)), boxRadius2f = (float) (a.cutoff2 * (1.0 + 3e-5));
    const bool floatFrame = tileFrame && a.cutoff2 >= 0.0;               // block-uniform
    // the candidate tile after the current one is loaded while the current one is tested (a staging is two dependent global loads and
    // the kernel runs at one or two wavefronts per SIMD: nothing else would cover them)
    auto stage_load = [&](int tj) -> double4 {              // the loads only: what is done with them waits until the tile is its turn
        const int j = pl_scan_atom(a, tj * PL_BLOCK + t);
        double4 p = make_double4(0, 0, 0, -1.0);
        if (j >= 0) { p = a.pos[j]; p.w = 1.0; }
        return p;
    };
    for (int chunk = 0; chunk < numTiles; chunk += PL_FAR_CHUNK) {
        // which tiles of this chunk are this workgroup's to look at: its share (every PL_PARTS-th) of those within reach -- all threads
        // test in parallel (one tile each per round) instead of every thread loading every tile's box in turn
        __syncthreads();
        for (int w = t; w < PL_FAR_CHUNK / 32; w += PL_BLOCK) sNear[w] = 0u;
        __syncthreads();
        const int chunkTiles = min(PL_FAR_CHUNK, numTiles - chunk);
        for (int tj = chunk + t; tj < chunk + chunkTiles; tj += PL_BLOCK)
            if (tj % PL_PARTS == part && !pl_tiles_far(a, tileI, tj)) atomicOr(&sNear[(tj - chunk) >> 5], 1u << ((tj - chunk) & 31));
        __syncthreads();
        const int words = (chunkTiles + 31) / 32;
        int w = 0;
        unsigned bits = sNear[0];                                        // block-uniform, as everything that steers the loop
        auto next_tile = [&]() -> int {
            while (bits == 0u) { if (++w >= words) return -1; bits = sNear[w]; }
            const int b = __ffs((int) bits) - 1;
            bits &= bits - 1u;
            return chunk + w * 32 + b;
        };
        int tileJ = next_tile();
        double4 staged = make_double4(0, 0, 0, -1.0);
        if (tileJ >= 0) staged = stage_load(tileJ);
        while (tileJ >= 0) {
            const int j0 = tileJ * PL_BLOCK;
            const int tileNext = next_tile();
            if (tileFrame && staged.w >= 0.0) {
                double dx = staged.x - c.x, dy = staged.y - c.y, dz = staged.z - c.z;
                min_image_d(dx, dy, dz, a.box);
                staged.x = dx; staged.y = dy; staged.z = dz;
            }
            __syncthreads();
            sj[t] = staged;
            if (floatFrame) {
                const bool here = staged.w >= 0.0;
                const float px = here ? (float) staged.x : 1e30f, py = here ? (float) staged.y : 1e30f, pz = here ? (float) staged.z : 1e30f;
                sxf[t] = px; syf[t] = py; szf[t] = pz;
                float lo[3] = {px, py, pz}, hi[3] = {here ? px : -1e30f, here ? py : -1e30f, here ? pz : -1e30f};
#pragma unroll
                for (int m = 16; m >= 1; m >>= 1)
#pragma unroll
                    for (int d = 0; d < 3; d++) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], m)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], m)); }
                if ((t & 31) == 0) for (int d = 0; d < 3; d++) { sbLo[t >> 5][d] = lo[d]; sbHi[t >> 5][d] = hi[d]; }
            }
            __syncthreads();
            if (tileNext >= 0) staged = stage_load(tileNext);            // in flight during the tests below
            tileJ = tileNext;
            auto append = [&](int j) {
                while (next < j) { cursor++; next = PL_PARTNER(cursor); }
                int tag = 0;
                if (next == j) {
                    if (a.excludeListed) return;
                    tag = cursor - rowBegin + 1;
                    if (tag > PL_MAX_ROW) tagOver = true;
                }
                if (cnt < a.subcap && !(a.debug & 2)) myList[(size_t) cnt * a.stride] = j | (tag << 24);          // (debug 2: profiling without the stores)
                cnt++;
            };
            if (floatFrame) {
                const v2f x2 = bc2(xf), y2 = bc2(yf), z2 = bc2(zf);
                const int self = j0 == tileI * PL_BLOCK ? t : -1;       // the own position, when the own tile is the candidate
                for (int kb = 0; kb < PL_BLOCK; kb += 32) {
                    // distance from the own atom to the block's box: no owner of this wavefront within the list radius -> the block is left out
                    const int sb = kb >> 5;
                    const float gx = fmaxf(fmaxf(sbLo[sb][0] - xf, xf - sbHi[sb][0]), 0.f), gy = fmaxf(fmaxf(sbLo[sb][1] - yf, yf - sbHi[sb][1]), 0.f),
                                gz = fmaxf(fmaxf(sbLo[sb][2] - zf, zf - sbHi[sb][2]), 0.f);
                    if (!__any(active && r2_of(gx, gy, gz) <= boxRadius2f)) continue;          // (every lane takes part: wave-uniform control flow)
                    // eight candidates per round, two per packed instruction; the partners in the block as a bit mask, appended in ascending order
                    // (one append loop per block, not per round: a wavefront runs it as often as its busiest lane has partners)
                    unsigned blockMask = 0u;
#pragma unroll
                    for (int k = kb; k < kb + 32; k += 8) {
                        const float4 xa = *(const float4*) &sxf[k], xb = *(const float4*) &sxf[k + 4], ya = *(const float4*) &syf[k], yb = *(const float4*) &syf[k + 4],
                                     za = *(const float4*) &szf[k], zb = *(const float4*) &szf[k + 4];
                        const v2f d0 = r2_of(mk2(xa.x, xa.y) - x2, mk2(ya.x, ya.y) - y2, mk2(za.x, za.y) - z2), d1 = r2_of(mk2(xa.z, xa.w) - x2, mk2(ya.z, ya.w) - y2, mk2(za.z, za.w) - z2),
                                  d2 = r2_of(mk2(xb.x, xb.y) - x2, mk2(yb.x, yb.y) - y2, mk2(zb.x, zb.y) - z2), d3 = r2_of(mk2(xb.z, xb.w) - x2, mk2(yb.z, yb.w) - y2, mk2(zb.z, zb.w) - z2);
                        unsigned mask = (d0.x <= radius2f ? 1u : 0u) | (d0.y <= radius2f ? 2u : 0u) | (d1.x <= radius2f ? 4u : 0u) | (d1.y <= radius2f ? 8u : 0u)
                                      | (d2.x <= radius2f ? 16u : 0u) | (d2.y <= radius2f ? 32u : 0u) | (d3.x <= radius2f ? 64u : 0u) | (d3.y <= radius2f ? 128u : 0u);
                        blockMask |= mask << (k - kb);
                    }
                    if ((unsigned) (self - kb) < 32u) blockMask &= ~(1u << (self - kb));
                    if (!active || (a.debug & 1)) blockMask = 0u;                                 // (debug 1: profiling without the appends)
                    while (blockMask != 0u) {
                        const int u = __ffs((int) blockMask) - 1;
                        blockMask &= blockMask - 1u;
                        append(j0 + kb + u);
                    }
                }
                continue;
            }
            // boxes the tile frame does not serve (triclinic, or a tile too large for the box), and the scan without a cutoff: double, every pair
            // reduced to its nearest image; four candidates per round
            const int nj = min(PL_BLOCK, a.numScan - j0);
            for (int k = 0; k < nj; k += 4) {
                double4 p[4];
#pragma unroll
                for (int u = 0; u < 4; u++) p[u] = sj[min(k + u, PL_BLOCK - 1)];
                bool inside[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int j = j0 + k + u;
                    inside[u] = active && k + u < nj && j != g && p[u].w >= 0.0;
                    if (a.cutoff2 >= 0.0) {
                        double dx = p[u].x - xi.x, dy = p[u].y - xi.y, dz = p[u].z - xi.z;
                        if (!tileFrame) min_image_d(dx, dy, dz, a.box);
                        inside[u] = inside[u] && !(dx * dx + dy * dy + dz * dz > a.cutoff2);
                    }
                }
                if ((a.debug & 1) && p[0].x != 12345.0) continue;          // profiling: no appends
                if (!(inside[0] || inside[1] || inside[2] || inside[3])) continue;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (inside[u]) append(j0 + k + u);
            }
        }
    }
#undef PL_PARTNER
#ifndef OMMHIP_EMU
    if (a.trace != nullptr && threadIdx.x == 0) a.trace[3 * blockIdx.x + 1] = (long long) wall_clock64();
#endif
    if (g < a.numScan) a.count[(size_t) part * a.stride + g] = min(cnt, a.subcap);
    if (cnt > a.subcap) atomicMax(a.overflow, cnt);                           // the sub-list length that would have been enough
    if (tagOver) atomicMax(a.overflow, 0x7fffffff);                           // or: a row too long to index
}

}  // namespace

// Consumers walk the four sub-lists as ONE list (a loop per sub-list would run to the longest sub-list of the wave four times over):
//     const PlSpan span = pl_span(count, stride, g);   for (int k = 0; k < span.total; k++) entry = pl_at(list, stride, subcap, span, k, g);
struct PlSpan { int c1, c2, c3, total; };          // where sub-lists 1, 2, 3 begin in the concatenation, and its length
__device__ __forceinline__ PlSpan pl_span(const int* count, int stride, int g) {
    PlSpan s;
    s.c1 = count[g]; s.c2 = s.c1 + count[(size_t) stride + g]; s.c3 = s.c2 + count[2 * (size_t) stride + g]; s.total = s.c3 + count[3 * (size_t) stride + g];
    return s;
}
__device__ __forceinline__ int pl_at(const int* list, int stride, int subcap, const PlSpan& s, int k, int g) {
    const int p = (k >= s.c1) + (k >= s.c2) + (k >= s.c3);
    const int kk = k - (p == 0 ? 0 : (p == 1 ? s.c1 : (p == 2 ? s.c2 : s.c3)));
    return list[((size_t) p * subcap + kk) * stride + g];
}

// Host side: re-key the rows, bound the tiles, build the list.  Returns 0, a hipError_t, or -2 when the list did not fit
// (*needed = the longest list, or 0x7fffffff when a row of listed partners is too long to index).  `overflowHost` = pinned or plain host int.
// `between` (optional) runs after the builder's kernels have been enqueued and before the host waits for the overflow word: work that does not
// need the lists goes there, on this stream or another.
template <class Between>
static inline int pl_launch(PairListArgs a, int* needed, hipStream_t st, int* buildsHost, Between between, int* deferred = nullptr, bool deferredCopies = true) {
    if (a.numScan > PL_POS_MASK) return 1;
    static const int debugMode = getenv("OPENMM_HIP_PL_DEBUG") != nullptr ? atoi(getenv("OPENMM_HIP_PL_DEBUG")) : 0;     // profiling only: wrong results
    a.debug = debugMode;
    if (a.state == nullptr || a.refPos == nullptr) { a.state = nullptr; a.refPos = nullptr; a.forceRebuild = 1; }
    const int tiles = (a.numScan + PL_BLOCK - 1) / PL_BLOCK;
    {
        const int boundTiles = a.skipTiles ? tiles : 0, checkBlocks = a.state != nullptr ? (a.n + PL_BLOCK - 1) / PL_BLOCK : 0;
        hipLaunchKernelGGL(pl_prepare, dim3(boundTiles + checkBlocks > 0 ? boundTiles + checkBlocks : 1), dim3(PL_BLOCK), 0, st, a, boundTiles);
    }
    // the rows of listed partners depend on the slot order and the parameters only: the caller forces a rebuild when either changed
    if (a.rowStart != nullptr && a.forceRebuild) hipLaunchKernelGGL(pl_sort_rows, dim3((a.n + 127) / 128), dim3(128), 0, st, a);
    a.trace = nullptr;
#ifndef OMMHIP_EMU
    static long long* traceBuf = nullptr;
    if (debugMode & 4) { if (traceBuf == nullptr) hipMalloc((void**) &traceBuf, sizeof(long long) * 3 * 65536); if (tiles * PL_PARTS <= 65536) a.trace = traceBuf; }
#endif
    hipLaunchKernelGGL(pl_build, dim3(tiles * PL_PARTS), dim3(PL_BLOCK), 0, st, a);
    if (a.state != nullptr) hipLaunchKernelGGL(pl_finish, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
#ifndef OMMHIP_EMU
    if (a.trace != nullptr) {
        // wall_clock64 ticks at 100 MHz: start / end of every workgroup relative to the first start, and where it ran
        const int n = tiles * PL_PARTS;
        long long* h = (long long*) malloc(sizeof(long long) * 3 * n);
        hipStreamSynchronize(st);
        hipMemcpy(h, traceBuf, sizeof(long long) * 3 * n, hipMemcpyDeviceToHost);
        long long t0 = h[0], t1 = h[1], life = 0, lifeMax = 0;
        for (int b = 0; b < n; b++) { if (h[3 * b] < t0) t0 = h[3 * b]; if (h[3 * b + 1] > t1) t1 = h[3 * b + 1]; life += h[3 * b + 1] - h[3 * b]; if (h[3 * b + 1] - h[3 * b] > lifeMax) lifeMax = h[3 * b + 1] - h[3 * b]; }
        int late = 0; for (int b = 0; b < n; b++) if (h[3 * b] - t0 > (t1 - t0) / 10) late++;
        fprintf(stderr, "pl_build trace: %d workgroups, span %.1f us, mean life %.1f us, longest %.1f us, %d started after 10 %% of the span;", n, (t1 - t0) * 0.01, life * 0.01 / n, lifeMax * 0.01, late);
        for (int b = 0; b < n; b += n / 12) fprintf(stderr, " [wg %d: %.0f-%.0f hw %llx]", b, (h[3 * b] - t0) * 0.01, (h[3 * b + 1] - t0) * 0.01, (unsigned long long) h[3 * b + 2]);
        fprintf(stderr, "\n");
        free(h);
    }
#endif
    between();
    if (deferred != nullptr) {
        // deferred check (round 5): the overflow word and the build counter travel to PINNED host words behind the builder's kernels and the
        // call returns at once -- the caller looks at them (pl_deferred_result) after its next wait on this stream, before anything is added
        // to the forces.  Lists that overflowed are truncated, never overrun: what walks them meanwhile fills work arrays with numbers
        // that the repeated call overwrites.
        // (deferredCopies = false: the caller has the two words carried to the host by a copy of its own -- the multipole solver reads them with its sums)
        if (!deferredCopies) return 0;
        hipError_t e = hipMemcpyAsync(&deferred[0], a.overflow, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && a.state != nullptr) e = hipMemcpyAsync(&deferred[1], a.state + 2, sizeof(int), hipMemcpyDeviceToHost, st);
        return (int) e;
    }
    int over = 0;
    hipError_t e = hipMemcpyAsync(&over, a.overflow, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && buildsHost != nullptr && a.state != nullptr) e = hipMemcpyAsync(buildsHost, a.state + 2, sizeof(int), hipMemcpyDeviceToHost, st);      // diagnostics: builds so far
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return (int) e;
    if (over != 0) { if (needed != nullptr) *needed = over == 0x7fffffff ? over : over * PL_PARTS; return -2; }       // in entries per atom, as the caller sizes the list
    return 0;
}
static inline int pl_launch(PairListArgs a, int* needed, hipStream_t st, int* buildsHost = nullptr) { return pl_launch(a, needed, st, buildsHost, [] {}); }
// What a deferred pl_launch found, once the stream has been waited for: 0, or -2 with *needed as pl_launch leaves it.
static inline int pl_deferred_result(const int* deferred, int* needed, int* buildsHost, bool haveState) {
    if (buildsHost != nullptr && haveState) *buildsHost = deferred[1];
    const int over = deferred[0];
    if (over != 0) { if (needed != nullptr) *needed = over == 0x7fffffff ? over : over * PL_PARTS; return -2; }
    return 0;
}

}  // namespace omm
#endif

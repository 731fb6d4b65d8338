#include "HipContext.h"
#include "openmm/System.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

using namespace OpenMM;
using namespace std;

namespace {
struct D4 { double x, y, z, w; };
struct I4 { int x, y, z, w; };

/* Hilbert-curve index of an integer cell (x,y,z) with `bits` bits per axis.
 * Skilling, "Programming the Hilbert curve", AIP Conf. Proc. 707 (2004): axes -> transposed index. */
unsigned long long hilbertIndex(unsigned x, unsigned y, unsigned z, int bits) {
    unsigned X[3] = {x, y, z};
    const unsigned M = 1u << (bits - 1);
    for (unsigned Q = M; Q > 1; Q >>= 1) {
        const unsigned P = Q - 1;
        for (int i = 0; i < 3; i++) {
            if (X[i] & Q) X[0] ^= P;
            else { unsigned t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    for (int i = 1; i < 3; i++) X[i] ^= X[i - 1];
    unsigned t = 0;
    for (unsigned Q = M; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    for (int i = 0; i < 3; i++) X[i] ^= t;
    unsigned long long key = 0;
    for (int b = bits - 1; b >= 0; b--)
        for (int i = 0; i < 3; i++)
            key = (key << 1) | ((X[i] >> b) & 1u);
    return key;
}

/* Host-side helpers of the re-sort (every few hundred steps, but O(N) serial work shows at a million atoms). */
int hostThreads(int ranksOnNode) {
    static const int forced = getenv("OPENMM_HIP_HOST_THREADS") != NULL ? atoi(getenv("OPENMM_HIP_HOST_THREADS")) : 0;
    if (forced > 0) return forced;
    const int hw = (int) std::thread::hardware_concurrency();
    return std::max(1, std::min(16, hw / std::max(1, ranksOnNode)));
}
template <class F>
void parallelFor(int n, int threads, const F& body) {          // body(begin, end)
    if (threads <= 1 || n < 50000) { body(0, n); return; }
    std::vector<std::thread> pool;
    const int chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        const int b = t * chunk, e = std::min(n, b + chunk);
        if (b >= e) break;
        pool.push_back(std::thread([&body, b, e]() { body(b, e); }));
    }
    for (size_t t = 0; t < pool.size(); t++) pool[t].join();
}
/* A space-filling curve through an arbitrary W x H x D grid of cells: the generalised Hilbert curve of J. Cerveny ("gilbert",
 * https://github.com/jakubcerveny/gilbert, BSD-2) -- the grid is split recursively into (nearly) halves whose sub-curves join end to
 * start, with even-sized pieces preferred so that almost every step goes to a face neighbour (a few diagonal steps when a size is odd;
 * never a jump).  The classical Hilbert curve needs a cube of 2^b cells per axis; the sections of a slab decomposition are thin plates
 * (a few cells along x, the whole box along y and z), and a cube's curve clipped to such a plate leaves and re-enters it all the time:
 * atoms that follow each other in the order then lie a box apart and their 32-atom block gets a box-sized bounding box -- the list
 * builder's slowest workgroups (one block took as long as the whole launch, profiles/r07f_nl_trace_*.txt).
 * rankOfCell[(x * H + y) * D + z] = position of the cell along the curve. */
struct Gilbert {
    std::vector<int>& rank; int H, D, next;
    static int sgn(int v) { return v < 0 ? -1 : (v > 0 ? 1 : 0); }
    static int half(int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }          // floor(v / 2)
    void put(int x, int y, int z) { rank[((size_t) x * H + y) * D + z] = next++; }
    void walk(int x, int y, int z, int ax, int ay, int az, int bx, int by, int bz, int cx, int cy, int cz) {
        const int w = std::abs(ax + ay + az), h = std::abs(bx + by + bz), d = std::abs(cx + cy + cz);
        const int dax = sgn(ax), day = sgn(ay), daz = sgn(az), dbx = sgn(bx), dby = sgn(by), dbz = sgn(bz), dcx = sgn(cx), dcy = sgn(cy), dcz = sgn(cz);
        if (h == 1 && d == 1) { for (int i = 0; i < w; i++) { put(x, y, z); x += dax; y += day; z += daz; } return; }
        if (w == 1 && d == 1) { for (int i = 0; i < h; i++) { put(x, y, z); x += dbx; y += dby; z += dbz; } return; }
        if (w == 1 && h == 1) { for (int i = 0; i < d; i++) { put(x, y, z); x += dcx; y += dcy; z += dcz; } return; }
        int ax2 = half(ax), ay2 = half(ay), az2 = half(az), bx2 = half(bx), by2 = half(by), bz2 = half(bz), cx2 = half(cx), cy2 = half(cy), cz2 = half(cz);
        const int w2 = std::abs(ax2 + ay2 + az2), h2 = std::abs(bx2 + by2 + bz2), d2 = std::abs(cx2 + cy2 + cz2);
        if ((w2 % 2) && w > 2) { ax2 += dax; ay2 += day; az2 += daz; }
        if ((h2 % 2) && h > 2) { bx2 += dbx; by2 += dby; bz2 += dbz; }
        if ((d2 % 2) && d > 2) { cx2 += dcx; cy2 += dcy; cz2 += dcz; }
        if (2 * w > 3 * h && 2 * w > 3 * d) {             // long: split along the major axis only
            walk(x, y, z, ax2, ay2, az2, bx, by, bz, cx, cy, cz);
            walk(x + ax2, y + ay2, z + az2, ax - ax2, ay - ay2, az - az2, bx, by, bz, cx, cy, cz);
        }
        else if (3 * h > 4 * d) {                         // flat: not split along the third axis
            walk(x, y, z, bx2, by2, bz2, cx, cy, cz, ax2, ay2, az2);
            walk(x + bx2, y + by2, z + bz2, ax, ay, az, bx - bx2, by - by2, bz - bz2, cx, cy, cz);
            walk(x + (ax - dax) + (bx2 - dbx), y + (ay - day) + (by2 - dby), z + (az - daz) + (bz2 - dbz), -bx2, -by2, -bz2, cx, cy, cz, -(ax - ax2), -(ay - ay2), -(az - az2));
        }
        else if (3 * d > 4 * h) {                         // ... or along the second
            walk(x, y, z, cx2, cy2, cz2, ax2, ay2, az2, bx, by, bz);
            walk(x + cx2, y + cy2, z + cz2, ax, ay, az, bx, by, bz, cx - cx2, cy - cy2, cz - cz2);
            walk(x + (ax - dax) + (cx2 - dcx), y + (ay - day) + (cy2 - dcy), z + (az - daz) + (cz2 - dcz), -cx2, -cy2, -cz2, -(ax - ax2), -(ay - ay2), -(az - az2), bx, by, bz);
        }
        else {                                            // the regular case: five pieces
            walk(x, y, z, bx2, by2, bz2, cx2, cy2, cz2, ax2, ay2, az2);
            walk(x + bx2, y + by2, z + bz2, cx, cy, cz, ax2, ay2, az2, bx - bx2, by - by2, bz - bz2);
            walk(x + (bx2 - dbx) + (cx - dcx), y + (by2 - dby) + (cy - dcy), z + (bz2 - dbz) + (cz - dcz), ax, ay, az, -bx2, -by2, -bz2, -(cx - cx2), -(cy - cy2), -(cz - cz2));
            walk(x + (ax - dax) + bx2 + (cx - dcx), y + (ay - day) + by2 + (cy - dcy), z + (az - daz) + bz2 + (cz - dcz), -cx, -cy, -cz, -(ax - ax2), -(ay - ay2), -(az - az2), bx - bx2, by - by2, bz - bz2);
            walk(x + (ax - dax) + (bx2 - dbx), y + (ay - day) + (by2 - dby), z + (az - daz) + (bz2 - dbz), -bx2, -by2, -bz2, cx2, cy2, cz2, -(ax - ax2), -(ay - ay2), -(az - az2));
        }
    }
};
void gilbertOrder(int W, int H, int D, std::vector<int>& rankOfCell) {
    rankOfCell.assign((size_t) W * H * D, -1);
    Gilbert g = {rankOfCell, H, D, 0};
    if (W >= H && W >= D) g.walk(0, 0, 0, W, 0, 0, 0, H, 0, 0, 0, D);
    else if (H >= W && H >= D) g.walk(0, 0, 0, 0, H, 0, W, 0, 0, 0, 0, D);
    else g.walk(0, 0, 0, 0, 0, D, W, 0, 0, 0, H, 0);
}

/* `count` independent tasks (the slabs of a decomposed order) on up to `threads` threads; an exception of a task is rethrown by the caller's thread. */
template <class F>
void parallelTasks(int count, int threads, const F& task) {          // task(index)
    if (threads <= 1 || count <= 1) { for (int i = 0; i < count; i++) task(i); return; }
    std::vector<std::thread> pool;
    std::vector<std::string> errors(count);
    std::vector<char> failed(count, 0);
    const int workers = std::min(threads, count);
    for (int t = 0; t < workers; t++)
        pool.push_back(std::thread([&, t]() {
            for (int i = t; i < count; i += workers) {
                try { task(i); } catch (const std::exception& e) { failed[i] = 1; errors[i] = e.what(); }
            }
        }));
    for (size_t t = 0; t < pool.size(); t++) pool[t].join();
    for (int i = 0; i < count; i++)
        if (failed[i]) throw OpenMMException(errors[i]);
}
/* Stable LSD radix sort of (key, value) pairs on the low `bits` bits of the key, 11 bits per pass. */
void radixSortPairs(std::vector<std::pair<unsigned long long, int> >& v, int bits) {
    std::vector<std::pair<unsigned long long, int> > tmp(v.size());
    const int digit = 11, buckets = 1 << digit;
    std::vector<size_t> count(buckets);
    for (int shift = 0; shift < bits; shift += digit) {
        std::fill(count.begin(), count.end(), 0);
        for (size_t i = 0; i < v.size(); i++) count[(v[i].first >> shift) & (buckets - 1)]++;
        size_t run = 0;
        for (int b = 0; b < buckets; b++) { const size_t c = count[b]; count[b] = run; run += c; }
        for (size_t i = 0; i < v.size(); i++) tmp[count[(v[i].first >> shift) & (buckets - 1)]++] = v[i];
        v.swap(tmp);
    }
}
}  // namespace

HipContext::HipContext(const System& system, int deviceIndex, bool hostMode, const HipDomain& domain) : domain(domain), numAtoms(system.getNumParticles()), hostMode(hostMode),
        stream(NULL), usePeriodic(false), sortCutoff(0.0), positionsValid(false), hasFallbackForces(false), stepsSinceReorder(0),
        reorderInterval(getenv("OPENMM_HIP_REORDER_INTERVAL") != NULL ? atoi(getenv("OPENMM_HIP_REORDER_INTERVAL")) : 500), cmRemovalPending(false), momentumValid(false), reorderRequested(true), deviceIndex(deviceIndex), pinnedResult(NULL) {
    int count = 0;
    HIP_CHECK(ommhip_device_count(&count));
    if (deviceIndex < 0 || deviceIndex >= count)
        throw OpenMMException("HIP platform: illegal DeviceIndex");
    HIP_CHECK(ommhip_set_device(deviceIndex));
    HIP_CHECK(ommhip_stream_create(&stream));
    // reciprocal space consists of small, latency-bound launches: give them priority over the pair kernel's waves
    HIP_CHECK(ommhip_stream_create_priority(&pmeStream, 1));
    HIP_CHECK(ommhip_event_create_untimed(&pmeForkEvent));
    HIP_CHECK(ommhip_event_create_untimed(&pmeDoneEvent));
    usePmeStream = true;
    paddedAtoms = ((numAtoms + OMMHIP_TILE - 1) / OMMHIP_TILE) * OMMHIP_TILE;
    if (paddedAtoms == 0) paddedAtoms = OMMHIP_TILE;
    findUnits(system);
    slotsPerRank = paddedAtoms; ownSlot0 = 0; ownSlot1 = paddedAtoms; trailerSlot = -1;
    if (decomposed()) {
        if (hostMode)
            throw OpenMMException("HIP platform: a Context spread over several GPUs needs a native integrator (Verlet, Langevin, LangevinMiddle) and no host-side state changes (barostat, virtual sites)");
        // every rank gets the same number of slots: its share of the atoms, room to keep the last unit whole, the trailer slot
        const int R = domain.ranks;
        const int share = (numAtoms + R - 1) / R + maxUnitSize + 2 + 4 * OMMHIP_TILE;     // + the padding between the four sections of a range (halo mode)
        slotsPerRank = ((share + OMMHIP_TILE - 1) / OMMHIP_TILE) * OMMHIP_TILE;
        paddedAtoms = R * slotsPerRank;
        ownSlot0 = domain.rank * slotsPerRank; ownSlot1 = ownSlot0 + slotsPerRank; trailerSlot = slotsPerRank - 2;
    }
    masses.resize(numAtoms);
    for (int i = 0; i < numAtoms; i++) masses[i] = system.getParticleMass(i);
    const size_t n4 = sizeof(double) * 4 * (size_t) max(numAtoms, 1);
    pos.allocate(n4); vel.allocate(n4); xp.allocate(n4); oldx.allocate(n4); tempVel.allocate(n4);
    wrap.allocate(sizeof(int) * 4 * (size_t) max(numAtoms, 1));
    force.allocate(sizeof(long long) * 3 * (size_t) paddedAtoms);
    atomOfSlot.allocate(sizeof(int) * (size_t) paddedAtoms);
    slotOfAtom.allocate(sizeof(int) * (size_t) max(numAtoms, 1));
    energyBuffer.allocate(sizeof(double) * EnergySlots);
    energyResult.allocate(sizeof(double) * (8 + OMMHIP_KE_SCRATCH));     // [0] potential, [1] kinetic, [8..] partial sums of the kinetic energy
    forceDouble.allocate(sizeof(double) * 3 * (size_t) max(numAtoms, 1));
    if (decomposed()) {
        // reciprocal space (with its all-to-alls) runs on the side stream beside the pair kernel: it gets a communicator of its own
        if (ommhip_comm_duplicate(domain.comm, &pmeComm) != 0) pmeComm = NULL;
        posWire.allocate(sizeof(unsigned) * 4 * (size_t) paddedAtoms);
        HIP_CHECK(ommhip_memset(posWire.ptr, 0, posWire.bytes, stream));
        posSlot.allocate(sizeof(double) * 4 * (size_t) paddedAtoms);
        velSlot.allocate(sizeof(double) * 4 * (size_t) paddedAtoms);
        HIP_CHECK(ommhip_memset(posSlot.ptr, 0, posSlot.bytes, stream));
        HIP_CHECK(ommhip_memset(velSlot.ptr, 0, velSlot.bytes, stream));
        wireRef.allocate(posWire.bytes);
        HIP_CHECK(ommhip_memset(wireRef.ptr, 0, wireRef.bytes, stream));
        ddFlags.allocate(sizeof(int) * 4);
        HIP_CHECK(ommhip_memset(ddFlags.ptr, 0, ddFlags.bytes, stream));
        HIP_CHECK(ommhip_host_malloc((void**) &pinnedDdFlags, sizeof(int) * 4));
        memset(pinnedDdFlags, 0, sizeof(int) * 4);
        HIP_CHECK(ommhip_event_create_untimed(&ddFlagsEvent));
    }
    // x drift of a unit's first atom between two re-sorts that the halo is cut for (nm): the upper limit; computeOrderDecomposed takes
    // what the slab widths allow.  At 0.75 of the margin the ranks re-sort early, together (pollDriftFlags).
    // steps the host keeps enqueueing between the snapshot of a due re-sort and its application: they must outlast the host's work
    // (35-55 ms at a million atoms = 15-25 steps there, ~100 steps of a rank of an 8-GPU run; 2 ms = 20 steps at DHFR size)
    reorderLag = getenv("OPENMM_HIP_REORDER_LAG") != NULL ? atoi(getenv("OPENMM_HIP_REORDER_LAG")) : (decomposed() ? 128 : 48);
    // Off the step a re-sort costs the GPU nothing, and a fresher order means fewer rows: one GPU re-sorts every 250 steps (985 527 atoms
    // 2.155 against 2.175 ms per step at 500 and 2.165 at 125; 92 224 atoms 0.2809 against 0.2833; DHFR unchanged: profiles/r06c_ab_reorder_interval.txt)
    if (getenv("OPENMM_HIP_REORDER_INTERVAL") == NULL && !decomposed()) reorderInterval = 250;
    // The margin trades halo against re-sorts: the fastest of a million water molecules moves 0.2 nm along x -- the warning level of a 0.4 nm
    // margin -- within ~60 steps, and every re-sort costs the host 10-20 ms plus the gather of the exact state; a rank of the 8-rank 1M-atom
    // box would convert 2.0 x its own slots per step instead of 2.4 x and pay for it with a re-sort every 60-75 steps (measured: +0.5 ms per
    // step in the serialised 8-rank run, profiles/r07c_*).  0.75 nm asks for one every ~300 steps.
    haloDriftMax = getenv("OPENMM_HIP_DD_DRIFT") != NULL ? atof(getenv("OPENMM_HIP_DD_DRIFT")) : 0.75;
    haloDriftMin = min(haloDriftMax, 0.25);          // slabs that leave less than this: positions stay replicated
    haloDrift = haloDriftMax;
    if (decomposed()) {
        vector<unsigned char> first(max(numAtoms, 1), 0);
        for (size_t u = 0; u + 1 < unitStart.size(); u++) first[unitAtomList[unitStart[u]]] = 1;
        guardAtom.allocate(first.size());
        HIP_CHECK(ommhip_memcpy_h2d(guardAtom.ptr, first.data(), first.size(), stream));
        HIP_CHECK(ommhip_stream_sync(stream));
    }
    memset(&haloPlan, 0, sizeof(haloPlan));
    HIP_CHECK(ommhip_host_malloc((void**) &pinnedResult, sizeof(double) * 8));
    HIP_CHECK(ommhip_memset(energyBuffer.ptr, 0, energyBuffer.bytes, stream));
    HIP_CHECK(ommhip_memset(force.ptr, 0, force.bytes, stream));
    HIP_CHECK(ommhip_memset(pos.ptr, 0, pos.bytes, stream));
    HIP_CHECK(ommhip_memset(xp.ptr, 0, xp.bytes, stream));
    HIP_CHECK(ommhip_memset(oldx.ptr, 0, oldx.bytes, stream));
    HIP_CHECK(ommhip_memset(wrap.ptr, 0, wrap.bytes, stream));
    // velocities start at zero with w = 1/m
    vector<Vec3> zero(numAtoms, Vec3());
    uploadVelocities(zero);
    // identity order
    hostAtomOfSlot.assign(paddedAtoms, -1);
    hostSlotOfAtom.resize(numAtoms);
    for (int i = 0; i < numAtoms; i++) { hostAtomOfSlot[i] = i; hostSlotOfAtom[i] = i; }
    HIP_CHECK(ommhip_memcpy_h2d(atomOfSlot.ptr, hostAtomOfSlot.data(), sizeof(int) * paddedAtoms, stream));
    if (numAtoms > 0)
        HIP_CHECK(ommhip_memcpy_h2d(slotOfAtom.ptr, hostSlotOfAtom.data(), sizeof(int) * numAtoms, stream));
    for (int i = 0; i < 6; i++) box[i] = 0;
    Vec3 a, b, c;
    system.getDefaultPeriodicBoxVectors(a, b, c);
    setBox(a, b, c);
    sync();
}

HipContext::~HipContext() {
    if (pmeComm != NULL) ommhip_comm_destroy(pmeComm);
    if (domain.comm != NULL) ommhip_comm_destroy(domain.comm);
    if (pinnedResult != NULL) ommhip_host_free(pinnedResult);
    if (pinnedDdFlags != NULL) ommhip_host_free(pinnedDdFlags);
    if (pinnedSnapshot != NULL) ommhip_host_free(pinnedSnapshot);
    if (snapshotEvent != NULL) ommhip_event_destroy(snapshotEvent);
    if (ddFlagsEvent != NULL) ommhip_event_destroy(ddFlagsEvent);
    if (pmeForkEvent != NULL) ommhip_event_destroy(pmeForkEvent);
    if (pmeDoneEvent != NULL) ommhip_event_destroy(pmeDoneEvent);
    if (pmeStream != NULL) ommhip_stream_destroy(pmeStream);
    if (stream != NULL) ommhip_stream_destroy(stream);
}

void HipContext::setAsCurrent() {
    HIP_CHECK(ommhip_set_device(deviceIndex));
}

void HipContext::sync() {
    joinPme();
    HIP_CHECK(ommhip_stream_sync(stream));
}

void HipContext::removeListener(HipContextListener* l) {
    listeners.erase(std::remove(listeners.begin(), listeners.end(), l), listeners.end());
}

void HipContext::uploadPositions(const vector<Vec3>& positions) {
    recoverIfFrozen();          // a frozen device state is brought up to date first: skipped steps must not be replayed on top of the new state
    vector<D4> tmp(numAtoms);
    for (int i = 0; i < numAtoms; i++) {
        if (positions[i][0] != positions[i][0] || positions[i][1] != positions[i][1] || positions[i][2] != positions[i][2])
            throw OpenMMException("Particle coordinate is NaN.  For more information, see https://github.com/openmm/openmm/wiki/Frequently-Asked-Questions#nan");
        tmp[i].x = positions[i][0]; tmp[i].y = positions[i][1]; tmp[i].z = positions[i][2]; tmp[i].w = 0;
    }
    if (numAtoms > 0) HIP_CHECK(ommhip_memcpy_h2d(pos.ptr, tmp.data(), sizeof(D4) * numAtoms, stream));
    if (decomposed()) fillWireFromPos();         // every rank was handed all positions: no communication
    sync();
    positionsValid = true;
    positionsVersion++;
    noteStateMutation();
}

void HipContext::recoverIfFrozen() {
    if (inRecovery || !listRecovery || !replaySteps) return;
    inRecovery = true;
    try {
        // steps the device skipped: those the synchronous check finds now plus those a force evaluation already found and left
        // for the integrator (pendingReplay) -- a getState between the two must not see the frozen state under the current time
        const int skipped = listRecovery() + pendingReplay;
        pendingReplay = 0;
        if (skipped > 0) replaySteps(skipped);
    } catch (...) { inRecovery = false; throw; }
    inRecovery = false;
}

void HipContext::downloadPositions(vector<Vec3>& positions) {
    recoverIfFrozen();
    if (decomposed()) gatherState();
    vector<D4> tmp(numAtoms);
    if (numAtoms > 0) HIP_CHECK(ommhip_memcpy_d2h(tmp.data(), pos.ptr, sizeof(D4) * numAtoms, stream));
    sync();
    positions.resize(numAtoms);
    for (int i = 0; i < numAtoms; i++) positions[i] = Vec3(tmp[i].x, tmp[i].y, tmp[i].z);
}

void HipContext::uploadVelocities(const vector<Vec3>& velocities) {
    recoverIfFrozen();
    vector<D4> tmp(numAtoms);
    for (int i = 0; i < numAtoms; i++) {
        tmp[i].x = velocities[i][0]; tmp[i].y = velocities[i][1]; tmp[i].z = velocities[i][2];
        tmp[i].w = masses[i] == 0.0 ? 0.0 : 1.0 / masses[i];
    }
    if (numAtoms > 0) HIP_CHECK(ommhip_memcpy_h2d(vel.ptr, tmp.data(), sizeof(D4) * numAtoms, stream));
    momentumValid = false;
    velocitiesConstrained = false;
    noteStateMutation();
    sync();
}

void HipContext::downloadVelocities(vector<Vec3>& velocities) {
    recoverIfFrozen();
    if (decomposed()) gatherState();
    vector<D4> tmp(numAtoms);
    if (numAtoms > 0) HIP_CHECK(ommhip_memcpy_d2h(tmp.data(), vel.ptr, sizeof(D4) * numAtoms, stream));
    sync();
    velocities.resize(numAtoms);
    for (int i = 0; i < numAtoms; i++) velocities[i] = Vec3(tmp[i].x, tmp[i].y, tmp[i].z);
}

void HipContext::downloadForces(vector<Vec3>& forces) {
    recoverIfFrozen();
    forces.resize(numAtoms);
    if (numAtoms == 0) return;
    if (decomposed()) {
        // every rank holds the forces on its own slots: all-gather each component of the slot-ordered buffer in place
        for (int c = 0; c < 3; c++)
            HIP_CHECK(ommhip_comm_all_gather(domain.comm, force.as<long long>() + (size_t) c * paddedAtoms, sizeof(long long) * (size_t) slotsPerRank, stream));
    }
    HIP_CHECK(ommhip_forces_to_double(force.as<long long>(), slotOfAtom.as<int>(), numAtoms, paddedAtoms, forceDouble.as<double>(), stream));
    vector<double> tmp(3 * (size_t) numAtoms);
    HIP_CHECK(ommhip_memcpy_d2h(tmp.data(), forceDouble.ptr, sizeof(double) * tmp.size(), stream));
    sync();
    for (int i = 0; i < numAtoms; i++) forces[i] = Vec3(tmp[3 * i], tmp[3 * i + 1], tmp[3 * i + 2]);
}

void HipContext::addHostForces(const vector<Vec3>& forces) {
    if (numAtoms == 0) return;
    vector<double> tmp(3 * (size_t) numAtoms);
    for (int i = 0; i < numAtoms; i++) { tmp[3 * i] = forces[i][0]; tmp[3 * i + 1] = forces[i][1]; tmp[3 * i + 2] = forces[i][2]; }
    HIP_CHECK(ommhip_memcpy_h2d(forceDouble.ptr, tmp.data(), sizeof(double) * tmp.size(), stream));
    HIP_CHECK(ommhip_add_forces_from_double(forceDouble.as<double>(), slotOfAtom.as<int>(), numAtoms, paddedAtoms, force.as<long long>(), stream));
    sync();   // tmp goes out of scope
}

void HipContext::setBox(const Vec3& a, const Vec3& b, const Vec3& c) {
    double nb[6] = {a[0], b[0], b[1], c[0], c[1], c[2]};
    bool changed = false;
    for (int i = 0; i < 6; i++)
        if (nb[i] != box[i]) changed = true;
    boxVectors[0] = a; boxVectors[1] = b; boxVectors[2] = c;
    if (!changed) return;
    for (int i = 0; i < 6; i++) box[i] = nb[i];
    boxVersion++;
    reorderRequested = true;       // wrap indices depend on the box
    for (size_t i = 0; i < listeners.size(); i++) listeners[i]->boxChanged();
}

void HipContext::ensureCleared() {
    if (!clearPending) return;
    clearPending = false;
    // the force accumulator and (if a PME kernel registered one) the charge grid are zeroed by one launch
    HIP_CHECK(ommhip_clear2(force.ptr, force.bytes, extraClearPtr, extraClearBytes, stream));
}

void HipContext::forkPme() {
    // (a fork point recorded earlier -- preparePmeFork -- stands: the side stream then waits for what the main stream held THEN)
    if (!pmeForkRecorded) HIP_CHECK(ommhip_event_record(pmeForkEvent, stream));
    pmeForkRecorded = false;
    HIP_CHECK(ommhip_stream_wait_event(pmeStream, pmeForkEvent));
}

void HipContext::preparePmeFork() {
    HIP_CHECK(ommhip_event_record(pmeForkEvent, stream));
    pmeForkRecorded = true;
}

void HipContext::markPmeDone() {
    HIP_CHECK(ommhip_event_record(pmeDoneEvent, pmeStream));
    pmeJoinPending = true;
}

void HipContext::joinPme() {
    if (!pmeJoinPending) return;
    HIP_CHECK(ommhip_stream_wait_event(stream, pmeDoneEvent));
    pmeJoinPending = false;
}

void HipContext::addTerms(const ommhip_term_batch& batch, bool includeEnergy, int id) {
    if (batch.terms.num_terms <= 0) return;
    if (!pendingTerms.empty() && (pendingTermsEnergy != includeEnergy || pendingTerms.size() == OMMHIP_MAX_TERM_LISTS))
        flushTerms();
    pendingTermsEnergy = includeEnergy;
    pendingTerms.push_back(batch);
    pendingTermIds.push_back(id);
}

void HipContext::flushTerms() {
    if (pendingTerms.empty()) return;
    ensureCleared();
    for (size_t i = 0; i < pendingTerms.size(); i++) stampOwnership(pendingTerms[i]);
    HIP_CHECK(ommhip_term_forces_multi((int) pendingTerms.size(), pendingTerms.data(), pos.ptr, slotOfAtom.as<int>(), paddedAtoms, box,
                                       force.as<long long>(), energyBuffer.as<double>(), EnergySlots, pendingTermsEnergy ? 1 : 0, stream));
    pendingTerms.clear();
    pendingTermIds.clear();
}

void HipContext::addValence(const ommhip_valence_list& list, bool includeEnergy) {
    if (list.num_terms <= 0) return;
    if (!pendingValence.empty() && (pendingValenceEnergy != includeEnergy || pendingValence.size() == OMMHIP_MAX_VALENCE_LISTS))
        flushValence();
    pendingValenceEnergy = includeEnergy;
    pendingValence.push_back(list);
}

static long long valenceListsLaunched = 0;
/* test hook: lists of kernels/valence.hip launched by this process so far */
extern "C" __attribute__((visibility("default"))) long long ommhip_plugin_valence_lists_launched() { return valenceListsLaunched; }

void HipContext::flushValence() {
    if (pendingValence.empty()) return;
    ensureCleared();
    valenceListsLaunched += (long long) pendingValence.size();
    HIP_CHECK(ommhip_valence_forces((int) pendingValence.size(), pendingValence.data(), pos.ptr, slotOfAtom.as<int>(), paddedAtoms,
                                    force.as<long long>(), energyBuffer.as<double>(), EnergySlots, pendingValenceEnergy ? 1 : 0, stream));
    pendingValence.clear();
}

void HipContext::stampOwnership(ommhip_term_batch& batch) const {
    // halo mode: a rank evaluates the terms that touch its atoms and counts the energy of those whose first atom it owns (bonded.hip)
    batch.own_slot0 = batch.own_slot1 = 0;
    batch.half_shell = 0; batch.eval_slot0 = batch.eval_slot1 = batch.up_slot0 = batch.up_slot1 = 0; batch.rank = domain.rank; batch.ranks = domain.ranks; batch.slots_per_rank = slotsPerRank;
    batch.error_flags = NULL;
    if (haloMode) { batch.own_slot0 = ownSlot0; batch.own_slot1 = ownSlot1; }
    if (haloMode && halfShell) {
        // one rank evaluates a term that crosses a slab boundary -- the upper one, which sees the lower one's section -- and returns the forces
        batch.half_shell = 1; batch.eval_slot0 = evalRange[0]; batch.eval_slot1 = evalRange[1]; batch.up_slot0 = returnRange[0]; batch.up_slot1 = returnRange[1];
        batch.error_flags = ddFlags.as<int>();
    }
}

bool HipContext::countsTermEnergy() const { return !decomposed() || haloMode || domain.rank == 0; }

int HipContext::registerTerms(int group, const ommhip_term_batch& batch) {
    TermRegistration r = {nextTermId++, group, batch};
    termRegistry.push_back(r);
    foreignPositionsNeeded = true;        // every rank evaluates every registered term (bonds, angles, torsions): all positions must be current
    return r.id;
}

void HipContext::updateTerms(int id, const ommhip_term_batch& batch) {
    for (size_t i = 0; i < termRegistry.size(); i++)
        if (termRegistry[i].id == id) termRegistry[i].batch = batch;
}

void HipContext::unregisterTerms(int id) {
    for (size_t i = 0; i < termRegistry.size(); i++)
        if (termRegistry[i].id == id) { termRegistry.erase(termRegistry.begin() + i); return; }
}

bool HipContext::termsLaunched(int id) const {
    return std::find(launchedTermIds.begin(), launchedTermIds.end(), id) != launchedTermIds.end();
}

void HipContext::collectFrontTerms(vector<ommhip_term_batch>& out, bool includeEnergy) {
    if (!pendingTerms.empty() && pendingTermsEnergy != includeEnergy)
        flushTerms();
    // queued lists first (their owners have executed already), then registered lists whose owners will execute later
    while (!pendingTerms.empty() && out.size() < OMMHIP_MAX_TERM_LISTS) {
        stampOwnership(pendingTerms.back());
        out.push_back(pendingTerms.back());
        if (pendingTermIds.back() >= 0) launchedTermIds.push_back(pendingTermIds.back());
        pendingTerms.pop_back();
        pendingTermIds.pop_back();
    }
    for (size_t i = 0; i < termRegistry.size() && out.size() < OMMHIP_MAX_TERM_LISTS; i++) {
        const TermRegistration& r = termRegistry[i];
        if (r.batch.terms.num_terms <= 0 || ((currentGroups >> r.group) & 1) == 0 || termsLaunched(r.id)) continue;
        if (std::find(pendingTermIds.begin(), pendingTermIds.end(), r.id) != pendingTermIds.end()) continue;
        out.push_back(r.batch);
        stampOwnership(out.back());
        launchedTermIds.push_back(r.id);
    }
}

int HipContext::registerEarlyWork(int group, const EarlyLaunch& launch) {
    EarlyWork w = {nextTermId++, group, launch};
    earlyWork.push_back(w);
    return w.id;
}

void HipContext::unregisterEarlyWork(int id) {
    for (size_t i = 0; i < earlyWork.size(); i++)
        if (earlyWork[i].id == id) { earlyWork.erase(earlyWork.begin() + i); return; }
}

bool HipContext::earlyWorkLaunched(int id) const {
    return std::find(launchedEarlyIds.begin(), launchedEarlyIds.end(), id) != launchedEarlyIds.end();
}

void HipContext::launchEarlyWork(ContextImpl& context, bool includeForces, bool includeEnergy) {
    for (size_t i = 0; i < earlyWork.size(); i++) {
        const EarlyWork w = earlyWork[i];
        if (((currentGroups >> w.group) & 1) == 0 || earlyWorkLaunched(w.id)) continue;
        launchedEarlyIds.push_back(w.id);
        w.launch(context, includeForces, includeEnergy);
    }
}

void HipContext::saveForces() {
    if (savedForce.ptr == NULL) savedForce.allocate(force.bytes);
    HIP_CHECK(ommhip_memcpy_d2d(savedForce.ptr, force.ptr, force.bytes, stream));
}

void HipContext::restoreForces() {
    if (savedForce.ptr != NULL)
        HIP_CHECK(ommhip_memcpy_d2d(force.ptr, savedForce.ptr, force.bytes, stream));
}

double HipContext::reduceEnergy() {
    HIP_CHECK(ommhip_reduce_energy(energyBuffer.as<double>(), EnergySlots, energyResult.as<double>(), stream));
    // the kinetic energy the caller is about to ask for (Context::getState) rides on the same copy and the same wait
    static const bool noPrefetch = getenv("OPENMM_HIP_NO_KE_PREFETCH") != NULL;          // A/B and test knob
    const bool withKinetic = !decomposed() && !hostMode && !noPrefetch && prefetchKineticEnergy && prefetchKineticEnergy();
    HIP_CHECK(ommhip_memcpy_d2h(pinnedResult, energyResult.ptr, sizeof(double) * (withKinetic ? 2 : 1), stream));
    sync();
    kineticEnergyPrefetched = withKinetic;
    if (withKinetic) { prefetchedKineticEnergy = pinnedResult[1]; kineticEnergyMutations = stateMutations; }
    return decomposed() ? sumOverRanks(pinnedResult[0]) : pinnedResult[0];
}

double HipContext::sumOverRanks(double v) {
    if (!decomposed()) return v;
    vector<double> all(domain.ranks);
    HIP_CHECK(ommhip_comm_all_gather_host(domain.comm, &v, all.data(), sizeof(double), stream));
    double sum = 0;
    for (int r = 0; r < domain.ranks; r++) sum += all[r];
    return sum;
}

void HipContext::exchangePositions() {
    if (haloMode) HIP_CHECK(ommhip_comm_halo_exchange(domain.comm, posWire.ptr, &haloPlan, stream));
    else HIP_CHECK(ommhip_comm_all_gather(domain.comm, posWire.ptr, sizeof(unsigned) * 4 * (size_t) slotsPerRank, stream));
}

void HipContext::returnHaloForces() {
    if (!halfShell) return;
    HIP_CHECK(ommhip_comm_halo_return(domain.comm, force.as<long long>(), paddedAtoms, &returnPlan, returnStaging.as<long long>(), stream));
}

unsigned HipContext::ddWarnFraction() const {
    // the re-sort is requested at this part of the margin; what is left of it must outlast the (at most 12) steps until every rank has
    // seen the flag plus the reorderLag steps until the new order applies: 0.3 nm for 0.28 ps at the defaults, where a water molecule
    // diffuses 0.05 nm (RMS, one axis)
    static const double warn = getenv("OPENMM_HIP_DD_WARN") != NULL ? atof(getenv("OPENMM_HIP_DD_WARN")) : 0.5;
    return (unsigned) (warn * haloDrift / box[0] * 4294967296.0);
}
unsigned HipContext::ddMaxFraction() const { return (unsigned) (haloDrift / box[0] * 4294967296.0); }

void HipContext::pollDriftFlags() {
    if (!haloMode) return;
    const long long n = ddEvaluations++;
    if ((n & 7) == 0) {
        HIP_CHECK(ommhip_memcpy_d2h(pinnedDdFlags, ddFlags.ptr, sizeof(int) * 4, stream));
        HIP_CHECK(ommhip_event_record(ddFlagsEvent, stream));
    }
    else if ((n & 7) == 4) {
        HIP_CHECK(ommhip_event_sync(ddFlagsEvent));          // recorded four evaluations ago: long complete
        if ((pinnedDdFlags[2] & 8) != 0)
            throw OpenMMException("HIP platform: a bonded term (or an exclusion / 1-4 pair) reaches across more than the halo of the domain decomposition: no rank holds all of its atoms");
        if ((pinnedDdFlags[2] & 4) != 0) {
            // the hard limit travels in the trailers like the warning levels: every rank finds it at the same evaluation and throws here,
            // none is left waiting in a collective ([0]: it was one of this rank's atoms; [3]: the largest drift this rank saw)
            char detail[256];
            snprintf(detail, sizeof(detail), " (rank %d%s: largest drift of its own atoms %.3f nm, margin %.3f nm, order %d steps old, re-sort %lld, interval %d, lag %d)", domain.rank,
                     pinnedDdFlags[0] != 0 ? ", where it happened" : "", 2.0 * pinnedDdFlags[3] / 4294967296.0 * box[0], haloDrift, stepsSinceReorder, reorderCount, reorderInterval, reorderLag);
            throw OpenMMException(string("HIP platform: an atom drifted further along x between two re-sorts than the halo of the domain decomposition allows; "
                                         "lower OPENMM_HIP_REORDER_INTERVAL or raise OPENMM_HIP_DD_DRIFT") + detail);
        }
        // every rank reads the same word at the same evaluation: they re-sort together -- off the step while the margin lasts, at once
        // (the next step waits for it) when an atom has used 80 % of it: a hot system, e.g. a lattice start that melts
        if ((pinnedDdFlags[2] & 2) != 0) reorderRequested = true;
        else if ((pinnedDdFlags[2] & 3) != 0) reorderDue = true;
    }
}

void HipContext::fillWireFromPos() {
    HIP_CHECK(ommhip_encode_wire(pos.ptr, atomOfSlot.as<int>(), 0, paddedAtoms, box, posWire.ptr, stream));
}

void HipContext::gatherState() {
    // ("alone" transport, diagnostics: the peers do not exist -- their atoms keep the state this rank was given at the start)
    if (domain.comm != NULL && strcmp(ommhip_comm_transport(domain.comm), "alone") == 0) return;
    // positions: the per-step wire records are 32-bit fractions; what leaves the platform (and what a re-sort redistributes) are
    // the owners' exact doubles -- staged in slot order, all-gathered, scattered back to atom order
    HIP_CHECK(ommhip_pack_slots(pos.ptr, atomOfSlot.as<int>(), ownSlot0, ownSlot1, posSlot.ptr, stream));
    HIP_CHECK(ommhip_comm_all_gather(domain.comm, posSlot.ptr, sizeof(double) * 4 * (size_t) slotsPerRank, stream));
    HIP_CHECK(ommhip_unpack_slots(posSlot.ptr, atomOfSlot.as<int>(), 0, paddedAtoms, pos.ptr, stream));
    // velocities: only the owner's are current
    HIP_CHECK(ommhip_pack_slots(vel.ptr, atomOfSlot.as<int>(), ownSlot0, ownSlot1, velSlot.ptr, stream));
    HIP_CHECK(ommhip_comm_all_gather(domain.comm, velSlot.ptr, sizeof(double) * 4 * (size_t) slotsPerRank, stream));
    HIP_CHECK(ommhip_unpack_slots(velSlot.ptr, atomOfSlot.as<int>(), 0, paddedAtoms, vel.ptr, stream));
}

void HipContext::findUnits(const System& system) {
    // constraint-connected groups of atoms (waters, X-H clusters, ...): they are integrated by one thread and must not be
    // split between ranks.  Union-find over the constraints; units are numbered by their lowest atom.
    vector<int> parent(numAtoms);
    for (int i = 0; i < numAtoms; i++) parent[i] = i;
    struct Find { static int root(vector<int>& p, int i) { while (p[i] != i) { p[i] = p[p[i]]; i = p[i]; } return i; } };
    for (int c = 0; c < system.getNumConstraints(); c++) {
        int a, b; double d;
        system.getConstraintParameters(c, a, b, d);
        int ra = Find::root(parent, a), rb = Find::root(parent, b);
        if (ra != rb) parent[max(ra, rb)] = min(ra, rb);
    }
    unitOfAtom.assign(numAtoms, -1);
    vector<int> count;
    for (int i = 0; i < numAtoms; i++) {
        const int r = Find::root(parent, i);
        if (unitOfAtom[r] < 0) { unitOfAtom[r] = (int) count.size(); count.push_back(0); }
        unitOfAtom[i] = unitOfAtom[r];
        count[unitOfAtom[i]]++;
    }
    unitStart.assign(count.size() + 1, 0);
    maxUnitSize = 1;
    for (size_t u = 0; u < count.size(); u++) { unitStart[u + 1] = unitStart[u] + count[u]; maxUnitSize = max(maxUnitSize, count[u]); }
    unitAtomList.resize(numAtoms);
    vector<int> cursor(unitStart.begin(), unitStart.end() - 1);
    for (int i = 0; i < numAtoms; i++) unitAtomList[cursor[unitOfAtom[i]]++] = i;
}

void HipContext::stepTaken() {
    noteStateMutation();
    velocitiesConstrained = true;
    stepsSinceReorder++;
    stepsSinceSnapshot++;
}

void HipContext::partitionBlocks(vector<int>& atomOfSlotLike) const {
    // inside every 32-slot block: atoms flagged by setBlockTailAtoms() after the others (stable), empty slots (-1) last
    if (blockTailAtom.empty()) return;
    static const bool off = getenv("OPENMM_HIP_NO_LJ_PARTITION") != NULL;        // A/B knob
    if (off) return;
    const int numBlocks = (int) ((atomOfSlotLike.size() + OMMHIP_TILE - 1) / OMMHIP_TILE);
    // (the blocks are independent; parallelFor splits ranges of at least 50 000 items, so the range is given in slots and cut at block boundaries)
    parallelFor(numBlocks * OMMHIP_TILE, hostThreads(domain.ranks), [&](int begin, int end) {
        int tmp[OMMHIP_TILE];
        const size_t first = (size_t) (begin + OMMHIP_TILE - 1) / OMMHIP_TILE * OMMHIP_TILE, last = min(atomOfSlotLike.size(), (size_t) (end + OMMHIP_TILE - 1) / OMMHIP_TILE * OMMHIP_TILE);
        for (size_t b = first; b < last; b += OMMHIP_TILE) {
            const int n = (int) min((size_t) OMMHIP_TILE, atomOfSlotLike.size() - b);
            int k = 0;
            for (int pass = 0; pass < 3; pass++)
                for (int i = 0; i < n; i++) {
                    const int atom = atomOfSlotLike[b + i];
                    const int cls = atom < 0 ? 2 : (blockTailAtom[atom] ? 1 : 0);
                    if (cls == pass) tmp[k++] = atom;
                }
            for (int i = 0; i < n; i++) atomOfSlotLike[b + i] = tmp[i];
        }
    });
    static const bool diag = getenv("OPENMM_HIP_DIAG_BLOCKS") != NULL;          // diagnostics: blocks by their number of head (non-tail) atoms
    if (diag) {
        int hist[OMMHIP_TILE + 1] = {0};
        for (size_t b = 0; b < atomOfSlotLike.size(); b += OMMHIP_TILE) {
            int head = 0;
            for (size_t i = b; i < min(b + OMMHIP_TILE, atomOfSlotLike.size()); i++)
                if (atomOfSlotLike[i] >= 0 && !blockTailAtom[atomOfSlotLike[i]]) head++;
            hist[head]++;
        }
        fprintf(stderr, "HIP platform: blocks by number of head atoms:");
        for (int h = 0; h <= OMMHIP_TILE; h++) if (hist[h] > 0) fprintf(stderr, " %d:%d", h, hist[h]);
        fprintf(stderr, "\n");
    }
}

void HipContext::computeOrder(const vector<Vec3>& positions, vector<int>& order, vector<int>& wrapOut) {
    wrapOut.assign(4 * (size_t) numAtoms, 0);
    order.resize(numAtoms);
    for (int i = 0; i < numAtoms; i++) order[i] = i;
    vector<Vec3> wrapped(positions);
    if (usePeriodic) {
        // periodic image such that the reduced position lies in the primary cell (triclinic aware)
        for (int i = 0; i < numAtoms; i++) {
            Vec3 p = positions[i];
            int iz = (int) floor(p[2] / box[5]);
            p[0] -= iz * box[3]; p[1] -= iz * box[4]; p[2] -= iz * box[5];
            int iy = (int) floor(p[1] / box[2]);
            p[0] -= iy * box[1]; p[1] -= iy * box[2];
            int ix = (int) floor(p[0] / box[0]);
            p[0] -= ix * box[0];
            wrapOut[4 * i] = ix; wrapOut[4 * i + 1] = iy; wrapOut[4 * i + 2] = iz;
            wrapped[i] = p;
        }
    }
    if (sortCutoff <= 0.0 || numAtoms <= OMMHIP_TILE)
        return;
    // bin into cells of ~0.3 nm and walk the cells along a Hilbert curve
    static const double binWidth = getenv("OPENMM_HIP_SORT_BIN") != NULL ? atof(getenv("OPENMM_HIP_SORT_BIN")) : 0.3;   // tuning knob (nm)
    Vec3 lo(1e300, 1e300, 1e300), hi(-1e300, -1e300, -1e300);
    for (int i = 0; i < numAtoms; i++)
        for (int k = 0; k < 3; k++) { lo[k] = min(lo[k], wrapped[i][k]); hi[k] = max(hi[k], wrapped[i][k]); }
    int maxCells = 1;
    int ncell[3];
    for (int k = 0; k < 3; k++) {
        ncell[k] = max(1, min(1023, (int) floor((hi[k] - lo[k]) / binWidth) + 1));
        maxCells = max(maxCells, ncell[k]);
    }
    int bits = 1;
    while ((1 << bits) < maxCells) bits++;
    // The curve visits all (2^bits)^3 cells of its cube; where the occupied cells are only part of that cube (21 of 32 per axis for DHFR at
    // 0.3 nm) it leaves and re-enters the occupied region, and two atoms that follow each other in the order can lie far apart: blocks and
    // tiles with box-sized bounding boxes.  So the cells are fitted to the extent: 2^bits of them along every axis (narrower than binWidth,
    // anisotropic in a non-cubic box), and consecutive cells of the curve are always neighbours.  OPENMM_HIP_SORT_FIT=0: cells of binWidth.
    static const bool fitCells = getenv("OPENMM_HIP_SORT_FIT") == NULL || atoi(getenv("OPENMM_HIP_SORT_FIT")) != 0;
    double width[3] = {binWidth, binWidth, binWidth};
    if (fitCells)
        for (int k = 0; k < 3; k++) { ncell[k] = 1 << bits; width[k] = max((hi[k] - lo[k]) * (1.0 + 1e-9), 1e-9) / ncell[k]; }
    vector<pair<unsigned long long, int> > keyed(numAtoms);
    const int threads = hostThreads(domain.ranks);
    parallelFor(numAtoms, threads, [&](int begin, int end) {
        for (int i = begin; i < end; i++) {
            unsigned c[3];
            for (int k = 0; k < 3; k++) {
                int v = (int) floor((wrapped[i][k] - lo[k]) / width[k]);
                c[k] = (unsigned) max(0, min(ncell[k] - 1, v));
            }
            keyed[i] = make_pair(hilbertIndex(c[0], c[1], c[2], bits), i);
        }
    });
    radixSortPairs(keyed, 3 * bits);          // stable: equal keys stay in atom order, as a sort of (key, atom) pairs would leave them
    for (int i = 0; i < numAtoms; i++) order[i] = keyed[i].second;
    partitionBlocks(order);
}

const vector<int>& HipContext::curveThroughGrid(int W, int H, int D) {
    const long long key = ((long long) W << 40) | ((long long) H << 20) | (long long) D;
    {
        std::lock_guard<std::mutex> lock(curveCacheMutex);
        map<long long, vector<int> >::const_iterator found = curveCache.find(key);
        if (found != curveCache.end()) return found->second;
    }
    vector<int> rank;
    gilbertOrder(W, H, D, rank);          // outside the lock: other slabs go on with grids that are there already
    std::lock_guard<std::mutex> lock(curveCacheMutex);
    vector<int>& slot = curveCache[key];  // (entries are never removed while a re-sort runs: references stay valid; the cache is trimmed between re-sorts)
    if (slot.empty()) slot.swap(rank);
    return slot;
}

void HipContext::computeOrderDecomposed(const vector<Vec3>& positions, vector<int>& newAtomOfSlot, vector<int>& wrapOut) {
    // Slabs along x with equal numbers of atoms, whole units only; inside a slab the units follow a Hilbert curve, so
    // 32 consecutive slots are a compact group of atoms (the neighbour list's i-blocks) just as on one GPU.
    // A triclinic box (reduced vectors a = (ax, 0, 0), b = (bx, by, 0), c = (cx, cy, cz)) is cut by planes parallel to b and c: the
    // slab coordinate is xi = ax * (coefficient of a), which is x in a rectangular box, is periodic with ax whatever image an atom is in,
    // and is what the first index of the PME grid counts -- so the planes of the grid a rank owns are a slab of xi as before.  Two
    // points a distance d apart differ by at most kappa * d in xi (kappa = |grad xi| >= 1): every Cartesian length that enters the halo
    // widths below -- list reach, unit extent -- is stretched by kappa; the drift margin is a length in xi to begin with (the guard
    // of nl_prepare compares the first box fraction), and the stencil reaches are multiples of the grid spacing along xi.
    if (!usePeriodic)
        throw OpenMMException("HIP platform: multi-GPU runs need a periodic box");
    const bool triclinic = box[1] != 0.0 || box[3] != 0.0 || box[4] != 0.0;
    const double skewY = box[1] / box[2], skewZ = (box[4] * box[1] - box[3] * box[2]) / (box[2] * box[5]);      // d xi / dy, d xi / dz
    const double kappa = sqrt(1.0 + skewY * skewY + skewZ * skewZ);
    const int R = domain.ranks, numUnits = (int) unitStart.size() - 1;
    const double L[3] = {box[0], box[2], box[5]};
    if (curveCache.size() > 96) curveCache.clear();
    static const bool phaseTiming = getenv("OPENMM_HIP_TIMING") != NULL && getenv("OPENMM_HIP_TIMING")[0] == '2';          // diagnostics: phases of this function on stderr
    std::chrono::steady_clock::time_point tPhase = std::chrono::steady_clock::now();
    auto phase = [&](const char* name) {
        if (!phaseTiming) return;
        const std::chrono::steady_clock::time_point now = std::chrono::steady_clock::now();
        fprintf(stderr, "  order phase %-28s %.2f ms\n", name, 1e-3 * std::chrono::duration_cast<std::chrono::microseconds>(now - tPhase).count());
        tPhase = now;
    };
    wrapOut.assign(4 * (size_t) numAtoms, 0);
    vector<Vec3> ref(numUnits);                      // wrapped position of the unit's first atom
    const int threads = hostThreads(domain.ranks);
    vector<double> extentOfThread(max(threads, 1) + 1, 0.0);      // largest distance of an atom from the first atom of its unit, per chunk of the loop below
    parallelFor(numUnits, threads, [&](int begin, int end) {
      double chunkExtent = 0.0;
      for (int u = begin; u < end; u++) {
        const int a0 = unitAtomList[unitStart[u]];
        Vec3 p0 = positions[a0];
        if (!triclinic)
            for (int k = 0; k < 3; k++) {
                const int w = (int) floor(p0[k] / L[k]);
                wrapOut[4 * a0 + k] = w;
                p0[k] -= w * L[k];
            }
        else {
            // the image whose coefficients of c, b, a lie in [0, 1) (computeOrder does the same)
            const int iz = (int) floor(p0[2] / box[5]);
            p0[0] -= iz * box[3]; p0[1] -= iz * box[4]; p0[2] -= iz * box[5];
            const int iy = (int) floor(p0[1] / box[2]);
            p0[0] -= iy * box[1]; p0[1] -= iy * box[2];
            const int ix = (int) floor(p0[0] / box[0]);
            p0[0] -= ix * box[0];
            wrapOut[4 * a0] = ix; wrapOut[4 * a0 + 1] = iy; wrapOut[4 * a0 + 2] = iz;
        }
        ref[u] = p0;
        // the other atoms go to the image nearest to the first one, so the unit stays in one piece
        for (int i = unitStart[u] + 1; i < unitStart[u + 1]; i++) {
            const int a = unitAtomList[i];
            double d2 = 0;
            if (!triclinic)
                for (int k = 0; k < 3; k++) {
                    double d = positions[a][k] - positions[a0][k];
                    d -= floor(d / L[k] + 0.5) * L[k];
                    wrapOut[4 * a + k] = (int) floor((positions[a][k] - (p0[k] + d)) / L[k] + 0.5);
                    d2 += d * d;
                }
            else {
                Vec3 d = positions[a] - positions[a0];
                const int nz = (int) floor(d[2] / box[5] + 0.5);
                d[0] -= nz * box[3]; d[1] -= nz * box[4]; d[2] -= nz * box[5];
                const int ny = (int) floor(d[1] / box[2] + 0.5);
                d[0] -= ny * box[1]; d[1] -= ny * box[2];
                const int nx = (int) floor(d[0] / box[0] + 0.5);
                d[0] -= nx * box[0];
                wrapOut[4 * a] = wrapOut[4 * a0] + nx; wrapOut[4 * a + 1] = wrapOut[4 * a0 + 1] + ny; wrapOut[4 * a + 2] = wrapOut[4 * a0 + 2] + nz;
                d2 = d.dot(d);
            }
            chunkExtent = max(chunkExtent, sqrt(d2));
        }
        // from here on the first coordinate of ref is the slab coordinate xi (a rectangular box: x itself)
        if (triclinic) {
            const double sz = p0[2] / box[5], sy = (p0[1] - sz * box[4]) / box[2];
            double xi = p0[0] - sy * box[1] - sz * box[3];          // (the image above has x, y, z in [0, edge), not the three coefficients)
            xi -= floor(xi / L[0]) * L[0];
            ref[u][0] = max(0.0, min(L[0], xi));
        }
      }
      // (chunks are numbered by where they begin: parallelFor hands out equal pieces)
      const int chunk = (numUnits + max(threads, 1) - 1) / max(threads, 1);
      extentOfThread[min((int) extentOfThread.size() - 1, chunk > 0 ? begin / chunk : 0)] = chunkExtent;
    });
    phase("wrap offsets, unit extents");
    // ---- cut along x into R groups of (nearly) equal atom count
    vector<int> byX(numUnits);
    {
        vector<pair<unsigned long long, int> > xs(numUnits);
        const double scale = 4294967295.0 / L[0];
        parallelFor(numUnits, threads, [&](int begin, int end) {
            for (int u = begin; u < end; u++) xs[u] = make_pair((unsigned long long) (max(0.0, min(L[0], ref[u][0])) * scale), u);
        });
        radixSortPairs(xs, 33);                  // 32-bit fixed-point x; ties stay in unit order
        for (int u = 0; u < numUnits; u++) byX[u] = xs[u].second;
    }
    vector<int> groupStart(R + 1, numUnits);
    groupStart[0] = 0;
    {
        long long cumulative = 0;
        int g = 0;
        for (int i = 0; i < numUnits; i++) {
            // unit i opens the next group once the groups so far hold their share
            while (g + 1 < R && cumulative >= (long long) (g + 1) * numAtoms / R) groupStart[++g] = i;
            cumulative += unitStart[byX[i] + 1] - unitStart[byX[i]];
        }
        while (g + 1 < R) groupStart[++g] = numUnits;
    }
    // ---- halo mode?  A rank needs the positions of every atom that can come within the list cutoff of one of its own before the
    //      next re-sort, and of every atom whose PME stencil can reach its planes: everything within Tdn below / Tup above its slab along x
    //      (computed further down).  With more than two ranks that must lie inside the two neighbouring slabs; the decision is a pure
    //      function of the gathered positions, so all ranks take it alike.  Otherwise positions stay replicated through the all-gather.
    phase("cut along x");
    vector<double> bound(R + 1, L[0]);           // slab of group g = [bound[g], bound[g + 1])
    bound[0] = 0.0;
    for (int g = 1; g < R; g++) bound[g] = groupStart[g] < numUnits ? ref[byX[groupStart[g]]][0] : L[0];
    double extent = 0.0;                          // largest distance of an atom from the first atom of its unit (a rigid unit may rotate: any of it can turn into x)
    for (size_t t = 0; t < extentOfThread.size(); t++) extent = max(extent, extentOfThread[t]);
    extent *= 1.1 * kappa;                        // constraints hold distances to the first atom; flexible units get a little room
    const double pairReach = haloReach * kappa;   // (both in xi)
    // The drift margin (how far along x the first atom of a unit may move between two re-sorts): haloDriftMax unless the narrowest slab
    // leaves less.  Round 3 let it grow to whatever the slabs allowed (0.7 nm on 8 ranks of the 1M-atom box: three slabs' worth of slots
    // converted per step); now it is an upper limit chosen for the re-sort cadence (the re-sorts are off the step), and the halo is what
    // the pairs and the PME stencils need.
    double minWidth = L[0];
    for (int g = 0; g < R; g++) minWidth = min(minWidth, bound[g + 1] - bound[g]);
    haloDrift = haloDriftMax;
    if (R > 2) haloDrift = max(0.0, min(haloDriftMax, 0.5 * (0.98 * minWidth - pairReach - extent)));
    // What a rank must see below / above its slab: the partners of its pairs (list cutoff + the drift of both atoms + a unit's reach) and the
    // atoms whose PME stencils touch its planes g L / R ... (g + 1) L / R (one atom's drift; the stencil looks forward, so mostly below).
    const double reachBelow = pmeReachBelow > 0.0 ? pmeReachBelow : pmeReachX, reachAbove = pmeReachAbove > 0.0 ? pmeReachAbove : pmeReachX;
    const double Tpair = pairReach + 2.0 * haloDrift + extent;
    double pmeBelow = 0.0, pmeAbove = 0.0;
    if (reachBelow > 0.0 || reachAbove > 0.0)
        for (int g = 0; g < R; g++) {
            pmeBelow = max(pmeBelow, bound[g] - (g * L[0] / R - reachBelow - haloDrift - extent));
            pmeAbove = max(pmeAbove, ((g + 1) * L[0] / R + reachAbove + haloDrift + extent) - bound[g + 1]);
        }
    static const bool noHalo = getenv("OPENMM_HIP_DD_REPLICATE") != NULL;          // A/B knob: always replicate positions (round-2 behaviour)
    static const bool bothSides = getenv("OPENMM_HIP_DD_BOTH_SIDES") != NULL;      // A/B knob: cross-boundary pairs on both sides, symmetric halo (round-3 behaviour)
    bool halo = R > 1 && haloReach > 0.0 && !noHalo && R <= OMMHIP_MAX_RANKS && haloDrift >= haloDriftMin;
    // Half-shell: a pair across a boundary is evaluated by the rank above it, which therefore needs Tup of its lower neighbour (pairs and
    // stencils) and only Tdn of its upper one (stencils).  Needs slabs that hold what their upper neighbour asks for; with two ranks (one
    // peer on both sides) also boundaries far enough apart that no pair could be claimed across both.
    double Tup = max(Tpair, pmeBelow), Tdn = max(0.0, pmeAbove);
    bool half = halo && !bothSides && (R > 2 ? minWidth >= Tup : minWidth >= Tup + max(Tdn, Tpair) + 0.05);
    if (!half) { Tup = Tdn = max(Tpair, max(pmeBelow, pmeAbove)); }
    if (halo && R > 2 && minWidth < Tup) halo = false;           // the neighbours do not hold everything a rank needs: positions stay replicated
    if (!halo) half = false;
    haloMode = halo;
    halfShell = half;
    static const bool ddDebug = getenv("OPENMM_HIP_DD_DEBUG") != NULL;           // diagnostics: the decision and what it was made from
    if (ddDebug && domain.rank == 0)
        fprintf(stderr, "HIP platform: decomposition over %d ranks: narrowest slab %.3f nm, list reach %.3f, unit extent %.3f, drift margin %.3f (limit %.3f, floor %.3f), pairs need %.3f, "
                        "stencils need %.3f below / %.3f above -> halo %d, half-shell %d, sections %.3f up / %.3f down\n", R, minWidth, haloReach, extent, haloDrift, haloDriftMax, haloDriftMin,
                Tpair, pmeBelow, pmeAbove, halo ? 1 : 0, half ? 1 : 0, Tup, Tdn);
    if (ddDebug && domain.rank == 0 && triclinic) fprintf(stderr, "HIP platform: triclinic box: slabs of the first box fraction, lengths stretched by %.4f\n", kappa);
    // ---- the order inside each slab: a space-filling curve through each of its (up to four) sections separately -- the generalised Hilbert
    //      curve through a grid of ~binWidth cells FITTED to the section's extent (a plate: thin along x), see gilbertOrder
    static const double binWidth = getenv("OPENMM_HIP_SORT_BIN") != NULL ? atof(getenv("OPENMM_HIP_SORT_BIN")) : 0.3;
    static const bool cubeCurve = getenv("OPENMM_HIP_DD_CUBE_CURVE") != NULL;          // A/B knob: the box-wide 2^b cube of round 3 clipped to the sections
    int maxCells = 1, ncell[3];
    for (int k = 0; k < 3; k++) { ncell[k] = max(1, min(1023, (int) floor(L[k] / binWidth) + 1)); maxCells = max(maxCells, ncell[k]); }
    int bits = 1;
    while ((1 << bits) < maxCells) bits++;
    const int gridY = max(1, min(1023, (int) ceil(L[1] / binWidth))), gridZ = max(1, min(1023, (int) ceil(L[2] / binWidth)));
    newAtomOfSlot.assign(paddedAtoms, -1);
    ownedUnits.clear();
    // sections of every rank's range, in slots relative to its start: down = [0, sectionEnd[1]), up = [sectionEnd[0], sectionEnd[2])
    vector<int> sectionEnd(4 * (size_t) R, 0);
    // (the slabs are independent of each other: one task per slab, each writing its own slot range)
    parallelTasks(R, threads, [&](int g) {
        const int first = groupStart[g], count = groupStart[g + 1] - groupStart[g];
        vector<pair<unsigned long long, int> > keyed(count);
        // sections: 0 = the rank below needs it, 1 = both neighbours, 2 = the rank above, 3 = nobody (most significant key bits)
        vector<unsigned char> sectionOf(count, 0);
        double lo[4] = {1e300, 1e300, 1e300, 1e300}, hi[4] = {-1e300, -1e300, -1e300, -1e300};
        // A section that comes out as a sliver -- a fraction of a nanometre thick, a few hundred atoms spread over the whole y-z face: its
        // 32-atom blocks cannot be compact whatever the order -- is merged into its neighbour by asking for a little more than needed
        // (sending an atom nobody needs is always safe): "nobody" joins "up only", a thin "both" is widened on either side.
        double TupG = Tup, TdnG = Tdn;
        if (halo) {
            const double width = bound[g + 1] - bound[g], thin = 0.6;
            const double nobody = width - TdnG - TupG;
            if (nobody > 0.0 && nobody < thin) TupG = width - TdnG;
            const double both = TdnG + TupG - width;
            if (both > 0.0 && both < thin && width > thin) { const double grow = 0.5 * (thin - both); TdnG = min(width, TdnG + grow); TupG = min(width, TupG + grow); }
        }
        for (int i = 0; i < count; i++) {
            const int u = byX[first + i];
            int section = 0;
            if (halo) {
                const bool down = ref[u][0] - bound[g] < TdnG, up = bound[g + 1] - ref[u][0] < TupG;
                section = down ? (up ? 1 : 0) : (up ? 2 : 3);
            }
            sectionOf[i] = (unsigned char) section;
            lo[section] = min(lo[section], ref[u][0]); hi[section] = max(hi[section], ref[u][0]);
        }
        int keyBits = 3 * bits;
        if (!cubeCurve) {
            const vector<int>* curve[4] = {NULL, NULL, NULL, NULL};
            int gridX[4] = {1, 1, 1, 1};
            double widthX[4] = {1, 1, 1, 1};
            keyBits = 1;
            for (int sct = 0; sct < 4; sct++) {
                if (hi[sct] < lo[sct]) continue;
                gridX[sct] = max(1, min(1023, (int) ceil((hi[sct] - lo[sct]) / binWidth)));
                widthX[sct] = max((hi[sct] - lo[sct]) * (1.0 + 1e-9), 1e-9) / gridX[sct];
                curve[sct] = &curveThroughGrid(gridX[sct], gridY, gridZ);
                while ((1ull << keyBits) < (unsigned long long) gridX[sct] * gridY * gridZ) keyBits++;
            }
            for (int i = 0; i < count; i++) {
                const int u = byX[first + i], sct = sectionOf[i];
                const int cx = max(0, min(gridX[sct] - 1, (int) floor((ref[u][0] - lo[sct]) / widthX[sct])));
                const int cy = max(0, min(gridY - 1, (int) floor(ref[u][1] / L[1] * gridY))), cz = max(0, min(gridZ - 1, (int) floor(ref[u][2] / L[2] * gridZ)));
                keyed[i] = make_pair(((unsigned long long) sct << keyBits) | (unsigned long long) (*curve[sct])[((size_t) cx * gridY + cy) * gridZ + cz], u);
            }
        }
        else
            for (int i = 0; i < count; i++) {
                const int u = byX[first + i];
                unsigned c[3];
                for (int k = 0; k < 3; k++) c[k] = (unsigned) max(0, min(ncell[k] - 1, (int) floor(ref[u][k] / binWidth)));
                keyed[i] = make_pair(((unsigned long long) sectionOf[i] << keyBits) | hilbertIndex(c[0], c[1], c[2], bits), u);
            }
        radixSortPairs(keyed, keyBits + 2);
        int slot = g * slotsPerRank, current = 0;
        for (size_t i = 0; i < keyed.size(); i++) {
            const int u = keyed[i].second;
            const int section = (int) (keyed[i].first >> keyBits);
            while (current < section) {          // a section ends: the next one starts on a block boundary
                slot = (slot + OMMHIP_TILE - 1) / OMMHIP_TILE * OMMHIP_TILE;
                sectionEnd[4 * g + current++] = slot - g * slotsPerRank;
            }
            if (slot + (unitStart[u + 1] - unitStart[u]) > g * slotsPerRank + trailerSlot)      // the two trailer records stay free
                throw OpenMMException("HIP platform: internal error: a rank's slot range overflowed in the domain decomposition");
            for (int j = unitStart[u]; j < unitStart[u + 1]; j++) newAtomOfSlot[slot++] = unitAtomList[j];
            if (g == domain.rank) ownedUnits.push_back(u);
        }
        while (current < 4) {
            slot = (slot + OMMHIP_TILE - 1) / OMMHIP_TILE * OMMHIP_TILE;
            sectionEnd[4 * g + current++] = min(slot - g * slotsPerRank, slotsPerRank);
        }
    });
    phase("Hilbert order of the slabs");
    numActiveRanges = 0;
    memset(&haloPlan, 0, sizeof(haloPlan));
    if (halo) {
        const size_t rec = sizeof(unsigned) * 4;
        haloPlan.rank_stride = rec * (size_t) slotsPerRank;
        for (int g = 0; g < R; g++) {
            haloPlan.down_offset[g] = 0; haloPlan.down_bytes[g] = rec * (size_t) sectionEnd[4 * g + 1];
            haloPlan.up_offset[g] = rec * (size_t) sectionEnd[4 * g]; haloPlan.up_bytes[g] = rec * (size_t) (sectionEnd[4 * g + 2] - sectionEnd[4 * g]);
        }
        haloPlan.trailer_offset = rec * (size_t) trailerSlot; haloPlan.trailer_bytes = 2 * rec;
        // what this rank holds current wire records for: its own range, the down section of the rank above, the up section of the rank below
        const int me = domain.rank, above = (me + 1) % R, below = (me + R - 1) % R;
        int ranges[3][2] = {{ownSlot0, ownSlot1}, {above * slotsPerRank, above * slotsPerRank + sectionEnd[4 * above + 1]},
                            {below * slotsPerRank + sectionEnd[4 * below], below * slotsPerRank + sectionEnd[4 * below + 2]}};
        if (above == below) {      // two ranks: one neighbour, its two sections may overlap or touch
            if (ranges[1][1] >= ranges[2][0]) { ranges[1][1] = ranges[2][1]; ranges[2][0] = ranges[2][1]; }          // merged into one range
        }
        for (int r = 0; r < 3; r++)
            if (ranges[r][1] > ranges[r][0]) { activeRange[2 * numActiveRanges] = ranges[r][0]; activeRange[2 * numActiveRanges + 1] = ranges[r][1]; numActiveRanges++; }
        // half-shell: the partners from below, and what comes back from above
        evalRange[0] = evalRange[1] = returnRange[0] = returnRange[1] = 0;
        memset(&returnPlan, 0, sizeof(returnPlan));
        if (halfShell) {
            evalRange[0] = below * slotsPerRank + sectionEnd[4 * below]; evalRange[1] = below * slotsPerRank + sectionEnd[4 * below + 2];
            returnRange[0] = me * slotsPerRank + sectionEnd[4 * me]; returnRange[1] = me * slotsPerRank + sectionEnd[4 * me + 2];
            int largest = 0;
            for (int g = 0; g < R; g++) {
                returnPlan.first_slot[g] = g * slotsPerRank + sectionEnd[4 * g]; returnPlan.num_slots[g] = sectionEnd[4 * g + 2] - sectionEnd[4 * g];
                largest = max(largest, returnPlan.num_slots[g]);
            }
            returnStaging.allocate(sizeof(long long) * 3 * (size_t) max(largest, 1));
        }
    }
    else { halfShell = false; evalRange[0] = evalRange[1] = returnRange[0] = returnRange[1] = 0; }
    partitionBlocks(newAtomOfSlot);          // a rank's range is a whole number of blocks: no atom changes owner
    phase("block partition");
}

double HipContext::timeDecomposedOrder(const vector<Vec3>& positions, int repeats, vector<int>* atomOfSlotOut) {
    double best = 1e30;
    for (int k = 0; k < repeats; k++) {
        vector<int> newAtomOfSlot, wrapHost;
        const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        computeOrderDecomposed(positions, newAtomOfSlot, wrapHost);
        best = min(best, 1e-3 * std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count());
        if (atomOfSlotOut != NULL) atomOfSlotOut->swap(newAtomOfSlot);
    }
    return best;
}

void HipContext::takeSnapshot() {
    // everything on the stream, nothing waits: (decomposed: exact positions of all atoms on every rank), a device copy -- the
    // reference the drift guard will measure from -- and its download into pinned memory, then an event
    if (decomposed()) gatherState();
    const size_t bytes = sizeof(double) * 4 * (size_t) max(numAtoms, 1);
    if (posSnapshot.ptr == NULL) {
        posSnapshot.allocate(bytes);
        HIP_CHECK(ommhip_host_malloc((void**) &pinnedSnapshot, bytes));
        HIP_CHECK(ommhip_event_create_untimed(&snapshotEvent));
    }
    HIP_CHECK(ommhip_memcpy_d2d(posSnapshot.ptr, pos.ptr, bytes, stream));
    HIP_CHECK(ommhip_memcpy_d2h(pinnedSnapshot, posSnapshot.ptr, bytes, stream));
    HIP_CHECK(ommhip_event_record(snapshotEvent, stream));
    snapshotPending = true;
    stepsSinceSnapshot = 0;
}

bool HipContext::reorderIfNeeded() {
    if (!reorderRequested) {
        if (snapshotPending) {
            if (stepsSinceSnapshot < reorderLag) return false;
            HIP_CHECK(ommhip_event_sync(snapshotEvent));      // the GPU has reached the snapshot; reorderLag steps are queued behind it
            snapshotPending = false;
            vector<Vec3> positions(numAtoms);
            for (int i = 0; i < numAtoms; i++) positions[i] = Vec3(pinnedSnapshot[4 * (size_t) i], pinnedSnapshot[4 * (size_t) i + 1], pinnedSnapshot[4 * (size_t) i + 2]);
            return applyOrder(positions, true);
        }
        if (!reorderDue && stepsSinceReorder < reorderInterval) return false;
        if (!usePeriodic && sortCutoff <= 0.0 && !decomposed()) { reorderDue = false; stepsSinceReorder = 0; return false; }
        if (reorderLag > 0 && !hostMode && positionsValid) {
            reorderDue = false;
            takeSnapshot();
            return false;
        }
    }
    snapshotPending = false;                // superseded by a re-sort that cannot wait
    reorderRequested = false;
    reorderDue = false;
    if (!usePeriodic && sortCutoff <= 0.0 && !decomposed()) {
        stepsSinceReorder = 0;
        return false;                       // identity order and no wrapping: nothing to do
    }
    vector<Vec3> positions;
    downloadPositions(positions);           // decomposed: also completes pos[] and vel[] on this rank (units change owner below)
    return applyOrder(positions, false);
}

bool HipContext::applyOrder(const vector<Vec3>& positions, bool fromSnapshot) {
    static const bool timing = getenv("OPENMM_HIP_TIMING") != NULL;          // diagnostics: wall time of the re-sort on stderr
    const std::chrono::steady_clock::time_point tStart = std::chrono::steady_clock::now();
    struct Report { bool on; std::chrono::steady_clock::time_point t0; int n;
                    ~Report() { if (on) fprintf(stderr, "HIP platform: re-sort of %d atoms took %.2f ms\n", n, 1e-3 * std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count()); } } report = {timing, tStart, numAtoms};
    stepsSinceReorder = fromSnapshot ? stepsSinceSnapshot : 0;       // the age of the order
    reorderCount++;
    vector<int> wrapHost;
    bool orderChanged = false;
    if (decomposed()) {
        vector<int> newAtomOfSlot;
        computeOrderDecomposed(positions, newAtomOfSlot, wrapHost);
        orderChanged = newAtomOfSlot != hostAtomOfSlot;
        hostAtomOfSlot.swap(newAtomOfSlot);
        for (int s = 0; s < paddedAtoms; s++)
            if (hostAtomOfSlot[s] >= 0) hostSlotOfAtom[hostAtomOfSlot[s]] = s;
    }
    else {
        vector<int> order;
        computeOrder(positions, order, wrapHost);
        for (int s = 0; s < numAtoms; s++)
            if (hostAtomOfSlot[s] != order[s]) { orderChanged = true; break; }
        for (int s = 0; s < numAtoms; s++) { hostAtomOfSlot[s] = order[s]; hostSlotOfAtom[order[s]] = s; }
    }
    if (orderChanged) orderVersion++;
    const std::chrono::steady_clock::time_point tOrdered = std::chrono::steady_clock::now();
    // a lagged re-sort of a decomposed run: the units change owner NOW -- every rank needs the current exact state of all atoms
    if (decomposed() && fromSnapshot) gatherState();
    HIP_CHECK(ommhip_memcpy_h2d(wrap.ptr, wrapHost.data(), sizeof(int) * wrapHost.size(), stream));
    HIP_CHECK(ommhip_memcpy_h2d(atomOfSlot.ptr, hostAtomOfSlot.data(), sizeof(int) * paddedAtoms, stream));
    HIP_CHECK(ommhip_memcpy_h2d(slotOfAtom.ptr, hostSlotOfAtom.data(), sizeof(int) * numAtoms, stream));
    if (decomposed()) {
        // the drift guard measures from the positions the sections were cut for: the snapshot's when the order comes from one
        if (fromSnapshot) HIP_CHECK(ommhip_encode_wire(posSnapshot.ptr, atomOfSlot.as<int>(), 0, paddedAtoms, box, wireRef.ptr, stream));
        fillWireFromPos();
        if (!fromSnapshot) HIP_CHECK(ommhip_memcpy_d2d(wireRef.ptr, posWire.ptr, posWire.bytes, stream));
        HIP_CHECK(ommhip_memset(ddFlags.ptr, 0, ddFlags.bytes, stream));
        HIP_CHECK(ommhip_clear_trailer_flags(posWire.ptr, domain.ranks, slotsPerRank, trailerSlot, stream));
        memset(pinnedDdFlags, 0, sizeof(int) * 4);
        ddEvaluations = 0;
    }
    sync();
    const std::chrono::steady_clock::time_point tUploaded = std::chrono::steady_clock::now();
    // wrap offsets may have changed even when the order did not: listeners rebuild their slot data either way
    for (size_t i = 0; i < listeners.size(); i++) listeners[i]->atomsReordered();
    if (timing) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return 1e-3 * std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        fprintf(stderr, "HIP platform: re-sort phases (%s): order %.2f ms, upload %.2f ms, listeners %.2f ms%s\n", fromSnapshot ? "from a snapshot taken earlier" : "at once", ms(tStart, tOrdered),
                ms(tOrdered, tUploaded), ms(tUploaded, std::chrono::steady_clock::now()), decomposed() ? (haloMode ? " (halo mode)" : " (replicated)") : "");
    }
    return orderChanged;
}

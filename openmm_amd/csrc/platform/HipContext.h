#ifndef OPENMM_HIPCONTEXT_H_
#define OPENMM_HIPCONTEXT_H_
/* Per-Context device state of the OpenMM "HIP" platform (MI355X).  Host-side only: talks to the GPU
 * exclusively through the C ABI of include/openmm_hip_kernels.h.
 *
 * Layout in HBM:
 *   atom order  : pos double4[N], vel double4[N] (w = 1/m), xp, oldx, wrap int4[N]
 *   slot order  : force int64[3*P] (fixed point, SoA), atomOfSlot int[P], slotOfAtom int[N]
 * Each nonbonded kernel owns its own float posq/sigEps/neighbour list in slot order.
 */
#include "openmm_hip_kernels.h"
#include "openmm_hip_comm.h"
#include "openmm/OpenMMException.h"
#include "openmm/Vec3.h"
#include <cstddef>
#include <functional>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

namespace OpenMM {

class System;
class ContextImpl;

#define HIP_CHECK(call) do { int rc__ = (call); if (rc__ != 0) { std::stringstream s__; \
    s__ << "HIP platform: " << #call << " failed with code " << rc__ << " (" << ommhip_error_string(rc__) << ") at " << __FILE__ << ":" << __LINE__; \
    throw OpenMM::OpenMMException(s__.str()); } } while (0)

/** RAII device allocation. */
class DeviceBuffer {
public:
    DeviceBuffer() : ptr(NULL), bytes(0) {}
    ~DeviceBuffer() { release(); }
    void allocate(size_t nbytes) {
        if (ptr != NULL && bytes == nbytes) return;       // same size: keep the allocation (contents unspecified either way)
        release();
        HIP_CHECK(ommhip_malloc(&ptr, nbytes));
        bytes = nbytes;
    }
    void release() {
        if (ptr != NULL) ommhip_free(ptr);   // may run after runtime teardown: ignore errors
        ptr = NULL; bytes = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(ptr); }
    void* ptr;
    size_t bytes;
private:
    DeviceBuffer(const DeviceBuffer&);
    DeviceBuffer& operator=(const DeviceBuffer&);
};

/** Something that must react when atoms are re-sorted or the periodic box changes. */
class HipContextListener {
public:
    virtual ~HipContextListener() {}
    virtual void atomsReordered() = 0;
    virtual void boxChanged() = 0;
    virtual void positionsSet() = 0;
};

/** Place of this process in a domain-decomposed run (one box on several GPUs); ranks == 1 is the ordinary case.
 *  The communicator is owned by the HipContext that receives it. */
struct HipDomain {
    HipDomain() : ranks(1), rank(0), comm(NULL) {}
    int ranks, rank;
    ommhip_comm* comm;
};

class HipContext {
public:
    HipContext(const System& system, int deviceIndex, bool hostMode, const HipDomain& domain = HipDomain());
    ~HipContext();
    void setAsCurrent();
    void sync();

    // ---- state transfer (all blocking)
    void uploadPositions(const std::vector<Vec3>& positions);
    void downloadPositions(std::vector<Vec3>& positions);
    void uploadVelocities(const std::vector<Vec3>& velocities);
    void downloadVelocities(std::vector<Vec3>& velocities);
    void downloadForces(std::vector<Vec3>& forces);            // from the fixed-point buffer
    void addHostForces(const std::vector<Vec3>& forces);       // host forces -> fixed-point buffer
    void setBox(const Vec3& a, const Vec3& b, const Vec3& c);
    void getBox(Vec3& a, Vec3& b, Vec3& c) const { a = boxVectors[0]; b = boxVectors[1]; c = boxVectors[2]; }
    /** Counters of state edits from outside the integration: every uploadPositions (setPositions, a checkpoint; in host mode every
     *  evaluation) and every change of the box.  Kernels that carry information from one evaluation to the next (the first guess of the
     *  AMOEBA dipole solver) compare them with the values they saw last. */
    long long positionsVersion = 0, boxVersion = 0;
    long long orderVersion = 0;              // counts the re-sorts that changed the slot order (slot-keyed data of other plugins' kernels: the AMOEBA pair lists)

    // ---- per-evaluation
    /** Request the start-of-evaluation clear of the force accumulator (+ extraClear buffer).  The clear is lazy: the
     *  nonbonded kernel folds it into its first launch (takePendingClear); everybody else calls ensureCleared()
     *  before touching the force buffer. */
    void clearForces() { clearPending = true; }
    void ensureCleared();
    bool takePendingClear() { bool p = clearPending; clearPending = false; return p; }
    bool clearPending = false;
    /** One extra buffer (the PME charge grid) zeroed together with the forces at the start of every evaluation. */
    void* extraClearPtr = NULL;
    size_t extraClearBytes = 0;
    /** Queue a per-term force list; all queued lists go out in one launch (flushTerms, called by finishComputation). */
    /** Side stream for PME reciprocal space (runs concurrently with the direct-space kernels; cf. the DisablePmeStream
     *  property of the CUDA platform, CudaKernels.cpp:728,851-868).  forkPme(): the side stream waits for everything
     *  enqueued so far on the main stream; joinPme(): the main stream waits for the side stream's work. */
    void* pmeStream;
    bool usePmeStream;
    void forkPme();
    /** Records the fork point NOW; the next forkPme() waits for this point instead of recording its own -- for work that is enqueued on the main
     *  stream before the side stream's work is launched but does not have to precede it (the AMOEBA multipole list build, HipAmoebaKernels.cpp). */
    void preparePmeFork();
    bool pmeForkRecorded = false;
    void markPmeDone();
    void joinPme();
    void addTerms(const ommhip_term_batch& batch, bool includeEnergy, int id = -1);
    /** Decomposed runs: which rank's term energies count.  Replicated positions: every rank evaluates every term, rank 0 counts;
     *  halo mode: the term kernel itself counts a term on the rank that owns its first atom (stampOwnership). */
    bool countsTermEnergy() const;
    void stampOwnership(ommhip_term_batch& batch) const;
    /** Term lists that persist across evaluations (bonds, angles, torsions) are registered once with their force group, so
     *  that whoever opens an evaluation can launch all lists of the evaluated groups together with its own work
     *  (ommhip_force_front) before the owning Force objects have executed.  Their execute() then finds them launched. */
    int registerTerms(int group, const ommhip_term_batch& batch);
    void updateTerms(int id, const ommhip_term_batch& batch);
    void unregisterTerms(int id);
    bool termsLaunched(int id) const;
    /** Moves every term list of this evaluation that can still be launched into `out` (at most OMMHIP_MAX_TERM_LISTS in
     *  total, `out` may already hold some): the queued ones and the registered ones of the evaluated force groups. */
    void collectFrontTerms(std::vector<ommhip_term_batch>& out, bool includeEnergy);
    int currentGroups = -1;
    /** Start of an evaluation: which force groups are evaluated; forgets what the previous evaluation launched. */
    void beginEvaluation(int groups) { currentGroups = groups; launchedTermIds.clear(); launchedEarlyIds.clear(); }
    /** Work of one Force that another Force's kernel may start before the owner executes -- the same idea as registerTerms, for whole
     *  kernels: a force whose evaluation blocks the host for a while (the AMOEBA dipole solver waits for its convergence measure) first
     *  launches what the forces after it in the System would otherwise only enqueue once it has finished (the AMOEBA vdW kernel, on the
     *  side stream).  The owner's execute() asks earlyWorkLaunched() and launches itself when nobody did. */
    typedef std::function<void(ContextImpl&, bool, bool)> EarlyLaunch;        // (context, includeForces, includeEnergy)
    int registerEarlyWork(int group, const EarlyLaunch& launch);
    void unregisterEarlyWork(int id);
    bool earlyWorkLaunched(int id) const;
    void noteEarlyWorkLaunched(int id) { launchedEarlyIds.push_back(id); }
    void launchEarlyWork(ContextImpl& context, bool includeForces, bool includeEnergy);
    void flushTerms();
    /** The same for the lists of kernels/valence.hip (the AMOEBA valence terms): queued by their Forces, one launch per evaluation. */
    void addValence(const ommhip_valence_list& list, bool includeEnergy);
    void flushValence();
    /** Energy queries (Context::getState(Energy) = one evaluation with energies + the kinetic energy) with ONE host round trip: an integrator
     *  whose kinetic energy is that of the velocities as they stand (LangevinMiddle) registers how to enqueue that sum; reduceEnergy() then
     *  enqueues it behind the evaluation and reads both numbers with one copy, and the integrator's computeKineticEnergy() -- which
     *  Context::getState calls next -- takes the number from here unless the state was touched in between (stateMutations).  Consumed once. */
    std::function<bool()> prefetchKineticEnergy;              // enqueues the sum into energyResult[1]; false: not applicable right now
    long long stateMutations = 0;                             // counts steps, uploads, constraint passes, CM removals: anything that changes velocities
    void noteStateMutation() { stateMutations++; }
    bool kineticEnergyPrefetched = false;
    long long kineticEnergyMutations = 0;
    double prefetchedKineticEnergy = 0.0;
    /** The native Verlet / Langevin / LangevinMiddle integrators evaluate the forces anew at the start of every step, so an energy-only
     *  evaluation need not put the previous forces back (two copies of the force buffer per energy query); false: save and restore
     *  (ReferenceKernels.cpp:190-199), as a CustomIntegrator's force caches need it. */
    bool forcesRecomputedEveryStep = false;
    void saveForces();                                         // device copy of the force buffer (energy-only evaluations)
    void restoreForces();
    double reduceEnergy();                                      // blocking; also zeroes the buffer
    /** Spatially re-sort atoms into slots if requested or due.  Returns true if the order changed.
     *  A re-sort that is merely DUE (the interval ran out, an atom neared the drift margin of a decomposed run) is taken off the
     *  step: the positions are snapshotted on the stream (takeSnapshot) and the host goes on enqueueing `reorderLag` more steps; only
     *  then does it wait for the snapshot, compute the order from it and apply it -- while the GPU works through those queued steps
     *  instead of idling for the 35-55 ms the host needs at a million atoms.  The order is `reorderLag` steps old when it takes
     *  effect (it is a locality heuristic; the drift guard of decomposed runs measures from the snapshot).  A re-sort that is
     *  REQUESTED (new positions, a new box, host mode) happens at once.  Decomposed runs: every rank takes the same path at the
     *  same step (the triggers are step counts and flags every rank sees alike). */
    bool reorderIfNeeded();
    void requestReorder() { reorderRequested = true; }
    void stepTaken();                                           // counts steps towards the next reorder
    void requestReorderSoon() { reorderDue = true; }           // off the step (see reorderIfNeeded); decomposed runs: called on every rank at the same evaluation

    /** Diagnostics (tools/time_resort_host.py): wall time in ms of the host-side order computation of a decomposed run for these positions. */
    double timeDecomposedOrder(const std::vector<Vec3>& positions, int repeats, std::vector<int>* atomOfSlotOut = NULL);
    int getDeviceIndex() const { return deviceIndex; }
    void addListener(HipContextListener* l) { listeners.push_back(l); }
    /** Atoms that should sit at the END of their 32-slot block (e.g. atoms without Lennard-Jones parameters: the pair kernel
     *  skips that part of the arithmetic for the tail of a block whose atoms have none).  Takes effect at the next re-sort. */
    void setBlockTailAtoms(const std::vector<char>& tail) { blockTailAtom = tail; reorderRequested = true; }
    void removeListener(HipContextListener* l);

    // ---- domain decomposition (ranks > 1): this rank owns the slots [ownSlot0, ownSlot1) -- a slab of the box along x --
    //      integrates the atoms in them and computes the forces on them; positions of ALL atoms are replicated through posWire
    //      (uint4, slot order: 32-bit fixed-point fractions of the box edges, 16 bytes per atom), which the ranks all-gather in
    //      place after every integration step.  The last two records of each rank's range (trailerSlot, trailerSlot + 1) never
    //      hold an atom: they carry the rank's momentum (three doubles) through the same all-gather.
    HipDomain domain;
    int slotsPerRank, ownSlot0, ownSlot1, trailerSlot;
    bool decomposed() const { return domain.comm != NULL; }      // also with ONE rank when a communicator was given (single-GPU test of the whole path)
    DeviceBuffer posWire, posSlot, velSlot;      // posSlot / velSlot: double4 staging of exact positions / velocities for downloads and re-sorts
    /** Communicator of the reciprocal-space stream (a duplicate of domain.comm; NULL = share domain.comm, single stream). */
    ommhip_comm* pmeComm = NULL;
    /** Enqueue the per-step position exchange on the main stream (after the integration kernel wrote this rank's part): in halo
     *  mode the grouped send/recv of the boundary sections with the two neighbouring slabs (+ the momentum trailers of all ranks),
     *  else the all-gather of the whole buffer. */
    void exchangePositions();
    // ---- halo mode: decided at every re-sort, identically on all ranks.  Inside a rank's slot range the units are
    //      laid out by what the neighbours need: [needed below only | needed on both sides | needed above only | needed by nobody],
    //      each section a whole number of 32-slot blocks and Hilbert-sorted on its own, so "what rank r - 1 needs" and "what rank
    //      r + 1 needs" are two contiguous (overlapping) runs of slots that travel without packing.
    bool haloMode = false;
    ommhip_halo_plan haloPlan;
    // ---- half-shell mode (a refinement of halo mode, decided with it): a pair of atoms of two neighbouring slabs is evaluated ONCE, by the
    //      rank above; that rank keeps the force on the lower rank's atom in its own buffer and hands it back after the force kernels
    //      (returnHaloForces), so a rank needs its lower neighbour's "up" section for its pairs and only a thin "down" section of its
    //      upper neighbour (for charge spreading: atoms whose PME stencil reaches its planes).  evalRange = the slots of the lower
    //      neighbour's up section: partners of this rank's pair list, forces on them kept and returned; returnRange = this rank's own up
    //      section, for which the upper neighbour returns forces.  Both-sides evaluation (round 3) remains the fallback: two ranks whose
    //      two boundaries are too close, or OPENMM_HIP_DD_BOTH_SIDES=1.
    bool halfShell = false;
    int evalRange[2] = {0, 0}, returnRange[2] = {0, 0};
    ommhip_halo_return_plan returnPlan;
    DeviceBuffer returnStaging;                  // long long[3 * slots of the largest up section]: forces on this rank's atoms as its upper neighbour computed them
    void returnHaloForces();                     // enqueued on the main stream by whoever completes the forces of an evaluation (finishComputation)
    double pmeReachBelow = 0.0, pmeReachAbove = 0.0;      // how far (nm) below / above its PME planes a rank must see atoms for charge spreading (forward-only stencil: mostly below)
    int numActiveRanges = 0;              // slot ranges this rank has current wire records for (own + received sections); 0 = all
    int activeRange[8];
    double haloReach = 0.0;               // list cutoff (cutoff + padding) of the nonbonded force, nm; 0 = no halo mode
    double pmeReachX = 0.0;               // how far (nm) beyond its PME planes a rank must see atoms for charge spreading
    double haloDrift;                     // x drift since the re-sort an atom is allowed before the run must have re-sorted (nm); set at every re-sort
    double haloDriftMax, haloDriftMin;
    DeviceBuffer guardAtom;               // unsigned char[N]: 1 for the first atom of every integration unit (the atoms the drift guard watches)
    DeviceBuffer wireRef, ddFlags;        // wire records of the last re-sort; int[4] flags (ommhip_neighbor_list::dd_flags)
    int* pinnedDdFlags = NULL;
    void* ddFlagsEvent = NULL;
    long long ddEvaluations = 0;
    long long reorderCount = 0;            // re-sorts so far (diagnostics)
    /** Called once per force evaluation on decomposed runs: reads the drift flags back every 8 evaluations (asynchronously),
     *  looks at them 4 evaluations later -- the same evaluation on every rank, and every rank finds the same words -- and
     *  requests a re-sort, or ends the run (on every rank alike) when the hard limit was exceeded. */
    void pollDriftFlags();
    unsigned ddWarnFraction() const;      // thresholds in units of 2^-32 box lengths
    unsigned ddMaxFraction() const;
    /** Make pos[] and vel[] (atom order) complete and current on this rank: before a re-sort and before downloads. */
    void gatherState();
    /** Wire records of every slot from pos[]: after an upload of all positions or a re-sort (every rank holds them all; no communication). */
    void fillWireFromPos();
    /** Sum of one double over the ranks, the same bits everywhere (host all-gather + fixed-order sum). */
    double sumOverRanks(double v);
    /** Integration units (constraint-connected groups of atoms): CSR over units in atom-index order of their first atom. */
    std::vector<int> unitStart, unitAtomList, unitOfAtom;
    int maxUnitSize;
    /** Decomposed runs: does anything read the double-precision positions of atoms this rank does not own (term lists of bonded
     *  forces, 1-4s, exclusion pairs across integration units)?  If not, nl_prepare skips the per-step refresh of those positions. */
    bool foreignPositionsNeeded = false;
    /** Units owned by this rank after the last re-sort (indices into unitStart). */
    std::vector<int> ownedUnits;

    // ---- neighbour-list overflow (a device-triggered rebuild needed more rows than allocated).  The device freezes the
    //      integration while the list's overflow word is set (ommhip_integrator_state::freeze_state); the host notices at the
    //      latest 16 evaluations later (or at the next download of positions / velocities / forces), grows the list and
    //      replays the skipped steps.  freezeState: device pointer to the list's state array (NULL: no freezing, e.g. decomposed runs).
    bool deterministicForces = false;                        // platform property DeterministicForces: reproducible charge-grid sums (ommhip_pme::deterministic)
    int* freezeState = NULL;
    int pendingReplay = 0;                                   // skipped steps the integrator still has to redo
    long long currentStepIndex = -1;                         // index of the step whose updateContextState is running (-1: none); differs from stepCount while skipped steps are redone
    long long overflowRecoveries = 0;                        // how often listRecovery found an overflowed list and grew it
    std::function<int()> listRecovery;                       // set by the nonbonded kernel: synchronous check; fixes the list, returns skipped steps
    std::function<bool()> listOverflowSeen;                  // set by the nonbonded kernel: do the state words last copied back (valid after a sync) show an overflow?
    std::function<void(int)> replaySteps;                    // set by the integrator kernel: redo that many steps
    void recoverIfFrozen();
    bool recovering() const { return inRecovery; }          // inside recoverIfFrozen (the redone steps evaluate forces themselves)

    // ---- immutable after construction
    int numAtoms, paddedAtoms;
    bool hostMode;                 // true: host vectors are authoritative (Reference integrator etc.)
    void* stream;
    std::vector<double> masses;

    // ---- device arrays
    DeviceBuffer savedForce;
    DeviceBuffer pos, vel, xp, oldx, tempVel, wrap, force, atomOfSlot, slotOfAtom, energyBuffer, energyResult, forceDouble;
    static const int EnergySlots = 2048;

    // ---- host mirrors
    double box[6];
    Vec3 boxVectors[3];
    bool usePeriodic;              // any force uses periodic boundary conditions
    double sortCutoff;             // > 0 when a cutoff-based nonbonded force wants spatial sorting
    std::vector<int> hostAtomOfSlot, hostSlotOfAtom;
    bool positionsValid;
    bool hasFallbackForces;        // some force kernels are Reference ones (need host positions/forces each evaluation)
    int stepsSinceReorder, reorderInterval;
    // CM-motion removal folded into the fused step (HipConstraints::fusedStep): the remover only raises `pending`;
    // `momentumValid` says the device holds the total momentum of the current velocities.
    bool cmRemovalPending, momentumValid;
    bool velocitiesConstrained = false;       // the velocities are what a native integration step left (they satisfy the constraints); false after an upload

private:
    void computeOrder(const std::vector<Vec3>& positions, std::vector<int>& order, std::vector<int>& wrapOut);
    void computeOrderDecomposed(const std::vector<Vec3>& positions, std::vector<int>& newAtomOfSlot, std::vector<int>& wrapOut);
    /** Position of every cell of a W x H x D grid along the space-filling curve the sections of a slab are ordered by (HipContext.cpp:
     *  gilbertOrder); kept from re-sort to re-sort -- the same few grid sizes come back.  Safe to call from the threads of one re-sort. */
    const std::vector<int>& curveThroughGrid(int W, int H, int D);
    std::map<long long, std::vector<int> > curveCache;
    std::mutex curveCacheMutex;
    void findUnits(const System& system);
    std::vector<HipContextListener*> listeners;
    std::vector<char> blockTailAtom;
    void partitionBlocks(std::vector<int>& atomOfSlotLike) const;
    bool inRecovery = false;
    void* pmeForkEvent = NULL;
    void* pmeDoneEvent = NULL;
    bool pmeJoinPending = false;
    std::vector<ommhip_term_batch> pendingTerms;
    std::vector<int> pendingTermIds;
    bool pendingTermsEnergy = false;
    std::vector<ommhip_valence_list> pendingValence;
    bool pendingValenceEnergy = false;
    struct TermRegistration { int id, group; ommhip_term_batch batch; };
    std::vector<TermRegistration> termRegistry;
    std::vector<int> launchedTermIds;          // ids whose terms already went out this evaluation (fused front launch)
    struct EarlyWork { int id, group; EarlyLaunch launch; };
    std::vector<EarlyWork> earlyWork;
    std::vector<int> launchedEarlyIds;
    int nextTermId = 1;
    bool reorderRequested;
    bool reorderDue = false;
    // lagged re-sort
    bool snapshotPending = false;
    int stepsSinceSnapshot = 0, reorderLag;
    DeviceBuffer posSnapshot;                // double4[N] positions at the snapshot (device copy: reference of the drift guard)
    double* pinnedSnapshot = NULL;           // the same on the host
    void* snapshotEvent = NULL;
    void takeSnapshot();
    bool applyOrder(const std::vector<Vec3>& positions, bool fromSnapshot);
    int deviceIndex;
    double* pinnedResult;
};

}  // namespace OpenMM
#endif

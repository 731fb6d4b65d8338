#include "HipKernels.h"
#include <chrono>
#include "ReferenceCCMAAlgorithm.h"
#include "ReferenceVirtualSites.h"
#include "SimTKOpenMMRealType.h"
#include "SimTKOpenMMUtilities.h"
#include "openmm/CMMotionRemover.h"
#include "openmm/HarmonicAngleForce.h"
#include "openmm/HarmonicBondForce.h"
#include "openmm/Integrator.h"
#include "openmm/LangevinIntegrator.h"
#include "openmm/LangevinMiddleIntegrator.h"
#include "openmm/NonbondedForce.h"
#include "openmm/PeriodicTorsionForce.h"
#include "openmm/VerletIntegrator.h"
#include "openmm/internal/NonbondedForceImpl.h"
#include "openmm/internal/OSRngSeed.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <mutex>
#include <thread>
#include <sstream>

using namespace OpenMM;
using namespace std;

namespace {
template <class T>
void uploadVector(DeviceBuffer& buf, const vector<T>& v, void* stream) {
    buf.allocate(max(sizeof(T) * v.size(), (size_t) 16));
    if (!v.empty()) {
        HIP_CHECK(ommhip_memcpy_h2d(buf.ptr, v.data(), sizeof(T) * v.size(), stream));
        HIP_CHECK(ommhip_stream_sync(stream));
    }
}
}  // namespace

// ================================================================================================
// Constraints
// ================================================================================================
/* Which constraints form rigid three-atom molecules that SETTLE can treat: the rule of ReferenceConstraints.cpp:44-148, restated.
 *   - only constraints with at least one massive end count (:54-66);
 *   - a candidate atom takes part in exactly two of them, both leading to atoms that also take part in exactly two (:72-78);
 *   - the three atoms must close a triangle (:83-95); every triangle is reported once, ordered by its lowest atom (:91-92);
 *   - the central atom is the one whose two distances are equal AS FLOATS (:76-77, :114-137), the distances handed on are those
 *     floats widened again (the Reference platform runs SETTLE with exactly these values); a triangle without two equal sides
 *     is left to the general solver (:138-139).
 * Output: atoms[4c .. 4c+2] = (central, second, third) as the Reference orders them, atoms[4c+3] = 0; dist[2c] = centre-to-other,
 * dist[2c+1] = the distance between the two others. */
void HipConstraints::findSettleClusters(const System& system, vector<int>& atoms, vector<double>& dist) {
    const int numParticles = system.getNumParticles(), numConstraints = system.getNumConstraints();
    vector<int> degree(numParticles, 0);
    for (int c = 0; c < numConstraints; c++) {
        int a, b; double d;
        system.getConstraintParameters(c, a, b, d);
        if (system.getParticleMass(a) != 0 || system.getParticleMass(b) != 0) { degree[a]++; degree[b]++; }
    }
    // the (at most two) distinct partners of every degree-2 atom among degree-2 atoms; a repeated pair keeps its last distance
    struct Links { int n, other[2]; float d[2]; };
    vector<Links> links(numParticles);
    for (int i = 0; i < numParticles; i++) links[i].n = 0;
    struct Add { static void link(Links& l, int other, float d) {
        for (int k = 0; k < l.n && k < 2; k++)
            if (l.other[k] == other) { l.d[k] = d; return; }
        if (l.n < 2) { l.other[l.n] = other; l.d[l.n] = d; }
        l.n++;                                  // a third distinct partner cannot happen at degree 2; kept for safety: n != 2 disqualifies
    } };
    for (int c = 0; c < numConstraints; c++) {
        int a, b; double d;
        system.getConstraintParameters(c, a, b, d);
        if (system.getParticleMass(a) == 0 && system.getParticleMass(b) == 0) continue;
        if (degree[a] != 2 || degree[b] != 2) continue;
        Add::link(links[a], b, (float) d);
        Add::link(links[b], a, (float) d);
    }
    struct Lookup { static bool find(const Links& l, int other, float& d) {
        for (int k = 0; k < 2 && k < l.n; k++)
            if (l.other[k] == other) { d = l.d[k]; return true; }
        return false;
    } };
    atoms.clear(); dist.clear();
    for (int p1 = 0; p1 < numParticles; p1++) {
        if (links[p1].n != 2) continue;
        const int p2 = min(links[p1].other[0], links[p1].other[1]), p3 = max(links[p1].other[0], links[p1].other[1]);
        if (p1 > p2) continue;                                  // reported from its lowest atom
        float d12 = 0, d13 = 0, d23 = 0;
        if (links[p2].n != 2 || links[p3].n != 2 || !Lookup::find(links[p2], p3, d23)) continue;      // open chain, not a triangle
        Lookup::find(links[p1], p2, d12);
        Lookup::find(links[p1], p3, d13);
        int order[3];
        float dCentre, dOthers;
        if (d12 == d13)      { order[0] = p1; order[1] = p2; order[2] = p3; dCentre = d12; dOthers = d23; }
        else if (d12 == d23) { order[0] = p2; order[1] = p1; order[2] = p3; dCentre = d12; dOthers = d13; }
        else if (d13 == d23) { order[0] = p3; order[1] = p1; order[2] = p2; dCentre = d13; dOthers = d12; }
        else continue;
        atoms.insert(atoms.end(), order, order + 3); atoms.push_back(0);
        dist.push_back((double) dCentre); dist.push_back((double) dOthers);
    }
}

HipConstraints::HipConstraints(const System& system, HipPlatform::PlatformData& data) : hip(*data.hip), numSettle(0), numShake(0), numCcma(0) {
    const int numParticles = system.getNumParticles();
    vector<double> masses(numParticles);
    for (int i = 0; i < numParticles; i++) masses[i] = system.getParticleMass(i);
    vector<bool> isSettleAtom(numParticles, false);

    // ---- SETTLE clusters: found by findSettleClusters below, an own restatement of ReferenceConstraints.cpp:44-148 (same
    //      waters, same central atom, same float-rounded distances as the Reference platform: tests/hip/TestHip.cpp compares the two)
    vector<int> settleAtomsHost;
    vector<double> settleDistHost;
    findSettleClusters(system, settleAtomsHost, settleDistHost);
    numSettle = (int) settleAtomsHost.size() / 4;
    for (int i = 0; i < numSettle; i++)
        for (int k = 0; k < 3; k++) isSettleAtom[settleAtomsHost[4 * i + k]] = true;
    if (numSettle > 0) {
        uploadVector(settleAtoms, settleAtomsHost, hip.stream);
        uploadVector(settleDist, settleDistHost, hip.stream);
    }

    // ---- everything else (ReferenceConstraints.cpp:150-184 sends these to CCMA)
    vector<int> c1, c2;
    vector<double> cd;
    vector<vector<int> > atomConstraints(numParticles);
    for (int i = 0; i < system.getNumConstraints(); i++) {
        int p1, p2;
        double d;
        system.getConstraintParameters(i, p1, p2, d);
        if (masses[p1] == 0 && masses[p2] == 0) continue;
        if (isSettleAtom[p1]) continue;
        atomConstraints[p1].push_back((int) c1.size());
        atomConstraints[p2].push_back((int) c1.size());
        c1.push_back(p1); c2.push_back(p2); cd.push_back(d);
    }
    const int numOther = (int) c1.size();
    // SHAKE clusters: a centre with <= 3 satellites, each satellite constrained only to the centre,
    // the centre constrained only to its satellites, no massless atoms.
    vector<bool> inShake(numOther, false);
    vector<int> shakeAtomsHost;
    vector<double> shakeDistHost;
    for (int atom = 0; atom < numParticles; atom++) {
        const vector<int>& cons = atomConstraints[atom];
        if (cons.empty() || cons.size() > 3) continue;
        bool ok = masses[atom] != 0;
        vector<int> sats;
        for (size_t k = 0; k < cons.size() && ok; k++) {
            int other = c1[cons[k]] == atom ? c2[cons[k]] : c1[cons[k]];
            if (atomConstraints[other].size() != 1 || masses[other] == 0) ok = false;
            sats.push_back(other);
        }
        if (!ok) continue;
        if (cons.size() == 1) {
            // a lone pair of atoms: both ends look like a centre; keep the lower index as centre
            int other = sats[0];
            if (atomConstraints[other].size() == 1 && other < atom) continue;
        }
        int base[4] = {atom, -1, -1, -1};
        double d[4] = {0, 0, 0, 0};
        for (size_t k = 0; k < cons.size(); k++) { base[k + 1] = sats[k]; d[k] = cd[cons[k]]; inShake[cons[k]] = true; }
        for (int k = 0; k < 4; k++) { shakeAtomsHost.push_back(base[k]); shakeDistHost.push_back(d[k]); }
    }
    numShake = (int) shakeAtomsHost.size() / 4;
    if (numShake > 0) {
        uploadVector(shakeAtoms, shakeAtomsHost, hip.stream);
        uploadVector(shakeDist, shakeDistHost, hip.stream);
    }
    // CCMA for the rest
    vector<pair<int, int> > ccmaIndices;
    vector<double> ccmaDistance;
    for (int i = 0; i < numOther; i++)
        if (!inShake[i]) { ccmaIndices.push_back(make_pair(c1[i], c2[i])); ccmaDistance.push_back(cd[i]); }
    numCcma = (int) ccmaIndices.size();
    memset(&ccma, 0, sizeof(ccma));
    if (numCcma > 0) {
        vector<ReferenceCCMAAlgorithm::AngleInfo> angles;
        for (int i = 0; i < system.getNumForces(); i++) {
            const HarmonicAngleForce* force = dynamic_cast<const HarmonicAngleForce*>(&system.getForce(i));
            if (force != NULL)
                for (int j = 0; j < force->getNumAngles(); j++) {
                    int a1, a2, a3;
                    double angle, k;
                    force->getAngleParameters(j, a1, a2, a3, angle, k);
                    angles.push_back(ReferenceCCMAAlgorithm::AngleInfo(a1, a2, a3, angle));
                }
        }
        // The coupling-matrix inverse is built by the reference's own host code (ReferenceCCMAAlgorithm.cpp:57-196).
        ReferenceCCMAAlgorithm host(numParticles, numCcma, ccmaIndices, ccmaDistance, masses, angles, 0.02);
        const vector<vector<pair<int, double> > >& matrix = host.getMatrix();
        vector<int> rowStart(numCcma + 1, 0), col, atoms(2 * (size_t) numCcma);
        vector<double> value;
        for (int i = 0; i < numCcma; i++) {
            atoms[2 * i] = ccmaIndices[i].first; atoms[2 * i + 1] = ccmaIndices[i].second;
            for (size_t j = 0; j < matrix[i].size(); j++) { col.push_back(matrix[i][j].first); value.push_back(matrix[i][j].second); }
            rowStart[i + 1] = (int) col.size();
        }
        uploadVector(ccmaAtoms, atoms, hip.stream);
        uploadVector(ccmaDist, ccmaDistance, hip.stream);
        uploadVector(ccmaRowStart, rowStart, hip.stream);
        uploadVector(ccmaCol, col, hip.stream);
        uploadVector(ccmaValue, value, hip.stream);
        ccmaDelta.allocate(sizeof(double) * numCcma);
        ccmaDelta2.allocate(sizeof(double) * numCcma);
        ccmaConverged.allocate(sizeof(int) * 4);
        ccma.num_constraints = numCcma;
        ccma.atoms = ccmaAtoms.as<int>(); ccma.distance = ccmaDist.as<double>();
        ccma.delta = ccmaDelta.as<double>(); ccma.delta2 = ccmaDelta2.as<double>();
        ccma.row_start = ccmaRowStart.as<int>(); ccma.col = ccmaCol.as<int>(); ccma.value = ccmaValue.as<double>();
        ccma.converged = ccmaConverged.as<int>();
    }

    {   // which algorithm got how many constraints: Context.getPlatform().getPropertyValue(context, "ConstraintPartition") (tests)
        stringstream partition;
        partition << "settle " << numSettle << " shake " << numShake << " ccma " << numCcma;
        data.propertyValues[HipPlatform::HipConstraintPartition()] = partition.str();
    }

    // ---- integration units of the fused step: waters, SHAKE clusters, then every remaining atom on its own
    numUnits = 0;
    totalMass = 0;
    for (int i = 0; i < numParticles; i++) totalMass += masses[i];
    const char* noFused = getenv("OPENMM_HIP_DISABLE_FUSED_STEP");
    if (numCcma == 0 && numParticles > 0 && !(noFused != NULL && noFused[0] == '1')) {
        vector<int> unitAtomsHost;
        vector<double> unitDistHost;
        vector<bool> covered(numParticles, false);
        if (numSettle > 0) {
            for (int i = 0; i < numSettle; i++) {
                const int at[4] = {settleAtomsHost[4 * i], settleAtomsHost[4 * i + 1], settleAtomsHost[4 * i + 2], -1};
                const double d[4] = {settleDistHost[2 * i], settleDistHost[2 * i + 1], 0.0, 1.0};
                unitAtomsHost.insert(unitAtomsHost.end(), at, at + 4);
                unitDistHost.insert(unitDistHost.end(), d, d + 4);
                covered[at[0]] = covered[at[1]] = covered[at[2]] = true;
            }
        }
        for (int i = 0; i < numShake; i++) {
            for (int k = 0; k < 4; k++) {
                const int atom = shakeAtomsHost[4 * i + k];
                unitAtomsHost.push_back(atom);
                unitDistHost.push_back(k < 3 ? shakeDistHost[4 * i + k] : 2.0);
                if (atom >= 0) covered[atom] = true;
            }
        }
        for (int i = 0; i < numParticles; i++)
            if (!covered[i]) {
                const int at[4] = {i, -1, -1, -1};
                const double d[4] = {0.0, 0.0, 0.0, 0.0};
                unitAtomsHost.insert(unitAtomsHost.end(), at, at + 4);
                unitDistHost.insert(unitDistHost.end(), d, d + 4);
            }
        allUnitAtoms.swap(unitAtomsHost);
        allUnitDist.swap(unitDistHost);
        // no SHAKE cluster (kind 2 in dist.w) anywhere: every unit is a SETTLE water or a free atom, at most three atoms -- the fused
        // step can run its register-lean variant (integrate.hip, k_step_units<KIND, SMALL>)
        smallUnits = true;
        for (size_t i = 3; i < allUnitDist.size(); i += 4)
            if (allUnitDist[i] == 2.0 || allUnitAtoms[i] >= 0) { smallUnits = false; break; }
        const size_t maxUnits = allUnitAtoms.size() / 4;
        unitAtoms.allocate(sizeof(int) * 4 * max(maxUnits, (size_t) 1));
        unitDist.allocate(sizeof(double) * 4 * max(maxUnits, (size_t) 1));
        cmScratch.allocate(sizeof(double) * (4 + 4 * ((maxUnits + 127) / 128)));
        HIP_CHECK(ommhip_memset(cmScratch.ptr, 0, cmScratch.bytes, hip.stream));
        uploadOwnedUnits();
    }
    hip.addListener(this);
}

HipConstraints::~HipConstraints() {
    hip.removeListener(this);
}

void HipConstraints::atomsReordered() {
    // (the momentum trailers stay valid: ownership changed, the velocities and hence the sum over the ranks did not)
    if (hip.decomposed()) uploadOwnedUnits();
}

void HipConstraints::uploadOwnedUnits() {
    const size_t total = allUnitAtoms.size() / 4;
    if (total == 0) return;
    if (!hip.decomposed()) {
        numUnits = (int) total;
        HIP_CHECK(ommhip_memcpy_h2d(unitAtoms.ptr, allUnitAtoms.data(), sizeof(int) * allUnitAtoms.size(), hip.stream));
        HIP_CHECK(ommhip_memcpy_h2d(unitDist.ptr, allUnitDist.data(), sizeof(double) * allUnitDist.size(), hip.stream));
        HIP_CHECK(ommhip_stream_sync(hip.stream));
        return;
    }
    // a unit belongs to the rank that holds its first atom (constraint-connected atoms are never split between ranks)
    std::vector<int> atoms;
    std::vector<double> dist;
    for (size_t u = 0; u < total; u++) {
        const int s = hip.hostSlotOfAtom[allUnitAtoms[4 * u]];
        if (s < hip.ownSlot0 || s >= hip.ownSlot1) continue;
        atoms.insert(atoms.end(), allUnitAtoms.begin() + 4 * u, allUnitAtoms.begin() + 4 * u + 4);
        dist.insert(dist.end(), allUnitDist.begin() + 4 * u, allUnitDist.begin() + 4 * u + 4);
    }
    numUnits = (int) atoms.size() / 4;
    if (numUnits > 0) {
        HIP_CHECK(ommhip_memcpy_h2d(unitAtoms.ptr, atoms.data(), sizeof(int) * atoms.size(), hip.stream));
        HIP_CHECK(ommhip_memcpy_h2d(unitDist.ptr, dist.data(), sizeof(double) * dist.size(), hip.stream));
    }
    HIP_CHECK(ommhip_stream_sync(hip.stream));
}

void HipConstraints::fusedStep(int integrator, const ommhip_integrator_state& state, double tol) {
    ommhip_step_units u;
    u.num_units = numUnits;
    u.atoms = unitAtoms.as<int>(); u.dist = unitDist.as<double>();
    u.tol = tol; u.max_iterations = 150;
    u.remove_cm = hip.cmRemovalPending && hip.momentumValid ? 1 : 0;
    u.inv_total_mass = totalMass > 0 ? 1.0 / totalMass : 0.0;
    u.cm_scratch = cmScratch.as<double>();
    u.pos_wire = NULL; u.ranks = 1; u.rank = 0; u.slots_per_rank = 0; u.trailer_slot = 0; u.dd_flags = NULL;
    static const bool noSmall = getenv("OPENMM_HIP_NO_SMALL_UNITS") != NULL;         // A/B knob
    u.small_units = smallUnits && !noSmall ? 1 : 0;
    u.box_len[0] = hip.box[0]; u.box_len[1] = hip.box[2]; u.box_len[2] = hip.box[5];
    u.box_skew[0] = hip.box[1]; u.box_skew[1] = hip.box[3]; u.box_skew[2] = hip.box[4];
    if (hip.decomposed()) {
        u.pos_wire = hip.posWire.ptr; u.ranks = hip.domain.ranks; u.rank = hip.domain.rank;
        u.slots_per_rank = hip.slotsPerRank; u.trailer_slot = hip.trailerSlot;
        u.dd_flags = hip.haloMode ? hip.ddFlags.as<int>() : NULL;
    }
    HIP_CHECK(ommhip_integrate_fused(integrator, &state, &u, hip.stream));
    // the new positions travel on the same stream: the boundary sections to the two neighbouring slabs (halo mode; the momentum
    // trailers go to everybody in the same group), or everything to everybody (in-place all-gather)
    if (hip.decomposed()) hip.exchangePositions();
    hip.cmRemovalPending = false;
    hip.momentumValid = true;
}

void HipConstraints::runCcma(void* target, bool velocities, double tol, void* reference) {
    // ReferenceCCMAAlgorithm.cpp:240-310 as a device-resident loop: batches of four iterations are enqueued without waiting;
    // the kernels of an iteration leave at once when an earlier one found every constraint converged, and the host looks at that
    // flag once per batch (the reference's GPU platforms poll a mapped flag every few iterations, CudaIntegrationUtilities.cpp:94-130).
    HIP_CHECK(ommhip_memset(ccma.converged, 0, sizeof(int) * 4, hip.stream));
    const int batch = 4;
    for (int done = 0; done < 150; done += batch) {
        HIP_CHECK(ommhip_ccma_iterations(&ccma, reference, target, hip.vel.ptr, velocities ? 1 : 0, tol, batch, hip.stream));
        int state[4] = {0, 0, 0, 0};
        HIP_CHECK(ommhip_memcpy_d2h(state, ccma.converged, sizeof(state), hip.stream));
        hip.sync();
        if (state[2] != 0) break;
    }
}

void HipConstraints::apply(void* target, double tol, void* reference) {
    hip.noteStateMutation();
    if (reference == NULL) reference = hip.pos.ptr;
    if (numCcma > 0) runCcma(target, false, tol, reference);
    HIP_CHECK(ommhip_constrain_clusters(numShake, shakeAtoms.as<int>(), shakeDist.as<double>(), numSettle, settleAtoms.as<int>(), settleDist.as<double>(),
                                        reference, target, hip.vel.ptr, 0, tol, 150, hip.stream));
}

void HipConstraints::applyToVelocities(void* target, double tol, void* reference) {
    if (target == hip.vel.ptr) hip.noteStateMutation();          // (the kinetic-energy query constrains a COPY of the velocities: not a mutation)
    if (reference == NULL) reference = hip.pos.ptr;
    if (numCcma > 0) runCcma(target, true, tol, reference);
    HIP_CHECK(ommhip_constrain_clusters(numShake, shakeAtoms.as<int>(), shakeDist.as<double>(), numSettle, settleAtoms.as<int>(), settleDist.as<double>(),
                                        reference, target, hip.vel.ptr, 1, tol, 150, hip.stream));
}

// ================================================================================================
// CalcForcesAndEnergy
// ================================================================================================
void HipCalcForcesAndEnergyKernel::initialize(const System& system) {
}

void HipCalcForcesAndEnergyKernel::beginComputation(ContextImpl& context, bool includeForce, bool includeEnergy, int groups) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    if (hip.hostMode) {
        // Host vectors are authoritative: bring positions and box to the device for the native force kernels.
        hip.setBox(data.periodicBoxVectors[0], data.periodicBoxVectors[1], data.periodicBoxVectors[2]);
        hip.uploadPositions(*data.positions);
        hip.requestReorder();
    }
    hip.reorderIfNeeded();
    if (hip.hostMode || hip.hasFallbackForces) {
        // what ReferenceCalcForcesAndEnergyKernel::beginComputation does (ReferenceKernels.cpp:181-195)
        if (!hip.hostMode) hip.downloadPositions(*data.positions);
        vector<Vec3>& forces = *data.forces;
        if (includeForce)
            for (size_t i = 0; i < forces.size(); i++) forces[i] = Vec3();
        else
            savedHostForces = forces;
    }
    for (map<string, double>::const_iterator it = context.getParameters().begin(); it != context.getParameters().end(); ++it)
        (*data.energyParameterDerivatives)[it->first] = 0;
    // An energy-only evaluation must leave the forces of the last full evaluation intact (ReferenceKernels.cpp:190-191,198-199):
    // the leapfrog kinetic energy reads them afterwards.
    if (!includeForce && !hip.hostMode && !hip.forcesRecomputedEveryStep)
        hip.saveForces();
    hip.clearForces();
    hip.beginEvaluation(groups);
}

double HipCalcForcesAndEnergyKernel::finishComputation(ContextImpl& context, bool includeForce, bool includeEnergy, int groups, bool& valid) {
    HipContext& hip = *data.hip;
    hip.ensureCleared();          // nobody folded the start-of-evaluation clear into a launch of its own
    hip.flushTerms();
    hip.flushValence();
    hip.joinPme();
    if (hip.decomposed()) hip.returnHaloForces();      // half-shell evaluation: what this rank computed on its lower neighbour's atoms goes home (same call order on every rank)
    double energy = 0;
    if (includeEnergy) energy = hip.reduceEnergy();
    if (includeEnergy && !hip.hostMode && !hip.recovering() && !hip.decomposed() && hip.listOverflowSeen && hip.listOverflowSeen()) {
        // The neighbour list overflowed at a device-triggered rebuild (the device has been skipping the integration since): an energy
        // summed over an incomplete list must not leave the platform.  Undo this evaluation, grow the list, redo the skipped steps, and
        // let ContextImpl::calcForcesAndEnergy (ContextImpl.cpp:298-307) evaluate again -- the reference's own remedy for an overflow.
        if (!includeForce) {
            if (!hip.forcesRecomputedEveryStep) hip.restoreForces();
            // the fallback (Reference) kernels of this pass added to the host array: the retry must start from what was saved
            if (hip.hasFallbackForces) *data.forces = savedHostForces;
        }
        if (hip.replaySteps) hip.recoverIfFrozen();
        else hip.listRecovery();                  // no step was ever taken: nothing to redo, the list is grown and rebuilt
        valid = false;
        return 0;
    }
    if (hip.hostMode) {
        if (includeForce) {
            vector<Vec3> deviceForces;
            hip.downloadForces(deviceForces);
            vector<Vec3>& forces = *data.forces;
            for (size_t i = 0; i < forces.size(); i++) forces[i] += deviceForces[i];
            // ReferenceKernels.cpp:200: virtual-site forces go to their parent atoms (host mode only has sites)
            ReferenceVirtualSites::distributeForces(context.getSystem(), *data.positions, forces);
        }
        else
            *data.forces = savedHostForces;
    }
    else {
        if (!includeForce) {
            if (!hip.forcesRecomputedEveryStep) hip.restoreForces();
            if (hip.hasFallbackForces) *data.forces = savedHostForces;
        }
        else if (hip.hasFallbackForces)
            hip.addHostForces(*data.forces);
    }
    valid = true;
    return energy;
}

// ================================================================================================
// UpdateStateData
// ================================================================================================
void HipUpdateStateDataKernel::initialize(const System& system) {
}
double HipUpdateStateDataKernel::getTime(const ContextImpl& context) const { return data.time; }
void HipUpdateStateDataKernel::setTime(ContextImpl& context, double time) { data.time = time; }
void HipUpdateStateDataKernel::getPositions(ContextImpl& context, vector<Vec3>& positions) {
    data.hip->setAsCurrent();
    data.hip->downloadPositions(positions);
}
void HipUpdateStateDataKernel::setPositions(ContextImpl& context, const vector<Vec3>& positions) {
    data.hip->setAsCurrent();
    data.hip->uploadPositions(positions);
    data.hip->requestReorder();
    *data.positions = positions;
}
void HipUpdateStateDataKernel::getVelocities(ContextImpl& context, vector<Vec3>& velocities) {
    data.hip->setAsCurrent();
    data.hip->downloadVelocities(velocities);
}
void HipUpdateStateDataKernel::setVelocities(ContextImpl& context, const vector<Vec3>& velocities) {
    data.hip->setAsCurrent();
    data.hip->uploadVelocities(velocities);
}
void HipUpdateStateDataKernel::getForces(ContextImpl& context, vector<Vec3>& forces) {
    data.hip->setAsCurrent();
    data.hip->downloadForces(forces);
}
void HipUpdateStateDataKernel::getEnergyParameterDerivatives(ContextImpl& context, map<string, double>& derivs) {
    derivs = *data.energyParameterDerivatives;
}
void HipUpdateStateDataKernel::getPeriodicBoxVectors(ContextImpl& context, Vec3& a, Vec3& b, Vec3& c) const {
    data.hip->getBox(a, b, c);
}
void HipUpdateStateDataKernel::setPeriodicBoxVectors(ContextImpl& context, const Vec3& a, const Vec3& b, const Vec3& c) {
    data.hip->setBox(a, b, c);
    data.periodicBoxVectors[0] = a; data.periodicBoxVectors[1] = b; data.periodicBoxVectors[2] = c;
    *data.periodicBoxSize = Vec3(a[0], b[1], c[2]);
}
void HipUpdateStateDataKernel::createCheckpoint(ContextImpl& context, ostream& stream) {
    // same content as ReferenceUpdateStateDataKernel::createCheckpoint (ReferenceKernels.cpp:282-294).  The thermostat noise is a
    // pure function of (seed, stepCount, atom): instead of a generator state the checkpoint carries the resolved seed, so a run
    // restarted in another Context or process (where seed 0 would resolve differently) continues the same noise stream.
    // Version 3 adds what a CustomIntegrator's noise hangs on: the draw counter of its device computations and the state of the host
    // generator its ComputeGlobal steps (and the barostat, and Reference kernels in host mode) draw from -- the last item is what
    // ReferenceUpdateStateDataKernel::createCheckpoint saves too.
    int version = 3;
    stream.write((char*) &version, sizeof(int));
    stream.write((char*) &data.time, sizeof(double));
    stream.write((char*) &data.stepCount, sizeof(int));
    stream.write((char*) &data.integratorSeed, sizeof(unsigned long long));
    stream.write((char*) &data.customDraws, sizeof(unsigned long long));
    vector<Vec3> pos, vel;
    data.hip->downloadPositions(pos);
    data.hip->downloadVelocities(vel);
    stream.write((char*) pos.data(), sizeof(Vec3) * pos.size());
    stream.write((char*) vel.data(), sizeof(Vec3) * vel.size());
    Vec3 box[3];
    data.hip->getBox(box[0], box[1], box[2]);
    stream.write((char*) box, 3 * sizeof(Vec3));
    SimTKOpenMMUtilities::createCheckpoint(stream);
}
void HipUpdateStateDataKernel::loadCheckpoint(ContextImpl& context, istream& stream) {
    int version;
    stream.read((char*) &version, sizeof(int));
    if (version != 2 && version != 3) throw OpenMMException("Checkpoint was created with a different version of OpenMM");
    stream.read((char*) &data.time, sizeof(double));
    stream.read((char*) &data.stepCount, sizeof(int));
    stream.read((char*) &data.integratorSeed, sizeof(unsigned long long));
    if (version >= 3) stream.read((char*) &data.customDraws, sizeof(unsigned long long));
    vector<Vec3> pos(data.hip->numAtoms), vel(data.hip->numAtoms);
    stream.read((char*) pos.data(), sizeof(Vec3) * pos.size());
    stream.read((char*) vel.data(), sizeof(Vec3) * vel.size());
    Vec3 box[3];
    stream.read((char*) box, 3 * sizeof(Vec3));
    if (!stream.good()) throw OpenMMException("HIP platform: the checkpoint is truncated");
    // Version 3 carries the state of the host generator (SimTKOpenMMUtilities: ONE generator per process, as on the Reference platform --
    // loading a checkpoint in one Context therefore also sets the host draws of every other Context of the process).  A version-2 blob has
    // neither that nor the draw counter of a device CustomIntegrator: both restart from the checkpoint's own seed, so that a restart from
    // such a blob is at least reproducible instead of continuing from whatever the live Context had drawn.
    if (version >= 3) SimTKOpenMMUtilities::loadCheckpoint(stream);
    else {
        data.customDraws = 0;
        SimTKOpenMMUtilities::setRandomNumberSeed((uint32_t) data.integratorSeed);
    }
    setPeriodicBoxVectors(context, box[0], box[1], box[2]);
    setPositions(context, pos);
    setVelocities(context, vel);
}

// ================================================================================================
// ApplyConstraints
// ================================================================================================
void HipApplyConstraintsKernel::initialize(const System& system) {
}
void HipApplyConstraintsKernel::apply(ContextImpl& context, double tol) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    HipConstraints& constraints = data.getDeviceConstraints(context.getSystem());
    if (!constraints.hasConstraints()) return;
    if (hip.decomposed()) hip.gatherState();       // every rank constrains all atoms (identical results), then refills the all-gather buffer
    // ReferenceKernels.cpp:318-324: constrain the current positions in place
    HIP_CHECK(ommhip_memcpy_d2d(hip.xp.ptr, hip.pos.ptr, hip.pos.bytes, hip.stream));
    constraints.apply(hip.xp.ptr, tol);
    HIP_CHECK(ommhip_memcpy_d2d(hip.pos.ptr, hip.xp.ptr, hip.pos.bytes, hip.stream));
    if (hip.decomposed()) hip.fillWireFromPos();
}
void HipApplyConstraintsKernel::applyToVelocities(ContextImpl& context, double tol) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    HipConstraints& constraints = data.getDeviceConstraints(context.getSystem());
    if (!constraints.hasConstraints()) return;
    if (hip.decomposed()) hip.gatherState();
    constraints.applyToVelocities(hip.vel.ptr, tol);
    hip.momentumValid = false;
}

// ================================================================================================
// NonbondedForce
// ================================================================================================
static vector<HipCalcNonbondedForceKernel*> liveNonbondedKernels;
static std::mutex liveNonbondedMutex;          // (the inner Contexts of a device list are created and destroyed on threads of their own)

HipCalcNonbondedForceKernel::HipCalcNonbondedForceKernel(string name, const Platform& platform, HipPlatform::PlatformData& data) :
        CalcNonbondedForceKernel(name, platform), data(data), hip(*data.hip), numParticles(0), num14(0), numExclusionPairs(0),
        slotParamsDirty(true), forceRebuild(true), etermDirty(true), hasInitializedParams(false), pinnedState(NULL), stateCopyPending(false),
        maxCharge(0.0), maxChargeDirty(true) {
    memset(&nl, 0, sizeof(nl));
    memset(&params, 0, sizeof(params));
    memset(&pme, 0, sizeof(pme));
    memset(&pmeDisp, 0, sizeof(pmeDisp));
    hip.addListener(this);
    { std::lock_guard<std::mutex> lock(liveNonbondedMutex); liveNonbondedKernels.push_back(this); }
}

/* Diagnostics for bench.py: neighbour-list occupancy of the most recently created native NonbondedForce kernel. */
extern "C" __attribute__((visibility("default"))) int ommhip_plugin_nl_stats(long long* out) {
    if (liveNonbondedKernels.empty()) return 1;
    try { liveNonbondedKernels.back()->getNeighborListStats(out); } catch (...) { return 2; }
    return 0;
}

/* Diagnostics for tests and bench.py: how the most recently created decomposed Context exchanges positions.
 * out[0] ranks, [1] 1 = halo mode (sections to the two neighbouring slabs) / 0 = replicated (all-gather), [2] slots per rank,
 * [3] slots this rank converts per step (own + received sections; all slots when replicated), [4] bytes it sends per step,
 * [5] bytes it receives per step, [6] re-sorts so far */
extern "C" __attribute__((visibility("default"))) int ommhip_plugin_dd_info(long long* out) {
    if (liveNonbondedKernels.empty()) return 1;
    try { liveNonbondedKernels.back()->getDomainInfo(out); } catch (...) { return 2; }
    return 0;
}

void HipCalcNonbondedForceKernel::getDomainInfo(long long* out) {
    const int R = hip.domain.ranks, me = hip.domain.rank;
    out[0] = R; out[1] = hip.haloMode ? 1 : 0; out[2] = hip.slotsPerRank;
    const long long rec = 16;
    if (hip.haloMode) {
        long long active = 0;
        for (int r = 0; r < hip.numActiveRanges; r++) active += hip.activeRange[2 * r + 1] - hip.activeRange[2 * r];
        const int above = (me + 1) % R, below = (me + R - 1) % R;
        out[3] = active;
        out[4] = (long long) hip.haloPlan.down_bytes[me] + (long long) hip.haloPlan.up_bytes[me] + (long long) (R - 1) * hip.haloPlan.trailer_bytes;
        out[5] = (long long) hip.haloPlan.down_bytes[above] + (long long) hip.haloPlan.up_bytes[below] + (long long) (R - 1) * hip.haloPlan.trailer_bytes;
    }
    else { out[3] = hip.paddedAtoms; out[4] = rec * hip.slotsPerRank * (R - 1); out[5] = out[4]; }
    out[6] = hip.reorderCount;
    out[7] = hip.halfShell ? 1 + (long long) (hip.evalRange[1] - hip.evalRange[0]) : 0;       // 0: pairs across a boundary on both sides; else 1 + the slots of the lower neighbour's section (forces returned)
}

/* Diagnostics: per i-block cost of the last list build (clock ticks, candidate blocks), as left by nl_find_interactions. */
/* Diagnostics (tools/time_resort_host.py): host time of the order computation of a decomposed run -- a HipContext of `ranks` ranks over a
 * communicator that is never used, reach / PME reach as a PME NonbondedForce with a 0.9 nm cutoff on a grid of `nx` planes would set them. */
static int unusedAllGather(void*, const void*, void*, size_t) { return 1; }
extern "C" __attribute__((visibility("default"))) double ommhip_plugin_time_decomposed_order(const void* system, const double* xyz, int ranks, int rank, int nx, int repeats, long long* info) {
    try {
        const System& sys = *(const System*) system;
        HipDomain domain;
        domain.ranks = ranks; domain.rank = rank;
        if (ommhip_comm_create_callback(unusedAllGather, NULL, rank, ranks, &domain.comm) != 0) return -1.0;
        HipContext hip(sys, 0, false, domain);
        hip.usePeriodic = true; hip.sortCutoff = 0.9; hip.haloReach = 0.9 * 1.15;
        hip.pmeReachX = 6.0 * hip.box[0] / nx; hip.pmeReachBelow = 6.0 * hip.box[0] / nx; hip.pmeReachAbove = 2.0 * hip.box[0] / nx;
        vector<Vec3> positions(sys.getNumParticles());
        for (int i = 0; i < sys.getNumParticles(); i++) positions[i] = Vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        vector<int> atomOfSlot;
        const double ms = hip.timeDecomposedOrder(positions, repeats, &atomOfSlot);
        if (info != NULL) {
            // how compact are this rank's 32-slot blocks?  Largest edge of each block's bounding box (minimum image relative to its first atom), in pm
            vector<long long> edges;
            for (int b = hip.ownSlot0 / OMMHIP_TILE; b < hip.ownSlot1 / OMMHIP_TILE; b++) {
                double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
                int first = -1;
                for (int k = 0; k < OMMHIP_TILE; k++) {
                    const int atom = atomOfSlot[(size_t) b * OMMHIP_TILE + k];
                    if (atom < 0) continue;
                    if (first < 0) first = atom;
                    for (int c = 0; c < 3; c++) {
                        const double len = c == 0 ? hip.box[0] : (c == 1 ? hip.box[2] : hip.box[5]);
                        double d = positions[atom][c] - positions[first][c];
                        d -= floor(d / len + 0.5) * len;
                        lo[c] = min(lo[c], d); hi[c] = max(hi[c], d);
                    }
                }
                if (first >= 0) edges.push_back((long long) (1000.0 * max(hi[0] - lo[0], max(hi[1] - lo[1], hi[2] - lo[2]))));
                if (first >= 0 && getenv("OPENMM_HIP_DD_DEBUG") != NULL && max(hi[0] - lo[0], max(hi[1] - lo[1], hi[2] - lo[2])) > 5.0)
                    fprintf(stderr, "  big block %d of the rank's %d (sections end at blocks %d %d %d): extent %.2f %.2f %.2f\n", b - hip.ownSlot0 / OMMHIP_TILE, hip.slotsPerRank / OMMHIP_TILE,
                            (int) (hip.haloPlan.up_offset[rank] / 16 / OMMHIP_TILE), (int) (hip.haloPlan.down_bytes[rank] / 16 / OMMHIP_TILE), (int) ((hip.haloPlan.up_offset[rank] + hip.haloPlan.up_bytes[rank]) / 16 / OMMHIP_TILE),
                            hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]);
            }
            std::sort(edges.begin(), edges.end());
            if (!edges.empty()) { info[6] = edges[edges.size() / 2]; info[7] = edges[edges.size() * 99 / 100]; info[8] = edges.back(); info[9] = (long long) edges.size(); }
            info[0] = hip.haloMode ? 1 : 0; info[1] = hip.halfShell ? 1 : 0; info[2] = hip.slotsPerRank;
            long long active = 0;
            for (int r = 0; r < hip.numActiveRanges; r++) active += hip.activeRange[2 * r + 1] - hip.activeRange[2 * r];
            info[3] = active; info[4] = hip.evalRange[1] - hip.evalRange[0]; info[5] = (long long) (1e6 * hip.haloDrift);
        }
        return ms;
    } catch (const std::exception& e) { fprintf(stderr, "ommhip_plugin_time_decomposed_order: %s\n", e.what()); return -1.0; }
}

/* Test hook: the SETTLE partition this platform makes of a System (HipConstraints::findSettleClusters); returns the number of
 * clusters, fills at most `capacity` of them. */
extern "C" __attribute__((visibility("default"))) int ommhip_plugin_settle_clusters(const void* system, int* atoms, double* dist, int capacity) {
    vector<int> a;
    vector<double> d;
    HipConstraints::findSettleClusters(*(const System*) system, a, d);
    const int n = (int) a.size() / 4;
    for (int i = 0; i < n && i < capacity; i++) {
        for (int k = 0; k < 3; k++) atoms[3 * i + k] = a[4 * i + k];
        dist[2 * i] = d[2 * i]; dist[2 * i + 1] = d[2 * i + 1];
    }
    return n;
}

extern "C" __attribute__((visibility("default"))) int ommhip_plugin_nl_block_costs(float* ticks, float* candidates, int maxBlocks) {
    if (liveNonbondedKernels.empty()) return -1;
    try { return liveNonbondedKernels.back()->getBlockCosts(ticks, candidates, maxBlocks); } catch (...) { return -2; }
}

extern "C" __attribute__((visibility("default"))) int ommhip_plugin_nl_block_diag(float* out, int columns, int maxBlocks) {
    // per i-block diagnostics of the last list build, `columns` (<= 8) floats each: builder ticks, candidate blocks, ticks until the
    // candidates were collected, ticks inside flushes, entries written
    if (liveNonbondedKernels.empty()) return -1;
    try { return liveNonbondedKernels.back()->getBlockDiag(out, columns, maxBlocks); } catch (...) { return -2; }
}

extern "C" __attribute__((visibility("default"))) int ommhip_plugin_nl_block_halves(float* out, int maxBlocks) {
    // half extents (x, y, z, flag) of the i-blocks' bounding boxes at the last evaluation
    if (liveNonbondedKernels.empty()) return -1;
    try { return liveNonbondedKernels.back()->getBlockHalves(out, maxBlocks); } catch (...) { return -2; }
}

int HipCalcNonbondedForceKernel::getBlockHalves(float* out, int maxBlocks) {
    hip.setAsCurrent();
    const int n = min(hip.paddedAtoms / OMMHIP_TILE, maxBlocks);
    HIP_CHECK(ommhip_memcpy_d2h(out, blockHalf.ptr, sizeof(float) * 4 * (size_t) n, hip.stream));
    hip.sync();
    return n;
}

int HipCalcNonbondedForceKernel::getBlockDiag(float* out, int columns, int maxBlocks) {
    hip.setAsCurrent();
    const int numBlocks = hip.paddedAtoms / OMMHIP_TILE;
    vector<float> ref(4 * (size_t) hip.paddedAtoms);
    HIP_CHECK(ommhip_memcpy_d2h(ref.data(), posqRef.ptr, sizeof(float) * ref.size(), hip.stream));
    hip.sync();
    const int n = min(numBlocks, maxBlocks), cols = min(columns, 8);
    for (int b = 0; b < n; b++)
        for (int c = 0; c < cols; c++) out[(size_t) b * columns + c] = ref[4 * (size_t) (b * OMMHIP_TILE + c) + 3];
    return n;
}

int HipCalcNonbondedForceKernel::getBlockCosts(float* ticks, float* candidates, int maxBlocks) {
    hip.setAsCurrent();
    const int numBlocks = hip.paddedAtoms / OMMHIP_TILE;
    vector<float> ref(4 * (size_t) hip.paddedAtoms);
    HIP_CHECK(ommhip_memcpy_d2h(ref.data(), posqRef.ptr, sizeof(float) * ref.size(), hip.stream));
    hip.sync();
    const int n = min(numBlocks, maxBlocks);
    for (int b = 0; b < n; b++) { ticks[b] = ref[4 * (size_t) (b * OMMHIP_TILE) + 3]; candidates[b] = ref[4 * (size_t) (b * OMMHIP_TILE + 1) + 3]; }
    return n;
}

void HipCalcNonbondedForceKernel::getNeighborListStats(long long* out) {
    hip.setAsCurrent();
    int state[OMMHIP_NL_STATE_INTS];
    HIP_CHECK(ommhip_memcpy_d2h(state, nlState.ptr, sizeof(state), hip.stream));
    hip.sync();
    // [2], [3]: chunks and rows the pair kernel walks (the per-step pruned list when there is one); [6], [7]: those of the list as built
    const bool pruned = nl.row_j_inner != NULL && state[10] == 0 && getenv("OPENMM_HIP_NO_PRUNE") == NULL && nl.pbc != 2;
    long long count[2][2] = {{0, 0}, {0, 0}};
    for (int which = 0; which < 2; which++) {
        const int chunks = min(which == 0 ? state[1] : state[7], nl.max_chunks);
        if (which == 1 && !pruned) { count[1][0] = count[0][0]; count[1][1] = count[0][1]; break; }
        vector<int> info(2 * (size_t) max(chunks, 1));
        if (chunks > 0) {
            HIP_CHECK(ommhip_memcpy_d2h(info.data(), which == 0 ? chunkInfo.ptr : chunkInfoInner.ptr, sizeof(int) * 2 * (size_t) chunks, hip.stream));
            hip.sync();
        }
        long long rows = 0;
        for (int c = 0; c < chunks; c++) rows += info[2 * c + 1] & 0xff;
        count[which][0] = chunks; count[which][1] = rows;
    }
    out[0] = numParticles; out[1] = hip.paddedAtoms; out[2] = count[1][0]; out[3] = count[1][1]; out[4] = nl.max_chunks; out[5] = state[4];
    out[6] = count[0][0]; out[7] = count[0][1];
}

static vector<double> bsplineModuli(int n);

void HipCalcNonbondedForceKernel::setupPmeDecomposed() {
    // buffers of ommhip_pme_reciprocal_dd (layouts: include/openmm_hip_kernels.h, ommhip_pme)
    const int R = hip.domain.ranks, nx = gridSize[0], ny = gridSize[1], nz = gridSize[2], nzc = nz / 2 + 1, nxl = nx / R, nyl = ny / R;
    ddHalo = getenv("OPENMM_HIP_DD_HALO") != NULL ? atoi(getenv("OPENMM_HIP_DD_HALO")) : 6;     // potential planes kept beyond the slab on each side
    ddHalo = max(0, min(ddHalo, nxl - 4));
    if (nxl < 5)
        throw OpenMMException("HIP platform: the PME grid is too coarse along x for this many GPUs (fewer than 5 planes per rank)");
    uploadVector(moduliX, bsplineModuli(nx), hip.stream);
    uploadVector(moduliY, bsplineModuli(ny), hip.stream);
    uploadVector(moduliZ, bsplineModuli(nz), hip.stream);
    DeviceBuffer* tw[3] = {&twiddleX, &twiddleY, &twiddleZ};
    for (int d = 0; d < 3; d++) {
        const int n = gridSize[d];
        vector<float> t(2 * (size_t) n);
        for (int k = 0; k < n; k++) { t[2 * k] = (float) cos(2.0 * M_PI * k / n); t[2 * k + 1] = (float) -sin(2.0 * M_PI * k / n); }
        uploadVector(*tw[d], t, hip.stream);
    }
    eterm.allocate(sizeof(float) * (size_t) nx * nyl * nzc);
    const size_t gridBytes = (sizeof(float) * (size_t) (nxl + 2 * ddHalo + 4) * ny * nz + 15) / 16 * 16;
    gridReal.allocate(gridBytes);
    if (!enableTileSpread(nx, ny, nz) && hip.extraClearPtr == NULL) { hip.extraClearPtr = gridReal.ptr; hip.extraClearBytes = gridBytes; pme.grid_precleared = 1; }
    gridComplex.allocate(sizeof(float) * 2 * (size_t) nxl * ny * nzc);
    gridComplex2.allocate(sizeof(float) * 2 * (size_t) nx * nyl * nzc);
    ddError.allocate(sizeof(int) * 4);
    HIP_CHECK(ommhip_memset(ddError.ptr, 0, ddError.bytes, hip.stream));
    HIP_CHECK(ommhip_host_malloc((void**) &pinnedDdError, sizeof(int) * 4));
    pinnedDdError[0] = 0;
    pme.nx = nx; pme.ny = ny; pme.nz = nz; pme.alpha = ewaldAlpha;
    pme.moduli_x = moduliX.as<double>(); pme.moduli_y = moduliY.as<double>(); pme.moduli_z = moduliZ.as<double>();
    pme.eterm = eterm.ptr; pme.grid_real = gridReal.ptr; pme.grid_complex = gridComplex.ptr; pme.grid_complex2 = gridComplex2.ptr;
    pme.twiddle_x = twiddleX.ptr; pme.twiddle_y = twiddleY.ptr; pme.twiddle_z = twiddleZ.ptr;
    pme.dd_ranks = R; pme.dd_rank = hip.domain.rank; pme.dd_halo = ddHalo; pme.comm = hip.domain.comm; pme.dd_error = ddError.as<int>();
    // a rank spreads every atom whose order-5 stencil touches its planes: atoms up to five cells below and one above them
    hip.pmeReachX = max(hip.pmeReachX, 6.0 * hip.box[0] / nx);
    hip.pmeReachBelow = max(hip.pmeReachBelow, 6.0 * hip.box[0] / nx); hip.pmeReachAbove = max(hip.pmeReachAbove, 2.0 * hip.box[0] / nx);
    etermDirty = true;
}

void HipCalcNonbondedForceKernel::checkDecomposedFlags() {
    // filled by an asynchronous copy queued a few evaluations ago: an owned atom left the potential planes this rank holds
    if (pinnedDdError != NULL && pinnedDdError[0] != 0)
        throw OpenMMException("HIP platform: an atom drifted out of the PME planes its rank holds between two re-sorts; "
                              "lower OPENMM_HIP_REORDER_INTERVAL or raise OPENMM_HIP_DD_HALO");
}

double HipCalcNonbondedForceKernel::executeDecomposed(ContextImpl& context, bool includeForces, bool includeEnergy, bool includeDirect, bool includeReciprocal) {
    // One rank of a decomposed evaluation.  Separate launches on one stream: positions (replicated) -> posq,
    // list for the owned blocks, charge spreading on the owned planes, slab FFT with two all-to-alls, pair kernel for the
    // owned blocks, halo planes, interpolation for the owned atoms.
    if (nonbondedMethod != PME || !includeDirect || !includeReciprocal)
        throw OpenMMException("HIP platform: multi-GPU runs support NonbondedForce with PME, direct and reciprocal space in one force group");
    {
        // test hook, "rank:evaluation": that rank fails at that evaluation -- what the others make of a rank that never reaches a collective
        // (tests/hip/TestHipParallel.cpp, testFailingRank)
        const char* failAt = getenv("OPENMM_HIP_DEBUG_FAIL_RANK");
        int failRank = -1, failEvaluation = -1;
        if (failAt != NULL && sscanf(failAt, "%d:%d", &failRank, &failEvaluation) == 2 && failRank == hip.domain.rank && failEvaluation == (int) evaluationCount)
            throw OpenMMException("HIP platform: rank " + std::to_string(failRank) + " fails here (OPENMM_HIP_DEBUG_FAIL_RANK)");
    }
    const int ie = includeEnergy ? 1 : 0;
    nl.pbc = (hip.box[1] != 0.0 || hip.box[3] != 0.0 || hip.box[4] != 0.0) ? 2 : 1;
    nl.dd_mode = 1;
    nl.first_block = hip.ownSlot0 / OMMHIP_TILE; nl.owned_blocks = hip.slotsPerRank / OMMHIP_TILE;
    // double-precision positions of foreign atoms (atom order) are refreshed per step only if something reads them: term lists of
    // bonded forces (evaluated by every rank), 1-4 terms, exclusion partners in another integration unit.  A water box needs none
    // of it -- 64 B of scattered read-modify-write per foreign atom and step saved.
    nl.pos_wire = hip.posWire.ptr;
    nl.pos_scatter = (hip.foreignPositionsNeeded || exclusionsSpanUnits || num14 > 0) ? hip.pos.ptr : NULL;
    // halo mode: only this rank's slots and the sections its two neighbours send are current (HipContext::computeOrderDecomposed)
    nl.num_active_ranges = hip.haloMode ? hip.numActiveRanges : 0;
    for (int i = 0; i < 8; i++) nl.active_range[i] = hip.activeRange[i];
    nl.wire_ref = hip.haloMode ? hip.wireRef.ptr : NULL;
    nl.dd_guard_atom = hip.guardAtom.as<unsigned char>();
    nl.dd_warn = hip.ddWarnFraction(); nl.dd_max = hip.ddMaxFraction();
    nl.dd_flags = hip.haloMode ? hip.ddFlags.as<int>() : NULL;
    nl.dd_ranks = hip.domain.ranks; nl.dd_slots_per_rank = hip.slotsPerRank; nl.dd_trailer_slot = hip.trailerSlot;
    // half-shell: the pairs with the lower neighbour's section are evaluated here (forces on its atoms kept, returned in finishComputation)
    nl.dd_half_shell = hip.haloMode && hip.halfShell ? 1 : 0; nl.dd_eval_slot0 = hip.evalRange[0]; nl.dd_eval_slot1 = hip.evalRange[1];
    hip.pollDriftFlags();
    foldExclusions = numExclusionPairs > 0;
    checkDecomposedFlags();
    if (nl.max_chunks == 0) allocateNeighborList((int) (estimateChunks() * 1.4 / hip.domain.ranks) + 256);
    // test hook: at the n-th evaluation pretend the allocation holds only 8 % more than the list -- past the 7/8 at which nl_prepare
    // asks for a common re-sort (tests/test_multirank_cpu.py::test_nearly_full_list_triggers_a_common_resort_that_grows_it_on_emulator)
    const int debugTightAfter = getenv("OPENMM_HIP_DEBUG_TIGHT_LIST_AFTER") != NULL ? atoi(getenv("OPENMM_HIP_DEBUG_TIGHT_LIST_AFTER")) : 0;      // read per evaluation
    if (debugTightAfter > 0 && !debugShrinkDone && (int) evaluationCount == debugTightAfter && nl.max_chunks > 0) {
        HIP_CHECK(ommhip_memcpy_d2h(pinnedState, nlState.ptr, sizeof(int) * OMMHIP_NL_STATE_INTS, hip.stream));
        hip.sync();
        nl.max_chunks = std::min(nl.max_chunks, pinnedState[1] * 100 / 92 + 1);
        debugShrinkDone = true;
        if (getenv("OPENMM_HIP_DD_DEBUG") != NULL) fprintf(stderr, "HIP platform: rank %d: list of %d chunks, allocation set to %d\n", hip.domain.rank, pinnedState[1], nl.max_chunks);
    }
    if (stateCopyPending && (pinnedState[2] != 0 || pinnedState[1] > nl.max_chunks)) {
        // the ranks step in lockstep through the all-gather, so one of them cannot freeze and redo steps on its own: a list that
        // overflows on a decomposed run ends the run instead of producing wrong forces.  In halo mode it does not get that far: a list
        // that fills 7/8 of its allocation makes every rank re-sort at once (nl_prepare, the drift flags' channel), and the rebuild
        // after a re-sort is verified below and given 1.5 x its size.
        stringstream msg;
        msg << "HIP platform: the neighbour list of rank " << hip.domain.rank << " overflowed (" << pinnedState[1] << " chunks needed, " << nl.max_chunks
            << " allocated) on a multi-GPU run; the last steps are invalid";
        throw OpenMMException(msg.str());
    }
    // diagnostics (bench.py --rank-alone, frozen dynamics): a device-side rebuild every n-th evaluation, as the displacement check would ask for one
    static const int debugRebuildEvery = getenv("OPENMM_HIP_DEBUG_REBUILD_EVERY") != NULL ? atoi(getenv("OPENMM_HIP_DEBUG_REBUILD_EVERY")) : 0;
    if (debugRebuildEvery > 0 && !forceRebuild && evaluationCount > 0 && evaluationCount % debugRebuildEvery == 0) {
        static const int one = 1;
        HIP_CHECK(ommhip_memcpy_h2d(nlState.ptr, &one, sizeof(int), hip.stream));
    }
    while (true) {
        if (forceRebuild) {
            const int request[3] = {1, 0, 0};
            HIP_CHECK(ommhip_memcpy_h2d(nlState.ptr, request, sizeof(request), hip.stream));
        }
        const bool clear = hip.takePendingClear();
        HIP_CHECK(ommhip_nl_prepare(&nl, hip.pos.ptr, hip.wrap.ptr, clear ? hip.force.ptr : NULL, clear ? hip.force.bytes : 0,
                                    clear ? hip.extraClearPtr : NULL, clear ? hip.extraClearBytes : 0, hip.stream));
        if (!forceRebuild) break;            // the (device-decided) rebuild is queued below, after reciprocal space was forked
        HIP_CHECK(ommhip_nl_rebuild_if_requested(&nl, hip.stream));
        HIP_CHECK(ommhip_memcpy_d2h(pinnedState, nlState.ptr, sizeof(int) * OMMHIP_NL_STATE_INTS, hip.stream));
        hip.sync();
        if (pinnedState[2] == 0 && pinnedState[1] * 1.5 <= nl.max_chunks) { forceRebuild = false; break; }
        if (getenv("OPENMM_HIP_DD_DEBUG") != NULL) fprintf(stderr, "HIP platform: rank %d: list of %d chunks, allocation %d -> %d\n", hip.domain.rank, pinnedState[1], nl.max_chunks, (int) (pinnedState[1] * 1.6) + 64);
        allocateNeighborList((int) (pinnedState[1] * 1.6) + 64);
    }
    fillPmeStruct();
    // spreading looks at the slots this rank holds positions for (halo mode), not at every slot of the box
    pme.dd_num_active_ranges = hip.haloMode ? hip.numActiveRanges : 0;
    for (int i = 0; i < 8; i++) pme.dd_active_range[i] = hip.activeRange[i];
    const bool sideStream = hip.usePmeStream && hip.pmeComm != NULL;
    if (sideStream) {
        // Two streams: reciprocal space -- spreading, slab FFT with its two all-to-alls, halo planes, interpolation -- on the
        // side stream with a communicator of its own, the list rebuild and the pair kernel on the main stream; the latency of
        // the collectives hides behind the pair kernel.  finishComputation joins the streams before the forces are used.
        hip.forkPme();
        pme.comm = hip.pmeComm;
        pme.phases = OMMHIP_PME_ALL;
        HIP_CHECK(ommhip_pme_reciprocal_dd(&pme, posq.ptr, hip.paddedAtoms, hip.ownSlot0, hip.ownSlot1, blockCenter.ptr, blockHalf.ptr,
                                           hip.force.as<long long>(), hip.energyBuffer.as<double>(), HipContext::EnergySlots, ie, hip.pmeStream));
        hip.markPmeDone();
        HIP_CHECK(ommhip_nl_rebuild_if_requested(&nl, hip.stream));
        params.direct_grid = pairGridBesideSideStream();
        HIP_CHECK(ommhip_nb_direct(&nl, &params, sigEps.ptr, hip.force.as<long long>(), hip.energyBuffer.as<double>(), HipContext::EnergySlots, ie, hip.stream));
        params.direct_grid = directGridOverride > 0 ? directGridOverride : 0;
    }
    else {
        pme.comm = hip.domain.comm;
        HIP_CHECK(ommhip_nl_rebuild_if_requested(&nl, hip.stream));
        pme.phases = OMMHIP_PME_SPREAD_ONLY;
        HIP_CHECK(ommhip_pme_reciprocal_dd(&pme, posq.ptr, hip.paddedAtoms, hip.ownSlot0, hip.ownSlot1, blockCenter.ptr, blockHalf.ptr,
                                           hip.force.as<long long>(), hip.energyBuffer.as<double>(), HipContext::EnergySlots, ie, hip.stream));
        pme.phases = OMMHIP_PME_AFTER_SPREAD;
        HIP_CHECK(ommhip_nb_direct(&nl, &params, sigEps.ptr, hip.force.as<long long>(), hip.energyBuffer.as<double>(), HipContext::EnergySlots, ie, hip.stream));
        HIP_CHECK(ommhip_pme_reciprocal_dd(&pme, posq.ptr, hip.paddedAtoms, hip.ownSlot0, hip.ownSlot1, blockCenter.ptr, blockHalf.ptr,
                                           hip.force.as<long long>(), hip.energyBuffer.as<double>(), HipContext::EnergySlots, ie, hip.stream));
    }
    pme.phases = OMMHIP_PME_ALL;
    if ((++evaluationCount & 15) == 0) {
        HIP_CHECK(ommhip_memcpy_d2h(pinnedState, nlState.ptr, sizeof(int) * OMMHIP_NL_STATE_INTS, hip.stream));
        HIP_CHECK(ommhip_memcpy_d2h(pinnedDdError, ddError.ptr, sizeof(int), hip.stream));
        stateCopyPending = true;
    }
    // 1-4 exceptions: evaluated by every rank, energy from rank 0 (as the bonded terms, HipTermForce::execute)
    ommhip_term_batch t14 = {OMMHIP_TERM_EXCEPTION14, {num14, exceptionAtomsD.as<int>(), exceptionParamsD.as<double>()},
                             exceptionsArePeriodic ? 1 : 0, chargeD.as<double>(), ewaldAlpha};
    hip.addTerms(t14, includeEnergy && hip.countsTermEnergy());
    double energy = 0;
    // host-side constants: returned by every rank (the device energies are summed over the ranks in finishComputation)
    if (includeEnergy)
        energy += dispersionCoefficient / (hip.box[0] * hip.box[2] * hip.box[5]) + selfEnergy;
    return energy;
}

HipCalcNonbondedForceKernel::~HipCalcNonbondedForceKernel() {
    { std::lock_guard<std::mutex> lock(liveNonbondedMutex); liveNonbondedKernels.erase(std::remove(liveNonbondedKernels.begin(), liveNonbondedKernels.end(), this), liveNonbondedKernels.end()); }
    hip.removeListener(this);
    if (hip.extraClearPtr != NULL && hip.extraClearPtr == gridReal.ptr) { hip.extraClearPtr = NULL; hip.extraClearBytes = 0; }
    if (pinnedState != NULL) ommhip_host_free(pinnedState);
    if (pinnedDdError != NULL) ommhip_host_free(pinnedDdError);
}

int HipCalcNonbondedForceKernel::recoverFromOverflow() {
    if (nl.max_chunks == 0) return 0;
    hip.setAsCurrent();
    HIP_CHECK(ommhip_memcpy_d2h(pinnedState, nlState.ptr, sizeof(int) * OMMHIP_NL_STATE_INTS, hip.stream));
    hip.sync();
    stateCopyPending = false;
    if (pinnedState[OMMHIP_NL_STATE_OVERFLOW] == 0 && pinnedState[1] <= nl.max_chunks) return 0;
    const int skipped = pinnedState[OMMHIP_NL_STATE_FROZEN];
    hip.overflowRecoveries++;
    fprintf(stderr, "HIP platform: neighbour list overflowed (%d chunks needed, %d allocated); growing it and redoing %d step(s)\n", pinnedState[1], nl.max_chunks, skipped);
    allocateNeighborList((int) (std::max(pinnedState[1], nl.max_chunks) * 1.5) + 64);
    const int zero[OMMHIP_NL_STATE_INTS] = {1, 0, 0, 0, pinnedState[4], 0, 0, 0, 0, 0, 0, 0};          // rebuild requested, overflow and frozen counters cleared
    HIP_CHECK(ommhip_memcpy_h2d(nlState.ptr, zero, sizeof(zero), hip.stream));
    hip.sync();
    forceRebuild = true;
    debugShrinkDone = true;
    return skipped;
}

void HipCalcNonbondedForceKernel::atomsReordered() { slotParamsDirty = true; forceRebuild = true; }
void HipCalcNonbondedForceKernel::boxChanged() {
    etermDirty = true; dispersionEtermDirty = true; forceRebuild = true;
    if (hip.decomposed() && gridSize[0] > 0) {       // see setupPmeDecomposed
        hip.pmeReachX = 6.0 * hip.box[0] / gridSize[0]; hip.pmeReachBelow = 6.0 * hip.box[0] / gridSize[0]; hip.pmeReachAbove = 2.0 * hip.box[0] / gridSize[0];
    }
}
void HipCalcNonbondedForceKernel::positionsSet() { forceRebuild = true; }

/* The "transposed" order of HipTermForce::upload for a list of term ids: neighbouring 1-4s share atoms, the lanes of a wavefront
 * should not (their fixed-point atomics to one address would serialise). */
static void transposeTermOrder(vector<int>& ids) {
    static const bool transpose = getenv("OPENMM_HIP_TRANSPOSE_TERMS") != NULL && getenv("OPENMM_HIP_TRANSPOSE_TERMS")[0] == '1';
    if (!transpose) return;        // off by default: measured slower, see HipTermForce::upload
    const int n = (int) ids.size(), waves = (n + 63) / 64;
    vector<int> reordered;
    for (int w = 0; w < waves; w++)
        for (int l = 0; l < 64; l++)
            if (l * waves + w < n) reordered.push_back(ids[l * waves + w]);
    ids.swap(reordered);
}

static int findLegalFftDimension(int minimum, int multipleOf = 1) {
    // smallest size >= minimum that the LDS FFT handles (2,3,5,7-smooth); same role as CudaFFT3D::findLegalDimension.
    // Slab-decomposed runs also need the x and y sizes to be multiples of the number of ranks.
    int n = max(minimum, 2);
    while (!ommhip_fft_supported_size(n) || n % multipleOf != 0) n++;
    return n;
}

static vector<double> bsplineModuli(int n) {
    // ReferencePME.cpp:98-193 for order 5: B-spline values at the knots, then |DFT|^2 with the small-modulus fix-up.
    const int order = 5;
    double data[order] = {1, 0, 0, 0, 0};
    for (int k = 3; k < order; k++) {
        double div = 1.0 / (k - 1.0);
        data[k - 1] = 0;
        for (int l = 1; l < k - 1; l++) data[k - l - 1] = div * (l * data[k - l - 2] + (k - l) * data[k - l - 1]);
        data[0] = div * data[0];
    }
    double div = 1.0 / (order - 1);
    data[order - 1] = 0;
    for (int l = 1; l < order - 1; l++) data[order - l - 1] = div * (l * data[order - l - 2] + (order - l) * data[order - l - 1]);
    data[0] = div * data[0];
    vector<double> bs(max(n, order + 1), 0.0), mod(n);
    for (int i = 1; i <= order; i++) bs[i] = data[i - 1];
    for (int i = 0; i < n; i++) {
        double sc = 0, ss = 0;
        for (int j = 0; j < n; j++) {
            double arg = (2.0 * M_PI * i * j) / n;
            sc += bs[j] * cos(arg);
            ss += bs[j] * sin(arg);
        }
        mod[i] = sc * sc + ss * ss;
    }
    for (int i = 0; i < n; i++)
        if (mod[i] < 1.0e-7) mod[i] = (mod[(i - 1 + n) % n] + mod[(i + 1) % n]) / 2;
    return mod;
}

void HipCalcNonbondedForceKernel::initialize(const System& system, const NonbondedForce& force) {
    hip.setAsCurrent();
    numParticles = force.getNumParticles();
    if (numParticles != hip.numAtoms)
        throw OpenMMException("NonbondedForce must have exactly as many particles as the System it belongs to.");

    // ---- exceptions: every one is an exclusion; those with interactions are "1-4s" (ReferenceKernels.cpp:869-896)
    set<int> exceptionsWithOffsets;
    for (int i = 0; i < force.getNumExceptionParameterOffsets(); i++) {
        string param; int exception; double charge, sigma, epsilon;
        force.getExceptionParameterOffset(i, param, exception, charge, sigma, epsilon);
        exceptionsWithOffsets.insert(exception);
    }
    vector<set<int> > exclusions(numParticles);
    vector<int> nb14s;
    map<int, int> nb14Index;
    vector<int> exclusionPairs;
    for (int i = 0; i < force.getNumExceptions(); i++) {
        int p1, p2; double chargeProd, sigma, epsilon;
        force.getExceptionParameters(i, p1, p2, chargeProd, sigma, epsilon);
        if (exclusions[p1].insert(p2).second) { exclusionPairs.push_back(min(p1, p2)); exclusionPairs.push_back(max(p1, p2)); }
        exclusions[p2].insert(p1);
        if (chargeProd != 0.0 || epsilon != 0.0 || exceptionsWithOffsets.find(i) != exceptionsWithOffsets.end()) {
            nb14Index[i] = (int) nb14s.size();
            nb14s.push_back(i);
        }
    }
    transposeTermOrder(nb14s);
    for (size_t i = 0; i < nb14s.size(); i++) nb14Index[nb14s[i]] = (int) i;
    num14 = (int) nb14s.size();
    numExclusionPairs = (int) exclusionPairs.size() / 2;
    exclusionsSpanUnits = false;
    for (size_t i = 0; i + 1 < exclusionPairs.size() && !hip.unitOfAtom.empty(); i += 2)
        if (hip.unitOfAtom[exclusionPairs[i]] != hip.unitOfAtom[exclusionPairs[i + 1]]) { exclusionsSpanUnits = true; break; }
    baseParticleParams.assign(numParticles, vector<double>(3));
    baseExceptionParams.assign(num14, vector<double>(3));
    exceptionAtoms.resize(num14);
    for (int i = 0; i < numParticles; i++)
        force.getParticleParameters(i, baseParticleParams[i][0], baseParticleParams[i][1], baseParticleParams[i][2]);
    vector<int> exceptionAtomsFlat(2 * (size_t) num14);
    for (int i = 0; i < num14; i++) {
        int p1, p2;
        force.getExceptionParameters(nb14s[i], p1, p2, baseExceptionParams[i][0], baseExceptionParams[i][1], baseExceptionParams[i][2]);
        exceptionAtoms[i] = make_pair(p1, p2);
        exceptionAtomsFlat[2 * i] = p1; exceptionAtomsFlat[2 * i + 1] = p2;
    }
    for (int i = 0; i < force.getNumParticleParameterOffsets(); i++) {
        string param; int particle; double charge, sigma, epsilon;
        force.getParticleParameterOffset(i, param, particle, charge, sigma, epsilon);
        particleParamOffsets[make_pair(param, particle)] = {charge, sigma, epsilon};
    }
    for (int i = 0; i < force.getNumExceptionParameterOffsets(); i++) {
        string param; int exception; double charge, sigma, epsilon;
        force.getExceptionParameterOffset(i, param, exception, charge, sigma, epsilon);
        exceptionParamOffsets[make_pair(param, nb14Index[exception])] = {charge, sigma, epsilon};
    }

    // ---- method and derived constants (ReferenceKernels.cpp:926-964)
    nonbondedMethod = CalcNonbondedForceKernel::NonbondedMethod(force.getNonbondedMethod());
    nonbondedCutoff = force.getCutoffDistance();
    useSwitchingFunction = nonbondedMethod != NoCutoff && force.getUseSwitchingFunction();
    switchingDistance = force.getSwitchingDistance();
    rfDielectric = force.getReactionFieldDielectric();
    ewaldAlpha = 0;
    kmax[0] = kmax[1] = kmax[2] = 0;
    gridSize[0] = gridSize[1] = gridSize[2] = 0;
    if (nonbondedMethod == LJPME) {
        // ReferenceKernels.cpp:948-955: two sets of PME parameters, no switching function
        if (hip.decomposed()) throw OpenMMException("HIP platform: LJPME is not available on multi-GPU runs");
        NonbondedForceImpl::calcPMEParameters(system, force, ewaldAlpha, gridSize[0], gridSize[1], gridSize[2], false);
        NonbondedForceImpl::calcPMEParameters(system, force, dispersionAlpha, dispersionGridSize[0], dispersionGridSize[1], dispersionGridSize[2], true);
        for (int k = 0; k < 3; k++) { gridSize[k] = findLegalFftDimension(gridSize[k]); dispersionGridSize[k] = findLegalFftDimension(dispersionGridSize[k]); }
        useSwitchingFunction = false;
    }
    if (nonbondedMethod == Ewald) {
        NonbondedForceImpl::calcEwaldParameters(system, force, ewaldAlpha, kmax[0], kmax[1], kmax[2]);
    }
    else if (nonbondedMethod == PME) {
        NonbondedForceImpl::calcPMEParameters(system, force, ewaldAlpha, gridSize[0], gridSize[1], gridSize[2], false);
        for (int k = 0; k < 3; k++) gridSize[k] = findLegalFftDimension(gridSize[k], k < 2 ? hip.domain.ranks : 1);
    }
    usesPeriodic = nonbondedMethod == CutoffPeriodic || nonbondedMethod == Ewald || nonbondedMethod == PME || nonbondedMethod == LJPME;
    exceptionsArePeriodic = usesPeriodic && force.getExceptionsUsePeriodicBoundaryConditions();
    if (force.getUseDispersionCorrection() && usesPeriodic)
        dispersionCoefficient = NonbondedForceImpl::calcDispersionCorrection(system, force);
    else
        dispersionCoefficient = 0.0;
    if (usesPeriodic) hip.usePeriodic = true;
    if (nonbondedMethod != NoCutoff) hip.sortCutoff = max(hip.sortCutoff, nonbondedCutoff);
    {
        // atoms without Lennard-Jones parameters (water hydrogens: two thirds of a solvated system) go to the end of their
        // 32-slot block, so that the pair kernel can leave the LJ arithmetic out for the tail of such blocks (nonbonded.hip,
        // OMM_LJ_HEAD).  Only an ordering hint: the kernel looks at the parameters themselves.
        vector<char> noLJ(numParticles);
        for (int i = 0; i < numParticles; i++) noLJ[i] = baseParticleParams[i][2] == 0.0 ? 1 : 0;
        hip.setBlockTailAtoms(noLJ);
    }
    hip.requestReorder();
    // List padding as a fraction of the cutoff.  Small systems do not fill the chip: the pair launches are bound by the latency
    // of a chunk, extra rows ride along for free (DHFR size: 7.3 k -> 8.7 k rows, same 59 us) while every rebuild avoided
    // saves 50 us -- 0.2 measured best there (1433 vs 1388 ns/day; 0.25 pushes the list past what two rounds of wavefronts
    // cover and loses again).  At a million atoms the pair kernel is throughput-bound (+16 % time for +20 % rows), but an atom
    // somewhere crosses half the padding every 2-3 steps and a rebuild costs two pair-kernel launches: 0.1 / 0.15 / 0.2 gave
    // 4.20 / 3.84 / 3.67 ms per step on one MI355X (profiles/r02c), so 0.2 it is at every size until the rebuild gets cheaper.
    // Round 3, with the builder of today (resident workgroups, cell-sorted candidates) and a liquid at 300 K instead of the melting
    // lattice those figures were taken on: 985 527 atoms 2.193 / 2.164 / 2.199 ms per step at 0.1 / 0.15 / 0.2, 92 224 atoms
    // 0.284 / 0.286 / 0.289, DHFR (fused launches, where rows ride along) 0.1188 (0.12) / 0.1184 (0.16) / 0.1180 (0.2): 0.15 above
    // the size of the fused front launch, 0.2 below (same-box A/B, profiles/r06a_ab_list_padding.txt).
    const int fusedFrontMaxAtoms = getenv("OPENMM_HIP_FUSED_FRONT_MAX_ATOMS") != NULL ? atoi(getenv("OPENMM_HIP_FUSED_FRONT_MAX_ATOMS")) : 60000;
    double paddingFraction = numParticles > fusedFrontMaxAtoms ? 0.15 : 0.2;
    if (getenv("OPENMM_HIP_NL_PADDING") != NULL) paddingFraction = atof(getenv("OPENMM_HIP_NL_PADDING"));   // tuning knob, fraction of the cutoff
    padding = nonbondedMethod == NoCutoff ? 0.0 : paddingFraction * nonbondedCutoff;
    if (getenv("OPENMM_HIP_DIRECT_GRID") != NULL) directGridOverride = atoi(getenv("OPENMM_HIP_DIRECT_GRID"));
    // decomposed runs: how far a rank must see beyond its slab (halo mode, HipContext::computeOrderDecomposed)
    if (nonbondedMethod != NoCutoff) hip.haloReach = max(hip.haloReach, nonbondedCutoff + padding);

    // ---- device arrays
    const int P = hip.paddedAtoms;
    chargeD.allocate(sizeof(double) * max(numParticles, 1));
    sigmaD.allocate(sizeof(double) * max(numParticles, 1));
    epsilonD.allocate(sizeof(double) * max(numParticles, 1));
    posq.allocate(sizeof(float) * 4 * P);
    posqRef.allocate(sizeof(float) * 4 * P);
    posqRel.allocate(sizeof(float) * 4 * P);
    posqRelLo.allocate(sizeof(float) * 4 * P);
    HIP_CHECK(ommhip_memset(posqRelLo.ptr, 0, posqRelLo.bytes, hip.stream));
    sigEps.allocate(sizeof(float) * 2 * P);
    HIP_CHECK(ommhip_memset(posq.ptr, 0, posq.bytes, hip.stream));
    HIP_CHECK(ommhip_memset(posqRef.ptr, 0, posqRef.bytes, hip.stream));
    HIP_CHECK(ommhip_memset(posqRel.ptr, 0, posqRel.bytes, hip.stream));
    vector<int> start(numParticles + 1, 0), flat;
    for (int i = 0; i < numParticles; i++) {
        for (set<int>::const_iterator it = exclusions[i].begin(); it != exclusions[i].end(); ++it) flat.push_back(*it);
        start[i + 1] = (int) flat.size();
    }
    uploadVector(exclStart, start, hip.stream);
    uploadVector(exclAtoms, flat, hip.stream);
    hostExclStart = start;
    hostExclAtoms = flat;
    exclBlockRange.allocate(sizeof(int) * 2 * (P / OMMHIP_TILE));
    uploadVector(exceptionAtomsD, exceptionAtomsFlat, hip.stream);
    exceptionParamsD.allocate(sizeof(double) * 3 * max(num14, 1));
    uploadVector(exclusionPairsD, exclusionPairs, hip.stream);
    nlState.allocate(sizeof(int) * OMMHIP_NL_STATE_INTS);
    HIP_CHECK(ommhip_memset(nlState.ptr, 0, nlState.bytes, hip.stream));
    blockCenter.allocate(sizeof(float) * 4 * (P / OMMHIP_TILE));
    blockHalf.allocate(sizeof(float) * 4 * (P / OMMHIP_TILE));
    HIP_CHECK(ommhip_host_malloc((void**) &pinnedState, sizeof(int) * OMMHIP_NL_STATE_INTS));
    memset(pinnedState, 0, sizeof(int) * OMMHIP_NL_STATE_INTS);

    nl.num_atoms = numParticles; nl.padded_atoms = P;
    nl.pbc = 0;
    nl.cutoff = nonbondedMethod == NoCutoff ? 0.0 : nonbondedCutoff;
    nl.padding = padding;
    nl.posq = posq.ptr; nl.posq_ref = posqRef.ptr; nl.posq_rel = posqRel.ptr; nl.posq_rel_lo = posqRelLo.ptr;
    nl.atom_of_slot = hip.atomOfSlot.as<int>(); nl.slot_of_atom = hip.slotOfAtom.as<int>();
    nl.excl_start = exclStart.as<int>(); nl.excl_atoms = exclAtoms.as<int>();
    nl.state = nlState.as<int>();
    nl.block_center = blockCenter.ptr; nl.block_half = blockHalf.ptr;
    // scratch of the cell-binned candidate search (used by the builder on large rectangular systems only)
    nl.max_cells = P / OMMHIP_TILE + 64;
    cellStart.allocate(sizeof(int) * (2 * (size_t) nl.max_cells + 2));
    cellBlocks.allocate(sizeof(int) * 2 * (size_t) (P / OMMHIP_TILE));
    cellBoxes.allocate(sizeof(float) * 8 * (size_t) (P / OMMHIP_TILE));
    cellMeta.allocate(sizeof(float) * 4);
    nl.cell_start = cellStart.as<int>(); nl.cell_blocks = cellBlocks.as<int>(); nl.cell_boxes = cellBoxes.ptr; nl.cell_meta = cellMeta.as<float>();
    nl.cell_min_blocks = getenv("OPENMM_HIP_NL_CELL_MIN_BLOCKS") != NULL ? atoi(getenv("OPENMM_HIP_NL_CELL_MIN_BLOCKS")) : 0;   // 0 = default
    nl.max_chunks = 0;
    if (!hip.decomposed() && !hip.hostMode) {
        // single-GPU device mode: the integration kernels freeze while this list's overflow word is set, and the skipped
        // steps are replayed after recoverFromOverflow() has grown the list (decomposed runs treat an overflow as an error)
        hip.freezeState = nlState.as<int>();
        hip.listRecovery = [this]() { return recoverFromOverflow(); };
        hip.listOverflowSeen = [this]() { return stateCopyPending && (pinnedState[OMMHIP_NL_STATE_OVERFLOW] != 0 || pinnedState[1] > nl.max_chunks); };
    }

    params.ewald = (nonbondedMethod == Ewald || nonbondedMethod == PME || nonbondedMethod == LJPME) ? 1 : 0;
    params.ljpme = nonbondedMethod == LJPME ? 1 : 0;
    params.dispersion_alpha = dispersionAlpha;
    params.use_switch = useSwitchingFunction ? 1 : 0;
    params.ewald_alpha = ewaldAlpha;
    params.krf = params.crf = 0;
    if (nonbondedMethod == CutoffNonPeriodic || nonbondedMethod == CutoffPeriodic) {
        // ReferenceLJCoulombIxn.cpp:78-79
        params.krf = pow(nonbondedCutoff, -3.0) * (rfDielectric - 1.0) / (2.0 * rfDielectric + 1.0);
        params.crf = (1.0 / nonbondedCutoff) * (3.0 * rfDielectric) / (2.0 * rfDielectric + 1.0);
    }
    params.switch_distance = switchingDistance;
    params.direct_grid = directGridOverride > 0 ? directGridOverride : 0;
    if (nonbondedMethod == PME) { if (hip.decomposed()) setupPmeDecomposed(); else setupPme(); }
    if (nonbondedMethod == LJPME) { setupPme(); setupDispersionPme(); }
    if (nonbondedMethod == Ewald)
        ewaldStructure.allocate(sizeof(double) * 2 * (size_t) kmax[0] * (2 * kmax[1] - 1) * (2 * kmax[2] - 1));
    hip.sync();
}

int HipCalcNonbondedForceKernel::pairGridBesideSideStream() {
    // Pair kernel while reciprocal space runs on the side stream: with a long list, 2 wavefronts per SIMD that walk through the
    // chunks (8 per CU x 4 SIMDs... = 8 x CUs workgroups of one wavefront) instead of one short-lived wavefront per chunk -- the
    // small launches of the side stream then find a chip in a steady state instead of queueing behind a dispatch backlog of
    // 10^5 workgroups.  Same-box A/B on one GPU (profiles/r03h_ab_persistent_grid_two_streams.txt): 2.69 / 2.65 -> 2.60 ms per step
    // at 985 k atoms, 0.321 / 0.312 -> 0.309 / 0.311 at 92 k; fewer wavefronts than that (slots left free on purpose) lose.
    // Decomposed runs take the same setting (same mechanism, not measured there).
    if (directGridOverride != 0) return directGridOverride > 0 ? directGridOverride : 0;       // OPENMM_HIP_DIRECT_GRID=-1: one wavefront per chunk everywhere (A/B)
    static int resident = 0;
    if (resident == 0) {
        int cus = 0;
        if (ommhip_device_info(hip.getDeviceIndex(), NULL, 0, &cus, NULL) != 0 || cus <= 0) cus = 256;
        resident = 8 * cus;
    }
    return nl.max_chunks >= 4 * resident ? resident : 0;
}

bool HipCalcNonbondedForceKernel::enableTileSpread(int nx, int ny, int nz) {
    // Opt-in (OPENMM_HIP_TILE_SPREAD_MIN_ATOMS=n): spread the charges by grid tiles (ommhip_pme::spread_mode 2) -- one workgroup
    // per 16^3 cells gathers what lands in it, no global atomics and no grid to clear.  Correct and tested, but measured SLOWER
    // than the brick kernel in this form ("tried and not kept": 446 vs 50 us at 92 k atoms, 837 vs 433 us at 1M), so
    // it is off unless asked for.  The kernel library falls back to the brick kernel by itself when the evaluation does not
    // qualify (triclinic box, fewer than 32 cells on an axis).
    const int minAtoms = getenv("OPENMM_HIP_TILE_SPREAD_MIN_ATOMS") != NULL ? atoi(getenv("OPENMM_HIP_TILE_SPREAD_MIN_ATOMS")) : 2147483647;
    if (numParticles < minAtoms || hip.deterministicForces || getenv("OPENMM_HIP_PME_SPREAD_DIRECT") != NULL || min(nx, min(ny, nz)) < 32) return false;
    const int tiles = ((nx + 15) / 16) * ((ny + 15) / 16) * ((nz + 15) / 16);
    const int cap = 384;        // blocks per tile list; water: ~110
    tileCount.allocate(sizeof(int) * (size_t) tiles);
    tileBlocks.allocate(sizeof(int) * (size_t) tiles * cap);
    pme.spread_mode = 2; pme.tile_count = tileCount.as<int>(); pme.tile_blocks = tileBlocks.as<int>(); pme.tile_cap = cap; pme.max_tiles = tiles;
    return true;
}

void HipCalcNonbondedForceKernel::setupPme() {
    const int nx = gridSize[0], ny = gridSize[1], nz = gridSize[2], nzc = nz / 2 + 1;
    uploadVector(moduliX, bsplineModuli(nx), hip.stream);
    uploadVector(moduliY, bsplineModuli(ny), hip.stream);
    uploadVector(moduliZ, bsplineModuli(nz), hip.stream);
    DeviceBuffer* tw[3] = {&twiddleX, &twiddleY, &twiddleZ};
    for (int d = 0; d < 3; d++) {
        const int n = gridSize[d];
        vector<float> t(2 * (size_t) n);
        for (int k = 0; k < n; k++) { t[2 * k] = (float) cos(2.0 * M_PI * k / n); t[2 * k + 1] = (float) -sin(2.0 * M_PI * k / n); }
        uploadVector(*tw[d], t, hip.stream);
    }
    eterm.allocate(sizeof(float) * (size_t) nx * ny * nzc);
    const size_t gridBytes = (sizeof(float) * (size_t) nx * ny * nz + 15) / 16 * 16;
    gridReal.allocate(gridBytes);
    pme.spread_mode = getenv("OPENMM_HIP_PME_SPREAD_DIRECT") != NULL ? 1 : 0;    // A/B knob: direct global atomics
    if (!enableTileSpread(nx, ny, nz) && hip.extraClearPtr == NULL) {
        // the context zeroes this grid together with the force buffer at the start of every evaluation
        hip.extraClearPtr = gridReal.ptr;
        hip.extraClearBytes = gridBytes;
        pme.grid_precleared = 1;
    }
    gridComplex.allocate(sizeof(float) * 2 * (size_t) nx * ny * nzc);
    pme.nx = nx; pme.ny = ny; pme.nz = nz; pme.alpha = ewaldAlpha;
    pme.moduli_x = moduliX.as<double>(); pme.moduli_y = moduliY.as<double>(); pme.moduli_z = moduliZ.as<double>();
    pme.eterm = eterm.ptr; pme.grid_real = gridReal.ptr; pme.grid_complex = gridComplex.ptr;
    pme.twiddle_x = twiddleX.ptr; pme.twiddle_y = twiddleY.ptr; pme.twiddle_z = twiddleZ.ptr;
    etermDirty = true;
}

void HipCalcNonbondedForceKernel::setupDispersionPme() {
    // the second grid of LJPME: same kernels, C6 factors instead of charges, the influence function of ReferencePME.cpp:518-614
    const int nx = dispersionGridSize[0], ny = dispersionGridSize[1], nz = dispersionGridSize[2], nzc = nz / 2 + 1;
    uploadVector(dModuliX, bsplineModuli(nx), hip.stream);
    uploadVector(dModuliY, bsplineModuli(ny), hip.stream);
    uploadVector(dModuliZ, bsplineModuli(nz), hip.stream);
    DeviceBuffer* tw[3] = {&dTwiddleX, &dTwiddleY, &dTwiddleZ};
    for (int d = 0; d < 3; d++) {
        const int n = dispersionGridSize[d];
        vector<float> t(2 * (size_t) n);
        for (int k = 0; k < n; k++) { t[2 * k] = (float) cos(2.0 * M_PI * k / n); t[2 * k + 1] = (float) -sin(2.0 * M_PI * k / n); }
        uploadVector(*tw[d], t, hip.stream);
    }
    dEterm.allocate(sizeof(float) * (size_t) nx * ny * nzc);
    dGridReal.allocate((sizeof(float) * (size_t) nx * ny * nz + 15) / 16 * 16);
    dGridComplex.allocate(sizeof(float) * 2 * (size_t) nx * ny * nzc);
    c6D.allocate(sizeof(double) * max(numParticles, 1));
    posqDisp.allocate(sizeof(float) * 4 * hip.paddedAtoms);
    pmeDisp.nx = nx; pmeDisp.ny = ny; pmeDisp.nz = nz; pmeDisp.alpha = dispersionAlpha;
    pmeDisp.moduli_x = dModuliX.as<double>(); pmeDisp.moduli_y = dModuliY.as<double>(); pmeDisp.moduli_z = dModuliZ.as<double>();
    pmeDisp.eterm = dEterm.ptr; pmeDisp.grid_real = dGridReal.ptr; pmeDisp.grid_complex = dGridComplex.ptr;
    pmeDisp.twiddle_x = dTwiddleX.ptr; pmeDisp.twiddle_y = dTwiddleY.ptr; pmeDisp.twiddle_z = dTwiddleZ.ptr;
    pmeDisp.dispersion = 1;
    dispersionEtermDirty = true;
}

void HipCalcNonbondedForceKernel::updateExclusionBlockRanges() {
    // For every i-block: the lowest/highest block that holds an exclusion partner of one of its atoms; and the exclusion CSR
    // keyed by slot with partners as slots (the builder then needs one gather per partner instead of three).  Runs after every
    // re-sort; block-parallel on the host (a million atoms take tens of milliseconds on one thread).
    const int numBlocks = hip.paddedAtoms / OMMHIP_TILE;
    vector<int> range(2 * (size_t) numBlocks);
    vector<int> slotStart(hip.paddedAtoms + 1, 0), slots(max((size_t) 1, hostExclAtoms.size()));
    for (int s = 0; s < hip.paddedAtoms; s++) {
        const int atom = hip.hostAtomOfSlot[s];
        slotStart[s + 1] = slotStart[s] + (atom >= 0 ? hostExclStart[atom + 1] - hostExclStart[atom] : 0);
    }
    auto body = [&](int begin, int end) {
        for (int b = begin; b < end; b++) {
            int lo = numBlocks, hi = -1;
            for (int s = b * OMMHIP_TILE; s < (b + 1) * OMMHIP_TILE; s++) {
                const int atom = hip.hostAtomOfSlot[s];
                if (atom < 0) continue;
                int n = slotStart[s];
                for (int e = hostExclStart[atom]; e < hostExclStart[atom + 1]; e++) {
                    const int partner = hip.hostSlotOfAtom[hostExclAtoms[e]];
                    slots[n++] = partner;
                    lo = min(lo, partner / OMMHIP_TILE);
                    hi = max(hi, partner / OMMHIP_TILE);
                }
            }
            range[2 * b] = lo; range[2 * b + 1] = hi;
        }
    };
    const int threads = numBlocks >= 4096 ? max(1, min(16, (int) std::thread::hardware_concurrency() / max(1, hip.domain.ranks))) : 1;
    if (threads <= 1) body(0, numBlocks);
    else {
        vector<std::thread> pool;
        const int chunk = (numBlocks + threads - 1) / threads;
        for (int t = 0; t < threads && t * chunk < numBlocks; t++) pool.push_back(std::thread(body, t * chunk, min(numBlocks, (t + 1) * chunk)));
        for (size_t t = 0; t < pool.size(); t++) pool[t].join();
    }
    HIP_CHECK(ommhip_memcpy_h2d(exclBlockRange.ptr, range.data(), sizeof(int) * range.size(), hip.stream));
    exclSlotStart.allocate(sizeof(int) * slotStart.size());
    exclSlots.allocate(sizeof(int) * slots.size());
    HIP_CHECK(ommhip_memcpy_h2d(exclSlotStart.ptr, slotStart.data(), sizeof(int) * slotStart.size(), hip.stream));
    HIP_CHECK(ommhip_memcpy_h2d(exclSlots.ptr, slots.data(), sizeof(int) * slots.size(), hip.stream));
    hip.sync();
    nl.excl_block_range = exclBlockRange.ptr;
    nl.excl_slot_start = exclSlotStart.as<int>(); nl.excl_slots = exclSlots.as<int>();
}

double HipCalcNonbondedForceKernel::innerPaddingFraction() {
    // padding of the pruned list as a fraction of the cutoff (tuning knob OPENMM_HIP_NL_INNER_PADDING)
    static const double f = getenv("OPENMM_HIP_NL_INNER_PADDING") != NULL ? atof(getenv("OPENMM_HIP_NL_INNER_PADDING")) : 0.06;
    return f;
}

int HipCalcNonbondedForceKernel::estimateChunks() const {
    const int numBlocks = hip.paddedAtoms / OMMHIP_TILE;
    double rows;
    if (nonbondedMethod == NoCutoff || !usesPeriodic)
        rows = 0.5 * (double) numBlocks * hip.paddedAtoms / OMMHIP_ROW + numBlocks;     // all pairs (upper bound for non-periodic cutoffs)
    else {
        const double volume = hip.box[0] * hip.box[2] * hip.box[5];
        const double density = numParticles / volume;
        const double a = pow(OMMHIP_TILE / density, 1.0 / 3.0), r = nonbondedCutoff + padding;
        const double minkowski = a * a * a + 6 * a * a * r + 3 * M_PI * a * r * r + 4.0 / 3.0 * M_PI * r * r * r;
        rows = numBlocks * (0.5 * min((double) hip.paddedAtoms, density * minkowski) / OMMHIP_ROW + 1.0);
    }
    double chunks = rows / OMMHIP_CHUNK_ROWS + numBlocks;
    return (int) min(chunks * 1.3 + 64, 2.0e8);
}

void HipCalcNonbondedForceKernel::allocateNeighborList(int maxChunks) {
    chunkInfo.allocate(sizeof(int) * 2 * (size_t) maxChunks);
    rowJ.allocate(sizeof(int) * (size_t) maxChunks * OMMHIP_CHUNK_ROWS * OMMHIP_ROW);
    rowMask.allocate(sizeof(unsigned) * (size_t) maxChunks * OMMHIP_CHUNK_ROWS * OMMHIP_ROW);
    nl.max_chunks = maxChunks;
    nl.chunk_info = chunkInfo.ptr; nl.row_j = rowJ.as<int>(); nl.row_mask = rowMask.as<unsigned>();
    // The pruned ("inner") list of include/openmm_hip_kernels.h (chunk_info_inner; same capacity -- it never holds more than the list
    // it is cut from): OFF unless OPENMM_HIP_PRUNE=1.  Measured on one MI355X (docs/EXPERIMENTS.md, round 3): it removes 25-30 % of the
    // rows the pair kernel walks (4.9 -> 3.6 evaluations per pair inside the cutoff), but the pair kernel gets only 12-20 % faster
    // (the rows that go are the ones it skipped through quickest) and a cut costs 0.6 ms at a million atoms every 2-4 steps: 2.25 ms
    // per step at best against 2.18 without.  At DHFR size the pair work rides on the FFT launches and 30 % fewer rows shorten them by 4 %.
    bool prune = false;
    if (getenv("OPENMM_HIP_PRUNE") != NULL) prune = atoi(getenv("OPENMM_HIP_PRUNE")) != 0;
    if (nonbondedMethod != NoCutoff && prune) {
        chunkInfoInner.allocate(chunkInfo.bytes);
        rowJInner.allocate(rowJ.bytes);
        rowMaskInner.allocate(rowMask.bytes);
        if (blockRuns.ptr == NULL) {
            blockRuns.allocate(sizeof(int) * 17 * (size_t) (hip.paddedAtoms / OMMHIP_TILE));
            HIP_CHECK(ommhip_memset(blockRuns.ptr, 0, blockRuns.bytes, hip.stream));
            posqRefInner.allocate(sizeof(float) * 4 * (size_t) hip.paddedAtoms);
            HIP_CHECK(ommhip_memset(posqRefInner.ptr, 0, posqRefInner.bytes, hip.stream));
        }
        nl.chunk_info_inner = chunkInfoInner.ptr; nl.row_j_inner = rowJInner.as<int>(); nl.row_mask_inner = rowMaskInner.as<unsigned>(); nl.block_runs = blockRuns.as<int>();
        nl.posq_ref_inner = posqRefInner.ptr;
        nl.inner_padding = innerPaddingFraction() * nonbondedCutoff;
    }
}

void HipCalcNonbondedForceKernel::computeParameters(ContextImpl& context, bool forceUpdate) {
    // ReferenceKernels.cpp:1077-1121; re-evaluated only when a global parameter with an offset changed.
    bool changed = forceUpdate || !hasInitializedParams;
    for (map<pair<string, int>, vector<double> >::const_iterator it = particleParamOffsets.begin(); it != particleParamOffsets.end(); ++it) {
        double v = context.getParameter(it->first.first);
        map<string, double>::iterator last = lastGlobalValues.find(it->first.first);
        if (last == lastGlobalValues.end() || last->second != v) { changed = true; lastGlobalValues[it->first.first] = v; }
    }
    for (map<pair<string, int>, vector<double> >::const_iterator it = exceptionParamOffsets.begin(); it != exceptionParamOffsets.end(); ++it) {
        double v = context.getParameter(it->first.first);
        map<string, double>::iterator last = lastGlobalValues.find(it->first.first);
        if (last == lastGlobalValues.end() || last->second != v) { changed = true; lastGlobalValues[it->first.first] = v; }
    }
    if (!changed) return;
    hasInitializedParams = true;
    vector<double> sigmas(numParticles), epsilons(numParticles);
    charges.resize(numParticles);
    maxChargeDirty = true;
    for (int i = 0; i < numParticles; i++) { charges[i] = baseParticleParams[i][0]; sigmas[i] = baseParticleParams[i][1]; epsilons[i] = baseParticleParams[i][2]; }
    for (map<pair<string, int>, vector<double> >::const_iterator it = particleParamOffsets.begin(); it != particleParamOffsets.end(); ++it) {
        double value = lastGlobalValues[it->first.first];
        int index = it->first.second;
        charges[index] += value * it->second[0]; sigmas[index] += value * it->second[1]; epsilons[index] += value * it->second[2];
    }
    vector<double> ex(3 * (size_t) num14);
    for (int i = 0; i < num14; i++) { ex[3 * i] = baseExceptionParams[i][0]; ex[3 * i + 1] = baseExceptionParams[i][1]; ex[3 * i + 2] = baseExceptionParams[i][2]; }
    for (map<pair<string, int>, vector<double> >::const_iterator it = exceptionParamOffsets.begin(); it != exceptionParamOffsets.end(); ++it) {
        double value = lastGlobalValues[it->first.first];
        int index = it->first.second;
        ex[3 * index] += value * it->second[0]; ex[3 * index + 1] += value * it->second[1]; ex[3 * index + 2] += value * it->second[2];
    }
    if (numParticles > 0) {
        HIP_CHECK(ommhip_memcpy_h2d(chargeD.ptr, charges.data(), sizeof(double) * numParticles, hip.stream));
        HIP_CHECK(ommhip_memcpy_h2d(sigmaD.ptr, sigmas.data(), sizeof(double) * numParticles, hip.stream));
        HIP_CHECK(ommhip_memcpy_h2d(epsilonD.ptr, epsilons.data(), sizeof(double) * numParticles, hip.stream));
    }
    if (num14 > 0)
        HIP_CHECK(ommhip_memcpy_h2d(exceptionParamsD.ptr, ex.data(), sizeof(double) * ex.size(), hip.stream));
    hip.sync();
    // Ewald self energy (ReferenceLJCoulombIxn.cpp:220-233)
    selfEnergy = 0;
    if (nonbondedMethod == Ewald || nonbondedMethod == PME || nonbondedMethod == LJPME)
        for (int i = 0; i < numParticles; i++) selfEnergy -= ONE_4PI_EPS0 * charges[i] * charges[i] * ewaldAlpha / sqrt(M_PI);
    if (nonbondedMethod == LJPME) {
        // per-atom C6 factors 8 (sigma/2)^3 (2 sqrt(eps)) and the dispersion self term (ReferenceLJCoulombIxn.cpp:224-227,257)
        vector<double> c6(numParticles);
        for (int i = 0; i < numParticles; i++) {
            const double s = 0.5 * sigmas[i], e = 2.0 * sqrt(epsilons[i]);
            c6[i] = 8.0 * s * s * s * e;
            selfEnergy += pow(dispersionAlpha, 6.0) * c6[i] * c6[i] / 12.0;
        }
        HIP_CHECK(ommhip_memcpy_h2d(c6D.ptr, c6.data(), sizeof(double) * numParticles, hip.stream));
        hip.sync();
    }
    slotParamsDirty = true;
}

void HipCalcNonbondedForceKernel::fillPmeStruct() {
    for (int i = 0; i < 6; i++) pme.box[i] = hip.box[i];
    pme.excl_start = foldExclusions ? exclStart.as<int>() : NULL;
    pme.excl_atoms = exclAtoms.as<int>(); pme.atom_of_slot = hip.atomOfSlot.as<int>();
    pme.pos = hip.pos.ptr; pme.charge = chargeD.as<double>(); pme.excl_periodic = exceptionsArePeriodic ? 1 : 0;
    pme.deterministic = hip.deterministicForces ? 1 : 0;
    pme.block_center = blockCenter.ptr; pme.block_half = blockHalf.ptr;
    if (maxChargeDirty) {          // fixed-point scales of the deterministic grid and of the tile spreading
        maxCharge = 0;
        for (size_t i = 0; i < charges.size(); i++) maxCharge = max(maxCharge, fabs(charges[i]));
        maxChargeDirty = false;
    }
    pme.max_charge = maxCharge;
    if (etermDirty) {
        HIP_CHECK(ommhip_pme_build_eterm(&pme, hip.stream));
        etermDirty = false;
    }
}

void HipCalcNonbondedForceKernel::launchPme(int includeEnergy, bool spreadDone, bool fftDone) {
    fillPmeStruct();
    if (spreadDone) {
        // the charges were spread by the fused front launch of this evaluation (and the FFTs may have travelled with the pair kernel)
        pme.phases = fftDone ? OMMHIP_PME_INTERPOLATE_ONLY : OMMHIP_PME_AFTER_SPREAD;
        HIP_CHECK(ommhip_pme_reciprocal(&pme, posq.ptr, hip.paddedAtoms, hip.force.as<long long>(), hip.energyBuffer.as<double>(), HipContext::EnergySlots, includeEnergy, hip.stream));
        pme.phases = OMMHIP_PME_ALL;
        return;
    }
    if (hip.usePmeStream) {
        // The whole reciprocal-space chain goes to the side stream and runs concurrently with the list rebuild and the
        // pair kernel.  (Keeping the spread on the main stream, where it does not compete with the pair kernel for wave
        // slots, was measured slower: 1126 vs 1154 ns/day -- it lengthens the main chain by more than it shortens the other.)
        static const bool spreadOnSide = getenv("OPENMM_HIP_SPREAD_ON_MAIN_STREAM") == NULL;       // A/B knob
        pme.phases = spreadOnSide ? OMMHIP_PME_ALL : OMMHIP_PME_SPREAD_ONLY;
        if (!spreadOnSide)
            HIP_CHECK(ommhip_pme_reciprocal(&pme, posq.ptr, hip.paddedAtoms, hip.force.as<long long>(), hip.energyBuffer.as<double>(), HipContext::EnergySlots, includeEnergy, hip.stream));
        hip.forkPme();
        pme.phases = spreadOnSide ? OMMHIP_PME_ALL : OMMHIP_PME_AFTER_SPREAD;
        HIP_CHECK(ommhip_pme_reciprocal(&pme, posq.ptr, hip.paddedAtoms, hip.force.as<long long>(), hip.energyBuffer.as<double>(), HipContext::EnergySlots, includeEnergy, hip.pmeStream));
        hip.markPmeDone();
        pme.phases = OMMHIP_PME_ALL;
    }
    else {
        pme.phases = OMMHIP_PME_ALL;
        HIP_CHECK(ommhip_pme_reciprocal(&pme, posq.ptr, hip.paddedAtoms, hip.force.as<long long>(), hip.energyBuffer.as<double>(), HipContext::EnergySlots, includeEnergy, hip.stream));
    }
}

double HipCalcNonbondedForceKernel::execute(ContextImpl& context, bool includeForces, bool includeEnergy, bool includeDirect, bool includeReciprocal) {
    hip.setAsCurrent();
    computeParameters(context, false);
    if (numParticles == 0) return 0.0;
    for (int i = 0; i < 6; i++) nl.box[i] = hip.box[i];
    if (usesPeriodic) {
        // ReferenceKernels.cpp:981-985
        double minAllowedSize = 1.999999 * nonbondedCutoff;
        if (hip.box[0] < minAllowedSize || hip.box[2] < minAllowedSize || hip.box[5] < minAllowedSize)
            throw OpenMMException("The periodic box size has decreased to less than twice the nonbonded cutoff.");
        nl.pbc = (hip.box[1] != 0.0 || hip.box[3] != 0.0 || hip.box[4] != 0.0) ? 2 : 1;
    }
    if (slotParamsDirty) {
        static const bool timing = getenv("OPENMM_HIP_TIMING") != NULL;
        const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        struct Report { bool on; std::chrono::steady_clock::time_point t0;
                        ~Report() { if (on) fprintf(stderr, "HIP platform: slot data of the nonbonded force after a re-sort: %.2f ms\n", 1e-3 * std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count()); } } report = {timing, t0};
        if (hip.decomposed()) {
            // every block starts without a bounding box (a hugely negative half extent: no block test can pass against it);
            // ommhip_nl_prepare gives the blocks it converts their boxes -- all of them, or in halo mode those this rank sees
            vector<float> none(4 * (size_t) (hip.paddedAtoms / OMMHIP_TILE), 0.f);
            for (size_t i = 0; i < none.size(); i += 4) none[i] = none[i + 1] = none[i + 2] = -1e30f;
            HIP_CHECK(ommhip_memcpy_h2d(blockHalf.ptr, none.data(), sizeof(float) * none.size(), hip.stream));
            hip.sync();
        }
        updateExclusionBlockRanges();
        HIP_CHECK(ommhip_set_slot_params(chargeD.as<double>(), sigmaD.as<double>(), epsilonD.as<double>(), hip.atomOfSlot.as<int>(), hip.paddedAtoms, posq.ptr, sigEps.ptr, hip.stream));
        slotParamsDirty = false;
    }
    if (hip.decomposed())
        return executeDecomposed(context, includeForces, includeEnergy, includeDirect, includeReciprocal);
    double energy = 0;
    const int ie = includeEnergy ? 1 : 0;
    bool pmeLaunched = false, frontLaunched = false, fftLaunched = false;
    // The exclusion correction belongs to the direct-space group (ReferenceLJCoulombIxn.cpp:373,462); when both halves
    // are evaluated together it is computed by the PME interpolation launch instead of a term list of its own.
    const char* noFoldEnv = getenv("OPENMM_HIP_NO_FOLDED_EXCLUSIONS");          // test/A-B knob, read per evaluation
    const bool noFold = noFoldEnv != NULL && noFoldEnv[0] == '1';
    foldExclusions = includeDirect && includeReciprocal && nonbondedMethod == PME && numExclusionPairs > 0 && !noFold;
    if (!includeDirect) hip.ensureCleared();
    if (!includeDirect)
        HIP_CHECK(ommhip_positions_to_posq(hip.pos.ptr, hip.wrap.ptr, hip.atomOfSlot.as<int>(), hip.paddedAtoms, hip.box, posq.ptr, hip.stream));
    else {
        if (nl.max_chunks == 0) allocateNeighborList(estimateChunks());
        // A list that overflowed during an earlier (device-triggered) rebuild shows up here late (the state is read back
        // asynchronously every 16th evaluation).  The device has been skipping the integration since then; the list is grown
        // and rebuilt now, and the integrator redoes the skipped steps (HipIntegratorBase::replaySkippedSteps).
        if (stateCopyPending && (pinnedState[OMMHIP_NL_STATE_OVERFLOW] != 0 || pinnedState[1] > nl.max_chunks)) {
            const int skipped = recoverFromOverflow();
            if (hip.freezeState != NULL) hip.pendingReplay += skipped;
        }
        // test hook: after a few evaluations pretend the allocation is a little smaller than the list and request a rebuild the
        // way the displacement check does (no host verification), so that it overflows (tests/test_gpu_platform.py::test_neighbour_list_overflow_is_recovered)
        const int debugShrinkAfter = getenv("OPENMM_HIP_DEBUG_SHRINK_LIST_AFTER") != NULL ? atoi(getenv("OPENMM_HIP_DEBUG_SHRINK_LIST_AFTER")) : 0;      // read per evaluation
        if (debugShrinkAfter > 0 && !debugShrinkDone && (int) evaluationCount == debugShrinkAfter && nl.max_chunks > 0) {
            HIP_CHECK(ommhip_memcpy_d2h(pinnedState, nlState.ptr, sizeof(int) * OMMHIP_NL_STATE_INTS, hip.stream));
            hip.sync();
            nl.max_chunks = std::max(1, pinnedState[1] - 8);
            const int one = 1;
            HIP_CHECK(ommhip_memcpy_h2d(nlState.ptr, &one, sizeof(int), hip.stream));      // ... and make this evaluation rebuild on the device's own path
            hip.sync();
            debugShrinkDone = true;
        }
        while (true) {
            if (forceRebuild) {
                const int request[3] = {1, 0, 0};     // REBUILD = 1, NUM_CHUNKS = 0, OVERFLOW = 0
                HIP_CHECK(ommhip_memcpy_h2d(nlState.ptr, request, sizeof(request), hip.stream));
            }
            // positions -> posq, displacement check, bounds; then the device-conditional rebuild (2 launches)
            static const bool noFront = getenv("OPENMM_HIP_NO_FUSED_FRONT") != NULL;          // A/B knob
            // (small systems only: at a million atoms the builder workgroups of a rebuild get in each other's way with the spread
            // workgroups of the same launch -- 3.5 ms against 2.1 + 0.3 ms as launches of their own, profiles/r02a vs r02c)
            static const int frontMaxAtoms = getenv("OPENMM_HIP_FUSED_FRONT_MAX_ATOMS") != NULL ? atoi(getenv("OPENMM_HIP_FUSED_FRONT_MAX_ATOMS")) : 60000;
            if (!forceRebuild && includeReciprocal && nonbondedMethod == PME && !hip.usePmeStream && !noFront && pme.grid_precleared && pme.spread_mode == 0 && numParticles <= frontMaxAtoms && !hip.deterministicForces) {
                // single-stream mode: list rebuild (if requested), charge spreading and every per-term force list of this
                // evaluation are independent once the positions are converted -- they go out as ONE launch
                const bool clear = hip.takePendingClear();
                HIP_CHECK(ommhip_nl_prepare(&nl, hip.pos.ptr, hip.wrap.ptr, clear ? hip.force.ptr : NULL, clear ? hip.force.bytes : 0,
                                            clear ? hip.extraClearPtr : NULL, clear ? hip.extraClearBytes : 0, hip.stream));
                fillPmeStruct();
                vector<ommhip_term_batch> lists;
                ommhip_term_batch t14 = {OMMHIP_TERM_EXCEPTION14, {num14, exceptionAtomsD.as<int>(), exceptionParamsD.as<double>()},
                                         exceptionsArePeriodic ? 1 : 0, chargeD.as<double>(), ewaldAlpha};
                if (num14 > 0) lists.push_back(t14);
                if (params.ewald && !foldExclusions && numExclusionPairs > 0) {
                    ommhip_term_batch tex = {OMMHIP_TERM_EWALD_EXCLUSION, {numExclusionPairs, exclusionPairsD.as<int>(), NULL},
                                             exceptionsArePeriodic ? 1 : 0, chargeD.as<double>(), ewaldAlpha};
                    lists.push_back(tex);
                }
                hip.collectFrontTerms(lists, includeEnergy);
                HIP_CHECK(ommhip_force_front(&nl, &pme, (int) lists.size(), lists.empty() ? NULL : lists.data(), hip.pos.ptr, hip.force.as<long long>(),
                                             hip.energyBuffer.as<double>(), HipContext::EnergySlots, ie, hip.stream));
                frontLaunched = true;
            }
            else if (!forceRebuild && includeReciprocal && nonbondedMethod == PME && hip.usePmeStream) {
                // side-stream mode: reciprocal space only needs posq, so it is forked BEFORE the (possible) list rebuild
                // and overlaps with it as well as with the pair kernel
                const bool clear = hip.takePendingClear();
                HIP_CHECK(ommhip_nl_prepare(&nl, hip.pos.ptr, hip.wrap.ptr, clear ? hip.force.ptr : NULL, clear ? hip.force.bytes : 0,
                                            clear ? hip.extraClearPtr : NULL, clear ? hip.extraClearBytes : 0, hip.stream));
                launchPme(ie);
                pmeLaunched = true;
                HIP_CHECK(ommhip_nl_rebuild_if_requested(&nl, hip.stream));
                // (Joining the streams again before the pair kernel -- so that only the latency-bound rebuild overlaps with
                // reciprocal space -- was measured 14 % slower: a cross-stream wait on the critical path costs more than the
                // contention between the pair kernel and the small PME launches.)
            }
            else if (hip.takePendingClear())
                HIP_CHECK(ommhip_nl_step_clear(&nl, hip.pos.ptr, hip.wrap.ptr, hip.force.ptr, hip.force.bytes, hip.extraClearPtr, hip.extraClearBytes, hip.stream));
            else
                HIP_CHECK(ommhip_nl_step(&nl, hip.pos.ptr, hip.wrap.ptr, hip.stream));
            if (!forceRebuild) break;
            // host-requested rebuild: verify the capacity synchronously
            HIP_CHECK(ommhip_memcpy_d2h(pinnedState, nlState.ptr, sizeof(int) * OMMHIP_NL_STATE_INTS, hip.stream));
            hip.sync();
            // Keep 1.5x headroom over the measured list length: later (device-triggered) rebuilds are only checked
            // lazily, so the list must never come close to its capacity through ordinary density fluctuations.
            if (pinnedState[2] == 0 && pinnedState[1] * 1.5 <= nl.max_chunks) { forceRebuild = false; break; }
            allocateNeighborList((int) (pinnedState[1] * 1.6) + 64);
        }
        // posq is ready: start reciprocal space on the side stream BEFORE queueing the pair kernel, so the two overlap
        if (!pmeLaunched && includeReciprocal && nonbondedMethod == PME && hip.usePmeStream) { launchPme(ie); pmeLaunched = true; }
        // Single-stream mode: the pair kernel travels with the three FFT launches of reciprocal space (a third of its chunks
        // in each), which fills the compute units the 56-112 FFT workgroups leave idle.
        const char* noPairsFft = getenv("OPENMM_HIP_NO_PAIRS_WITH_FFT");          // A/B and test knob, read per evaluation
        const bool pairsWithFft = !(noPairsFft != NULL && noPairsFft[0] == '1');
        if (frontLaunched && pairsWithFft) {
            int rc = ommhip_pairs_with_fft(&nl, &params, sigEps.ptr, &pme, hip.force.as<long long>(), hip.energyBuffer.as<double>(), HipContext::EnergySlots, ie, hip.stream);
            if (rc > 0) HIP_CHECK(rc);
            fftLaunched = rc == 0;
        }
        if (!fftLaunched) {
            if (pmeLaunched && hip.usePmeStream) params.direct_grid = pairGridBesideSideStream();
            HIP_CHECK(ommhip_nb_direct(&nl, &params, sigEps.ptr, hip.force.as<long long>(), hip.energyBuffer.as<double>(), HipContext::EnergySlots, ie, hip.stream));
            params.direct_grid = directGridOverride > 0 ? directGridOverride : 0;
        }
        // the list's state words come back every 16th evaluation -- and with every energy: that evaluation ends in a host
        // synchronisation anyway (HipContext::reduceEnergy), after which HipCalcForcesAndEnergyKernel::finishComputation looks at them
        if ((++evaluationCount & 15) == 0 || includeEnergy) {
            HIP_CHECK(ommhip_memcpy_d2h(pinnedState, nlState.ptr, sizeof(int) * OMMHIP_NL_STATE_INTS, hip.stream));
            stateCopyPending = true;
        }
        // 1-4 exceptions and the Ewald exclusion correction are queued; they leave in one launch together with the
        // bonded terms of the other forces (HipContext::flushTerms)
        ommhip_term_batch t14 = {OMMHIP_TERM_EXCEPTION14, {num14, exceptionAtomsD.as<int>(), exceptionParamsD.as<double>()},
                                 exceptionsArePeriodic ? 1 : 0, chargeD.as<double>(), ewaldAlpha};
        if (!frontLaunched) hip.addTerms(t14, includeEnergy);
        if (params.ewald && !foldExclusions && !frontLaunched) {
            ommhip_term_batch tex = {OMMHIP_TERM_EWALD_EXCLUSION, {numExclusionPairs, exclusionPairsD.as<int>(), NULL},
                                     exceptionsArePeriodic ? 1 : 0, chargeD.as<double>(), ewaldAlpha};
            hip.addTerms(tex, includeEnergy);
            if (nonbondedMethod == LJPME) {
                ommhip_term_batch tdx = {OMMHIP_TERM_DISPERSION_EXCLUSION, {numExclusionPairs, exclusionPairsD.as<int>(), NULL},
                                         exceptionsArePeriodic ? 1 : 0, c6D.as<double>(), dispersionAlpha};
                hip.addTerms(tdx, includeEnergy);
            }
        }
        if (includeEnergy && usesPeriodic && nonbondedMethod != LJPME)         // ReferenceKernels.cpp:1008-1011 (not for LJPME)
            energy += dispersionCoefficient / (hip.box[0] * hip.box[2] * hip.box[5]);
    }
    if (includeReciprocal) {
        if (nonbondedMethod == PME) {
            if (!pmeLaunched) launchPme(ie, frontLaunched, fftLaunched);
        }
        else if (nonbondedMethod == LJPME) {
            // two grids back to back on the main stream: charges, then C6 factors at the same positions
            fillPmeStruct();
            pme.phases = OMMHIP_PME_ALL;
            HIP_CHECK(ommhip_pme_reciprocal(&pme, posq.ptr, hip.paddedAtoms, hip.force.as<long long>(), hip.energyBuffer.as<double>(), HipContext::EnergySlots, ie, hip.stream));
            for (int i = 0; i < 6; i++) pmeDisp.box[i] = hip.box[i];
            if (dispersionEtermDirty) { HIP_CHECK(ommhip_pme_build_eterm(&pmeDisp, hip.stream)); dispersionEtermDirty = false; }
            HIP_CHECK(ommhip_posq_with_weights(posq.ptr, c6D.as<double>(), hip.atomOfSlot.as<int>(), hip.paddedAtoms, posqDisp.ptr, hip.stream));
            HIP_CHECK(ommhip_pme_reciprocal(&pmeDisp, posqDisp.ptr, hip.paddedAtoms, hip.force.as<long long>(), hip.energyBuffer.as<double>(), HipContext::EnergySlots, ie, hip.stream));
        }
        else if (nonbondedMethod == Ewald) {
            if (hip.box[1] != 0.0 || hip.box[3] != 0.0 || hip.box[4] != 0.0)
                throw OpenMMException("Ewald (as opposed to PME) requires a rectangular periodic box");   // ReferenceLJCoulombIxn.cpp:99-100
            HIP_CHECK(ommhip_ewald_reciprocal(hip.pos.ptr, chargeD.as<double>(), hip.slotOfAtom.as<int>(), numParticles, hip.paddedAtoms, hip.box, ewaldAlpha,
                                              kmax[0], kmax[1], kmax[2], ewaldStructure.ptr, hip.force.as<long long>(), hip.energyBuffer.as<double>(), HipContext::EnergySlots, ie, hip.stream));
        }
        if (includeEnergy && params.ewald) energy += selfEnergy;
    }
    return energy;
}

void HipCalcNonbondedForceKernel::copyParametersToContext(ContextImpl& context, const NonbondedForce& force) {
    // ReferenceKernels.cpp:1016-1060
    if (force.getNumParticles() != numParticles)
        throw OpenMMException("updateParametersInContext: The number of particles has changed");
    set<int> exceptionsWithOffsets;
    for (int i = 0; i < force.getNumExceptionParameterOffsets(); i++) {
        string param; int exception; double charge, sigma, epsilon;
        force.getExceptionParameterOffset(i, param, exception, charge, sigma, epsilon);
        exceptionsWithOffsets.insert(exception);
    }
    vector<int> nb14s;
    for (int i = 0; i < force.getNumExceptions(); i++) {
        int p1, p2; double chargeProd, sigma, epsilon;
        force.getExceptionParameters(i, p1, p2, chargeProd, sigma, epsilon);
        if (chargeProd != 0.0 || epsilon != 0.0 || exceptionsWithOffsets.find(i) != exceptionsWithOffsets.end())
            nb14s.push_back(i);
    }
    if ((int) nb14s.size() != num14)
        throw OpenMMException("updateParametersInContext: The number of non-excluded exceptions has changed");
    transposeTermOrder(nb14s);
    for (int i = 0; i < numParticles; i++)
        force.getParticleParameters(i, baseParticleParams[i][0], baseParticleParams[i][1], baseParticleParams[i][2]);
    for (int i = 0; i < num14; i++) {
        int p1, p2;
        force.getExceptionParameters(nb14s[i], p1, p2, baseExceptionParams[i][0], baseExceptionParams[i][1], baseExceptionParams[i][2]);
        if (p1 != exceptionAtoms[i].first || p2 != exceptionAtoms[i].second)
            throw OpenMMException("updateParametersInContext: The set of particles in an exception has changed");
    }
    NonbondedForce::NonbondedMethod method = force.getNonbondedMethod();
    if (force.getUseDispersionCorrection() && (method == NonbondedForce::CutoffPeriodic || method == NonbondedForce::Ewald || method == NonbondedForce::PME))
        dispersionCoefficient = NonbondedForceImpl::calcDispersionCorrection(context.getSystem(), force);
    hip.setAsCurrent();
    computeParameters(context, true);
}

void HipCalcNonbondedForceKernel::getPMEParameters(double& alpha, int& nx, int& ny, int& nz) const {
    if (nonbondedMethod != PME && nonbondedMethod != LJPME)
        throw OpenMMException("getPMEParametersInContext: This Context is not using PME or LJPME");
    alpha = ewaldAlpha; nx = gridSize[0]; ny = gridSize[1]; nz = gridSize[2];
}

void HipCalcNonbondedForceKernel::getLJPMEParameters(double& alpha, int& nx, int& ny, int& nz) const {
    if (nonbondedMethod != LJPME)
        throw OpenMMException("getPMEParametersInContext: This Context is not using LJPME");
    alpha = dispersionAlpha; nx = dispersionGridSize[0]; ny = dispersionGridSize[1]; nz = dispersionGridSize[2];
}

// ================================================================================================
// CalcPmeReciprocalForce (kernels.h:1493-1560): reciprocal space alone, host arrays in and out
// ================================================================================================
HipCalcPmeReciprocalForceKernel::HipCalcPmeReciprocalForceKernel(std::string name, const Platform& platform, int deviceIndex) : CalcPmeReciprocalForceKernel(name, platform),
        deviceIndex(deviceIndex), numParticles(0), paddedAtoms(0), alpha(0.0), deterministic(false), includeEnergy(false), started(false), stream(NULL), pinnedEnergy(NULL) {
    grid[0] = grid[1] = grid[2] = 0;
    memset(&pme, 0, sizeof(pme));
    for (int i = 0; i < 6; i++) lastBox[i] = 0.0;
}

HipCalcPmeReciprocalForceKernel::~HipCalcPmeReciprocalForceKernel() {
    if (stream != NULL) { ommhip_stream_sync(stream); ommhip_stream_destroy(stream); }
    if (pinnedEnergy != NULL) ommhip_host_free(pinnedEnergy);
}

void HipCalcPmeReciprocalForceKernel::initialize(int gridx, int gridy, int gridz, int numParticles, double alpha, bool deterministic) {
    // the caller chose the grid (NonbondedForceImpl::calcPMEParameters rounds to FFT-friendly sizes of ITS transform): this one takes
    // lengths that factor into 2, 3, 5 and 7 and fit a line buffer in LDS
    if (!ommhip_fft_supported_size(gridx) || !ommhip_fft_supported_size(gridy) || !ommhip_fft_supported_size(gridz))
        throw OpenMMException("HIP platform: CalcPmeReciprocalForceKernel: grid dimensions must factor into 2, 3, 5 and 7");
    if (deviceIndex >= 0) HIP_CHECK(ommhip_set_device(deviceIndex));
    if (stream == NULL) HIP_CHECK(ommhip_stream_create(&stream));
    if (pinnedEnergy == NULL) HIP_CHECK(ommhip_host_malloc((void**) &pinnedEnergy, sizeof(double) * 8));
    this->numParticles = numParticles; this->alpha = alpha; this->deterministic = deterministic;
    grid[0] = gridx; grid[1] = gridy; grid[2] = gridz;
    paddedAtoms = max(OMMHIP_TILE, (numParticles + OMMHIP_TILE - 1) / OMMHIP_TILE * OMMHIP_TILE);
    const int nzc = gridz / 2 + 1;
    uploadVector(moduliX, bsplineModuli(gridx), stream);
    uploadVector(moduliY, bsplineModuli(gridy), stream);
    uploadVector(moduliZ, bsplineModuli(gridz), stream);
    DeviceBuffer* tw[3] = {&twiddleX, &twiddleY, &twiddleZ};
    for (int d = 0; d < 3; d++) {
        const int n = grid[d];
        vector<float> t(2 * (size_t) n);
        for (int k = 0; k < n; k++) { t[2 * k] = (float) cos(2.0 * M_PI * k / n); t[2 * k + 1] = (float) -sin(2.0 * M_PI * k / n); }
        uploadVector(*tw[d], t, stream);
    }
    eterm.allocate(sizeof(float) * (size_t) gridx * gridy * nzc);
    gridReal.allocate((sizeof(float) * (size_t) gridx * gridy * gridz + 15) / 16 * 16);
    gridComplex.allocate(sizeof(float) * 2 * (size_t) gridx * gridy * nzc);
    posq.allocate(sizeof(float) * 4 * (size_t) paddedAtoms);
    force.allocate(sizeof(long long) * 3 * (size_t) paddedAtoms);
    forceDouble.allocate(sizeof(double) * 3 * (size_t) max(numParticles, 1));
    energyBuffer.allocate(sizeof(double) * HipContext::EnergySlots);
    energyResult.allocate(sizeof(double) * 8);
    HIP_CHECK(ommhip_memset(energyBuffer.ptr, 0, energyBuffer.bytes, stream));
    vector<int> identity(max(numParticles, 1));
    for (int i = 0; i < numParticles; i++) identity[i] = i;          // slot order = atom order: the caller's posq is taken as it comes
    uploadVector(slotOfAtom, identity, stream);
    memset(&pme, 0, sizeof(pme));
    pme.nx = gridx; pme.ny = gridy; pme.nz = gridz; pme.alpha = alpha;
    pme.moduli_x = moduliX.as<double>(); pme.moduli_y = moduliY.as<double>(); pme.moduli_z = moduliZ.as<double>();
    pme.eterm = eterm.ptr; pme.grid_real = gridReal.ptr; pme.grid_complex = gridComplex.ptr;
    pme.twiddle_x = twiddleX.ptr; pme.twiddle_y = twiddleY.ptr; pme.twiddle_z = twiddleZ.ptr;
    pme.phases = OMMHIP_PME_ALL;
    for (int i = 0; i < 6; i++) lastBox[i] = 0.0;
    HIP_CHECK(ommhip_stream_sync(stream));
}

void HipCalcPmeReciprocalForceKernel::beginComputation(IO& io, const Vec3* periodicBoxVectors, bool includeEnergy) {
    if (stream == NULL) throw OpenMMException("HIP platform: CalcPmeReciprocalForceKernel::beginComputation before initialize");
    if (deviceIndex >= 0) HIP_CHECK(ommhip_set_device(deviceIndex));
    this->includeEnergy = includeEnergy;
    const double box[6] = {periodicBoxVectors[0][0], periodicBoxVectors[1][0], periodicBoxVectors[1][1], periodicBoxVectors[2][0], periodicBoxVectors[2][1], periodicBoxVectors[2][2]};
    bool boxChanged = false;
    for (int i = 0; i < 6; i++) { if (box[i] != lastBox[i]) boxChanged = true; pme.box[i] = lastBox[i] = box[i]; }
    if (boxChanged) HIP_CHECK(ommhip_pme_build_eterm(&pme, stream));
    // posq: x, y, z, q per atom, as the reference's GPU platforms hand it over (CudaKernels.cpp PmeIO); padding slots carry no charge
    vector<float> hostPosq(4 * (size_t) paddedAtoms, 0.f);
    const float* in = io.getPosq();
    double maxCharge = 0.0;
    for (int i = 0; i < numParticles; i++) {
        for (int k = 0; k < 4; k++) hostPosq[4 * (size_t) i + k] = in[4 * (size_t) i + k];
        maxCharge = max(maxCharge, (double) fabs(in[4 * (size_t) i + 3]));
    }
    for (int i = numParticles; i < paddedAtoms; i++)
        for (int k = 0; k < 3; k++) hostPosq[4 * (size_t) i + k] = numParticles > 0 ? in[k] : 0.f;
    pme.deterministic = deterministic ? 1 : 0; pme.max_charge = maxCharge;
    HIP_CHECK(ommhip_memcpy_h2d(posq.ptr, hostPosq.data(), sizeof(float) * hostPosq.size(), stream));
    HIP_CHECK(ommhip_memset(force.ptr, 0, force.bytes, stream));
    HIP_CHECK(ommhip_pme_reciprocal(&pme, posq.ptr, paddedAtoms, force.as<long long>(), energyBuffer.as<double>(), HipContext::EnergySlots, includeEnergy ? 1 : 0, stream));
    HIP_CHECK(ommhip_forces_to_double(force.as<long long>(), slotOfAtom.as<int>(), numParticles, paddedAtoms, forceDouble.as<double>(), stream));
    hostForceDouble.resize(3 * (size_t) max(numParticles, 1));
    HIP_CHECK(ommhip_stream_sync(stream));          // (hostPosq goes out of scope)
    if (numParticles > 0) HIP_CHECK(ommhip_memcpy_d2h(hostForceDouble.data(), forceDouble.ptr, sizeof(double) * 3 * (size_t) numParticles, stream));
    if (includeEnergy) {
        HIP_CHECK(ommhip_reduce_energy(energyBuffer.as<double>(), HipContext::EnergySlots, energyResult.as<double>(), stream));
        HIP_CHECK(ommhip_memcpy_d2h(pinnedEnergy, energyResult.ptr, sizeof(double), stream));
    }
    started = true;
}

double HipCalcPmeReciprocalForceKernel::finishComputation(IO& io) {
    if (!started) throw OpenMMException("HIP platform: CalcPmeReciprocalForceKernel::finishComputation without beginComputation");
    started = false;
    if (deviceIndex >= 0) HIP_CHECK(ommhip_set_device(deviceIndex));
    HIP_CHECK(ommhip_stream_sync(stream));
    hostForce.assign(4 * (size_t) max(numParticles, 1), 0.f);
    for (int i = 0; i < numParticles; i++)
        for (int k = 0; k < 3; k++) hostForce[4 * (size_t) i + k] = (float) hostForceDouble[3 * (size_t) i + k];
    io.setForce(hostForce.data());
    return includeEnergy ? pinnedEnergy[0] : 0.0;
}

void HipCalcPmeReciprocalForceKernel::getPMEParameters(double& alpha, int& nx, int& ny, int& nz) const {
    alpha = this->alpha; nx = grid[0]; ny = grid[1]; nz = grid[2];
}

// ================================================================================================
// Bonded terms
// ================================================================================================
ommhip_term_batch HipTermForce::batch() const {
    ommhip_term_batch b = {kind, {numTerms, atomsD.as<int>(), paramsD.as<double>()}, periodic ? 1 : 0, NULL, 0.0};
    return b;
}
HipTermForce::~HipTermForce() {
    if (registrationId >= 0 && data.hip != NULL) data.hip->unregisterTerms(registrationId);
}
void HipTermForce::upload(const vector<int>& atoms, const vector<double>& params, bool usesPeriodic, int forceGroup) {
    data.hip->setAsCurrent();
    numTerms = (int) atoms.size() / atomsPerTerm;
    periodic = usesPeriodic;
    if (usesPeriodic) data.hip->usePeriodic = true;
    // Device order = the caller's order by default: neighbouring terms share atoms, so the position gathers of a wavefront touch
    // few cache lines.  (Storing the lists "transposed" -- lane l of wavefront w takes term l * W + w, so that no two lanes send a
    // fixed-point atomic to the same address -- was measured on the real DHFR System: k_terms 16.9 us against 14.1 us; the
    // scattered gathers cost more than the serialised atomics.  OPENMM_HIP_TRANSPOSE_TERMS=1 keeps it for A/B.)
    static const bool transpose = getenv("OPENMM_HIP_TRANSPOSE_TERMS") != NULL && getenv("OPENMM_HIP_TRANSPOSE_TERMS")[0] == '1';
    order.resize(numTerms);
    const int waves = (numTerms + 63) / 64;
    int t = 0;
    for (int w = 0; w < waves && transpose; w++)
        for (int l = 0; l < 64; l++) {
            const int original = l * waves + w;
            if (original < numTerms) order[t++] = original;
        }
    if (!transpose)
        for (int i = 0; i < numTerms; i++) order[i] = i;
    vector<int> atomsP(atoms.size());
    vector<double> paramsP(params.size());
    for (int i = 0; i < numTerms; i++) {
        for (int k = 0; k < atomsPerTerm; k++) atomsP[(size_t) i * atomsPerTerm + k] = atoms[(size_t) order[i] * atomsPerTerm + k];
        for (int k = 0; k < paramsPerTerm; k++) paramsP[(size_t) i * paramsPerTerm + k] = params[(size_t) order[i] * paramsPerTerm + k];
    }
    uploadVector(atomsD, atomsP, data.hip->stream);
    uploadVector(paramsD, paramsP, data.hip->stream);
    if (registrationId < 0) registrationId = data.hip->registerTerms(forceGroup, batch());
    else data.hip->updateTerms(registrationId, batch());
}
void HipTermForce::uploadParams(const vector<double>& params) {
    data.hip->setAsCurrent();
    if ((int) params.size() != numTerms * paramsPerTerm)
        throw OpenMMException("updateParametersInContext: The number of terms has changed");
    vector<double> paramsP(params.size());
    for (int i = 0; i < numTerms; i++)
        for (int k = 0; k < paramsPerTerm; k++) paramsP[(size_t) i * paramsPerTerm + k] = params[(size_t) order[i] * paramsPerTerm + k];
    uploadVector(paramsD, paramsP, data.hip->stream);
    if (registrationId >= 0) data.hip->updateTerms(registrationId, batch());      // uploadVector may have moved the buffer
}
void HipTermForce::execute(bool includeEnergy) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    if (registrationId >= 0 && hip.termsLaunched(registrationId)) return;         // went out with the front launch of this evaluation
    // decomposed runs: every rank evaluates all bonded terms (they are few; forces on atoms it does not own are never read),
    // their energy is taken from rank 0 only
    hip.addTerms(batch(), includeEnergy && hip.countsTermEnergy(), registrationId);
}

void HipCalcHarmonicBondForceKernel::initialize(const System& system, const HarmonicBondForce& force) {
    vector<int> atoms; vector<double> params;
    for (int i = 0; i < force.getNumBonds(); i++) {
        int p1, p2; double length, k;
        force.getBondParameters(i, p1, p2, length, k);
        atoms.push_back(p1); atoms.push_back(p2); params.push_back(length); params.push_back(k);
    }
    terms.upload(atoms, params, force.usesPeriodicBoundaryConditions(), force.getForceGroup());
}
double HipCalcHarmonicBondForceKernel::execute(ContextImpl& context, bool includeForces, bool includeEnergy) {
    terms.execute(includeEnergy);
    return 0.0;
}
void HipCalcHarmonicBondForceKernel::copyParametersToContext(ContextImpl& context, const HarmonicBondForce& force) {
    vector<double> params;
    for (int i = 0; i < force.getNumBonds(); i++) {
        int p1, p2; double length, k;
        force.getBondParameters(i, p1, p2, length, k);
        params.push_back(length); params.push_back(k);
    }
    terms.uploadParams(params);
}

void HipCalcHarmonicAngleForceKernel::initialize(const System& system, const HarmonicAngleForce& force) {
    vector<int> atoms; vector<double> params;
    for (int i = 0; i < force.getNumAngles(); i++) {
        int p1, p2, p3; double angle, k;
        force.getAngleParameters(i, p1, p2, p3, angle, k);
        atoms.push_back(p1); atoms.push_back(p2); atoms.push_back(p3); params.push_back(angle); params.push_back(k);
    }
    terms.upload(atoms, params, force.usesPeriodicBoundaryConditions(), force.getForceGroup());
}
double HipCalcHarmonicAngleForceKernel::execute(ContextImpl& context, bool includeForces, bool includeEnergy) {
    terms.execute(includeEnergy);
    return 0.0;
}
void HipCalcHarmonicAngleForceKernel::copyParametersToContext(ContextImpl& context, const HarmonicAngleForce& force) {
    vector<double> params;
    for (int i = 0; i < force.getNumAngles(); i++) {
        int p1, p2, p3; double angle, k;
        force.getAngleParameters(i, p1, p2, p3, angle, k);
        params.push_back(angle); params.push_back(k);
    }
    terms.uploadParams(params);
}

void HipCalcPeriodicTorsionForceKernel::packTorsions(const PeriodicTorsionForce& force, vector<int>& atoms, vector<double>& params) {
    // every (periodicity, phase, k) on the same four atoms -- force fields use up to four per dihedral -- becomes a sub-term of ONE
    // device term (OMMHIP_TORSION_SUBTERMS slots, k = 0 when unused); a fifth one opens another device term on the same atoms
    map<vector<int>, int> open;            // atoms -> device term that still has a free slot
    vector<int> used;                       // sub-terms filled per device term
    for (int i = 0; i < force.getNumTorsions(); i++) {
        int p1, p2, p3, p4, periodicity; double phase, k;
        force.getTorsionParameters(i, p1, p2, p3, p4, periodicity, phase, k);
        const vector<int> key = {p1, p2, p3, p4};
        map<vector<int>, int>::iterator it = open.find(key);
        int term;
        if (it == open.end() || used[it->second] == OMMHIP_TORSION_SUBTERMS) {
            term = (int) used.size();
            used.push_back(0);
            atoms.insert(atoms.end(), key.begin(), key.end());
            params.resize(params.size() + 4 * OMMHIP_TORSION_SUBTERMS, 0.0);
            open[key] = term;
        }
        else term = it->second;
        double* slot = &params[(size_t) term * 4 * OMMHIP_TORSION_SUBTERMS + 4 * used[term]++];
        slot[0] = k; slot[1] = cos(phase); slot[2] = sin(phase); slot[3] = periodicity;
    }
}
void HipCalcPeriodicTorsionForceKernel::initialize(const System& system, const PeriodicTorsionForce& force) {
    vector<int> atoms; vector<double> params;
    packTorsions(force, atoms, params);
    terms.upload(atoms, params, force.usesPeriodicBoundaryConditions(), force.getForceGroup());
}
double HipCalcPeriodicTorsionForceKernel::execute(ContextImpl& context, bool includeForces, bool includeEnergy) {
    terms.execute(includeEnergy);
    return 0.0;
}
void HipCalcPeriodicTorsionForceKernel::copyParametersToContext(ContextImpl& context, const PeriodicTorsionForce& force) {
    // the atoms of a torsion cannot change (PeriodicTorsionForce.h: updateParametersInContext), so the grouping is the same
    vector<int> atoms; vector<double> params;
    packTorsions(force, atoms, params);
    terms.uploadParams(params);
}

// ================================================================================================
// Integrators
// ================================================================================================
void HipIntegratorBase::fillState(ommhip_integrator_state& s, double dt) {
    HipContext& hip = *data.hip;
    memset(&s, 0, sizeof(s));
    s.num_atoms = hip.numAtoms; s.padded_atoms = hip.paddedAtoms; s.dt = dt;
    s.pos = hip.pos.ptr; s.vel = hip.vel.ptr; s.xp = hip.xp.ptr; s.oldx = hip.oldx.ptr;
    s.force = hip.force.as<long long>(); s.slot_of_atom = hip.slotOfAtom.as<int>();
    s.step = (unsigned long long) data.stepCount;
    s.freeze_state = hip.freezeState;
}

void HipIntegratorBase::runSteps(ContextImpl& context, const Integrator& integrator, const std::function<void(long long)>& launch,
                                 long long firstIndex, long long endIndex, bool haveForces) {
    // Steps firstIndex .. endIndex-1, each a force evaluation plus launch(index) -- the first one without the evaluation if the
    // current forces already belong to it.  This is the ordinary single step (firstIndex + 1 == endIndex) and, after a
    // neighbour-list overflow, the steps the device skipped (HipContext::pendingReplay, filled by the nonbonded kernel while
    // it evaluates the forces): they are redone in order with the step indices they would have had (the thermostat noise is
    // keyed by the index), so the trajectory is the one an unlimited list would have given.
    HipContext& hip = *data.hip;
    long long next = firstIndex;
    while (next < endIndex) {
        if (!haveForces) {
            hip.currentStepIndex = next;             // what a force's updateContextState decides by (CMMotionRemover's frequency): the step being (re)done
            context.updateContextState();
            hip.currentStepIndex = -1;
            context.calcForcesAndEnergy(true, false, integrator.getIntegrationForceGroups());
            if (hip.pendingReplay > 0) {             // overflowed again meanwhile: the last pendingReplay launches did nothing
                next -= hip.pendingReplay;
                hip.pendingReplay = 0;
            }
        }
        haveForces = false;
        launch(next++);
    }
}

double HipIntegratorBase::kineticEnergy(double timeShift) {
    // ReferenceKernels.cpp:146-176
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    // Decomposed run: every rank sums over its own atoms (their velocities, and the forces a time-shifted estimate needs, are
    // local; constraint-connected units are never split between ranks) and the partial sums are added in rank order on every
    // rank.  The kernels below also run over the atoms of the other ranks -- on stale data whose results are not read.
    HipConstraints& constraints = data.getDeviceConstraints(*data.system);
    // No time shift and velocities as a native step left them: they are what the Reference would get back from its constraint pass (it
    // leaves velocities inside its tolerance alone), so the sum runs over them directly -- two launches fewer in every energy query.
    const bool asTheyAre = timeShift == 0.0 && (!constraints.hasConstraints() || hip.velocitiesConstrained) && getenv("OPENMM_HIP_KE_ALWAYS_CONSTRAIN") == NULL;
    if (asTheyAre && hip.kineticEnergyPrefetched && hip.kineticEnergyMutations == hip.stateMutations) {
        // summed behind the energy evaluation this query began with and read with its energy (HipContext::reduceEnergy): no second round trip
        hip.kineticEnergyPrefetched = false;
        return hip.prefetchedKineticEnergy;
    }
    hip.kineticEnergyPrefetched = false;
    void* const velocities = asTheyAre ? hip.vel.ptr : hip.tempVel.ptr;
    if (!asTheyAre) {
        ommhip_integrator_state s;
        fillState(s, 0.0);
        HIP_CHECK(ommhip_shifted_velocities(&s, timeShift, hip.tempVel.ptr, hip.stream));
        if (constraints.hasConstraints()) constraints.applyToVelocities(hip.tempVel.ptr, 1e-4);
    }
    double* const result_d = hip.energyResult.as<double>() + 1;
    double* const scratch_d = hip.energyResult.as<double>() + 8;
    if (hip.decomposed())
        HIP_CHECK(ommhip_kinetic_energy(velocities, hip.atomOfSlot.as<int>(), hip.ownSlot0, hip.ownSlot1, scratch_d, result_d, hip.stream));
    else
        HIP_CHECK(ommhip_kinetic_energy(velocities, NULL, 0, hip.numAtoms, scratch_d, result_d, hip.stream));
    double result = 0;
    HIP_CHECK(ommhip_memcpy_d2h(&result, result_d, sizeof(double), hip.stream));
    hip.sync();
    return hip.sumOverRanks(result);
}

void HipIntegratorBase::finishStep(double dt) {
    data.time += dt;
    data.stepCount++;
    data.hip->stepTaken();
}

void HipIntegrateVerletStepKernel::launchStep(ContextImpl& context, const VerletIntegrator& integrator, long long stepIndex) {
    // ReferenceVerletDynamics.cpp:76-119
    HipContext& hip = *data.hip;
    const double dt = integrator.getStepSize();
    ommhip_integrator_state s;
    fillState(s, dt);
    s.step = (unsigned long long) stepIndex;
    HipConstraints& constraints = data.getDeviceConstraints(context.getSystem());
    if (constraints.fusedStepAvailable()) constraints.fusedStep(OMMHIP_INTEGRATOR_VERLET, s, integrator.getConstraintTolerance());
    else if (hip.decomposed()) throw OpenMMException("HIP platform: multi-GPU runs need constraints that form SETTLE waters or X-H clusters (no CCMA)");
    else {
        HIP_CHECK(ommhip_integrate_stage(OMMHIP_STAGE_VERLET_1, &s, hip.stream));
        if (constraints.hasConstraints()) constraints.apply(hip.xp.ptr, integrator.getConstraintTolerance());
        HIP_CHECK(ommhip_integrate_stage(OMMHIP_STAGE_FINISH_POSITIONS, &s, hip.stream));
        hip.momentumValid = false;
    }
}
void HipIntegrateVerletStepKernel::execute(ContextImpl& context, const VerletIntegrator& integrator) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    const std::function<void(long long)> launch = [this, &context, &integrator](long long index) { launchStep(context, integrator, index); };
    hip.replaySteps = [this, &context, &integrator, launch](int steps) { runSteps(context, integrator, launch, (long long) data.stepCount - steps, data.stepCount, false); };
    // the forces just evaluated belong to the oldest step not done yet: this one, or the first of those the device skipped
    const int skipped = hip.pendingReplay;
    hip.pendingReplay = 0;
    runSteps(context, integrator, launch, (long long) data.stepCount - skipped, (long long) data.stepCount + 1, true);
    finishStep(integrator.getStepSize());
}
double HipIntegrateVerletStepKernel::computeKineticEnergy(ContextImpl& context, const VerletIntegrator& integrator) {
    return kineticEnergy(0.5 * integrator.getStepSize());
}

static unsigned long long resolveSeed(int seed) {
    // seed 0 means "pick one" (LangevinIntegrator.h); same convention as the other platforms
    if (seed == 0) return (unsigned long long) osrngseed();
    return (unsigned long long) (unsigned int) seed;
}

void HipIntegrateLangevinStepKernel::initialize(const System& system, const LangevinIntegrator& integrator) {
    data.integratorSeed = data.forcedSeed != 0 ? data.forcedSeed : resolveSeed(integrator.getRandomNumberSeed());          // (forcedSeed: the ranks of a device list share one)
    data.hip->forcesRecomputedEveryStep = true;
}
void HipIntegrateLangevinStepKernel::launchStep(ContextImpl& context, const LangevinIntegrator& integrator, long long stepIndex) {
    // ReferenceStochasticDynamics.cpp:89-194
    HipContext& hip = *data.hip;
    const double dt = integrator.getStepSize(), friction = integrator.getFriction(), kT = BOLTZ * integrator.getTemperature();
    ommhip_integrator_state s;
    fillState(s, dt);
    s.step = (unsigned long long) stepIndex;
    s.seed = data.integratorSeed;
    s.vscale = exp(-dt * friction);
    s.fscale = friction == 0 ? dt : (1 - s.vscale) / friction;
    s.noisescale = sqrt(kT * (1 - s.vscale * s.vscale));
    HipConstraints& constraints = data.getDeviceConstraints(context.getSystem());
    if (constraints.fusedStepAvailable()) constraints.fusedStep(OMMHIP_INTEGRATOR_LANGEVIN, s, integrator.getConstraintTolerance());
    else if (hip.decomposed()) throw OpenMMException("HIP platform: multi-GPU runs need constraints that form SETTLE waters or X-H clusters (no CCMA)");
    else {
        HIP_CHECK(ommhip_integrate_stage(OMMHIP_STAGE_LANGEVIN_1, &s, hip.stream));
        if (constraints.hasConstraints()) constraints.apply(hip.xp.ptr, integrator.getConstraintTolerance());
        HIP_CHECK(ommhip_integrate_stage(OMMHIP_STAGE_FINISH_POSITIONS, &s, hip.stream));
        hip.momentumValid = false;
    }
}
void HipIntegrateLangevinStepKernel::execute(ContextImpl& context, const LangevinIntegrator& integrator) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    const std::function<void(long long)> launch = [this, &context, &integrator](long long index) { launchStep(context, integrator, index); };
    hip.replaySteps = [this, &context, &integrator, launch](int steps) { runSteps(context, integrator, launch, (long long) data.stepCount - steps, data.stepCount, false); };
    // the forces just evaluated belong to the oldest step not done yet: this one, or the first of those the device skipped
    const int skipped = hip.pendingReplay;
    hip.pendingReplay = 0;
    runSteps(context, integrator, launch, (long long) data.stepCount - skipped, (long long) data.stepCount + 1, true);
    finishStep(integrator.getStepSize());
}
double HipIntegrateLangevinStepKernel::computeKineticEnergy(ContextImpl& context, const LangevinIntegrator& integrator) {
    return kineticEnergy(0.5 * integrator.getStepSize());
}

void HipIntegrateLangevinMiddleStepKernel::initialize(const System& system, const LangevinMiddleIntegrator& integrator) {
    data.integratorSeed = data.forcedSeed != 0 ? data.forcedSeed : resolveSeed(integrator.getRandomNumberSeed());
    data.hip->forcesRecomputedEveryStep = true;
    // the kinetic energy of this integrator is that of the velocities as they stand (ReferenceKernels.cpp:2450-2453, time shift 0): an
    // energy evaluation can sum it behind its own kernels (HipContext::reduceEnergy), one host round trip per Context::getState(Energy)
    HipPlatform::PlatformData* const d = &data;
    data.hip->prefetchKineticEnergy = [d]() -> bool {
        HipContext& hip = *d->hip;
        HipConstraints& constraints = d->getDeviceConstraints(*d->system);
        if (constraints.hasConstraints() && !hip.velocitiesConstrained) return false;
        if (getenv("OPENMM_HIP_KE_ALWAYS_CONSTRAIN") != NULL) return false;
        return ommhip_kinetic_energy(hip.vel.ptr, NULL, 0, hip.numAtoms, hip.energyResult.as<double>() + 8, hip.energyResult.as<double>() + 1, hip.stream) == 0;
    };
}
void HipIntegrateLangevinMiddleStepKernel::launchStep(ContextImpl& context, const LangevinMiddleIntegrator& integrator, long long stepIndex) {
    // ReferenceLangevinMiddleDynamics.cpp:92-127
    HipContext& hip = *data.hip;
    const double dt = integrator.getStepSize(), friction = integrator.getFriction(), kT = BOLTZ * integrator.getTemperature();
    const double tol = integrator.getConstraintTolerance();
    ommhip_integrator_state s;
    fillState(s, dt);
    s.step = (unsigned long long) stepIndex;
    s.seed = data.integratorSeed;
    s.vscale = exp(-dt * friction);
    s.noisescale = sqrt(kT * (1 - s.vscale * s.vscale));
    HipConstraints& constraints = data.getDeviceConstraints(context.getSystem());
    if (constraints.fusedStepAvailable()) constraints.fusedStep(OMMHIP_INTEGRATOR_LANGEVIN_MIDDLE, s, tol);
    else if (hip.decomposed()) throw OpenMMException("HIP platform: multi-GPU runs need constraints that form SETTLE waters or X-H clusters (no CCMA)");
    else {
        HIP_CHECK(ommhip_integrate_stage(OMMHIP_STAGE_LMIDDLE_1, &s, hip.stream));
        if (constraints.hasConstraints()) constraints.applyToVelocities(hip.vel.ptr, tol);
        HIP_CHECK(ommhip_integrate_stage(OMMHIP_STAGE_LMIDDLE_2, &s, hip.stream));
        if (constraints.hasConstraints()) constraints.apply(hip.xp.ptr, tol);
        HIP_CHECK(ommhip_integrate_stage(OMMHIP_STAGE_LMIDDLE_3, &s, hip.stream));
        hip.momentumValid = false;
    }
}
void HipIntegrateLangevinMiddleStepKernel::execute(ContextImpl& context, const LangevinMiddleIntegrator& integrator) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    const std::function<void(long long)> launch = [this, &context, &integrator](long long index) { launchStep(context, integrator, index); };
    hip.replaySteps = [this, &context, &integrator, launch](int steps) { runSteps(context, integrator, launch, (long long) data.stepCount - steps, data.stepCount, false); };
    // the forces just evaluated belong to the oldest step not done yet: this one, or the first of those the device skipped
    const int skipped = hip.pendingReplay;
    hip.pendingReplay = 0;
    runSteps(context, integrator, launch, (long long) data.stepCount - skipped, (long long) data.stepCount + 1, true);
    finishStep(integrator.getStepSize());
}
double HipIntegrateLangevinMiddleStepKernel::computeKineticEnergy(ContextImpl& context, const LangevinMiddleIntegrator& integrator) {
    return kineticEnergy(0.0);
}

// ================================================================================================
// CMMotionRemover
// ================================================================================================
void HipRemoveCMMotionKernel::initialize(const System& system, const CMMotionRemover& force) {
    frequency = force.getFrequency();
    data.hip->setAsCurrent();
    scratch.allocate(sizeof(double) * 4 * 64);
}
void HipRemoveCMMotionKernel::execute(ContextImpl& context) {
    HipContext& hip = *data.hip;
    // a step that is redone after a neighbour-list overflow carries the index it had the first time: the removal falls on the
    // same steps as in an undisturbed run, not on all or none of the replayed ones
    if ((hip.currentStepIndex >= 0 ? hip.currentStepIndex : (long long) data.stepCount) % frequency != 0) return;
    hip.setAsCurrent();
    // The fused step (one launch per step) already holds the momentum of the current velocities and subtracts the
    // centre-of-mass velocity itself; velocities play no role in the force evaluation in between.
    if (!hip.hostMode && hip.momentumValid && data.getDeviceConstraints(context.getSystem()).fusedStepAvailable()) {
        hip.cmRemovalPending = true;
        return;
    }
    if (hip.decomposed()) hip.gatherState();     // all velocities on every rank; each removes the same centre-of-mass motion
    HIP_CHECK(ommhip_remove_cm_motion(hip.vel.ptr, hip.numAtoms, scratch.as<double>(), hip.stream));
    hip.noteStateMutation();
    hip.momentumValid = false;
}

// ================================================================================================
// MonteCarloBarostat
// ================================================================================================
void HipApplyMonteCarloBarostatKernel::initialize(const System& system, const Force& barostat) {
}
void HipApplyMonteCarloBarostatKernel::scaleCoordinates(ContextImpl& context, double scaleX, double scaleY, double scaleZ) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    if (numMolecules == 0) {
        // molecules as ContextImpl finds them (bond- and constraint-connected atoms), as ReferenceKernels.cpp does on first use
        const vector<vector<int> >& molecules = context.getMolecules();
        vector<int> start(1, 0), atoms;
        for (size_t m = 0; m < molecules.size(); m++) {
            atoms.insert(atoms.end(), molecules[m].begin(), molecules[m].end());
            start.push_back((int) atoms.size());
        }
        numMolecules = (int) molecules.size();
        uploadVector(molStart, start, hip.stream);
        uploadVector(molAtoms, atoms, hip.stream);
        savedPos.allocate(hip.pos.bytes);
    }
    hip.recoverIfFrozen();
    // One box on several GPUs: every rank scales ALL atoms, from the owners' exact positions -- the same doubles, the same result on
    // every rank; the re-sort that follows (the box changes) cuts the slabs and sections for the new box.  The trial energies are sums
    // over the ranks formed in rank order on every rank, and the barostat's random numbers come from the force's own seed: all ranks
    // take the same decision.
    if (hip.decomposed()) hip.gatherState();
    HIP_CHECK(ommhip_memcpy_d2d(savedPos.ptr, hip.pos.ptr, hip.pos.bytes, hip.stream));
    HIP_CHECK(ommhip_scale_molecule_centers(numMolecules, molStart.as<int>(), molAtoms.as<int>(), hip.pos.ptr, hip.box, scaleX, scaleY, scaleZ, hip.stream));
    hip.requestReorder();          // wrap counts and slot order refer to the old box
}
void HipApplyMonteCarloBarostatKernel::restoreCoordinates(ContextImpl& context) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    HIP_CHECK(ommhip_memcpy_d2d(hip.pos.ptr, savedPos.ptr, hip.pos.bytes, hip.stream));
    hip.requestReorder();
}

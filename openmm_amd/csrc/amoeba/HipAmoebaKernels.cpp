#include "HipAmoebaKernels.h"
#include "openmm/AmoebaVdwForce.h"
#include "openmm/AmoebaMultipoleForce.h"
#include "openmm/NonbondedForce.h"
#include "openmm/internal/AmoebaVdwForceImpl.h"
#include "openmm/internal/NonbondedForceImpl.h"
#include "openmm/internal/ContextImpl.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <sstream>
#include <string>
#include <map>
#include <set>
#include <vector>

using namespace OpenMM;
using namespace std;

namespace {
template <class T>
void uploadVector(DeviceBuffer& buffer, const vector<T>& v, void* stream) {
    buffer.allocate(sizeof(T) * max(v.size(), (size_t) 1));
    if (!v.empty()) HIP_CHECK(ommhip_memcpy_h2d(buffer.ptr, v.data(), sizeof(T) * v.size(), stream));
    HIP_CHECK(ommhip_stream_sync(stream));       // v may be a temporary
}
long long nativeEvaluations[2] = {0, 0};
long long solverIterations[2] = {0, 0};      // mutual-polarization solves and their iterations, summed
// Conjugate gradients stop on the residual of the dipoles they return; the Reference's iteration stops on the size of an update it has
// already applied, so at the same nominal epsilon its dipoles are the better converged ones (profiles/r11/reference_platform_at_run_epsilon.txt:
// 2.1e-5 of the RMS force at 1e-5 D on DHFR).  The solver therefore aims at a fraction of the force's mutualInducedTargetEpsilon
// (OPENMM_HIP_AMOEBA_EPSILON_SCALE, A/B in profiles/r11) and finishes with the update mu += alpha r (k_mp_cg stage 6).
double solverTargetScale() {
    static const double scale = getenv("OPENMM_HIP_AMOEBA_EPSILON_SCALE") != NULL ? atof(getenv("OPENMM_HIP_AMOEBA_EPSILON_SCALE")) : 1.0;
    return scale > 0 ? scale : 1.0;
}
int listBuilds[2] = {0, 0};        // pair-list builds of the most recently used vdW / multipole kernel (written by the kernel library at every call)
// Verlet skin of the AMOEBA pair lists (nm): the lists reach this far beyond the cutoff and are rebuilt when an atom has moved by half of it.
// A wider skin means fewer rebuilds and more list entries for every pair kernel to skip: at 1 fs steps of liquid water the optimum is broad
// around 0.04-0.06 nm (a rebuild every 5-8 steps, +20 % entries).  0 = rebuild at every evaluation (the behaviour up to round 3).
double listSkin() {
    static const double skin = getenv("OPENMM_HIP_AMOEBA_SKIN") != NULL ? atof(getenv("OPENMM_HIP_AMOEBA_SKIN")) : 0.05;
    return skin > 0.0 ? skin : 0.0;
}
}

/* Diagnostics for the tests: how many force evaluations went through the native kernels ([0] vdW, [1] multipole) -- a Context that
 * silently fell back to the AMOEBA plugin's Reference kernels leaves these at zero. */
extern "C" __attribute__((visibility("default"))) void ommhip_amoeba_native_evaluations(long long* out) {
    out[0] = nativeEvaluations[0]; out[1] = nativeEvaluations[1];
}
/* ... and how often the pair lists of the most recently used kernels were built ([0] vdW, [1] multipole): with the Verlet skin fewer than evaluations. */
/* ... and the number of mutual-polarization solves and the sum of their iterations. */
extern "C" __attribute__((visibility("default"))) void ommhip_amoeba_solver_iterations(long long* out) {
    out[0] = solverIterations[0]; out[1] = solverIterations[1];
}
extern "C" __attribute__((visibility("default"))) void ommhip_amoeba_list_builds(long long* out) {
    out[0] = listBuilds[0]; out[1] = listBuilds[1];
}

// ================================================================================================
// AmoebaVdwForce
// ================================================================================================
HipCalcAmoebaVdwForceKernel::~HipCalcAmoebaVdwForceKernel() {
    if (earlyId >= 0) data.hip->unregisterEarlyWork(earlyId);
}

void HipCalcAmoebaVdwForceKernel::initialize(const System& system, const AmoebaVdwForce& force) {
    // AmoebaReferenceKernels.cpp:656-667
    numParticles = system.getNumParticles();
    if (force.getNonbondedMethod() != AmoebaVdwForce::NoCutoff && force.getNonbondedMethod() != AmoebaVdwForce::CutoffPeriodic)
        throw OpenMMException("HIP platform: AmoebaVdwForce supports NoCutoff and CutoffPeriodic");
    usePBC = force.getNonbondedMethod() == AmoebaVdwForce::CutoffPeriodic;
    cutoff = force.getCutoffDistance();
    dispersionCoefficient = force.getUseDispersionCorrection() ? AmoebaVdwForceImpl::calcDispersionCorrection(system, force) : 0.0;
    if (usePBC) { data.hip->usePeriodic = true; data.hip->sortCutoff = max(data.hip->sortCutoff, cutoff); }      // spatial slot order: the pair lists are built tile by tile
    upload(force);
    if (earlyId < 0 && getenv("OPENMM_HIP_AMOEBA_VDW_MAIN_STREAM") == NULL)          // (A/B knob: the evaluation stays where execute() finds it, on the main stream)
        earlyId = data.hip->registerEarlyWork(force.getForceGroup(), [this](ContextImpl& c, bool f, bool e) { launch(c, f, e); });
}

void HipCalcAmoebaVdwForceKernel::upload(const AmoebaVdwForce& force) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    // AmoebaReferenceVdwForce::initialize (AmoebaReferenceVdwForce.cpp:40-67)
    vector<int> particleType;
    vector<vector<double> > sigmaMatrix, epsilonMatrix;
    AmoebaVdwForceImpl::createParameterMatrix(force, particleType, sigmaMatrix, epsilonMatrix);
    const int numTypes = (int) sigmaMatrix.size();
    vector<double> sig((size_t) numTypes * numTypes), eps((size_t) numTypes * numTypes);
    for (int i = 0; i < numTypes; i++)
        for (int j = 0; j < numTypes; j++) { sig[(size_t) i * numTypes + j] = sigmaMatrix[i][j]; eps[(size_t) i * numTypes + j] = epsilonMatrix[i][j]; }
    vector<int> parents(numParticles), start(numParticles + 1, 0), flat;
    vector<double> reductions(numParticles);
    vector<unsigned char> alch(numParticles);
    for (int i = 0; i < numParticles; i++) {
        int type;
        double sigma, epsilon;
        bool isAlchemical;
        force.getParticleParameters(i, parents[i], sigma, epsilon, reductions[i], isAlchemical, type);
        alch[i] = isAlchemical ? 1 : 0;
        vector<int> exclusions;
        force.getParticleExclusions(i, exclusions);
        set<int> sorted(exclusions.begin(), exclusions.end());        // ascending, each once: the kernel consumes the row with one cursor
        flat.insert(flat.end(), sorted.begin(), sorted.end());
        start[i + 1] = (int) flat.size();
    }
    uploadVector(parent, parents, hip.stream); uploadVector(reduction, reductions, hip.stream); uploadVector(type, particleType, hip.stream);
    uploadVector(sigma, sig, hip.stream); uploadVector(epsilon, eps, hip.stream);
    uploadVector(exclStart, start, hip.stream); uploadVector(exclAtoms, flat, hip.stream); uploadVector(alchemical, alch, hip.stream);
    reduced.allocate(sizeof(double) * 4 * (size_t) max(numParticles, 1));
    memset(&vdw, 0, sizeof(vdw));
    vdw.num_atoms = numParticles; vdw.num_types = numTypes;
    vdw.parent = parent.as<int>(); vdw.reduction = reduction.as<double>(); vdw.type = type.as<int>();
    vdw.sigma = sigma.as<double>(); vdw.epsilon = epsilon.as<double>();
    vdw.excl_start = exclStart.as<int>(); vdw.excl_atoms = exclAtoms.as<int>(); vdw.alchemical = alchemical.as<unsigned char>();
    vdw.alchemical_method = force.getAlchemicalMethod() == AmoebaVdwForce::Decouple ? 1 : (force.getAlchemicalMethod() == AmoebaVdwForce::Annihilate ? 2 : 0);
    softcorePower = force.getSoftcorePower(); softcoreAlpha = force.getSoftcoreAlpha();
    vdw.lennard_jones = force.getPotentialFunction() == AmoebaVdwForce::LennardJones ? 1 : 0;
    vdw.periodic = usePBC ? 1 : 0;
    vdw.mixed_precision = getenv("OPENMM_HIP_AMOEBA_PRECISION") != NULL && string(getenv("OPENMM_HIP_AMOEBA_PRECISION")) == "double" ? 0 : 1;
    // AmoebaReferenceVdwForce::setTaperCoefficients (:69-81), taper between 0.9 cutoff and the cutoff
    vdw.cutoff = cutoff; vdw.taper_cutoff = 0.9 * cutoff;
    if (usePBC && vdw.taper_cutoff != cutoff) {
        vdw.taper_c3 = 10.0 / pow(vdw.taper_cutoff - cutoff, 3.0);
        vdw.taper_c4 = 15.0 / pow(vdw.taper_cutoff - cutoff, 4.0);
        vdw.taper_c5 = 6.0 / pow(vdw.taper_cutoff - cutoff, 5.0);
    }
    vdw.reduced = reduced.as<double>();
    // CutoffPeriodic: per-atom pair lists in the platform's slot order, rebuilt at every evaluation (amoeba_pairs.h)
    if (usePBC) {
        tileBounds.allocate(sizeof(double) * 4 * 2 * max(((size_t) hip.paddedAtoms + 127) / 128, (size_t) 1));
        exclPos.allocate(sizeof(int) * max(flat.size(), (size_t) 1));
        pairCount.allocate(sizeof(int) * 4 * (size_t) hip.paddedAtoms);
        pairOverflow.allocate(sizeof(int));
        vdw.tile_bounds = tileBounds.as<double>(); vdw.excl_pos = exclPos.as<int>(); vdw.pair_count = pairCount.as<int>(); vdw.pair_overflow = pairOverflow.as<int>();
        vdw.pair_needed = &pairNeeded;
        vdw.skin = listSkin();
        refPos.allocate(sizeof(double) * 4 * (size_t) max(numParticles, 1));
        if (listState.ptr == NULL) { listState.allocate(sizeof(int) * 4); HIP_CHECK(ommhip_memset(listState.ptr, 0, listState.bytes, hip.stream)); }
        vdw.ref_pos = refPos.as<double>(); vdw.list_state = listState.as<int>(); vdw.force_rebuild = 1; vdw.list_builds = &listBuilds[0];
        listDirty = true;
        // capacity: the partners of an atom at the density of the box, with room for fluctuations; a list that does not fit grows it
        const double volume = hip.box[0] * hip.box[2] * hip.box[5];
        // (OPENMM_HIP_AMOEBA_PAIR_CAP: a deliberately small first capacity, for the test of the growth path)
        // (updateParametersInContext comes through here again: a capacity the lists have grown to is kept)
        allocatePairList(max(grownPairCap, getenv("OPENMM_HIP_AMOEBA_PAIR_CAP") != NULL ? atoi(getenv("OPENMM_HIP_AMOEBA_PAIR_CAP")) : (int) min(4.0 * numParticles, 2.0 * 4.18879 * pow(cutoff + listSkin(), 3.0) * numParticles / volume + 128.0)));
    }
}

void HipCalcAmoebaVdwForceKernel::allocatePairList(int cap) {
    HipContext& hip = *data.hip;
    vdw.pair_cap = (max(cap, 4) + 3) / 4 * 4;            // four sub-lists per atom
    listDirty = true;
    pairList.allocate(sizeof(int) * (size_t) vdw.pair_cap * hip.paddedAtoms);
    vdw.pair_list = pairList.as<int>();
}

double HipCalcAmoebaVdwForceKernel::execute(ContextImpl& context, bool includeForces, bool includeEnergy) {
    HipContext& hip = *data.hip;
    if (earlyId < 0 || !hip.earlyWorkLaunched(earlyId)) {
        if (earlyId >= 0) hip.noteEarlyWorkLaunched(earlyId);
        launch(context, includeForces, includeEnergy);
    }
    // the pair energy is summed on the device (HipCalcForcesAndEnergyKernel::finishComputation); the host adds the constant
    return includeEnergy && usePBC ? dispersionCoefficient / (hip.box[0] * hip.box[2] * hip.box[5]) : 0.0;
}

void HipCalcAmoebaVdwForceKernel::launch(ContextImpl& context, bool includeForces, bool includeEnergy) {
    // AmoebaReferenceKernels.cpp:669-690
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    const double lambda = context.getParameter(AmoebaVdwForce::Lambda());
    vdw.epsilon_scale = pow(lambda, softcorePower);
    vdw.softcore = softcoreAlpha * (1.0 - lambda) * (1.0 - lambda);
    if (usePBC) {
        const double minAllowedSize = 1.999999 * cutoff;
        if (hip.box[0] < minAllowedSize || hip.box[2] < minAllowedSize || hip.box[5] < minAllowedSize)
            throw OpenMMException("The periodic box size has decreased to less than twice the cutoff.");
    }
    hip.ensureCleared();
    vdw.atom_of_slot = hip.atomOfSlot.as<int>();
    // on the side stream (it waits for what the main stream holds so far: the clear of the force buffer); the main stream joins it in finishComputation
    void* stream = earlyId >= 0 ? hip.pmeStream : hip.stream;
    if (earlyId >= 0) hip.forkPme();
    for (int attempt = 0; ; attempt++) {
        vdw.force_rebuild = listDirty || listOrderVersion != hip.orderVersion || listBoxVersion != hip.boxVersion ? 1 : 0;
        const int rc = ommhip_amoeba_vdw_forces(&vdw, hip.pos.ptr, hip.box, hip.slotOfAtom.as<int>(), hip.paddedAtoms, hip.force.as<long long>(),
                                                hip.energyBuffer.as<double>(), HipContext::EnergySlots, includeEnergy ? 1 : 0, stream);
        if (rc != -2) { HIP_CHECK(rc); listDirty = false; listOrderVersion = hip.orderVersion; listBoxVersion = hip.boxVersion; break; }
        // the pair lists did not fit (nothing has been added to the forces yet: the list is built before the pair kernel runs)
        if (attempt == 3 || pairNeeded == 0x7fffffff) throw OpenMMException("AmoebaVdwForce: the pair lists of the HIP platform cannot hold this System");
        grownPairCap = (int) min((long long) numParticles * 4, (long long) pairNeeded * 5 / 4 + 16);
        allocatePairList(grownPairCap);
    }
    if (earlyId >= 0) hip.markPmeDone();
    nativeEvaluations[0]++;
}

void HipCalcAmoebaVdwForceKernel::copyParametersToContext(ContextImpl& context, const AmoebaVdwForce& force) {
    if (numParticles != force.getNumParticles())
        throw OpenMMException("updateParametersInContext: The number of particles has changed");
    upload(force);
}

// ================================================================================================
// AmoebaMultipoleForce (PME; direct, mutual and extrapolated polarization)
// ================================================================================================
namespace {
vector<double> bsplineModuli(int n) {
    // order-5 B-spline moduli, ReferencePME.cpp:98-193 (the convention of the platform's spread / FFT / influence-function kernels,
    // which the AMOEBA kernels share): B-spline values at the knots, |DFT|^2, small moduli replaced by the mean of their neighbours
    const int order = 5;
    double data[order] = {1, 0, 0, 0, 0};
    for (int k = 3; k <= order; k++) {
        const double div = 1.0 / (k - 1.0);
        data[k - 1] = 0;
        for (int l = 1; l < k - 1; l++) data[k - l - 1] = div * (l * data[k - l - 2] + (k - l) * data[k - l - 1]);
        data[0] = div * data[0];
    }
    vector<double> bs(max(n, order + 1), 0.0), mod(n);
    for (int i = 1; i <= order; i++) bs[i] = data[i - 1];
    for (int i = 0; i < n; i++) {
        double sc = 0, ss = 0;
        for (int j = 0; j < n; j++) {
            const double arg = (2.0 * M_PI * i * j) / n;
            sc += bs[j] * cos(arg);
            ss += bs[j] * sin(arg);
        }
        mod[i] = sc * sc + ss * ss;
    }
    for (int i = 0; i < n; i++)
        if (mod[i] < 1.0e-7) mod[i] = (mod[(i - 1 + n) % n] + mod[(i + 1) % n]) / 2;
    // AMOEBA (Tinker) additionally applies the "optimal zeta" correction of the interpolation error to every modulus
    // (AmoebaReferencePmeMultipoleForce::initializeBSplineModuli, AmoebaReferenceMultipoleForce.cpp:5048-5075)
    const int jcut = 50;
    for (int i = 0; i < n; i++) {
        const int k = i + 1 > n / 2 ? i - n : i;
        if (k == 0) continue;
        double sum1 = 1.0, sum2 = 1.0;
        const double factor = M_PI * k / n;
        for (int j = 1; j <= jcut; j++) {
            const double up = factor / (factor + M_PI * j), down = factor / (factor - M_PI * j);
            sum1 += pow(up, order) + pow(down, order);
            sum2 += pow(up, 2 * order) + pow(down, 2 * order);
        }
        const double zeta = sum2 / sum1;
        mod[i] *= zeta * zeta;
    }
    return mod;
}
}

HipCalcAmoebaMultipoleForceKernel::HipCalcAmoebaMultipoleForceKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data, KernelImpl* referenceKernel) :
        CalcAmoebaMultipoleForceKernel(name, platform), data(data), reference(dynamic_cast<CalcAmoebaMultipoleForceKernel*>(referenceKernel)) {
    memset(&mp, 0, sizeof(mp));
    memset(&pme, 0, sizeof(pme));
}

HipCalcAmoebaMultipoleForceKernel::~HipCalcAmoebaMultipoleForceKernel() {
    delete reference;
    if (sideStream != NULL) { data.hip->setAsCurrent(); ommhip_stream_destroy(sideStream); ommhip_event_destroy(eventA); ommhip_event_destroy(eventB); }
}

bool HipCalcAmoebaMultipoleForceKernel::supports(const AmoebaMultipoleForce& force, const System& system) {
    if (getenv("OPENMM_HIP_REFERENCE_AMOEBA_MULTIPOLE") != NULL) return false;          // A/B knob: always the Reference kernel
    if (force.getNonbondedMethod() != AmoebaMultipoleForce::PME) return false;
    if (force.getPolarizationType() != AmoebaMultipoleForce::Direct && force.getPolarizationType() != AmoebaMultipoleForce::Mutual &&
            force.getPolarizationType() != AmoebaMultipoleForce::Extrapolated) return false;
    if (force.getPolarizationType() == AmoebaMultipoleForce::Extrapolated &&
            (force.getExtrapolationCoefficients().empty() || force.getExtrapolationCoefficients().size() > OMMHIP_AMOEBA_MAX_EXT_ORDERS || getenv("OPENMM_HIP_REFERENCE_AMOEBA_EXTRAPOLATED") != NULL)) return false;
    double alpha; int nx, ny, nz;
    force.getPMEParameters(alpha, nx, ny, nz);
    if (nx == 0 || alpha == 0.0) {
        NonbondedForce nb;
        nb.setEwaldErrorTolerance(force.getEwaldErrorTolerance());
        nb.setCutoffDistance(force.getCutoffDistance());
        NonbondedForceImpl::calcPMEParameters(system, nb, alpha, nx, ny, nz, false);
    }
    // the Reference transforms whatever grid it is given (fftpack); the platform's FFT wants 2-3-5-7-smooth sizes, even along z
    return ommhip_fft_supported_size(nx) && ommhip_fft_supported_size(ny) && ommhip_fft_supported_size(nz) && nz % 2 == 0;
}

void HipCalcAmoebaMultipoleForceKernel::initialize(const System& system, const AmoebaMultipoleForce& force) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    numParticles = force.getNumMultipoles();
    if (numParticles != system.getNumParticles())
        throw OpenMMException("AmoebaMultipoleForce must have exactly as many particles as the System it belongs to.");
    // AmoebaReferenceKernels.cpp:241-262
    force.getPMEParameters(alphaEwald, gridSize[0], gridSize[1], gridSize[2]);
    cutoff = force.getCutoffDistance();
    if (gridSize[0] == 0 || alphaEwald == 0.0) {
        NonbondedForce nb;
        nb.setEwaldErrorTolerance(force.getEwaldErrorTolerance());
        nb.setCutoffDistance(force.getCutoffDistance());
        NonbondedForceImpl::calcPMEParameters(system, nb, alphaEwald, gridSize[0], gridSize[1], gridSize[2], false);
    }
    hip.usePeriodic = true;
    hip.sortCutoff = max(hip.sortCutoff, cutoff);         // spatial slot order: the pair lists are built tile by tile
    if (reference != NULL) reference->initialize(system, force);
    // ---- the platform's PME machinery on a grid of its own
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
    gridReal.allocate((sizeof(float) * (size_t) nx * ny * nz + 15) / 16 * 16);
    gridComplex.allocate(sizeof(float) * 2 * (size_t) nx * ny * nzc);
    pme.nx = nx; pme.ny = ny; pme.nz = nz; pme.alpha = alphaEwald;
    pme.moduli_x = moduliX.as<double>(); pme.moduli_y = moduliY.as<double>(); pme.moduli_z = moduliZ.as<double>();
    pme.eterm = eterm.ptr; pme.grid_real = gridReal.ptr; pme.grid_complex = gridComplex.ptr;
    pme.twiddle_x = twiddleX.ptr; pme.twiddle_y = twiddleY.ptr; pme.twiddle_z = twiddleZ.ptr;
    etermBuilt = false;
    const size_t n = (size_t) max(numParticles, 1);
    labDipole.allocate(sizeof(double) * 3 * n); labQuad.allocate(sizeof(double) * 6 * n);
    fieldD.allocate(sizeof(double) * 3 * n); fieldP.allocate(sizeof(double) * 3 * n);
    indD.allocate(sizeof(double) * 3 * n); indP.allocate(sizeof(double) * 3 * n);
    phi.allocate(sizeof(double) * 20 * n); phiInd.allocate(sizeof(double) * 20 * n); torque.allocate(sizeof(double) * 3 * n);
    mutual = force.getPolarizationType() == AmoebaMultipoleForce::Mutual;
    extrapolated = force.getPolarizationType() == AmoebaMultipoleForce::Extrapolated;
    if (extrapolated) {
        const size_t K = force.getExtrapolationCoefficients().size();
        phiIndP.allocate(sizeof(double) * 20 * n); solver.allocate(sizeof(double) * (24 * n + 16));
        extDipoles.allocate(sizeof(double) * 6 * n * K); extGradients.allocate(sizeof(double) * 12 * n * max(K - 1, (size_t) 1));
    }
    if (mutual) {
        phiIndP.allocate(sizeof(double) * 20 * n); solver.allocate(sizeof(double) * (24 * n + 16));
        if (getenv("OPENMM_HIP_AMOEBA_NO_PREDICTOR") == NULL) history.allocate(sizeof(double) * 6 * n * HistorySlots);       // (A/B knob: every solve starts from the direct dipoles)
    }
    upload(force);
}

void HipCalcAmoebaMultipoleForceKernel::upload(const AmoebaMultipoleForce& force) {
    HipContext& hip = *data.hip;
    // AmoebaReferenceKernels.cpp:170-231 (per-particle data) and AmoebaReferenceMultipoleForce::setupScaleMaps (:181-241)
    vector<double> q(numParticles), d(3 * (size_t) numParticles), quad(6 * (size_t) numParticles), th(numParticles), dampF(numParticles), pol(numParticles);
    vector<int> ax(4 * (size_t) numParticles);
    vector<vector<vector<int> > > covalent(numParticles);
    for (int i = 0; i < numParticles; i++) {
        int axisType, atomZ, atomX, atomY;
        vector<double> dip, qd;
        force.getMultipoleParameters(i, q[i], dip, qd, axisType, atomZ, atomX, atomY, th[i], dampF[i], pol[i]);
        for (int k = 0; k < 3; k++) d[3 * i + k] = dip[k];
        quad[6 * i] = qd[0]; quad[6 * i + 1] = qd[1]; quad[6 * i + 2] = qd[2]; quad[6 * i + 3] = qd[4]; quad[6 * i + 4] = qd[5]; quad[6 * i + 5] = qd[8];
        ax[4 * i] = axisType; ax[4 * i + 1] = atomZ; ax[4 * i + 2] = atomX; ax[4 * i + 3] = atomY;
        force.getCovalentMaps(i, covalent[i]);
    }
    // scale factors of the covalently related pairs: (m, p, d, u), defined by the lists of the atom with the lower index (as the Reference
    // looks them up), stored for both atoms of a pair
    const double mScale[5] = {0.0, 0.0, 0.0, 0.4, 0.8}, pScale[5] = {0.0, 0.0, 0.0, 1.0, 1.0}, dScale[4] = {0.0, 1.0, 1.0, 1.0}, uScale[4] = {1.0, 1.0, 1.0, 1.0};
    vector<map<int, vector<double> > > special(numParticles);
    for (int i = 0; i < numParticles; i++) {
        const vector<vector<int> >& info = covalent[i];
        const vector<int>& p11 = info[AmoebaMultipoleForce::PolarizationCovalent11];
        for (int list = 0; list < AmoebaMultipoleForce::PolarizationCovalent11; list++)
            for (int j : info[list]) {
                if (j < i) continue;
                const bool half = list == AmoebaMultipoleForce::Covalent14 && find(p11.begin(), p11.end(), j) != p11.end();
                vector<double>& s = special[i].insert(make_pair(j, vector<double>(4, 1.0))).first->second;
                s[0] = mScale[list + 1]; s[1] = half ? 0.5 * pScale[list + 1] : pScale[list + 1];
            }
        for (int list = AmoebaMultipoleForce::PolarizationCovalent11; list < (int) info.size(); list++)
            for (int j : info[list]) {
                if (j < i) continue;
                vector<double>& s = special[i].insert(make_pair(j, vector<double>(4, 1.0))).first->second;
                s[2] = dScale[list - 4]; s[3] = uScale[list - 4];
            }
    }
    for (int i = 0; i < numParticles; i++)
        for (map<int, vector<double> >::const_iterator it = special[i].begin(); it != special[i].end(); ++it)
            if (it->first > i) special[it->first][i] = it->second;
    vector<int> start(numParticles + 1, 0), atoms;
    vector<double> scales;
    for (int i = 0; i < numParticles; i++) {
        for (map<int, vector<double> >::const_iterator it = special[i].begin(); it != special[i].end(); ++it) {
            if (it->first == i) continue;
            atoms.push_back(it->first);
            scales.insert(scales.end(), it->second.begin(), it->second.end());
        }
        start[i + 1] = (int) atoms.size();
    }
    uploadVector(charge, q, hip.stream); uploadVector(molDipole, d, hip.stream); uploadVector(molQuad, quad, hip.stream); uploadVector(axis, ax, hip.stream);
    uploadVector(thole, th, hip.stream); uploadVector(damping, dampF, hip.stream); uploadVector(polarity, pol, hip.stream);
    uploadVector(specStart, start, hip.stream); uploadVector(specAtom, atoms, hip.stream); uploadVector(specScale, scales, hip.stream);
    mp.num_atoms = numParticles;
    mp.charge = charge.as<double>(); mp.mol_dipole = molDipole.as<double>(); mp.mol_quadrupole = molQuad.as<double>(); mp.axis = axis.as<int>();
    mp.thole = thole.as<double>(); mp.damping = damping.as<double>(); mp.polarity = polarity.as<double>();
    mp.special_start = specStart.as<int>(); mp.special_atom = specAtom.as<int>(); mp.special_scale = specScale.as<double>();
    mp.cutoff = cutoff; mp.alpha = alphaEwald;
    mp.lab_dipole = labDipole.as<double>(); mp.lab_quadrupole = labQuad.as<double>(); mp.field_d = fieldD.as<double>(); mp.field_p = fieldP.as<double>();
    mp.induced_d = indD.as<double>(); mp.induced_p = indP.as<double>(); mp.phi = phi.as<double>(); mp.phi_induced = phiInd.as<double>(); mp.torque = torque.as<double>();
    mp.pme = &pme;
    mp.mutual = mutual ? 1 : 0;
    // pair arithmetic: float by default ("mixed", what the reference's GPU platforms do in their mixed mode); OPENMM_HIP_AMOEBA_PRECISION=double keeps everything in double
    mp.mixed_precision = getenv("OPENMM_HIP_AMOEBA_PRECISION") != NULL && string(getenv("OPENMM_HIP_AMOEBA_PRECISION")) == "double" ? 0 : 1;
    mp.max_iterations = force.getMutualInducedMaxIterations(); mp.target_epsilon = force.getMutualInducedTargetEpsilon() * solverTargetScale();
    mp.phi_induced_p = mutual || extrapolated ? phiIndP.as<double>() : NULL; mp.solver = mutual || extrapolated ? solver.as<double>() : NULL; mp.status = solverStatus;
    mp.extrapolation_orders = 0; mp.ext_dipoles = NULL; mp.ext_gradients = NULL;
    for (int k = 0; k < OMMHIP_AMOEBA_MAX_EXT_ORDERS; k++) mp.ext_coefficients[k] = 0.0;
    if (extrapolated) {
        const vector<double>& c = force.getExtrapolationCoefficients();
        mp.extrapolation_orders = (int) c.size();
        for (size_t k = 0; k < c.size(); k++) mp.ext_coefficients[k] = c[k];
        mp.ext_dipoles = extDipoles.as<double>(); mp.ext_gradients = extGradients.as<double>();
    }
    mp.solver_gather = NULL;
    if (mutual && getenv("OPENMM_HIP_AMOEBA_NO_GATHER_COPY") == NULL) {        // (A/B knob: the field kernel gathers the atom-ordered doubles)
        solverGather.allocate(sizeof(float) * 6 * (size_t) hip.paddedAtoms);
        HIP_CHECK(ommhip_memset(solverGather.ptr, 0, solverGather.bytes, hip.stream));
        mp.solver_gather = solverGather.as<float>();
    }
    mp.history = mutual && history.ptr != NULL ? history.as<double>() : NULL; mp.history_slots = HistorySlots; mp.history_newest = 0; mp.history_store = -1; mp.history_use = 0; mp.expected_iterations = 0;
    historyValid = 0;                         // new parameters: the earlier solutions belong to another Hamiltonian
    // mutual polarization: a second grid set and a side stream, so that the potentials of the two dipole sets are computed side by side
    mp.pme2 = NULL; mp.stream2 = NULL; mp.event_a = mp.event_b = NULL;
    if (mutual && getenv("OPENMM_HIP_AMOEBA_ONE_GRID") == NULL) {
        const size_t nx = gridSize[0], ny = gridSize[1], nz = gridSize[2], nzc = nz / 2 + 1;
        gridReal2.allocate((sizeof(float) * nx * ny * nz + 15) / 16 * 16);
        gridComplex2.allocate(sizeof(float) * 2 * nx * ny * nzc);
        // (high priority: its chains of small launches -- spreading, transforms, read-back -- run beside the long pair kernels of the main stream
        // and are the longer side of every solver iteration: profiles/r11/r11ao_amoeba_dhfr_timeline.txt)
        if (sideStream == NULL) { HIP_CHECK(ommhip_stream_create_priority(&sideStream, getenv("OPENMM_HIP_AMOEBA_SIDE_NORMAL") == NULL ? 1 : 0)); HIP_CHECK(ommhip_event_create_untimed(&eventA)); HIP_CHECK(ommhip_event_create_untimed(&eventB)); }
        mp.pme2 = &pme2; mp.stream2 = sideStream; mp.event_a = eventA; mp.event_b = eventB;
    }
    // per-atom pair lists in the platform's slot order, rebuilt at every evaluation (amoeba_pairs.h); the order itself is set per evaluation
    const size_t tiles = ((size_t) hip.paddedAtoms + 127) / 128;
    tileBounds.allocate(sizeof(double) * 4 * 2 * max(tiles, (size_t) 1));
    specPos.allocate(sizeof(int) * max(atoms.size(), (size_t) 1));
    specScaleSorted.allocate(sizeof(double) * 4 * max(atoms.size(), (size_t) 1));
    pairCount.allocate(sizeof(int) * 4 * (size_t) hip.paddedAtoms);
    pairOverflow.allocate(sizeof(int));
    mp.tile_bounds = tileBounds.as<double>(); mp.special_pos = specPos.as<int>(); mp.special_scale_sorted = specScaleSorted.as<double>();
    mp.pair_count = pairCount.as<int>(); mp.pair_overflow = pairOverflow.as<int>(); mp.pair_needed = &pairNeeded;
    mp.skin = listSkin();
    refPos.allocate(sizeof(double) * 4 * (size_t) max(numParticles, 1));
    if (listState.ptr == NULL) { listState.allocate(sizeof(int) * 4); HIP_CHECK(ommhip_memset(listState.ptr, 0, listState.bytes, hip.stream)); }
    mp.ref_pos = refPos.as<double>(); mp.list_state = listState.as<int>(); mp.force_rebuild = 1; mp.list_builds = &listBuilds[1];
    listDirty = true;
    mp.atom_of_slot = NULL; mp.slot_of_atom = NULL; mp.scan_slots = 0;
    const double volume = hip.box[0] * hip.box[2] * hip.box[5];
    // (OPENMM_HIP_AMOEBA_PAIR_CAP: a deliberately small first capacity, for the test of the growth path)
    // (updateParametersInContext comes through here again: a capacity the lists have grown to is kept)
    allocatePairList(max(grownPairCap, getenv("OPENMM_HIP_AMOEBA_PAIR_CAP") != NULL ? atoi(getenv("OPENMM_HIP_AMOEBA_PAIR_CAP")) : (int) min(4.0 * numParticles, 2.0 * 4.18879 * pow(cutoff + listSkin(), 3.0) * numParticles / volume + 128.0)));
}

void HipCalcAmoebaMultipoleForceKernel::allocatePairList(int cap) {
    HipContext& hip = *data.hip;
    mp.pair_cap = (max(cap, 4) + 3) / 4 * 4;             // four sub-lists per atom
    listDirty = true;
    pairList.allocate(sizeof(int) * (size_t) mp.pair_cap * hip.paddedAtoms);
    mp.pair_list = pairList.as<int>();
    // mutual polarization: 20 bytes per list entry that save the solver iterations their erfc / exp / Thole arithmetic (0.3 GB at 36 k atoms;
    // left out when it would take more than 8 GB)
    mp.pair_cache = NULL;
    const size_t cacheBytes = sizeof(float) * 5 * (size_t) mp.pair_cap * hip.paddedAtoms;
    if (mutual && cacheBytes <= ((size_t) 8 << 30) && getenv("OPENMM_HIP_AMOEBA_NO_PAIR_CACHE") == NULL) { pairCache.allocate(cacheBytes); mp.pair_cache = pairCache.as<float>(); }
}

bool HipCalcAmoebaMultipoleForceKernel::growPairList(int rc, int attempt) {
    if (rc != -2) return false;
    if (attempt == 3 || pairNeeded == 0x7fffffff) throw OpenMMException("AmoebaMultipoleForce: the pair lists of the HIP platform cannot hold this System");
    grownPairCap = (int) min((long long) numParticles * 4, (long long) pairNeeded * 5 / 4 + 16);
    allocatePairList(grownPairCap);
    return true;
}

void HipCalcAmoebaMultipoleForceKernel::setScanOrder() {
    HipContext& hip = *data.hip;
    mp.atom_of_slot = hip.atomOfSlot.as<int>(); mp.slot_of_atom = hip.slotOfAtom.as<int>(); mp.scan_slots = hip.paddedAtoms;
    // the pair lists are keyed by slots and built for one box: a new order or box (or new parameters / a new capacity: listDirty) asks for a rebuild
    mp.force_rebuild = listDirty || listOrderVersion != hip.orderVersion || listBoxVersion != hip.boxVersion ? 1 : 0;
}

void HipCalcAmoebaMultipoleForceKernel::listBuilt() {
    listDirty = false; listOrderVersion = data.hip->orderVersion; listBoxVersion = data.hip->boxVersion;
}

void HipCalcAmoebaMultipoleForceKernel::prepareGrid() {
    HipContext& hip = *data.hip;
    bool boxChanged = !etermBuilt;
    for (int k = 0; k < 6; k++) if (lastBox[k] != hip.box[k]) boxChanged = true;
    for (int k = 0; k < 6; k++) { pme.box[k] = hip.box[k]; lastBox[k] = hip.box[k]; }
    if (boxChanged) { HIP_CHECK(ommhip_pme_build_eterm(&pme, hip.stream)); etermBuilt = true; }
    pme2 = pme; pme2.grid_real = gridReal2.ptr; pme2.grid_complex = gridComplex2.ptr;          // the twin: same box, tables and influence function, grids of its own
    const double minAllowedSize = 1.999999 * cutoff;
    if (hip.box[0] < minAllowedSize || hip.box[2] < minAllowedSize || hip.box[5] < minAllowedSize)
        throw OpenMMException("The periodic box size has decreased to less than twice the nonbonded cutoff.");
}

double HipCalcAmoebaMultipoleForceKernel::execute(ContextImpl& context, bool includeForces, bool includeEnergy) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    prepareGrid();
    hip.ensureCleared();
    // What other forces have to launch goes before the solver (which waits for the device) -- and, since round 5, AFTER this force's list build and
    // list-free work have been enqueued: the call below hands control back through `after_lists_enqueued` at that point.  The AmoebaVdwForce's
    // launch ends in a host wait for its own list build on the side stream; the two builds now run side by side.  The side stream waits for the
    // fork point recorded here (the cleared force buffer), not for what this force enqueues in between.
    struct EarlyLaunch { HipContext* hip; ContextImpl* context; bool forces, energy; std::exception_ptr error; } early = {&hip, &context, includeForces, includeEnergy, nullptr};
    static const bool earlyFirst = getenv("OPENMM_HIP_AMOEBA_EARLY_FIRST") != NULL;       // A/B: the order before round 5
    if (earlyFirst) hip.launchEarlyWork(context, includeForces, includeEnergy);
    else hip.preparePmeFork();
    mp.after_lists_enqueued = [](void* arg) {
        // (an exception of the other force's launch -- "the periodic box size has decreased ..." -- does not travel through the C ABI: kept, rethrown below)
        EarlyLaunch* e = (EarlyLaunch*) arg;
        try { e->hip->launchEarlyWork(*e->context, e->forces, e->energy); }
        catch (...) { if (!e->error) e->error = std::current_exception(); }
    };
    mp.after_lists_arg = &early;
    setScanOrder();
    chooseFirstGuess();
    int rc;
    for (int attempt = 0; ; attempt++) {
        rc = ommhip_amoeba_multipole_forces(&mp, hip.pos.ptr, hip.box, hip.slotOfAtom.as<int>(), hip.paddedAtoms, hip.force.as<long long>(),
                                            hip.energyBuffer.as<double>(), HipContext::EnergySlots, includeEnergy ? 1 : 0, hip.stream);
        if (!growPairList(rc, attempt)) break;        // -2: the pair lists did not fit (they are built before anything is added to the forces)
        setScanOrder();
    }
    mp.after_lists_enqueued = NULL; mp.after_lists_arg = NULL;
    hip.pmeForkRecorded = false;
    if (early.error) std::rethrow_exception(early.error);
    hip.launchEarlyWork(context, includeForces, includeEnergy);       // (already launched unless the call failed before its hook)
    if (rc == 0 || rc == -1) listBuilt();
    checkSolver(rc);
    recordSolve();
    nativeEvaluations[1]++;
    return 0.0;        // summed on the device (HipCalcForcesAndEnergyKernel::finishComputation)
}

void HipCalcAmoebaMultipoleForceKernel::induce() {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    if (hip.hostMode) {
        // host mode (virtual sites, a custom integrator, ...): the host vectors are authoritative and reach the device only at the start
        // of a force evaluation -- a dipole query right after setPositions / setPeriodicBoxVectors must not see the old geometry
        // (the CUDA platform re-evaluates when its multipoles are stale, CudaCalcAmoebaMultipoleForceKernel::ensureMultipolesValid)
        hip.setBox(data.periodicBoxVectors[0], data.periodicBoxVectors[1], data.periodicBoxVectors[2]);
        hip.uploadPositions(*data.positions);
    }
    prepareGrid();
    setScanOrder();
    chooseFirstGuess();
    int rc;
    for (int attempt = 0; ; attempt++) {
        rc = ommhip_amoeba_multipole_induce(&mp, hip.pos.ptr, hip.box, hip.stream);
        if (!growPairList(rc, attempt)) break;
        setScanOrder();
    }
    if (rc == 0 || rc == -1) listBuilt();
    checkSolver(rc);
    recordSolve();
}

void HipCalcAmoebaMultipoleForceKernel::chooseFirstGuess() {
    // Which earlier solutions may the solver start from?  Those of the immediately preceding steps of an undisturbed run: the step count
    // moved on by one since the newest record, nobody edited positions or box in between (device mode: every uploadPositions is an edit;
    // host mode uploads at every evaluation, there only the step count speaks), the parameters are the same (upload() resets).
    // The same step again (an energy query after the step) starts from the last solution itself.
    HipContext& hip = *data.hip;
    const int newest = (historyNext + HistorySlots - 1) % HistorySlots;
    mp.history_use = 0; mp.history_newest = newest; mp.history_store = historyNext; mp.expected_iterations = (int) solverStatus[1];
    historySameStep = false;
    if (!mutual || mp.history == NULL) return;
    const bool edited = (!hip.hostMode && hip.positionsVersion != historyPositionsVersion) || hip.boxVersion != historyBoxVersion;
    if (edited) historyValid = 0;
    if (historyValid > 0 && (long long) data.stepCount == historyStep) {
        historySameStep = true;
        mp.history_use = 1;
        mp.history_store = newest;              // replaces the newest record
    }
    else if (historyValid > 0 && (long long) data.stepCount == historyStep + 1) mp.history_use = historyValid;
    else historyValid = 0;
    // Coefficients.  One record: the last solution itself.  Consecutive steps: the predictor of Kolafa's always-stable scheme (J. Comput.
    // Chem. 25, 335 (2004); Tinker's polpred ASPC is the same family), c_k = (-1)^(k+1) k C(2m, m - k) / C(2m - 2, m - 1) for m records --
    // it damps the noise that solutions converged to epsilon carry, which plain polynomial extrapolation (OPENMM_HIP_AMOEBA_PREDICTOR=poly)
    // amplifies.  OPENMM_HIP_AMOEBA_PREDICTOR_POINTS limits the number of records used (default 4).
    static const bool poly = getenv("OPENMM_HIP_AMOEBA_PREDICTOR") != NULL && string(getenv("OPENMM_HIP_AMOEBA_PREDICTOR")) == "poly";
    static const int maxPoints = getenv("OPENMM_HIP_AMOEBA_PREDICTOR_POINTS") != NULL ? max(1, min((int) HistorySlots, atoi(getenv("OPENMM_HIP_AMOEBA_PREDICTOR_POINTS")))) : 4;
    if (!historySameStep) mp.history_use = min(mp.history_use, maxPoints);
    for (int k = 0; k < OMMHIP_AMOEBA_MAX_HISTORY; k++) mp.history_coeff[k] = 0.0;
    const int m = mp.history_use;
    if (m == 1) mp.history_coeff[0] = 1.0;
    else if (m > 1) {
        struct Binomial { static double of(int n, int k) { if (k < 0 || k > n) return 0.0; double b = 1.0; for (int i = 1; i <= k; i++) b = b * (n - k + i) / i; return b; } };
        for (int k = 1; k <= m; k++)
            mp.history_coeff[k - 1] = poly ? ((k % 2) ? 1.0 : -1.0) * Binomial::of(m, k)
                                           : ((k % 2) ? 1.0 : -1.0) * k * Binomial::of(2 * m, m - k) / Binomial::of(2 * m - 2, m - 1);
    }
    // iterations are only enqueued ahead of the convergence check when the last solve started from the same kind of guess
    if (mp.history_use != lastHistoryUse) mp.expected_iterations = 0;
    lastHistoryUse = mp.history_use;
}

void HipCalcAmoebaMultipoleForceKernel::recordSolve() {
    if (mutual) { solverIterations[0]++; solverIterations[1] += (long long) solverStatus[1]; }
    if (!mutual || mp.history == NULL) return;
    HipContext& hip = *data.hip;
    if (!historySameStep) {
        historyNext = (historyNext + 1) % HistorySlots;
        historyValid = min(historyValid + 1, (int) HistorySlots);
    }
    historyStep = data.stepCount; historyPositionsVersion = hip.positionsVersion; historyBoxVersion = hip.boxVersion;
}

void HipCalcAmoebaMultipoleForceKernel::checkSolver(int rc) {
    if (rc == -1) {
        // AmoebaReferenceMultipoleForce::setup (AmoebaReferenceMultipoleForce.cpp:1811-1817)
        std::stringstream message;
        message << "Induced dipoles did not converge:  iterations=" << (int) solverStatus[1] << " eps=" << solverStatus[0];
        throw OpenMMException(message.str());
    }
    HIP_CHECK(rc);
}

void HipCalcAmoebaMultipoleForceKernel::download3(DeviceBuffer& buffer, vector<Vec3>& out) {
    HipContext& hip = *data.hip;
    vector<double> tmp(3 * (size_t) max(numParticles, 1));
    HIP_CHECK(ommhip_memcpy_d2h(tmp.data(), buffer.ptr, sizeof(double) * 3 * (size_t) numParticles, hip.stream));
    hip.sync();
    out.resize(numParticles);
    for (int i = 0; i < numParticles; i++) out[i] = Vec3(tmp[3 * i], tmp[3 * i + 1], tmp[3 * i + 2]);
}

void HipCalcAmoebaMultipoleForceKernel::getLabFramePermanentDipoles(ContextImpl& context, vector<Vec3>& dipoles) {
    induce();
    download3(labDipole, dipoles);
}

void HipCalcAmoebaMultipoleForceKernel::getInducedDipoles(ContextImpl& context, vector<Vec3>& dipoles) {
    induce();
    download3(indD, dipoles);
}

void HipCalcAmoebaMultipoleForceKernel::getTotalDipoles(ContextImpl& context, vector<Vec3>& dipoles) {
    vector<Vec3> induced;
    induce();
    download3(labDipole, dipoles);
    download3(indD, induced);
    for (int i = 0; i < numParticles; i++) dipoles[i] += induced[i];
}

void HipCalcAmoebaMultipoleForceKernel::syncHostPositions(ContextImpl& context) {
    // the Reference kernel reads the host copy of the state (ReferencePlatform::PlatformData)
    HipContext& hip = *data.hip;
    if (!hip.hostMode) {
        hip.downloadPositions(*data.positions);
        Vec3 a, b, c;
        hip.getBox(a, b, c);
        data.periodicBoxVectors[0] = a; data.periodicBoxVectors[1] = b; data.periodicBoxVectors[2] = c;
        *data.periodicBoxSize = Vec3(a[0], b[1], c[2]);
    }
}

void HipCalcAmoebaMultipoleForceKernel::getElectrostaticPotential(ContextImpl& context, const vector<Vec3>& inputGrid, vector<double>& outputElectrostaticPotential) {
    if (reference == NULL) throw OpenMMException("HIP platform: getElectrostaticPotential needs the AMOEBA plugin's Reference kernels");
    syncHostPositions(context);
    reference->getElectrostaticPotential(context, inputGrid, outputElectrostaticPotential);
}

void HipCalcAmoebaMultipoleForceKernel::getSystemMultipoleMoments(ContextImpl& context, vector<double>& outputMultipoleMoments) {
    if (reference == NULL) throw OpenMMException("HIP platform: getSystemMultipoleMoments needs the AMOEBA plugin's Reference kernels");
    syncHostPositions(context);
    reference->getSystemMultipoleMoments(context, outputMultipoleMoments);
}

void HipCalcAmoebaMultipoleForceKernel::copyParametersToContext(ContextImpl& context, const AmoebaMultipoleForce& force) {
    if (numParticles != force.getNumMultipoles())
        throw OpenMMException("updateParametersInContext: The number of multipoles has changed");
    data.hip->setAsCurrent();
    upload(force);
    if (reference != NULL) reference->copyParametersToContext(context, force);
}

void HipCalcAmoebaMultipoleForceKernel::getPMEParameters(double& alpha, int& nx, int& ny, int& nz) const {
    alpha = alphaEwald; nx = gridSize[0]; ny = gridSize[1]; nz = gridSize[2];
}


// ================================================================================================
// AmoebaTorsionTorsionForce
// ================================================================================================
bool HipCalcAmoebaTorsionTorsionForceKernel::supports(const AmoebaTorsionTorsionForce& force) {
    static const bool off = getenv("OPENMM_HIP_REFERENCE_CUSTOM_FORCES") != NULL && getenv("OPENMM_HIP_REFERENCE_CUSTOM_FORCES")[0] == '1';      // A/B knob, as for the Custom*Forces
    if (off || force.usesPeriodicBoundaryConditions() || force.getNumTorsionTorsionGrids() == 0) return false;
    for (int g = 0; g < force.getNumTorsionTorsionGrids(); g++) {
        const TorsionTorsionGrid& grid = force.getTorsionTorsionGrid(g);
        if (grid.size() == 0) continue;                           // an index nobody defined
        if (grid.size() < 2) return false;
        for (size_t x = 0; x < grid.size(); x++) {
            if (grid[x].size() != grid.size()) return false;
            for (size_t y = 0; y < grid.size(); y++)
                if (grid[x][y].size() != 6) return false;
        }
        if (grid[1][0][0] == grid[0][0][0]) return false;       // first index must run along the first angle
    }
    return true;
}

void HipCalcAmoebaTorsionTorsionForceKernel::initialize(const System& system, const AmoebaTorsionTorsionForce& force) {
    if (!supports(force)) throw OpenMMException("HIP platform: internal error, an AmoebaTorsionTorsionForce the native kernel does not take reached it");
    // the maps with the spline derivatives the Force computed (AmoebaTorsionTorsionForce.cpp:113-190), one after the other:
    // [x][y][angle1, angle2, f, fx, fy, fxy]
    vector<double> grids;
    vector<size_t> offset;
    vector<int> size;
    for (int g = 0; g < force.getNumTorsionTorsionGrids(); g++) {
        const TorsionTorsionGrid& grid = force.getTorsionTorsionGrid(g);
        offset.push_back(grids.size());
        size.push_back((int) grid.size());
        for (size_t x = 0; x < grid.size(); x++)
            for (size_t y = 0; y < grid.size(); y++)
                grids.insert(grids.end(), grid[x][y].begin(), grid[x][y].end());
    }
    vector<int> atoms; vector<double> params;
    for (int i = 0; i < force.getNumTorsionTorsions(); i++) {
        int p[5], chiral, grid;
        force.getTorsionTorsionParameters(i, p[0], p[1], p[2], p[3], p[4], chiral, grid);
        if (grid < 0 || grid >= force.getNumTorsionTorsionGrids() || size[grid] < 2) throw OpenMMException("AmoebaTorsionTorsionForce: a torsion-torsion refers to a grid that was not set");
        atoms.insert(atoms.end(), p, p + 5);
        atoms.push_back(chiral);
        params.push_back((double) offset[grid]);
        params.push_back(size[grid]);
    }
    HipValenceForm form;
    form.kind = OMMHIP_VALENCE_TORSION_TORSION;
    terms.upload(form, 6, atoms, params);
    terms.uploadGrids(grids);
}

double HipCalcAmoebaTorsionTorsionForceKernel::execute(ContextImpl& context, bool includeForces, bool includeEnergy) {
    terms.execute(includeEnergy);
    return 0.0;
}

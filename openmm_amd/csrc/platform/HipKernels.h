#ifndef OPENMM_HIPKERNELS_H_
#define OPENMM_HIPKERNELS_H_
/* KernelImpl subclasses of the "HIP" platform.  Each class derives from the abstract kernel
 * interface in olla/include/openmm/kernels.h that it replaces (cited per class) and mirrors the
 * behaviour of the corresponding Reference kernel (platforms/reference/src/ReferenceKernels.cpp).
 */
#include "HipPlatform.h"
#include "HipContext.h"
#include "openmm/kernels.h"
#include "openmm/System.h"
#include <functional>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

namespace OpenMM {

/** Device-side constraint data shared by the integrators and ApplyConstraints. */
class HipConstraints : public HipContextListener {
public:
    HipConstraints(const System& system, HipPlatform::PlatformData& data);
    ~HipConstraints();
    void atomsReordered();
    void boxChanged() {}
    void positionsSet() {}
    /** Constrain trial positions `target` (double4[N]) against the reference positions ctx.pos. */
    /** reference: the constrained positions the directions are taken from; NULL = the Context's current positions */
    void apply(void* target, double tol, void* reference = NULL);
    /** Remove constrained components from velocities `target` (double4[N], w = 1/m). */
    void applyToVelocities(void* target, double tol, void* reference = NULL);
    bool hasConstraints() const { return numSettle + numShake + numCcma > 0; }
    /** True when every constraint sits in a SETTLE water or a SHAKE cluster, so a whole step fits one launch
     *  (ommhip_integrate_fused); OPENMM_HIP_DISABLE_FUSED_STEP=1 forces the staged kernels. */
    bool fusedStepAvailable() const { return !allUnitAtoms.empty(); }
    /** Launch the fused step; consumes a pending CM-motion removal and leaves the new total momentum on the device. */
    void fusedStep(int integrator, const ommhip_integrator_state& state, double tol);
    /** The rigid three-atom molecules SETTLE treats (ReferenceConstraints.cpp:44-148 restated): int4 (centre, second, third, 0) and
     *  two distances (centre - other, other - other) per cluster. */
    static void findSettleClusters(const System& system, std::vector<int>& atoms, std::vector<double>& dist);
private:
    /** Upload the integration units this rank owns (all of them on one GPU). */
    void uploadOwnedUnits();
    std::vector<int> allUnitAtoms;        // int4 per unit
    std::vector<double> allUnitDist;      // double4 per unit
    bool smallUnits = false;              // no SHAKE cluster among the units (fusedStep launches the three-atom variant)
    int numUnits;
    double totalMass;
    DeviceBuffer unitAtoms, unitDist, cmScratch;
    void runCcma(void* target, bool velocities, double tol, void* reference);
    HipContext& hip;
    int numSettle, numShake, numCcma;
    DeviceBuffer settleAtoms, settleDist, shakeAtoms, shakeDist;
    DeviceBuffer ccmaAtoms, ccmaDist, ccmaDelta, ccmaDelta2, ccmaRowStart, ccmaCol, ccmaValue, ccmaConverged;
    ommhip_ccma ccma;
};

/** kernels.h:81-119 CalcForcesAndEnergyKernel; Reference: ReferenceKernels.cpp:178-203. */
class HipCalcForcesAndEnergyKernel : public CalcForcesAndEnergyKernel {
public:
    HipCalcForcesAndEnergyKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data) : CalcForcesAndEnergyKernel(name, platform), data(data) {}
    void initialize(const System& system);
    void beginComputation(ContextImpl& context, bool includeForce, bool includeEnergy, int groups);
    double finishComputation(ContextImpl& context, bool includeForce, bool includeEnergy, int groups, bool& valid);
private:
    HipPlatform::PlatformData& data;
    std::vector<Vec3> savedHostForces;
};

/** kernels.h:125-215 UpdateStateDataKernel; Reference: ReferenceKernels.cpp:205-308. */
class HipUpdateStateDataKernel : public UpdateStateDataKernel {
public:
    HipUpdateStateDataKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data) : UpdateStateDataKernel(name, platform), data(data) {}
    void initialize(const System& system);
    double getTime(const ContextImpl& context) const;
    void setTime(ContextImpl& context, double time);
    void getPositions(ContextImpl& context, std::vector<Vec3>& positions);
    void setPositions(ContextImpl& context, const std::vector<Vec3>& positions);
    void getVelocities(ContextImpl& context, std::vector<Vec3>& velocities);
    void setVelocities(ContextImpl& context, const std::vector<Vec3>& velocities);
    void getForces(ContextImpl& context, std::vector<Vec3>& forces);
    void getEnergyParameterDerivatives(ContextImpl& context, std::map<std::string, double>& derivs);
    void getPeriodicBoxVectors(ContextImpl& context, Vec3& a, Vec3& b, Vec3& c) const;
    void setPeriodicBoxVectors(ContextImpl& context, const Vec3& a, const Vec3& b, const Vec3& c);
    void createCheckpoint(ContextImpl& context, std::ostream& stream);
    void loadCheckpoint(ContextImpl& context, std::istream& stream);
private:
    HipPlatform::PlatformData& data;
};

/** kernels.h:220-247 ApplyConstraintsKernel; Reference: ReferenceKernels.cpp:310-333. */
class HipApplyConstraintsKernel : public ApplyConstraintsKernel {
public:
    HipApplyConstraintsKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data) : ApplyConstraintsKernel(name, platform), data(data) {}
    void initialize(const System& system);
    void apply(ContextImpl& context, double tol);
    void applyToVelocities(ContextImpl& context, double tol);
private:
    HipPlatform::PlatformData& data;
};

/** kernels.h:252-271 VirtualSitesKernel.  Systems with virtual sites run in host mode (Reference kernel); this is the no-site case. */
class HipVirtualSitesKernel : public VirtualSitesKernel {
public:
    HipVirtualSitesKernel(std::string name, const Platform& platform) : VirtualSitesKernel(name, platform) {}
    void initialize(const System& system) {}
    void computePositions(ContextImpl& context) {}
};

/** kernels.h:1493-1560 CalcPmeReciprocalForceKernel + ::IO -- reciprocal space on its own (the optional sub-boundary of SURVEY.md 8(b): the
 *  reference's GPU platforms can hand reciprocal space to such a kernel, plugins/cpupme is its CPU implementation).  Host posq in, host float
 *  forces out through the IO object; in between ommhip_pme_reciprocal: spreading, the hand-written 3-D FFT with the influence function, interpolation.
 *  Usable without a Context (constructed directly, as plugins/cpupme/tests/TestCpuPme.cpp does) on the current device, or through
 *  HipKernelFactory on the Context's device. */
class HipCalcPmeReciprocalForceKernel : public CalcPmeReciprocalForceKernel {
public:
    HipCalcPmeReciprocalForceKernel(std::string name, const Platform& platform, int deviceIndex = -1);
    ~HipCalcPmeReciprocalForceKernel();
    void initialize(int gridx, int gridy, int gridz, int numParticles, double alpha, bool deterministic);
    void beginComputation(IO& io, const Vec3* periodicBoxVectors, bool includeEnergy);
    double finishComputation(IO& io);
    void getPMEParameters(double& alpha, int& nx, int& ny, int& nz) const;
private:
    int deviceIndex, numParticles, paddedAtoms, grid[3];
    double alpha;
    bool deterministic, includeEnergy, started;
    void* stream;
    ommhip_pme pme;
    DeviceBuffer moduliX, moduliY, moduliZ, twiddleX, twiddleY, twiddleZ, eterm, gridReal, gridComplex, posq, force, forceDouble, slotOfAtom, energyBuffer, energyResult;
    std::vector<float> hostForce;
    std::vector<double> hostForceDouble;
    double lastBox[6];
    double* pinnedEnergy;
};

/** kernels.h:556-614 CalcNonbondedForceKernel; Reference: ReferenceKernels.cpp:864-1121. */
class HipCalcNonbondedForceKernel : public CalcNonbondedForceKernel, public HipContextListener {
public:
    HipCalcNonbondedForceKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data);
    ~HipCalcNonbondedForceKernel();
    void initialize(const System& system, const NonbondedForce& force);
    double execute(ContextImpl& context, bool includeForces, bool includeEnergy, bool includeDirect, bool includeReciprocal);
    void copyParametersToContext(ContextImpl& context, const NonbondedForce& force);
    void getPMEParameters(double& alpha, int& nx, int& ny, int& nz) const;
    void getLJPMEParameters(double& alpha, int& nx, int& ny, int& nz) const;
    void atomsReordered();
    void boxChanged();
    void positionsSet();
    /** out[0..6) = atoms, padded atoms, chunks in use, rows in use, chunk capacity, rebuilds so far (blocking). */
    void getNeighborListStats(long long* out);
    void getDomainInfo(long long* out);
    int getBlockDiag(float* out, int columns, int maxBlocks);
    int getBlockHalves(float* out, int maxBlocks);
    int getBlockCosts(float* ticks, float* candidates, int maxBlocks);
private:
    void computeParameters(ContextImpl& context, bool force);
    void allocateNeighborList(int maxChunks);
    static double innerPaddingFraction();
    void setupPme();
    void fillPmeStruct();
    void launchPme(int includeEnergy, bool spreadDone = false, bool fftDone = false);
    /** Synchronous overflow check (HipContext::listRecovery): grows the list and requests a rebuild if a device-triggered
     *  rebuild ran out of rows; returns the number of integration steps the device skipped meanwhile. */
    int recoverFromOverflow();
    /** The evaluation on one rank of a domain-decomposed run (PME only). */
    double executeDecomposed(ContextImpl& context, bool includeForces, bool includeEnergy, bool includeDirect, bool includeReciprocal);
    void setupPmeDecomposed();
    void checkDecomposedFlags();
    // LJPME: the dispersion grid (second set of PME buffers), per-atom C6 factors, posq with the C6 factor in .w
    void setupDispersionPme();
    double dispersionAlpha = 0.0, dispersionSelfEnergy = 0.0;
    int dispersionGridSize[3] = {0, 0, 0};
    DeviceBuffer dModuliX, dModuliY, dModuliZ, dEterm, dGridReal, dGridComplex, dTwiddleX, dTwiddleY, dTwiddleZ, c6D, posqDisp;
    ommhip_pme pmeDisp;
    bool dispersionEtermDirty = true;
    DeviceBuffer gridComplex2, ddError;
    int* pinnedDdError = NULL;
    int ddHalo = 0;
    void rebuildEterm();
    int estimateChunks() const;
    HipPlatform::PlatformData& data;
    HipContext& hip;
    int numParticles, num14, numExclusionPairs;
    bool exclusionsSpanUnits = false;
    NonbondedMethod nonbondedMethod;
    double nonbondedCutoff, switchingDistance, rfDielectric, ewaldAlpha, dispersionCoefficient, selfEnergy, padding;
    bool useSwitchingFunction, exceptionsArePeriodic, usesPeriodic;
    int kmax[3], gridSize[3], directGridOverride = 0;
    int pairGridBesideSideStream();
    unsigned evaluationCount = 0;
    std::vector<std::vector<double> > baseParticleParams, baseExceptionParams;   // (charge, sigma, epsilon)
    std::vector<std::pair<int, int> > exceptionAtoms;
    std::map<std::pair<std::string, int>, std::vector<double> > particleParamOffsets, exceptionParamOffsets;
    std::map<std::string, double> lastGlobalValues;
    std::vector<double> charges;                                             // current, atom order
    std::vector<int> hostExclStart, hostExclAtoms;                           // exclusion CSR (atom indices)
    DeviceBuffer exclBlockRange, exclSlotStart, exclSlots, cellStart, cellBlocks, cellBoxes, cellMeta;
    void updateExclusionBlockRanges();
    bool slotParamsDirty, forceRebuild, etermDirty, hasInitializedParams;
    bool foldExclusions;       // this evaluation: the Ewald exclusion correction rides in the PME interpolation launch
    // device
    DeviceBuffer chargeD, sigmaD, epsilonD, posq, posqRef, posqRel, posqRelLo, sigEps, exclStart, exclAtoms, nlState, blockCenter, blockHalf, chunkInfo, rowJ, rowMask, chunkInfoInner, rowJInner, rowMaskInner, blockRuns, posqRefInner;
    DeviceBuffer exceptionAtomsD, exceptionParamsD, exclusionPairsD, ewaldStructure;
    DeviceBuffer moduliX, moduliY, moduliZ, eterm, gridReal, gridComplex, twiddleX, twiddleY, twiddleZ, tileCount, tileBlocks;
    double maxCharge;
    bool maxChargeDirty;
    bool enableTileSpread(int nx, int ny, int nz);
    ommhip_neighbor_list nl;
    ommhip_nonbonded_params params;
    ommhip_pme pme;
    int* pinnedState;
    bool stateCopyPending;
    bool debugShrinkDone = false;
};

/** Common code of the per-term bonded kernels. */
class HipTermForce {
public:
    HipTermForce(HipPlatform::PlatformData& data, int kind, int atomsPerTerm, int paramsPerTerm) : data(data), kind(kind), atomsPerTerm(atomsPerTerm), paramsPerTerm(paramsPerTerm), numTerms(0), periodic(false), registrationId(-1) {}
    ~HipTermForce();
    void upload(const std::vector<int>& atoms, const std::vector<double>& params, bool usesPeriodic, int forceGroup);
    void uploadParams(const std::vector<double>& params);
    void execute(bool includeEnergy);
    int getNumTerms() const { return numTerms; }
private:
    HipPlatform::PlatformData& data;
    int kind, atomsPerTerm, paramsPerTerm, numTerms;
    bool periodic;
    int registrationId;
    ommhip_term_batch batch() const;
    DeviceBuffer atomsD, paramsD;
    std::vector<int> order;        // term t of the device lists is term order[t] of the caller (see upload)
};

/** kernels.h:276-341 CalcHarmonicBondForceKernel; Reference: ReferenceKernels.cpp:343-391. */
class HipCalcHarmonicBondForceKernel : public CalcHarmonicBondForceKernel {
public:
    HipCalcHarmonicBondForceKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data) : CalcHarmonicBondForceKernel(name, platform), terms(data, OMMHIP_TERM_HARMONIC_BOND, 2, 2) {}
    void initialize(const System& system, const HarmonicBondForce& force);
    double execute(ContextImpl& context, bool includeForces, bool includeEnergy);
    void copyParametersToContext(ContextImpl& context, const HarmonicBondForce& force);
private:
    HipTermForce terms;
};

/** kernels.h:346-411 CalcHarmonicAngleForceKernel; Reference: ReferenceKernels.cpp:473-522. */
class HipCalcHarmonicAngleForceKernel : public CalcHarmonicAngleForceKernel {
public:
    HipCalcHarmonicAngleForceKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data) : CalcHarmonicAngleForceKernel(name, platform), terms(data, OMMHIP_TERM_HARMONIC_ANGLE, 3, 2) {}
    void initialize(const System& system, const HarmonicAngleForce& force);
    double execute(ContextImpl& context, bool includeForces, bool includeEnergy);
    void copyParametersToContext(ContextImpl& context, const HarmonicAngleForce& force);
private:
    HipTermForce terms;
};

/** kernels.h:416-481 CalcPeriodicTorsionForceKernel; Reference: ReferenceKernels.cpp:603-650. */
class HipCalcPeriodicTorsionForceKernel : public CalcPeriodicTorsionForceKernel {
public:
    HipCalcPeriodicTorsionForceKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data) : CalcPeriodicTorsionForceKernel(name, platform), terms(data, OMMHIP_TERM_PERIODIC_TORSION, 4, 4 * OMMHIP_TORSION_SUBTERMS) {}
    void initialize(const System& system, const PeriodicTorsionForce& force);
    double execute(ContextImpl& context, bool includeForces, bool includeEnergy);
    void copyParametersToContext(ContextImpl& context, const PeriodicTorsionForce& force);
private:
    /** Groups the force's torsions by their four atoms and packs each group's (k, phase, periodicity) into sub-terms. */
    void packTorsions(const PeriodicTorsionForce& force, std::vector<int>& atoms, std::vector<double>& params);
    HipTermForce terms;
};

/** Shared implementation of the three native integrators. */
class HipIntegratorBase {
public:
    HipIntegratorBase(HipPlatform::PlatformData& data) : data(data) {}
protected:
    void fillState(ommhip_integrator_state& s, double dt);
    double kineticEnergy(double timeShift);
    void finishStep(double dt);
    /** Steps firstIndex .. endIndex-1 (force evaluation + launch each; the first without the evaluation if haveForces). */
    void runSteps(ContextImpl& context, const Integrator& integrator, const std::function<void(long long)>& launch,
                  long long firstIndex, long long endIndex, bool haveForces);
    HipPlatform::PlatformData& data;
};

/** kernels.h:1033-1061 IntegrateVerletStepKernel; Reference: ReferenceKernels.cpp:2063-2096. */
class HipIntegrateVerletStepKernel : public IntegrateVerletStepKernel, public HipIntegratorBase {
public:
    HipIntegrateVerletStepKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data) : IntegrateVerletStepKernel(name, platform), HipIntegratorBase(data) {}
    void initialize(const System& system, const VerletIntegrator& integrator) { data.hip->forcesRecomputedEveryStep = true; }
    void execute(ContextImpl& context, const VerletIntegrator& integrator);
    double computeKineticEnergy(ContextImpl& context, const VerletIntegrator& integrator);
private:
    void launchStep(ContextImpl& context, const VerletIntegrator& integrator, long long stepIndex);
};

/** kernels.h:1160-1188 IntegrateLangevinStepKernel; Reference: ReferenceKernels.cpp:2360-2402. */
class HipIntegrateLangevinStepKernel : public IntegrateLangevinStepKernel, public HipIntegratorBase {
public:
    HipIntegrateLangevinStepKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data) : IntegrateLangevinStepKernel(name, platform), HipIntegratorBase(data), seed(0) {}
    void initialize(const System& system, const LangevinIntegrator& integrator);
    void execute(ContextImpl& context, const LangevinIntegrator& integrator);
    double computeKineticEnergy(ContextImpl& context, const LangevinIntegrator& integrator);
private:
    void launchStep(ContextImpl& context, const LangevinIntegrator& integrator, long long stepIndex);
    unsigned long long seed;
};

/** kernels.h:1193-1221 IntegrateLangevinMiddleStepKernel; Reference: ReferenceKernels.cpp:2404-2445. */
class HipIntegrateLangevinMiddleStepKernel : public IntegrateLangevinMiddleStepKernel, public HipIntegratorBase {
public:
    HipIntegrateLangevinMiddleStepKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data) : IntegrateLangevinMiddleStepKernel(name, platform), HipIntegratorBase(data), seed(0) {}
    void initialize(const System& system, const LangevinMiddleIntegrator& integrator);
    void execute(ContextImpl& context, const LangevinMiddleIntegrator& integrator);
    double computeKineticEnergy(ContextImpl& context, const LangevinMiddleIntegrator& integrator);
private:
    void launchStep(ContextImpl& context, const LangevinMiddleIntegrator& integrator, long long stepIndex);
    unsigned long long seed;
};

/** kernels.h:1464-1488 RemoveCMMotionKernel; Reference: ReferenceKernels.cpp:2705-2740. */
class HipRemoveCMMotionKernel : public RemoveCMMotionKernel {
public:
    HipRemoveCMMotionKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data) : RemoveCMMotionKernel(name, platform), data(data), frequency(1) {}
    void initialize(const System& system, const CMMotionRemover& force);
    void execute(ContextImpl& context);
private:
    HipPlatform::PlatformData& data;
    int frequency;
    DeviceBuffer scratch;
};

/** kernels.h:1425-1459 ApplyMonteCarloBarostatKernel; Reference: ReferenceKernels.cpp ReferenceApplyMonteCarloBarostatKernel,
 *  ReferenceMonteCarloBarostat.cpp:68-120.  Keeps NPT runs in device mode: positions are scaled and restored in HBM. */
class HipApplyMonteCarloBarostatKernel : public ApplyMonteCarloBarostatKernel {
public:
    HipApplyMonteCarloBarostatKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data) : ApplyMonteCarloBarostatKernel(name, platform), data(data), numMolecules(0) {}
    void initialize(const System& system, const Force& barostat);
    void scaleCoordinates(ContextImpl& context, double scaleX, double scaleY, double scaleZ);
    void restoreCoordinates(ContextImpl& context);
private:
    HipPlatform::PlatformData& data;
    int numMolecules;
    DeviceBuffer molStart, molAtoms, savedPos;
};

}  // namespace OpenMM
#endif

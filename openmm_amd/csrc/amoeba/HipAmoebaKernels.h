#ifndef OPENMM_HIPAMOEBAKERNELS_H_
#define OPENMM_HIPAMOEBAKERNELS_H_
/* Native AMOEBA kernels of the OpenMM "HIP" platform (BASELINE.json configs[4]) -- the contents of
 * libOpenMMAmoebaHIP.so, the counterpart of the reference's libOpenMMAmoebaCUDA: a plugin of its own that links the AMOEBA
 * plugin's API library (libOpenMMAmoeba) and the HIP platform (libOpenMMHIP), so that neither of those depends on the other.
 * Each class derives from the abstract kernel of plugins/amoeba/openmmapi/include/openmm/amoebaKernels.h and states the Reference
 * implementation it is checked against.
 */
#include "HipPlatform.h"
#include "HipValenceKernels.h"
#include "openmm/AmoebaTorsionTorsionForce.h"
#include "HipContext.h"
#include "openmm/amoebaKernels.h"
#include "openmm_hip_amoeba.h"

namespace OpenMM {

/** amoebaKernels.h:182-219 CalcAmoebaVdwForceKernel; Reference: AmoebaReferenceKernels.cpp:643-696 + AmoebaReferenceVdwForce.cpp. */
class HipCalcAmoebaVdwForceKernel : public CalcAmoebaVdwForceKernel {
public:
    HipCalcAmoebaVdwForceKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data) : CalcAmoebaVdwForceKernel(name, platform), data(data) {}
    ~HipCalcAmoebaVdwForceKernel();
    void initialize(const System& system, const AmoebaVdwForce& force);
    double execute(ContextImpl& context, bool includeForces, bool includeEnergy);
    void copyParametersToContext(ContextImpl& context, const AmoebaVdwForce& force);
private:
    void upload(const AmoebaVdwForce& force);
    /** The evaluation itself, on the platform's side stream: started by execute(), or earlier in the same evaluation by a kernel that is
     *  about to keep the host busy (HipContext::launchEarlyWork: the multipole kernel's dipole solver) -- the pair kernel then runs beside
     *  the solver's small launches instead of after them. */
    void launch(ContextImpl& context, bool includeForces, bool includeEnergy);
    int earlyId = -1;
    HipPlatform::PlatformData& data;
    int numParticles = 0;
    bool usePBC = false;
    double cutoff = 0, dispersionCoefficient = 0, softcorePower = 0, softcoreAlpha = 0;
    ommhip_amoeba_vdw vdw;
    void allocatePairList(int cap);
    int pairNeeded = 0, grownPairCap = 0;      // grownPairCap: the capacity a list that did not fit was grown to (0: never)
    // Verlet skin of the pair lists: rebuilt when an atom has moved by half of it (device side) or when slot order / box / parameters / capacity changed
    bool listDirty = true;
    long long listOrderVersion = -1, listBoxVersion = -1;
    DeviceBuffer refPos, listState;
    DeviceBuffer parent, reduction, type, sigma, epsilon, exclStart, exclAtoms, alchemical, reduced, tileBounds, exclPos, pairList, pairCount, pairOverflow;
};

/** amoebaKernels.h:49-77 CalcAmoebaTorsionTorsionForceKernel; Reference: AmoebaReferenceKernels.cpp (ReferenceCalcAmoebaTorsionTorsionForceKernel) +
 *  AmoebaReferenceTorsionTorsionForce.cpp:283-530.  One list of kernels/valence.hip (OMMHIP_VALENCE_TORSION_TORSION): it goes out in the same
 *  launch as the other AMOEBA valence terms (HipContext::addValence). */
class HipCalcAmoebaTorsionTorsionForceKernel : public CalcAmoebaTorsionTorsionForceKernel {
public:
    HipCalcAmoebaTorsionTorsionForceKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data) : CalcAmoebaTorsionTorsionForceKernel(name, platform), terms(data) {}
    void initialize(const System& system, const AmoebaTorsionTorsionForce& force);
    double execute(ContextImpl& context, bool includeForces, bool includeEnergy);
    /** Can the native kernel take this Force?  (all maps square, of one size, with derivatives; no periodic boundary conditions) */
    static bool supports(const AmoebaTorsionTorsionForce& force);
private:
    HipValenceTerms terms;
};

/** amoebaKernels.h:82-139 CalcAmoebaMultipoleForceKernel for PME with direct, mutual or extrapolated polarization; Reference: AmoebaReferenceKernels.cpp:170-520 +
 *  AmoebaReferencePmeMultipoleForce.  The factory hands every other configuration (NoCutoff, grids
 *  the platform's FFT does not take) to the AMOEBA plugin's own Reference kernel; so does this class for the two queries it does not
 *  compute itself (electrostatic potential on a grid of points, system multipole moments). */
class HipCalcAmoebaMultipoleForceKernel : public CalcAmoebaMultipoleForceKernel {
public:
    HipCalcAmoebaMultipoleForceKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data, KernelImpl* referenceKernel);
    ~HipCalcAmoebaMultipoleForceKernel();
    /** Can this force be computed natively?  (PME, direct / mutual / extrapolated polarization, FFT-friendly grid, rectangular or triclinic box.) */
    static bool supports(const AmoebaMultipoleForce& force, const System& system);
    void initialize(const System& system, const AmoebaMultipoleForce& force);
    double execute(ContextImpl& context, bool includeForces, bool includeEnergy);
    void getLabFramePermanentDipoles(ContextImpl& context, std::vector<Vec3>& dipoles);
    void getInducedDipoles(ContextImpl& context, std::vector<Vec3>& dipoles);
    void getTotalDipoles(ContextImpl& context, std::vector<Vec3>& dipoles);
    void getElectrostaticPotential(ContextImpl& context, const std::vector<Vec3>& inputGrid, std::vector<double>& outputElectrostaticPotential);
    void getSystemMultipoleMoments(ContextImpl& context, std::vector<double>& outputMultipoleMoments);
    void copyParametersToContext(ContextImpl& context, const AmoebaMultipoleForce& force);
    void getPMEParameters(double& alpha, int& nx, int& ny, int& nz) const;
private:
    void upload(const AmoebaMultipoleForce& force);
    void prepareGrid();
    void induce();
    void setScanOrder();
    void listBuilt();
    void allocatePairList(int cap);
    bool growPairList(int rc, int attempt);
    int pairNeeded = 0, grownPairCap = 0;
    bool listDirty = true;                     // Verlet skin of the pair lists, as in the vdW kernel
    long long listOrderVersion = -1, listBoxVersion = -1;
    DeviceBuffer refPos, listState;
    void checkSolver(int rc);
    void download3(DeviceBuffer& buffer, std::vector<Vec3>& out);
    void syncHostPositions(ContextImpl& context);
    HipPlatform::PlatformData& data;
    CalcAmoebaMultipoleForceKernel* reference;      // the AMOEBA plugin's Reference kernel (for the two queries above), owned
    int numParticles = 0, gridSize[3] = {0, 0, 0};
    double alphaEwald = 0, cutoff = 0, lastBox[6] = {0, 0, 0, 0, 0, 0};
    bool etermBuilt = false, mutual = false, extrapolated = false;
    double solverStatus[2] = {0, 0};          // epsilon reached and iterations of the last mutual-polarization solve
    // first guess of the solver from the solutions of the previous steps (ommhip_amoeba_multipole::history): a ring of HistorySlots records
    static const int HistorySlots = OMMHIP_AMOEBA_MAX_HISTORY;
    int historyNext = 0, historyValid = 0;    // slot the next solution goes to; how many of the most recent slots hold solutions of consecutive steps
    long long historyStep = -1, historyPositionsVersion = -1, historyBoxVersion = -1;      // when the newest record was made
    bool historySameStep = false;
    int lastHistoryUse = -1;
    void chooseFirstGuess();                  // before a solve: history_use / history_next / expected_iterations
    void recordSolve();                       // after a successful one
    ommhip_amoeba_multipole mp;
    ommhip_pme pme, pme2;
    void* sideStream = NULL; void* eventA = NULL; void* eventB = NULL;
    DeviceBuffer charge, molDipole, molQuad, axis, thole, damping, polarity, specStart, specAtom, specScale;
    DeviceBuffer labDipole, labQuad, fieldD, fieldP, indD, indP, phi, phiInd, phiIndP, solver, solverGather, history, extDipoles, extGradients, torque, tileBounds, specPos, specScaleSorted, pairList, pairCount, pairOverflow, pairCache;
    DeviceBuffer moduliX, moduliY, moduliZ, twiddleX, twiddleY, twiddleZ, eterm, gridReal, gridComplex, gridReal2, gridComplex2;
};

}  // namespace OpenMM
#endif

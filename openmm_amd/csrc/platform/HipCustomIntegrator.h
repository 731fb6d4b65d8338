#ifndef OPENMM_HIP_CUSTOM_INTEGRATOR_H_
#define OPENMM_HIP_CUSTOM_INTEGRATOR_H_
/* IntegrateCustomStepKernel (kernels.h:1329-1395) in device mode.  The sequence of computations is walked on the host exactly as
 * ReferenceCustomDynamics::update() does (ReferenceCustomDynamics.cpp:227-355: which step needs which force group, what invalidates the
 * forces, if / while blocks, global computations); everything per degree of freedom runs on the device as small interpreted programs
 * (kernels/custom_integrator.hip), consecutive ComputePerDof steps in one launch.  Forces are kept per force group (device copies in atom
 * order) for as long as the positions they belong to stand -- across steps as well, which the reference's own cache (one map per call of
 * update()) does not do: an MTS integrator that ends a step and begins the next with the same force group evaluates it once. */
#include "HipKernels.h"
#include "openmm/CustomIntegrator.h"
#include "openmm/internal/CustomIntegratorUtilities.h"
#include "lepton/ParsedExpression.h"
#include <map>
#include <string>
#include <vector>

namespace OpenMM {

class HipIntegrateCustomStepKernel : public IntegrateCustomStepKernel, public HipIntegratorBase {
public:
    HipIntegrateCustomStepKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data) : IntegrateCustomStepKernel(name, platform), HipIntegratorBase(data) {}
    void initialize(const System& system, const CustomIntegrator& integrator);
    void execute(ContextImpl& context, CustomIntegrator& integrator, bool& forcesAreValid);
    double computeKineticEnergy(ContextImpl& context, CustomIntegrator& integrator, bool& forcesAreValid);
    void getGlobalVariables(ContextImpl& context, std::vector<double>& values) const;
    void setGlobalVariables(ContextImpl& context, const std::vector<double>& values);
    void getPerDofVariable(ContextImpl& context, int variable, std::vector<Vec3>& values) const;
    void setPerDofVariable(ContextImpl& context, int variable, const std::vector<Vec3>& values);
    /** Can every expression of this integrator be turned into a device program?  (no tabulated functions, no vector functions, no deriv())
     *  If not, the Context runs in host mode with the Reference kernel. */
    static bool supports(const CustomIntegrator& integrator);
private:
    struct Program { int first, count, usesRandom; bool usesForce, usesEnergy; };
    void compile(ContextImpl& context, const CustomIntegrator& integrator);
    /** postfix translation of an expression tree; throws if it holds something the device cannot evaluate */
    Program translate(const Lepton::ExpressionTreeNode& root);
    void emit(const Lepton::ExpressionTreeNode& node, int& depth, int& maxDepth, Program& program);
    int globalIndex(const std::string& name);
    /** host values of the globals -> the device array, if anything changed since the last time */
    void syncGlobals();
    void loadContextParameters(ContextImpl& context);
    void recordChangedParameters(ContextImpl& context);
    double evaluateOnHost(const Lepton::ParsedExpression& expression, double energy);
    /** forces (and / or energy) of the force groups `flags` at the current positions, from the cache or evaluated now */
    void ensureForces(ContextImpl& context, int flags, bool needForces, bool needEnergy, bool computeForces, bool computeEnergy, bool& forcesAreValid);
    void invalidateForces();
    void enqueue(const Program& program, int target, int flags, bool needsForces);
    void flush();
    double runSum(const Program& program, int flags, bool needsForces);
    ommhip_vm_state vmState();

    bool compiled = false;
    int numAtoms = 0, numPerDof = 0, numIntegratorGlobals = 0;
    std::map<const Lepton::ParsedExpression*, int> hostRandomUse;      // bit 0: the expression reads `uniform`, bit 1: `gaussian`
    // the integrator's definition, analysed (CustomIntegratorUtilities::analyzeComputations)
    std::vector<CustomIntegrator::ComputationType> stepType;
    std::vector<std::string> stepVariable, perDofNames;
    std::vector<std::vector<Lepton::ParsedExpression> > expressions;
    std::vector<CustomIntegratorUtilities::Comparison> comparisons;
    std::vector<int> blockEnd, forceGroupFlags, stepTarget, stepGlobal;
    std::vector<bool> invalidatesForces, needsForces, needsEnergy, computeBoth;
    std::vector<Program> stepProgram;
    Program kineticProgram;
    bool kineticNeedsForce = false;
    std::vector<ommhip_vm_instruction> instructions;
    // globals: [0] dt, [1 .. G] the integrator's, then the Context parameters, last the energy of the step being executed
    std::vector<std::string> globalNames;
    std::vector<double> globalValues;
    std::map<std::string, int> globalSlot;
    int energySlot = 0;
    bool globalsDirty = true;
    DeviceBuffer globalsD, programD, perDofD, oldPosD, sumScratchD, sumResultD;
    // per force-group-flags caches
    std::map<int, DeviceBuffer*> forceCache;
    std::map<int, bool> forceCached;
    std::map<int, double> energyCache;
    long long cachePositionsVersion = -1, cacheBoxVersion = -1;
    std::vector<ommhip_vm_step> pending;
    std::vector<std::vector<Vec3> > initialPerDof;
public:
    ~HipIntegrateCustomStepKernel();
};

}  // namespace OpenMM

#endif

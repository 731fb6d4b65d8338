/* See HipCustomIntegrator.h. */
#include "HipCustomIntegrator.h"
#include "SimTKOpenMMUtilities.h"
#include "openmm/internal/OSRngSeed.h"
#include "openmm/OpenMMException.h"
#include "openmm/internal/ContextImpl.h"
#include "lepton/Operation.h"
#include "lepton/Parser.h"
#include <cstdio>
#include <cstdlib>
#include <sstream>

using namespace OpenMM;
using namespace Lepton;
using namespace std;

namespace {
// a variable name of the form <prefix><0..31> -> the number, else -1
int numberedVariable(const string& name, const string& prefix) {
    if (name.size() <= prefix.size() || name.compare(0, prefix.size(), prefix) != 0) return -1;
    int value = 0;
    for (size_t i = prefix.size(); i < name.size(); i++) {
        if (!isdigit((unsigned char) name[i])) return -1;
        value = 10 * value + (name[i] - '0');
    }
    return value < 32 ? value : -1;
}

bool treeIsTranslatable(const ExpressionTreeNode& node) {
    if (node.getOperation().getId() == Operation::CUSTOM) return false;
    for (size_t i = 0; i < node.getChildren().size(); i++)
        if (!treeIsTranslatable(node.getChildren()[i])) return false;
    return true;
}

// Stack slots the postfix program of a tree needs (the count emit() arrives at): the children are evaluated left to right, child i
// while i results are already waiting.
int treeStackDepth(const ExpressionTreeNode& node) {
    int need = 1;
    for (size_t i = 0; i < node.getChildren().size(); i++)
        need = max(need, (int) i + treeStackDepth(node.getChildren()[i]));
    return need;
}

bool expressionIsTranslatable(const string& expression) {
    try {
        // the tree compile() will translate: CustomIntegratorUtilities::analyzeComputations hands out the OPTIMIZED expression
        // (CustomIntegratorUtilities.cpp:108-114), whose depth may differ from the parsed one's
        const ParsedExpression parsed = Parser::parse(expression).optimize();
        // deeper than the interpreter's stack: host mode (the Reference kernel), not an exception at the first step.  An unknown variable
        // is not looked for here: the Reference kernel rejects it as well.
        return treeIsTranslatable(parsed.getRootNode()) && treeStackDepth(parsed.getRootNode()) <= OMMHIP_VM_STACK;
    }
    catch (...) { return false; }      // an unknown function: tabulated, vector-valued, deriv()
}

bool treeUsesVariable(const ExpressionTreeNode& node, const string& name) {
    if (node.getOperation().getId() == Operation::VARIABLE && node.getOperation().getName() == name) return true;
    for (size_t i = 0; i < node.getChildren().size(); i++)
        if (treeUsesVariable(node.getChildren()[i], name)) return true;
    return false;
}
}  // namespace

bool HipIntegrateCustomStepKernel::supports(const CustomIntegrator& integrator) {
    static const bool off = getenv("OPENMM_HIP_REFERENCE_CUSTOM_INTEGRATOR") != NULL && getenv("OPENMM_HIP_REFERENCE_CUSTOM_INTEGRATOR")[0] == '1';      // A/B knob: host mode
    if (off || integrator.getNumTabulatedFunctions() > 0) return false;
    for (int i = 0; i < integrator.getNumComputations(); i++) {
        CustomIntegrator::ComputationType type;
        string variable, expression;
        integrator.getComputationStep(i, type, variable, expression);
        if (type == CustomIntegrator::ComputeGlobal || type == CustomIntegrator::ComputePerDof || type == CustomIntegrator::ComputeSum) {
            if (!expressionIsTranslatable(expression)) return false;
        }
        else if (type == CustomIntegrator::IfBlockStart || type == CustomIntegrator::WhileBlockStart) {
            string lhs, rhs;
            CustomIntegratorUtilities::Comparison comparison;
            try { CustomIntegratorUtilities::parseCondition(expression, lhs, rhs, comparison); } catch (...) { return false; }
            if (!expressionIsTranslatable(lhs) || !expressionIsTranslatable(rhs)) return false;
        }
    }
    return expressionIsTranslatable(integrator.getKineticEnergyExpression());
}

HipIntegrateCustomStepKernel::~HipIntegrateCustomStepKernel() {
    for (map<int, DeviceBuffer*>::iterator it = forceCache.begin(); it != forceCache.end(); ++it) delete it->second;
}

void HipIntegrateCustomStepKernel::initialize(const System& system, const CustomIntegrator& integrator) {
    numAtoms = system.getNumParticles();
    numPerDof = integrator.getNumPerDofVariables();
    numIntegratorGlobals = integrator.getNumGlobalVariables();
    // The per-DOF noise is a pure function of (data.integratorSeed, data.customDraws, atom): both words travel in a checkpoint
    // (HipUpdateStateDataKernel::createCheckpoint), so a restart -- in this Context or another -- continues the stream.  The random numbers of
    // ComputeGlobal steps come from the host generator of the Reference platform, whose state the checkpoint carries as well.
    const int s = integrator.getRandomNumberSeed();
    data.integratorSeed = s == 0 ? (unsigned long long) osrngseed() : (unsigned long long) (unsigned int) s;
    data.customDraws = 0;
    SimTKOpenMMUtilities::setRandomNumberSeed((unsigned int) data.integratorSeed);
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    perDofD.allocate(max((size_t) 16, sizeof(double) * 3 * numAtoms * max(numPerDof, 1)));
    HIP_CHECK(ommhip_memset(perDofD.ptr, 0, perDofD.bytes, hip.stream));
    sumScratchD.allocate(sizeof(double) * OMMHIP_KE_SCRATCH * 64);
    sumResultD.allocate(16);
    globalNames.assign(1, "dt");
    globalValues.assign(1, integrator.getStepSize());
    for (int i = 0; i < numIntegratorGlobals; i++) {
        globalNames.push_back(integrator.getGlobalVariableName(i));
        globalValues.push_back(integrator.getGlobalVariable(i));
    }
    for (size_t i = 0; i < globalNames.size(); i++) globalSlot[globalNames[i]] = (int) i;
}

int HipIntegrateCustomStepKernel::globalIndex(const string& name) {
    map<string, int>::const_iterator it = globalSlot.find(name);
    return it == globalSlot.end() ? -1 : it->second;
}

void HipIntegrateCustomStepKernel::emit(const ExpressionTreeNode& node, int& depth, int& maxDepth, Program& program) {
    const Operation& op = node.getOperation();
    for (size_t i = 0; i < node.getChildren().size(); i++) emit(node.getChildren()[i], depth, maxDepth, program);
    ommhip_vm_instruction in = {0, 0, 0.0};
    const int arity = (int) node.getChildren().size();
    switch (op.getId()) {
        case Operation::CONSTANT:
            in.op = OMMHIP_VM_CONSTANT; in.value = dynamic_cast<const Operation::Constant&>(op).getValue();
            break;
        case Operation::VARIABLE: {
            const string name = op.getName();
            in.op = OMMHIP_VM_VARIABLE;
            if (name == "x") in.arg = 0;
            else if (name == "v") in.arg = 1;
            else if (name == "f" || numberedVariable(name, "f") >= 0) { in.arg = 2; program.usesForce = true; }
            else if (name == "m") in.arg = 3;
            else if (name == "gaussian") { in.arg = 4; program.usesRandom |= 1; }
            else if (name == "uniform") { in.arg = 5; program.usesRandom |= 2; }
            else if (name == "energy" || numberedVariable(name, "energy") >= 0) { in.op = OMMHIP_VM_GLOBAL; in.arg = energySlot; program.usesEnergy = true; }
            else {
                int perDof = -1;
                for (int k = 0; k < numPerDof; k++)
                    if (perDofNames[k] == name) perDof = k;
                if (perDof >= 0) in.arg = 6 + perDof;
                else {
                    const int g = globalIndex(name);
                    if (g < 0) throw OpenMMException("CustomIntegrator: unknown variable '" + name + "'");
                    in.op = OMMHIP_VM_GLOBAL; in.arg = g;
                }
            }
            break;
        }
        case Operation::ADD_CONSTANT: in.op = OMMHIP_VM_ADD_CONSTANT; in.value = dynamic_cast<const Operation::AddConstant&>(op).getValue(); break;
        case Operation::MULTIPLY_CONSTANT: in.op = OMMHIP_VM_MULTIPLY_CONSTANT; in.value = dynamic_cast<const Operation::MultiplyConstant&>(op).getValue(); break;
        case Operation::POWER_CONSTANT: in.op = OMMHIP_VM_POWER_CONSTANT; in.value = dynamic_cast<const Operation::PowerConstant&>(op).getValue(); break;
        case Operation::CUSTOM: throw OpenMMException("CustomIntegrator: function '" + op.getName() + "' has no device form");
        default:
            // the VM's operation codes follow Lepton's ids from ADD on (include/openmm_hip_kernels.h)
            in.op = OMMHIP_VM_ADD + ((int) op.getId() - (int) Operation::ADD);
            break;
    }
    program.count++;
    instructions.push_back(in);
    depth += 1 - arity;
    maxDepth = max(maxDepth, depth);
}

HipIntegrateCustomStepKernel::Program HipIntegrateCustomStepKernel::translate(const ExpressionTreeNode& root) {
    Program p = {(int) instructions.size(), 0, 0, false, false};
    int depth = 0, maxDepth = 0;
    emit(root, depth, maxDepth, p);
    if (maxDepth > OMMHIP_VM_STACK) throw OpenMMException("CustomIntegrator: an expression is too deeply nested for the device interpreter");
    return p;
}

void HipIntegrateCustomStepKernel::loadContextParameters(ContextImpl& context) {
    for (map<string, double>::const_iterator it = context.getParameters().begin(); it != context.getParameters().end(); ++it) {
        int g = globalIndex(it->first);
        if (g < 0) {
            g = (int) globalNames.size();
            globalNames.push_back(it->first); globalValues.push_back(it->second); globalSlot[it->first] = g;
            globalsDirty = true;
        }
        else if (g > numIntegratorGlobals && globalValues[g] != it->second) { globalValues[g] = it->second; globalsDirty = true; }
    }
}

void HipIntegrateCustomStepKernel::recordChangedParameters(ContextImpl& context) {
    // ReferenceCustomDynamics.cpp:425-432: a global computation may have assigned to a Context parameter
    for (map<string, double>::const_iterator it = context.getParameters().begin(); it != context.getParameters().end(); ++it) {
        const int g = globalIndex(it->first);
        if (g >= 0 && globalValues[g] != it->second) context.setParameter(it->first, globalValues[g]);
    }
}

void HipIntegrateCustomStepKernel::compile(ContextImpl& context, const CustomIntegrator& integrator) {
    perDofNames.clear();
    for (int k = 0; k < numPerDof; k++) perDofNames.push_back(integrator.getPerDofVariableName(k));
    loadContextParameters(context);
    energySlot = (int) globalNames.size();
    globalNames.push_back("energy"); globalValues.push_back(0.0);          // not in globalSlot: reached through the variables energy / energyN only
    const int numSteps = integrator.getNumComputations();
    stepType.resize(numSteps); stepVariable.resize(numSteps);
    for (int i = 0; i < numSteps; i++) {
        string expression;
        integrator.getComputationStep(i, stepType[i], stepVariable[i], expression);
    }
    vector<int> forceGroup;
    map<string, CustomFunction*> functions;
    CustomIntegratorUtilities::analyzeComputations(context, integrator, expressions, comparisons, blockEnd, invalidatesForces, needsForces, needsEnergy, computeBoth, forceGroup, functions);
    forceGroupFlags.assign(numSteps, integrator.getIntegrationForceGroups());
    stepTarget.assign(numSteps, -2); stepGlobal.assign(numSteps, -1);
    stepProgram.resize(numSteps);
    instructions.clear();
    for (int i = 0; i < numSteps; i++) {
        if (forceGroup[i] > -1) forceGroupFlags[i] = 1 << forceGroup[i];
        if (stepType[i] == CustomIntegrator::WhileBlockStart) blockEnd[blockEnd[i]] = i;       // where the end of the block branches back to
        if (stepType[i] == CustomIntegrator::ComputePerDof || stepType[i] == CustomIntegrator::ComputeSum) {
            stepProgram[i] = translate(expressions[i][0].getRootNode());
            if (stepType[i] == CustomIntegrator::ComputePerDof) {
                if (stepVariable[i] == "x") stepTarget[i] = 0;
                else if (stepVariable[i] == "v") stepTarget[i] = 1;
                else
                    for (int k = 0; k < numPerDof; k++)
                        if (perDofNames[k] == stepVariable[i]) stepTarget[i] = 2 + k;
                if (stepTarget[i] == -2) throw OpenMMException("Illegal per-DOF output variable: " + stepVariable[i]);
            }
            else stepTarget[i] = -1;
        }
        if (stepType[i] == CustomIntegrator::ComputeGlobal || stepType[i] == CustomIntegrator::ComputeSum) {
            stepGlobal[i] = globalIndex(stepVariable[i]);
            if (stepGlobal[i] < 0) throw OpenMMException("Illegal global output variable: " + stepVariable[i]);
        }
    }
    ParsedExpression kinetic = Parser::parse(integrator.getKineticEnergyExpression()).optimize();
    kineticProgram = translate(kinetic.getRootNode());
    kineticNeedsForce = kineticProgram.usesForce;
    HipContext& hip = *data.hip;
    programD.allocate(max((size_t) 16, sizeof(ommhip_vm_instruction) * instructions.size()));
    if (!instructions.empty()) HIP_CHECK(ommhip_memcpy_h2d(programD.ptr, instructions.data(), sizeof(ommhip_vm_instruction) * instructions.size(), hip.stream));
    hip.sync();
    if (data.getDeviceConstraints(context.getSystem()).hasConstraints()) oldPosD.allocate(sizeof(double) * 4 * numAtoms);
    globalsDirty = true;
    compiled = true;
}

void HipIntegrateCustomStepKernel::syncGlobals() {
    if (!globalsDirty) return;
    HipContext& hip = *data.hip;
    if (globalsD.bytes < sizeof(double) * globalValues.size()) globalsD.allocate(sizeof(double) * (globalValues.size() + 16));
    HIP_CHECK(ommhip_memcpy_h2d(globalsD.ptr, globalValues.data(), sizeof(double) * globalValues.size(), hip.stream));
    hip.sync();         // globalValues may change again before the copy has left the host
    globalsDirty = false;
}

double HipIntegrateCustomStepKernel::evaluateOnHost(const ParsedExpression& expression, double energy) {
    map<string, double> variables;
    for (size_t g = 0; g < globalNames.size(); g++) variables[globalNames[g]] = globalValues[g];
    variables["energy"] = energy;
    for (int i = 0; i < 32; i++) { stringstream name; name << "energy" << i; variables[name.str()] = energy; }
    // drawn only when asked for (ReferenceCustomDynamics binds them lazily too): the host sequence stays the Reference kernel's
    map<const ParsedExpression*, int>::iterator uses = hostRandomUse.find(&expression);
    if (uses == hostRandomUse.end())
        uses = hostRandomUse.insert(make_pair(&expression, (treeUsesVariable(expression.getRootNode(), "uniform") ? 1 : 0) |
                                                              (treeUsesVariable(expression.getRootNode(), "gaussian") ? 2 : 0))).first;
    if (uses->second & 1) variables["uniform"] = SimTKOpenMMUtilities::getUniformlyDistributedRandomNumber();
    if (uses->second & 2) variables["gaussian"] = SimTKOpenMMUtilities::getNormallyDistributedRandomNumber();
    return expression.evaluate(variables);
}

ommhip_vm_state HipIntegrateCustomStepKernel::vmState() {
    HipContext& hip = *data.hip;
    ommhip_vm_state s;
    s.num_atoms = numAtoms; s.num_per_dof = numPerDof;
    s.pos = hip.pos.ptr; s.vel = hip.vel.ptr; s.per_dof = perDofD.as<double>();
    s.globals = globalsD.as<double>(); s.program = programD.as<ommhip_vm_instruction>();
    s.seed = data.integratorSeed; s.sum_scratch = sumScratchD.as<double>(); s.sum_result = sumResultD.as<double>();
    return s;
}

void HipIntegrateCustomStepKernel::invalidateForces() {
    forceCached.clear();
    energyCache.clear();
}

void HipIntegrateCustomStepKernel::ensureForces(ContextImpl& context, int flags, bool needForces, bool needEnergy, bool computeForces, bool computeEnergy, bool& forcesAreValid) {
    const bool haveForces = forceCached.find(flags) != forceCached.end(), haveEnergy = energyCache.find(flags) != energyCache.end();
    if ((!needForces || haveForces) && (!needEnergy || haveEnergy)) return;
    flush();                         // the evaluation reads the positions the queued computations are about to write
    HipContext& hip = *data.hip;
    recordChangedParameters(context);
    double e = 0;
    for (int attempt = 0; ; attempt++) {
        const long long recoveriesBefore = hip.overflowRecoveries;
        e = context.calcForcesAndEnergy(computeForces, computeEnergy, flags);
        // a neighbour list that overflowed at this evaluation left incomplete forces: the integration kernels of the native integrators
        // wait for the host on their own (ommhip_integrator_state::freeze_state); here the host looks at once -- a synchronisation per
        // evaluation, only for Systems with a NonbondedForce -- and evaluates again with the list the recovery has grown
        if (hip.listRecovery) hip.listRecovery();
        if (hip.overflowRecoveries == recoveriesBefore) break;
        if (attempt == 4)
            throw OpenMMException("HIP platform: the neighbour list overflowed at five evaluations in a row of one CustomIntegrator step; the forces would be incomplete");
    }
    hip.pendingReplay = 0;
    if (computeForces) {
        DeviceBuffer*& buffer = forceCache[flags];
        if (buffer == NULL) { buffer = new DeviceBuffer(); buffer->allocate(max((size_t) 16, sizeof(double) * 3 * numAtoms)); }
        HIP_CHECK(ommhip_forces_to_atom_order(hip.force.as<long long>(), hip.slotOfAtom.as<int>(), numAtoms, hip.paddedAtoms, buffer->as<double>(), hip.stream));
        forceCached[flags] = true;
    }
    if (computeEnergy) energyCache[flags] = e;
    forcesAreValid = true;
}

void HipIntegrateCustomStepKernel::enqueue(const Program& program, int target, int flags, bool usesForces) {
    if (pending.size() == OMMHIP_VM_MAX_STEPS) flush();
    ommhip_vm_step st;
    st.first = program.first; st.count = program.count; st.target = target; st.uses_random = program.usesRandom;
    st.force = usesForces ? forceCache[flags]->as<double>() : NULL;
    st.draw = program.usesRandom ? data.customDraws++ : 0;
    pending.push_back(st);
}

void HipIntegrateCustomStepKernel::flush() {
    if (pending.empty()) return;
    syncGlobals();
    const ommhip_vm_state s = vmState();
    HIP_CHECK(ommhip_vm_per_dof(&s, (int) pending.size(), pending.data(), data.hip->stream));
    pending.clear();
}

double HipIntegrateCustomStepKernel::runSum(const Program& program, int flags, bool usesForces) {
    flush();
    enqueue(program, -1, flags, usesForces);
    flush();
    double result = 0;
    HipContext& hip = *data.hip;
    HIP_CHECK(ommhip_memcpy_d2h(&result, sumResultD.ptr, sizeof(double), hip.stream));
    hip.sync();
    return result;
}

void HipIntegrateCustomStepKernel::execute(ContextImpl& context, CustomIntegrator& integrator, bool& forcesAreValid) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    if (!compiled) compile(context, integrator);
    HipConstraints& constraints = data.getDeviceConstraints(context.getSystem());
    const double tolerance = integrator.getConstraintTolerance();
    if (globalValues[0] != integrator.getStepSize()) { globalValues[0] = integrator.getStepSize(); globalsDirty = true; }
    loadContextParameters(context);
    // forces kept from earlier steps stand as long as nobody touched the state from outside (ContextImpl tells the integrator: forcesAreValid)
    if (!forcesAreValid || cachePositionsVersion != hip.positionsVersion || cacheBoxVersion != hip.boxVersion) invalidateForces();
    cachePositionsVersion = hip.positionsVersion; cacheBoxVersion = hip.boxVersion;
    if (oldPosD.ptr != NULL) HIP_CHECK(ommhip_memcpy_d2d(oldPosD.ptr, hip.pos.ptr, sizeof(double) * 4 * numAtoms, hip.stream));
    const int numSteps = (int) stepType.size();
    for (int step = 0; step < numSteps; ) {
        const int flags = forceGroupFlags[step];
        if (needsForces[step] || needsEnergy[step])
            ensureForces(context, flags, needsForces[step], needsEnergy[step], needsForces[step] || computeBoth[step], needsEnergy[step] || computeBoth[step], forcesAreValid);
        const double energy = needsEnergy[step] ? energyCache[flags] : 0.0;
        int nextStep = step + 1;
        bool stepInvalidatesForces = invalidatesForces[step];
        switch (stepType[step]) {
            case CustomIntegrator::ComputeGlobal: {
                flush();
                const double result = evaluateOnHost(expressions[step][0], energy);
                if (globalValues[stepGlobal[step]] != result) { globalValues[stepGlobal[step]] = result; globalsDirty = true; }
                break;
            }
            case CustomIntegrator::ComputePerDof: {
                if (stepProgram[step].usesEnergy && globalValues[energySlot] != energy) { flush(); globalValues[energySlot] = energy; globalsDirty = true; }
                if (globalsDirty) flush();          // the queued computations belong to the old values
                enqueue(stepProgram[step], stepTarget[step], flags, needsForces[step]);
                if (stepTarget[step] <= 1) hip.momentumValid = false;
                break;
            }
            case CustomIntegrator::ComputeSum: {
                if (stepProgram[step].usesEnergy && globalValues[energySlot] != energy) { flush(); globalValues[energySlot] = energy; globalsDirty = true; }
                const double sum = runSum(stepProgram[step], flags, needsForces[step]);
                if (globalValues[stepGlobal[step]] != sum) { globalValues[stepGlobal[step]] = sum; globalsDirty = true; }
                break;
            }
            case CustomIntegrator::ConstrainPositions: {
                if (constraints.hasConstraints()) {
                    flush();
                    constraints.apply(hip.pos.ptr, tolerance, oldPosD.ptr);
                    HIP_CHECK(ommhip_memcpy_d2d(oldPosD.ptr, hip.pos.ptr, sizeof(double) * 4 * numAtoms, hip.stream));
                }
                break;
            }
            case CustomIntegrator::ConstrainVelocities: {
                if (constraints.hasConstraints()) {
                    flush();
                    constraints.applyToVelocities(hip.vel.ptr, tolerance, oldPosD.ptr);
                    hip.momentumValid = false;
                }
                break;
            }
            case CustomIntegrator::UpdateContextState: {
                flush();
                recordChangedParameters(context);
                stepInvalidatesForces = context.updateContextState();
                loadContextParameters(context);
                break;
            }
            case CustomIntegrator::IfBlockStart:
            case CustomIntegrator::WhileBlockStart: {
                flush();
                const double lhs = evaluateOnHost(expressions[step][0], energy), rhs = evaluateOnHost(expressions[step][1], energy);
                bool holds = false;
                switch (comparisons[step]) {
                    case CustomIntegratorUtilities::EQUAL: holds = lhs == rhs; break;
                    case CustomIntegratorUtilities::LESS_THAN: holds = lhs < rhs; break;
                    case CustomIntegratorUtilities::GREATER_THAN: holds = lhs > rhs; break;
                    case CustomIntegratorUtilities::NOT_EQUAL: holds = lhs != rhs; break;
                    case CustomIntegratorUtilities::LESS_THAN_OR_EQUAL: holds = lhs <= rhs; break;
                    case CustomIntegratorUtilities::GREATER_THAN_OR_EQUAL: holds = lhs >= rhs; break;
                }
                if (!holds) nextStep = blockEnd[step] + 1;
                break;
            }
            case CustomIntegrator::BlockEnd:
                if (blockEnd[step] != -1) nextStep = blockEnd[step];
                break;
        }
        if (stepInvalidatesForces) {
            forcesAreValid = false;
            invalidateForces();
        }
        step = nextStep;
    }
    flush();
    recordChangedParameters(context);
    if (globalValues[0] != integrator.getStepSize()) integrator.setStepSize(globalValues[0]);
    hip.velocitiesConstrained = false;
    finishStep(globalValues[0]);
}

double HipIntegrateCustomStepKernel::computeKineticEnergy(ContextImpl& context, CustomIntegrator& integrator, bool& forcesAreValid) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    if (!compiled) compile(context, integrator);
    loadContextParameters(context);
    const int flags = integrator.getIntegrationForceGroups();
    if (kineticNeedsForce) {
        if (!forcesAreValid || cachePositionsVersion != hip.positionsVersion || cacheBoxVersion != hip.boxVersion) invalidateForces();
        cachePositionsVersion = hip.positionsVersion; cacheBoxVersion = hip.boxVersion;
        ensureForces(context, flags, true, false, true, false, forcesAreValid);
    }
    return runSum(kineticProgram, flags, kineticNeedsForce);
}

void HipIntegrateCustomStepKernel::getGlobalVariables(ContextImpl& context, vector<double>& values) const {
    values.assign(globalValues.begin() + 1, globalValues.begin() + 1 + numIntegratorGlobals);
}

void HipIntegrateCustomStepKernel::setGlobalVariables(ContextImpl& context, const vector<double>& values) {
    for (int i = 0; i < numIntegratorGlobals && i < (int) values.size(); i++)
        if (globalValues[1 + i] != values[i]) { globalValues[1 + i] = values[i]; globalsDirty = true; }
}

void HipIntegrateCustomStepKernel::getPerDofVariable(ContextImpl& context, int variable, vector<Vec3>& values) const {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    vector<double> tmp(3 * (size_t) numAtoms);
    if (numAtoms > 0) HIP_CHECK(ommhip_memcpy_d2h(tmp.data(), perDofD.as<double>() + (size_t) variable * 3 * numAtoms, sizeof(double) * tmp.size(), hip.stream));
    hip.sync();
    values.resize(numAtoms);
    for (int i = 0; i < numAtoms; i++) values[i] = Vec3(tmp[3 * i], tmp[3 * i + 1], tmp[3 * i + 2]);
}

void HipIntegrateCustomStepKernel::setPerDofVariable(ContextImpl& context, int variable, const vector<Vec3>& values) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    vector<double> tmp(3 * (size_t) numAtoms, 0.0);
    for (int i = 0; i < numAtoms && i < (int) values.size(); i++) { tmp[3 * i] = values[i][0]; tmp[3 * i + 1] = values[i][1]; tmp[3 * i + 2] = values[i][2]; }
    if (numAtoms > 0) HIP_CHECK(ommhip_memcpy_h2d(perDofD.as<double>() + (size_t) variable * 3 * numAtoms, tmp.data(), sizeof(double) * tmp.size(), hip.stream));
    hip.sync();
}

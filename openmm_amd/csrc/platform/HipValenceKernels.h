#ifndef OPENMM_HIP_VALENCE_KERNELS_H_
#define OPENMM_HIP_VALENCE_KERNELS_H_
/* Native kernels for the Custom*Forces an AMOEBA force field is made of.  The reference's platforms compile ANY energy expression at run
 * time (CUDA: nvrtc); this platform has hand-written kernels (kernels/valence.hip) for the expressions that
 * wrappers/python/openmm/app/forcefield.py writes for the AMOEBA valence terms, recognises those expressions (HipValenceForm) and leaves
 * every other Custom*Force to its Reference kernel as a fallback force.  A kernel object learns which Force it serves only in
 * initialize(), so each of the classes below carries the Reference kernel and hands everything over to it when the expression is not one
 * of the known forms (HipPlatform's classification asks the same question, HipValenceForm::isNative, to decide whether the Context has
 * fallback forces). */
#include "HipPlatform.h"
#include "HipContext.h"
#include "openmm/CustomAngleForce.h"
#include "openmm/CustomBondForce.h"
#include "openmm/CustomCompoundBondForce.h"
#include "openmm/kernels.h"
#include <string>
#include <vector>

namespace OpenMM {

/** What an energy expression was recognised as: the kind of kernels/valence.hip, its per-Force coefficients, and where each per-term
 *  parameter of the kernel is found among the Force's per-bond parameters. */
struct HipValenceForm {
    int kind;                       // OMMHIP_VALENCE_*, -1: not recognised
    double coefficients[6];
    std::vector<int> paramIndex;    // kernel parameter p = per-bond parameter paramIndex[p] of the Force
    HipValenceForm() : kind(-1) { for (int i = 0; i < 6; i++) coefficients[i] = 0; }
    static HipValenceForm recognise(const CustomBondForce& force);
    /** kind of a CustomBondForce whose expression is none of the known forms but can be interpreted on the device (HipInterpretedBonds) */
    static const int INTERPRETED_BOND = 100, INTERPRETED_ANGLE = 101;
    static HipValenceForm recognise(const CustomAngleForce& force);
    static HipValenceForm recognise(const CustomCompoundBondForce& force);
    /** Is this Force one of the three classes above with a recognised expression? */
    static bool isNative(const Force& force);
};

/** One list of terms on the device and its launch (through HipContext::addValence: all lists of an evaluation share a launch). */
class HipValenceTerms {
public:
    HipValenceTerms(HipPlatform::PlatformData& data) : data(data), numTerms(0) {}
    void upload(const HipValenceForm& form, int atomsPerTerm, const std::vector<int>& atoms, const std::vector<double>& params);
    void uploadParams(const std::vector<double>& params);
    void uploadGrids(const std::vector<double>& grids);
    void execute(bool includeEnergy);
private:
    HipPlatform::PlatformData& data;
    HipValenceForm form;
    int numTerms, paramsPerTerm = 0;
    DeviceBuffer atomsD, paramsD, gridsD;
};

/** A CustomBondForce with ANY expression of r, per-bond and global parameters (no tabulated functions, no energy parameter derivatives):
 *  the energy and its symbolic derivative with respect to r as two programs of the CustomIntegrator's interpreter
 *  (kernels/custom_integrator.hip, ommhip_vm_bond_forces).  The reference's GPU platforms compile such expressions at run time. */
class HipInterpretedBonds {
public:
    HipInterpretedBonds(HipPlatform::PlatformData& data) : data(data) {}
    /** Can this Force be interpreted?  (parses, only built-in operations, shallow enough for the interpreter's stack) */
    static bool supports(const CustomBondForce& force);
    void initialize(const CustomBondForce& force);
    void uploadParams(const CustomBondForce& force);
    /** the same for a CustomAngleForce: an expression of theta (ommhip_vm_angle_forces) */
    static bool supports(const CustomAngleForce& force);
    void initialize(const CustomAngleForce& force);
    void uploadParams(const CustomAngleForce& force);
    void execute(ContextImpl& context, bool includeEnergy);
private:
    /** what the two Force classes have in common, as the translation needs it */
    struct Description {
        std::string expression, variable;
        std::vector<std::string> perTerm, globals;
        std::vector<double> globalDefaults;
        int atomsPerTerm, numDerivatives;
        bool periodic;
    };
    static Description describe(const CustomBondForce& force);
    static Description describe(const CustomAngleForce& force);
    static bool translate(const Description& d, std::vector<ommhip_vm_instruction>& program, int counts[4]);
    void setup(const Description& d, const std::vector<int>& atoms);
    void uploadParamTable(const std::vector<std::vector<double> >& perTerm);
    HipPlatform::PlatformData& data;
    int numBonds = 0, numParams = 0, stride = 3, atomsPerTerm = 2, counts[4] = {0, 0, 0, 0};
    bool periodic = false;
    std::vector<std::string> globalNames;
    std::vector<double> globalValues;
    DeviceBuffer atomsD, paramsD, programD, globalsD;
};

class HipCalcCustomBondForceKernel : public CalcCustomBondForceKernel {
public:
    HipCalcCustomBondForceKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data, KernelImpl* referenceKernel) : CalcCustomBondForceKernel(name, platform), reference(dynamic_cast<CalcCustomBondForceKernel*>(referenceKernel)), terms(data), interpreted(data) {}
    ~HipCalcCustomBondForceKernel() { delete reference; }
    void initialize(const System& system, const CustomBondForce& force);
    double execute(ContextImpl& context, bool includeForces, bool includeEnergy);
    void copyParametersToContext(ContextImpl& context, const CustomBondForce& force);
private:
    void collect(const CustomBondForce& force, std::vector<int>* atoms, std::vector<double>& params) const;
    CalcCustomBondForceKernel* reference;      // the Reference kernel: every expression without a native form is its business (a fallback force), owned
    HipValenceForm form;
    HipValenceTerms terms;
    HipInterpretedBonds interpreted;
};

class HipCalcCustomAngleForceKernel : public CalcCustomAngleForceKernel {
public:
    HipCalcCustomAngleForceKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data, KernelImpl* referenceKernel) : CalcCustomAngleForceKernel(name, platform), reference(dynamic_cast<CalcCustomAngleForceKernel*>(referenceKernel)), terms(data), interpreted(data) {}
    ~HipCalcCustomAngleForceKernel() { delete reference; }
    void initialize(const System& system, const CustomAngleForce& force);
    double execute(ContextImpl& context, bool includeForces, bool includeEnergy);
    void copyParametersToContext(ContextImpl& context, const CustomAngleForce& force);
private:
    void collect(const CustomAngleForce& force, std::vector<int>* atoms, std::vector<double>& params) const;
    CalcCustomAngleForceKernel* reference;      // the Reference kernel: every expression without a native form is its business (a fallback force), owned
    HipValenceForm form;
    HipValenceTerms terms;
    HipInterpretedBonds interpreted;
};

class HipCalcCustomCompoundBondForceKernel : public CalcCustomCompoundBondForceKernel {
public:
    HipCalcCustomCompoundBondForceKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data, KernelImpl* referenceKernel) : CalcCustomCompoundBondForceKernel(name, platform), reference(dynamic_cast<CalcCustomCompoundBondForceKernel*>(referenceKernel)), terms(data) {}
    ~HipCalcCustomCompoundBondForceKernel() { delete reference; }
    void initialize(const System& system, const CustomCompoundBondForce& force);
    double execute(ContextImpl& context, bool includeForces, bool includeEnergy);
    void copyParametersToContext(ContextImpl& context, const CustomCompoundBondForce& force);
private:
    void collect(const CustomCompoundBondForce& force, std::vector<int>* atoms, std::vector<double>& params) const;
    CalcCustomCompoundBondForceKernel* reference;      // the Reference kernel: every expression without a native form is its business (a fallback force), owned
    HipValenceForm form;
    HipValenceTerms terms;
};

}  // namespace OpenMM

#endif

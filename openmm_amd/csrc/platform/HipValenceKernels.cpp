/* See HipValenceKernels.h.  Recognition: an energy expression is reduced to its SHAPE -- white space removed, every numeric literal
 * replaced by '#' (a sign that follows an operator or an opening parenthesis belongs to the literal) -- and the literals are collected in
 * order; a shape equal to one of the shapes below, with the fixed literals (the exponents) in place, is that form, and the remaining
 * literals are its coefficients.  The shapes are those of the strings wrappers/python/openmm/app/forcefield.py builds (:3368, :3502, :3565,
 * :3730, :4039, :4428). */
#include "HipValenceKernels.h"
#include "lepton/CompiledExpression.h"
#include "lepton/ExpressionTreeNode.h"
#include "lepton/Operation.h"
#include "lepton/ParsedExpression.h"
#include "lepton/Parser.h"
#include "openmm/OpenMMException.h"
#include "openmm/internal/ContextImpl.h"
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>

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

/* -> the shape of an expression; its literals go to `numbers` */
string shapeOf(const string& expression, vector<double>& numbers) {
    string text;
    for (size_t i = 0; i < expression.size(); i++)
        if (!isspace((unsigned char) expression[i])) text += expression[i];
    string shape;
    numbers.clear();
    size_t i = 0;
    while (i < text.size()) {
        const char c = text[i];
        if (isalpha((unsigned char) c) || c == '_') {            // an identifier (may contain digits)
            while (i < text.size() && (isalnum((unsigned char) text[i]) || text[i] == '_')) shape += text[i++];
            continue;
        }
        const bool signStart = (c == '-' || c == '+') && i + 1 < text.size() && (isdigit((unsigned char) text[i + 1]) || text[i + 1] == '.') &&
                               (shape.empty() || string("+-*/^(,=;").find(shape[shape.size() - 1]) != string::npos);
        if (isdigit((unsigned char) c) || c == '.' || signStart) {
            char* end = NULL;
            numbers.push_back(strtod(text.c_str() + i, &end));
            i = end - text.c_str();
            shape += '#';
            continue;
        }
        shape += c;
        i++;
    }
    return shape;
}

const char* PLANE = "projx=x2-nx*dot;projy=y2-ny*dot;projz=z2-nz*dot;dot=nx*(x2-x3)+ny*(y2-y3)+nz*(z2-z3);nx=px/norm;ny=py/norm;nz=pz/norm;"
                    "norm=sqrt(px*px+py*py+pz*pz);px=(d1y*d2z-d1z*d2y);py=(d1z*d2x-d1x*d2z);pz=(d1x*d2y-d1y*d2x);"
                    "d1x=x1-x4;d1y=y1-y4;d1z=z1-z4;d2x=x3-x4;d2y=y3-y4;d2z=z3-z4";

bool isRadian(double v) { return fabs(v - 180.0 / M_PI) < 1e-9; }

/* parameter names -> indices in the Force's list; false if one is missing or the Force has others */
bool mapParameters(const vector<string>& have, const vector<string>& want, vector<int>& index) {
    if (have.size() != want.size()) return false;
    index.clear();
    for (size_t w = 0; w < want.size(); w++) {
        int found = -1;
        for (size_t h = 0; h < have.size(); h++)
            if (have[h] == want[w]) found = (int) h;
        if (found < 0) return false;
        index.push_back(found);
    }
    return true;
}

vector<string> names(const char* a, const char* b = NULL, const char* c = NULL, const char* d = NULL, const char* e = NULL) {
    vector<string> v;
    const char* all[5] = {a, b, c, d, e};
    for (int i = 0; i < 5; i++) if (all[i] != NULL) v.push_back(all[i]);
    return v;
}

/* "k*(x^2+#*x^3+#*x^4+#*x^5+#*x^6)" literals: 2 c0 3 c1 4 c2 5 c3 6 from position `at` */
bool sexticCoefficients(const vector<double>& n, size_t at, double* c) {
    if (n.size() < at + 9) return false;
    if (n[at] != 2 || n[at + 2] != 3 || n[at + 4] != 4 || n[at + 6] != 5 || n[at + 8] != 6) return false;
    c[0] = n[at + 1]; c[1] = n[at + 3]; c[2] = n[at + 5]; c[3] = n[at + 7];
    return true;
}

}  // namespace

HipValenceForm HipValenceForm::recognise(const CustomBondForce& force) {
    HipValenceForm f, none;
    static const bool noInterpreter = getenv("OPENMM_HIP_NO_INTERPRETED_FORCES") != NULL;       // A/B knob: only the hand-written forms
    if (!noInterpreter && HipInterpretedBonds::supports(force)) none.kind = INTERPRETED_BOND;      // what an expression is when it is not the form below
    if (force.getNumGlobalParameters() > 0 || force.getNumEnergyParameterDerivatives() > 0 || force.usesPeriodicBoundaryConditions()) return none;
    vector<double> n;
    if (shapeOf(force.getEnergyFunction(), n) != "k*(d^#+#*d^#+#*d^#);d=r-r0" || n.size() != 5 || n[0] != 2 || n[2] != 3 || n[4] != 4) return none;
    vector<string> have;
    for (int i = 0; i < force.getNumPerBondParameters(); i++) have.push_back(force.getPerBondParameterName(i));
    if (!mapParameters(have, names("r0", "k"), f.paramIndex)) return none;
    f.kind = OMMHIP_VALENCE_POLY_BOND;
    f.coefficients[0] = n[1]; f.coefficients[1] = n[3];
    return f;
}

HipValenceForm HipValenceForm::recognise(const CustomAngleForce& force) {
    HipValenceForm f, none;
    static const bool noInterpreter = getenv("OPENMM_HIP_NO_INTERPRETED_FORCES") != NULL;
    if (!noInterpreter && HipInterpretedBonds::supports(force)) none.kind = INTERPRETED_ANGLE;
    if (force.getNumGlobalParameters() > 0 || force.getNumEnergyParameterDerivatives() > 0 || force.usesPeriodicBoundaryConditions()) return none;
    vector<double> n;
    if (shapeOf(force.getEnergyFunction(), n) != "k*(d^#+#*d^#+#*d^#+#*d^#+#*d^#);d=#*theta-theta0" || n.size() != 10) return none;
    if (!sexticCoefficients(n, 0, f.coefficients) || !isRadian(n[9])) return none;
    vector<string> have;
    for (int i = 0; i < force.getNumPerAngleParameters(); i++) have.push_back(force.getPerAngleParameterName(i));
    if (!mapParameters(have, names("theta0", "k"), f.paramIndex)) return none;
    f.kind = OMMHIP_VALENCE_POLY_ANGLE;
    f.coefficients[4] = n[9];
    return f;
}

HipValenceForm HipValenceForm::recognise(const CustomCompoundBondForce& force) {
    HipValenceForm f, none;
    if (force.getNumGlobalParameters() > 0 || force.getNumEnergyParameterDerivatives() > 0 || force.usesPeriodicBoundaryConditions() ||
            force.getNumTabulatedFunctions() > 0)
        return none;
    vector<double> n;
    const string shape = shapeOf(force.getEnergyFunction(), n);
    vector<string> have;
    for (int i = 0; i < force.getNumPerBondParameters(); i++) have.push_back(force.getPerBondParameterName(i));
    const int particles = force.getNumParticlesPerBond();
    const string plane = PLANE;
    if (particles == 4 && shape == "k*(d^#+#*d^#+#*d^#+#*d^#+#*d^#);d=theta-theta0;theta=#*pointangle(x1,y1,z1,projx,projy,projz,x3,y3,z3);" + plane) {
        if (n.size() != 10 || !sexticCoefficients(n, 0, f.coefficients) || !isRadian(n[9]) || !mapParameters(have, names("theta0", "k"), f.paramIndex)) return none;
        f.kind = OMMHIP_VALENCE_INPLANE_ANGLE;
        f.coefficients[4] = n[9];
        return f;
    }
    if (particles == 4 && shape == "k*(theta^#+#*theta^#+#*theta^#+#*theta^#+#*theta^#);theta=#*pointangle(x2,y2,z2,x4,y4,z4,projx,projy,projz);" + plane) {
        if (n.size() != 10 || !sexticCoefficients(n, 0, f.coefficients) || !isRadian(n[9]) || !mapParameters(have, names("k"), f.paramIndex)) return none;
        f.kind = OMMHIP_VALENCE_OUT_OF_PLANE_BEND;
        f.coefficients[4] = n[9];
        return f;
    }
    if (particles == 3 && shape == "(k1*(distance(p1,p2)-r12)+k2*(distance(p2,p3)-r23))*(#*(angle(p1,p2,p3)-theta0))") {
        if (n.size() != 1 || !mapParameters(have, names("r12", "r23", "theta0", "k1", "k2"), f.paramIndex)) return none;
        f.kind = OMMHIP_VALENCE_STRETCH_BEND;
        f.coefficients[0] = n[0];
        return f;
    }
    if (particles == 6 && shape == "#*k*sin(phi)^#;phi=pointdihedral(x3+c1x,y3+c1y,z3+c1z,x3,y3,z3,x4,y4,z4,x4+c2x,y4+c2y,z4+c2z);"
                                   "c1x=(d14y*d24z-d14z*d24y);c1y=(d14z*d24x-d14x*d24z);c1z=(d14x*d24y-d14y*d24x);"
                                   "c2x=(d53y*d63z-d53z*d63y);c2y=(d53z*d63x-d53x*d63z);c2z=(d53x*d63y-d53y*d63x);"
                                   "d14x=x1-x4;d14y=y1-y4;d14z=z1-z4;d24x=x2-x4;d24y=y2-y4;d24z=z2-z4;"
                                   "d53x=x5-x3;d53y=y5-y3;d53z=z5-z3;d63x=x6-x3;d63y=y6-y3;d63z=z6-z3") {
        if (n.size() != 2 || n[0] != 2 || n[1] != 2 || !mapParameters(have, names("k"), f.paramIndex)) return none;
        f.kind = OMMHIP_VALENCE_PI_TORSION;
        return f;
    }
    return none;
}

bool HipValenceForm::isNative(const Force& force) {
    static const bool off = getenv("OPENMM_HIP_REFERENCE_CUSTOM_FORCES") != NULL && getenv("OPENMM_HIP_REFERENCE_CUSTOM_FORCES")[0] == '1';      // A/B knob: everything to the Reference kernels
    if (off) return false;
    if (const CustomBondForce* b = dynamic_cast<const CustomBondForce*>(&force)) return recognise(*b).kind >= 0;
    if (const CustomAngleForce* a = dynamic_cast<const CustomAngleForce*>(&force)) return recognise(*a).kind >= 0;
    if (const CustomCompoundBondForce* c = dynamic_cast<const CustomCompoundBondForce*>(&force)) return recognise(*c).kind >= 0;
    return false;
}

// ================================================================================================
void HipValenceTerms::upload(const HipValenceForm& f, int atomsPerTerm, const vector<int>& atoms, const vector<double>& params) {
    data.hip->setAsCurrent();
    form = f;
    numTerms = (int) atoms.size() / atomsPerTerm;
    paramsPerTerm = numTerms > 0 ? (int) params.size() / numTerms : 0;
    uploadVector(atomsD, atoms, data.hip->stream);
    uploadVector(paramsD, params, data.hip->stream);
}
void HipValenceTerms::uploadParams(const vector<double>& params) {
    data.hip->setAsCurrent();
    if ((int) params.size() != numTerms * paramsPerTerm)
        throw OpenMMException("updateParametersInContext: The number of terms has changed");
    uploadVector(paramsD, params, data.hip->stream);
}
void HipValenceTerms::uploadGrids(const vector<double>& grids) {
    data.hip->setAsCurrent();
    uploadVector(gridsD, grids, data.hip->stream);
}
void HipValenceTerms::execute(bool includeEnergy) {
    if (numTerms == 0) return;
    ommhip_valence_list l;
    l.kind = form.kind; l.num_terms = numTerms; l.atoms = atomsD.as<int>(); l.params = paramsD.as<double>();
    for (int i = 0; i < 6; i++) l.coefficients[i] = form.coefficients[i];
    l.grids = gridsD.as<double>();
    data.hip->addValence(l, includeEnergy);
}

// ================================================================================================
// ================================================================================================ interpreted CustomBondForce
namespace {
/* postfix form of an expression tree (the interpreter's operation codes follow Lepton's ids from ADD on); false if something has no device form */
bool emitBondProgram(const Lepton::ExpressionTreeNode& node, const string& variable, const vector<string>& perBond, const vector<string>& globals, vector<ommhip_vm_instruction>& out, int& depth, int& maxDepth) {
    using Lepton::Operation;
    for (size_t i = 0; i < node.getChildren().size(); i++)
        if (!emitBondProgram(node.getChildren()[i], variable, perBond, globals, out, depth, maxDepth)) return false;
    const Operation& op = node.getOperation();
    ommhip_vm_instruction in = {0, 0, 0.0};
    switch (op.getId()) {
        case Operation::CONSTANT: in.op = OMMHIP_VM_CONSTANT; in.value = dynamic_cast<const Operation::Constant&>(op).getValue(); break;
        case Operation::VARIABLE: {
            const string name = op.getName();
            in.op = OMMHIP_VM_VARIABLE; in.arg = -1;
            if (name == variable) in.arg = 0;
            for (size_t k = 0; k < perBond.size() && in.arg < 0; k++) if (perBond[k] == name) in.arg = 6 + (int) k;
            for (size_t k = 0; k < globals.size() && in.arg < 0; k++) if (globals[k] == name) { in.op = OMMHIP_VM_GLOBAL; in.arg = (int) k; }
            if (in.arg < 0) return false;
            break;
        }
        case Operation::ADD_CONSTANT: in.op = OMMHIP_VM_ADD_CONSTANT; in.value = dynamic_cast<const Operation::AddConstant&>(op).getValue(); break;
        case Operation::MULTIPLY_CONSTANT: in.op = OMMHIP_VM_MULTIPLY_CONSTANT; in.value = dynamic_cast<const Operation::MultiplyConstant&>(op).getValue(); break;
        case Operation::POWER_CONSTANT: in.op = OMMHIP_VM_POWER_CONSTANT; in.value = dynamic_cast<const Operation::PowerConstant&>(op).getValue(); break;
        case Operation::CUSTOM: return false;
        default:
            if ((int) op.getId() < (int) Operation::ADD || (int) op.getId() > (int) Operation::SELECT) return false;
            in.op = OMMHIP_VM_ADD + ((int) op.getId() - (int) Operation::ADD);
            break;
    }
    out.push_back(in);
    depth += 1 - (int) node.getChildren().size();
    maxDepth = max(maxDepth, depth);
    return true;
}
}  // namespace

HipInterpretedBonds::Description HipInterpretedBonds::describe(const CustomBondForce& force) {
    Description d;
    d.expression = force.getEnergyFunction(); d.variable = "r"; d.atomsPerTerm = 2;
    d.numDerivatives = force.getNumEnergyParameterDerivatives(); d.periodic = force.usesPeriodicBoundaryConditions();
    for (int i = 0; i < force.getNumPerBondParameters(); i++) d.perTerm.push_back(force.getPerBondParameterName(i));
    for (int i = 0; i < force.getNumGlobalParameters(); i++) { d.globals.push_back(force.getGlobalParameterName(i)); d.globalDefaults.push_back(force.getGlobalParameterDefaultValue(i)); }
    return d;
}
HipInterpretedBonds::Description HipInterpretedBonds::describe(const CustomAngleForce& force) {
    Description d;
    d.expression = force.getEnergyFunction(); d.variable = "theta"; d.atomsPerTerm = 3;
    d.numDerivatives = force.getNumEnergyParameterDerivatives(); d.periodic = force.usesPeriodicBoundaryConditions();
    for (int i = 0; i < force.getNumPerAngleParameters(); i++) d.perTerm.push_back(force.getPerAngleParameterName(i));
    for (int i = 0; i < force.getNumGlobalParameters(); i++) { d.globals.push_back(force.getGlobalParameterName(i)); d.globalDefaults.push_back(force.getGlobalParameterDefaultValue(i)); }
    return d;
}

bool HipInterpretedBonds::translate(const Description& d, vector<ommhip_vm_instruction>& program, int counts[4]) {
    if (d.numDerivatives > 0) return false;
    try {
        // the expressions the Reference kernel evaluates (ReferenceKernels.cpp, CalcCustomBondForceKernel / CalcCustomAngleForceKernel::initialize):
        // E and its derivative with respect to r / theta, optimised
        Lepton::ParsedExpression energy = Lepton::Parser::parse(d.expression).optimize();
        Lepton::ParsedExpression deriv = energy.differentiate(d.variable).optimize();
        program.clear();
        int depth = 0, maxDepth = 0;
        counts[0] = 0;
        if (!emitBondProgram(energy.getRootNode(), d.variable, d.perTerm, d.globals, program, depth, maxDepth)) return false;
        counts[1] = (int) program.size();
        counts[2] = counts[1];
        depth = 0;
        if (!emitBondProgram(deriv.getRootNode(), d.variable, d.perTerm, d.globals, program, depth, maxDepth)) return false;
        counts[3] = (int) program.size() - counts[2];
        return maxDepth <= OMMHIP_VM_STACK;
    }
    catch (const std::exception&) { return false; }
}

bool HipInterpretedBonds::supports(const CustomBondForce& force) {
    vector<ommhip_vm_instruction> program;
    int counts[4];
    return translate(describe(force), program, counts);
}
bool HipInterpretedBonds::supports(const CustomAngleForce& force) {
    vector<ommhip_vm_instruction> program;
    int counts[4];
    return translate(describe(force), program, counts);
}

void HipInterpretedBonds::setup(const Description& d, const vector<int>& atoms) {
    data.hip->setAsCurrent();
    vector<ommhip_vm_instruction> program;
    if (!translate(d, program, counts)) throw OpenMMException("HIP platform: internal error: a Custom*Force that cannot be interpreted");
    atomsPerTerm = d.atomsPerTerm;
    numBonds = (int) atoms.size() / atomsPerTerm; numParams = (int) d.perTerm.size();
    stride = (max(numBonds, 1) + 2) / 3 * 3;
    periodic = d.periodic;
    uploadVector(atomsD, atoms, data.hip->stream);
    uploadVector(programD, program, data.hip->stream);
    globalNames = d.globals; globalValues = d.globalDefaults;
    uploadVector(globalsD, globalValues, data.hip->stream);
}

void HipInterpretedBonds::uploadParamTable(const vector<vector<double> >& perTerm) {
    data.hip->setAsCurrent();
    if ((int) perTerm.size() != numBonds) throw OpenMMException("updateParametersInContext: The number of terms has changed");
    vector<double> params((size_t) max(numParams, 1) * stride, 0.0);
    for (int i = 0; i < numBonds; i++)
        for (int k = 0; k < numParams; k++) params[(size_t) k * stride + i] = perTerm[i][k];
    uploadVector(paramsD, params, data.hip->stream);
}

void HipInterpretedBonds::initialize(const CustomBondForce& force) {
    vector<int> atoms(2 * (size_t) force.getNumBonds());
    for (int i = 0; i < force.getNumBonds(); i++) { vector<double> p; force.getBondParameters(i, atoms[2 * i], atoms[2 * i + 1], p); }
    setup(describe(force), atoms);
    uploadParams(force);
}
void HipInterpretedBonds::uploadParams(const CustomBondForce& force) {
    vector<vector<double> > table(force.getNumBonds());
    for (int i = 0; i < force.getNumBonds(); i++) { int p1, p2; force.getBondParameters(i, p1, p2, table[i]); }
    uploadParamTable(table);
}
void HipInterpretedBonds::initialize(const CustomAngleForce& force) {
    vector<int> atoms(3 * (size_t) force.getNumAngles());
    for (int i = 0; i < force.getNumAngles(); i++) { vector<double> p; force.getAngleParameters(i, atoms[3 * i], atoms[3 * i + 1], atoms[3 * i + 2], p); }
    setup(describe(force), atoms);
    uploadParams(force);
}
void HipInterpretedBonds::uploadParams(const CustomAngleForce& force) {
    vector<vector<double> > table(force.getNumAngles());
    for (int i = 0; i < force.getNumAngles(); i++) { int p1, p2, p3; force.getAngleParameters(i, p1, p2, p3, table[i]); }
    uploadParamTable(table);
}

static long long interpretedBondLaunches = 0;
/* test hook: launches of the interpreted CustomBondForce kernel by this process so far */
extern "C" __attribute__((visibility("default"))) long long ommhip_plugin_interpreted_bond_launches() { return interpretedBondLaunches; }

void HipInterpretedBonds::execute(ContextImpl& context, bool includeEnergy) {
    if (numBonds == 0) return;
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    bool changed = false;
    for (size_t i = 0; i < globalNames.size(); i++) {
        const double v = context.getParameter(globalNames[i]);
        if (v != globalValues[i]) { globalValues[i] = v; changed = true; }
    }
    if (changed) uploadVector(globalsD, globalValues, hip.stream);
    hip.ensureCleared();
    ommhip_vm_bonds b;
    b.num_bonds = numBonds; b.num_params = numParams; b.param_stride = stride; b.periodic = periodic ? 1 : 0;
    b.atoms = atomsD.as<int>(); b.params = paramsD.as<double>(); b.program = (const ommhip_vm_instruction*) programD.ptr;
    b.energy_first = counts[0]; b.energy_count = counts[1]; b.deriv_first = counts[2]; b.deriv_count = counts[3];
    b.globals = globalsD.as<double>();
    for (int k = 0; k < 6; k++) b.box[k] = hip.box[k];
    interpretedBondLaunches++;
    HIP_CHECK((atomsPerTerm == 2 ? ommhip_vm_bond_forces : ommhip_vm_angle_forces)(&b, hip.pos.ptr, hip.slotOfAtom.as<int>(), hip.paddedAtoms, hip.force.as<long long>(),
                                                                                    hip.energyBuffer.as<double>(), HipContext::EnergySlots, includeEnergy ? 1 : 0, hip.stream));
}

void HipCalcCustomBondForceKernel::collect(const CustomBondForce& force, vector<int>* atoms, vector<double>& params) const {
    for (int i = 0; i < force.getNumBonds(); i++) {
        int p1, p2; vector<double> p;
        force.getBondParameters(i, p1, p2, p);
        if (atoms != NULL) { atoms->push_back(p1); atoms->push_back(p2); }
        for (size_t k = 0; k < form.paramIndex.size(); k++) params.push_back(p[form.paramIndex[k]]);
    }
}
void HipCalcCustomBondForceKernel::initialize(const System& system, const CustomBondForce& force) {
    form = HipValenceForm::isNative(force) ? HipValenceForm::recognise(force) : HipValenceForm();
    if (getenv("OPENMM_HIP_VALENCE_DEBUG") != NULL)          // diagnostics: which of the three ways this Force goes
        fprintf(stderr, "HIP platform: CustomBondForce \"%s\": %s\n", force.getEnergyFunction().c_str(),
                form.kind < 0 ? "Reference kernel (fallback force)" : (form.kind == HipValenceForm::INTERPRETED_BOND ? "interpreted on the device" : "hand-written kernel"));
    if (form.kind < 0) {
        if (reference == NULL) throw OpenMMException("HIP platform: no kernel for this CustomBondForce");
        reference->initialize(system, force);
        return;
    }
    if (form.kind == HipValenceForm::INTERPRETED_BOND) { interpreted.initialize(force); return; }
    vector<int> atoms; vector<double> params;
    collect(force, &atoms, params);
    terms.upload(form, 2, atoms, params);
}
double HipCalcCustomBondForceKernel::execute(ContextImpl& context, bool includeForces, bool includeEnergy) {
    if (form.kind < 0) return reference->execute(context, includeForces, includeEnergy);
    if (form.kind == HipValenceForm::INTERPRETED_BOND) { interpreted.execute(context, includeEnergy); return 0.0; }
    terms.execute(includeEnergy);
    return 0.0;
}
void HipCalcCustomBondForceKernel::copyParametersToContext(ContextImpl& context, const CustomBondForce& force) {
    if (form.kind < 0) { reference->copyParametersToContext(context, force); return; }
    if (form.kind == HipValenceForm::INTERPRETED_BOND) { interpreted.uploadParams(force); return; }
    vector<double> params;
    collect(force, NULL, params);
    terms.uploadParams(params);
}

void HipCalcCustomAngleForceKernel::collect(const CustomAngleForce& force, vector<int>* atoms, vector<double>& params) const {
    for (int i = 0; i < force.getNumAngles(); i++) {
        int p1, p2, p3; vector<double> p;
        force.getAngleParameters(i, p1, p2, p3, p);
        if (atoms != NULL) { atoms->push_back(p1); atoms->push_back(p2); atoms->push_back(p3); }
        for (size_t k = 0; k < form.paramIndex.size(); k++) params.push_back(p[form.paramIndex[k]]);
    }
}
void HipCalcCustomAngleForceKernel::initialize(const System& system, const CustomAngleForce& force) {
    form = HipValenceForm::isNative(force) ? HipValenceForm::recognise(force) : HipValenceForm();
    if (getenv("OPENMM_HIP_VALENCE_DEBUG") != NULL)
        fprintf(stderr, "HIP platform: CustomAngleForce \"%s\": %s\n", force.getEnergyFunction().c_str(),
                form.kind < 0 ? "Reference kernel (fallback force)" : (form.kind == HipValenceForm::INTERPRETED_ANGLE ? "interpreted on the device" : "hand-written kernel"));
    if (form.kind < 0) {
        if (reference == NULL) throw OpenMMException("HIP platform: no kernel for this CustomAngleForce");
        reference->initialize(system, force);
        return;
    }
    if (form.kind == HipValenceForm::INTERPRETED_ANGLE) { interpreted.initialize(force); return; }
    vector<int> atoms; vector<double> params;
    collect(force, &atoms, params);
    terms.upload(form, 3, atoms, params);
}
double HipCalcCustomAngleForceKernel::execute(ContextImpl& context, bool includeForces, bool includeEnergy) {
    if (form.kind < 0) return reference->execute(context, includeForces, includeEnergy);
    if (form.kind == HipValenceForm::INTERPRETED_ANGLE) { interpreted.execute(context, includeEnergy); return 0.0; }
    terms.execute(includeEnergy);
    return 0.0;
}
void HipCalcCustomAngleForceKernel::copyParametersToContext(ContextImpl& context, const CustomAngleForce& force) {
    if (form.kind < 0) { reference->copyParametersToContext(context, force); return; }
    if (form.kind == HipValenceForm::INTERPRETED_ANGLE) { interpreted.uploadParams(force); return; }
    vector<double> params;
    collect(force, NULL, params);
    terms.uploadParams(params);
}

void HipCalcCustomCompoundBondForceKernel::collect(const CustomCompoundBondForce& force, vector<int>* atoms, vector<double>& params) const {
    for (int i = 0; i < force.getNumBonds(); i++) {
        vector<int> particles; vector<double> p;
        force.getBondParameters(i, particles, p);
        if (atoms != NULL) atoms->insert(atoms->end(), particles.begin(), particles.end());
        for (size_t k = 0; k < form.paramIndex.size(); k++) params.push_back(p[form.paramIndex[k]]);
    }
}
void HipCalcCustomCompoundBondForceKernel::initialize(const System& system, const CustomCompoundBondForce& force) {
    form = HipValenceForm::isNative(force) ? HipValenceForm::recognise(force) : HipValenceForm();
    if (form.kind < 0) {
        if (reference == NULL) throw OpenMMException("HIP platform: no kernel for this CustomCompoundBondForce");
        reference->initialize(system, force);
        return;
    }
    vector<int> atoms; vector<double> params;
    collect(force, &atoms, params);
    terms.upload(form, force.getNumParticlesPerBond(), atoms, params);
}
double HipCalcCustomCompoundBondForceKernel::execute(ContextImpl& context, bool includeForces, bool includeEnergy) {
    if (form.kind < 0) return reference->execute(context, includeForces, includeEnergy);
    terms.execute(includeEnergy);
    return 0.0;
}
void HipCalcCustomCompoundBondForceKernel::copyParametersToContext(ContextImpl& context, const CustomCompoundBondForce& force) {
    if (form.kind < 0) { reference->copyParametersToContext(context, force); return; }
    vector<double> params;
    collect(force, NULL, params);
    terms.uploadParams(params);
}

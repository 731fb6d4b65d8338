/* libommharness.so -- a small C API over the OpenMM C++ API (System / Force / Integrator / Context),
 * bound from Python with ctypes (openmm_amd/harness.py).  It exists because the reference's SWIG
 * Python module cannot be built in this environment (no swig/doxygen); tests and bench.py use it
 * to build Systems and to run the same Context on the "HIP", "CPU" and "Reference" platforms.
 * It mirrors the public API names it wraps (openmmapi/include/openmm/*.h); it contains no physics.
 */
#include "OpenMM.h"
#include "openmm/serialization/XmlSerializer.h"
#include "ReferenceConstraints.h"
#include "ReferenceSETTLEAlgorithm.h"
#include "ReferenceCCMAAlgorithm.h"
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <vector>

using namespace OpenMM;
using namespace std;

static string lastError;

#define GUARD(...) try { __VA_ARGS__; return 0; } catch (const std::exception& e) { lastError = e.what(); return 1; } catch (...) { lastError = "unknown error"; return 1; }

extern "C" {

const char* omm_last_error() { return lastError.c_str(); }

int omm_load_plugin(const char* path) { GUARD(Platform::loadPluginLibrary(path)) }
int omm_load_plugins_from_directory(const char* dir) { GUARD(Platform::loadPluginsFromDirectory(dir)) }
int omm_num_platforms() { return Platform::getNumPlatforms(); }
const char* omm_platform_name(int i) { return Platform::getPlatform(i).getName().c_str(); }
double omm_platform_speed(int i) { return Platform::getPlatform(i).getSpeed(); }
const char* omm_version() { static string v; v = Platform::getOpenMMVersion(); return v.c_str(); }

/* ---- System */
void* omm_system_create() { return new System(); }
void omm_system_destroy(void* s) { delete (System*) s; }
int omm_system_add_particles(void* s, int n, const double* masses) { GUARD(for (int i = 0; i < n; i++) ((System*) s)->addParticle(masses[i])) }
int omm_system_num_particles(void* s) { return ((System*) s)->getNumParticles(); }
int omm_system_set_box(void* s, const double* b) { GUARD(((System*) s)->setDefaultPeriodicBoxVectors(Vec3(b[0], b[1], b[2]), Vec3(b[3], b[4], b[5]), Vec3(b[6], b[7], b[8]))) }
int omm_system_add_constraints(void* s, int n, const int* pairs, const double* dist) { GUARD(for (int i = 0; i < n; i++) ((System*) s)->addConstraint(pairs[2 * i], pairs[2 * i + 1], dist[i])) }
int omm_system_num_constraints(void* s) { return ((System*) s)->getNumConstraints(); }

/* XmlSerializer::deserialize<System> / serialize (serialization/include/openmm/serialization/XmlSerializer.h:60-76) */
void* omm_system_from_xml(const char* text) {
    try {
        std::stringstream in(text);
        return XmlSerializer::deserialize<System>(in);
    } catch (const std::exception& e) { lastError = e.what(); return NULL; }
}
int omm_system_to_xml(void* s, char** text, long* size) {
    GUARD(
        std::stringstream out;
        XmlSerializer::serialize<System>((System*) s, "System", out);
        const std::string data = out.str();
        *size = (long) data.size();
        *text = (char*) malloc(data.size() + 1);
        memcpy(*text, data.c_str(), data.size() + 1);
    )
}
int omm_system_num_forces(void* s) { return ((System*) s)->getNumForces(); }

/* The SETTLE clusters the Reference platform finds in a System (ReferenceConstraints.cpp:44-148): the checker of the HIP platform's
 * own partition (tests only).  Returns the number of clusters; fills at most `capacity`: atoms[3i..], dist[2i..]. */
int omm_reference_settle_clusters(void* s, int* atoms, double* dist, int capacity) {
    try {
        ReferenceConstraints constraints(*(System*) s);
        ReferenceSETTLEAlgorithm* settle = dynamic_cast<ReferenceSETTLEAlgorithm*>(constraints.settle);
        if (settle == NULL) return 0;
        const int n = settle->getNumClusters();
        for (int i = 0; i < n && i < capacity; i++)
            settle->getClusterParameters(i, atoms[3 * i], atoms[3 * i + 1], atoms[3 * i + 2], dist[2 * i], dist[2 * i + 1]);
        return n;
    } catch (const std::exception& e) { lastError = e.what(); return -1; }
}

/* The thresholded inverse coupling matrix the Reference platform's CCMA builds for a System (ReferenceCCMAAlgorithm.cpp:42-196), as
 * (row, column, value) triplets over the CCMA constraints in the reference's order; constraint[2i..] = their atoms.  The checker of
 * oracle/constraints.py::ccma_matrix (tests only).  Returns the number of triplets (at most `capacity` are written), -1 on error. */
int omm_reference_ccma_matrix(void* s, int* rows, int* cols, double* values, int capacity, int* constraintAtoms, int constraintCapacity, int* numConstraints) {
    try {
        ReferenceConstraints constraints(*(System*) s);
        ReferenceCCMAAlgorithm* ccma = dynamic_cast<ReferenceCCMAAlgorithm*>(constraints.ccma);
        *numConstraints = 0;
        if (ccma == NULL) return 0;
        const System& system = *(System*) s;
        // the CCMA constraints are the System's constraints that touch no SETTLE atom, in System order (ReferenceConstraints.cpp:150-165)
        std::vector<char> isSettle(system.getNumParticles(), 0);
        ReferenceSETTLEAlgorithm* settle = dynamic_cast<ReferenceSETTLEAlgorithm*>(constraints.settle);
        if (settle != NULL)
            for (int i = 0; i < settle->getNumClusters(); i++) {
                int a, b, c2; double d1, d2;
                settle->getClusterParameters(i, a, b, c2, d1, d2);
                isSettle[a] = isSettle[b] = isSettle[c2] = 1;
            }
        int n = 0;
        for (int i = 0; i < system.getNumConstraints(); i++) {
            int a, b; double d;
            system.getConstraintParameters(i, a, b, d);
            if (isSettle[a]) continue;
            if (n < constraintCapacity) { constraintAtoms[2 * n] = a; constraintAtoms[2 * n + 1] = b; }
            n++;
        }
        *numConstraints = n;
        int count = 0;
        const auto& m = ccma->getMatrix();
        for (int i = 0; i < (int) m.size(); i++)
            for (const auto& e : m[i]) {
                if (count < capacity) { rows[count] = i; cols[count] = e.first; values[count] = e.second; }
                count++;
            }
        return count;
    } catch (const std::exception& e) { lastError = e.what(); return -1; }
}

/* ---- NonbondedForce (openmmapi/include/openmm/NonbondedForce.h) */
void* omm_nonbonded_create(void* s, int method, double cutoff, double ewaldTol, int useDispersion, int useSwitch, double switchDist) {
    try {
        NonbondedForce* f = new NonbondedForce();
        f->setNonbondedMethod((NonbondedForce::NonbondedMethod) method);
        f->setCutoffDistance(cutoff);
        f->setEwaldErrorTolerance(ewaldTol);
        f->setUseDispersionCorrection(useDispersion != 0);
        f->setUseSwitchingFunction(useSwitch != 0);
        f->setSwitchingDistance(switchDist);
        ((System*) s)->addForce(f);
        return f;
    } catch (const std::exception& e) { lastError = e.what(); return NULL; }
}
int omm_nonbonded_add_particles(void* f, int n, const double* q, const double* sig, const double* eps) { GUARD(for (int i = 0; i < n; i++) ((NonbondedForce*) f)->addParticle(q[i], sig[i], eps[i])) }
int omm_nonbonded_add_exceptions(void* f, int n, const int* pairs, const double* qq, const double* sig, const double* eps) {
    GUARD(for (int i = 0; i < n; i++) ((NonbondedForce*) f)->addException(pairs[2 * i], pairs[2 * i + 1], qq[i], sig[i], eps[i], true))
}
int omm_nonbonded_create_exceptions_from_bonds(void* f, int n, const int* pairs, double coulomb14, double lj14) {
    GUARD(vector<pair<int, int> > bonds; for (int i = 0; i < n; i++) bonds.push_back(make_pair(pairs[2 * i], pairs[2 * i + 1]));
          ((NonbondedForce*) f)->createExceptionsFromBonds(bonds, coulomb14, lj14))
}
int omm_nonbonded_num_exceptions(void* f) { return ((NonbondedForce*) f)->getNumExceptions(); }
int omm_nonbonded_set_pme_parameters(void* f, double alpha, int nx, int ny, int nz) { GUARD(((NonbondedForce*) f)->setPMEParameters(alpha, nx, ny, nz)) }
int omm_nonbonded_set_ljpme_parameters(void* f, double alpha, int nx, int ny, int nz) { GUARD(((NonbondedForce*) f)->setLJPMEParameters(alpha, nx, ny, nz)) }
int omm_nonbonded_get_ljpme_parameters_in_context(void* f, void* ctx, double* alpha, int* n) {
    GUARD(((NonbondedForce*) f)->getLJPMEParametersInContext(*(Context*) ctx, *alpha, n[0], n[1], n[2]))
}
int omm_nonbonded_set_reaction_field_dielectric(void* f, double d) { GUARD(((NonbondedForce*) f)->setReactionFieldDielectric(d)) }
int omm_nonbonded_set_reciprocal_force_group(void* f, int g) { GUARD(((NonbondedForce*) f)->setReciprocalSpaceForceGroup(g)) }
int omm_nonbonded_set_exceptions_use_periodic(void* f, int p) { GUARD(((NonbondedForce*) f)->setExceptionsUsePeriodicBoundaryConditions(p != 0)) }
int omm_nonbonded_get_pme_parameters_in_context(void* f, void* ctx, double* alpha, int* n) {
    GUARD(((NonbondedForce*) f)->getPMEParametersInContext(*(Context*) ctx, *alpha, n[0], n[1], n[2]))
}
int omm_force_set_group(void* f, int g) { GUARD(((Force*) f)->setForceGroup(g)) }

/* ---- bonded forces and CMMotionRemover */
void* omm_add_harmonic_bonds(void* s, int n, const int* atoms, const double* length, const double* k) {
    try { HarmonicBondForce* f = new HarmonicBondForce(); for (int i = 0; i < n; i++) f->addBond(atoms[2 * i], atoms[2 * i + 1], length[i], k[i]); ((System*) s)->addForce(f); return f; }
    catch (const std::exception& e) { lastError = e.what(); return NULL; }
}
void* omm_add_harmonic_angles(void* s, int n, const int* atoms, const double* angle, const double* k) {
    try { HarmonicAngleForce* f = new HarmonicAngleForce(); for (int i = 0; i < n; i++) f->addAngle(atoms[3 * i], atoms[3 * i + 1], atoms[3 * i + 2], angle[i], k[i]); ((System*) s)->addForce(f); return f; }
    catch (const std::exception& e) { lastError = e.what(); return NULL; }
}
void* omm_add_periodic_torsions(void* s, int n, const int* atoms, const int* periodicity, const double* phase, const double* k) {
    try { PeriodicTorsionForce* f = new PeriodicTorsionForce(); for (int i = 0; i < n; i++) f->addTorsion(atoms[4 * i], atoms[4 * i + 1], atoms[4 * i + 2], atoms[4 * i + 3], periodicity[i], phase[i], k[i]); ((System*) s)->addForce(f); return f; }
    catch (const std::exception& e) { lastError = e.what(); return NULL; }
}
/* ---- Custom forces defined by an energy expression.  paramNames: the per-bond / per-angle parameter names joined by ','; params [n x numParams].
 *      CustomBondForce / CustomAngleForce / CustomCompoundBondForce (openmmapi/include/openmm/Custom*Force.h) */
static vector<string> splitNames(const char* names) {
    vector<string> out;
    string cur;
    for (const char* c = names; c != NULL && *c; c++) {
        if (*c == ',') { if (!cur.empty()) out.push_back(cur); cur.clear(); }
        else cur += *c;
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}
void* omm_add_custom_bond_force(void* s, const char* energy, const char* paramNames, int n, const int* atoms, const double* params) {
    try {
        CustomBondForce* f = new CustomBondForce(energy);
        vector<string> names = splitNames(paramNames);
        for (size_t i = 0; i < names.size(); i++) f->addPerBondParameter(names[i]);
        const int np = (int) names.size();
        for (int i = 0; i < n; i++) f->addBond(atoms[2 * i], atoms[2 * i + 1], vector<double>(params + (size_t) np * i, params + (size_t) np * (i + 1)));
        ((System*) s)->addForce(f);
        return f;
    } catch (const std::exception& e) { lastError = e.what(); return NULL; }
}
void* omm_add_custom_angle_force(void* s, const char* energy, const char* paramNames, int n, const int* atoms, const double* params) {
    try {
        CustomAngleForce* f = new CustomAngleForce(energy);
        vector<string> names = splitNames(paramNames);
        for (size_t i = 0; i < names.size(); i++) f->addPerAngleParameter(names[i]);
        const int np = (int) names.size();
        for (int i = 0; i < n; i++) f->addAngle(atoms[3 * i], atoms[3 * i + 1], atoms[3 * i + 2], vector<double>(params + (size_t) np * i, params + (size_t) np * (i + 1)));
        ((System*) s)->addForce(f);
        return f;
    } catch (const std::exception& e) { lastError = e.what(); return NULL; }
}
void* omm_add_custom_compound_bond_force(void* s, int particlesPerBond, const char* energy, const char* paramNames, int n, const int* atoms, const double* params) {
    try {
        CustomCompoundBondForce* f = new CustomCompoundBondForce(particlesPerBond, energy);
        vector<string> names = splitNames(paramNames);
        for (size_t i = 0; i < names.size(); i++) f->addPerBondParameter(names[i]);
        const int np = (int) names.size();
        for (int i = 0; i < n; i++)
            f->addBond(vector<int>(atoms + (size_t) particlesPerBond * i, atoms + (size_t) particlesPerBond * (i + 1)), vector<double>(params + (size_t) np * i, params + (size_t) np * (i + 1)));
        ((System*) s)->addForce(f);
        return f;
    } catch (const std::exception& e) { lastError = e.what(); return NULL; }
}
int omm_force_set_name(void* f, const char* name) { GUARD(((Force*) f)->setName(name)) }

void* omm_add_gbsa_obc(void* s, int n, const double* charge, const double* radius, const double* scale, int method, double cutoff, double solventDielectric, double soluteDielectric) {
    try {
        GBSAOBCForce* f = new GBSAOBCForce();
        for (int i = 0; i < n; i++) f->addParticle(charge[i], radius[i], scale[i]);
        f->setNonbondedMethod((GBSAOBCForce::NonbondedMethod) method);
        f->setCutoffDistance(cutoff);
        f->setSolventDielectric(solventDielectric);
        f->setSoluteDielectric(soluteDielectric);
        ((System*) s)->addForce(f);
        return f;
    } catch (const std::exception& e) { lastError = e.what(); return NULL; }
}
void* omm_add_cmmotion_remover(void* s, int frequency) {
    try { CMMotionRemover* f = new CMMotionRemover(frequency); ((System*) s)->addForce(f); return f; }
    catch (const std::exception& e) { lastError = e.what(); return NULL; }
}

void* omm_add_monte_carlo_barostat(void* s, double pressure, double temperature, int frequency, int seed) {
    try { MonteCarloBarostat* f = new MonteCarloBarostat(pressure, temperature, frequency); f->setRandomNumberSeed(seed); ((System*) s)->addForce(f); return f; }
    catch (const std::exception& e) { lastError = e.what(); return NULL; }
}

/* ---- Integrators: kind 0 Verlet, 1 Langevin, 2 LangevinMiddle */
void* omm_integrator_create(int kind, double dt, double temperature, double friction, int seed, double constraintTol) {
    try {
        Integrator* integ;
        if (kind == 0) integ = new VerletIntegrator(dt);
        else if (kind == 1) { LangevinIntegrator* l = new LangevinIntegrator(temperature, friction, dt); l->setRandomNumberSeed(seed); integ = l; }
        else if (kind == 2) { LangevinMiddleIntegrator* l = new LangevinMiddleIntegrator(temperature, friction, dt); l->setRandomNumberSeed(seed); integ = l; }
        else throw OpenMMException("unknown integrator kind");
        integ->setConstraintTolerance(constraintTol);
        return integ;
    } catch (const std::exception& e) { lastError = e.what(); return NULL; }
}
/* ---- CustomIntegrator (openmmapi/include/openmm/CustomIntegrator.h): the computation steps are added one by one.
 *      kind: 0 global variable, 1 per-dof variable (declarations); 2 ComputeGlobal, 3 ComputePerDof, 4 ComputeSum (result, expression);
 *      5 ConstrainPositions, 6 ConstrainVelocities, 7 UpdateContextState */
void* omm_custom_integrator_create(double dt, int seed, double constraintTol) {
    try { CustomIntegrator* c = new CustomIntegrator(dt); c->setRandomNumberSeed(seed); c->setConstraintTolerance(constraintTol); return c; }
    catch (const std::exception& e) { lastError = e.what(); return NULL; }
}
int omm_custom_integrator_add(void* i, int kind, const char* name, const char* expression, double value) {
    GUARD(CustomIntegrator* c = dynamic_cast<CustomIntegrator*>((Integrator*) i);
          if (c == NULL) throw OpenMMException("not a CustomIntegrator");
          switch (kind) {
              case 0: c->addGlobalVariable(name, value); break;
              case 1: c->addPerDofVariable(name, value); break;
              case 2: c->addComputeGlobal(name, expression); break;
              case 3: c->addComputePerDof(name, expression); break;
              case 4: c->addComputeSum(name, expression); break;
              case 5: c->addConstrainPositions(); break;
              case 6: c->addConstrainVelocities(); break;
              case 7: c->addUpdateContextState(); break;
              default: throw OpenMMException("unknown CustomIntegrator step kind");
          })
}
int omm_custom_integrator_get_per_dof(void* i, int index, double* out) {
    GUARD(CustomIntegrator* c = dynamic_cast<CustomIntegrator*>((Integrator*) i);
          if (c == NULL) throw OpenMMException("not a CustomIntegrator");
          vector<Vec3> values; c->getPerDofVariable(index, values);
          for (size_t k = 0; k < values.size(); k++) { out[3 * k] = values[k][0]; out[3 * k + 1] = values[k][1]; out[3 * k + 2] = values[k][2]; })
}
int omm_custom_integrator_get_global(void* i, int index, double* out) {
    GUARD(CustomIntegrator* c = dynamic_cast<CustomIntegrator*>((Integrator*) i);
          if (c == NULL) throw OpenMMException("not a CustomIntegrator");
          *out = c->getGlobalVariable(index))
}
int omm_custom_integrator_set_global(void* i, int index, double value) {
    GUARD(CustomIntegrator* c = dynamic_cast<CustomIntegrator*>((Integrator*) i);
          if (c == NULL) throw OpenMMException("not a CustomIntegrator");
          c->setGlobalVariable(index, value))
}
void omm_integrator_destroy(void* i) { delete (Integrator*) i; }
int omm_integrator_step(void* i, int steps) { GUARD(((Integrator*) i)->step(steps)) }
int omm_integrator_set_step_size(void* i, double dt) { GUARD(((Integrator*) i)->setStepSize(dt)) }

/* ---- Context.  properties = "key=value;key=value" */
void* omm_context_create(void* s, void* integ, const char* platformName, const char* properties) {
    try {
        Platform& platform = Platform::getPlatformByName(platformName);
        map<string, string> props;
        string p = properties == NULL ? "" : properties;
        stringstream ss(p);
        string item;
        while (getline(ss, item, ';')) {
            size_t eq = item.find('=');
            if (eq != string::npos) props[item.substr(0, eq)] = item.substr(eq + 1);
        }
        return new Context(*(System*) s, *(Integrator*) integ, platform, props);
    } catch (const std::exception& e) { lastError = e.what(); return NULL; }
}
void omm_context_destroy(void* c) { delete (Context*) c; }
const char* omm_context_platform_name(void* c) { return ((Context*) c)->getPlatform().getName().c_str(); }
const char* omm_context_platform_property(void* c, const char* name) {
    static string v;
    try { v = ((Context*) c)->getPlatform().getPropertyValue(*(Context*) c, name); } catch (const std::exception& e) { lastError = e.what(); v = ""; }
    return v.c_str();
}
int omm_context_set_positions(void* c, int n, const double* xyz) {
    GUARD(vector<Vec3> p(n); for (int i = 0; i < n; i++) p[i] = Vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]); ((Context*) c)->setPositions(p))
}
int omm_context_set_velocities(void* c, int n, const double* xyz) {
    GUARD(vector<Vec3> p(n); for (int i = 0; i < n; i++) p[i] = Vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]); ((Context*) c)->setVelocities(p))
}
int omm_context_set_velocities_to_temperature(void* c, double temperature, int seed) { GUARD(((Context*) c)->setVelocitiesToTemperature(temperature, seed)) }
int omm_context_set_box(void* c, const double* b) { GUARD(((Context*) c)->setPeriodicBoxVectors(Vec3(b[0], b[1], b[2]), Vec3(b[3], b[4], b[5]), Vec3(b[6], b[7], b[8]))) }
int omm_context_apply_constraints(void* c, double tol) { GUARD(((Context*) c)->applyConstraints(tol)) }
int omm_context_apply_velocity_constraints(void* c, double tol) { GUARD(((Context*) c)->applyVelocityConstraints(tol)) }
/* Context::createCheckpoint / loadCheckpoint (openmmapi/include/openmm/Context.h): the blob is returned through a malloc'ed buffer the
 * caller hands back to omm_free. */
int omm_context_create_checkpoint(void* c, char** blob, long* size) {
    GUARD(
        std::stringstream stream(std::ios_base::out | std::ios_base::in | std::ios_base::binary);
        ((Context*) c)->createCheckpoint(stream);
        const std::string data = stream.str();
        *size = (long) data.size();
        *blob = (char*) malloc(data.size() > 0 ? data.size() : 1);
        memcpy(*blob, data.data(), data.size());
    )
}
int omm_context_load_checkpoint(void* c, const char* blob, long size) {
    GUARD(
        std::stringstream stream(std::string(blob, (size_t) size), std::ios_base::out | std::ios_base::in | std::ios_base::binary);
        ((Context*) c)->loadCheckpoint(stream);
    )
}
void omm_free(void* p) { free(p); }
int omm_context_set_parameter(void* c, const char* name, double v) { GUARD(((Context*) c)->setParameter(name, v)) }
int omm_context_minimize(void* c, double tolerance, int maxIterations) { GUARD(LocalEnergyMinimizer::minimize(*(Context*) c, tolerance, maxIterations)) }
int omm_context_reinitialize(void* c, int preserveState) { GUARD(((Context*) c)->reinitialize(preserveState != 0)) }
/* flags: 1 positions, 2 velocities, 4 forces, 8 energy.  energies[0] = potential, [1] = kinetic, [2] = time. groups = -1 for all */
int omm_context_get_box(void* c, double* b) {
    GUARD(
        Vec3 v[3];
        ((Context*) c)->getState(0).getPeriodicBoxVectors(v[0], v[1], v[2]);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) b[3 * i + j] = v[i][j];
    )
}
int omm_context_get_state(void* c, int flags, int groups, double* pos, double* vel, double* forces, double* energies) {
    GUARD(
        int types = 0;
        if (flags & 1) types |= State::Positions;
        if (flags & 2) types |= State::Velocities;
        if (flags & 4) types |= State::Forces;
        if (flags & 8) types |= State::Energy;
        State st = ((Context*) c)->getState(types, false, groups);
        int n = ((Context*) c)->getSystem().getNumParticles();
        if (flags & 1) { const vector<Vec3>& v = st.getPositions(); for (int i = 0; i < n; i++) { pos[3 * i] = v[i][0]; pos[3 * i + 1] = v[i][1]; pos[3 * i + 2] = v[i][2]; } }
        if (flags & 2) { const vector<Vec3>& v = st.getVelocities(); for (int i = 0; i < n; i++) { vel[3 * i] = v[i][0]; vel[3 * i + 1] = v[i][1]; vel[3 * i + 2] = v[i][2]; } }
        if (flags & 4) { const vector<Vec3>& v = st.getForces(); for (int i = 0; i < n; i++) { forces[3 * i] = v[i][0]; forces[3 * i + 1] = v[i][1]; forces[3 * i + 2] = v[i][2]; } }
        if (flags & 8) { energies[0] = st.getPotentialEnergy(); energies[1] = st.getKineticEnergy(); }
        if (energies != NULL) energies[2] = st.getTime();
    )
}

}

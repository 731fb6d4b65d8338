/* libommharness_amoeba.so -- the C API of libommharness.so extended to the two AMOEBA forces the HIP platform computes natively
 * (AmoebaMultipoleForce, AmoebaVdwForce; plugins/amoeba/openmmapi/include/openmm/AmoebaMultipoleForce.h, AmoebaVdwForce.h), bound
 * with ctypes by openmm_amd/harness.py.  A library of its own so that libommharness.so does not depend on the AMOEBA plugin.
 * It mirrors the API names it wraps and contains no physics.
 */
#include "OpenMM.h"
#include "openmm/AmoebaGeneralizedKirkwoodForce.h"
#include "openmm/AmoebaMultipoleForce.h"
#include "openmm/AmoebaTorsionTorsionForce.h"
#include "openmm/AmoebaVdwForce.h"
#include "openmm/AmoebaWcaDispersionForce.h"
#include <string>
#include <vector>

using namespace OpenMM;
using namespace std;

static string lastError;
#define GUARD(...) try { __VA_ARGS__; return 0; } catch (const std::exception& e) { lastError = e.what(); return 1; } catch (...) { lastError = "unknown error"; return 1; }

extern "C" {

const char* omm_amoeba_last_error() { return lastError.c_str(); }

/* method: 0 NoCutoff, 1 PME; polarization: 0 Mutual, 1 Direct, 2 Extrapolated (the enums of AmoebaMultipoleForce.h) */
void* omm_amoeba_multipole_create(void* system, int method, int polarization, double cutoff, double aEwald, const int* grid, double ewaldTol,
                                  double mutualEpsilon, int mutualMaxIterations) {
    try {
        AmoebaMultipoleForce* f = new AmoebaMultipoleForce();
        f->setNonbondedMethod((AmoebaMultipoleForce::NonbondedMethod) method);
        f->setPolarizationType((AmoebaMultipoleForce::PolarizationType) polarization);
        f->setCutoffDistance(cutoff);
        f->setAEwald(aEwald);
        if (grid != NULL && grid[0] > 0) f->setPmeGridDimensions(vector<int>(grid, grid + 3));
        f->setEwaldErrorTolerance(ewaldTol);
        f->setMutualInducedTargetEpsilon(mutualEpsilon);
        f->setMutualInducedMaxIterations(mutualMaxIterations);
        ((System*) system)->addForce(f);
        return f;
    } catch (const std::exception& e) { lastError = e.what(); return NULL; }
}

/* dipoles [3n], quadrupoles [9n] (molecular frame), axes [4n] = (axis type, z, x, y atoms) */
int omm_amoeba_multipole_add(void* f, int n, const double* charge, const double* dipole, const double* quadrupole, const int* axes,
                             const double* thole, const double* damping, const double* polarity) {
    GUARD(for (int i = 0; i < n; i++)
              ((AmoebaMultipoleForce*) f)->addMultipole(charge[i], vector<double>(dipole + 3 * i, dipole + 3 * i + 3), vector<double>(quadrupole + 9 * i, quadrupole + 9 * i + 9),
                                                        axes[4 * i], axes[4 * i + 1], axes[4 * i + 2], axes[4 * i + 3], thole[i], damping[i], polarity[i]))
}

/* CSR over (atom, covalent type): entry e sets the map `type[e]` of atom `atom[e]` to list[start[e] .. start[e + 1]) */
int omm_amoeba_multipole_set_covalent_maps(void* f, int entries, const int* atom, const int* type, const int* start, const int* list) {
    GUARD(for (int e = 0; e < entries; e++)
              ((AmoebaMultipoleForce*) f)->setCovalentMap(atom[e], (AmoebaMultipoleForce::CovalentType) type[e], vector<int>(list + start[e], list + start[e + 1])))
}

int omm_amoeba_multipole_get_induced_dipoles(void* f, void* context, double* out) {
    GUARD(vector<Vec3> d; ((AmoebaMultipoleForce*) f)->getInducedDipoles(*(Context*) context, d);
          for (size_t i = 0; i < d.size(); i++) { out[3 * i] = d[i][0]; out[3 * i + 1] = d[i][1]; out[3 * i + 2] = d[i][2]; })
}

/* AmoebaTorsionTorsionForce (plugins/amoeba/openmmapi/include/openmm/AmoebaTorsionTorsionForce.h): atoms [6n] = five chain atoms + the
 * chirality marker atom (or -1); grids [numGrids][nx][ny][columns] with columns = 3 (angle1, angle2, f: the force derives the spline
 * derivatives itself) or 6 (+ fx, fy, fxy) */
void* omm_amoeba_torsion_torsion_create(void* system, int n, const int* atoms, const int* gridIndex) {
    try {
        AmoebaTorsionTorsionForce* f = new AmoebaTorsionTorsionForce();
        for (int i = 0; i < n; i++)
            f->addTorsionTorsion(atoms[6 * i], atoms[6 * i + 1], atoms[6 * i + 2], atoms[6 * i + 3], atoms[6 * i + 4], atoms[6 * i + 5], gridIndex[i]);
        ((System*) system)->addForce(f);
        return f;
    } catch (const std::exception& e) { lastError = e.what(); return NULL; }
}
int omm_amoeba_torsion_torsion_set_grid(void* f, int index, int nx, int ny, int columns, const double* values) {
    GUARD(vector<vector<vector<double> > > grid(nx, vector<vector<double> >(ny, vector<double>(columns)));
          for (int x = 0; x < nx; x++) for (int y = 0; y < ny; y++) for (int c = 0; c < columns; c++) grid[x][y][c] = values[((size_t) x * ny + y) * columns + c];
          ((AmoebaTorsionTorsionForce*) f)->setTorsionTorsionGrid(index, grid))
}

/* sigmaRule / epsilonRule: the strings of AmoebaVdwForce ("CUBIC-MEAN", "HHG", ...); method: 0 NoCutoff, 1 CutoffPeriodic */
void* omm_amoeba_vdw_create(void* system, const char* sigmaRule, const char* epsilonRule, int method, double cutoff, int dispersionCorrection) {
    try {
        AmoebaVdwForce* f = new AmoebaVdwForce();
        f->setSigmaCombiningRule(sigmaRule);
        f->setEpsilonCombiningRule(epsilonRule);
        f->setNonbondedMethod((AmoebaVdwForce::NonbondedMethod) method);
        f->setCutoff(cutoff);
        f->setUseDispersionCorrection(dispersionCorrection != 0);
        ((System*) system)->addForce(f);
        return f;
    } catch (const std::exception& e) { lastError = e.what(); return NULL; }
}

int omm_amoeba_vdw_add(void* f, int n, const int* parent, const double* sigma, const double* epsilon, const double* reduction) {
    GUARD(for (int i = 0; i < n; i++) ((AmoebaVdwForce*) f)->addParticle(parent[i], sigma[i], epsilon[i], reduction[i]))
}

/* AmoebaGeneralizedKirkwoodForce (plugins/amoeba/openmmapi/include/openmm/AmoebaGeneralizedKirkwoodForce.h) */
void* omm_amoeba_gk_create(void* system, int n, const double* charge, const double* radius, const double* scale, double solventDielectric, double soluteDielectric,
                           int includeCavityTerm, double probeRadius, double surfaceAreaFactor) {
    try {
        AmoebaGeneralizedKirkwoodForce* f = new AmoebaGeneralizedKirkwoodForce();
        f->setSolventDielectric(solventDielectric);
        f->setSoluteDielectric(soluteDielectric);
        f->setIncludeCavityTerm(includeCavityTerm);
        f->setProbeRadius(probeRadius);
        f->setSurfaceAreaFactor(surfaceAreaFactor);
        for (int i = 0; i < n; i++) f->addParticle(charge[i], radius[i], scale[i]);
        ((System*) system)->addForce(f);
        return f;
    } catch (const std::exception& e) { lastError = e.what(); return NULL; }
}

/* AmoebaWcaDispersionForce (plugins/amoeba/openmmapi/include/openmm/AmoebaWcaDispersionForce.h); globals = (epso, epsh, rmino, rminh, awater, slevy, dispoff, shctd) */
void* omm_amoeba_wca_create(void* system, int n, const double* radius, const double* epsilon, const double* globals) {
    try {
        AmoebaWcaDispersionForce* f = new AmoebaWcaDispersionForce();
        f->setEpso(globals[0]); f->setEpsh(globals[1]); f->setRmino(globals[2]); f->setRminh(globals[3]);
        f->setAwater(globals[4]); f->setSlevy(globals[5]); f->setDispoff(globals[6]); f->setShctd(globals[7]);
        for (int i = 0; i < n; i++) f->addParticle(radius[i], epsilon[i]);
        ((System*) system)->addForce(f);
        return f;
    } catch (const std::exception& e) { lastError = e.what(); return NULL; }
}

/* CSR: the exclusions of atom i are list[start[i] .. start[i + 1]) */
int omm_amoeba_vdw_set_exclusions(void* f, int n, const int* start, const int* list) {
    GUARD(for (int i = 0; i < n; i++) ((AmoebaVdwForce*) f)->setParticleExclusions(i, vector<int>(list + start[i], list + start[i + 1])))
}

}

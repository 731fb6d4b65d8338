// HipCalcPmeReciprocalForceKernel (kernels.h:1493-1560 CalcPmeReciprocalForceKernel + ::IO) checked the way the reference checks its CPU
// implementation of that kernel (plugins/cpupme/tests/TestCpuPme.cpp:568-638 testPME): a cloud of random point charges, rectangular and
// triclinic box, reciprocal-space forces and energy of the Reference platform (force group of its own) against the kernel driven directly
// through the IO object -- same tolerances (1e-3).  Grid sizes are rounded up to lengths this platform's transform takes.
#include "HipTests.h"
#include "HipKernels.h"
#include "openmm/internal/AssertionUtilities.h"
#include "openmm/Context.h"
#include "openmm/NonbondedForce.h"
#include "openmm/internal/NonbondedForceImpl.h"
#include "openmm/System.h"
#include "openmm/VerletIntegrator.h"
#include "sfmt/SFMT.h"
#include <cmath>
#include <iostream>
#include <vector>

using namespace OpenMM;
using namespace std;

static const double ONE_4PI_EPS0_ = 138.935456;

class IO : public CalcPmeReciprocalForceKernel::IO {
public:
    vector<float> posq;
    float* force;
    float* getPosq() { return &posq[0]; }
    void setForce(float* f) { force = f; }
};

static void testPME(bool triclinic, int numParticles) {
    const double boxWidth = 5.0, cutoff = 1.0;
    Vec3 boxVectors[3];
    boxVectors[0] = Vec3(boxWidth, 0, 0);
    boxVectors[1] = triclinic ? Vec3(0.2 * boxWidth, boxWidth, 0) : Vec3(0, boxWidth, 0);
    boxVectors[2] = triclinic ? Vec3(-0.3 * boxWidth, -0.1 * boxWidth, boxWidth) : Vec3(0, 0, boxWidth);
    System system;
    system.setDefaultPeriodicBoxVectors(boxVectors[0], boxVectors[1], boxVectors[2]);
    NonbondedForce* force = new NonbondedForce();
    system.addForce(force);
    vector<Vec3> positions(numParticles);
    OpenMM_SFMT::SFMT sfmt;
    init_gen_rand(0, sfmt);
    for (int i = 0; i < numParticles; i++) {
        system.addParticle(1.0);
        force->addParticle(-1.0 + i * 2.0 / (numParticles - 1), 1.0, 0.0);
        positions[i] = Vec3(boxWidth * genrand_real2(sfmt), boxWidth * genrand_real2(sfmt), boxWidth * genrand_real2(sfmt));
    }
    force->setNonbondedMethod(NonbondedForce::PME);
    force->setCutoffDistance(cutoff);
    force->setReciprocalSpaceForceGroup(1);
    force->setEwaldErrorTolerance(1e-4);
    double alpha;
    int gx, gy, gz;
    NonbondedForceImpl::calcPMEParameters(system, *force, alpha, gx, gy, gz, false);
    while (!ommhip_fft_supported_size(gx)) gx++;
    while (!ommhip_fft_supported_size(gy)) gy++;
    while (!ommhip_fft_supported_size(gz)) gz++;
    force->setPMEParameters(alpha, gx, gy, gz);          // both sides on the same grid

    Platform& reference = Platform::getPlatformByName("Reference");
    VerletIntegrator integrator(0.01);
    Context context(system, integrator, reference);
    context.setPositions(positions);
    State refState = context.getState(State::Forces | State::Energy, false, 1 << 1);

    HipCalcPmeReciprocalForceKernel pme(CalcPmeReciprocalForceKernel::Name(), platform);
    IO io;
    double sumSquaredCharges = 0;
    for (int i = 0; i < numParticles; i++) {
        double charge, sigma, epsilon;
        force->getParticleParameters(i, charge, sigma, epsilon);
        io.posq.push_back((float) positions[i][0]); io.posq.push_back((float) positions[i][1]); io.posq.push_back((float) positions[i][2]); io.posq.push_back((float) charge);
        sumSquaredCharges += charge * charge;
    }
    const double ewaldSelfEnergy = -ONE_4PI_EPS0_ * alpha * sumSquaredCharges / sqrt(M_PI);
    for (int pass = 0; pass < 2; pass++) {          // twice: the kernel object is reused from evaluation to evaluation
        pme.initialize(gx, gy, gz, numParticles, alpha, pass == 1);
        pme.beginComputation(io, boxVectors, true);
        const double energy = pme.finishComputation(io);
        ASSERT_EQUAL_TOL(refState.getPotentialEnergy(), energy + ewaldSelfEnergy, 1e-3);
        for (int i = 0; i < numParticles; i++)
            ASSERT_EQUAL_VEC(refState.getForces()[i], Vec3(io.force[4 * i], io.force[4 * i + 1], io.force[4 * i + 2]), 1e-3);
    }
    double a; int nx, ny, nz;
    pme.getPMEParameters(a, nx, ny, nz);
    ASSERT(a == alpha && nx == gx && ny == gy && nz == gz);
    // forces only: the energy comes back as zero and the forces are the same
    vector<float> first(io.force, io.force + 4 * numParticles);
    pme.beginComputation(io, boxVectors, false);
    ASSERT(pme.finishComputation(io) == 0.0);
    for (int i = 0; i < 4 * numParticles; i++) ASSERT_EQUAL_TOL(first[i], io.force[i], 1e-4);
}

static void testThroughTheFactory() {
    // the platform hands the kernel out under the reference's name (what a platform that outsources reciprocal space asks for: CudaKernels.cpp:1679-1700)
    System system;
    system.addParticle(1.0);
    VerletIntegrator integrator(0.001);
    Context context(system, integrator, platform);
    // (Platform::createKernel needs the ContextImpl; reaching it through a Kernel object is what ContextImpl's own clients do)
    ASSERT(platform.supportsKernels(vector<string>(1, CalcPmeReciprocalForceKernel::Name())));
}

int main(int argc, char* argv[]) {
    try {
        initializeTests(argc, argv);
        testPME(false, 51);
        testPME(true, 51);
        testPME(false, 1500);
        testThroughTheFactory();
    }
    catch (const exception& e) {
        cout << "exception: " << e.what() << endl;
        return 1;
    }
    cout << "Done" << endl;
    return 0;
}

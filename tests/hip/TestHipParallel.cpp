// One Context over a LIST of devices (`DeviceIndex = "d,d"`): the HIP twin of the reference's testParallelComputation
// (platforms/cuda/tests/TestCudaNonbondedForce.cpp:37-96 -- one Context on a single device, one on the same device named twice, forces and
// energy must agree, also after updateParametersInContext), for the method this platform decomposes (PME), plus what the reference's test
// does not look at: dynamics -- a box of rigid water under LangevinMiddle with a CMMotionRemover, the trajectory of the device-list Context
// against the single-device one, a checkpoint taken and put back, the state queries (positions, velocities, forces, energies).
#include "HipTests.h"
#include "openmm/internal/AssertionUtilities.h"
#include "openmm/CMMotionRemover.h"
#include "openmm/Context.h"
#include "openmm/HarmonicBondForce.h"
#include "openmm/LangevinMiddleIntegrator.h"
#include "openmm/NonbondedForce.h"
#include "openmm/System.h"
#include "openmm/VerletIntegrator.h"
#include "sfmt/SFMT.h"
#include <cmath>
#include <iostream>
#include <map>
#include <sstream>
#include <vector>

using namespace OpenMM;
using namespace std;

static map<string, string> listOf(Context& single, int copies) {
    const string d = platform.getPropertyValue(single, HipPlatform::HipDeviceIndex());
    string list = d;
    for (int i = 1; i < copies; i++) list += "," + d;
    map<string, string> props;
    props[HipPlatform::HipDeviceIndex()] = list;
    return props;
}

void testParallelComputation() {
    System system;
    const int numParticles = 200;
    for (int i = 0; i < numParticles; i++)
        system.addParticle(1.0);
    NonbondedForce* force = new NonbondedForce();
    for (int i = 0; i < numParticles; i++)
        force->addParticle(i % 2 - 0.5, 0.5, 1.0);
    force->setNonbondedMethod(NonbondedForce::PME);
    system.addForce(force);
    system.setDefaultPeriodicBoxVectors(Vec3(5, 0, 0), Vec3(0, 5, 0), Vec3(0, 0, 5));
    OpenMM_SFMT::SFMT sfmt;
    init_gen_rand(0, sfmt);
    vector<Vec3> positions(numParticles);
    for (int i = 0; i < numParticles; i++)
        positions[i] = Vec3(5 * genrand_real2(sfmt), 5 * genrand_real2(sfmt), 5 * genrand_real2(sfmt));
    for (int i = 0; i < numParticles; ++i)
        for (int j = 0; j < i; ++j) {
            Vec3 delta = positions[i] - positions[j];
            if (delta.dot(delta) < 0.1)
                force->addException(i, j, 0, 1, 0);
        }

    // Create two contexts, one with a single device and one with two devices.

    VerletIntegrator integrator1(0.01);
    Context context1(system, integrator1, platform);
    context1.setPositions(positions);
    State state1 = context1.getState(State::Forces | State::Energy);
    VerletIntegrator integrator2(0.01);
    Context context2(system, integrator2, platform, listOf(context1, 2));
    ASSERT(platform.getPropertyValue(context2, HipPlatform::HipRanks()) == "2");
    context2.setPositions(positions);
    State state2 = context2.getState(State::Forces | State::Energy);

    // See if they agree.  (The reference asks for 1e-5 of its double-precision-accumulated single-precision forces; here the two Contexts
    // sort the atoms into different blocks, whose centres the pair arithmetic is relative to: separations differ by ~1e-7 nm, and the
    // random cloud holds pairs at 0.1 nm whose Lennard-Jones force -- thousands of kJ/mol/nm -- turns that into 2e-5.  The platform's
    // stated tolerance is 1e-4.)

    const double tol = 5e-5;
    ASSERT_EQUAL_TOL(state1.getPotentialEnergy(), state2.getPotentialEnergy(), 1e-5);
    for (int i = 0; i < numParticles; i++)
        ASSERT_EQUAL_VEC(state1.getForces()[i], state2.getForces()[i], tol);

    // Modify some particle parameters and see if they still agree.

    for (int i = 0; i < numParticles; i += 5) {
        double charge, sigma, epsilon;
        force->getParticleParameters(i, charge, sigma, epsilon);
        force->setParticleParameters(i, 0.9 * charge, sigma, epsilon);
    }
    force->updateParametersInContext(context1);
    force->updateParametersInContext(context2);
    state1 = context1.getState(State::Forces | State::Energy);
    state2 = context2.getState(State::Forces | State::Energy);
    ASSERT_EQUAL_TOL(state1.getPotentialEnergy(), state2.getPotentialEnergy(), 1e-5);
    for (int i = 0; i < numParticles; i++)
        ASSERT_EQUAL_VEC(state1.getForces()[i], state2.getForces()[i], tol);
}

static void buildWater(System& system, vector<Vec3>& positions, int side) {
    // side^3 TIP3P molecules on a lattice (0.31 nm), every molecule turned a little differently; rigid (three constraints per molecule)
    const double spacing = 0.31, dOH = 0.09572, angle = 104.52 * M_PI / 180.0, dHH = 2 * dOH * sin(0.5 * angle);
    const double L = side * spacing;
    system.setDefaultPeriodicBoxVectors(Vec3(L, 0, 0), Vec3(0, L, 0), Vec3(0, 0, L));
    NonbondedForce* nb = new NonbondedForce();
    nb->setNonbondedMethod(NonbondedForce::PME);
    nb->setCutoffDistance(0.9);
    OpenMM_SFMT::SFMT sfmt;
    init_gen_rand(7, sfmt);
    for (int x = 0; x < side; x++)
        for (int y = 0; y < side; y++)
            for (int z = 0; z < side; z++) {
                const int o = system.addParticle(15.9994);
                system.addParticle(1.008); system.addParticle(1.008);
                nb->addParticle(-0.834, 0.315075, 0.635968); nb->addParticle(0.417, 1.0, 0.0); nb->addParticle(0.417, 1.0, 0.0);
                nb->addException(o, o + 1, 0, 1, 0); nb->addException(o, o + 2, 0, 1, 0); nb->addException(o + 1, o + 2, 0, 1, 0);
                system.addConstraint(o, o + 1, dOH); system.addConstraint(o, o + 2, dOH); system.addConstraint(o + 1, o + 2, dHH);
                const double phi = 2 * M_PI * genrand_real2(sfmt), c = cos(phi), s = sin(phi);
                const Vec3 centre((x + 0.5) * spacing, (y + 0.5) * spacing, (z + 0.5) * spacing);
                const double hx = dOH * sin(0.5 * angle), hz = dOH * cos(0.5 * angle);
                positions.push_back(centre);
                positions.push_back(centre + Vec3(c * hx, s * hx, hz));
                positions.push_back(centre + Vec3(-c * hx, -s * hx, hz));
            }
    system.addForce(nb);
    system.addForce(new CMMotionRemover(1));
}

void testDynamics(int ranks, int side = 14, int steps = 30) {
    System system;
    vector<Vec3> positions;
    buildWater(system, positions, side);          // side 14: 8 232 atoms, L = 4.34 nm
    const int numParticles = system.getNumParticles();
    LangevinMiddleIntegrator integrator1(300.0, 1.0, 0.002), integrator2(300.0, 1.0, 0.002);
    integrator1.setRandomNumberSeed(11); integrator2.setRandomNumberSeed(11);
    Context context1(system, integrator1, platform);
    Context context2(system, integrator2, platform, listOf(context1, ranks));
    ASSERT(platform.getPropertyValue(context2, HipPlatform::HipIntegrationMode()) == "device");
    for (Context* c : {&context1, &context2}) {
        c->setPositions(positions);
        c->setVelocitiesToTemperature(300.0, 5);
    }
    integrator1.step(steps);
    integrator2.step(steps);
    State s1 = context1.getState(State::Positions | State::Velocities | State::Forces | State::Energy);
    State s2 = context2.getState(State::Positions | State::Velocities | State::Forces | State::Energy);
    double worst = 0;
    for (int i = 0; i < numParticles; i++) {
        const Vec3 d = s1.getPositions()[i] - s2.getPositions()[i];
        worst = max(worst, sqrt(d.dot(d)));
    }
    ASSERT(worst < 1e-4);          // 30 steps of a chaotic system in float arithmetic: the two runs sum their forces in different orders
    ASSERT_EQUAL_TOL(s1.getPotentialEnergy(), s2.getPotentialEnergy(), 1e-4);
    ASSERT_EQUAL_TOL(s1.getKineticEnergy(), s2.getKineticEnergy(), 1e-3);
    ASSERT_EQUAL_TOL(s1.getTime(), s2.getTime(), 1e-12);
    // a checkpoint of the device-list Context, more steps, the checkpoint put back, the same steps again: the same state
    stringstream checkpoint;
    context2.createCheckpoint(checkpoint);
    integrator2.step(steps / 3);
    State after = context2.getState(State::Positions);
    context2.loadCheckpoint(checkpoint);
    integrator2.step(steps / 3);
    State again = context2.getState(State::Positions);
    for (int i = 0; i < numParticles; i++)
        ASSERT_EQUAL_VEC(after.getPositions()[i], again.getPositions()[i], 2e-5);          // (float atomics on the charge grid: two runs of ten steps agree to ~1e-6, not bit for bit)
    // the constraints hold on every rank's molecules
    for (int i = 0; i < numParticles; i += 3) {
        const Vec3 d = again.getPositions()[i] - again.getPositions()[i + 1];
        ASSERT_EQUAL_TOL(0.09572, sqrt(d.dot(d)), 1e-4);
    }
}

/* A rank that fails in the middle of a run (OPENMM_HIP_DEBUG_FAIL_RANK = "rank:evaluation": its nonbonded kernel throws there) never reaches
 * the collectives the others wait in.  With the host-staged transport between the threads the waits end with an error, the user's call throws
 * an OpenMMException that names the rank which failed first, and the Context can be destroyed -- no thread is left waiting. */
void testFailingRank(int failing, int side) {
    System system;
    vector<Vec3> positions;
    buildWater(system, positions, side);
    LangevinMiddleIntegrator integrator0(300.0, 1.0, 0.002), integrator(300.0, 1.0, 0.002);
    Context single(system, integrator0, platform);          // (only to learn the device the tests run on)
    const string failAt = to_string(failing) + ":5";
    setenv("OPENMM_HIP_DEBUG_FAIL_RANK", failAt.c_str(), 1);
    string message;
    {
        Context context(system, integrator, platform, listOf(single, 2));
        context.setPositions(positions);
        context.setVelocitiesToTemperature(300.0, 5);
        try {
            integrator.step(40);                                           // (the inner ranks' errors surface at the next join: every 32 steps, or with an energy)
            context.getState(State::Energy);
        }
        catch (const OpenMMException& e) { message = e.what(); }
    }                                                                      // ... and the destructor returns
    unsetenv("OPENMM_HIP_DEBUG_FAIL_RANK");
    ASSERT(message.find("rank " + to_string(failing) + " fails here") != string::npos);
}

void testRefusals() {
    System system;
    system.addParticle(1.0);
    VerletIntegrator integrator(0.001);
    map<string, string> props;
    props[HipPlatform::HipDeviceIndex()] = "0,0";
    props[HipPlatform::HipRanks()] = "2";
    bool threw = false;
    try { Context context(system, integrator, platform, props); } catch (const OpenMMException&) { threw = true; }
    ASSERT(threw);
    // a device that does not exist is refused before any rank waits for it
    map<string, string> far;
    far[HipPlatform::HipDeviceIndex()] = "0,4096";
    threw = false;
    try { Context context(system, integrator, platform, far); } catch (const OpenMMException&) { threw = true; }
    ASSERT(threw);
}

int main(int argc, char* argv[]) {
    try {
        initializeTests(1, argv);
        const bool quick = argc > 1 && string(argv[1]) == "quick";          // the CPU emulator: the force comparison only
        testParallelComputation();
        testRefusals();
        testFailingRank(1, 8);
        testFailingRank(0, 8);
        if (!quick) {
            testDynamics(2);
            testDynamics(3);
        }
        else testDynamics(2, 8, 6);          // 1 536 atoms, six steps (slabs too thin for a halo: positions replicated)
    }
    catch (const exception& e) {
        cout << "exception: " << e.what() << endl;
        return 1;
    }
    cout << "Done" << endl;
    return 0;
}

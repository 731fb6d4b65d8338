// Native AMOEBA multipole kernel (libOpenMMAmoebaHIP.so) against the AMOEBA plugin's own Reference kernel on the systems of the
// reference's test body (plugins/amoeba/tests/TestAmoebaMultipoleForce.h, included at build time, never copied): the same Context
// is created twice, once with OPENMM_HIP_REFERENCE_AMOEBA_MULTIPOLE=1 (the factory then hands the force to the Reference kernel,
// which runs as a fallback force on the HIP platform).  Prints energies and the largest force difference relative to the RMS force;
// exit code 1 when a case is off by more than 1e-4.
#include "HipAmoebaTests.h"
#define main reference_test_body_main
#include "TestAmoebaMultipoleForce.h"
#undef main
#include <cmath>
#include <cstdio>

void runPlatformTests() {}

typedef void (*Case)(std::vector<Vec3>& forces, double& energy);
static AmoebaMultipoleForce::PolarizationType gPolarization = AmoebaMultipoleForce::Direct;
static void water4(std::vector<Vec3>& f, double& e) { setupAndGetForcesEnergyMultipoleWater(AmoebaMultipoleForce::PME, gPolarization, 0.70, 20, f, e); }
static void ionsAndWater(std::vector<Vec3>& f, double& e) { setupAndGetForcesEnergyMultipoleIonsAndWater(AmoebaMultipoleForce::PME, gPolarization, 0.70, 20, "parity", f, e); }
static void water648(std::vector<Vec3>& f, double& e) {
    std::string name = "parity";
    std::vector<double> moments, potential;
    std::vector<Vec3> grid;
    setupAndGetForcesEnergyMultipoleLargeWater(AmoebaMultipoleForce::PME, gPolarization, 0.70, 24, name, f, e, moments, grid, potential);
}

// point multipoles given in the lab frame (NoAxisType), no covalent maps: which rank of the expansion / which part of the polarization is off?
static int gLevel = 0;      // bit 0 charges, 1 dipoles, 2 quadrupoles, 3 polarizable
static void randomSites(std::vector<Vec3>& forces, double& energy) {
    const int n = 24;
    const double L = 2.0;
    System system;
    AmoebaMultipoleForce* mf = new AmoebaMultipoleForce();
    mf->setNonbondedMethod(AmoebaMultipoleForce::PME);
    mf->setPolarizationType(AmoebaMultipoleForce::Direct);
    mf->setCutoffDistance(0.8);
    mf->setAEwald(5.4459052);
    std::vector<int> grid(3, 24);
    mf->setPmeGridDimensions(grid);
    system.setDefaultPeriodicBoxVectors(Vec3(L, 0, 0), Vec3(0, L, 0), Vec3(0, 0, L));
    std::vector<Vec3> pos(n);
    unsigned long long state = 12345;
    auto rnd = [&]() { state = state * 6364136223846793005ULL + 1442695040888963407ULL; return (double) ((state >> 33) & 0xffffff) / 16777216.0; };
    double qsum = 0;
    for (int i = 0; i < n; i++) {
        system.addParticle(1.0);
        pos[i] = Vec3(L * rnd(), L * rnd(), L * rnd());
        double q = (gLevel & 1) ? (i % 2 ? 0.5 : -0.5) : 0.0;
        std::vector<double> d(3, 0.0), quad(9, 0.0);
        if (gLevel & 2) for (int k = 0; k < 3; k++) d[k] = 0.02 * (rnd() - 0.5);
        if (gLevel & 4) {
            double a = 0.002 * (rnd() - 0.5), b = 0.002 * (rnd() - 0.5), xy = 0.002 * (rnd() - 0.5), xz = 0.002 * (rnd() - 0.5), yz = 0.002 * (rnd() - 0.5);
            quad[0] = a; quad[4] = b; quad[8] = -a - b; quad[1] = quad[3] = xy; quad[2] = quad[6] = xz; quad[5] = quad[7] = yz;
        }
        mf->addMultipole(q, d, quad, AmoebaMultipoleForce::NoAxisType, -1, -1, -1, 0.39, (gLevel & 8) ? 0.3 : 0.0, (gLevel & 8) ? 0.001 : 0.0);
        qsum += q;
    }
    // keep atoms apart
    for (int it = 0; it < 200; it++)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < i; j++) {
                Vec3 d = pos[i] - pos[j];
                for (int k = 0; k < 3; k++) d[k] -= L * std::floor(d[k] / L + 0.5);
                double r = std::sqrt(d.dot(d));
                if (r < 0.25) { pos[i] += d * (0.05 / r); }
            }
    system.addForce(mf);
    LangevinIntegrator integrator(0.0, 0.1, 0.01);
    Context context(system, integrator, platform);
    context.setPositions(pos);
    State st = context.getState(State::Forces | State::Energy);
    forces = st.getForces();
    energy = st.getPotentialEnergy();
}

int main(int argc, char* argv[]) {
    int bad = 0;
    if (argc > 1) {
        try {
            setupKernels(argc, argv);
            for (int level : {1, 2, 4, 3, 5, 6, 7, 9, 10, 12, 15}) {
                gLevel = level;
                std::vector<Vec3> fn, fr;
                double en = 0, er = 0;
                unsetenv("OPENMM_HIP_REFERENCE_AMOEBA_MULTIPOLE");
                randomSites(fn, en);
                setenv("OPENMM_HIP_REFERENCE_AMOEBA_MULTIPOLE", "1", 1);
                randomSites(fr, er);
                double rms = 0, worst = 0;
                for (size_t i = 0; i < fr.size(); i++) rms += fr[i].dot(fr[i]);
                rms = std::sqrt(rms / fr.size());
                for (size_t i = 0; i < fr.size(); i++) { Vec3 d = fn[i] - fr[i]; worst = std::max(worst, std::sqrt(d.dot(d)) / rms); }
                printf("level %2d (q%d d%d Q%d pol%d)  E native %.8f  E reference %.8f  diff %.2e   force max diff / rms %.2e\n", level, level & 1, (level >> 1) & 1, (level >> 2) & 1, (level >> 3) & 1, en, er, en - er, worst);
            }
        } catch (const std::exception& e) { printf("exception: %s\n", e.what()); return 2; }
        return 0;
    }
    try {
        setupKernels(argc, argv);
        const char* names[] = {"4 waters", "2 ions + 2 waters", "216 waters", "4 waters, mutual", "2 ions + 2 waters, mutual", "216 waters, mutual",
                               "4 waters, extrapolated", "2 ions + 2 waters, extrap.", "216 waters, extrapolated"};
        Case cases[] = {water4, ionsAndWater, water648, water4, ionsAndWater, water648, water4, ionsAndWater, water648};
        for (int c = 0; c < 9; c++) {
            // (Extrapolated: the force's default coefficients, OPT3 -- four perturbation orders)
            gPolarization = c < 3 ? AmoebaMultipoleForce::Direct : (c < 6 ? AmoebaMultipoleForce::Mutual : AmoebaMultipoleForce::Extrapolated);
            std::vector<Vec3> fn, fr;
            double en = 0, er = 0;
            unsetenv("OPENMM_HIP_REFERENCE_AMOEBA_MULTIPOLE");
            cases[c](fn, en);
            setenv("OPENMM_HIP_REFERENCE_AMOEBA_MULTIPOLE", "1", 1);
            cases[c](fr, er);
            double rms = 0, worst = 0;
            for (size_t i = 0; i < fr.size(); i++) rms += fr[i].dot(fr[i]);
            rms = std::sqrt(rms / fr.size());
            for (size_t i = 0; i < fr.size(); i++) { Vec3 d = fn[i] - fr[i]; worst = std::max(worst, std::sqrt(d.dot(d)) / rms); }
            printf("%-26s atoms %4d  E native %.8f  E reference %.8f  rel %.2e   force max diff / rms %.2e\n", names[c], (int) fr.size(), en, er, std::fabs(en - er) / std::max(std::fabs(er), 1.0), worst);
            if (getenv("AMOEBA_PARITY_VERBOSE") != NULL)
                for (size_t i = 0; i < fr.size() && i < 12; i++) printf("   %2d  native % .6f % .6f % .6f   reference % .6f % .6f % .6f\n", (int) i, fn[i][0], fn[i][1], fn[i][2], fr[i][0], fr[i][1], fr[i][2]);
            if (worst > 1e-4 || std::fabs(en - er) > 1e-4 * std::max(std::fabs(er), 1.0)) bad++;
        }
    }
    catch (const std::exception& e) {
        printf("exception: %s\n", e.what());
        return 2;
    }
    printf(bad == 0 ? "Done\n" : "MISMATCH\n");
    return bad == 0 ? 0 : 1;
}

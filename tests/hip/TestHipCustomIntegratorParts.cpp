// The reference's TestCustomIntegrator.h (included from $(REF)/tests at build time, never copied) with a driver that runs the tests named
// on the command line (all of them without arguments) and reports each: the CPU emulator is too slow for the long-running ones, so the
// emulator suite picks, the GPU suite runs the reference's own main() (TestHipCustomIntegrator).
#include "HipTests.h"
#define main reference_main
#include "TestCustomIntegrator.h"
#undef main
#include <chrono>
#include <cstdio>
#include <cstring>

void runPlatformTests() {
}

struct NamedTest { const char* name; void (*run)(); };
#define T(t) {#t, t}
static const NamedTest tests[] = {T(testSingleBond), T(testConstraints), T(testVelocityConstraints), T(testConstrainedMasslessParticles), T(testWithThermostat), T(testMonteCarlo), T(testSum),
                                  T(testParameter), T(testRandomDistributions), T(testPerDofVariables), T(testForceGroups), T(testRespa), T(testIfBlock), T(testWhileBlock), T(testChangingGlobal),
                                  T(testEnergyParameterDerivatives), T(testChangeDT), T(testTabulatedFunction), T(testAlternatingGroups), T(testUpdateContextState), T(testVectorFunctions),
                                  T(testRecordEnergy), T(testInitialTemperature), T(testCheckpoint), T(testSaveParameters)};

int main(int argc, char** argv) {
    initializeTests(1, argv);
    int failures = 0, ran = 0;
    for (const NamedTest& t : tests) {
        bool wanted = argc == 1;
        for (int i = 1; i < argc; i++) wanted = wanted || strcmp(argv[i], t.name) == 0;
        if (!wanted) continue;
        const auto t0 = std::chrono::steady_clock::now();
        try { t.run(); printf("%-36s ok   %.1f s\n", t.name, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); }
        catch (const std::exception& e) { printf("%-36s FAIL %s\n", t.name, e.what()); failures++; }
        fflush(stdout);
        ran++;
    }
    printf("%d tests, %d failures\n", ran, failures);
    if (failures == 0 && ran > 0) printf("Done\n");
    return failures == 0 && ran > 0 ? 0 : 1;
}

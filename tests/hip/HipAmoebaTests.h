// Platform header for the shared AMOEBA test bodies (plugins/amoeba/tests/TestAmoeba*.h of the OpenMM tree), the HIP twin of
// plugins/amoeba/platforms/reference/tests/ReferenceAmoebaTests.h:34-41: a global `platform` and setupKernels().
//
// Two plugins are loaded, the way an installation's plugin directory would bring them in:
//   libOpenMMAmoebaReference.so  the AMOEBA plugin's own Reference kernels; registerAmoebaReferenceKernelFactories() adds them to every
//                                platform derived from ReferencePlatform (AmoebaReferenceKernelFactory.cpp:47-58) -- on a HIP Context
//                                they are "fallback forces": host copy of the positions in, forces added to the device's buffer;
//   libOpenMMAmoebaHIP.so        the native kernels of this repository (openmm_amd/csrc/amoeba): AmoebaVdwForce and
//                                AmoebaMultipoleForce (PME) computed on the device; the forces they do not cover (torsion-torsion,
//                                bonded AMOEBA terms, ...) stay with the fallback.
// HIP_AMOEBA_FALLBACK_ONLY=1 leaves the native plugin out (the round-2 behaviour; kept as a test of the fallback path).  At exit the
// executable prints how many evaluations the native kernels performed -- tests/test_gpu_platform.py asserts that the count is not
// zero, so that a silent fallback to the Reference kernels cannot pass for a native run.
// A plugin library is loaded the way an installation loads it -- Platform::loadPluginLibrary (dlopen + the library's own
// registerKernelFactories) -- not linked: every OpenMM plugin exports the same two C symbols, so two of them cannot be linked
// into one executable.
#include "HipPlatform.h"
#include "openmm/Platform.h"
#include "openmm/OpenMMException.h"
#include <string>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <unistd.h>

OpenMM::HipPlatform platform;

static void reportNativeEvaluations() {
    void* self = dlopen(NULL, RTLD_NOW);
    void (*fn)(long long*) = self != NULL ? (void (*)(long long*)) dlsym(self, "ommhip_amoeba_native_evaluations") : NULL;
    long long n[2] = {0, 0};
    if (fn != NULL) fn(n);
    printf("native AMOEBA kernel evaluations: vdw %lld multipole %lld\n", n[0], n[1]);
    fflush(stdout);
}

void setupKernels(int argc, char* argv[]) {
    bool registered = false;
    for (int i = 0; i < OpenMM::Platform::getNumPlatforms(); i++)
        if (OpenMM::Platform::getPlatform(i).getName() == "HIP") registered = true;
    if (!registered) OpenMM::Platform::registerPlatform(new OpenMM::HipPlatform());
    // build/openmm/lib/libOpenMMAmoebaReference.so, relative to this executable (build/tests/ or tests/emu/_build/tests/)
    char exe[4096];
    const ssize_t n = readlink("/proc/self/exe", exe, sizeof(exe) - 1);
    std::string dir = n > 0 ? std::string(exe, n) : std::string(".");
    dir = dir.substr(0, dir.find_last_of('/'));
    const char* candidates[] = {"/../openmm/lib/", "/../../../../build/openmm/lib/"};
    bool loaded = false;
    if (getenv("OPENMM_AMOEBA_REFERENCE_LIB") != NULL) { OpenMM::Platform::loadPluginLibrary(getenv("OPENMM_AMOEBA_REFERENCE_LIB")); loaded = true; }
    for (int i = 0; i < 2 && !loaded; i++) {
        const std::string path = dir + candidates[i] + "libOpenMMAmoebaReference.so";
        if (access(path.c_str(), R_OK) == 0) { OpenMM::Platform::loadPluginLibrary(path); loaded = true; }
    }
    if (!loaded) throw OpenMM::OpenMMException("libOpenMMAmoebaReference.so not found next to the test executable");
    // the native kernels: libOpenMMAmoebaHIP.so sits next to the libOpenMMHIP.so this executable is linked to
    if (getenv("HIP_AMOEBA_FALLBACK_ONLY") == NULL) {
        const char* native[] = {"/../../openmm_amd/lib/libOpenMMAmoebaHIP.so", "/../libOpenMMAmoebaHIP.so"};      // build/tests/ -> product, tests/emu/_build/tests/ -> emulated twin
        bool nativeLoaded = false;
        for (int i = 0; i < 2 && !nativeLoaded; i++) {
            const std::string path = dir + native[i];
            if (access(path.c_str(), R_OK) == 0) {
                // RTLD_GLOBAL copy first, so that the evaluation counter can be found at exit; the plugin loader then registers its kernels
                dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
                OpenMM::Platform::loadPluginLibrary(path);
                nativeLoaded = true;
            }
        }
        if (!nativeLoaded) throw OpenMM::OpenMMException("libOpenMMAmoebaHIP.so not found");
    }
    atexit(reportNativeEvaluations);
    platform = dynamic_cast<OpenMM::HipPlatform&>(OpenMM::Platform::getPlatformByName("HIP"));
}

// Platform header for the shared AMOEBA test bodies (plugins/amoeba/tests/TestAmoeba*.h of the OpenMM tree), the HIP twin of
// plugins/amoeba/platforms/reference/tests/ReferenceAmoebaTests.h:34-41: a global `platform` and setupKernels().
//
// The HIP platform has no native AMOEBA kernels.  What it offers is what any platform derived from
// ReferencePlatform gets from the AMOEBA plugin itself: registerAmoebaReferenceKernelFactories() walks the registered
// platforms and adds the plugin's Reference kernels to every ReferencePlatform subclass (AmoebaReferenceKernelFactory.cpp:47-58).
// On a HIP Context they are "fallback forces": evaluated on the host copy of the positions, their forces added to the
// device's fixed-point buffer, while NonbondedForce / bonded terms / integration of the same System stay on the GPU.
// The plugin library is loaded the way an installation loads it -- Platform::loadPluginLibrary (dlopen + the library's own
// registerKernelFactories) -- not linked: every OpenMM plugin exports the same two C symbols, so two of them cannot be linked
// into one executable.
#include "HipPlatform.h"
#include "openmm/Platform.h"
#include "openmm/OpenMMException.h"
#include <string>
#include <cstdlib>
#include <unistd.h>

OpenMM::HipPlatform platform;

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
    platform = dynamic_cast<OpenMM::HipPlatform&>(OpenMM::Platform::getPlatformByName("HIP"));
}

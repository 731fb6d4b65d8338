// Platform header for the reference's own shared test bodies (tests/Test*.h of the OpenMM tree),
// the HIP twin of platforms/cuda/tests/CudaTests.h:35-43: a global `platform` object and
// initializeTests(argc, argv) where argv[1], if present, selects the Precision property.
#include "HipPlatform.h"
#include <string>

OpenMM::HipPlatform platform;

void initializeTests(int argc, char* argv[]) {
    if (argc > 1)
        platform.setPropertyDefaultValue("Precision", std::string(argv[1]));
}

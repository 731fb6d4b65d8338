/* libOpenMMAmoebaHIP.so: registers the native AMOEBA kernels with the "HIP" platform (olla/include/openmm/PluginInitializer.h:45-57;
 * the pattern of plugins/amoeba/platforms/cuda/src/AmoebaCudaKernelFactory.cpp). */
#include "HipAmoebaKernels.h"
#include "openmm/KernelFactory.h"
#include "openmm/OpenMMException.h"
#include "openmm/Platform.h"
#include "openmm/internal/ContextImpl.h"

using namespace OpenMM;

namespace {
class HipAmoebaKernelFactory : public KernelFactory {
public:
    KernelImpl* createKernelImpl(std::string name, const Platform& platform, ContextImpl& context) const {
        HipPlatform::PlatformData& data = HipPlatform::getData(context);
        if (name == CalcAmoebaVdwForceKernel::Name())
            return new HipCalcAmoebaVdwForceKernel(name, platform, data);
        throw OpenMMException((std::string("Tried to create kernel with illegal kernel name '") + name + "'").c_str());
    }
};
}

extern "C" __attribute__((visibility("default"))) void registerPlatforms() {
}

extern "C" __attribute__((visibility("default"))) void registerKernelFactories() {
    try {
        Platform& platform = Platform::getPlatformByName("HIP");
        if (dynamic_cast<HipPlatform*>(&platform) == NULL) return;
        HipAmoebaKernelFactory* factory = new HipAmoebaKernelFactory();
        HipPlatform::registerNativeKernel(CalcAmoebaVdwForceKernel::Name(), "AmoebaVdwForce", factory);
        platform.registerKernelFactory(CalcAmoebaVdwForceKernel::Name(), factory);
    }
    catch (std::exception&) {
        // no HIP platform in this process: nothing to register
    }
}

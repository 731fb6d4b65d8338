/* libOpenMMAmoebaHIP.so: registers the native AMOEBA kernels with the "HIP" platform (olla/include/openmm/PluginInitializer.h:45-57;
 * the pattern of plugins/amoeba/platforms/cuda/src/AmoebaCudaKernelFactory.cpp). */
#include "HipAmoebaKernels.h"
#include "AmoebaReferenceKernelFactory.h"
#include "openmm/AmoebaMultipoleForce.h"
#include "openmm/AmoebaTorsionTorsionForce.h"
#include "openmm/KernelFactory.h"
#include "openmm/System.h"
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
        if (name == CalcAmoebaTorsionTorsionForceKernel::Name()) {
            const System& system = context.getSystem();
            for (int i = 0; i < system.getNumForces(); i++) {
                const AmoebaTorsionTorsionForce* force = dynamic_cast<const AmoebaTorsionTorsionForce*>(&system.getForce(i));
                if (force != NULL && HipCalcAmoebaTorsionTorsionForceKernel::supports(*force))
                    return new HipCalcAmoebaTorsionTorsionForceKernel(name, platform, data);
            }
            return reference.createKernelImpl(name, platform, context);
        }
        if (name == CalcAmoebaMultipoleForceKernel::Name()) {
            // native for PME with direct polarization; everything else is the AMOEBA plugin's own Reference kernel (a fallback force:
            // HipPlatform's classification asks the same question through nativeMultipole below)
            const System& system = context.getSystem();
            for (int i = 0; i < system.getNumForces(); i++) {
                const AmoebaMultipoleForce* force = dynamic_cast<const AmoebaMultipoleForce*>(&system.getForce(i));
                if (force != NULL && HipCalcAmoebaMultipoleForceKernel::supports(*force, system))
                    return new HipCalcAmoebaMultipoleForceKernel(name, platform, data, reference.createKernelImpl(name, platform, context));
            }
            return reference.createKernelImpl(name, platform, context);
        }
        throw OpenMMException((std::string("Tried to create kernel with illegal kernel name '") + name + "'").c_str());
    }
private:
    AmoebaReferenceKernelFactory reference;
};

bool nativeTorsionTorsion(const Force& force, const System& system) {
    const AmoebaTorsionTorsionForce* tt = dynamic_cast<const AmoebaTorsionTorsionForce*>(&force);
    return tt != NULL && HipCalcAmoebaTorsionTorsionForceKernel::supports(*tt);
}

bool nativeMultipole(const Force& force, const System& system) {
    const AmoebaMultipoleForce* mp = dynamic_cast<const AmoebaMultipoleForce*>(&force);
    return mp != NULL && HipCalcAmoebaMultipoleForceKernel::supports(*mp, system);
}
}

extern "C" __attribute__((visibility("default"))) void registerPlatforms() {
}

extern "C" __attribute__((visibility("default"))) void registerKernelFactories() {
    try {
        Platform& platform = Platform::getPlatformByName("HIP");
        if (dynamic_cast<HipPlatform*>(&platform) == NULL) return;
        HipAmoebaKernelFactory* factory = new HipAmoebaKernelFactory();
        HipPlatform::registerNativeKernel(CalcAmoebaVdwForceKernel::Name(), "AmoebaVdwForce", factory);
        HipPlatform::registerNativeKernel(CalcAmoebaMultipoleForceKernel::Name(), "AmoebaMultipoleForce", factory, nativeMultipole);
        HipPlatform::registerNativeKernel(CalcAmoebaTorsionTorsionForceKernel::Name(), "AmoebaTorsionTorsionForce", factory, nativeTorsionTorsion);
        platform.registerKernelFactory(CalcAmoebaTorsionTorsionForceKernel::Name(), factory);
        platform.registerKernelFactory(CalcAmoebaVdwForceKernel::Name(), factory);
        platform.registerKernelFactory(CalcAmoebaMultipoleForceKernel::Name(), factory);
    }
    catch (std::exception&) {
        // no HIP platform in this process: nothing to register
    }
}

#ifndef OPENMM_HIPAMOEBAKERNELS_H_
#define OPENMM_HIPAMOEBAKERNELS_H_
/* Native AMOEBA kernels of the OpenMM "HIP" platform (BASELINE.json configs[4]) -- the contents of
 * libOpenMMAmoebaHIP.so, the counterpart of the reference's libOpenMMAmoebaCUDA: a plugin of its own that links the AMOEBA
 * plugin's API library (libOpenMMAmoeba) and the HIP platform (libOpenMMHIP), so that neither of those depends on the other.
 * Each class derives from the abstract kernel of plugins/amoeba/openmmapi/include/openmm/amoebaKernels.h and states the Reference
 * implementation it is checked against.
 */
#include "HipPlatform.h"
#include "HipContext.h"
#include "openmm/amoebaKernels.h"
#include "openmm_hip_amoeba.h"

namespace OpenMM {

/** amoebaKernels.h:182-219 CalcAmoebaVdwForceKernel; Reference: AmoebaReferenceKernels.cpp:643-696 + AmoebaReferenceVdwForce.cpp. */
class HipCalcAmoebaVdwForceKernel : public CalcAmoebaVdwForceKernel {
public:
    HipCalcAmoebaVdwForceKernel(std::string name, const Platform& platform, HipPlatform::PlatformData& data) : CalcAmoebaVdwForceKernel(name, platform), data(data) {}
    void initialize(const System& system, const AmoebaVdwForce& force);
    double execute(ContextImpl& context, bool includeForces, bool includeEnergy);
    void copyParametersToContext(ContextImpl& context, const AmoebaVdwForce& force);
private:
    void upload(const AmoebaVdwForce& force);
    HipPlatform::PlatformData& data;
    int numParticles = 0;
    bool usePBC = false;
    double cutoff = 0, dispersionCoefficient = 0, softcorePower = 0, softcoreAlpha = 0;
    ommhip_amoeba_vdw vdw;
    DeviceBuffer parent, reduction, type, sigma, epsilon, exclStart, exclAtoms, alchemical, reduced;
};

}  // namespace OpenMM
#endif

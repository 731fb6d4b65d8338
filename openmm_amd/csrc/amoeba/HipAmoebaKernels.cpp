#include "HipAmoebaKernels.h"
#include "openmm/AmoebaVdwForce.h"
#include "openmm/internal/AmoebaVdwForceImpl.h"
#include "openmm/internal/ContextImpl.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <set>
#include <vector>

using namespace OpenMM;
using namespace std;

namespace {
template <class T>
void uploadVector(DeviceBuffer& buffer, const vector<T>& v, void* stream) {
    buffer.allocate(sizeof(T) * max(v.size(), (size_t) 1));
    if (!v.empty()) HIP_CHECK(ommhip_memcpy_h2d(buffer.ptr, v.data(), sizeof(T) * v.size(), stream));
    HIP_CHECK(ommhip_stream_sync(stream));       // v may be a temporary
}
long long nativeEvaluations[2] = {0, 0};
}

/* Diagnostics for the tests: how many force evaluations went through the native kernels ([0] vdW, [1] multipole) -- a Context that
 * silently fell back to the AMOEBA plugin's Reference kernels leaves these at zero. */
extern "C" __attribute__((visibility("default"))) void ommhip_amoeba_native_evaluations(long long* out) {
    out[0] = nativeEvaluations[0]; out[1] = nativeEvaluations[1];
}

// ================================================================================================
// AmoebaVdwForce
// ================================================================================================
void HipCalcAmoebaVdwForceKernel::initialize(const System& system, const AmoebaVdwForce& force) {
    // AmoebaReferenceKernels.cpp:656-667
    numParticles = system.getNumParticles();
    if (force.getNonbondedMethod() != AmoebaVdwForce::NoCutoff && force.getNonbondedMethod() != AmoebaVdwForce::CutoffPeriodic)
        throw OpenMMException("HIP platform: AmoebaVdwForce supports NoCutoff and CutoffPeriodic");
    usePBC = force.getNonbondedMethod() == AmoebaVdwForce::CutoffPeriodic;
    cutoff = force.getCutoffDistance();
    dispersionCoefficient = force.getUseDispersionCorrection() ? AmoebaVdwForceImpl::calcDispersionCorrection(system, force) : 0.0;
    if (usePBC) data.hip->usePeriodic = true;
    upload(force);
}

void HipCalcAmoebaVdwForceKernel::upload(const AmoebaVdwForce& force) {
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    // AmoebaReferenceVdwForce::initialize (AmoebaReferenceVdwForce.cpp:40-67)
    vector<int> particleType;
    vector<vector<double> > sigmaMatrix, epsilonMatrix;
    AmoebaVdwForceImpl::createParameterMatrix(force, particleType, sigmaMatrix, epsilonMatrix);
    const int numTypes = (int) sigmaMatrix.size();
    vector<double> sig((size_t) numTypes * numTypes), eps((size_t) numTypes * numTypes);
    for (int i = 0; i < numTypes; i++)
        for (int j = 0; j < numTypes; j++) { sig[(size_t) i * numTypes + j] = sigmaMatrix[i][j]; eps[(size_t) i * numTypes + j] = epsilonMatrix[i][j]; }
    vector<int> parents(numParticles), start(numParticles + 1, 0), flat;
    vector<double> reductions(numParticles);
    vector<unsigned char> alch(numParticles);
    for (int i = 0; i < numParticles; i++) {
        int type;
        double sigma, epsilon;
        bool isAlchemical;
        force.getParticleParameters(i, parents[i], sigma, epsilon, reductions[i], isAlchemical, type);
        alch[i] = isAlchemical ? 1 : 0;
        vector<int> exclusions;
        force.getParticleExclusions(i, exclusions);
        set<int> sorted(exclusions.begin(), exclusions.end());        // ascending, each once: the kernel consumes the row with one cursor
        flat.insert(flat.end(), sorted.begin(), sorted.end());
        start[i + 1] = (int) flat.size();
    }
    uploadVector(parent, parents, hip.stream); uploadVector(reduction, reductions, hip.stream); uploadVector(type, particleType, hip.stream);
    uploadVector(sigma, sig, hip.stream); uploadVector(epsilon, eps, hip.stream);
    uploadVector(exclStart, start, hip.stream); uploadVector(exclAtoms, flat, hip.stream); uploadVector(alchemical, alch, hip.stream);
    reduced.allocate(sizeof(double) * 4 * (size_t) max(numParticles, 1));
    memset(&vdw, 0, sizeof(vdw));
    vdw.num_atoms = numParticles; vdw.num_types = numTypes;
    vdw.parent = parent.as<int>(); vdw.reduction = reduction.as<double>(); vdw.type = type.as<int>();
    vdw.sigma = sigma.as<double>(); vdw.epsilon = epsilon.as<double>();
    vdw.excl_start = exclStart.as<int>(); vdw.excl_atoms = exclAtoms.as<int>(); vdw.alchemical = alchemical.as<unsigned char>();
    vdw.alchemical_method = force.getAlchemicalMethod() == AmoebaVdwForce::Decouple ? 1 : (force.getAlchemicalMethod() == AmoebaVdwForce::Annihilate ? 2 : 0);
    softcorePower = force.getSoftcorePower(); softcoreAlpha = force.getSoftcoreAlpha();
    vdw.lennard_jones = force.getPotentialFunction() == AmoebaVdwForce::LennardJones ? 1 : 0;
    vdw.periodic = usePBC ? 1 : 0;
    // AmoebaReferenceVdwForce::setTaperCoefficients (:69-81), taper between 0.9 cutoff and the cutoff
    vdw.cutoff = cutoff; vdw.taper_cutoff = 0.9 * cutoff;
    if (usePBC && vdw.taper_cutoff != cutoff) {
        vdw.taper_c3 = 10.0 / pow(vdw.taper_cutoff - cutoff, 3.0);
        vdw.taper_c4 = 15.0 / pow(vdw.taper_cutoff - cutoff, 4.0);
        vdw.taper_c5 = 6.0 / pow(vdw.taper_cutoff - cutoff, 5.0);
    }
    vdw.reduced = reduced.as<double>();
}

double HipCalcAmoebaVdwForceKernel::execute(ContextImpl& context, bool includeForces, bool includeEnergy) {
    // AmoebaReferenceKernels.cpp:669-690
    HipContext& hip = *data.hip;
    hip.setAsCurrent();
    const double lambda = context.getParameter(AmoebaVdwForce::Lambda());
    vdw.epsilon_scale = pow(lambda, softcorePower);
    vdw.softcore = softcoreAlpha * (1.0 - lambda) * (1.0 - lambda);
    if (usePBC) {
        const double minAllowedSize = 1.999999 * cutoff;
        if (hip.box[0] < minAllowedSize || hip.box[2] < minAllowedSize || hip.box[5] < minAllowedSize)
            throw OpenMMException("The periodic box size has decreased to less than twice the cutoff.");
    }
    hip.ensureCleared();
    HIP_CHECK(ommhip_amoeba_vdw_forces(&vdw, hip.pos.ptr, hip.box, hip.slotOfAtom.as<int>(), hip.paddedAtoms, hip.force.as<long long>(),
                                       hip.energyBuffer.as<double>(), HipContext::EnergySlots, includeEnergy ? 1 : 0, hip.stream));
    nativeEvaluations[0]++;
    // the pair energy is summed on the device (HipCalcForcesAndEnergyKernel::finishComputation); the host adds the constant
    return includeEnergy && usePBC ? dispersionCoefficient / (hip.box[0] * hip.box[2] * hip.box[5]) : 0.0;
}

void HipCalcAmoebaVdwForceKernel::copyParametersToContext(ContextImpl& context, const AmoebaVdwForce& force) {
    if (numParticles != force.getNumParticles())
        throw OpenMMException("updateParametersInContext: The number of particles has changed");
    upload(force);
}

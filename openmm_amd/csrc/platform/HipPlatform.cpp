/* HipPlatform: registration, per-Context data, kernel factory.
 *
 * Drop-in boundary: this shared object exports extern "C" registerPlatforms(), which
 * olla/src/Platform.cpp:237-252 resolves with dlsym after dlopen; it registers one HipPlatform whose
 * KernelFactory hands out the native kernels of HipKernels.h and, for every other kernel name, the
 * Reference implementation (the platforms/cpu precedent, CpuPlatform.cpp:63-77).
 */
#include "HipPlatform.h"
#include <typeinfo>
#include <vector>
#include "HipContext.h"
#include "HipKernels.h"
#include "HipValenceKernels.h"
#include "HipCustomIntegrator.h"
#include "HipParallel.h"
#include "ReferenceKernelFactory.h"
#include "openmm/Context.h"
#include "openmm/KernelFactory.h"
#include "openmm/PluginInitializer.h"
#include "openmm/System.h"
#include "openmm/VirtualSite.h"
#include "openmm/CMMotionRemover.h"
#include "openmm/HarmonicAngleForce.h"
#include "openmm/HarmonicBondForce.h"
#include "openmm/NonbondedForce.h"
#include "openmm/PeriodicTorsionForce.h"
#include "openmm/AndersenThermostat.h"
#include "openmm/MonteCarloBarostat.h"
#include "openmm/MonteCarloAnisotropicBarostat.h"
#include "openmm/MonteCarloMembraneBarostat.h"
#include "openmm/CustomCVForce.h"
#include "openmm/LangevinIntegrator.h"
#include "openmm/LangevinMiddleIntegrator.h"
#include "openmm/VerletIntegrator.h"
#include "openmm/kernels.h"
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <sstream>

using namespace OpenMM;
using namespace std;

namespace OpenMM {

/* How a Context runs, decided once in contextCreated():
 *   device mode  -- native integrator, constraints and state kernels; positions/velocities live in HBM.
 *                   Forces without a native kernel run as Reference kernels on a host copy of the
 *                   positions ("fallback forces") and their result is added to the device buffer.
 *   host mode    -- anything that needs to mutate the state on the host (non-native integrators,
 *                   barostats, thermostats, virtual sites, unknown Force subclasses): all state and
 *                   integration kernels are the Reference ones; the native force kernels still compute
 *                   on the GPU from an uploaded copy of the positions.
 */
struct HipModeInfo {
    bool hasBarostat;
    bool hostMode;
    bool referenceNonbonded;     // LJPME: NonbondedForce itself falls back to Reference
    bool hasFallbackForces;
    bool hasPluginNativeForces;  // forces evaluated by a native kernel of another plugin (registerNativeKernel), e.g. the AMOEBA forces
    bool hasValenceForces;       // Custom*Forces with a recognised expression (the AMOEBA valence terms): native, single GPU only
    bool customIntegrator;       // a CustomIntegrator run by the device interpreter (single GPU only)
    std::string fallbackNames;   // class names of the Forces that run as Reference kernels (the FallbackForces property)
};

namespace {
struct NativeKernel { string kernelName, forceType; KernelFactory* factory; HipPlatform::NativeForceTest supported; };
vector<NativeKernel>& nativeKernels() { static vector<NativeKernel> v; return v; }
}

void HipPlatform::registerNativeKernel(const string& kernelName, const string& forceType, KernelFactory* factory, NativeForceTest supported) {
    for (size_t i = 0; i < nativeKernels().size(); i++)
        if (nativeKernels()[i].kernelName == kernelName) { nativeKernels()[i].forceType = forceType; nativeKernels()[i].factory = factory; nativeKernels()[i].supported = supported; return; }
    NativeKernel k = {kernelName, forceType, factory, supported};
    nativeKernels().push_back(k);
}

bool HipPlatform::isNativeForce(const Force& force, const System& system) {
    const string typeName = typeid(force).name();
    for (size_t i = 0; i < nativeKernels().size(); i++)
        if (!nativeKernels()[i].forceType.empty() && typeName.find(nativeKernels()[i].forceType) != string::npos)
            return nativeKernels()[i].supported == NULL || nativeKernels()[i].supported(force, system);
    return false;
}

static HipModeInfo classifyContext(ContextImpl& context) {
    HipModeInfo info = {false, false, false, false, false, false, false, ""};
    const System& system = context.getSystem();
    const Integrator& integrator = context.getIntegrator();
    if (dynamic_cast<const VerletIntegrator*>(&integrator) == NULL && dynamic_cast<const LangevinIntegrator*>(&integrator) == NULL &&
            dynamic_cast<const LangevinMiddleIntegrator*>(&integrator) == NULL) {
        // a CustomIntegrator whose expressions all have a device form runs natively (HipCustomIntegrator.h); anything else integrates on the host
        const CustomIntegrator* custom = dynamic_cast<const CustomIntegrator*>(&integrator);
        if (custom == NULL || !HipIntegrateCustomStepKernel::supports(*custom)) info.hostMode = true;
        else info.customIntegrator = true;
    }
    for (int i = 0; i < system.getNumParticles(); i++)
        if (system.isVirtualSite(i)) info.hostMode = true;
    for (int i = 0; i < system.getNumForces(); i++) {
        const Force& f = system.getForce(i);
        if (const NonbondedForce* nb = dynamic_cast<const NonbondedForce*>(&f)) {
            // LJPME is native too (dispersion grid + direct-space correction); OPENMM_HIP_REFERENCE_LJPME=1 restores the Reference kernel (A/B)
            char* refLj = getenv("OPENMM_HIP_REFERENCE_LJPME");
            if (nb->getNonbondedMethod() == NonbondedForce::LJPME && refLj != NULL && string(refLj) == "1") { info.referenceNonbonded = true; info.hasFallbackForces = true; info.fallbackNames += string(info.fallbackNames.empty() ? "" : ",") + "NonbondedForce"; }
            continue;
        }
        if (dynamic_cast<const HarmonicBondForce*>(&f) != NULL || dynamic_cast<const HarmonicAngleForce*>(&f) != NULL ||
                dynamic_cast<const PeriodicTorsionForce*>(&f) != NULL || dynamic_cast<const CMMotionRemover*>(&f) != NULL)
            continue;
        if (dynamic_cast<const MonteCarloBarostat*>(&f) != NULL || dynamic_cast<const MonteCarloAnisotropicBarostat*>(&f) != NULL) {
            info.hasBarostat = true;        // native: positions are scaled / restored on the device (HipApplyMonteCarloBarostatKernel)
            continue;
        }
        if (dynamic_cast<const AndersenThermostat*>(&f) != NULL ||
                dynamic_cast<const MonteCarloAnisotropicBarostat*>(&f) != NULL || dynamic_cast<const MonteCarloMembraneBarostat*>(&f) != NULL ||
                dynamic_cast<const CustomCVForce*>(&f) != NULL) {
            info.hostMode = true;       // these change state (or own an inner Context) on the host
            continue;
        }
        if (HipPlatform::isNativeForce(f, system)) {
            info.hasPluginNativeForces = true;       // a native kernel from a plugin of its own (registerNativeKernel)
            continue;
        }
        if (HipValenceForm::isNative(f)) {
            info.hasValenceForces = true;            // a Custom*Force whose expression has a hand-written kernel (HipValenceKernels.h)
            continue;
        }
        // Any other Force: its Reference kernel only reads positions and adds forces.
        info.hasFallbackForces = true;
        {
            // (the mangled class name without its length prefixes: "N6OpenMM12GBSAOBCForceE" -> "GBSAOBCForce")
            string n = typeid(f).name();
            const size_t at = n.find("OpenMM");
            if (at != string::npos) n = n.substr(at + 6);
            while (!n.empty() && isdigit((unsigned char) n[0])) n.erase(0, 1);
            if (!n.empty() && n[n.size() - 1] == 'E') n.erase(n.size() - 1);
            info.fallbackNames += (info.fallbackNames.empty() ? "" : ",") + n;
        }
    }
    char* forceHost = getenv("OPENMM_HIP_FORCE_HOST_MODE");
    if (forceHost != NULL && string(forceHost) == "1") info.hostMode = true;
    return info;
}

class HipKernelFactory : public KernelFactory {
public:
    KernelImpl* createKernelImpl(std::string name, const Platform& platform, ContextImpl& context) const {
        HipPlatform::PlatformData& data = HipPlatform::getData(context);
        KernelImpl* kernel = createOwnKernel(name, platform, context, data);
        data.kernelsByName[name].push_back(kernel);
        if (data.group != NULL) {
            // the user's Context of a device list: its kernels also drive the peer kernels of the inner ranks (HipParallel.h)
            KernelImpl* wrapped = hipMakeParallelKernel(name, platform, data, kernel);
            if (wrapped != NULL) return wrapped;
        }
        return kernel;
    }
private:
    KernelImpl* createOwnKernel(const std::string& name, const Platform& platform, ContextImpl& context, HipPlatform::PlatformData& data) const {
        const bool hostMode = data.hip->hostMode;
        if (name == CalcForcesAndEnergyKernel::Name())
            return new HipCalcForcesAndEnergyKernel(name, platform, data);
        if (name == CalcNonbondedForceKernel::Name() && !data.referenceNonbonded)
            return new HipCalcNonbondedForceKernel(name, platform, data);
        if (name == CalcPmeReciprocalForceKernel::Name())
            return new HipCalcPmeReciprocalForceKernel(name, platform, data.hip->getDeviceIndex());
        if (name == CalcHarmonicBondForceKernel::Name())
            return new HipCalcHarmonicBondForceKernel(name, platform, data);
        if (name == CalcHarmonicAngleForceKernel::Name())
            return new HipCalcHarmonicAngleForceKernel(name, platform, data);
        if (name == CalcPeriodicTorsionForceKernel::Name())
            return new HipCalcPeriodicTorsionForceKernel(name, platform, data);
        // Custom*Forces: native for the expressions of the AMOEBA valence terms, the Reference kernel (held inside) for everything else
        if (name == CalcCustomBondForceKernel::Name())
            return new HipCalcCustomBondForceKernel(name, platform, data, reference.createKernelImpl(name, platform, context));
        if (name == CalcCustomAngleForceKernel::Name())
            return new HipCalcCustomAngleForceKernel(name, platform, data, reference.createKernelImpl(name, platform, context));
        if (name == CalcCustomCompoundBondForceKernel::Name())
            return new HipCalcCustomCompoundBondForceKernel(name, platform, data, reference.createKernelImpl(name, platform, context));
        if (!hostMode) {
            if (name == UpdateStateDataKernel::Name())
                return new HipUpdateStateDataKernel(name, platform, data);
            if (name == ApplyConstraintsKernel::Name())
                return new HipApplyConstraintsKernel(name, platform, data);
            if (name == VirtualSitesKernel::Name())
                return new HipVirtualSitesKernel(name, platform);
            if (name == IntegrateVerletStepKernel::Name())
                return new HipIntegrateVerletStepKernel(name, platform, data);
            if (name == IntegrateLangevinStepKernel::Name())
                return new HipIntegrateLangevinStepKernel(name, platform, data);
            if (name == IntegrateLangevinMiddleStepKernel::Name())
                return new HipIntegrateLangevinMiddleStepKernel(name, platform, data);
            if (name == IntegrateCustomStepKernel::Name())
                return new HipIntegrateCustomStepKernel(name, platform, data);
            if (name == RemoveCMMotionKernel::Name())
                return new HipRemoveCMMotionKernel(name, platform, data);
            if (name == ApplyMonteCarloBarostatKernel::Name())
                return new HipApplyMonteCarloBarostatKernel(name, platform, data);
        }
        return reference.createKernelImpl(name, platform, context);
    }
    ReferenceKernelFactory reference;
};

}  // namespace OpenMM

extern "C" __attribute__((visibility("default"))) void registerPlatforms() {
    Platform::registerPlatform(new HipPlatform());
}

extern "C" __attribute__((visibility("default"))) void registerKernelFactories() {
}

HipPlatform::HipPlatform() {
    HipKernelFactory* factory = new HipKernelFactory();
    registerKernelFactory(CalcForcesAndEnergyKernel::Name(), factory);
    registerKernelFactory(UpdateStateDataKernel::Name(), factory);
    registerKernelFactory(ApplyConstraintsKernel::Name(), factory);
    registerKernelFactory(VirtualSitesKernel::Name(), factory);
    registerKernelFactory(CalcNonbondedForceKernel::Name(), factory);
    registerKernelFactory(CalcPmeReciprocalForceKernel::Name(), factory);
    registerKernelFactory(CalcHarmonicBondForceKernel::Name(), factory);
    registerKernelFactory(CalcHarmonicAngleForceKernel::Name(), factory);
    registerKernelFactory(CalcPeriodicTorsionForceKernel::Name(), factory);
    registerKernelFactory(CalcCustomBondForceKernel::Name(), factory);
    registerKernelFactory(CalcCustomAngleForceKernel::Name(), factory);
    registerKernelFactory(CalcCustomCompoundBondForceKernel::Name(), factory);
    registerKernelFactory(IntegrateVerletStepKernel::Name(), factory);
    registerKernelFactory(IntegrateLangevinStepKernel::Name(), factory);
    registerKernelFactory(IntegrateLangevinMiddleStepKernel::Name(), factory);
    registerKernelFactory(IntegrateCustomStepKernel::Name(), factory);
    registerKernelFactory(RemoveCMMotionKernel::Name(), factory);
    registerKernelFactory(ApplyMonteCarloBarostatKernel::Name(), factory);
    platformProperties.push_back(HipDeviceIndex());
    platformProperties.push_back(HipDeviceName());
    platformProperties.push_back(HipPrecision());
    platformProperties.push_back(HipDeterministicForces());
    platformProperties.push_back(HipDisablePmeStream());
    platformProperties.push_back(HipIntegrationMode());
    platformProperties.push_back(HipConstraintPartition());
    platformProperties.push_back(HipFallbackForces());
    platformProperties.push_back(HipRanks());
    platformProperties.push_back(HipRank());
    platformProperties.push_back(HipCommId());
    setPropertyDefaultValue(HipRanks(), "1");
    setPropertyDefaultValue(HipIntegrationMode(), "");
    setPropertyDefaultValue(HipConstraintPartition(), "");
    setPropertyDefaultValue(HipFallbackForces(), "");
    setPropertyDefaultValue(HipRank(), "0");
    setPropertyDefaultValue(HipCommId(), "");
    setPropertyDefaultValue(HipDeviceIndex(), "");
    setPropertyDefaultValue(HipDeviceName(), "");
    setPropertyDefaultValue(HipPrecision(), "mixed");
    setPropertyDefaultValue(HipDeterministicForces(), "false");
    // Default: ONE stream.  The work that is independent at the start of an evaluation (list rebuild, charge spreading,
    // per-term forces) overlaps inside one fused launch (ommhip_force_front) instead of across streams: a cross-stream
    // dependency costs ~13 us on this stack, twice per step.  Measured on MI355X, DHFR-size workload, same box:
    // fused single stream 1275 ns/day, side stream (DisablePmeStream=false) 1214, single stream without fusion 1182.
    const char* streamDefault = getenv("OPENMM_HIP_DEFAULT_DISABLE_PME_STREAM");      // test / A-B knob for the default
    setPropertyDefaultValue(HipDisablePmeStream(), streamDefault != NULL && streamDefault[0] == '0' ? "false" : "true");
}

double HipPlatform::getSpeed() const {
    return 150;    // above CUDA's 100 (CudaPlatform.cpp:149-151): preferred when a MI355X is present
}

bool HipPlatform::supportsDoublePrecision() const {
    return false;  // forces are evaluated in single precision; integration state is double ("mixed")
}

const string& HipPlatform::getPropertyValue(const Context& context, const string& property) const {
    const ContextImpl& impl = getContextImpl(context);
    const PlatformData* data = reinterpret_cast<const PlatformData*>(impl.getPlatformData());
    map<string, string>::const_iterator value = data->propertyValues.find(property);
    if (value != data->propertyValues.end())
        return value->second;
    return ReferencePlatform::getPropertyValue(context, property);
}

void HipPlatform::setPropertyValue(Context& context, const string& property, const string& value) const {
    // every property of this platform shapes the Context at creation (device, streams, communicator): none can change afterwards
    throw OpenMMException("HIP platform: the property '" + property + "' cannot be changed after the Context has been created");
}

void HipPlatform::contextCreated(ContextImpl& context, const map<string, string>& properties) const {
    const string& devicePropValue = (properties.find(HipDeviceIndex()) == properties.end() ?
            getPropertyDefaultValue(HipDeviceIndex()) : properties.find(HipDeviceIndex())->second);
    string precision = (properties.find(HipPrecision()) == properties.end() ?
            getPropertyDefaultValue(HipPrecision()) : properties.find(HipPrecision())->second);
    if (precision != "mixed")
        throw OpenMMException("HIP platform: Precision must be 'mixed' (single-precision pair and PME arithmetic, 64-bit fixed-point force accumulation, "
                              "double-precision integration and constraints); 'single' and 'double' modes do not exist on this platform");
    int deviceIndex = 0;
    vector<int> deviceList;
    if (devicePropValue.find(',') != string::npos) {
        // a list of devices: ONE box over all of them, each device a rank of the decomposition (HipParallel.h)
        stringstream list(devicePropValue);
        string item;
        while (getline(list, item, ',')) {
            int d = -1;
            if (!(stringstream(item) >> d) || d < 0) throw OpenMMException("HIP platform: illegal DeviceIndex list '" + devicePropValue + "'");
            deviceList.push_back(d);
        }
        if (deviceList.size() < 2) throw OpenMMException("HIP platform: illegal DeviceIndex list '" + devicePropValue + "'");
        if (properties.find(HipRanks()) != properties.end() || properties.find(HipCommId()) != properties.end())
            throw OpenMMException("HIP platform: give either a list of devices (one process) or Ranks / Rank / CommId (one process per device), not both");
        deviceIndex = deviceList[0];
    }
    else if (!devicePropValue.empty())
        stringstream(devicePropValue) >> deviceIndex;
    else if (getenv("LOCAL_RANK") != NULL) {
        // one process per GPU under torch.distributed.run: default to this rank's device
        int count = 0;
        if (ommhip_device_count(&count) == 0 && count > 0)
            deviceIndex = atoi(getenv("LOCAL_RANK")) % count;
    }
    // native kernels of other plugins take precedence over Reference kernels registered for the same names (see registerNativeKernel)
    // (a device-list Context creates its inner Contexts on threads of their own: the ranks must not re-assert side by side -- and need not, the
    // user's Context has done it before it started them)
    static std::mutex reassertMutex;
    if (!hipCreatingInnerRank()) {
        std::lock_guard<std::mutex> lock(reassertMutex);
        for (size_t i = 0; i < nativeKernels().size(); i++)
            const_cast<HipPlatform*>(this)->registerKernelFactory(nativeKernels()[i].kernelName, nativeKernels()[i].factory);
    }
    HipModeInfo mode = classifyContext(context);
    // ---- one box on several GPUs (one process per GPU)
    HipDomain domain;
    string commId;
    if (properties.find(HipRanks()) != properties.end()) stringstream(properties.find(HipRanks())->second) >> domain.ranks;
    if (properties.find(HipRank()) != properties.end()) stringstream(properties.find(HipRank())->second) >> domain.rank;
    if (properties.find(HipCommId()) != properties.end()) commId = properties.find(HipCommId())->second;
    if (domain.ranks < 1 || domain.rank < 0 || domain.rank >= domain.ranks)
        throw OpenMMException("HIP platform: illegal Ranks/Rank properties");
    HipRankGroup* group = NULL;
    if (!deviceList.empty()) {
        if (mode.customIntegrator || mode.hostMode)
            throw OpenMMException("HIP platform: a Context over a list of devices supports the Verlet, Langevin and LangevinMiddle integrators");
        // (every device of the list is checked HERE: a rank that failed before it reached the communicator would leave the others waiting in its creation)
        int visible = 0;
        HIP_CHECK(ommhip_device_count(&visible));
        for (size_t i = 0; i < deviceList.size(); i++)
            if (deviceList[i] >= visible) {
                stringstream msg;
                msg << "HIP platform: DeviceIndex list '" << devicePropValue << "' names device " << deviceList[i] << ", but only " << visible << " device(s) are visible";
                throw OpenMMException(msg.str());
            }
        if (mode.hasValenceForces || mode.hasPluginNativeForces || mode.hasFallbackForces || mode.referenceNonbonded)
            throw OpenMMException("HIP platform: a Context over a list of devices supports NonbondedForce (PME), HarmonicBond/Angle, PeriodicTorsion, CMMotionRemover and the MonteCarloBarostat");
        // the threads of the other devices start creating their Contexts now; this thread goes on as rank 0 and meets them in the communicator
        group = new HipRankGroup(*this, context, deviceList, properties);
        domain.ranks = (int) deviceList.size(); domain.rank = 0; commId = group->commId();
    }
    unsigned long long forcedSeed = group != NULL ? group->seed() : 0;
    if (domain.ranks > 1 || !commId.empty()) {
        // Kernels of other plugins (the AMOEBA forces) know nothing of the decomposition: each rank would evaluate the whole system
        // from positions that are current only for its own atoms and its halo, and add the full energy on every rank.
        if (mode.customIntegrator)
            throw OpenMMException("HIP platform: a multi-GPU Context supports the Verlet, Langevin and LangevinMiddle integrators (a CustomIntegrator runs on one GPU)");
        if (mode.hasValenceForces)
            throw OpenMMException("HIP platform: a multi-GPU Context cannot hold the Custom*Forces of an AMOEBA force field (their native kernels evaluate the whole system on one GPU)");
        if (mode.hasPluginNativeForces)
            throw OpenMMException("HIP platform: a multi-GPU Context cannot hold forces whose native kernels come from another plugin (AmoebaVdwForce, AmoebaMultipoleForce): they evaluate the whole system on one GPU");
        if (mode.hostMode || mode.hasFallbackForces || mode.referenceNonbonded)
            throw OpenMMException("HIP platform: a multi-GPU Context supports NonbondedForce (PME), HarmonicBond/Angle, PeriodicTorsion, CMMotionRemover and the (isotropic or anisotropic) MonteCarloBarostat with the Verlet, Langevin and LangevinMiddle integrators");
        int count = 0;
        HIP_CHECK(ommhip_device_count(&count));
        if (deviceIndex < 0 || deviceIndex >= count) throw OpenMMException("HIP platform: illegal DeviceIndex");
        HIP_CHECK(ommhip_set_device(deviceIndex));       // the communicator binds to the current device
        if (commId.compare(0, 10, "inprocess:") == 0) {
            // the ranks of a device list that share a device (or have no RCCL): host-staged all-gather between the threads of this process
            const int token = atoi(commId.c_str() + 10);
            void* user = hipInProcessUser(token, domain.rank);
            if (user == NULL) throw OpenMMException("HIP platform: unknown in-process communicator");
            HIP_CHECK(ommhip_comm_create_callback(hipInProcessAllGather, user, domain.rank, domain.ranks, &domain.comm));
        }
        else if (commId == "alone") {
            // diagnostics (bench.py --rank-alone): this rank without its peers, collectives that do nothing -- refused unless the process asks for it
            const char* allow = getenv("OPENMM_HIP_ALLOW_ALONE_COMM");
            if (allow == NULL || allow[0] != '1')
                throw OpenMMException("HIP platform: CommId \"alone\" (a rank without its peers: timing diagnostics, meaningless forces) needs OPENMM_HIP_ALLOW_ALONE_COMM=1");
            HIP_CHECK(ommhip_comm_create_alone(domain.rank, domain.ranks, &domain.comm));
        }
        else if (commId.compare(0, 9, "callback:") == 0) {
            // the host-staged transport of the tests and of bench.py's rehearsals: the property carries the address of a function this
            // library will call -- accepted only from a process that says so itself, never from a property string alone
            const char* allow = getenv("OPENMM_HIP_ALLOW_CALLBACK_COMM");
            if (allow == NULL || allow[0] != '1')
                throw OpenMMException("HIP platform: a \"callback:\" CommId (host-staged test transport) needs OPENMM_HIP_ALLOW_CALLBACK_COMM=1 in the environment; production runs pass the RCCL id from ommhip_comm_unique_id()");
            unsigned long long fn = 0, user = 0;
            if (sscanf(commId.c_str() + 9, "%llu:%llu", &fn, &user) < 1 || fn == 0)
                throw OpenMMException("HIP platform: malformed callback CommId");
            HIP_CHECK(ommhip_comm_create_callback((ommhip_host_all_gather_fn) (size_t) fn, (void*) (size_t) user, domain.rank, domain.ranks, &domain.comm));
        }
        else {
            int rc = ommhip_comm_create_rccl(commId.c_str(), domain.rank, domain.ranks, &domain.comm);
            if (rc != 0) {
                stringstream msg;
                msg << "HIP platform: could not create the RCCL communicator (code " << rc << ": " << ommhip_error_string(rc) << "); CommId must be the 256-character hex id from ommhip_comm_unique_id()";
                throw OpenMMException(msg.str());
            }
        }
    }
    PlatformData* data = NULL;
    try {
        data = new PlatformData(context.getSystem(), deviceIndex, mode.hostMode, domain);
        data->group = group;
        data->forcedSeed = forcedSeed;
    } catch (...) {
        delete group;
        // the HipContext did not come to own the communicator (it destroys it otherwise): the peers must not be left waiting in it
        if (domain.comm != NULL) ommhip_comm_destroy(domain.comm);
        throw;
    }
    { stringstream v; v << domain.ranks; data->propertyValues[HipRanks()] = v.str(); }
    { stringstream v; v << domain.rank; data->propertyValues[HipRank()] = v.str(); }
    data->propertyValues[HipCommId()] = domain.comm != NULL ? ommhip_comm_transport(domain.comm) : "";
    data->referenceNonbonded = mode.referenceNonbonded;
    data->hip->hasFallbackForces = mode.hasFallbackForces;
    stringstream dev;
    dev << deviceIndex;
    data->propertyValues[HipDeviceIndex()] = deviceList.empty() ? dev.str() : devicePropValue;
    char name[256];
    name[0] = 0;
    ommhip_device_info(deviceIndex, name, 256, NULL, NULL);
    data->propertyValues[HipDeviceName()] = name;
    data->propertyValues[HipPrecision()] = "mixed";
    data->propertyValues[HipIntegrationMode()] = mode.hostMode ? "host" : (mode.customIntegrator ? "device, custom integrator" : "device");
    data->propertyValues[HipFallbackForces()] = mode.hostMode ? "host mode" : (mode.fallbackNames.empty() ? "none" : mode.fallbackNames);
    data->propertyValues[HipDeterministicForces()] = (properties.find(HipDeterministicForces()) == properties.end() ?
            getPropertyDefaultValue(HipDeterministicForces()) : properties.find(HipDeterministicForces())->second);
    data->propertyValues[HipDisablePmeStream()] = (properties.find(HipDisablePmeStream()) == properties.end() ?
            getPropertyDefaultValue(HipDisablePmeStream()) : properties.find(HipDisablePmeStream())->second);
    data->hip->usePmeStream = data->propertyValues[HipDisablePmeStream()] != "true";
    data->hip->deterministicForces = data->propertyValues[HipDeterministicForces()] == "true";
    // decomposed runs overlap reciprocal space (and its collectives) with the pair kernel unless told otherwise -- and so do
    // single-GPU runs of systems above the size up to which the fused single-stream launches are used (HipKernels.cpp,
    // OPENMM_HIP_FUSED_FRONT_MAX_ATOMS): there the pair kernel is a launch of its own that fills the chip for 0.1-1 ms while
    // the FFT launches need a few hundred workgroups; same-box A/B (profiles/r03f_ab_pme_stream.txt): +2.5 % at 92 k atoms,
    // +2.8 % at 98 k, +2 % at 985 k
    static const int fusedMaxAtoms = getenv("OPENMM_HIP_FUSED_FRONT_MAX_ATOMS") != NULL ? atoi(getenv("OPENMM_HIP_FUSED_FRONT_MAX_ATOMS")) : 60000;
    if (properties.find(HipDisablePmeStream()) == properties.end() &&
            (data->hip->decomposed() || (!mode.hostMode && context.getSystem().getNumParticles() > fusedMaxAtoms))) {
        data->hip->usePmeStream = true;
        data->propertyValues[HipDisablePmeStream()] = "false";
    }
    context.setPlatformData(data);
}

void HipPlatform::contextDestroyed(ContextImpl& context) const {
    PlatformData* data = reinterpret_cast<PlatformData*>(context.getPlatformData());
    delete data;
}

HipPlatform::PlatformData::PlatformData(const System& system, int deviceIndex, bool hostMode, const HipDomain& domain) : ReferencePlatform::PlatformData(system),
        hip(NULL), system(&system), referenceNonbonded(false), integratorSeed(0), deviceConstraints(NULL) {
    hip = new HipContext(system, deviceIndex, hostMode, domain);
}

HipPlatform::PlatformData::~PlatformData() {
    delete group;          // the inner ranks of a device list end with the user's Context
    delete deviceConstraints;
    delete hip;
}

HipConstraints& HipPlatform::PlatformData::getDeviceConstraints(const System& sys) {
    if (deviceConstraints == NULL)
        deviceConstraints = new HipConstraints(sys, *this);
    return *deviceConstraints;
}

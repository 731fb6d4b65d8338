#ifndef OPENMM_HIPPLATFORM_H_
#define OPENMM_HIPPLATFORM_H_
/* The OpenMM "HIP" Platform for AMD Instinct MI355X (gfx950).
 *
 * Registered through olla's plugin interface (olla/include/openmm/PluginInitializer.h:45-57,
 * olla/src/Platform.cpp:237-252): this library exports extern "C" registerPlatforms().
 *
 * HipPlatform subclasses ReferencePlatform (the platforms/cpu precedent, CpuPlatform.h:50) so that
 * every kernel it does not implement natively resolves to the Reference implementation; the
 * per-Context data object subclasses ReferencePlatform::PlatformData so those kernels find their
 * host vectors where they expect them.
 */
#include "ReferencePlatform.h"
#include "openmm/internal/ContextImpl.h"
#include <map>
#include <string>
#include <vector>

namespace OpenMM {

class HipContext;
class HipConstraints;
class HipRankGroup;
struct HipDomain;

class HipPlatform : public ReferencePlatform {
public:
    class PlatformData;
    HipPlatform();
    const std::string& getName() const {
        static const std::string name = "HIP";
        return name;
    }
    double getSpeed() const;
    bool supportsDoublePrecision() const;
    const std::string& getPropertyValue(const Context& context, const std::string& property) const;
    void setPropertyValue(Context& context, const std::string& property, const std::string& value) const;
    void contextCreated(ContextImpl& context, const std::map<std::string, std::string>& properties) const;
    void contextDestroyed(ContextImpl& context) const;
    static const std::string& HipDeviceIndex() { static const std::string key = "DeviceIndex"; return key; }
    static const std::string& HipDeviceName() { static const std::string key = "DeviceName"; return key; }
    static const std::string& HipPrecision() { static const std::string key = "Precision"; return key; }
    static const std::string& HipDeterministicForces() { static const std::string key = "DeterministicForces"; return key; }
    static const std::string& HipDisablePmeStream() { static const std::string key = "DisablePmeStream"; return key; }
    /** Read-only: "device" (native integrator, state in HBM), "device, custom integrator" or "host" (Reference integration kernels; see HipPlatform.cpp). */
    static const std::string& HipIntegrationMode() { static const std::string key = "IntegrationMode"; return key; }
    /** read-only, filled when the Context's constraints are first needed: "settle <clusters> shake <clusters> ccma <constraints>" */
    /** Read-only: the Forces of the System that run as Reference kernels on a host copy of the positions ("none" when every Force has a native kernel). */
    static const std::string& HipFallbackForces() { static const std::string key = "FallbackForces"; return key; }
    static const std::string& HipConstraintPartition() { static const std::string key = "ConstraintPartition"; return key; }
    /** One box on several GPUs, one process per GPU: "Ranks" = number of processes, "Rank" = this one's index, "CommId" = the
     *  ncclUniqueId (hex) created with ommhip_comm_unique_id() on rank 0 and distributed by the launcher, or
     *  "callback:<address of an ommhip_host_all_gather_fn>:<user pointer>" for the host-staged test transport. */
    static const std::string& HipRanks() { static const std::string key = "Ranks"; return key; }
    static const std::string& HipRank() { static const std::string key = "Rank"; return key; }
    static const std::string& HipCommId() { static const std::string key = "CommId"; return key; }
    static PlatformData& getData(ContextImpl& context) {
        return *reinterpret_cast<PlatformData*>(context.getPlatformData());
    }
    /** The ContextImpl behind a Context (Platform::getContextImpl is protected): the inner Contexts of a device list (HipParallel.h). */
    ContextImpl& implOf(Context& context) const { return getContextImpl(context); }
    /** Native kernels that live in a plugin of their own (libOpenMMAmoebaHIP.so, the counterpart of the reference's
     *  libOpenMMAmoebaCUDA): the plugin's registerKernelFactories() announces them here.  `forceType` is a fragment of the C++
     *  type name of the Force the kernel serves ("AmoebaVdwForce"): a System holding such a Force then needs no fallback for it
     *  (the kernel computes on the device state).  The announcement is repeated at every Context creation, because the AMOEBA
     *  plugin's own Reference kernels register themselves with every platform derived from ReferencePlatform
     *  (AmoebaReferenceKernelFactory.cpp:47-58) -- whichever of the two plugins was loaded last would otherwise win. */
    typedef bool (*NativeForceTest)(const Force& force, const System& system);
    /** `supported` (optional): does the native kernel take THIS force object?  (A factory may hand configurations it does not cover
     *  to the Reference kernel -- the Force is then a fallback force like any other.) */
    static void registerNativeKernel(const std::string& kernelName, const std::string& forceType, KernelFactory* factory, NativeForceTest supported = NULL);
    static bool isNativeForce(const Force& force, const System& system);
};

/** ReferencePlatform::PlatformData (host vectors for fallback kernels) + the device context. */
class HipPlatform::PlatformData : public ReferencePlatform::PlatformData {
public:
    PlatformData(const System& system, int deviceIndex, bool hostMode, const HipDomain& domain);
    ~PlatformData();
    HipConstraints& getDeviceConstraints(const System& system);
    HipContext* hip;
    const System* system;
    bool referenceNonbonded;
    unsigned long long integratorSeed;      // resolved seed of the Langevin thermostat noise (part of a checkpoint)
    unsigned long long customDraws = 0;     // random per-DOF computations a device CustomIntegrator has issued so far: the counter its noise is keyed by (part of a checkpoint)
    std::map<std::string, std::string> propertyValues;
    /** Every native kernel of this Context by name, in creation order: how the wrapper kernels of a Context over a device list find
     *  their peers in the inner Contexts (HipParallel.h). */
    std::map<std::string, std::vector<KernelImpl*> > kernelsByName;
    HipRankGroup* group = NULL;             // the user's Context of a device list only: the inner ranks (owned)
    unsigned long long forcedSeed = 0;      // != 0: the thermostat seed every rank of a device list uses (a seed of 0 would be drawn per rank)
private:
    HipConstraints* deviceConstraints;
};

}  // namespace OpenMM
#endif

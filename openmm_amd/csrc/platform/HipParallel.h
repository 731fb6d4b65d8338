#ifndef OPENMM_HIPPARALLEL_H_
#define OPENMM_HIPPARALLEL_H_
/* One Context over a list of devices: `DeviceIndex = "0,1,2,3"` -- the reference's own multi-device interface
 * (examples/benchmark.py:146-150; platforms/cuda/src/CudaParallelKernels.cpp:177-254; test platforms/cuda/tests/TestCudaNonbondedForce.cpp:37-96).
 *
 * The reference's GPU platforms replicate all atoms on every device, split the pairs, and add the forces up through device 0.  This platform
 * decomposes the BOX: every device of the list becomes one rank of the slab decomposition that `Ranks` / `Rank` / `CommId`
 * give a multi-process run.  The user's Context is rank 0.  For every further device the platform creates an inner Context of the same
 * System (a copy of the integrator, `Ranks` / `Rank` / `CommId` set) on a host thread of its own -- the `CudaContext::WorkThread`
 * precedent -- and the kernels the user's Context gets are thin wrappers: each call runs on rank 0 and is posted, in order, to the
 * peer kernel of every inner Context.  The ranks meet in the collectives of the decomposed step exactly as processes do.
 *
 * Transport: devices that are all different -> RCCL (an ncclUniqueId made in the process); a device named twice ("0,0": the reference's
 * own way of testing the path on one GPU) or OPENMM_HIP_INPROCESS_TRANSPORT=staged -> a host-staged all-gather between the threads. */
#include "HipPlatform.h"
#include "openmm/Context.h"
#include "openmm/Integrator.h"
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace OpenMM {

class HipRankGroup {
public:
    /** devices[0] is the user's Context (rank 0); spawns one thread per further device, each creating its inner Context at once (the
     *  communicator's creation is a collective: the caller must go on to create rank 0's). */
    HipRankGroup(const HipPlatform& platform, ContextImpl& primary, const std::vector<int>& devices, const std::map<std::string, std::string>& properties);
    ~HipRankGroup();
    int size() const { return (int) devices.size(); }
    const std::string& commId() const { return commIdValue; }
    unsigned long long seed() const { return sharedSeed; }
    /** Posts task(r) to every inner rank r = 1 .. size-1 (each runs it on its own thread, in posting order); does not wait. */
    void post(const std::function<void(int)>& task);
    /** Waits until every inner rank has finished what was posted; rethrows the first exception one of them met. */
    void join();
    /** The same wait; returns the first inner rank's error message ("" if none) instead of throwing. */
    std::string firstError();
    /** A rank has failed (its thread calls this; the wrapper kernels do for rank 0): ranks waiting for it in a collective of the host-staged
     *  transport return with an error instead of waiting for ever.  The Context is lost; its destruction no longer hangs. */
    void abortCollectives(int rank, const std::string& why);
    /** The rank whose failure was reported first (-1: none) and its message. */
    int failedFirst() { std::lock_guard<std::mutex> lock(causeMutex); return causeRank; }
    std::string firstCause() { std::lock_guard<std::mutex> lock(causeMutex); return cause; }
    ContextImpl& impl(int rank);
    Integrator& integrator(int rank);
    /** The ordinal-th kernel named `name` of an inner Context (kernels are created in the same order in every Context of one System). */
    KernelImpl* peer(int rank, const std::string& name, int ordinal);
private:
    struct Worker {
        std::thread thread;
        std::mutex mutex;
        std::condition_variable wake, idle;
        std::deque<std::function<void()> > queue;
        bool busy = false, quit = false;
        std::string error;
        Context* context = NULL;
        Integrator* integrator = NULL;
    };
    void workerMain(int rank, const System* system, std::map<std::string, std::string> props);
    const HipPlatform& platform;
    std::vector<int> devices;
    std::vector<std::unique_ptr<Worker> > workers;          // [0] unused
    std::string commIdValue;
    unsigned long long sharedSeed;
    int stagedToken;
    std::mutex causeMutex;
    int causeRank = -1;
    std::string cause;
};

/** True on a thread of a HipRankGroup while it creates its inner Context (what the platform does once per user Context is then left out). */
bool& hipCreatingInnerRank();

/** Host-staged all-gather between the threads of one process ("inprocess:<token>" CommId): registry of the groups alive. */
int hipInProcessCreate(int ranks);                                    // -> token
void hipInProcessDestroy(int token);
void hipInProcessAbort(int token);                                     // every present and future wait of the group ends with an error
/** The callback transport's function and the `user` pointer of one rank of a group (valid until hipInProcessDestroy). */
void* hipInProcessUser(int token, int rank);
extern "C" int hipInProcessAllGather(void* user, const void* send, void* recv, size_t bytes);

/** A kernel of the user's Context when it drives a device list, wrapping rank 0's own kernel (created by `make`); NULL: not a wrapped kind. */
KernelImpl* hipMakeParallelKernel(const std::string& name, const Platform& platform, HipPlatform::PlatformData& data, KernelImpl* own);

}  // namespace OpenMM
#endif

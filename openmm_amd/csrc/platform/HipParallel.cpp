/* One Context over a list of devices (HipParallel.h): the rank group with its worker threads, the host-staged all-gather between the threads
 * of one process, and the wrapper kernels of the user's Context. */
#include "HipParallel.h"
#include "HipContext.h"
#include "HipKernels.h"
#include "openmm/LangevinIntegrator.h"
#include "openmm/LangevinMiddleIntegrator.h"
#include "openmm/VerletIntegrator.h"
#include "openmm/kernels.h"
#include <cstdlib>
#include <cstring>
#include <random>
#include <sstream>

using namespace OpenMM;
using namespace std;

// ================================================================================================
// host-staged all-gather between threads
// ================================================================================================
namespace {
struct StagedGroup;
struct StagedRank { StagedGroup* group; int rank; };
struct StagedGroup {
    int ranks;
    std::mutex mutex;
    std::condition_variable cv;
    int arrived = 0, copied = 0;
    long long generationIn = 0, generationOut = 0;
    bool broken = false;               // a rank failed: nobody waits for it any more (hipInProcessAbort)
    vector<const void*> send;
    vector<StagedRank> users;
};
std::mutex registryMutex;
map<int, StagedGroup*> registry;
int nextToken = 1;
}  // namespace

int OpenMM::hipInProcessCreate(int ranks) {
    std::lock_guard<std::mutex> lock(registryMutex);
    StagedGroup* g = new StagedGroup();
    g->ranks = ranks;
    g->send.assign(ranks, NULL);
    g->users.resize(ranks);
    for (int r = 0; r < ranks; r++) { g->users[r].group = g; g->users[r].rank = r; }
    registry[nextToken] = g;
    return nextToken++;
}

void OpenMM::hipInProcessDestroy(int token) {
    std::lock_guard<std::mutex> lock(registryMutex);
    map<int, StagedGroup*>::iterator found = registry.find(token);
    if (found == registry.end()) return;
    delete found->second;
    registry.erase(found);
}

void OpenMM::hipInProcessAbort(int token) {
    std::lock_guard<std::mutex> lock(registryMutex);
    map<int, StagedGroup*>::iterator found = registry.find(token);
    if (found == registry.end()) return;
    { std::lock_guard<std::mutex> inner(found->second->mutex); found->second->broken = true; }
    found->second->cv.notify_all();
}

void* OpenMM::hipInProcessUser(int token, int rank) {
    std::lock_guard<std::mutex> lock(registryMutex);
    map<int, StagedGroup*>::iterator found = registry.find(token);
    if (found == registry.end() || rank < 0 || rank >= found->second->ranks) return NULL;
    return &found->second->users[rank];
}

extern "C" int OpenMM::hipInProcessAllGather(void* user, const void* send, void* recv, size_t bytes) {
    // every rank's thread arrives with its piece; when all are there each copies all pieces; nobody leaves (and lets its send buffer go)
    // before all have copied.  The ranks issue their collectives in the same order, one at a time per rank.
    StagedRank* me = (StagedRank*) user;
    if (me == NULL) return 1;
    StagedGroup& g = *me->group;
    // A rank that failed never arrives: whoever learns of it (hipInProcessAbort) marks the group broken, and every rank waiting here -- now or
    // later -- returns an error instead (its own call then fails the way a failed collective does; the Context is lost either way).
    {
        std::unique_lock<std::mutex> lock(g.mutex);
        if (g.broken) return 1;
        g.send[me->rank] = send;
        const long long gen = g.generationIn;
        if (++g.arrived == g.ranks) { g.arrived = 0; g.generationIn++; g.cv.notify_all(); }
        else g.cv.wait(lock, [&] { return g.generationIn != gen || g.broken; });
        if (g.generationIn == gen) return 1;          // woken by the abort: the pieces are not all there
    }
    for (int r = 0; r < g.ranks; r++) memcpy((char*) recv + (size_t) r * bytes, g.send[r], bytes);
    {
        std::unique_lock<std::mutex> lock(g.mutex);
        const long long gen = g.generationOut;
        if (++g.copied == g.ranks) { g.copied = 0; g.generationOut++; g.cv.notify_all(); }
        else g.cv.wait(lock, [&] { return g.generationOut != gen || g.broken; });
        if (g.generationOut == gen) return 1;         // (the others' copies may be unfinished: they fail as well)
    }
    return 0;
}

bool& OpenMM::hipCreatingInnerRank() {
    static thread_local bool creating = false;
    return creating;
}

// ================================================================================================
// the rank group
// ================================================================================================
static Integrator* cloneIntegrator(const Integrator& in, unsigned long long seed) {
    Integrator* out = NULL;
    if (const VerletIntegrator* v = dynamic_cast<const VerletIntegrator*>(&in)) out = new VerletIntegrator(v->getStepSize());
    else if (const LangevinMiddleIntegrator* l = dynamic_cast<const LangevinMiddleIntegrator*>(&in)) {
        LangevinMiddleIntegrator* c = new LangevinMiddleIntegrator(l->getTemperature(), l->getFriction(), l->getStepSize());
        c->setRandomNumberSeed((int) seed);
        out = c;
    }
    else if (const LangevinIntegrator* l = dynamic_cast<const LangevinIntegrator*>(&in)) {
        LangevinIntegrator* c = new LangevinIntegrator(l->getTemperature(), l->getFriction(), l->getStepSize());
        c->setRandomNumberSeed((int) seed);
        out = c;
    }
    else throw OpenMMException("HIP platform: a Context over a list of devices supports the Verlet, Langevin and LangevinMiddle integrators");
    out->setConstraintTolerance(in.getConstraintTolerance());
    out->setIntegrationForceGroups(in.getIntegrationForceGroups());
    return out;
}

HipRankGroup::HipRankGroup(const HipPlatform& platform, ContextImpl& primary, const vector<int>& devices, const map<string, string>& properties) : platform(platform), devices(devices), sharedSeed(0), stagedToken(0) {
    const int ranks = (int) devices.size();
    // the thermostat noise of every rank is keyed by ONE seed (each rank draws for the atoms it integrates): the user's, or one picked here
    const Integrator& integ = primary.getIntegrator();
    int userSeed = 0;
    if (const LangevinMiddleIntegrator* l = dynamic_cast<const LangevinMiddleIntegrator*>(&integ)) userSeed = l->getRandomNumberSeed();
    else if (const LangevinIntegrator* l = dynamic_cast<const LangevinIntegrator*>(&integ)) userSeed = l->getRandomNumberSeed();
    sharedSeed = (unsigned long long) (unsigned int) userSeed;
    if (sharedSeed == 0) { std::random_device rd; sharedSeed = (rd() & 0x3fffffffu) | 1u; }
    // transport: RCCL between different devices; a device named twice (or the knob) -> the host-staged all-gather between the threads
    bool repeated = false;
    for (int i = 0; i < ranks; i++)
        for (int j = 0; j < i; j++) repeated = repeated || devices[i] == devices[j];
    const char* forced = getenv("OPENMM_HIP_INPROCESS_TRANSPORT");
    bool staged = repeated || (forced != NULL && string(forced) == "staged");
    if (!staged) {
        char hex[OMMHIP_COMM_ID_HEX_LEN];
        if (ommhip_comm_unique_id(hex) == 0) commIdValue = hex;
        else staged = true;          // no RCCL in this build (the CPU emulator): threads can still meet through the host
    }
    if (staged) {
        stagedToken = hipInProcessCreate(ranks);
        stringstream id;
        id << "inprocess:" << stagedToken;
        commIdValue = id.str();
    }
    workers.resize(ranks);
    for (int r = 1; r < ranks; r++) {
        map<string, string> props;
        for (map<string, string>::const_iterator it = properties.begin(); it != properties.end(); ++it)
            if (it->first != HipPlatform::HipDeviceIndex()) props[it->first] = it->second;
        { stringstream v; v << devices[r]; props[HipPlatform::HipDeviceIndex()] = v.str(); }
        { stringstream v; v << ranks; props[HipPlatform::HipRanks()] = v.str(); }
        { stringstream v; v << r; props[HipPlatform::HipRank()] = v.str(); }
        props[HipPlatform::HipCommId()] = commIdValue;
        workers[r].reset(new Worker());
        workers[r]->integrator = cloneIntegrator(integ, sharedSeed);
        workers[r]->busy = true;          // until its Context stands
        workers[r]->thread = std::thread(&HipRankGroup::workerMain, this, r, &primary.getSystem(), props);
    }
}

void HipRankGroup::workerMain(int rank, const System* system, map<string, string> props) {
    Worker& w = *workers[rank];
    try {
        hipCreatingInnerRank() = true;
        w.context = new Context(*system, *w.integrator, const_cast<HipPlatform&>(platform), props);
        hipCreatingInnerRank() = false;
    } catch (const std::exception& e) {
        hipCreatingInnerRank() = false;
        std::lock_guard<std::mutex> lock(w.mutex);
        w.error = string("creating the Context of device-list rank ") + to_string(rank) + ": " + e.what();
        abortCollectives(rank, w.error);
    }
    while (true) {
        std::function<void()> task;
        {
            std::unique_lock<std::mutex> lock(w.mutex);
            w.busy = false;
            if (w.queue.empty()) w.idle.notify_all();
            w.wake.wait(lock, [&] { return w.quit || !w.queue.empty(); });
            if (w.queue.empty()) break;          // quit
            task = w.queue.front();
            w.queue.pop_front();
            w.busy = true;
            if (!w.error.empty()) continue;      // after a failure nothing more is attempted: join() reports it
        }
        try { task(); }
        catch (const std::exception& e) {
            std::lock_guard<std::mutex> lock(w.mutex);
            if (w.error.empty()) w.error = string("device-list rank ") + to_string(rank) + ": " + e.what();
            abortCollectives(rank, w.error);          // the other ranks must not wait for this one in a collective it will never reach
        }
    }
    delete w.context;
    w.context = NULL;
}

void HipRankGroup::abortCollectives(int rank, const std::string& why) {
    {
        std::lock_guard<std::mutex> lock(causeMutex);
        if (causeRank < 0) { causeRank = rank; cause = why; }
    }
    if (stagedToken != 0) hipInProcessAbort(stagedToken);
    // (RCCL: a rank stuck in a device-side collective cannot be released from here; the process has to end, as with one process per GPU)
}

HipRankGroup::~HipRankGroup() {
    for (size_t r = 1; r < workers.size(); r++) {
        { std::lock_guard<std::mutex> lock(workers[r]->mutex); workers[r]->quit = true; }
        workers[r]->wake.notify_all();
    }
    for (size_t r = 1; r < workers.size(); r++) {
        if (workers[r]->thread.joinable()) workers[r]->thread.join();
        delete workers[r]->integrator;
    }
    if (stagedToken != 0) hipInProcessDestroy(stagedToken);
}

void HipRankGroup::post(const std::function<void(int)>& task) {
    for (size_t r = 1; r < workers.size(); r++) {
        Worker& w = *workers[r];
        const int rank = (int) r;
        std::unique_lock<std::mutex> lock(w.mutex);
        // (a bounded run-ahead: the host of rank 0 only enqueues GPU work and could get hundreds of steps ahead of the other threads)
        if (w.queue.size() > 512) w.idle.wait(lock, [&] { return w.queue.size() < 64; });
        w.queue.push_back([task, rank]() { task(rank); });
        w.wake.notify_all();
    }
}

std::string HipRankGroup::firstError() {
    string error;
    for (size_t r = 1; r < workers.size(); r++) {
        Worker& w = *workers[r];
        std::unique_lock<std::mutex> lock(w.mutex);
        w.idle.wait(lock, [&] { return w.queue.empty() && !w.busy; });
        if (error.empty() && !w.error.empty()) error = w.error;
    }
    return error;
}

void HipRankGroup::join() {
    const string error = firstError();
    if (!error.empty()) throw OpenMMException("HIP platform: " + error);
}

ContextImpl& HipRankGroup::impl(int rank) {
    Context* c = workers[rank]->context;
    if (c == NULL) throw OpenMMException("HIP platform: the inner Context of a device-list rank does not exist");
    return platform.implOf(*c);
}

Integrator& HipRankGroup::integrator(int rank) { return *workers[rank]->integrator; }

KernelImpl* HipRankGroup::peer(int rank, const string& name, int ordinal) {
    HipPlatform::PlatformData& d = HipPlatform::getData(impl(rank));
    map<string, vector<KernelImpl*> >::iterator found = d.kernelsByName.find(name);
    if (found == d.kernelsByName.end() || ordinal < 0 || ordinal >= (int) found->second.size())
        throw OpenMMException("HIP platform: no peer for kernel " + name + " on a device-list rank");
    return found->second[ordinal];
}

// ================================================================================================
// wrapper kernels of the user's Context
// ================================================================================================
namespace {

template <class K>
class Par : public K {
public:
    Par(const string& name, const Platform& platform, HipPlatform::PlatformData& data, KernelImpl* ownImpl) : K(name, platform), data(data), own(dynamic_cast<K*>(ownImpl)) {
        ordinal = (int) data.kernelsByName[name].size() - 1;          // `own` was registered just before
        if (own == NULL) throw OpenMMException("HIP platform: internal error: kernel " + name + " has an unexpected type");
    }
    ~Par() {
        // tasks still queued on the inner ranks' threads refer to this object
        try { if (data.group != NULL) data.group->join(); } catch (...) {}
        delete own;
    }
protected:
    HipRankGroup& group() const { return *data.group; }
    /** Rank 0's own share of a call.  If it throws, the inner ranks are released from the collectives in which they would wait for it. */
    template <class F>
    auto guard(F f) const -> decltype(f()) {
        try { return f(); }
        catch (const std::exception& e) {
            group().abortCollectives(0, e.what());
            // was it an inner rank that failed first (rank 0 then only saw a collective return an error)?  Its message is the one to report
            group().firstError();          // (waits until the inner ranks have run out of the collectives)
            if (group().failedFirst() > 0) throw OpenMMException("HIP platform: " + group().firstCause());
            throw;
        }
    }
    K& peer(int rank) const {
        K* k = dynamic_cast<K*>(group().peer(rank, this->getName(), ordinal));
        if (k == NULL) throw OpenMMException("HIP platform: internal error: peer kernel of an unexpected type");
        return *k;
    }
    HipPlatform::PlatformData& data;
    K* own;
    int ordinal;
};

class ParCalcForcesAndEnergy : public Par<CalcForcesAndEnergyKernel> {
public:
    using Par::Par;
    void initialize(const System& system) { own->initialize(system); }
    void beginComputation(ContextImpl& context, bool includeForce, bool includeEnergy, int groups) {
        group().post([=](int r) { peer(r).beginComputation(group().impl(r), includeForce, includeEnergy, groups); });
        guard([&] { own->beginComputation(context, includeForce, includeEnergy, groups); });
    }
    double finishComputation(ContextImpl& context, bool includeForce, bool includeEnergy, int groups, bool& valid) {
        group().post([=](int r) { bool v = true; peer(r).finishComputation(group().impl(r), includeForce, includeEnergy, groups, v); });
        const double energy = guard([&] { return own->finishComputation(context, includeForce, includeEnergy, groups, valid); });
        if (includeEnergy) group().join();          // an energy leaves the platform: whatever went wrong on another rank is reported with it
        return energy;
    }
};

class ParUpdateStateData : public Par<UpdateStateDataKernel> {
public:
    using Par::Par;
    void initialize(const System& system) { own->initialize(system); }
    double getTime(const ContextImpl& context) const { return own->getTime(context); }
    void setTime(ContextImpl& context, double time) {
        group().post([=](int r) { peer(r).setTime(group().impl(r), time); });
        own->setTime(context, time);
    }
    // the downloads of a decomposed run are collectives (every rank gathers the exact state): the inner ranks take part and drop the copy
    void getPositions(ContextImpl& context, vector<Vec3>& positions) {
        group().post([=](int r) { vector<Vec3> tmp; peer(r).getPositions(group().impl(r), tmp); });
        guard([&] { own->getPositions(context, positions); });
        group().join();
    }
    void setPositions(ContextImpl& context, const vector<Vec3>& positions) {
        std::shared_ptr<vector<Vec3> > copy(new vector<Vec3>(positions));
        group().post([=](int r) { peer(r).setPositions(group().impl(r), *copy); });
        guard([&] { own->setPositions(context, positions); });
    }
    void getVelocities(ContextImpl& context, vector<Vec3>& velocities) {
        group().post([=](int r) { vector<Vec3> tmp; peer(r).getVelocities(group().impl(r), tmp); });
        guard([&] { own->getVelocities(context, velocities); });
        group().join();
    }
    void setVelocities(ContextImpl& context, const vector<Vec3>& velocities) {
        std::shared_ptr<vector<Vec3> > copy(new vector<Vec3>(velocities));
        group().post([=](int r) { peer(r).setVelocities(group().impl(r), *copy); });
        guard([&] { own->setVelocities(context, velocities); });
    }
    void getForces(ContextImpl& context, vector<Vec3>& forces) {
        group().post([=](int r) { vector<Vec3> tmp; peer(r).getForces(group().impl(r), tmp); });
        guard([&] { own->getForces(context, forces); });
        group().join();
    }
    void getEnergyParameterDerivatives(ContextImpl& context, map<string, double>& derivs) { own->getEnergyParameterDerivatives(context, derivs); }
    void getPeriodicBoxVectors(ContextImpl& context, Vec3& a, Vec3& b, Vec3& c) const { own->getPeriodicBoxVectors(context, a, b, c); }
    void setPeriodicBoxVectors(ContextImpl& context, const Vec3& a, const Vec3& b, const Vec3& c) {
        group().post([=](int r) { peer(r).setPeriodicBoxVectors(group().impl(r), a, b, c); });
        guard([&] { own->setPeriodicBoxVectors(context, a, b, c); });
    }
    void createCheckpoint(ContextImpl& context, ostream& stream) {
        group().post([=](int r) { stringstream drop; peer(r).createCheckpoint(group().impl(r), drop); });
        guard([&] { own->createCheckpoint(context, stream); });
        group().join();
    }
    void loadCheckpoint(ContextImpl& context, istream& stream) {
        // rank 0 reads its part of the stream; the same bytes are then handed to the inner ranks
        const std::streampos before = stream.tellg();
        guard([&] { own->loadCheckpoint(context, stream); });
        const std::streampos after = stream.tellg();
        if (before == std::streampos(-1) || after == std::streampos(-1))
            throw OpenMMException("HIP platform: loading a checkpoint into a Context over a list of devices needs a seekable stream");
        std::shared_ptr<string> bytes(new string((size_t) (after - before), '\0'));
        stream.seekg(before);
        stream.read(&(*bytes)[0], (std::streamsize) bytes->size());
        group().post([=](int r) { stringstream s(*bytes); peer(r).loadCheckpoint(group().impl(r), s); });
        group().join();
    }
};

class ParApplyConstraints : public Par<ApplyConstraintsKernel> {
public:
    using Par::Par;
    void initialize(const System& system) { own->initialize(system); }
    void apply(ContextImpl& context, double tol) {
        group().post([=](int r) { peer(r).apply(group().impl(r), tol); });
        guard([&] { own->apply(context, tol); });
    }
    void applyToVelocities(ContextImpl& context, double tol) {
        group().post([=](int r) { peer(r).applyToVelocities(group().impl(r), tol); });
        guard([&] { own->applyToVelocities(context, tol); });
    }
};

class ParCalcNonbondedForce : public Par<CalcNonbondedForceKernel> {
public:
    using Par::Par;
    void initialize(const System& system, const NonbondedForce& force) { own->initialize(system, force); }
    double execute(ContextImpl& context, bool includeForces, bool includeEnergy, bool includeDirect, bool includeReciprocal) {
        group().post([=](int r) { peer(r).execute(group().impl(r), includeForces, includeEnergy, includeDirect, includeReciprocal); });
        return guard([&] { return own->execute(context, includeForces, includeEnergy, includeDirect, includeReciprocal); });
    }
    void copyParametersToContext(ContextImpl& context, const NonbondedForce& force) {
        const NonbondedForce* f = &force;
        group().post([=](int r) { peer(r).copyParametersToContext(group().impl(r), *f); });
        guard([&] { own->copyParametersToContext(context, force); });
        group().join();          // the Force object is the caller's
    }
    void getPMEParameters(double& alpha, int& nx, int& ny, int& nz) const { own->getPMEParameters(alpha, nx, ny, nz); }
    void getLJPMEParameters(double& alpha, int& nx, int& ny, int& nz) const { own->getLJPMEParameters(alpha, nx, ny, nz); }
};

template <class K, class F>
class ParTermForce : public Par<K> {
public:
    using Par<K>::Par;
    void initialize(const System& system, const F& force) { this->own->initialize(system, force); }
    double execute(ContextImpl& context, bool includeForces, bool includeEnergy) {
        this->group().post([=](int r) { this->peer(r).execute(this->group().impl(r), includeForces, includeEnergy); });
        return this->guard([&] { return this->own->execute(context, includeForces, includeEnergy); });
    }
    void copyParametersToContext(ContextImpl& context, const F& force) {
        const F* f = &force;
        this->group().post([=](int r) { this->peer(r).copyParametersToContext(this->group().impl(r), *f); });
        this->guard([&] { this->own->copyParametersToContext(context, force); });
        this->group().join();
    }
};

/* The inner copies of the integrator take the user's current settings with every step (they travel inside the task). */
template <class K, class I>
class ParIntegrate : public Par<K> {
public:
    using Par<K>::Par;
    void initialize(const System& system, const I& integrator) { this->own->initialize(system, integrator); }
    void execute(ContextImpl& context, const I& integrator) {
        const double dt = integrator.getStepSize(), tol = integrator.getConstraintTolerance();
        double temperature = 0, friction = 0;
        settings(integrator, temperature, friction);
        this->group().post([=](int r) {
            I& mine = dynamic_cast<I&>(this->group().integrator(r));
            mine.setStepSize(dt); mine.setConstraintTolerance(tol);
            apply(mine, temperature, friction);
            this->peer(r).execute(this->group().impl(r), mine);
        });
        this->guard([&] { this->own->execute(context, integrator); });
        if ((++steps & 31) == 0) this->group().join();          // errors of the other ranks surface within a few steps
    }
    double computeKineticEnergy(ContextImpl& context, const I& integrator) {
        this->group().post([=](int r) { this->peer(r).computeKineticEnergy(this->group().impl(r), dynamic_cast<I&>(this->group().integrator(r))); });
        const double ke = this->guard([&] { return this->own->computeKineticEnergy(context, integrator); });
        this->group().join();
        return ke;
    }
private:
    static void settings(const VerletIntegrator&, double&, double&) {}
    static void settings(const LangevinIntegrator& i, double& t, double& f) { t = i.getTemperature(); f = i.getFriction(); }
    static void settings(const LangevinMiddleIntegrator& i, double& t, double& f) { t = i.getTemperature(); f = i.getFriction(); }
    static void apply(VerletIntegrator&, double, double) {}
    static void apply(LangevinIntegrator& i, double t, double f) { i.setTemperature(t); i.setFriction(f); }
    static void apply(LangevinMiddleIntegrator& i, double t, double f) { i.setTemperature(t); i.setFriction(f); }
    long long steps = 0;
};

class ParRemoveCMMotion : public Par<RemoveCMMotionKernel> {
public:
    using Par::Par;
    void initialize(const System& system, const CMMotionRemover& force) { own->initialize(system, force); }
    void execute(ContextImpl& context) {
        group().post([=](int r) { peer(r).execute(group().impl(r)); });
        guard([&] { own->execute(context); });
    }
};

class ParApplyMonteCarloBarostat : public Par<ApplyMonteCarloBarostatKernel> {
public:
    using Par::Par;
    void initialize(const System& system, const Force& barostat) { own->initialize(system, barostat); }
    void scaleCoordinates(ContextImpl& context, double scaleX, double scaleY, double scaleZ) {
        group().post([=](int r) { peer(r).scaleCoordinates(group().impl(r), scaleX, scaleY, scaleZ); });
        guard([&] { own->scaleCoordinates(context, scaleX, scaleY, scaleZ); });
    }
    void restoreCoordinates(ContextImpl& context) {
        group().post([=](int r) { peer(r).restoreCoordinates(group().impl(r)); });
        guard([&] { own->restoreCoordinates(context); });
    }
};

}  // namespace

KernelImpl* OpenMM::hipMakeParallelKernel(const string& name, const Platform& platform, HipPlatform::PlatformData& data, KernelImpl* own) {
    if (name == CalcForcesAndEnergyKernel::Name()) return new ParCalcForcesAndEnergy(name, platform, data, own);
    if (name == UpdateStateDataKernel::Name()) return new ParUpdateStateData(name, platform, data, own);
    if (name == ApplyConstraintsKernel::Name()) return new ParApplyConstraints(name, platform, data, own);
    if (name == CalcNonbondedForceKernel::Name()) return new ParCalcNonbondedForce(name, platform, data, own);
    if (name == CalcHarmonicBondForceKernel::Name()) return new ParTermForce<CalcHarmonicBondForceKernel, HarmonicBondForce>(name, platform, data, own);
    if (name == CalcHarmonicAngleForceKernel::Name()) return new ParTermForce<CalcHarmonicAngleForceKernel, HarmonicAngleForce>(name, platform, data, own);
    if (name == CalcPeriodicTorsionForceKernel::Name()) return new ParTermForce<CalcPeriodicTorsionForceKernel, PeriodicTorsionForce>(name, platform, data, own);
    if (name == IntegrateVerletStepKernel::Name()) return new ParIntegrate<IntegrateVerletStepKernel, VerletIntegrator>(name, platform, data, own);
    if (name == IntegrateLangevinStepKernel::Name()) return new ParIntegrate<IntegrateLangevinStepKernel, LangevinIntegrator>(name, platform, data, own);
    if (name == IntegrateLangevinMiddleStepKernel::Name()) return new ParIntegrate<IntegrateLangevinMiddleStepKernel, LangevinMiddleIntegrator>(name, platform, data, own);
    if (name == RemoveCMMotionKernel::Name()) return new ParRemoveCMMotion(name, platform, data, own);
    if (name == ApplyMonteCarloBarostatKernel::Name()) return new ParApplyMonteCarloBarostat(name, platform, data, own);
    return NULL;
}

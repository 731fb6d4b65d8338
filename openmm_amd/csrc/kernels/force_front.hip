// One translation unit for the three kernels that open a force evaluation, and the launch that runs them side by side.
//
// After nl_prepare three pieces of work are independent of each other: the (possible) neighbour-list rebuild, the PME
// charge spreading and the per-term forces (bonds, angles, torsions, 1-4s).  Each is latency-bound on its own and uses
// a fraction of the chip; as separate launches they cost 1-53 + 16 + 8 us back to back, and running them on separate
// streams costs ~13 us per cross-stream dependency on this stack.  force_front runs them as three groups of workgroups
// of ONE launch (same workgroup size, LDS aliased through a union), so they overlap without any synchronisation object.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "neighbor.hip"
#include "pme.hip"
#include "bonded.hip"
#include "nonbonded.hip"
#ifndef OMMHIP_EMU
#include <hip/hip_ext.h>          // hipExtLaunchKernelGGL: events stamped by a kernel's own dispatch (timed launches)
#endif

namespace {

struct FrontArgs {
    NlArgs nl;
    PmeArgs pme;
    TermArgs terms;
    int nlBlocks, spreadBlocks, termBlocks;
};

union FrontShared {
    NlShared nl;
    SpreadShared spread;
    double termPartial[4];
};

template <int PBC>
__global__ __launch_bounds__(256) void force_front(FrontArgs f) {
    __shared__ FrontShared sh;
    const int b = blockIdx.x;
#ifndef OMM_FRONT_ORDER
#define OMM_FRONT_ORDER 0
#endif
#if OMM_FRONT_ORDER == 0
    // heavy, rare work first in the grid so that it starts first
    if (b < f.nlBlocks) nl_find_body<PBC>(f.nl, f.nl.firstBlock + b, f.nlBlocks, sh.nl);
    else if (b < f.nlBlocks + f.spreadBlocks) pme_spread_body<false>(f.pme, b - f.nlBlocks, sh.spread);
    else terms_body(f.terms, b - f.nlBlocks - f.spreadBlocks, sh.termPartial);
#else
    // the work of every step first: the builder's workgroups, which leave at once on all steps but one in thirty, are dispatched behind it
    if (b < f.termBlocks) terms_body(f.terms, b, sh.termPartial);
    else if (b < f.termBlocks + f.spreadBlocks) pme_spread_body<false>(f.pme, b - f.termBlocks, sh.spread);
    else nl_find_body<PBC>(f.nl, f.nl.firstBlock + b - f.termBlocks - f.spreadBlocks, f.nlBlocks, sh.nl);
#endif
}

}  // namespace

extern "C" int ommhip_force_front(const ommhip_neighbor_list* nl, const ommhip_pme* pme, int num_lists, const ommhip_term_batch* lists,
                                  const void* pos_d, long long* force_d, double* energy_buffer_d, int energy_slots, int include_energy, void* stream) {
    static_assert(NL_THREADS == 256, "the fused launch uses 256-thread workgroups for all three kinds of work");
    hipStream_t st = (hipStream_t) stream;
    FrontArgs f;
    f.nl = make_nl_args(nl);
    if (f.nl.cellMode) hipLaunchKernelGGL(nl_bin_blocks, dim3(1), dim3(1024), 0, st, f.nl);
    f.nlBlocks = f.nl.ownedBlocks;
    f.spreadBlocks = 0;
    if (pme != nullptr) {
        if (pme->spread_mode == 1 || !pme->grid_precleared || pme->deterministic) return 1;          // the direct-atomics variant, un-cleared grids and fixed-point grids are not fused
        f.pme = make_pme_args(pme, nl->posq, nl->padded_atoms, force_d, energy_buffer_d, energy_slots, include_energy);
        f.spreadBlocks = (nl->padded_atoms + SPREAD_ATOMS - 1) / SPREAD_ATOMS;
    }
    else
        std::memset(&f.pme, 0, sizeof(f.pme));
    f.termBlocks = make_term_args(f.terms, num_lists, lists, pos_d, nl->slot_of_atom, nl->padded_atoms, nl->box, force_d, energy_buffer_d, energy_slots, include_energy);
    if (f.termBlocks < 0) return 1;
    const dim3 grid(f.nlBlocks + f.spreadBlocks + f.termBlocks);
    ommhip_profile_begin(OMMHIP_TIMER_NL_UPDATE, stream);
    if (nl->pbc == 0) hipLaunchKernelGGL(force_front<0>, grid, dim3(256), 0, st, f);
    else if (nl->pbc == 1) hipLaunchKernelGGL(force_front<1>, grid, dim3(256), 0, st, f);
    else hipLaunchKernelGGL(force_front<2>, grid, dim3(256), 0, st, f);
    launch_prune(f.nl, st);
    ommhip_profile_end(OMMHIP_TIMER_NL_UPDATE, stream);
    return (int) hipGetLastError();
}

// ================================================================================================
// Pair kernel + reciprocal-space FFTs, three launches: stage 0 = forward plane transforms, stage 1 = x transform with
// the convolution, stage 2 = backward plane transforms.  The FFT stages are latency-bound and need only 56-112
// workgroups; each launch is [FFT workgroups | pair workgroups], the pair workgroups of stage s working on the part
// [split[s], split[s+1]) / 64 of the chunk list, one chunk per wavefront.
//
// Tried and not kept (DHFR-sized system, pair kernel 37 us + FFT chain 31 us as separate launches):
//  * pair wavefronts drawing chunks from device-wide ticket queues until the FFT workgroups of the launch signal
//    completion: every draw is a device-scope atomic round trip (1-2 us per 18 us chunk; same-address atomics also
//    serialise at ~8 ns each): 88 us;
//  * ONE launch in which the FFT workgroups walk through the three stages with a device-wide barrier of their own
//    while the pair workgroups stream through the rest of the chip: the XCDs' L2s are not coherent with each other
//    inside a kernel, so the grid has to be handed from stage to stage through memory.  With __threadfence() (L2
//    write-back + invalidate, ~10 us per barrier, and it evicts the pair code's working set): 170 us; with device-scope
//    atomic loads/stores of the grid elements (serialised by the compiler, one round trip each): 72 us; with the complex
//    grid in uncached memory: 67 us (FFT chain alone 60 us) -- no better than separate launches.
//  * single rows instead of two-row chunks as the pair work unit (shorter unit latency, but the i-block set-up and force
//    reduction twice per chunk): 63 us against 59 us.
// A launch with a share of the pair work takes ~20 us whatever the share (thirds, or one chunk per SIMD): the pair code
// is bound by the latency of a chunk, not by throughput, at this size.  FFT workgroups alone: 13 us per launch.
// Passing the pair kernel's arguments as a kernel parameter of their own matters: as a member of one big argument
// struct whose other members are passed on by reference they end up in scratch, the loads turn into flat loads, and
// the pair code runs at 60 % of its speed.
// ================================================================================================
namespace {

#define PF_THREADS 256
#define PF_PLANE_CAP 4160          // planes up to 64 x 64 (nz * (ny + 1) complex elements)

struct PairsFftStage {
    int fftBlocks;
    int fracLo, fracHi;
};

// Work units of a pair wavefront in workgroup b of a fused launch (the number of pair workgroups is a multiple of 8)
__device__ __forceinline__ ChunkSchedule pair_schedule(const NbArgs& nb, const PairsFftStage& s, const int b) {
    const int pairBlock = b - s.fftBlocks, pairBlocks = (int) gridDim.x - s.fftBlocks, waveInBlock = threadIdx.x >> 6;
    ChunkSchedule sched = {pairBlock * (PF_THREADS / 64) + waveInBlock, pairBlocks * (PF_THREADS / 64), s.fracLo, s.fracHi, -1};
    if (nb.xcdAware) {
        // workgroup b runs on XCD b % 8; of the pair workgroups there it is number pairBlock / 8
        sched.xcd = b % OMM_NUM_XCD;
        sched.first = (pairBlock / OMM_NUM_XCD) * (PF_THREADS / 64) + waveInBlock;
        sched.stride = (pairBlocks / OMM_NUM_XCD) * (PF_THREADS / 64);
    }
    return sched;
}

template <int METHOD, bool ENERGY>
__global__ __launch_bounds__(PF_THREADS, 2) void pairs_fft_plane(NbArgs nb, PlaneArgs plane, PairsFftStage s, const float4* __restrict__ posqI, const float2* __restrict__ sigEpsI) {
    __shared__ PlaneShared<PF_PLANE_CAP> sh;
    const int b = blockIdx.x;
    if (b < s.fftBlocks) fft_plane_body<PF_THREADS, PF_PLANE_CAP>(plane, b, sh);
    else {
        const int wave = (b - s.fftBlocks) * (PF_THREADS / 64) + (threadIdx.x >> 6);
        nb_direct_body<METHOD, 1, ENERGY>(nb, posqI, sigEpsI, pair_schedule(nb, s, b), wave);
    }
}

template <int METHOD, bool ENERGY>
__global__ __launch_bounds__(PF_THREADS, 2) void pairs_fft_lines(NbArgs nb, FftArgs fft, PairsFftStage s, const float4* __restrict__ posqI, const float2* __restrict__ sigEpsI) {
    __shared__ FftShared sh;
    const int b = blockIdx.x;
    if (b < s.fftBlocks) fft_body_mode<PF_THREADS, 3, false, false>(fft, b, sh, 0, 0);
    else {
        const int wave = (b - s.fftBlocks) * (PF_THREADS / 64) + (threadIdx.x >> 6);
        nb_direct_body<METHOD, 1, ENERGY>(nb, posqI, sigEpsI, pair_schedule(nb, s, b), wave);
    }
}

template <int METHOD, bool ENERGY>
void launch_pairs_fft(int stage, int pairBlocks, hipStream_t st, const NbArgs& nb, const PlaneArgs& plane, const FftArgs& fft, PairsFftStage s, hipEvent_t evStart, hipEvent_t evStop) {
    const int grid = s.fftBlocks + pairBlocks;
#ifndef OMMHIP_EMU
    if (evStart != nullptr || evStop != nullptr) {
        // a timed launch: the events ride on the kernel's own dispatch packet
        if (stage == 1) hipExtLaunchKernelGGL((pairs_fft_lines<METHOD, ENERGY>), dim3(grid), dim3(PF_THREADS), 0, st, evStart, evStop, 0, nb, fft, s, nb.posq, nb.sigEps);
        else hipExtLaunchKernelGGL((pairs_fft_plane<METHOD, ENERGY>), dim3(grid), dim3(PF_THREADS), 0, st, evStart, evStop, 0, nb, plane, s, nb.posq, nb.sigEps);
        return;
    }
#else
    if (evStart != nullptr) hipEventRecord(evStart, st);
#endif
    if (stage == 1) hipLaunchKernelGGL((pairs_fft_lines<METHOD, ENERGY>), dim3(grid), dim3(PF_THREADS), 0, st, nb, fft, s, nb.posq, nb.sigEps);
    else hipLaunchKernelGGL((pairs_fft_plane<METHOD, ENERGY>), dim3(grid), dim3(PF_THREADS), 0, st, nb, plane, s, nb.posq, nb.sigEps);
#ifdef OMMHIP_EMU
    if (evStop != nullptr) hipEventRecord(evStop, st);
#endif
}

}  // namespace

extern "C" int ommhip_pairs_with_fft(const ommhip_neighbor_list* nl, const ommhip_nonbonded_params* p, const void* sig_eps_d, const ommhip_pme* pme,
                                     long long* force_d, double* energy_buffer_d, int energy_slots, int include_energy, void* stream) {
    const int nx = pme->nx, ny = pme->ny, nz = pme->nz;
    if (nl->pbc != 1 || !p->ewald || p->ljpme || nz * (ny + 1) > PF_PLANE_CAP || ny > 256 || nz > 256 || pme->fft_mode == 1) return -1;
    if (nl->posq_rel == nullptr) return 1;
    hipStream_t st = (hipStream_t) stream;
    // list fractions (in 64ths) at which stages 1 and 2 begin; tuning knob OPENMM_HIP_PAIRS_FFT_SPLIT="a,b"
    static int split[4] = {0, -1, -1, 64};
    if (split[1] < 0) {
        int a = 21, b = 43;
        const char* env = getenv("OPENMM_HIP_PAIRS_FFT_SPLIT");
        if (env != nullptr && sscanf(env, "%d,%d", &a, &b) != 2) { a = 21; b = 43; }
        if (a < 0) a = 0;
        if (b < a) b = a;
        if (b > 64) b = 64;
        split[2] = b; split[1] = a;
    }
    const NbArgs nb = make_nb_args(nl, p, sig_eps_d, force_d, energy_buffer_d, energy_slots);
    const FftArgs fft = make_xconv_args(pme, energy_buffer_d, energy_slots, include_energy);
    void* evStart = nullptr; void* evStop = nullptr;
    ommhip_profile_take(OMMHIP_TIMER_NB_DIRECT, &evStart, &evStop);
    hipEvent_t perLaunchStop = nullptr, bracketStop = nullptr;
    (void) perLaunchStop; (void) bracketStop;
    for (int stage = 0; stage < 3; stage++) {
        hipEvent_t e0 = stage == 0 ? (hipEvent_t) evStart : nullptr, e1 = stage == 2 ? (hipEvent_t) evStop : nullptr;
#ifndef OMMHIP_EMU
        // ... and every launch of a timed call with a pair of its own (hipExtLaunchKernelGGL takes one start and one stop event: the
        // bracketing pair gets the first start and the last stop from plain records around the loop instead)
        void* ks = nullptr; void* ke = nullptr;
        ommhip_profile_take_if(OMMHIP_TIMER_PAIRS_FFT_STAGE0 + stage, OMMHIP_TIMER_NB_DIRECT, &ks, &ke);
        if (ks != nullptr && ke != nullptr) {
            if (e0 != nullptr) hipEventRecord(e0, st);
            e0 = (hipEvent_t) ks;
            perLaunchStop = (hipEvent_t) ke; bracketStop = e1;
            e1 = perLaunchStop;
        }
        else { perLaunchStop = nullptr; bracketStop = nullptr; }
#endif
        const PlaneArgs plane = make_plane_args(pme, stage == 0);
        PairsFftStage s;
        s.fftBlocks = stage == 1 ? fft.numOuter * ((fft.numInner + fft.B - 1) / fft.B) : nx;
        s.fracLo = split[stage]; s.fracHi = split[stage + 1];
        // one chunk per wavefront for this stage's share of the list capacity (surplus wavefronts find no chunk and leave)
        const long long shareChunks = ((long long) nl->max_chunks * (s.fracHi - s.fracLo) + 63) / 64;
        int pairBlocks = (int) ((shareChunks + PF_THREADS / 64 - 1) / (PF_THREADS / 64));
        pairBlocks = (pairBlocks + OMM_NUM_XCD - 1) / OMM_NUM_XCD * OMM_NUM_XCD;       // the same number on every XCD
        const bool energy = include_energy != 0;
        if (use_ewald_poly(nb, nl, p, include_energy)) launch_pairs_fft<9, false>(stage, pairBlocks, st, nb, plane, fft, s, e0, e1);
        else if (p->use_switch) { if (energy) launch_pairs_fft<3, true>(stage, pairBlocks, st, nb, plane, fft, s, e0, e1); else launch_pairs_fft<3, false>(stage, pairBlocks, st, nb, plane, fft, s, e0, e1); }
        else { if (energy) launch_pairs_fft<1, true>(stage, pairBlocks, st, nb, plane, fft, s, e0, e1); else launch_pairs_fft<1, false>(stage, pairBlocks, st, nb, plane, fft, s, e0, e1); }
#ifndef OMMHIP_EMU
        if (perLaunchStop != nullptr && bracketStop != nullptr) hipEventRecord(bracketStop, st);
#endif
    }
    return (int) hipGetLastError();
}

// One translation unit for the three kernels that open a force evaluation, and the launch that runs them side by side.
//
// After nl_prepare three pieces of work are independent of each other: the (possible) neighbour-list rebuild, the PME
// charge spreading and the per-term forces (bonds, angles, torsions, 1-4s).  Each is latency-bound on its own and uses
// a fraction of the chip; as separate launches they cost 1-53 + 16 + 8 us back to back, and running them on separate
// streams costs ~13 us per cross-stream dependency on this stack.  force_front runs them as three groups of workgroups
// of ONE launch (same workgroup size, LDS aliased through a union), so they overlap without any synchronisation object.
#include <cstring>
#include "neighbor.hip"
#include "pme.hip"
#include "bonded.hip"

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
    // heavy, rare work first in the grid so that it starts first
    if (b < f.nlBlocks) nl_find_body<PBC>(f.nl, b, f.nlBlocks, sh.nl);
    else if (b < f.nlBlocks + f.spreadBlocks) pme_spread_body(f.pme, b - f.nlBlocks, sh.spread);
    else terms_body(f.terms, b - f.nlBlocks - f.spreadBlocks, sh.termPartial);
}

}  // namespace

extern "C" int ommhip_force_front(const ommhip_neighbor_list* nl, const ommhip_pme* pme, int num_lists, const ommhip_term_batch* lists,
                                  const void* pos_d, long long* force_d, double* energy_buffer_d, int energy_slots, int include_energy, void* stream) {
    static_assert(NL_THREADS == 256, "the fused launch uses 256-thread workgroups for all three kinds of work");
    hipStream_t st = (hipStream_t) stream;
    FrontArgs f;
    f.nl = make_nl_args(nl);
    if (f.nl.cellMode) hipLaunchKernelGGL(nl_bin_blocks, dim3(1), dim3(1024), 0, st, f.nl);
    f.nlBlocks = f.nl.numBlocks;
    f.spreadBlocks = 0;
    if (pme != nullptr) {
        if (pme->spread_mode == 1 || !pme->grid_precleared) return 1;          // the direct-atomics variant and un-cleared grids are not fused
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
    ommhip_profile_end(OMMHIP_TIMER_NL_UPDATE, stream);
    return (int) hipGetLastError();
}

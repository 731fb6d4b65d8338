/* openmm_hip_amoeba.h -- C ABI of the AMOEBA kernels of the OpenMM "HIP" platform (MI355X / gfx950), part of
 * libopenmm_hip_kernels.so like openmm_hip_kernels.h and under the same rules: plain C, device pointers from ommhip_malloc(),
 * `stream` = hipStream_t as void*, 0 or a hipError_t as the return value, nothing throws.
 *
 * Replaces, for a Context on the HIP platform, the kernels the AMOEBA plugin declares in
 * plugins/amoeba/openmmapi/include/openmm/amoebaKernels.h: CalcAmoebaVdwForceKernel (:182-219) and
 * CalcAmoebaMultipoleForceKernel (:82-139).  The oracle is the plugin's own Reference implementation
 * (plugins/amoeba/platforms/reference/src/SimTKReference/AmoebaReferenceVdwForce.cpp, AmoebaReferenceMultipoleForce.cpp).
 *
 * All arithmetic is double precision on the atom-ordered positions (pos_d: double4[num_atoms], HipContext::pos); forces are added
 * to the platform's 64-bit fixed-point buffer (slot order, through slot_of_atom_d), energies to energy_buffer_d[0..energy_slots).
 */
#ifndef OPENMM_HIP_AMOEBA_H_
#define OPENMM_HIP_AMOEBA_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * AmoebaVdwForce: buffered 14-7 (or Lennard-Jones) between "reduced" sites.
 * Reference: AmoebaReferenceVdwForce::calculatePairIxn (AmoebaReferenceVdwForce.cpp:106-171), setReducedPositions (:173-189),
 * calculateForceAndEnergy (:191-323), kernel wrapper AmoebaReferenceKernels.cpp:656-690.
 *   - the interaction site of atom i is  red_i (x_i - x_parent(i)) + x_parent(i)  (hydrogens sit closer to their heavy atom);
 *     the force on a site is shared between the atom (red_i) and its parent (1 - red_i)                        (:92-104)
 *   - sigma / epsilon come from a type x type matrix built by AmoebaVdwForceImpl::createParameterMatrix (combining rules)
 *   - with a cutoff, pairs are selected by the distance between the ATOMS (the Reference builds its neighbour list on the
 *     atom positions, AmoebaReferenceKernels.cpp:676) and the energy is tapered between taper_cutoff and cutoff  (:157-163)
 *   - alchemical scaling: pairs selected by the method get epsilon * lambda^n and the softcore term alpha (1 - lambda)^2 (:221-227)
 * ------------------------------------------------------------------------------------------ */
typedef struct ommhip_amoeba_vdw {
    int num_atoms;
    const int* parent;             /* device int[num_atoms]: indexIV (the atom itself when it is its own site) */
    const double* reduction;       /* device double[num_atoms] */
    const int* type;               /* device int[num_atoms]: row/column of the parameter matrices */
    int num_types;
    const double* sigma;           /* device double[num_types * num_types] */
    const double* epsilon;         /* device double[num_types * num_types] */
    const int* excl_start;         /* device int[num_atoms + 1]: CSR of excluded partners, ascending within a row */
    const int* excl_atoms;
    const unsigned char* alchemical;   /* device [num_atoms] */
    int alchemical_method;         /* 0 none, 1 decouple (one of the two atoms alchemical), 2 annihilate (either) */
    double epsilon_scale;          /* lambda^n */
    double softcore;               /* alpha (1 - lambda)^2 */
    int lennard_jones;             /* 1: 12-6 instead of buffered 14-7 */
    int periodic;                  /* 1: CutoffPeriodic (box given to the call), 0: NoCutoff */
    double cutoff, taper_cutoff, taper_c3, taper_c4, taper_c5;
    double* reduced;               /* device double4[num_atoms] scratch: the interaction sites */
} ommhip_amoeba_vdw;

int ommhip_amoeba_vdw_forces(const ommhip_amoeba_vdw* vdw, const void* pos_d, const double box[6], const int* slot_of_atom_d, int padded_atoms,
                             long long* force_d, double* energy_buffer_d, int energy_slots, int include_energy, void* stream);

#ifdef __cplusplus
}
#endif
#endif

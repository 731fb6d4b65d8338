/* openmm_hip_amoeba.h -- C ABI of the AMOEBA kernels of the OpenMM "HIP" platform (MI355X / gfx950), part of
 * libopenmm_hip_kernels.so like openmm_hip_kernels.h and under the same rules: plain C, device pointers from ommhip_malloc(),
 * `stream` = hipStream_t as void*, 0 or a hipError_t as the return value, nothing throws.
 *
 * Replaces, for a Context on the HIP platform, the kernels the AMOEBA plugin declares in
 * plugins/amoeba/openmmapi/include/openmm/amoebaKernels.h: CalcAmoebaVdwForceKernel (:182-219) and
 * CalcAmoebaMultipoleForceKernel (:82-139).  The oracle is the plugin's own Reference implementation
 * (plugins/amoeba/platforms/reference/src/SimTKReference/AmoebaReferenceVdwForce.cpp, AmoebaReferenceMultipoleForce.cpp).
 *
 * Positions are the atom-ordered doubles (pos_d: double4[num_atoms], HipContext::pos); sums, frames, reciprocal-space read-back and
 * the solver are double precision, the pair arithmetic float or double (mixed_precision), the grids float; forces are added
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
    /* CutoffPeriodic: pair lists (amoeba_pairs.h; see ommhip_amoeba_multipole below; rebuilt by every call unless a skin is given) -- partners within the cutoff by
     * atom distance, exclusions left out -- when pair_list is given; without it (and for NoCutoff) every thread scans all atoms.
     * S = padded_atoms with atom_of_slot, num_atoms without. */
    const int* atom_of_slot;       /* device int[padded_atoms]: atom at each slot, -1 = padding (optional: spatial order -> far tiles skipped) */
    double* tile_bounds;           /* device double4[2 * ceil(S / 128)] work array */
    int* excl_pos;                 /* device int[entries of excl_atoms] work array: the excluded partners as scan positions, rows sorted */
    int* pair_list;                /* device int[pair_cap * S] work array */
    int* pair_count;               /* device int[4 * S] work array (four sub-lists per atom) */
    int pair_cap;                  /* list entries per atom (a multiple of 4: four sub-lists of pair_cap / 4); a call that needs more returns -2 */
    int* pair_overflow;            /* device int work word */
    int* pair_needed;              /* HOST int written with the return code -2 (or NULL) */
    /* Verlet skin, optional (skin > 0 with both arrays): the lists hold the partners within cutoff + skin and are rebuilt only when some atom
     * has moved by more than skin / 2 since the last build (decided on the device) or when the caller says so -- after a change of the slot
     * order, the box, the parameters or the list capacity; the pair kernel re-tests the cutoff. */
    double skin;
    double* ref_pos;               /* device double4[num_atoms] work array: positions at the last build */
    int* list_state;               /* device int[4] work words, zeroed by the caller once; [2] counts the builds */
    int force_rebuild;
    int* list_builds;              /* HOST int (or NULL), written by every call: list builds so far (diagnostics) */
    int mixed_precision;           /* 1: the pair arithmetic of the list kernel in float (separation formed in double, sums in double) */
} ommhip_amoeba_vdw;

int ommhip_amoeba_vdw_forces(const ommhip_amoeba_vdw* vdw, const void* pos_d, const double box[6], const int* slot_of_atom_d, int padded_atoms,
                             long long* force_d, double* energy_buffer_d, int energy_slots, int include_energy, void* stream);

/* ------------------------------------------------------------------------------------------
 * AmoebaMultipoleForce with PME: permanent atomic multipoles (charge, dipole, traceless quadrupole in a local frame) and induced
 * dipoles (polarisability alpha_i, Thole damping), Ewald-split.
 * Reference: AmoebaReferencePmeMultipoleForce (AmoebaReferenceMultipoleForce.cpp:4800-6818) and its base class: frames
 * :396-561 (applyRotationMatrixToParticle), chirality :354-378, scale maps :181-266, fixed field :5079-5201, self terms :6294-6333,
 * reciprocal space :5204-6006, torque -> force :1476-1712, driver :1775-1850 / :6753-6775.
 *
 * Formulation of this implementation (not the Reference's quasi-internal spherical-harmonic one): every pair term is
 * L_A L_B f(r) with the multipole operator L = q - mu.grad + Q:grad grad of each site acting on a radial kernel f given through
 * the chain B_0 = f, B_n = -(1/r) dB_(n-1)/dr.  Three chains per pair: erfc-screened Coulomb minus (1 - m_ij) of the bare kernel
 * (permanent-permanent), and the same with (1 - p_ij lambda_(2n+1)) / (1 - d_ij lambda_(2n+1)) for permanent multipoles against the
 * two induced-dipole sets (lambda: Thole damping, a consistent derivative chain as well).  One device function evaluates energy,
 * force and torque of L_A L_B f for any chain (amoeba_multipole.hip: mpole_pair).  Reciprocal space: the multipoles are spread
 * with B-spline derivatives, convolved by the platform's own 3-D FFT (ommhip_pme_convolve), and the potential and its first three
 * derivatives are read back at every atom.
 *
 * Supported: PME with polarization Direct (induced dipoles = alpha x field of the permanent multipoles) or Mutual (the induced dipoles
 * also polarize each other: (1/alpha - T) mu = E, solved for both dipole sets by preconditioned conjugate gradients -- each step
 * one induced-dipole field evaluation in real and reciprocal space -- until the Reference's own measure, debye x the RMS of
 * alpha (E + T mu) - mu (AmoebaReferenceMultipoleForce::convergeInduceDipolesByDIIS :939-1005), falls below the target; the energy
 * keeps its form and the force gains the term -1/2 mu_d (dT/dx) mu_p) or Extrapolated (a fixed number of perturbation orders, see
 * extrapolation_orders below).  The host falls back to the AMOEBA plugin's Reference kernel for everything else (NoCutoff).
 * ------------------------------------------------------------------------------------------ */
#define OMMHIP_AMOEBA_MAX_HISTORY 6
#define OMMHIP_AMOEBA_MAX_EXT_ORDERS 8
typedef struct ommhip_amoeba_multipole {
    int num_atoms;
    /* per atom, device */
    const double* charge;          /* [n] */
    const double* mol_dipole;      /* [3n] local-frame dipole */
    const double* mol_quadrupole;  /* [6n] local-frame quadrupole xx, xy, xz, yy, yz, zz */
    const int* axis;               /* int4[n]: (axis type as AmoebaMultipoleForce::MultipoleAxisTypes, z atom, x atom, y atom; -1 = none) */
    const double* thole;           /* [n] */
    const double* damping;         /* [n] */
    const double* polarity;        /* [n] */
    /* pairs with scale factors different from one: CSR per atom (both directions), partners ascending; scales = (m, p, d, u) */
    const int* special_start;      /* [n + 1] */
    const int* special_atom;
    const double* special_scale;   /* double4 per entry */
    double cutoff, alpha;
    /* work arrays, device, allocated by the caller */
    double* lab_dipole;            /* [3n] */
    double* lab_quadrupole;        /* [6n] */
    double* field_d;               /* [3n] field of the permanent multipoles, d-scaled */
    double* field_p;               /* [3n] ... p-scaled */
    double* induced_d;             /* [3n] */
    double* induced_p;             /* [3n] */
    double* phi;                   /* [20n] reciprocal potential of the permanent multipoles and its derivatives up to third order (Cartesian) */
    double* phi_induced;           /* [20n] the same for the induced dipoles (mu_d + mu_p) / 2 */
    double* torque;                /* [3n] */
    /* mutual polarization (mutual = 1) */
    int mutual, max_iterations;
    double target_epsilon;         /* AmoebaMultipoleForce::getMutualInducedTargetEpsilon() */
    double* phi_induced_p;         /* [20n]: with mutual polarization phi_induced holds the potential of mu_d, this one that of mu_p */
    double* solver;                /* [24n + 16] work vectors of the conjugate-gradient solver */
    double* status;                /* HOST double[2], written by the calls: [0] epsilon reached, [1] iterations (or NULL) */
    void* pme;                     /* const ommhip_pme*: grid sizes, box, moduli, eterm, real / complex grids, twiddles of the platform's PME */
    /* Pair lists (amoeba_pairs.h; rebuilt by every call unless a skin is given): per atom the partners within the cutoff, found by a scan that only tests
     * distances; the kernels with the long pair arithmetic (fixed field, induced-dipole field, forces) walk these lists.  With the
     * platform's slot order (optional: atom_of_slot / slot_of_atom / scan_slots, NULL / 0 = atom order) the scan works on 128-slot tiles
     * with bounding boxes and skips the tiles farther apart than the cutoff (rectangular boxes).  S = scan_slots, or num_atoms. */
    const int* atom_of_slot;       /* device int[scan_slots]: atom at each slot, -1 = padding */
    const int* slot_of_atom;       /* device int[num_atoms] */
    int scan_slots;                /* the platform's padded atom count */
    double* tile_bounds;           /* device double4[2 * ceil(S / 128)] work array: tile centres, then half extents */
    int* special_pos;              /* device int[entries of special_atom] work array: the partners as scan positions, rows sorted */
    double* special_scale_sorted;  /* device double4[entries] work array */
    int* pair_list;                /* device int[pair_cap * S] work array */
    int* pair_count;               /* device int[4 * S] work array (four sub-lists per atom) */
    int pair_cap;                  /* list entries per atom (a multiple of 4: four sub-lists of pair_cap / 4); a call that needs more returns -2 */
    int* pair_overflow;            /* device int work word */
    int* pair_needed;              /* HOST int written with the return code -2: the capacity that would have been enough (or NULL) */
    /* mutual polarization, optional: a second grid set (an ommhip_pme that shares everything but grid_real / grid_complex with `pme`), a side
     * stream and two ordering events.  With them the potentials of the two dipole sets travel through the same launches on two grids, and
     * (round 5) the side stream carries what needs neither the pair lists nor each other's results beside the long pair kernels of `stream`:
     * the reciprocal potential of the permanent multipoles, the reciprocal chain and the vector stages of every solver iteration, the
     * potentials of the converged dipoles.  Best created with a higher priority than `stream`. */
    void* pme2; void* stream2; void* event_a; void* event_b;
    float* pair_cache;             /* device float[5 * pair_cap * S] or NULL (mutual polarization): per list entry the separation and the two
                                    * coefficients of the damped dipole-dipole chain, written once per evaluation and read by every solver iteration */
    /* Mutual polarization, optional: solutions of earlier calls, from which the solver extrapolates its first guess (the CUDA platform's and
     * Tinker's remedy for the ~10 iterations a solve from the direct dipoles takes; the converged dipoles are the same to the tolerance).
     * The caller owns the ring and says what it holds: a call starts from sum_k history_coeff[k] x record k, k < history_use, record 0
     * = slot history_newest, record k = slot history_newest - k (mod history_slots); history_use = 0: from the direct dipoles.  It stores
     * its converged (mu_d, mu_p) in slot history_store (-1: nowhere). */
    double* history;               /* device double[history_slots * 6n] or NULL */
    int history_slots, history_newest, history_store, history_use;
    double history_coeff[OMMHIP_AMOEBA_MAX_HISTORY];
    /* Verlet skin of the pair lists, optional: as in ommhip_amoeba_vdw */
    double skin;
    double* ref_pos;
    int* list_state;
    int force_rebuild;
    int* list_builds;
    /* Extrapolated polarization (mutual = 0, extrapolation_orders = K > 0; AmoebaMultipoleForce::Extrapolated with its K coefficients c_n):
     * mu_0 = alpha E, mu_(n+1) = alpha T mu_n, induced dipoles = sum_n (sum_(j >= n) c_j) mu_n; K - 1 field evaluations, no iteration.
     * Needs solver and phi_induced_p as the mutual solver does. */
    int extrapolation_orders;
    double ext_coefficients[OMMHIP_AMOEBA_MAX_EXT_ORDERS];
    double* ext_dipoles;           /* device double[K * 6n]: (mu_d, mu_p) of every order */
    double* ext_gradients;         /* device double[(K - 1) * 12n]: gradient (xx yy zz xy xz yz) of the field of orders 0 .. K - 2, d set then p set */
    float* solver_gather;          /* device float[6 * S] or NULL (mutual polarization): the solver keeps the vectors whose field it takes as six floats per scan
                                    * position here as well -- the induced-dipole field kernel gathers them from this copy */
    int mixed_precision;           /* 1: the pair arithmetic of ordinary pairs in float (separations formed in double, sums in double; covalently related
                                    * pairs stay in double) -- the "mixed" mode of the reference's GPU platforms; 0: everything in double */
    int expected_iterations;       /* iterations the previous solve took (0 = unknown): that many minus one are enqueued before the host first waits for the
                                    * convergence measure, which the device forms itself; status[1] reports what this call took */
    /* Optional hook (NULL = none): called once per call of ommhip_amoeba_multipole_forces, on the calling thread, after the pair-list build and the
     * work that does not need the lists (frames, reciprocal potential of the permanent multipoles) have been ENQUEUED on `stream` and before the
     * host waits for the build's overflow word.  A caller with independent work for another stream (the platform's AmoebaVdwForce, whose own list
     * build ends in a host wait as well) launches it here, so that the two builds run side by side instead of one after the other. */
    void (*after_lists_enqueued)(void* arg);
    void* after_lists_arg;
} ommhip_amoeba_multipole;

/* Whole evaluation: frames -> reciprocal and real-space field -> induced dipoles -> energy, forces, torques -> forces. */
int ommhip_amoeba_multipole_forces(const ommhip_amoeba_multipole* mp, const void* pos_d, const double box[6], const int* slot_of_atom_d, int padded_atoms,
                                   long long* force_d, double* energy_buffer_d, int energy_slots, int include_energy, void* stream);
/* Only as far as the induced dipoles (lab_dipole, induced_d, induced_p are left on the device): getInducedDipoles() and friends. */
int ommhip_amoeba_multipole_induce(const ommhip_amoeba_multipole* mp, const void* pos_d, const double box[6], void* stream);

#ifdef __cplusplus
}
#endif
#endif

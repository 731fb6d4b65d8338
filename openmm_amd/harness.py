"""ctypes front end of libommharness.so: just enough of the OpenMM API (System, NonbondedForce,
bonded forces, integrators, Context, State) to build test/benchmark systems from numpy arrays and run
them on the "HIP", "CPU" and "Reference" platforms.  Names follow the OpenMM Python API
(wrappers/python/openmm) so the tests read like the reference's own.

The HIP platform is loaded from openmm_amd/lib/libOpenMMHIP.so through OpenMM's own plugin loader
(Platform::loadPluginLibrary -> registerPlatforms()), i.e. through the drop-in boundary.  Tests that
check host logic without a GPU pass `emulated=True`, which loads the CPU-emulated twin from
tests/emu/_build instead; the two can not be mixed in one process.
"""
import ctypes as C
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_DIR = os.path.join(_HERE, "lib")
EMU_DIR = os.path.join(ROOT, "tests", "emu", "_build")
ORACLE_DIR = os.path.join(ROOT, "oracle", "_ref")          # libOpenMMCPU.so: CPU baseline / second checker (tests and bench only)
HOST_LIB_DIR = os.path.join(os.environ.get("OPENMM_DIR", os.path.join(ROOT, "build", "openmm")), "lib")      # the host OpenMM library the plugin links

NoCutoff, CutoffNonPeriodic, CutoffPeriodic, Ewald, PME, LJPME = range(6)
VERLET, LANGEVIN, LANGEVIN_MIDDLE = 0, 1, 2

_lib = None
_loaded_plugins = set()


class OpenMMError(RuntimeError):
    pass


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(LIB_DIR, "libommharness.so")
        if not os.path.exists(path):
            raise OpenMMError("harness library missing: %s (run __graft_entry__.build())" % path)
        C.CDLL(os.path.join(HOST_LIB_DIR, "libOpenMM.so"), mode=C.RTLD_GLOBAL)
        _lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        for name in ("omm_last_error", "omm_platform_name", "omm_version", "omm_context_platform_name", "omm_context_platform_property"):
            getattr(_lib, name).restype = C.c_char_p
        for name in ("omm_system_create", "omm_nonbonded_create", "omm_add_harmonic_bonds", "omm_add_harmonic_angles",
                     "omm_add_periodic_torsions", "omm_add_cmmotion_remover", "omm_add_monte_carlo_barostat", "omm_integrator_create", "omm_context_create",
                     "omm_add_custom_bond_force", "omm_add_custom_angle_force", "omm_add_custom_compound_bond_force", "omm_custom_integrator_create"):
            getattr(_lib, name).restype = C.c_void_p
        _lib.omm_platform_speed.restype = C.c_double
    return _lib


def _check(rc):
    if rc != 0:
        raise OpenMMError(lib().omm_last_error().decode())


def _handle(p):
    if not p:
        raise OpenMMError(lib().omm_last_error().decode())
    return C.c_void_p(p)


def domain_info():
    """Diagnostics of the most recently created decomposed Context (ommhip_plugin_dd_info): [ranks, halo mode, slots per rank,
    slots converted per step, bytes sent per step, bytes received per step, re-sorts so far, half-shell: 0, or 1 + the slots of the
    lower neighbour's section whose forces this rank computes and returns]."""
    path = next(iter(_loaded_plugins))
    plugin = C.CDLL(path)
    out = (C.c_longlong * 8)()
    if plugin.ommhip_plugin_dd_info(out) != 0:
        raise OpenMMError("no decomposed Context")
    return [int(v) for v in out[:8]]


def valence_lists_launched():
    """Lists of kernels/valence.hip (AMOEBA valence terms) launched so far by the loaded HIP plugin."""
    plugin = C.CDLL(next(p for p in _loaded_plugins if p.endswith("libOpenMMHIP.so")))
    plugin.ommhip_plugin_valence_lists_launched.restype = C.c_longlong
    return int(plugin.ommhip_plugin_valence_lists_launched())


def interpreted_bond_launches():
    """Launches of the interpreted CustomBondForce kernel (any expression; kernels/custom_integrator.hip) by the loaded HIP plugin so far."""
    plugin = C.CDLL(next(p for p in _loaded_plugins if p.endswith("libOpenMMHIP.so")))
    plugin.ommhip_plugin_interpreted_bond_launches.restype = C.c_longlong
    return int(plugin.ommhip_plugin_interpreted_bond_launches())


def load_hip_platform(emulated=False):
    """Register the HIP platform through OpenMM's plugin loader.  Raises if the plugin is missing."""
    path = os.path.join(EMU_DIR if emulated else LIB_DIR, "libOpenMMHIP.so")
    if path in _loaded_plugins:
        return
    if _loaded_plugins:
        raise OpenMMError("a different HIP plugin build is already loaded in this process")
    if not os.path.exists(path):
        raise OpenMMError("HIP platform plugin not found: %s" % path)
    _check(lib().omm_load_plugin(path.encode()))
    _loaded_plugins.add(path)


def load_cpu_platform():
    path = os.path.join(ORACLE_DIR, "libOpenMMCPU.so")
    if path in _loaded_plugins:
        return
    _check(lib().omm_load_plugin(path.encode()))
    _loaded_plugins.add(path)


def platform_names():
    L = lib()
    return [L.omm_platform_name(i).decode() for i in range(L.omm_num_platforms())]


class System:
    def __init__(self):
        self.h = _handle(lib().omm_system_create())
        self.forces = []

    @classmethod
    def from_xml(cls, text):
        """XmlSerializer::deserialize<System> of the reference (serialization/include/openmm/serialization/XmlSerializer.h:74-76)"""
        lib().omm_system_from_xml.restype = C.c_void_p
        self = cls.__new__(cls)
        self.h = _handle(lib().omm_system_from_xml(text.encode() if isinstance(text, str) else text))
        self.forces = []
        return self

    def to_xml(self):
        blob, size = C.c_void_p(), C.c_long(0)
        _check(lib().omm_system_to_xml(self.h, C.byref(blob), C.byref(size)))
        text = C.string_at(blob, size.value).decode()
        lib().omm_free(blob)
        return text

    def getNumForces(self):
        return lib().omm_system_num_forces(self.h)

    def getNumConstraints(self):
        return lib().omm_system_num_constraints(self.h)

    def addParticles(self, masses):
        m = np.ascontiguousarray(masses, dtype=np.float64)
        _check(lib().omm_system_add_particles(self.h, len(m), _dp(m)))

    def getNumParticles(self):
        return lib().omm_system_num_particles(self.h)

    def setDefaultPeriodicBoxVectors(self, a, b, c):
        box = np.ascontiguousarray(np.array([a, b, c], dtype=np.float64).reshape(9))
        _check(lib().omm_system_set_box(self.h, _dp(box)))

    def addConstraints(self, pairs, distances):
        p = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1)
        d = np.ascontiguousarray(distances, dtype=np.float64)
        _check(lib().omm_system_add_constraints(self.h, len(d), _ip(p), _dp(d)))

    def addHarmonicBondForce(self, atoms, length, k):
        a = np.ascontiguousarray(atoms, dtype=np.int32).reshape(-1)
        l = np.ascontiguousarray(length, dtype=np.float64)
        kk = np.ascontiguousarray(k, dtype=np.float64)
        return _handle(lib().omm_add_harmonic_bonds(self.h, len(l), _ip(a), _dp(l), _dp(kk)))

    def addHarmonicAngleForce(self, atoms, angle, k):
        a = np.ascontiguousarray(atoms, dtype=np.int32).reshape(-1)
        t = np.ascontiguousarray(angle, dtype=np.float64)
        kk = np.ascontiguousarray(k, dtype=np.float64)
        return _handle(lib().omm_add_harmonic_angles(self.h, len(t), _ip(a), _dp(t), _dp(kk)))

    def addPeriodicTorsionForce(self, atoms, periodicity, phase, k):
        a = np.ascontiguousarray(atoms, dtype=np.int32).reshape(-1)
        n = np.ascontiguousarray(periodicity, dtype=np.int32)
        ph = np.ascontiguousarray(phase, dtype=np.float64)
        kk = np.ascontiguousarray(k, dtype=np.float64)
        return _handle(lib().omm_add_periodic_torsions(self.h, len(n), _ip(a), _ip(n), _dp(ph), _dp(kk)))

    def _custom(self, fn, per, energy, names, atoms, params, *lead):
        a = np.ascontiguousarray(atoms, dtype=np.int32).reshape(-1, per)
        p = np.ascontiguousarray(params, dtype=np.float64).reshape(len(a), len(names))
        return _handle(fn(self.h, *lead, energy.encode(), ",".join(names).encode(), len(a), _ip(a), _dp(p)))

    def addCustomBondForce(self, energy, names, atoms, params):
        return self._custom(lib().omm_add_custom_bond_force, 2, energy, names, atoms, params)

    def addCustomAngleForce(self, energy, names, atoms, params):
        return self._custom(lib().omm_add_custom_angle_force, 3, energy, names, atoms, params)

    def addCustomCompoundBondForce(self, particlesPerBond, energy, names, atoms, params):
        return self._custom(lib().omm_add_custom_compound_bond_force, particlesPerBond, energy, names, atoms, params, particlesPerBond)

    def addGBSAOBCForce(self, charge, radius, scale, method=0, cutoff=1.0, solventDielectric=78.3, soluteDielectric=1.0):
        """openmmapi/include/openmm/GBSAOBCForce.h; method: 0 NoCutoff, 1 CutoffNonPeriodic, 2 CutoffPeriodic"""
        q, r, sc = (np.ascontiguousarray(a, dtype=np.float64) for a in (charge, radius, scale))
        lib().omm_add_gbsa_obc.restype = C.c_void_p
        h = _handle(lib().omm_add_gbsa_obc(self.h, len(q), _dp(q), _dp(r), _dp(sc), method, C.c_double(cutoff), C.c_double(solventDielectric), C.c_double(soluteDielectric)))
        self.forces.append(h)
        return h

    def addCMMotionRemover(self, frequency=1):
        return _handle(lib().omm_add_cmmotion_remover(self.h, frequency))

    def addMonteCarloBarostat(self, pressure, temperature, frequency=25, seed=1):
        return _handle(lib().omm_add_monte_carlo_barostat(self.h, C.c_double(pressure), C.c_double(temperature), frequency, seed))


class NonbondedForce:
    """Created already attached to `system` (the System owns it, as in OpenMM)."""

    def __init__(self, system, method=NoCutoff, cutoff=1.0, ewaldErrorTolerance=5e-4, useDispersionCorrection=True,
                 switchingDistance=None):
        use_switch = switchingDistance is not None
        self.h = _handle(lib().omm_nonbonded_create(system.h, method, C.c_double(cutoff), C.c_double(ewaldErrorTolerance),
                                                    int(useDispersionCorrection), int(use_switch),
                                                    C.c_double(switchingDistance if use_switch else -1.0)))

    def addParticles(self, charge, sigma, epsilon):
        q = np.ascontiguousarray(charge, dtype=np.float64)
        s = np.ascontiguousarray(sigma, dtype=np.float64)
        e = np.ascontiguousarray(epsilon, dtype=np.float64)
        _check(lib().omm_nonbonded_add_particles(self.h, len(q), _dp(q), _dp(s), _dp(e)))

    def addExceptions(self, pairs, chargeProd, sigma, epsilon):
        p = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1)
        qq = np.ascontiguousarray(chargeProd, dtype=np.float64)
        s = np.ascontiguousarray(sigma, dtype=np.float64)
        e = np.ascontiguousarray(epsilon, dtype=np.float64)
        _check(lib().omm_nonbonded_add_exceptions(self.h, len(qq), _ip(p), _dp(qq), _dp(s), _dp(e)))

    def createExceptionsFromBonds(self, bonds, coulomb14Scale, lj14Scale):
        p = np.ascontiguousarray(bonds, dtype=np.int32).reshape(-1)
        _check(lib().omm_nonbonded_create_exceptions_from_bonds(self.h, len(p) // 2, _ip(p), C.c_double(coulomb14Scale), C.c_double(lj14Scale)))

    def getNumExceptions(self):
        return lib().omm_nonbonded_num_exceptions(self.h)

    def setPMEParameters(self, alpha, nx, ny, nz):
        _check(lib().omm_nonbonded_set_pme_parameters(self.h, C.c_double(alpha), nx, ny, nz))

    def setLJPMEParameters(self, alpha, nx, ny, nz):
        _check(lib().omm_nonbonded_set_ljpme_parameters(self.h, C.c_double(alpha), nx, ny, nz))

    def getLJPMEParametersInContext(self, context):
        alpha = C.c_double()
        n = (C.c_int * 3)()
        _check(lib().omm_nonbonded_get_ljpme_parameters_in_context(self.h, context.h, C.byref(alpha), n))
        return alpha.value, n[0], n[1], n[2]

    def setReactionFieldDielectric(self, dielectric):
        _check(lib().omm_nonbonded_set_reaction_field_dielectric(self.h, C.c_double(dielectric)))

    def setReciprocalSpaceForceGroup(self, group):
        _check(lib().omm_nonbonded_set_reciprocal_force_group(self.h, group))

    def setExceptionsUsePeriodicBoundaryConditions(self, periodic):
        _check(lib().omm_nonbonded_set_exceptions_use_periodic(self.h, int(periodic)))

    def getPMEParametersInContext(self, context):
        alpha = C.c_double()
        n = (C.c_int * 3)()
        _check(lib().omm_nonbonded_get_pme_parameters_in_context(self.h, context.h, C.byref(alpha), n))
        return alpha.value, n[0], n[1], n[2]


# ---- the AMOEBA forces with native kernels (libommharness_amoeba.so over plugins/amoeba/openmmapi)
_alib = None
Mutual, Direct, Extrapolated = 0, 1, 2                                   # AmoebaMultipoleForce::PolarizationType
ZThenX, Bisector, ZBisect, ThreeFold, ZOnly, NoAxisType = range(6)       # AmoebaMultipoleForce::MultipoleAxisTypes
Covalent12, Covalent13, Covalent14, Covalent15, PolarizationCovalent11 = range(5)


def amoeba_lib():
    global _alib
    if _alib is None:
        lib()
        path = os.path.join(LIB_DIR, "libommharness_amoeba.so")
        if not os.path.exists(path):
            raise OpenMMError("AMOEBA harness library missing: %s (run __graft_entry__.build())" % path)
        C.CDLL(os.path.join(HOST_LIB_DIR, "libOpenMMAmoeba.so"), mode=C.RTLD_GLOBAL)
        _alib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        _alib.omm_amoeba_last_error.restype = C.c_char_p
        _alib.omm_amoeba_multipole_create.restype = C.c_void_p
        _alib.omm_amoeba_vdw_create.restype = C.c_void_p
        _alib.omm_amoeba_torsion_torsion_create.restype = C.c_void_p
    return _alib


def load_amoeba_plugins(emulated=False, native=True):
    """The AMOEBA plugin's Reference kernels (they attach to every ReferencePlatform-derived platform, the HIP platform included, as
    fallback forces) and -- native=True -- libOpenMMAmoebaHIP.so, whose kernels then take AmoebaVdwForce and AmoebaMultipoleForce (PME)."""
    load_hip_platform(emulated)
    paths = [os.path.join(HOST_LIB_DIR, "libOpenMMAmoebaReference.so")]
    if native:
        paths.append(os.path.join(EMU_DIR if emulated else LIB_DIR, "libOpenMMAmoebaHIP.so"))
    for path in paths:
        if path not in _loaded_plugins:
            if path.endswith("AmoebaHIP.so"):
                C.CDLL(path, mode=C.RTLD_GLOBAL)      # so that ommhip_amoeba_native_evaluations can be found
            _check(lib().omm_load_plugin(path.encode()))
            _loaded_plugins.add(path)


def amoeba_native_evaluations():
    """(vdw, multipole) evaluations the native AMOEBA kernels performed in this process: a silent fallback to the Reference kernels shows as 0."""
    fn = C.CDLL(None).ommhip_amoeba_native_evaluations
    out = (C.c_longlong * 2)()
    fn(out)
    return int(out[0]), int(out[1])


def amoeba_list_builds():
    """(vdw, multipole) pair-list builds of the most recently used native AMOEBA kernels (Verlet skin: fewer than evaluations)."""
    fn = C.CDLL(None).ommhip_amoeba_list_builds
    out = (C.c_longlong * 2)()
    fn(out)
    return int(out[0]), int(out[1])


def amoeba_solver_iterations():
    """(solves, iterations summed) of the mutual-polarization solver of the native multipole kernels in this process."""
    fn = C.CDLL(None).ommhip_amoeba_solver_iterations
    out = (C.c_longlong * 2)()
    fn(out)
    return int(out[0]), int(out[1])


def _acheck(rc):
    if rc != 0:
        raise OpenMMError(amoeba_lib().omm_amoeba_last_error().decode())


class AmoebaMultipoleForce:
    NoCutoff, PME = 0, 1

    def __init__(self, system, method=1, polarization=Mutual, cutoff=0.7, aEwald=0.0, grid=None, ewaldErrorTolerance=5e-4, mutualInducedTargetEpsilon=1e-5,
                 mutualInducedMaxIterations=60):
        g = np.ascontiguousarray(grid if grid is not None else [0, 0, 0], dtype=np.int32)
        h = amoeba_lib().omm_amoeba_multipole_create(system.h, method, polarization, C.c_double(cutoff), C.c_double(aEwald), _ip(g), C.c_double(ewaldErrorTolerance),
                                                     C.c_double(mutualInducedTargetEpsilon), mutualInducedMaxIterations)
        if not h:
            raise OpenMMError(amoeba_lib().omm_amoeba_last_error().decode())
        self.h = C.c_void_p(h)

    def addMultipoles(self, charge, dipole, quadrupole, axes, thole, damping, polarity):
        """dipole (n, 3), quadrupole (n, 3, 3) in the molecular frame, axes (n, 4) = (axis type, z, x, y atom)"""
        q = np.ascontiguousarray(charge, dtype=np.float64)
        _acheck(amoeba_lib().omm_amoeba_multipole_add(self.h, len(q), _dp(q), _dp(np.ascontiguousarray(dipole, dtype=np.float64).reshape(-1)),
                                                      _dp(np.ascontiguousarray(quadrupole, dtype=np.float64).reshape(-1)), _ip(np.ascontiguousarray(axes, dtype=np.int32).reshape(-1)),
                                                      _dp(np.ascontiguousarray(thole, dtype=np.float64)), _dp(np.ascontiguousarray(damping, dtype=np.float64)),
                                                      _dp(np.ascontiguousarray(polarity, dtype=np.float64))))

    def setCovalentMaps(self, atoms, types, lists):
        """entry e: setCovalentMap(atoms[e], types[e], lists[e])"""
        start = np.zeros(len(lists) + 1, dtype=np.int32)
        start[1:] = np.cumsum([len(l) for l in lists])
        flat = np.ascontiguousarray(np.concatenate([np.asarray(l, dtype=np.int32) for l in lists]) if len(lists) else np.zeros(1, np.int32), dtype=np.int32)
        _acheck(amoeba_lib().omm_amoeba_multipole_set_covalent_maps(self.h, len(lists), _ip(np.ascontiguousarray(atoms, dtype=np.int32)), _ip(np.ascontiguousarray(types, dtype=np.int32)),
                                                                    _ip(start), _ip(flat)))

    def getInducedDipoles(self, context):
        out = np.zeros((context.n, 3))
        _acheck(amoeba_lib().omm_amoeba_multipole_get_induced_dipoles(self.h, context.h, _dp(out)))
        return out


class AmoebaVdwForce:
    NoCutoff, CutoffPeriodic = 0, 1

    def __init__(self, system, sigmaCombiningRule="CUBIC-MEAN", epsilonCombiningRule="HHG", method=1, cutoff=0.9, useDispersionCorrection=True):
        h = amoeba_lib().omm_amoeba_vdw_create(system.h, sigmaCombiningRule.encode(), epsilonCombiningRule.encode(), method, C.c_double(cutoff), int(useDispersionCorrection))
        if not h:
            raise OpenMMError(amoeba_lib().omm_amoeba_last_error().decode())
        self.h = C.c_void_p(h)

    def addParticles(self, parent, sigma, epsilon, reduction):
        p = np.ascontiguousarray(parent, dtype=np.int32)
        _acheck(amoeba_lib().omm_amoeba_vdw_add(self.h, len(p), _ip(p), _dp(np.ascontiguousarray(sigma, dtype=np.float64)), _dp(np.ascontiguousarray(epsilon, dtype=np.float64)),
                                                _dp(np.ascontiguousarray(reduction, dtype=np.float64))))

    def setParticleExclusions(self, lists):
        start = np.zeros(len(lists) + 1, dtype=np.int32)
        start[1:] = np.cumsum([len(l) for l in lists])
        flat = np.ascontiguousarray(np.concatenate([np.asarray(l, dtype=np.int32) for l in lists]), dtype=np.int32)
        _acheck(amoeba_lib().omm_amoeba_vdw_set_exclusions(self.h, len(lists), _ip(start), _ip(flat)))


def addAmoebaGeneralizedKirkwoodForce(system, charge, radius, scale, solventDielectric=78.3, soluteDielectric=1.0, includeCavityTerm=1, probeRadius=0.14, surfaceAreaFactor=-170.351730663):
    """plugins/amoeba/openmmapi/include/openmm/AmoebaGeneralizedKirkwoodForce.h"""
    q, r, sc = (np.ascontiguousarray(a, dtype=np.float64) for a in (charge, radius, scale))
    amoeba_lib().omm_amoeba_gk_create.restype = C.c_void_p
    h = amoeba_lib().omm_amoeba_gk_create(system.h, len(q), _dp(q), _dp(r), _dp(sc), C.c_double(solventDielectric), C.c_double(soluteDielectric), int(includeCavityTerm),
                                          C.c_double(probeRadius), C.c_double(surfaceAreaFactor))
    if not h:
        raise OpenMMError(amoeba_lib().omm_amoeba_last_error().decode())
    return C.c_void_p(h)


def addAmoebaWcaDispersionForce(system, radius, epsilon, epso, epsh, rmino, rminh, awater, slevy, dispoff, shctd):
    """plugins/amoeba/openmmapi/include/openmm/AmoebaWcaDispersionForce.h"""
    r, e = (np.ascontiguousarray(a, dtype=np.float64) for a in (radius, epsilon))
    g = np.ascontiguousarray([epso, epsh, rmino, rminh, awater, slevy, dispoff, shctd], dtype=np.float64)
    amoeba_lib().omm_amoeba_wca_create.restype = C.c_void_p
    h = amoeba_lib().omm_amoeba_wca_create(system.h, len(r), _dp(r), _dp(e), _dp(g))
    if not h:
        raise OpenMMError(amoeba_lib().omm_amoeba_last_error().decode())
    return C.c_void_p(h)


class AmoebaTorsionTorsionForce:
    """plugins/amoeba/openmmapi/include/openmm/AmoebaTorsionTorsionForce.h; created attached to `system`.  atoms [n, 6] = the five chain
    atoms and the chirality marker (-1: none); grids {index: array [nx, ny, 3 or 6]}."""

    def __init__(self, system, atoms, grid_index, grids):
        a = np.ascontiguousarray(atoms, dtype=np.int32).reshape(-1, 6)
        g = np.ascontiguousarray(grid_index, dtype=np.int32)
        h = amoeba_lib().omm_amoeba_torsion_torsion_create(system.h, len(a), _ip(a), _ip(g))
        if not h:
            raise OpenMMError(amoeba_lib().omm_amoeba_last_error().decode())
        self.h = C.c_void_p(h)
        for index, grid in sorted(grids.items()):
            v = np.ascontiguousarray(grid, dtype=np.float64)
            _acheck(amoeba_lib().omm_amoeba_torsion_torsion_set_grid(self.h, int(index), v.shape[0], v.shape[1], v.shape[2], _dp(v)))


class Integrator:
    def __init__(self, kind, stepSize, temperature=300.0, friction=1.0, seed=1, constraintTolerance=1e-5):
        self.h = _handle(lib().omm_integrator_create(kind, C.c_double(stepSize), C.c_double(temperature), C.c_double(friction),
                                                     seed, C.c_double(constraintTolerance)))
        self.stepSize = stepSize

    def step(self, steps):
        _check(lib().omm_integrator_step(self.h, steps))


class CustomIntegrator(Integrator):
    """openmmapi/include/openmm/CustomIntegrator.h: the steps are appended in the order of the calls."""

    def __init__(self, stepSize, seed=1, constraintTolerance=1e-5):
        self.h = _handle(lib().omm_custom_integrator_create(C.c_double(stepSize), seed, C.c_double(constraintTolerance)))
        self.stepSize = stepSize

    def _add(self, kind, name="", expression="", value=0.0):
        _check(lib().omm_custom_integrator_add(self.h, kind, name.encode(), expression.encode(), C.c_double(value)))

    def addGlobalVariable(self, name, value):
        self._add(0, name, value=value)

    def addPerDofVariable(self, name, value):
        self._add(1, name, value=value)

    def addComputeGlobal(self, name, expression):
        self._add(2, name, expression)

    def addComputePerDof(self, name, expression):
        self._add(3, name, expression)

    def addComputeSum(self, name, expression):
        self._add(4, name, expression)

    def addConstrainPositions(self):
        self._add(5)

    def addConstrainVelocities(self):
        self._add(6)

    def addUpdateContextState(self):
        self._add(7)

    def getPerDofVariable(self, index, num_atoms):
        out = np.zeros((num_atoms, 3))
        _check(lib().omm_custom_integrator_get_per_dof(self.h, index, _dp(out)))
        return out

    def getGlobalVariable(self, index):
        out = C.c_double(0.0)
        _check(lib().omm_custom_integrator_get_global(self.h, index, C.byref(out)))
        return out.value

    def setGlobalVariable(self, index, value):
        _check(lib().omm_custom_integrator_set_global(self.h, index, C.c_double(value)))


class MTSIntegrator(CustomIntegrator):
    """wrappers/python/openmm/mtsintegrator.py:30-110 (rRESPA, no thermostat): groups = [(force group, evaluations per step), ...]."""

    def __init__(self, dt, groups, constraintTolerance=1e-5):
        if len(groups) == 0:
            raise ValueError("No force groups specified")
        groups = sorted(groups, key=lambda g: g[1])
        CustomIntegrator.__init__(self, dt, 1, constraintTolerance)
        self.addPerDofVariable("x1", 0)
        self.addUpdateContextState()
        self._substeps(1, groups)
        self.addConstrainVelocities()

    def _substeps(self, parent, groups):
        group, substeps = groups[0]
        per_parent = substeps / parent
        if per_parent < 1 or per_parent != int(per_parent):
            raise ValueError("The number for substeps for each group must be a multiple of the number for the previous group")
        if group < 0 or group > 31:
            raise ValueError("Force group must be between 0 and 31")
        kick = "v+0.5*(dt/%s)*f%s/m" % (substeps, group)
        for _ in range(int(per_parent)):
            self.addComputePerDof("v", kick)
            if len(groups) == 1:
                self.addComputePerDof("x", "x+(dt/%s)*v" % substeps)
                self.addComputePerDof("x1", "x")
                self.addConstrainPositions()
                self.addComputePerDof("v", "v+(x-x1)/(dt/%s)" % substeps)
                self.addConstrainVelocities()
            else:
                self._substeps(substeps, groups[1:])
            self.addComputePerDof("v", kick)


class MTSLangevinIntegrator(CustomIntegrator):
    """wrappers/python/openmm/mtsintegrator.py:112-199 (BAOAB-RESPA): groups = [(force group, evaluations per step), ...].  As there, the
    friction factors a and b are those of the FULL step and are applied once per innermost substep."""

    def __init__(self, temperature, friction, dt, groups, seed=1, constraintTolerance=1e-5):
        if len(groups) == 0:
            raise ValueError("No force groups specified")
        groups = sorted(groups, key=lambda g: g[1])
        CustomIntegrator.__init__(self, dt, seed, constraintTolerance)
        self.addGlobalVariable("a", math.exp(-friction * dt))
        self.addGlobalVariable("b", math.sqrt(1 - math.exp(-2 * friction * dt)))
        self.addGlobalVariable("kT", 8.31446261815324e-3 * temperature)
        self.addPerDofVariable("x1", 0)
        self.addUpdateContextState()
        self._substeps(1, groups)
        self.addConstrainVelocities()

    def _substeps(self, parent, groups):
        group, substeps = groups[0]
        per_parent = substeps / parent
        if per_parent < 1 or per_parent != int(per_parent):
            raise ValueError("The number for substeps for each group must be a multiple of the number for the previous group")
        if group < 0 or group > 31:
            raise ValueError("Force group must be between 0 and 31")
        kick = "v+0.5*(dt/%s)*f%s/m" % (substeps, group)
        for _ in range(int(per_parent)):
            self.addComputePerDof("v", kick)
            if len(groups) == 1:
                self.addComputePerDof("x", "x+(dt/%s)*v" % (2 * substeps))
                self.addComputePerDof("v", "a*v + b*sqrt(kT/m)*gaussian")
                self.addComputePerDof("x", "x+(dt/%s)*v" % (2 * substeps))
                self.addComputePerDof("x1", "x")
                self.addConstrainPositions()
                self.addComputePerDof("v", "v+(x-x1)/(dt/%s)" % substeps)
                self.addConstrainVelocities()
            else:
                self._substeps(substeps, groups[1:])
            self.addComputePerDof("v", kick)


class State:
    pass


class Context:
    def __init__(self, system, integrator, platformName, properties=None):
        props = ";".join("%s=%s" % kv for kv in (properties or {}).items())
        self.system, self.integrator = system, integrator
        self.h = _handle(lib().omm_context_create(system.h, integrator.h, platformName.encode(), props.encode()))
        self.n = system.getNumParticles()

    def close(self):
        if self.h is not None:
            lib().omm_context_destroy(self.h)
            self.h = None

    def getPlatformName(self):
        return lib().omm_context_platform_name(self.h).decode()

    def getPlatformProperty(self, name):
        return lib().omm_context_platform_property(self.h, name.encode()).decode()

    def setPositions(self, positions):
        p = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1)
        _check(lib().omm_context_set_positions(self.h, len(p) // 3, _dp(p)))

    def setVelocities(self, velocities):
        p = np.ascontiguousarray(velocities, dtype=np.float64).reshape(-1)
        _check(lib().omm_context_set_velocities(self.h, len(p) // 3, _dp(p)))

    def setVelocitiesToTemperature(self, temperature, seed=1):
        _check(lib().omm_context_set_velocities_to_temperature(self.h, C.c_double(temperature), seed))

    def getPeriodicBoxVectors(self):
        box = np.zeros(9)
        _check(lib().omm_context_get_box(self.h, _dp(box)))
        return box.reshape(3, 3)

    def setPeriodicBoxVectors(self, a, b, c):
        box = np.ascontiguousarray(np.array([a, b, c], dtype=np.float64).reshape(9))
        _check(lib().omm_context_set_box(self.h, _dp(box)))

    def minimizeEnergy(self, tolerance=10.0, maxIterations=0):
        _check(lib().omm_context_minimize(self.h, C.c_double(tolerance), maxIterations))

    def applyConstraints(self, tol):
        _check(lib().omm_context_apply_constraints(self.h, C.c_double(tol)))

    def createCheckpoint(self):
        blob, size = C.c_void_p(), C.c_long(0)
        _check(lib().omm_context_create_checkpoint(self.h, C.byref(blob), C.byref(size)))
        data = C.string_at(blob, size.value)
        lib().omm_free(blob)
        return data

    def loadCheckpoint(self, data):
        _check(lib().omm_context_load_checkpoint(self.h, C.c_char_p(data), C.c_long(len(data))))

    def applyVelocityConstraints(self, tol):
        _check(lib().omm_context_apply_velocity_constraints(self.h, C.c_double(tol)))

    def getState(self, getPositions=False, getVelocities=False, getForces=False, getEnergy=False, groups=-1):
        flags = (1 if getPositions else 0) | (2 if getVelocities else 0) | (4 if getForces else 0) | (8 if getEnergy else 0)
        pos = np.zeros((self.n, 3)) if getPositions else np.zeros((1, 3))
        vel = np.zeros((self.n, 3)) if getVelocities else np.zeros((1, 3))
        frc = np.zeros((self.n, 3)) if getForces else np.zeros((1, 3))
        en = np.zeros(3)
        _check(lib().omm_context_get_state(self.h, flags, groups, _dp(pos), _dp(vel), _dp(frc), _dp(en)))
        s = State()
        s.positions = pos if getPositions else None
        s.velocities = vel if getVelocities else None
        s.forces = frc if getForces else None
        s.potentialEnergy, s.kineticEnergy, s.time = (en[0], en[1], en[2])
        return s

"""which constraints SETTLE treats.  The HIP platform's own partition (HipConstraints::findSettleClusters,
reached through the plugin's test hook) against the Reference platform's (ReferenceConstraints.cpp:44-148, reached through the
harness) on the benchmark System, the systems of the reference's TestSettle.h / TestVerletIntegrator.h and cases built to hit
every branch of the rule.  No GPU: the function is host code and runs before any device call."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import EMU_BUILD, ROOT
from openmm_amd import harness as H, testsystems as T

# the CPU suite's other tests load the emulated kernel library into the same (xdist worker) process, and the emulated and the product
# builds export the same C ABI: take the emulated twin of the plugin (identical host code) and keep its symbols local
PLUGIN = os.path.join(EMU_BUILD, "libOpenMMHIP.so")
if not os.path.exists(PLUGIN):
    PLUGIN = os.path.join(ROOT, "openmm_amd", "lib", "libOpenMMHIP.so")
needs_plugin = pytest.mark.skipif(not os.path.exists(PLUGIN), reason="plugin missing (run __graft_entry__.build())")


def _clusters(fn, system, capacity):
    atoms = np.full(3 * capacity, -1, dtype=np.int32)
    dist = np.zeros(2 * capacity, dtype=np.float64)
    n = fn(system.h, atoms.ctypes.data_as(C.POINTER(C.c_int)), dist.ctypes.data_as(C.POINTER(C.c_double)), capacity)
    assert 0 <= n <= capacity
    return atoms[:3 * n].reshape(n, 3), dist[:2 * n].reshape(n, 2)


def both_partitions(system, capacity):
    H.lib()                                                    # libOpenMM + the harness, RTLD_GLOBAL
    plugin = C.CDLL(PLUGIN)                                    # dlopen only: no platform is registered, no device is touched
    return _clusters(plugin.ommhip_plugin_settle_clusters, system, capacity), _clusters(H.lib().omm_reference_settle_clusters, system, capacity)


def make_system(masses, constraints):
    s = H.System()
    s.addParticles(masses)
    if len(constraints):
        c = np.array(constraints, dtype=np.float64)
        s.addConstraints(c[:, :2].astype(np.int32), c[:, 2])
    return s


def assert_same(system, capacity, expect=None):
    (a_hip, d_hip), (a_ref, d_ref) = both_partitions(system, capacity)
    assert np.array_equal(a_hip, a_ref), (a_hip, a_ref)
    assert np.array_equal(d_hip, d_ref)                        # bit-identical: both hand on the float-rounded distances
    if expect is not None:
        assert len(a_hip) == expect, a_hip
    return a_hip, d_hip


@needs_plugin
def test_dhfr_waters_are_the_settle_clusters_of_the_reference():
    w = T.dhfr()
    system, _ = w.build()
    atoms, dist = assert_same(system, 8000, expect=7023)
    assert np.all(atoms[:, 0] >= 2489)                          # no protein atom
    assert np.all(np.abs(w.charge[atoms[:, 0]] + 0.834) < 1e-6)  # the oxygen is the central atom
    assert np.all(dist[:, 0] == np.float64(np.float32(0.09572)))


@needs_plugin
def test_systems_of_the_reference_tests():
    # TestSettle.h:44-98: ten waters, O (16) H H (1, 1), constraints O-H 0.1, O-H 0.1, H-H 0.163
    cons = []
    for m in range(10):
        cons += [(3 * m, 3 * m + 1, 0.1), (3 * m, 3 * m + 2, 0.1), (3 * m + 1, 3 * m + 2, 0.163)]
    assert_same(make_system([16.0, 1.0, 1.0] * 10, cons), 32, expect=10)
    # TestVerletIntegrator.h testConstraints: a chain of constraints -- nothing for SETTLE
    n = 8
    assert_same(make_system([10.0] * n, [(i, i + 1, 1.0) for i in range(n - 1)]), 8, expect=0)
    # TestVerletIntegrator.h testConstrainedClusters: a centre with three satellites, and two-atom pairs
    assert_same(make_system([5.0, 1.0, 1.0, 1.0, 3.0, 1.0, 1.0], [(0, 1, 1.0), (0, 2, 1.0), (0, 3, 1.0), (4, 5, 1.2), (4, 6, 1.2)]), 8, expect=0)


@needs_plugin
def test_every_branch_of_the_rule():
    # the central atom may be the lowest, the middle or the highest index of the triangle
    a, d = assert_same(make_system([1.0, 16.0, 1.0], [(0, 1, 0.1), (1, 2, 0.1), (0, 2, 0.16)]), 4, expect=1)
    assert list(a[0]) == [1, 0, 2] and d[0, 1] == np.float64(np.float32(0.16))
    a, d = assert_same(make_system([1.0, 1.0, 16.0], [(0, 1, 0.16), (1, 2, 0.1), (0, 2, 0.1)]), 4, expect=1)
    assert list(a[0]) == [2, 0, 1]
    a, d = assert_same(make_system([16.0, 1.0, 1.0], [(1, 2, 0.16), (0, 2, 0.1), (1, 0, 0.1)]), 4, expect=1)
    assert list(a[0]) == [0, 1, 2]
    # three different sides: left to the general solver
    assert_same(make_system([16.0, 1.0, 1.0], [(0, 1, 0.1), (0, 2, 0.11), (1, 2, 0.16)]), 4, expect=0)
    # equal only as floats (ReferenceConstraints.cpp:76-77,114): still a SETTLE water, and the float value is what is handed on
    a, d = assert_same(make_system([16.0, 1.0, 1.0], [(0, 1, 0.1), (0, 2, 0.1 + 1e-10), (1, 2, 0.16)]), 4, expect=1)
    assert d[0, 0] == np.float64(np.float32(0.1))
    # equilateral: the first test (d12 == d13) wins, the lowest atom is the centre
    a, d = assert_same(make_system([1.0, 1.0, 1.0], [(0, 1, 0.1), (1, 2, 0.1), (0, 2, 0.1)]), 4, expect=1)
    assert list(a[0]) == [0, 1, 2]
    # an open chain of three, a ring of four: no closed triangle
    assert_same(make_system([1.0] * 3, [(0, 1, 0.1), (1, 2, 0.1)]), 4, expect=0)
    assert_same(make_system([1.0] * 4, [(0, 1, 0.1), (1, 2, 0.1), (2, 3, 0.1), (3, 0, 0.1)]), 4, expect=0)
    # a triangle with a fourth atom hanging on it: one corner takes part in three constraints
    assert_same(make_system([1.0] * 4, [(0, 1, 0.1), (1, 2, 0.1), (0, 2, 0.16), (2, 3, 0.1)]), 4, expect=0)
    # the same constraint listed twice: the atom takes part in three, the Reference does not call it a water either
    assert_same(make_system([16.0, 1.0, 1.0], [(0, 1, 0.1), (0, 2, 0.1), (1, 2, 0.16), (1, 2, 0.16)]), 4, expect=0)
    # both constraints of an atom lead to the same partner
    assert_same(make_system([1.0] * 2, [(0, 1, 0.1), (0, 1, 0.1)]), 4, expect=0)
    # constraints between two massless atoms do not count (ReferenceConstraints.cpp:59): the triangle 0-1-2 is still closed by
    # massive ends, the massless pair 3-4 hanging on nothing changes nothing
    assert_same(make_system([16.0, 1.0, 1.0, 0.0, 0.0], [(0, 1, 0.1), (0, 2, 0.1), (1, 2, 0.16), (3, 4, 0.1)]), 4, expect=1)
    # a massless corner: its constraints to massive atoms count
    assert_same(make_system([16.0, 1.0, 0.0], [(0, 1, 0.1), (0, 2, 0.1), (1, 2, 0.16)]), 4, expect=1)
    # two waters and a chain between them, clusters reported by their lowest atom
    cons = [(5, 6, 0.1), (5, 7, 0.1), (6, 7, 0.16), (0, 1, 0.16), (0, 2, 0.1), (1, 2, 0.1), (3, 4, 0.12)]
    a, d = assert_same(make_system([1.0, 1.0, 16.0, 12.0, 1.0, 16.0, 1.0, 1.0], cons), 4, expect=2)
    assert list(a[0]) == [2, 0, 1] and list(a[1]) == [5, 6, 7]


@needs_plugin
def test_random_constraint_graphs_agree_with_the_reference():
    rng = np.random.default_rng(5)
    for trial in range(200):
        n = int(rng.integers(3, 14))
        masses = rng.choice([0.0, 1.0, 12.0, 16.0], size=n, p=[0.1, 0.4, 0.2, 0.3])
        cons = []
        for _ in range(int(rng.integers(0, 2 * n))):
            i, j = rng.choice(n, size=2, replace=False)
            cons.append((int(i), int(j), float(rng.choice([0.1, 0.1 + 1e-10, 0.16, 0.12]))))
        assert_same(make_system(masses, cons), 16)

"""CPU-side checks of the drop-in boundary: the gfx950 kernel library loads and exports every symbol that
include/openmm_hip_kernels.h declares; the plugin exports OpenMM's plugin entry points; and the product
path fails loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADERS = [os.path.join(ROOT, "include", "openmm_hip_kernels.h"), os.path.join(ROOT, "include", "openmm_hip_comm.h"), os.path.join(ROOT, "include", "openmm_hip_amoeba.h")]
LIB = os.path.join(ROOT, "openmm_amd", "lib")


def declared_symbols():
    text = "".join(open(h).read() for h in HEADERS)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ommhip_\w+)\s*\(", text)) - {"ommhip_host_all_gather_fn"})


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert len(syms) >= 40
    for must in ("ommhip_nb_direct", "ommhip_nl_update", "ommhip_pme_reciprocal", "ommhip_integrate_stage", "ommhip_settle",
                 "ommhip_pme_reciprocal_dd", "ommhip_comm_create_rccl", "ommhip_comm_all_gather", "ommhip_comm_all_to_all"):
        assert must in syms


def test_kernel_library_exports_every_declared_symbol():
    lib = C.CDLL(os.path.join(LIB, "libopenmm_hip_kernels.so"))
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, "declared in the header but not exported: %s" % missing


def test_kernel_library_contains_gfx950_code_objects():
    out = subprocess.run(["strings", "-a", os.path.join(LIB, "libopenmm_hip_kernels.so")], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_plugin_exports_openmm_entry_points():
    # olla/include/openmm/PluginInitializer.h:45-57
    C.CDLL(os.path.join(ROOT, "build", "openmm", "lib", "libOpenMM.so"), mode=C.RTLD_GLOBAL)      # the host library the plugin links
    plugin = C.CDLL(os.path.join(LIB, "libOpenMMHIP.so"))
    assert hasattr(plugin, "registerPlatforms")
    assert hasattr(plugin, "registerKernelFactories")


def test_fft_size_rule_needs_no_device():
    lib = C.CDLL(os.path.join(LIB, "libopenmm_hip_kernels.so"))
    ok = [n for n in range(2, 200) if lib.ommhip_fft_supported_size(n)]
    for n in (56, 64, 98, 70, 192, 18, 25, 28, 30):
        assert n in ok
    for n in (17, 19, 22, 23, 121):
        assert n not in ok


def test_product_path_fails_loudly_without_gpu():
    """Context("HIP") must raise when no device is usable; it must never fall back to a CPU path silently."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from openmm_amd import harness as H, testsystems as T, capi\n"
        "import ctypes as C\n"
        "k = capi.load(); n = C.c_int(0)\n"
        "rc = k.lib.ommhip_device_count(C.byref(n))\n"
        "if rc == 0 and n.value > 0: print('HAVE_GPU'); sys.exit(0)\n"
        "H.load_hip_platform()\n"
        "w = T.argon_box(2)\n"
        "s, nb = w.build()\n"
        "try:\n"
        "    H.Context(s, H.Integrator(H.VERLET, 0.001), 'HIP')\n"
        "    print('CREATED')\n"
        "except H.OpenMMError as e:\n"
        "    print('RAISED', e)\n" % ROOT)
    out = subprocess.run(["python", "-c", code], capture_output=True, text=True, timeout=300).stdout
    if "HAVE_GPU" in out:
        pytest.skip("a GPU is present")
    assert "RAISED" in out and "CREATED" not in out, out

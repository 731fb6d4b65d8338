"""Write a benchmark System as an XML file the reference's `XmlSerializer::deserialize<System>` loads.

    python tools/system_to_xml.py dhfr      out.xml     # examples/benchmark.py `pme`: fixture tests/golden/dhfr_5dfr_amber99sb_tip3p.npz
    python tools/system_to_xml.py lysozyme  out.xml     # TestForceField.py test_Forces: needs /root/reference (PDB + force-field files)
    python tools/system_to_xml.py apoa1     out.xml     # the apoa1-sized water box of bench.py

A C++ client then does   std::ifstream in("out.xml"); System* system = XmlSerializer::deserialize<System>(in);   and runs it on any platform,
"HIP" included; openmm_amd/harness.py::System.from_xml does the same through the harness.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from openmm_amd import system_xml, testsystems as T
    which, out = sys.argv[1], sys.argv[2]
    if which == "dhfr":
        w = T.dhfr()
    elif which == "lysozyme":
        from openmm_amd import forcefield as FF
        w = FF.lysozyme_implicit()
    elif which == "apoa1":
        w = T.apoa1_like()
    else:
        raise SystemExit("unknown system '%s' (dhfr, lysozyme, apoa1)" % which)
    text = system_xml.workload_to_xml(w)
    with open(out, "w") as f:
        f.write(text)
    print("wrote %s: %d atoms, %.1f MB" % (out, w.num_atoms, len(text) / 1e6))


if __name__ == "__main__":
    main()

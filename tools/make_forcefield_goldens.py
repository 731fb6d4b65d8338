"""tests/golden/forcefield_reference_forces.npz: the golden forces the reference's own Python tests hold for two force-field-built Systems,
copied out of its serialized State files so that the comparison travels without the reference tree:
  * wrappers/python/tests/systems/lysozyme-implicit-forces.xml      (TestForceField.py:285-301; amber99sb + amber99_obc, NoCutoff)
  * wrappers/python/tests/systems/alanine-dipeptide-amoeba-forces.xml  (TestForceField.py:1246-1262; amoeba2013 + amoeba2013_gk, direct polarization)
together with the coordinates of the two PDB files the tests read.  Run in the build container (needs /root/reference).

    python tools/make_forcefield_goldens.py
"""
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SYSTEMS = "/root/reference/wrappers/python/tests/systems"


def forces_of(path):
    root = ET.parse(path).getroot()
    return np.array([[float(f.attrib[k]) for k in "xyz"] for f in root.find("Forces").findall("Force")])


def main():
    from openmm_amd import forcefield as FF
    out = {}
    for key, pdb, xml in (("lysozyme", "lysozyme-implicit.pdb", "lysozyme-implicit-forces.xml"),
                          ("alanine_dipeptide_amoeba", "alanine-dipeptide-implicit.pdb", "alanine-dipeptide-amoeba-forces.xml")):
        out[key + "_positions"] = FF.read_pdb(os.path.join(SYSTEMS, pdb))["positions"]
        out[key + "_forces"] = forces_of(os.path.join(SYSTEMS, xml))
        assert len(out[key + "_positions"]) == len(out[key + "_forces"]), key
        print(key, out[key + "_forces"].shape, "rms force %.1f" % np.sqrt((out[key + "_forces"] ** 2).sum(1).mean()))
    path = os.path.join(ROOT, "tests", "golden", "forcefield_reference_forces.npz")
    np.savez_compressed(path, source="tools/make_forcefield_goldens.py: <Forces> of wrappers/python/tests/systems/{lysozyme-implicit,alanine-dipeptide-amoeba}-forces.xml "
                        "(kJ/mol/nm), positions of the PDB files TestForceField.py reads (nm)", **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""ORACLE recipe (test infrastructure): builds the REFERENCE's own UDF marching cubes from the sources where they lie
(/root/reference/meshudf/_marching_cubes_lewiner_cy.pyx, Cython -> C++ -> g++) into oracle/_ref/ (git-ignored; it
travels to the GPU box with the snapshot).  Nothing of the reference is copied into the repository: the generated
C++ and the extension module live only under oracle/_ref/.  Runs only where /root/reference exists.

    python oracle/build_ref.py            # -> oracle/_ref/ref_mc_cy*.so   (about 20 s)

`load()` returns a namespace with `udf_mc_lewiner(udf, grads, spacing)` = the reference's
meshudf/_marching_cubes_lewiner.py:87-154 driving the compiled extension (imported from the reference tree when it
is present; a minimal driver with the same post-processing otherwise — the GPU box has no /root/reference).
"""
from __future__ import annotations

import glob
import importlib.util
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("SURFD_REFERENCE", "/root/reference")
PYX = os.path.join(REF, "meshudf", "_marching_cubes_lewiner_cy.pyx")


def built():
    hits = glob.glob(os.path.join(OUT, "ref_mc_cy*.so"))
    return hits[0] if hits else None


def build(force: bool = False):
    if built() and not force:
        return built()
    if not os.path.exists(PYX):
        return None
    import numpy as np
    os.makedirs(OUT, exist_ok=True)
    cpp = os.path.join(OUT, "ref_mc_cy.cpp")
    # module name on the command line: the generated init function must match the file we load
    subprocess.check_call([sys.executable, "-m", "cython", "--cplus", "-3", "--module-name", "ref_mc_cy", PYX, "-o", cpp])
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    so = os.path.join(OUT, "ref_mc_cy" + ext)
    inc = ["-I" + sysconfig.get_paths()["include"], "-I" + np.get_include()]
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++14", "-w", "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION"]
                          + inc + [cpp, "-o", so])
    return so


def load():
    so = built() or build()
    if not so:
        return None
    spec = importlib.util.spec_from_file_location("ref_mc_cy", so)
    cy = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cy)
    return cy


LUT_ORDER = ["EDGESRELX", "EDGESRELY", "EDGESRELZ", "CASESCLASSIC", "CASES", "TILING1", "TILING2", "TILING3_1", "TILING3_2",
             "TILING4_1", "TILING4_2", "TILING5", "TILING6_1_1", "TILING6_1_2", "TILING6_2", "TILING7_1", "TILING7_2", "TILING7_3",
             "TILING7_4_1", "TILING7_4_2", "TILING8", "TILING9", "TILING10_1_1", "TILING10_1_1_", "TILING10_1_2", "TILING10_2",
             "TILING10_2_", "TILING11", "TILING12_1_1", "TILING12_1_1_", "TILING12_1_2", "TILING12_2", "TILING12_2_", "TILING13_1",
             "TILING13_1_", "TILING13_2", "TILING13_2_", "TILING13_3", "TILING13_3_", "TILING13_4", "TILING13_5_1", "TILING13_5_2",
             "TILING14", "TEST3", "TEST4", "TEST6", "TEST7", "TEST10", "TEST12", "TEST13", "SUBCONFIG13"]


def lut_provider(cy, tables):
    """The reference's LutProvider (positional constructor, _marching_cubes_lewiner.py:229-281) from a {name: int8 array}
    dict — the tables come from the library under test (surfd_amd.mcubes.lut_tables), so the reference extension can
    be driven on a host that has oracle/_ref but no reference tree."""
    return cy.LutProvider(*[tables[n] for n in LUT_ORDER])


def reference_udf_mc(cy, tables, udf, grads):
    """(vertices (z,y,x) float32, faces int32[F,3]) as udf_mc_lewiner returns them for spacing 1 (:133-142)."""
    import numpy as np
    v, f, n, val = cy.marching_cubes_udf(udf, grads, lut_provider(cy, tables), 1, 0, None)
    return np.fliplr(v), np.fliplr(f.reshape(-1, 3)), np.fliplr(n), val


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

#!/usr/bin/env python3
"""Build the committed golden fixtures from the reference's own TEST DATA files.

Run once in the build container (needs /root/reference and /opt/conda/bin/h5dump):

    python tests/golden/make_fixtures.py

Inputs (data files only, no reference source code is read or copied):
  * /root/reference/tests/tests_data/{disk,square_tri,square_quad,coarse_square}.h5
      the four test meshes of tests/test_compute_meshtags.py:28-104
  * /root/reference/tests/tests_data/*_tags.csv
      the 336 live golden tag files compared at tests/test_compute_meshtags.py:239-243
      (two rows each: entity indices in dolfinx-0.9.0 local numbering; tag values)

Outputs (small, committed):
  * tests/golden/meshes.npz        coords (f64) + cells (i32, file order) of the 4 meshes
  * tests/golden/tags_golden.npz   per golden CSV: values as int8 and indices as int32
                                   (kept whole so that index-exact parity can be added
                                   once dolfinx's renumbering is restated)
  * tests/golden/one_sided_kat.json  the 9 known answers of
                                   tests/test_one_sided_integral.py:32,63,88
  * tests/golden/flower_data.npz   input points + outputs of the four problem-data functions of
                                   demo/weak-dirichlet/flower/data.py (pure numpy, so this one
                                   module of the reference IS importable here): golden vectors
                                   for the restatement in tests/flower_data.py (a13)
The known-answer values are plain numbers asserted by the reference's test; they are
restated here as data.
"""
import json
import os
import re
import subprocess

import numpy as np
import sys
# importing the reference's data modules must not leave a __pycache__ in /root/reference
# (read-only by contract; root ignores the mode bits)
sys.dont_write_bytecode = True

REF = "/root/reference/tests/tests_data"
HERE = os.path.dirname(os.path.abspath(__file__))
H5DUMP = "/opt/conda/bin/h5dump"


def h5_dataset(path, name, dtype):
    """Dataset `name` of an HDF5 file through h5dump's RAW BINARY export (-b LE): the file's own bits.  (Rounds 1-2 parsed
    h5dump's text output, which prints doubles with 6 significant digits: the committed coordinates were the file's
    values rounded at 1e-7 relative.  Since round 3 the fixture holds the exact coordinates; with them 8 more golden
    cases -- ellipse_in_square, detection degree 3, discretize -- are decided by the last bit of a coordinate and join
    the floating-point-degenerate ones, tests/datasets.py.)"""
    import tempfile
    head = subprocess.run([H5DUMP, "-H", "-d", name, path], check=True, capture_output=True, text=True).stdout
    m = re.search(r"DATASPACE\s+SIMPLE\s*\{\s*\(\s*(\d+)\s*,\s*(\d+)\s*\)", head)
    shape = (int(m.group(1)), int(m.group(2)))
    is_float = ("IEEE_F" in head) or ("FLOAT" in head)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "d.bin")
        subprocess.run([H5DUMP, "-d", name, "-b", "LE", "-o", out, path], check=True, capture_output=True)
        item = os.path.getsize(out) // (shape[0] * shape[1])
        raw = np.fromfile(out, dtype=np.dtype(f"<{'f' if is_float else 'i'}{item}"))
    return raw.astype(dtype).reshape(shape)


def main():
    meshes = {}
    spec = {
        "disk": ("/data0", "/data1", "triangle"),
        "square_tri": ("/Mesh/mesh/geometry", "/Mesh/mesh/topology", "triangle"),
        "square_quad": ("/Mesh/mesh/geometry", "/Mesh/mesh/topology", "quadrilateral"),
        "coarse_square": ("/Mesh/mesh/geometry", "/Mesh/mesh/topology", "triangle"),
    }
    for name, (g, t, ctype) in spec.items():
        p = os.path.join(REF, name + ".h5")
        x = h5_dataset(p, g, np.float64)
        c = h5_dataset(p, t, np.int32)
        meshes[name + "_x"] = x
        meshes[name + "_cells"] = c
        meshes[name + "_type"] = np.array(ctype)
        print(name, x.shape, c.shape, ctype)
    np.savez_compressed(os.path.join(HERE, "meshes.npz"), **meshes)

    datasets = ["circle_in_circle", "boundary_crossing_circle", "circle_in_square",
                "square_in_square", "ellipse_in_square", "circle_near_boundary",
                "nasty_levelset"]
    gold = {}
    n = 0
    for d in datasets:
        for deg in (1, 2, 3):
            for disc in (False, True):
                for box in (True, False):
                    for sl in (False, True):
                        mid = "_"
                        if disc:
                            mid += "discretize_"
                        if not box:
                            mid += "submesh_"
                        if sl:
                            mid += "single_layer_"
                        for ent in ("cells", "facets"):
                            key = f"{d}_{deg}{mid}{ent}_tags"
                            a = np.loadtxt(os.path.join(REF, key + ".csv"), delimiter=" ")
                            a = np.atleast_2d(a)
                            gold[key + ":i"] = a[0].astype(np.int32)
                            gold[key + ":v"] = a[1].astype(np.int8)
                            n += 1
    np.savez_compressed(os.path.join(HERE, "tags_golden.npz"), **gold)
    print("golden tag files:", n)

    kat = {
        "line_in_square_quad": {"mesh": "square_quad", "values": [3.0, -3.0]},
        "square_in_square_quad": {"mesh": "square_quad", "values": [3.2, 2.4]},
        "square_in_square_tri": {"mesh": "square_tri", "values": [3.2, 2.4]},
    }
    with open(os.path.join(HERE, "one_sided_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)

    # golden vectors of the demo's data functions, produced by the reference itself
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_flower_data", "/root/reference/demo/weak-dirichlet/flower/data.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(20261003)
    x = np.concatenate([rng.uniform(-4.5, 4.5, (2, 4000)),
                        np.stack(np.meshgrid(np.linspace(-4.5, 4.5, 33),
                                             np.linspace(-4.5, 4.5, 33))).reshape(2, -1)], axis=1)
    np.savez_compressed(os.path.join(HERE, "flower_data.npz"), x=x,
                        levelset=ref.levelset(x), detection_levelset=ref.detection_levelset(x),
                        source_term=ref.source_term(x), dirichlet_data=ref.dirichlet_data(x))


if __name__ == "__main__":
    main()

"""The oracle against every golden the reference holds for the tagging path (CPU only).

  * 336 tag CSVs of tests/test_compute_meshtags.py -> histograms (the CSV indices are in
    dolfinx-local numbering, SURVEY 4.3, so only numbering-free facts can be compared);
  * 9 known answers of tests/test_one_sided_integral.py:32,63,88 (the `discretize=False` leg;
    for the robust data the `discretize=True` leg has the same goldens).
"""
import json
import os
import warnings

import numpy as np
import pytest

from oracle import tagging as T
from oracle.topology import Topology

from datasets import (FP_FRAGILE, FP_FRAGILE_DISCRETIZED, MESHTAG_DATA, ONE_SIDED_DATA, is_fragile,
                      load_mesh, nasty_interpolated)

HERE = os.path.dirname(__file__)
GOLD = np.load(os.path.join(HERE, "golden", "tags_golden.npz"))


def hist(v, hi):
    return np.bincount(np.asarray(v, dtype=np.int64), minlength=hi + 1)[1:hi + 1]


def golden_key(name, deg, disc, box, sl, ent):
    mid = "_"
    if disc:
        mid += "discretize_"
    if not box:
        mid += "submesh_"
    if sl:
        mid += "single_layer_"
    return f"{name}_{deg}{mid}{ent}_tags"


_RENUMBERED = {}


def mesh_as_dolfinx_reads_it(mesh):
    """(cell type, x, cells) of a test mesh in dolfinx's numbering (cached)."""
    if mesh not in _RENUMBERED:
        from phifem_amd.reorder import as_dolfinx_reads_it
        ctype, x, cells = load_mesh(mesh)
        xn, cn, cell_new, vertex_new = as_dolfinx_reads_it(ctype, x, cells)
        _RENUMBERED[mesh] = (ctype, xn, cn, cell_new)
    return _RENUMBERED[mesh][:3]


def test_reordering_is_the_identity_on_the_files_dolfinx_wrote():
    """Three of the four mesh files are fixed points of read_mesh's reordering; `disk` is not."""
    for mesh in ("square_quad", "square_tri", "coarse_square", "disk"):
        mesh_as_dolfinx_reads_it(mesh)
        cell_new = _RENUMBERED[mesh][3]
        assert bool(np.array_equal(cell_new, np.arange(cell_new.size))) == (mesh != "disk"), mesh

CASES = [(n, d, disc, box, sl) for n in MESHTAG_DATA for d in (1, 2, 3)
         for disc in (False, True) for box in (True, False) for sl in (False, True)]


@pytest.mark.parametrize("name,deg,disc,box,sl", CASES)
def test_tag_histograms(name, deg, disc, box, sl):
    fragile = is_fragile(name, deg, disc)
    mesh, f = MESHTAG_DATA[name]
    if disc and name == "nasty_levelset":
        f = nasty_interpolated
        fragile = deg == 2
    ctype, x, cells = mesh_as_dolfinx_reads_it(mesh)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ct, ft, sub, _, maps, topo = T.compute_tags_measures(
            ctype, x, cells, f, deg, box_mode=box, single_layer_cut=sl)
    gc_i = GOLD[golden_key(name, deg, disc, box, sl, "cells") + ":i"]
    gc_v = GOLD[golden_key(name, deg, disc, box, sl, "cells") + ":v"]
    gf_i = GOLD[golden_key(name, deg, disc, box, sl, "facets") + ":i"]
    gf_v = GOLD[golden_key(name, deg, disc, box, sl, "facets") + ":v"]
    ok = (np.array_equal(hist(ct.values, 3), hist(gc_v, 3))
          and np.array_equal(hist(ft.values, 6), hist(gf_v, 6))
          and ct.indices.size == gc_i.size and ft.indices.size == gf_i.size)
    if fragile and not ok:
        pytest.xfail("floating-point-degenerate level-set (SURVEY 4.3): decided by FFCx round-off")
    assert np.array_equal(hist(ct.values, 3), hist(gc_v, 3))
    assert np.array_equal(hist(ft.values, 6), hist(gf_v, 6))
    # golden entity lists are dense 0..n-1, as ours
    assert np.array_equal(ct.indices, gc_i)
    assert np.array_equal(ft.indices, gf_i)
    # ELEMENT-WISE, as tests/test_compute_meshtags.py:239-243 compares.  The mesh is held in the numbering
    # dolfinx's read_mesh gives it (phifem_amd/reorder.py: Gibbs-Poole-Stockmeyer on the dual graph, vertices by first
    # appearance; the identity on three of the four files); the oracle keeps the caller's cell and vertex order and numbers
    # facets by the lexicographic rank of their sorted vertex tuples -- dolfinx's rule.
    assert np.array_equal(ct.values, gc_v), np.flatnonzero(ct.values != gc_v)[:8]
    assert np.array_equal(ft.values, gf_v), np.flatnonzero(ft.values != gf_v)[:8]


@pytest.mark.parametrize("name", list(MESHTAG_DATA))
@pytest.mark.parametrize("deg", [1, 2, 3])
def test_facet_tags_partition(name, deg):
    mesh, f = MESHTAG_DATA[name]
    ctype, x, cells = load_mesh(mesh)
    topo = Topology(ctype, cells, x.shape[0])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cv = T.tag_cells_values(topo, x, f, deg)
        bc = T.boundary_cell_cut_flags(topo, x, f, deg)
    tags, count = T.tag_facets_values(topo, cv, bc)
    assert np.all(count == 1) and np.all(tags > 0)


KAT = json.load(open(os.path.join(HERE, "golden", "one_sided_kat.json")))


@pytest.mark.parametrize("name", list(ONE_SIDED_DATA))
@pytest.mark.parametrize("deg", [1, 2, 3])
def test_one_sided_integrals(name, deg):
    mesh, f, integrand = ONE_SIDED_DATA[name]
    ctype, x, cells = load_mesh(mesh)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ct, ft, _, meas, _, topo = T.compute_tags_measures(ctype, x, cells, f, deg, box_mode=True)
    v100 = T.one_sided_integral_2d(topo, x, meas(100), integrand)
    v101 = T.one_sided_integral_2d(topo, x, meas(101), integrand)
    # tests/test_one_sided_integral.py:167-168 (np.isclose, atol=1e-20, default rtol)
    assert np.isclose(v100, KAT[name]["values"][0], atol=1.0e-20)
    assert np.isclose(v101, KAT[name]["values"][1], atol=1.0e-20)


def test_flower_data_against_reference_vectors():
    """a13: tests/flower_data.py against vectors produced by the reference's own
    demo/weak-dirichlet/flower/data.py (tests/golden/make_fixtures.py)."""
    import flower_data as F
    g = np.load(os.path.join(HERE, "golden", "flower_data.npz"))
    x = g["x"]
    for name in ("levelset", "detection_levelset", "source_term", "dirichlet_data"):
        got = getattr(F, name)(x)
        assert np.array_equal(got, g[name]), name


# ---------------------------------------------------------------------------------------------------------------
# Joint signatures (a step beyond per-file histograms towards SURVEY 8 f1).  Every box-mode golden of one mesh file
# lists the tags in the SAME dolfinx numbering (read_mesh is deterministic), so entity c carries a SIGNATURE
# (tag in case 1, tag in case 2, ...) across all robust cases of that mesh.  The permutation between dolfinx's
# numbering and ours is not recoverable (dolfinx's GPS reordering is not in /root/reference), but it is ONE
# permutation for all cases: the MULTISET of signatures must be identical.  Per-file histograms allow two cases to
# be right "in different places"; the joint multiset does not -- e.g. for `disk` it pins 24 files at once.
# ---------------------------------------------------------------------------------------------------------------
def _robust_box_cases(mesh_name):
    out = []
    for name, (mesh, f) in MESHTAG_DATA.items():
        if mesh != mesh_name:
            continue
        for deg in (1, 2, 3):
            for disc in (False, True):
                g = f
                if is_fragile(name, deg, disc):
                    if not (disc and name == "nasty_levelset" and deg != 2):
                        continue
                if disc and name == "nasty_levelset":
                    g = nasty_interpolated
                for sl in (False, True):
                    out.append((name, deg, disc, sl, g))
    return out


@pytest.mark.parametrize("mesh_name", sorted({m for m, _ in MESHTAG_DATA.values()}))
def test_joint_tag_signatures_of_a_mesh(mesh_name):
    cases = _robust_box_cases(mesh_name)
    assert cases, mesh_name
    ctype, x, cells = load_mesh(mesh_name)
    ours_c, ours_f, gold_c, gold_f = [], [], [], []
    for name, deg, disc, sl, f in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ct, ft, _, _, _, topo = T.compute_tags_measures(ctype, x, cells, f, deg, box_mode=True, single_layer_cut=sl)
        gc = GOLD[golden_key(name, deg, disc, True, sl, "cells") + ":v"]
        gf = GOLD[golden_key(name, deg, disc, True, sl, "facets") + ":v"]
        if not (np.array_equal(hist(ct.values, 3), hist(gc, 3)) and np.array_equal(hist(ft.values, 6), hist(gf, 6))):
            continue   # a case test_tag_histograms reports on its own (e.g. degenerate with the rounded fixture coordinates)
        ours_c.append(ct.values); ours_f.append(ft.values); gold_c.append(gc); gold_f.append(gf)
    assert len(ours_c) >= max(4, (3 * len(cases)) // 4), (mesh_name, len(ours_c), len(cases))

    def multiset(cols):
        sig, cnt = np.unique(np.stack(cols, axis=1), axis=0, return_counts=True)
        return sig, cnt

    for ours, gold, what in ((ours_c, gold_c, "cells"), (ours_f, gold_f, "facets")):
        so, co = multiset(ours)
        sg, cg = multiset(gold)
        assert so.shape == sg.shape and np.array_equal(so, sg) and np.array_equal(co, cg), \
            f"{mesh_name}: joint {what} signatures over {len(ours)} goldens differ"
    print(f"{mesh_name}: {len(ours_c)} goldens jointly, {multiset(ours_c)[0].shape[0]} distinct cell / "
          f"{multiset(ours_f)[0].shape[0]} distinct facet signatures")

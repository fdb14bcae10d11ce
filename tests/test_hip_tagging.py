"""GPU parity of the tagging path (a1-a8) through the C ABI: HIP vs the numpy oracle, bit-exact,
and HIP vs the reference's committed goldens."""
import json
import os
import warnings

import numpy as np
import pytest

from oracle import meshgen
from oracle import tagging as T
from oracle.topology import Topology

from datasets import (FP_FRAGILE, FP_FRAGILE_DISCRETIZED, MESHTAG_DATA, ONE_SIDED_DATA, is_fragile,
                      load_mesh)

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)


@pytest.fixture(scope="module")
def P():
    import phifem_amd
    assert phifem_amd._lib.device_count() > 0, "no GPU: the HIP path cannot run"
    return phifem_amd


_mesh_cache = {}


def get_mesh(P, name):
    """The test mesh in the numbering dolfinx's read_mesh gives it (phifem_amd/reorder.py; the identity on the three
    files dolfinx wrote, a Gibbs-Poole-Stockmeyer reordering of `disk`): tag arrays then compare with the reference's
    golden files index by index."""
    if name not in _mesh_cache:
        from phifem_amd.reorder import as_dolfinx_reads_it
        ctype, x, cells = load_mesh(name)
        x, cells = as_dolfinx_reads_it(ctype, x, cells)[:2]
        m = P.Mesh.from_arrays(ctype, x, cells)
        topo = Topology(ctype, cells, x.shape[0])
        _mesh_cache[name] = (m, topo, x)
    return _mesh_cache[name]


def test_topology_upload_matches_oracle(P):
    for name in ("disk", "square_tri", "square_quad", "coarse_square"):
        m, topo, x = get_mesh(P, name)
        assert m.nf == topo.nf and m.nbf == topo.boundary_facets.size
        assert np.array_equal(m.c2f, topo.c2f)
        assert np.array_equal(m.f2c, topo.f2c)
        bf = m.boundary_facets
        assert np.array_equal(topo.c2f[bf[:, 0], bf[:, 1]], topo.boundary_facets)


def levelset_variants(P, f, x):
    from phifem_amd.mesh_scripts import NodalFunction, Quadric
    out = [("callable", f, f)]
    with np.errstate(all="ignore"):
        nod = np.asarray(f(x.T), dtype=np.float64)
    if np.all(np.isfinite(nod)):
        out.append(("nodal", NodalFunction(nod), T.NodalP1(nod)))
    if hasattr(f, "quadric"):
        x0, a, x1, b, c = f.quadric
        out.append(("quadric", Quadric([x0, x1], [a, b], c), f))
    return out


@pytest.mark.parametrize("name", list(MESHTAG_DATA))
@pytest.mark.parametrize("deg", [1, 2, 3])
@pytest.mark.parametrize("sl", [False, True])
def test_tags_bit_exact_vs_oracle(P, name, deg, sl):
    mesh_name, f = MESHTAG_DATA[name]
    m, topo, x = get_mesh(P, mesh_name)
    for label, ls_hip, ls_ora in levelset_variants(P, f, x):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                oc, of, _, omeas, _, _ = T.compute_tags_measures(
                    topo.cell_type, x, topo, ls_ora, deg, box_mode=True, single_layer_cut=sl)
            except ValueError:
                with pytest.raises(ValueError):
                    P.compute_tags_measures(m, ls_hip, deg, box_mode=True, single_layer_cut=sl)
                continue
            hc, hf, sub, hmeas, maps = P.compute_tags_measures(
                m, ls_hip, deg, box_mode=True, single_layer_cut=sl)
        assert sub is None and maps is None
        assert np.array_equal(hc.indices, oc.indices), label
        assert np.array_equal(hc.values, oc.values), label
        assert np.array_equal(hf.indices, of.indices), label
        assert np.array_equal(hf.values, of.values), label
        assert hc.values.dtype == np.int32 and hc.indices.dtype == np.int32
        for tag in (100, 101):
            assert np.array_equal(hmeas(tag), omeas(tag)), (label, tag)
        assert np.array_equal(hc.find(2), oc.find(2))


@pytest.mark.parametrize("name", ["circle_in_circle", "boundary_crossing_circle",
                                  "ellipse_in_square", "circle_near_boundary"])
@pytest.mark.parametrize("deg", [1, 3])
def test_submesh_mode_vs_oracle(P, name, deg):
    mesh_name, f = MESHTAG_DATA[name]
    m, topo, x = get_mesh(P, mesh_name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        oc, of, osub, _, omaps, _ = T.compute_tags_measures(
            topo.cell_type, x, topo, f, deg, box_mode=False, single_layer_cut=True)
        hc, hf, hsub, hmeas, hmaps = P.compute_tags_measures(
            m, f, deg, box_mode=False, single_layer_cut=True)
    assert np.array_equal(hmaps[0], omaps[0]) and np.array_equal(hmaps[1], omaps[1])
    assert np.array_equal(hsub.cells, osub.topology.cells)
    assert np.array_equal(hsub.x, osub.x)
    assert np.array_equal(hsub.c2f, osub.topology.c2f)
    assert np.array_equal(hc.values, oc.values) and np.array_equal(hc.indices, oc.indices)
    assert np.array_equal(hf.values, of.values) and np.array_equal(hf.indices, of.indices)


# ----------------------------------------------------------------------------------------------
GOLD = np.load(os.path.join(HERE, "golden", "tags_golden.npz"))


def hist(v, hi):
    return np.bincount(np.asarray(v, dtype=np.int64), minlength=hi + 1)[1:hi + 1]


@pytest.mark.parametrize("name", list(MESHTAG_DATA))
@pytest.mark.parametrize("deg", [1, 2, 3])
@pytest.mark.parametrize("box", [True, False])
@pytest.mark.parametrize("sl", [False, True])
def test_hip_vs_reference_goldens(P, name, deg, box, sl):
    """tests/test_compute_meshtags.py:239-243 on the HIP path (histograms: SURVEY 4.3)."""
    mesh_name, f = MESHTAG_DATA[name]
    m, topo, x = get_mesh(P, mesh_name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hc, hf = P.compute_tags_measures(m, f, deg, box_mode=box, single_layer_cut=sl)[:2]
    missed = []
    for disc in (False, True):
        mid = "_" + ("discretize_" if disc else "") + ("" if box else "submesh_") + \
            ("single_layer_" if sl else "")
        gc = GOLD[f"{name}_{deg}{mid}cells_tags:v"]
        gf = GOLD[f"{name}_{deg}{mid}facets_tags:v"]
        ok = np.array_equal(hist(hc.values, 3), hist(gc, 3)) and \
            np.array_equal(hist(hf.values, 6), hist(gf, 6))
        if is_fragile(name, deg, disc) or (disc and name == "nasty_levelset"):
            # decided by FFCx / basix round-off [3P] (the discretised `nasty` leg interpolates a NaN, see
            # datasets.nasty_interpolated and the oracle test): REPORTED as an expected failure, never skipped silently
            if not ok:
                missed.append(f"{name}_{deg}{mid}: cells {tuple(hist(hc.values, 3))} / golden {tuple(hist(gc, 3))}")
            continue
        assert ok, (name, deg, box, sl, disc)
        assert np.array_equal(hc.indices, GOLD[f"{name}_{deg}{mid}cells_tags:i"])
        assert np.array_equal(hf.indices, GOLD[f"{name}_{deg}{mid}facets_tags:i"])
        # ELEMENT-WISE equality with the reference's golden files, as tests/test_compute_meshtags.py:239-243
        # compares (the mesh is held in dolfinx's numbering, see get_mesh)
        assert np.array_equal(hc.values, gc), (name, deg, box, sl, disc, np.flatnonzero(hc.values != gc)[:8])
        assert np.array_equal(hf.values, gf), (name, deg, box, sl, disc, np.flatnonzero(hf.values != gf)[:8])
    if missed:
        pytest.xfail("floating-point-degenerate level-set (SURVEY 4.3), decided by FFCx round-off: " + "; ".join(missed))


KAT = json.load(open(os.path.join(HERE, "golden", "one_sided_kat.json")))


@pytest.mark.parametrize("name", list(ONE_SIDED_DATA))
@pytest.mark.parametrize("deg", [1, 2, 3])
def test_one_sided_integrals_hip(P, name, deg):
    """tests/test_one_sided_integral.py:135-168 with the HIP integration entities."""
    mesh_name, f, integrand = ONE_SIDED_DATA[name]
    m, topo, x = get_mesh(P, mesh_name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, _, _, meas, _ = P.compute_tags_measures(m, f, deg, box_mode=True)
    v100 = T.one_sided_integral_2d(topo, x, meas(100), integrand)
    v101 = T.one_sided_integral_2d(topo, x, meas(101), integrand)
    assert np.isclose(v100, KAT[name]["values"][0], atol=1.0e-20)
    assert np.isclose(v101, KAT[name]["values"][1], atol=1.0e-20)


# ----------------------------------------------------------------------------------------------
def facet_keys(cells, c2f, nf, fv):
    keys = np.zeros((nf, fv.shape[1]), dtype=np.int64)
    for lf in range(fv.shape[0]):
        keys[c2f[:, lf]] = np.sort(cells[:, fv[lf]], axis=1)
    return keys


@pytest.mark.parametrize("d,n", [(2, (7, 5)), (3, (4, 3, 5))])
def test_device_box_generator(P, d, n):
    from oracle.points import FACET_VERTS
    lo, hi = [-1.5, -1.0, -0.5][:d], [1.5, 2.0, 1.0][:d]
    m = P.create_box(lo, hi, n)
    xo, co = meshgen.create_box(lo, hi, n)
    assert np.array_equal(m.x, xo)
    assert np.array_equal(m.cells, co)
    ctype = "triangle" if d == 2 else "tetrahedron"
    topo = Topology(ctype, co, xo.shape[0])
    assert m.nf == topo.nf and m.nbf == topo.boundary_facets.size
    # numbering-free comparison through the facets' sorted vertex tuples
    fv = FACET_VERTS[ctype]
    hk = facet_keys(m.cells.astype(np.int64), m.c2f, m.nf, fv)
    assert np.unique(hk, axis=0).shape[0] == m.nf
    okm = topo.facet_key_map()
    to_oracle = np.array([okm[tuple(k)] for k in hk])
    assert np.array_equal(to_oracle[m.c2f], topo.c2f)
    assert np.array_equal(m.f2c, topo.f2c[to_oracle])
    bf = m.boundary_facets
    assert np.array_equal(np.sort(to_oracle[m.c2f[bf[:, 0], bf[:, 1]]]), topo.boundary_facets)


def test_box_slab_coordinates_are_bit_identical(P):
    full = P.create_box([-1.5] * 3, [1.5] * 3, [6, 5, 8])
    slab = P.create_box([-1.5] * 3, [1.5] * 3, [6, 5, 3], offset=[0, 0, 2], n_global=[6, 5, 8])
    xf = full.x.reshape(9, 6, 7, 3)
    assert np.array_equal(slab.x.reshape(4, 6, 7, 3), xf[2:6])


@pytest.mark.parametrize("d,n,deg", [(2, 24, 1), (2, 16, 3), (3, 10, 1), (3, 6, 2), (3, 5, 3)])
@pytest.mark.parametrize("mode", ["nodal", "quadric"])
def test_box_tags_vs_oracle(P, d, n, deg, mode):
    from oracle.points import FACET_VERTS
    from phifem_amd.mesh_scripts import NodalFunction, Quadric
    lo, hi = [-1.5] * d, [1.5] * d
    m = P.create_box(lo, hi, [n] * d)
    x = m.x
    ctype = "triangle" if d == 2 else "tetrahedron"
    topo = Topology(ctype, m.cells.astype(np.int64), x.shape[0])
    # run the oracle on the library's own facet numbering
    topo.c2f = m.c2f.astype(np.int64)
    topo.f2c = m.f2c.astype(np.int64)
    topo.boundary_facets = np.flatnonzero(topo.f2c[:, 1] < 0)
    centre = [0.1, -0.2, 0.05][:d]

    def f(xx):
        acc = (1.0 * xx[0] - centre[0]) ** 2 + (1.0 * xx[1] - centre[1]) ** 2
        if d == 3:
            acc = acc + (1.0 * xx[2] - centre[2]) ** 2
        return acc + (-1.0)

    if mode == "nodal":
        nod = f(x.T)
        lh, lo_ = NodalFunction(nod), T.NodalP1(nod)
    else:
        lh, lo_ = Quadric(centre, [1.0] * d, -1.0), f
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for sl in (False, True):
            oc, of, _, om, _, _ = T.compute_tags_measures(ctype, x, topo, lo_, deg, box_mode=True,
                                                          single_layer_cut=sl)
            hc, hf, _, hm, _ = P.compute_tags_measures(m, lh, deg, box_mode=True,
                                                       single_layer_cut=sl)
            assert np.array_equal(hc.values, oc.values)
            assert np.array_equal(hf.values, of.values)
            assert np.array_equal(hm(100), om(100)) and np.array_equal(hm(101), om(101))
            assert set(np.unique(hc.values)) == {1, 2, 3}


def test_overwrite_and_errors(P):
    from phifem_amd import MeshTags
    mesh_name, f = MESHTAG_DATA["circle_in_square"]
    m, topo, x = get_mesh(P, mesh_name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ow = {"cells": MeshTags(2, [3, 10], [7, 9]), "facets": MeshTags(1, [0, 5], [11, 12])}
        hc, hf = P.compute_tags_measures(m, f, 1, box_mode=True, overwrite_tags=ow)[:2]
        assert list(hc.values[[3, 10]]) == [7, 9] and list(hf.values[[0, 5]]) == [11, 12]
        # mesh_scripts.py:608-614
        with pytest.raises(ValueError, match="Cannot overwrite cells tags"):
            P.compute_tags_measures(m, f, 1, box_mode=True,
                                    overwrite_tags={"cells": MeshTags(2, [0], [2])})
        with pytest.raises(ValueError, match="Cannot overwrite facets tags"):
            P.compute_tags_measures(m, f, 1, box_mode=True,
                                    overwrite_tags={"facets": MeshTags(1, [0], [100])})
    with pytest.raises(NotImplementedError):  # mesh_scripts.py:326-329
        P.Mesh.from_arrays("hexahedron", x, topo.cells)


def test_zero_denominator_warning(P):
    # mesh_scripts.py:129-133
    m, topo, x = get_mesh(P, "coarse_square")
    from phifem_amd.mesh_scripts import NodalFunction
    with pytest.warns(RuntimeWarning, match="zero everywhere on a cell"):
        P.compute_tags_measures(m, NodalFunction(np.zeros(m.nv)), 1, box_mode=True)
    assert np.all(m.cell_tag_values() == 2)


@pytest.mark.parametrize("mesh_name", ["disk", "square_quad"])
def test_degree_zero_detection(P, mesh_name):
    """N = 0: one point at the barycentre (mesh_scripts.py:38-39,63-64,90-91): no cell is cut."""
    m, topo, x = get_mesh(P, mesh_name)
    f = MESHTAG_DATA["circle_in_circle"][1]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        oc, of, _, om, _, _ = T.compute_tags_measures(topo.cell_type, x, topo, f, 0, box_mode=True)
        hc, hf, _, hm, _ = P.compute_tags_measures(m, f, 0, box_mode=True)
    assert np.array_equal(hc.values, oc.values) and np.array_equal(hf.values, of.values)
    # a cell is "cut" only when phi vanishes exactly at its barycentre (zero denominator -> 0.5)
    assert np.count_nonzero(hc.values == 2) <= 4
    assert np.any(hf.values == 6)          # inside and outside cells touch directly


def test_all_inside_and_all_outside(P):
    """Level-sets of one sign: `len(exterior_cells) == 0` branch (mesh_scripts.py:469-470) and the
    empty Omega_h."""
    from phifem_amd.mesh_scripts import NodalFunction
    m, topo, x = get_mesh(P, "coarse_square")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for sign, ctag in ((-1.0, 1), (1.0, 3)):
            nod = np.full(m.nv, sign)
            oc, of, _, om, _, _ = T.compute_tags_measures(topo.cell_type, x, topo, T.NodalP1(nod), 1,
                                                          box_mode=True)
            hc, hf, _, hm, _ = P.compute_tags_measures(m, NodalFunction(nod), 1, box_mode=True)
            assert np.all(hc.values == ctag)
            assert np.array_equal(hf.values, of.values)
            assert np.array_equal(hm(100), om(100)) and np.array_equal(hm(101), om(101))
        # all inside: every background-boundary facet becomes Gamma_h (tag 4)
        nod = np.full(m.nv, -1.0)
        hc, hf, _, hm, _ = P.compute_tags_measures(m, NodalFunction(nod), 1, box_mode=True)
        assert np.count_nonzero(hf.values == 4) == m.nbf and hm(100).size == 2 * m.nbf
        # all outside: no cell is tagged 1 or 2 -> no sub-mesh, no system
        nod = np.full(m.nv, 1.0)
        with pytest.raises(ValueError):
            P.compute_tags_measures(m, NodalFunction(nod), 1, box_mode=False)
        P.compute_tags_measures(m, NodalFunction(nod), 1, box_mode=True)
        with pytest.raises(ValueError):
            P.PhiFEMSolver(m).assemble(nod, nod, nod)


@pytest.mark.parametrize("ctype,n,deg", [("triangle", 12, 1), ("triangle", 9, 3), ("tetrahedron", 5, 2),
                                         ("quadrilateral", 10, 2), ("quadrilateral", 7, 3)])
def test_degree2_levelset_is_tabulated_on_the_device(P, ctype, n, deg):
    """VERDICT r1 missing #8: a P2 / Q2 nodal level-set is evaluated at the detection points by the library
    (phx_levelset_eval_points), not by numpy on the host.  The device values equal the host tabulation of round 1
    (kept in mesh_scripts as the reference) to round-off, in the PHX_PHI_POINTS layout (cells, then boundary
    facets), from a numpy array and from a tensor on the GPU; the tags that follow are identical."""
    import torch
    from phifem_amd import mesh_scripts as MS
    if ctype == "quadrilateral":
        from test_oracle_flux_quad import quad_mesh
        x0, cells0 = quad_mesh(n)
        m = P.Mesh.from_arrays("quadrilateral", x0, cells0.astype(np.int32))
        pts = m.q2_dof_points()
    else:
        d = 2 if ctype == "triangle" else 3
        m = P.create_box([-1.5] * d, [1.5] * d, [n] * d)
        pts = m.p2_dof_points()
    nod = (pts ** 2).sum(axis=1) - 1.0 + 0.05 * np.sin(3.0 * pts[:, 0])
    ref = MS._evaluate_p2(m, nod, deg)
    kind, p, loc, keep = MS._levelset_args(m, MS.NodalFunction(nod, degree=2), deg)
    dev_vals = keep[0].cpu().numpy()
    assert loc == P._lib.DEVICE and dev_vals.shape == ref.shape
    assert np.abs(dev_vals - ref).max() <= 1e-14 * max(1.0, np.abs(ref).max())
    nod_t = torch.from_numpy(nod).to(f"cuda:{m.device}")
    _, _, _, keep_t = MS._levelset_args(m, MS.NodalFunction(nod_t, degree=2), deg)
    assert np.array_equal(keep_t[0].cpu().numpy(), dev_vals)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        c_dev, f_dev = P.compute_tags_measures(m, MS.NodalFunction(nod, degree=2), deg, box_mode=True)[:2]
        # the same values handed over from the host, as round 1 did
        staged = (P._lib.PHI_POINTS,) + P._lib.ptr(ref) + (ref,)
        warn = __import__("ctypes").c_int(0)
        m._flush_lazy_tags()     # tags are about to change through the raw C ABI: MeshTags handed out keep their state
        P._lib.check(P._lib.lib.phx_tag_cells(m._h, staged[0], staged[1], staged[2], deg, 0, __import__("ctypes").byref(warn)))
        MS._tag_facets(m, staged, deg)
        c_host, f_host = m.cell_tag_values().copy(), m.facet_tag_values().copy()
    cd = np.zeros(m.nc, dtype=np.int8); cd[c_dev.indices] = c_dev.values
    fd = np.zeros(m.nf, dtype=np.int8); fd[f_dev.indices] = f_dev.values
    assert np.array_equal(cd, c_host) and np.array_equal(fd, f_host)
    with pytest.raises(ValueError):
        MS._levelset_args(m, MS.NodalFunction(nod[:-1], degree=2), deg)


@pytest.mark.parametrize("d,n,deg", [(2, 20, 2), (3, 8, 1), (3, 6, 3)])
def test_device_expression_equals_host_callable(P, d, n, deg):
    """The "UFL expression" leg on the device: DeviceExpression(f) evaluates f on a torch tensor of the physical
    detection points the library produced (phx_detection_points_physical); tags and measures equal those of the
    same expression evaluated by numpy on the host (and hence the oracle's, test_tags_bit_exact_vs_oracle)."""
    import torch
    from phifem_amd.mesh_scripts import DeviceExpression
    m = P.create_box([-1.5] * d, [1.5] * d, [n] * d)

    def f(x):     # numpy and torch alike: products and sums only, so both give the same bits
        acc = (x[0] - 0.1) * (x[0] - 0.1) + (x[1] + 0.2) * (x[1] + 0.2)
        if d == 3:
            acc = acc + x[2] * x[2]
        return acc - 1.0

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hc, hf, _, hm, _ = P.compute_tags_measures(m, f, deg, box_mode=True, single_layer_cut=True)
        dc, df, _, dm, _ = P.compute_tags_measures(m, DeviceExpression(f), deg, box_mode=True, single_layer_cut=True)
    assert np.array_equal(hc.values, dc.values) and np.array_equal(hc.indices, dc.indices)
    assert np.array_equal(hf.values, df.values) and np.array_equal(hf.indices, df.indices)
    assert np.array_equal(hm(100), dm(100)) and np.array_equal(hm(101), dm(101))
    assert set(np.unique(dc.values)) == {1, 2, 3}
    with pytest.raises(ValueError):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            P.compute_tags_measures(m, DeviceExpression(lambda x: np.zeros(3)), deg, box_mode=True)


def test_device_topology_rejects_bad_meshes_and_matches_the_host_helper(P):
    """The facet numbering of a caller-supplied mesh is built on the device (phx_topology.inc.hip): same c2f / f2c as
    the exported host helper phx_topology_build_host on a shuffled unstructured mesh, and the same errors -- a vertex
    index out of range, a facet shared by three cells."""
    import ctypes as C
    L = P._lib
    ctype, x, cells = load_mesh("coarse_square")
    rng = np.random.default_rng(3)
    cells = cells[rng.permutation(cells.shape[0])].astype(np.int32)
    m = P.Mesh.from_arrays(ctype, x, cells)
    nfpc = m.c2f.shape[1]
    c2f = np.empty((cells.shape[0], nfpc), dtype=np.int32)
    f2c = np.empty((cells.shape[0] * nfpc, 2), dtype=np.int32)
    nf = C.c_int64(0)
    L.check(L.lib.phx_topology_build_host(L.CELL_TYPES[ctype], x.shape[0], cells.shape[0],
                                          cells.ctypes.data_as(C.c_void_p), c2f.ctypes.data_as(C.c_void_p),
                                          f2c.ctypes.data_as(C.c_void_p), C.byref(nf)))
    assert m.nf == nf.value
    assert np.array_equal(m.c2f, c2f) and np.array_equal(m.f2c, f2c[:nf.value])
    bad = cells.copy()
    bad[0, 0] = x.shape[0] + 5
    with pytest.raises(ValueError):
        P.Mesh.from_arrays(ctype, x, bad)
    # three triangles on one edge: non-manifold
    xt = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [0.5, -1.0]])
    tri = np.array([[0, 1, 2], [0, 1, 3], [0, 1, 4]], dtype=np.int32)
    with pytest.raises(ValueError):
        P.Mesh.from_arrays("triangle", xt, tri)


def test_meshtags_keep_the_state_they_were_created_in(P):
    """compute_tags_measures hands out MeshTags whose host arrays are fetched on first use; tagging the same mesh again
    must not change what an earlier, still unread result shows."""
    from phifem_amd.mesh_scripts import Quadric
    m = P.create_box([-1.5] * 2, [1.5] * 2, [24, 24])
    small, big = Quadric([0.0, 0.0], [1.0, 1.0], -0.25), Quadric([0.0, 0.0], [1.0, 1.0], -1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        c_small, f_small = P.compute_tags_measures(m, small, 1, box_mode=True)[:2]      # not read yet
        c_big, f_big = P.compute_tags_measures(m, big, 1, box_mode=True)[:2]
        c_ref, f_ref = P.compute_tags_measures(m, small, 1, box_mode=True)[:2]
    n_in_big = int((c_big.values == 1).sum())
    assert np.array_equal(c_small.values, c_ref.values) and np.array_equal(c_small.indices, c_ref.indices)
    assert np.array_equal(f_small.values, f_ref.values) and np.array_equal(f_small.indices, f_ref.indices)
    assert int((c_small.values == 1).sum()) < n_in_big
    assert c_small.find(2).size > 0 and c_small.dim == 2 and f_small.dim == 1


@pytest.mark.gpu
@pytest.mark.parametrize("gdim,n,single_layer", [(3, (19, 23, 17), True), (3, (40, 33, 35), False), (2, (67, 53), True)])
def test_counts_left_by_the_tagging_kernels_equal_a_recount(P, gdim, n, single_layer):
    """The kernels that write the tags also count them (histograms, and the per-chunk counts the cut-cell, ghost-facet and
    tag-3/4 selections of an assembly start from): the histogram equals a recount of the tag arrays, the entity lists
    built from the kept counts equal the lists derived from the tags, and a solve that starts from them converges."""
    import ctypes as C
    import warnings
    from phifem_amd.mesh_scripts import NodalFunction, _tag_cells, _tag_facets
    lo, hi = [-1.5] * gdim, [1.5] * gdim
    m = P.create_box(lo, hi, n)
    x = m.x
    phi = (x ** 2).sum(axis=1) - 1.0 + 0.05 * np.sin(5.0 * x[:, 0])
    staged = _tag_cells(m, NodalFunction(phi), 1, single_layer_cut=single_layer)
    _tag_facets(m, staged, 1)
    hc, hf = (C.c_int64 * 4)(), (C.c_int64 * 7)()
    P._lib.check(P._lib.lib.phx_mesh_tag_histogram(m._h, hc, hf))
    ct, ft = m.cell_tag_values(), m.facet_tag_values()
    assert list(hc) == [int((ct == b).sum()) for b in range(4)]
    assert list(hf) == [int((ft == b).sum()) for b in range(7)]
    # ds(100) / ds(101) entities come from the selection of the facets tagged 3 / 4
    f2c = m.f2c
    from phifem_amd import _lib as L_
    for which, tag, cells_ok in ((100, 4, (1, 2)), (101, 3, (2, 3))):
        npairs = C.c_int64(0)
        L_.check(L_.lib.phx_integration_entities(m._h, which, None, C.byref(npairs)))
        facets = np.flatnonzero(ft == tag)
        expect = sum(int(ct[c] in cells_ok) for f in facets for c in f2c[f] if c >= 0)
        assert npairs.value == expect
    # and the assembly (cut-cell / ghost-facet selections with the kept counts) yields a system that solves
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s = P.PhiFEMSolver(m)
        s.assemble(phi, np.ones(m.nv), np.zeros(m.nv))
        u = s.solve(rtol=1e-9)
    assert s.stats["converged"] and np.isfinite(u).all()

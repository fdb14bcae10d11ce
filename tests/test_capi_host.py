"""CPU-only checks of the C-ABI library: it loads, exports every symbol the header declares,
and its host-only helpers agree bit for bit with the oracle.  No GPU is touched."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import points as OP
from oracle.topology import Topology

from datasets import load_mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "phifem_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(phx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from phifem_amd import _lib as L
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L.lib, s), f"{s} declared in include/phifem_hip.h but not exported"
        assert s in L.SIGNATURES, f"{s} has no ctypes signature"
    assert sorted(L.SIGNATURES) == syms
    assert L.lib.phx_version() == 1


@pytest.mark.parametrize("ctype", ["triangle", "quadrilateral", "tetrahedron"])
@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_detection_points_bit_exact(ctype, deg):
    from phifem_amd import _lib as L
    from phifem_amd.mesh_scripts import _ref_points
    got_c = _ref_points(ctype, deg, 0)
    got_f = _ref_points(ctype, deg, 1)
    assert np.array_equal(got_c, OP.cell_detection_points(ctype, deg))
    assert np.array_equal(got_f, OP.facet_detection_points(ctype, deg))


def test_point_counts():
    # mesh_scripts.py:43-92: 3N / 4N points, N+1 on the segment; tetrahedron boundary lattice
    assert [len(OP.triangle_boundary_points(n)) for n in (0, 1, 2, 3)] == [1, 3, 6, 9]
    assert [len(OP.square_boundary_points(n)) for n in (0, 1, 2, 3)] == [1, 4, 8, 12]
    assert [len(OP.segment_points(n)) for n in (0, 1, 2, 3)] == [1, 2, 3, 4]
    assert [len(OP.tetrahedron_boundary_points(n)) for n in (0, 1, 2, 3, 4)] == [1, 4, 10, 20, 34]


@pytest.mark.parametrize("mesh", ["disk", "square_tri", "square_quad", "coarse_square"])
def test_host_topology_matches_oracle(mesh):
    from phifem_amd import _lib as L
    ctype, x, cells = load_mesh(mesh)
    cells32 = np.ascontiguousarray(cells, dtype=np.int32)
    nc, nvpc = cells32.shape
    c2f = np.empty((nc, nvpc), dtype=np.int32)
    f2c = np.empty((nc * nvpc, 2), dtype=np.int32)
    nf = C.c_int64(0)
    L.check(L.lib.phx_topology_build_host(L.CELL_TYPES[ctype], x.shape[0], nc,
                                          cells32.ctypes.data_as(C.c_void_p),
                                          c2f.ctypes.data_as(C.c_void_p),
                                          f2c.ctypes.data_as(C.c_void_p), C.byref(nf)))
    topo = Topology(ctype, cells, x.shape[0])
    assert nf.value == topo.nf
    assert np.array_equal(c2f, topo.c2f)
    assert np.array_equal(f2c[:nf.value], topo.f2c)


def test_host_topology_rejects_bad_input():
    from phifem_amd import _lib as L
    cells = np.array([[0, 1, 7]], dtype=np.int32)
    c2f = np.empty((1, 3), dtype=np.int32)
    f2c = np.empty((3, 2), dtype=np.int32)
    nf = C.c_int64(0)
    rc = L.lib.phx_topology_build_host(0, 3, 1, cells.ctypes.data_as(C.c_void_p),
                                       c2f.ctypes.data_as(C.c_void_p),
                                       f2c.ctypes.data_as(C.c_void_p), C.byref(nf))
    with pytest.raises(ValueError):
        L.check(rc)
    rc = L.lib.phx_detection_points(9, 1, 0, None, C.byref(nf))
    with pytest.raises(NotImplementedError):  # mesh_scripts.py:326-329
        L.check(rc)


def test_device_entry_points_fail_loudly_without_gpu():
    from phifem_amd import _lib as L
    if L.device_count() > 0:
        pytest.skip("a GPU is present")
    import phifem_amd as P
    with pytest.raises(RuntimeError):
        P.create_box([0, 0], [1, 1], [2, 2])


def test_reshape_map_matches_oracle():
    """a5 (_reshape_map, mesh_scripts.py:195-214): reversed links, -1 padding."""
    from oracle import tagging as OT
    from phifem_amd.mesh_scripts import _reshape_map
    rng = np.random.default_rng(3)
    num = rng.integers(1, 6, size=40)
    offsets = np.concatenate([[0], num.cumsum()])
    array = rng.integers(0, 100, size=offsets[-1])
    got, w = _reshape_map(offsets, array)
    ref, w2 = OT.reshape_map(offsets, array)
    assert w == w2 == num.max() and np.array_equal(got, ref)
    assert np.array_equal(got[0, :num[0]], array[:num[0]][::-1])
    # f->c of a real mesh: boundary facets have one link and a -1
    ctype, x, cells = load_mesh("coarse_square")
    topo = Topology(ctype, cells, x.shape[0])
    cnt = (topo.f2c >= 0).sum(axis=1)
    off = np.concatenate([[0], cnt.cumsum()])
    arr = topo.f2c[topo.f2c >= 0]
    got, w = _reshape_map(off, arr)
    assert w == 2 and np.array_equal(got[cnt == 1][:, 1], -np.ones((cnt == 1).sum(), dtype=np.int64))
    assert np.array_equal(got[cnt == 2], topo.f2c[cnt == 2][:, ::-1])

"""ctypes binding of libphifem_hip.so (the C ABI of include/phifem_hip.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libphifem_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C phifem_amd/csrc`).  phifem_amd has no CPU fallback.")

# One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7).
# Loading torch FIRST makes the dynamic linker resolve this library's DT_NEEDED libamdhip64.so.7
# to the copy torch already mapped; the other order maps two runtimes and the second one finds
# "No HIP GPUs".  A pure C/C++ host links /opt/rocm's runtime directly and skips this.
if os.environ.get("PHIFEM_NO_TORCH", "") != "1":
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover
        pass

lib = C.CDLL(LIB_PATH)

HOST, DEVICE = 0, 1
TRIANGLE, QUADRILATERAL, TETRAHEDRON = 0, 1, 2
CELL_TYPES = {"triangle": TRIANGLE, "quadrilateral": QUADRILATERAL, "tetrahedron": TETRAHEDRON}
CELL_NAMES = {v: k for k, v in CELL_TYPES.items()}
PHI_NODAL_P1, PHI_POINTS, PHI_QUADRIC = 0, 1, 2
(OPT_PROFILE_SPMV, OPT_HAS_EXTERIOR, OPT_SPMV_XCD_GROUP, OPT_SPMV_VALUE_INDEX, OPT_PRECOND, OPT_ALLOW_EMPTY,
 OPT_EXPORT_CSR, OPT_STRUCTURED, OPT_DETERMINISTIC, OPT_EL_COARSE, OPT_STENCIL_PLANE_ROWS) = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11)
(ARR_COORDS, ARR_CELLS, ARR_C2F, ARR_F2C, ARR_CELL_TAGS, ARR_FACET_TAGS, ARR_BFACETS, ARR_C2E,
 ARR_EDGES) = range(9)

OK, ERR_VALUE, ERR_NOT_IMPLEMENTED, ERR_HIP, ERR_PARTITION, ERR_CAPACITY, ERR_BREAKDOWN, ERR_TIMEOUT = (
    0, -1, -2, -3, -4, -5, -6, -7)

_vp, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double
_pi, _pi64, _pd = C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_double)

# every symbol the header declares, with its signature (tests check the export list against
# include/phifem_hip.h)
SIGNATURES = {
    "phx_version": ([], _i),
    "phx_last_error": ([], C.c_char_p),
    "phx_device_count": ([_pi], _i),
    "phx_pool_release": ([], _i),
    "phx_detection_points": ([_i, _i, _i, _vp, _pi64], _i),
    "phx_levelset_points_count": ([_vp, _i, _pi64], _i),
    "phx_levelset_eval_points": ([_vp, _i, _vp, _i, _vp], _i),
    "phx_detection_points_physical": ([_vp, _i, _vp], _i),
    "phx_topology_build_host": ([_i, _i64, _i64, _vp, _vp, _vp, _pi64], _i),
    "phx_mesh_create": ([_i, _i, _i64, _vp, _i64, _vp, _i, C.POINTER(_vp)], _i),
    "phx_mesh_create_box": ([_i, _vp, _vp, _vp, _vp, _vp, _i, C.POINTER(_vp)], _i),
    "phx_mesh_set_slab_faces": ([_vp, _i, _i], _i),
    "phx_mesh_destroy": ([_vp], _i),
    "phx_mesh_counts": ([_vp, _pi64], _i),
    "phx_mesh_edge_count": ([_vp, _pi64], _i),
    "phx_mesh_get_array": ([_vp, _i, _vp, _i], _i),
    "phx_mesh_stream": ([_vp, C.POINTER(C.c_uint64)], _i),
    "phx_mesh_synchronize": ([_vp], _i),
    "phx_mesh_set_stream": ([_vp, C.c_uint64], _i),
    "phx_set_option": ([_vp, _i, _i64], _i),
    "phx_event_pair_overhead": ([_vp, _pd], _i),
    "phx_box_poisson_solve": ([_i, _pi, _pd, _i, _vp], _i),
    "phx_box_dst_bench": ([_i, _pi, _i, _i, _pd], _i),
    "phx_mesh_tag_histogram": ([_vp, _pi64, _pi64], _i),
    "phx_krylov_precond_active": ([_vp, _pi], _i),
    "phx_krylov_precond_disable": ([_vp], _i),
    "phx_precond_info": ([_vp, _pd], _i),
    "phx_precond_local_bbox": ([_vp, _pi64], _i),
    "phx_precond_setup_global": ([_vp, _pi64, _i, _i, _pi64, _pi64], _i),
    "phx_precond_set_carry_buffers": ([_vp, _vp, _vp], _i),
    "phx_precond_dist_info": ([_vp, _pi64], _i),
    "phx_krylov_attach": ([_vp, _vp, _vp, _vp], _i),
    "phx_krylov_phase": ([_vp, _i], _i),
    "phx_krylov_finish": ([_vp, _vp, _i], _i),
    "phx_krylov_profile": ([_vp, _i, _pd, _pi64], _i),
    "phx_system_get_perm": ([_vp, _vp, _vp, _vp, _i], _i),
    "phx_comm_unique_id": ([_vp], _i),
    "phx_comm_create": ([_i, _i, _vp, _i, C.POINTER(_vp)], _i),
    "phx_comm_destroy": ([_vp], _i),
    "phx_dense_inverse": ([_i, C.c_int64, _vp, C.POINTER(_i)], _i),
    "phx_comm_library": ([C.c_char_p, C.c_int64], _i),
    "phx_comm_overlap": ([_vp, C.POINTER(_i)], _i),
    "phx_solve_distributed": ([_vp, _vp, _i, _vp, _vp, _vp, _d, _i64, _vp, _i, _pd], _i),
    "phx_halo_selftest": ([_vp, _vp, _i, _vp, _vp, _vp, _vp], _i),
    "phx_tag_cells": ([_vp, _i, _vp, _i, _i, _i, _pi], _i),
    "phx_tag_facets": ([_vp, _i, _vp, _i, _i], _i),
    "phx_overwrite_tags": ([_vp, _i, _i64, _vp, _vp], _i),
    "phx_set_tags": ([_vp, _i, _vp, _i], _i),
    "phx_integration_entities": ([_vp, _i, _vp, _pi64], _i),
    "phx_submesh_create": ([_vp, C.POINTER(_vp)], _i),
    "phx_submesh_maps": ([_vp, _vp, _vp], _i),
    "phx_assemble_poisson_wd": ([_vp, _d, _d, _vp, _vp, _vp, _i, C.POINTER(_vp)], _i),
    "phx_assemble_poisson_wd_p2": ([_vp, _d, _d, _vp, _i, _vp, _vp, _i, C.POINTER(_vp)], _i),
    "phx_assemble_poisson_sd": ([_vp, _d, _i, _vp, _i, _vp, _i, C.POINTER(_vp)], _i),
    "phx_assemble_poisson_flux": ([_vp, _vp, _i, _i, _vp, _vp, _vp, _i, C.POINTER(_vp)], _i),
    "phx_assemble_elasticity_if": ([_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, C.POINTER(_vp)], _i),
    "phx_reference_nodes": ([_i, _i, _vp, C.POINTER(C.c_int)], _i),
    "phx_cell_errors": ([_vp, _i, _i, _vp, _vp, _i64, _vp, _i, _vp, _vp, _pd], _i),
    "phx_system_destroy": ([_vp], _i),
    "phx_system_info": ([_vp, _pi64], _i),
    "phx_system_export": ([_vp, _vp, _vp, _vp, _vp, _vp], _i),
    "phx_solve": ([_vp, _i, _d, _i64, _vp, _i, _pd], _i),
    "phx_spmv": ([_vp, _vp, _vp, _i], _i),
    "phx_spmv_bench": ([_vp, _i, _pd], _i),
    "phx_last_timings": ([_vp, _pd], _i),
}
for _name, (_args, _res) in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.argtypes = _args
    _fn.restype = _res


class PartitionError(ValueError):
    """The reference's facet sets overlap: dolfinx MeshTags would reject the duplicated entities."""


def check(rc):
    """Map a phx_status to the exception type the reference raises at that point."""
    if rc == OK:
        return
    msg = lib.phx_last_error().decode()
    if rc == ERR_VALUE:
        raise ValueError(msg)                      # mesh_scripts.py:242,262,609,614
    if rc == ERR_NOT_IMPLEMENTED:
        raise NotImplementedError(msg)             # mesh_scripts.py:326-329
    if rc == ERR_PARTITION:
        raise PartitionError(msg)
    if rc == ERR_CAPACITY:
        raise MemoryError(msg)
    if rc == ERR_BREAKDOWN:
        raise ArithmeticError(msg)
    if rc == ERR_TIMEOUT:
        raise TimeoutError(msg)
    raise RuntimeError(msg)


def device_count():
    n = C.c_int(0)
    lib.phx_device_count(C.byref(n))
    return n.value


def sync_torch_stream(device):
    """The library's kernels run on the mesh's own NON-BLOCKING stream, which does not wait for torch's streams:
    whatever torch still has queued for a tensor must have run before its raw pointer crosses the C ABI."""
    import torch
    st = torch.cuda.current_stream(device)
    if not st.query():        # nothing queued (every pointer after the first of an API call): no host wait
        st.synchronize()


def ptr(a):
    """void* of a numpy array (host) or a torch tensor (host or device) + its location flag."""
    if a is None:
        return None, HOST
    if hasattr(a, "data_ptr"):  # torch tensor
        if a.is_cuda:
            sync_torch_stream(a.device)
        return C.c_void_p(a.data_ptr()), (DEVICE if a.is_cuda else HOST)
    return a.ctypes.data_as(C.c_void_p), HOST

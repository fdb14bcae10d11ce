"""phifem_amd -- MI355X-native hot path of phi-FEM behind phiFEM's Python API.

Host side (this package) mirrors the reference's interface for the path:
  * `phifem_amd.mesh_scripts.compute_tags_measures`  <- src/phifem/mesh_scripts.py:571-653
  * `phifem_amd.solver.PhiFEMSolver`                 <- the "define form -> assemble -> solve"
    sequence of demo/weak-dirichlet/flower/main.py:102-186
  * `phifem_amd.solver.StrongDirichletSolver`        <- demo/strong-dirichlet/flower/main.py:83-182
  * `phifem_amd.solver.NeumannRobinSolver`           <- demo/robin/square/main.py:98-190 (simplices) and
    demo/neumann/square/main.py:49-158 (quadrilaterals)
  * `phifem_amd.io`                                  <- XDMFFile.write_mesh / write_function / read_mesh
Everything numerical runs in `libphifem_hip.so` (hand-written HIP for gfx950) through the C ABI
declared in `include/phifem_hip.h`.  There is no CPU fallback.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)
from .mesh import Mesh, MeshTags, create_box, create_rectangle  # noqa: F401
from . import io  # noqa: F401
from .mesh_scripts import DeviceExpression, NodalFunction, Quadric, compute_tags_measures  # noqa: F401
from .solver import (InterfaceElasticitySolver, NeumannRobinSolver, PhiFEMSolver,  # noqa: F401
                     StrongDirichletSolver)

/* phifem_hip.h -- C ABI of the MI355X-native phi-FEM hot path (libphifem_hip.so).
 *
 * The reference (PhiFEM/phiFEM v0.7.0) has no FFI: its boundary is the Python call
 *   compute_tags_measures(mesh, discrete_levelset, detection_degree, box_mode,
 *                         single_layer_cut, overwrite_tags)      src/phifem/mesh_scripts.py:571-653
 * followed, in every demo, by dolfinx `assemble_matrix` / `assemble_vector` and a PETSc/MUMPS
 * solve (demo/weak-dirichlet/flower/main.py:137-139,153-154,162-182).  Each entry point below
 * names the reference lines it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every buffer it passes in;
 *   - every function returns 0 on success or a negative phx_status; phx_last_error() gives
 *     the message of the last failure on the calling thread;
 *   - arguments called `loc` say where a buffer lives: PHX_HOST or PHX_DEVICE (a device
 *     pointer of the mesh's GPU, e.g. a torch tensor's data_ptr);
 *   - one mesh handle is bound to one GPU and one HIP stream; handles are not thread-safe,
 *     different handles may be used concurrently;
 *   - there is NO CPU fallback: without a usable GPU the device entry points fail with
 *     PHX_ERR_HIP.  The few host-only helpers are marked [host].
 *
 * Numbering contract (also restated in oracle/topology.py and oracle/meshgen.py)
 *   cells/vertices keep caller order; simplex local facet i is opposite local vertex i;
 *   quadrilateral vertices are in tensor-product order, facets (0,1),(0,2),(1,3),(2,3);
 *   u-DoF of vertex v is v, p-DoF is nv + v.
 */
#ifndef PHIFEM_HIP_H
#define PHIFEM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct phx_mesh phx_mesh;     /* mesh + topology + current tags, resident in HBM */
typedef struct phx_system phx_system; /* assembled matrix + rhs + solver workspace       */

enum phx_status {
  PHX_OK = 0,
  PHX_ERR_VALUE = -1,          /* -> ValueError            (mesh_scripts.py:242,262,609,614) */
  PHX_ERR_NOT_IMPLEMENTED = -2,/* -> NotImplementedError   (mesh_scripts.py:326-329)         */
  PHX_ERR_HIP = -3,            /* -> RuntimeError: HIP runtime failure / no GPU              */
  PHX_ERR_PARTITION = -4,      /* -> ValueError: facet tag sets overlap (dolfinx MeshTags
                                     rejects duplicated entities, mesh_scripts.py:554-556)    */
  PHX_ERR_CAPACITY = -5,       /* row-slot capacity exceeded during assembly; retry larger   */
  PHX_ERR_BREAKDOWN = -6,      /* Krylov breakdown (rho or omega vanished)                   */
  PHX_ERR_TIMEOUT = -7         /* multi-GPU: a collective did not complete within PHX_DIST_TIMEOUT_S (a rank
                                  never arrived); the stream is wedged -- the process must exit        */
};

enum phx_loc { PHX_HOST = 0, PHX_DEVICE = 1 };

enum phx_cell_type { PHX_TRIANGLE = 0, PHX_QUADRILATERAL = 1, PHX_TETRAHEDRON = 2 };

/* How the level-set reaches the tagging kernels (mesh_scripts.py:571-574: a P_k Function or a
 * UFL expression). */
enum phx_phi_kind {
  PHX_PHI_NODAL_P1 = 0, /* phi[nv]: values at the mesh vertices                               */
  PHX_PHI_POINTS = 1,   /* phi_cells[nc*npts_cell] then phi_bfacets[nbf*npts_facet]: values the
                           host evaluated at the physical detection points (expression mode)  */
  PHX_PHI_QUADRIC = 2   /* 7 doubles {cx,cy,cz, sx,sy,sz, c0}: sum_a (s_a x_a - c_a)^2 + c0,
                           evaluated on the device at the physical detection points           */
};

enum phx_array {          /* phx_mesh_get_array selectors */
  PHX_ARR_COORDS = 0,     /* f64 [nv*gdim]                                */
  PHX_ARR_CELLS = 1,      /* i32 [nc*nvpc]                                */
  PHX_ARR_C2F = 2,        /* i32 [nc*nfpc]                                */
  PHX_ARR_F2C = 3,        /* i32 [nf*2], ascending cell index, -1 padded  */
  PHX_ARR_CELL_TAGS = 4,  /* i32 [nc]  values 1,2,3 (0 = unclassified)    */
  PHX_ARR_FACET_TAGS = 5, /* i32 [nf]  values 1..6                        */
  PHX_ARR_BFACETS = 6,    /* i32 [nbf*2] (cell, local facet) of the background-boundary facets,
                             ascending facet index                        */
  PHX_ARR_C2E = 7,        /* i32 [nc*nepc] cell -> edges (P2 DoFs), local edge order of basix:
                             triangle (1,2),(0,2),(0,1); tetrahedron (2,3),(1,3),(1,2),(0,3),(0,2),(0,1) */
  PHX_ARR_EDGES = 8       /* i32 [ne*2] vertex pair of every edge, ascending */
};

/* ------------------------------------------------------------------ misc ------------- */
int phx_version(void);                     /* [host] ABI version (1) */
const char *phx_last_error(void);          /* [host] */
int phx_device_count(int *n);              /* [host] 0 GPUs is not an error here */
/* Return the device memory cached by the library's allocator to the driver. */
int phx_pool_release(void);

/* [host] Detection points on the reference cell / reference facet: restates
 * _reference_{segment,triangle_boundary,square_boundary}_points (mesh_scripts.py:28-92) and
 * extends them to the tetrahedron.  which = 0: cell points (tdim coords each), 1: facet points
 * (tdim-1 coords each).  Count-then-fill: call with out == NULL to get *npts. */
int phx_detection_points(int cell_type, int degree, int which, double *out, int64_t *npts);

/* Level-sets that are neither P1-nodal nor a quadric reach phx_tag_cells / phx_tag_facets as PHX_PHI_POINTS: one
 * value per detection point, the cells first ([nc][points per cell]), then the background-boundary facets in
 * ascending facet id ([nbf][points per facet]); *count = the number of values (doubles) of that layout.
 * phx_levelset_eval_points: a DEGREE-2 nodal level-set (mesh_scripts.py:95-134 with a P2 `discrete_levelset`, the
 * Robin / Neumann demos' phi_h) evaluated on the device -- nodal[nv + ne] (vertices, then edges) on simplices,
 * nodal[nv + nf + nc] (vertices, facets, cells) on quadrilaterals, at `loc`; out_device[count].
 * phx_detection_points_physical: the physical detection points x_q themselves, out_device[count][gdim], so that a
 * caller's expression (the "UFL expression" mode of tests/test_compute_meshtags.py:159-161) can be evaluated on
 * device arrays without a host round trip. */
int phx_levelset_points_count(phx_mesh *m, int detection_degree, int64_t *count);
int phx_levelset_eval_points(phx_mesh *m, int detection_degree, const double *nodal, int loc,
                             double *out_device);
int phx_detection_points_physical(phx_mesh *m, int detection_degree, double *out_device);

/* [host] Facet numbering and connectivities of an unstructured mesh: stands in for dolfinx
 * create_connectivity (mesh_scripts.py:151-153,419-422) and locate_entities_boundary (:430).
 * c2f[nc*nfpc], f2c[(max)nc*nfpc*2] are caller buffers; *nf receives the facet count. */
int phx_topology_build_host(int cell_type, int64_t nv, int64_t nc, const int32_t *cells,
                            int32_t *c2f, int32_t *f2c, int64_t *nf);

/* ------------------------------------------------------------------ mesh ------------- */
/* Unstructured mesh from caller arrays (host): replaces XDMFFile.read_mesh + connectivities
 * (tests/test_compute_meshtags.py:136-137).  Quadrilaterals in tensor-product order. */
int phx_mesh_create(int gdim, int cell_type, int64_t nv, const double *coords, int64_t nc,
                    const int32_t *cells, int device, phx_mesh **out);

/* Structured Kuhn/Freudenthal simplicial box generated ON THE DEVICE: replaces
 * dolfinx.mesh.create_rectangle / create_box (demo/weak-dirichlet/flower/main.py:45-46).
 * n[gdim] cubes per axis in this (sub-)box; vertex (i,j,k) sits at
 *   lo[a] + (hi[a]-lo[a]) * (double)(offset[a]+i) / (double)n_global[a]
 * so that a slab of a larger box (multi-GPU partition) reproduces the global coordinates bit
 * for bit.  offset/n_global may be NULL (= 0 / n). */
int phx_mesh_create_box(int gdim, const double *lo, const double *hi, const int64_t *n,
                        const int64_t *offset, const int64_t *n_global, int device,
                        phx_mesh **out);

/* Multi-GPU slabs: declare that the lower / upper end plane of the LAST axis of a device-generated
 * box is a cut through the global mesh, not part of its boundary.  Facets on such a plane stay
 * untagged (0) and take no part in the `ds` detection of mesh_scripts.py:434-461. */
int phx_mesh_set_slab_faces(phx_mesh *m, int lower_is_cut, int upper_is_cut);
int phx_mesh_destroy(phx_mesh *m);
/* Number of edges (builds the edge numbering on first use; needed for P2 spaces). */
int phx_mesh_edge_count(phx_mesh *m, int64_t *ne);
/* counts[6] = {gdim, cell_type, nv, nc, nf, nbf} */
int phx_mesh_counts(const phx_mesh *m, int64_t *counts);
/* Copy a mesh array out (to host or to a device buffer of the same GPU). */
int phx_mesh_get_array(phx_mesh *m, int which, void *out, int loc);
/* The HIP stream all work of this mesh is enqueued on (as uintptr). */
int phx_mesh_stream(phx_mesh *m, uint64_t *stream);
int phx_mesh_synchronize(phx_mesh *m);
/* Enqueue all further work of this mesh on a caller stream (e.g. torch's current stream, so that
 * torch.distributed collectives and the kernels order without extra synchronisation). */
int phx_mesh_set_stream(phx_mesh *m, uint64_t stream);
enum phx_option {
  PHX_OPT_PROFILE_SPMV = 1, /* k > 0: bracket every k-th SpMV launch of a solve with HIP events  */
  PHX_OPT_HAS_EXTERIOR = 2, /* -1: `len(exterior_cells) == 0` (mesh_scripts.py:469) is decided
                               from this mesh's tags; 0/1: imposed by a multi-GPU driver that
                               reduced it over all slabs                                        */
  PHX_OPT_SPMV_XCD_GROUP = 3, /* G > 0: SpMV blocks are regrouped so that each XCD (blockIdx % 8)
                               walks runs of G consecutive blocks; 0: plain order (default)      */
  PHX_OPT_PRECOND = 5, /* 1 (default): P1 Poisson systems on Kuhn boxes (2-D, 3-D) are preconditioned with the
                               lattice Laplacian of a box around the active vertices, inverted by f64 sine
                               transforms (u block; p block: Jacobi); 2: the same with f32 transforms -- 25 %
                               faster per application, but the half-length sine transform amplifies rounding
                               by O(L) and BiCGStab is not a flexible method: erratic on some problems
                               (2-D flower: 100-580 iterations against 38); 0: Jacobi everywhere          */
  PHX_OPT_EXPORT_CSR = 7, /* 1: systems assembled from now on ALSO keep the CSR copy phx_system_export
                               hands out (sorted columns, dolfinx sparsity pattern with its explicit zeros).
                               Default 0: the solver formats are built straight from the row slots and the CSR
                               is never formed for P1 weak-Dirichlet systems on Kuhn boxes -- export then
                               returns PHX_ERR_VALUE and the host re-assembles with this option set         */
  PHX_OPT_STRUCTURED = 8, /* 1 (default): P1 weak-Dirichlet systems on Kuhn boxes apply their
                               translation-invariant interior rows from a 7-point stencil over runs of
                               consecutive rows (no stored columns or values); only the rows near Gamma_h keep
                               SELL storage; P2 weak-Dirichlet systems on 3-D Kuhn boxes likewise (eight class
                               stencils, interior rows never assembled).  0: every row stored (SELL)          */
  PHX_OPT_DETERMINISTIC = 9, /* 1: bit-reproducible results for the scattering assemblies (P2 weak Dirichlet,
                               interface elasticity) and the Krylov solve: the element kernels run twice and
                               accumulate exactly (per-slot exponent, two accumulators), the dot products are
                               summed in a fixed order.  Costs one more pass of the element kernels and 12 bytes
                               per row slot while assembling (skipped beyond PHX_DET_LIMIT_GB, default a fifth of the device memory).
                               Default 0: f64 atomics in arrival order (results equal to round-off)           */
  PHX_OPT_EL_COARSE = 10, /* coarse-space correction of the interface-elasticity solve on generated boxes (one rank):
                               Galerkin problem on trilinear functions of spacing H = value * h per displacement
                               block, added to the vertex-block Jacobi (the iteration count then follows H / h
                               instead of growing with the box).  -1 (default): on from 80 cubes per axis with
                               H ~ n / 8 (H = 16 h beyond 128 cubes); 0: off; >= 5: this ratio.  The dense inverse
                               of the coarse matrix is the library's own (phx_dense_inverse)                   */
  PHX_OPT_STENCIL_PLANE_ROWS = 11, /* structured P1 systems on 3-D boxes: from this many interior rows per lattice
                               plane on (default 32768: three planes of the vector no longer fit an XCD's L2 next to the
                               streams) the stencil blocks of the SpMV walk the k-th eighth of EVERY plane on XCD k
                               instead of the k-th eighth of all rows; 0: never.  Placement only: same product   */
  PHX_OPT_ALLOW_EMPTY = 6, /* 1: phx_assemble_poisson_wd returns an EMPTY system (n_active = 0) when no cell
                               is tagged 1 / 2 instead of PHX_ERR_VALUE: a slab of a partitioned box that
                               does not touch the domain still joins every collective of the solve          */
  PHX_OPT_SPMV_VALUE_INDEX = 4 /* 1 (default): systems assembled from now on store SELL slices whose
                               values take <= 64 distinct doubles as dictionary + byte codes
                               (bit-identical products, 5 instead of 12 bytes per entry); 0: raw */
};
int phx_set_option(phx_mesh *m, int option, int64_t value);
/* Direct solve of the 7-point lattice Laplacian K = sum_a (h_b h_c / h_a) tridiag(-1, 2, -1)_a with
 * homogeneous Dirichlet faces on an (L0-1) x (L1-1) x (L2-1) interior lattice (x fastest), by type-I sine
 * transforms on the device (f32 != 0: lattice array and transforms in single precision);
 * L_a in {64, 128, 192, 256, 384, 512, 768, 1024}.  u overwrites f (host).
 * This is the kernel sequence of the fictitious-domain preconditioner (PHX_OPT_PRECOND), exposed for tests. */
int phx_box_poisson_solve(int device, const int *L, const double *h, int f32, double *f_host);
/* Timing aid: average microseconds of the x, y and z (forward + divide + inverse) transform passes on an
 * (L0-1) x (L1-1) x (L2-1) lattice (tools/dst_bench.py). */
int phx_box_dst_bench(int device, const int *L, int f32, int reps, double *out_us3);
/* a_host (n x n, row-major, n <= 20000) := its inverse: the library's own dense inverse (blocked Gauss-Jordan elimination
 * with partial pivoting, phx_dense.inc.hip) that the elasticity coarse correction applies to its Galerkin matrix -- exposed
 * for tests.  *singular = 1: a pivot column vanished (a_host is then undefined).  The reference has no counterpart (MUMPS
 * factorises the whole system, demo/interface-elasticity/main.py:283-289). */
int phx_dense_inverse(int device, int64_t n, double *a_host, int *singular);
/* Mean elapsed time of an empty HIP event pair on the mesh stream: the cost the bracketing of
 * PHX_OPT_PROFILE_SPMV adds to each timed launch (measurement aid of bench.py). */
int phx_event_pair_overhead(phx_mesh *m, double *seconds);
/* Tag counts of the current tagging: cells4[t] for t = 0..3, facets7[t] for t = 0..6. */
int phx_mesh_tag_histogram(const phx_mesh *m, int64_t *cells4, int64_t *facets7);

/* ------------------------------------------------------------------ tagging ---------- */
/* _tag_cells (mesh_scripts.py:284-390) incl. _compute_detection_vector (:95-134):
 * tags 1 inside / 2 cut / 3 outside, optional single-layer demotion (:349-358).
 * warn_zero_denominator (may be NULL) receives 1 when some cell's denominator is ~0 (the
 * RuntimeWarning of :129-133). */
int phx_tag_cells(phx_mesh *m, int phi_kind, const double *phi, int loc, int detection_degree,
                  int single_layer_cut, int *warn_zero_denominator);

/* _tag_facets (mesh_scripts.py:393-558), tags 1..6, from the cell tags currently held by the
 * mesh; the `ds` detection of the background-boundary facets (:434-461) uses the same phi.
 * Fails with PHX_ERR_PARTITION when the reference's sets do not partition the facets. */
int phx_tag_facets(phx_mesh *m, int phi_kind, const double *phi, int loc, int detection_degree);

/* User overwrite of tags (_overwrite_tags, mesh_scripts.py:561-568, value checks :606-615). */
int phx_overwrite_tags(phx_mesh *m, int entity_is_facet, int64_t n, const int32_t *indices,
                       const int32_t *values);
/* Install externally computed dense tags (tests, multi-GPU ghost exchange). */
int phx_set_tags(phx_mesh *m, int entity_is_facet, const int32_t *values, int loc);

/* _compute_integration_entities (mesh_scripts.py:137-192) for the two one-sided measures of
 * compute_tags_measures (:617-626): which = 100 -> facets tagged 4 seen from cells {1,2};
 * which = 101 -> facets tagged 3 seen from cells {2,3}.  Flat [cell, local facet, ...] in the
 * reference's order.  Count-then-fill on *n_pairs (out may be NULL). */
int phx_integration_entities(phx_mesh *m, int which, int32_t *out, int64_t *n_pairs);

/* Sub-mesh of the cells tagged 1 or 2 with transferred tags: dolfinx create_submesh +
 * _transfer_tags (mesh_scripts.py:217-281,636-645).  c_map[ncs], v_map[nvs] are count-then-fill
 * (pass NULL first, sizes come back in counts of *sub). */
int phx_submesh_create(phx_mesh *m, phx_mesh **sub);
int phx_submesh_maps(phx_mesh *sub, int32_t *c_map, int32_t *v_map);

/* ------------------------------------------------------------------ assembly --------- */
/* Weak-Dirichlet phi-FEM Poisson, mixed (u,p) in P1 x P1: bilinear form
 * demo/weak-dirichlet/flower/main.py:112-135 + assemble_matrix :137-139, linear form :142-151 +
 * assemble_vector :153-154, on the tags currently held by the mesh (dx((1,2)), dx(2), dS((2,3)),
 * ds = ds(100) in box mode or all exterior facets on a sub-mesh).
 * phi_h, f_h, u_D: nodal P1 arrays [nv].  Only the active DoFs (rows touched by an integral)
 * are stored. */
int phx_assemble_poisson_wd(phx_mesh *m, double pen_coef, double stab_coef, const double *phi_h,
                            const double *f_h, const double *u_D, int loc, phx_system **out);
/* The same forms with primal_degree = auxiliary_degree = 2 (BASELINE configs[2]); the
 * div(grad(.)) terms of main.py:123-128,150 are live.  DoFs per field: vertex v -> v, edge e ->
 * nv + e (PHX_ARR_EDGES / PHX_ARR_C2E), p block shifted by nv + ne.  f_h, u_D: nodal P2 arrays
 * [nv + ne]; phi_h: [nv] if phi_degree == 1, [nv + ne] if 2. */
int phx_assemble_poisson_wd_p2(phx_mesh *m, double pen_coef, double stab_coef, const double *phi_h,
                               int phi_degree, const double *f_h, const double *u_D, int loc,
                               phx_system **out);
/* Strong-Dirichlet ("direct") phi-FEM Poisson, u_h = phi_h w_h, one scalar field w of Lagrange
 * degree 1 or 2: bilinear form demo/strong-dirichlet/flower/main.py:104-117 + assemble_matrix
 * :120-121, linear form :125-129, on the tags held by the mesh (dx((1,2)), dx(2), dS((2,3)), ds =
 * ds_bdy(100) on the background mesh :60-65 or all exterior facets of a sub-mesh :70).  f_h: nodal
 * array of the w space ([nv], or [nv + ne] at degree 2); phi_h: [nv] if phi_degree == 1, [nv + ne]
 * if 2.  The system holds the active w DoFs only (n_active_u == n_active, n_full = DoFs of the
 * space); the caller forms u_h = w_h phi_h at the nodes of its solution space (main.py:176-182). */
int phx_assemble_poisson_sd(phx_mesh *m, double stab_coef, int degree, const double *phi_h,
                            int phi_degree, const double *f_h, int loc, phx_system **out);
/* Neumann / Robin phi-FEM Poisson (-lap u + u = f, du/dn + kappa u = g on the boundary), mixed (u, y, p) in
 * P1 x P1^d x DG0 with a P2 level-set, on simplices: forms demo/robin/square/main.py:112-168 + assemble_matrix /
 * assemble_vector :145-147,167-168; kappa = 0 with facet_tag = 3 is the formulation of
 * demo/neumann/square/main.py:113-158.  QUADRILATERAL meshes (the cell type of that demo, :49-50; axis-parallel
 * rectangles in tensor-product vertex order, else PHX_ERR_NOT_IMPLEMENTED) are assembled as Q1 x Q1^2 x DG0 with a Q2
 * level-set: phi_h[nv + nf + nc] = vertex values, edge-midpoint values by facet id, cell-centre values; the cut-cell
 * integrals then use a tensor Gauss rule of quadrature_degree / 2 + 1 points per direction.
 * params = {pen_coef, stab_coef, robin_coef}; facet_tag: the interior facets carrying the gradient-jump term
 * (2 in the Robin demo :140, 3 in the Neumann demo :134); quadrature_degree: degree of the cut-cell rule
 * (|grad phi_h| is not polynomial; 10 = UFL's estimate for the Robin integrand).  phi_h: [nv + ne] (P2: vertex
 * values, then edge values), f_h, g_h (u_N / u_R): [nv].  DoFs in the full numbering: u at vertex v -> v,
 * y_k at vertex v -> (1 + k) nv + v, p on cell c -> (1 + gdim) nv + c; n_full = (1 + gdim) nv + nc. */
int phx_assemble_poisson_flux(phx_mesh *m, const double *params, int facet_tag, int quadrature_degree,
                              const double *phi_h, const double *f_h, const double *g_h, int loc,
                              phx_system **out);
/* Interface linear elasticity, 5-field mixed phi-FEM (u_in, u_out, y_in, y_out, p), all P1:
 * demo/interface-elasticity/main.py:179-235 (bilinear form) + assemble_matrix(bcs) :237-239, linear
 * form :255-269 + apply_lifting / bc.set :271-277, material law data.py:5-36, on the tags held by
 * the mesh (box mode: dx((1,2)), dx((2,3)), dx(2), dS(3), dS(4), d_bdry(100), d_bdry(101)).
 *   params[6] = {E_in, nu_in, E_out, nu_out, penalization_coefficient, stabilization_coefficient}
 *   phi_h[nv]; f_h, u_D: [d*nv] component-major nodal vector fields; bc_vertices[nbc]: vertices
 *   where u_in = u_D is imposed (main.py:158-177).
 * DoF layout: component-major blocks of nv: u_in[a] -> a, u_out[a] -> d+a, y_in[a][b] -> 2d+a d+b,
 * y_out[a][b] -> 2d+d^2+a d+b, p[a] -> 2d+2d^2+a; phx_solve returns x[(2d+2d^2+d)*nv]. */
int phx_assemble_elasticity_if(phx_mesh *m, const double *params, const double *phi_h,
                               const double *f_h, const double *u_D, const int32_t *bc_vertices,
                               int64_t nbc, int loc, phx_system **out);
int phx_system_destroy(phx_system *s);
/* info[14] = {n_active, n_active_u, nnz (structural, CSR), n_full (= 2*nv), sell_padded_nnz,
 *             slot_capacity, sell_nnz (explicit zeros dropped), n_slices, value-indexed slices,
 *             matrix bytes one SpMV of the solve streams (columns + value stream + slice table),
 *             value-indexed slices whose dictionary exceeds 64 entries (LDS look-up),
 *             rows applied from the stencil (structured systems, else 0), stencil runs,
 *             1 if the system holds its CSR copy (phx_system_export can return the matrix)} */
int phx_system_info(const phx_system *s, int64_t *info);
/* CSR of the active system in ORIGINAL active numbering (sorted columns) + the map active row ->
 * full DoF index; host buffers: rowptr[n_active+1], col[nnz], val[nnz], rhs[n_active],
 * dof[n_active].  Any pointer may be NULL. */
int phx_system_export(phx_system *s, int64_t *rowptr, int32_t *col, double *val, double *rhs,
                      int64_t *dof);

/* ------------------------------------------------------------------ solve ------------ */
enum phx_method { PHX_BICGSTAB_JACOBI = 0 };
/* Replaces KSP preonly + LU/MUMPS with null-pivot detection (main.py:162-182): solves the active
 * system, returns x in FULL numbering [2*nv] with inactive DoFs = 0 (what ICNTL(24)=1 yields).
 * stats[8] = {iterations, relative residual ||b-Ax||/||b||, seconds, spmv_count,
 *            average SpMV seconds and launches timed (PHX_OPT_PROFILE_SPMV),
 *            converged (1: relative residual <= rtol; 0: max_iter reached -- the reference solves directly,
 *            its callers assume an accurate x: treat 0 as a failure), breakdown restarts}.
 * Returns PHX_OK also when max_iter was reached: check stats[6].
 * The residual that is reported (and tested against rtol) is the TRUE one: when the recurrences announce
 * convergence, b - A x is evaluated once and, if it does not meet rtol (drift after thousands of iterations on
 * ill-conditioned systems), the iteration restarts from it. */
int phx_solve(phx_system *s, int method, double rtol, int64_t max_iter, double *x, int loc,
              double *stats);

/* --- pieces of the solve for externally driven (multi-GPU) iterations ------------------------
 * The driver owns the loop, does the halo exchange of the SpMV inputs before phases 2 / 4 and
 * all-reduces the 8 reduction scalars (scal[8..15]) after phases 0, 2, 4 and 5.  Vectors are in SOLVER
 * order (row i of the solver = active row perm[i]).
 *   work: 10 vectors of n doubles {r, rhat, p, v, s, t, y, b, phat, shat};  scal: 8208 doubles (16
 *         scalars + 2 x 8 x 64 dot-product slots of 64 B)
 *   own : n bytes in solver order, 1 = this rank owns the row (NULL = all)
 * SpMV inputs: p / s, or -- when phx_krylov_precond_active says 1 after phase 0 -- phat / shat, which
 * phases 7 / 8 compute from p / s (rank-local block preconditioner on the owned rows). */
int phx_krylov_attach(phx_system *s, double *work, double *scal, const uint8_t *own);
int phx_krylov_precond_active(const phx_system *s, int *active);
/* --- slab-exact preconditioner of a z-partitioned box -----------------------------------------------------
 * Every rank holds whole x-y planes of ONE global lattice box around the active vertices of ALL ranks: sine
 * transforms in x and y stay rank-local, the tridiagonal z solves continue across ranks through two carries per
 * lattice column and rank (one all-gather per application).  Protocol, after phx_krylov_attach:
 *   phx_precond_local_bbox   -> out6 = min / max GLOBAL vertex indices of this rank's owned active u DoFs
 *                               (INT64_MAX / -1 when it owns none); the driver reduces MIN / MAX over the ranks
 *   phx_precond_setup_global -> builds this rank's share; zb[nranks + 1]: rank r owns the global vertex planes
 *                               [zb[r], zb[r+1]);  *ncol = 0: not applicable (every rank gets the same answer)
 *   phx_precond_set_carry_buffers -> send[2 ncol], recv[nranks * 2 ncol] doubles on the device (caller-owned)
 *   per application: phase 7 (or 8), all-gather send -> recv over the ranks, phase 9 (or 10).
 * phx_precond_dist_info: out4 = {active, doubles per rank in the all-gather, planes held here, planes of the
 * global column}. */
int phx_precond_local_bbox(phx_system *s, int64_t *out6);
int phx_precond_setup_global(phx_system *s, const int64_t *bbox6, int nranks, int rank, const int64_t *zb,
                             int64_t *ncol);
int phx_precond_set_carry_buffers(phx_system *s, double *send, double *recv);
int phx_precond_dist_info(const phx_system *s, int64_t *out4);
/* Multi-GPU drivers choose the preconditioner collectively: phase 0 leaves this rank's veto in scal[8 + 5]
 * (1: box preconditioner configured out or impossible here; 0: built, or the rank owns no u row); the driver
 * all-reduces scal[8 + 4 .. 8 + 5] (SUM) after phase 0 and, when the vetoes add up to > 0, calls this on
 * every rank before the first iteration: Jacobi everywhere, same vectors exchanged, same check cadence. */
int phx_krylov_precond_disable(phx_system *s);
/* After a solve: out[8] = {preconditioner (0: Jacobi, 1: lattice solve, 2: vertex-block Jacobi of the
 * elasticity system), transform lengths L0, L1, L2, lattice points of
 * the box, sampled average seconds of one y-pass launch of the sine transforms (PHX_OPT_PROFILE_SPMV),
 * launches sampled, bytes per lattice value (4: f32 transforms, 8: f64)}. */
int phx_precond_info(phx_system *s, double *out);
/* phase 0 begin, 1 begin2, 2 v=A phat, 3 s-update, 4 t=A shat, 5 x/r-update, 6 p-update + roll,
 * 7 phat = P p, 8 shat = P s (no-ops without a preconditioner); with the slab-exact preconditioner
 * (phx_precond_setup_global) 7 / 8 run its first half and 9 / 10 the second, the driver all-gathering the
 * carry buffer in between.  Multi-GPU overlap: 20 | 21 (40 | 41) = phase 2 (4) in two launches -- the rows that
 * read no halo entry, then (after the driver has unpacked the halo) the rows that do; they need the row flags
 * phx_solve_distributed builds.  True-residual verification: 11 t = A y (after a halo exchange of y), 12 r = b - t and
 * (r, r) -> scal[8 + 5] (all-reduced by the driver), 13 restart of the recurrences from r. */
int phx_krylov_phase(phx_system *s, int phase);
int phx_krylov_finish(phx_system *s, double *x, int loc);
/* reset != 0: arm the SpMV event profile; else collect {average seconds, launches timed}. */
int phx_krylov_profile(phx_system *s, int reset, double *avg_seconds, int64_t *count);
/* perm[n] (solver position -> active row), dof_u[nv] / dof_p[nv] (vertex -> active row or -1). */
int phx_system_get_perm(phx_system *s, int32_t *perm, int32_t *dof_u, int32_t *dof_p, int loc);

/* --- native multi-GPU solve: RCCL on the solver's stream (bound with dlopen at run time) -------
 * One communicator per process; the 128-byte id comes from rank 0 (phx_comm_unique_id) and is
 * broadcast by the host (torch.distributed).  phx_solve_distributed runs the same phase sequence
 * as phx_solve with a point-to-point halo of p and s (ncclSend/ncclRecv with <= 2 neighbours, on the communicator's
 * own stream, overlapped with the rows of the SpMV that read no halo entry; PHX_DIST_OVERLAP=0: in series) and
 * three all-reduces of 1, 2, 2 doubles per iteration.  Convergence checks are scheduled as in phx_solve; relres /
 * converged refer to the TRUE residual (one more halo exchange + SpMV + all-reduce, restart when it misses rtol).
 * Every host wait is bounded by PHX_DIST_TIMEOUT_S (default 300 s) -> PHX_ERR_TIMEOUT.
 * Interface-elasticity systems (PHX_OPT_EL_COARSE): the coarse correction is built collectively before the first
 * iteration (all-reduces of the used-node flags, of a veto and of the coarse matrix: every rank restricts the rows it
 * owns), and every application adds one all-reduce of the coarse right-hand side (phases 30 / 32 restrict p / s,
 * 31 / 33 add the prolonged correction to phat / shat).  The phase API alone keeps the vertex blocks.
 * Buffers attached with phx_krylov_attach.
 *   peers[npeers]; counts[2*npeers] = {n_send, n_recv}; idx[2*npeers] = device int64 arrays
 *   {send positions, recv positions} in solver order. */
typedef struct phx_comm phx_comm;
int phx_comm_unique_id(void *out128);
int phx_comm_create(int nranks, int rank, const void *uid128, int device, phx_comm **out);
int phx_comm_destroy(phx_comm *c);
/* Path of the shared object the ten RCCL entry points were bound from (dladdr of ncclAllReduce), zero-terminated into
 * out[len].  The library binds RCCL at run time (librccl.so.1 of the process; PHX_RCCL_LIB names another file -- the
 * one-GPU tests load a host-staged stand-in that way -- and says so on stderr): a driver prints this next to its
 * result so that what carried the collectives is on record.  No counterpart in the reference (serial). */
int phx_comm_library(char *out, int64_t len);
/* *out = 1 when phx_solve_distributed overlaps the halo exchanges of this communicator with the SpMV rows that read no
 * halo entry (second stream, two events): phx_halo_selftest has then seen the overlapped exchange deliver, bit for bit, what
 * the exchange in series delivers, and PHX_DIST_OVERLAP is not 0.  Otherwise the exchanges stay on the solver stream. */
int phx_comm_overlap(const phx_comm *c, int *out);
int phx_solve_distributed(phx_system *s, phx_comm *c, int npeers, const int *peers,
                          const int64_t *counts, const int64_t *const *idx, double rtol,
                          int64_t max_iter, double *x, int loc, double *stats);
/* stats[8] as phx_solve, except stats[7] = 1 when all ranks kept the box preconditioner, 0 when one vetoed it. */
/* One halo exchange of `vec` (solver order) through the solver's own code path: wiring test. */
int phx_halo_selftest(phx_system *s, phx_comm *c, int npeers, const int *peers,
                      const int64_t *counts, const int64_t *const *idx, double *vec);

/* y = A x on the active system (solver ordering is internal; x, y are in active numbering).
 * For tests and halo-exchange driven (multi-GPU) solvers. */
int phx_spmv(phx_system *s, const double *x, double *y, int loc);
/* Cell-wise discretisation errors, demo/interface-elasticity/main.py:327-383: the exact solution
 * and u_h are interpolated into the Lagrange space of degree 3 (= primal_degree + 2 for P1, basix's default
 * GLL-warped variant [3P]), e = I(u_ex) - u_h, and per listed cell K
 *   l2_local[i]  = int_K e . e            (main.py:369-377)      h10_local[i] = int_K grad e : grad e   (:347-356)
 * norms[4] = { sum l2_local, sum h10_local, int |I u_ex|^2 (:363-367), int |grad I u_ex|^2 (:339-345) } over the
 * listed cells, summed in a fixed order.  u_h: component-major nodal values [ncomp][nv] (degree_h 1) or
 * [ncomp][nv + ne] (2); u_ref: the exact solution at the reference nodes of each listed cell,
 * [ncells][n_nodes][ncomp], nodes as `phx_reference_nodes` lists them (physical point = sum_m bary[m] x_vertex_m);
 * cell_list: ncells cell indices, or NULL for all cells (ncells == nc).  All arrays live at `loc`. */
int phx_reference_nodes(int gdim, int degree, double *bary, int *n_nodes);
int phx_cell_errors(phx_mesh *m, int ncomp, int degree_h, const double *u_h, const double *u_ref,
                    int64_t ncells, const int32_t *cell_list, int loc, double *l2_local,
                    double *h10_local, double *norms);

/* Times `reps` launches of the SpMV kernel with HIP events on the mesh's stream.
 * out[3] = {avg ms per launch, algorithmic bytes per launch (12 nnz + 20 n), padded bytes}. */
int phx_spmv_bench(phx_system *s, int reps, double *out);

/* Per-stage device seconds of the last calls, measured with HIP events on the mesh's stream:
 * t[8] = {tag_cells, tag_facets, assemble, solve, spmv_avg, n/a, n/a, n/a}. */
int phx_last_timings(const phx_mesh *m, double *t);

#ifdef __cplusplus
}
#endif
#endif /* PHIFEM_HIP_H */

"""CPU oracle for the phi-FEM hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain numpy/scipy restatement of the reference algorithm
(PhiFEM/phiFEM v0.7.0, `src/phifem/mesh_scripts.py` for tagging and
`demo/weak-dirichlet/flower/main.py:102-186` for the weak-Dirichlet Poisson forms).
It is the *checker*: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import it.  Nothing under `phifem_amd/` imports it and the
product path never falls back to it.

Pinning status (see DESIGN.md "Oracle"):
  * tagging (a1-a8): pinned to the reference's own goldens -- tag histograms of the
    336 CSVs of tests/test_compute_meshtags.py (numbering-free; the CSV indices are
    dolfinx-local and cannot be reproduced without dolfinx) and the 9 known answers
    of tests/test_one_sided_integral.py.
  * assembly / solve (a9-a11): PARITY UNPINNED -- the reference holds no test,
    fixture or golden for any assembled matrix, vector or solution; the reference
    itself cannot be imported here (dolfinx is not installed).  The restatement is
    checked by manufactured solutions and algebraic identities instead.
"""

"""Generates tests/golden/neumann_data.npz by importing the reference's plain-numpy problem data
(/root/reference/demo/neumann/square/data.py, importable without dolfinx) at seeded random points.
Run in the build container only; the .npz is the committed fixture."""
import importlib.util
import sys

# importing the reference's data modules must not leave a __pycache__ in /root/reference
sys.dont_write_bytecode = True
import os

import numpy as np

spec = importlib.util.spec_from_file_location("refdata_neumann", "/root/reference/demo/neumann/square/data.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
x = np.random.default_rng(20260701).uniform(-1.0, 1.0, size=(2, 400))
out = {"x": x}
for name in ("detection_levelset", "levelset", "exact_solution", "source_term", "neumann_data"):
    out[name] = getattr(ref, name)(x.copy())
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "neumann_data.npz"), **out)

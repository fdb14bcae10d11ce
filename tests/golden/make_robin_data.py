"""Generates tests/golden/robin_data.npz by importing the reference's plain-numpy problem data
(/root/reference/demo/robin/square/data.py, importable without dolfinx) at seeded random points.
Run in the build container only; the .npz is the committed fixture."""
import importlib.util
import sys

# importing the reference's data modules must not leave a __pycache__ in /root/reference
# (read-only by contract; root ignores the mode bits)
sys.dont_write_bytecode = True
import os

import numpy as np

spec = importlib.util.spec_from_file_location("refdata", "/root/reference/demo/robin/square/data.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
x = np.random.default_rng(20260630).uniform(-1.0, 1.0, size=(2, 400))
out = {"x": x, "robin_coef": np.float64(ref.robin_coef)}
for name in ("detection_levelset", "levelset", "exact_solution", "source_term", "robin_data"):
    out[name] = getattr(ref, name)(x.copy())
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "robin_data.npz"), **out)

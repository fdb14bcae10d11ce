"""Elasticity solve with / without the coarse correction (development aid): iterations and seconds.
usage: el_coarse_probe.py n [ratio ...]   (ratio 0 = vertex blocks alone)"""
import os, sys, subprocess, json
if len(sys.argv) > 2 and sys.argv[1] != "--child":
    n = sys.argv[1]
    for ratio in sys.argv[2:]:
        env = dict(os.environ, PHX_EL_COARSE=ratio)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n], env=env, capture_output=True, text=True)
        print(f"ratio {ratio}: {r.stdout.strip()} {r.stderr.strip()[-300:] if r.returncode else ''}", flush=True)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings
import torch
from phifem_amd.distributed import ElasticitySlabProblem
n = int(sys.argv[2])
p = ElasticitySlabProblem(n, n, rtol=1e-8)
p.setup()
warnings.simplefilter("ignore")
p.step()
res = p.step()
print(json.dumps({"it": res["iterations"], "conv": res["converged"], "relres": res["relres"], "precond": res["precond"],
                  "ratio": res["precond_L"][0], "nc": res["precond_points"],
                  "stage_ms": {k: round(1e3 * v, 1) for k, v in res["stage_s"].items()}}))

#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
ADMM_HIP_OC_DEBUG=1 python - <<'PY' 2>&1 | tail -12
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, scenes
sc = scenes.cloth_scene(40)
o = sc.make_oracle()
b = o.A @ np.random.default_rng(3).standard_normal(o.dof)
for aff in ("1", "0"):
    os.environ["ADMM_HIP_OC_AFFINE"] = aff; os.environ["ADMM_HIP_OC_CHEB"] = "0"
    s = sc.make_solver(pcg_tol=1e-8, pcg_max_iters=5000)
    x, it = s.global_solve(b, np.zeros(o.dof))
    print("affine", aff, "iters", it, flush=True)
    s.close()
PY

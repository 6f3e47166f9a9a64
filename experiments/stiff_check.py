"""Nearly incompressible rubber (E = 1e7, nu = 0.499: lambda / mu ~ 500) on the on-chip PCG vs the launch path."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import admm_elastic_amd as pkg
from admm_elastic_amd.solver import Lame
import scenes
for kind in (pkg.TET_NEOHOOKEAN, pkg.TET_STVK):
    res = []
    for launches in ("0", "1"):
        os.environ["ADMM_HIP_PCG_LAUNCHES"] = launches
        sc = scenes.cube_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 30, kind, lame=Lame.rubber(), admm_iters=20, linsolver=0)
        s = sc.make_solver(pcg_tol=1e-8, pcg_max_iters=5000)
        inner = 0; unconv = 0
        for f in range(3):
            s.step(); rd = s.runtime_data(); inner += rd.inner_iters; unconv += rd.unconverged_solves
        res.append(s.m_x.copy())
        print("kind", kind, "launch path" if launches == "1" else "on-chip   ", "inner iterations", inner, "unconverged", unconv, "finite", np.isfinite(s.m_x).all(), "max disp %.4f" % np.abs(s.m_x - sc.x.ravel()).max())
        s.close()
    os.environ.pop("ADMM_HIP_PCG_LAUNCHES", None)
    print("   on-chip vs launch path:", scenes.rel_err(res[0], res[1]))

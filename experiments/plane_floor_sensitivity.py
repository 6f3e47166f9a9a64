import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np, scenes
import admm_elastic_amd as pkg
from admm_elastic_amd.solver import Plane
def run(obst, floor_y=-0.01, steps=4):
    sc2 = scenes.cube_scene(4, pkg.TET_NEOHOOKEAN, pin_face=False, admm_iters=8, linsolver=2, size=0.5)
    sc2.pins.clear()
    sol = sc2.make_solver(init=False)
    sol.add_obstacle(pkg.Floor(floor_y) if obst == "floor" else Plane([0.0, 2.0, 0.0], 2 * floor_y))
    assert sol.initialize(sc2.product_settings)
    out = []
    for _ in range(steps):
        sol.step(); out.append(sol.m_x.copy())
    sol.close()
    return out
for fy in (-0.01, -0.011, -0.0123, -0.02):
    a = run("floor", fy); b = run("plane", fy)
    print(os.environ.get("ADMM_HIP_UZ_PERSIST", "1"), os.environ.get("ADMM_HIP_UZ_COMPACT", "1"), "floor", fy, "max |floor - plane| per step:", " ".join("%.1e" % np.abs(x - y).max() for x, y in zip(a, b)), flush=True)

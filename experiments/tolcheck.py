import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, scenes
from test_gpu_parity import KINDS, deformed
for kind in KINDS:
    for amp in (0.0, 0.01, 0.12, 0.3):
        sc = scenes.cube_scene(4, KINDS[kind], pin_face=False)
        s = sc.make_solver(); o = sc.make_oracle(mode=1)
        x = deformed(sc, amp, 11); R = o.R
        u0 = 0.05 * np.random.default_rng(12).standard_normal(R)
        z, u = s.local_step(x, u0)
        zo = np.zeros(R); uo = u0.copy(); o.local_step(x, zo, uo)
        print(kind, amp, "max|z-zo| %.2e  max|u-uo| %.2e" % (np.abs(z - zo).max(), np.abs(u - uo).max()))
        s.close()
# the largest difference at amp 0.3: which element, and what are the stretches of its q = D x + u ?
sc = scenes.cube_scene(4, KINDS["neohookean"], pin_face=False)
s = sc.make_solver(); o = sc.make_oracle(mode=1)
x = deformed(sc, 0.3, 11); R = o.R
u0 = 0.05 * np.random.default_rng(12).standard_normal(R)
z, u = s.local_step(x, u0)
zo = np.zeros(R); uo = u0.copy(); o.local_step(x, zo, uo)
d = np.abs(z - zo).reshape(-1, 9).max(axis=1)
for t in np.argsort(d)[-3:]:
    q = (zo + uo).reshape(-1, 3, 3)[t]
    print("tet", t, "diff %.2e" % d[t], "stretches of q", np.linalg.svd(q, compute_uv=False), "det %.3e" % np.linalg.det(q))

"""Debug helper: Uzawa solve with dynamic + passive rows, GPU vs oracle, for several iteration caps."""
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, scenes
for floor in (None, 0.02):
    for uzit in (20, 100):
        sc = scenes.two_blocks_scene(3, floor=floor)
        s = sc.make_solver(pcg_tol=1e-12, pcg_max_iters=600, uzawa_max_iters=uzit)
        o = sc.make_oracle(uzawa_max_iters=uzit)
        rng = np.random.default_rng(2)
        x = sc.x.ravel().copy()
        b = o.A @ (x + 0.001 * rng.standard_normal(x.size))
        hits = o.detect_passive(x); o._dhits = o.detect_dynamic(x)
        Cm, c = o.make_matrix(hits, o._dhits)
        xo, ito = o.solve_uzawa(x, b, hits)
        xg, itg = s.global_solve(b, x)
        print(floor, uzit, 'its', ito, itg, 'diff', np.abs(xg - xo).max(), 'res o/g', np.linalg.norm(Cm @ xo - c), np.linalg.norm(Cm @ xg - c),
              'nrows', Cm.shape[0])

import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, bench
sc, nt, nv = bench.build_scene(bench.WORKLOADS["cube1m_nh"], 60)
s = sc.make_solver(pcg_tol=1e-8, pcg_max_iters=800)
s.upload()
g = 0.0; inner = 0; unc = 0
for f in range(4):
    s.step_device(stats=True); rd = s.runtime_data()
    if f >= 1: g += rd.global_ms; inner += rd.inner_iters; unc += rd.unconverged_solves
s.download()
print('n=60 verts', nv, 'global ms/iter %.3f' % (g / 60), 'its/solve %.1f' % (inner / 60), 'unconverged', unc, 'finite', np.isfinite(s.m_x).all())

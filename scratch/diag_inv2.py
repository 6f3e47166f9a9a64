import sys, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from test_known_answers import product_tet, oracle_tet, VERTS
from admm_elastic_amd.solver import Lame
lame = Lame(mu=100.0, lambda_=100.0)
iters = 27
s, st = product_tet(lame, timestep_s=0.7, linsolver=0, admm_iters=iters)
st.pcg_tol = 1e-12; st.pcg_max_iters = 500
assert s.initialize(st)
o = oracle_tet(100.0, 100.0, dt=0.7, gravity=0.0, admm_iters=iters, linsolver=0)
s.m_x[0:3] = (1.0,1.0,1.0); o.x[0:3] = (1.0,1.0,1.0)
for f in range(10):
    s.step(); o.step()
    print("frame", f, "err %.2e" % np.abs(s.m_x - o.x).max(), "its", s.runtime_data().pcg_iters_per_solve[:27])

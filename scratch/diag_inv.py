import sys, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from test_known_answers import product_tet, volume, VERTS
from admm_elastic_amd.solver import Lame
lame = Lame(mu=100.0, lambda_=100.0)
for tol in (1e-10, 1e-12):
    last=None; worst=0
    for iters in range(20, 40):
        s, st = product_tet(lame, timestep_s=0.7, linsolver=0, admm_iters=iters)
        st.pcg_tol = tol; st.pcg_max_iters = 500
        assert s.initialize(st)
        s.m_x[0:3] = (1.0,1.0,1.0)
        its=[]
        for _ in range(10):
            s.step(); its.append(s.runtime_data().inner_iters)
        if last is not None: worst=max(worst, np.linalg.norm(last - s.m_x[0:3]))
        last = s.m_x[0:3].copy(); s.close()
    print("recycle off" if os.environ.get("ADMM_HIP_NO_RECYCLE")=="1" else "recycle on ", "tol", tol, "worst consecutive diff %.3e" % worst, "inner its last", its[-3:], "unconv", s.runtime_data().unconverged_solves)

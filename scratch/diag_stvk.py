import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import admm_elastic_amd as pkg, scenes
from test_gpu_parity import deformed
from oracle import oracle as orc
sc = scenes.cube_scene(4, pkg.TET_STVK, pin_face=False)
s = sc.make_solver(); o = sc.make_oracle(mode=1)
x = deformed(sc, 0.12, 11)
u0 = 0.05*np.random.default_rng(12).standard_normal(o.R)
z,u = s.local_step(x,u0)
zo=np.zeros(o.R); uo=u0.copy(); o.local_step(x,zo,uo)
q = (uo - u0 + zo)  # Dx
q = q + u0
d = np.abs(z-zo).reshape(-1,9).max(axis=1)
bad = np.nonzero(d>1e-7)[0]
print("n tets", len(d), "bad", len(bad))
mu,la = o.t_mu[0], o.t_la[0]; k = o.t_k[0]
def E(zz, qq):
    Z = zz.reshape(3,3).T; Q = qq.reshape(3,3).T
    sg = np.linalg.svd(Z, compute_uv=False)
    st = 0.5*(sg**2-1)
    return mu*(st**2).sum() + 0.5*la*st.sum()**2 + 0.5*k*((Z-Q)**2).sum()
for t in bad[:20]:
    Q = q[9*t:9*t+9].reshape(3,3).T
    U,S,V = orc.signed_svd3(Q)
    print(t, "sig0", np.round(S,4), "E gpu %.6e  E orc %.6e" % (E(z[9*t:9*t+9], q[9*t:9*t+9]), E(zo[9*t:9*t+9], q[9*t:9*t+9])),
          "sv gpu", np.round(np.linalg.svd(z[9*t:9*t+9].reshape(3,3).T, compute_uv=False),4), "sv orc", np.round(np.linalg.svd(zo[9*t:9*t+9].reshape(3,3).T, compute_uv=False),4))

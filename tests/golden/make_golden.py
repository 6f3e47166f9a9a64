"""Generates tests/golden/ref_vectors.npz from the REAL reference sources compiled into
oracle/_ref/libadmm_ref.so (run in the build container, where /root/reference exists):
    python tests/golden/make_golden.py
Inputs are seeded; outputs are what the reference's own code returns for them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc  # noqa: E402
import test_oracle_vs_ref as T  # noqa: E402


def main():
    ref = orc.ref_lib()
    assert ref is not None, "build oracle/_ref first (needs /root/reference)"
    Fs = np.array(T.svd_cases())
    Ss = []
    for F in Fs:
        a = np.ascontiguousarray(F.T).copy()
        U = np.zeros(9); S = np.zeros(3); V = np.zeros(9)
        ref.ref_signed_svd(T._p(a), T._p(S), T._p(U), T._p(V))
        Ss.append(S)
    verts, tris, x, u, mu, la, limits = T.tri_case()
    n, nv = len(tris), len(verts)
    z = np.zeros(6 * n); uu = u.copy(); w = np.zeros(n)
    ref.ref_tri_local_step(n, T._i(np.ascontiguousarray(tris)), nv, T._p(np.ascontiguousarray(verts)), mu, la, limits[0], limits[1],
                           T._p(np.ascontiguousarray(x)), T._p(z), T._p(uu), T._p(w), None, None, None)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_vectors.npz")
    np.savez_compressed(out, svd_F=Fs, svd_S=np.array(Ss), tri_z=z, tri_u=uu, tri_w=w)
    print("wrote", out)


if __name__ == "__main__":
    main()

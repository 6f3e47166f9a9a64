"""Generates tests/golden/ref_vectors.npz from the REAL reference sources compiled into oracle/_ref/libadmm_ref.so (run in
the build container, where /root/reference exists):
    python tests/golden/make_golden.py
Inputs are seeded (tests/ref_cases.py); outputs are what the reference's own code returns for them: signed SVD factors
(FastSVD.hpp), triangle local step + reduction (TriEnergyTerm.cpp, three strain-limit settings), SpringPin local step
(SpringEnergyTerm.hpp), floor constraint rows (Collider::detect + ConstraintSet::make_matrix), Eigen::SimplicialLDLT
solve, the three xu:: splines (f, g, h, df, dg, dh; kappa = 0 and kappa != 0) and Lame."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc  # noqa: E402
import ref_cases as R  # noqa: E402


def main():
    L = orc.ref_lib()
    assert L is not None, "build oracle/_ref first (needs /root/reference)"
    data = {}
    for name, fn in R.CASES.items():
        for k, v in fn(L).items():
            data[name + "/" + k] = np.asarray(v)
    np.savez_compressed(R.GOLD, **data)
    print("wrote", R.GOLD, "(%d arrays, %d bytes)" % (len(data), os.path.getsize(R.GOLD)))


if __name__ == "__main__":
    main()

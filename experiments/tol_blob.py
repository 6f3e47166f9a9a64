"""Which pcg_tol keeps the 1e-5 parity bar on the benchmarked unstructured body (blob1m_mix, n = 118)?  GPU(tol) vs GPU(1e-12, verified)
after 2 frames: python experiments/tol_blob.py [n]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import bench, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 118
sc, nt, nv = bench.build_scene(bench.WORKLOADS["blob1m_mix"], n)
def run(tol, mx, verify, frames=2):
    os.environ["ADMM_HIP_OC_VERIFY"] = verify
    s = sc.make_solver(pcg_tol=tol, pcg_max_iters=mx)
    os.environ.pop("ADMM_HIP_OC_VERIFY")
    out = []
    for _ in range(frames):
        s.step(); out.append(s.m_x.copy())
    rd = s.runtime_data(); s.close()
    return out, rd.inner_iters / 20.0
ref, _ = run(1e-12, 3000, "1")
ref13, _ = run(1e-13, 3000, "1")
print("1e-13 vs 1e-12:", [scenes.rel_err(a, b) for a, b in zip(ref13, ref)])
for tol, ver in ((1e-8, "1"), (1e-8, "0"), (3e-9, "0"), (1e-9, "0"), (1e-10, "0")):
    xs, its = run(tol, 1500, ver)
    print("tol %.0e verify %s: rel err per frame" % (tol, ver), ["%.2e" % scenes.rel_err(a, b) for a, b in zip(xs, ref)], "its/solve (last frame) %.2f" % its, flush=True)

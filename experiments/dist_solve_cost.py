"""What one PCG iteration of the DISTRIBUTED solve (ADMM_HIP_DIST_SOLVE=1, launch_pcg_dist) costs on ONE GPU with a world of one
and a real RCCL communicator (ADMM_HIP_FORCE_COMM=1): the kernels of the launch path for the WHOLE body + the in-place
ncclAllReduce calls (identity over one rank: their launch and synchronisation cost, no wire time) + the per-iteration host
read of the convergence flag.  This is the floor of the per-iteration cost of a rank at any N (the kernels shrink with 1/N, the
collectives grow with the wire time); the on-chip solver's figure is printed beside it.  torch is imported FIRST, as in bench.py,
so the run also checks that the library's lazily bound RCCL coexists with torch's at exit.
    python experiments/dist_solve_cost.py [n]"""
import ctypes as C
import os
import sys
import time

import torch  # noqa: F401  (first: see above)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from admm_elastic_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else None
sc, nt, nv = bench.build_scene(dict(bench.WORKLOADS["blob1m_mix"], linsolver=0), n)      # (linsolver 0: inner_iters = PCG iterations)
out = {}
for mode in ("onchip", "dist"):
    if mode == "dist":
        os.environ["ADMM_HIP_FORCE_COMM"] = "1"; os.environ["ADMM_HIP_DIST_SOLVE"] = "1"
    s = sc.make_solver(pcg_tol=bench.PCG_TOL, pcg_max_iters=3000)
    if mode == "dist":
        buf = C.create_string_buffer(128)
        capi.check(capi.lib().admm_hip_comm_unique_id(buf))
        capi.check(capi.lib().admm_hip_comm_init(s._ctx, bytes(buf.raw), 0, 1))
    for _ in range(3):
        s.step()
    t0 = time.perf_counter(); its = 0; frames = 4
    for _ in range(frames):
        s.step(); its += s.runtime_data().inner_iters
        assert s.runtime_data().unconverged_solves == 0
    dt = time.perf_counter() - t0
    out[mode] = (1e3 * dt / frames, its / frames)
    print("%-7s %d tets: %.2f ms/frame, %.0f PCG iterations/frame, %.1f us per PCG iteration (whole frame / iterations)" % (mode, nt, out[mode][0], out[mode][1], 1e3 * out[mode][0] / out[mode][1]), flush=True)
    s.close()

#!/bin/bash
# round 3, final code: PCG iterations of every solve of a frame (steady state)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
python - <<'PY'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import bench, numpy as np
for wl in ("blob1m_mix", "cube1m_mix"):
    sc, nt, nv = bench.build_scene(bench.WORKLOADS[wl], None)
    s = sc.make_solver(pcg_tol=1e-8, pcg_max_iters=600)
    s.upload()
    for f in range(9):
        s.step_device(stats=True)
        if f >= 5: print(wl, "frame", f, "iterations per solve:", list(s.runtime_data().pcg_iters_per_solve)[:20], "sum", sum(list(s.runtime_data().pcg_iters_per_solve)[:20]))
PY

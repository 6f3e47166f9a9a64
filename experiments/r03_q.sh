#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_edge_cases.py -x -q -m gpu -k "not big_rot and not big_rest and not big_trans" > gpurun_out/r03/q_tests.txt 2>&1
tail -4 gpurun_out/r03/q_tests.txt
STEPS=10 bash experiments/env_ab.sh "blob1m_mix cube1m_mix" "ADMM_HIP_RC_PREV=0" "X=1" > gpurun_out/r03/q_ab.txt 2>&1
cat gpurun_out/r03/q_ab.txt
python - <<'PY'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import bench, numpy as np
for wl in ("blob1m_mix", "cube1m_mix"):
    sc, nt, nv = bench.build_scene(bench.WORKLOADS[wl], None)
    s = sc.make_solver(pcg_tol=1e-8, pcg_max_iters=600)
    s.upload()
    for f in range(4):
        s.step_device(stats=True)
        print(wl, "frame", f, "iterations per solve:", s.runtime_data().pcg_iters_per_solve, "sum", sum(s.runtime_data().pcg_iters_per_solve))
PY

#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
for b in 0 17 50 101 128 160 200 255; do
  echo "== block $b"
  ADMM_HIP_OC_PROF_BLOCK=$b python experiments/oc_prof.py blob1m_mix 2>&1 | grep "n=" | tail -2
done > gpurun_out/r03/p_blocks.txt
cat gpurun_out/r03/p_blocks.txt
python - <<'PY'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import bench, numpy as np
sc, nt, nv = bench.build_scene(bench.WORKLOADS["blob1m_mix"], None)
s = sc.make_solver(pcg_tol=1e-8, pcg_max_iters=600)
s.upload()
for _ in range(3): s.step_device(stats=True)
print("iterations per solve of a frame:", s.runtime_data().pcg_iters_per_solve)
PY

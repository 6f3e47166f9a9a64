#!/bin/bash
# round 3: Binv recomputed from the rest positions inside the local step (-DADMM_TET_REST=1) vs streamed (A/B, same box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
STEPS=10 bash experiments/ab_libs.sh "blob1m_mix cube1m_mix" "cur=" "rest=-DADMM_TET_REST=1" > gpurun_out/r03/r_ab.txt 2>&1
cat gpurun_out/r03/r_ab.txt
for name in cur rest; do
ADMM_HIP_LIB=/tmp/ab/$name.so python - <<'PY'
import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import bench, numpy as np
sc, nt, nv = bench.build_scene(bench.WORKLOADS["blob1m_mix"], None)
s = sc.make_solver(pcg_tol=1e-11, pcg_max_iters=600)
s.upload()
for f in range(2): s.step_device(stats=True)
x, v = s.download()
np.save("/tmp/x_%s.npy" % os.path.basename(os.environ["ADMM_HIP_LIB"]), x)
print(os.environ["ADMM_HIP_LIB"], float(np.abs(x).sum()))
PY
done
python -c "
import numpy as np
a=np.load('/tmp/x_cur.so.npy'); b=np.load('/tmp/x_rest.so.npy'); print('max |x_cur - x_rest| after 2 frames at tol 1e-11:', float(np.abs(a-b).max()))"

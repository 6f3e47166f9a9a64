#!/bin/bash
# round 3, first call: what do the recycled pairs buy with the two-level solver (0 / 1 / 2 / 4 pairs, recycling off)?  + phase table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
STEPS=10 bash experiments/env_ab.sh "blob1m_mix" "ADMM_HIP_NO_RECYCLE=1" "ADMM_HIP_RC_PAIRS=0" "ADMM_HIP_RC_PAIRS=1" "ADMM_HIP_RC_PAIRS=2" "ADMM_HIP_RC_PAIRS=4" > gpurun_out/r03/a_pairs_blob.txt 2>&1
for cfg in "ADMM_HIP_NO_RECYCLE=1" "ADMM_HIP_RC_PAIRS=1" "ADMM_HIP_RC_PAIRS=4"; do
  env $cfg python bench.py --workload cube1m_mix --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('roofline_global') or {}
print('[$cfg]', 'cube1m_mix', '| it/s', round(d['value'],1), 'its/solve', g.get('iterations_per_solve'), 'solve us', round(g.get('solve_us',0),1), 'unconverged', d.get('unconverged_solves_in_timed_region'))"
done > gpurun_out/r03/a_pairs_cube.txt 2>&1
python experiments/oc_prof.py blob1m_mix > gpurun_out/r03/a_ocprof_blob.txt 2>&1
ADMM_HIP_NO_RECYCLE=1 python experiments/oc_prof.py blob1m_mix > gpurun_out/r03/a_ocprof_blob_norc.txt 2>&1
cat gpurun_out/r03/a_pairs_blob.txt gpurun_out/r03/a_pairs_cube.txt; tail -8 gpurun_out/r03/a_ocprof_blob.txt; tail -8 gpurun_out/r03/a_ocprof_blob_norc.txt

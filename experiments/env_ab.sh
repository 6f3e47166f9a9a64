#!/bin/bash
# Same-box A/B of environment switches: bash experiments/env_ab.sh "<workloads>" "ENV1=.. ENV2=.." "ENV=.." ...   (2 rounds)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
WLS=$1; shift
for rep in 1 2; do
for cfg in "$@"; do
  for w in $WLS; do
  env $cfg python bench.py --workload $w --steps ${STEPS:-10} --warmup ${WARMUP:-5} --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('roofline_global') or {}
print('[$cfg]', '$w', '| it/s', round(d['value'],1), 'its/solve', g.get('iterations_per_solve'), 'solve us', round(g.get('solve_us',0),1), 'local us', round(1000*d['split_ms_per_admm_iter']['local'],1), 'rhs us', round(1000*d['split_ms_per_admm_iter']['rhs'],1), 'unconverged', d.get('unconverged_solves_in_timed_region'))"
  done
done
done

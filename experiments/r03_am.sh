#!/bin/bash
# round 3: the 1024-thread instance of k_pcg2 (more than 768 rows per block: bodies above ~1.1 M tets) next to the 768-thread one
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
for n in 118 124 130; do
  ADMM_BENCH_N=$n python bench.py --workload blob1m_mix --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['roofline_global']
print('n=$n', d['config']['elements'], 'tets', d['config']['verts'], 'verts | it/s', round(d['value'],1), 'its/solve', g['iterations_per_solve'], 'solve us', round(g['solve_us'],1), 'local us', round(1000*d['split_ms_per_admm_iter']['local'],1), 'unconv', d['unconverged_solves_in_timed_region'], g['kernel'][:20])"
done

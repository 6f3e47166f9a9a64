#!/bin/bash
# Times the NH local step for compile-time variants of the Newton driver (kernel experiments; results of
# non-default variants are NOT parity-checked).  Usage: bash experiments/local_variants.sh "<flags A>" "<flags B>" ...
export OMP_NUM_THREADS=8
cd "$GRAFT_REPO_ROOT"
for flags in "$@"; do
  ADMM_HIP_EXTRA_FLAGS="$flags" python -c "
import sys; sys.path.insert(0,'.')
from admm_elastic_amd import build; build.build_library(force=True)"
  python bench.py --workload cube1m_nh --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$flags', '| local us', round(1000*d['split_ms_per_admm_iter']['local'],1), 'it/s', round(d['value'],1), 'inner', d['inner_iters_per_admm_iter'])"
done

#!/bin/bash
# Like local_variants.sh, on the default (NH + StVK mix) workload.  Usage: bash experiments/local_variants2.sh "<flags A>" ...
export OMP_NUM_THREADS=8
cd "$GRAFT_REPO_ROOT"
for flags in "$@"; do
  ADMM_HIP_EXTRA_FLAGS="$flags" python -c "
import sys; sys.path.insert(0,'.')
from admm_elastic_amd import build; build.build_library(force=True)"
  for w in cube1m_mix cube1m_linear; do
  python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$flags', '$w', '| local us', round(1000*d['split_ms_per_admm_iter']['local'],1), 'rhs us', round(1000*d['split_ms_per_admm_iter']['rhs'],1), 'it/s', round(d['value'],1))"
  done
done

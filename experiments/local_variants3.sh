#!/bin/bash
# Build variants (extra compile flags) and report, on the default workload: local-step time (event pair), VALU instructions per wave.
# Usage: bash experiments/local_variants3.sh "<flags A>" "<flags B>" ...
export OMP_NUM_THREADS=8 TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python -c "import torch" > /dev/null 2>&1
for flags in "$@"; do
  ADMM_HIP_EXTRA_FLAGS="$flags" python -c "
import sys; sys.path.insert(0,'.')
from admm_elastic_amd import build; build.build_library(force=True)" > /dev/null 2>&1
  for w in blob1m_mix; do
  python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$flags]', '$w', '| local us', round(1000*d['split_ms_per_admm_iter']['local'],1), 'rhs us', round(1000*d['split_ms_per_admm_iter']['rhs'],1), 'it/s', round(d['value'],1), 'frac', round(d['roofline']['frac'],3))"
  bash experiments/local_insts.sh $w 2>&1 | grep local_tets
  done
done

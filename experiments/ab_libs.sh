#!/bin/bash
# Same-box A/B of prebuilt libraries ab/libadmm_hip_<name>.so (boxes differ by a few %: only same-box numbers compare).
# Usage: bash experiments/ab_libs.sh "<workloads>" name1 name2 ...   (2 rounds, interleaved)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
WLS=$1; shift
cp admm-elastic_amd/libadmm_hip.so /tmp/keep.so
for rep in 1 2; do
for name in "$@"; do
  cp ab/libadmm_hip_$name.so admm-elastic_amd/libadmm_hip.so
  for w in $WLS; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$name]', '$w', '| local us', round(1000*d['split_ms_per_admm_iter']['local'],2), 'rhs us', round(1000*d['split_ms_per_admm_iter']['rhs'],1), 'it/s', round(d['value'],1), 'frac', round(d['roofline']['frac'],3))"
  done
done
done
cp /tmp/keep.so admm-elastic_amd/libadmm_hip.so

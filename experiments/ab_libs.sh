#!/bin/bash
# Same-box A/B of library VARIANTS (boxes differ by a few %: only same-box numbers compare).  The variants are built ON THE
# BOX, out of tree (/tmp/ab/<name>.so), from the current sources with extra compiler flags, and loaded through ADMM_HIP_LIB:
# the in-tree library is never overwritten.
# Usage: bash experiments/ab_libs.sh "<workloads>" "name1=<flags>" "name2=<flags>" ...   (2 rounds, interleaved; "cur=" = no flags)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
WLS=$1; shift
mkdir -p /tmp/ab
names=""
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ADMM_HIP_EXTRA_FLAGS="$flags" python -c "
import sys; sys.path.insert(0,'.')
from admm_elastic_amd import build; build.build_library(out='/tmp/ab/$name.so')" > /dev/null 2>&1 || echo "build of $name failed"
  echo "$name = [$flags]"
  names="$names $name"
done
for rep in 1 2; do
for name in $names; do
  for w in $WLS; do
  ADMM_HIP_LIB=/tmp/ab/$name.so python bench.py --workload $w --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('roofline_global') or {}
print('[$name]', '$w', '| local us', round(1000*d['split_ms_per_admm_iter']['local'],2), 'rhs us', round(1000*d['split_ms_per_admm_iter']['rhs'],1), 'solve us', round(g.get('solve_us',0),1), 'its/solve', g.get('iterations_per_solve'), 'it/s', round(d['value'],1), 'frac', round(d['roofline']['frac'],3), 'unconv', d.get('unconverged_solves_in_timed_region'))"
  done
done
done

#!/bin/bash
# A/B of compile-time variants of the on-chip PCG (kernel experiments).  Usage: bash experiments/oc_variants.sh "<flags A>" "<flags B>" ...
export OMP_NUM_THREADS=8
cd "$GRAFT_REPO_ROOT"
for flags in "$@"; do
  ADMM_HIP_EXTRA_FLAGS="$flags" python -c "
import sys; sys.path.insert(0,'.')
from admm_elastic_amd import build; build.build_library(force=True)"
  for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$flags', '| it/s', round(d['value'],1), 'global ms', round(d['split_ms_per_admm_iter']['global'],4), 'inner', round(d['inner_iters_per_admm_iter'],1), 'unconv', d['unconverged_solves_in_timed_region'])"; done
  python experiments/oc_stats.py 2>&1 | grep "\[oc\]" | python -c "
import sys,re
tot=pip=n=vf=phs=0
for l in sys.stdin:
    m=re.search(r'iters (\d+) \(pipelined (\d+)\) verifications (\d+) .*? phases (\d+)', l)
    if m:
        a,b,c,d=map(int,m.groups()); tot+=a; pip+=b; n+=1; vf+=c; phs+=d
print('   solves',n,'iterations',tot,'CG-CG',tot-pip,'verifications',vf,'extra phases per solve',(phs-tot)/max(n,1))"
done

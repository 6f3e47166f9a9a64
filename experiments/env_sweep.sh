#!/bin/bash
# Same-box sweep of environment knobs of the on-chip PCG on a bench workload: bash experiments/env_sweep.sh <workload> "<VAR=val ...>" "<...>" ...
cd $GRAFT_REPO_ROOT
WL=$1; shift
python -c "import torch" > /dev/null 2>&1
for cfg in "" "$@"; do
  line=$(env $cfg python bench.py --workload $WL --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1)
  python - "$cfg" <<PY
import json, sys
d = json.loads('''$line''') if '''$line''' else None
print("%-44s" % (sys.argv[1] or "(default)"), "ERR" if d is None else "%.0f ADMM it/s  its/solve %.2f  global %.3f ms  unconv %d" % (d["value"], d["inner_iters_per_admm_iter"], d["split_ms_per_admm_iter"]["global"], d["unconverged_solves_in_timed_region"]))
PY
done

#!/bin/bash
# Round 6, fourth session: an ERROR-shaped stop rule for k_pcg2 -- a pass ends on sum_axes r . M^-1 r / b . D^-1 b <= kappa tol^2
# (gamma of the iteration's record, the two-level preconditioner's norm) instead of the Jacobi norm r . D^-1 r: 200-frame drift and iterations per solve by kappa
# (ADMM_HIP_OC_STOP_M=kappa, experiments only), against the bench setting.  + same-box A/B that the switch itself (off) costs nothing.
cd "$(dirname "$0")/.." || exit 1
R=$PWD
O=$R/gpurun_out/r06stopm; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
export ADMM_HIP_LIB=$R/experiments/_build/oc_stopm.so
ADMM_DRIFT_FRAMES=${FRAMES:-200} ADMM_DRIFT_VARIANTS="${VARIANTS:-7e-10:SOFTSET=24;7e-10:SOFTSET=24:ADMM_HIP_OC_STOP_M=1;7e-10:SOFTSET=24:ADMM_HIP_OC_STOP_M=4;7e-10:SOFTSET=24:ADMM_HIP_OC_STOP_M=16;7e-10:SOFTSET=24:ADMM_HIP_OC_STOP_M=64;7e-10:SOFTSET=24:ADMM_HIP_OC_STOP_M=256;7e-10:SOFTSET=24:ADMM_HIP_OC_STOP_M=1024}" timeout 1500 python experiments/r05_drift.py 2>&1 | grep -v "^\[" | tee $O/drift.txt
unset ADMM_HIP_LIB
summ() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s value %.1f  ms/frame %.3f  inner timed %.3f  unconv %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["inner_iters_per_admm_iter"], d.get("unconverged_solves_in_timed_region")))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for rep in 1 2; do
  for v in oc_base oc_stopm; do
    ADMM_HIP_LIB=$R/experiments/_build/$v.so timeout 300 python bench.py --workload blob1m_mix --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_${v}_$rep.json; summ $O/bench_${v}_$rep.json "[$v]"
  done
done | tee $O/ab.txt

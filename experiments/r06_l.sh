#!/bin/bash
# Round 6, third session: k_gs_persist with polls IN FLIGHT (ADMM_GSP_PIPE; gs_persist.hpp).  Same-box A/B of prebuilt variants
# (experiments/build_variant.sh -> experiments/_build/<name>.so, loaded through ADMM_HIP_LIB), two interleaved rounds, + the phase split.
# usage: r06_l.sh "<variant names>" ["<workloads>"]
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${OUT:-r06l}; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
NAMES=$1; WLS=${2:-"cube100k_gs cloth200k_gs_floor"}
for rep in 1 2; do
for name in $NAMES; do
  for w in $WLS; do
  ADMM_HIP_LIB=$PWD/experiments/_build/$name.so timeout 300 python bench.py --workload $w --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_${name}_${w}_$rep.json
  python - $O/bench_${name}_${w}_$rep.json $name $w <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("[%s] %s it/s %.1f ms/frame %.3f split %s finite %s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], {k: round(1000 * v, 1) for k, v in d["split_ms_per_admm_iter"].items()}, d.get("finite")))
except Exception as e:
    print("[%s] %s no line: %s" % (sys.argv[2], sys.argv[3], e))
PY
  done
done
done | tee $O/ab.txt
for name in $NAMES; do
  for w in $WLS; do
    echo "[$name] $w" >> $O/gspprof.txt
    ADMM_HIP_LIB=$PWD/experiments/_build/$name.so ADMM_HIP_GSP_PROF=1 ADMM_HIP_GSP_PROF_BLOCK=20 timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep gsp_prof | tail -2 >> $O/gspprof.txt
  done
done
cat $O/gspprof.txt

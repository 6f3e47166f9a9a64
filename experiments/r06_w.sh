#!/bin/bash
# Round 6, third session: rows per block of the persistent GS plan (ADMM_HIP_GS_ROWS) with the two-granule / whole-sector hand-off
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06w; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
for rep in 1 2; do
for rows in 96 128 192 256 320 384 448; do
  for w in cube100k_gs; do
  ADMM_HIP_GS_ROWS=$rows ADMM_HIP_OC_DIAG=1 timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2> $O/err.txt | tail -1 > $O/bench_${rows}_${w}_$rep.json
  python - $O/bench_${rows}_${w}_$rep.json $rows $w <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("[rows %s] %s it/s %.1f ms/frame %.3f split %s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], {k: round(1000 * v, 1) for k, v in d["split_ms_per_admm_iter"].items()}))
except Exception as e:
    print("[rows %s] %s no line: %s" % (sys.argv[2], sys.argv[3], e))
PY
  grep gs_plan $O/err.txt | tail -1
  done
done
done | tee $O/sweep.txt
for rows in 256 320 384 448; do ADMM_HIP_GS_ROWS=$rows timeout 300 python bench.py --workload cloth200k_gs_floor --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[rows $rows] cloth200k_gs_floor it/s %.1f' % d['value'])"; done | tee -a $O/sweep.txt

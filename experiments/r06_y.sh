#!/bin/bash
# Round 6, third session: the persistent GS plan with TWO blocks per CU for bodies beyond 256 x 192 rows (ADMM_HIP_GS_BLOCKS_PER_CU=1|2, in-tree library)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06y; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
for rep in 1 2; do
for bpc in 1 2; do
  for w in cloth200k_gs_floor cube100k_gs; do
  ADMM_HIP_GS_BLOCKS_PER_CU=$bpc ADMM_HIP_OC_DIAG=1 timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2> $O/err.txt | tail -1 > $O/bench_${bpc}_${w}_$rep.json
  python - $O/bench_${bpc}_${w}_$rep.json $bpc $w <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("[blocks per CU %s] %s it/s %.1f ms/frame %.3f split %s finite %s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], {k: round(1000 * v, 1) for k, v in d["split_ms_per_admm_iter"].items()}, d.get("finite")))
except Exception as e:
    print("[blocks per CU %s] %s no line: %s" % (sys.argv[2], sys.argv[3], e))
PY
  grep gs_plan $O/err.txt | tail -1
  done
done
done | tee $O/ab.txt
for bpc in 1 2; do echo "[blocks per CU $bpc]"; ADMM_HIP_GS_BLOCKS_PER_CU=$bpc ADMM_HIP_GSP_PROF=1 ADMM_HIP_GSP_PROF_BLOCK=20 timeout 300 python bench.py --workload cloth200k_gs_floor --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep gsp_prof | tail -2; done | tee $O/gspprof.txt
timeout 1200 python -m pytest tests/test_gs_persist.py tests/test_bench_parity.py tests/test_gpu_parity.py -m gpu -q -x -k "gs or cloth or floor" > $O/t_gs.txt 2>&1; tail -3 $O/t_gs.txt

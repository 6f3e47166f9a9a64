#!/bin/bash
# round-5, UzawaCG look-ahead columns: tests, then the touchdown-window bench with and without (ADMM_HIP_UZ_AHEAD=0)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05x2
rm -rf $O; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -x -q -k "uzawa" > $O/uz_tests.txt 2>&1; grep -n "passed\|failed\|^E " $O/uz_tests.txt | tail -8
for A in 4 0 2 8; do
  ADMM_HIP_UZ_LANES_DEBUG=1 ADMM_HIP_UZ_AHEAD=$A timeout 200 python bench.py --workload cube100k_uzawa_floor --steps 24 --warmup 1 --no-cpu-baseline 2> $O/err_ahead$A.txt | tail -1 > $O/bench_touchdown_window_ahead$A.json
  grep "uz_ahead\|uz_lanes" $O/err_ahead$A.txt | head -5
done
ADMM_HIP_UZ_AHEAD_LANES=6 timeout 200 python bench.py --workload cube100k_uzawa_floor --steps 24 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_touchdown_window_ahead4_lanes6.json
timeout 200 python bench.py --workload cube100k_uzawa_floor --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_20_5.json
for f in $O/bench_*.json; do echo $(basename $f); python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  value %.1f  ms/frame %.2f  median it/s %.1f  uzawa %s  rows %s" % (d["value"], d["ms_per_step"], d.get("median_admm_it_per_s_statistics_frames") or 0, d.get("uzawa"), d.get("rows_projected_in_timed_region")))
PY
done

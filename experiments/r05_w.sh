#!/bin/bash
# round-5, UzawaCG column lanes: bench lines of cube100k_uzawa_floor (the usual window; a window that holds the first touchdown, with
# the lanes and on the main stream alone) and the rocprofv3 kernel stats of the touchdown window.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05w
rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
python bench.py --workload cube100k_uzawa_floor --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_cube100k_uzawa_floor.json
python bench.py --workload cube100k_uzawa_floor --steps 24 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cube100k_uzawa_floor_touchdown_window.json
ADMM_HIP_UZ_LANES=1 python bench.py --workload cube100k_uzawa_floor --steps 24 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cube100k_uzawa_floor_touchdown_window_one_stream.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $GRAFT_REPO_ROOT/bench.py --workload cube100k_uzawa_floor --steps 24 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof_touchdown_window.json 2> $O/stats_err.txt )
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_cube100k_uzawa_floor_touchdown_window.csv
rm -rf $O/stats
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  value %.1f  ms/frame %.2f  median it/s %.1f  uzawa %s  rows %s" % (d["value"], d["ms_per_step"], d.get("median_admm_it_per_s_statistics_frames") or 0, d.get("uzawa"), d.get("rows_projected_in_timed_region")))
PY
done
head -6 $O/kernel_stats_cube100k_uzawa_floor_touchdown_window.csv | cut -c1-200

#!/bin/bash
# refresh of the GS evidence after the reciprocal-diagonal change (the rest of r05_profiles.sh is unaffected)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05p2; rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
for wl in cube100k_gs cloth200k_gs_floor; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof_$wl.json 2> $O/stats_$wl.err )
  cp $(find $O/stats_$wl -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$wl.csv
  rm -rf $O/stats_$wl
  python bench.py --workload $wl --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_$wl.json
  ADMM_HIP_GSP_PROF=1 ADMM_HIP_GSP_PROF_BLOCK=20 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep gsp_prof | tail -2 > $O/gspprof_$wl.txt
done
ls $O

#!/bin/bash
# round-2 evidence: rocprofv3 kernel stats of the default bench (unstructured body) and the cube, PMC passes (FETCH_SIZE and
# WRITE_SIZE separately, --kernel-trace only), bench JSON lines of every workload.  Writes under gpurun_out/r02/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02
mkdir -p $O
python -c "import torch" > /dev/null 2>&1
for wl in blob1m_mix cube1m_mix; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof_$wl.json 2> $O/stats_$wl.err )
done
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${C} -o p -- python $GRAFT_REPO_ROOT/bench.py --workload blob1m_mix --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> $O/pmc_$C.err )
done
for wl in blob1m_mix cube1m_mix cube1m_nh cube100k_gs cloth200k_gs_floor cube100k_uzawa_floor; do
  python bench.py --workload $wl --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_$wl.json
done
find $O -name "*kernel_stats.csv" | head; find $O -name "*counter_collection.csv" | head
ls -la $O | head -30

#!/bin/bash
# round-3 evidence: rocprofv3 kernel stats of the default bench (unstructured body) and the cube, PMC passes (FETCH_SIZE and
# WRITE_SIZE separately, --kernel-trace only), the phase table of the persistent PCG kernel, bench JSON lines of every workload.
# Writes under gpurun_out/r03p/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03p
rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
for wl in blob1m_mix cube1m_mix; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof_$wl.json 2> $O/stats_$wl.err )
done
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${C} -o p -- python $GRAFT_REPO_ROOT/bench.py --workload blob1m_mix --steps 2 --warmup 3 --no-cpu-baseline --no-roofline > $O/pmc_bench_$C.json 2> $O/pmc_$C.err )
done
for wl in blob1m_mix cube1m_mix cube1m_nh cube100k_gs cloth200k_gs_floor cube100k_uzawa_floor; do
  python bench.py --workload $wl --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_$wl.json
done
python bench.py 2>/dev/null | tail -1 > $O/bench_default_driver_flags.json
for b in 0 100; do ADMM_HIP_OC_PROF_BLOCK=$b python experiments/oc_prof.py blob1m_mix 2>&1 | grep oc_prof | tail -6 > $O/ocprof_blob_block$b.txt; done
find $O -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head

#!/bin/bash
# round-5 evidence, final code: PMC passes first (FETCH_SIZE and WRITE_SIZE separately, --kernel-trace only) -> the JSON bench.py takes
# `roofline.traffic` from; rocprofv3 kernel stats of the default bench, the GS workloads, the UzawaCG workload and the launch-path PCG at
# 2 M tets; phase tables of the persistent kernels; bench JSON lines of every workload (contact regimes included); the size curve.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05p
rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${C} -o p -- python $GRAFT_REPO_ROOT/bench.py --workload blob1m_mix --steps 2 --warmup 3 --no-cpu-baseline --no-roofline > $O/pmc_bench_$C.json 2> $O/pmc_$C.err )
done
python experiments/pmc_to_json_r03.py $O $O/pmc_hbm_blob1m.json blob1m_mix
cp $O/pmc_hbm_blob1m.json profiles/r05_e_pmc_hbm_blob1m.json      # (this box's copy of the repo: the bench lines below quote it)
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
for wl in blob1m_mix cube100k_gs cloth200k_gs_floor cube100k_uzawa_floor; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof_$wl.json 2> $O/stats_$wl.err )
  cp $(find $O/stats_$wl -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$wl.csv
  if [ $wl = blob1m_mix ]; then python experiments/pcg2_from_trace.py $O/stats_$wl $O/kernel_trace_split_blob1m_mix.txt; fi
done
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_blob2m -o p -- python $GRAFT_REPO_ROOT/bench.py --workload blob1m_mix --n 148 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_under_rocprof_blob2m.json 2> $O/stats_blob2m.err )
cp $(find $O/stats_blob2m -name "*kernel_stats.csv" | head -1) $O/kernel_stats_blob2m_launch_path.csv
rm -rf $O/stats_*
for wl in blob1m_mix cube1m_mix cube1m_nh cube100k_gs cloth200k_gs_floor cube100k_uzawa_floor; do
  python bench.py --workload $wl --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_$wl.json
done
python bench.py 2>/dev/null | tail -1 > $O/bench_default_driver_flags.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --soft-modes 0 --pcg-tol 5e-10 2>/dev/null | tail -1 > $O/bench_blob1m_mix_round4_settings.json
for b in 0 100; do ADMM_HIP_OC_PROF_BLOCK=$b python experiments/oc_prof.py blob1m_mix 2>&1 | grep oc_prof | tail -8 > $O/ocprof_blob_block$b.txt; done
for wl in cube100k_gs cloth200k_gs_floor; do ADMM_HIP_GSP_PROF=1 ADMM_HIP_GSP_PROF_BLOCK=20 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep gsp_prof | tail -2 > $O/gspprof_$wl.txt; done
bash experiments/r05_size_curve.sh $O/size > $O/size_log.txt 2>&1
cp $O/size/size_curve.txt $O/size_curve.txt
ls -la $O

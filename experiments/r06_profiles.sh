#!/bin/bash
# Round-6 evidence, final code: PMC passes (FETCH_SIZE and WRITE_SIZE separately, --kernel-trace only) restricted to the ADMM loop -> the JSON
# bench.py takes `roofline.traffic` from; rocprofv3 kernel stats of the default bench (whole run AND the ADMM loop alone) and of the other
# workloads; phase tables of the persistent kernels; bench JSON lines of every workload at its own settings.
cd "$(dirname "$0")/.." || exit 1
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r06p
rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${C} -o p -- python $R/bench.py --workload blob1m_mix --steps 2 --warmup 3 --no-cpu-baseline --no-roofline > $O/pmc_bench_$C.json 2> $O/pmc_$C.err )
done
python experiments/pmc_to_json_r06.py $O $O/pmc_hbm_blob1m.json blob1m_mix
cp $O/pmc_hbm_blob1m.json profiles/r06_e_pmc_hbm_blob1m.json      # (this box's copy of the repo: the bench lines below quote it)
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
for wl in blob1m_mix cube1m_nh cube100k_gs cloth200k_gs_floor; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -o p -- python $R/bench.py --workload $wl --steps 5 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof_$wl.json 2> $O/stats_$wl.err )
  cp $(find $O/stats_$wl -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$wl.csv
  python experiments/loop_stats_from_trace.py $O/stats_$wl $O/kernel_stats_admm_loop_$wl.csv
  if [ $wl = blob1m_mix ]; then python experiments/pcg2_from_trace.py $O/stats_$wl $O/kernel_trace_split_blob1m_mix.txt; fi
done
rm -rf $O/stats_*
for wl in blob1m_mix cube1m_mix cube1m_nh cube100k_gs cloth200k_gs_floor cube100k_uzawa_floor; do
  python bench.py --workload $wl --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_$wl.json
done
python bench.py 2>/dev/null | tail -1 > $O/bench_default_driver_flags.json
for b in 0 100; do ADMM_HIP_OC_PROF_BLOCK=$b python experiments/oc_prof.py blob1m_mix 2>&1 | grep oc_prof | tail -8 > $O/ocprof_blob_block$b.txt; done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys, os
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-44s value %.1f  ms/frame %.3f  inner %.3f  roofline.frac %s  global.frac %s" % (os.path.basename(sys.argv[1]), d["value"], d["ms_per_step"], d["inner_iters_per_admm_iter"],
          (d.get("roofline") or {}).get("frac"), (d.get("roofline_global") or {}).get("frac")))
except Exception as e:
    print(os.path.basename(sys.argv[1]), "no line", e)
PY
done
ls -la $O

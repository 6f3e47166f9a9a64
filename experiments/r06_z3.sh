#!/bin/bash
# Round 6, fourth session: launch path -- the end projection on the soft modes from the solve's own residual (k_defl_dots_r) and k_defl_solve with
# its partials in flight, vs the product form (ADMM_HIP_DEFL_RESID=0); tests of the launch path, the soft modes and the distributed solve.
cd "$(dirname "$0")/.." || exit 1
R=$PWD
O=$R/gpurun_out/r06z3; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
timeout 1200 python -m pytest tests/test_big_pcg.py tests/test_soft_modes.py -m gpu -q -x > $O/t_big_soft.txt 2>&1; tail -3 $O/t_big_soft.txt
timeout 1500 python -m pytest tests/test_multi_gpu.py -m gpu -q -x -k "distributed_solve" > $O/t_dist.txt 2>&1; tail -3 $O/t_dist.txt
summ() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-34s %8d tets: %7.1f ADMM it/s, %6.2f ms/frame, %5.2f inner its per ADMM it, split %s, unconverged %s" % (sys.argv[2], d["config"]["elements"], d["value"], d["ms_per_step"],
          d["inner_iters_per_admm_iter"], {k: round(v, 3) for k, v in d["split_ms_per_admm_iter"].items()}, d.get("unconverged_solves_in_timed_region")))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for rep in 1 2; do
  for n in 148 187; do
    for v in "ADMM_HIP_DEFL_RESID=0" "ADMM_HIP_DEFL_RESID=1"; do
      env $v timeout 900 python bench.py --workload blob1m_mix --n $n --steps 10 --warmup 5 --no-cpu-baseline 2> $O/err_$n.txt | tail -1 > $O/bench_${n}_${v}_$rep.json
      summ $O/bench_${n}_${v}_$rep.json "[$v] n=$n"
    done
  done
done | tee $O/ab.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_148 -o p -- python $R/bench.py --workload blob1m_mix --n 148 --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof_148.json 2> $O/stats_148.err )
python experiments/loop_stats_from_trace.py $O/stats_148 $O/kernel_stats_admm_loop_148.csv
rm -rf $O/stats_148
head -24 $O/kernel_stats_admm_loop_148.csv

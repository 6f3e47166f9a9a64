#!/bin/bash
# Round 6, fourth session: launch-path two-level PCG (bodies beyond the chip) -- first chunk of a solve = what the solve at the same position of
# the previous frame needed, short chunks behind it (default) vs the fixed chunks of 8 of round 5 (ADMM_HIP_BIG_CHUNK=8); rocprofv3 kernel
# statistics of the 2 M-tet body (where do the 0.6 ms per solve outside the iterations go?); the launch-path tests.
cd "$(dirname "$0")/.." || exit 1
R=$PWD
O=$R/gpurun_out/r06z; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
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
    for v in "ADMM_HIP_BIG_CHUNK=8" "ADMM_HIP_BIG_TAIL=2" "ADMM_HIP_BIG_TAIL=3"; do
      [ $n = 187 ] && [ "$v" = "ADMM_HIP_BIG_TAIL=3" ] && continue
      env $v timeout 900 python bench.py --workload blob1m_mix --n $n --steps 10 --warmup 5 --no-cpu-baseline 2> $O/err_$n.txt | tail -1 > $O/bench_${n}_${v}_$rep.json
      summ $O/bench_${n}_${v}_$rep.json "[$v] n=$n"
    done
  done
done | tee $O/ab.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_148 -o p -- python $R/bench.py --workload blob1m_mix --n 148 --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof_148.json 2> $O/stats_148.err )
cp $(find $O/stats_148 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_148.csv
python experiments/loop_stats_from_trace.py $O/stats_148 $O/kernel_stats_admm_loop_148.csv
rm -rf $O/stats_148
head -30 $O/kernel_stats_admm_loop_148.csv
timeout 900 python -m pytest tests/test_big_pcg.py -m gpu -q -x > $O/t_big.txt 2>&1; tail -3 $O/t_big.txt

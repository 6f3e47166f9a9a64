#!/bin/bash
# launch-path two-level PCG after the last-block reductions / loads-first vector kernel / new soft-mode kernels: tests, kernel stats at 2 M and 4 M tets
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05m; rm -rf $O; mkdir -p $O
python -m pytest tests/test_big_pcg.py tests/test_soft_modes.py "tests/test_multi_gpu.py::test_distributed_solve_matches_single_context" tests/test_multi_gpu.py::test_distributed_solve_collectives_over_rccl_on_one_gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5 > $O/tests.txt
for n in 148 187; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$n -o p -- python $GRAFT_REPO_ROOT/bench.py --workload blob1m_mix --n $n --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_under_rocprof_n$n.json 2> $O/stats_$n.err )
  cp $(find $O/stats_$n -name "*kernel_stats.csv" | head -1) $O/kernel_stats_n$n.csv
  rm -rf $O/stats_$n
  python bench.py --workload blob1m_mix --n $n --steps 6 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_n$n.json
done
for ag in 700 512; do ADMM_HIP_BIG_AGGREGATES=$ag python bench.py --workload blob1m_mix --n 187 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_n187_agg$ag.json; done
cat $O/tests.txt

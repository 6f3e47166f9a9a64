#!/bin/bash
# k_pcg2 epilogue with all loads issued up front: tests, phase table, bench (two runs)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05n; rm -rf $O; mkdir -p $O
python -m pytest tests/test_soft_modes.py tests/test_bench_parity.py::test_blob1m_two_frames_vs_oracle_exact_solves -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 > $O/tests.txt
ADMM_HIP_OC_PROF_BLOCK=100 python experiments/oc_prof.py blob1m_mix 2>&1 | grep oc_prof | tail -8 > $O/ocprof_block100.txt
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$i.json; done
python bench.py --workload cube1m_mix --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cube.json
cat $O/tests.txt; cat $O/ocprof_block100.txt | cut -c1-200

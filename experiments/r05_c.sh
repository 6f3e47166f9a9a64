#!/bin/bash
# Round 5, third GPU session: drift with the soft-mode end projection, launch-path two-level PCG tests, size curve, whole suite.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_big_pcg.py tests/test_cpp_api.py -m gpu -q -s > $O/t_big.txt 2>&1
tail -25 $O/t_big.txt
ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="4e-10;3e-10;5e-10:SOFT=16;5e-10:SOFT=32;1e-9:SOFT=32;1e-9:SOFT=64;2e-9:SOFT=32;2e-9:SOFT=64;3e-9:SOFT=64;5e-9:SOFT=64" timeout 1800 python experiments/r05_drift.py > $O/drift_soft.txt 2>&1
cat $O/drift_soft.txt
bash experiments/r05_size_curve.sh $O/size > $O/size_log.txt 2>&1
cat $O/size/size_curve.txt
timeout 3000 python -m pytest tests -m gpu -q --deselect tests/test_bench_parity.py::test_blob1m_drift_200_frames_bench_tolerance_vs_tight_solve --deselect tests/test_big_pcg.py > $O/suite.txt 2>&1
tail -40 $O/suite.txt

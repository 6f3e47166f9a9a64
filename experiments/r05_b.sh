#!/bin/bash
# Round 5, second GPU session: the whole -m gpu suite on the new code, and the 200-frame drift by tolerance.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="4e-10;3e-10;2e-10" timeout 1500 python experiments/r05_drift.py > $O/drift_200.txt 2>&1
cat $O/drift_200.txt
timeout 3000 python -m pytest tests -m gpu -q -x --deselect tests/test_bench_parity.py::test_blob1m_drift_200_frames_bench_tolerance_vs_tight_solve > $O/suite.txt 2>&1
tail -15 $O/suite.txt

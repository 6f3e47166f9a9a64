#!/bin/bash
# Round 5, fourth GPU session: profile of the launch-path two-level PCG at 2 M tets, the drift floor at 200 frames, soft modes on the last solve only,
# re-run of the tests that failed.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05d; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_big_pcg.py tests/test_cpp_api.py "tests/test_gpu_parity.py::test_onchip_pcg_unconverged_and_cloth" "tests/test_gpu_parity.py::test_onchip_pcg_preconditioner_modes" "tests/test_gpu_parity.py::test_onchip_pcg_matches_launch_path_and_exact" "tests/test_samples.py::test_curtain_sample_bending_slide_stable_nh" tests/test_multi_gpu.py -m gpu -q -s > $O/t_fix.txt 2>&1
tail -30 $O/t_fix.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof_big -- python $OLDPWD/bench.py --workload blob1m_mix --n 148 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OLDPWD/$O/prof_big.log 2>&1)
f=$(find $O/prof_big -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="3e-10;2e-10;1e-10;1e-11;5e-10:SOFT=32:ADMM_HIP_DEFL_EVERY=20;5e-10:SOFT=32:ADMM_HIP_DEFL_EVERY=5;1e-9:SOFT=32:ADMM_HIP_DEFL_EVERY=20" timeout 1800 python experiments/r05_drift.py > $O/drift_floor.txt 2>&1
cat $O/drift_floor.txt | cut -c1-330

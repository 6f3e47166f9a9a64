#!/bin/bash
# Round 5, first GPU session: the new tests (f3 terms, solver params, contact counters, parity chain at 1 M tets, 200-frame drift), the bench
# line, and the tolerance-schedule drift experiment.  Everything under gpurun_out/r05a/.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
timeout 1500 python -m pytest tests/test_f3_terms.py tests/test_edge_cases.py tests/test_cpp_api.py tests/test_gs_persist.py "tests/test_samples.py::test_curtain_sample_bending_slide_stable_nh" -m gpu -x -q > $O/t1_new_tests.txt 2>&1
tail -5 $O/t1_new_tests.txt
timeout 1500 python -m pytest tests/test_bench_parity.py -m gpu -x -q -s -k "two_frames or 200_frames or contact_counters" > $O/t2_parity_chain.txt 2>&1
tail -8 $O/t2_parity_chain.txt
cp gpurun_out/drift_blob1m_frames.txt $O/ 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_blob1m.json 2> $O/bench_blob1m.err
tail -c 600 $O/bench_blob1m.json
ADMM_DRIFT_FRAMES=100 ADMM_DRIFT_VARIANTS="5e-10;5e-10:ADMM_HIP_TOL_SCHED=20,20;5e-10:ADMM_HIP_TOL_SCHED=20,20,5,5;5e-10:ADMM_HIP_TOL_SCHED=40,40,10,10,3,3;7e-10:ADMM_HIP_TOL_SCHED=20,20;3e-10:ADMM_HIP_TOL_SCHED=30,30,10,5,2" timeout 1200 python experiments/r05_drift.py > $O/drift_sched.txt 2>&1
cat $O/drift_sched.txt

#!/bin/bash
# Round 6, final code: the whole GPU suite, then the evidence of experiments/r06_profiles.sh again (PMC passes restricted to the ADMM loop, rocprofv3
# kernel statistics, bench lines of every workload, phase tables of k_pcg2) + the phase tables of k_gs_persist + the size curve of one body.
cd "$(dirname "$0")/.." || exit 1
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r06final
rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_test_suite.txt 2>&1; tail -4 $O/gpu_test_suite.txt
bash experiments/r06_profiles.sh > $O/profiles_log.txt 2>&1; tail -12 $O/profiles_log.txt
for wl in cube100k_gs cloth200k_gs_floor; do
  echo "[$wl]"; ADMM_HIP_OC_DIAG=1 ADMM_HIP_GSP_PROF=1 ADMM_HIP_GSP_PROF_BLOCK=20 timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -E "gsp_prof|gs_plan" | tail -3
done > $O/gspprof.txt
cat $O/gspprof.txt
bash experiments/r05_size_curve.sh $O/size > $O/size_log.txt 2>&1; cat $O/size/size_curve.txt

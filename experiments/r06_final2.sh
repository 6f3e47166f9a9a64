#!/bin/bash
# Round 6, the very last code: smoke(), the whole GPU suite, bench lines of every workload, rocprofv3 / PMC / phase tables (experiments/r06_profiles.sh)
cd "$(dirname "$0")/.." || exit 1
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r06final2; rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_test_suite.txt 2>&1; tail -4 $O/gpu_test_suite.txt
bash experiments/r06_profiles.sh > $O/profiles_log.txt 2>&1; grep value $O/profiles_log.txt

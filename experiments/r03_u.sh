#!/bin/bash
# round 3: UzawaCG with cached columns of K^-1: tests, then same-box A/B against ADMM_HIP_UZ_CACHE=0
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_dynamic_collision.py tests/test_samples.py tests/test_edge_cases.py -x -q -m gpu -k "uzawa or collision or torus or boxes or edge" > gpurun_out/r03/u_tests.txt 2>&1
grep -E "passed|failed|Error" gpurun_out/r03/u_tests.txt | tail -5
for cfg in "ADMM_HIP_UZ_COMPACT=0" "ADMM_HIP_UZ_RECYCLE=0" "X=1"; do
  env $cfg python bench.py --workload cube100k_uzawa_floor --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$cfg]', 'it/s', round(d['value'],1), 'ms/frame', round(d['ms_per_step'],2), 'global ms/it', round(d['split_ms_per_admm_iter']['global'],3), d.get('uzawa'), 'finite', d['finite'])"
done

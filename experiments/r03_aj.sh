#!/bin/bash
# round 3: GS with >= 3 colours, residual test fused into the colour kernels (k_gs_colorN) vs colour kernels + residual SpMV, same box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dynamic_collision.py tests/test_samples.py -x -q -m gpu -k "gs or GS or cloth or signorini or boxes or trianglestrain" 2>&1 | tail -3
for rep in 1 2; do for cfg in "ADMM_HIP_GS_THREE_KERNELS=1" "X=1"; do
  env $cfg python bench.py --workload cloth200k_gs_floor --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg] cloth200k_gs_floor it/s', round(d['value'],1), 'global ms', round(d['split_ms_per_admm_iter']['global'],4), 'inner', d['inner_iters_per_admm_iter'])"
done; done

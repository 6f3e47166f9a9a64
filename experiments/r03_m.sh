#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
for cfg in "ADMM_HIP_OC_AFFINE=0" "X=1"; do
env $cfg python bench.py --workload cube100k_uzawa_floor --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg]', 'it/s', round(d['value'],1), 'ms/frame', round(d['ms_per_frame'],1), 'split', d['split_ms_per_admm_iter'], 'schur its/admm', d['inner_iters_per_admm_iter'])"
done
ADMM_HIP_OC_DEBUG=1 python bench.py --workload cube100k_uzawa_floor --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep "^\[oc\]" | tail -45 | awk '{print $3,$4,$5,$6,$7,$8,$9}' | tr '\n' ';'

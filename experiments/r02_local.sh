#!/bin/bash
# local-step check on one box: kernel-level parity tests, then time + instruction counts of the local kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_known_answers.py -q -m gpu -x 2>&1 | tail -3
for w in blob1m_mix cube1m_nh; do
python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '| local us', round(1000*d['split_ms_per_admm_iter']['local'],1), 'rhs us', round(1000*d['split_ms_per_admm_iter']['rhs'],1), 'it/s', round(d['value'],1), 'frac', round(d['roofline']['frac'],3))"
bash experiments/local_insts.sh $w 2>&1 | grep local_tets
done

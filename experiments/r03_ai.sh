#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
for i in 1 2; do python bench.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default flags:', round(d['value'],1), d['steps'], d['warmup'], round(d['roofline']['frac'],3), d['roofline_global']['iterations_per_solve'], round(d['roofline_global']['frac'],3))"; done
python bench.py --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps 10 warmup 2:', round(d['value'],1), d['roofline_global']['iterations_per_solve'])"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "big or unstructured or bit or determin" 2>&1 | tail -1

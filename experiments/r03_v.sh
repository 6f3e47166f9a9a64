#!/bin/bash
# round 3: full GPU suite + smoke + default bench after retiring the round-1 on-chip kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r03/v_tests.txt 2>&1
grep -E "passed|failed|Error" gpurun_out/r03/v_tests.txt | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py 2>/dev/null | tail -1 > gpurun_out/r03/v_bench_default.json
python -c "
import json; d=json.loads(open('gpurun_out/r03/v_bench_default.json').read()); print(d['value'], d['roofline']['frac'], d['roofline_global']['frac'], d['cpu_baseline']['value'])"

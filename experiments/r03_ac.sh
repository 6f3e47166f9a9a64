#!/bin/bash
# round 3: the 8-rank weak-scaling path end to end on ONE GPU (ADMM_BENCH_SHARE_GPU=1: functional only), under the driver's own launcher
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python -c "import torch" > /dev/null 2>&1
ADMM_BENCH_SHARE_GPU=1 ADMM_BENCH_N=30 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/r03/ac_err.txt | tail -1 > gpurun_out/r03/ac_bench8.json
python -c "
import json; d=json.loads(open('gpurun_out/r03/ac_bench8.json').read()); print(d['n_gpus'], d['scaling'], d['config']['workload'], d['config']['parallelism'], round(d['value'],1), d['unconverged_solves_in_timed_region'], d['finite'], d.get('expected_speedup'))"
tail -3 gpurun_out/r03/ac_err.txt

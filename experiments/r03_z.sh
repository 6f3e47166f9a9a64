#!/bin/bash
# round 3: PMC passes of the final code on the Kuhn cube (roofline.traffic of cube1m_mix) + the UzawaCG cache test with its path knobs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03z
rm -rf $O; mkdir -p $O
python -c "import torch" > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dynamic_collision.py -x -q -m gpu -k "uzawa or collision" 2>&1 | tail -2
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${C} -o p -- python $GRAFT_REPO_ROOT/bench.py --workload cube1m_mix --steps 2 --warmup 3 --no-cpu-baseline --no-roofline > $O/pmc_bench_$C.json 2> $O/pmc_$C.err )
done
python experiments/pmc_to_json_r03.py $O $O/pmc_hbm_cube1m.json cube1m_mix
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE

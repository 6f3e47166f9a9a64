#!/bin/bash
# round-2 A/B on one box: legacy layout / plan+Jacobi / plan+two-level on the cube and the unstructured body
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02a
python -c "import torch" 2>/dev/null
run() { # name, env..., -- args
  name=$1; shift
  env "$@" > /dev/null 2>&1
}
for wl in cube1m_mix blob1m_mix; do
  for cfg in "legacy ADMM_HIP_OC_PLAN=0" "plan_jacobi ADMM_HIP_OC_COARSE=0" "plan_2lvl X=1"; do
    set -- $cfg
    name=$1; kv=$2
    echo "== $wl $name"
    env $kv timeout 600 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02a/${wl}_${name}.json 2> gpurun_out/r02a/${wl}_${name}.err
    tail -c 1500 gpurun_out/r02a/${wl}_${name}.json
    tail -3 gpurun_out/r02a/${wl}_${name}.err
  done
done

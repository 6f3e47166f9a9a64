#!/bin/bash
# per-phase stamps of k_pcg2 for prebuilt libraries: bash experiments/prof_libs.sh workload name1 name2 ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
WL=$1; shift
cp admm-elastic_amd/libadmm_hip.so /tmp/keep.so
for name in "$@"; do
  cp ab/libadmm_hip_$name.so admm-elastic_amd/libadmm_hip.so
  echo "== $name"
  ADMM_HIP_OC_PROF=1 python bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "oc_prof" | grep -E "n=[0-9][0-9]|n=[5-9]" | tail -4
done
cp /tmp/keep.so admm-elastic_amd/libadmm_hip.so

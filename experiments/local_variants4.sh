#!/bin/bash
# Same-box comparison of compile-time variants of the library (built on the box).  Usage: local_variants4.sh "<flags>" ...
export OMP_NUM_THREADS=8 TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python -c "import torch" > /dev/null 2>&1
mkdir -p ab
i=0
for flags in "$@"; do
  ADMM_HIP_EXTRA_FLAGS="$flags" python -c "
import sys; sys.path.insert(0,'.')
from admm_elastic_amd import build; build.build_library(force=True)" > /dev/null 2>&1
  cp admm-elastic_amd/libadmm_hip.so ab/libadmm_hip_v$i.so
  echo "v$i = [$flags]"
  i=$((i+1))
done
names=""; for j in $(seq 0 $((i-1))); do names="$names v$j"; done
bash experiments/ab_libs.sh "${WLS:-blob1m_mix}" $names 2>&1 | grep "^\["

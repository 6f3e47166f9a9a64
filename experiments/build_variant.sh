#!/bin/bash
# build a library VARIANT here (cross-compile, no GPU minutes) into experiments/_build/<name>.so; optional ISA of one kernel
# usage: build_variant.sh <name> "<extra flags>" [asm]
cd /root/repo
name=$1; flags=$2
mkdir -p experiments/_build
if [ "$3" = "asm" ]; then
  mkdir -p /tmp/asm_$name
  ( cd /tmp/asm_$name && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -S --cuda-device-only /root/repo/admm-elastic_amd/csrc/admm_hip.hip -o $name.s )
  echo /tmp/asm_$name/$name.s
else
  ADMM_HIP_EXTRA_FLAGS="$flags" python -c "
import sys; sys.path.insert(0,'.')
from admm_elastic_amd import build; build.build_library(out='experiments/_build/$name.so')"
fi

#!/bin/bash
# builds ab/libadmm_hip_<name>.so from a git revision (default HEAD) for same-box A/B runs (experiments/ab_libs.sh)
# usage: bash experiments/build_ref_lib.sh name [rev] [extra flags]
set -e
name=$1; rev=${2:-HEAD}; shift; shift || true
rm -rf /tmp/ab_wt && git -C /root/repo worktree add /tmp/ab_wt $rev > /dev/null 2>&1
mkdir -p /root/repo/ab
( cd /tmp/ab_wt/admm-elastic_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result "$@" csrc/admm_hip.hip csrc/host_setup.cpp csrc/oc_plan.cpp -o /root/repo/ab/libadmm_hip_$name.so )
git -C /root/repo worktree remove --force /tmp/ab_wt
ls -la /root/repo/ab/libadmm_hip_$name.so

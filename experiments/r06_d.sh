#!/bin/bash
# Round 6, fourth GPU session: the overlapped exchange of k_pcg2 (local part of a product behind the neighbours' hand-off; rows local-first):
# parity, then same-box A/B against the library built from the previous commit (experiments/_build/libadmm_hip_r06base.so), phase tables.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06d; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_soft_modes.py tests/test_oc_plan.py -m gpu -q -x > $O/t_parity.txt 2>&1; tail -5 $O/t_parity.txt
summ() { python - "$1" <<'PY'
import json, sys, os
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(os.path.basename(sys.argv[1]), "no line", e); sys.exit(0)
print("%-44s value %.1f  ms/frame %.3f  stats-frames %.3f  inner timed %.3f stats %s  split %s" % (
    os.path.basename(sys.argv[1]), d["value"], d["ms_per_step"], d["stats_frames_ms_per_step"], d["inner_iters_per_admm_iter"],
    d.get("inner_iters_per_admm_iter_statistics_frames"), {k: round(v, 4) for k, v in d["split_ms_per_admm_iter"].items()}))
PY
}
BASE=$PWD/experiments/_build/libadmm_hip_r06base.so
for rep in 1 2; do
  for WL in blob1m_mix cube1m_nh; do
    timeout 300 python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_new_${WL}_$rep.json; summ $O/bench_new_${WL}_$rep.json
    ADMM_HIP_LIB=$BASE timeout 300 python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_base_${WL}_$rep.json; summ $O/bench_base_${WL}_$rep.json
  done
done | tee $O/ab.txt
ADMM_HIP_OC_DIAG=1 timeout 300 python experiments/oc_prof.py blob1m_mix 2>&1 | grep "oc_prof\|oc_plan" | tail -9 > $O/ocprof_blob_overlap.txt; cat $O/ocprof_blob_overlap.txt
ADMM_HIP_LIB=$BASE timeout 300 python experiments/oc_prof.py blob1m_mix 2>&1 | grep oc_prof | tail -8 > $O/ocprof_blob_base.txt; cat $O/ocprof_blob_base.txt

#!/bin/bash
# Round 6: the look-ahead stop test of k_pcg2: parity, same-box A/B (library of the previous commit; ADMM_HIP_OC_LOOKAHEAD=0), phase table.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06k; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_soft_modes.py -m gpu -q -x > $O/t_parity.txt 2>&1; tail -4 $O/t_parity.txt
summ() { python - "$1" <<'PY'
import json, sys, os
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(os.path.basename(sys.argv[1]), "no line", e); sys.exit(0)
print("%-40s value %.1f  ms/frame %.3f  stats-frames %.3f  inner timed %.3f stats %s  split %s" % (
    os.path.basename(sys.argv[1]), d["value"], d["ms_per_step"], d["stats_frames_ms_per_step"], d["inner_iters_per_admm_iter"],
    d.get("inner_iters_per_admm_iter_statistics_frames"), {k: round(v, 4) for k, v in d["split_ms_per_admm_iter"].items()}))
PY
}
BASE=$PWD/experiments/_build/libadmm_hip_r06base.so
for rep in 1 2; do
  for WL in blob1m_mix; do
  timeout 300 python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_new_${WL}_$rep.json; summ $O/bench_new_${WL}_$rep.json
  ADMM_HIP_OC_LOOKAHEAD=0 timeout 300 python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_off_${WL}_$rep.json; summ $O/bench_off_${WL}_$rep.json
  ADMM_HIP_LIB=$BASE timeout 300 python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_base_${WL}_$rep.json; summ $O/bench_base_${WL}_$rep.json
  done
done | tee $O/ab.txt
timeout 300 python experiments/oc_prof.py blob1m_mix 2>&1 | grep "oc_prof" | tail -8 > $O/ocprof_blob_new.txt; cat $O/ocprof_blob_new.txt


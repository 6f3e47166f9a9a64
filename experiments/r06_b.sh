#!/bin/bash
# Round 6, second GPU session: (1) parity of the solve that sums its own right-hand side (k_pcg2 with Oc2Args::g_inc) -- step-parity, pin,
# slide and recovery tests; (2) same-box A/B of it on the default bench line (ADMM_HIP_FUSE_RHS=0 / 1) + the phase table;
# (3) PCG iterations per solve of frames 0-25 (where do the driver's timed frames 6-25 spend their 8.7 iterations per solve?);
# (4) the cubes' drift at LOOSER tolerances (their 200-frame drift at the blob's settings is 5e-9: three decades of margin).
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_f3_terms.py tests/test_soft_modes.py -m gpu -q -x > $O/t_parity.txt 2>&1; tail -5 $O/t_parity.txt
for F in 1 0 1 0; do
  ADMM_HIP_FUSE_RHS=$F timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/err_f$F.txt | tail -1 > $O/bench_fuse${F}_$RANDOM.json
done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys, os
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(os.path.basename(sys.argv[1]), "no line", e); sys.exit(0)
g = d.get("roofline_global", {})
print("%-28s value %.1f  ms/frame %.3f  stats-frames %.3f  inner timed %.3f stats %s  split %s  local %.1f us" % (
    os.path.basename(sys.argv[1]), d["value"], d["ms_per_step"], d["stats_frames_ms_per_step"], d["inner_iters_per_admm_iter"],
    d.get("inner_iters_per_admm_iter_statistics_frames"), {k: round(v, 4) for k, v in d["split_ms_per_admm_iter"].items()}, d["roofline"]["avg_launch_us"]))
PY
done | tee $O/ab_summary.txt
timeout 300 python experiments/oc_prof.py blob1m_mix 2>&1 | grep oc_prof | tail -8 > $O/ocprof_blob_fused_rhs.txt; cat $O/ocprof_blob_fused_rhs.txt
timeout 300 python experiments/iters_log.py blob1m_mix 27 > $O/iters_blob_frames_0_26.txt 2>&1; cat $O/iters_blob_frames_0_26.txt
for WL in cube1m_nh cube1m_mix; do
  ADMM_DRIFT_WORKLOAD=$WL ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="2e-9;5e-9;1e-8;2e-8;5e-8;1e-7;1e-8:SOFTSET=24;5e-8:SOFTSET=24" timeout 1500 python experiments/r05_drift.py > $O/drift_$WL.txt 2>&1
  cat $O/drift_$WL.txt
done

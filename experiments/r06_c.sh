#!/bin/bash
# Round 6, third GPU session: (1) the reference's own test binary on the GPU; (2) the interleaved right-hand-side gather of k_pcg2, same-box
# A/B + phase table; (3) deeper / other history bases for the first two solves of a frame (ADMM_HIP_RC_DEPTH, ADMM_HIP_RC_ORDER0/1);
# (4) the cubes at their own settings: drift neighbours, bench lines, the 200-frame tests.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06c; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_cpp_api.py -m gpu -q -x -k "reference or lineartet" > $O/t_ref.txt 2>&1; tail -3 $O/t_ref.txt
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
for F in 1 0 1 0; do
  ADMM_HIP_FUSE_RHS=$F timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/err_f$F.txt | tail -1 > $O/bench_fuse${F}_$RANDOM.json
done
for f in $O/bench_fuse*.json; do summ $f; done | tee $O/ab_fuse.txt
timeout 300 python experiments/oc_prof.py blob1m_mix 2>&1 | grep oc_prof | tail -8 > $O/ocprof_blob_fused_rhs_interleaved.txt; cat $O/ocprof_blob_fused_rhs_interleaved.txt
i=0
while read -r ENVS; do
  i=$((i+1))
  env $ENVS timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_rc_$i.json
  echo "$i: $ENVS"; summ $O/bench_rc_$i.json
done <<'LIST' | tee $O/ab_rc.txt
ADMM_X=0
ADMM_HIP_RC_DEPTH=4 ADMM_HIP_RC_ORDER0=p0.1,p0.2,p0.3,p1.1
ADMM_HIP_RC_DEPTH=5 ADMM_HIP_RC_ORDER0=p0.1,p0.2,p0.3,p0.4
ADMM_HIP_RC_DEPTH=4 ADMM_HIP_RC_ORDER0=p0.1,p0.2,p0.3,p1.1 ADMM_HIP_RC_ORDER1=o1,p0.1,p0.2,p0.3
ADMM_HIP_RC_ORDER0=p0.1,p0.2,p1.1,p2.1
ADMM_HIP_RC_ORDER0=p0.1,p1.1,p0.2,p2.1 ADMM_HIP_RC_ORDER1=o1,p0.1,p1.1,p0.2
ADMM_HIP_RC_DEPTH=5 ADMM_HIP_RC_ORDER0=p0.1,p0.2,p0.3,p0.4 ADMM_HIP_RC_ORDER1=o1,p0.1,p0.2,p0.3
LIST
for WL in cube1m_nh; do
  ADMM_DRIFT_WORKLOAD=$WL ADMM_DRIFT_FRAMES=200 ADMM_DRIFT_VARIANTS="2e-7;4e-7" timeout 900 python experiments/r05_drift.py > $O/drift_$WL.txt 2>&1
  cut -c1-220 $O/drift_$WL.txt
done
for WL in cube1m_nh cube1m_mix; do
  timeout 300 python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$WL.json; summ $O/bench_$WL.json
done
timeout 1500 python -m pytest tests/test_bench_parity.py -m gpu -q -s -k "drift_200 or two_frames" > $O/t_drift.txt 2>&1; grep -v "^$" $O/t_drift.txt | cut -c1-400 | tail -12
cp gpurun_out/drift_*_frames.txt $O/ 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "big" > $O/t_big.txt 2>&1; tail -3 $O/t_big.txt

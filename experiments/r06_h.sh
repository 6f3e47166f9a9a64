#!/bin/bash
# Round 6: the new defaults of the recycled start (history for every solve, four pairs): 200-frame drift tests, bench lines, and basis orders of the late solves.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06h; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_parity.py -m gpu -q -s -k "drift_200 or blob52k" > $O/t_drift.txt 2>&1; grep -v "^$" $O/t_drift.txt | cut -c1-330 | tail -8
summ() { python - "$1" <<'PY'
import json, sys, os
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(os.path.basename(sys.argv[1]), "no line", e); sys.exit(0)
print("   value %.1f  ms/frame %.3f  stats-frames %.3f  inner timed %.3f stats %s  unconverged %s" % (
    d["value"], d["ms_per_step"], d["stats_frames_ms_per_step"], d["inner_iters_per_admm_iter"], d.get("inner_iters_per_admm_iter_statistics_frames"), d.get("unconverged_solves_in_timed_region")))
PY
}
i=0
while read -r WL ENVS; do
  i=$((i+1))
  env $ENVS timeout 300 python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_$i.json
  echo "$i: $WL $ENVS"; summ $O/bench_$i.json
done <<'LIST' | tee $O/sweep.txt
blob1m_mix ADMM_X=0
blob1m_mix ADMM_HIP_RC_ORDER=o1,p0.1,p1.1,o2
blob1m_mix ADMM_HIP_RC_ORDER=o1,p0.1,p0.2,p1.1
blob1m_mix ADMM_HIP_RC_DEPTH=4 ADMM_HIP_RC_ORDER=o1,p0.1,p0.2,p0.3
blob1m_mix ADMM_HIP_RC_ORDER=o1,o2,p0.1,p0.2
blob1m_mix ADMM_HIP_RC_ORDER=p0.1,o1,p0.2,o2
blob1m_mix ADMM_HIP_DEFL_START=0
blob1m_mix ADMM_HIP_DEFL_START=6
cube1m_nh ADMM_X=0
cube1m_nh ADMM_HIP_RC_ORDER=o1,p0.1,p1.1,o2
cube100k_uzawa_floor ADMM_X=0
cube100k_uzawa_floor ADMM_HIP_RC_HIST_N=5
LIST
timeout 300 python experiments/iters_log.py blob1m_mix 27 > $O/iters_blob_frames_0_26.txt 2>&1; tail -8 $O/iters_blob_frames_0_26.txt

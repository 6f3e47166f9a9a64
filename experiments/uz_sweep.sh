cd $GRAFT_REPO_ROOT
for cfg in "--pcg-tol 5e-10" "--pcg-tol 1e-8"; do for e in "" "ADMM_HIP_RC_HIST=0"; do
  line=$(env $e python bench.py --workload cube100k_uzawa_floor --steps 10 --warmup 4 --no-cpu-baseline $cfg 2>/dev/null | grep '^{' | tail -1)
  python - "$cfg $e" <<PY
import json, sys
d = json.loads('''$line''')
print("%-40s" % sys.argv[1], "%.0f it/s global %.3f inner %.2f" % (d["value"], d["split_ms_per_admm_iter"]["global"], d["inner_iters_per_admm_iter"]), d["uzawa"])
PY
done; done

"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (gpurun_out/pmc_*) into profiles/<name>.json"""
import csv, collections, json, sys
name, workload = sys.argv[1], sys.argv[2]
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f'gpurun_out/pmc_{C}/p_counter_collection.csv')))
    agg = collections.defaultdict(list)
    for r in rows:
        if r.get('Counter_Name') == C:
            agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    for k, v in agg.items():
        out.setdefault(k, {})[C + "_KB_mean"] = sum(v) / len(v); out[k]["launches"] = len(v)
loc = [v for k, v in out.items() if "k_local_tets" in k]
# per ADMM iteration = one launch of every constitutive-model kernel; FETCH_SIZE x2 (gfx950 calibration)
per_launch = sum(2.0 * v["FETCH_SIZE_KB_mean"] * 1024 + v["WRITE_SIZE_KB_mean"] * 1024 for v in loc)
json.dump({"workload": workload,
           "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace (bench.py --steps 1 --warmup 1 "
                   "--pcg-max-iters 64); KB per dispatch.  gfx950: FETCH_SIZE reports 1/2 of the bytes read (calibrated on k_predict: "
                   "12.6 MB read -> ~6.0 MB reported; k_finish 8.4 -> ~4.0), WRITE_SIZE is 1:1.",
           "local_step_bytes_per_launch": per_launch, "kernels": out}, open(f'profiles/{name}.json', 'w'), indent=1)
print(name, "local step bytes/launch", per_launch / 1e6, "MB")

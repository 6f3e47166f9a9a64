"""Summarise the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of experiments/r03_profiles.sh (separate passes, --kernel-trace only)
into one JSON: bytes per launch of the local step (`roofline.traffic`), the RHS gather and the persistent PCG kernel.
python experiments/pmc_to_json_r03.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/ pmc_bench_*.json> <out.json> <workload>"""
import csv, collections, glob, json, os, sys
root, out_path, workload = sys.argv[1], sys.argv[2], sys.argv[3]
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, "pmc_" + C, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == C:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[C + "_KB_mean"] = sum(v) / len(v); out[k]["launches"] = len(v)
def per_launch(pat):   # FETCH_SIZE x 2 (gfx950 calibration on k_predict / k_finish), WRITE_SIZE 1:1
    return sum(2.0 * v.get("FETCH_SIZE_KB_mean", 0.0) * 1024 + v.get("WRITE_SIZE_KB_mean", 0.0) * 1024 for k, v in out.items() if pat in k)
its = None
for f in glob.glob(os.path.join(root, "pmc_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); its = d["roofline_global"]["iterations_per_solve"] if d.get("roofline_global") else d["inner_iters_per_admm_iter"]
    except Exception:
        pass
res = {"workload": workload,
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only (bench.py --steps 2 --warmup 3 --no-roofline); "
               "KB per dispatch, mean over all dispatches of the run.  gfx950: FETCH_SIZE reports 1/2 of the bytes read (calibrated on k_predict / "
               "k_finish, round 1), WRITE_SIZE is 1:1.  k_pcg2: every 16-byte write-through (sc1) store is counted as its sector.",
       "local_step_bytes_per_launch": per_launch("k_local_tets"), "gather_bytes_per_launch": per_launch("k_gather_rhs"),
       "pcg_bytes_per_launch": per_launch("k_pcg2"), "pcg_iterations_per_launch_in_the_statistics_frames": its,
       "pcg_bytes_per_iteration": per_launch("k_pcg2") / (its + 1.0) if its is not None else None,
       "pcg_bytes_per_iteration_note": "bytes per launch / (iterations per solve + 1): the launch's fixed traffic (slab fill 30 MB, recycled pairs 35 MB, x / b / new pair) is NOT subtracted",
       "kernels": out}
json.dump(res, open(out_path, "w"), indent=1)
print(out_path, "local step MB/launch", res["local_step_bytes_per_launch"] / 1e6, "gather", res["gather_bytes_per_launch"] / 1e6, "pcg", res["pcg_bytes_per_launch"] / 1e6)

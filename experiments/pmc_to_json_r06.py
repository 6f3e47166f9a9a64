"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only) -> one JSON of HBM bytes per launch, RESTRICTED TO THE
ADMM LOOP: `admm_hip_compute_soft_modes` (at initialize) solves 96 systems to a tight tolerance with the same k_pcg2 -- 2.9 GB per launch -- and
the round-5 file averaged them in (round-5 review, weak item 7).  A dispatch belongs to the loop when it follows the first local-step dispatch.
python experiments/pmc_to_json_r06.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/ pmc_bench_*.json> <out.json> <workload>"""
import csv, collections, glob, json, os, sys
root, out_path, workload = sys.argv[1], sys.argv[2], sys.argv[3]
out, setup = {}, {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, "pmc_" + C, "**", "*counter_collection.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == C:
                rows.append((int(r.get("Dispatch_Id", len(rows))), r["Kernel_Name"].split("(")[0], float(r["Counter_Value"])))
    rows.sort()
    first_local = next((i for i, r in enumerate(rows) if "k_local_" in r[1]), len(rows))
    for dst, part in ((setup, rows[:first_local]), (out, rows[first_local:])):
        agg = collections.defaultdict(list)
        for _, k, v in part:
            agg[k].append(v)
        for k, v in agg.items():
            dst.setdefault(k, {})[C + "_KB_mean"] = sum(v) / len(v); dst[k]["launches"] = len(v)
def per_launch(tab, pat):   # FETCH_SIZE x 2 (gfx950 calibration on k_predict / k_finish, MI355X_MICROARCH.md), WRITE_SIZE 1:1
    return sum(2.0 * v.get("FETCH_SIZE_KB_mean", 0.0) * 1024 + v.get("WRITE_SIZE_KB_mean", 0.0) * 1024 for k, v in tab.items() if pat in k)
its = None
for f in glob.glob(os.path.join(root, "pmc_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); its = d["inner_iters_per_admm_iter_statistics_frames"] or d["inner_iters_per_admm_iter"]
    except Exception:
        pass
res = {"workload": workload,
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only (bench.py --steps 2 --warmup 3 --no-roofline); "
               "KB per dispatch, mean over the dispatches of the ADMM LOOP (after the first local-step dispatch; the soft-mode computation's solves are "
               "listed under `before_the_loop`).  gfx950: FETCH_SIZE reports 1/2 of the bytes read (calibrated on k_predict / k_finish, round 1), "
               "WRITE_SIZE is 1:1.  k_pcg2: every 16-byte write-through (sc1) store is counted as its sector.",
       "local_step_bytes_per_launch": per_launch(out, "k_local_tets"), "gather_bytes_per_launch": per_launch(out, "k_gather_rhs"),
       "pcg_bytes_per_launch": per_launch(out, "k_pcg2"), "pcg_iterations_per_launch_in_the_statistics_frames": its,
       "pcg_bytes_per_iteration": per_launch(out, "k_pcg2") / (its + 1.0) if its is not None else None,
       "pcg_bytes_per_iteration_note": "bytes per launch / (iterations per solve + 1): the launch's fixed traffic (slab fill 30 MB, recycled pairs 35 MB, soft modes 17.6 MB, x / b / new pair) is NOT subtracted",
       "kernels": out, "before_the_loop": {"pcg_bytes_per_launch": per_launch(setup, "k_pcg2"), "kernels": setup}}
json.dump(res, open(out_path, "w"), indent=1)
print(out_path, "local step MB/launch", res["local_step_bytes_per_launch"] / 1e6, "gather", res["gather_bytes_per_launch"] / 1e6, "pcg (loop)", res["pcg_bytes_per_launch"] / 1e6,
      "pcg (soft-mode computation)", res["before_the_loop"]["pcg_bytes_per_launch"] / 1e6)

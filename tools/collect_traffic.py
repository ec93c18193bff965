"""HBM traffic per launch of one kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same
command) -> profiles/<out>.json in the format bench.py's pmc_traffic() reads (bench_key, hbm_bytes_per_launch_corrected).
FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md: the counter reports half of a wide coalesced read stream).
usage: collect_traffic.py <fetch dir> <write dir> <kernel-name substring> <bench key> <out.json> [min workgroups]"""
import csv, glob, json, os, sys

fetch_dir, write_dir, sub, key, out = sys.argv[1:6]
min_wg = int(sys.argv[6]) if len(sys.argv) > 6 else 1
vals, n = {}, 0
for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
    f = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
         if sub in r["Kernel_Name"] and r["Counter_Name"] == name
         and int(r["Grid_Size"]) // int(r["Workgroup_Size"]) >= min_wg]
    vals[name] = sum(v) / max(1, len(v))
    n = len(v)
res = {"kernel": sub, "bench_key": key, "launches": n, "FETCH_SIZE_KB_per_launch": vals["FETCH_SIZE"],
       "WRITE_SIZE_KB_per_launch": vals["WRITE_SIZE"],
       "hbm_bytes_per_launch_corrected": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
       "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); WRITE_SIZE uncalibrated"}
json.dump(res, open(out, "w"), indent=1)
print(res)

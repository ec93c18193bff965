"""Summarise tools/pmc_clock.sh: per kernel, GRBM_GUI_ACTIVE / duration = effective GFX clock (GHz).
usage: pmc_clock.py <gpurun_out dir> <tag>   -> prints JSON (commit under profiles/)"""
import csv, glob, json, os, sys, collections

root, tag = sys.argv[1], sys.argv[2]
out = {"_how": "rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -- <program> (tools/pmc_clock.sh); "
               "clock_ghz = GRBM_GUI_ACTIVE / (End - Start of the same dispatch in the kernel trace); GRBM_GUI_ACTIVE is "
               "reported per XCD and summed by rocprofv3 over the 8 XCDs -> divided by 8"}
for prog in ("mfma", "chain", "nng"):
    d = os.path.join(root, f"{tag}_{prog}")
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not cc or not kt:
        continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        ns, name = dur.get(r["Dispatch_Id"], (None, r["Kernel_Name"]))
        key = name.split("(")[0][-70:]
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if ns is not None and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            acc[key]["duration_ns"].append(ns)
    res = {}
    for k, c in acc.items():
        if "duration_ns" not in c or min(c["duration_ns"]) < 200000:
            continue
        n = len(c["duration_ns"])
        gui, ns = sum(c["GRBM_GUI_ACTIVE"]) / n, sum(c["duration_ns"]) / n
        res[k] = {"launches": n, "avg_duration_us": ns / 1e3, "GRBM_GUI_ACTIVE": gui,
                  "SQ_BUSY_CYCLES": sum(c.get("SQ_BUSY_CYCLES", [0])) / n,
                  "clock_ghz_if_summed_over_8_xcd": gui / 8 / ns, "clock_ghz_if_single": gui / ns}
    out[prog] = res
print(json.dumps(out, indent=1))

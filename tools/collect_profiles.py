"""Turn the raw output of tools/measure_round.sh (gpurun_out/final_*) into the committed summaries under profiles/.
usage: python tools/collect_profiles.py <tag>     e.g. r02_final"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
P = os.path.join(ROOT, "profiles")

bench = json.load(open(os.path.join(G, "final_bench.json")))
json.dump(bench, open(os.path.join(P, f"{tag}_bench.json"), "w"), indent=1)
with open(os.path.join(P, f"{tag}_igemm_classes.txt"), "w") as f:
    for line in open(os.path.join(G, "final_bench.err")):
        if line.startswith("[bench]"):
            f.write(line)
stats = sorted(glob.glob(os.path.join(G, "final_prof", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1]
shutil.copy(stats, os.path.join(P, f"{tag}_bench_kernel_stats.csv"))

# HBM traffic of the dominant kernel: FETCH_SIZE / WRITE_SIZE passes (KB per launch), FETCH doubled on gfx950
key = bench["roofline"]["kernel"]
# bench key -> kernel template instantiation (only the direct-to-LDS family is ever dominant)
amode = {"A_ROW": 0, "A_COL": 1, "A_CONV": 2, "A_WGRAD": 3}
bmode = {"B_ROW": 0, "B_COL": 1, "B_DGRAD": 2}
parts = key[len("igemm<"):-1].split(",")
sub = None
if parts[2].startswith("glds"):
    sub = f"igemm_glds_kernel<{amode[parts[0]]}, {bmode[parts[1]]}, {'true' if 'gated' in parts else 'false'}, 8, false>"
elif parts[2].startswith("ires_chain"):
    sub = "ires_chain_kernel<"      # the instantiation is selected by chain_ok() below (JB = 1: J <= 32, JB = 4: J <= 128)


def chain_ok(name):
    """ires_chain_kernel<JB, BF16, T2OUT>: keep the fp32 instantiations of the class bench.py reports (with or
    without the stored intermediate); every other kernel family passes"""
    if "ires_chain_kernel<" not in name:
        return True
    args = name.split("ires_chain_kernel<")[-1].split(">")[0].replace(" ", "").split(",")
    jb = "4" if "x128" in parts[2] else "1"
    return args[0] == jb and args[1] == "false" and args[2] == ("true" if "t2out" in parts else "false")


out = {"kernel": f"{sub} = {key}", "bench_key": key,
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 1 --warmup 1 "
                  "--no-cpu-baseline (tools/measure_round.sh)"}
vals = {}
for name, d in (("FETCH_SIZE", "final_fetch"), ("WRITE_SIZE", "final_write")):
    f = sorted(glob.glob(os.path.join(G, d, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
    # the split-K launches of the same instantiation (scale-1 edge GEMMs, 384 blocks) are their own bench class
    # ("...,splitk,gated"): keep only launches with >= 512 workgroups, i.e. the class bench.py reports
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
         if sub and sub in r["Kernel_Name"] and r["Counter_Name"] == name
         and int(r["Grid_Size"]) // int(r["Workgroup_Size"]) >= 512 and chain_ok(r["Kernel_Name"])]
    vals[name] = sum(v) / max(1, len(v))
    out["launches"] = len(v)
    out[f"{name}_KB_per_launch"] = vals[name]
out["hbm_bytes_per_launch_corrected"] = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
out["note"] = ("FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); "
               "WRITE_SIZE uncalibrated")
json.dump(out, open(os.path.join(P, f"{tag}_traffic_dominant_kernel.json"), "w"), indent=1)
if bench["roofline"].get("traffic") is None:  # first measurement of a new dominant class: the PMC passes of this same set
    bench["roofline"]["traffic"] = out["hbm_bytes_per_launch_corrected"]
    bench["roofline"]["traffic_source"] = f"profiles/{tag}_traffic_dominant_kernel.json"
    json.dump(bench, open(os.path.join(P, f"{tag}_bench.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

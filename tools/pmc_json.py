"""The three counter passes of tools/pmc_kernels.sh -> JSON: per kernel instantiation (its LARGEST launch grid only: the production
shape of the micro-benchmark) the per-launch counter averages and the derived shares used in DESIGN.md.
usage: python tools/pmc_json.py <gpurun_out/tag>   (reads <tag>_1, _2, _3)"""
import collections
import csv
import glob
import json
import sys

base = sys.argv[1]


def kname(n):   # "void (anonymous namespace)::k<...>(args)" -> "k<...>"
    n = n.replace("(anonymous namespace)::", "")
    if n.startswith("void "):
        n = n[5:]
    depth = 0
    for i, ch in enumerate(n):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return n[:i]
    return n


acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for i in (1, 2, 3):
    for f in glob.glob(f"{base}_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = (kname(r["Kernel_Name"]), int(r["Grid_Size"]))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{base}_{i}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            dur[(kname(r["Kernel_Name"]), g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
largest = {}
for (name, grid) in acc:
    if name not in largest or grid > largest[name]:
        largest[name] = grid
out = {"_how": "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass>, three passes (tools/pmc_kernels.sh); per-launch averages over the launches "
               "with the kernel's largest grid.  mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES) (the normalisation of "
               "profiles/r04_pmc_f16_kernels.json); *_share = counter / SQ_WAVE_CYCLES; us = launch duration under the counters."}
for name, grid in sorted(largest.items()):
    if "at::native" in name or "rocclr" in name or "philox" in name:
        continue
    c = {k: sum(v) / len(v) for k, v in acc[(name, grid)].items()}
    d = dur.get((name, grid), [])
    e = {"kernel": name, "grid_size": grid,
         "launches": len(next(iter(acc[(name, grid)].values()))), "us": round(sum(d) / len(d) / 1e3, 1) if d else None, "per_launch": c}
    der = {}
    g = lambda k: c.get(k, 0.0)  # noqa: E731
    if g("SQ_BUSY_CYCLES"):
        der["mfma_pipe_busy"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (32 * g("SQ_BUSY_CYCLES"))
    if g("SQ_INSTS_MFMA"):
        der["non_mfma_valu_per_mfma"] = (g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA")) / g("SQ_INSTS_MFMA")
        der["mfma_per_wave"] = g("SQ_INSTS_MFMA") / max(1.0, g("SQ_WAVES"))
    if g("SQ_WAVE_CYCLES"):
        der["wave_time_waiting_any"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
        der["wave_time_waiting_on_instruction_dependencies"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_LDS_IDX_ACTIVE"):
        der["lds_bank_conflict_share_of_lds_active"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    if g("SQ_VALU_MFMA_BUSY_CYCLES"):
        der["valu_mfma_coexec_share_of_mfma_busy"] = g("SQ_VALU_MFMA_COEXEC_CYCLES") / g("SQ_VALU_MFMA_BUSY_CYCLES")
    e["derived"] = {k: round(v, 4) for k, v in der.items()}
    out[e["kernel"][:90]] = e
print(json.dumps(out, indent=1))

"""Kernel classes of ONE steady training step out of a rocprofv3 --kernel-trace of bench.py: the launches between the last two optimiser
steps (adamw_ema_kernel marks the end of a step), so warm-up passes, graph captures and the NELBO / sampling legs do not dilute it.
usage: python tools/last_step_classes.py <kernel_trace.csv>"""
import collections
import csv
import os
import subprocess
import sys
import tempfile

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
opt = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r["Kernel_Name"]]
# optimiser launches come in runs (score network, flow): a step ends with the last launch of a run
ends = [i for k, i in enumerate(opt) if k + 1 == len(opt) or opt[k + 1] - i > 50]
if len(ends) < 2:
    sys.exit("fewer than two optimiser steps in the trace")
seg = rows[ends[-2] + 1:ends[-1] + 1]
d = collections.defaultdict(lambda: [0, 0])
for r in seg:
    d[r["Kernel_Name"]][0] += 1
    d[r["Kernel_Name"]][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
with tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False) as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs"])
    for n, v in d.items():
        w.writerow([n, v[0], v[1], v[1] / v[0]])
print(f"last step of {os.path.basename(sys.argv[1])}: {len(seg)} launches, {wall:.1f} ms from first launch to last (under the profiler)")
print(subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "kernel_classes.py"), f.name, "1"],
                     capture_output=True, text=True).stdout, end="")
os.unlink(f.name)

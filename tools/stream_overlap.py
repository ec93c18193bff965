"""How much of the kernel time of a rocprofv3 --kernel-trace ran concurrently with a kernel of ANOTHER stream / queue.
usage: python tools/stream_overlap.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
per = collections.Counter()
for r in rows:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (r.get("Stream_Id"), r.get("Queue_Id"))
    per[q] += e - s
    ev.append((s, 1, q))
    ev.append((e, -1, q))
ev.sort()
active = collections.Counter()
last = None
busy1 = busy2 = 0
for t, d, q in ev:
    if last is not None:
        n = sum(1 for v in active.values() if v > 0)
        if n == 1:
            busy1 += t - last
        elif n > 1:
            busy2 += t - last
    active[q] += d
    last = t
print("kernel time per (stream, queue), ms:", {k: round(v / 1e6, 1) for k, v in per.items()})
print(f"wall time with exactly one stream busy {busy1 / 1e6:.1f} ms, with two or more busy {busy2 / 1e6:.1f} ms")

"""Flake hunting, second half: take the log of a whole-suite `pytest -n 3 -v` run in which a test failed, recover the exact
sequence of tests the failing xdist worker had executed up to and including the failure, and replay that sequence
SERIALLY in one process.  If the failure reproduces with nothing else on the GPU it is a process-state leak; if not, the
same order is replayed beside background GPU load (co-tenancy).

usage: flake_replay.py <hunt log> <failing test substring> <out prefix> [--load N]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker_order(log, needle):
    """-> (worker id, [node ids in the order that worker reported them, ending with the failing one])"""
    lines = open(log, errors="replace").read().splitlines()
    rep = re.compile(r"^\[(gw\d+)\] \[\s*\d+%\] (PASSED|FAILED|SKIPPED|ERROR|XFAIL|XPASS) (\S+)")
    per, failing = {}, None
    for ln in lines:
        m = rep.match(ln)
        if not m:
            continue
        w, status, node = m.groups()
        per.setdefault(w, []).append(node)
        if status == "FAILED" and needle in node and failing is None:
            failing = (w, len(per[w]))
    if failing is None:
        return None, []
    w, n = failing
    return w, per[w][:n]


def main():
    log, needle, prefix = sys.argv[1:4]
    load = int(sys.argv[sys.argv.index("--load") + 1]) if "--load" in sys.argv else 0
    w, order = worker_order(log, needle)
    if not order:
        print("no failing", needle, "in", log)
        return 2
    open(prefix + "_order.txt", "w").write("\n".join(order) + "\n")
    print(f"worker {w}: {len(order)} tests up to the failure; order in {prefix}_order.txt", flush=True)
    bg = []
    for i in range(load):   # co-tenancy: full-size training jobs that keep every CU busy
        bg.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "gpu_noise.py"), "1500"], cwd=ROOT,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
    try:
        cmd = [sys.executable, "-m", "pytest", "-p", "no:cacheprovider", "-p", "no:xdist", "-q", "-m", "gpu"] + order
        out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    finally:
        for p in bg:
            p.kill()
    open(prefix + ".log", "w").write(out.stdout[-200000:] + "\n---- stderr ----\n" + out.stderr[-20000:])
    tail = [ln for ln in out.stdout.splitlines() if ln.strip()][-1:]
    print("replay:", tail, flush=True)
    return 0 if needle not in "".join(ln for ln in out.stdout.splitlines() if ln.startswith("FAILED")) else 1


if __name__ == "__main__":
    sys.exit(main())

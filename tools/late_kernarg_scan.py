"""ISA check: kernel arguments must be loaded at the START of a kernel.
hipcc loads a kernel argument with a scalar load from the kernarg segment where it is first needed -- in a kernel with a
long phase before that use, many microseconds after the launch.  Round 4 traced the reproducibility flake of the test suite
to exactly such a kernel (round 3's fused GroupNorm: gamma / beta / output pointers loaded after its statistics phase); on a GPU
shared by more processes than it has hardware queue slots those late loads occasionally returned another launch's values.
This scanner reports, for every kernel in `hipcc -S` output, scalar loads from the kernarg base register pair that come
AFTER the kernel's first s_barrier (or after `--max-insn` instructions), i.e. arguments that are not resident in SGPRs
before the kernel's first synchronisation point.
usage: late_kernarg_scan.py [--max-insn N] file.s [...]      exit code 1 if anything is reported"""
import re
import sys


def scan(path, max_insn=None):
    out, fn, base, n, barrier_seen = [], None, None, 0, False
    for ln, line in enumerate(open(path), 1):
        t = line.split(";")[0].strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            fn, base, n, barrier_seen = m.group(1), None, 0, False
            continue
        if fn is None or not t or t.startswith(".") or t.endswith(":"):
            continue
        if t.startswith("s_endpgm"):
            fn = None
            continue
        n += 1
        if t.startswith("s_barrier"):
            barrier_seen = True
            continue
        m = re.match(r"s_load_dword\w*\s+\S+,\s*(s\[\d+:\d+\]),", t)
        if not m:
            continue
        if base is None:
            base = m.group(1)      # the first scalar load of a kernel reads its kernarg segment
            continue
        if m.group(1) == base and (barrier_seen or (max_insn is not None and n > max_insn)):
            out.append((fn, ln, t, "after the first barrier" if barrier_seen else f"instruction {n}"))
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    max_insn = None
    if args and args[0] == "--max-insn":
        max_insn, args = int(args[1]), args[2:]
    total = 0
    for f in args:
        by = {}
        for fn, ln, t, why in scan(f, max_insn):
            by.setdefault(fn, []).append((ln, t, why))
        for fn, v in by.items():
            total += len(v)
            print(f"{f}: {fn[:110]}: {len(v)} late kernel-argument loads; first: {v[0]}")
    print("total", total)
    sys.exit(1 if total else 0)
